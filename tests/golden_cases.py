"""Inputs of the golden cases whose OUTPUTS were produced by the reference's own code (oracle/make_golden.py), rebuilt from the seeds
and small arrays the fixtures carry — shared by the CPU (oracle) and GPU (HIP path) tests. Data only: nothing of the reference runs here."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def model_forward_case(cfg):
    """tests/golden/lisa_model_forward_tiny.npz (LISA.py:175-430 run by oracle/make_golden.py::lisa_model_forward_golden):
    -> (state dict, training batch of three single-conversation samples, inference batch of one image with two conversations, golden)."""
    from haff import weights as hw
    g = np.load(os.path.join(GOLD, "lisa_model_forward_tiny.npz"))
    seed = int(g["seed"])
    sd = hw.make_state_dict(cfg, seed)
    rng = np.random.default_rng(seed + 7000)
    b, S = 3, cfg.sam.img_size
    H0, W0 = (int(v) for v in g["label_hw"])
    images = torch.from_numpy(rng.standard_normal((b, 3, S, S), dtype=np.float32))
    images_clip = torch.from_numpy(rng.standard_normal((b, 3, cfg.clip.image, cfg.clip.image), dtype=np.float32))
    assert abs(float(images.double().sum()) - float(g["images_checksum"])) < 1e-6
    rng.integers(3, 300, size=(4, 10))                      # (the generator drew the prompt ids here: they are stored)
    ids, labels = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["labels"])
    n_masks = [2, 1, 1]
    masks_l = [(torch.from_numpy(rng.random((n, H0, W0))) > 0.5).float() for n in n_masks]
    masks_r = [(torch.from_numpy(rng.random((n, H0, W0))) > 0.6).float() for n in n_masks]
    tax = torch.from_numpy(g["taxonomies"])
    resize = [tuple(int(v) for v in r) for r in g["resize_list"]]
    label_list = [{"left": torch.zeros(H0, W0), "right": torch.zeros(H0, W0)} for _ in range(b)]
    am = torch.ones_like(ids, dtype=torch.bool)
    train = dict(images=images, images_clip=images_clip, input_ids=ids[1:], labels=labels[1:], attention_masks=am[1:],
                 offset=torch.tensor([0, 1, 2, 3]), masks_list_left=[m[:1] for m in masks_l], masks_list_right=[m[:1] for m in masks_r],
                 taxonomies_list=tax, label_list=label_list, resize_list=resize, inference=False)
    infer = dict(images=images[:1], images_clip=images_clip[:1], input_ids=ids[:2], labels=labels[:2], attention_masks=am[:2],
                 offset=torch.tensor([0, 2]), masks_list_left=masks_l[:1], masks_list_right=masks_r[:1], taxonomies_list=tax[:1],
                 label_list=label_list[:1], resize_list=resize[:1], inference=True)
    return sd, train, infer, g


class StubSpTokenizer:
    """A deterministic stand-in with the PROPERTIES of the sentencepiece Llama tokenizer that the reference's collate_fn arithmetic
    relies on (utils/dataset.py:95-140): every call prepends BOS, "</s>" is ONE token, every other character is one token; pad = 0.
    Used identically by oracle/make_golden.py::collate_goldens (the reference's collate_fn) and by the tests (the product's)."""
    bos_token_id, eos_token_id, pad_token_id, unk_token_id = 1, 2, 0, 0
    model_max_length = 2048

    def __call__(self, text):
        ids = [self.bos_token_id]
        parts = text.split("</s>")
        for i, part in enumerate(parts):
            ids.extend(3 + (ord(ch) % 500) for ch in part)
            if i + 1 < len(parts):
                ids.append(self.eos_token_id)
        return type("Enc", (), {"input_ids": ids})()


def collate_samples(conv_factory):
    """Two 12-tuples in the layout the datasets emit (utils/aff_dataset.py:267-280): sample 0 with two conversations, sample 1 with one;
    conv_factory() -> a fresh conversation object of the template under test (the reference's or the product's)."""
    cases = [([("open the drawer", "It is [SEG]."), ("cut the bread with the knife", "Sure, [SEG].")], (8, 10)),
             ([("pour water", "[SEG].")], (6, 8))]
    out = []
    for n, (qa, (h, w)) in enumerate(cases):
        convs = []
        for q, a in qa:
            c = conv_factory()
            c.messages = []
            c.append_message(c.roles[0], "<image>\n" + "How can I perform the action '%s' in this image? Please output segmentation mask." % q)
            c.append_message(c.roles[1], a)
            convs.append(c.get_prompt())
        g = torch.Generator().manual_seed(100 + n)
        out.append(("path/%d.jpg" % n, torch.randn((3, 16, 16), generator=g), torch.randn((3, 8, 8), generator=g), convs,
                    (torch.rand((len(convs), h, w), generator=g) > 0.5), (torch.rand((len(convs), h, w), generator=g) > 0.5),
                    [0.0, 0.0, 1.0, 0.0] if n == 0 else [1.0, 0.0, 0.0, 0.0], {"left": torch.zeros(h, w), "right": torch.zeros(h, w)},
                    (16, 12), [q for q, _ in qa], [q for q, _ in qa], False))
    return out
