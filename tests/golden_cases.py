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
