"""2HANDS sample loader — the caller BEFORE the training path.

Mirrors `2Haff/utils/aff_dataset.py:48-346` (`AffDataset`) for records of the public HF dataset layout
(`_load_from_huggingface`, :117-150): each record carries `narration` (or `text`), `image` (or `inpainted`), `taxonomy`
and `masks = {aff_left: [contour, ...], aff_right: [...], original_size: (h, w)}` with OpenCV-style contours
(lists of (x, y) points). `__getitem__` reproduces :198-280: a RANDOM record per call (the reference ignores `idx`),
masks re-drawn from the contours, the question/answer templates of :29-46, one llava_v1 conversation, CLIP and SAM
preprocessing, and the 11/12-tuple that `collate_fn` (utils/dataset.py:30-169 = train_ds.collate_fn here) consumes.

Differences, both forced by this image (parity unpinned): contours are filled with PIL's polygon rasteriser instead of
`cv2.drawContours(..., FILLED)` (identical interior, boundary pixels may differ by one), and the local h5/json layout
(`_load_from_local`, :152-183) needs h5py, which is not installed — `from_local` raises with that message.
"""
import random

import numpy as np
import torch

from . import preprocess
from . import prompt as hprompt

SHORT_QUESTION_LIST = [
    hprompt.DEFAULT_IMAGE_TOKEN + "\n" + "Can you show me where I have to interact with the objects to perform the following task: {class_name}?",
    hprompt.DEFAULT_IMAGE_TOKEN + "\n" + "Please segment the region to perform the action '{class_name}' in this image.",
    hprompt.DEFAULT_IMAGE_TOKEN + "\n" + "How can I perform the action '{class_name}' in this image? Please respond with segmentation mask.",
    hprompt.DEFAULT_IMAGE_TOKEN + "\n" + "How can I perform the action '{class_name}' in this image? Please output segmentation mask.",
]
ANSWER_LIST = ["It is [SEG].", "Sure, [SEG].", "Sure, it is [SEG].", "Sure, the segmentation result is [SEG].", "[SEG]."]


def recreate_mask_from_contours(contours, shape):
    """aff_dataset.py:340-346: binary uint8 mask of `shape` = (h, w) with every contour filled."""
    from PIL import Image, ImageDraw
    h, w = int(shape[0]), int(shape[1])
    img = Image.new("L", (w, h), 0)
    draw = ImageDraw.Draw(img)
    for contour in contours or []:
        pts = np.asarray(contour, dtype=np.int64).reshape(-1, 2)
        if len(pts) == 1:
            draw.point([tuple(pts[0])], fill=1)
        elif len(pts) == 2:
            draw.line([tuple(p) for p in pts], fill=1)
        elif len(pts) > 2:
            draw.polygon([tuple(p) for p in pts], fill=1, outline=1)
    return np.array(img, dtype=np.uint8)


def _taxonomy_vector(t):
    """Records store the 4-way soft target [left-only, right-only, both, both] (LISA.py:359-361); a bare class index
    becomes its one-hot."""
    if isinstance(t, bytes):
        t = int(t.decode("utf-8"))
    if isinstance(t, (int, np.integer)):
        v = [0.0] * 4
        v[int(t)] = 1.0
        return v
    v = [float(x) for x in np.asarray(t, dtype=np.float64).reshape(-1)]
    assert len(v) == 4, f"taxonomy must have 4 entries, got {len(v)}"
    return v


class AffRecordsDataset(torch.utils.data.Dataset):
    """AffDataset over in-memory records (what `load_dataset(name, split="train")` yields)."""

    def __init__(self, records, cfg, samples_per_epoch=500 * 8 * 2 * 10, inference=False, seed=None):
        self.records = list(records)
        if not self.records:
            raise ValueError("no records")
        self.cfg, self.samples_per_epoch, self.inference = cfg, samples_per_epoch, inference
        self.size = len(self.records)
        self.original_size = None
        for r in self.records:
            m = r.get("masks") or {}
            if "original_size" in m:
                self.original_size = tuple(int(x) for x in m["original_size"])
                break
        self.rng = random.Random(seed)

    @classmethod
    def from_hf(cls, name, cfg, **kw):
        try:
            from datasets import load_dataset
        except ImportError as e:  # aff_dataset.py:108-113
            raise ImportError(f"'{name}' looks like a HuggingFace dataset id but the 'datasets' library is missing") from e
        return cls(load_dataset(name, split="train"), cfg, **kw)

    @classmethod
    def from_local(cls, base_image_dir, cfg, **kw):
        raise ImportError("the local 2HANDS layout (h5/ + jsons/, aff_dataset.py:152-183) needs h5py, which this image "
                          "does not have; export the records (narration, inpainted, taxonomy, masks) and use AffRecordsDataset")

    def __len__(self):
        return self.samples_per_epoch

    def __getitem__(self, idx):
        item = self.records[self.rng.randint(0, self.size - 1)]        # the reference draws a random sample per call
        text = item.get("narration", item.get("text", ""))
        if isinstance(text, bytes):
            text = text.decode("utf-8")
        image = np.array(item["image"] if "image" in item else item["inpainted"])
        if image.ndim == 2:
            image = np.stack([image] * 3, -1)
        image = np.ascontiguousarray(image[..., :3]).astype(np.uint8)
        taxonomy = _taxonomy_vector(item.get("taxonomy", 2))
        m = item.get("masks") or {}
        shape = tuple(int(x) for x in m.get("original_size", self.original_size or image.shape[:2]))
        left = recreate_mask_from_contours(m.get("aff_left", []), shape)
        right = recreate_mask_from_contours(m.get("aff_right", []), shape)
        label = {"left": torch.from_numpy((left == 0).astype(np.int64) * 255),
                 "right": torch.from_numpy((right == 0).astype(np.int64) * 255)}
        cfg = self.cfg
        image_clip = preprocess.clip_preprocess(torch.from_numpy(image.copy()), cfg.clip.image)
        resized = preprocess.resize_longest_side(torch.from_numpy(image.copy()), cfg.sam.img_size)
        resize = tuple(resized.shape[:2])
        image_t = preprocess.sam_preprocess(resized, cfg.sam.img_size)
        question = self.rng.choice(SHORT_QUESTION_LIST).format(class_name=text.lower())
        answer = self.rng.choice(ANSWER_LIST)
        conv = hprompt.conv_llava_v1()
        conv.append_message(conv.roles[0], question)
        conv.append_message(conv.roles[1], answer)
        out = (None, image_t, image_clip, [conv.get_prompt()], torch.from_numpy(left).unsqueeze(0),
               torch.from_numpy(right).unsqueeze(0), taxonomy, label, resize, [question], [text])
        return out + (self.inference,)
