"""2HANDS sample loader — the caller BEFORE the training path.

Mirrors `2Haff/utils/aff_dataset.py:48-346` (`AffDataset`) for records of the public HF dataset layout
(`_load_from_huggingface`, :117-150): each record carries `narration` (or `text`), `image` (or `inpainted`), `taxonomy`
and `masks = {aff_left: [contour, ...], aff_right: [...], original_size: (h, w)}` with OpenCV-style contours
(lists of (x, y) points). `__getitem__` reproduces :198-280: a RANDOM record per call (the reference ignores `idx`),
masks re-drawn from the contours, the question/answer templates of :29-46, one conversation of the default template (llava_v1 unless train_ds.py --conv_type says otherwise), CLIP and SAM
preprocessing, and the 11/12-tuple that `collate_fn` (utils/dataset.py:30-169 = train_ds.collate_fn here) consumes.

`AffValDataset` mirrors `AffDatasetVal` (:350-544): the benchmark folder walk (`<root>/<video>/<frame>/{inpainting.png,
aff_left.png, aff_right.png, annotation.json}`, a missing hand is an all-zero mask, frames without an image / annotation /
any mask are skipped), the same 12-tuple with `inference=True`.

Contours are filled by `cvlite.draw_contours_filled`, a restatement of `cv2.drawContours(..., FILLED)` from OpenCV's published
sources (this image has no cv2; tests/test_cvlite_cpu.py holds hand-derived fixtures — parity with cv2 itself is unpinned).
The local h5/json layout (`_load_from_local`, :152-183, :307-338) is `AffRecordsDataset.from_local`: all of its index / ordering
logic is here; the HDF5 reader itself is h5py, which is not installed in this image (ImportError naming it unless `h5_open=` is given).
"""
import random

import numpy as np
import torch

from . import cvlite
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
    """aff_dataset.py:340-346: binary uint8 mask of `shape` = (h, w) with every contour filled
    (`cv2.drawContours(mask, [np.array(contour, np.int32)], -1, 1, thickness=cv2.FILLED)` per contour)."""
    return cvlite.draw_contours_filled(shape, contours, 1)


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
    def from_local(cls, base_image_dir, cfg, h5_open=None, **kw):
        """The local 2HANDS layout of `_load_from_local` / `extract_index_from_h5` (aff_dataset.py:152-183, 307-338):

            <dir>/h5/<start>-<end>_*.h5     group `data` with datasets `inpainted` [n, H, W, 3] uint8, `narration` [n] bytes,
                                            `taxonomy` [n]; a file holds the samples start..end (inclusive) of the global index
            <dir>/jsons/<start>-<end>_*.json  {"0": {"original_size": [h, w], "aff_left": [contours], "aff_right": [contours]}, ...}

        Everything the reference does with it is here: the json files are read in the order of the first number in their names
        and their entries appended in key order to ONE global list of contour masks (:160-183), `original_size` comes from entry
        "0" of the first file, the number of samples is the sum of the `inpainted` lengths, and sample i is fetched from the
        file whose name range contains i (:307-321) at row i - start. The one thing this image lacks is the HDF5 reader:
        `h5_open(path)` must return a mapping like `h5py.File(path, "r")`; left None it is h5py's, and its absence raises
        ImportError naming the dependency (h5py, any version that reads the files written by 2HANDS/scripts)."""
        import json
        import os
        import re
        if h5_open is None:
            try:
                import h5py
            except ImportError as e:
                raise ImportError("the local 2HANDS layout (<dir>/h5/*.h5 + <dir>/jsons/*.json, aff_dataset.py:152-183) is read "
                                  "through h5py, which is not installed here: `pip install h5py`, or pass h5_open=, or export "
                                  "the records and use AffRecordsDataset / from_hf") from e
            h5_open = lambda path: h5py.File(path, "r")   # noqa: E731
        image_dir, json_dir = os.path.join(base_image_dir, "h5"), os.path.join(base_image_dir, "jsons")

        def first_number(name):
            m = re.search(r"(\d+)", name)
            return int(m.group(1)) if m else float("inf")
        h5_names = [f for f in os.listdir(image_dir) if f.endswith(".h5")]
        ranges = []
        for f in h5_names:
            m = re.match(r"(\d+)-(\d+)_", f)
            if m:
                ranges.append((int(m.group(1)), int(m.group(2)), os.path.join(image_dir, f)))
        size = 0
        for f in h5_names:
            h = h5_open(os.path.join(image_dir, f))
            size += int(h["data"]["inpainted"].shape[0])
            if hasattr(h, "close"):
                h.close()
        masks_left, masks_right, original_size = [], [], None
        for name in sorted(os.listdir(json_dir), key=first_number):
            with open(os.path.join(json_dir, name)) as fh:
                data = json.load(fh)
            if original_size is None:
                original_size = [int(x) for x in data["0"]["original_size"]]
            for key in data:
                masks_left.append(data[key].get("aff_left", []))
                masks_right.append(data[key].get("aff_right", []))
        if len(masks_left) < size:
            raise ValueError(f"{size} samples in h5/ but contours for only {len(masks_left)} in jsons/")

        class _LocalRecords:
            """records[i] for AffRecordsDataset: fetched from the h5 file whose name range holds i (extract_index_from_h5)."""

            def __len__(self):
                return size

            def __getitem__(self, i):
                for lo, hi, path in ranges:
                    if lo <= i <= hi:
                        h = h5_open(path)
                        d = h["data"]
                        rec = {"narration": d["narration"][i - lo], "inpainted": np.asarray(d["inpainted"][i - lo]),
                               "taxonomy": d["taxonomy"][i - lo],
                               "masks": {"original_size": original_size, "aff_left": masks_left[i], "aff_right": masks_right[i]}}
                        if hasattr(h, "close"):
                            h.close()
                        return rec
                raise ValueError(f"Index {i} not found in any file in {image_dir}.")

            def __iter__(self):
                return (self[i] for i in range(size))
        ds = cls.__new__(cls)
        ds.records = _LocalRecords()
        ds.cfg, ds.samples_per_epoch, ds.inference = cfg, kw.get("samples_per_epoch", 500 * 8 * 2 * 10), kw.get("inference", False)
        ds.size, ds.original_size, ds.rng = size, tuple(original_size), random.Random(kw.get("seed"))
        return ds

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
        conv = hprompt.default_conversation()
        conv.append_message(conv.roles[0], question)
        conv.append_message(conv.roles[1], answer)
        out = (None, image_t, image_clip, [conv.get_prompt()], torch.from_numpy(left).unsqueeze(0),
               torch.from_numpy(right).unsqueeze(0), taxonomy, label, resize, [question], [text])
        return out + (self.inference,)


class AffValDataset(torch.utils.data.Dataset):
    """AffDatasetVal (aff_dataset.py:350-544): the ActAffordance-style benchmark folders as validation samples."""

    def __init__(self, base_image_dir, cfg, seed=None):
        self.cfg = cfg
        self.images, self.affs_left, self.affs_right, self.narrations, self.taxonomies = self.load_data_from_nested_folders(base_image_dir)
        self.size = len(self.images)
        self.rng = random.Random(seed)

    def __len__(self):
        return self.size

    @staticmethod
    def load_data_from_nested_folders(root_folder):
        """:457-544. Two directory levels; a leaf needs inpainting.png, annotation.json and at least one of aff_left.png /
        aff_right.png (the missing hand becomes zeros of the other's shape). Masks are read as 8-bit grayscale
        (cv2.imread(..., IMREAD_GRAYSCALE): identical to PIL's "L" for the single-channel PNGs the benchmark ships)."""
        import json
        import os
        from PIL import Image
        images, lefts, rights, narrations, taxonomies = [], [], [], [], []
        for l1 in os.listdir(root_folder):
            p1 = os.path.join(root_folder, l1)
            if not os.path.isdir(p1):
                continue
            for l2 in os.listdir(p1):
                p2 = os.path.join(p1, l2)
                if not os.path.isdir(p2):
                    continue
                files = set(os.listdir(p2))
                has_l, has_r = "aff_left.png" in files, "aff_right.png" in files
                if "inpainting.png" not in files or "annotation.json" not in files or not (has_l or has_r):
                    continue
                images.append(np.array(Image.open(os.path.join(p2, "inpainting.png"))))
                left = np.array(Image.open(os.path.join(p2, "aff_left.png")).convert("L")) if has_l else None
                right = np.array(Image.open(os.path.join(p2, "aff_right.png")).convert("L")) if has_r else None
                lefts.append(left if left is not None else np.zeros_like(right))
                rights.append(right if right is not None else np.zeros_like(left))
                with open(os.path.join(p2, "annotation.json")) as f:
                    ann = json.load(f)
                narrations.append(ann.get("narration", ""))
                taxonomies.append(ann.get("taxonomy", ""))
        return images, lefts, rights, narrations, taxonomies

    def __getitem__(self, idx):
        idx = self.rng.randint(0, self.size - 1)                         # :394 — the reference draws a random sample here too
        text, image, taxonomy = self.narrations[idx], self.images[idx], self.taxonomies[idx]
        if image.ndim == 2:
            image = np.stack([image] * 3, -1)
        image = np.ascontiguousarray(image[..., :3]).astype(np.uint8)
        left, right = self.affs_left[idx], self.affs_right[idx]
        label = {"left": torch.from_numpy((left == 0).astype(np.int64) * 255),
                 "right": torch.from_numpy((right == 0).astype(np.int64) * 255)}
        cfg = self.cfg
        image_clip = preprocess.clip_preprocess(torch.from_numpy(image.copy()), cfg.clip.image)
        resized = preprocess.resize_longest_side(torch.from_numpy(image.copy()), cfg.sam.img_size)
        resize = tuple(resized.shape[:2])
        image_t = preprocess.sam_preprocess(resized, cfg.sam.img_size)
        question = self.rng.choice(SHORT_QUESTION_LIST).format(class_name=text.lower())
        answer = self.rng.choice(ANSWER_LIST)
        conv = hprompt.default_conversation()
        conv.append_message(conv.roles[0], question)
        conv.append_message(conv.roles[1], answer)
        return (None, image_t, image_clip, [conv.get_prompt()], torch.from_numpy(left).unsqueeze(0),
                torch.from_numpy(right).unsqueeze(0), taxonomy, label, resize, [question], [text], True)
