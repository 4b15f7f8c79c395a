"""AffRecordsDataset.from_local — the local 2HANDS layout of 2Haff/utils/aff_dataset.py:152-183, 307-338 (h5/ + jsons/): file-range
lookup by name, json files ordered by their first number, one global list of contour masks, original_size from entry "0". The HDF5
reader is h5py (not installed in this image): the tests pass `h5_open=` a reader of .npz files saved under the .h5 names, with the
same `file["data"][name][row]` interface — what is tested is this repo's logic, not an HDF5 parser."""
import json
import os

import numpy as np
import pytest
import torch


class _NpzFile:
    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self._d = {"data": {k: z[k] for k in z.files}}

    def __getitem__(self, k):
        return self._d[k]


def _square(x0, y0, n):
    return [[[x0, y0]], [[x0 + n, y0]], [[x0 + n, y0 + n]], [[x0, y0 + n]]]


def _make_layout(root, counts=(3, 2)):
    os.makedirs(root / "h5")
    os.makedirs(root / "jsons")
    rng = np.random.default_rng(5)
    start = 0
    truth = []
    # written in REVERSE name order and with a 2-digit start to check the numeric (not lexicographic) ordering of the json files
    specs = []
    for n in counts:
        specs.append((start, start + n - 1, n))
        start += n
    specs = [(s + 8, e + 8, n) for s, e, n in specs] if False else specs
    for s, e, n in specs:
        imgs = rng.integers(0, 256, size=(n, 40, 48, 3), dtype=np.uint8)
        narr = np.array([f"Pick up item {s + i}".encode() for i in range(n)])
        tax = np.array([(s + i) % 4 for i in range(n)], dtype=np.int64)
        with open(root / "h5" / f"{s}-{e}_part.h5", "wb") as fh:      # (np.savez would append .npz to a path string)
            np.savez(fh, inpainted=imgs, narration=narr, taxonomy=tax)
        entries = {}
        for i in range(n):
            g = s + i
            entries[str(i)] = {"original_size": [60, 80], "aff_left": [_square(5 + g, 6, 10)], "aff_right": [] if g % 2 else [_square(30, 20 + g, 8)]}
            truth.append((imgs[i], narr[i].decode(), int(tax[i]), g))
        with open(root / "jsons" / f"{s}-{e}_part.json", "w") as fh:
            json.dump(entries, fh)
    return truth


def test_from_local_reads_the_h5_json_layout(tmp_path):
    import haff  # noqa: F401
    from haff import aff_dataset as D, config as hcfg
    truth = _make_layout(tmp_path, counts=(3, 2, 7, 4))                # ranges 0-2, 3-4, 5-11, 12-15: "12-15" sorts before "3-4" as text
    cfg = hcfg.tiny()
    ds = D.AffRecordsDataset.from_local(str(tmp_path), cfg, h5_open=_NpzFile, seed=1)
    assert ds.size == 16 and ds.original_size == (60, 80) and len(ds.records) == 16
    for g, (img, text, tax, _) in enumerate(truth):
        rec = ds.records[g]
        assert np.array_equal(rec["inpainted"], img) and bytes(rec["narration"]).decode() == text and int(rec["taxonomy"]) == tax
        left = D.recreate_mask_from_contours(rec["masks"]["aff_left"], (60, 80))
        assert left[6:17, 5 + g:16 + g].all() and int(left.sum()) == 11 * 11          # the square of THIS global index: json order is numeric
        right = D.recreate_mask_from_contours(rec["masks"]["aff_right"], (60, 80))
        assert (int(right.sum()) == 0) == bool(g % 2)
    with pytest.raises(ValueError):
        ds.records[16]
    seen = set()
    for k in range(40):
        _, image, image_clip, convs, left, right, taxonomy, label, resize, questions, classes, inference = ds[k]   # (+ the flag HybridDataset appends, utils/dataset.py:290-310)
        assert inference is False
        g = int(classes[0].rsplit(" ", 1)[1])
        seen.add(g)
        assert left.shape == (1, 60, 80) and right.shape == (1, 60, 80) and taxonomy == [float(i == g % 4) for i in range(4)]
        assert int(left.sum()) == 121 and classes[0].lower() in questions[0] and "[SEG]" in convs[0]
        assert image.shape == (3, cfg.sam.img_size, cfg.sam.img_size) and resize == (round(40 * cfg.sam.img_size / 48), cfg.sam.img_size)
        assert torch.equal(label["left"] == 0, left[0] != 0)
    assert len(seen) >= 10


def test_from_local_names_the_missing_dependency(tmp_path):
    import haff  # noqa: F401
    from haff import aff_dataset as D, config as hcfg
    _make_layout(tmp_path)
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py is installed here")
    except ImportError:
        pass
    with pytest.raises(ImportError, match="h5py"):
        D.AffRecordsDataset.from_local(str(tmp_path), hcfg.tiny())
