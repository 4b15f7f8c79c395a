"""Rows f3 / f4 on data the REFERENCE ships: four leaf folders of ActAffordance/data_zipped/masks/*.tar.gz copied by
oracle/make_actaffordance_sample.py into tests/golden/actaffordance_sample/ (data files only: annotation.json, inpainting.png
256 x 256, aff_*.png / obj_*.png 855 x 855). The loader is 2Haff/utils/aff_dataset.py:350-544 (AffDatasetVal), the scorer
ActAffordance/scripts/evaluation/calculate_iou.py:26-41, 96-114, 117-337. OpenCV is not installed here: `findContours` behind the
Hausdorff numbers is cvlite's restatement (parity with cv2 itself unpinned), IoU / IoCM / union rules are exact."""
import json
import os
import shutil

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "actaffordance_sample")
EGO = "8f91bc0d-9ce7-4b31-aba7-dd59791917df"
LEAVES = [(EGO, "00000029"), ("P14_05", "0001413"), ("P14_05", "0002976"), ("P14_05", "0003558")]


def _gray(*parts):
    return np.asarray(Image.open(os.path.join(ROOT, *parts)).convert("L"))


def test_sample_is_what_the_reference_ships():
    for sub, leaf in LEAVES:
        assert Image.open(os.path.join(ROOT, sub, leaf, "inpainting.png")).size == (256, 256)
        ann = json.load(open(os.path.join(ROOT, sub, leaf, "annotation.json")))
        assert len(ann["taxonomy"]) == 4 and sum(ann["taxonomy"]) == 1 and ann["narration"]
    assert _gray(EGO, "00000029", "aff_left.png").shape == (855, 855)
    assert set(np.unique(_gray("P14_05", "0003558", "obj_right.png"))) == {0, 255}
    assert _gray("P14_05", "0003558", "aff_right.png").max() > 1          # grey-level maps: the reference thresholds them at > 0


def test_val_dataset_walks_the_benchmark_folders():
    """aff_dataset.py:457-544: a leaf needs inpainting.png + annotation.json + at least one hand; the missing hand is zeros of
    the other's shape; masks stay at 855 x 855 while the image is 256 x 256 (the model's masks are resized to the LABEL's
    shape, LISA.py:306-325)."""
    import haff  # noqa: F401
    from haff import aff_dataset as D, config as hcfg
    cfg = hcfg.tiny()
    ds = D.AffValDataset(ROOT, cfg, seed=0)
    assert len(ds) == 4
    by_text = {n: i for i, n in enumerate(ds.narrations)}
    assert set(by_text) == {"#C C puts rice in a bowl with a serving spoon", "put cereals in bowl", "open refrigerator", "put down milk"}
    i = by_text["put down milk"]                                          # P14_05/0001413: left hand only
    assert ds.taxonomies[i] == [1, 0, 0, 0] and ds.images[i].shape == (256, 256, 3)
    assert ds.affs_left[i].shape == (855, 855) and int((ds.affs_left[i] > 0).sum()) == 3260
    assert ds.affs_right[i].shape == (855, 855) and not ds.affs_right[i].any()
    i = by_text["open refrigerator"]                                      # P14_05/0002976: taxonomy "both", only aff_right.png exists
    assert ds.taxonomies[i] == [0, 0, 0, 1] and not ds.affs_left[i].any() and int((ds.affs_right[i] > 0).sum()) == 1753
    i = by_text["#C C puts rice in a bowl with a serving spoon"]          # both hands, annotation without obj_* keys
    assert int((ds.affs_left[i] > 0).sum()) == 4787 and int((ds.affs_right[i] > 0).sum()) == 3682
    assert np.array_equal(ds.affs_left[i], _gray(EGO, "00000029", "aff_left.png"))
    seen = set()
    for k in range(12):                                                   # __getitem__ draws a random sample (:394)
        _, image, image_clip, convs, left, right, tax, label, resize, questions, classes, inference = ds[k]
        assert image.shape == (3, cfg.sam.img_size, cfg.sam.img_size) and image_clip.shape == (3, cfg.clip.image, cfg.clip.image)
        assert left.shape == (1, 855, 855) and right.shape == (1, 855, 855) and inference is True
        assert resize == (cfg.sam.img_size, cfg.sam.img_size)             # 256 x 256 -> longest side = img_size
        assert label["left"].shape == (855, 855) and set(torch.unique(label["left"]).tolist()) <= {0, 255}
        assert torch.equal(label["right"] == 0, right[0] != 0)            # ignore-label convention: 255 where the mask is empty
        assert classes[0] in by_text and classes[0].lower() in questions[0] and convs[0].endswith("[SEG].</s>")
        assert tax == ds.taxonomies[by_text[classes[0]]]
        seen.add(classes[0])
    assert len(seen) >= 3


def _comparison_from(tmp_path, prefix, transform=None):
    """A prediction tree <tmp>/<video>/<frame>/aff_{side}.png built from the sample's own `prefix`_{side}.png files."""
    for sub, leaf in LEAVES:
        for side in ("left", "right"):
            src = os.path.join(ROOT, sub, leaf, f"{prefix}_{side}.png")
            if not os.path.exists(src):
                continue
            dst = tmp_path / sub / leaf
            dst.mkdir(parents=True, exist_ok=True)
            if transform is None:
                shutil.copy(src, dst / f"aff_{side}.png")
            else:
                transform(np.asarray(Image.open(src).convert("L"))).save(dst / f"aff_{side}.png")
    return str(tmp_path)


def _np_scores(bench, comp):
    inter, union = np.logical_and(bench, comp).sum(), np.logical_or(bench, comp).sum()
    return inter / union, inter / comp.sum()


def test_scorer_on_ground_truth_against_itself():
    import haff  # noqa: F401
    from haff import evaluation as E
    res = E.evaluate_folders(ROOT, ROOT, verbose=False)
    b = res["best"]
    assert b["count"] == 4 and b["failed"] == 0 and b["iou"] == 1.0 and b["iocm"] == 1.0 and b["hd"] == 0.0 and b["directed_hd"] == 0.0


def test_scorer_with_the_object_masks_as_predictions(tmp_path):
    """GT affordance regions against the annotated OBJECT masks of the same frames (a prediction tree built from obj_*.png):
    every number from the reference's formulas recomputed here in plain numpy on the unions of both hands, the averages and the
    --only filters of calculate_iou.py:122-126, and the values pinned as they came out on this sample."""
    import haff  # noqa: F401
    from haff import evaluation as E
    comp = _comparison_from(tmp_path, "obj")
    per_leaf = {}
    for sub, leaf in LEAVES:
        def union(prefix):
            ms = [_gray(sub, leaf, f"{prefix}_{s}.png") > 0 for s in ("left", "right") if os.path.exists(os.path.join(ROOT, sub, leaf, f"{prefix}_{s}.png"))]
            return np.logical_or.reduce(ms)
        per_leaf[(sub, leaf)] = _np_scores(union("aff"), union("obj"))
        got = E.score_frame(os.path.join(ROOT, sub, leaf), os.path.join(comp, sub, leaf), (855, 855))
        assert got[0] == pytest.approx(per_leaf[(sub, leaf)][0], abs=1e-12) and got[1] == pytest.approx(per_leaf[(sub, leaf)][1], abs=1e-12)
        assert 0 < got[2] <= got[3] < np.sqrt(2) * 855                    # directed <= symmetric Hausdorff, inside the image
    assert per_leaf[("P14_05", "0003558")][0] == pytest.approx(0.3480, abs=5e-5)
    assert per_leaf[("P14_05", "0002976")][0] == pytest.approx(0.0101, abs=5e-5)   # the annotated objects: fridge + bottle; the region: a handle
    b = E.evaluate_folders(ROOT, comp, verbose=False)["best"]
    assert b["count"] == 4 and b["failed"] == 0
    assert b["iou"] == pytest.approx(np.mean([v[0] for v in per_leaf.values()]), abs=1e-12) == pytest.approx(0.218683, abs=1e-6)
    assert b["iocm"] == pytest.approx(np.mean([v[1] for v in per_leaf.values()]), abs=1e-12) == pytest.approx(0.223327, abs=1e-6)
    assert b["hd"] == pytest.approx(203.4836, abs=1e-3) and b["directed_hd"] == pytest.approx(194.7555, abs=1e-3)
    epic = E.evaluate_folders(ROOT, comp, only="epic", verbose=False)["best"]
    ego = E.evaluate_folders(ROOT, comp, only="ego", verbose=False)["best"]
    assert epic["count"] == 3 and ego["count"] == 1 and ego["iou"] == pytest.approx(per_leaf[(EGO, "00000029")][0], abs=1e-12)
    # --intersection (:207-215): predictions ANDed with the object mask — a no-op for predictions that ARE the object mask
    inter = E.evaluate_folders(ROOT, comp, take_intersection=True, verbose=False)["best"]
    assert inter["iou"] == b["iou"] and inter["count"] == 4


def test_scorer_resizes_predictions_of_another_size(tmp_path):
    """The live edge case of calculate_iou.py:139,199-201: the benchmark masks are 855 x 855, a model run on inpainting.png writes
    256 x 256 planes — predictions go through the resize to the benchmark size before anything is compared. Nearest-downsampled
    GT as the prediction: bilinear upsampling + `> 0` can only grow a region, so the benchmark region is (nearly) contained in
    the prediction (IoCM == IoU up to pixels the 3.3x down-sampling dropped) and the IoU stays high."""
    import haff  # noqa: F401
    from haff import evaluation as E
    comp = _comparison_from(tmp_path, "aff", lambda a: Image.fromarray(((a > 0) * 255).astype(np.uint8)).resize((256, 256), Image.NEAREST))
    assert Image.open(os.path.join(comp, "P14_05", "0003558", "aff_right.png")).size == (256, 256)
    res = E.evaluate_folders(ROOT, comp, verbose=False)["best"]
    assert res["count"] == 4 and res["failed"] == 0
    assert res["iou"] == pytest.approx(0.8632, abs=2e-3) and res["iocm"] >= res["iou"] and res["iocm"] == pytest.approx(0.8632, abs=2e-3)
    assert res["hd"] < 15.0                                                # contours move by a few pixels, not across the image
    # --cropped (:178-182) takes the size from inpainting.png instead: 256 x 256 predictions against 855 x 855 masks do not
    # line up and every frame is skipped (the reference would raise on the shape mismatch inside np.logical_and)
    cropped = E.evaluate_folders(ROOT, comp, is_cropped=True, verbose=False)["best"]
    assert cropped["count"] == 0
