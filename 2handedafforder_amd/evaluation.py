"""ActAffordance benchmark scoring — the harness AFTER the hot path.

Mirrors `ActAffordance/scripts/evaluation/calculate_iou.py`: walks `<benchmark>/<video>/<frame>/{aff_left,aff_right}.png`
against `<comparison>[/<threshold>]/<video>/<frame>/{aff_left,aff_right}.png` (what `inference.py:294-334` writes),
takes the union of both hands on each side (a missing hand counts as empty, :243-261), and reports IoU, IoCM
("precision": intersection over the predicted area), Hausdorff and directed Hausdorff averages; with `--map` the
threshold folders are swept and the best-IoCM one is reported together with the mean over thresholds (:321-343).

Exact restatements: `calculate_iou` (:26-41), `calculate_iocm` (:97-114), the union / missing-hand rules, the
averaging and threshold selection, the CLI flags, and `calculate_hausdorff` (:9-24) on the point set the reference uses: the
vertices of the FIRST external contour OpenCV returns (`findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)[0][0]`, first contour
only) — through `cvlite.find_contours_external`, a restatement of OpenCV's border follower (no cv2 in this image: fixtures
hand-derived, tests/test_cvlite_cpu.py; parity with cv2 itself unpinned). Approximation: the reference resizes predictions with
`cv2.resize` (bilinear, no antialias -> here `F.interpolate(bilinear, align_corners=False)`). Visualisation overlays (:43-95) are
not reproduced.
"""
import argparse
import os

import numpy as np


def calculate_iou(mask1, mask2):
    """calculate_iou.py:26-41 — None when either mask is the empty placeholder, 0 when the union is empty."""
    if mask1.size == 0 or mask2.size == 0:
        return None
    inter = np.logical_and(mask1, mask2).sum()
    union = np.logical_or(mask1, mask2).sum()
    return float(inter) / float(union) if union != 0 else 0.0


def calculate_iocm(benchmark_mask, comparison_mask):
    """calculate_iou.py:97-114 — intersection over the comparison (predicted) area."""
    if comparison_mask.size == 0 or benchmark_mask.size == 0:
        return None
    inter = np.logical_and(benchmark_mask, comparison_mask).sum()
    area = comparison_mask.sum()
    return float(inter) / float(area) if area != 0 else 0.0


def calculate_hausdorff(mask1, mask2):
    """calculate_iou.py:9-24: (directed mask2 -> mask1, symmetric) Hausdorff distance between the CHAIN_APPROX_SIMPLE vertices of
    the first external contour of each mask. Empty prediction (mask2) -> the image diagonal for both; empty benchmark -> 0."""
    from scipy.spatial.distance import directed_hausdorff
    from . import cvlite
    shp = mask1.shape
    c1 = cvlite.find_contours_external(np.asarray(mask1).astype(np.uint8))
    c2 = cvlite.find_contours_external(np.asarray(mask2).astype(np.uint8))
    if len(c2) == 0:
        d = float(np.sqrt(shp[0] ** 2 + shp[1] ** 2))
        return d, d
    if len(c1) == 0:
        return 0, 0
    p1 = c1[0].reshape(-1, 2).astype(np.float64)      # np.vstack(contours[0]).squeeze(), a single point kept 2-D
    p2 = c2[0].reshape(-1, 2).astype(np.float64)
    d21 = directed_hausdorff(p2, p1)[0]
    return float(d21), float(max(directed_hausdorff(p1, p2)[0], d21))


def _read_gray(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"))


def _resize_bilinear(img_u8, size_wh):
    """cv2.resize(img, (w, h)) stand-in: bilinear on half-pixel centres, no antialiasing."""
    import torch
    import torch.nn.functional as F
    w, h = size_wh
    if img_u8.shape == (h, w):
        return img_u8
    t = torch.from_numpy(img_u8.astype(np.float32))[None, None]
    return F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)[0, 0].round().clamp(0, 255).numpy().astype(np.uint8)


def score_frame(bench_dir, comp_dir, orig_shape_wh, take_intersection=False):
    """One leaf folder: (iou, iocm, directed_hd, hd) of the two-hand unions, or None when the frame is skipped."""
    empty = np.zeros((0, 0))
    b_left = b_right = c_left = c_right = empty
    p = os.path.join(bench_dir, "aff_left.png")
    if os.path.exists(p):
        b_left = _read_gray(p) > 0
    p = os.path.join(bench_dir, "aff_right.png")
    if os.path.exists(p):
        b_right = _read_gray(p) > 0
    for side in ("left", "right"):
        p = os.path.join(comp_dir, f"aff_{side}.png")
        if not os.path.exists(p):
            continue
        m = _resize_bilinear(_read_gray(p), orig_shape_wh)
        if take_intersection:  # calculate_iou.py:207-215 — restrict the prediction to the annotated object mask
            op = os.path.join(bench_dir, f"obj_{side}.png")
            if not os.path.exists(op):
                return None
            obj = _read_gray(op)
            if obj.shape != m.shape:
                return None
            m = np.bitwise_and(m, obj)
        if side == "left":
            c_left = m > 0
        else:
            c_right = m > 0

    def union(a, b):
        if a.size and b.size:
            return np.logical_or(a, b)
        return a if a.size else b
    bench_union, comp_union = union(b_left, b_right), union(c_left, c_right)
    if bench_union.size and comp_union.size and bench_union.shape != comp_union.shape:
        return None   # (the reference's np.logical_and raises here: e.g. --cropped with full-resolution masks; the frame is skipped)
    iou, iocm = calculate_iou(bench_union, comp_union), calculate_iocm(bench_union, comp_union)
    if iou is None or iocm is None:
        return None
    dhd, hd = calculate_hausdorff(bench_union, comp_union)
    return iou, iocm, dhd, hd


def evaluate_folders(benchmark_folder, comparison_folder, only=None, calc_map=False, is_cropped=False,
                     take_intersection=False, n_examples=float("inf"), verbose=True):
    """calculate_iou.py:117-343 without the overlays. Returns a dict with the per-threshold averages and the pick."""
    subfolders = sorted(os.listdir(benchmark_folder))
    if only == "ego":
        subfolders = [s for s in subfolders if not s.startswith("P")]
    if only == "epic":
        subfolders = [s for s in subfolders if s.startswith("P")]
    thresholds = sorted(os.listdir(comparison_folder)) if calc_map else ["."]
    per_th = []
    for th in thresholds:
        th_dir = os.path.join(comparison_folder, th)
        tot = np.zeros(4)
        count = zero = 0
        for sub in subfolders:
            bsub, csub = os.path.join(benchmark_folder, sub), os.path.join(th_dir, sub)
            if not (os.path.isdir(bsub) and os.path.isdir(csub)):
                continue
            for leaf in sorted(os.listdir(bsub)):
                bleaf, cleaf = os.path.join(bsub, leaf), os.path.join(csub, leaf)
                if not (os.path.isdir(bleaf) and os.path.isdir(cleaf)):
                    continue
                inpaint = os.path.join(bleaf, "inpainting.png")
                if not os.path.exists(inpaint):
                    continue
                shape_wh = (855, 855)                      # calculate_iou.py:139: the uncropped benchmark resolution
                if is_cropped:
                    from PIL import Image
                    shape_wh = Image.open(inpaint).size     # (w, h)
                res = score_frame(bleaf, cleaf, shape_wh, take_intersection)
                if res is None:
                    continue
                tot += np.asarray(res)
                zero += int(res[0] == 0 and res[1] == 0)
                count += 1
                if verbose:
                    print(f"IoU for {sub}/{leaf}: {res[0]:.4f}\nIoCM for {sub}/{leaf}: {res[1]:.4f}")
                if count >= n_examples:
                    break
        avg = tot / max(count, 1)
        per_th.append({"threshold": th, "count": count, "failed": zero, "iou": avg[0], "iocm": avg[1],
                       "directed_hd": avg[2], "hd": avg[3]})
    best = max(per_th, key=lambda r: r["iocm"])
    out = {"per_threshold": per_th, "best": best}
    if calc_map:
        out["mean_average_precision"] = float(np.mean([r["iocm"] for r in per_th]))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="IoU / IoCM / Hausdorff between benchmark and prediction folders")
    ap.add_argument("--benchmark_folder", type=str, default="../../data/cropped")
    ap.add_argument("--comparison_folder", type=str, required=True)
    ap.add_argument("--num-examples", type=int, default=None)
    ap.add_argument("--only", default=None)
    ap.add_argument("--map", default=None, action="store_true")
    ap.add_argument("--cropped", default=None, action="store_true")
    ap.add_argument("--intersection", default=None, action="store_true")
    args = ap.parse_args(argv)
    res = evaluate_folders(args.benchmark_folder, args.comparison_folder, only=args.only, calc_map=bool(args.map),
                           is_cropped=bool(args.cropped), take_intersection=bool(args.intersection),
                           n_examples=args.num_examples or float("inf"))
    b = res["best"]
    if not args.map:
        print(f"Total Failed Predictions: {b['failed']}")
        print(f"Total Averaged IoU: {b['iou']}")
        print(f"Total Averaged IoCM: {b['iocm']}")
        print(f"Total Averaged Hausdorff Distance: {b['hd']}")
        print(f"Total Averaged Directed Hausdorff Distance: {b['directed_hd']}")
    else:
        print(f"mean average precision: {res['mean_average_precision']}")
        print(f"Best performing threshold was {b['threshold']}")
        print(f"IoU: {b['iou']}\nPrecision: {b['iocm']}\nHausdorff-Distance: {b['hd']}\nDirected Hausdorff-Distance: {b['directed_hd']}")
    return res


if __name__ == "__main__":
    main()
