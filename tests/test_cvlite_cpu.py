"""cvlite = the two OpenCV routines of rows f3 / f4 restated without cv2 (not installable here). The expected values below are
HAND-DERIVED from the published algorithms (OpenCV drawing.cpp LineIterator / CollectPolyEdges / FillEdgeCollection,
contours.cpp icvFetchContour), not produced by running cv2: shapes for which the definitions leave no freedom."""
import numpy as np


def _cv():
    import haff  # noqa: F401
    from haff import cvlite
    return cvlite


def test_filled_rectangle_diamond_point_line():
    cv = _cv()
    m = cv.draw_contours_filled((8, 10), [[(2, 1), (6, 1), (6, 4), (2, 4)]])
    exp = np.zeros((8, 10), np.uint8)
    exp[1:5, 2:7] = 1                                    # boundary lines included: rows 1..4, columns 2..6
    assert np.array_equal(m, exp)
    d = cv.draw_contours_filled((9, 9), [[(4, 0), (8, 4), (4, 8), (0, 4)]])
    yy, xx = np.mgrid[0:9, 0:9]
    assert np.array_equal(d, (abs(xx - 4) + abs(yy - 4) <= 4).astype(np.uint8))   # 45-degree edges: dx = +-1.0 exactly
    assert cv.draw_contours_filled((5, 5), [[(3, 2)]]).sum() == 1 and cv.draw_contours_filled((5, 5), [[(3, 2)]])[2, 3] == 1
    # 8-connected Bresenham, always walked from the smaller-x end, err = dx - 2 dy: (0,0)-(4,2) and its reverse are ONE pixel set
    for pts in ([(0, 0), (4, 2)], [(4, 2), (0, 0)]):
        ln = cv.draw_contours_filled((4, 6), [pts])
        assert sorted(zip(*np.nonzero(ln)[::-1])) == [(0, 0), (1, 0), (2, 1), (3, 1), (4, 2)]
    # OpenCV contour nesting: [n, 1, 2] arrays and several contours per mask are accepted; out-of-image parts are dropped
    m2 = cv.draw_contours_filled((6, 6), [np.array([[[1, 1]], [[3, 1]], [[3, 3]], [[1, 3]]]), [(4, 4), (9, 4), (9, 9), (4, 9)]])
    exp2 = np.zeros((6, 6), np.uint8)
    exp2[1:4, 1:4] = 1
    exp2[4:, 4:] = 1
    assert np.array_equal(m2, exp2)


def _inside_and_distance(pts, h, w):
    """even-odd point-in-polygon test of every pixel centre (ray casting) and its distance to the polygon's boundary"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    inside = np.zeros((h, w), bool)
    dist = np.full((h, w), np.inf)
    n = len(pts)
    for i in range(n):
        (x0, y0), (x1, y1) = pts[i], pts[(i + 1) % n]
        if y0 != y1:
            cross = ((y0 <= yy) != (y1 <= yy)) & (xx < x0 + (yy - y0) * (x1 - x0) / (y1 - y0))
            inside ^= cross
        dx, dy = x1 - x0, y1 - y0
        L2 = dx * dx + dy * dy
        t = np.clip(((xx - x0) * dx + (yy - y0) * dy) / L2, 0, 1) if L2 else np.zeros_like(xx)
        dist = np.minimum(dist, np.hypot(xx - (x0 + t * dx), yy - (y0 + t * dy)))
    return inside, dist


def test_fill_matches_even_odd_interior_on_random_polygons():
    """Pixels strictly inside (farther than 0.75 px from the boundary) are set, pixels outside and farther than one pixel from it
    are not; the band in between belongs to the boundary lines and the fixed-point rounding of the scanline crossings. Star-shaped
    polygons (sorted angles, random radii): convex and concave vertices."""
    cv = _cv()
    rng = np.random.default_rng(0)
    for _ in range(30):
        k = rng.integers(3, 10)
        ang = np.sort(rng.uniform(0, 2 * np.pi, size=k))
        rad = rng.uniform(6, 18, size=k)
        pts = np.stack([20 + rad * np.cos(ang), 20 + rad * np.sin(ang)], 1).round().astype(int)
        m = cv.draw_contours_filled((41, 41), [pts]).astype(bool)
        inside, dist = _inside_and_distance(pts.tolist(), 41, 41)
        assert m[inside & (dist > 0.75)].all()
        assert not m[~inside & (dist > 1.0)].any()


def test_external_contours_of_simple_shapes():
    cv = _cv()
    m = np.zeros((8, 10), np.uint8)
    m[1:5, 2:7] = 1
    c = cv.find_contours_external(m)
    assert len(c) == 1 and c[0].dtype == np.int32 and c[0].shape == (4, 1, 2)
    assert c[0][:, 0].tolist() == [[2, 1], [2, 4], [6, 4], [6, 1]]        # top-left first, then down the left side
    p = np.zeros((5, 5), np.uint8)
    p[3, 1] = 7
    assert [a[:, 0].tolist() for a in cv.find_contours_external(p)] == [[[1, 3]]]
    ln = np.zeros((6, 9), np.uint8)
    ln[3, 2:7] = 1
    assert cv.find_contours_external(ln)[0][:, 0].tolist() == [[2, 3], [6, 3]]
    yy, xx = np.mgrid[0:9, 0:9]
    d = (abs(xx - 4) + abs(yy - 4) <= 4).astype(np.uint8)
    assert cv.find_contours_external(d)[0][:, 0].tolist() == [[4, 0], [0, 4], [4, 8], [8, 4]]
    # image-border pixels are part of the shape (the image is padded, not zeroed)
    full = np.ones((3, 4), np.uint8)
    assert cv.find_contours_external(full)[0][:, 0].tolist() == [[0, 0], [0, 2], [3, 2], [3, 0]]
    assert cv.find_contours_external(np.zeros((4, 4), np.uint8)) == []


def test_external_retrieval_order_and_nesting():
    cv = _cv()
    m = np.zeros((12, 12), np.uint8)
    m[1:3, 1:3] = 1            # found first in raster order ...
    m[6:11, 2:10] = 1          # ... found second: OpenCV returns it FIRST (contours are prepended)
    m[7:10, 3:9] = 0           # a hole in the second blob ...
    m[8, 5:7] = 1              # ... with an island inside: not external, never returned
    c = cv.find_contours_external(m)
    assert len(c) == 2
    assert c[0][:, 0].tolist() == [[2, 6], [2, 10], [9, 10], [9, 6]] and c[1][:, 0].tolist() == [[1, 1], [1, 2], [2, 2], [2, 1]]


def test_contour_of_a_filled_polygon_refills_to_the_same_mask():
    cv = _cv()
    rng = np.random.default_rng(1)
    for _ in range(10):
        ang = np.sort(rng.uniform(0, 2 * np.pi, size=rng.integers(3, 8)))
        pts = np.stack([30 + 20 * np.cos(ang), 25 + 18 * np.sin(ang)], 1).round().astype(int)
        m = cv.draw_contours_filled((52, 62), [pts])
        c = cv.find_contours_external(m)
        assert len(c) == 1
        again = cv.draw_contours_filled((52, 62), [c[0]])
        assert np.array_equal(again, m)


def test_dataset_masks_and_validation_folders(tmp_path):
    """f4: AffRecordsDataset re-draws the contour lists as drawContours(FILLED) would (cvlite), AffValDataset walks the benchmark
    folders as AffDatasetVal does (aff_dataset.py:457-544: missing hand -> zeros, incomplete leaves skipped) and both hand
    collate_fn the reference's 12-tuple."""
    import json
    import torch
    from PIL import Image
    import haff  # noqa: F401
    from haff import config as hcfg
    from haff.aff_dataset import AffRecordsDataset, AffValDataset, recreate_mask_from_contours
    cfg = hcfg.tiny()
    m = recreate_mask_from_contours([[(2, 1), (6, 1), (6, 4), (2, 4)], [[8, 8]]], (12, 14))
    assert m.dtype == np.uint8 and m.sum() == 20 + 1 and m[8, 8] == 1 and m[1:5, 2:7].all()
    rec = {"narration": "Cut The Bread", "inpainted": np.full((12, 14, 3), 90, np.uint8), "taxonomy": [0, 0, 1, 0],
           "masks": {"aff_left": [[(2, 1), (6, 1), (6, 4), (2, 4)]], "aff_right": [], "original_size": (12, 14)}}
    item = AffRecordsDataset([rec], cfg, seed=0)[0]
    assert len(item) == 12 and item[4].shape == (1, 12, 14) and int(item[4].sum()) == 20 and int(item[5].sum()) == 0
    assert item[7]["left"][0, 0] == 255 and item[7]["left"][2, 3] == 0 and "cut the bread" in item[9][0] and item[11] is False
    root = tmp_path / "bench"
    for vid, frame, hands in (("v1", "f1", ("left", "right")), ("v1", "f2", ("right",)), ("v2", "f1", ())):
        d = root / vid / frame
        d.mkdir(parents=True)
        Image.fromarray(np.full((12, 14, 3), 50, np.uint8)).save(d / "inpainting.png")
        (d / "annotation.json").write_text(json.dumps({"narration": f"{vid} {frame}", "taxonomy": [0, 1, 0, 0]}))
        for h in hands:
            a = np.zeros((12, 14), np.uint8)
            a[3:6, 4:9] = 255
            Image.fromarray(a).save(d / f"aff_{h}.png")
    (root / "v2" / "broken").mkdir()
    val = AffValDataset(str(root), cfg, seed=1)
    assert len(val) == 2                                   # v2/f1 has no mask, v2/broken has nothing
    by_text = dict(zip(val.narrations, zip(val.affs_left, val.affs_right)))
    assert by_text["v1 f2"][0].sum() == 0 and by_text["v1 f2"][1].sum() == 15 * 255
    item = val[0]
    assert len(item) == 12 and item[11] is True and item[1].shape == (3, cfg.sam.img_size, cfg.sam.img_size)
    assert item[4].shape == (1, 12, 14) and isinstance(item[8], tuple) and torch.is_tensor(item[7]["right"])
