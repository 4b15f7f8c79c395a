"""The two OpenCV routines the reference's data / scoring code calls, restated in numpy + Python (this image has no cv2):

  draw_contours_filled   cv2.drawContours(mask, [contour], -1, 1, thickness=cv2.FILLED)       (2Haff/utils/aff_dataset.py:340-346)
  find_contours_external cv2.findContours(mask, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)   (ActAffordance/scripts/evaluation/
                                                                                               calculate_iou.py:9-24)

Both are integer algorithms, restated from OpenCV's published sources (opencv/modules/imgproc/src/drawing.cpp: CollectPolyEdges,
FillEdgeCollection, LineIterator; contours.cpp: the Suzuki-Abe border follower icvFetchContour and the external-retrieval rule):

* fill: every polygon edge is first drawn as an 8-connected Bresenham line (LineIterator with leftToRight: the line always runs
  from its smaller-x end, err = dx - 2 dy on the major axis), then the interior is filled scanline by scanline from a
  16.16 fixed-point active-edge list: an edge covers rows y0 <= y < y1, its x advances by dx = (x1 - x0) * 65536 / (y1 - y0)
  (C integer division) per row, crossings are paired in x order and every pair fills x_left >> 16 .. x_right >> 16 inclusive.
* contours: raster scan; an outer border starts at a foreground pixel whose left neighbour is background; the follower looks
  for the first foreground neighbour CLOCKWISE from "left", then walks the border COUNTER-clockwise (codes 0 = +x, 1 = +x-y,
  2 = -y, ... 7 = +x+y, y down) until it returns to the start pixel in the start configuration; CHAIN_APPROX_SIMPLE keeps a
  point only where the outgoing direction changes; RETR_EXTERNAL keeps the outer borders whose surrounding background is
  connected to the image frame; OpenCV hands contours back in REVERSE order of discovery (each is inserted as the first child
  of the frame), so `[0]` — the one the reference scores — is the one whose start pixel comes LAST in raster order.

PARITY: unpinned against cv2 itself (not installable here). tests/test_cvlite_cpu.py holds hand-derived fixtures: shapes whose
cv2 output follows from the definitions above without running it (axis-aligned rectangles, 45-degree diamonds, single pixels
and lines, the convex-polygon interior against the even-odd rule, nested blobs, discovery order) plus structural properties
(the contour of a filled polygon re-fills to the same mask). Polygons that leave the image are clipped per pixel here;
OpenCV clips the SEGMENT first, which can shift the Bresenham phase of such an edge by a pixel.
"""
import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def _cdiv(a, b):
    """C integer division (truncation toward zero) on Python ints."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _line8(img, x0, y0, x1, y1, color):
    """LineIterator(img, pt1, pt2, 8, leftToRight=True) + Line(): every pixel of the 8-connected Bresenham line."""
    h, w = img.shape
    dx, dy = x1 - x0, y1 - y0
    if dx < 0:                      # leftToRight: start from the smaller-x end
        x0, y0, x1, y1 = x1, y1, x0, y0
        dx, dy = -dx, -dy
    sy = 1 if dy >= 0 else -1
    dy = abs(dy)
    steep = dy > dx
    if steep:                       # walk the major axis
        major, minor = dy, dx
    else:
        major, minor = dx, dy
    err = major - 2 * minor
    plus_delta, minus_delta = 2 * major, -2 * minor
    x, y = x0, y0
    for _ in range(major + 1):
        if 0 <= x < w and 0 <= y < h:
            img[y, x] = color
        neg = err < 0
        err += minus_delta + (plus_delta if neg else 0)
        if steep:
            y += sy
            if neg:
                x += 1
        else:
            x += 1
            if neg:
                y += sy


def fill_poly(img, pts, color=1):
    """cv::fillPoly / drawContours(FILLED) of ONE polygon (integer vertices, lineType 8, shift 0) into a 2-D uint8 array."""
    pts = np.asarray(pts, dtype=np.int64).reshape(-1, 2)
    n = len(pts)
    if n == 0:
        return img
    h, w = img.shape
    edges = []
    px, py = int(pts[-1][0]), int(pts[-1][1])
    for i in range(n):
        qx, qy = int(pts[i][0]), int(pts[i][1])
        _line8(img, px, py, qx, qy, color)
        if py != qy:
            dx = _cdiv((qx - px) << XY_SHIFT, qy - py)
            if py < qy:
                edges.append([py, qy, px << XY_SHIFT, dx])
            else:
                edges.append([qy, py, qx << XY_SHIFT, dx])
        px, py = qx, qy
    if len(edges) < 2:
        return img
    y_min = min(e[0] for e in edges)
    y_max = min(max(e[1] for e in edges), h)
    edges.sort(key=lambda e: (e[0], e[2], e[3]))            # CmpEdges: y0, then x, then dx
    active, nxt = [], 0
    for y in range(y_min, y_max):
        active = [e for e in active if e[1] != y]           # an edge leaves when y reaches its lower end
        while nxt < len(edges) and edges[nxt][0] == y:
            active.append(edges[nxt])
            nxt += 1
        active.sort(key=lambda e: e[2])                     # (stable: the bubble sort of the active list)
        for j in range(0, len(active) - 1, 2):
            a, b = active[j], active[j + 1]
            if y >= 0:
                x1, x2 = a[2] >> XY_SHIFT, b[2] >> XY_SHIFT
                if x1 < w and x2 >= 0:
                    img[y, max(x1, 0):min(x2, w - 1) + 1] = color
        for e in active:
            e[2] += e[3]
    return img


def draw_contours_filled(shape, contours, color=1):
    """aff_dataset.py:340-346: a zero uint8 mask of `shape` = (h, w) with every contour filled, one drawContours call each."""
    mask = np.zeros((int(shape[0]), int(shape[1])), dtype=np.uint8)
    for c in contours or []:
        fill_poly(mask, np.asarray(c, dtype=np.int32).reshape(-1, 2), color)
    return mask


_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_DY = (0, -1, -1, -1, 0, 1, 1, 1)


def _outside_background(fg):
    """Background pixels connected to the image frame (4-connectivity: the complement of 8-connected foreground)."""
    h, w = fg.shape
    pad = np.zeros((h + 2, w + 2), dtype=bool)
    pad[1:-1, 1:-1] = fg
    out = np.zeros_like(pad)
    out[0, :] = out[-1, :] = out[:, 0] = out[:, -1] = True
    frontier = out.copy()
    free = ~pad
    while frontier.any():
        grow = np.zeros_like(out)
        grow[1:, :] |= frontier[:-1, :]
        grow[:-1, :] |= frontier[1:, :]
        grow[:, 1:] |= frontier[:, :-1]
        grow[:, :-1] |= frontier[:, 1:]
        frontier = grow & free & ~out
        out |= frontier
    return out          # padded by one pixel on every side


def _follow(fg, x0, y0):
    """icvFetchContour with CHAIN_APPROX_SIMPLE on a padded foreground map: the outer border that starts at (x0, y0)
    (padded coordinates), as a list of (x, y). Also returns every border pixel visited."""
    s = 4
    s_end = 4
    while True:                                  # first foreground neighbour, clockwise from "left"
        s = (s - 1) & 7
        x1, y1 = x0 + _DX[s], y0 + _DY[s]
        if fg[y1, x1] or s == s_end:
            break
    if s == s_end:                               # isolated pixel
        return [(x0, y0)], [(x0, y0)]
    pts, visited = [], []
    x3, y3 = x0, y0
    prev_s = s ^ 4
    while True:
        while True:                              # next border pixel, counter-clockwise
            s = (s + 1) & 7
            x4, y4 = x3 + _DX[s], y3 + _DY[s]
            if fg[y4, x4]:
                break
        visited.append((x3, y3))
        if s != prev_s:
            pts.append((x3, y3))
            prev_s = s
        if x4 == x0 and y4 == y0 and x3 == x1 and y3 == y1:
            break
        x3, y3 = x4, y4
        s = (s + 4) & 7
    return pts, visited


def find_contours_external(mask):
    """cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)[0]: list of int32 arrays [n, 1, 2] (x, y), OpenCV's order."""
    fg0 = np.asarray(mask) != 0
    h, w = fg0.shape
    fg = np.zeros((h + 2, w + 2), dtype=bool)
    fg[1:-1, 1:-1] = fg0
    outside = _outside_background(fg0)
    traced = np.zeros_like(fg)
    found = []
    ys, xs = np.nonzero(fg & ~np.roll(fg, 1, axis=1))          # foreground with a background pixel on its left
    for y, x in zip(ys.tolist(), xs.tolist()):                 # raster order (np.nonzero is row-major)
        if traced[y, x] or not outside[y, x - 1]:
            continue
        pts, visited = _follow(fg, x, y)
        for vx, vy in visited:
            traced[vy, vx] = True
        found.append(np.array([(px - 1, py - 1) for px, py in pts], dtype=np.int32).reshape(-1, 1, 2))
    return found[::-1]
