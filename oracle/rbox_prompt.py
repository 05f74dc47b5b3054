"""CPU restatement of the rotated-box MASK PROMPT construction -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product path (samrs_amd.transforms.rbox_mask_prompts ->
libsamrs_hip `samrs_rbox_mask_prompt`) never does.

What it restates: `Generate Dataset/main_sam_rbox_mask_instance.py:125-141` -- per rotated box
    canvas = cv2.fillPoly(zeros, [box_pts.astype(int32)], 255)            (:126-129)
    box_mask = -1000 everywhere, +1000 where the canvas is white          (:130-133)
    box_mask = cv2.resize(box_mask, (tw, th), INTER_LINEAR)               (:135-136, th/tw = ResizeLongestSide shape)
    box_mask = cv2.copyMakeBorder(... bottom / right, value=-1000)        (:137-139)
    box_mask = cv2.resize(box_mask, (256, 256), INTER_LINEAR)             (:140)
    -> torch.float32 [256, 256], fed as `mask_input`                      (:141,159-164)

PARITY UNPINNED: the algorithm lives in a third-party dependency that is absent from this image and from
/root/reference (OpenCV; the reference pins no version, requirements name `opencv-python`).  The functions
below restate the PUBLISHED OpenCV 4.x algorithms from modules/imgproc/src/drawing.cpp (`FillPoly` ->
`CollectPolyEdges` + `FillEdgeCollection`, `Line` -> `LineIterator`, 8-connected, shift = 0) and
modules/imgproc/src/resize.cpp (`INTER_LINEAR` for CV_64F: float coefficients, double accumulation,
horizontal pass then vertical pass).  They could not be checked against cv2 here; what IS pinned:
  * the resize against torch F.interpolate(bilinear, align_corners=False) in float64 (same sampling rule;
    cv2 rounds the two tap weights to float32, so agreement is ~1e-4 on the +-1000 range),
  * the polygon fill against a brute-force even-odd / on-boundary rule on random convex quads
    (tests/test_rbox_prompt.py),
  * round 4: both against hand-derived known answers of the published rules (tests/golden/opencv_known_answers.json:
    inclusive rectangle, 45-degree diamond, degenerate slivers, clipping, a 45-degree hypotenuse; INTER_LINEAR up- and
    down-scaling with clamped ends) -- tests/test_rbox_prompt.py::test_fill_poly_and_resize_known_answers.
  * round 5: rotated rectangles at arbitrary angles (18.4 / 21.8 / 71.6 degrees) and a sub-pixel sliver, with the derivation
    of every line pixel and scanline crossing committed next to the answer (oracle/derive_fillpoly_cases.py, independent of
    this module).  OpenCV changed the span rule in 4.5.2 (ceil..floor -> round..round on edge x + 0.5); the pinned cases of round 5
    are ones where both rules give the same picture.  For a general polygon the two can differ by boundary pixels the 8-connected
    edge line does not cover -- the reference pins no version.
  * round 6: `fill_poly(rule=...)` restates BOTH rules, and three more derived cases (`differs_*` in the JSON) are ones where the
    rules give DIFFERENT pictures, with one expected picture per rule.
One known deviation is documented in `line8`: cv2 clips a boundary line to the image before walking it.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def _tdiv(a: int, b: int) -> int:
    """C++ integer division (truncates toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def line8(mask: np.ndarray, p1: Tuple[int, int], p2: Tuple[int, int]) -> None:
    """cv::LineIterator(img, p1, p2, connectivity=8, leftToRight=true) walk, marking pixels inside the image.

    drawing.cpp: the iterator starts at the LEFT end point, steps along the major axis and moves along the minor
    axis whenever the error term is negative: err0 = dx - 2dy; err += -2dy (+ 2dx when a minor step was taken).
    Deviation: cv::Line first calls clipLine() and walks between the CLIPPED end points; here the unclipped line
    is walked and out-of-image pixels are skipped, which can differ by one pixel along an edge that leaves the image.
    """
    h, w = mask.shape
    (x1, y1), (x2, y2) = p1, p2
    if x2 < x1:                                  # leftToRight: start from the left end point
        x1, y1, x2, y2 = x2, y2, x1, y1
    dx, dy = x2 - x1, y2 - y1
    ystep = -1 if dy < 0 else 1
    dy = abs(dy)
    x, y = x1, y1
    if dy > dx:                                  # y is the major axis
        err, plus, minus, count = dy - 2 * dx, 2 * dy, -2 * dx, dy + 1
        for _ in range(count):
            if 0 <= x < w and 0 <= y < h:
                mask[y, x] = True
            neg = err < 0
            err += minus + (plus if neg else 0)
            y += ystep
            x += 1 if neg else 0
    else:
        err, plus, minus, count = dx - 2 * dy, 2 * dx, -2 * dy, dx + 1
        for _ in range(count):
            if 0 <= x < w and 0 <= y < h:
                mask[y, x] = True
            neg = err < 0
            err += minus + (plus if neg else 0)
            x += 1
            y += ystep if neg else 0


FILL_RULES = ("cv2_le_451", "cv2_ge_452")


def fill_poly(h: int, w: int, pts: np.ndarray, rule: str = "cv2_le_451") -> np.ndarray:
    """cv2.fillPoly(zeros(h, w), [pts int32 [V, 2]], 255) != 0, lineType = LINE_8, shift = 0.

    ``rule``: the scanline span rule -- "cv2_le_451": pixels ceil(x_left) .. floor(x_right) (OpenCV 2.4 - 4.5.1, the text below);
    "cv2_ge_452": both crossings rounded half up, (x + XY_ONE / 2) >> XY_SHIFT (OpenCV >= 4.5.2: CollectPolyEdges adds XY_ONE >> 1
    to the edge x and FillEdgeCollection's delta is 0).  The boundary lines are identical under both.

    CollectPolyEdges: every edge is drawn with Line(); non-horizontal edges are collected as
    (y0 < y1, x at y0 in 16.16 fixed point, dx = ((x1 - x0) << 16) / (y1 - y0), C++ truncating division).
    FillEdgeCollection: for every scanline y in [y0, y1) of the active edges, sorted by x and paired,
    pixels ceil(x_left) .. floor(x_right) are set; x advances by dx per scanline.
    """
    if rule not in FILL_RULES:
        raise ValueError(rule)
    pts = np.asarray(pts, dtype=np.int64).reshape(-1, 2)
    mask = np.zeros((h, w), dtype=bool)
    edges = []
    v = len(pts)
    for i in range(v):
        x0, y0 = int(pts[i - 1][0]), int(pts[i - 1][1])
        x1, y1 = int(pts[i][0]), int(pts[i][1])
        line8(mask, (x0, y0), (x1, y1))
        if y0 == y1:
            continue
        dxf = _tdiv((x1 - x0) << XY_SHIFT, y1 - y0)
        if y0 < y1:
            edges.append((y0, y1, x0 << XY_SHIFT, dxf))
        else:
            edges.append((y1, y0, x1 << XY_SHIFT, dxf))
    if not edges:
        return mask
    ymin = min(e[0] for e in edges)
    ymax = max(e[1] for e in edges)
    for y in range(max(ymin, 0), min(ymax, h)):
        xs = sorted(e[2] + (y - e[0]) * e[3] for e in edges if e[0] <= y < e[1])
        for k in range(0, len(xs) - 1, 2):
            if rule == "cv2_le_451":
                xa = (xs[k] + XY_ONE - 1) >> XY_SHIFT
                xb = xs[k + 1] >> XY_SHIFT
            else:
                xa = (xs[k] + (XY_ONE >> 1)) >> XY_SHIFT
                xb = (xs[k + 1] + (XY_ONE >> 1)) >> XY_SHIFT
            if xa < w and xb >= 0:
                xa, xb = max(xa, 0), min(xb, w - 1)
                if xa <= xb:
                    mask[y, xa:xb + 1] = True
    return mask


def linear_taps(n_in: int, n_out: int):
    """cv::resize INTER_LINEAR coordinate table for one axis (resize.cpp, `resize` -> xofs / alpha):
    scale = 1 / (n_out / n_in); f = float((d + 0.5) * scale - 0.5); s = floor(f); f -= s; clamped at both ends.
    Returns (i0, i1, w0, w1) with float32 weights."""
    inv = float(n_out) / float(n_in)
    scale = 1.0 / inv
    d = np.arange(n_out, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= n_in - 1
    f[hi] = 0.0
    s[hi] = n_in - 1
    i1 = np.minimum(s + 1, n_in - 1)
    return s, i1, (np.float32(1.0) - f).astype(np.float32), f


def resize_linear_f64(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img float64 [H, W], (out_w, out_h), interpolation=INTER_LINEAR)."""
    img = np.asarray(img, dtype=np.float64)
    h, w = img.shape
    x0, x1, a0, a1 = linear_taps(w, out_w)
    y0, y1, b0, b1 = linear_taps(h, out_h)
    tmp = img[:, x0] * a0.astype(np.float64) + img[:, x1] * a1.astype(np.float64)          # horizontal pass
    return tmp[y0, :] * b0.astype(np.float64)[:, None] + tmp[y1, :] * b1.astype(np.float64)[:, None]


def preprocess_shape(h: int, w: int, long_side: int) -> Tuple[int, int]:
    """segment_anything/utils/transforms.py:93-102"""
    scale = long_side * 1.0 / max(h, w)
    return int(h * scale + 0.5), int(w * scale + 0.5)


def rbox_mask_prompt(pts: np.ndarray, h: int, w: int, img_size: int = 1024, out: int = 256, rule: str = "cv2_le_451") -> np.ndarray:
    """main_sam_rbox_mask_instance.py:125-141 for one rotated box; returns float32 [out, out].  ``rule``: see fill_poly."""
    inside = fill_poly(h, w, np.asarray(pts).astype(np.int32), rule)
    m = np.where(inside, 1000.0, -1000.0)
    th, tw = preprocess_shape(h, w, img_size)
    m = resize_linear_f64(m, th, tw)
    full = np.full((img_size, img_size), -1000.0)
    full[:th, :tw] = m
    return resize_linear_f64(full, out, out).astype(np.float32)
