"""Derivation of the rotated-rectangle known answers in tests/golden/opencv_known_answers.json -- TEST INFRASTRUCTURE ONLY.

cv2 cannot be installed in this image, so `oracle/rbox_prompt.py` (the restatement of cv2.fillPoly used by
`Generate Dataset/main_sam_rbox_mask_instance.py:125-129`) is pinned by known answers derived from the PUBLISHED rules of
OpenCV's modules/imgproc/src/drawing.cpp.  Rounds 3 / 4 covered horizontal / vertical / 45-degree edges only, where the line
walk has no rounding choice to make; C4's rotated boxes have theta ~ U[0, pi).  This script derives four cases with arbitrary
slopes WITHOUT importing rbox_prompt.py, and writes every intermediate a reader needs to re-do the derivation with pencil and
paper next to each answer (`derivation` in the JSON):

  * per polygon edge, the cv::LineIterator walk (connectivity 8, leftToRight): start at the LEFT end point, a = |d major|,
    b = |d minor|, err_0 = a - 2b; at every major step: "minor step iff err < 0", then err += -2b (+ 2a after a minor step).
    The JSON lists err before every step and the pixel visited;
  * per scanline y in [y_min, y_max) the two edge crossings in 16.16 fixed point, x(y) = (x_top << 16) + (y - y_top) * dx with
    dx = ((x1 - x0) << 16) / (y1 - y0) in C++ truncating division (CollectPolyEdges), and the span filled between them
    (FillEdgeCollection).  OpenCV has published TWO span rules for non-antialiased polygons: ceil(x_left) .. floor(x_right)
    (2.4 - 4.5.1: `(x1 + XY_ONE - 1) >> XY_SHIFT`, `x2 >> XY_SHIFT`) and, since 4.5.2, round-half-up on both sides (the edge
    x gets `XY_ONE >> 1` added in CollectPolyEdges and `delta = 0`).  The reference pins no OpenCV version, so every case here
    is one where BOTH rules give the same picture (the pixels they disagree on are covered by the boundary lines); the script
    asserts it.

Round 6 adds three `differs_*` cases on which the two rules DISAGREE, with one expected picture per rule (CASES_DIFFER below).

Run:  python oracle/derive_fillpoly_cases.py   (rewrites the `rotated_*` and `differs_*` entries of the JSON in place).
"""
from __future__ import annotations

import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XY_SHIFT = 16


def c_div(a: int, b: int) -> int:
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def line_walk(p1, p2):
    (x1, y1), (x2, y2) = p1, p2
    if x2 < x1:
        x1, y1, x2, y2 = x2, y2, x1, y1
    dx, dy = x2 - x1, y2 - y1
    ystep = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    a, b = (dy, dx) if steep else (dx, dy)
    err, x, y = a - 2 * b, x1, y1
    pixels, errs = [], []
    for _ in range(a + 1):
        pixels.append([x, y])
        errs.append(err)
        minor = err < 0
        err += -2 * b + (2 * a if minor else 0)
        if steep:
            y += ystep
            x += 1 if minor else 0
        else:
            x += 1
            y += ystep if minor else 0
    return pixels, errs


def derive(h: int, w: int, pts, must_agree: bool = True):
    lines, edges, pix = [], [], set()
    for i in range(len(pts)):
        p0, p1 = pts[i - 1], pts[i]
        walk, errs = line_walk(p0, p1)
        lines.append({"from": list(p0), "to": list(p1), "err_before_each_step": errs, "pixels": walk})
        pix |= {(x, y) for x, y in walk if 0 <= x < w and 0 <= y < h}
        if p0[1] == p1[1]:
            continue
        dxf = c_div((p1[0] - p0[0]) << XY_SHIFT, p1[1] - p0[1])
        top, bot = (p0, p1) if p0[1] < p1[1] else (p1, p0)
        edges.append((top[1], bot[1], top[0] << XY_SHIFT, dxf))
    spans, filled = [], {"ceil_floor": set(pix), "round_round": set(pix)}
    if edges:
        for y in range(max(min(e[0] for e in edges), 0), min(max(e[1] for e in edges), h)):
            xs = sorted(e[2] + (y - e[0]) * e[3] for e in edges if e[0] <= y < e[1])
            for k in range(0, len(xs) - 1, 2):
                old = ((xs[k] + (1 << XY_SHIFT) - 1) >> XY_SHIFT, xs[k + 1] >> XY_SHIFT)
                new = ((xs[k] + (1 << (XY_SHIFT - 1))) >> XY_SHIFT, (xs[k + 1] + (1 << (XY_SHIFT - 1))) >> XY_SHIFT)
                spans.append({"y": y, "x_left_16_16": xs[k], "x_right_16_16": xs[k + 1], "span_ceil_floor": list(old),
                              "span_round_round": list(new)})
                for name, (a, b) in (("ceil_floor", old), ("round_round", new)):
                    filled[name] |= {(x, y) for x in range(max(a, 0), min(b, w - 1) + 1)}
    def as_rows(pixels):
        rows = {}
        for y in range(h):
            xs = sorted(x for (x, yy) in pixels if yy == y)
            if xs:
                assert xs == list(range(xs[0], xs[-1] + 1)), "convex: one run per row"
                rows[str(y)] = [xs[0], xs[-1]]
        return rows

    if must_agree:
        assert filled["ceil_floor"] == filled["round_round"], "pick a case on which the two published span rules agree"
        return as_rows(filled["ceil_floor"]), {"boundary_lines": lines, "scanlines": spans}
    assert filled["ceil_floor"] != filled["round_round"], "pick a case on which the two published span rules DISAGREE"
    only_new = sorted(filled["round_round"] - filled["ceil_floor"])
    only_old = sorted(filled["ceil_floor"] - filled["round_round"])
    return (as_rows(filled["ceil_floor"]), as_rows(filled["round_round"]),
            {"boundary_lines": lines, "scanlines": spans, "pixels_only_under_cv2_ge_452": [list(p) for p in only_new],
             "pixels_only_under_cv2_le_451": [list(p) for p in only_old]})


CASES = [
    # a 6.3 x 3.2 rectangle at 18.4 degrees (edge vectors (6, 2) and (-1, 3) are orthogonal)
    ("rotated_rectangle_18_degrees", 8, 10, [[2, 1], [8, 3], [7, 6], [1, 4]]),
    # a square of side sqrt(29) at 21.8 degrees: dx per scanline 2.5 and -0.4 (-26214 in 16.16 after truncation)
    ("rotated_square_22_degrees", 9, 10, [[3, 0], [8, 2], [6, 7], [1, 5]]),
    # a steep 9.5 x 3.2 rectangle at 71.6 degrees (edge vectors (3, 9) and (-3, 1))
    ("rotated_rectangle_72_degrees", 12, 9, [[4, 0], [7, 9], [4, 10], [1, 1]]),
    # what a 8.6 x 0.6 pixel rotated box becomes after .astype(np.int32) (main_sam_rbox_mask_instance.py:128): two of the four
    # vertices collapse, the polygon is a sliver at 14 degrees that is nowhere a full pixel thick
    ("rotated_sub_pixel_sliver", 6, 11, [[1, 1], [9, 3], [9, 4], [1, 1]]),
]


# Round 6 (VERDICT r05 "next round" 8): cases on which the two span rules give DIFFERENT pictures -- the rasteriser now takes the rule as
# a parameter (`fill_rule` = "cv2_le_451" | "cv2_ge_452"), so each rule needs known answers of its own where it matters.  Found by
# a search over small rotated rectangles (this script's own derive(), not the module under test); the extra pixels of the newer
# rule sit on the right-hand edge where x_right's fraction is >= 0.5 and the boundary line has stepped inwards.
CASES_DIFFER = [
    # 3.2 x 4.1 at 23 degrees: two pixels differ
    ("differs_rectangle_23_degrees", 12, 12, [[3, 4], [6, 5], [4, 9], [2, 8]]),
    # 6.7 x 1.4 at 60 degrees: three pixels differ
    ("differs_rectangle_60_degrees", 12, 12, [[3, 1], [6, 7], [5, 8], [1, 2]]),
    # 8.9 x 2.2 at 118 degrees (one horizontal edge after the int32 truncation): four pixels differ
    ("differs_rectangle_118_degrees", 12, 12, [[7, 0], [3, 8], [1, 7], [5, 0]]),
]


def main() -> None:
    path = os.path.join(ROOT, "tests", "golden", "opencv_known_answers.json")
    known = json.load(open(path))
    keep = [c for c in known["fill_poly"] if not c["name"].startswith(("rotated_", "differs_"))]
    for name, h, w, pts in CASES:
        rows, derivation = derive(h, w, [tuple(p) for p in pts])
        keep.append({"name": name, "h": h, "w": w, "pts": pts, "rows": rows, "derivation": derivation})
    for name, h, w, pts in CASES_DIFFER:
        rows_old, rows_new, derivation = derive(h, w, [tuple(p) for p in pts], must_agree=False)
        keep.append({"name": name, "h": h, "w": w, "pts": pts, "rows": rows_old, "rows_cv2_ge_452": rows_new, "derivation": derivation})
    known["fill_poly"] = keep
    known["_doc_differs"] = (
        "Round 6: the three differs_* cases are rotated rectangles on which OpenCV's two published span rules give different pictures: "
        "`rows` is the picture under ceil(x_left) .. floor(x_right) (OpenCV <= 4.5.1, fill_rule cv2_le_451), `rows_cv2_ge_452` the one "
        "with both crossings rounded half up (OpenCV >= 4.5.2).  Every other case has one picture: both rules agree on it.  The "
        "derivation lists both spans per scanline and the pixels that exist under one rule only.")
    known["_doc_rotated"] = (
        "Round 5: the four rotated_* cases have edges at arbitrary slopes (18.4, 21.8, 71.6 degrees and a sub-pixel sliver at 14 "
        "degrees).  Each carries its derivation: the cv::LineIterator walk of every edge (err before each step; a minor-axis step "
        "is taken iff err < 0) and, per scanline, the two 16.16 fixed-point crossings with the span under BOTH span rules OpenCV "
        "has published (ceil..floor up to 4.5.1, round..round since 4.5.2); the cases are chosen so that the rules agree.  "
        "Generated by oracle/derive_fillpoly_cases.py, which does not import oracle/rbox_prompt.py.")
    with open(path, "w") as f:
        json.dump(known, f, indent=1)
        f.write("\n")
    for c in keep[-len(CASES) - len(CASES_DIFFER):]:
        print(c["name"], c["rows"], c.get("rows_cv2_ge_452", ""))


if __name__ == "__main__":
    main()
