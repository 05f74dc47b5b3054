"""N3: rotated-box mask prompts.

CPU: pins the numpy restatement (oracle/rbox_prompt.py) of cv2.fillPoly / cv2.resize as far as possible without
cv2 (absent in this image -> "parity unpinned" in the oracle header).  GPU: the HIP kernel against that oracle,
bit for bit, and through the predictor as a `mask_input`."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rbox_prompt as rp  # noqa: E402


def random_rbox(rng, h, w, allow_outside=False):
    cx, cy = rng.uniform(0.1 * w, 0.9 * w), rng.uniform(0.1 * h, 0.9 * h)
    bw, bh = rng.uniform(4, 0.6 * w), rng.uniform(4, 0.6 * h)
    th = rng.uniform(0, math.pi)
    c, s = math.cos(th), math.sin(th)
    pts = np.array([[-bw / 2, -bh / 2], [bw / 2, -bh / 2], [bw / 2, bh / 2], [-bw / 2, bh / 2]])
    pts = pts @ np.array([[c, s], [-s, c]]) + [cx, cy]
    if not allow_outside:
        pts[:, 0] = np.clip(pts[:, 0], 0, w - 1)
        pts[:, 1] = np.clip(pts[:, 1], 0, h - 1)
    return pts


def test_axis_aligned_rectangle_is_inclusive():
    m = rp.fill_poly(40, 50, np.array([[5, 7], [30, 7], [30, 20], [5, 20]]))
    ref = np.zeros((40, 50), bool)
    ref[7:21, 5:31] = True                         # cv2.fillPoly includes the boundary
    assert np.array_equal(m, ref)


def _known():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "opencv_known_answers.json")))


def test_fill_poly_and_resize_known_answers():
    """VERDICT r03 "missing" 5: oracle/rbox_prompt.py restates cv2.fillPoly / cv2.resize(INTER_LINEAR) and cv2 cannot be
    installed, so its rules are pinned by hand-derived cases (tests/golden/opencv_known_answers.json): an integer-vertex
    rectangle is filled INCLUDING its boundary (scanline fill ceil(x_left) .. floor(x_right) + the boundary lines), a 45-degree
    diamond is exactly |dx| + |dy| <= r, zero-area polygons leave their 8-connected boundary line (a row, a column, a point),
    the fill is clipped by the image, a 45-degree hypotenuse loses one pixel per row; INTER_LINEAR samples at
    (d + 0.5) * in / out - 0.5 with clamped ends and does not anti-alias when shrinking.
    Round 5: four `rotated_*` cases with edges at the arbitrary angles C4's rotated boxes have (18.4, 21.8,
    71.6 degrees and the sub-pixel sliver a thin box becomes after `.astype(np.int32)`), each with its pencil-and-paper
    derivation in the JSON (LineIterator error terms, 16.16 crossings, the span under both span rules OpenCV has published),
    produced by oracle/derive_fillpoly_cases.py without importing the module under test."""
    k = _known()
    n_differ = 0
    for c in k["fill_poly"]:
        # round 6: both span rules OpenCV has published (fill_rule); `rows_cv2_ge_452` exists where the newer rule's picture differs
        for rule in rp.FILL_RULES:
            rows = c.get("rows_cv2_ge_452", c["rows"]) if rule == "cv2_ge_452" else c["rows"]
            want = np.zeros((c["h"], c["w"]), bool)
            for y, (a, b) in rows.items():
                want[int(y), a:b + 1] = True
            got = rp.fill_poly(c["h"], c["w"], np.asarray(c["pts"]), rule)
            assert np.array_equal(got, want), (c["name"], rule)
        n_differ += "rows_cv2_ge_452" in c
    assert n_differ >= 3
    for c in k["resize_linear"]:
        got = rp.resize_linear_f64(np.asarray(c["src"], dtype=np.float64), c["out_h"], c["out_w"])
        assert np.allclose(got, np.asarray(c["dst"]), atol=1e-4, rtol=0), c["name"]


@pytest.mark.gpu
def test_hip_rbox_prompts_on_the_known_answer_polygons():
    """The HIP rasteriser on the known-answer polygons (scaled onto a 1024^2 canvas, where every edge is still horizontal,
    vertical or at 45 degrees): bit-exact with the oracle, whose fill rule the CPU test above pins."""
    from samrs_amd import synth, transforms
    k = _known()
    polys = [np.asarray(c["pts"], dtype=np.float32) * 100.0 for c in k["fill_poly"] if len(c["pts"]) == 4]
    got = transforms.rbox_mask_prompts(np.stack(polys), (1024, 1024), img_size=1024).cpu().numpy()
    for j, p in enumerate(polys):
        want = rp.rbox_mask_prompt(p.astype(np.int32), 1024, 1024).astype(np.float32)
        assert np.array_equal(got[j], want), k["fill_poly"][j]["name"]
        assert (got[j] < 0).any()             # (a one-pixel sliver can vanish in the 4:1 bilinear reduction to 256 x 256: no > 0 assertion)
    # round 5: every case again at its NATIVE size (a canvas of a few pixels blown up to img_size = 64, prompt 16 x 16), where one
    # wrongly filled pixel moves a sixteenth of the prompt: the rotated_* cases have edges at 18 / 22 / 72 degrees and a
    # sub-pixel sliver, i.e. the rounding choices of the line walk and of the 16.16 scanline crossings all matter
    for c in k["fill_poly"]:
        p = np.asarray(c["pts"], dtype=np.float32)
        if len(p) != 4:
            continue
        per_rule = {}
        for rule in rp.FILL_RULES:              # round 6: the kernel takes the span rule as a parameter (samrs_rbox_mask_prompt_rule)
            got1 = transforms.rbox_mask_prompts(p[None], (c["h"], c["w"]), img_size=64, out_size=16, fill_rule=rule).cpu().numpy()[0]
            want1 = rp.rbox_mask_prompt(p.astype(np.int32), c["h"], c["w"], img_size=64, out=16, rule=rule).astype(np.float32)
            assert np.array_equal(got1, want1), (c["name"], rule)
            per_rule[rule] = got1
        if c["name"].startswith("rotated_"):
            assert (got1 > 0).any() and (got1 < 0).any(), c["name"]
        # the rules must give different prompts exactly on the cases whose known answers differ
        assert np.array_equal(per_rule["cv2_le_451"], per_rule["cv2_ge_452"]) == ("rows_cv2_ge_452" not in c), c["name"]
    # FAIR1M-shaped rotated boxes at full size under both rules: bit-exact with the oracle, and the rules do differ on some of them
    polys, _ = synth.make_rboxes(77, 24)
    n_diff = 0
    for rule in rp.FILL_RULES:
        got = transforms.rbox_mask_prompts(polys, (1024, 1024), fill_rule=rule).cpu().numpy()
        for j in range(len(polys)):
            assert np.array_equal(got[j], rp.rbox_mask_prompt(polys[j].astype(np.int32), 1024, 1024, rule=rule)), (rule, j)
        per_rule[rule] = got
    n_diff = int(sum(not np.array_equal(a, b) for a, b in zip(per_rule["cv2_le_451"], per_rule["cv2_ge_452"])))
    print(f"24 FAIR1M-shaped rboxes: {n_diff} prompts differ between the two fill rules")


def test_line8_closed_form_matches_walk():
    """the kernel uses floor((2*dminor*i + dmajor - 1) / (2*dmajor)) minor steps after i major steps"""
    rng = np.random.default_rng(3)
    for _ in range(300):
        x1, y1, x2, y2 = (int(v) for v in rng.integers(0, 40, 4))
        m = np.zeros((40, 40), bool)
        rp.line8(m, (x1, y1), (x2, y2))
        a, b, c, d = (x1, y1, x2, y2) if x2 >= x1 else (x2, y2, x1, y1)
        dx, dy, ys = c - a, abs(d - b), (-1 if d - b < 0 else 1)
        ref = np.zeros((40, 40), bool)
        if dy > dx:
            for i in range(dy + 1):
                ref[b + ys * i, a + (2 * dx * i + dy - 1) // (2 * dy)] = True
        else:
            for i in range(dx + 1):
                ref[b + (ys * ((2 * dy * i + dx - 1) // (2 * dx)) if dx else 0), a + i] = True
        assert np.array_equal(m, ref), (x1, y1, x2, y2)


def test_fill_contains_interior_and_stays_inside_dilated_polygon():
    rng = np.random.default_rng(5)
    h, w = 120, 160
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(40):
        pts = random_rbox(rng, h, w).astype(np.int32)
        m = rp.fill_poly(h, w, pts)
        # exact signed distances to the 4 edges (integer vertices): interior = same sign for all edges
        sgn = []
        for i in range(4):
            (x0, y0), (x1, y1) = pts[i - 1], pts[i]
            sgn.append((x1 - x0) * (yy - y0) - (y1 - y0) * (xx - x0))
        sgn = np.stack(sgn)
        strictly_inside = np.all(sgn > 0, 0) | np.all(sgn < 0, 0)
        closed = np.all(sgn >= 0, 0) | np.all(sgn <= 0, 0)
        assert not np.any(strictly_inside & ~m), "an interior pixel is not filled"
        # everything filled is within one pixel of the closed polygon (Bresenham boundary pixels)
        dil = closed.copy()
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                dil |= np.roll(np.roll(closed, dy, 0), dx, 1)
        assert not np.any(m & ~dil), "filled pixel far outside the polygon"


@pytest.mark.parametrize("shape,out", [((600, 800), (768, 1024)), ((1024, 1024), (256, 256)), ((37, 53), (101, 64))])
def test_resize_matches_torch_bilinear(shape, out):
    rng = np.random.default_rng(1)
    img = np.where(rng.random(shape) > 0.5, 1000.0, -1000.0)
    got = rp.resize_linear_f64(img, *out)
    ref = F.interpolate(torch.from_numpy(img)[None, None], size=out, mode="bilinear", align_corners=False)[0, 0].numpy()
    # same sampling rule; cv2 computes the source coordinate and the two tap weights in float32: a coordinate of
    # magnitude c carries an error of c * 2^-24, times the 2000-wide value range -> < 1e-2 for sizes up to ~1000
    assert np.abs(got - ref).max() < 2e-2


def test_prompt_shape_range_and_padding():
    pts = np.array([[100.7, 50.2], [300.1, 80.9], [280.5, 200.3], [80.2, 170.8]])
    p = rp.rbox_mask_prompt(pts, 600, 800)
    assert p.shape == (256, 256) and p.dtype == np.float32
    assert p.max() <= 1000.0 and p.min() >= -1000.0 and p.max() > 900
    assert np.all(p[200:, :] == -1000.0)             # 600x800 -> 768x1024: rows >= 192 are border
    th, tw = rp.preprocess_shape(600, 800, 1024)
    assert (th, tw) == (768, 1024)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(1024, 1024), (600, 800), (333, 517)])
def test_hip_rbox_prompts_bit_exact_with_oracle(hw):
    from samrs_amd import transforms
    h, w = hw
    rng = np.random.default_rng(h + w)
    polys = np.stack([random_rbox(rng, h, w, allow_outside=False) for _ in range(6)]
                     + [np.array([[10.0, 10.0], [12.0, 10.0], [12.0, 11.0], [10.0, 11.0]]),          # tiny box
                        np.array([[0.0, 0.0], [w - 1.0, 0.0], [w - 1.0, h - 1.0], [0.0, h - 1.0]])])  # whole image
    got = transforms.rbox_mask_prompts(polys, (h, w)).cpu().numpy()
    for i, p in enumerate(polys):
        ref = rp.rbox_mask_prompt(p, h, w)
        assert np.array_equal(got[i], ref), f"box {i}: max diff {np.abs(got[i] - ref).max()}"


@pytest.mark.gpu
def test_hip_rbox_prompts_partially_outside_fill_part():
    """vertices outside the image: the span fill is restated exactly; the boundary walk is unclipped in both the
    oracle and the kernel (documented deviation from cv2.clipLine), so they still agree bit for bit"""
    from samrs_amd import transforms
    h, w = 400, 500
    rng = np.random.default_rng(9)
    polys = np.stack([random_rbox(rng, h, w, allow_outside=True) for _ in range(8)])
    got = transforms.rbox_mask_prompts(polys, (h, w)).cpu().numpy()
    for i, p in enumerate(polys):
        assert np.array_equal(got[i], rp.rbox_mask_prompt(p, h, w)), f"box {i}"


@pytest.mark.gpu
def test_rbox_mask_prompt_flow_matches_oracle():
    """main_sam_rbox_mask_instance.py:125-164 end to end: rboxes -> mask prompts -> predict_torch(mask_input only).
    HIP prompts + HIP decoder against oracle prompts + oracle decoder on the oracle's embedding."""
    from oracle import sam_oracle as so
    from samrs_amd import SamPredictor, sam_model_registry, synth, transforms
    name = "vit_tiny"
    cfg = synth.CONFIGS[name]
    sam = sam_model_registry[name](precision="f16", max_images=1, max_prompts=8, max_points=1)      # seeded synthetic weights
    sam.to(device="cuda")
    pred = SamPredictor(sam)
    orc = so.OraclePredictor(synth.make_state_dict(cfg, 0), cfg)
    h, w = 600, 800
    img = synth.make_image(1, h, w)
    orc.set_image(img)
    pred.set_image(img)
    pred.model.engine.set_embedding(orc.features.cuda(), pred.slot)
    rng = np.random.default_rng(21)
    polys = np.stack([random_rbox(rng, h, w) for _ in range(5)])
    prompts = transforms.rbox_mask_prompts(polys, (h, w))
    ref_prompts = torch.from_numpy(np.stack([rp.rbox_mask_prompt(p, h, w) for p in polys]))
    assert torch.equal(prompts.cpu(), ref_prompts)
    m, q, low = pred.predict_torch(point_coords=None, point_labels=None, boxes=None, mask_input=prompts[:, None],
                                   multimask_output=False)
    m0, q0, l0 = orc.predict_torch(None, None, None, ref_prompts[:, None], multimask_output=False)
    assert m.shape == m0.shape == (5, 1, h, w)
    l2 = ((low.cpu() - l0).norm() / l0.norm()).item()
    inter = (m.cpu() & m0).flatten(1).sum(1).float()
    union = (m.cpu() | m0).flatten(1).sum(1).float().clamp(min=1)
    print(f"rbox mask-prompt flow: low-res rel L2 {l2:.3e}; IoU min {(inter / union).min():.5f}; quality err {(q.cpu() - q0).abs().max():.2e}")
    assert l2 < 1.5e-3                       # same bar as the decoder-alone test (f16 operands)
    assert (inter / union).min() >= 0.999 or union.max() < 16


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["point", "rbox_mask", "box"])
def test_instance_prompter_modes_match_oracle(mode):
    """driver.InstancePrompter (the three HRSC instance-driver prompt recipes) against the oracle predictor fed with the
    same prompts built on the host, on the oracle's embedding; 5 objects through an engine capped at 2 prompts per call
    (exercises the chunking)."""
    from oracle import sam_oracle as so
    from samrs_amd import SamPredictor, driver, sam_model_registry, synth
    name = "vit_tiny"
    cfg = synth.CONFIGS[name]
    sam = sam_model_registry[name](precision="f16", max_images=1, max_prompts=2, max_points=1)
    sam.to(device="cuda")
    pred = SamPredictor(sam)
    orc = so.OraclePredictor(synth.make_state_dict(cfg, 0), cfg)
    h, w = 600, 800
    img = synth.make_image(2, h, w)
    orc.set_image(img)
    pred.set_image(img)
    pred.model.engine.set_embedding(orc.features.cuda(), pred.slot)
    rng = np.random.default_rng(33)
    rboxes = np.stack([random_rbox(rng, h, w) for _ in range(5)])
    hboxes = np.concatenate([rboxes.min(1), rboxes.max(1)], axis=1).astype(np.float32)       # enclosing hbox (x0, y0, x1, y1)
    points = rboxes.mean(1).astype(np.float32)                                               # object centres
    m, q = driver.InstancePrompter(pred).predict(img, mode, hboxes=hboxes, rboxes=rboxes, points=points, already_set=True)
    if mode == "point":
        m0, q0, _ = orc.predict_torch(torch.from_numpy(points)[:, None, :], torch.ones(5, 1), None, None, multimask_output=False)
    elif mode == "rbox_mask":
        pr = torch.from_numpy(np.stack([rp.rbox_mask_prompt(p, h, w) for p in rboxes]))
        m0, q0, _ = orc.predict_torch(None, None, None, pr[:, None], multimask_output=False)
    else:
        m0, q0, _ = orc.predict_torch(None, None, so.apply_boxes(torch.from_numpy(hboxes), (h, w)), None, multimask_output=False)
    assert m.shape == (5, h, w) and q.shape == (5,)
    inter = (m.cpu() & m0[:, 0]).flatten(1).sum(1).float()
    union = (m.cpu() | m0[:, 0]).flatten(1).sum(1).float().clamp(min=1)
    print(f"instance prompter {mode}: IoU min {(inter / union).min():.5f}, quality err {(q.cpu() - q0[:, 0]).abs().max():.2e}")
    assert (inter / union).min() >= 0.999 or union.max() < 16
    assert (q.cpu() - q0[:, 0]).abs().max() < 2e-3
