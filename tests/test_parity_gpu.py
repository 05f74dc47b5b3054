"""GPU: the drop-in ``SamPredictor`` (HIP path through the C ABI) against the CPU oracle and the
reference-generated golden fixtures, on the same seeded inputs.

Tolerance (floating point path; BASELINE.json north_star: "IoU >= 0.999 vs reference, argmax
class map identical"):
  * f16 MFMA operands (default precision): per-mask IoU >= 0.999 and painted class map mismatch
    <= 0.1 % of pixels on the seeded synthetic weights.  Random-init logits are tiny (std ~0.1) so
    every boundary is ill-conditioned; the remaining flips are pixels whose fp32 logit is within
    the operand-rounding noise of zero, which the margin-filtered IoU (pixels with
    |oracle logit| >= 2 % of the logit std) makes explicit: that one must be >= 0.9999.
  * bf16 MFMA operands: IoU >= 0.99 (8 mantissa bits; measured ~0.997), margin-filtered >= 0.999.
  * decoder alone (oracle embedding installed into the engine): low-res logits rel L2 <= 1.5e-3
    and max abs <= 6e-3 of the logit std for f16 (12e-3 / 5e-2 for bf16).
"""
import json
import os

import numpy as np
import pytest
import torch

from samrs_amd import synth

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import sam_oracle
    return sam_oracle


_CACHE = {}


def get_predictor(name, precision, max_prompts=16, max_images=1):
    key = (name, precision, max_prompts, max_images)
    if key not in _CACHE:
        import samrs_amd
        sam = samrs_amd.sam_model_registry[name](precision=precision, max_prompts=max_prompts, max_images=max_images,
                                                 max_points=4)
        sam.to(device="cuda")
        _CACHE[key] = samrs_amd.SamPredictor(sam)
    return _CACHE[key]


def get_oracle(name):
    key = ("oracle", name)
    if key not in _CACHE:
        so = _oracle()
        cfg = synth.CONFIGS[name]
        _CACHE[key] = so.OraclePredictor(synth.make_state_dict(cfg, 0), cfg)
    return _CACHE[key]


def iou_stats(m, m0, low0=None, margin=0.0):
    m, m0 = m.flatten(1), m0.flatten(1)
    inter = (m & m0).sum(1).double()
    union = (m | m0).sum(1).double().clamp(min=1)
    return (inter / union)


def margin_iou(masks, masks0, logits0, frac=0.02):
    """IoU over pixels whose oracle logit is at least `frac` * std away from the threshold."""
    keep = logits0.abs() >= frac * logits0.std()
    m, m0 = (masks & keep).flatten(1), (masks0 & keep).flatten(1)
    inter = (m & m0).sum(1).double()
    union = (m | m0).sum(1).double().clamp(min=1)
    return inter / union


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["vit_tiny", "vit_tiny80", "vit_tiny1280"])
@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_encoder_blockwise(name, precision):
    """Residual stream after patch embed and after each block vs the oracle (same weights)."""
    so = _oracle()
    cfg = synth.CONFIGS[name]
    pred = get_predictor(name, precision)
    img = synth.make_image(0)
    taps = {}
    orc = get_oracle(name)
    with torch.no_grad():
        so.image_encoder(orc.sd, cfg, so.preprocess(img), taps=taps)
    t = torch.as_tensor(img, device="cuda")[None].contiguous()
    tol = 3e-3 if precision == "f16" else 3e-2
    for nb in range(cfg.depth + 1):
        x = pred.model.engine.debug_encoder_prefix(t, nb).cpu()[0]
        ref = taps["patch" if nb == 0 else f"block{nb - 1}"][0]
        rel = ((x - ref).norm() / ref.norm()).item()
        print(f"{name} {precision} after {nb} blocks: rel L2 {rel:.3e} max abs {(x - ref).abs().max().item():.3e}")
        assert rel < tol, f"residual stream diverges after {nb} blocks"


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_layernorm_tail_of_the_gemms_matches_the_standalone_kernel(precision):
    """Round 5: in the 1x-rate modes the LayerNorm behind proj (norm2) and behind lin2 (the next block's norm1) runs as a TAIL of
    those GEMM launches (option "ln_tail", gemm.hip LnTail): the block that stores the last of the four 256 x 320 tiles of a
    256-row panel normalises the panel -- release / counter / acquire between the four blocks.  embed_dim 1280, a batch of 4
    tiles (the smallest that fills a round of tiles).  Checked: (1) the residual stream after every block against the oracle at
    the stand-alone path's tolerance; (2) against the stand-alone LayerNorm path of the same engine: the same expressions, so
    equal up to hipcc's contraction choices (asserted: relative difference < 1e-5; printed: whether bit-identical);
    (3) the protocol: launch to launch identical, and identical tiles at different batch positions give identical rows -- a
    stale or torn read of another block's tile could not hide from that."""
    so = _oracle()
    import samrs_amd
    name = "vit_tiny1280"
    cfg = synth.CONFIGS[name]
    sam = samrs_amd.sam_model_registry[name](precision=precision, max_prompts=8, max_images=4, max_points=1, options={"split": 15})
    sam.to(device="cuda")
    eng = sam.engine
    assert eng.get_option("ln_tail") in (0, -1)                # off unless asked for (measured slower: profiles/r05_ln_tail.txt)
    imgs = [synth.make_image(0), synth.make_image(1)]
    taps = [{}, {}]
    with torch.no_grad():
        for k in range(2):
            so.image_encoder(get_oracle(name).sd, cfg, so.preprocess(imgs[k]), taps=taps[k])
    t = torch.as_tensor(np.stack([imgs[0], imgs[1], imgs[0], imgs[1]]), device="cuda").contiguous()
    tol = 3e-3 if precision == "f16" else 3e-2
    try:
        for nb in range(1, cfg.depth + 1):
            eng.set_option("ln_tail", 0)
            x0 = eng.debug_encoder_prefix(t, nb).cpu()
            eng.set_option("ln_tail", 1)
            x1 = eng.debug_encoder_prefix(t, nb).cpu()
            x1b = eng.debug_encoder_prefix(t, nb).cpu()
            assert torch.equal(x1, x1b), f"LayerNorm tail: not reproducible from launch to launch after {nb} blocks"
            assert torch.equal(x1[0], x1[2]) and torch.equal(x1[1], x1[3]), f"LayerNorm tail: batch position matters after {nb} blocks"
            d = ((x1 - x0).norm() / x0.norm()).item()
            e1 = max(((x1[k] - taps[k][f"block{nb - 1}"][0]).norm() / taps[k][f"block{nb - 1}"][0].norm()).item() for k in range(2))
            print(f"{name} {precision} after {nb} blocks: LayerNorm tail vs oracle {e1:.3e}; vs stand-alone LayerNorm {d:.3e}"
                  f"{' (bit-identical)' if torch.equal(x0, x1) else ''}")
            assert e1 < tol and d < 1e-5, (nb, e1, d)
    finally:
        eng.set_option("ln_tail", 0)
        eng.close()


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_folded_layernorm_matches_unfolded(precision):
    """embed_dim 1280: the encoder blocks run without LayerNorm launches (the LayerNorm is folded into the qkv / lin1 GEMMs and
    its statistics come out of the proj / lin2 epilogues).  Both paths against the oracle after every block: the folded path's
    error must be in the same class as the stand-alone-LayerNorm path's (it rounds x instead of LN(x) to the operand type), and
    the two paths must agree with each other to the same tolerance."""
    from samrs_amd import engine as _eng
    _lib = _eng.load_library()
    _lib.samrs_debug_has_experiments.restype = __import__("ctypes").c_int
    if not _lib.samrs_debug_has_experiments():
        pytest.skip("the LayerNorm fold is built with make EXPERIMENTS=1 only (measured slower: DESIGN.md 6)")
    so = _oracle()
    import samrs_amd
    from samrs_amd import engine
    name = "vit_tiny1280"
    cfg = synth.CONFIGS[name]
    # the option must be on before the weights are finalized: the folded copies are prepared at load
    sam = samrs_amd.sam_model_registry[name](precision=precision, max_prompts=8, max_images=1, max_points=4,
                                             options={"ln_fold": 1, "split": 15})    # the fold covers the 1x-rate block GEMMs only
    sam.to(device="cuda")
    pred = samrs_amd.SamPredictor(sam)
    eng = sam.engine
    img = synth.make_image(1)
    taps = {}
    with torch.no_grad():
        so.image_encoder(get_oracle(name).sd, cfg, so.preprocess(img), taps=taps)
    t = torch.as_tensor(img, device="cuda")[None].contiguous()
    tol = 3e-3 if precision == "f16" else 3e-2
    try:
        for nb in range(1, cfg.depth + 1):
            ref = taps[f"block{nb - 1}"][0]
            eng.set_option("ln_fold", 0)
            x0 = pred.model.engine.debug_encoder_prefix(t, nb).cpu()[0]
            eng.set_option("ln_fold", 1)
            x1 = pred.model.engine.debug_encoder_prefix(t, nb).cpu()[0]
            x1b = pred.model.engine.debug_encoder_prefix(t, nb).cpu()[0]
            assert torch.equal(x1, x1b), "folded path is not run-to-run reproducible"
            e0 = ((x0 - ref).norm() / ref.norm()).item()
            e1 = ((x1 - ref).norm() / ref.norm()).item()
            d = ((x1 - x0).norm() / ref.norm()).item()
            print(f"{name} {precision} after {nb} blocks: unfolded {e0:.3e}, folded {e1:.3e} vs oracle; folded vs unfolded {d:.3e}")
            assert not torch.equal(x0, x1), "the switch did nothing"
            assert e1 < tol and e1 < 1.5 * e0 + 1e-5 and d < tol
        # the folded engine end to end: embedding and box masks against the oracle
        eng.set_option("ln_fold", 1)
        orc = get_oracle(name)
        pred.set_image(img)
        orc.set_image(img)
        rel = ((pred.get_image_embedding().cpu() - orc.features).norm() / orc.features.norm()).item()
        assert rel < tol, rel
        boxes = torch.from_numpy(synth.C1_BOXES)
        tb = pred.transform.apply_boxes_torch(boxes.cuda(), img.shape[:2])
        masks, _, _ = pred.predict_torch(None, None, tb, None, multimask_output=False)
        m0, _, _ = orc.predict_torch(None, None, tb.cpu(), None, multimask_output=False)
        iou = iou_stats(masks.cpu(), m0)
        print(f"{name} {precision} folded: embedding rel L2 {rel:.3e}, box-mask IoU min {iou.min().item():.5f}")
        assert iou.min().item() >= (0.999 if precision == "f16" else 0.99)
    finally:
        eng.set_option("ln_fold", 0)
        eng.close()


@pytest.mark.parametrize("name", ["vit_tiny", "vit_tiny80", "vit_tiny1280"])
@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_embedding_and_masks_vs_oracle(name, precision):
    so = _oracle()
    pred = get_predictor(name, precision)
    orc = get_oracle(name)
    img = synth.make_image(0)
    pred.set_image(img)
    orc.set_image(img)
    f, f0 = pred.get_image_embedding().cpu(), orc.features
    rel = ((f - f0).norm() / f0.norm()).item()
    print(f"{name} {precision}: embedding rel L2 {rel:.3e}")
    assert rel < (3e-3 if precision == "f16" else 3e-2)
    boxes, labels = synth.make_boxes(0, 12)
    boxes = torch.from_numpy(np.concatenate([synth.C1_BOXES, boxes]))
    tb = pred.transform.apply_boxes_torch(boxes.cuda(), img.shape[:2])
    assert torch.equal(tb.cpu(), so.apply_boxes(boxes, img.shape[:2]))
    masks, iou, low = pred.predict_torch(None, None, tb, None, multimask_output=False)
    m0, i0, l0 = orc.predict_torch(None, None, tb.cpu(), None, multimask_output=False)
    assert masks.dtype == torch.bool and masks.shape == m0.shape and low.shape == l0.shape
    ious = iou_stats(masks.cpu(), m0)
    l0_up = so.postprocess_masks(l0, orc.input_size, orc.original_size)
    mi = margin_iou(masks.cpu(), m0, l0_up)
    print(f"{name} {precision}: IoU min {ious.min():.5f} mean {ious.mean():.5f}; margin-IoU min {mi.min():.6f}; "
          f"low-res max abs {(low.cpu() - l0).abs().max():.3e} (std {l0.std():.3e}); iou-pred max abs {(iou.cpu() - i0).abs().max():.3e}")
    if precision == "f16":
        assert ious.min() >= 0.999 and mi.min() >= 0.9999
    else:
        assert ious.min() >= 0.99 and mi.min() >= 0.999


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_decoder_alone_all_prompt_types(precision):
    """Install the ORACLE's fp32 embedding, then compare the decoder for every prompt combination the
    SAMRS drivers use (box-only, point-only with pad point, mask-only, point+box+mask)."""
    from oracle.make_golden import cases, run_predictor
    so = _oracle()
    name = "vit_tiny"
    pred = get_predictor(name, precision)
    orc = get_oracle(name)
    img = synth.make_image(0)
    orc.set_image(img)
    pred.set_image(img)                              # sets sizes / state
    pred.model.engine.set_embedding(orc.features.cuda(), pred.slot)
    tol_l2, tol_max = (1.5e-3, 6e-3) if precision == "f16" else (1.2e-2, 5e-2)
    bad = []
    for tag, kw, labels in cases(name):
        m, i, l = run_predictor(pred, lambda b, s: pred.transform.apply_boxes_torch(b.cuda(), s),
                                lambda c, s: pred.transform.apply_coords_torch(c.cuda(), s), img.shape[:2], kw)
        m0, i0, l0 = run_predictor(orc, so.apply_boxes, so.apply_coords, img.shape[:2], kw)
        err = (l.cpu() - l0).abs().max().item() / l0.std().item()
        l2 = ((l.cpu() - l0).norm() / l0.norm()).item()
        ierr = (i.cpu() - i0).abs().max().item()
        ious = iou_stats(m.cpu(), m0)
        print(f"decoder {precision} {tag}: low-res rel L2 {l2:.3e} max err / std {err:.3e}; iou-pred err {ierr:.3e}; mask IoU min {ious.min():.5f}")
        assert m.shape == m0.shape and l.shape == l0.shape and i.shape == i0.shape
        if not (l2 < tol_l2 and err < tol_max and ierr < (2e-3 if precision == "f16" else 2e-2)):
            bad.append(tag)
    assert not bad, bad


def test_operand_split_buys_what_the_error_budget_says():
    """The "split" option (two-term operand split of patch embed, neck, decoder out-projection and upscaler): against the fp32
    oracle the embedding error and the decoder's own error must drop by the factors oracle/error_budget.py predicts for these
    rounding points, and switching it off must reproduce the round-2 arithmetic's error class."""
    import samrs_amd
    so = _oracle()
    name = "vit_tiny80"
    sam = samrs_amd.sam_model_registry[name](precision="f16", max_prompts=8, max_points=1).to("cuda")
    pred = samrs_amd.SamPredictor(sam)
    eng = sam.engine
    orc = get_oracle(name)
    img = synth.make_image(0)
    orc.set_image(img)
    boxes = torch.from_numpy(synth.C1_BOXES)
    tb = so.apply_boxes(boxes, img.shape[:2])
    _, _, l0 = orc.predict_torch(None, None, tb, None, multimask_output=True)
    res = {}
    for split in (0, 15):
        eng.set_option("split", split)
        assert eng.get_option("split") == split
        pred.set_image(img)
        e_emb = ((pred.get_image_embedding().cpu() - orc.features).norm() / orc.features.norm()).item()
        eng.set_embedding(orc.features.cuda(), pred.slot)                  # decoder alone on the oracle's embedding
        _, _, l = pred.predict_torch(None, None, tb.cuda(), None, multimask_output=True)
        e_dec = ((l.cpu() - l0).norm() / l0.norm()).item()
        res[split] = (e_emb, e_dec)
        print(f"split={split}: embedding rel L2 {e_emb:.3e}, decoder-alone low-res rel L2 {e_dec:.3e}")
    eng.close()
    assert res[15][0] < 0.75 * res[0][0], res          # two-block encoder: patch embed + neck dominate its operand rounding
    assert res[15][1] < 0.5 * res[0][1], res           # out-projection + upscaler are ~90 % of the decoder's error variance


def test_split_block_gemms_one_launch_equals_three_passes():
    """Reference-grade mode at ViT-H: the three terms of every split block GEMM as one launch over a three-segment K axis
    (samrs_k_gemm_split3) against the generic route (three accumulating launches through an fp32 scratch).  Same products, same
    fp32 accumulation class: the embeddings agree far inside the mode's own error against the oracle (~1e-4 rel. L2)."""
    import samrs_amd
    sam = samrs_amd.sam_model_registry["vit_h"](precision="f16", max_prompts=8, max_points=1, options={"split": 63}).to("cuda")
    pred = samrs_amd.SamPredictor(sam)
    eng = sam.engine
    img = synth.make_image(3)
    emb = {}
    assert eng.get_option("lo_format") == 4                 # ViT-H default: the attention-side lo terms on MXFP4 operands
    eng.set_option("lo_format", 0)                          # same products on both routes: f16 lo terms
    for passes in (1, 0):
        eng.set_option("split_passes", passes)
        assert eng.get_option("split_passes") == passes
        pred.set_image(img)
        emb[passes] = pred.get_image_embedding().float().cpu()
    eng.set_option("lo_format", 4)                          # round 4: the lo terms of all four block GEMMs on the block-scaled fp4 MFMA
    pred.set_image(img)
    emb["mx"] = pred.get_image_embedding().float().cpu()
    eng.set_option("split", 15)
    pred.set_image(img)
    plain = pred.get_image_embedding().float().cpu()
    eng.close()
    d = ((emb[0] - emb[1]).norm() / emb[1].norm()).item()
    d_mx = ((emb["mx"] - emb[1]).norm() / emb[1].norm()).item()
    d_plain = ((plain - emb[1]).norm() / emb[1].norm()).item()
    print(f"one launch vs three passes: rel L2 {d:.2e}; MXFP4 lo terms vs exact lo terms: {d_mx:.2e}; default split vs reference-grade: {d_plain:.2e}")
    assert d < 2e-5 and d < 0.1 * d_plain
    assert d_mx < 0.5 * d_plain             # the fp4 corrections (now of ALL four block GEMMs) keep most of what the split buys (error budget plans9)


@pytest.mark.parametrize("name", ["vit_tiny", "vit_tiny80", "vit_b", "vit_l", "vit_h"])
def test_against_reference_golden(name, golden_dir):
    """Replay the committed fixtures produced by the REAL reference (oracle/make_golden.py).  vit_l (embed_dim 1024,
    build_sam.py:27-34) takes GEMM tile mixes no other registry entry reaches (N = 1024 / 3072 / 4096: not multiples of 320)."""
    from oracle.make_golden import cases, run_predictor
    so = _oracle()
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    pred = get_predictor(name, "f16")
    shapes = [(1024, 1024)] + ([(600, 800)] if name.startswith("vit_tiny") else [])
    for si, (h, w) in enumerate(shapes):
        img = synth.make_image(si, h, w)
        pred.set_image(img)
        f = pred.get_image_embedding().cpu()
        ref = torch.from_numpy(g[f"s{si}_emb_sample"])
        rel = ((f[0, ::16, ::4, ::4] - ref).norm() / ref.norm()).item()
        print(f"golden {name} {h}x{w}: embedding sample rel L2 {rel:.3e}")
        assert rel < 5e-3
        for tag, kw, labels in cases(name):
            kw = dict(kw)
            if (h, w) != (1024, 1024):
                for key in ("boxes", "point_coords"):
                    if key in kw:
                        kw[key] = kw[key] * np.float32(min(h, w) / 1024.0)
            m, i, l = run_predictor(pred, lambda b, s: pred.transform.apply_boxes_torch(b.cuda(), s),
                                    lambda c, s: pred.transform.apply_coords_torch(c.cuda(), s), (h, w), kw)
            k = f"s{si}_{tag}"
            lg = torch.from_numpy(g[k + "_low"])
            err = (l.cpu()[:, :, ::4, ::4] - lg).abs().max().item() / lg.std().item()
            l2 = ((l.cpu()[:, :, ::4, ::4] - lg).norm() / lg.norm()).item()
            area = m.flatten(2).sum(-1).cpu().numpy().astype(np.int64)
            ga = g[k + "_area"]
            rel_area = np.abs(area - ga) / np.maximum(ga, 1)
            print(f"golden {name} {h}x{w} {tag}: low-res rel L2 {l2:.3e} max err/std {err:.3e}; max rel area diff {rel_area.max():.3e}")
            assert l2 < 2.5e-3 and err < 1.5e-2, (name, tag)
            assert m.shape[-2:] == (h, w)
            if labels is not None:
                seg, _ = so.paint_semantic(m[:, 0].cpu().numpy(), labels, (h, w))
                frac = (seg != g[k + "_seg"]).mean()
                print(f"golden {name} {h}x{w} {tag}: painted class map mismatch {frac:.3e}")
                assert frac < 2e-3


def _unpack(bits, shape):
    return np.unpackbits(bits, axis=-1).reshape(*bits.shape[:-1], *shape).astype(bool)


# split 79 = 15 | 64 = the ViT-H DEFAULT: the cheap rounding points + the v third of qkv and proj on hi + lo operands in the
# leading three quarters of the blocks (split_depth 0 = automatic); built here WITHOUT options, so the default is what is tested
VIT_H_DEFAULT_SPLIT = 79


@pytest.mark.parametrize("name,split,variant", [("vit_b", 15, 0), ("vit_h", 15, 0), ("vit_h", 31, 0), ("vit_h", 63, 0),
                                                ("vit_h", 15, 1), ("vit_h", 31, 1), ("vit_h", 79, 0), ("vit_h", 79, 1),
                                                # a third draw, generated AFTER split 79 / depth 24 had been chosen on the first two
                                                ("vit_h", 79, 2), ("vit_h", 15, 2),
                                                # 207 = 79 | 128: the default + lin2 of every block with MXFP4 lo terms (the "margin" mode)
                                                ("vit_h", 207, 0), ("vit_h", 207, 1), ("vit_h", 207, 2)])
def test_c2_c4_against_reference_golden(name, split, variant, golden_dir):
    """BASELINE.json configs[1] (32 hboxes on one tile) and configs[3] (rbox -> enclosing hbox / rbox -> mask prompt,
    multimask_output=True) against FULL-RESOLUTION masks produced by the real reference on the realistic-margin weights
    (oracle/make_golden.py `extended`, fixtures tests/golden/<name>_c2c4.npz).  Asserted, at ViT-H too:
      * per-mask IoU >= 0.9995 on the C2 path (north_star asks 0.999) and relative area error <= 1.5e-3; C4 (the three
        multimask tokens: smaller masks with as many near-threshold pixels): >= 0.999 at ViT-B, >= 0.998 at ViT-H -- the
        four block GEMMs of the 32-block encoder in f16, with EVERYTHING else exact, already cost 0.99878 there
        (oracle/error_budget.py vit_h plans2, row "floor"), so no 1x-rate f16 path can promise 0.999 on
        this fixture; every MASK pixel whose reference logit has a margin of tau is reproduced exactly;
      * the painted class map is IDENTICAL to the reference's on every pixel whose reference decision has a margin of
        tau = 1 % of the logit spread (`c2_unstable` = the complement, computed from the reference's own logits), and
        the engine's logit error stays below tau -- i.e. the only pixels that may differ are the ones where the
        reference's own answer is decided by less than the f16 operand rounding noise (identity on ALL
        pixels is not attainable by any reduced-precision path on a continuous logit field; the count is printed);
      * low-res logits and IoU predictions within the f16 tolerances of this file's header.
    `split` = the engine's operand-split option: 15 = the cheap rounding points (patch embed, neck, decoder out-projection and
    upscaler on hi + lo operands: the default below ViT-H, and the "1x rate" mode at ViT-H); 79 = the ViT-H default = 15 + the v
    third of qkv and proj in the leading 24 blocks; 31 adds all of qkv + proj in every block, 63 every block GEMM.  From 79 on the C4
    fixtures clear the north star's 0.999 at ViT-H too, which shows that the 0.9983 - 0.9992 of split 15 is the price of running
    the block GEMMs at the 1x f16 rate and nothing else."""
    import samrs_amd
    from samrs_amd import transforms
    from oracle.make_golden import extended_inputs
    # variant 1: a second, independent draw of the fixture (other tile, other boxes) -- the floors below were set on variant 0
    g = np.load(os.path.join(golden_dir, name + "_c2c4" + (f"_v{variant}" if variant else "") + ".npz"))
    cfg = synth.CONFIGS[name]
    sd = synth.make_state_dict(cfg, 0, logit_scale=float(g["logit_scale"]))
    sam = samrs_amd.sam_model_registry[name](state_dict=sd, precision="f16", max_prompts=32, max_points=1,
                                             options=None if split == VIT_H_DEFAULT_SPLIT else {"split": split}).to("cuda")
    pred = samrs_amd.SamPredictor(sam)
    eng = sam.engine
    assert eng.get_option("split") == split and eng.get_option("split_depth") == 0
    inp = extended_inputs(variant)
    img = synth.make_image(inp["image_index"])
    hw = img.shape[:2]
    pred.set_image(img)
    f = pred.get_image_embedding().cpu()
    ref = torch.from_numpy(g["emb_sample"])
    rel = ((f[0, ::16, ::4, ::4] - ref).norm() / ref.norm()).item()
    print(f"c2c4 {name}: embedding sample rel L2 {rel:.3e}")
    assert rel < 5e-3
    tau_frac = float(g["tau_frac"])

    def check(tag, m, q, l, iou_floor=0.999):
        gm = torch.from_numpy(_unpack(g[tag + "_masks"], hw))
        lg = torch.from_numpy(g[tag + "_low"])
        std = float(g[tag + "_low_std"])
        err = (l.cpu()[:, :, ::4, ::4] - lg).abs().max().item() / std
        l2 = ((l.cpu()[:, :, ::4, ::4] - lg).norm() / lg.norm()).item()
        ious = iou_stats(m.cpu().flatten(0, 1), gm.flatten(0, 1))
        area = m.flatten(2).sum(-1).cpu().numpy().astype(np.int64)
        rel_area = (np.abs(area - g[tag + "_area"]) / np.maximum(g[tag + "_area"], 1)).max()
        qerr = (q.cpu() - torch.from_numpy(g[tag + "_iou"])).abs().max().item()
        flip = m.cpu() != gm
        flips = flip.flatten(2).sum(-1)
        near = torch.from_numpy(_unpack(g[tag + "_nearmask"], hw))          # reference pixels with |logit| < tau
        outside = int((flip & ~near).sum())
        print(f"c2c4 {name} split={split} v{variant} {tag}: IoU min {ious.min():.5f} mean {ious.mean():.5f}; max rel area diff {rel_area:.2e}; low-res rel L2 {l2:.2e} "
              f"max err/std {err:.2e} (tau {tau_frac:.0e}); iou-pred err {qerr:.2e}; flipped pixels per mask max {int(flips.max())} "
              f"(reference pixels within tau: max {int(g[tag + '_near'].max())})")
        assert outside == 0, f"{tag}: {outside} mask pixels differ where the reference's logit has margin"
        assert ious.min() >= iou_floor, (tag, ious.min().item())
        assert rel_area <= 1.5e-3 and l2 < 2.5e-3 and err < tau_frac and qerr < 5e-3, tag
        return gm

    # ---- C2: 32 hboxes, one call here == the reference's 20 + 12 chunks (bit-identical by construction, tested above) ----
    tb = pred.transform.apply_boxes_torch(torch.from_numpy(inp["boxes"]).cuda(), hw)
    m, q, l = pred.predict_torch(None, None, tb, None, multimask_output=False)
    check("c2", m, q, l, iou_floor=0.9995)
    seg = torch.full(hw, 255, dtype=torch.uint8, device="cuda")
    eng.paint(m[:, 0], torch.from_numpy(inp["labels"]), seg)
    seg = seg.cpu().numpy()
    unstable = _unpack(g["c2_unstable"], hw)
    diff = seg != g["c2_seg"]
    print(f"c2c4 {name} split={split} v{variant} class map: {int(diff.sum())} of {diff.size} pixels differ ({diff.mean():.2e}); reference-unstable pixels "
          f"{int(unstable.sum())} ({unstable.mean():.2e}); differing pixels outside the unstable set: {int((diff & ~unstable).sum())}")
    assert np.array_equal(seg[~unstable], g["c2_seg"][~unstable]), "class map differs where the reference's decision has margin"
    # round 2 (no operand split): 849 / 896 pixels at ViT-B / ViT-H; with the split rounding points the error budget
    # predicts 394 / 476 (oracle/error_budget.py plans2, row E1)
    # (error budget at ViT-H: 307 with the qkv + proj GEMMs split as well, 89 with every block GEMM split)
    # round 4: the lo terms of the block-GEMM splits run on MXFP4 operands by default (lo_format 4): error budget plans9 / plans10 predict
    # 148 pixels for split 63 (90 with exact lo terms), unchanged counts for 79 / 31
    # split 207 (79 + lin2 alone): statistical sample 312 px max against 392 for split 79 (profiles/r04_parity_stats.md)
    assert diff.sum() <= {0: 1000, 15: 620, 31: 420, 63: 200, 79: 480, 207: 400}[split], int(diff.sum())
    # ---- C4: enclosing hbox prompt, multimask ----
    # measured at ViT-H: 0.99874 / 0.99877 (split 15), 0.99911 / 0.99927 (31), 0.99974 / 0.99984 (63); the error budget
    # predicted 0.99886 / 0.99881, 0.99919 / 0.99928, 0.99976 / 0.99984
    # second draw (variant 1): 0.99924 / 0.99834 (split 15; round-2 arithmetic = split 0: see profiles/r03_ab.txt), 0.99952 / 0.99929 (31):
    # the C4 minimum is set by one small mask and moves by +-5e-4 between draws, so the default's floor is 0.998
    # split 79 (v + proj, 24 leading blocks; error budget plans6: 0.99911 / 0.99926; measured 0.99920 / 0.99928 and, second draw,
    # 0.99951 / 0.99914, at +6.7 ms per step instead of +21.6 for split 31): the ViT-H default holds the north star's 0.999 on C4
    # split 207: 0.99936 / 0.99942 over 96 + 96 sampled masks (margin 3.6e-4): the three fixture draws are asked for 0.9991
    c4_floor = 0.9991 if split == 207 else 0.999 if (name == "vit_b" or split >= 31) else 0.998
    tb = pred.transform.apply_boxes_torch(torch.from_numpy(inp["hboxes"]).cuda(), hw)
    m, q, l = pred.predict_torch(None, None, tb, None, multimask_output=True)
    assert m.shape[1] == 3
    # the three multimask tokens: smaller, noisier masks than token 0 on random weights; the zero-tolerance statement is
    # the margin one inside check()
    check("c4box", m, q, l, iou_floor=c4_floor)
    # ---- C4: rbox mask prompt (GPU rasteriser; bit-exact with the oracle that built the golden input), multimask ----
    prompts = transforms.rbox_mask_prompts(inp["polys"], hw, img_size=1024, device=torch.device("cuda"))
    assert abs(prompts.double().sum().item() - float(g["c4mask_prompt_sum"])) < 1e-6 * abs(float(g["c4mask_prompt_sum"])) + 1e-3
    m, q, l = pred.predict_torch(None, None, None, prompts[:, None], multimask_output=True)
    check("c4mask", m, q, l, iou_floor=c4_floor)
    sam.engine.close()


@pytest.mark.parametrize("name", ["vit_b", "vit_h"])
def test_instance_recipes_as_scripted_against_reference_golden(name, golden_dir):
    """VERDICT r03 "missing" 2: every reference instance driver calls ``multimask_output=False`` --
    main_sam_hbox_mask_instance.py:160-165 (point only, the point NOT run through apply_coords),
    main_sam_rbox_mask_instance.py:159-164 (mask prompt only), main_sam_rhbox_mask_instance.py:163-168 (enclosing hbox only).
    The three recipes through ``driver.InstancePrompter`` (the drop-in predictor surface) in the mode
    ``InstancePipeline(multimask=False)`` selects (split 15: single-mask output), against full-resolution masks of the REAL
    reference (tests/golden/<name>_inst.npz, oracle/make_golden.py instances): per-mask IoU >= 0.9995 -- the single-mask floor
    of the C2 path --, every pixel whose reference logit is at least tau = 2.5e-3 x std from the threshold reproduced
    exactly, low-res logits and IoU predictions within the f16 tolerances."""
    import samrs_amd
    from samrs_amd import driver, transforms
    from oracle.make_golden import instance_inputs
    g = np.load(os.path.join(golden_dir, name + "_inst.npz"))
    cfg = synth.CONFIGS[name]
    sd = synth.make_state_dict(cfg, 0, logit_scale=float(g["logit_scale"]))
    sam = samrs_amd.sam_model_registry[name](state_dict=sd, precision="f16", max_prompts=8, max_points=1, max_images=2).to("cuda")
    eng = sam.engine
    inp = instance_inputs(0)
    img = synth.make_image(inp["image_index"])
    hw = img.shape[:2]
    # the mode the product's single-mask instance pipeline runs its calls in
    mode = driver.InstancePipeline(sam, 37, prompt="point", multimask=False, batch=1, box_batch=8, max_boxes=8).split_mode
    assert mode == 15
    pred = samrs_amd.SamPredictor(sam)
    tau_frac = float(g["tau_frac"])
    with eng.options(split=mode):
        pred.set_image(img)
        assert eng.get_slot_info(pred.slot)["split"] == 15
        rel = ((pred.get_image_embedding().cpu()[0, ::16, ::4, ::4] - torch.from_numpy(g["emb_sample"])).norm() / torch.from_numpy(g["emb_sample"]).norm()).item()
        assert rel < 5e-3
        prompts = transforms.rbox_mask_prompts(inp["polys"], hw, img_size=1024, device=torch.device("cuda"))
        assert abs(prompts.double().sum().item() - float(g["mask_prompt_sum"])) < 1e-6 * abs(float(g["mask_prompt_sum"])) + 1e-3
        n = len(inp["polys"])
        calls = {
            "inst_point": dict(point_coords=torch.from_numpy(inp["points"]).cuda()[:, None, :], point_labels=torch.ones(n, 1, device="cuda")),
            "inst_mask": dict(point_coords=None, point_labels=None, mask_input=prompts[:, None]),
            "inst_rhbox": dict(point_coords=None, point_labels=None,
                               boxes=pred.transform.apply_boxes_torch(torch.from_numpy(inp["hboxes"]).cuda(), hw)),
        }
        prompter = driver.InstancePrompter(pred)
        for tag, kw in calls.items():
            m, q, l = pred.predict_torch(multimask_output=False, **kw)
            gm = torch.from_numpy(_unpack(g[tag + "_masks"], hw))
            near = torch.from_numpy(_unpack(g[tag + "_nearmask"], hw))
            ious = iou_stats(m.cpu().flatten(0, 1), gm.flatten(0, 1))
            flip = m.cpu() != gm
            lg = torch.from_numpy(g[tag + "_low"])
            err = (l.cpu()[:, :, ::4, ::4] - lg).abs().max().item() / float(g[tag + "_low_std"])
            qerr = (q.cpu() - torch.from_numpy(g[tag + "_iou"])).abs().max().item()
            print(f"instance recipe {name} {tag}: IoU min {ious.min():.5f} mean {ious.mean():.5f}; flipped px per mask max {int(flip.flatten(2).sum(-1).max())} "
                  f"(reference px within tau: max {int(g[tag + '_near'].max())}); flips outside tau {int((flip & ~near).sum())}; "
                  f"low-res max err / std {err:.2e} (tau {tau_frac:.1e}); iou-pred err {qerr:.2e}")
            assert int((flip & ~near).sum()) == 0, f"{tag}: mask pixels differ where the reference's logit has margin"
            # the mask-only recipe is the weakest single-mask workload (its dense prompt embedding replaces no_mask_embed on all
            # 4096 keys): 0.9993 at ViT-H in the 1x-rate mode on this fixture and on the 32-object sample of
            # test_vit_h_statistical_parity_sample, 0.99955 in the ViT-H default; the north star's 0.999 is its floor
            assert ious.min() >= (0.999 if (tag == "inst_mask" and name == "vit_h") else 0.9995), (tag, ious.min().item())
            assert err < 2 * tau_frac and qerr < 5e-3, tag
            # the product's prompter (chunking, point / mask / box plumbing) returns the same masks
            mode_name = {"inst_point": "point", "inst_mask": "rbox_mask", "inst_rhbox": "box"}[tag]
            pm, pq = prompter.predict(img, mode_name, hboxes=inp["hboxes"], rboxes=inp["polys"], points=inp["points"], already_set=True)
            assert torch.equal(pm, m[:, 0]) and torch.equal(pq, q[:, 0])
    eng.close()


def test_vit_h_statistical_parity_sample():
    """the ViT-H parity claims on a sample that can carry a "min" -- 8 tiles x 32 hboxes (256 single masks,
    8 painted class maps), 32 FAIR1M-shaped rboxes x 3 multimask outputs per prompt type (96 + 96 masks), the three instance
    drivers' scripted recipes (32 masks each), an 800 x 800 and a ragged 771 x 1163 tile -- checked against the PINNED oracle run
    on this machine's CPU (oracle/parity_sample.py; ~2 min, almost all of it the oracle's eleven ViT-H encoder passes), plus one
    long-tail tile of 128 boxes (C3).
    Asserted per precision mode (measured on MI355X, profiles/r04_parity_stats.md):
      split 15 (single-mask pipelines): every single-mask workload >= 0.9995 except the mask-prompt recipe (>= 0.999: measured
        min 0.99931), class-map pixels differing <= 620 per tile (measured max 537, mean 460 of 1 048 576);
      split 79 (the ViT-H default; multimask): c4box / c4mask >= 0.999 on all 96 + 96 masks (measured min 0.99927 / 0.99920);
      every mode: ZERO flipped mask pixels where the oracle's logit is further than tau = 2.5e-3 x std from the threshold, ZERO
        differing class-map pixels outside the set that tau makes unstable (1.5 % of a tile; round 3 excluded 4.3 - 4.9 % at
        tau = 1e-2), and the low-res logit error bounded."""
    import json
    import samrs_amd
    from oracle import parity_sample as ps
    so = _oracle()
    cfg = synth.CONFIGS["vit_h"]
    sd = synth.make_state_dict(cfg, 0, logit_scale=synth.MARGIN_LOGIT_SCALE)
    # round 6: built with the lo copies of EVERY block GEMM so that the sample also covers mode 63 (all four block GEMMs on hi + lo operands,
    # MXFP4 lo terms): the mode closest to the fp32 floor, reported by bench.py as `all_split_mode`.  (That an engine built WITHOUT options
    # starts in 79 is asserted by test_c2_c4_against_reference_golden; modes 15 / 79 do not depend on which extra copies exist.)
    sam = samrs_amd.sam_model_registry["vit_h"](state_dict=sd, precision="f16", max_prompts=32, max_points=1, options={"split": 63}).to("cuda")
    assert sam.engine.get_option("split") == 63 and sam.engine.get_option("lo_format") == 4
    sam.engine.set_option("allow_reduced", 1)              # the sample also runs the multimask workloads in mode 15, to report them
    pred = samrs_amd.SamPredictor(sam)
    # round 6: the floor.  The SAME fp32 oracle code in torch eager on this GPU (rocBLAS / MIOpen behind torch) against the host CPU:
    # what two fp32 backends of the reference algorithm disagree on is what "bit-identical class map" cannot be asked to beat
    orc_gpu = so.OraclePredictor({k: v.cuda() for k, v in sd.items()}, cfg)
    rec = ps.run(pred, so.OraclePredictor(sd, cfg), [15, 79, 63], orc_other=orc_gpu)
    summ = ps.summarise(rec)
    print(ps.table(summ))
    floor = summ.pop(ps.REF_BACKEND)
    fc2 = floor["c2"]
    print(f"reference-backend floor (fp32 oracle, torch eager on {torch.cuda.get_device_name(0)} vs host CPU), 8 C2 tiles: class-map pixels "
          f"differing mean {fc2['classmap_diff_mean']:.1f} max {fc2['classmap_diff_max']}, mask flips max {fc2['flips_max']}, IoU min "
          f"{fc2['iou_min']:.6f}; the engine: {summ[15]['c2']['classmap_diff_mean']:.0f} (mode 15) / {summ[79]['c2']['classmap_diff_mean']:.0f} (mode 79) "
          f"= {summ[15]['c2']['classmap_diff_mean'] / max(fc2['classmap_diff_mean'], 1e-9):.0f}x / "
          f"{summ[79]['c2']['classmap_diff_mean'] / max(fc2['classmap_diff_mean'], 1e-9):.0f}x that floor")
    if os.path.isdir("gpurun_out"):
        # what bench.py's `parity` object reads (copied to profiles/parity_stats.json): pinned to a hash of the device sources
        import bench
        json.dump({"tau_frac": ps.TAU_FRAC, "csrc_sha16": bench.csrc_sha(), "device": torch.cuda.get_device_name(0),
                   "summary": {str(k): v for k, v in summ.items()}, "reference_backend_floor": floor},
                  open("gpurun_out/parity_stats_test.json", "w"))
    assert fc2["n_masks"] == 256 and fc2["flips_outside_tau"] == 0 and fc2["classmap_diff_outside_unstable"] == 0
    for mode, tags in summ.items():
        for tag, s in tags.items():
            assert s["flips_outside_tau"] == 0, (mode, tag)
            assert s.get("classmap_diff_outside_unstable", 0) == 0, (mode, tag)
            assert s["low_err_over_std_max"] < (6e-3 if mode == 15 else 5e-3), (mode, tag, s["low_err_over_std_max"])
    m15, m79, m63 = summ[15], summ[79], summ[63]
    # mode 63: every workload, multimask included, with an order of magnitude of margin on the north star's bar
    for tag, t in m63.items():
        assert t["iou_min"] >= 0.9995, (tag, t["iou_min"])
    assert m63["c2"]["classmap_diff_max"] <= 200 and m63["c2"]["classmap_diff_mean"] < 0.5 * m15["c2"]["classmap_diff_mean"]
    assert m15["c2"]["n_masks"] == 256 and m79["c4box"]["n_masks"] == 96 and m79["c4mask"]["n_masks"] == 96
    # c3_long: one tile with 128 boxes (the long tail of the DOTA-shaped stream; the engine walks it in chunks of max_prompts = 32,
    # the reference in chunks of 20) -- same single-mask floor, and its 128-mask class map obeys the same zero-outside-tau rule above
    assert m15["c3_long"]["n_masks"] == ps.LONG_TAIL_BOXES
    for tag in ("c2", "c2_800", "c2_ragged", "inst_point", "inst_rhbox", "c3_long"):
        assert m15[tag]["iou_min"] >= 0.9995, (tag, m15[tag]["iou_min"])
        assert m79[tag]["iou_min"] >= 0.9995, (tag, m79[tag]["iou_min"])
    assert m15["inst_mask"]["iou_min"] >= 0.999 and m79["inst_mask"]["iou_min"] >= 0.9995
    assert m15["c2"]["classmap_diff_max"] <= 620 and m79["c2"]["classmap_diff_max"] <= 480
    for tag in ("c4box", "c4mask"):
        assert m79[tag]["iou_min"] >= 0.999 and m79[tag]["n_below_0999"] == 0, (tag, m79[tag])
        assert m15[tag]["iou_min"] >= 0.998, (tag, m15[tag])        # reported, not the mode this output is served in
    sam.engine.close()


def test_vit_b_c1_config_vs_oracle():
    """BASELINE.json configs[0]: ViT-B, one 1024^2 tile, 4 hboxes, CPU reference path."""
    so = _oracle()
    pred = get_predictor("vit_b", "f16")
    orc = get_oracle("vit_b")
    img = synth.make_image(0)
    pred.set_image(img)
    orc.set_image(img)
    boxes = torch.from_numpy(synth.C1_BOXES)
    tb = pred.transform.apply_boxes_torch(boxes.cuda(), img.shape[:2])
    masks, iou, low = pred.predict_torch(None, None, tb, None, multimask_output=False)
    m0, i0, l0 = orc.predict_torch(None, None, tb.cpu(), None, multimask_output=False)
    ious = iou_stats(masks.cpu(), m0)
    f, f0 = pred.get_image_embedding().cpu(), orc.features
    print(f"vit_b C1: embedding rel {((f - f0).norm() / f0.norm()).item():.3e}; IoU {ious.tolist()}")
    assert ious.min() >= 0.999


def test_paint_and_statistics_on_device():
    so = _oracle()
    pred = get_predictor("vit_tiny", "f16")
    img = synth.make_image(1)
    pred.set_image(img)
    boxes, labels = synth.make_boxes(1, 16)
    tb = pred.transform.apply_boxes_torch(torch.from_numpy(boxes).cuda(), img.shape[:2])
    masks, _, _ = pred.predict_torch(None, None, tb, None, multimask_output=False)
    eng = pred.model.engine
    seg = torch.full(img.shape[:2], 255, dtype=torch.uint8, device="cuda")
    cpix = torch.zeros(18, dtype=torch.int64, device="cuda")
    cins = torch.zeros(18, dtype=torch.int64, device="cuda")
    # two chunks, like the reference's 20-box batches (main_sam_hbox_semantic.py:157-181)
    a1 = eng.paint(masks[:10, 0], torch.from_numpy(labels[:10]), seg, cpix, cins)
    a2 = eng.paint(masks[10:, 0], torch.from_numpy(labels[10:]), seg, cpix, cins)
    seg0, areas0 = so.paint_semantic(masks[:, 0].cpu().numpy(), labels, img.shape[:2])
    pix0, ins0 = so.class_statistics(areas0, labels, 18)
    assert np.array_equal(seg.cpu().numpy(), seg0)                 # integer work: bit exact
    assert np.array_equal(torch.cat([a1, a2]).cpu().numpy(), areas0)
    assert np.array_equal(cpix.cpu().numpy(), pix0) and np.array_equal(cins.cpu().numpy(), ins0)


@pytest.mark.parametrize("h,w,n", [(64, 48, 70), (33, 21, 5), (1024, 1024, 3)])
def test_paint_kernel_paths_are_bit_exact(h, w, n):
    """samrs_paint on arbitrary u8 masks (any non-zero byte = set), more masks than one LDS counter chunk (64), sizes
    that take the 16-byte fused kernel (h*w % 16 == 0) and the byte-wise fallback: seg, areas and class counters equal
    the driver loop of main_sam_hbox_semantic.py:183-206 / statistic.py:15-21 (numpy restatement below)."""
    eng = get_predictor("vit_tiny", "f16").model.engine
    rng = np.random.default_rng(h * 1000 + n)
    masks = (rng.random((n, h, w)) < 0.15).astype(np.uint8) * rng.choice(np.array([1, 2, 128, 255], np.uint8), (n, h, w))
    masks[n // 2] = 0                                       # an empty mask: no instance, no pixels (statistic.py:18)
    labels = rng.integers(0, 18, n).astype(np.int32)
    seg0 = np.full((h, w), 255, np.uint8)
    for j in range(n):
        seg0[masks[j] != 0] = labels[j]
    areas0 = (masks != 0).reshape(n, -1).sum(1).astype(np.int64)
    pix0, ins0 = np.zeros(18, np.int64), np.zeros(18, np.int64)
    for j in range(n):
        if areas0[j] > 0:
            pix0[labels[j]] += areas0[j]
            ins0[labels[j]] += 1
    seg = torch.full((h, w), 255, dtype=torch.uint8, device="cuda")
    cpix = torch.zeros(18, dtype=torch.int64, device="cuda")
    cins = torch.zeros(18, dtype=torch.int64, device="cuda")
    areas = eng.paint(torch.from_numpy(masks).cuda(), torch.from_numpy(labels), seg, cpix, cins)
    assert np.array_equal(seg.cpu().numpy(), seg0)
    assert np.array_equal(areas.cpu().numpy(), areas0)
    assert np.array_equal(cpix.cpu().numpy(), pix0) and np.array_equal(cins.cpu().numpy(), ins0)


def test_error_behaviour_matches_reference():
    import samrs_amd
    pred = get_predictor("vit_tiny", "f16")
    pred.reset_image()
    with pytest.raises(RuntimeError, match="An image must be set"):
        pred.predict_torch(None, None, torch.zeros(1, 4).cuda(), None)
    with pytest.raises(RuntimeError, match="An image must be set"):
        pred.get_image_embedding()
    with pytest.raises(AssertionError):
        pred.set_image(synth.make_image(0), image_format="XYZ")
    with pytest.raises(AssertionError):
        pred.set_torch_image(torch.zeros(1, 3, 512, 512), (512, 512))
    pred.set_image(synth.make_image(0))
    with pytest.raises(AssertionError):
        pred.predict(point_coords=np.zeros((1, 2)), point_labels=None)
    # numpy convenience wrapper (predictor.py:92-166)
    m, q, l = pred.predict(box=np.array([100, 100, 400, 300]), multimask_output=True)
    assert m.shape == (3, 1024, 1024) and m.dtype == bool and q.shape == (3,) and l.shape == (3, 256, 256)
    # strict weight loading (build_sam.py:106)
    sd = dict(synth.make_state_dict(synth.CONFIGS["vit_tiny"], 0))
    sd.pop("mask_decoder.iou_token.weight")
    with pytest.raises(Exception, match="missing tensor"):
        samrs_amd.sam_model_registry["vit_tiny"](state_dict=sd).to("cuda")


def test_engine_options_are_per_handle():
    """samrs_set_option: every handle has its own options (no process-wide switches on the product path), unknown names are
    refused, and the reference-grade split bits cannot be switched on once the weights were finalized without their lo copies."""
    import samrs_amd
    a = samrs_amd.sam_model_registry["vit_tiny"](max_prompts=4).to("cuda").engine
    b = samrs_amd.sam_model_registry["vit_tiny"](max_prompts=4, options={"split": 63, "decoder_fusion": 0}).to("cuda").engine
    assert (a.get_option("split"), a.get_option("decoder_fusion"), a.get_option("upscaler_fused")) == (15, 1, 1)
    assert (b.get_option("split"), b.get_option("decoder_fusion")) == (63, 0)
    a.set_option("split", 3)
    a.set_option("gemm_variant", 7)
    assert a.get_option("split") == 3 and b.get_option("split") == 63 and b.get_option("gemm_variant") == -1
    with pytest.raises(AssertionError, match="unknown option"):
        a.set_option("no_such_option", 1)
    with pytest.raises(AssertionError, match="lo weights"):
        a.set_option("split", 31)                       # bits 16 / 32 need lo copies of the block weights, prepared at load
    b.set_option("split", 15); b.set_option("split", 63)    # ... which b has: can be cleared and set again
    # the two handles give different embeddings for the same tile (different arithmetic), each reproducibly
    t = torch.as_tensor(synth.make_image(2), device="cuda")[None].contiguous()
    a.set_option("split", 15); a.set_option("gemm_variant", -1)
    a.set_images(t, 0); b.set_images(t, 0)
    ea, eb = a.get_embedding(0).clone(), b.get_embedding(0).clone()
    a.set_images(t, 0)
    assert torch.equal(a.get_embedding(0), ea) and not torch.equal(ea, eb)
    assert ((ea - eb).norm() / eb.norm()).item() < 2e-3
    a.close(); b.close()


def test_encoder_batch_equals_single():
    """Batched set_images (config 2: 8 tiles per encoder pass) == one tile at a time."""
    pred = get_predictor("vit_tiny", "f16", max_images=4)
    eng = pred.model.engine
    imgs = torch.stack([torch.as_tensor(synth.make_noise_image(i)) for i in range(3)]).cuda()
    eng.set_images(imgs, 0)
    batch = [eng.get_embedding(i).clone() for i in range(3)]
    for i in range(3):
        eng.set_images(imgs[i:i + 1].contiguous(), 3)
        single = eng.get_embedding(3)
        assert torch.equal(single, batch[i]), f"image {i}: batched encode differs from single encode"


def test_generation_driver_end_to_end(tmp_path):
    """N2 row: images + boxes in, gray/color PNG + ins/*.pkl out (main_sam_hbox_semantic.py:195-216),
    checked against the oracle's painting of the oracle's masks."""
    import json
    import pickle
    from PIL import Image
    from samrs_amd import generate, rle
    so = _oracle()
    img_dir, out_dir = tmp_path / "img", tmp_path / "out"
    img_dir.mkdir()
    ann = {}
    for i in range(2):
        Image.fromarray(synth.make_image(10 + i)).save(img_dir / f"P{i:04d}.png")
        b, l = synth.make_boxes(10 + i, 23)          # 23 boxes -> chunks of 20 + 3
        ann[f"P{i:04d}"] = {"boxes": b.tolist(), "labels": l.tolist()}
    (tmp_path / "boxes.json").write_text(json.dumps(ann))
    stats = generate.run(generate.argparse.Namespace(
        images=str(img_dir), boxes=str(tmp_path / "boxes.json"), out=str(out_dir), model="vit_tiny", checkpoint=None,
        precision="f16", classes=None, n_classes=18, palette=None, box_batch=20, no_rle=False, log=str(tmp_path / "run.jsonl")))
    lines = [json.loads(l) for l in open(tmp_path / "run.jsonl")]
    assert lines[0]["images_todo"] == 2 and sum(len(l.get("images", [])) for l in lines[1:]) == 2 and lines[-1]["done"] == 2
    assert sorted(b for l in lines[1:] for b in l["boxes"]) == [23, 23]
    orc = get_oracle("vit_tiny")
    tot_pix = np.zeros(18, np.int64)
    for i in range(2):
        stem = f"P{i:04d}"
        gray = np.array(Image.open(out_dir / "gray" / (stem + ".png")))
        info = pickle.load(open(out_dir / "ins" / (stem + ".pkl"), "rb"))
        boxes, labels = np.asarray(ann[stem]["boxes"], np.float32), np.asarray(ann[stem]["labels"])
        img = synth.make_image(10 + i)
        orc.set_image(img)
        m0 = torch.cat([orc.predict_torch(None, None, so.apply_boxes(torch.from_numpy(boxes[s:e]), img.shape[:2]), None,
                                          multimask_output=False)[0] for s, e in so.box_chunks(len(labels), 20)])[:, 0].numpy()
        seg0, areas0 = so.paint_semantic(m0, labels, img.shape[:2])
        assert (gray != seg0).mean() < 2e-3
        assert len(info) == len(labels)
        for j, d in enumerate(info):
            m = rle.decode(d["mask"])
            assert int(m.sum()) == d["size"] and d["label"] == int(labels[j])
            assert abs(d["size"] - areas0[j]) <= max(8, 2e-3 * areas0[j])
            if d["size"] > 0:
                tot_pix[d["label"]] += d["size"]
    assert stats["class_pixel_num"] == tot_pix.tolist()
    # N4 (BASELINE.json configs[4]): the files this run wrote go through the reference's training read path
    # (oracle/consumer_check.py: PF/End_to_End/datasets.py:185-273 + CrossEntropyLoss(ignore_index=255), main_pretrain.py:321);
    # the two-rank DDP form of the same check runs on CPU (tests/test_consumer_gloo.py)
    from oracle import consumer_check as cc
    (tmp_path / "train.txt").write_text("P0000\nP0001\n")
    (tmp_path / "valid.txt").write_text("P0001\n")
    ds = cc.SegmentationDataset(str(tmp_path), str(img_dir), str(out_dir / "gray"), flag="trn")
    losses, _, drawn, hist = cc.train_steps(ds, 18, 0, 1, steps=2, batch_size=1)
    assert drawn == [0, 1] and all(np.isfinite(losses)) and hist[18:255].sum() == 0 and hist[:18].sum() > 0
    # --resume: nothing is missing -> nothing is recomputed, nothing is rewritten; remove one pickle -> exactly that image
    # (a second run() in the same process also exercises the per-run work-queue key)
    before = {p: os.stat(p).st_mtime_ns for p in sorted(str(x) for x in out_dir.rglob("*.p*"))}
    ns = generate.argparse.Namespace(
        images=str(img_dir), boxes=str(tmp_path / "boxes.json"), out=str(out_dir), model="vit_tiny", checkpoint=None,
        precision="f16", classes=None, n_classes=18, palette=None, box_batch=20, no_rle=False, resume=True)
    # round 4: a resumed run's statistics cover the WHOLE output directory (the completed images contribute through their
    # ins/*.pkl, as Generate Dataset/statistic.py:12-21,44-49 computes them) instead of overwriting class_stats.json with
    # the statistics of the re-run images only
    full = {k: stats[k] for k in ("class_pixel_num", "class_instance_num", "mask_num")}
    s2 = generate.run(ns)
    assert {k: s2[k] for k in full} == full and before == {p: os.stat(p).st_mtime_ns for p in before}
    os.remove(out_dir / "ins" / "P0001.pkl")
    s3 = generate.run(ns)
    assert {k: s3[k] for k in full} == full                          # the same image recomputed: the same totals
    assert json.load(open(out_dir / "statistic" / "class_stats.json"))["mask_num"] == full["mask_num"]
    assert os.path.exists(out_dir / "ins" / "P0001.pkl")
    assert os.stat(out_dir / "ins" / "P0000.pkl").st_mtime_ns == before[str(out_dir / "ins" / "P0000.pkl")]
    assert not [f for f in os.listdir(out_dir / "ins") if ".tmp" in f]


def test_vit_h_full_size_properties():
    """BASELINE.json configs[1] at full size (ViT-H, 8 x 1024^2 tiles, 32 boxes per tile), through
    size-independent properties (the CPU oracle needs ~10 s per ViT-H tile, the goldens cover one tile):
      * batched encoder == one tile at a time, bit for bit (rows are independent in every kernel);
      * 32 boxes in one call == the reference's 20 + 12 chunking (main_sam_hbox_semantic.py:157-181), bit for
        bit (no atomics anywhere on the path);
      * on-device painting / areas == recomputation from the returned masks (integer, exact);
      * predictions are bit-reproducible across repeated calls."""
    pred = get_predictor("vit_h", "f16", max_prompts=32, max_images=8)
    eng = pred.model.engine
    tiles = torch.stack([torch.as_tensor(synth.make_noise_image(40 + i)) for i in range(8)]).cuda()
    eng.set_images(tiles, 0)
    emb3 = eng.get_embedding(3).clone()
    emb7 = eng.get_embedding(7).clone()
    eng.set_images(tiles[3:4].contiguous(), 0)
    assert torch.equal(eng.get_embedding(0), emb3), "batched encode differs from single-tile encode"
    eng.set_images(tiles, 0)
    assert torch.equal(eng.get_embedding(7), emb7), "encoder is not deterministic"
    boxes, labels = synth.make_boxes(40, 32)
    b = torch.from_numpy(boxes).cuda()
    size = (1024, 1024)
    m_all, q_all, l_all = eng.predict(3, b, None, None, None, False, False, size, size)
    parts = [eng.predict(3, b[s:e], None, None, None, False, False, size, size) for s, e in [(0, 20), (20, 32)]]
    m_ch = torch.cat([p[0] for p in parts]); l_ch = torch.cat([p[2] for p in parts]); q_ch = torch.cat([p[1] for p in parts])
    scale = l_all.std().item()
    print(f"vit_h 32 boxes vs 20+12: low-res max diff / std {(l_all - l_ch).abs().max().item() / scale:.2e}, "
          f"mask pixels differing {(m_all != m_ch).sum().item()} of {m_all.numel()}")
    assert torch.equal(l_all, l_ch) and torch.equal(q_all, q_ch) and torch.equal(m_all, m_ch)
    m_rep, q_rep, l_rep = eng.predict(3, b, None, None, None, False, False, size, size)
    assert torch.equal(l_rep, l_all) and torch.equal(m_rep, m_all), "predict is not reproducible"
    seg = torch.full(size, 255, dtype=torch.uint8, device="cuda")
    cp = torch.zeros(18, dtype=torch.int64, device="cuda"); ci = torch.zeros(18, dtype=torch.int64, device="cuda")
    areas = eng.paint(m_all[:, 0], torch.from_numpy(labels), seg, cp, ci)
    assert torch.equal(areas, m_all[:, 0].flatten(1).sum(1))
    seg_ref = torch.full(size, 255, dtype=torch.uint8, device="cuda")
    for j in range(32):
        seg_ref[m_all[j, 0]] = int(labels[j])
    assert torch.equal(seg, seg_ref)
    assert int(cp.sum()) == int(areas.sum()) and int(ci.sum()) == int((areas > 0).sum())


def test_decoder_fused_kernels_match_unfused():
    """The fused decoder kernels (i2t attention + projection + residual + LayerNorm; ConvT1 GEMM + LayerNorm2d + GELU;
    ConvT2 + GELU + hypernetwork product) against the same computation as separate launches, ViT-H tile, 32 boxes:
    identical rounding points except the fp32 summation order of the LayerNorm statistics and the un-rounded GELU
    output in the last product."""
    pred = get_predictor("vit_h", "f16", max_prompts=32, max_images=8)
    eng = pred.model.engine
    tile = torch.as_tensor(synth.make_noise_image(40))[None].cuda()
    eng.set_images(tile, 0)
    boxes, _ = synth.make_boxes(40, 32)
    b = torch.from_numpy(boxes).cuda()
    size = (1024, 1024)
    split = eng.get_option("split")
    try:
        # like with like: the un-fused kernels never split their operands, so the fused ones run un-split here too
        # (the split itself is tested kernel by kernel and against the oracle / the reference fixtures)
        eng.set_option("split", split & 3)
        eng.set_option("decoder_fusion", 0)
        m0, q0, l0 = eng.predict(0, b, None, None, None, False, False, size, size)
        m0, q0, l0 = m0.clone(), q0.clone(), l0.clone()
        eng.set_option("decoder_fusion", 1)
        m1, q1, l1 = eng.predict(0, b, None, None, None, False, False, size, size)
    finally:
        eng.set_option("decoder_fusion", 1)
        eng.set_option("split", split)
    rel = ((l1 - l0).norm() / l0.norm()).item()
    mx = ((l1 - l0).abs().max() / l0.std()).item()
    diff = (m1 != m0).sum().item()
    print(f"fused vs unfused decoder: low-res rel L2 {rel:.2e}, max / std {mx:.2e}, mask pixels differing {diff} of {m0.numel()}")
    assert rel < 5e-4 and mx < 5e-3
    # random weights put many logits next to the 0 threshold: a 1e-4-relative logit change flips a few hundred of 33.5 M pixels
    assert diff < 1e-4 * m0.numel()
    assert (q1 - q0).abs().max().item() < 1e-3


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 path -- barrier + MAX-over-ranks timing, the shared-counter work queue named per phase, the statistics
    all-reduce on device tensors -- with both ranks on this GPU over gloo (SAMRS_BENCH_SHARE_GPU=1): the control flow the
    driver's multi-GPU run takes, exercised on every 1-GPU box.  RCCL itself needs a multi-GPU node and is not covered."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SAMRS_BENCH_SHARE_GPU="1", PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--model", "vit_tiny", "--workload", "c3",
           "--steps", "3", "--warmup", "1", "--batch", "2", "--no-cpu-baseline", "--no-alt-dtype", "--no-pcie-leg", "--no-rle-leg"]
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["tiles_per_step"] == 4
    assert d["value"] > 0 and abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]      # whole-job tiles / MAX time
    assert d["stats_allreduce"]["total_instances"] > 0


def test_bench_self_launches_for_two_gpus():
    """``python bench.py --gpus 2 ...`` -- the form the driver uses for N = 1, with no torchrun around it and
    no WORLD_SIZE in the environment -- must launch its own ranks (bench._self_launch: torch.distributed.run on 127.0.0.1, a free
    port) instead of asserting.  Both ranks share this GPU over gloo (SAMRS_BENCH_SHARE_GPU=1).  Checked on the ONE JSON line:
    n_gpus 2, image-parallel x2, and the statistics all-reduce (the one collective of the path, statistic.py:15-21) equals the
    serial sum of what each rank painted locally."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SAMRS_BENCH_SHARE_GPU="1", PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "vit_tiny",
           "--batch", "2", "--no-cpu-baseline", "--no-alt-dtype", "--no-pcie-leg", "--no-rle-leg"]
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "image-parallel x2" and d["config"]["tiles_per_step"] == 4
    st = d["stats_allreduce"]
    assert len(st["per_rank_local"]) == 2 and all(p["instances"] > 0 for p in st["per_rank_local"])     # both ranks did work
    assert st["total_pixels"] == sum(p["pixels"] for p in st["per_rank_local"])
    assert st["total_instances"] == sum(p["instances"] for p in st["per_rank_local"])
    assert 0 < st["total_instances"] <= 2 * (2 + 1) * 2 * 32        # ranks x (steps + warm-up) x batch x boxes; empty masks are not instances


def test_generate_two_ranks_on_one_gpu_under_a_four_cpu_mask(tmp_path):
    """The generation CLI's N > 1 path on a ONE-GPU box: two ranks of ``python -m samrs_amd.generate`` share
    cuda:0 over gloo (SAMRS_SHARE_GPU=1) inside a four-CPU affinity mask -- the share eight ranks get on the pool's 16-CPU
    containers -- with the shared-counter schedule, then a --resume pass after one image's pickle was removed.  Every file must
    be byte-identical to the single-rank run's (a tile's outputs do not depend on which rank or batch position computed it),
    the merged statistics equal, both ranks must have done work, and the thread pools must have been sized from the mask."""
    import json
    import shutil
    import subprocess
    import sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    img_dir = tmp_path / "img"
    img_dir.mkdir()
    ann = {}
    for i in range(7):                                              # 7 images in batches of 2: a ragged last batch
        Image.fromarray(synth.make_image(40 + i)).save(img_dir / f"T{i:04d}.png")
        b, l = synth.make_boxes(40 + i, 5 + 3 * i)
        ann[f"T{i:04d}"] = {"boxes": b.tolist(), "labels": l.tolist()}
    (tmp_path / "boxes.json").write_text(json.dumps(ann))
    cpus = sorted(os.sched_getaffinity(0))[:4]
    mask = ",".join(str(c) for c in cpus)
    common = ["--images", str(img_dir), "--boxes", str(tmp_path / "boxes.json"), "--model", "vit_tiny", "--batch", "2", "--box-batch", "8",
              "--timing"]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    taskset = ["taskset", "-c", mask] if shutil.which("taskset") else []

    def launch(world, out, extra=()):
        cmd = list(taskset)
        if world > 1:
            cmd += [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                    "--master-port", "29547", "-m", "samrs_amd.generate"]
        else:
            cmd += [sys.executable, "-m", "samrs_amd.generate"]
        cmd += common + ["--out", str(out), "--run-log", str(out) + ".jsonl"] + list(extra)      # --log: ambiguous to torchrun's parser
        r = subprocess.run(cmd, env=dict(env, SAMRS_SHARE_GPU="1"), cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return r

    one, two = tmp_path / "one", tmp_path / "two"
    launch(1, one)
    launch(2, two, ["--schedule", "dynamic"])

    def tree(d):
        return {os.path.relpath(os.path.join(b, f), d): open(os.path.join(b, f), "rb").read()
                for b, _, fs in os.walk(d) for f in fs if "statistic" not in b}
    t1, t2 = tree(one), tree(two)
    assert sorted(t1) == sorted(t2) and len(t1) == 3 * 7
    assert all(t1[k] == t2[k] for k in t1), [k for k in t1 if t1[k] != t2[k]]
    s1 = json.load(open(one / "statistic" / "class_stats.json"))
    s2 = json.load(open(two / "statistic" / "class_stats.json"))
    for k in ("class_pixel_num", "class_instance_num", "mask_num"):
        assert s1[k] == s2[k], k
    assert sorted(np.load(one / "statistic" / "all_mask_size.npy").tolist()) == sorted(np.load(two / "statistic" / "all_mask_size.npy").tolist())
    per_rank = []
    for r in range(2):
        lines = [json.loads(l) for l in open(f"{two}.jsonl.rank{r}")]
        assert lines[0]["world"] == 2 and lines[0]["rank"] == r
        per_rank.append(sum(len(l.get("images", [])) for l in lines[1:]))
    assert sum(per_rank) == 7 and min(per_rank) >= 1, per_rank          # the shared counter hands every rank some of the batches
    if taskset:
        t = s2["timing"]                                                # rank 0's pools: 4 CPUs / 2 ranks = 2 -> the floor of 2 + 2
        assert t["cpu_budget"] <= 2.0 and t["readers"] == 2 and t["writers"] == 2, t
    # --resume across ranks: rank 0 lists the directory, both ranks index the same todo list; the statistics still cover all 7
    os.remove(two / "ins" / "T0003.pkl")
    keep = {k: os.stat(two / k).st_mtime_ns for k in t2 if "T0003" not in k}
    launch(2, two, ["--resume"])
    assert tree(two) == t2
    assert keep == {k: os.stat(two / k).st_mtime_ns for k in keep}     # nothing else was rewritten
    s3 = json.load(open(two / "statistic" / "class_stats.json"))
    for k in ("class_pixel_num", "class_instance_num", "mask_num"):
        assert s1[k] == s3[k], k


@pytest.mark.parametrize("batch", [2, 3, 5])
def test_vit_h_odd_batches_equal_single_tile(batch):
    """The GEMM tile shape is chosen per (M, N, K): 2, 3 and 5 tiles per encoder pass take other mixes of the
    256x320 / 256x256 / 256x128 kernels than 1 or 8 tiles do.  Every mix must give the same bits per tile
    (each output element is accumulated over k in the same order by every tile shape)."""
    pred = get_predictor("vit_h", "f16", max_prompts=32, max_images=8)
    eng = pred.model.engine
    tiles = torch.stack([torch.as_tensor(synth.make_noise_image(60 + i)) for i in range(batch)]).cuda()
    eng.set_images(tiles, 0)
    embs = [eng.get_embedding(i).clone() for i in range(batch)]
    for i in (0, batch - 1):
        eng.set_images(tiles[i:i + 1].contiguous(), 7)
        assert torch.equal(eng.get_embedding(7), embs[i]), f"batch {batch}, tile {i}: differs from single-tile encode"


@pytest.mark.parametrize("split", [15, 79])
def test_operand_row_padding_is_bit_identical(split):
    """Round 5: at ViT-H the K = 1280 operands of the plain qkv / lin1 launches (the LayerNorm output, the weights) are stored with a row
    stride of 1408 elements instead of 1280 (option "operand_pad", engine field ldk; the persistent ET kernels take the stride:
    gemm.hip tl_gemm_ld) -- a 2560-byte row is ten 256-byte units and lands the rows of a tile on half of the memory channels.  Where the
    bytes live cannot change what is computed: the embeddings of 4 and 8 tiles (the batches whose shapes fill whole rounds of tiles and
    therefore take the padded layout) must equal the dense layout's bit for bit, in the single-mask mode and in the multimask default
    (whose lin1, and qkv of the last blocks, run plain)."""
    pred = get_predictor("vit_h", "f16", max_prompts=32, max_images=8)
    eng = pred.model.engine
    tiles = torch.stack([torch.as_tensor(synth.make_noise_image(90 + i)) for i in range(8)]).cuda()
    with eng.options(split=split):
        for n in (8, 4):
            embs = {}
            for pad in (1, 0, 1):
                with eng.options(operand_pad=pad):
                    assert eng.get_option("operand_pad") == pad
                    eng.set_images(tiles[:n].contiguous(), 0)
                    got = torch.stack([eng.get_embedding(i).clone() for i in range(n)])
                if pad in embs:
                    assert torch.equal(got, embs[pad]), f"split {split}, {n} tiles: the padded layout is not reproducible from launch to launch"
                embs[pad] = got
            assert embs[1].abs().max() > 0
            assert torch.equal(embs[0], embs[1]), f"split {split}, {n} tiles: padded operand rows changed the embedding"


def test_f16_operand_range_stress_and_saturation_counter():
    """VERDICT r04 item 4 / "missing" 6: every f16 conversion on the path saturates at 65504 -- silently -- and every other
    test runs on N(0, sigma) weights.  Here the weights get the outlier structure checkpoints have (synth.heavy_tailed: a few
    lin1 rows / v rows x 3e3, LayerNorm gammas x 30, so that GELU(lin1), v and the attention output reach 1e4 - 1e5) and the
    SAME dict is evaluated by the fp32 oracle:
      * activations at 2.7e4 - 3.4e4 (inside f16): the operand-range check counts nothing and parity holds at the bar of the
        ordinary weights (f16's relative precision does not depend on magnitude);
      * activations at 8e5 - 1.1e6 (outside): the counter (option "range_check" = 1, read through "saturated") fires, the
        embedding is visibly wrong, "range_check" = 2 turns the pass into SAMRS_ERR_RANGE (engine.OperandRangeError), and the
        documented remedy -- the bf16 operand type -- evaluates the same weights with nothing saturated at bf16's own tolerance."""
    import samrs_amd
    from samrs_amd import engine
    so = _oracle()
    name = "vit_tiny"
    cfg = synth.CONFIGS[name]
    base = synth.make_state_dict(cfg, 0)
    img = synth.make_image(0)
    x = so.preprocess(img, cfg.img_size)
    boxes = torch.from_numpy(synth.C1_BOXES)

    def oracle_maxima(sd):
        mx = {}

        def rec(k):
            def f(t):
                mx[k] = max(mx.get(k, 0.0), float(t.abs().max()))
                return t
            return f
        with torch.no_grad():
            so.image_encoder(sd, cfg, x, so.Rounding(points={k: rec(k) for k in so.ENC_POINTS}))
        return max(mx[k] for k in ("enc.qkv_in", "enc.qkv_out", "enc.proj_in", "enc.lin1_in", "enc.lin2_in", "enc.neck0"))

    def build(sd, precision, check):
        sam = samrs_amd.sam_model_registry[name](state_dict=sd, precision=precision, max_prompts=8, max_points=1,
                                                 options={"range_check": check}).to("cuda")
        return sam, samrs_amd.SamPredictor(sam)

    def compare(pred, sd):
        orc = so.OraclePredictor(sd, cfg)
        orc.set_image(img)
        pred.set_image(img)
        rel = ((pred.get_image_embedding().cpu() - orc.features).norm() / orc.features.norm()).item()
        tb = pred.transform.apply_boxes_torch(boxes.cuda(), img.shape[:2])
        m, _, _ = pred.predict_torch(None, None, tb, None, multimask_output=False)
        m0, _, _ = orc.predict_torch(None, None, tb.cpu(), None, multimask_output=False)
        return rel, iou_stats(m.cpu(), m0).min().item()

    # ---- inside the range ----
    sd_in = synth.heavy_tailed(base, cfg, 0, hidden_scale=3e3, v_scale=3e3, gamma_scale=30.0)
    top = oracle_maxima(sd_in)
    assert 1e4 < top < 6e4, top                       # the stress reaches the decade below f16's maximum
    sam, pred = build(sd_in, "f16", 1)
    assert sam.engine.get_option("range_check") == 1 and sam.engine.get_option("saturated") == 0
    # (round 6: the engine now finds the outlier columns at load time and carries their hi + lo terms -- tests/test_outlier_gpu.py; this
    # test is about RANGE, so it looks at the plain f16 arithmetic first and at the remedy second)
    sam.engine.set_option("outlier_cols", 0)
    rel, iou = compare(pred, sd_in)
    print(f"heavy-tailed weights, operands up to {top:.3g}: f16 embedding rel L2 {rel:.3e}, box-mask IoU min {iou:.5f}, "
          f"saturated {sam.engine.get_option('saturated')}")
    assert sam.engine.get_option("saturated") == 0
    # Nothing saturates, but parity is NOT what it is on the ordinary weights (measured: embedding rel L2 3.8e-3 against 1e-3, IoU
    # min 0.9980): an operand's rounding error is relative to ITS magnitude, so one 3e4-sized element of a dot product carries an
    # absolute error of 8 -- more than the O(1) terms next to it contribute at all.  That is f16's 11 bits meeting outlier
    # channels, not a range problem; the bound asserted here is the measured class, and the two-term operand split (the
    # reference-grade bits of "split": 2^-22 operand error) is the remedy that is asserted to help below.
    assert rel < 1e-2 and iou >= 0.995, (rel, iou)
    sam.engine.set_option("outlier_cols", 7)
    rel_oc, iou_oc = compare(pred, sd_in)
    print(f"the same weights with the outlier columns' hi + lo terms (the default): embedding rel L2 {rel_oc:.3e}, box-mask IoU min {iou_oc:.5f}")
    assert sam.engine.get_option("saturated") == 0 and rel_oc < 0.6 * rel and iou_oc >= iou, (rel_oc, iou_oc)
    sam.engine.close()
    sam = samrs_amd.sam_model_registry[name](state_dict=sd_in, precision="f16", max_prompts=8, max_points=1,
                                             options={"split": 63, "range_check": 1, "outlier_cols": 0}).to("cuda")
    rel63, iou63 = compare(samrs_amd.SamPredictor(sam), sd_in)
    print(f"the same weights with every MFMA operand split (63): embedding rel L2 {rel63:.3e}, box-mask IoU min {iou63:.5f}")
    # (not all of the error is operand rounding of the split GEMMs: q / k / v and the softmax probabilities are STORED in f16 in
    # every mode; measured 2.0e-3 / 0.99897 against 3.8e-3 / 0.99802)
    assert sam.engine.get_option("saturated") == 0 and rel63 < 0.7 * rel and iou63 >= iou, (rel63, iou63)
    sam.engine.close()

    # ---- outside ----
    sd_out = synth.heavy_tailed(base, cfg, 0, hidden_scale=1e5, v_scale=1e5, gamma_scale=30.0)
    top = oracle_maxima(sd_out)
    assert top > 2e5, top
    sam, pred = build(sd_out, "f16", 1)
    rel, iou = compare(pred, sd_out)
    n_sat = sam.engine.get_option("saturated")
    print(f"heavy-tailed weights, operands up to {top:.3g}: f16 embedding rel L2 {rel:.3e}, box-mask IoU min {iou:.5f}, saturated {n_sat}")
    assert n_sat > 0 and rel > 1e-2, (n_sat, rel)    # counted, and the damage is real
    sam.engine.set_option("saturated", 0)
    assert sam.engine.get_option("saturated") == 0    # write = reset
    sam.engine.set_option("range_check", 2)
    with pytest.raises(engine.OperandRangeError, match="saturated"):
        pred.set_image(img)
    sam.engine.close()
    sam, pred = build(sd_out, "bf16", 2)               # the remedy: fp32 exponent range; would raise if anything overflowed
    rel, iou = compare(pred, sd_out)
    print(f"the same weights on bf16 operands: embedding rel L2 {rel:.3e}, box-mask IoU min {iou:.5f}, saturated {sam.engine.get_option('saturated')}")
    assert sam.engine.get_option("saturated") == 0 and rel < 3e-2 and iou >= 0.99, (rel, iou)
    sam.engine.close()


def test_checkpoint_argument_round_trip(tmp_path):
    """VERDICT r05 "missing" 5: ``sam_model_registry[...](checkpoint=path)`` -- what the reference's drivers call
    (main_sam_hbox_semantic.py:83-87; build_sam.py:103-106: torch.load + strict load_state_dict).  A reference-shaped state dict saved
    with torch.save and loaded through that argument must give the engine built from the in-memory dict, bit for bit; a file with
    a wrong key, a missing key or a wrong shape must raise like the strict load does (RuntimeError naming the tensor)."""
    import samrs_amd
    cfg = synth.CONFIGS["vit_tiny"]
    sd = synth.make_state_dict(cfg, 3)
    path = tmp_path / "sam_vit_tiny_seed3.pth"
    torch.save(sd, path)
    img = synth.make_image(5)
    embs = []
    for kw in (dict(checkpoint=str(path)), dict(state_dict=sd)):
        sam = samrs_amd.sam_model_registry["vit_tiny"](precision="f16", max_prompts=4, max_points=1, **kw).to(device="cuda")
        pred = samrs_amd.SamPredictor(sam)
        pred.set_image(img)
        embs.append(pred.get_image_embedding().cpu())
        tb = pred.transform.apply_boxes_torch(torch.from_numpy(synth.C1_BOXES).cuda(), img.shape[:2])
        embs.append(pred.predict_torch(None, None, tb, None, multimask_output=False)[2].cpu())
        sam.engine.close()
    assert torch.equal(embs[0], embs[2]) and torch.equal(embs[1], embs[3])
    k = "image_encoder.blocks.0.attn.qkv.weight"
    bad_key = {("module." + n if n == k else n): v for n, v in sd.items()}                  # the classic DataParallel prefix
    missing = {n: v for n, v in sd.items() if n != "mask_decoder.iou_token.weight"}
    bad_shape = dict(sd); bad_shape[k] = sd[k][:, :-1].contiguous()
    for name, blob, what in (("bad_key", bad_key, "unexpected tensor module." + k + "|missing tensor " + k),
                             ("missing", missing, "missing tensor mask_decoder.iou_token.weight"),
                             ("bad_shape", bad_shape, "shape mismatch for " + k)):
        f = tmp_path / (name + ".pth")
        torch.save(blob, f)
        with pytest.raises(RuntimeError, match=what):
            samrs_amd.sam_model_registry["vit_tiny"](checkpoint=str(f), precision="f16", max_prompts=4, max_points=1).to(device="cuda")
