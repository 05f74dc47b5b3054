"""GPU: parity on checkpoint-like weights (VERDICT r05 "missing" 2, "next round" 1).

Every other parity test runs on seeded-normal weights.  A real ViT checkpoint has outlier channels -- a few LayerNorm gammas
10 - 100x the rest, MLP hidden units and v channels that run thousands of times hotter -- and an f16 operand's rounding error is
relative to ITS magnitude, so those few K-columns carry most of the operand error of a block GEMM.  ``synth.heavy_tailed`` plants
that structure; the engine picks the columns from the weights at load time (engine.hip pick_outlier_columns; option
"outlier_cols") and carries their hi + lo split as 64 more K columns of the SAME qkv / lin1 launch (the LayerNorm writes the
operand side).  The fp32 oracle evaluates the same state dict; ``oracle/outlier_budget.py`` is the CPU emulation that priced it.
"""
import json
import os
import time

import numpy as np
import pytest
import torch

from samrs_amd import synth

pytestmark = pytest.mark.gpu

HEAVY = dict(hidden_scale=3e3, v_scale=3e3, gamma_scale=30.0)


def _iou(m, m0):
    m, m0 = m.flatten(1), m0.flatten(1)
    return ((m & m0).sum(1).double() / (m | m0).sum(1).double().clamp(min=1))


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_outlier_columns_are_picked_from_the_weights_and_cut_the_error(precision):
    """vit_tiny / vit_tiny1280 (the padded-stride route needs width 1280), heavy-tailed weights, a batch of 4 tiles so that the 1280-wide
    model runs the persistent 256 x 320 kernels: (1) the load-time rule finds exactly the planted columns -- 4 per block GEMM and
    block, none on the seeded-normal weights; (2) the residual stream after every block is closer to the oracle's with the
    extension on than off; (3) off is bit-identical with an engine that never picked anything (the extension is the only change)."""
    import samrs_amd
    from oracle import sam_oracle as so
    for name, n_img in (("vit_tiny", 1), ("vit_tiny1280", 4), ("vit_tiny1280", 1)):
        cfg = synth.CONFIGS[name]
        base = synth.make_state_dict(cfg, 0)
        sd = synth.heavy_tailed(base, cfg, 0, **HEAVY)
        sam0 = samrs_amd.sam_model_registry[name](state_dict=base, precision=precision, max_prompts=8, max_points=1, max_images=n_img).to("cuda")
        assert sam0.engine.get_option("outlier_blocks") == 0 and sam0.engine.get_option("outlier_columns") == 0
        sam0.engine.close()
        sam = samrs_amd.sam_model_registry[name](state_dict=sd, precision=precision, max_prompts=8, max_points=1, max_images=n_img,
                                                 options={"split": 15}).to("cuda")
        eng = sam.engine
        assert eng.get_option("outlier_cols") == 7                                # bit 0: qkv / lin1, bit 1: lin2, bit 2: proj
        assert eng.get_option("outlier_blocks") == cfg.depth                      # heavy_tailed's default blocks: first, middle, last = both
        assert eng.get_option("outlier_columns") == 4 * 4 * cfg.depth             # 4 planted columns x 4 block GEMMs x blocks
        # ... and they are the SAME columns the CPU restatement of the rule picks (oracle/outlier_budget.py: what lets that tool price the
        # engine's behaviour on a new checkpoint before a GPU sees it)
        from oracle.outlier_budget import outlier_columns
        want = outlier_columns(sd, cfg)
        for blk in range(cfg.depth):
            for gi, point in enumerate(("enc.qkv_in", "enc.lin1_in", "enc.lin2_in", "enc.proj_in")):
                assert eng.outlier_columns(blk, gi) == want[(blk, point)].tolist(), (name, blk, point)
        from samrs_amd import outliers
        host = outliers.outlier_columns(sd, cfg)
        assert eng.get_option("outlier_dominant_blocks") == sum(host[(b_, "qkv")][1] > 0.5 or host[(b_, "proj")][1] > 0.5 for b_ in range(cfg.depth))
        never = samrs_amd.sam_model_registry[name](state_dict=sd, precision=precision, max_prompts=8, max_points=1, max_images=n_img,
                                                   options={"split": 15, "outlier_cols": 0}).to("cuda")
        assert never.engine.get_option("outlier_columns") == 0
        imgs = [synth.make_image(i) for i in range(n_img)]
        t = torch.as_tensor(np.stack(imgs), device="cuda").contiguous()
        taps = {}
        with torch.no_grad():
            so.image_encoder(sd, cfg, so.preprocess(imgs[0]), taps=taps)
        for nb in range(1, cfg.depth + 1):
            ref = taps[f"block{nb - 1}"][0]
            rel = {}
            for mask in (7, 1, 0):
                eng.set_option("outlier_cols", mask)
                x = eng.debug_encoder_prefix(t, nb).cpu()
                rel[mask] = ((x[0] - ref).norm() / ref.norm()).item()
                if mask == 0:
                    assert torch.equal(x, never.engine.debug_encoder_prefix(t, nb).cpu()), (name, nb)
            print(f"{name} {precision} x{n_img} after {nb} blocks, heavy-tailed weights: residual-stream rel L2 vs oracle {rel[0]:.3e} (off) -> "
                  f"{rel[1]:.3e} (qkv / lin1 columns) -> {rel[7]:.3e} (+ lin2 / proj columns)")
            # (bf16: 8 significand bits everywhere, so the non-outlier columns' rounding is a larger share of the error than at f16)
            assert rel[1] < (0.75 if precision == "f16" else 0.9) * rel[0] and rel[7] < rel[1], (name, precision, nb, rel)
        eng.set_option("outlier_cols", 7)
        eng.close()
        never.engine.close()


def _c2_compare(pred, orc_masks, orc_low, img, boxes, batch=None):
    """batch = None: SamPredictor.set_image (one tile: M = 4096 rows, the generic GEMM routes).  batch = u8 [8, 1024, 1024, 3] with
    the fixture tile at index 0: one 8-tile encoder pass (the padded-stride qkv / lin1 launches, proj / lin2 on the EXT stage of the
    256 x 320 kernel), decoded from slot 0."""
    tb = pred.transform.apply_boxes_torch(torch.from_numpy(boxes).cuda(), img.shape[:2])
    if batch is None:
        pred.set_image(img)
        m, _, low = pred.predict_torch(None, None, tb, None, multimask_output=False)
    else:
        eng = pred.model.engine
        eng.set_images(batch)
        m, _, low = eng.predict(0, tb, None, None, None, False, False, (1024, 1024), (1024, 1024))
    iou = _iou(m.cpu(), orc_masks)
    rel = ((low.cpu() - orc_low).norm() / orc_low.norm()).item()
    return float(iou.min()), float(iou.mean()), rel


@pytest.mark.parametrize("every", [False, True])
def test_heavy_tailed_vit_h_holds_the_iou_bar_in_modes_15_and_79(every):
    """ViT-H, ``synth.heavy_tailed`` (outlier channels in the first / middle / last block -- or, harsher, in EVERY block), the C2
    fixture inputs (make_golden.extended_inputs: tile 0, 32 hboxes, 20 + 12 chunks on the oracle side): per-mask IoU >= 0.999 against the
    fp32 oracle on the SAME weights, in the 1x-rate mode (15) and the ViT-H default (79), with the outlier-column extension on;
    the same with it off is measured and reported beside it.  Also the cost: an 8-tile encoder pass with and without it."""
    import samrs_amd
    from oracle import make_golden
    from oracle import sam_oracle as so
    cfg = synth.CONFIGS["vit_h"]
    base = synth.make_state_dict(cfg, 0, logit_scale=synth.MARGIN_LOGIT_SCALE)
    sd = synth.heavy_tailed(base, cfg, 0, blocks=list(range(cfg.depth)) if every else None, **HEAVY)
    inp = make_golden.extended_inputs()
    img = synth.make_image(inp["image_index"])
    orc = so.OraclePredictor(sd, cfg)
    t0 = time.time()
    orc.set_image(img)
    tb = so.apply_boxes(torch.as_tensor(inp["boxes"]), (1024, 1024))
    ms, lows = [], []
    for s0, s1 in so.box_chunks(32, 20):
        m, _, low = orc.predict_torch(None, None, tb[s0:s1], None, multimask_output=False)
        ms.append(m); lows.append(low)
    m0, low0 = torch.cat(ms), torch.cat(lows)
    # C4 (BASELINE configs[3]): the fixture's four rotated boxes -> enclosing hbox / +-1000 mask prompt, multimask_output=True (12 + 12 masks)
    from oracle import rbox_prompt
    from samrs_amd import transforms
    hb = so.apply_boxes(torch.as_tensor(inp["hboxes"]), (1024, 1024))
    c4box0 = orc.predict_torch(None, None, hb, None, multimask_output=True)[0]
    pr0 = torch.from_numpy(np.stack([rbox_prompt.rbox_mask_prompt(p.astype(np.int32), 1024, 1024) for p in inp["polys"]]).astype(np.float32))[:, None]
    c4mask0 = orc.predict_torch(None, None, None, pr0, multimask_output=True)[0]
    print(f"oracle on heavy-tailed ViT-H weights: {time.time() - t0:.0f} s")
    sam = samrs_amd.sam_model_registry["vit_h"](state_dict=sd, precision="f16", max_prompts=32, max_points=1, max_images=8).to("cuda")
    eng = sam.engine
    n_blocks = cfg.depth if every else 3
    assert eng.get_option("split") == 79 and eng.get_option("outlier_blocks") == n_blocks
    assert eng.get_option("outlier_columns") == 16 * n_blocks
    pred = samrs_amd.SamPredictor(sam)
    out = {"weights": "synth.heavy_tailed(hidden 3e3, v 3e3, gamma 30), " + ("every block" if every else "blocks 0 / 16 / 31"),
           "n_masks": 32, "outlier_blocks": n_blocks}
    tiles = torch.as_tensor(np.stack([img] + [synth.make_noise_image(i) for i in range(7)]), device="cuda").contiguous()
    for mode in (15, 79):
        eng.set_option("split", mode)
        for tag, mask in (("off", 0), ("qkv_lin1", 1), ("on", 7)):
            eng.set_option("outlier_cols", mask)
            imin, imean, rel = _c2_compare(pred, m0, low0, img, inp["boxes"], batch=tiles)
            out[f"mode{mode}_{tag}"] = {"iou_min": imin, "iou_mean": imean, "low_res_rel_l2": rel}
            print(f"heavy-tailed ViT-H ({'every block' if every else '3 blocks'}), 8-tile pass, mode {mode}, outlier columns {tag:8s}: "
                  f"C2 IoU min {imin:.5f} mean {imean:.5f}, low-res rel L2 {rel:.2e}")
        # one tile at a time (SamPredictor.set_image): other GEMM routes, the same claim
        imin, imean, rel = _c2_compare(pred, m0, low0, img, inp["boxes"])
        out[f"mode{mode}_on_single_tile"] = {"iou_min": imin, "iou_mean": imean, "low_res_rel_l2": rel}
        print(f"   ... one tile (set_image), mode {mode}, outlier columns on: C2 IoU min {imin:.5f}, low-res rel L2 {rel:.2e}")
        assert imin >= 0.999 and rel < 1.25 * out[f"mode{mode}_on"]["low_res_rel_l2"], (mode, imin, rel)
        # the multimask outputs (three smaller masks per object: the thin side of the parity statement) on the same embedding
        eng.set_option("allow_reduced", 1)
        for tag, mask in (("off", 0), ("on", 7)):
            eng.set_option("outlier_cols", mask)
            pred.set_image(img)
            tb4 = pred.transform.apply_boxes_torch(torch.from_numpy(inp["hboxes"]).cuda(), img.shape[:2])
            mb = pred.predict_torch(None, None, tb4, None, multimask_output=True)[0].cpu()
            pm = transforms.rbox_mask_prompts(inp["polys"], (1024, 1024), fill_rule="cv2_le_451")[:, None]
            mm = pred.predict_torch(None, None, None, pm, multimask_output=True)[0].cpu()
            i_box, i_mask = float(_iou(mb.flatten(0, 1), c4box0.flatten(0, 1)).min()), float(_iou(mm.flatten(0, 1), c4mask0.flatten(0, 1)).min())
            out[f"mode{mode}_{tag}_c4"] = {"hbox_iou_min": i_box, "mask_prompt_iou_min": i_mask, "n_masks": 24}
            print(f"   ... C4 multimask (12 + 12 masks), mode {mode}, outlier columns {tag:3s}: IoU min hbox {i_box:.5f} / mask prompt {i_mask:.5f}")
        eng.set_option("outlier_cols", 7)
    # cost of the extension: 8-tile encoder passes in mode 15, alternated
    eng.set_option("split", 15)

    def enc_ms(on, reps=6):
        eng.set_option("outlier_cols", on)
        eng.set_images(tiles)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            eng.set_images(tiles)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    t_off, t_on = [], []
    for _ in range(3):
        t_off.append(enc_ms(0)); t_on.append(enc_ms(7))
    out["encoder_ms_8_tiles"] = {"off": min(t_off), "on": min(t_on), "cost": min(t_on) / min(t_off) - 1.0}
    print(f"8-tile encoder pass, mode 15: {min(t_off):.2f} ms off -> {min(t_on):.2f} ms with the outlier columns "
          f"({100 * (min(t_on) / min(t_off) - 1):+.1f} %; {n_blocks} of 32 blocks carry them)")
    eng.set_option("outlier_cols", 7)
    if os.path.isdir("gpurun_out"):
        path = "gpurun_out/heavy_tailed_parity.json"
        blob = json.load(open(path)) if os.path.exists(path) else {}
        import bench
        blob["csrc_sha16"] = bench.csrc_sha()
        blob["every_block" if every else "three_blocks"] = out
        json.dump(blob, open(path, "w"), indent=1)
    for mode in (15, 79):
        assert out[f"mode{mode}_on"]["iou_min"] >= 0.999, out
        assert out[f"mode{mode}_on"]["low_res_rel_l2"] < out[f"mode{mode}_off"]["low_res_rel_l2"], out
    # <= 3 % step cost on synth.heavy_tailed as it is (three blocks); with outlier columns in EVERY block of all
    # four GEMMs (measured + 8.9 %: lin1 leaves the four-wave kernel for K = 1344, + 5 % K on qkv / lin1, the attention kernel's lo
    # output, the side GEMM, the gather: ~130 us per block) it is bounded and reported
    assert out["encoder_ms_8_tiles"]["cost"] < (0.12 if every else 0.03), out
    eng.close()


def test_heavy_tailed_statistical_sample():
    """The parity sample of oracle/parity_sample.py (what the N(0, sigma) claims rest on) on CHECKPOINT-LIKE weights: synth.heavy_tailed with
    outlier channels in every block (the harsher variant), 4 tiles x 32 hboxes (128 single masks, 4 painted class maps) and 4 x 8
    FAIR1M-shaped rboxes (96 + 96 multimask masks, the three scripted instance recipes), engine (outlier columns on) against the fp32 oracle on
    the SAME weights, modes 15 and 79.  Asserted: every single-mask workload >= 0.9995 (the mask-only recipe >= 0.999) in both modes; the
    multimask outputs >= 0.999 on all 96 + 96 masks in BOTH modes -- on these weights the 1x-rate mode clears the bar the N(0, sigma) weights
    need mode 79 for, because the error sits in columns the extension reaches, and mode 79 hands its outlier-dominated blocks to the same
    exact lo terms (before that rule: 0.99848, 16 masks under 0.999); zero flips outside the tau band."""
    import samrs_amd
    from oracle import parity_sample as ps
    from oracle import sam_oracle as so
    cfg = synth.CONFIGS["vit_h"]
    base = synth.make_state_dict(cfg, 0, logit_scale=synth.MARGIN_LOGIT_SCALE)
    sd = synth.heavy_tailed(base, cfg, 0, blocks=list(range(cfg.depth)), **HEAVY)
    sam = samrs_amd.sam_model_registry["vit_h"](state_dict=sd, precision="f16", max_prompts=32, max_points=1).to("cuda")
    eng = sam.engine
    assert eng.get_option("outlier_cols") == 7 and eng.get_option("outlier_blocks") == cfg.depth
    # every block's qkv / proj operand error is dominated by its outlier columns: in mode 79 these blocks take the exact lo terms of those
    # columns (plain launches + the extension) instead of the MXFP4 lo terms of all columns (engine.hip oc_dominant)
    assert eng.get_option("outlier_dominant_blocks") == cfg.depth
    eng.set_option("allow_reduced", 1)
    pred = samrs_amd.SamPredictor(sam)
    rec = ps.run(pred, so.OraclePredictor(sd, cfg), [15, 79], tile_iter=ps.tiles(n_c2=4, n_c4=4, odd=False, long_tail=False))
    summ = ps.summarise(rec)
    print(ps.table(summ))
    if os.path.isdir("gpurun_out"):
        path = "gpurun_out/heavy_tailed_parity.json"
        blob = json.load(open(path)) if os.path.exists(path) else {}
        import bench
        blob["csrc_sha16"] = bench.csrc_sha()
        blob["sample_every_block"] = {str(k): v for k, v in summ.items()}
        json.dump(blob, open(path, "w"), indent=1)
    for mode, tags in summ.items():
        for tag, s in tags.items():
            assert s["flips_outside_tau"] == 0 and s.get("classmap_diff_outside_unstable", 0) == 0, (mode, tag)
            if tag in ("c2", "inst_point", "inst_rhbox"):
                assert s["iou_min"] >= 0.9995, (mode, tag, s["iou_min"])
            if tag == "inst_mask":                      # the mask-only recipe: the weakest single-mask workload on the seeded-normal weights too (floor 0.999 there)
                assert s["iou_min"] >= 0.999, (mode, tag, s["iou_min"])
    assert summ[15]["c2"]["n_masks"] == 128 and summ[15]["c4box"]["n_masks"] == 96
    for mode in (15, 79):
        for tag in ("c4box", "c4mask"):
            assert summ[mode][tag]["iou_min"] >= 0.999 and summ[mode][tag]["n_below_0999"] == 0, (mode, tag, summ[mode][tag])
    eng.close()


def test_more_outlier_candidates_than_a_stage_holds():
    """40 LayerNorm gammas x 30 in one block of the two-block model: 40 candidate columns for qkv / lin1, one 64-wide K stage carries 32 -- the
    engine keeps the 32 largest scores (the same ones the host rule keeps) and still improves on the untreated arithmetic."""
    import samrs_amd
    from oracle import sam_oracle as so
    from samrs_amd import outliers
    cfg = synth.CONFIGS["vit_tiny"]
    sd = synth.heavy_tailed(synth.make_state_dict(cfg, 0), cfg, 0, gamma_scale=30.0, n_channels=40, blocks=[0])
    sam = samrs_amd.sam_model_registry["vit_tiny"](state_dict=sd, precision="f16", max_prompts=8, max_points=1, options={"split": 15}).to("cuda")
    eng = sam.engine
    host = outliers.outlier_columns(sd, cfg)
    assert len(host[(0, "qkv")][0]) == 32
    for gi, g in enumerate(outliers.GEMMS):
        assert eng.outlier_columns(0, gi) == host[(0, g)][0].tolist(), g
    img = synth.make_image(0)
    taps = {}
    with torch.no_grad():
        so.image_encoder(sd, cfg, so.preprocess(img), taps=taps)
    t = torch.as_tensor(img, device="cuda")[None].contiguous()
    rel = {}
    for mask in (7, 0):
        eng.set_option("outlier_cols", mask)
        x = eng.debug_encoder_prefix(t, cfg.depth).cpu()[0]
        rel[mask] = ((x - taps[f"block{cfg.depth - 1}"][0]).norm() / taps[f"block{cfg.depth - 1}"][0].norm()).item()
    print(f"40 hot LayerNorm channels, 32 carried: residual stream rel L2 vs oracle {rel[0]:.3e} -> {rel[7]:.3e}")
    assert rel[7] < 0.8 * rel[0], rel
    eng.close()
