"""TEST INFRASTRUCTURE ONLY -- where does the engine's mask error come from?  (CPU, no GPU needed.)

Runs the fp32 oracle on the C2 / C4 fixture inputs (``make_golden.extended_inputs``, realistic-margin weights) once
exactly and then with single ``Rounding`` points (or groups of them) switched to the engine's operand type, and
reports per configuration, against the exact run:

  * C2 (32 hboxes, 20 + 12 chunks): pixels of the painted class map that differ, min per-mask IoU, rel. L2 of the
    low-res logits;
  * C4 (4 rotated boxes, multimask): min IoU over the 12 masks for the enclosing-hbox prompt and for the mask prompt.

    python -m oracle.error_budget vit_b [coarse|dec|enc|plans|all]      (vit_h: ~20 s per encoder pass on 8 cores)
    python -m oracle.error_budget vit_h plans2     floor (only the four block GEMMs in f16), E1 = the engine's default split, ...
    python -m oracle.error_budget vit_h plans3     which block GEMMs must be split as well for the C4 fixture to clear 0.999
    python -m oracle.error_budget vit_h plans4     the v third of the qkv product on its own
    python -m oracle.error_budget vit_h plans10    the lo terms on MXFP4 (e2m1, 32-element scale blocks) operands, per mode
    python -m oracle.error_budget vit_h plans11    lin2's MXFP4 lo terms in the leading k blocks only, on top of split 79

Encoder passes are cached under $SAMRS_EB_CACHE (default /tmp/samrs_error_budget); delete it after changing the oracle.

The table in DESIGN.md section 2 comes from here; ``profiles/r03_error_budget_*.txt`` hold the raw output.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from samrs_amd import synth
from oracle import make_golden, rbox_prompt
from oracle import sam_oracle as so

F16 = torch.float16
CACHE = os.environ.get("SAMRS_EB_CACHE", "/tmp/samrs_error_budget")


def decode_all(orc: so.OraclePredictor, inp, prompts):
    """low-res logits + full-resolution masks for the three fixture workloads."""
    h = w = 1024
    out = {}
    tb = so.apply_boxes(torch.as_tensor(inp["boxes"]), (h, w))
    lows, lgs = [], []
    for s0, s1 in so.box_chunks(32, 20):
        lg, _, low = orc.predict_torch(None, None, tb[s0:s1], None, multimask_output=False, return_logits=True)
        lows.append(low); lgs.append(lg)
    out["c2"] = (torch.cat(lows), torch.cat(lgs) > 0)
    tb = so.apply_boxes(torch.as_tensor(inp["hboxes"]), (h, w))
    lg, _, low = orc.predict_torch(None, None, tb, None, multimask_output=True, return_logits=True)
    out["c4box"] = (low, lg > 0)
    lg, _, low = orc.predict_torch(None, None, None, prompts, multimask_output=True, return_logits=True)
    out["c4mask"] = (low, lg > 0)
    return out


def iou_min(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.flatten(2), b.flatten(2)
    inter = (a & b).sum(-1).double()
    union = (a | b).sum(-1).double().clamp(min=1)
    return float((inter / union).min())


def metrics(ref, got, labels):
    m = {}
    seg0, _ = so.paint_semantic(ref["c2"][1][:, 0].numpy(), labels, (1024, 1024))
    seg1, _ = so.paint_semantic(got["c2"][1][:, 0].numpy(), labels, (1024, 1024))
    m["c2_map_diff"] = int((seg0 != seg1).sum())
    m["c2_iou"] = iou_min(ref["c2"][1], got["c2"][1])
    m["c2_flips"] = int((ref["c2"][1] ^ got["c2"][1]).sum())
    m["c2_relL2"] = float((got["c2"][0] - ref["c2"][0]).norm() / ref["c2"][0].norm())
    for k in ("c4box", "c4mask"):
        m[k + "_iou"] = iou_min(ref[k][1], got[k][1])
        m[k + "_relL2"] = float((got[k][0] - ref[k][0]).norm() / ref[k][0].norm())
    return m


def main(argv):
    name = argv[0] if argv else "vit_b"
    what = argv[1] if len(argv) > 1 else "all"
    torch.set_num_threads(os.cpu_count() or 1)
    os.makedirs(CACHE, exist_ok=True)
    cfg = synth.CONFIGS[name]
    sd = synth.make_state_dict(cfg, 0, logit_scale=synth.MARGIN_LOGIT_SCALE)
    inp = make_golden.extended_inputs()
    img = synth.make_image(0)
    prompts = torch.from_numpy(np.stack([rbox_prompt.rbox_mask_prompt(p.astype(np.int32), 1024, 1024)
                                         for p in inp["polys"]]).astype(np.float32))[:, None]
    x = so.preprocess(img)

    def embedding(tag: str, rd: so.Rounding) -> torch.Tensor:
        """Encoder pass (cached on disk: the decoder sweeps reuse the exact one)."""
        path = os.path.join(CACHE, f"{name}_emb_{tag}.pt")
        if os.path.exists(path):
            return torch.load(path)
        t0 = time.time()
        with torch.no_grad():
            e = so.image_encoder(sd, cfg, x, rd)
        torch.save(e, path)
        print(f"# encoder pass [{tag}] {time.time() - t0:.0f}s", flush=True)
        return e

    def run(enc_tag: str, rd: so.Rounding):
        orc = so.OraclePredictor(sd, cfg, rd)
        orc.features = embedding(enc_tag, rd)
        orc.original_size = orc.input_size = (1024, 1024)
        orc.is_image_set = True
        with torch.no_grad():
            return decode_all(orc, inp, prompts)

    ref = run("exact", so.Rounding())
    hdr = f"{'configuration':58s} {'C2 map diff':>11s} {'C2 flips':>9s} {'C2 IoU min':>10s} {'C2 relL2':>9s} {'c4box IoU':>10s} {'c4mask IoU':>10s} {'c4box L2':>9s}"
    print(f"# {name}: error budget against the exact fp32 oracle (operand type f16)\n{hdr}", flush=True)

    def report(label, enc_tag, rd):
        m = metrics(ref, run(enc_tag, rd), inp["labels"])
        print(f"{label:58s} {m['c2_map_diff']:11d} {m['c2_flips']:9d} {m['c2_iou']:10.5f} {m['c2_relL2']:9.2e} "
              f"{m['c4box_iou']:10.5f} {m['c4mask_iou']:10.5f} {m['c4box_relL2']:9.2e}", flush=True)

    # the fused engine path keeps the GELU output of ConvT #2 in fp32 registers: "dec.prod" is not a rounding point there
    eng_dec = {"dec.prod": None}
    if what in ("coarse", "all"):
        report("engine r02: encoder f16 + decoder f16", "f16", so.Rounding(enc=F16, dec=F16, points=dict(eng_dec)))
        report("encoder f16 only (decoder exact)", "f16", so.Rounding(enc=F16))
        report("decoder f16 only (encoder exact)", "exact", so.Rounding(dec=F16, points=dict(eng_dec)))
    if what in ("dec", "all"):
        for pt in so.DEC_POINTS:
            report(f"only {pt} in f16", "exact", so.Rounding(points={pt: F16}))
    if what in ("enc", "all"):
        for pt in so.ENC_POINTS:
            report(f"only {pt} in f16", "only_" + pt.replace(".", "_"), so.Rounding(points={pt: F16}))
    if what in ("plans2",):
        sp = so.split2(F16)
        cheap = {"enc.patch": sp, "enc.neck0": sp, "enc.neck2": sp, "dec.prod": None}
        report("floor: only the four block GEMMs' operands in f16", "f16_blocks_only",
               so.Rounding(points={"enc.qkv_in": F16, "enc.proj_in": F16, "enc.lin1_in": F16, "enc.lin2_in": F16}))
        report("E1: enc f16, patch/neck split; dec oi/up1/up2 split", "f16_patchnecksplit",
               so.Rounding(enc=F16, dec=F16, points=dict(cheap, **{"dec.oi": sp, "dec.up1": sp, "dec.up2": sp})))
        report("E2: E1 + dec.keys split", "f16_patchnecksplit",
               so.Rounding(enc=F16, dec=F16, points=dict(cheap, **{"dec.oi": sp, "dec.up1": sp, "dec.up2": sp, "dec.keys": sp})))
        report("E3: E2 + proj_in split", "f16_patchneckprojsplit",
               so.Rounding(enc=F16, dec=F16, points=dict(cheap, **{"dec.oi": sp, "dec.up1": sp, "dec.up2": sp, "dec.keys": sp,
                                                                    "enc.proj_in": sp})))
    if what in ("plans3",):
        # which block GEMMs would have to run split as well for the C4 fixture to clear 0.999 at ViT-H (cost: +2x that GEMM's FLOPs)
        sp = so.split2(F16)
        e1 = {"enc.patch": sp, "enc.neck0": sp, "enc.neck2": sp, "dec.prod": None, "dec.oi": sp, "dec.up1": sp, "dec.up2": sp}
        for label, pts in (("E1 + proj + lin2 split", ("enc.proj_in", "enc.lin2_in")),
                           ("E1 + proj + qkv split", ("enc.proj_in", "enc.qkv_in")),
                           ("E1 + proj + lin1 + lin2 split", ("enc.proj_in", "enc.lin1_in", "enc.lin2_in")),
                           ("E1 + all four block GEMMs split", ("enc.proj_in", "enc.qkv_in", "enc.lin1_in", "enc.lin2_in"))):
            report(label, "p3_" + "_".join(q.split(".")[1] for q in pts),
                   so.Rounding(enc=F16, dec=F16, points=dict(e1, **{q: sp for q in pts})))
    if what in ("plans4",):
        # is the v third of the qkv product what makes qkv_in expensive?  (q and k pass through the softmax)
        sp = so.split2(F16)
        e1 = {"enc.patch": sp, "enc.neck0": sp, "enc.neck2": sp, "dec.prod": None, "dec.oi": sp, "dec.up1": sp, "dec.up2": sp}
        report("E1 + proj + v (of qkv) split", "p4_proj_v", so.Rounding(enc=F16, dec=F16, points=dict(e1, **{"enc.proj_in": sp, "enc.v_in": sp})))
        report("E1 + v (of qkv) split only", "p4_v", so.Rounding(enc=F16, dec=F16, points=dict(e1, **{"enc.v_in": sp})))
    if what in ("plans5",):
        # which BLOCKS carry the block-GEMM error: the four GEMMs split in a range of blocks only (cost: 2 extra passes of those blocks)
        sp = so.split2(F16)
        e1 = {"enc.patch": sp, "enc.neck0": sp, "enc.neck2": sp, "dec.prod": None, "dec.oi": sp, "dec.up1": sp, "dec.up2": sp}
        four = ("enc.proj_in", "enc.qkv_in", "enc.lin1_in", "enc.lin2_in")
        n = cfg.depth
        for label, blocks in ((f"E1 + block GEMMs split in blocks {n * 3 // 4}..{n - 1}", range(n * 3 // 4, n)),
                              (f"E1 + block GEMMs split in blocks {n // 2}..{n - 1}", range(n // 2, n)),
                              (f"E1 + block GEMMs split in blocks 0..{n // 4 - 1}", range(0, n // 4)),
                              (f"E1 + block GEMMs split in blocks 0..{n // 2 - 1}", range(0, n // 2)),
                              ("E1 + block GEMMs split in the global-attention blocks", tuple(cfg.global_attn_indexes))):
            tag = "p5_" + "_".join(str(b) for b in (blocks[0], blocks[-1], len(blocks)))
            report(label, tag, so.Rounding(enc=F16, dec=F16, points=dict(e1), block_points={q: (blocks, sp) for q in four}))
    if what in ("plans6",):
        # attention-side split (qkv or only its v third, + proj) confined to the leading blocks: what does the 1x rate in the late blocks cost?
        sp = so.split2(F16)
        e1 = {"enc.patch": sp, "enc.neck0": sp, "enc.neck2": sp, "dec.prod": None, "dec.oi": sp, "dec.up1": sp, "dec.up2": sp}
        n = cfg.depth
        for label, pts, blocks in (("E1 + qkv + proj split in blocks 0..%d" % (n // 2 - 1), ("enc.qkv_in", "enc.proj_in"), range(0, n // 2)),
                                   ("E1 + v + proj split in blocks 0..%d" % (n // 2 - 1), ("enc.v_in", "enc.proj_in"), range(0, n // 2)),
                                   ("E1 + qkv + proj split in blocks 0..%d" % (3 * n // 4 - 1), ("enc.qkv_in", "enc.proj_in"), range(0, 3 * n // 4)),
                                   ("E1 + v + proj split in blocks 0..%d" % (3 * n // 4 - 1), ("enc.v_in", "enc.proj_in"), range(0, 3 * n // 4))):
            tag = "p6_" + "_".join(q.split(".")[1] for q in pts) + "_%d" % len(blocks)
            report(label, tag, so.Rounding(enc=F16, dec=F16, points=dict(e1), block_points={q: (blocks, sp) for q in pts}))
    if what in ("plans7",):
        # around the ViT-H default (v + proj, blocks 0..23): is either half or a shorter prefix enough?
        sp = so.split2(F16)
        e1 = {"enc.patch": sp, "enc.neck0": sp, "enc.neck2": sp, "dec.prod": None, "dec.oi": sp, "dec.up1": sp, "dec.up2": sp}
        n = cfg.depth
        for label, pts, blocks in (("E1 + v + proj split in blocks 0..%d" % (5 * n // 8 - 1), ("enc.v_in", "enc.proj_in"), range(0, 5 * n // 8)),
                                   ("E1 + v split in blocks 0..%d" % (3 * n // 4 - 1), ("enc.v_in",), range(0, 3 * n // 4)),
                                   ("E1 + proj split in blocks 0..%d" % (3 * n // 4 - 1), ("enc.proj_in",), range(0, 3 * n // 4)),
                                   ("E1 + v + proj split in blocks 0..%d, lin2 in 0..%d" % (3 * n // 4 - 1, n // 4 - 1), ("enc.v_in", "enc.proj_in", "enc.lin2_in"), None)):
            tag = "p7_" + "_".join(q.split(".")[1] for q in pts) + "_%s" % (len(blocks) if blocks is not None else "mix")
            if blocks is None:
                bp = {"enc.v_in": (range(0, 3 * n // 4), sp), "enc.proj_in": (range(0, 3 * n // 4), sp), "enc.lin2_in": (range(0, n // 4), sp)}
            else:
                bp = {q: (blocks, sp) for q in pts}
            report(label, tag, so.Rounding(enc=F16, dec=F16, points=dict(e1), block_points=bp))
    if what in ("plans8",):
        # the ViT-H default (v + proj, blocks 0..23) with the two correction terms on fp8 operands (block-scaled fp8 MFMA: 2x the f16 rate)
        sp, sp8 = so.split2(F16), so.split_fp8_lo(F16)
        e1 = {"enc.patch": sp, "enc.neck0": sp, "enc.neck2": sp, "dec.prod": None, "dec.oi": sp, "dec.up1": sp, "dec.up2": sp}
        n = cfg.depth
        blocks = range(0, 3 * n // 4)
        report("E1 + v + proj split in blocks 0..%d, lo terms exact (= the default)" % (3 * n // 4 - 1), "p6_v_in_proj_in_%d" % len(blocks),
               so.Rounding(enc=F16, dec=F16, points=dict(e1), block_points={q: (blocks, sp) for q in ("enc.v_in", "enc.proj_in")}))
        report("  ... lo terms on fp8 (e4m3) operands", "p8_fp8lo_%d" % len(blocks),
               so.Rounding(enc=F16, dec=F16, points=dict(e1), block_points={q: (blocks, sp8) for q in ("enc.v_in", "enc.proj_in")}))
        report("all four block GEMMs in every block, lo terms on fp8 operands", "p8_fp8lo_all4",
               so.Rounding(enc=F16, dec=F16, points=dict(e1, **{q: sp8 for q in ("enc.qkv_in", "enc.proj_in", "enc.lin1_in", "enc.lin2_in")})))
    if what in ("plans9",):
        # the same with the lo terms on fp6 / fp4 operands (block-scaled f8f6f4 MFMA: 4x the f16 rate), MX blocks of 32
        sp = so.split2(F16)
        e1 = {"enc.patch": sp, "enc.neck0": sp, "enc.neck2": sp, "dec.prod": None, "dec.oi": sp, "dec.up1": sp, "dec.up2": sp}
        four = ("enc.qkv_in", "enc.proj_in", "enc.lin1_in", "enc.lin2_in")
        for fmt in ("e2m3", "e2m1"):
            q = so.split_fp8_lo(F16, fmt=fmt, block=32)
            report(f"all four block GEMMs in every block, lo terms on {fmt} operands", f"p9_{fmt}_all4",
                   so.Rounding(enc=F16, dec=F16, points=dict(e1, **{k: q for k in four})))
    if what in ("plans10",):
        # round 4: the lo terms on MXFP4 operands (e2m1, one E8M0 scale per 32 k: the format whose LDS stage has the geometry of
        # the f16 stage, gemm.hip) in the modes the engine offers -- which blocks / GEMMs have to take them for the C4 floor?
        sp = so.split2(F16)
        q4 = so.split_fp8_lo(F16, fmt="e2m1", block=32)
        e1 = {"enc.patch": sp, "enc.neck0": sp, "enc.neck2": sp, "dec.prod": None, "dec.oi": sp, "dec.up1": sp, "dec.up2": sp}
        n = cfg.depth
        for label, pts, blocks in (("E1 + v + proj, blocks 0..%d, lo terms e2m1 / 32" % (3 * n // 4 - 1), ("enc.v_in", "enc.proj_in"), range(0, 3 * n // 4)),
                                   ("E1 + v + proj, every block, lo terms e2m1 / 32", ("enc.v_in", "enc.proj_in"), range(0, n)),
                                   ("E1 + qkv + proj, every block, lo terms e2m1 / 32", ("enc.qkv_in", "enc.proj_in"), range(0, n)),
                                   ("E1 + v + proj + lin2, every block, lo terms e2m1 / 32", ("enc.v_in", "enc.proj_in", "enc.lin2_in"), range(0, n))):
            tag = "p10_e2m1_" + "_".join(k.split(".")[1] for k in pts) + "_%d" % len(blocks)
            report(label, tag, so.Rounding(enc=F16, dec=F16, points=dict(e1), block_points={k: (blocks, q4) for k in pts}))
    if what in ("plans11",):
        # round 4, after the 576-mask sample (profiles/r04_parity_stats.md): is there a mode between split 79 (v + proj, blocks
        # 0..23) and 207 (+ lin2 in every block)?  lin2's lo terms in the LEADING k blocks only, on top of 79
        sp = so.split2(F16)
        q4 = so.split_fp8_lo(F16, fmt="e2m1", block=32)
        e1 = {"enc.patch": sp, "enc.neck0": sp, "enc.neck2": sp, "dec.prod": None, "dec.oi": sp, "dec.up1": sp, "dec.up2": sp}
        n = cfg.depth
        att = range(0, 3 * n // 4)
        for k in (n // 4, n // 2, 3 * n // 4):
            bp = {"enc.v_in": (att, q4), "enc.proj_in": (att, q4), "enc.lin2_in": (range(0, k), q4)}
            report("E1 + v + proj (blocks 0..%d) + lin2 (blocks 0..%d), e2m1 / 32" % (3 * n // 4 - 1, k - 1), "p11_lin2_%d" % k,
                   so.Rounding(enc=F16, dec=F16, points=dict(e1), block_points=bp))
        # ... or in the TRAILING blocks (lin2's output goes straight into the residual stream the neck reads)
        for k0 in (3 * n // 4, n // 2, 0):
            bp = {"enc.v_in": (att, q4), "enc.proj_in": (att, q4), "enc.lin2_in": (range(k0, n), q4)}
            report("E1 + v + proj (blocks 0..%d) + lin2 (blocks %d..%d), e2m1 / 32" % (3 * n // 4 - 1, k0, n - 1), "p11_lin2_from%d" % k0,
                   so.Rounding(enc=F16, dec=F16, points=dict(e1), block_points=bp))
    if what in ("plans", "all"):
        sp = so.split2(F16)
        report("plan A: encoder f16, decoder operands split f16 (hi+lo)", "f16",
               so.Rounding(enc=F16, dec=sp, points=dict(eng_dec)))
        report("plan B: A + K/V/Q storage, OI, U1 kept f16", "f16",
               so.Rounding(enc=F16, dec=sp, points={"dec.prod": None, "dec.kvq_out": F16, "dec.oi": F16, "dec.up2": F16}))
        report("plan C: A + neck in split f16", "f16_necksplit",
               so.Rounding(enc=F16, dec=sp, points={"dec.prod": None, "enc.neck0": sp, "enc.neck2": sp}))
        report("plan D: C + lin2 operand (MLP hidden) split", "f16_necklin2split",
               so.Rounding(enc=F16, dec=sp, points={"dec.prod": None, "enc.neck0": sp, "enc.neck2": sp, "enc.lin2_in": sp}))


if __name__ == "__main__":
    main(sys.argv[1:])
