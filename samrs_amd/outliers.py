"""Which K-columns of a checkpoint's block GEMMs are OUTLIERS -- the host-side statement of the rule the engine applies on the device at
``samrs_finalize_weights`` (``samrs_amd/csrc/engine.hip pick_outlier_columns``; option ``"outlier_cols"``), usable without a GPU:

    python -m samrs_amd.outliers sam_vit_h_4b8939.pth [--model vit_h] [--ratio-pct 400]

prints, per encoder block and GEMM, how many columns the engine will give hi + lo terms, the share of the squared-score mass they carry
(> 1/2: the block leaves the MXFP4 route in the v-third modes) and the LayerNorm gammas / row norms behind them.  A real ViT checkpoint has
a few such channels (LayerNorm gammas 10 - 100x the rest, hidden units / v channels that run hot); seeded-normal weights have none.

score_c = (operand-magnitude proxy of column c) x || W[:, c] ||; a column is picked when score_c > ratio x the median score of its GEMM, at
most 32 per GEMM (the largest), ascending:
    qkv / lin1 (A = a LayerNorm output)            |gamma_c| + |beta_c|
    lin2 (A = GELU(lin1))                          || W1[c, :] || rms(gamma2) + |b1_c|
    proj (A = the attention output)                || Wv[c, :] || rms(gamma1) + |bv_c|
``tests/test_outlier_gpu.py`` asserts that the engine picks exactly these columns.
"""
from __future__ import annotations

import sys
from typing import Dict, Tuple

import torch

GEMMS = ("qkv", "lin1", "lin2", "proj")
MAX_COLS = 32


def block_scores(sd, cfg, i: int) -> Dict[str, torch.Tensor]:
    D = cfg.embed_dim
    p = f"image_encoder.blocks.{i}"
    f = lambda k: sd[p + k].detach().to(torch.float32)
    g1, b1, g2, b2 = f(".norm1.weight").abs(), f(".norm1.bias").abs(), f(".norm2.weight").abs(), f(".norm2.bias").abs()
    wqkv, bqkv, w1, bb1, w2, wp = f(".attn.qkv.weight"), f(".attn.qkv.bias"), f(".mlp.lin1.weight"), f(".mlp.lin1.bias"), f(".mlp.lin2.weight"), f(".attn.proj.weight")
    rms1, rms2 = float(g1.square().mean().sqrt()), float(g2.square().mean().sqrt())
    return {"qkv": (g1 + b1) * wqkv.norm(dim=0),
            "lin1": (g2 + b2) * w1.norm(dim=0),
            "lin2": (w1.norm(dim=1) * rms2 + bb1.abs()) * w2.norm(dim=0),
            "proj": (wqkv[2 * D:].norm(dim=1) * rms1 + bqkv[2 * D:].abs()) * wp.norm(dim=0)}


def pick(score: torch.Tensor, ratio: float = 4.0, max_cols: int = MAX_COLS) -> Tuple[torch.Tensor, float]:
    """(ascending column indices, their share of the squared-score mass)"""
    idx = torch.nonzero(score > ratio * score.median()).flatten()
    if len(idx) > max_cols:
        idx = idx[torch.argsort(score[idx], descending=True)[:max_cols]]
    idx = torch.sort(idx).values
    share = float(score[idx].square().sum() / score.square().sum().clamp(min=1e-30)) if len(idx) else 0.0
    return idx, share


def outlier_columns(sd, cfg, ratio: float = 4.0, max_cols: int = MAX_COLS):
    """{(block, gemm): (indices, share)} for every encoder block and gemm in GEMMS."""
    out = {}
    for i in range(cfg.depth):
        for g, s in block_scores(sd, cfg, i).items():
            out[(i, g)] = pick(s, ratio, max_cols)
    return out


def main(argv) -> int:
    import argparse
    from .synth import CONFIGS
    ap = argparse.ArgumentParser(prog="python -m samrs_amd.outliers", description=__doc__.split("\n\n")[0])
    ap.add_argument("checkpoint", help="a SAM state_dict saved with torch.save (sam_vit_h_4b8939.pth ...)")
    ap.add_argument("--model", default="vit_h", choices=sorted(CONFIGS))
    ap.add_argument("--ratio-pct", type=int, default=400, help='the engine option "outlier_ratio_pct"')
    a = ap.parse_args(argv)
    sd = torch.load(a.checkpoint, map_location="cpu")
    cfg = CONFIGS[a.model]
    oc = outlier_columns(sd, cfg, a.ratio_pct / 100.0)
    total = dominant = 0
    print(f"{a.checkpoint}: {a.model}, a column is an outlier above {a.ratio_pct} % of its GEMM's median score")
    print("block  " + "  ".join(f"{g:>14s}" for g in GEMMS) + "   (columns picked / share of the squared-score mass)")
    for i in range(cfg.depth):
        cells = []
        for g in GEMMS:
            idx, share = oc[(i, g)]
            total += len(idx)
            cells.append(f"{len(idx):5d} / {share:6.3f}")
        dom = oc[(i, "qkv")][1] > 0.5 or oc[(i, "proj")][1] > 0.5
        dominant += dom
        print(f"{i:5d}  " + "  ".join(f"{c:>14s}" for c in cells) + ("   outlier-dominated (plain launches + exact lo terms in modes 79 / 207)" if dom else ""))
    print(f"{total} outlier columns in all; {dominant} of {cfg.depth} blocks outlier-dominated.  "
          + ("The engine will run bit-identically to its seeded-normal-weight tests." if total == 0 else
             'The engine carries their hi + lo terms automatically (option "outlier_cols"); python -m oracle.outlier_budget prices them on the CPU.'))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
