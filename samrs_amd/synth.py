"""Seeded synthetic weights / images / boxes for the SAM box->mask hot path.

There is no SAM checkpoint and no dataset on the build or GPU machines, so every test,
the smoke run and ``bench.py`` use inputs generated here.  The generators are pure
torch-CPU / numpy so that the same seed gives bit-identical tensors in the authoring
container (where golden vectors are produced with the real reference) and on the GPU box.

Shapes follow the reference's ``state_dict`` contract:
  image encoder  : Generate Dataset/segment_anything/modeling/image_encoder.py:17-104
  prompt encoder : Generate Dataset/segment_anything/modeling/prompt_encoder.py:16-60,176-188
  mask decoder   : Generate Dataset/segment_anything/modeling/mask_decoder.py:16-69
  two-way xfmr   : Generate Dataset/segment_anything/modeling/transformer.py:16-60,109-149,185-216
  registry       : Generate Dataset/segment_anything/build_sam.py:14-107
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch


@dataclass(frozen=True)
class SamConfig:
    """Hyper-parameters of one SAM variant (build_sam.py:14-21,37-44,55-98)."""

    name: str
    embed_dim: int
    depth: int
    num_heads: int
    global_attn_indexes: Tuple[int, ...]
    img_size: int = 1024
    patch_size: int = 16
    window_size: int = 14
    out_chans: int = 256          # == prompt_embed_dim == decoder transformer_dim
    mlp_ratio: int = 4
    mask_in_chans: int = 16
    dec_depth: int = 2
    dec_heads: int = 8
    dec_mlp_dim: int = 2048
    num_mask_tokens: int = 4      # 3 multimask outputs + 1
    iou_hidden: int = 256

    @property
    def grid(self) -> int:        # 64 tokens per side
        return self.img_size // self.patch_size

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads


CONFIGS: Dict[str, SamConfig] = {
    "vit_h": SamConfig("vit_h", 1280, 32, 16, (7, 15, 23, 31)),
    "vit_l": SamConfig("vit_l", 1024, 24, 16, (5, 11, 17, 23)),
    "vit_b": SamConfig("vit_b", 768, 12, 12, (2, 5, 8, 11)),
    # Not in the reference registry: a 2-block encoder with the same 1024^2 / 64x64 geometry,
    # small enough that the CPU oracle finishes in ~1 s.  Used by fast parity tests and smoke().
    # Block 0 is windowed, block 1 is global, so both attention kernels are exercised.
    "vit_tiny": SamConfig("vit_tiny", 128, 2, 2, (1,)),
    # head_dim 80 like ViT-H (the awkward MFMA K size); width 640 keeps every GEMM dim a multiple of 128.
    "vit_tiny80": SamConfig("vit_tiny80", 640, 2, 8, (1,)),
    # ViT-H's width (1280 = the embed_dim whose LayerNorms the engine folds into the qkv / lin1 GEMMs), head_dim 80, two blocks.
    "vit_tiny1280": SamConfig("vit_tiny1280", 1280, 2, 16, (1,)),
}


def _uniform(gen: torch.Generator, shape, bound: float) -> torch.Tensor:
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * bound


def _linear(sd, gen, prefix: str, out_f: int, in_f: int, bias: bool = True) -> None:
    bound = 1.0 / math.sqrt(in_f)
    sd[prefix + ".weight"] = _uniform(gen, (out_f, in_f), bound)
    if bias:
        sd[prefix + ".bias"] = _uniform(gen, (out_f,), bound)


def _norm(sd, gen, prefix: str, n: int) -> None:
    # LayerNorm affine: near (1, 0) but not exactly, so gamma/beta paths are exercised.
    sd[prefix + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=gen)
    sd[prefix + ".bias"] = 0.05 * torch.randn(n, generator=gen)


def _dec_attention(sd, gen, prefix: str, dim: int, internal: int) -> None:
    _linear(sd, gen, prefix + ".q_proj", internal, dim)
    _linear(sd, gen, prefix + ".k_proj", internal, dim)
    _linear(sd, gen, prefix + ".v_proj", internal, dim)
    _linear(sd, gen, prefix + ".out_proj", dim, internal)


# "Realistic-margin" weights: the last layer of every hypernetwork MLP scaled so that the
# low-res logits have a checkpoint-like spread (std ~ 4 instead of ~ 0.1).  A power of two, so every logit is
# EXACTLY 32 x the unscaled one in fp32 and in the engine alike.  NB: a positive scale cannot move a zero
# crossing -- mask = (logit > 0) -- so this mode changes magnitudes (f16 range, IoU-head inputs, thresholds
# written in absolute units), not which pixels sit next to the threshold; DESIGN.md 2 has the arithmetic.
MARGIN_LOGIT_SCALE = 32.0


def make_state_dict(cfg: SamConfig, seed: int = 0, logit_scale: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    """A full SAM ``state_dict`` (fp32, CPU) with the reference's key names and shapes.

    Every tensor is random, including the ones the reference zero-initialises
    (``pos_embed``, ``rel_pos_h/w`` -- image_encoder.py:68-70,221-222) so that the rel-pos and
    abs-pos paths are never tested against zeros.
    ``logit_scale`` multiplies the output layer of the four hypernetwork MLPs (see MARGIN_LOGIT_SCALE).
    """
    gen = torch.Generator().manual_seed(1234567 + seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    D, g, hd = cfg.embed_dim, cfg.grid, cfg.head_dim
    P, C = cfg.patch_size, cfg.out_chans

    # ---- image encoder -------------------------------------------------------------
    sd["image_encoder.pos_embed"] = 0.02 * torch.randn(1, g, g, D, generator=gen)
    fan = 3 * P * P
    sd["image_encoder.patch_embed.proj.weight"] = _uniform(gen, (D, 3, P, P), 1.0 / math.sqrt(fan))
    sd["image_encoder.patch_embed.proj.bias"] = _uniform(gen, (D,), 1.0 / math.sqrt(fan))
    for i in range(cfg.depth):
        p = f"image_encoder.blocks.{i}"
        s = g if i in cfg.global_attn_indexes else cfg.window_size
        _norm(sd, gen, p + ".norm1", D)
        sd[p + ".attn.rel_pos_h"] = 0.02 * torch.randn(2 * s - 1, hd, generator=gen)
        sd[p + ".attn.rel_pos_w"] = 0.02 * torch.randn(2 * s - 1, hd, generator=gen)
        _linear(sd, gen, p + ".attn.qkv", 3 * D, D)
        _linear(sd, gen, p + ".attn.proj", D, D)
        _norm(sd, gen, p + ".norm2", D)
        _linear(sd, gen, p + ".mlp.lin1", cfg.mlp_ratio * D, D)
        _linear(sd, gen, p + ".mlp.lin2", D, cfg.mlp_ratio * D)
    sd["image_encoder.neck.0.weight"] = _uniform(gen, (C, D, 1, 1), 1.0 / math.sqrt(D))
    _norm(sd, gen, "image_encoder.neck.1", C)
    sd["image_encoder.neck.2.weight"] = _uniform(gen, (C, C, 3, 3), 1.0 / math.sqrt(9 * C))
    _norm(sd, gen, "image_encoder.neck.3", C)

    # ---- prompt encoder ------------------------------------------------------------
    sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"] = torch.randn(2, C // 2, generator=gen)
    for i in range(4):
        sd[f"prompt_encoder.point_embeddings.{i}.weight"] = torch.randn(1, C, generator=gen)
    sd["prompt_encoder.not_a_point_embed.weight"] = torch.randn(1, C, generator=gen)
    m4, m = cfg.mask_in_chans // 4, cfg.mask_in_chans
    sd["prompt_encoder.mask_downscaling.0.weight"] = _uniform(gen, (m4, 1, 2, 2), 0.5)
    sd["prompt_encoder.mask_downscaling.0.bias"] = _uniform(gen, (m4,), 0.5)
    _norm(sd, gen, "prompt_encoder.mask_downscaling.1", m4)
    sd["prompt_encoder.mask_downscaling.3.weight"] = _uniform(gen, (m, m4, 2, 2), 1.0 / math.sqrt(4 * m4))
    sd["prompt_encoder.mask_downscaling.3.bias"] = _uniform(gen, (m,), 1.0 / math.sqrt(4 * m4))
    _norm(sd, gen, "prompt_encoder.mask_downscaling.4", m)
    sd["prompt_encoder.mask_downscaling.6.weight"] = _uniform(gen, (C, m, 1, 1), 1.0 / math.sqrt(m))
    sd["prompt_encoder.mask_downscaling.6.bias"] = _uniform(gen, (C,), 1.0 / math.sqrt(m))
    sd["prompt_encoder.no_mask_embed.weight"] = torch.randn(1, C, generator=gen)

    # ---- mask decoder --------------------------------------------------------------
    for i in range(cfg.dec_depth):
        p = f"mask_decoder.transformer.layers.{i}"
        _dec_attention(sd, gen, p + ".self_attn", C, C)
        _norm(sd, gen, p + ".norm1", C)
        _dec_attention(sd, gen, p + ".cross_attn_token_to_image", C, C // 2)
        _norm(sd, gen, p + ".norm2", C)
        _linear(sd, gen, p + ".mlp.lin1", cfg.dec_mlp_dim, C)
        _linear(sd, gen, p + ".mlp.lin2", C, cfg.dec_mlp_dim)
        _norm(sd, gen, p + ".norm3", C)
        _norm(sd, gen, p + ".norm4", C)
        _dec_attention(sd, gen, p + ".cross_attn_image_to_token", C, C // 2)
    _dec_attention(sd, gen, "mask_decoder.transformer.final_attn_token_to_image", C, C // 2)
    _norm(sd, gen, "mask_decoder.transformer.norm_final_attn", C)
    sd["mask_decoder.iou_token.weight"] = torch.randn(1, C, generator=gen)
    sd["mask_decoder.mask_tokens.weight"] = torch.randn(cfg.num_mask_tokens, C, generator=gen)
    sd["mask_decoder.output_upscaling.0.weight"] = _uniform(gen, (C, C // 4, 2, 2), 1.0 / math.sqrt(C))
    sd["mask_decoder.output_upscaling.0.bias"] = _uniform(gen, (C // 4,), 1.0 / math.sqrt(C))
    _norm(sd, gen, "mask_decoder.output_upscaling.1", C // 4)
    sd["mask_decoder.output_upscaling.3.weight"] = _uniform(gen, (C // 4, C // 8, 2, 2), 1.0 / math.sqrt(C // 4))
    sd["mask_decoder.output_upscaling.3.bias"] = _uniform(gen, (C // 8,), 1.0 / math.sqrt(C // 4))
    for i in range(cfg.num_mask_tokens):
        p = f"mask_decoder.output_hypernetworks_mlps.{i}.layers"
        _linear(sd, gen, p + ".0", C, C)
        _linear(sd, gen, p + ".1", C, C)
        _linear(sd, gen, p + ".2", C // 8, C)
    p = "mask_decoder.iou_prediction_head.layers"
    _linear(sd, gen, p + ".0", cfg.iou_hidden, C)
    _linear(sd, gen, p + ".1", cfg.iou_hidden, cfg.iou_hidden)
    _linear(sd, gen, p + ".2", cfg.num_mask_tokens, cfg.iou_hidden)
    if logit_scale != 1.0:
        for i in range(cfg.num_mask_tokens):
            for leaf in ("weight", "bias"):
                sd[f"mask_decoder.output_hypernetworks_mlps.{i}.layers.2.{leaf}"] *= float(logit_scale)
    return sd


def heavy_tailed(sd, cfg: SamConfig, seed: int = 0, hidden_scale: float = 1.0, v_scale: float = 1.0, gamma_scale: float = 1.0,
                 n_channels: int = 4, blocks=None):
    """A copy of ``sd`` with the outlier structure checkpoints have and N(0, sigma) draws do not (every test
    ran on seeded-normal weights, and the f16 operand type saturates silently at 65504).  In the chosen encoder blocks
    (default: first, middle, last), for ``n_channels`` seeded channels each:

      * ``hidden_scale``: rows of ``mlp.lin1`` (weight and bias) x s, the matching columns of ``mlp.lin2`` / sqrt(s) -- those
        hidden units run s times hotter through the GELU and lin2's operand, and what they add to the residual stream grows by
        sqrt(s): the "massive activation" pattern (dividing by s itself would push the lin2 weights into f16's subnormals, a
        range problem of its own that checkpoints do not have).  The pre-GELU values of the seeded weights reach ~3, so
        s = 1e4 puts GELU(lin1) at ~3e4 (inside f16) and s = 1e5 beyond 65504;
      * ``v_scale``: rows of the v third of ``attn.qkv`` x s, the matching columns of ``attn.proj`` / sqrt(s) (v and the
        attention output, a convex combination of v rows, run s times hotter);
      * ``gamma_scale``: entries of ``norm1.weight`` / ``norm2.weight`` x s (the LayerNorm outputs = the qkv / lin1 operands).

    The oracle (fp32 torch) evaluates the same dict: parity is engine vs oracle on THESE weights."""
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    gen = torch.Generator().manual_seed(424242 + seed)
    D, H = cfg.embed_dim, cfg.mlp_ratio * cfg.embed_dim
    if blocks is None:
        blocks = sorted({0, cfg.depth // 2, cfg.depth - 1})
    for i in blocks:
        p = f"image_encoder.blocks.{i}"
        hid = torch.randperm(H, generator=gen)[:n_channels]
        vch = torch.randperm(D, generator=gen)[:n_channels]
        gch = torch.randperm(D, generator=gen)[:n_channels]
        if hidden_scale != 1.0:
            out[p + ".mlp.lin1.weight"][hid] *= hidden_scale
            out[p + ".mlp.lin1.bias"][hid] *= hidden_scale
            out[p + ".mlp.lin2.weight"][:, hid] /= math.sqrt(hidden_scale)
        if v_scale != 1.0:
            out[p + ".attn.qkv.weight"][2 * D + vch] *= v_scale
            out[p + ".attn.qkv.bias"][2 * D + vch] *= v_scale
            out[p + ".attn.proj.weight"][:, vch] /= math.sqrt(v_scale)
        if gamma_scale != 1.0:
            out[p + ".norm1.weight"][gch] *= gamma_scale
            out[p + ".norm2.weight"][gch] *= gamma_scale
    return out


def make_image(index: int, h: int = 1024, w: int = 1024) -> np.ndarray:
    """uint8 HWC "blob" tile: 40 random filled discs + N(0, 8) noise."""
    rng = np.random.default_rng(1000 + index)
    img = np.full((h, w, 3), rng.integers(40, 200, size=3), dtype=np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(40):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        r = rng.uniform(8, 160)
        col = rng.integers(0, 256, size=3).astype(np.float32)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = col
    img += rng.normal(0.0, 8.0, size=img.shape).astype(np.float32)
    return np.clip(img + 0.5, 0, 255).astype(np.uint8)


def make_noise_image(index: int, h: int = 1024, w: int = 1024) -> np.ndarray:
    """Cheap white-noise tile (throughput runs; generation cost matters there)."""
    rng = np.random.default_rng(1000 + index)
    return rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)


def make_boxes(index: int, n: int = 32, h: int = 1024, w: int = 1024, n_classes: int = 18):
    """``n`` xyxy float32 hboxes + integer labels (DOTA-v2 has 18 classes, GD/mapping.py:46-50)."""
    rng = np.random.default_rng(2000 + index)
    cx, cy = rng.uniform(0, w, n), rng.uniform(0, h, n)
    bw = np.exp(rng.uniform(np.log(8), np.log(512), n))
    bh = np.exp(rng.uniform(np.log(8), np.log(512), n))
    x0, x1 = np.clip(cx - bw / 2, 0, w - 1), np.clip(cx + bw / 2, 0, w - 1)
    y0, y1 = np.clip(cy - bh / 2, 0, h - 1), np.clip(cy + bh / 2, 0, h - 1)
    boxes = np.stack([x0, y0, x1, y1], axis=1).astype(np.float32)
    labels = rng.integers(0, n_classes, n).astype(np.int64)
    return boxes, labels


def make_rboxes(index: int, n: int = 12, h: int = 1024, w: int = 1024, n_classes: int = 37):
    """``n`` rotated boxes as 4-corner polygons [n, 4, 2] (x, y) float32 + labels (FAIR1M has 37 classes,
    GD/mapping.py:58-63): centres uniform, sides log-uniform in [8, 256] px, angle uniform in [0, pi)
    (SURVEY.md 8d, config C4).  Corners may leave the image, like real annotations near a tile border."""
    rng = np.random.default_rng(3000 + index)
    cx, cy = rng.uniform(0, w, n), rng.uniform(0, h, n)
    bw = np.exp(rng.uniform(np.log(8), np.log(256), n))
    bh = np.exp(rng.uniform(np.log(8), np.log(256), n))
    th = rng.uniform(0, np.pi, n)
    c, s_ = np.cos(th), np.sin(th)
    corners = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]])
    polys = np.empty((n, 4, 2), dtype=np.float32)
    for k, (ux, uy) in enumerate(corners):
        polys[:, k, 0] = cx + ux * bw * c - uy * bh * s_
        polys[:, k, 1] = cy + ux * bw * s_ + uy * bh * c
    labels = rng.integers(0, n_classes, n).astype(np.int64)
    return polys, labels


def enclosing_hboxes(polys: np.ndarray) -> np.ndarray:
    """rbox -> hbox by min / max of the corners (main_sam_rhbox_mask_instance.py:125-130)."""
    p = np.asarray(polys)
    return np.stack([p[:, :, 0].min(1), p[:, :, 1].min(1), p[:, :, 0].max(1), p[:, :, 1].max(1)], axis=1).astype(np.float32)


def long_tailed_box_counts(n_images: int, seed: int = 0, mean: float = 32.0, cap: int = 400) -> np.ndarray:
    """Boxes per image of a DOTA-v2-shaped stream: geometric with the given mean,
    at least 1, capped."""
    rng = np.random.default_rng(4000 + seed)
    return np.clip(rng.geometric(1.0 / mean, size=n_images), 1, cap).astype(np.int64)


# BASELINE.json configs[0]: ViT-B, one tile, 4 hboxes (SURVEY.md 8d "C1").
C1_BOXES = np.array(
    [[100, 100, 300, 300], [10, 20, 500, 400], [600, 600, 900, 1000], [0, 0, 1023, 1023]],
    dtype=np.float32,
)
