"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the SAM box->mask path (the parity oracle).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker.  The product (``samrs_amd``) never imports it and fails
loudly when the HIP library is missing.

It is a *functional* fp32 torch-CPU restatement (plain tensors + a ``state_dict``; no
``nn.Module``) of the reference's algorithm, each function citing the reference lines it
follows (paths relative to ``/root/reference/Generate Dataset/segment_anything``).

Pinning: the reference ships no tests / golden vectors for this path, so this
oracle is pinned against outputs of the *reference itself*, run in the authoring container by
``oracle/make_golden.py`` and committed under ``tests/golden/``; ``tests/test_oracle_golden.py``
re-checks it on every CPU run.

``Rounding`` lets a test emulate the engine's reduced-precision points (MFMA operands) on the
CPU, which separates "precision noise" from "logic bug" when the HIP path disagrees with fp32.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

PIXEL_MEAN = (123.675, 116.28, 103.53)   # modeling/sam.py:27
PIXEL_STD = (58.395, 57.12, 57.375)      # modeling/sam.py:28


# ------------------------------------------------------------------------------------------
# optional precision emulation
# ------------------------------------------------------------------------------------------
def split2(dt: torch.dtype) -> Callable[[Tensor], Tensor]:
    """Two-term split of an operand (hi + lo, both in ``dt``): what a 3-MFMA split product sees."""
    def r(x: Tensor) -> Tensor:
        hi = x.to(dt).to(torch.float32)
        return hi + (x - hi).to(dt).to(torch.float32)
    return r


@dataclass
class Rounding:
    """Where the engine feeds MFMA operands (or stores a tensor in the operand type), round like the engine does.

    ``enc`` applies to every encoder point, ``dec`` to every decoder image-side point; ``points`` overrides single
    named points (``"enc.qkv_in"``, ``"dec.keys"``, ...: the names at the call sites below).  A value is a
    ``torch.dtype``, a callable (e.g. ``split2(torch.float16)``) or ``None`` = exact fp32 (the oracle proper).
    ``oracle/error_budget.py`` sweeps them one by one.
    """

    enc: Optional[object] = None
    dec: Optional[object] = None
    points: Optional[Dict[str, Optional[object]]] = None
    # per-block overrides for encoder points: name -> (container of block indices, rounding); ``image_encoder`` sets
    # ``cur_block`` while it walks the blocks (error_budget.py plans5: which BLOCKS carry the error)
    block_points: Optional[Dict[str, tuple]] = None
    cur_block: int = -1

    @staticmethod
    def _r(x: Tensor, dt) -> Tensor:
        if dt is None:
            return x
        if callable(dt) and not isinstance(dt, torch.dtype):
            return dt(x)
        return x.to(dt).to(torch.float32)

    def p(self, name: str) -> Callable[[Tensor], Tensor]:
        """The rounding applied at the named point."""
        if self.block_points is not None and name in self.block_points and self.cur_block in self.block_points[name][0]:
            dt = self.block_points[name][1]
        elif self.points is not None and name in self.points:
            dt = self.points[name]
        else:
            dt = self.enc if name.startswith("enc.") else self.dec
        if hasattr(dt, "matmul"):                 # term-level emulation (split_fp8_lo): _linear needs the object itself
            return dt
        return lambda x: self._r(x, dt)

    def e(self, x: Tensor) -> Tensor:
        return self._r(x, self.enc)

    def d(self, x: Tensor) -> Tensor:
        return self._r(x, self.dec)


# every named rounding point, in data-flow order (error_budget.py iterates over these)
ENC_POINTS = ("enc.patch", "enc.qkv_in", "enc.qkv_out", "enc.relpos", "enc.P", "enc.proj_in", "enc.lin1_in", "enc.lin2_in",
              "enc.neck0", "enc.neck2")
DEC_POINTS = ("dec.keys", "dec.kvq_out", "dec.oi", "dec.up1", "dec.up2", "dec.prod")


_EXACT = Rounding()


def _linear(x: Tensor, sd: SD, name: str, r: Callable[[Tensor], Tensor]) -> Tensor:
    if hasattr(r, "matmul"):                      # a term-level emulation (split_fp8_lo): needs both operands at once
        return r.matmul(x, sd[name + ".weight"]) + sd[name + ".bias"]
    w = r(sd[name + ".weight"])
    return r(x) @ w.t() + sd[name + ".bias"]


class split_fp8_lo:
    """Error-budget probe for a cheaper operand split: x w^T = x_hi w_hi^T + x_lo w_hi^T + x_hi w_lo^T with the hi x hi term on
    `dt` operands and the two CORRECTION terms on fp8 (e4m3, 3 mantissa bits) operands with a power-of-two scale per row --
    what gfx950's block-scaled fp8 MFMA would compute at twice the f16 rate.  The corrections are 2^-11 of the product, so
    their operands' 2^-4 rounding leaves ~2^-15 instead of the plain path's 2^-12."""

    def __init__(self, dt=torch.float16, fmt: str = "e4m3", block: int = 0):
        """fmt: "e4m3" (fp8), "e2m3" / "e3m2" (the two fp6 formats of the f8f6f4 MFMA, 4x the f16 rate), "e2m1" (fp4);
        block: elements along k that share one power-of-two scale (MX: 32; 0 = the whole row)."""
        self.dt, self.fmt, self.block = dt, fmt, block

    _FMT = {"e4m3": (4, 3, 7, 448.0), "e2m3": (2, 3, 1, 7.5), "e3m2": (3, 2, 3, 28.0), "e2m1": (2, 1, 1, 6.0)}   # ebits, mbits, bias, max

    def _q8(self, t: Tensor) -> Tensor:
        ebits, mbits, bias, vmax = self._FMT[self.fmt]
        shp = t.shape
        if self.block and shp[-1] % self.block == 0:
            t = t.reshape(*shp[:-1], shp[-1] // self.block, self.block)
        amax = t.abs().amax(dim=-1, keepdim=True).clamp(min=1e-30)
        # E8M0 scale: the block maximum lands in the top binade of the format
        scale = torch.exp2(torch.floor(torch.log2(amax)) - float(2 ** ebits - 1 - bias - (1 if self.fmt == "e4m3" else 0)))
        v = t / scale
        e = torch.floor(torch.log2(v.abs().clamp(min=1e-30))).clamp(min=float(1 - bias))       # subnormals share the lowest exponent
        step = torch.exp2(e - mbits)
        q = (torch.round(v / step) * step).clamp(-vmax, vmax)
        return (q * scale).reshape(shp)

    def __call__(self, t: Tensor) -> Tensor:      # used where a plain rounding function is expected (weights of a conv, ...)
        hi = t.to(self.dt).to(torch.float32)
        return hi + (t - hi).to(self.dt).to(torch.float32)

    def matmul(self, x: Tensor, w: Tensor) -> Tensor:
        xh, wh = x.to(self.dt).to(torch.float32), w.to(self.dt).to(torch.float32)
        xl, wl = x - xh, w - wh
        return xh @ wh.t() + self._q8(xl) @ self._q8(wh).t() + self._q8(xh) @ self._q8(wl).t()


# ------------------------------------------------------------------------------------------
# image side
# ------------------------------------------------------------------------------------------
def preprocess(image_u8_hwc: np.ndarray, img_size: int = 1024) -> Tensor:
    """uint8 HWC (already long-side == img_size) -> [1,3,S,S] fp32.

    predictor.py:56-58 (HWC -> 1x3xHxW), modeling/sam.py:164-174 ((x-mean)/std, then zero pad
    bottom/right -- the pad value is 0 *after* normalisation).
    """
    x = torch.from_numpy(np.ascontiguousarray(image_u8_hwc)).permute(2, 0, 1)[None].to(torch.float32)
    mean = torch.tensor(PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(PIXEL_STD).view(1, 3, 1, 1)
    x = (x - mean) / std
    h, w = x.shape[-2:]
    return F.pad(x, (0, img_size - w, 0, img_size - h))


def _rel_tables(q_size: int, rel_pos: Tensor) -> Tensor:
    """R[q, k, :] = rel_pos[(q - k) + (S - 1)]  (image_encoder.py:292-322 with q_size == k_size;
    the table already has 2S-1 rows so no interpolation happens, :304-306)."""
    idx = torch.arange(q_size)[:, None] - torch.arange(q_size)[None, :] + (q_size - 1)
    return rel_pos[idx.to(rel_pos.device)]


def _attention(x: Tensor, sd: SD, p: str, heads: int, rd: Rounding) -> Tensor:
    """Multi-head attention with decomposed rel-pos on a [B, S, S, D] grid.

    image_encoder.py:224-240 (qkv split order [3][heads][d], scale on q before QK^T) and
    :325-361 (rel-pos from the UNSCALED q; bias = rel_h[..., kh, None] + rel_w[..., None, kw]).
    """
    B, S, _, D = x.shape
    d = D // heads
    qkv = _linear(x.reshape(B, S * S, D), sd, p + ".qkv", rd.p("enc.qkv_in"))  # [B, N, 3D]
    if (rd.points is not None and "enc.v_in" in rd.points) or (rd.block_points is not None and "enc.v_in" in rd.block_points):      # error-budget probe: the v third of the qkv product on its own rounding
        qkv = torch.cat([qkv[..., :2 * D], _linear(x.reshape(B, S * S, D), sd, p + ".qkv", rd.p("enc.v_in"))[..., 2 * D:]], dim=-1)
    qkv = qkv.reshape(B, S * S, 3, heads, d).permute(2, 0, 3, 1, 4)         # [3, B, h, N, d]
    q, k, v = (t.reshape(B * heads, S * S, d) for t in qkv)
    rq_ = rd.p("enc.qkv_out")
    q, k, v = rq_(q), rq_(k), rq_(v)
    attn = (q * d ** -0.5) @ k.transpose(1, 2)                              # [Bh, N, N]
    Rh = rd.p("enc.relpos")(_rel_tables(S, sd[p + ".rel_pos_h"]))           # [S, S, d]
    Rw = rd.p("enc.relpos")(_rel_tables(S, sd[p + ".rel_pos_w"]))
    rq = q.reshape(B * heads, S, S, d)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    attn = (attn.view(-1, S, S, S, S) + rel_h[..., :, None] + rel_w[..., None, :]).view(-1, S * S, S * S)
    attn = rd.p("enc.P")(attn.softmax(dim=-1))
    o = (attn @ v).view(B, heads, S, S, d).permute(0, 2, 3, 1, 4).reshape(B, S, S, D)
    return _linear(o, sd, p + ".proj", rd.p("enc.proj_in"))


def _block(x: Tensor, sd: SD, p: str, heads: int, window: int, rd: Rounding) -> Tensor:
    """image_encoder.py:166-182; window_partition/unpartition :243-289 (zero pad AFTER norm1,
    bottom/right; padded rows are dropped after attention)."""
    B, H, W, D = x.shape
    y = F.layer_norm(x, (D,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps=1e-6)
    if window > 0:
        ph, pw = (-H) % window, (-W) % window
        y = F.pad(y, (0, 0, 0, pw, 0, ph))
        Hp, Wp = H + ph, W + pw
        y = y.view(B, Hp // window, window, Wp // window, window, D).permute(0, 1, 3, 2, 4, 5)
        y = y.reshape(-1, window, window, D)
        y = _attention(y, sd, p + ".attn", heads, rd)
        y = y.view(B, Hp // window, Wp // window, window, window, D).permute(0, 1, 3, 2, 4, 5)
        y = y.reshape(B, Hp, Wp, D)[:, :H, :W]
    else:
        y = _attention(y, sd, p + ".attn", heads, rd)
    x = x + y
    z = F.layer_norm(x, (D,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps=1e-6)
    z = _linear(z, sd, p + ".mlp.lin1", rd.p("enc.lin1_in"))
    z = F.gelu(z)                                                            # exact erf, common.py:18-26
    z = _linear(z, sd, p + ".mlp.lin2", rd.p("enc.lin2_in"))
    return x + z


def _layernorm2d_cl(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """LayerNorm2d (common.py:31-43) on a channels-LAST tensor == per-pixel LN over C, biased var."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps=eps)


def image_encoder(sd: SD, cfg, x: Tensor, rd: Rounding = _EXACT, taps: Optional[dict] = None) -> Tensor:
    """[B,3,S,S] fp32 -> [B,256,64,64] fp32 (image_encoder.py:106-116).  ``taps`` (if given)
    receives the residual stream after patch-embed and after every block, channels-last."""
    P, D = cfg.patch_size, cfg.embed_dim
    x = F.conv2d(rd.p("enc.patch")(x), rd.p("enc.patch")(sd["image_encoder.patch_embed.proj.weight"]),
                 sd["image_encoder.patch_embed.proj.bias"], stride=P)        # :387-395
    x = x.permute(0, 2, 3, 1) + sd["image_encoder.pos_embed"]                 # :107-109
    if taps is not None:
        taps["patch"] = x.clone()
    for i in range(cfg.depth):
        win = 0 if i in cfg.global_attn_indexes else cfg.window_size
        if rd.block_points is not None:
            rd.cur_block = i
        x = _block(x, sd, f"image_encoder.blocks.{i}", cfg.num_heads, win, rd)
        if taps is not None:
            taps[f"block{i}"] = x.clone()
    # neck :88-104 -- 1x1 conv (no bias), LN2d, 3x3 conv pad 1 (no bias), LN2d
    w0 = sd["image_encoder.neck.0.weight"][:, :, 0, 0]
    y = rd.p("enc.neck0")(x) @ rd.p("enc.neck0")(w0).t()
    y = _layernorm2d_cl(y, sd["image_encoder.neck.1.weight"], sd["image_encoder.neck.1.bias"])
    y = F.conv2d(rd.p("enc.neck2")(y.permute(0, 3, 1, 2)), rd.p("enc.neck2")(sd["image_encoder.neck.2.weight"]), None, padding=1)
    y = _layernorm2d_cl(y.permute(0, 2, 3, 1), sd["image_encoder.neck.3.weight"], sd["image_encoder.neck.3.bias"])
    return y.permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------
# prompt side
# ------------------------------------------------------------------------------------------
def _pe_encoding(sd: SD, coords01: Tensor) -> Tensor:
    """prompt_encoder.py:190-197: 2c-1, @G, *2pi, cat(sin, cos)."""
    g = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = (2.0 * coords01.to(g.dtype) - 1.0) @ g      # (a no-op cast in fp32; lets oracle/ref_noise_floor.py run the same code in fp64)
    c = 2.0 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(sd: SD, cfg) -> Tensor:
    """[1,256,64,64] grid PE at pixel centres (i+0.5)/64 (prompt_encoder.py:62-71,199-209)."""
    g = cfg.grid
    dev = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].device
    t = (torch.arange(g, dtype=torch.float32, device=dev) + 0.5) / g
    yy, xx = torch.meshgrid(t, t, indexing="ij")
    pe = _pe_encoding(sd, torch.stack([xx, yy], dim=-1))
    return pe.permute(2, 0, 1)[None]


def prompt_encoder(sd: SD, cfg, points: Optional[Tuple[Tensor, Tensor]], boxes: Optional[Tensor],
                   masks: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """prompt_encoder.py:128-173.  Returns sparse [B,N,256], dense [B,256,64,64]."""
    C, S = cfg.out_chans, float(cfg.img_size)
    if points is not None:
        bs = points[0].shape[0]
    elif boxes is not None:
        bs = boxes.shape[0]
    elif masks is not None:
        bs = masks.shape[0]
    else:
        bs = 1
    dev = sd["prompt_encoder.no_mask_embed.weight"].device
    sparse = torch.empty(bs, 0, C, device=dev)
    if points is not None:                                                    # :73-91
        coords, labels = points
        coords = coords.to(torch.float32) + 0.5
        if boxes is None:
            coords = torch.cat([coords, torch.zeros(bs, 1, 2, device=coords.device)], dim=1)
            labels = torch.cat([labels, -torch.ones(bs, 1, dtype=labels.dtype, device=labels.device)], dim=1)
        emb = _pe_encoding(sd, coords / S)
        emb[labels == -1] = 0.0
        emb[labels == -1] += sd["prompt_encoder.not_a_point_embed.weight"]
        emb[labels == 0] += sd["prompt_encoder.point_embeddings.0.weight"]
        emb[labels == 1] += sd["prompt_encoder.point_embeddings.1.weight"]
        sparse = torch.cat([sparse, emb], dim=1)
    if boxes is not None:                                                     # :93-100
        c = (boxes.to(torch.float32) + 0.5).reshape(-1, 2, 2)
        emb = _pe_encoding(sd, c / S)
        emb[:, 0, :] += sd["prompt_encoder.point_embeddings.2.weight"]
        emb[:, 1, :] += sd["prompt_encoder.point_embeddings.3.weight"]
        sparse = torch.cat([sparse, emb], dim=1)
    if masks is not None:                                                     # :51-59,102-105
        p = "prompt_encoder.mask_downscaling"
        y = F.conv2d(masks.to(torch.float32), sd[p + ".0.weight"], sd[p + ".0.bias"], stride=2)
        y = _layernorm2d_cl(y.permute(0, 2, 3, 1), sd[p + ".1.weight"], sd[p + ".1.bias"]).permute(0, 3, 1, 2)
        y = F.gelu(y)
        y = F.conv2d(y, sd[p + ".3.weight"], sd[p + ".3.bias"], stride=2)
        y = _layernorm2d_cl(y.permute(0, 2, 3, 1), sd[p + ".4.weight"], sd[p + ".4.bias"]).permute(0, 3, 1, 2)
        y = F.gelu(y)
        dense = F.conv2d(y, sd[p + ".6.weight"], sd[p + ".6.bias"])
    else:                                                                     # :167-171
        dense = sd["prompt_encoder.no_mask_embed.weight"].reshape(1, C, 1, 1).expand(bs, C, cfg.grid, cfg.grid)
    return sparse, dense


# ------------------------------------------------------------------------------------------
# mask decoder
# ------------------------------------------------------------------------------------------
def _dec_attn(sd: SD, p: str, q: Tensor, k: Tensor, v: Tensor, heads: int,
              rq: Callable, rk: Callable, oq: Callable = lambda t: t, ok: Callable = lambda t: t) -> Tensor:
    """transformer.py:218-240: project, split heads, QK^T / sqrt(d) (scale AFTER the product),
    softmax, @V, merge, out_proj.  ``rq`` / ``rk`` round the operands of the q-side and the
    k/v-side projections (image-side ones are MFMA GEMMs in the engine), ``oq`` / ``ok`` their
    outputs (the engine stores image-side projections in the operand type)."""
    q = oq(_linear(q, sd, p + ".q_proj", rq))
    k = ok(_linear(k, sd, p + ".k_proj", rk))
    v = ok(_linear(v, sd, p + ".v_proj", rk))
    b, n, c = q.shape
    d = c // heads

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], heads, d).transpose(1, 2)

    a = split(q) @ split(k).transpose(2, 3)
    a = torch.softmax(a / math.sqrt(d), dim=-1)
    o = (a @ split(v)).transpose(1, 2).reshape(b, n, c)
    return o, p


def _ln(x: Tensor, sd: SD, name: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps=1e-5)


def two_way_transformer(sd: SD, cfg, src: Tensor, pos: Tensor, tokens: Tensor,
                        rd: Rounding = _EXACT) -> Tuple[Tensor, Tensor]:
    """transformer.py:62-106 (outer) and :151-182 (block).  src/pos [B,256,64,64], tokens [B,T,256]."""
    H = cfg.dec_heads
    keys = src.flatten(2).permute(0, 2, 1)
    kpe = pos.flatten(2).permute(0, 2, 1)
    queries, qpe = tokens, tokens
    ident = lambda t: t
    for i in range(cfg.dec_depth):
        p = f"mask_decoder.transformer.layers.{i}"
        if i == 0:                                                            # :154-156
            o, _ = _dec_attn(sd, p + ".self_attn", queries, queries, queries, H, ident, ident)
            queries = _linear(o, sd, p + ".self_attn.out_proj", ident)
        else:
            q = queries + qpe
            o, _ = _dec_attn(sd, p + ".self_attn", q, q, queries, H, ident, ident)
            queries = queries + _linear(o, sd, p + ".self_attn.out_proj", ident)
        queries = _ln(queries, sd, p + ".norm1")
        # tokens -> image                                                     :163-169
        o, _ = _dec_attn(sd, p + ".cross_attn_token_to_image", queries + qpe, keys + kpe, keys, H, ident, rd.p("dec.keys"),
                         ok=rd.p("dec.kvq_out"))
        queries = _ln(queries + _linear(o, sd, p + ".cross_attn_token_to_image.out_proj", ident), sd, p + ".norm2")
        # MLP (ReLU)                                                          :171-174
        m = _linear(F.relu(_linear(queries, sd, p + ".mlp.lin1", ident)), sd, p + ".mlp.lin2", ident)
        queries = _ln(queries + m, sd, p + ".norm3")
        # image -> tokens                                                     :176-181
        o, _ = _dec_attn(sd, p + ".cross_attn_image_to_token", keys + kpe, queries + qpe, queries, H, rd.p("dec.keys"), ident,
                         oq=rd.p("dec.kvq_out"))
        keys = _ln(keys + _linear(o, sd, p + ".cross_attn_image_to_token.out_proj", rd.p("dec.oi")), sd, p + ".norm4")
    p = "mask_decoder.transformer.final_attn_token_to_image"                  # :98-104
    o, _ = _dec_attn(sd, p, queries + qpe, keys + kpe, keys, H, ident, rd.p("dec.keys"), ok=rd.p("dec.kvq_out"))
    queries = _ln(queries + _linear(o, sd, p + ".out_proj", ident), sd, "mask_decoder.transformer.norm_final_attn")
    return queries, keys


def _mlp3(sd: SD, p: str, x: Tensor) -> Tensor:
    """mask_decoder.py:179-201 (ReLU between layers, none after the last)."""
    x = F.relu(x @ sd[p + ".layers.0.weight"].t() + sd[p + ".layers.0.bias"])
    x = F.relu(x @ sd[p + ".layers.1.weight"].t() + sd[p + ".layers.1.bias"])
    return x @ sd[p + ".layers.2.weight"].t() + sd[p + ".layers.2.bias"]


def mask_decoder(sd: SD, cfg, emb: Tensor, pos: Tensor, sparse: Tensor, dense: Tensor,
                 multimask_output: bool, rd: Rounding = _EXACT) -> Tuple[Tensor, Tensor]:
    """mask_decoder.py:71-174.  emb [1,256,64,64]; returns low-res [B,C,256,256], iou [B,C]."""
    B = sparse.shape[0]
    out_tok = torch.cat([sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]], dim=0)
    tokens = torch.cat([out_tok[None].expand(B, -1, -1), sparse], dim=1)      # :127-129
    src = emb.expand(B, -1, -1, -1) + dense                                   # :136-137
    pos_src = pos.expand(B, -1, -1, -1)
    b, c, h, w = src.shape
    hs, keys = two_way_transformer(sd, cfg, src, pos_src, tokens, rd)
    iou_tok = hs[:, 0]
    mask_toks = hs[:, 1:1 + cfg.num_mask_tokens]
    up = keys.transpose(1, 2).reshape(b, c, h, w)
    p = "mask_decoder.output_upscaling"                                       # :53-59
    up = F.conv_transpose2d(rd.p("dec.up1")(up), rd.p("dec.up1")(sd[p + ".0.weight"]), sd[p + ".0.bias"], stride=2)
    up = _layernorm2d_cl(up.permute(0, 2, 3, 1), sd[p + ".1.weight"], sd[p + ".1.bias"]).permute(0, 3, 1, 2)
    up = F.gelu(up)
    up = F.conv_transpose2d(rd.p("dec.up2")(up), rd.p("dec.up2")(sd[p + ".3.weight"]), sd[p + ".3.bias"], stride=2)
    up = F.gelu(up)
    hyper = torch.stack([_mlp3(sd, f"mask_decoder.output_hypernetworks_mlps.{i}", mask_toks[:, i])
                         for i in range(cfg.num_mask_tokens)], dim=1)         # :156-159
    b, c, h, w = up.shape
    masks = (hyper @ rd.p("dec.prod")(up).reshape(b, c, h * w)).reshape(b, -1, h, w)      # :167
    iou = _mlp3(sd, "mask_decoder.iou_prediction_head", iou_tok)              # :172
    sl = slice(1, None) if multimask_output else slice(0, 1)                  # :102-107
    return masks[:, sl], iou[:, sl]


def postprocess_masks(low_res: Tensor, input_size: Tuple[int, int], original_size: Tuple[int, int],
                      img_size: int = 1024) -> Tensor:
    """modeling/sam.py:133-162: bilinear (align_corners=False) to 1024^2, crop, bilinear to original."""
    m = F.interpolate(low_res, (img_size, img_size), mode="bilinear", align_corners=False)
    m = m[..., : input_size[0], : input_size[1]]
    return F.interpolate(m, tuple(original_size), mode="bilinear", align_corners=False)


# ------------------------------------------------------------------------------------------
# predictor-shaped wrapper (predictor.py:17-271)
# ------------------------------------------------------------------------------------------
def get_preprocess_shape(oldh: int, oldw: int, long_side: int) -> Tuple[int, int]:
    """utils/transforms.py:93-102."""
    scale = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


def apply_image(image: np.ndarray, long_side: int = 1024) -> np.ndarray:
    """utils/transforms.py:26-31 (PIL bilinear; identity when the long side already matches)."""
    h, w = get_preprocess_shape(image.shape[0], image.shape[1], long_side)
    if (h, w) == image.shape[:2]:
        return image
    from PIL import Image
    return np.array(Image.fromarray(image).resize((w, h), Image.BILINEAR))


def apply_coords(coords: Tensor, original_size: Tuple[int, int], long_side: int = 1024) -> Tensor:
    """utils/transforms.py:67-81 (copy -> float32, scale x by new_w/old_w, y by new_h/old_h)."""
    oh, ow = original_size
    nh, nw = get_preprocess_shape(oh, ow, long_side)
    c = coords.detach().clone().to(torch.float32)
    c[..., 0] = c[..., 0] * (nw / ow)
    c[..., 1] = c[..., 1] * (nh / oh)
    return c


def apply_boxes(boxes: Tensor, original_size: Tuple[int, int], long_side: int = 1024) -> Tensor:
    """utils/transforms.py:83-91."""
    return apply_coords(boxes.reshape(-1, 2, 2), original_size, long_side).reshape(-1, 4)


class OraclePredictor:
    """Same surface as the reference ``SamPredictor`` (predictor.py), fp32 on the device of the state dict: CPU in every
    test, ``cuda`` only in bench.py's eager-on-GPU baseline leg (the reference as its users run it: torch eager)."""

    mask_threshold = 0.0

    def __init__(self, sd: SD, cfg, rounding: Rounding = _EXACT):
        self.sd, self.cfg, self.rd = sd, cfg, rounding
        self.pe = dense_pe(sd, cfg)
        self.features: Optional[Tensor] = None
        self.is_image_set = False

    @torch.no_grad()
    def set_image(self, image: np.ndarray, image_format: str = "RGB") -> None:
        assert image_format in ("RGB", "BGR")
        if image_format != "RGB":
            image = image[..., ::-1]
        inp = apply_image(image, self.cfg.img_size)
        self.original_size = tuple(image.shape[:2])
        self.input_size = tuple(inp.shape[:2])
        x = preprocess(inp, self.cfg.img_size).to(self.pe.device)     # cpu unless the state dict lives elsewhere
        self.features = image_encoder(self.sd, self.cfg, x, self.rd)
        self.is_image_set = True

    @torch.no_grad()
    def predict_torch(self, point_coords, point_labels, boxes=None, mask_input=None,
                      multimask_output: bool = True, return_logits: bool = False):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        points = (point_coords, point_labels) if point_coords is not None else None
        sparse, dense = prompt_encoder(self.sd, self.cfg, points, boxes, mask_input)
        low, iou = mask_decoder(self.sd, self.cfg, self.features, self.pe, sparse, dense, multimask_output, self.rd)
        masks = postprocess_masks(low, self.input_size, self.original_size, self.cfg.img_size)
        if not return_logits:
            masks = masks > self.mask_threshold
        return masks, iou, low


# ------------------------------------------------------------------------------------------
# driver-loop restatement (Generate Dataset/main_sam_hbox_semantic.py:148-206) and statistics
# ------------------------------------------------------------------------------------------
def paint_semantic(masks: np.ndarray, labels: np.ndarray, shape: Tuple[int, int]):
    """Ordered painting: ``seg_mask`` starts at 255 and later boxes overwrite earlier ones
    (main_sam_hbox_semantic.py:162,195-199); area = number of mask pixels (:204)."""
    seg = np.full(shape, 255, dtype=np.uint8)
    areas = np.zeros(len(labels), dtype=np.int64)
    for j in range(len(labels)):
        m = masks[j].astype(bool)
        seg[m] = labels[j]
        areas[j] = int(m.sum())
    return seg, areas


def class_statistics(areas: np.ndarray, labels: np.ndarray, n_classes: int):
    """Generate Dataset/statistic.py:15-21: per-class pixel and instance counts, skipping area 0."""
    pix = np.zeros(n_classes, dtype=np.int64)
    ins = np.zeros(n_classes, dtype=np.int64)
    for a, l in zip(areas, labels):
        if a > 0:
            pix[int(l)] += int(a)
            ins[int(l)] += 1
    return pix, ins


def box_chunks(n: int, batch_size: int = 20):
    """The reference's chunking: part_num = n // bs + 1, empty tail skipped
    (main_sam_hbox_semantic.py:157-181)."""
    out, start = [], 0
    end = min(n, start + batch_size)
    for _ in range(n // batch_size + 1):
        if start < end:
            out.append((start, end))
        start = end
        end = min(n, start + batch_size)
    return out
