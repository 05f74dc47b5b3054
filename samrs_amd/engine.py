"""ctypes binding of ``libsamrs_hip.so`` (C ABI: ``include/samrs_hip.h``).

PyTorch is plumbing here: it owns device memory and the HIP stream; every computation happens in
the library.  There is NO fallback: if the shared library is missing, or there is no HIP device,
importing / constructing fails loudly.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .synth import SamConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsamrs_hip.so")

PREC_BF16, PREC_F16 = 0, 1
PRECISIONS = {"bf16": PREC_BF16, "f16": PREC_F16, "fp16": PREC_F16}

ABI_VERSION = 5
# "split" option bits (include/samrs_hip.h): rounding points that run as a two-term operand split
SPLIT_PATCH, SPLIT_NECK, SPLIT_OI, SPLIT_UP, SPLIT_DEFAULT = 1, 2, 4, 8, 15
SPLIT_ATTN, SPLIT_MLP, SPLIT_ATTN_V, SPLIT_LIN2, SPLIT_ALL = 16, 32, 64, 128, 255          # reference-grade bits: set before the weights are loaded
# (64 = the attention-side split restricted to the v third of qkv + proj; option "split_depth" = leading blocks they apply to)

OK, ERR_NOT_SET, ERR_BAD_SHAPE, ERR_BAD_ARG, ERR_HIP, ERR_BAD_WEIGHTS, ERR_CAPACITY, ERR_PRECISION, ERR_RANGE = 0, -1, -2, -3, -4, -5, -6, -7, -8


class samrs_config(C.Structure):
    _fields_ = [
        ("embed_dim", C.c_int32), ("depth", C.c_int32), ("num_heads", C.c_int32),
        ("n_global", C.c_int32), ("global_attn_indexes", C.c_int32 * 8),
        ("img_size", C.c_int32), ("patch_size", C.c_int32), ("window_size", C.c_int32),
        ("out_chans", C.c_int32), ("max_images", C.c_int32), ("max_prompts", C.c_int32),
        ("max_points", C.c_int32), ("precision", C.c_int32),
    ]


_lib = None


def load_library() -> C.CDLL:
    """Loads the in-tree shared library; raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SAMRS_LIB_PATH", LIB_PATH)          # A/B builds of the same ABI (tools/, never the tests)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `make -C samrs_amd/csrc` (or __graft_entry__.build()). "
            "samrs_amd has no CPU / PyTorch fallback by design.")
    lib = C.CDLL(path)
    vp, ip, fp = C.c_void_p, C.c_int, C.c_float
    lib.samrs_abi_version.restype = ip
    lib.samrs_create.restype = vp
    lib.samrs_create.argtypes = [C.POINTER(samrs_config), ip, C.c_char_p, ip]
    lib.samrs_destroy.argtypes = [vp]
    lib.samrs_destroy.restype = None
    lib.samrs_last_error.argtypes = [vp]
    lib.samrs_last_error.restype = C.c_char_p
    lib.samrs_load_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), ip]
    lib.samrs_finalize_weights.argtypes = [vp, vp]
    lib.samrs_set_images.argtypes = [vp, vp, ip, ip, ip, ip, vp]
    lib.samrs_set_images_ragged.argtypes = [vp, C.POINTER(vp), C.POINTER(ip), C.POINTER(ip), ip, ip, vp]
    lib.samrs_get_embedding.argtypes = [vp, ip, vp, vp]
    lib.samrs_set_embedding.argtypes = [vp, ip, vp, vp]
    lib.samrs_reset_image.argtypes = [vp, ip]
    lib.samrs_predict.argtypes = [vp, ip, ip, vp, vp, vp, ip, vp, ip, ip, ip, ip, ip, ip, vp, vp, vp, vp]
    lib.samrs_paint.argtypes = [vp, vp, vp, ip, ip, ip, vp, vp, vp, vp, ip, vp]
    lib.samrs_debug_encoder_prefix.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp]
    lib.samrs_debug_encoder_prefix.restype = ip
    lib.samrs_debug_time_dominant_kernel.argtypes = [vp, ip]
    lib.samrs_debug_time_dominant_kernel.restype = ip
    lib.samrs_debug_dominant_kernel_time.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(ip), C.POINTER(ip), C.POINTER(ip), C.POINTER(ip)]
    lib.samrs_debug_dominant_kernel_time.restype = ip
    lib.samrs_resample_pass_u8.argtypes = [vp, vp, vp, vp, ip, ip, ip, ip, ip, vp]
    lib.samrs_resample_pass_u8.restype = ip
    lib.samrs_rbox_mask_prompt.argtypes = [vp, ip, ip, ip, ip, ip, ip, ip, ip, vp, vp]
    lib.samrs_rbox_mask_prompt.restype = ip
    lib.samrs_rbox_mask_prompt_rule.argtypes = [vp, ip, ip, ip, ip, ip, ip, ip, ip, ip, vp, vp]
    lib.samrs_rbox_mask_prompt_rule.restype = ip
    lib.samrs_k_gemm.argtypes = [ip, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, ip, vp]
    lib.samrs_k_gemm_f32.argtypes = [vp, ip, vp, vp, vp, ip, ip, ip, ip, ip, ip, vp]
    lib.samrs_k_gemm_stats.argtypes = [ip, vp, vp, vp, vp, vp, vp, ip, ip, ip, vp]
    lib.samrs_k_gemm_fold.argtypes = [ip, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, vp]
    lib.samrs_k_ln_rowstat.argtypes = [vp, vp, ip, fp, vp]
    lib.samrs_k_ln_fold_weight.argtypes = [ip, vp, vp, vp, vp, vp, vp, vp, ip, ip, vp]
    lib.samrs_k_rowstats_convert.argtypes = [ip, vp, vp, vp, ip, ip, vp]
    lib.samrs_set_option.argtypes = [vp, C.c_char_p, ip]
    lib.samrs_get_option.argtypes = [vp, C.c_char_p, C.POINTER(ip)]
    lib.samrs_rle_encode.argtypes = [vp, vp, ip, ip, ip, vp, C.c_int64, vp, vp, vp]
    lib.samrs_k_convert_split.argtypes = [ip, vp, vp, vp, C.c_int64, vp]
    lib.samrs_select_best.argtypes = [vp, vp, vp, ip, ip, ip, ip, vp, vp, vp, vp]
    lib.samrs_k_upscaler_fused.argtypes = [ip, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, vp]
    lib.samrs_k_convert.argtypes = [ip, vp, vp, C.c_int64, vp]
    lib.samrs_k_layernorm.argtypes = [ip, vp, vp, vp, fp, vp, vp, ip, ip, ip, ip, ip, ip, vp]
    lib.samrs_k_window_attention.argtypes = [ip, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, vp]
    lib.samrs_k_global_attention.argtypes = [ip, vp, vp, vp, vp, ip, ip, ip, ip, vp]
    lib.samrs_k_postprocess.argtypes = [vp, ip, ip, ip, ip, ip, ip, ip, vp, vp]
    lib.samrs_k_neck_im2col.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.samrs_k_neck_im2col.restype = ip
    lib.samrs_k_gemm_gln.argtypes = [ip, vp, vp, vp, vp, vp, ip, ip, ip, vp, vp, vp]
    lib.samrs_k_upscale2_masks.argtypes = [ip, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, vp]
    lib.samrs_k_gemm_split3.argtypes = [ip, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, vp]
    i64 = C.c_int64
    lib.samrs_k_mx_scale_bytes.argtypes = [ip, ip, ip]
    lib.samrs_k_mx_scale_bytes.restype = i64
    lib.samrs_k_mx4_pack.argtypes = [ip, vp, vp, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, vp]
    lib.samrs_k_gemm_mx.argtypes = [ip, vp, vp, vp, vp, ip, ip, ip, ip, vp, vp, vp, vp, vp, vp, vp, vp, ip, ip, ip, vp]
    lib.samrs_k_layernorm_mx.argtypes = [ip, vp, vp, vp, fp, vp, ip, ip, vp, vp, vp, vp, vp]
    lib.samrs_k_attention_mx.argtypes = [ip, ip, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, vp, vp, vp, vp, vp]
    lib.samrs_k_gemm_mx_gelu_mxout.argtypes = [ip, vp, vp, vp, vp, ip, ip, ip, ip, vp, vp, vp, vp, vp, vp, vp, vp, ip, vp, vp, vp, vp, vp]
    for name in ("samrs_k_mx4_pack", "samrs_k_gemm_mx", "samrs_k_layernorm_mx", "samrs_k_attention_mx", "samrs_k_gemm_mx_gelu_mxout"):
        getattr(lib, name).restype = ip
    lib.samrs_get_slot_info.argtypes = [vp, ip, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    for name in ("samrs_load_weight", "samrs_finalize_weights", "samrs_set_images", "samrs_set_images_ragged", "samrs_get_embedding",
                 "samrs_set_embedding", "samrs_reset_image", "samrs_predict", "samrs_paint", "samrs_k_gemm",
                 "samrs_k_gemm_f32", "samrs_k_convert", "samrs_k_layernorm", "samrs_k_window_attention",
                 "samrs_k_global_attention", "samrs_k_postprocess", "samrs_k_gemm_gln", "samrs_k_upscale2_masks",
                 "samrs_k_gemm_stats", "samrs_k_gemm_fold", "samrs_k_ln_fold_weight", "samrs_k_rowstats_convert", "samrs_k_ln_rowstat",
                 "samrs_set_option", "samrs_get_option", "samrs_rle_encode", "samrs_k_convert_split", "samrs_select_best",
                 "samrs_k_upscaler_fused", "samrs_k_gemm_split3", "samrs_get_slot_info"):
        getattr(lib, name).restype = ip
    if lib.samrs_abi_version() != ABI_VERSION:
        raise ImportError("libsamrs_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class EngineError(RuntimeError):
    pass


class PrecisionError(EngineError):
    """``multimask_output=True`` on an embedding that was encoded below the operand-split mode this model's multimask
    outputs need (SAMRS_ERR_PRECISION; ``Engine.get_slot_info``).  Re-encode the image in the engine's default mode
    (``SamPredictor.set_image`` does), or accept the reduced mode with ``engine.set_option("allow_reduced", 1)``."""


class OperandRangeError(EngineError):
    """Option ``range_check`` = 2: the encoder pass of ``set_image`` / ``set_images`` saturated values of an MFMA operand tensor
    (f16 tops out at 65504 and every conversion on the path saturates there; SAMRS_ERR_RANGE).  The checkpoint's activations do
    not fit the f16 operand type: build the model with ``precision="bf16"``."""


class Engine:
    """One engine handle = one GPU's weights + workspaces (``samrs_engine_t``)."""

    def __init__(self, cfg: SamConfig, device: torch.device, precision: str = "f16", max_images: int = 1,
                 max_prompts: int = 64, max_points: int = 4):
        if device.type != "cuda":
            raise EngineError("samrs_amd runs on a HIP device only (torch device type 'cuda'); there is no CPU path")
        if not torch.cuda.is_available():
            raise EngineError("no HIP device visible to PyTorch")
        self.lib = load_library()
        self.cfg = cfg
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        self.precision = precision
        self.max_images, self.max_prompts, self.max_points = max_images, max_prompts, max_points
        c = samrs_config()
        c.embed_dim, c.depth, c.num_heads = cfg.embed_dim, cfg.depth, cfg.num_heads
        c.n_global = len(cfg.global_attn_indexes)
        for i, g in enumerate(cfg.global_attn_indexes):
            c.global_attn_indexes[i] = g
        c.img_size, c.patch_size, c.window_size, c.out_chans = cfg.img_size, cfg.patch_size, cfg.window_size, cfg.out_chans
        c.max_images, c.max_prompts, c.max_points = max_images, max_prompts, max_points
        c.precision = PRECISIONS[precision]
        err = C.create_string_buffer(512)
        with torch.cuda.device(self.device):
            self.handle = self.lib.samrs_create(C.byref(c), self.device.index, err, 512)
        if not self.handle:
            raise EngineError("samrs_create failed: " + err.value.decode())

    # -- error mapping: same exception types / messages as the reference (predictor.py:133-134 ...)
    def _check(self, rc: int) -> None:
        if rc == OK:
            return
        msg = self.lib.samrs_last_error(self.handle).decode()
        if rc == ERR_NOT_SET:
            raise RuntimeError(msg)
        if rc in (ERR_BAD_SHAPE, ERR_BAD_ARG):
            raise AssertionError(msg)
        if rc == ERR_PRECISION:
            raise PrecisionError(msg)
        if rc == ERR_RANGE:
            raise OperandRangeError(msg)
        raise EngineError(f"libsamrs_hip error {rc}: {msg}")

    # -- per-engine options (include/samrs_hip.h: "split", "decoder_fusion", "ln_fold", "gemm_variant")
    def set_option(self, name: str, value: int) -> None:
        self._check(self.lib.samrs_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int()
        self._check(self.lib.samrs_get_option(self.handle, name.encode(), C.byref(v)))
        return v.value

    @contextlib.contextmanager
    def options(self, **values: int):
        """Options in force for the calls made inside the block only (a pipeline's operand-split mode must not outlive the
        pipeline's own calls: the engine is shared with every SamPredictor built on the same model)."""
        old = {k: self.get_option(k) for k in values}
        try:
            for k, v in values.items():
                if v != old[k]:
                    self.set_option(k, v)
            yield self
        finally:
            for k, v in old.items():
                if self.get_option(k) != v:
                    self.set_option(k, v)

    def get_slot_info(self, slot: int = 0) -> Dict[str, int]:
        """{"is_set", "split", "split_depth"} of an embedding slot: the operand-split mode its image was encoded in (-1: the
        embedding was installed by set_embedding)."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.samrs_get_slot_info(self.handle, slot, C.byref(a), C.byref(b), C.byref(c)))
        return {"is_set": int(a.value), "split": int(b.value), "split_depth": int(c.value)}

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.samrs_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Strict load (build_sam.py:103-106): every key / shape of the reference state_dict."""
        for name, t in sd.items():
            a = np.ascontiguousarray(t.detach().to(torch.float32).cpu().numpy())
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.samrs_load_weight(self.handle, name.encode(), a.ctypes.data, shape, a.ndim))
        with torch.cuda.device(self.device):
            self._check(self.lib.samrs_finalize_weights(self.handle, _stream()))

    def set_images(self, images_u8: torch.Tensor, slot0: int = 0) -> None:
        """images_u8: uint8 [n, H, W, 3] on this device, long side == img_size."""
        assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3
        assert images_u8.is_cuda and images_u8.is_contiguous()
        n, h, w, _ = images_u8.shape
        with torch.cuda.device(self.device):
            self._check(self.lib.samrs_set_images(self.handle, images_u8.data_ptr(), n, h, w, slot0, _stream()))

    def set_images_ragged(self, images_u8, slot0: int = 0) -> None:
        """One encoder pass over tiles of DIFFERENT sizes: a sequence of uint8 [H_i, W_i, 3] device tensors, each with
        long side == img_size (samrs_set_images_ragged)."""
        n = len(images_u8)
        for t in images_u8:
            assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[-1] == 3 and t.is_cuda and t.is_contiguous()
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in images_u8])
        hs = (C.c_int * n)(*[int(t.shape[0]) for t in images_u8])
        ws = (C.c_int * n)(*[int(t.shape[1]) for t in images_u8])
        with torch.cuda.device(self.device):
            self._check(self.lib.samrs_set_images_ragged(self.handle, ptrs, hs, ws, n, slot0, _stream()))

    def debug_encoder_prefix(self, images_u8: torch.Tensor, n_blocks: int) -> torch.Tensor:
        """Test hook: residual stream [n, 64, 64, D] after patch embed + the first n_blocks blocks."""
        n, h, w, _ = images_u8.shape
        out = torch.empty(n, self.cfg.grid, self.cfg.grid, self.cfg.embed_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.samrs_debug_encoder_prefix(self.handle, images_u8.data_ptr(), n, h, w, n_blocks,
                                                            out.data_ptr(), _stream()))
        return out

    def outlier_columns(self, block: int, gemm: int):
        """Test hook: the outlier K-columns the engine picked for block GEMM `gemm` (0 qkv, 1 lin1, 2 lin2, 3 proj) of encoder block `block`."""
        buf = (C.c_int32 * 32)()
        self.lib.samrs_debug_outlier_columns.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        self.lib.samrs_debug_outlier_columns.restype = C.c_int
        n = self.lib.samrs_debug_outlier_columns(self.handle, block, gemm, buf)
        if n < 0:
            self._check(n)
        return [int(buf[i]) for i in range(n)]

    def time_dominant_kernel(self, enable: bool) -> None:
        self._check(self.lib.samrs_debug_time_dominant_kernel(self.handle, int(enable)))

    def dominant_kernel_time(self):
        """(average ms, launches, N, K) of the MLP lin1+GELU GEMM launches timed since the last call."""
        ms, n, m, nn, kk = C.c_float(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.samrs_debug_dominant_kernel_time(self.handle, C.byref(ms), C.byref(n), C.byref(m), C.byref(nn), C.byref(kk)))
        return ms.value, n.value, nn.value, kk.value

    def get_embedding(self, slot: int = 0) -> torch.Tensor:
        out = torch.empty(1, self.cfg.out_chans, self.cfg.grid, self.cfg.grid, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.samrs_get_embedding(self.handle, slot, out.data_ptr(), _stream()))
        return out

    def set_embedding(self, emb: torch.Tensor, slot: int = 0) -> None:
        emb = emb.to(self.device, torch.float32).contiguous()
        assert emb.numel() == self.cfg.out_chans * self.cfg.grid ** 2
        with torch.cuda.device(self.device):
            self._check(self.lib.samrs_set_embedding(self.handle, slot, emb.data_ptr(), _stream()))

    def reset_image(self, slot: int = 0) -> None:
        self._check(self.lib.samrs_reset_image(self.handle, slot))

    def predict(self, slot: int, boxes: Optional[torch.Tensor], point_coords: Optional[torch.Tensor],
                point_labels: Optional[torch.Tensor], mask_input: Optional[torch.Tensor], multimask_output: bool,
                return_logits: bool, input_size: Tuple[int, int], original_size: Tuple[int, int],
                want_masks: bool = True) -> Tuple[Optional[torch.Tensor], torch.Tensor, torch.Tensor]:
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        if point_coords is not None and point_labels is None:
            raise AssertionError("point_labels must be supplied if point_coords is supplied.")
        # shape contract of predictor.py:169-177 / prompt_encoder.py:73-105 -- the library reads raw pointers, so a
        # mis-shaped prompt must be rejected here (the reference fails with a shape error inside the prompt encoder)
        n = None
        if point_coords is not None:
            if point_coords.dim() != 3 or point_coords.shape[-1] != 2:
                raise ValueError(f"point_coords must be BxNx2, got {tuple(point_coords.shape)}")
            n, npts = int(point_coords.shape[0]), int(point_coords.shape[1])
            if tuple(point_labels.shape) != (n, npts):
                raise ValueError(f"point_labels must be BxN = {(n, npts)}, got {tuple(point_labels.shape)}")
            if npts < 1 or npts > self.max_points:
                raise ValueError(f"{npts} points per prompt, but this engine was created with max_points={self.max_points} "
                                 "(hard limit 8: the decoder keeps at most 16 tokens per prompt); pass max_points= to "
                                 "sam_model_registry[...]")
            point_coords = point_coords.to(**f32).contiguous()
            point_labels = point_labels.to(dtype=torch.int32, device=dev).contiguous()
        if boxes is not None:
            if boxes.numel() % 4 != 0 or boxes.shape[-1] != 4:
                raise ValueError(f"boxes must be Bx4, got {tuple(boxes.shape)}")
            boxes = boxes.to(**f32).reshape(-1, 4).contiguous()
            if n is not None and boxes.shape[0] != n:
                raise ValueError(f"boxes has {boxes.shape[0]} rows but the point prompts have {n}")
            n = int(boxes.shape[0])
        if mask_input is not None:
            side = 4 * self.cfg.grid
            if mask_input.dim() != 4 or tuple(mask_input.shape[1:]) != (1, side, side):
                raise ValueError(f"mask_input must be Bx1x{side}x{side}, got {tuple(mask_input.shape)}")
            if n is not None and mask_input.shape[0] != n:
                raise ValueError(f"mask_input has batch {mask_input.shape[0]} but the other prompts have {n}")
            mask_input = mask_input.to(**f32).contiguous()
            n = int(mask_input.shape[0])
        if n is None:
            raise AssertionError("at least one prompt (points, boxes or mask_input) is required")
        if n < 1:
            raise ValueError("empty prompt batch")
        npts = 0 if point_coords is None else point_coords.shape[1]
        c = 3 if multimask_output else 1
        oh, ow = int(original_size[0]), int(original_size[1])
        masks = None
        if want_masks:
            masks = torch.empty(n, c, oh, ow, dtype=torch.float32 if return_logits else torch.uint8, device=dev)
        iou = torch.empty(n, c, **f32)
        low = torch.empty(n, c, 256, 256, **f32)
        with torch.cuda.device(dev):
            self._check(self.lib.samrs_predict(
                self.handle, slot, n, _ptr(boxes), _ptr(point_coords), _ptr(point_labels), npts, _ptr(mask_input),
                int(bool(multimask_output)), int(bool(return_logits)), int(input_size[0]), int(input_size[1]), oh, ow,
                _ptr(masks), iou.data_ptr(), low.data_ptr(), _stream()))
        if masks is not None and not return_logits:
            masks = masks.view(torch.bool)
        return masks, iou, low

    def rle_encode(self, masks: torch.Tensor, out: torch.Tensor, cursor: torch.Tensor, table: torch.Tensor) -> None:
        """COCO RLE strings of `masks` ([n, H, W] bool / uint8 on this device) appended to the byte buffer `out` (uint8, device)
        behind `cursor` (int64 [1], device, in / out); `table` (int64 [n, 3], device) receives (offset, length, n_counts)
        per mask.  Asynchronous on the current stream; see samrs_rle_encode in samrs_hip.h."""
        m = masks.view(torch.uint8) if masks.dtype == torch.bool else masks
        m = m.reshape(-1, m.shape[-2], m.shape[-1]).contiguous()
        n, h, w = m.shape
        assert out.dtype == torch.uint8 and out.is_cuda and out.is_contiguous() and out.data_ptr() % 16 == 0
        assert cursor.dtype == torch.int64 and cursor.is_cuda and table.dtype == torch.int64 and table.is_cuda
        assert table.is_contiguous() and table.numel() >= 3 * n
        with torch.cuda.device(self.device):
            self._check(self.lib.samrs_rle_encode(self.handle, m.data_ptr(), n, h, w, out.data_ptr(), out.numel(),
                                                  cursor.data_ptr(), table.data_ptr(), _stream()))

    def select_best(self, masks: torch.Tensor, iou: torch.Tensor, best_out: Optional[torch.Tensor] = None,
                    quality_out: Optional[torch.Tensor] = None, areas_out: Optional[torch.Tensor] = None):
        """Best-of-C by predicted IoU on the device (samrs_select_best): masks [n, C, H, W] bool / uint8, iou [n, C] ->
        (best masks uint8 [n, H, W], quality fp32 [n], areas int64 [n]); the outputs may be caller-owned contiguous slices."""
        m = masks.view(torch.uint8) if masks.dtype == torch.bool else masks
        n, c, h, w = m.shape
        assert m.is_contiguous() and iou.is_contiguous() and iou.dtype == torch.float32 and tuple(iou.shape) == (n, c)
        best = best_out if best_out is not None else torch.empty(n, h, w, dtype=torch.uint8, device=self.device)
        qual = quality_out if quality_out is not None else torch.empty(n, dtype=torch.float32, device=self.device)
        areas = areas_out if areas_out is not None else torch.empty(n, dtype=torch.int64, device=self.device)
        assert best.is_contiguous() and qual.is_contiguous() and areas.is_contiguous() and areas.dtype == torch.int64
        with torch.cuda.device(self.device):
            self._check(self.lib.samrs_select_best(self.handle, m.data_ptr(), iou.data_ptr(), n, c, h, w, best.data_ptr(),
                                                   qual.data_ptr(), areas.data_ptr(), _stream()))
        return best, qual, areas

    def paint(self, masks: torch.Tensor, labels: torch.Tensor, seg: torch.Tensor,
              class_pixels: Optional[torch.Tensor] = None, class_instances: Optional[torch.Tensor] = None,
              areas_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Ordered painting + areas (+ class statistics) on device; see samrs_paint in samrs_hip.h.  `areas_out`: a caller-owned
        contiguous int64 [n] slice to receive the areas (the pipeline's [batch, max_boxes] table) instead of a new tensor."""
        m = masks.view(torch.uint8) if masks.dtype == torch.bool else masks
        m = m.reshape(-1, m.shape[-2], m.shape[-1]).contiguous()
        n, h, w = m.shape
        labels = labels.to(dtype=torch.int32, device=self.device).contiguous()
        if areas_out is not None:
            assert areas_out.dtype == torch.int64 and areas_out.is_contiguous() and areas_out.numel() == n and areas_out.is_cuda
        areas = areas_out if areas_out is not None else torch.empty(n, dtype=torch.int64, device=self.device)
        ncls = 0 if class_pixels is None else class_pixels.numel()
        with torch.cuda.device(self.device):
            self._check(self.lib.samrs_paint(self.handle, m.data_ptr(), labels.data_ptr(), n, h, w, seg.data_ptr(),
                                             areas.data_ptr(), _ptr(class_pixels), _ptr(class_instances), ncls, _stream()))
        return areas
