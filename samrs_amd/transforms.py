"""``ResizeLongestSide`` -- host-side mirror of the reference helper
(Generate Dataset/segment_anything/utils/transforms.py:16-102) without the torchvision dependency.
Image resize stays on the host (PIL bilinear, identity for 1024-long-side tiles); box / point
scaling is the part that sits on the hot path."""
from __future__ import annotations

from copy import deepcopy
from typing import Optional, Tuple

import numpy as np
import torch


PRECISION_BITS = 32 - 8 - 2      # Pillow Resample.c


def pil_bilinear_coeffs(in_size: int, out_size: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR (triangle, support 1) filter
    over the full input range: returns (bounds int32 [out,2], coef int32 [out,ksize], ksize)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        x = np.arange(xmax, dtype=np.float64)
        w = np.abs((x + xmin - center + 0.5) * ss)
        w = np.where(w < 1.0, 1.0 - w, 0.0)
        ww = 0.0
        for v in w:                     # same left-to-right summation order as the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        q = w * float(1 << PRECISION_BITS)
        coef[xx, :xmax] = np.where(q < 0, (q - 0.5).astype(np.int64), (q + 0.5).astype(np.int64)).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return bounds, coef, ksize


def resample_reference(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """numpy statement of Pillow's two-pass 8-bit resample (horizontal, then vertical); used by the
    tests to pin the coefficient tables against PIL itself."""
    def one_pass(a, out_len, axis):
        a = np.moveaxis(a, axis, 0).astype(np.int64)
        b, k, _ = pil_bilinear_coeffs(a.shape[0], out_len)
        out = np.empty((out_len,) + a.shape[1:], dtype=np.uint8)
        for o in range(out_len):
            k0, kn = b[o]
            ss = (1 << (PRECISION_BITS - 1)) + np.tensordot(k[o, :kn].astype(np.int64), a[k0:k0 + kn], axes=(0, 0))
            out[o] = np.clip(ss >> PRECISION_BITS, 0, 255)
        return np.moveaxis(out, 0, axis)
    h, w = img.shape[:2]
    tmp = one_pass(img, out_w, 1) if out_w != w else img
    return one_pass(tmp, out_h, 0) if out_h != h else tmp


_COEF_CACHE = {}


def _device_coeffs(in_size: int, out_size: int, dev: torch.device):
    """Pillow coefficient tables of one (in, out) pair on `dev`: built once (a Python loop over out_size rows) and kept,
    so that a stream of same-size images (DIOR 800^2, HRSC) pays for the tables and their upload once."""
    key = (in_size, out_size, dev.type, dev.index)
    hit = _COEF_CACHE.get(key)
    if hit is None:
        b, k, ks = pil_bilinear_coeffs(in_size, out_size)
        hit = (torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), ks)
        if len(_COEF_CACHE) > 64:
            _COEF_CACHE.clear()
        _COEF_CACHE[key] = hit
    return hit


class ResizeLongestSide:
    def __init__(self, target_length: int) -> None:
        self.target_length = target_length

    def apply_image(self, image: np.ndarray) -> np.ndarray:
        """HxWxC uint8 -> resized so that the long side == target_length (transforms.py:26-31)."""
        h, w = self.get_preprocess_shape(image.shape[0], image.shape[1], self.target_length)
        if (h, w) == tuple(image.shape[:2]):
            return image
        from PIL import Image
        return np.array(Image.fromarray(image).resize((w, h), Image.BILINEAR))

    def apply_image_device(self, image_u8: torch.Tensor) -> torch.Tensor:
        """Same result as ``apply_image`` (bit-exact with PIL BILINEAR) for a uint8 HWC tensor that is
        already on the GPU: two integer resample passes in libsamrs_hip (next-row N3)."""
        from . import engine
        assert image_u8.is_cuda and image_u8.dtype == torch.uint8 and image_u8.dim() == 3 and image_u8.shape[2] == 3
        h, w = int(image_u8.shape[0]), int(image_u8.shape[1])
        nh, nw = self.get_preprocess_shape(h, w, self.target_length)
        if (nh, nw) == (h, w):
            return image_u8
        lib = engine.load_library()
        dev = image_u8.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        cur = image_u8.contiguous()
        if nw != w:                                   # Pillow: horizontal pass first
            bd, kd, ks = _device_coeffs(w, nw, dev)
            out = torch.empty(h, nw, 3, dtype=torch.uint8, device=dev)
            rc = lib.samrs_resample_pass_u8(cur.data_ptr(), out.data_ptr(), bd.data_ptr(), kd.data_ptr(), ks, w, nw, h, 1, stream)
            assert rc == 0
            cur = out
        if nh != h:
            bd, kd, ks = _device_coeffs(h, nh, dev)
            out = torch.empty(nh, cur.shape[1], 3, dtype=torch.uint8, device=dev)
            rc = lib.samrs_resample_pass_u8(cur.data_ptr(), out.data_ptr(), bd.data_ptr(), kd.data_ptr(), ks, h, nh, cur.shape[1], 0, stream)
            assert rc == 0
            cur = out
        return cur

    def apply_coords(self, coords: np.ndarray, original_size: Tuple[int, ...]) -> np.ndarray:
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(old_h, old_w, self.target_length)
        coords = deepcopy(coords).astype(float)
        coords[..., 0] = coords[..., 0] * (new_w / old_w)
        coords[..., 1] = coords[..., 1] * (new_h / old_h)
        return coords

    def apply_boxes(self, boxes: np.ndarray, original_size: Tuple[int, ...]) -> np.ndarray:
        return self.apply_coords(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)

    def apply_coords_torch(self, coords: torch.Tensor, original_size: Tuple[int, ...]) -> torch.Tensor:
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(old_h, old_w, self.target_length)
        coords = deepcopy(coords).to(torch.float)
        coords[..., 0] = coords[..., 0] * (new_w / old_w)
        coords[..., 1] = coords[..., 1] * (new_h / old_h)
        return coords

    def apply_boxes_torch(self, boxes: torch.Tensor, original_size: Tuple[int, ...]) -> torch.Tensor:
        return self.apply_coords_torch(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)

    @staticmethod
    def get_preprocess_shape(oldh: int, oldw: int, long_side_length: int) -> Tuple[int, int]:
        scale = long_side_length * 1.0 / max(oldh, oldw)
        newh, neww = oldh * scale, oldw * scale
        return int(newh + 0.5), int(neww + 0.5)


FILL_RULES = {"cv2_le_451": 0, "cv2_ge_452": 1}


def resolve_fill_rule(fill_rule: str = "auto") -> str:
    """Which of the two scanline span rules OpenCV has published for ``cv2.fillPoly`` the rasteriser reproduces: ``"cv2_le_451"``
    (ceil(x_left) .. floor(x_right); OpenCV 2.4 - 4.5.1) or ``"cv2_ge_452"`` (both ends rounded half up; OpenCV >= 4.5.2).  The
    reference pins no OpenCV version (`Generate Dataset/main_sam_rbox_mask_instance.py:126-129`); ``"auto"`` takes the rule of the
    cv2 that is installed beside this package -- the one the reference script would have called -- and, when none is importable,
    ``SAMRS_FILL_RULE`` or the older rule (what every fixture of this repository was generated with)."""
    if fill_rule != "auto":
        if fill_rule not in FILL_RULES:
            raise ValueError(f"fill_rule must be 'auto' or one of {sorted(FILL_RULES)}, got {fill_rule!r}")
        return fill_rule
    import os
    env = os.environ.get("SAMRS_FILL_RULE")
    if env:
        return resolve_fill_rule(env)
    try:
        import cv2                                   # noqa: F401  (absent from the build and GPU images)
        ver = tuple(int(x) for x in cv2.__version__.split(".")[:3])
        return "cv2_ge_452" if ver >= (4, 5, 2) else "cv2_le_451"
    except Exception:
        return "cv2_le_451"


def rbox_mask_prompts(polys, original_size: Tuple[int, int], img_size: int = 1024, out_size: int = 256,
                      device: Optional["torch.device"] = None, fill_rule: str = "auto") -> torch.Tensor:
    """Rotated boxes -> SAM mask prompts on the GPU, replacing the cv2 pre-step of
    `Generate Dataset/main_sam_rbox_mask_instance.py:125-141` (fillPoly -> +-1000 -> resize to the
    ResizeLongestSide shape -> pad with -1000 -> resize to 256x256).

    polys: [n, V, 2] (x, y) vertices in original-image pixels (float or int; truncated like the reference's
    `.astype(np.int32)`), 3 <= V <= 8.  Returns fp32 [n, out_size, out_size] on the device; feed
    `prompts[:, None]` as `mask_input` of `SamPredictor.predict_torch` (main_sam_rbox_mask_instance.py:159-164).
    ``fill_rule``: see ``resolve_fill_rule``."""
    if not torch.cuda.is_available():
        raise RuntimeError("rbox_mask_prompts needs the HIP device (no CPU fallback)")
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    p = np.asarray(polys)
    if p.ndim != 3 or p.shape[2] != 2 or not (3 <= p.shape[1] <= 8):
        raise ValueError(f"polys must be [n, V, 2] with 3 <= V <= 8, got {p.shape}")
    pts = torch.from_numpy(np.ascontiguousarray(p.astype(np.int32))).to(dev)
    return _rbox_prompts_from_int_points(pts, original_size, img_size, out_size, fill_rule)


def rbox_mask_prompts_device(polys: torch.Tensor, original_size: Tuple[int, int], img_size: int = 1024, out_size: int = 256,
                             device: Optional["torch.device"] = None, fill_rule: str = "auto") -> torch.Tensor:
    """Same as ``rbox_mask_prompts`` for vertices that already live on the GPU ([n, V, 2] float or int tensor);
    float coordinates are truncated toward zero like the reference's ``.astype(np.int32)``."""
    if polys.dim() != 3 or polys.shape[2] != 2 or not (3 <= polys.shape[1] <= 8):
        raise ValueError(f"polys must be [n, V, 2] with 3 <= V <= 8, got {tuple(polys.shape)}")
    assert polys.is_cuda
    return _rbox_prompts_from_int_points(polys.to(torch.int32).contiguous(), original_size, img_size, out_size, fill_rule)


def _rbox_prompts_from_int_points(pts: torch.Tensor, original_size: Tuple[int, int], img_size: int, out_size: int,
                                  fill_rule: str = "auto") -> torch.Tensor:
    from . import engine as _engine
    lib = _engine.load_library()
    dev = pts.device
    n, nv = int(pts.shape[0]), int(pts.shape[1])
    h, w = int(original_size[0]), int(original_size[1])
    th, tw = ResizeLongestSide.get_preprocess_shape(h, w, img_size)
    out = torch.empty(n, out_size, out_size, dtype=torch.float32, device=dev)
    if n == 0:
        return out
    with torch.cuda.device(dev):
        rc = lib.samrs_rbox_mask_prompt_rule(pts.data_ptr(), n, nv, h, w, th, tw, img_size, out_size, FILL_RULES[resolve_fill_rule(fill_rule)],
                                             out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"samrs_rbox_mask_prompt_rule failed with code {rc}")
    return out
