"""``sam_model_registry`` -- same names / call shape as the reference
(Generate Dataset/segment_anything/build_sam.py:14-107), returning a ``Sam`` handle whose compute
lives in ``libsamrs_hip.so``.

    sam = sam_model_registry["vit_h"](checkpoint="sam_vit_h_4b8939.pth")   # or checkpoint=None
    sam = sam.to(device="cuda")
    predictor = SamPredictor(sam)

``checkpoint=None`` gives seeded random weights (the reference gives unseeded random weights);
``state_dict=`` lets a caller pass tensors directly.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from .engine import Engine
from .synth import CONFIGS, SamConfig, make_state_dict


class Sam:
    """Weights + (after ``.to('cuda')``) an engine handle.  Mirrors the attributes SAMRS's drivers
    touch: ``image_encoder.img_size`` (main_sam_rbox_mask_instance.py:135), ``device``,
    ``mask_threshold``, ``image_format`` (modeling/sam.py:19-20)."""

    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, cfg: SamConfig, state_dict: Dict[str, torch.Tensor], precision: str = "f16",
                 max_images: int = 1, max_prompts: int = 64, max_points: int = 4, options: Optional[Dict[str, int]] = None):
        """``options``: per-engine options applied BEFORE the weights are loaded (include/samrs_hip.h samrs_set_option:
        "split", "decoder_fusion", "ln_fold", "gemm_variant")."""
        self.cfg = cfg
        self.options = dict(options or {})
        self._state_dict = state_dict
        self.precision = precision
        self.max_images, self.max_prompts, self.max_points = max_images, max_prompts, max_points
        self.image_encoder = SimpleNamespace(img_size=cfg.img_size)
        self.engine: Optional[Engine] = None
        self._device = torch.device("cpu")

    @property
    def device(self) -> torch.device:
        return self._device

    def to(self, device=None, **kwargs) -> "Sam":
        device = torch.device(device if device is not None else kwargs.get("device", "cuda"))
        if device.type != "cuda":
            raise RuntimeError("samrs_amd.Sam can only live on a HIP device ('cuda'); there is no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self.engine is not None and self.engine.device == device:
            return self
        self.engine = Engine(self.cfg, device, self.precision, self.max_images, self.max_prompts, self.max_points)
        for k, v in self.options.items():
            self.engine.set_option(k, v)
        # an explicit operand-split mode for the whole engine is the caller's decision, multimask outputs included (otherwise
        # the engine refuses multimask predicts on embeddings encoded below its multimask grade: samrs_get_slot_info)
        if "split" in self.options and "allow_reduced" not in self.options:
            self.engine.set_option("allow_reduced", 1)
        self.engine.load_state_dict(self._state_dict)
        # what this model runs in when nobody says otherwise (ViT-H: 79, else 15): the multimask-safe mode; the pipelines of
        # driver.py run THEIR OWN calls in it or in the 1x-rate mode by output contract (TilePipeline precision="auto") and leave
        # the engine's option as it is
        self.default_split = self.engine.get_option("split")
        self._device = device
        return self

    def cuda(self, index: Optional[int] = None) -> "Sam":
        return self.to(torch.device("cuda", index) if index is not None else "cuda")

    def eval(self) -> "Sam":
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return self._state_dict


def _build_sam(cfg_name: str, checkpoint: Optional[str] = None, state_dict: Optional[Dict[str, torch.Tensor]] = None,
               **engine_kwargs) -> Sam:
    cfg = CONFIGS[cfg_name]
    if state_dict is None:
        if checkpoint is not None:
            with open(checkpoint, "rb") as f:
                state_dict = torch.load(f, map_location="cpu")
        else:
            state_dict = make_state_dict(cfg, int(os.environ.get("SAMRS_SEED", "0")))
    return Sam(cfg, state_dict, **engine_kwargs)


def build_sam_vit_h(checkpoint=None, **kw):
    return _build_sam("vit_h", checkpoint, **kw)


def build_sam_vit_l(checkpoint=None, **kw):
    return _build_sam("vit_l", checkpoint, **kw)


def build_sam_vit_b(checkpoint=None, **kw):
    return _build_sam("vit_b", checkpoint, **kw)


def build_sam_vit_tiny(checkpoint=None, **kw):      # test-only geometry, not in the reference registry
    return _build_sam("vit_tiny", checkpoint, **kw)


def build_sam_vit_tiny80(checkpoint=None, **kw):
    return _build_sam("vit_tiny80", checkpoint, **kw)


def build_sam_vit_tiny1280(checkpoint=None, **kw):
    return _build_sam("vit_tiny1280", checkpoint, **kw)


build_sam = build_sam_vit_h

sam_model_registry = {
    "default": build_sam_vit_h,
    "vit_h": build_sam_vit_h,
    "vit_l": build_sam_vit_l,
    "vit_b": build_sam_vit_b,
    "vit_tiny": build_sam_vit_tiny,
    "vit_tiny80": build_sam_vit_tiny80,
    "vit_tiny1280": build_sam_vit_tiny1280,
}
