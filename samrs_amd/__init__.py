"""samrs_amd -- MI355X-native SAM box->mask engine behind the reference's ``segment_anything``
surface (``sam_model_registry``, ``SamPredictor``) used by SAMRS's generation drivers."""
from .build_sam import (build_sam, build_sam_vit_b, build_sam_vit_h, build_sam_vit_l, sam_model_registry)
from .predictor import SamPredictor
from .transforms import ResizeLongestSide

__all__ = ["build_sam", "build_sam_vit_h", "build_sam_vit_l", "build_sam_vit_b", "sam_model_registry",
           "SamPredictor", "ResizeLongestSide"]
