"""``segment_anything.build_sam`` alias (Generate Dataset/segment_anything/build_sam.py)."""
from samrs_amd.build_sam import (build_sam, build_sam_vit_b, build_sam_vit_h, build_sam_vit_l,  # noqa: F401
                                 sam_model_registry)
