"""``segment_anything`` -- import-name alias of :mod:`samrs_amd`.

SAMRS's generation drivers do ``from segment_anything import sam_model_registry, SamPredictor``
(Generate Dataset/main_sam_hbox_semantic.py:9) and ``from segment_anything.utils.transforms import
ResizeLongestSide`` (main_sam_rbox_mask_instance.py:11).  With this directory on ``sys.path`` ahead of the
reference's vendored copy those lines resolve to the MI355X engine unchanged: same names, same call
shapes, same error behaviour -- the compute is in ``libsamrs_hip.so``.

Not provided (never called by a SAMRS driver): ``SamAutomaticMaskGenerator``,
the ONNX exporter.  Asking for them raises with a message instead of an ImportError deep inside a job.
"""
from samrs_amd import (ResizeLongestSide, SamPredictor, build_sam, build_sam_vit_b, build_sam_vit_h,  # noqa: F401
                       build_sam_vit_l, sam_model_registry)

__all__ = ["build_sam", "build_sam_vit_h", "build_sam_vit_l", "build_sam_vit_b", "sam_model_registry",
           "SamPredictor", "ResizeLongestSide"]


def __getattr__(name):
    if name in ("SamAutomaticMaskGenerator",):
        raise AttributeError(f"segment_anything.{name} is outside the SAMRS box->mask path and is not provided by samrs_amd")
    raise AttributeError(name)
