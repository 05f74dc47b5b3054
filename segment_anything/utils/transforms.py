"""``segment_anything.utils.transforms`` alias (main_sam_rbox_mask_instance.py:11)."""
from samrs_amd.transforms import ResizeLongestSide  # noqa: F401
