"""``segment_anything.predictor`` alias (Generate Dataset/segment_anything/predictor.py)."""
from samrs_amd.predictor import SamPredictor  # noqa: F401
