"""The ``segment_anything`` import name resolves to the MI355X engine (drop-in for
``from segment_anything import sam_model_registry, SamPredictor``, Generate Dataset/main_sam_hbox_semantic.py:9)."""
import importlib
import os
import sys

import pytest


def test_alias_reexports_samrs_amd():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert sys.path[0] == root or root in sys.path
    for name in [m for m in sys.modules if m == "segment_anything" or m.startswith("segment_anything.")]:
        del sys.modules[name]             # an earlier test may have imported the REFERENCE package under this name
    sys.path.insert(0, root)
    try:
        sa = importlib.import_module("segment_anything")
        import samrs_amd
        assert os.path.dirname(sa.__file__) == os.path.join(root, "segment_anything")
        assert sa.sam_model_registry is samrs_amd.sam_model_registry
        assert sa.SamPredictor is samrs_amd.SamPredictor
        from segment_anything.utils.transforms import ResizeLongestSide
        assert ResizeLongestSide is samrs_amd.ResizeLongestSide
        from segment_anything.build_sam import sam_model_registry as reg2
        from segment_anything.predictor import SamPredictor as P2
        assert reg2 is sa.sam_model_registry and P2 is sa.SamPredictor
        assert ResizeLongestSide.get_preprocess_shape(600, 800, 1024) == (768, 1024)
        with pytest.raises(AttributeError, match="outside the SAMRS"):
            sa.SamAutomaticMaskGenerator
    finally:
        sys.path.remove(root)
        for name in [m for m in sys.modules if m == "segment_anything" or m.startswith("segment_anything.")]:
            del sys.modules[name]
