"""TEST INFRASTRUCTURE ONLY -- import the real reference ``segment_anything`` read-only.

Works only where ``/root/reference`` exists (the authoring container); the GPU box does not
have it, so nothing under ``tests/ -m gpu``, ``smoke()`` or ``bench.py`` may call this.
It exists to (a) validate ``oracle/sam_oracle.py`` against the reference itself and (b)
generate the golden fixtures committed under ``tests/golden/`` (``oracle/make_golden.py``).

``torchvision`` is not installed and the reference imports it at module import time
(Generate Dataset/segment_anything/utils/transforms.py:10, automatic_mask_generator.py:9),
so a stub is injected.  For 1024-long-side inputs that are already 1024 the
stubbed ``resize`` is the identity.
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = "/root/reference/Generate Dataset"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "segment_anything"))


def _install_torchvision_stub() -> None:
    if "torchvision" in sys.modules:
        return
    import numpy as np
    from PIL import Image

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")
    ops = types.ModuleType("torchvision.ops")
    boxes = types.ModuleType("torchvision.ops.boxes")

    def to_pil_image(a):
        return Image.fromarray(np.asarray(a))

    def resize(img, size):
        h, w = size
        return img.resize((w, h), Image.BILINEAR)

    fn.to_pil_image, fn.resize = to_pil_image, resize
    boxes.batched_nms = boxes.box_area = None
    tv.transforms, tr.functional, tv.ops, ops.boxes = tr, fn, ops, boxes
    for name, mod in [("torchvision", tv), ("torchvision.transforms", tr),
                      ("torchvision.transforms.functional", fn), ("torchvision.ops", ops),
                      ("torchvision.ops.boxes", boxes)]:
        sys.modules[name] = mod


def import_reference():
    """Returns the reference's ``segment_anything`` module (never writes into /root/reference)."""
    if not reference_available():
        raise RuntimeError("reference tree not present on this machine")
    sys.dont_write_bytecode = True
    _install_torchvision_stub()
    # this repo ships an import-name alias package `segment_anything` (the drop-in boundary): if it is already imported,
    # `import segment_anything` would hand the ENGINE back and golden generation would validate the engine against itself
    for name in [n for n in sys.modules if n == "segment_anything" or n.startswith("segment_anything.")]:
        if not (getattr(sys.modules[name], "__file__", None) or "").startswith(REF_ROOT):
            del sys.modules[name]
    if sys.path[:1] != [REF_ROOT]:
        sys.path.insert(0, REF_ROOT)
    import segment_anything  # noqa: E402  (the reference's, not ours)
    assert (segment_anything.__file__ or "").startswith(REF_ROOT), segment_anything.__file__
    return segment_anything


def build_reference_sam(cfg, state_dict):
    """The reference ``Sam`` module for ``cfg`` with ``state_dict`` loaded strictly."""
    sa = import_reference()
    from segment_anything.build_sam import _build_sam  # type: ignore

    sam = _build_sam(
        encoder_embed_dim=cfg.embed_dim,
        encoder_depth=cfg.depth,
        encoder_num_heads=cfg.num_heads,
        encoder_global_attn_indexes=list(cfg.global_attn_indexes),
        checkpoint=None,
    )
    sam.load_state_dict(state_dict, strict=True)
    sam.eval()
    return sa, sam
