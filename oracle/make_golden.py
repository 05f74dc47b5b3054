"""TEST INFRASTRUCTURE ONLY -- generate ``tests/golden/*.npz`` by running the REAL reference.

Run in the authoring container (needs ``/root/reference``)::

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden vit_tiny vit_tiny80 vit_b vit_h

Weights / images / prompts come from ``samrs_amd.synth`` (seeded, reproducible anywhere), the
outputs come from the reference's own ``SamPredictor`` (Generate Dataset/segment_anything/
predictor.py) driven the way ``main_sam_hbox_semantic.py:148-206`` and the
``main_sam_*_mask_instance.py`` scripts drive it.  To keep fixtures small, low-res logits are
stored at stride 4 and embeddings at a fixed sub-lattice; full-resolution results are stored as
per-mask areas plus the painted class map (compressed).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from samrs_amd import synth
from oracle import ref_import, sam_oracle

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def cases(cfg_name: str):
    """(tag, image index, image shape, prompt kwargs) -- the same list the tests replay."""
    boxes, labels = synth.make_boxes(0, 6)
    boxes = np.concatenate([synth.C1_BOXES, boxes]).astype(np.float32)
    labels = np.concatenate([np.array([3, 1, 4, 1]), labels]).astype(np.int64)
    pts = np.array([[[200.0, 220.0]], [[512.0, 300.0]], [[900.0, 64.0]]], dtype=np.float32)
    plab = np.ones((3, 1), dtype=np.int32)
    gen = torch.Generator().manual_seed(3)
    mask_in = torch.where(torch.rand(2, 1, 256, 256, generator=gen) > 0.5, 1000.0, -1000.0).numpy()
    out = [
        ("box_single", dict(boxes=boxes, multimask_output=False), labels),
        ("box_multi", dict(boxes=boxes[:3], multimask_output=True), None),
        ("point_multi", dict(point_coords=pts, point_labels=plab, multimask_output=True), None),
        ("mask_single", dict(mask_input=mask_in, multimask_output=False), None),
        ("combo_multi", dict(point_coords=pts[:2], point_labels=plab[:2], boxes=boxes[:2],
                             mask_input=mask_in, multimask_output=True), None),
    ]
    return out


def run_predictor(pred, apply_boxes, apply_coords, img_shape, kw):
    """Drive a SamPredictor-shaped object with numpy prompts given in ORIGINAL-image pixels."""
    t = lambda a, dt: None if a is None else torch.as_tensor(a, dtype=dt)
    boxes = kw.get("boxes")
    pc = kw.get("point_coords")
    tb = None if boxes is None else apply_boxes(torch.as_tensor(boxes), img_shape)
    tp = None if pc is None else apply_coords(torch.as_tensor(pc), img_shape)
    return pred.predict_torch(tp, t(kw.get("point_labels"), torch.int), tb,
                              t(kw.get("mask_input"), torch.float32),
                              multimask_output=kw["multimask_output"])


def main(names):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    for name in names:
        cfg = synth.CONFIGS[name]
        sd = synth.make_state_dict(cfg, 0)
        sa, sam = ref_import.build_reference_sam(cfg, sd)
        pred = sa.SamPredictor(sam)
        shapes = [(1024, 1024)] + ([(600, 800)] if name.startswith("vit_tiny") else [])
        blob = {}
        for si, (h, w) in enumerate(shapes):
            img = synth.make_image(si, h, w)
            pred.set_image(img)
            f = pred.get_image_embedding()
            blob[f"s{si}_emb_sample"] = f[0, ::16, ::4, ::4].numpy().copy()
            blob[f"s{si}_emb_norm"] = np.float64(f.double().norm().item())
            for tag, kw, labels in cases(name):
                kw = dict(kw)
                if h != 1024 or w != 1024:            # keep prompts inside the smaller image
                    for key in ("boxes", "point_coords"):
                        if key in kw:
                            kw[key] = kw[key] * np.float32(min(h, w) / 1024.0)
                masks, iou, low = run_predictor(pred, pred.transform.apply_boxes_torch,
                                                pred.transform.apply_coords_torch, (h, w), kw)
                k = f"s{si}_{tag}"
                blob[k + "_low"] = low[:, :, ::4, ::4].numpy().copy()
                blob[k + "_iou"] = iou.numpy().copy()
                blob[k + "_area"] = masks.flatten(2).sum(-1).numpy().astype(np.int64)
                if labels is not None:
                    seg, areas = sam_oracle.paint_semantic(masks[:, 0].numpy(), labels, (h, w))
                    blob[k + "_seg"] = seg
            print(name, (h, w), "done", flush=True)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **blob)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["vit_tiny", "vit_tiny80", "vit_b"])
