"""TEST INFRASTRUCTURE ONLY -- generate ``tests/golden/*.npz`` by running the REAL reference.

Run in the authoring container (needs ``/root/reference``)::

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden vit_tiny vit_tiny80 vit_b vit_h
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden vit_b_c2c4 vit_h_c2c4      # C2 / C4 fixtures, margin weights
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden vit_h_c2c4_v1              # a second, independent draw of the same
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden vit_b_inst vit_h_inst      # the instance drivers' recipes as scripted

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


UNSTABLE_TAU = 0.01        # "decision not robust": |full-resolution logit| < UNSTABLE_TAU * std(low-res logits)


def _pack(mask_bool: np.ndarray) -> np.ndarray:
    """[..., H, W] bool -> [..., H*W/8] uint8 (np.packbits, row-major, MSB first)."""
    m = np.ascontiguousarray(mask_bool).reshape(*mask_bool.shape[:-2], -1)
    return np.packbits(m, axis=-1)


def unstable_class_map(logits: "torch.Tensor", tau: float) -> "torch.Tensor":
    """Pixels of the painted class map (later box wins, main_sam_hbox_semantic.py:195-199) whose value could change
    if every logit moved by less than tau: with J = the last box that is ON with margin (logit >= tau), the pixel is
    decided by J unless some LATER box is within tau of the threshold.  Boxes before J are overpainted anyway."""
    n = logits.shape[0]
    idx = torch.arange(1, n + 1).view(n, 1, 1)
    last_on = (idx * (logits >= tau)).amax(0)                 # 1-based index of J, 0 = none
    near = logits.abs() < tau
    return ((idx * near).amax(0) > last_on)                   # an uncertain box after J


def extended_inputs(variant: int = 0):
    """Inputs of the C2 / C4 fixtures: the same on the authoring machine and on the GPU box.  `variant` > 0: a
    second, independent draw (other tile, other boxes) -- the fixture `<name>_c2c4_v<variant>.npz`."""
    boxes, labels = synth.make_boxes(7 + 10 * variant, 32)        # C2: 32 hboxes on one 1024^2 tile, 18 classes
    polys, plabels = synth.make_rboxes(0 + 10 * variant, 4)       # C4: FAIR1M-shaped rotated boxes, 37 classes
    return dict(boxes=boxes, labels=labels, polys=polys, plabels=plabels, hboxes=synth.enclosing_hboxes(polys),
                image_index=variant)


def extended(name: str, variant: int = 0) -> None:
    """``tests/golden/<name>_c2c4.npz``: BASELINE.json configs[1] (32 hboxes per tile, 20 + 12 chunks,
    main_sam_hbox_semantic.py:157-181) and configs[3] (rbox -> enclosing hbox / rbox -> mask prompt with
    multimask_output=True, main_sam_rhbox_mask_instance.py:125-130, main_sam_rbox_mask_instance.py:125-164) run by the
    REAL reference on the realistic-margin weights.  Full-resolution masks are stored bit-packed; next to them the
    set of pixels whose reference decision is not robust (|logit| < UNSTABLE_TAU * std): outside that set the engine
    must reproduce every bit."""
    from oracle import rbox_prompt
    cfg = synth.CONFIGS[name]
    sd = synth.make_state_dict(cfg, 0, logit_scale=synth.MARGIN_LOGIT_SCALE)
    sa, sam = ref_import.build_reference_sam(cfg, sd)
    pred = sa.SamPredictor(sam)
    inp = extended_inputs(variant)
    h = w = 1024
    img = synth.make_image(inp["image_index"], h, w)
    pred.set_image(img)
    f = pred.get_image_embedding()
    blob = {"emb_sample": f[0, ::16, ::4, ::4].numpy().copy(), "emb_norm": np.float64(f.double().norm().item()),
            "logit_scale": np.float64(synth.MARGIN_LOGIT_SCALE), "tau_frac": np.float64(UNSTABLE_TAU)}

    def run(tag, chunks, multimask, **kw):
        logits, ious, lows = [], [], []
        for s0, s1 in chunks:
            sub = {k: (None if v is None else v[s0:s1]) for k, v in kw.items()}
            lg, iou, low = pred.predict_torch(sub.get("pc"), sub.get("pl"), sub.get("boxes"), sub.get("mask_input"),
                                              multimask_output=multimask, return_logits=True)
            logits.append(lg); ious.append(iou); lows.append(low)
        lg, iou, low = torch.cat(logits), torch.cat(ious), torch.cat(lows)
        masks = lg > sam.mask_threshold                                   # predictor.py:242-243
        tau = UNSTABLE_TAU * low.std().item()
        blob[tag + "_masks"] = _pack(masks.numpy())
        blob[tag + "_iou"] = iou.numpy().copy()
        blob[tag + "_low"] = low[:, :, ::4, ::4].numpy().copy()
        blob[tag + "_low_std"] = np.float64(low.std().item())
        blob[tag + "_area"] = masks.flatten(2).sum(-1).numpy().astype(np.int64)
        near = lg.abs() < tau
        blob[tag + "_near"] = near.flatten(2).sum(-1).numpy().astype(np.int64)                # per mask
        blob[tag + "_nearmask"] = _pack(near.numpy())         # the pixels of each mask whose reference decision has no margin
        return lg, masks, tau

    tb = pred.transform.apply_boxes_torch(torch.as_tensor(inp["boxes"]), (h, w))
    lg, masks, tau = run("c2", sam_oracle.box_chunks(32, 20), False, boxes=tb)
    seg, areas = sam_oracle.paint_semantic(masks[:, 0].numpy(), inp["labels"], (h, w))
    blob["c2_seg"] = seg
    blob["c2_unstable"] = _pack(unstable_class_map(lg[:, 0], tau).numpy())
    tb = pred.transform.apply_boxes_torch(torch.as_tensor(inp["hboxes"]), (h, w))
    run("c4box", [(0, 4)], True, boxes=tb)
    prompts = np.stack([rbox_prompt.rbox_mask_prompt(p.astype(np.int32), h, w) for p in inp["polys"]])
    blob["c4mask_prompt_sum"] = np.float64(prompts.astype(np.float64).sum())
    run("c4mask", [(0, 4)], True, mask_input=torch.from_numpy(prompts.astype(np.float32))[:, None])
    path = os.path.join(GOLDEN_DIR, name + "_c2c4" + (f"_v{variant}" if variant else "") + ".npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; unstable class-map pixels:",
          int(np.unpackbits(blob["c2_unstable"]).sum()), flush=True)


INSTANCE_TAU = 2.5e-3      # the engine's own logit-error bound (oracle/parity_sample.py TAU_FRAC), not the 1e-2 of the C2 / C4 fixtures


def instance_inputs(variant: int = 0, n: int = 8):
    """Inputs of the instance-recipe fixture: FAIR1M-shaped rboxes on one tile; the foreground point of an object is its centre."""
    polys, plabels = synth.make_rboxes(20 + 10 * variant, n)
    return dict(polys=polys, plabels=plabels, hboxes=synth.enclosing_hboxes(polys),
                points=polys.mean(axis=1).astype(np.float32), image_index=40 + variant)


def instances(name: str, variant: int = 0) -> None:
    """``tests/golden/<name>_inst.npz``: the three instance drivers' prompt recipes EXACTLY AS SCRIPTED, run by the REAL
    reference, all with ``multimask_output=False``:
      inst_point  main_sam_hbox_mask_instance.py:160-165  point_coords=gt_points[:, None, :] (NOT through apply_coords),
                  point_labels=ones, boxes=None, mask_input=None
      inst_mask   main_sam_rbox_mask_instance.py:159-164  mask_input=rbox_mask_prompts[:, None], nothing else
      inst_rhbox  main_sam_rhbox_mask_instance.py:163-168 boxes=apply_boxes_torch(enclosing hboxes), nothing else
    Stored like the C2 / C4 fixtures: bit-packed full-resolution masks, the per-mask set of pixels whose reference logit is
    within INSTANCE_TAU x std of the threshold, stride-4 low-res logits, IoU predictions, areas."""
    from oracle import rbox_prompt
    cfg = synth.CONFIGS[name]
    sd = synth.make_state_dict(cfg, 0, logit_scale=synth.MARGIN_LOGIT_SCALE)
    sa, sam = ref_import.build_reference_sam(cfg, sd)
    pred = sa.SamPredictor(sam)
    inp = instance_inputs(variant)
    h = w = 1024
    img = synth.make_image(inp["image_index"], h, w)
    pred.set_image(img)
    f = pred.get_image_embedding()
    blob = {"emb_sample": f[0, ::16, ::4, ::4].numpy().copy(), "logit_scale": np.float64(synth.MARGIN_LOGIT_SCALE),
            "tau_frac": np.float64(INSTANCE_TAU)}
    prompts = np.stack([rbox_prompt.rbox_mask_prompt(p.astype(np.int32), h, w) for p in inp["polys"]])
    blob["mask_prompt_sum"] = np.float64(prompts.astype(np.float64).sum())
    n = len(inp["polys"])
    recipes = {
        "inst_point": dict(point_coords=torch.from_numpy(inp["points"])[:, None, :], point_labels=torch.ones(n)[:, None]),
        "inst_mask": dict(mask_input=torch.from_numpy(prompts.astype(np.float32))[:, None]),
        "inst_rhbox": dict(boxes=pred.transform.apply_boxes_torch(torch.as_tensor(inp["hboxes"]), (h, w))),
    }
    for tag, kw in recipes.items():
        lg, iou, low = pred.predict_torch(kw.get("point_coords"), kw.get("point_labels"), kw.get("boxes"), kw.get("mask_input"),
                                          multimask_output=False, return_logits=True)
        masks = lg > sam.mask_threshold
        tau = INSTANCE_TAU * low.std().item()
        near = lg.abs() < tau
        blob[tag + "_masks"] = _pack(masks.numpy())
        blob[tag + "_nearmask"] = _pack(near.numpy())
        blob[tag + "_near"] = near.flatten(2).sum(-1).numpy().astype(np.int64)
        blob[tag + "_iou"] = iou.numpy().copy()
        blob[tag + "_low"] = low[:, :, ::4, ::4].numpy().copy()
        blob[tag + "_low_std"] = np.float64(low.std().item())
        blob[tag + "_area"] = masks.flatten(2).sum(-1).numpy().astype(np.int64)
        print(name, tag, "areas", blob[tag + "_area"].ravel().tolist(), "near", blob[tag + "_near"].ravel().tolist(), flush=True)
    path = os.path.join(GOLDEN_DIR, name + "_inst" + (f"_v{variant}" if variant else "") + ".npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", flush=True)


def main(names):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    import re
    ext = [re.fullmatch(r"(.+)_(c2c4|inst)(?:_v(\d+))?", n) for n in names]
    names = [n for n, m in zip(names, ext) if m is None]
    for m in ext:
        if m is not None:
            (extended if m.group(2) == "c2c4" else instances)(m.group(1), int(m.group(3) or 0))
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
