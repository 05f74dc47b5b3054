"""Drop-in for the reference's generation drivers (``Generate Dataset/main_sam_hbox_semantic.py`` and
``main_sam_rhbox_semantic.py``): box annotations + images in, ``gray/*.png`` (uint8 class map, 255 =
unlabeled), ``color/*.png`` (palette image) and ``ins/*.pkl`` (list of ``{mask: RLE, bbox, category,
label, size}``) out -- the on-disk contract read by ``Generate Dataset/statistic.py:12-21`` and the
training datasets.  Paths are flags instead of the reference's hard-coded module globals.

    torchrun --nproc-per-node 8 -m samrs_amd.generate --images DIR --boxes boxes.json --out OUT \
             --model vit_h --checkpoint sam_vit_h_4b8939.pth --classes classes.txt

``boxes.json``: ``{"<image stem>": {"boxes": [[x0,y0,x1,y1], ...], "labels": [int, ...]}, ...}`` (how
the dataset's own annotation format becomes boxes -- loaddata.py -- stays outside the hot path).
Rotated boxes are reduced to their enclosing hbox by the caller
(main_sam_rhbox_semantic.py:125-130).  One process per GPU, rank r takes ``sorted(stems)[r::world]``,
no collective on the data path, one int64 all-reduce for the class statistics at the end.
"""
from __future__ import annotations

import argparse
import json
import os
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import driver, rle


def default_palette(n_classes: int) -> np.ndarray:
    """Deterministic colour per class id (the reference's MAPPING tables, Generate Dataset/mapping.py,
    are dataset-specific constants; pass --palette to reproduce them exactly)."""
    rng = np.random.default_rng(12345)
    return rng.integers(0, 256, size=(n_classes, 3), dtype=np.uint8)


def write_outputs(out_dir: str, stem: str, seg: np.ndarray, masks: Optional[np.ndarray], boxes: np.ndarray,
                  labels: np.ndarray, areas: np.ndarray, palette: np.ndarray, class_names: Sequence[str]) -> None:
    from PIL import Image
    for sub in ("gray", "color", "ins"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    Image.fromarray(seg).save(os.path.join(out_dir, "gray", stem + ".png"))                 # :212,214
    color = np.full((*seg.shape, 3), 255, dtype=np.uint8)                                   # :163
    lab = seg != 255
    color[lab] = palette[seg[lab]]
    Image.fromarray(color).save(os.path.join(out_dir, "color", stem + ".png"))              # :213,215
    info = []
    for j in range(len(labels)):                                                            # :200-206
        entry = {"bbox": boxes[j], "category": class_names[int(labels[j])], "label": int(labels[j]), "size": int(areas[j])}
        if masks is not None:
            entry["mask"] = rle.encode(masks[j])
        info.append(entry)
    with open(os.path.join(out_dir, "ins", stem + ".pkl"), "wb") as f:
        pickle.dump(info, f)                                                                # :216


def run(args) -> Dict[str, List[int]]:
    import torch.distributed as dist
    from PIL import Image

    import samrs_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    ann = json.load(open(args.boxes))
    names = [l.strip() for l in open(args.classes)] if args.classes else [str(i) for i in range(args.n_classes)]
    n_classes = len(names)
    palette = np.load(args.palette) if args.palette else default_palette(n_classes)
    sam = samrs_amd.sam_model_registry[args.model](checkpoint=args.checkpoint, precision=args.precision,
                                                   max_prompts=args.box_batch).to(f"cuda:{local}")
    gen = driver.SemanticGenerator(samrs_amd.SamPredictor(sam), n_classes, box_batch=args.box_batch)
    exts = (".png", ".jpg", ".jpeg", ".tif", ".bmp")
    files = {os.path.splitext(f)[0]: f for f in os.listdir(args.images) if f.lower().endswith(exts)}
    stems = driver.shard([s for s in files if s in ann and len(ann[s]["boxes"]) > 0], rank, world)   # :126-129
    for k, stem in enumerate(stems):
        img = np.array(Image.open(os.path.join(args.images, files[stem])).convert("RGB"))           # :114
        boxes = np.asarray(ann[stem]["boxes"], dtype=np.float32)
        labels = np.asarray(ann[stem]["labels"], dtype=np.int64)
        res = gen.process_image(img, boxes, labels, keep_masks=not args.no_rle)
        masks = res.masks.cpu().numpy() if res.masks is not None else None
        write_outputs(args.out, stem, res.seg_mask.cpu().numpy(), masks, boxes, labels, res.areas.cpu().numpy(), palette, names)
        if rank == 0 and k % 50 == 0:
            print(f"[rank 0] {k}/{len(stems)} images", flush=True)
    pix, ins = driver.reduce_statistics(gen.class_pixels, gen.class_instances)
    stats = {"class_pixel_num": pix.cpu().tolist(), "class_instance_num": ins.cpu().tolist()}
    if rank == 0:
        os.makedirs(os.path.join(args.out, "statistic"), exist_ok=True)
        with open(os.path.join(args.out, "statistic", "class_stats.json"), "w") as f:       # statistic.py:28-31
            json.dump(stats, f)
    return stats


def main(argv=None):
    ap = argparse.ArgumentParser(description="SAM box -> semantic label generation (SAMRS) on MI355X")
    ap.add_argument("--images", required=True)
    ap.add_argument("--boxes", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--model", default="vit_h")
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--precision", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--classes", default=None, help="text file, one class name per line")
    ap.add_argument("--n-classes", type=int, default=18)
    ap.add_argument("--palette", default=None, help=".npy uint8 [n_classes, 3]")
    ap.add_argument("--box-batch", type=int, default=20)                                    # main_sam_hbox_semantic.py:91
    ap.add_argument("--no-rle", action="store_true", help="skip per-instance RLE (only class maps + areas)")
    run(ap.parse_args(argv))


if __name__ == "__main__":
    main()
