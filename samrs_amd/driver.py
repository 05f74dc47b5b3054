"""Image-parallel dataset generation loop + the one collective of the path.

Restates the loop of ``Generate Dataset/main_sam_hbox_semantic.py:110-216`` around the drop-in
predictor: per image ``set_image`` once, ``predict_torch`` on box chunks (box-only prompt,
``multimask_output=False``), ordered painting into ``seg_mask`` (init 255, later box wins), per-box
area.  Painting / areas / class statistics run on the device (``samrs_paint``) so that only the
1 MiB class map and the per-box areas cross PCIe instead of n full-resolution masks.

Multi-GPU: images are independent, so rank r takes ``sorted(files)[r::world]`` (one process per
GPU, full weight replica) and there is NO collective on the data path.  The only exchange is the
dataset statistic of ``Generate Dataset/statistic.py:15-21`` -- per-class pixel and instance
counts -- which every rank accumulates locally as int64 and all-reduces once (``reduce_statistics``;
backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in the CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def shard(items: Sequence, rank: int, world: int) -> List:
    """Static image-parallel sharding (sort first: ``os.listdir`` order is unspecified,
    main_sam_hbox_semantic.py:95)."""
    return list(sorted(items))[rank::world]


def box_chunks(n: int, batch_size: int) -> List[Tuple[int, int]]:
    """Same chunk boundaries as the reference (``part_num = n // bs + 1``; empty tail skipped,
    main_sam_hbox_semantic.py:157-181)."""
    out, start = [], 0
    end = min(n, start + batch_size)
    for _ in range(n // batch_size + 1):
        if start < end:
            out.append((start, end))
        start = end
        end = min(n, start + batch_size)
    return out


@dataclass
class ImageResult:
    seg_mask: torch.Tensor      # uint8 [H, W] on device, 255 = unlabeled
    areas: torch.Tensor         # int64 [n_boxes] on device
    masks: Optional[torch.Tensor] = None   # bool [n_boxes, H, W] when keep_masks


class SemanticGenerator:
    """hbox -> semantic label generation for one rank."""

    def __init__(self, predictor, n_classes: int, box_batch: int = 20):
        self.predictor = predictor
        self.n_classes = n_classes
        self.box_batch = box_batch
        dev = predictor.device
        self.class_pixels = torch.zeros(n_classes, dtype=torch.int64, device=dev)
        self.class_instances = torch.zeros(n_classes, dtype=torch.int64, device=dev)

    @torch.no_grad()
    def process_image(self, image: np.ndarray, boxes: np.ndarray, labels: np.ndarray, keep_masks: bool = False,
                      already_set: bool = False) -> ImageResult:
        p = self.predictor
        if not already_set:
            p.set_image(image)
        h, w = image.shape[:2]
        dev = p.device
        seg = torch.full((h, w), 255, dtype=torch.uint8, device=dev)              # :162
        gt = torch.from_numpy(np.asarray(boxes)).to(dev)
        lab = torch.from_numpy(np.asarray(labels).astype(np.int32)).to(dev)
        areas, kept = [], []
        for s, e in box_chunks(len(labels), self.box_batch):
            tb = p.transform.apply_boxes_torch(gt[s:e], (h, w))                   # :174
            masks, _, _ = p.predict_torch(None, None, tb, None, multimask_output=False)   # :176-181
            a = p.model.engine.paint(masks[:, 0], lab[s:e], seg, self.class_pixels, self.class_instances)  # :195-206
            areas.append(a)
            if keep_masks:
                kept.append(masks[:, 0])
        return ImageResult(seg, torch.cat(areas) if areas else torch.zeros(0, dtype=torch.int64, device=dev),
                           torch.cat(kept) if kept else None)


def reduce_statistics(class_pixels: torch.Tensor, class_instances: torch.Tensor, group=None):
    """SUM all-reduce of the two int64 count vectors (statistic.py:19-21 summed over all ranks).
    One message of 2*C int64 (<= 592 bytes): latency-bound, so a single fused all-reduce."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return class_pixels.clone(), class_instances.clone()
    buf = torch.cat([class_pixels, class_instances]).contiguous()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    c = class_pixels.numel()
    return buf[:c].clone(), buf[c:].clone()
