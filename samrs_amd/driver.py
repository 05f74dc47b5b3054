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


class InstancePrompter:
    """The prompt recipes of the HRSC2016 / DOTA instance drivers, one object per rank:

    * ``"point"``     -- ``Generate Dataset/main_sam_hbox_mask_instance.py:160-165``: one foreground point per object,
      ``point_coords = gt_points[:, None, :]`` passed AS IS (the reference does not run them through
      ``apply_coords``), labels all 1, no box, no mask;
    * ``"rbox_mask"`` -- ``main_sam_rbox_mask_instance.py:125-164``: the rotated box rasterised to a +-1000 mask prompt
      (``transforms.rbox_mask_prompts`` on the GPU instead of cv2), nothing else;
    * ``"box"``       -- ``main_sam_rhbox_mask_instance.py:160-168``: the enclosing horizontal box through
      ``apply_boxes_torch``.

    Boxes are processed in chunks of the engine's ``max_prompts`` (results are bit-identical to one call).
    Returns (masks bool [n, H, W], qualities fp32 [n]) on the device."""

    MODES = ("point", "rbox_mask", "box")

    def __init__(self, predictor):
        self.predictor = predictor

    @torch.no_grad()
    def predict(self, image: np.ndarray, mode: str, hboxes=None, rboxes=None, points=None, already_set: bool = False):
        from . import transforms
        if mode not in self.MODES:
            raise ValueError(f"mode must be one of {self.MODES}")
        p = self.predictor
        if not already_set:
            p.set_image(image)
        h, w = image.shape[:2]
        dev = p.device
        src = {"point": points, "rbox_mask": rboxes, "box": hboxes}[mode]
        if src is None:
            raise ValueError(f"mode {mode!r} needs its annotation array")
        n = len(src)
        cap = p.model.engine.max_prompts
        masks, quals = [], []
        for s, e in box_chunks(n, cap):
            if mode == "point":
                pc = torch.as_tensor(np.asarray(points[s:e]), dtype=torch.float32, device=dev)[:, None, :]
                pl = torch.ones(e - s, 1, device=dev)
                m, q, _ = p.predict_torch(point_coords=pc, point_labels=pl, boxes=None, mask_input=None, multimask_output=False)
            elif mode == "rbox_mask":
                prompts = transforms.rbox_mask_prompts(np.asarray(rboxes[s:e]), (h, w), img_size=p.model.image_encoder.img_size,
                                                       device=dev)
                m, q, _ = p.predict_torch(point_coords=None, point_labels=None, boxes=None, mask_input=prompts[:, None],
                                          multimask_output=False)
            else:
                tb = p.transform.apply_boxes_torch(torch.as_tensor(np.asarray(hboxes[s:e]), dtype=torch.float32, device=dev), (h, w))
                m, q, _ = p.predict_torch(point_coords=None, point_labels=None, boxes=tb, mask_input=None, multimask_output=False)
            masks.append(m[:, 0])
            quals.append(q[:, 0])
        return torch.cat(masks), torch.cat(quals)


def mean_iou(pred_masks, gt_masks):
    """``main_sam_rhbox_mask_instance.py:222-241``: per-instance IoU averaged over the instances whose union is not empty
    ("Average mIOU"), and total intersection / total union ("Area mIOU").  Lists of [n_i, H, W] arrays, one per image."""
    ious, inter_all, union_all = [], [], []
    for pm, gm in zip(pred_masks, gt_masks):
        pm = np.asarray(pm).astype(bool)
        gm = np.asarray(gm).astype(bool)
        for j in range(pm.shape[0]):
            inter = float(np.sum(pm[j] & gm[j]))
            union = float(np.sum(pm[j] | gm[j]))
            if union > 0:
                inter_all.append(inter)
                union_all.append(union)
                ious.append(inter / union)
    if not ious:
        return float("nan"), float("nan")
    return float(np.mean(ious)), float(np.sum(inter_all) / np.sum(union_all))
