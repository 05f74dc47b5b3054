"""Image-parallel dataset generation loop + the one collective of the path.

Restates the loop of ``Generate Dataset/main_sam_hbox_semantic.py:110-216`` around the drop-in
predictor: per image ``set_image`` once, ``predict_torch`` on box chunks (box-only prompt,
``multimask_output=False``), ordered painting into ``seg_mask`` (init 255, later box wins), per-box
area.  Painting / areas / class statistics run on the device (``samrs_paint``) so that only the
1 MiB class map and the per-box areas cross PCIe instead of n full-resolution masks.

``TilePipeline`` is the production loop (what ``samrs_amd.generate`` and ``bench.py`` run): batches of
tiles through ONE encoder pass, H2D of batch k+1 and decode + paint + D2H of batch k-1 overlapped with the
encoder of batch k on separate HIP streams.  ``SemanticGenerator`` is the same computation one image at a
time on one stream (the reference's shape); the two give bit-identical outputs (tests/test_pipeline_gpu.py).

Multi-GPU: images are independent, so rank r takes ``sorted(files)[r::world]`` (one process per
GPU, full weight replica) -- or, for long-tailed box counts, pulls batches of image indices from a
shared counter (``WorkQueue``) -- and there is NO collective on the data path.  The only exchange is the
dataset statistic of ``Generate Dataset/statistic.py:15-21`` -- per-class pixel and instance
counts -- which every rank accumulates locally as int64 and all-reduces once (``reduce_statistics``;
backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in the CPU tests).
"""
from __future__ import annotations

import contextlib
import os

import queue
import threading
from dataclasses import dataclass, field
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

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
    dev = buf.device
    if dist.get_backend(group) != "nccl":            # gloo (CPU tests, ranks sharing one GPU): the message goes through the host
        buf = buf.cpu()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    buf = buf.to(dev)
    c = class_pixels.numel()
    return buf[:c].clone(), buf[c:].clone()


def agree_on_list(items: List, group=None, src: int = 0) -> List:
    """Every rank gets RANK `src`'s copy of a Python list (``broadcast_object_list``; no-op outside a process group).
    Used wherever the work list is derived from something a rank could see differently from its peers -- e.g. ``generate
    --resume`` lists the output directory while faster ranks may already be writing into it: the shards
    (``sorted(files)[r::world]``, Generate Dataset/main_sam_hbox_semantic.py:110) are only disjoint and complete when all
    ranks index the SAME list."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(items)
    box = [list(items) if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


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

    def __init__(self, predictor, fill_rule: str = "auto"):
        """fill_rule: which cv2.fillPoly span rule the rbox rasteriser reproduces (transforms.resolve_fill_rule: "auto" = that of the
        cv2 installed beside this package, else the older one)."""
        from . import transforms
        self.predictor = predictor
        self.fill_rule = transforms.resolve_fill_rule(fill_rule)

    @torch.no_grad()
    def predict(self, image: np.ndarray, mode: str, hboxes=None, rboxes=None, points=None, already_set: bool = False,
                multimask_output: bool = False):
        """multimask_output=True (BASELINE.json configs[3]) decodes the three multimask outputs per object and keeps the
        one with the highest predicted IoU."""
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
                m, q, _ = p.predict_torch(point_coords=pc, point_labels=pl, boxes=None, mask_input=None, multimask_output=multimask_output)
            elif mode == "rbox_mask":
                prompts = transforms.rbox_mask_prompts(np.asarray(rboxes[s:e]), (h, w), img_size=p.model.image_encoder.img_size,
                                                       device=dev, fill_rule=self.fill_rule)
                m, q, _ = p.predict_torch(point_coords=None, point_labels=None, boxes=None, mask_input=prompts[:, None],
                                          multimask_output=multimask_output)
            else:
                tb = p.transform.apply_boxes_torch(torch.as_tensor(np.asarray(hboxes[s:e]), dtype=torch.float32, device=dev), (h, w))
                m, q, _ = p.predict_torch(point_coords=None, point_labels=None, boxes=tb, mask_input=None, multimask_output=multimask_output)
            best = q.argmax(1) if multimask_output else torch.zeros(e - s, dtype=torch.long, device=dev)
            rows = torch.arange(e - s, device=dev)
            masks.append(m[rows, best])
            quals.append(q[rows, best])
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


# ------------------------------------------------------------------------------------------------
# Work distribution across ranks
# ------------------------------------------------------------------------------------------------
class WorkQueue:
    """Hands out consecutive index ranges [start, end) of a sorted work list to the ranks of one job.

    * ``mode="static"``  -- rank r owns the chunks r, r + world, r + 2 world, ... (no communication; with
      ``chunk=1`` this is ``sorted(files)[r::world]``);
    * ``mode="dynamic"`` -- a shared counter: every pull is one atomic fetch-add of ``chunk`` on the job's
      rendezvous store (``torch.distributed`` TCPStore ``add``: one small round trip to rank 0's store thread,
      ~0.1 ms, against ~60 ms of GPU work per 8-tile chunk).  A rank that drew images with hundreds of boxes
      (DOTA-v2: a few per cent of the tiles) simply pulls less often, so the 8 GPUs finish together instead
      of waiting for the unluckiest static shard.  The data path itself still has no collective.
    """

    _seq = 0          # queues created by this process: every rank creates its queues in the same order

    def __init__(self, n_items: int, chunk: int = 1, rank: int = 0, world: int = 1, mode: str = "static",
                 store=None, name: Optional[str] = None):
        """`name` keys the shared counter on the store: it must be the same on every rank and UNIQUE per work list (a spent
        counter would hand a second queue of the same name nothing).  Default: a per-process sequence number."""
        if mode not in ("static", "dynamic"):
            raise ValueError("mode must be 'static' or 'dynamic'")
        if name is None:
            name = f"samrs_wq/{WorkQueue._seq}"
        WorkQueue._seq += 1
        self.n, self.chunk, self.rank, self.world, self.mode = int(n_items), int(chunk), rank, world, mode
        self._k = 0
        self._local = 0
        self._key = name + "/head"
        self._store = None
        if mode == "dynamic" and world > 1:
            if store is None:
                import torch.distributed as dist
                from torch.distributed.distributed_c10d import _get_default_store
                if not dist.is_initialized():
                    raise RuntimeError("dynamic WorkQueue over several ranks needs an initialised process group (its store)")
                store = _get_default_store()
            self._store = store

    def pull(self) -> Optional[Tuple[int, int]]:
        """Next range, or None when the list is exhausted."""
        if self.mode == "static":
            start = (self._k * self.world + self.rank) * self.chunk
            self._k += 1
        elif self._store is None:
            start = self._local
            self._local += self.chunk
        else:
            start = int(self._store.add(self._key, self.chunk)) - self.chunk
        if start >= self.n:
            return None
        return start, min(self.n, start + self.chunk)

    def __iter__(self) -> Iterator[Tuple[int, int]]:
        while True:
            r = self.pull()
            if r is None:
                return
            yield r


def gather_mask_sizes(local_sizes: Sequence[int], group=None) -> List[int]:
    """All ranks' per-instance mask sizes in rank order -- the list `Generate Dataset/statistic.py:34-53` builds from
    every ``ins/*.pkl`` (``all_mask_size``).  Variable length per rank: all-gather of the counts, then of the
    sizes padded to the longest rank (RCCL on the GPU box, gloo in the CPU tests)."""
    import torch.distributed as dist
    sizes = torch.as_tensor(list(local_sizes), dtype=torch.int64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sizes.tolist()
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    cnt = torch.tensor([sizes.numel()], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    m = max(int(c.item()) for c in counts)
    buf = torch.zeros(max(m, 1), dtype=torch.int64, device=dev)
    buf[: sizes.numel()] = sizes.to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    out: List[int] = []
    for c, b in zip(counts, bufs):
        out.extend(b[: int(c.item())].cpu().tolist())
    return out


# ------------------------------------------------------------------------------------------------
# The production loop
# ------------------------------------------------------------------------------------------------
@dataclass
class WorkItem:
    """One image of the stream: uint8 HWC pixels (host numpy array, host tensor or device tensor), its boxes in
    ORIGINAL-image pixels (xyxy) and their class labels; `key` travels to the sink untouched."""
    key: object
    image: object
    boxes: np.ndarray
    labels: np.ndarray


@dataclass
class TileResult:
    key: object
    seg_mask: np.ndarray                    # uint8 [H, W] host, 255 = unlabeled (a view of a pinned ring buffer)
    areas: np.ndarray                       # int64 [n_boxes] host
    boxes: np.ndarray
    labels: np.ndarray
    masks: Optional[np.ndarray] = None      # uint8 [n_boxes, H, W] host when keep_masks
    quality: Optional[np.ndarray] = None    # fp32 [n_boxes] predicted IoU of the kept mask (instance pipelines)
    rle_table: Optional[np.ndarray] = None  # int64 [n_boxes, 3] (offset, length, n_counts) into rle_data when rle=True
    rle_data: Optional[np.ndarray] = None   # uint8 view of the batch's pinned RLE byte buffer
    size: Optional[Tuple[int, int]] = None  # (H, W) of the tile

    def rle(self, j: int) -> dict:
        """COCO RLE of instance j exactly as the reference stores it (main_sam_hbox_semantic.py:201-202):
        ``{"size": [H, W], "counts": str}``; encoded on the device (samrs_rle_encode)."""
        off, n, _ = (int(v) for v in self.rle_table[j])
        return {"size": [int(self.size[0]), int(self.size[1])], "counts": self.rle_data[off:off + n].tobytes().decode("ascii")}


class _OutBuf:
    def __init__(self, batch: int, side: int, max_boxes: int, rle: bool = False):
        self.seg = torch.empty(batch, side, side, dtype=torch.uint8).pin_memory()
        self.areas = torch.empty(batch, max_boxes, dtype=torch.int64).pin_memory()
        self.done = torch.cuda.Event()
        self.masks: List[Optional[torch.Tensor]] = [None] * batch     # keep_masks: host copies of the full masks
        self.odd: dict = {}                                           # tiles that are not side x side: their class maps
        # rle: per-box (offset, length, n_counts), the number of bytes used, and the strings themselves (grown on demand)
        self.rle_tab = torch.zeros(batch * max_boxes, 3, dtype=torch.int64).pin_memory() if rle else None
        self.rle_cur = torch.zeros(1, dtype=torch.int64).pin_memory() if rle else None
        self.rle_bytes = torch.empty(1 << 20, dtype=torch.uint8).pin_memory() if rle else None


class TilePipeline:
    """hbox -> semantic labels for a stream of tiles (main_sam_hbox_semantic.py:110-216), restructured for the GPU:

        stream h2d : pinned staging -> HBM            tiles + boxes of batch k+1
        stream enc : samrs_set_images(_ragged)        batch k   (one encoder pass over `batch` tiles)
        stream dec : predict (box chunks) + paint     batch k-1 (reads the OTHER embedding slot set)
                     + D2H of class maps / areas
        host       : hands batch k-2 to `sink`        (PNG / pickle writers run in a thread pool there)

    Two embedding slot sets and two input staging sets, `out_depth` pinned output buffers.  What crosses PCIe per
    tile: 3 MiB in, 1 MiB class map + 8 B per box out (`samrs_paint` runs on the device), and with `rle=True` the
    per-instance COCO RLE strings of main_sam_hbox_semantic.py:201-202, encoded on the device (`samrs_rle_encode`:
    a few KB per mask on real data) -- the full-resolution masks themselves never leave HBM unless `keep_masks`.  Bit-identical to `SemanticGenerator` (no kernel depends on batch composition or on what
    runs next to it).  Tiles of one batch may differ in size (non-1024 tiles are resized on the GPU, bit-exact with
    PIL, and encoded through samrs_set_images_ragged)."""

    BOX_WIDTH = 4          # floats per annotation: xyxy

    def __init__(self, sam, n_classes: int, batch: int = 8, box_batch: int = 20, keep_masks: bool = False,
                 out_depth: int = 3, max_boxes: int = 512, device_inputs: bool = False, rle: bool = False,
                 rle_buffer_mb: int = 256, precision="auto", _multimask: bool = False):
        """precision: the operand-split mode (engine option "split") THIS PIPELINE'S OWN CALLS run in.  The option is set around
        each of the pipeline's encode / decode calls and restored afterwards (``Engine.options``), so the mode never outlives
        them: a ``SamPredictor`` built on the same model keeps the engine's own default (round 3 changed the engine's option for
        good, and whoever built the last pipeline decided everybody's precision).  The mode an image was encoded in is
        recorded with its embedding slot (``Engine.get_slot_info``).
        "auto" = by output contract: this pipeline only ever asks for the single mask of token 0, which holds IoU >= 0.9995 against
        the reference with every block GEMM at the 1x f16 rate (C2 fixtures): split 15; a multimask pipeline
        (InstancePipeline(multimask=True)) runs in the model's own default, which at ViT-H adds the v third of qkv + proj on
        hi + lo operands (split 79: what the three multimask tokens need for IoU >= 0.999, C4 fixtures; 0.90x the throughput).
        An explicit choice -- builder ``options={"split": ...}`` or SAMRS_SPLIT -- is never overridden.
        "engine" = the engine's option as it stands at each call; an int = that mode (and, for a multimask pipeline, the
        caller's consent to run its multimask predicts in it: option "allow_reduced")."""
        from .transforms import ResizeLongestSide
        eng = sam.engine
        if eng is None:
            raise RuntimeError("move the model to the GPU first: sam.to('cuda')")
        self.split_mode = self._choose_split(sam, precision, multimask=_multimask)
        # consent to reduced-precision multimask predicts only means something for a pipeline that issues them
        self.allow_reduced = _multimask and isinstance(precision, int) and not isinstance(precision, bool)
        self.eng = eng
        if self.split_mode is not None:
            # fail HERE when the engine cannot run the mode (e.g. a block-GEMM bit whose lo weights were not finalized), not
            # mid-run inside _encode after _stage has queued H2D copies and recorded events (round-4 advisor finding)
            with self._mode():
                pass
        if out_depth < 2:
            raise ValueError("out_depth must be >= 2: batch k-1's results are still on loan to the sink when batch k decodes")
        if eng.max_images < 2 * batch:
            raise ValueError(f"TilePipeline(batch={batch}) needs an engine with max_images >= {2 * batch} "
                             f"(two embedding slot sets); got {eng.max_images}")
        self.sam, self.eng, self.dev = sam, eng, eng.device
        self.batch, self.box_batch, self.keep_masks, self.max_boxes = batch, box_batch, keep_masks, max_boxes
        self.rle = rle
        self.side = sam.cfg.img_size
        self.transform = ResizeLongestSide(sam.image_encoder.img_size)
        self.class_pixels = torch.zeros(n_classes, dtype=torch.int64, device=self.dev)
        self.class_instances = torch.zeros(n_classes, dtype=torch.int64, device=self.dev)
        dev, side = self.dev, self.side
        # (round 5, measured: creating the encoder stream at the device's highest priority changes nothing -- 143.7 against 144.0
        # images/s over three alternations on one box; the streams stay at the default priority)
        self.s_h2d, self.s_enc, self.s_dec = (torch.cuda.Stream(dev) for _ in range(3))
        self.device_inputs = device_inputs
        self.pin_in = None
        if not device_inputs:
            self.pin_in = [torch.empty(batch, side, side, 3, dtype=torch.uint8).pin_memory() for _ in range(2)]
        if rle:      # per input set: the batch's RLE strings (packed, 16-byte aligned), a cursor, (offset, length, n_counts) per box
            self.s_d2h = torch.cuda.Stream(dev)
            self.rle_dev = [torch.empty(rle_buffer_mb << 20, dtype=torch.uint8, device=dev) for _ in range(2)]
            self.rle_cur = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)]
            self.rle_tab = [torch.zeros(batch * max_boxes, 3, dtype=torch.int64, device=dev) for _ in range(2)]
        self.dev_in = [torch.empty(batch, side, side, 3, dtype=torch.uint8, device=dev) for _ in range(2)]
        bw = self.BOX_WIDTH
        self.pin_box = [torch.empty(batch * max_boxes, bw, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.pin_lab = [torch.empty(batch * max_boxes, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.dev_box = [torch.empty(batch * max_boxes, bw, dtype=torch.float32, device=dev) for _ in range(2)]
        self.dev_lab = [torch.empty(batch * max_boxes, dtype=torch.int32, device=dev) for _ in range(2)]
        self.seg_dev = [torch.empty(batch, side, side, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.area_dev = [torch.zeros(batch, max_boxes, dtype=torch.int64, device=dev) for _ in range(2)]
        self.ev_h2d = [torch.cuda.Event() for _ in range(2)]
        self.ev_enc = [torch.cuda.Event() for _ in range(2)]
        self.ev_dec = [torch.cuda.Event() for _ in range(2)]
        self.ev_in_free = [torch.cuda.Event() for _ in range(2)]     # encoder has consumed input set b
        self.free_out: "queue.Queue[_OutBuf]" = queue.Queue()
        for _ in range(out_depth):
            self.free_out.put(_OutBuf(batch, side, max_boxes, rle))

    @staticmethod
    def _choose_split(sam, precision, multimask: bool) -> Optional[int]:
        """The "split" mode a pipeline's own calls run in; None = whatever the engine's option says at each call."""
        from .engine import SPLIT_DEFAULT
        if precision == "engine":
            return None
        if precision == "auto":
            if "split" in getattr(sam, "options", {}) or "SAMRS_SPLIT" in os.environ:
                return None                                          # an explicit choice for the whole engine stands
            return int(getattr(sam, "default_split", SPLIT_DEFAULT)) if multimask else SPLIT_DEFAULT
        return int(precision)

    @contextlib.contextmanager
    def _mode(self):
        """The engine in this pipeline's operand-split mode for the calls inside the block (host-side state read at launch)."""
        if self.split_mode is None:
            yield
            return
        kw = {"split": self.split_mode}
        if self.allow_reduced:
            kw["allow_reduced"] = 1
        with self.eng.options(**kw):
            yield

    # -- stage A: stage tiles + boxes of one batch, H2D on s_h2d ------------------------------------------------
    def _stage(self, b: int, items: List[WorkItem]):
        tiles = []         # per image: (device tensor uint8 [h, w, 3] with long side == img_size, original (H, W))
        same = True
        n_off = 0
        offs = []
        self.ev_in_free[b].synchronize()                  # pinned / device input set b is free again (host wait: the
        for i, it in enumerate(items):                    # pinned buffer must not be overwritten under an in-flight copy)
            nb = len(it.labels)
            if nb > self.max_boxes:
                raise ValueError(f"{nb} boxes on one image > max_boxes={self.max_boxes}")
            self.pin_box[b][n_off:n_off + nb] = torch.as_tensor(np.asarray(it.boxes, dtype=np.float32).reshape(-1, self.BOX_WIDTH))
            self.pin_lab[b][n_off:n_off + nb] = torch.as_tensor(np.asarray(it.labels).astype(np.int32))
            offs.append((n_off, nb))
            n_off += nb
        with torch.cuda.stream(self.s_h2d):
            self.s_h2d.wait_event(self.ev_dec[b])          # the decoder of batch k-2 read box / label set b
            self.dev_box[b][:n_off].copy_(self.pin_box[b][:n_off], non_blocking=True)
            self.dev_lab[b][:n_off].copy_(self.pin_lab[b][:n_off], non_blocking=True)
            for i, it in enumerate(items):
                img = it.image
                H, W = int(img.shape[0]), int(img.shape[1])
                native = (H == self.side and W == self.side)
                if isinstance(img, torch.Tensor) and img.is_cuda:
                    t = img
                    if native:
                        self.dev_in[b][i].copy_(t, non_blocking=True)
                        t = self.dev_in[b][i]
                elif native:
                    src = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))
                    if src.is_pinned():                                           # caller-owned pinned memory: straight H2D
                        self.dev_in[b][i].copy_(src, non_blocking=True)
                    else:
                        if self.pin_in is None:                                   # device_inputs=True promised resident tiles
                            self.pin_in = [torch.empty(self.batch, self.side, self.side, 3, dtype=torch.uint8).pin_memory()
                                           for _ in range(2)]
                        # host memcpy into pinned staging: numpy's (one thread, lock released), not Tensor.copy_, which fans a
                        # 3 MiB copy out over every OpenMP thread -- 5.6 ms instead of 0.1 ms on a 16-CPU slice of a 256-thread host
                        np.copyto(self.pin_in[b][i].numpy(), src.numpy())
                        self.dev_in[b][i].copy_(self.pin_in[b][i], non_blocking=True)
                    t = self.dev_in[b][i]
                else:
                    src = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))
                    t = src.to(self.dev)                                          # odd sizes: plain upload, then
                if not native:
                    t = self.transform.apply_image_device(t.contiguous())         # PIL-exact resize on the GPU
                    t.record_stream(self.s_enc)                                   # allocated on s_h2d, read by the encoder
                    same = False
                tiles.append((t, (H, W)))
            self.ev_h2d[b].record(self.s_h2d)
        return tiles, same, offs

    def _encode(self, b: int, tiles, same: bool):
        n = len(tiles)
        with torch.cuda.stream(self.s_enc):
            self.s_enc.wait_event(self.ev_h2d[b])
            self.s_enc.wait_event(self.ev_dec[b])                  # decoder is done with embedding slot set b
            with self._mode():
                if same:
                    self.eng.set_images(self.dev_in[b][:n], b * self.batch)
                else:
                    self.eng.set_images_ragged([t for t, _ in tiles], b * self.batch)
            self.ev_enc[b].record(self.s_enc)
            self.ev_in_free[b].record(self.s_enc)

    def _decode_tile(self, b: int, i: int, tile, hw, off: int, nb: int, out: _OutBuf) -> None:
        """Everything the reference does per image after set_image (main_sam_hbox_semantic.py:157-206), on s_dec."""
        eng, (H, W) = self.eng, hw
        in_size = (int(tile.shape[0]), int(tile.shape[1]))
        native = (H, W) == (self.side, self.side)
        seg = self.seg_dev[b][i] if native else torch.full((H, W), 255, dtype=torch.uint8, device=self.dev)
        kept = []
        for s, e in box_chunks(nb, self.box_batch):                          # :157-181
            tb = self._input_frame_boxes(self.dev_box[b][off + s:off + e], (H, W), in_size)   # :174
            masks, _, _ = eng.predict(b * self.batch + i, tb, None, None, None, False, False, in_size, (H, W))
            eng.paint(masks[:, 0], self.dev_lab[b][off + s:off + e], seg, self.class_pixels, self.class_instances,
                      areas_out=self.area_dev[b][i, s:e])
            if self.rle:                                                          # :201-202, on the device
                eng.rle_encode(masks[:, 0], self.rle_dev[b], self.rle_cur[b], self.rle_tab[b][off + s:off + e])
            if self.keep_masks:
                kept.append(masks[:, 0].view(torch.uint8))
        if native:
            out.seg[i].copy_(seg, non_blocking=True)
        else:
            out.odd[i] = seg.cpu()                    # other sizes (DIOR 800^2, HRSC): synchronous copy, rare path
        # keep_masks: the full-resolution masks themselves (n MiB per tile instead of 1 MiB, copied synchronously) -- a
        # debugging / evaluation mode; the reference's pkl contract needs only their RLE (rle=True, encoded on the device)
        out.masks[i] = torch.cat(kept).cpu() if (self.keep_masks and kept) else None

    def _input_frame_boxes(self, boxes: torch.Tensor, hw, in_size) -> torch.Tensor:
        """apply_boxes_torch (utils/transforms.py:83-91).  When the tile already has the encoder's input size both scale
        factors are exactly 1.0 and x * 1.0f == x bit for bit, so the boxes are handed over as they are (no launches)."""
        if tuple(hw) == tuple(in_size):
            return boxes
        return self.transform.apply_boxes_torch(boxes, hw)

    def _decode(self, b: int, items, tiles, offs, out: _OutBuf):
        with torch.cuda.stream(self.s_dec):
            self.s_dec.wait_event(self.ev_enc[b])
            self.seg_dev[b].fill_(255)                                               # main_sam_hbox_semantic.py:162
            if self.rle:
                self.rle_cur[b].zero_()
            with self._mode():
                for i, ((t, hw), (off, nb)) in enumerate(zip(tiles, offs)):
                    self._decode_tile(b, i, t, hw, off, nb, out)
            out.areas.copy_(self.area_dev[b], non_blocking=True)
            if self.rle:
                out.rle_tab.copy_(self.rle_tab[b], non_blocking=True)
                out.rle_cur.copy_(self.rle_cur[b], non_blocking=True)
            self._extra_outputs(b, out)
            self.ev_dec[b].record(self.s_dec)
            out.done.record(self.s_dec)

    def _extra_outputs(self, b: int, out: _OutBuf) -> None:
        pass

    def _fetch_rle(self, b: int, out: _OutBuf, n_boxes: int):
        """The batch's RLE strings: the table and the byte count are on the host (out.done), so the strings can be copied
        with their exact size -- on a copy stream, while the GPU works on the batches already queued."""
        total = int(out.rle_cur[0])
        tab = out.rle_tab[:n_boxes].numpy()
        if n_boxes and int(tab[:, 1].min()) < 0:
            need = int((-tab[:, 1] - 1).max())
            raise RuntimeError(f"RLE buffer too small: a mask needs {need} bytes and the batch already holds {total}; raise "
                               f"rle_buffer_mb (now {self.rle_dev[b].numel() >> 20})")
        if out.rle_bytes.numel() < total:
            out.rle_bytes = torch.empty(max(total, 2 * out.rle_bytes.numel()), dtype=torch.uint8).pin_memory()
        if total:
            with torch.cuda.stream(self.s_d2h):
                out.rle_bytes[:total].copy_(self.rle_dev[b][:total], non_blocking=True)
            self.s_d2h.synchronize()
        return tab, out.rle_bytes.numpy()

    def _finish(self, pending, sink):
        items, offs, out, b = pending
        out.done.synchronize()
        odd = out.odd
        res = []
        rtab = rdat = None
        if self.rle:
            rtab, rdat = self._fetch_rle(b, out, sum(nb for _, nb in offs))
        for i, (it, (off, nb)) in enumerate(zip(items, offs)):
            seg = (odd[i].numpy() if odd[i] is not None else None) if i in odd else out.seg[i].numpy()
            m = out.masks[i].numpy() if out.masks[i] is not None else None
            q = out.quality[i, :nb].numpy().copy() if getattr(out, "quality", None) is not None else None
            r = TileResult(it.key, seg, out.areas[i, :nb].numpy().copy(), np.asarray(it.boxes), np.asarray(it.labels), m, q)
            if seg is not None:
                r.size = (int(seg.shape[0]), int(seg.shape[1]))
            else:
                r.size = (int(it.image.shape[0]), int(it.image.shape[1]))
            if self.rle:
                r.rle_table, r.rle_data = rtab[off:off + nb], rdat
            res.append(r)
        out.odd = {}
        release = lambda o=out: self.free_out.put(o)
        sink(res, release)

    @torch.no_grad()
    def run(self, batches: Iterable[List[WorkItem]], sink: Callable[[List[TileResult], Callable[[], None]], None]) -> int:
        """Drives `batches` (lists of <= batch WorkItems) through the pipeline.  `sink(results, release)` is called on the
        calling thread, in order, once a batch's class maps and areas are on the host; the arrays are views of a pinned
        ring buffer, so the sink (or whatever it hands them to) must call `release()` when it is done with them.
        Returns the number of tiles processed."""
        cur = torch.cuda.current_stream(self.dev)
        for st in (self.s_h2d, self.s_enc, self.s_dec):
            st.wait_stream(cur)
        n_tiles = 0
        staged = None        # batch k+1: staged, not yet encoded
        encoded = None       # batch k  : encoder issued
        decoding = None      # batch k-1: decoder issued, results not yet handed over
        it = iter(batches)
        k = 0
        while True:
            nxt = next(it, None)
            if nxt is not None:
                if len(nxt) < 1 or len(nxt) > self.batch:
                    raise ValueError(f"a batch must hold 1..{self.batch} work items")
                b = k & 1
                tiles, same, offs = self._stage(b, nxt)
                self._encode(b, tiles, same)
                staged = (b, nxt, tiles, offs)
                n_tiles += len(nxt)
                k += 1
            if encoded is not None:
                b, items, tiles, offs = encoded
                out = self.free_out.get()                              # blocks while the writers hold every buffer
                self._decode(b, items, tiles, offs, out)
                if decoding is not None:
                    self._finish(decoding, sink)
                decoding = (items, offs, out, b)
            encoded, staged = staged, None
            if nxt is None and encoded is None:
                break
        if decoding is not None:
            self._finish(decoding, sink)
        for st in (self.s_h2d, self.s_enc, self.s_dec):
            cur.wait_stream(st)
        return n_tiles


def batched(items: Iterable[WorkItem], batch: int) -> Iterator[List[WorkItem]]:
    buf: List[WorkItem] = []
    for it in items:
        buf.append(it)
        if len(buf) == batch:
            yield buf
            buf = []
    if buf:
        yield buf


class InstancePipeline(TilePipeline):
    """The instance drivers' recipe (main_sam_rhbox_mask_instance.py:125-168 / main_sam_rbox_mask_instance.py:125-164)
    on the same three-stream pipeline.  ``multimask=True`` is BASELINE.json configs[3]; the reference's own instance scripts all pass
    ``multimask_output=False`` (main_sam_rhbox_mask_instance.py:168, main_sam_rbox_mask_instance.py:164,
    main_sam_hbox_mask_instance.py:165): ``multimask=False`` is their configuration.  Annotations are
    rotated boxes [n, 4, 2]; ``prompt="box"`` feeds the enclosing hbox (min / max of the corners, :125-130) through
    ``apply_boxes_torch``, ``prompt="rbox_mask"`` rasterises the rbox into a +-1000 mask prompt on the GPU
    (``transforms.rbox_mask_prompts``).  Of the three masks per object the one with the highest predicted IoU is kept
    (SAM's own selection rule, `samrs_select_best` on the device); per object the host receives its area and quality (and with
    ``rle=True`` its COCO RLE), the kept masks stay in HBM (``last_masks``) unless keep_masks.  No class map is painted:
    ``TileResult.seg_mask`` is None and the class statistics stay zero (the instance drivers write neither)."""

    BOX_WIDTH = 8          # four (x, y) corners

    def __init__(self, sam, n_classes: int, prompt: str = "box", multimask: bool = True, fill_rule: str = "auto", **kw):
        """prompt: "box" / "rbox_mask" (annotations = rotated boxes [n, 4, 2]) or "point"
        (main_sam_hbox_mask_instance.py:160-165: annotations = one foreground point [n, 2] per object, handed to the prompt
        encoder AS IS -- the reference does not run them through apply_coords -- labels all 1, no box, no mask;
        that driver uses multimask_output=False: pass multimask=False)."""
        if prompt not in ("box", "rbox_mask", "point"):
            raise ValueError("prompt must be 'box', 'rbox_mask' or 'point'")
        if prompt == "point":
            self.BOX_WIDTH = 2
        from . import transforms
        self.fill_rule = transforms.resolve_fill_rule(fill_rule)       # prompt="rbox_mask": the cv2.fillPoly span rule to reproduce
        super().__init__(sam, n_classes, precision=kw.pop("precision", "auto"), _multimask=bool(multimask), **kw)
        self.prompt, self.multimask = prompt, bool(multimask)
        self.qual_dev = [torch.zeros(self.batch, self.max_boxes, dtype=torch.float32, device=self.dev) for _ in range(2)]
        for _ in range(self.free_out.qsize()):
            o = self.free_out.get()
            o.quality = torch.empty(self.batch, self.max_boxes, dtype=torch.float32).pin_memory()
            self.free_out.put(o)
        self.last_masks = None

    def _decode_tile(self, b, i, tile, hw, off, nb, out) -> None:
        from . import transforms
        eng, (H, W) = self.eng, hw
        in_size = (int(tile.shape[0]), int(tile.shape[1]))
        slot, mm = b * self.batch + i, self.multimask
        kept = []
        for s, e in box_chunks(nb, self.box_batch):
            ann = self.dev_box[b][off + s:off + e]
            if self.prompt == "box":
                polys = ann.view(-1, 4, 2)
                hb = torch.cat([polys.amin(1), polys.amax(1)], dim=1)                                   # :125-130
                m, q, _ = eng.predict(slot, self._input_frame_boxes(hb, (H, W), in_size), None, None, None, mm, False, in_size, (H, W))
            elif self.prompt == "rbox_mask":
                pr = transforms.rbox_mask_prompts_device(ann.view(-1, 4, 2), (H, W), self.side, device=self.dev, fill_rule=self.fill_rule)
                m, q, _ = eng.predict(slot, None, None, None, pr[:, None], mm, False, in_size, (H, W))
            else:
                pl = torch.ones(e - s, 1, dtype=torch.int32, device=self.dev)
                m, q, _ = eng.predict(slot, None, ann.view(-1, 1, 2), pl, None, mm, False, in_size, (H, W))
            # best of the C masks by predicted IoU, its quality and area: one pass on the device, straight into the tables
            mk, _, _ = eng.select_best(m, q, None, self.qual_dev[b][i, s:e], self.area_dev[b][i, s:e])
            if self.rle:
                eng.rle_encode(mk, self.rle_dev[b], self.rle_cur[b], self.rle_tab[b][off + s:off + e])
            kept.append(mk)
        self.last_masks = kept[-1].view(torch.bool) if kept else None
        out.masks[i] = torch.cat(kept).cpu() if (self.keep_masks and kept) else None
        out.odd[i] = None                      # instance pipelines paint no class map

    def _extra_outputs(self, b: int, out: _OutBuf) -> None:
        out.quality.copy_(self.qual_dev[b], non_blocking=True)
