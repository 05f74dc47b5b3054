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
(main_sam_rhbox_semantic.py:125-130).  One process per GPU; the stems are sorted and handed out in chunks of
``--batch`` (statically, or from a shared counter with ``--schedule dynamic``); every rank runs
``driver.TilePipeline`` (batched encoder, decoder / painting / transfers overlapped on separate HIP streams,
reader + writer thread pools); no collective on the data path, one int64 all-reduce for the class statistics
and one all-gather for the mask-size list (statistic.py:34-53) at the end.
"""
from __future__ import annotations

import argparse
import json
import os
import pickle
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import driver, rle, tile_io


def default_palette(n_classes: int) -> np.ndarray:
    """Deterministic colour per class id (the reference's MAPPING tables, Generate Dataset/mapping.py,
    are dataset-specific constants; pass --palette to reproduce them exactly)."""
    rng = np.random.default_rng(12345)
    return rng.integers(0, 256, size=(n_classes, 3), dtype=np.uint8)


class StageClock:
    """Wall-clock per host stage, summed over threads (--timing): which side of the GPU loop the time goes to."""

    def __init__(self) -> None:
        self.t: Dict[str, float] = {}
        self.n: Dict[str, int] = {}

    def add(self, stage: str, t0: float) -> float:
        import time
        now = time.perf_counter()
        self.t[stage] = self.t.get(stage, 0.0) + (now - t0)        # dict updates under the GIL: good enough for a report
        self.n[stage] = self.n.get(stage, 0) + 1
        return now

    def report(self, n_images: int, wall: float) -> str:
        rows = [f"{k}: {1e3 * v / max(n_images, 1):.1f} ms/image (thread time)" for k, v in sorted(self.t.items())]
        return f"{n_images} images in {wall:.2f} s = {n_images / max(wall, 1e-9):.1f} images/s; " + "; ".join(rows)


def write_outputs(out_dir: str, stem: str, seg: np.ndarray, masks: Optional[np.ndarray], boxes: np.ndarray,
                  labels: np.ndarray, areas: np.ndarray, palette: np.ndarray, class_names: Sequence[str],
                  rles: Optional[Sequence[dict]] = None, clock: Optional[StageClock] = None, png_level: int = tile_io.LEVEL_LABELS) -> None:
    """`rles`: the per-instance COCO RLE dicts when they were encoded on the device (driver.TileResult.rle); otherwise they are
    encoded here from `masks` (host restatement), or left out when both are None (--no-rle)."""
    import time
    t0 = time.perf_counter()
    for sub in ("gray", "color", "ins"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    # native PNG encoders (libsamrs_io.so): no interpreter lock held while a tile is compressed; the colour image is the class
    # map seen through a 256-entry table (:163 white background, :199 palette colour per painted box)
    # png_level: LEVEL_LABELS (default) = the label-aware encoder for both images -- the colour image IS the class map seen through
    # the palette, so its deflate tokens are derived from the 1-byte map (4x less CPU than zlib on the RGB bytes); 1..9 = zlib for
    # color/*.png (and the run-length preset for gray/*.png), kept for A/B runs.  The decoded pixels never depend on it.
    if png_level == tile_io.LEVEL_LABELS:
        # one parse of the label runs, two deflate streams (:212-215)
        tile_io.write_label_pair(os.path.join(out_dir, "gray", stem + ".png"), os.path.join(out_dir, "color", stem + ".png"), seg,
                                 tile_io.class_lut(palette))
        if clock: t0 = clock.add("write.gray_color_png", t0)
    else:
        tile_io.write_gray(os.path.join(out_dir, "gray", stem + ".png"), seg, tile_io.LEVEL_RUNS)                          # :212,214
        if clock: t0 = clock.add("write.gray_png", t0)
        tile_io.write_lut_rgb(os.path.join(out_dir, "color", stem + ".png"), seg, tile_io.class_lut(palette), png_level)   # :213,215
        if clock: t0 = clock.add("write.color_png", t0)
    info = []
    for j in range(len(labels)):                                                            # :200-206
        entry = {"bbox": boxes[j], "category": class_names[int(labels[j])], "label": int(labels[j]), "size": int(areas[j])}
        if rles is not None:
            entry["mask"] = rles[j]
        elif masks is not None:
            entry["mask"] = rle.encode(masks[j])
        info.append(entry)
    # the pickle goes last and through a rename: --resume takes its presence as "this image is complete"
    tmp = os.path.join(out_dir, "ins", f"{stem}.pkl.tmp.{os.getpid()}")     # per process: two ranks can never share a tmp file
    with open(tmp, "wb") as f:
        pickle.dump(info, f)                                                                # :216
    os.replace(tmp, os.path.join(out_dir, "ins", stem + ".pkl"))
    if clock: clock.add("write.pickle", t0)


def host_cpu_budget(local_world: Optional[int] = None) -> float:
    """CPUs THIS rank may keep busy: the container's share (cgroup v2 ``cpu.max``, else the affinity mask) divided by the ranks on
    this node (``LOCAL_WORLD_SIZE``, set by torchrun).  On the pool's GPU boxes a container is a 16-CPU slice of 256 hardware
    threads: eight ranks get two CPUs each, and thread pools sized for the whole node would only add context switches."""
    cpus = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cpus = min(cpus, int(quota) / int(period))
    except (OSError, ValueError):
        pass
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    return max(1.0, cpus / max(1, local_world))


def io_threads(readers: int = 0, writers: int = 0, local_world: Optional[int] = None) -> Tuple[int, int]:
    """(reader threads, writer threads); 0 = sized from :func:`host_cpu_budget`.  Per image the host pays ~6 ms of decode and
    ~14 ms of encode + pickle, so a third of the budget reads and the rest writes; the caps (8 / 16) are what
    one rank at 140 images/s can use, the floor (2 / 2) keeps decode and encode overlapped with each other and with the GPU loop."""
    budget = int(host_cpu_budget(local_world))
    r = readers if readers > 0 else min(8, max(2, budget // 3))
    w = writers if writers > 0 else min(16, max(2, budget - min(8, max(2, budget // 3))))
    return r, w


HOST_MS_PER_IMAGE = 20.7       # host thread time per image of this driver: read.decode 5.9 + write.gray_color_png 11.7 + pickle 2.7 + rle 0.4 (BENCH_r04 cli_inclusive)


def host_bound_warning(readers: int, writers: int, gpu_images_per_s: float = 140.0, local_world: Optional[int] = None) -> Optional[str]:
    """A sentence for the operator when this rank's share of the host cannot keep up with its GPU (VERDICT r04 "what's weak" 11):
    a rank that produces ``gpu_images_per_s`` needs ``gpu_images_per_s x 20.7 ms`` = ~2.9 busy CPUs for decode + encode + pickle; on a
    16-CPU container shared by eight ranks it has two.  None when the budget suffices and the pools fit it."""
    budget = host_cpu_budget(local_world)
    need = gpu_images_per_s * HOST_MS_PER_IMAGE * 1e-3
    msgs = []
    if readers + writers > budget + 1e-9 and budget < need:
        msgs.append(f"{readers} reader + {writers} writer threads on a share of {budget:.1f} CPUs")
    if budget < need:
        msgs.append(f"this rank is HOST-bound at ~{budget / (HOST_MS_PER_IMAGE * 1e-3):.0f} images/s (its GPU loop does ~{gpu_images_per_s:.0f}): "
                    f"decode + PNG encode + pickle cost {HOST_MS_PER_IMAGE:.1f} ms of host thread time per image = {need:.1f} busy CPUs; "
                    "give the job more CPUs per rank, or run fewer ranks per container")
    return "; ".join(msgs) if msgs else None


def outputs_exist(out_dir: str, stem: str) -> bool:
    """All three files of an image are on disk (main_sam_hbox_semantic.py:214-216 writes gray, color, then ins)."""
    return all(os.path.exists(os.path.join(out_dir, sub, stem + ext)) for sub, ext in (("gray", ".png"), ("color", ".png"), ("ins", ".pkl")))


_RUN_SEQ = [0]          # run() calls of this process (part of the work queue's store key)


def resumed_statistics(out_dir: str, stems: List[str], n_classes: int):
    """(class_pixel_num, class_instance_num, mask sizes) of images completed by an earlier run, from their ins/*.pkl
    (`Generate Dataset/statistic.py:12-21,44-49`); a label outside 0..n_classes-1 raises, naming the file."""
    pix, ins, sizes = np.zeros(n_classes, np.int64), np.zeros(n_classes, np.int64), []
    for stem in stems:
        path = os.path.join(out_dir, "ins", stem + ".pkl")
        with open(path, "rb") as f:
            for entry in pickle.load(f):
                label, size = int(entry["label"]), int(entry["size"])
                if not 0 <= label < n_classes:
                    raise ValueError(f"--resume: {path} holds label {label}, outside 0..{n_classes - 1}: it was written with "
                                     "another class list (use a fresh --out or the same --classes / --n-classes)")
                if size > 0:                                                                         # statistic.py:18
                    pix[label] += size
                    ins[label] += 1
                    sizes.append(size)
    return pix, ins, sizes


def default_split_options(split, environ=None):
    """Engine options of this driver's model: an explicit --split wins; otherwise SAMRS_SPLIT (read by samrs_create) stands
    and NO option is passed; only when neither is given the single-mask default 15 is set before the weights are finalized."""
    environ = os.environ if environ is None else environ
    if split is not None:
        return {"split": int(split)}
    if environ.get("SAMRS_SPLIT", "") != "":
        return None
    return {"split": 15}


def run(args) -> Dict[str, List[int]]:
    import torch.distributed as dist
    from concurrent.futures import ThreadPoolExecutor
    import samrs_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # SAMRS_SHARE_GPU=1: every rank on cuda:0, collectives over gloo -- the N-rank control flow of this driver (shards, --resume
    # list broadcast, statistics all-reduce, mask-size all-gather, per-rank logs) on a ONE-GPU box; RCCL itself needs a multi-GPU
    # node.  The ranks' engines are separate handles with their own streams and slots: sharing a device changes nothing they compute.
    share = os.environ.get("SAMRS_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    ann = json.load(open(args.boxes))
    names = [l.strip() for l in open(args.classes)] if args.classes else [str(i) for i in range(args.n_classes)]
    n_classes = len(names)
    palette = np.load(args.palette) if args.palette else default_palette(n_classes)
    # 255 is the "unlabeled" value of gray/*.png (main_sam_hbox_semantic.py:162) and stays white in color/*.png (:163)
    if n_classes > 255 or len(palette) > 255:
        raise ValueError(f"{n_classes} classes / {len(palette)} palette rows: class ids must stay below 255 (255 = unlabeled)")
    bad = [s for s, a in ann.items() if len(a["labels"]) and not (0 <= min(a["labels"]) and max(a["labels"]) < n_classes)]
    if bad:
        raise ValueError(f"labels outside 0..{n_classes - 1} in the annotations of {bad[:3]}{' ...' if len(bad) > 3 else ''}")
    batch = getattr(args, "batch", 8)
    # --split: an explicit operand-split mode (engine option "split"); default: the pipeline picks by output contract -- this
    # driver asks for single masks only, which hold IoU >= 0.9995 with the block GEMMs at the 1x rate (split 15)
    # no --split: 15 explicitly, BEFORE the weights are finalized, so that a ViT-H engine does not allocate the lo copies of its
    # qkv / proj weights and the Ylo / AOlo workspaces (~0.75 GB of HBM this driver would never touch)
    # An operator's SAMRS_SPLIT is never overridden (samrs_hip.h option docs): the default is injected only when
    # neither --split nor the environment variable names a mode (round-4 advisor finding: options are applied AFTER samrs_create
    # has read the environment, so an unconditional 15 silently replaced SAMRS_SPLIT=63 / 79).
    opts = default_split_options(getattr(args, "split", None))
    sam = samrs_amd.sam_model_registry[args.model](checkpoint=args.checkpoint, precision=args.precision, options=opts,
                                                   max_images=2 * batch, max_prompts=args.box_batch).to(f"cuda:{local}")
    exts = (".png", ".jpg", ".jpeg", ".tif", ".bmp")
    files = {os.path.splitext(f)[0]: f for f in os.listdir(args.images) if f.lower().endswith(exts)}
    stems = sorted(s for s in files if s in ann and len(ann[s]["boxes"]) > 0)                       # :126-129
    n_all = len(stems)
    done_before: List[str] = []
    if getattr(args, "resume", False):
        # restart = re-run what is missing.  The output directory is listed ONCE, on rank 0, and the result is
        # broadcast: a rank that finished loading its weights earlier may already be writing files while a slower one would
        # still be listing, and ranks indexing different todo lists skip or repeat images (round-3 advisor finding).
        todo = [s for s in stems if not outputs_exist(args.out, s)] if rank == 0 else None
        todo = driver.agree_on_list(todo if todo is not None else [])
        keep = set(todo)
        done_before = [s for s in stems if s not in keep]
        stems = todo
        if rank == 0:
            print(f"[rank 0] --resume: {len(done_before)} of {n_all} images already complete", flush=True)
    # a resumed run's statistics cover the WHOLE output directory (see the end of this function): the earlier run's ins/*.pkl
    # are read HERE, before any GPU work, so that pickles written with another class list fail the run early and by name
    # (round-4 advisor finding: the labels indexed the counters unchecked, after every tile had been processed and written)
    old_pix, old_ins, old_sizes = resumed_statistics(args.out, done_before[rank::world], n_classes)
    max_boxes = max([len(ann[s]["labels"]) for s in stems] + [1])
    # per-instance RLE (main_sam_hbox_semantic.py:201-202) is encoded on the device; the full masks never cross PCIe
    pipe = driver.TilePipeline(sam, n_classes, batch=batch, box_batch=args.box_batch, rle=not args.no_rle,
                               rle_buffer_mb=getattr(args, "rle_buffer_mb", 256), max_boxes=max_boxes,
                               out_depth=getattr(args, "out_depth", 4))
    # rank r takes chunks of `batch` consecutive stems: statically (r, r + world, ...) or from the shared counter (whose
    # store key must be unique per work list: a second run() in the same process group must not find a spent counter)
    import zlib
    wq_name = "samrs_wq/%08x/%d" % (zlib.crc32("\n".join(stems).encode()), _RUN_SEQ[0])
    _RUN_SEQ[0] += 1
    wq = driver.WorkQueue(len(stems), chunk=batch, rank=rank, world=world, mode=getattr(args, "schedule", "static"), name=wq_name)

    import time
    tile_io.load_library()                                  # fail here, not on a worker thread, when libsamrs_io.so is missing
    n_readers, n_writers = io_threads(getattr(args, "readers", 0) or 0, getattr(args, "writers", 0) or 0)
    warn = host_bound_warning(n_readers, n_writers)
    if warn:
        import sys
        print(f"[rank {rank}] warning: {warn}", file=sys.stderr, flush=True)
    png_level = getattr(args, "png_level", tile_io.LEVEL_LABELS)
    clock = StageClock() if getattr(args, "timing", False) else None

    # Tiles of the native size are decoded straight into pinned buffers the pipeline can upload from (driver._stage: "caller-owned
    # pinned memory: straight H2D"), so a tile is written once by the decoder and read once by the DMA engine.  A buffer goes back
    # to the pool when its image's results reach the sink (its upload finished long before).  In flight at any time: two batches
    # on the reader side + three inside the pipeline; when the pool runs dry a tile simply takes the pageable path.
    import queue
    side = sam.image_encoder.img_size
    pool: "queue.SimpleQueue[torch.Tensor]" = queue.SimpleQueue()
    for _ in range(6 * batch):
        pool.put(torch.empty(side, side, 3, dtype=torch.uint8).pin_memory())
    loaned: Dict[str, torch.Tensor] = {}

    def load(stem: str) -> driver.WorkItem:
        t0 = time.perf_counter()
        path = os.path.join(args.images, files[stem])
        buf = None
        if path.lower().endswith(".png"):
            try:
                if tile_io.png_size(path) == (side, side):
                    buf = pool.get_nowait()
            except (tile_io.TileIOError, queue.Empty):
                buf = None
        if buf is not None:
            tile_io.read_rgb(path, out=buf.numpy())                                                  # :114
            loaned[stem] = buf
            img = buf
        else:
            img = tile_io.read_rgb(path)
        if clock: clock.add("read.decode", t0)
        return driver.WorkItem(stem, img, np.asarray(ann[stem]["boxes"], dtype=np.float32),
                               np.asarray(ann[stem]["labels"], dtype=np.int64))

    def batches():
        # image decode of the NEXT batch runs on a helper thread while the GPU works on this one
        with ThreadPoolExecutor(max_workers=n_readers) as readers:
            nxt = None
            for s0, s1 in wq:
                fut = [readers.submit(load, st) for st in stems[s0:s1]]
                if nxt is not None:
                    yield collect(nxt)
                nxt = fut
            if nxt is not None:
                yield collect(nxt)

    def collect(futs):
        t0 = time.perf_counter()
        items = [f.result() for f in futs]
        if clock: clock.add("loop.wait_readers", t0)
        return items

    done = [0]
    sizes: List[int] = []
    writers = ThreadPoolExecutor(max_workers=n_writers)
    pending: List = []

    def reap(block: bool) -> None:
        """Re-raise the first failure of a writer job (disk full, bad path, pickle error): a run whose files did not reach the
        disk must not go on to write statistics and exit 0."""
        while pending and (block or pending[0].done()):
            pending.pop(0).result()

    def sink(results, release):
        # PNG encode + pickle on the writer pool, ONE JOB PER IMAGE (zlib releases the GIL; a job per batch would serialise
        # 8 images' worth of encoding behind one thread and the pipeline would stall on its output ring); the pinned ring
        # buffers go back when the last image of the batch is on disk
        import threading
        left = [len(results)]
        lock = threading.Lock()

        def job(r):
            try:
                t0 = time.perf_counter()
                rles = [r.rle(j) for j in range(len(r.labels))] if r.rle_table is not None else None
                if clock: clock.add("write.rle_dicts", t0)
                write_outputs(args.out, r.key, r.seg_mask, None, r.boxes, r.labels, r.areas, palette, names, rles, clock, png_level)
            finally:
                with lock:
                    left[0] -= 1
                    last = left[0] == 0
                if last:
                    release()
        if not results:
            release()
        pending.extend(writers.submit(job, r) for r in results)
        if run_log is not None:                                  # one JSON line per batch handed to the writers (per-run log)
            run_log.write(json.dumps({"t": round(time.perf_counter() - t_run, 4), "rank": rank, "images": [str(r.key) for r in results],
                                      "boxes": [int(len(r.labels)) for r in results], "done": done[0] + len(results)}) + "\n")
            run_log.flush()
        for r in results:
            buf = loaned.pop(r.key, None)
            if buf is not None:
                pool.put(buf)
        reap(block=False)
        for r in results:
            sizes.extend(int(a) for a in r.areas if a > 0)                                           # statistic.py:44-49
        done[0] += len(results)
        if rank == 0 and (done[0] // batch) % 50 == 0:
            print(f"[rank 0] {done[0]} images", flush=True)

    t_run = time.perf_counter()
    run_log = None
    if getattr(args, "log", None):
        os.makedirs(os.path.dirname(os.path.abspath(args.log)) or ".", exist_ok=True)
        run_log = open(f"{args.log}.rank{rank}" if world > 1 else args.log, "a")
        run_log.write(json.dumps({"t": 0.0, "rank": rank, "world": world, "model": args.model, "split": pipe.split_mode if pipe.split_mode is not None else sam.engine.get_option("split"),
                                  "images_total": n_all, "images_todo": len(stems), "batch": batch, "box_batch": args.box_batch}) + "\n")
    try:
        pipe.run(batches(), sink)
        if clock: clock.add("loop.pipe_run_total", t_run)
    finally:
        writers.shutdown(wait=True)
        if run_log is not None:
            run_log.close()
    reap(block=True)
    wall = time.perf_counter() - t_run
    if clock and rank == 0:
        import sys
        print("[rank 0] --timing: " + clock.report(done[0], wall), file=sys.stderr, flush=True)
    # a resumed run's statistics cover the WHOLE output directory, like `Generate Dataset/statistic.py:12-21,44-49` computes
    # them: the images completed by an earlier run contribute through their ins/*.pkl (label + size per instance), read by
    # the ranks in shards and merged by the same all-reduce / all-gather as this run's counters
    if done_before:
        pipe.class_pixels += torch.from_numpy(old_pix).to(pipe.class_pixels.device)
        pipe.class_instances += torch.from_numpy(old_ins).to(pipe.class_instances.device)
        sizes.extend(old_sizes)
    pix, ins = driver.reduce_statistics(pipe.class_pixels, pipe.class_instances)
    all_sizes = driver.gather_mask_sizes(sizes)
    stats = {"class_pixel_num": pix.cpu().tolist(), "class_instance_num": ins.cpu().tolist(),
             "mask_num": len(all_sizes)}                                                             # statistic.py:53
    if clock:
        stats["timing"] = {"images": done[0], "loop_seconds": wall, "stage_thread_seconds": dict(clock.t),   # this rank's loop
                           "readers": n_readers, "writers": n_writers, "cpu_budget": round(host_cpu_budget(), 1)}
    if rank == 0:
        os.makedirs(os.path.join(args.out, "statistic"), exist_ok=True)
        with open(os.path.join(args.out, "statistic", "class_stats.json"), "w") as f:       # statistic.py:28-31
            json.dump(stats, f)
        np.save(os.path.join(args.out, "statistic", "all_mask_size.npy"), np.asarray(all_sizes, dtype=np.int64))
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
    # boxes per predict call.  The reference uses 20 (main_sam_hbox_semantic.py:91, a memory workaround); the masks do not depend
    # on the chunking (tests/test_parity_gpu.py: predict(32) == predict(20) + predict(12) bit for bit) and a DOTA-shaped stream runs
    # 8.5 % faster at 64 (bench.py --workload c3 --box-batch 64: 137.6 vs 126.8 images/s), so the generation CLI defaults to 64
    ap.add_argument("--box-batch", type=int, default=64)
    ap.add_argument("--no-rle", action="store_true", help="skip per-instance RLE (only class maps + areas)")
    ap.add_argument("--rle-buffer-mb", type=int, default=256, help="device buffer for one batch's RLE strings (real masks need KBs each)")
    ap.add_argument("--resume", action="store_true", help="skip images whose gray / color / ins outputs already exist")
    ap.add_argument("--batch", type=int, default=8, help="tiles per encoder pass")
    ap.add_argument("--schedule", default="static", choices=["static", "dynamic"],
                    help="static: rank r takes chunks r, r+world, ...; dynamic: shared-counter work queue (long-tailed box counts)")
    ap.add_argument("--readers", type=int, default=0, help="image decode threads (0 = from this rank's share of the container's CPUs: "
                    "cgroup cpu.max / LOCAL_WORLD_SIZE, between 2 and 8)")
    ap.add_argument("--writers", type=int, default=0, help="PNG / pickle writer threads (0 = from the same budget, between 2 and 16)")
    ap.add_argument("--split", type=int, default=None, help="engine operand-split mode (15 = block GEMMs at the 1x f16 rate, the default of "
                    "this single-mask driver; 79 = multimask-grade; 31 / 63 = reference-grade; DESIGN.md section 2)")
    ap.add_argument("--png-level", type=int, default=tile_io.LEVEL_LABELS,
                    help="-2 (default): the label-aware PNG encoder for gray/ and color/ (deflate tokens derived from the class map); 1..9: zlib "
                         "level of color/*.png, run-length preset for gray/*.png.  The decoded pixels are the same either way")
    ap.add_argument("--out-depth", type=int, default=4, help="pinned output buffers (batches on loan to the writers at once)")
    # --run-log: the spelling to use under `python -m torch.distributed.run`, whose own parser stops at `--log` as an ambiguous
    # prefix of its --log-dir / --logs-specs before it hands the remaining arguments to this module
    ap.add_argument("--log", "--run-log", dest="log", default=None, help="append one JSON line per batch (time, image stems, box counts) "
                    "to this file (<file>.rank<r> with more than one rank); spell it --run-log under torchrun")
    ap.add_argument("--timing", action="store_true", help="print the host-side time per stage (decode, PNG encode, pickle, waits)")
    run(ap.parse_args(argv))


if __name__ == "__main__":
    main()
