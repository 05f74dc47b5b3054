#!/usr/bin/env python3
"""End-to-end run of the generation CLI's loop (samrs_amd.generate.run) on synthetic tiles written to disk as PNG:
image decode (reader pool) -> TilePipeline (encoder / decoder / paint / device RLE) -> gray + color PNG + ins/*.pkl (writer pool).
What `python -m samrs_amd.generate` costs per image when the host side is included.  usage: generate_bench.py [n_tiles] [model]
env: READERS, WRITERS, PNG_LEVEL, OUT_DEPTH; NOWRITE=1 / NOREAD=1 take one side out (which side holds the loop back)."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import generate, synth  # noqa: E402


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    model = sys.argv[2] if len(sys.argv) > 2 else "vit_h"
    root = tempfile.mkdtemp(prefix="samrs_gen_")
    img_dir, out_dir = os.path.join(root, "img"), os.path.join(root, "out")
    os.makedirs(img_dir)
    base = [synth.make_image(i) for i in range(8)]                      # 8 distinct blob tiles, re-used with a roll
    ann = {}
    t0 = time.perf_counter()
    from concurrent.futures import ThreadPoolExecutor
    from samrs_amd import tile_io

    def put(i):
        tile_io.write_rgb(os.path.join(img_dir, f"T{i:05d}.png"), np.ascontiguousarray(np.roll(base[i % 8], 37 * (i // 8), axis=1)), 1)
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(put, range(n)))
    for i in range(n):
        b, l = synth.make_boxes(i, 32)
        ann[f"T{i:05d}"] = {"boxes": b.tolist(), "labels": l.tolist()}
    with open(os.path.join(root, "boxes.json"), "w") as f:
        json.dump(ann, f)
    print(f"wrote {n} tiles in {time.perf_counter() - t0:.1f}s", flush=True)
    # the rate is that of generate.run's loop (first batch read -> last file on disk); model build is outside it
    def one(no_rle, stems_n):
        sub = {k: ann[k] for k in sorted(ann)[:stems_n]}
        with open(os.path.join(root, "boxes_sub.json"), "w") as f:
            json.dump(sub, f)
        shutil.rmtree(out_dir, ignore_errors=True)
        ns = argparse.Namespace(images=img_dir, boxes=os.path.join(root, "boxes_sub.json"), out=out_dir, model=model, checkpoint=None,
                                precision="f16", classes=None, n_classes=18, palette=None, box_batch=64, no_rle=no_rle, batch=8,
                                schedule="static", readers=int(os.environ.get("READERS", "8")), writers=int(os.environ.get("WRITERS", "16")),
                                resume=False, rle_buffer_mb=512, timing=True,
                                png_level=int(os.environ.get("PNG_LEVEL", "6")), out_depth=int(os.environ.get("OUT_DEPTH", "4")))
        stats = generate.run(ns)
        return stats.get("timing")
    for no_rle in (False, True):
        one(no_rle, 16)                                     # first-call costs (kernel load, allocator) stay out of the measured run
        t = one(no_rle, n)
        per = {k: 1e3 * v / t["images"] for k, v in sorted(t["stage_thread_seconds"].items()) if not k.startswith("loop.")}
        print(f"generate.run {model} {'--no-rle' if no_rle else 'with RLE'}: {t['images']} tiles, loop {t['loop_seconds']:.2f} s = "
              f"{t['images'] / t['loop_seconds']:.1f} images/s; thread ms per image: "
              + ", ".join(f"{k} {v:.1f}" for k, v in per.items()), flush=True)
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
