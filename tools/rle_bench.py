#!/usr/bin/env python3
"""Device RLE timing: the masks of one 32-box predict on a synthetic ViT-H tile (noise-like on random weights: ~135 k runs per
mask) and a smooth-blob batch (real-mask-like: a few thousand runs), encoded in a loop.  Run under rocprofv3 --kernel-trace --stats
for the per-kernel split."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import samrs_amd  # noqa: E402
from samrs_amd import synth  # noqa: E402


def main() -> None:
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    model = sys.argv[2] if len(sys.argv) > 2 else "vit_h"
    dev = torch.device("cuda", 0)
    sam = samrs_amd.sam_model_registry[model](precision="f16", max_images=1, max_prompts=32, max_points=1).to(dev)
    eng = sam.engine
    eng.set_images(torch.from_numpy(synth.make_noise_image(0))[None].to(dev), 0)
    boxes = torch.from_numpy(synth.make_boxes(0, 32)[0]).to(dev)
    noisy, _, _ = eng.predict(0, boxes, None, None, None, False, False, (1024, 1024), (1024, 1024))
    yy, xx = torch.meshgrid(torch.arange(1024, device=dev), torch.arange(1024, device=dev), indexing="ij")
    smooth = torch.stack([((yy - 300 - 10 * i) ** 2 + (xx - 500 + 7 * i) ** 2) < (100 + 9 * i) ** 2 for i in range(32)])
    out = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    cur = torch.zeros(1, dtype=torch.int64, device=dev)
    tab = torch.zeros(32, 3, dtype=torch.int64, device=dev)
    for name, m in (("predicted (noise-like)", noisy[:, 0]), ("smooth blobs", smooth)):
        for _ in range(3):
            cur.zero_(); eng.rle_encode(m, out, cur, tab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            cur.zero_(); eng.rle_encode(m, out, cur, tab)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"rle_encode(32 x 1024^2, {name}): {dt * 1e6:.1f} us, {int(tab[:, 1].sum()) / 32:.0f} bytes / mask, {int(tab[:, 2].sum()) / 32:.0f} counts / mask")


if __name__ == "__main__":
    main()
