#!/usr/bin/env python3
"""Decoder-only timing: one ViT-H tile encoded once, then `predict` (32 boxes) in a loop on one stream.
Run under `rocprofv3 --kernel-trace --stats` for an uncontended per-kernel split of the prompt path."""
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
    cfg = synth.CONFIGS[model]
    sd = synth.make_state_dict(cfg, 0)
    sam = samrs_amd.sam_model_registry[model](state_dict=sd, precision="f16", max_images=1, max_prompts=32, max_points=1).to(dev)
    eng = sam.engine
    tile = torch.from_numpy(synth.make_noise_image(0))[None].to(dev)
    eng.set_images(tile, 0)
    b, _ = synth.make_boxes(0, 32)
    boxes = torch.from_numpy(b).to(dev)
    splits = [int(v) for v in os.environ["DEC_SPLITS"].split(",")] if os.environ.get("DEC_SPLITS") else [eng.get_option("split")]
    ufs = [int(v) for v in os.environ["DEC_UF"].split(",")] if os.environ.get("DEC_UF") else [eng.get_option("upscaler_fused")]
    for split, uf in [(s_, u_) for s_ in splits for u_ in ufs]:
        eng.set_option("split", split)
        eng.set_option("upscaler_fused", uf)
        for _ in range(3):
            eng.predict(0, boxes, None, None, None, False, False, (1024, 1024), (1024, 1024))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.predict(0, boxes, None, None, None, False, False, (1024, 1024), (1024, 1024))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"predict(32 boxes) {model} split={split} upscaler_fused={uf}: {dt * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
