#!/usr/bin/env python
"""8-tile ViT-H encoder passes on synth.heavy_tailed weights with outlier channels in EVERY block (the worst case for the
outlier-column extension: all 128 block-GEMM launches carry it), for rocprofv3:

    rocprofv3 --kernel-trace --stats -d gpurun_out/oc_prof -- python tools/outlier_profile.py [mask]

mask = the engine's "outlier_cols" option (7 default; 0 = off, for the A/B)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import samrs_amd                                        # noqa: E402
from samrs_amd import synth                             # noqa: E402

mask = int(sys.argv[1]) if len(sys.argv) > 1 else 7
cfg = synth.CONFIGS["vit_h"]
sd = synth.heavy_tailed(synth.make_state_dict(cfg, 0), cfg, 0, hidden_scale=3e3, v_scale=3e3, gamma_scale=30.0, blocks=list(range(cfg.depth)))
sam = samrs_amd.sam_model_registry["vit_h"](state_dict=sd, precision="f16", max_prompts=8, max_points=1, max_images=8,
                                            options={"split": 15}).to("cuda")
eng = sam.engine
eng.set_option("outlier_cols", mask)
tiles = torch.as_tensor(np.stack([synth.make_noise_image(i) for i in range(8)]), device="cuda").contiguous()
for _ in range(4):
    eng.set_images(tiles)
torch.cuda.synchronize()
print(f"outlier_cols={mask}: {eng.get_option('outlier_blocks')} blocks, {eng.get_option('outlier_columns')} columns")
