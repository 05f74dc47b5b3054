"""How well-posed is "bit-identical class map" on this data?  Runs the fp32 oracle (pinned to the reference) and the SAME
arithmetic in float64 on the C2 fixture inputs and counts the pixels whose thresholded decision differs: the
reference's own fp32 rounding noise, a floor no implementation can go below.
    python -m oracle.ref_noise_floor [vit_b|vit_h]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import synth                              # noqa: E402
from oracle import sam_oracle as so                     # noqa: E402
from oracle.make_golden import extended_inputs          # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
cfg = synth.CONFIGS[name]
sd32 = synth.make_state_dict(cfg, 0, logit_scale=synth.MARGIN_LOGIT_SCALE)
inp = extended_inputs()
img = synth.make_image(0)
hw = (1024, 1024)
tb = so.apply_boxes(torch.from_numpy(inp["boxes"]), hw)


def run(dtype):
    torch.set_default_dtype(dtype)          # the oracle builds its coordinate grids in the default dtype
    sd = {k: v.to(dtype) for k, v in sd32.items()}
    pred = so.OraclePredictor(sd, cfg)
    x = so.preprocess(img, cfg.img_size).to(dtype)
    with torch.no_grad():
        pred.features = so.image_encoder(sd, cfg, x)
        pred.is_image_set, pred.input_size, pred.original_size = True, hw, hw
        out = pred.predict_torch(None, None, tb.to(dtype), None, multimask_output=False, return_logits=True)
    return out[0][:, 0], out[2]


l32, low32 = run(torch.float32)
l64, low64 = run(torch.float64)
m32, m64 = l32 > 0, l64 > 0
flips = (m32 != m64).flatten(1).sum(1)
seg32, _ = so.paint_semantic(m32.numpy(), inp["labels"], hw)
seg64, _ = so.paint_semantic(m64.numpy(), inp["labels"], hw)
rel = ((low32.double() - low64).norm() / low64.norm()).item()
print(f"{name}: fp32 vs fp64 low-res logits rel L2 {rel:.2e}; flipped mask pixels per box: total {int(flips.sum())}, max {int(flips.max())}; "
      f"class-map pixels that differ: {int((seg32 != seg64).sum())} of {seg32.size}")
