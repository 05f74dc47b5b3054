"""GPU + host CPU: the statistical parity sample of oracle/parity_sample.py at ViT-H in the requested precision modes.
    python tools/parity_stats.py [--modes 15,79,63] [--c2 8] [--c4 4] [--out gpurun_out/parity_stats.json]
Prints the markdown table DESIGN.md 2 quotes and writes the raw per-mask records + summary as JSON."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import samrs_amd
from samrs_amd import synth
from oracle import parity_sample as ps, sam_oracle as so

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="15,79,63")
ap.add_argument("--c2", type=int, default=ps.N_C2_TILES)
ap.add_argument("--c4", type=int, default=ps.N_C4_TILES)
ap.add_argument("--no-odd", action="store_true")
ap.add_argument("--no-long", action="store_true")
ap.add_argument("--model", default="vit_h")
ap.add_argument("--create-split", type=int, default=127, help="the split mask the engine is created with (lo weight copies)")
ap.add_argument("--out", default="gpurun_out/parity_stats.json")
ap.add_argument("--weights", default="normal", choices=["normal", "heavy_tailed", "heavy_tailed_every_block"],
                help="round 6: the same sample on checkpoint-like weights (synth.heavy_tailed; the engine's outlier-column extension on)")
a = ap.parse_args()
modes = [int(m) if m.isdigit() else m for m in a.modes.split(",")]      # "79" or "79:4" = split 79 with lo_format 4
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
cfg = synth.CONFIGS[a.model]
sd = synth.make_state_dict(cfg, 0, logit_scale=synth.MARGIN_LOGIT_SCALE)
if a.weights != "normal":
    sd = synth.heavy_tailed(sd, cfg, 0, hidden_scale=3e3, v_scale=3e3, gamma_scale=30.0,
                            blocks=list(range(cfg.depth)) if a.weights == "heavy_tailed_every_block" else None)
sam = samrs_amd.sam_model_registry[a.model](state_dict=sd, precision="f16", max_prompts=32, max_points=1,
                                            options={"split": a.create_split}).to("cuda")
sam.engine.set_option("allow_reduced", 1)
print(f"weights: {a.weights}; outlier columns {sam.engine.get_option('outlier_columns')} in {sam.engine.get_option('outlier_blocks')} blocks "
      f"({sam.engine.get_option('outlier_dominant_blocks')} outlier-dominated), option outlier_cols = {sam.engine.get_option('outlier_cols')}")
pred = samrs_amd.SamPredictor(sam)
orc = so.OraclePredictor(sd, cfg)
t0 = time.time()
rec = ps.run(pred, orc, modes, ps.tiles(a.c2, a.c4, odd=not a.no_odd, long_tail=not a.no_long))
summ = ps.summarise(rec)
print(ps.table(summ))
print(f"total {time.time() - t0:.0f} s; tau = {ps.TAU_FRAC} x std(low-res logits); device {torch.cuda.get_device_name(0)}")
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
json.dump({"tau_frac": ps.TAU_FRAC, "model": a.model, "modes": modes, "summary": {str(k): v for k, v in summ.items()},
           "records": {str(k): v for k, v in rec.items()}}, open(a.out, "w"))
