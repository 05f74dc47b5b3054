#!/bin/bash
# What the outlier-column extension costs in the PRODUCT loop: bench.py --weights <w>, secondary legs off, arms alternated twice on one box
# (profiles/r06_outlier_loop_cost.txt is this script's output + a reading).
cd "${GRAFT_REPO_ROOT:-.}"
BQ="--steps 20 --warmup 3 --no-cpu-baseline --no-alt-dtype --no-pcie-leg --no-cli-leg --no-fast-leg --no-rle-leg"
row() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-64s %8.3f  %8.3f   %.4f' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" "$1"; }
echo "weights                                                          images/s   ms/step   lin1 in situ (ms)"
for i in 1 2; do
  for w in normal heavy_tailed heavy_tailed_every_block; do python bench.py $BQ --weights $w 2>/dev/null | row "$w"; done
done
SAMRS_OUTLIER_COLS=0 python bench.py $BQ --weights heavy_tailed_every_block 2>/dev/null | row "heavy_tailed_every_block, SAMRS_OUTLIER_COLS=0 (extension off)"
