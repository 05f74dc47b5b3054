#!/bin/bash
# A/B of library builds on the same box: rounds x (bench.py with SAMRS_LIB_PATH = each argument).  usage: ab_libs.sh rounds lib...
rounds=$1; shift
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    SAMRS_LIB_PATH=$v timeout 600 python bench.py --steps ${BENCH_STEPS:-6} --warmup 2 --no-cpu-baseline --no-alt-dtype --no-pcie-leg --no-cli-leg --no-fast-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'])
"
  done
done
