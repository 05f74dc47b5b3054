#!/bin/bash
# A/B of one environment knob on the same box: alternates `bench.py` runs with VAR=a and VAR=b.  usage: ab_env.sh VAR a b [rounds]
# BENCH_EXTRA='--workload c3' etc. selects another workload
var=$1; a=$2; b=$3; rounds=${4:-2}
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for v in $a $b; do
    env $var=$v timeout 600 python bench.py --steps ${BENCH_STEPS:-6} --warmup 2 --no-cpu-baseline --no-alt-dtype --no-pcie-leg --no-cli-leg --no-fast-leg ${BENCH_EXTRA:-} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('$var=$v', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'])
"
  done
done
