#!/bin/bash
# Samples socket power / sclk / temperature (rocm-smi) while bench.py runs.  usage: power_trace.sh [bench args]
mkdir -p gpurun_out
python bench.py --steps ${STEPS:-250} --warmup 2 --no-cpu-baseline --no-alt-dtype --no-pcie-leg "$@" > gpurun_out/power_bench.log 2>&1 &
pid=$!
for i in $(seq 1 200); do
  kill -0 $pid 2>/dev/null || break
  rocm-smi --showpower --showclocks --showtemp --showperflevel --json 2>/dev/null | python -c "
import sys, json
try:
    d = json.load(sys.stdin)['card0']
    keys = [k for k in d if any(t in k.lower() for t in ('power', 'sclk', 'mclk', 'junction', 'hotspot', 'edge'))]
    print({k: d[k] for k in keys})
except Exception as e:
    print('parse error', e)
"
done
wait $pid
tail -1 gpurun_out/power_bench.log | cut -c1-200
rocm-smi --showmaxpower --showpowercap 2>/dev/null | grep -i "power" | head -5
