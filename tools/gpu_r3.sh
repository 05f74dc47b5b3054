#!/bin/bash
# Round-3 GPU session helper (runs on the GPU box via gpurun): named steps, each with its own timeout and log.
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONDONTWRITEBYTECODE=1
run() { local name=$1; shift; local t=$1; shift
  echo "=== $name" | tee -a gpurun_out/summary.txt
  timeout "$t" "$@" > "gpurun_out/$name.log" 2>&1
  echo "rc=$?" | tee -a gpurun_out/summary.txt
  grep -E "passed|failed|error" "gpurun_out/$name.log" | tail -3 | tee -a gpurun_out/summary.txt; }
: > gpurun_out/summary.txt
PT="python -m pytest -m gpu -q -s -rA -p no:cacheprovider -x"
for step in "$@"; do
  case $step in
    rle)     run rle 600 $PT tests/test_rle_gpu.py ;;
    ksplit)  run ksplit 600 $PT tests/test_kernels_gpu.py -k "split or group_layernorm or upscale2" ;;
    psplit)  run psplit 900 $PT tests/test_parity_gpu.py -k "split or decoder_fused or decoder_alone or folded" ;;
    c2c4)    run c2c4 900 $PT tests/test_parity_gpu.py -k "c2_c4" ;;
    pipe)    run pipe 900 $PT tests/test_pipeline_gpu.py ;;
    gen)     run gen 600 $PT tests/test_parity_gpu.py -k "generation" ;;
    all)     run all 2400 python -m pytest tests -m gpu -q -rA -p no:cacheprovider ;;
    smoke)   run smoke 600 python -c "import __graft_entry__ as g; g.smoke()" ;;
    benchq)  run benchq 900 python bench.py --steps ${BENCH_STEPS:-8} --warmup 2 --no-cpu-baseline --no-alt-dtype --no-pcie-leg --no-cli-leg --no-fast-leg ${BENCH_EXTRA:-} ;;
    bench)   run bench 1500 python bench.py ${BENCH_ARGS:-} ;;
    c3)      run c3 600 python bench.py --workload c3 --steps ${BENCH_STEPS:-6} --warmup 2 --no-cpu-baseline --no-alt-dtype --no-pcie-leg ;;
    c4)      run c4 600 python bench.py --workload c4 --steps ${BENCH_STEPS:-6} --warmup 2 --no-cpu-baseline --no-alt-dtype --no-pcie-leg ;;
    decb)    run decb 300 python tools/dec_bench.py 20 ;;
    prof)    cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
             run prof 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r03 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-pcie-leg --no-rle-leg --no-cli-leg --no-fast-leg ;;
    *)       n=$((${n:-0} + 1)); echo "cmd$n: $step" >> gpurun_out/summary.txt; run "cmd$n" 1200 bash -c "$step" ;;
  esac
done
tail -n 6 gpurun_out/*.log 2>/dev/null | tail -150
cat gpurun_out/summary.txt
