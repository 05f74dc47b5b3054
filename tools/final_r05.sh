#!/bin/bash
# Round-5 final evidence sequence (one gpurun call): full GPU suite -> parity statistics pinned to this tree -> rocprofv3 stats + PMC passes of
# bench.py -> dominant-kernel summary pinned to gemm.hip -> the bench lines (c2 default, c3, c4) that read both.
set -u
mkdir -p gpurun_out; cd "${GRAFT_REPO_ROOT:-.}"; export PYTHONDONTWRITEBYTECODE=1
BQ="--no-cpu-baseline --no-alt-dtype --no-pcie-leg --no-cli-leg --no-fast-leg --no-rle-leg"
timeout 2300 python -m pytest tests -m gpu -q -rs -p no:cacheprovider > gpurun_out/all_final.log 2>&1; tail -4 gpurun_out/all_final.log
cp gpurun_out/parity_stats_test.json profiles/parity_stats.json
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
rm -rf gpurun_out/prof gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 gpurun_out/pmc4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r05 -- python bench.py --steps 3 --warmup 1 $BQ > gpurun_out/prof.log 2>&1
PC="python bench.py --steps 2 --warmup 1 $BQ"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -o p -- $PC > gpurun_out/pmc1.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc2 -o p -- $PC > gpurun_out/pmc2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc3 -o p -- $PC > gpurun_out/pmc3.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc4 -o p -- $PC > gpurun_out/pmc4.log 2>&1
python tools/pmc_summary.py gpurun_out gpurun_out/r05_pmc_per_kernel.json --dominant "gemm_et_w4x_kernel<1, false, 2, true" gpurun_out/dominant_kernel_pmc.json 2>&1 | tail -2
cp gpurun_out/dominant_kernel_pmc.json profiles/dominant_kernel_pmc.json
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
timeout 900 python bench.py > gpurun_out/bench_final.log 2> gpurun_out/bench_final.err; tail -1 gpurun_out/bench_final.log | cut -c1-400
timeout 400 python bench.py --workload c3 --steps 12 --warmup 2 $BQ > gpurun_out/c3.log 2>&1; tail -1 gpurun_out/c3.log | cut -c1-200
timeout 400 python bench.py --workload c4 --steps 12 --warmup 2 $BQ > gpurun_out/c4.log 2>&1; tail -1 gpurun_out/c4.log | cut -c1-200
