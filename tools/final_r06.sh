#!/bin/bash
# Round-6 final evidence sequence (ONE gpurun call): full GPU suite -> parity statistics (incl. the reference-backend floor and the
# heavy-tailed-weights sample) pinned to this tree -> rocprofv3 kernel trace (+ tools/trace_summary.py: per-(kernel, role, grid) rows)
# and PMC passes of bench.py -> dominant-kernel summary pinned to gemm.hip -> the bench lines (c2 default, c3, c4, self-launched x2).
# evidence files are copied into profiles/ ONLY from a green run of THIS call (stale files are removed first).
set -u
mkdir -p gpurun_out; cd "${GRAFT_REPO_ROOT:-.}"; export PYTHONDONTWRITEBYTECODE=1
BQ="--no-cpu-baseline --no-alt-dtype --no-pcie-leg --no-cli-leg --no-fast-leg --no-rle-leg"
rm -f gpurun_out/parity_stats_test.json gpurun_out/heavy_tailed_parity.json
timeout 2300 python -m pytest tests -m gpu -q -rs -p no:cacheprovider > gpurun_out/all_final.log 2>&1; RC=$?; tail -4 gpurun_out/all_final.log
if [ $RC -eq 0 ] && [ -s gpurun_out/parity_stats_test.json ] && [ -s gpurun_out/heavy_tailed_parity.json ]; then
  python - <<'PY'
import json
st = json.load(open("gpurun_out/parity_stats_test.json"))
st["heavy_tailed"] = json.load(open("gpurun_out/heavy_tailed_parity.json"))
json.dump(st, open("profiles/parity_stats.json", "w"))
print("profiles/parity_stats.json <- this run (csrc", st["csrc_sha16"], ")")
PY
  cp gpurun_out/all_final.log profiles/r06_gpu_tests.txt
else
  echo "GPU suite rc=$RC or parity files missing: profiles/parity_stats.json NOT updated"
fi
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
rm -rf gpurun_out/prof gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 gpurun_out/pmc4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r06 -- python bench.py --steps 6 --warmup 2 $BQ > gpurun_out/prof.log 2>&1
TR=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); ST=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$TR" ]; then python tools/trace_summary.py "$TR" --steps 8 --out profiles/r06_trace_summary | head -24; fi
if [ -n "$ST" ]; then cp "$ST" profiles/r06_kernel_stats.csv; fi
PC="python bench.py --steps 2 --warmup 1 $BQ"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -o p -- $PC > gpurun_out/pmc1.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc2 -o p -- $PC > gpurun_out/pmc2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc3 -o p -- $PC > gpurun_out/pmc3.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc4 -o p -- $PC > gpurun_out/pmc4.log 2>&1
python tools/pmc_summary.py gpurun_out gpurun_out/r06_pmc_per_kernel.json --dominant "gemm_et_w4x_kernel<1, false, 2, true" gpurun_out/dominant_kernel_pmc.json 2>&1 | tail -2
if [ -s gpurun_out/dominant_kernel_pmc.json ]; then cp gpurun_out/dominant_kernel_pmc.json profiles/dominant_kernel_pmc.json; cp gpurun_out/r06_pmc_per_kernel.json profiles/r06_pmc_per_kernel.json; fi
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
timeout 900 python bench.py > gpurun_out/bench_final.log 2> gpurun_out/bench_final.err; tail -1 gpurun_out/bench_final.log | cut -c1-400
tail -1 gpurun_out/bench_final.log > profiles/r06_bench.json
timeout 400 python bench.py --workload c3 --steps 12 --warmup 2 $BQ > gpurun_out/c3.log 2>&1; tail -1 gpurun_out/c3.log | cut -c1-200
timeout 400 python bench.py --workload c4 --steps 12 --warmup 2 $BQ > gpurun_out/c4.log 2>&1; tail -1 gpurun_out/c4.log | cut -c1-200
# the driver's N > 1 command form on this ONE-GPU box: bench.py launches its own ranks; both share cuda:0 over gloo
SAMRS_BENCH_SHARE_GPU=1 timeout 600 env -u WORLD_SIZE -u RANK -u LOCAL_RANK python bench.py --gpus 2 --steps 6 --warmup 2 $BQ > gpurun_out/x2_shared.log 2>&1; tail -1 gpurun_out/x2_shared.log | cut -c1-300
{ echo "# c3"; tail -1 gpurun_out/c3.log; echo "# c4"; tail -1 gpurun_out/c4.log; echo "# python bench.py --gpus 2 (self-launched; both ranks on ONE GPU over gloo: control flow, not a scaling number)"; tail -1 gpurun_out/x2_shared.log; } > profiles/r06_bench_other_workloads.txt
# only gpurun_out/ travels back from the GPU box: hand over what this run wrote into profiles/
cp profiles/r06_* profiles/parity_stats.json profiles/dominant_kernel_pmc.json gpurun_out/ 2>/dev/null
