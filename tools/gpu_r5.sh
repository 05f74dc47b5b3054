#!/bin/bash
# Round-5 GPU session helper (runs on the GPU box via gpurun): named steps, each with its own timeout and log.
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONDONTWRITEBYTECODE=1
run() { local name=$1; shift; local t=$1; shift
  echo "=== $name" | tee -a gpurun_out/summary.txt
  local t0=$SECONDS
  timeout "$t" "$@" > "gpurun_out/$name.log" 2>&1
  echo "rc=$? ($((SECONDS - t0)) s)" | tee -a gpurun_out/summary.txt
  grep -E "passed|failed|error" "gpurun_out/$name.log" | tail -3 | tee -a gpurun_out/summary.txt; }
: > gpurun_out/summary.txt
PT="python -m pytest -m gpu -q -s -rA -p no:cacheprovider -x"
BQ="--no-cpu-baseline --no-alt-dtype --no-pcie-leg --no-cli-leg --no-fast-leg --no-rle-leg"
prof_env() { cd /tmp; export TMPDIR=/tmp; cd - >/dev/null; }
for step in "$@"; do
  case $step in
    k_attn)  run k_attn 600 $PT tests/test_kernels_gpu.py -k "attention" ;;
    kern)    run kern 900 $PT tests/test_kernels_gpu.py ;;
    parity)  run parity 1500 $PT tests/test_parity_gpu.py ;;
    pipe)    run pipe 900 $PT tests/test_pipeline_gpu.py tests/test_rle_gpu.py ;;
    all)     run all 2400 python -m pytest tests -m gpu -q -rA -p no:cacheprovider ;;
    smoke)   run smoke 600 python -c "import __graft_entry__ as g; g.smoke()" ;;
    attnb)   run attnb_new 300 python tools/attn_bench.py
             SAMRS_LIB_PATH=${AB_LIB:-samrs_amd/csrc/libsamrs_hip_noskew.so} run attnb_old 300 python tools/attn_bench.py
             run attnb_new2 300 python tools/attn_bench.py ;;
    attnpmc) prof_env
             run attnpmc1 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/attnpmc1 -o p -- python tools/attn_bench.py
             SAMRS_LIB_PATH=${AB_LIB:-samrs_amd/csrc/libsamrs_hip_noskew.so} run attnpmc0 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/attnpmc0 -o p -- python tools/attn_bench.py
             run attnpmc2 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/attnpmc2 -o p -- python tools/attn_bench.py ;;
    k_gemm)  run k_gemm 600 $PT tests/test_kernels_gpu.py -k "gemm" ;;
    k_w4x)   run k_w4x 600 $PT tests/test_kernels_gpu.py -k "w4x" ;;
    energy)  run energy 300 python tools/gemm_energy.py ;;
    k_rbox)  run k_rbox 300 $PT tests/test_rbox_prompt.py ;;
    golden)  run golden 900 $PT tests/test_parity_gpu.py -k "reference_golden or vit_b_c1" ;;
    wint)    SAMRS_LIB_PATH=build/ab/libsamrs_hip_wt.so run wint 200 python tools/win_timeline.py ;;
    energy2) run energy_new 300 python tools/gemm_energy.py
             SAMRS_LIB_PATH=${AB_LIB:-build/ab/libsamrs_hip_as26.so} run energy_old 300 python tools/gemm_energy.py ;;
    attnab)  for v in "0 0" "1 1" "3 1" "0 0" "1 1" "3 1"; do read wp gp <<< "$v"; SAMRS_WIN_PIPE=$wp SAMRS_GLB_PIPE=$gp run "attnab_w${wp}g${gp}_$SECONDS" 200 python tools/attn_bench.py; done ;;
    wint1)   SAMRS_WIN_PIPE=1 SAMRS_LIB_PATH=build/ab/libsamrs_hip_wt.so run wint1 200 python tools/win_timeline.py
             SAMRS_WIN_PIPE=3 SAMRS_LIB_PATH=build/ab/libsamrs_hip_wt.so run wint3 200 python tools/win_timeline.py ;;
    range)   run range 600 $PT tests/test_parity_gpu.py -k "operand_range" ;;
    w4xab)   for r in 1 2; do for v in 0 1; do SAMRS_GEMM_W4X=$v run "bench_w4x${v}_r$r" 400 python bench.py --steps ${BENCH_STEPS:-12} --warmup 3 $BQ; done; done
             grep -h '"value"' gpurun_out/bench_w4x*.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('w4x A/B', d['value'], 'img/s', d['ms_per_step'], 'ms/step; lin1', d['roofline']['avg_launch_ms'], 'ms', d['roofline']['achieved'], 'TF')
" | tee -a gpurun_out/summary.txt ;;
    lntail)  run lntail 600 $PT tests/test_parity_gpu.py -k "layernorm_tail" ;;
    proflnt) prof_env
             SAMRS_LN_TAIL=1 run proflnt1 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proflnt1 -o t -- python bench.py --steps 3 --warmup 1 $BQ
             SAMRS_LN_TAIL=0 run proflnt0 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proflnt0 -o t -- python bench.py --steps 3 --warmup 1 $BQ
             for v in 0 1; do f=$(find gpurun_out/proflnt$v -name "*kernel_stats.csv" | head -1); echo "== LN_TAIL=$v $f"; head -12 "$f" | cut -c1-60,200-400 | awk -F, '{print $0}' ; done | tee -a gpurun_out/summary.txt ;;
    abenv)   # A/B of engine env switches on one box: ABENV="NAME=a NAME=b ..." alternated ABR times
             for r in $(seq 1 ${ABR:-2}); do for kv in ${ABENV}; do env $kv bash -c "timeout 400 python bench.py --steps ${BENCH_STEPS:-16} --warmup 4 $BQ" > "gpurun_out/ab_${kv}_r$r.log" 2>&1
               grep -h '"value"' "gpurun_out/ab_${kv}_r$r.log" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$kv r$r:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; lin1', d['roofline']['avg_launch_ms'], 'ms')
" | tee -a gpurun_out/summary.txt; done; done ;;
    pstat)   run pstat 900 $PT tests/test_parity_gpu.py -k "statistical_parity_sample" ;;
    kmx)     run kmx 600 $PT tests/test_kernels_gpu.py -k "mx" ;;
    c4ab)    SAMRS_LO_FORMAT=0 run c4_lo0 400 python bench.py --workload c4 --steps ${BENCH_STEPS:-12} --warmup 3 $BQ
             SAMRS_LO_FORMAT=4 run c4_lo4 400 python bench.py --workload c4 --steps ${BENCH_STEPS:-12} --warmup 3 $BQ
             SAMRS_LO_FORMAT=0 run c4_lo0b 400 python bench.py --workload c4 --steps ${BENCH_STEPS:-12} --warmup 3 $BQ
             SAMRS_LO_FORMAT=4 run c4_lo4b 400 python bench.py --workload c4 --steps ${BENCH_STEPS:-12} --warmup 3 $BQ ;;
    mxb)     run mxb 300 python tools/mx_bench.py ;;
    profc4)  prof_env
             SAMRS_LO_FORMAT=${LOF:-4} run profc4 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profc4 -o c4 -- python bench.py --workload c4 --steps 3 --warmup 1 $BQ ;;
    pstats)  run pstats 1500 python tools/parity_stats.py --modes ${PS_MODES:-15,79,63} ${PS_ARGS:-} ;;
    benchq)  run benchq 900 python bench.py --steps ${BENCH_STEPS:-8} --warmup 2 $BQ ${BENCH_EXTRA:-} ;;
    bench)   run bench 1500 python bench.py ${BENCH_ARGS:-} ;;
    ablib)   run ablib 1200 bash tools/ab_libs.sh ${AB_ROUNDS:-2} ${AB_LIB:-samrs_amd/csrc/libsamrs_hip_noskew.so} samrs_amd/csrc/libsamrs_hip.so ;;
    c3)      run c3 600 python bench.py --workload c3 --steps ${BENCH_STEPS:-6} --warmup 2 $BQ ;;
    c4)      run c4 600 python bench.py --workload c4 --steps ${BENCH_STEPS:-6} --warmup 2 $BQ ;;
    decb)    run decb 300 python tools/dec_bench.py 20 ;;
    gemmb)   run gemmb 900 python tools/gemm_bench.py ${GEMM_VARIANTS:-27,28} f16 ;;
    mxprobe) run mxprobe 120 bash -c "hipcc --offload-arch=gfx950 -O2 -o /tmp/mx_probe tools/mx_probe.hip && /tmp/mx_probe" ;;
    mfma)    run mfma 300 bash -c "hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/mfma_rate.hip && /tmp/mfma_rate" ;;
    prof)    prof_env
             run prof 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r05 -- python bench.py --steps 3 --warmup 1 $BQ ;;
    pmc)     prof_env
             PC="python bench.py --steps 2 --warmup 1 $BQ"
             run pmc1 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -o p -- $PC
             run pmc2 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc2 -o p -- $PC
             run pmc3 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc3 -o p -- $PC
             run pmc4 600 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc4 -o p -- $PC ;;
    *)       n=$((${n:-0} + 1)); echo "cmd$n: $step" >> gpurun_out/summary.txt; run "cmd$n" 1200 bash -c "$step" ;;
  esac
done
tail -n 8 gpurun_out/*.log 2>/dev/null | tail -200
cat gpurun_out/summary.txt
