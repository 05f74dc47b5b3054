#!/bin/bash
# Runs on the GPU box (via gpurun): each test group in its own process with its own timeout, so a
# fault in one kernel does not hide the others.  Logs land in gpurun_out/.
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONDONTWRITEBYTECODE=1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/gpu.txt
run() { # name timeout cmd...
  local name=$1; shift; local t=$1; shift
  echo "=== $name" | tee -a gpurun_out/summary.txt
  timeout "$t" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/summary.txt
  grep -E "passed|failed|error" "gpurun_out/$name.log" | tail -3 | tee -a gpurun_out/summary.txt
}
: > gpurun_out/summary.txt
PT="python -m pytest -m gpu -q -s -rA -p no:cacheprovider"
for grp in "$@"; do
  case $grp in
    k_basic)  run k_basic 600 $PT tests/test_kernels_gpu.py -k "convert or gemm or layernorm or postprocess or upscale2" ;;
    k_gemm)   run k_gemm 900 $PT tests/test_kernels_gpu.py -k "gemm" ;;
    k_m32)    run k_m32 600 $PT tests/test_kernels_gpu.py -k "m32" ;;
    k_fold)   run k_fold 600 $PT tests/test_kernels_gpu.py -k "fold or stats" ;;
    p_fold)   run p_fold 900 $PT tests/test_parity_gpu.py -k "vit_tiny1280 or folded" ;;
    foldb)    run foldb 600 python tools/fold_bench.py ;;
    abfold)   run abfold 900 bash tools/ab_env.sh SAMRS_LN_FOLD 0 1 ${AB_ROUNDS:-2} ;;
    abm32)    run abm32 900 bash tools/ab_env.sh SAMRS_GEMM_M32 ${M32_A:-0} ${M32_B:-3} ${AB_ROUNDS:-2} ;;
    p_c2c4)   run p_c2c4 900 $PT tests/test_parity_gpu.py -k "c2_c4" ;;
    k_win)    run k_win 600 $PT tests/test_kernels_gpu.py -k "window_attention" ;;
    k_glb)    run k_glb 600 $PT tests/test_kernels_gpu.py -k "global_attention" ;;
    p_enc)    run p_enc 900 $PT tests/test_parity_gpu.py -k "encoder_blockwise" ;;
    p_dec)    run p_dec 900 $PT tests/test_parity_gpu.py -k "decoder_alone" ;;
    p_e2e)    run p_e2e 900 $PT tests/test_parity_gpu.py -k "embedding_and_masks or paint or error or batch_equals" ;;
    p_gold)   run p_gold 1500 $PT tests/test_parity_gpu.py -k "golden or c1_config" ;;
    all)      run all 2400 python -m pytest tests -m gpu -q -rA -p no:cacheprovider ;;
    smoke)    run smoke 600 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench)    run bench 1200 python bench.py ${BENCH_ARGS:-} ;;
    c3)       run c3 600 python bench.py --workload c3 --steps ${BENCH_STEPS:-6} --warmup 2 --no-cpu-baseline --no-alt-dtype --no-pcie-leg ;;
    c4)       run c4 600 python bench.py --workload c4 --steps ${BENCH_STEPS:-6} --warmup 2 --no-cpu-baseline --no-alt-dtype --no-pcie-leg
              run c4m 600 python bench.py --workload c4 --c4-prompt rbox_mask --steps ${BENCH_STEPS:-6} --warmup 2 --no-cpu-baseline --no-alt-dtype --no-pcie-leg ;;
    b2)       SAMRS_BENCH_SHARE_GPU=1 run b2 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-pcie-leg ;;
    gemmb)    run gemmb 600 python tools/gemm_bench.py ${GEMM_VARIANTS:-0,1} f16 ;;
    pmc)      cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
              run pmc1 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -o p -- python ${PMC_CMD:-tools/gemm_bench.py 2 f16}
              run pmc2 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc2 -o p -- python ${PMC_CMD:-tools/gemm_bench.py 2 f16}
              run pmc3 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc3 -o p -- python ${PMC_CMD:-tools/gemm_bench.py 2 f16}
              run pmc4 600 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc4 -o p -- python ${PMC_CMD:-tools/gemm_bench.py 2 f16} ;;
    clk)      cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
              run clk 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/clk -o p -- python ${PMC_CMD:-tools/gemm_bench.py 25 f16} ;;
    pmcx)     cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
              run pmcx1 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmcx1 -o p -- python ${PMC_CMD:-tools/gemm_bench.py 25 f16}
              run pmcx2 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_CMD_FIFO_FULL --kernel-trace --output-format csv -d gpurun_out/pmcx2 -o p -- python ${PMC_CMD:-tools/gemm_bench.py 25 f16}
              run pmcx3 600 rocprofv3 --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TAGRAM0_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmcx3 -o p -- python ${PMC_CMD:-tools/gemm_bench.py 25 f16} ;;
    attnb)    run attnb 600 python tools/attn_bench.py ;;
    decb)     cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
              run decb 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/decb -o dec -- python tools/dec_bench.py 20 ;;
    benchq)   run benchq 900 python bench.py --steps ${BENCH_STEPS:-8} --warmup 2 --no-cpu-baseline --no-alt-dtype --no-pcie-leg ${BENCH_EXTRA:-} ;;
    prof)     cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
              run prof 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r02 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-pcie-leg ;;
  esac
done
tail -5 gpurun_out/*.log 2>/dev/null | tail -120
cat gpurun_out/summary.txt
