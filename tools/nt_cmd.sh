mkdir -p gpurun_out
python tools/gemm_hash.py > gpurun_out/nth_hash_new.log 2>&1
SAMRS_LIB_PATH=build/ab/libsamrs_hip_r5final.so python tools/gemm_hash.py > gpurun_out/nth_hash_base.log 2>&1
diff gpurun_out/nth_hash_base.log gpurun_out/nth_hash_new.log && echo HASHES_EQUAL
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" > gpurun_out/nth_kern.log 2>&1; tail -2 gpurun_out/nth_kern.log
BQ="--no-cpu-baseline --no-alt-dtype --no-pcie-leg --no-cli-leg --no-fast-leg --no-rle-leg"
for r in 1 2 3; do for v in 0 -1; do
  SAMRS_NT_HIDDEN=$v timeout 400 python bench.py --steps 20 --warmup 4 $BQ 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('NT_HIDDEN=$v', d['value'], 'ms/step', d['ms_per_step'], 'lin1', d['roofline']['avg_launch_ms'], d['roofline']['achieved'])
" | tee -a gpurun_out/nth_ab.log
done; done
