mkdir -p gpurun_out
B=build/ab/libsamrs_hip_r5final.so; P=build/ab/libsamrs_hip_pf.so
BQ="--no-cpu-baseline --no-alt-dtype --no-pcie-leg --no-cli-leg --no-fast-leg --no-rle-leg"
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
for r in 1 2; do
 for v in base pf; do
  L=$B; [ $v = pf ] && L=$P
  SAMRS_LIB_PATH=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pfprof_${v}_$r -o t -- python bench.py --steps 4 --warmup 1 $BQ > gpurun_out/pfprof_${v}_$r.log 2>&1
  f=$(find gpurun_out/pfprof_${v}_$r -name "*kernel_stats.csv" | head -1)
  echo "== $v $r"; grep -E "gemm_et_x64_kernel<1, true|gemm_et_w4x|gemm_et_x64p_kernel<1, false, 0|layernorm_kernel<1>" "$f" | awk -F'","' '{print substr($1,1,70), $2, $4}'
  find gpurun_out/pfprof_${v}_$r -name "*kernel_trace.csv" -delete
 done
done
for r in 1 2; do
 SAMRS_LIB_PATH=$B ONLY=lin2,proj timeout 200 python tools/gemm_bench.py 27 f16 2>/dev/null | tail -2
 SAMRS_LIB_PATH=$P ONLY=lin2,proj timeout 200 python tools/gemm_bench.py 27 f16 2>/dev/null | tail -2
done
BENCH_STEPS=20 bash tools/ab_libs.sh 4 $B $P > gpurun_out/pf_ab2.log 2>&1
cat gpurun_out/pf_ab2.log
