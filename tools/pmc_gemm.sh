#!/bin/bash
# Per-shape PMC comparison of one of our GEMM variants with the vendor kernel torch.matmul picks (both run inside
# tools/gemm_bench.py on the same operands).  usage: pmc_gemm.sh <shape-prefix> <variant>   -> gpurun_out/pmcg_<shape>_<pass>/
# Counters in their own runs (no trace domains besides --kernel-trace), three passes: SQ / TCC-fetch / TCC-write.
shape=$1; var=${2:-28}
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
P2="FETCH_SIZE GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VMEM_WR"
P3="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  ONLY=$shape timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/pmcg_${shape}_$i -o p -- python tools/gemm_bench.py $var f16 > gpurun_out/pmcg_${shape}_$i.log 2>&1
done
grep -h "TF" gpurun_out/pmcg_${shape}_1.log | cut -c1-220
