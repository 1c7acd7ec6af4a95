#!/bin/bash
# SQ counters of the conv kernels (scripts/bench_conv.py, layers in $ONLY): what the mode-3 implicit GEMM is busy with.
# Separate --pmc passes, kernel-trace only.  Run on the GPU box from the repo root: bash scripts/pmc_igemm.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export ONLY=${ONLY:-E3,D3}
export REPS=1
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_igemm
rm -rf $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/scripts/bench_conv.py 64 > $OUT.p$i.log 2>&1
done
python3 $ROOT/scripts/pmc_igemm.py $OUT > $ROOT/gpurun_out/pmc_igemm_summary.txt
