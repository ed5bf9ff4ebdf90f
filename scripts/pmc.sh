#!/bin/bash
# usage (on the GPU box): scripts/pmc.sh <tag> <gpu_diag mode> : rocprofv3 PMC passes over scripts/dev_gpu_diag.py
# writes gpurun_out/pmc_<tag>_<n>/..._counter_collection.csv ; summarise with scripts/pmc_sum.py
tag=$1; mode=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM" ; do
  n=$((n+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/pmc_${tag}_$n -- python $root/scripts/dev_gpu_diag.py $mode > $root/gpurun_out/pmc_${tag}_$n.log 2>&1
done
python $root/scripts/pmc_sum.py $root/gpurun_out/pmc_${tag}_*
