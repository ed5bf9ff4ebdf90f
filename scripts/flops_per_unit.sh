#!/bin/bash
# usage (on the GPU box): scripts/flops_per_unit.sh <tag>
# FP64 flops and vector instructions per voxel-step of every bench workload: two rocprofv3 --pmc passes (counters only, --kernel-trace)
# over scripts/unit_workload.py <key>; condensed locally by scripts/flops_sum.py gpurun_out <tag> into profiles/<tag>_flops_per_unit.json,
# which bench.py multiplies by a run's own rate (roofline.binding)
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for key in ${2:-headline cfg1 cfg3 cfg4 dense mixed}; do
    rm -rf $root/gpurun_out/prof_${tag}_unit_${key}_f64 $root/gpurun_out/prof_${tag}_unit_${key}_valu
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv \
        -d $root/gpurun_out/prof_${tag}_unit_${key}_f64 -- python $root/scripts/unit_workload.py $key > $root/gpurun_out/prof_${tag}_unit_${key}_f64.log 2>&1
    cp $root/gpurun_out/unit_${key}.json $root/gpurun_out/unit_${key}_f64.json
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv \
        -d $root/gpurun_out/prof_${tag}_unit_${key}_valu -- python $root/scripts/unit_workload.py $key > $root/gpurun_out/prof_${tag}_unit_${key}_valu.log 2>&1
    cp $root/gpurun_out/unit_${key}.json $root/gpurun_out/unit_${key}_valu.json
done
