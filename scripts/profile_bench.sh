#!/bin/bash
# usage (on the GPU box): scripts/profile_bench.sh <tag>   e.g. r01
# rocprofv3 passes over the default bench.py run (kernel trace + stats; HBM counters in their own passes) and over
# the 8-byte-per-lane calibration kernels; raw CSVs under gpurun_out/prof_<tag>_*, summary printed by profile_sum.py
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_stream $root/scripts/calib_stream.hip
BENCH="python $root/bench.py --no-cpu-baseline --no-other-configs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_stats -- $BENCH > $root/gpurun_out/prof_${tag}_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/prof_${tag}_fetch -- $BENCH > $root/gpurun_out/prof_${tag}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/prof_${tag}_write -- $BENCH > $root/gpurun_out/prof_${tag}_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/prof_${tag}_calfetch -- /tmp/calib_stream > $root/gpurun_out/prof_${tag}_calfetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/prof_${tag}_calwrite -- /tmp/calib_stream > $root/gpurun_out/prof_${tag}_calwrite.log 2>&1
python $root/scripts/profile_sum.py $root/gpurun_out $tag
# SQ counters of the dominant kernel (instruction mix, VALU activity, waits, LDS bank conflicts): a few counters per pass, each pass
# with --kernel-trace only; summed per kernel into profiles/<tag>_pmc_sq.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_IFETCH" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/prof_${tag}_sq$i -- $BENCH > $root/gpurun_out/prof_${tag}_sq$i.log 2>&1
done
python $root/scripts/profile_sum.py $root/gpurun_out $tag sq
# the command the round-end driver runs: one kernel trace (both regimes of DESIGN.md "The cost of a launch" in it)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_driver -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $root/gpurun_out/prof_${tag}_driver.log 2>&1
# SQ counters of the kernels of the other BASELINE configs (wide kernel on 64 x 6^3, wide MESH kernel on 64 x 8^3 swimmers, tiled kernel
# on the 20^3 lattice): scripts/dev_gpu_diag.py cfg1 / cfg3 / cfg4 each run one of them
for cfg in cfg1 cfg3 cfg4; do
    j=0
    for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
               "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
        j=$((j+1))
        timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/prof_${tag}_${cfg}_sq$j -- python $root/scripts/dev_gpu_diag.py $cfg > $root/gpurun_out/prof_${tag}_${cfg}_sq$j.log 2>&1
    done
done
python $root/scripts/pmc_sum.py $root/gpurun_out/prof_${tag}_cfg1_sq* > $root/gpurun_out/${tag}_pmc_sq_cfg1.txt
python $root/scripts/pmc_sum.py $root/gpurun_out/prof_${tag}_cfg3_sq* > $root/gpurun_out/${tag}_pmc_sq_cfg3.txt
python $root/scripts/pmc_sum.py $root/gpurun_out/prof_${tag}_cfg4_sq* > $root/gpurun_out/${tag}_pmc_sq_cfg4.txt
# the other BASELINE configs (multi-workgroup kernel k_tile_steps on the small populations and the 20^3 lattice, resident MESH kernel
# on the swimmers): kernel trace + stats of scripts/dev_gpu_diag.py tileprof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_tiled -- python $root/scripts/dev_gpu_diag.py tileprof > $root/gpurun_out/prof_${tag}_tiled.log 2>&1
python $root/scripts/profile_sum.py $root/gpurun_out $tag tiled
# FP64 flops / vector instructions per voxel-step of every bench workload (what bench.py's roofline.binding multiplies by a run's own rate):
# two counter passes per workload; condensed locally by scripts/flops_sum.py gpurun_out <tag> into profiles/<tag>_flops_per_unit.json
$root/scripts/flops_per_unit.sh $tag
