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
# the other BASELINE configs (multi-workgroup kernel k_tile_steps on the small populations and the 20^3 lattice, resident MESH kernel
# on the swimmers): kernel trace + stats of scripts/dev_gpu_diag.py tileprof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_tiled -- python $root/scripts/dev_gpu_diag.py tileprof > $root/gpurun_out/prof_${tag}_tiled.log 2>&1
python $root/scripts/profile_sum.py $root/gpurun_out $tag tiled
