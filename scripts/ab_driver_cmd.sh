#!/bin/bash
# same-box A/B of the round-end driver's command (bench.py --gpus 1 --steps 20 --warmup 5, side legs off) between the working library and a
# kept one (evosoro_amd/libvxhip_base.so), four times alternating; prints: 0 = working library, 1 = the other; value; ms per step.
# Run on the GPU box: gpurun -- 'bash scripts/ab_driver_cmd.sh'
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
for lib in libvxhip.so libvxhip_base.so; do
VXH_LIB_NAME=$lib python - <<PY
import os, sys, json, subprocess
sys.path.insert(0, os.getcwd())
from evosoro_amd import engine
engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), os.environ["VXH_LIB_NAME"])
import bench
sys.argv = ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs"]
import io, contextlib
bench.main()
PY
done; done 2>/dev/null | python -c "
import sys, json
for i,l in enumerate(sys.stdin):
    if l.startswith('{'):
        d=json.loads(l); print(i%2, d['value'], d['ms_per_step'])"
