#!/usr/bin/env python3
"""One bench workload, alone in a process, for the counter passes of scripts/flops_per_unit.sh: runs the headline population or one
of bench.py's other_configs entries exactly as bench.py does and writes the voxel-steps the WHOLE process took (pre-advance included:
the counters see every dispatch) to gpurun_out/unit_<key>.json.  FP64 flops per voxel-step = counted flops / that number."""
import json
import os
import sys
import tempfile
import shutil

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
key = sys.argv[1]
import torch  # noqa: E402
torch.cuda.init()          # (torch's HIP runtime first, as in bench.py)
import bench  # noqa: E402
from evosoro_amd import engine  # noqa: E402

if key == "headline":
    tmp = tempfile.mkdtemp(prefix="vxunit_")
    try:
        steps, warmup = 600, 200
        paths = bench.make_population(tmp, 512, 0, (10, 10, 10), max(0.5, bench.INIT_CM_TIME + (steps + warmup + 1100) * 7.2e-4), bench.INIT_CM_TIME)
        with engine.Engine(engine.VOXCAD, 0) as eng:
            eng.add_vxa_files(paths)
            dims = [eng.dims(i) for i in range(len(paths))]
            eng.step(int(max(bench.INIT_CM_TIME / d["dt"] for d in dims)) + 32 + warmup)
            eng.step(steps)
            c = eng.counters()
            out = {"key": key, "voxel_steps": c.voxel_steps, "kernel": bench.kernel_name(c.dominant_block), "kernel_seconds": c.kernel_seconds}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
else:
    r = bench.side_workload(engine, key, 0)
    out = {"key": key, "voxel_steps": r["_voxel_steps_process"], "kernel": r.get("kernel", r.get("kernels")), "value": r["value"]}
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
with open(os.path.join(REPO, "gpurun_out", "unit_%s.json" % key), "w") as f:
    json.dump(out, f)
print(json.dumps(out))
