#!/bin/bash
# registers / scratch / occupancy of every kernel of the engine, as the compiler reports them (no GPU needed)
# usage: scripts/kernel_resources.sh [extra hipcc flags]
cd "$(dirname "$0")/../evosoro_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm "$@" -Rpass-analysis=kernel-resource-usage -c engine.hip -o /tmp/vxh_engine_res.o 2>&1 |
python3 -c '
import re, sys, subprocess
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark: +([\w \[\]/]+?): (.*?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ("Function Name", "Name"):
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void vxh::", "")
    print("%-46s VGPR %4s AGPR %3s SGPR %4s scratch %5s B/lane  occupancy %s  LDS %s" % (name, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")) + "  spilled SGPRs %s" % r.get("SGPRs Spill"))
'
