#!/bin/bash
# registers / scratch / occupancy of every kernel of the engine, as the compiler reports them (no GPU needed)
# usage: scripts/kernel_resources.sh [-f family] [extra hipcc flags]     family: engine | launch_fused_land | launch_fused_mesh | launch_wide | launch_tiled | launch_pair (default: all)
cd "$(dirname "$0")/../evosoro_amd/csrc"
FAMS="engine launch_fused_land launch_fused_mesh launch_wide launch_tiled"
if [ "$1" = "-f" ]; then FAMS="$2"; shift 2; fi
for f in $FAMS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm "$@" --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/vxh_${f}_res.o 2>&1 &
done 2>&1 | cat |
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
    if not name.startswith("k_") and "::k_" not in name: continue
    print("%-46s VGPR %4s AGPR %3s SGPR %4s scratch %5s B/lane  occupancy %s  LDS %s" % (name, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")) + "  spilled SGPRs %s" % r.get("SGPRs Spill"))
'
