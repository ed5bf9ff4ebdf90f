#!/bin/bash
# code bytes of every kernel of the engine (no GPU needed): the hot loop of a resident kernel should stay well inside the 64 KB instruction cache
cd "$(dirname "$0")/../evosoro_amd/csrc"
for f in engine launch_fused_land launch_fused_mesh launch_wide launch_tiled; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm "$@" --cuda-device-only -c $f.hip -o /tmp/vxh_${f}_dev.co 2>/dev/null
  /opt/rocm/lib/llvm/bin/llvm-readelf -s -W /tmp/vxh_${f}_dev.co | awk '$4=="FUNC" {print $3, $8}' | while read size name; do echo "$size $(c++filt "$name" | sed 's/(.*//; s/void vxh:://')"; done
done | sort -n
