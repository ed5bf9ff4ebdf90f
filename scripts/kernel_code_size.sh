#!/bin/bash
# code bytes of every kernel of the engine (no GPU needed): the hot loop of a resident kernel should stay well inside the 64 KB instruction cache
cd "$(dirname "$0")/../evosoro_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm "$@" --cuda-device-only -c engine.hip -o /tmp/vxh_engine_dev.o 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/vxh_engine_dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/vxh_engine_dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf -s -W /tmp/vxh_engine_dev.co | awk '$4=="FUNC" {print $3, $8}' | while read size name; do echo "$size $(c++filt "$name" | sed 's/(.*//; s/void vxh:://')"; done | sort -n
