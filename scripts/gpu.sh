#!/bin/bash
# build both libraries, then run the given command line on an MI355X box; nothing is sent when the build fails
# usage: scripts/gpu.sh [--timeout S] '<command>'
set -e
cd "$(dirname "$0")/.."
T=900
if [ "$1" = "--timeout" ]; then T=$2; shift 2; fi
make -C evosoro_amd/csrc all prof > /tmp/vxh_build.log 2>&1 || { grep -E "error" -A4 /tmp/vxh_build.log | head -40; echo "BUILD FAILED"; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$1"
