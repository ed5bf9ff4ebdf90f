#!/bin/bash
# same-box A/B of the working library against a kept one, alternating: scripts/r6_ab.sh <other .so name under evosoro_amd/> <reps> <dev_gpu_diag mode...>
cd "$(dirname "$0")/.."
other=$1; reps=$2; shift 2
for rep in $(seq 1 $reps); do
  echo "== A libvxhip.so"; python scripts/ab_lib.py libvxhip.so "$@" 2>&1 | grep -v "^   broad\|^vxhip:"
  echo "== B $other"; python scripts/ab_lib.py "$other" "$@" 2>&1 | grep -v "^   broad\|^vxhip:"
done
