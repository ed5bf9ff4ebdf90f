#!/bin/bash
# same-box A/B of the working library against a kept one: scripts/r4_ab.sh <other .so name> <dev_gpu_diag mode...>
cd "$(dirname "$0")/.."
other=$1; shift
for rep in 1 2; do
  python scripts/ab_lib.py libvxhip.so "$@" 2>&1 | grep -v "^   broad"
  python scripts/ab_lib.py "$other" "$@" 2>&1 | grep -v "^   broad"
done
