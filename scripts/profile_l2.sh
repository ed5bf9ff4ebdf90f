#!/bin/bash
# usage (on the GPU box): scripts/profile_l2.sh <tag>
# L2 (TCC) request / hit / miss and memory-side request counters of the resident kernel on the bench population, with and without
# self-collision (scripts/dev_gpu_diag.py l2pop): what FETCH_SIZE / WRITE_SIZE (= L2 memory-side requests, MI355X_MICROARCH.md "HBM")
# are made of.  Counters in their own passes, --kernel-trace only.
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"; do
    i=$((i+1))
    for col in 1 0; do
        timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/prof_${tag}_l2_${i}_col$col -- python $root/scripts/dev_gpu_diag.py l2pop $col > $root/gpurun_out/prof_${tag}_l2_${i}_col$col.log 2>&1
    done
done
python - <<PY
import csv, glob, collections
for col in (1, 0):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob("$root/gpurun_out/prof_${tag}_l2_*_col%d/**/*counter_collection.csv" % col, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_robot_steps" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("self-collision", col)
    for c in sorted(tot): print("   %-26s %16.0f  (%d dispatches)" % (c, tot[c], n[c]))
    if tot.get("TCC_HIT_sum") or tot.get("TCC_MISS_sum"):
        print("   L2 hit rate %.3f" % (tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"])))
PY
grep -il "error\|invalid\|not found" $root/gpurun_out/prof_${tag}_l2_*.log
