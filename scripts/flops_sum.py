#!/usr/bin/env python3
"""scripts/flops_sum.py <gpurun_out> <tag>: condenses the counter passes of scripts/flops_per_unit.sh into
profiles/<tag>_flops_per_unit.json: per bench workload the FP64 flops (64 lanes x (ADD + MUL + 2 FMA + TRANS) wave-instructions) and the
vector wave-instructions per voxel-step, summed over every stepping kernel of the process, with the per-kernel split."""
import csv, glob, json, os, sys, collections

out_root, tag = sys.argv[1], sys.argv[2]
prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
table = {}
for key in ("headline", "cfg1", "cfg3", "cfg4", "dense", "mixed"):
    row = {}
    for kind in ("f64", "valu"):
        meta = os.path.join(out_root, "unit_%s_%s.json" % (key, kind))
        d = os.path.join(out_root, "prof_%s_unit_%s_%s" % (tag, key, kind))
        if not (os.path.exists(meta) and os.path.isdir(d)):
            continue
        vs = json.load(open(meta))["voxel_steps"]
        tot = collections.defaultdict(float)
        per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if not any(k in name for k in ("k_robot", "k_tile", "k_bonds", "k_voxels", "k_step_begin", "k_mesh", "k_facets")):
                    continue
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); per_kernel[name.split("(")[0].replace("void vxh::", "")][r["Counter_Name"]] += float(r["Counter_Value"])
        if kind == "f64":
            fl = lambda t: 64.0 * (t["SQ_INSTS_VALU_ADD_F64"] + t["SQ_INSTS_VALU_MUL_F64"] + 2.0 * t["SQ_INSTS_VALU_FMA_F64"] + t["SQ_INSTS_VALU_TRANS_F64"])
            row["fp64_flop_per_voxel_step"] = fl(tot) / vs
            row["fp64_wave_insts_per_voxel_step"] = (tot["SQ_INSTS_VALU_ADD_F64"] + tot["SQ_INSTS_VALU_MUL_F64"] + tot["SQ_INSTS_VALU_FMA_F64"] + tot["SQ_INSTS_VALU_TRANS_F64"]) / vs
            row["voxel_steps_counted"] = vs
            row["kernels"] = {k: fl(v) for k, v in per_kernel.items() if fl(v) > 0}
        else:
            row["valu_inst_per_voxel_step"] = tot["SQ_INSTS_VALU"] / vs
            row["salu_inst_per_voxel_step"] = tot["SQ_INSTS_SALU"] / vs
            row["lds_inst_per_voxel_step"] = tot["SQ_INSTS_LDS"] / vs
    if row:
        if row.get("valu_inst_per_voxel_step") and row.get("fp64_wave_insts_per_voxel_step"):
            row["non_fp64_share_of_valu"] = 1.0 - row["fp64_wave_insts_per_voxel_step"] / row["valu_inst_per_voxel_step"]
        table[key] = row
path = os.path.join(prof_dir, "%s_flops_per_unit.json" % tag)
with open(path, "w") as f:
    json.dump(table, f, indent=1)
print(json.dumps(table, indent=1))
