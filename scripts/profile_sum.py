#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by scripts/profile_bench.sh into the small files kept under profiles/:
<tag>_bench_kernel_stats.csv (per-kernel calls / total / average duration) and <tag>_hbm_traffic.json (FETCH_SIZE,
WRITE_SIZE per launch of the dominant kernel, with the calibration factors of the 8-byte-per-lane streaming kernels)."""
import csv, glob, json, os, sys, collections

out_root, tag = sys.argv[1], sys.argv[2]


def rows(sub, suffix):
    for f in glob.glob(os.path.join(out_root, "prof_%s_%s" % (tag, sub), "**", "*" + suffix), recursive=True):
        for r in csv.DictReader(open(f)):
            yield r


def counter_per_kernel(sub):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in rows(sub, "counter_collection.csv"):
        tot[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    return tot, n


prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
os.makedirs(prof_dir, exist_ok=True)
if len(sys.argv) > 3 and sys.argv[3] == "sq":      # `profile_sum.py <out_root> <tag> sq`: SQ counters summed per kernel over the passes
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for d in sorted(glob.glob(os.path.join(out_root, "prof_%s_sq*" % tag))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                tot[(r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"]); n[(r["Kernel_Name"], r["Counter_Name"])] += 1
    path = os.path.join(prof_dir, "%s_pmc_sq.txt" % tag)
    with open(path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc <a few SQ counters per pass> -- python bench.py --no-cpu-baseline --no-other-configs (N=1); "
                "sums over all dispatches of the kernel in that run\n")
        for kern in sorted({k for k, _ in tot}, key=lambda k: -sum(v for (kk, _), v in tot.items() if kk == k)):
            f.write(kern[:110] + "\n")
            for (kk, c) in sorted(tot):
                if kk == kern:
                    f.write("   %-28s %16.0f  (%d dispatches)\n" % (c, tot[(kk, c)], n[(kk, c)]))
    print("wrote", path)
    # what binds the dominant kernel, for bench.py's roofline.compute: SQ counters are in quad-cycles (ACTIVE / WAVE_CYCLES / WAIT), the
    # kernel's duration from the stats pass of the same command
    kinds = sorted({k for k, _ in tot}, key=lambda k: -tot.get((k, "SQ_INSTS_VALU"), 0.0))
    if kinds and tot.get((kinds[0], "SQ_WAVE_CYCLES")):
        dom = kinds[0]
        g = lambda c: tot.get((dom, c), 0.0)
        dur_ns = sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows("stats", "kernel_trace.csv") if r["Kernel_Name"] == dom)
        import re
        m = re.search(r"<(\d+)", dom)
        block = int(m.group(1)) if m else 0
        waves_per_simd = max(1, block // 256)
        f64 = {c: g(c) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")}
        flops = 64.0 * (f64["SQ_INSTS_VALU_ADD_F64"] + f64["SQ_INSTS_VALU_MUL_F64"] + 2.0 * f64["SQ_INSTS_VALU_FMA_F64"] + f64["SQ_INSTS_VALU_TRANS_F64"])
        comp = {"kernel": dom, "command": "python bench.py --no-cpu-baseline --no-other-configs", "dispatches": n.get((dom, "SQ_INSTS_VALU"), 0),
                "valu_busy": g("SQ_ACTIVE_INST_VALU") * waves_per_simd / g("SQ_WAVE_CYCLES"),
                "valu_busy_note": "SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / %d): share of the SIMD cycles (while the workgroup's %d wavefronts per SIMD are "
                                  "resident) in which a vector instruction was executing" % (waves_per_simd, waves_per_simd),
                "cycles_per_valu_inst": 4.0 * g("SQ_ACTIVE_INST_VALU") / max(1.0, g("SQ_INSTS_VALU")),
                "wait_share": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
                "lds_bank_conflict": g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE")),
                "bound": "fp64-issue"}
        if dur_ns > 0 and flops > 0:
            comp.update({"fp64_flops_per_s": flops / (dur_ns * 1e-9), "peak": 78.6e12, "unit": "FLOP/s", "frac": flops / (dur_ns * 1e-9) / 78.6e12,
                         "fp64_insts": f64,
                         "peak_note": "vector FP64 peak = half the FP32 vector peak of MI355X_MICROARCH.md (157.3 TFLOP/s): one FP64 wave-instruction per "
                                      "4 cycles and SIMD at 2.4 GHz (measured under load: 4.3-4.9 cycles at 2.05-2.2 GHz, scripts/ubench/fp64_issue.hip); "
                                      "flops = 64 lanes x (ADD + MUL + 2 FMA + TRANS) FP64 wave-instructions"})
        with open(os.path.join(prof_dir, "%s_compute.json" % tag), "w") as f:
            json.dump(comp, f, indent=1)
        print(json.dumps(comp, indent=1))
    sys.exit(0)
if len(sys.argv) > 3:          # `profile_sum.py <out_root> <tag> tiled`: only the per-kernel table of that trace
    sub = sys.argv[3]
    dur, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in rows(sub, "kernel_trace.csv"):
        dur[r["Kernel_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); cnt[r["Kernel_Name"]] += 1
    total = sum(dur.values()) or 1.0
    path = os.path.join(prof_dir, "%s_%s_kernel_stats.csv" % (tag, sub))
    what = {"tiled": "python scripts/dev_gpu_diag.py tileprof (the populations are printed in profiles/%s_%s_run.log)" % (tag, sub),
            "driver": "python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs (the command the round-end driver "
                      "runs; its JSON line is in profiles/%s_%s_run.log)" % (tag, sub)}.get(sub, sub)
    with open(path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s; durations in ns\n" % what)
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for k in sorted(dur, key=dur.get, reverse=True):
            f.write('"%s",%d,%.0f,%.1f,%.2f\n' % (k, cnt[k], dur[k], dur[k] / cnt[k], 100 * dur[k] / total))
        if sub == "driver" and dur:
            # every dispatch of the dominant kernel in time order: the pre-advance + warm-up launch, then the timed 20-step launch
            dom = max(dur, key=dur.get)
            disp = sorted((float(r["Start_Timestamp"]), float(r["End_Timestamp"])) for r in rows(sub, "kernel_trace.csv") if r["Kernel_Name"] == dom)
            f.write("# dispatches of the dominant kernel in time order, ns: %s  (the last one is the timed launch of --steps 20)\n" % ", ".join("%.0f" % (b - a) for a, b in disp))
    log = os.path.join(out_root, "prof_%s_%s.log" % (tag, sub))
    if os.path.exists(log):
        with open(os.path.join(prof_dir, "%s_%s_run.log" % (tag, sub)), "w") as f:
            f.write("".join(l for l in open(log) if l.startswith(("variant", "   broad", '{"metric"'))))
    print(open(path).read())
    sys.exit(0)

dur, cnt = collections.defaultdict(float), collections.defaultdict(int)
for r in rows("stats", "kernel_trace.csv"):
    dur[r["Kernel_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); cnt[r["Kernel_Name"]] += 1
total = sum(dur.values()) or 1.0
with open(os.path.join(prof_dir, "%s_bench_kernel_stats.csv" % tag), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-other-configs (N=1); durations in ns\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for k in sorted(dur, key=dur.get, reverse=True):
        f.write('"%s",%d,%.0f,%.1f,%.2f\n' % (k, cnt[k], dur[k], dur[k] / cnt[k], 100 * dur[k] / total))
dom = max(dur, key=dur.get) if dur else None
fetch, fn = counter_per_kernel("fetch")
write, wn = counter_per_kernel("write")
cf, _ = counter_per_kernel("calfetch")
cw, _ = counter_per_kernel("calwrite")
cal_bytes = float((1 << 28) * 8)
info = {"command": "python bench.py --no-cpu-baseline --no-other-configs", "dominant_kernel": dom,
        "units_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of 1024 B per count on this image; "
                      "raw counter sums are stored, bytes = raw * 1024"}
if dom:
    info["launches"] = cnt[dom]; info["avg_launch_ns"] = dur[dom] / cnt[dom]
    info["fetch_raw_per_launch"] = fetch.get(dom, 0.0) / max(1, fn.get(dom, 1))
    info["write_raw_per_launch"] = write.get(dom, 0.0) / max(1, wn.get(dom, 1))
for name, table, key in (("calib_read8", cf, "calib_fetch_raw"), ("calib_write8", cw, "calib_write_raw")):
    for k, v in table.items():
        if name in k:
            info[key] = v
info["calib_true_bytes"] = cal_bytes
with open(os.path.join(prof_dir, "%s_hbm_traffic.json" % tag), "w") as f:
    json.dump(info, f, indent=1)
print(json.dumps(info, indent=1))
print(open(os.path.join(prof_dir, "%s_bench_kernel_stats.csv" % tag)).read())
