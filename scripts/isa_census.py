"""Static census of one kernel's gfx950 ISA by instruction class and by source line (no GPU needed).
    cd evosoro_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm --cuda-device-only -S -gline-tables-only launch_tiled.hip -o /tmp/tiled_g.s
    (one translation unit per kernel family: launch_fused_land / _mesh, launch_wide, launch_tiled, engine = streaming kernels)
    python scripts/isa_census.py /tmp/engine_g.s 'k_robot_stepsILi768ELi2ELb0ELb0E' [--lines N] [--cls mov,cnd,...]
Every VALU wave-instruction costs a SIMD four cycles whether it is an FMA or a move, so for an FP64-issue-bound kernel the
non-FP64 share of the vector stream is pure overhead; this shows where it comes from."""
import re, sys, collections

def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
        if op.endswith("_f64") or "_f64_" in op: return "f64"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "mov"
        if op.startswith("v_cndmask"): return "cnd"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"): return "cmp"
        return "valu_other"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    return "other"

def main():
    path, pat = sys.argv[1], sys.argv[2]
    nlines = 40; want = None; root = False
    a = sys.argv[3:]
    while a:
        if a[0] == "--lines": nlines = int(a[1]); a = a[2:]
        elif a[0] == "--cls": want = set(a[1].split(",")); a = a[2:]
        elif a[0] == "--root": root = True; a = a[1:]      # attribute an instruction to the OUTERMOST call site (the line of the kernel that the inlined code belongs to)
        else: a = a[1:]
    files = {}
    inside = False
    cur = ("?", 0)
    tot = collections.Counter()
    per_line = collections.defaultdict(collections.Counter)
    with open(path) as f:
        for line in f:
            s = line.strip()
            m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
            if m:
                files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
                continue
            if not inside:
                if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", line): inside = True
                continue
            if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False: break
            if s.startswith(".loc"):
                p = s.split()
                cur = (files.get(int(p[1]), p[1]), int(p[2]))
                if root:
                    sites = re.findall(r"([\w.]+):(\d+):\d+ \]", s) or re.findall(r"@\[ \S*?([\w.]+):(\d+):\d+", s)
                    allsites = re.findall(r"@\[ \S*?/?([\w.]+\.h(?:pp)?):(\d+):\d+", s)
                    if allsites: cur = (allsites[-1][0], int(allsites[-1][1]))
                continue
            if not s or s.startswith((";", ".", "//")) or s.endswith(":"): continue
            op = s.split()[0]
            c = classify(op)
            tot[c] += 1
            per_line[cur][c] += 1
    valu = sum(tot[c] for c in ("f64", "mov", "cnd", "cmp", "lane", "valu_other"))
    print("VALU %d  f64 %d (%.0f%%)  mov %d  cnd %d  cmp %d  lane %d  other %d | SALU %d smem %d lds %d vmem %d scratch %d wait %d branch %d barrier %d" % (
        valu, tot["f64"], 100.0 * tot["f64"] / max(valu, 1), tot["mov"], tot["cnd"], tot["cmp"], tot["lane"], tot["valu_other"],
        tot["salu"], tot["smem"], tot["lds"], tot["vmem"], tot["scratch"], tot["wait"], tot["branch"], tot["barrier"]))
    key = (lambda c: sum(c[k] for k in want)) if want else (lambda c: sum(c[k] for k in ("mov", "cnd", "cmp", "lane", "valu_other")))
    rows = sorted(per_line.items(), key=lambda kv: -key(kv[1]))[:nlines]
    for (fn, ln), c in rows:
        print("%-22s %5d  nonf64 %4d  f64 %4d  mov %3d cnd %3d cmp %3d lane %3d oth %3d  salu %3d lds %3d vmem %3d scr %3d" % (
            fn, ln, sum(c[k] for k in ("mov", "cnd", "cmp", "lane", "valu_other")), c["f64"], c["mov"], c["cnd"], c["cmp"], c["lane"], c["valu_other"], c["salu"], c["lds"], c["vmem"], c["scratch"]))

main()
