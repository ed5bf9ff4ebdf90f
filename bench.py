#!/usr/bin/env python3
"""Benchmark of the hot path: voxel-timesteps/sec of a batch of random 10x10x10 soft robots (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload of the headline `value` = BASELINE.json configs[2], "population 512 of 10x10x10 robots": it fits one MI355X, so
a single GPU steps the whole population of 512; with N GPUs every rank steps its own 512 robots (WEAK scaling, seeds
offset by rank, no data-path collective).  A "step" is one Voxelyze TimeStep of every robot of the batch.  Robots:
material per voxel uniform on {0..4} with P(empty)=0.3, largest connected component, evosoro's default
materials/environment, self-collision ON, DtFrac 0.9.  Before anything is timed every robot is
advanced past InitCmTime (untimed), so that whatever window --steps selects, it samples the steady-state instruction mix:
actuation on, bonds in both the small- and the large-angle branch (`config.large_angle_bonds` = their share at the end
of the timed window).  After the timed region the fitness records of all ranks are gathered with one RCCL all_gather
(the path's only collective).

With N > 1 the same line carries `strong`: BASELINE configs[2] as stated -- ONE population of 512, partitioned over the
N ranks by cost (64 per GPU at N = 8) -- timed the same way right after the weak run.  At N = 1 the same statement is measured as
far as one GPU can (round 6): `other_configs` carries shard 0 of the 8 cost-balanced shards -- the 64 robots ONE of eight GPUs would
step -- with the default kernels and with tile_small, and the projected 8-GPU strong rate; `--shard-of K` times that shard as the
line's own workload.

READ THIS BEFORE COMPARING TWO LINES: `value` depends on --steps.  The timed region is cut into launches of at most 1024 steps (engine option steps_per_launch), a
launch of a self-colliding population carries ~0.07 ms of fixed cost (it ends with its slowest workgroup, and in every launch some
robots run a ~0.04 ms collision broad-phase; 0.27 and 0.17 ms until round 3), and a call ~0.04 ms on the host.  `--steps 20 --warmup 5`
(what the round-end driver runs) therefore reports ~1.4e10 voxel-steps/s by the HIP events (~25.5 us per step; ~1.3e10 by the host
clock, which was the line's `value` until round 4: 1.24e10 then, 1.03e10 in round 3, 7.4e9 in round 2), the default `--steps 2000`
~1.5e10 (~23.7-24.4 us) -- same kernel, same population; `timed_region` in the line says which case it is.

TIMING (round 5).  `value` / `ms_per_step` are taken from HIP events on the engine's own stream around the K timed steps
(vxh_counters.kernel_seconds of that one call), max over ranks -- the line says so itself: "clock": "hip_event" --; the host clock between the two barrier + torch.cuda.synchronize()
pairs the contract names is measured too and printed as `host_clock` (it adds ~0.04 ms of launch and wake-up latency per call, 8 %
of a 20-step region, none of it GPU work).

The JSON line also carries
  roofline      HBM: algorithmic bytes (224*Nvox + 144*Nbond per voxel-step) of the dominant kernel over its
                HIP-event time, in GB/s against the 8 TB/s peak; `traffic` = what the PMC counters saw for the same kernel on
                this workload (newest profiles/r*_hbm_traffic.json), also in GB/s -- rates, comparable whatever the launch length.
                `traffic` counts the L2's MEMORY-SIDE requests (FETCH_SIZE / WRITE_SIZE; Infinity-Cache hits included), i.e. an
                upper bound of the HBM bytes, and it lies far BELOW `achieved`: the bond history of the robots resident on an
                XCD fits its L2 (hit rate 0.97, profiles/r02_l2_counters.txt), so most algorithmic bytes never leave the chip.
                `achieved / peak` is therefore not a statement that the kernel is HBM-bound -- it is bound by FP64 issue
                (DESIGN.md section 4).
  roofline.binding  the bound that binds: FP64 flops per voxel-step of this (population, kernel) -- a property of the two, counted once (a STORED
                count: `live` is false for it; the rate it is multiplied by is this run's) --
                with the SQ_INSTS_VALU_*_F64 counters (profiles/r*_flops_per_unit.json, scripts/flops_per_unit.sh) -- times this run's own
                voxel-steps/s over its HIP-event time, against the 78.6 TFLOP/s vector-FP64 peak; every other_configs entry carries its own.
  ranks         N > 1: world size, backend, and per rank the device (index, PCI bus id) and its ms per step, gathered over the RCCL group
  other_configs the other BASELINE configs at their stated sizes (64 x 6^3 walkers, 64 x 8^3 swimmers, one 20^3 lattice), 512 dense 10^3,
                a mixed generation run to completion, and configs[2] as stated (one of 8 shards); N = 1 only: value, us per step,
                algorithmic roofline fraction, binding, kernel
  cpu_baseline  the reference C++ voxelyze (oracle/_ref/voxelyze_ref, built from the reference sources) on this box's host
                cores: `nproc` concurrent processes on 2 x nproc robots of the bench population (evaluation.py:89 launches
                one process per robot and lets the OS schedule them), plus the single-process figure, plus the single-process figure of the
                build the reference SHIPS (-O0 library objects; the others are -O3); rank 0, N = 1 only.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
INIT_CM_TIME = 0.05            # FitnessEvaluationInitTime of the bench robots


def make_population(tmp, count, first_seed, shape, sim_time, init_time, selfcol=True, env=None, variant_phase=False):
    from evosoro_amd import workloads
    from evosoro_amd.base import Sim, Env
    from evosoro_amd.tools.read_write_voxelyze import write_voxelyze_file
    for d in ("voxelyzeFiles", "fitnessFiles", "tempFiles"):
        os.makedirs(os.path.join(tmp, d), exist_ok=True)
    sim = Sim(self_collisions_enabled=selfcol, dt_frac=0.9, simulation_time=sim_time, fitness_eval_init_time=init_time)
    env = env or Env()
    paths = []
    for i in range(count):
        ind = workloads.random_robot(first_seed + i, shape, first_seed + i, phase_offset=variant_phase)
        write_voxelyze_file(sim, env, ind, tmp, "bench")
        paths.append(os.path.join(tmp, "voxelyzeFiles", "bench--id_%05i.vxa" % ind.id))
    return paths


def run_reference(ref, paths, tmp, concurrency):
    """the reference binary on every file, `concurrency` processes at a time; wall seconds"""
    t0 = time.time()
    running, queue = [], list(paths)
    while queue or running:
        while queue and len(running) < concurrency:
            running.append(subprocess.Popen(["timeout", "900", ref, "-f", queue.pop(0)], cwd=tmp,
                                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        running = [p for p in running if p.poll() is None]
        time.sleep(0.002)
    return time.time() - t0


def live_parity(paths, tmp, device):
    """The result files the reference binary has just written for the cpu_baseline sample (tmp/fitnessFiles), against the engine's for the
    same .vxa files, evaluated now on `device`: how many are byte-identical (every tag printed with six significant digits, as the
    reference prints them), and the largest difference of any numeric tag otherwise.  The reference's files are the checker; the
    engine's run is not part of any timed figure."""
    import re
    from evosoro_amd import engine
    tag = re.compile(r"<(\w+)>\s*([-+0-9.eE]+|nan|inf|-inf|-nan)\s*</\1>")
    with engine.Engine(engine.VOXCAD, device) as eng:
        eng.add_vxa_files(paths)
        names = [eng.fitness_file_name(i) for i in range(len(paths))]
        ref_text = []
        for n in names:
            with open(n if os.path.isabs(n) else os.path.join(tmp, n)) as f:
                ref_text.append(f.read())
        eng.run()
        steps = sum(eng.result(i).steps for i in range(len(paths)))
        out_dir = os.path.join(tmp, "engineFitness")
        os.makedirs(out_dir, exist_ok=True)
        identical, worst, worst_tag, tags = 0, 0.0, None, 0
        for i in range(len(paths)):
            mine = os.path.join(out_dir, "%05d.xml" % i)
            eng.write_result_xml(i, mine)
            with open(mine) as f:
                text = f.read()
            if text == ref_text[i]:
                identical += 1
            a, b = dict(tag.findall(ref_text[i])), dict(tag.findall(text))
            if set(a) != set(b):
                raise RuntimeError("result file %d: tags differ from the reference's: %s" % (i, sorted(set(a) ^ set(b))))
            tags = len(a)
            for k in a:
                d = abs(float(a[k]) - float(b[k]))
                if d > worst:
                    worst, worst_tag = d, k
    return {"live": True, "robots": len(paths), "steps": int(steps), "numeric_tags_per_file": tags, "result_files_byte_identical": identical,
            "worst_tag_difference": worst, "worst_tag": worst_tag,
            "against": "the reference binary's result files for the cpu_baseline sample (oracle/_ref/voxelyze_ref, this run)",
            "note": "tags are printed with six significant digits by both; a difference is in the tag's own unit (metres / voxels as the tag says)"}


def cpu_baseline(shape, device=None):
    """Reference voxelyze (oracle/_ref) on the host cores.  Sample: 2 x nproc robots of the bench population (at most 512),
    0.12 s of simulated time each (about 1900 steps), nproc processes at a time, like evaluation.py:89 -- sized for roughly
    10-30 s of wall clock; then ONE robot alone for the single-core figure.  With a `device`: the result files of that sample
    are compared with the engine's for the same robots (live_parity), returned under "parity_check"."""
    from evosoro_amd import engine
    ref = os.path.join(REPO, "oracle", "_ref", "voxelyze_ref")
    nproc = os.cpu_count() or 1
    sim_time = 0.12
    tmp = tempfile.mkdtemp(prefix="vxbench_cpu_")
    try:
        paths = make_population(tmp, min(512, 2 * nproc), 0, shape, sim_time, 0.05)

        def work_of(files):
            total = 0.0
            for p in files:
                d = engine.inspect_vxa(p)
                total += d.nvox * d.planned_steps
            return total
        if os.path.exists(ref):
            # all hardware threads busy, and one process per physical core (two threads of a core share its FP units): the better counts
            walls = {}
            for used in sorted(set([nproc, max(1, nproc // 2)]), reverse=True):
                for f in os.listdir(os.path.join(tmp, "fitnessFiles")):
                    os.remove(os.path.join(tmp, "fitnessFiles", f))
                walls[used] = run_reference(ref, paths, tmp, used)
                done = len([f for f in os.listdir(os.path.join(tmp, "fitnessFiles")) if f.endswith(".xml")])
                if done != len(paths):
                    raise RuntimeError("reference finished %d of %d robots" % (done, len(paths)))
            used = min(walls, key=walls.get)
            wall = walls[used]
            check = live_parity(paths, tmp, device) if device is not None else None
            one = run_reference(ref, paths[:1], tmp, 1)
            # context: the same robot with the optimisation level the reference SHIPS with (its library objects are compiled with an
            # empty CXXFLAGS = -O0); every other figure here is the -O3 build, which flatters the reference
            ref_O0 = ref + "_O0"
            one_O0 = run_reference(ref_O0, paths[:1], tmp, 1) if os.path.exists(ref_O0) else None
            return {"value": work_of(paths) / wall, "unit": "voxel-timesteps/s", "cores": used, "kind": "reference",
                    "nproc": nproc, "parity_check": check, "single_core_value": work_of(paths[:1]) / one,
                    "single_core_value_shipped_flags": (work_of(paths[:1]) / one_O0) if one_O0 else None,
                    "sample": "%d random %dx%dx%d robots of the bench population, %.2f s simulated each (%.3g voxel-steps), one process per "
                              "robot (evaluation.py:89), g++ -O3 build of the reference sources, host with %d hardware threads: %s; "
                              "value = the faster; single_core_value: one robot alone, %.1f s; single_core_value_shipped_flags: the same robot "
                              "with the reference's own -O0 library build, %s"
                              % (len(paths), shape[0], shape[1], shape[2], sim_time, work_of(paths), nproc,
                                 ", ".join("%d at a time %.1f s" % (u, w) for u, w in sorted(walls.items())), one,
                                 ("%.1f s" % one_O0) if one_O0 else "binary not built")}
        from oracle import vxoracle
        t0 = time.time()
        for p in paths[:4]:
            sim = vxoracle.OracleSim.from_vxa(p)
            sim.step(-1)
        wall = time.time() - t0
        return {"value": work_of(paths[:4]) / wall, "unit": "voxel-timesteps/s", "cores": 1, "kind": "port", "nproc": nproc,
                "single_core_value": work_of(paths[:4]) / wall,
                "sample": "4 robots of the bench population through the C restatement (oracle/), one core, %.1f s" % wall}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measured_traffic(robots_per_gpu, lattice):
    """GB/s between the L2 and the memory side (an upper bound of the HBM GB/s: Infinity-Cache hits are counted) that the dominant
    kernel really moved, from the newest rocprofv3 counter summary under profiles/
    (scripts/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate passes over this same command; FETCH_SIZE scaled by
    the calibration kernels of the same passes).  PMC counters cannot be read from inside this process, so the figure is
    only reported for the profiled workload; it is a RATE (bytes of a launch over that launch's duration in the profile)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hbm_traffic.json")))
    if not files or robots_per_gpu != 512 or lattice != 10:
        return {}
    with open(files[-1]) as f:
        t = json.load(f)
    if not t.get("avg_launch_ns"):
        return {}
    fetch_scale = t["calib_true_bytes"] / (t["calib_fetch_raw"] * 1024.0) if t.get("calib_fetch_raw") else 2.0
    write_scale = t["calib_true_bytes"] / (t["calib_write_raw"] * 1024.0) if t.get("calib_write_raw") else 1.0
    nbytes = t["fetch_raw_per_launch"] * 1024.0 * fetch_scale + t["write_raw_per_launch"] * 1024.0 * write_scale
    return {"traffic": nbytes / t["avg_launch_ns"],
            "traffic_source": os.path.relpath(files[-1], REPO) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              "command; FETCH x%.2f, WRITE x%.2f from the calibration kernels; L2 memory-side bytes incl. Infinity-Cache hits = "
                              "upper bound of HBM bytes)" % (fetch_scale, write_scale)}


def kernel_name(block):
    """vxh_counters.dominant_block -> kernel: 0 streaming, 1 tiled, workgroup size of the resident kernel, workgroup size + 1 of the wide one"""
    if block == 1:
        return "k_tile_steps"
    if block == 1026:
        return "k_robot_pair<512 x 2,...>"
    if block > 1 and block % 64 == 1:
        return "k_robot_wide<%d,...>" % (block - 1)
    return ("k_robot_steps<%d,...>" % block) if block else "k_bonds+k_voxels"


def measured_compute(kernel):
    """What the SQ counters say binds the dominant kernel (newest profiles/r*_compute.json, written by scripts/profile_sum.py from
    separate rocprofv3 --pmc passes over this command): share of the SIMD cycles in which a vector instruction was executing,
    share of wavefront time spent waiting, LDS bank-conflict ratio, and the FP64 rate those instructions amount to against the
    vector-FP64 peak.  Counters cannot be read from inside this process: reported only when the profile is of this kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_compute.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        c = json.load(f)
    if kernel.split("<")[0] not in c.get("kernel", "") or kernel.split("<")[-1].split(",")[0] not in c.get("kernel", ""):
        return None
    c["source"] = os.path.relpath(files[-1], REPO)
    c["live"] = False          # SQ counters of a committed rocprofv3 profile of this command, not of this very run
    return c


def measured_parity():
    """The other half of BASELINE's metric ("CoM-displacement err vs CPU"): the newest committed parity ledgers
    (tests/test_gpu_ledger.py -> profiles/r*_parity_{auto,narrow,tiles3}.json, one per kernel path): every golden case run to its stop
    condition on the engine, error of the final centre of mass against the reference binary's, in voxels.  A record of the last GPU
    test run, not of this bench run."""
    import glob
    out = None
    for path_name in ("auto", "narrow", "tiles3"):
        files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_parity_%s.json" % path_name)))
        if not files:
            continue
        with open(files[-1]) as f:
            led = json.load(f)
        rows = led.get("rows", [])
        strict_rows = [r for r in rows if r.get("strict_1e-9")]
        worst_row = max(rows, key=lambda r: max(r["err_cur_cm_vox"], r["err_ini_cm_vox"])) if rows else None
        entry = {"cases": led.get("cases"), "strict_1e-9": led.get("strict_1e-9"), "within_1e-12": led.get("within_1e-12"),
                 "within_tolerance": led.get("within_tolerance"),
                 "worst_err_vox": max([max(r["err_cur_cm_vox"], r["err_ini_cm_vox"]) for r in rows] or [None]),
                 "worst_case": worst_row["case"] if worst_row else None,
                 "worst_err_vox_of_the_strict_cases": max([max(r["err_cur_cm_vox"], r["err_ini_cm_vox"]) for r in strict_rows] or [None]),
                 "bench_robot_err_vox": max([max(r["err_cur_cm_vox"], r["err_ini_cm_vox"]) for r in rows if r["case"].startswith("bench10")] or [None]),
                 "source": os.path.relpath(files[-1], REPO)}
        if out is None:
            out = dict(entry)
            out["quantity"] = "max over x, y, z of |final centre of mass - reference binary's| (and IniCM) after the whole evaluation, voxels"
            out["live"] = False
            out["kernel_paths"] = {}
        out["kernel_paths"][path_name] = entry
    return out


FP64_PEAK_FLOPS = 78.6e12      # vector FP64: one FP64 wave-instruction per 4 cycles and SIMD at 2.4 GHz (half the FP32 vector peak of MI355X_MICROARCH.md)
VALU_ISSUE_PER_S = 256 * 4 * 2.4e9 / 4.0      # vector wave-instructions the chip can issue per second (1024 SIMDs, one per 4 cycles)


def flops_per_unit():
    """newest profiles/r*_flops_per_unit.json: workload key -> {fp64_flop_per_voxel_step, valu_inst_per_voxel_step, kernels}"""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_flops_per_unit.json")))
    if not files:
        return {}, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], REPO)


def binding(key, voxel_steps_per_s):
    """roofline.binding of a workload: its counted FP64 flops (and vector instructions) per voxel-step x THIS run's rate"""
    table, src = flops_per_unit()
    row = table.get(key)
    if not row or not voxel_steps_per_s:
        return None
    flops = row["fp64_flop_per_voxel_step"] * voxel_steps_per_s
    out = {"bound": "fp64-issue", "achieved": flops, "peak": FP64_PEAK_FLOPS, "unit": "FLOP/s", "frac": flops / FP64_PEAK_FLOPS,
           "fp64_flop_per_voxel_step": row["fp64_flop_per_voxel_step"],
           "live": False,     # (the rate is this run's; the flops per voxel-step are a stored count -- advisor, round 5)
           "live_part": "voxel-steps/s", "stored_part": "flops and vector instructions per voxel-step",
           "source": "%s (flops per voxel-step: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 of this workload) x this run's voxel-steps/s" % src}
    if row.get("valu_inst_per_voxel_step"):
        out["valu_issue_frac"] = row["valu_inst_per_voxel_step"] * voxel_steps_per_s / VALU_ISSUE_PER_S
        out["valu_inst_per_voxel_step"] = row["valu_inst_per_voxel_step"]
    return out


_TRACE = None


def trace_mark(what, t0, t1, rank):
    if _TRACE:
        with open(_TRACE, "a") as f:
            f.write(json.dumps({"rank": rank, "op": what, "group": "-", "t0": t0, "t1": t1}) + "\n")


def trace_collectives(dist, ctl, rank):
    """VXH_BENCH_TRACE=<file prefix>: every collective this process issues is logged with its group and wall-clock span, and so are the
    timed regions (tests/test_bench_cli.py: no collective of the default group -- RCCL on a GPU node -- may overlap a timed region on
    any rank).  Off by default; wraps the torch.distributed entry points bench.py and evosoro_amd.parallel use."""
    global _TRACE
    prefix = os.environ.get("VXH_BENCH_TRACE")
    if not prefix:
        return
    _TRACE = "%s.rank%d.jsonl" % (prefix, rank)

    def wrap(name):
        inner = getattr(dist, name)

        def traced(*a, **k):
            t0 = time.time()
            try:
                return inner(*a, **k)
            finally:
                g = k.get("group")
                with open(_TRACE, "a") as f:
                    f.write(json.dumps({"rank": rank, "op": name, "group": "ctl" if (g is not None and g is ctl) else "default", "t0": t0, "t1": time.time()}) + "\n")
        setattr(dist, name, traced)
    for name in ("barrier", "all_reduce", "broadcast_object_list", "all_gather_into_tensor", "all_gather", "broadcast", "gather", "all_gather_object"):
        if hasattr(dist, name):
            wrap(name)


def relaunch_under_launcher(n_gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks of this same command under
    torch.distributed.run and pass its exit code on -- instead of silently measuring one GPU."""
    import socket
    import torch
    share = os.environ.get("VXH_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not share and have < n_gpus:
        sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible (VXH_BENCH_SHARE_GPU=1 walks the N-rank path on one device)\n" % (n_gpus, have))
        raise SystemExit(2)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    entry = os.environ.get("VXH_BENCH_ENTRY", os.path.abspath(__file__))      # (tests launch a wrapper that stubs the engine)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), entry] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")))


def timed_steps(eng, steps, barrier=None):
    """EXACTLY `steps` time steps of every robot, bracketed the way the driver asks: barrier + device sync on both sides.  Returns
    (device seconds, host seconds): the HIP-event time of that one call on the engine's own stream (vxh_counters.kernel_seconds: first
    event in front of the call's first launch, last behind its last) and the host clock from behind the opening barrier + sync to
    behind the rank's own closing sync.  The line reports the MAX over ranks of each (reduce_stats); `value` is priced on the device
    time -- the host clock of a 20-step region carries ~0.04 ms of launch + wake-up latency and the two synchronisations, none of it
    work of the path (round-4 review, task 6)."""
    import torch
    if barrier:
        barrier()
    torch.cuda.synchronize()
    k0 = eng.counters().kernel_seconds
    w0 = time.time()
    t0 = time.perf_counter()
    eng.step(steps)                      # (returns after the engine's own stream synchronisation)
    torch.cuda.synchronize()
    host = time.perf_counter() - t0
    trace_mark("timed_region", w0, time.time(), int(os.environ.get("RANK", "0")))
    if barrier:
        barrier()
    return eng.counters().kernel_seconds - k0, host


def side_config(engine, name, variant, count, shape, env, device, steps, full=False, phase=False, init_time=0.02, key=None, options=None):
    """one of the other BASELINE configs at its stated size: pre-advanced past InitCmTime, then `steps` timed steps"""
    from evosoro_amd import workloads
    from evosoro_amd.base import Sim
    from evosoro_amd.tools.read_write_voxelyze import write_voxelyze_file
    tmp = tempfile.mkdtemp(prefix="vxbench_side_")
    try:
        os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
        # (long enough for the robot with the largest time step: nothing is run to its stop condition here)
        sim = Sim(dt_frac=0.9, simulation_time=init_time + (steps + 4000) * 7.2e-4, fitness_eval_init_time=init_time)
        with engine.Engine(variant, device) as eng:
            for k, val in (options or {}).items():
                eng.set_option(k, val)
            for i in range(count):
                if full:
                    ind = workloads.make_individual(i, workloads.full_material(shape[0], 1 + i))
                else:
                    ind = workloads.random_robot(i, shape, i, phase_offset=phase)
                write_voxelyze_file(sim, env, ind, tmp, "s")
                eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "s--id_%05i.vxa" % i))
            dims = [eng.dims(i) for i in range(count)]
            nvox = sum(d["nvox"] for d in dims)
            nbond = sum(d["nbond"] for d in dims)
            pre = int(max(init_time / d["dt"] for d in dims)) + 32
            eng.step(pre)
            c0 = eng.counters()
            elapsed, host = timed_steps(eng, steps)
            c1 = eng.counters()
            assert abs((c1.voxel_steps - c0.voxel_steps) - float(nvox) * steps) < 0.5, name
            alg = (224.0 * nvox + 144.0 * nbond) * steps
            large, total = eng.bond_modes()
            return {"workload": name, "value": nvox * steps / elapsed, "unit": "voxel-timesteps/s",
                    "us_per_step": elapsed / steps * 1e6, "us_per_step_host_clock": host / steps * 1e6, "steps": steps, "voxels": nvox, "bonds": nbond,
                    "kernel": kernel_name(c1.dominant_block),
                    # SURVEY 8(d)'s EFFECTIVE figure: algorithmic bytes over the kernel's time against 8 TB/s.  Not a roof for this design
                    # (the state is register / LDS / L2-resident: real traffic is a fifth of it), so it may exceed 1; `binding` is the bound
                    "effective_hbm": {"achieved_GBs": alg / c1.dominant_seconds / 1e9 if c1.dominant_seconds > 0 else None, "peak_GBs": HBM_PEAK_GBS,
                                      "frac_of_alg_bytes": alg / c1.dominant_seconds / 1e9 / HBM_PEAK_GBS if c1.dominant_seconds > 0 else None,
                                      "note": "algorithmic bytes, not traffic: not a bound"},
                    "binding": binding(key or name, nvox * steps / elapsed),
                    "large_angle_bonds": large / max(1, total), "_voxel_steps_process": c1.voxel_steps}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def mixed_generation(engine, device, count=512, lattice=10, sim_time=0.5, init_time=0.05, options=None):
    """A realistic generation (SURVEY.md section 7 "Heterogeneous robots in one batch", evosoro examples/basic.py:114-127): `count`
    robots on a lattice^3 grid whose fill is uniform on 30-100 %, every second one without bone -- its softest-material time step is
    ten times the others', so it needs a tenth of the steps for the same simulated time -- evaluated TO COMPLETION by one vxh_run.
    Robots of several size classes (kernel variants, launch groups on their own streams) that stop at very different step counts:
    the case the headline population (one variant, one step count) does not exercise.  Reported: voxel-steps/s of the whole run over
    the GPU's own time for it and over the wall clock of the call, and the TAIL EFFICIENCY = that rate over the rate the same engine
    reaches on the same population while every robot is still stepping (steps 100 .. 100 + `steady`, the robots without bone already
    actuating): what the run loses to launch groups draining at different times, to robots that stop in the middle
    of a launch, and to the part of the run in which only the long robots are left."""
    import numpy as np
    from evosoro_amd import workloads
    from evosoro_amd.base import Sim, Env
    from evosoro_amd.tools.read_write_voxelyze import write_voxelyze_file
    tmp = tempfile.mkdtemp(prefix="vxbench_mixed_")
    try:
        os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
        sim = Sim(self_collisions_enabled=True, dt_frac=0.9, simulation_time=sim_time, fitness_eval_init_time=init_time)
        rng = np.random.RandomState(4242)
        paths = []
        for i in range(count):
            fill = rng.uniform(0.3, 1.0)
            mat = workloads.random_material((lattice,) * 3, 7000 + i, p_empty=1.0 - fill)
            if i % 2:
                mat = np.where(mat == 2, 1, mat)          # no bone: fat instead
            write_voxelyze_file(sim, Env(), workloads.make_individual(i, mat), tmp, "g")
            paths.append(os.path.join(tmp, "voxelyzeFiles", "g--id_%05i.vxa" % i))
        steady = 256
        with engine.Engine(engine.VOXCAD, device) as eng:
            eng.add_vxa_files(paths)
            dims = [eng.dims(i) for i in range(count)]
            pre = 100          # (past InitCmTime of the fast-stepping half; the robots with bone get there at step ~780, when the others are done)
            assert pre + steady < min(d["planned_steps"] for d in dims), (pre, steady, min(d["planned_steps"] for d in dims))
            eng.step(pre)
            c0 = eng.counters()
            timed_steps(eng, steady)
            c1 = eng.counters()
            rate_steady = (c1.voxel_steps - c0.voxel_steps) / (c1.kernel_seconds - c0.kernel_seconds)
            vs_process = c1.voxel_steps
        with engine.Engine(engine.VOXCAD, device) as eng:
            for key, val in (options or {}).items():
                eng.set_option(key, val)
            eng.add_vxa_files(paths)
            import torch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run()
            wall = time.perf_counter() - t0
            c = eng.counters()
            finished = sum(1 for i in range(count) if eng.result(i).status == engine.ROBOT_FINISHED)
        # which kernel steps a robot is a function of the robot alone (engine.hip fused_variant): up to 512 voxels and 1023 bonds
        # k_robot_wide<512>, else k_robot_steps<512 / 768 / 1024> by voxel count
        blocks = {}
        for d in dims:
            k = ("k_robot_wide<512>" if (d["nvox"] <= 512 and d["nbond"] <= 1023) else
                 "k_robot_steps<%d>" % (512 if d["nvox"] <= 512 else (768 if d["nvox"] <= 768 else 1024)))
            blocks[k] = blocks.get(k, 0) + 1
        nv = np.array([d["nvox"] for d in dims]); st = np.array([d["planned_steps"] for d in dims])
        return {"workload": "mixed generation: %d random %d^3 robots, fill uniform 30-100 %%, every second one without bone (10x fewer steps), "
                            "self-collision on, %.2f s simulated, vxh_run to completion" % (count, lattice, sim_time),
                "value": c.voxel_steps / c.kernel_seconds, "unit": "voxel-timesteps/s", "value_wall": c.voxel_steps / wall,
                "voxel_steps": c.voxel_steps, "gpu_seconds": c.kernel_seconds, "wall_seconds": wall, "launches": int(c.launches),
                "steady_rate": rate_steady, "tail_efficiency": (c.voxel_steps / c.kernel_seconds) / rate_steady,
                "robots_finished": finished, "voxels_min_mean_max": [int(nv.min()), float(nv.mean()), int(nv.max())],
                "steps_min_max": [int(st.min()), int(st.max())], "kernels": blocks,
                "binding": binding("mixed", c.voxel_steps / c.kernel_seconds), "_voxel_steps_process": vs_process + c.voxel_steps,
                "note": "value = sum(nvox x steps) / GPU time of the whole run (first event to last); value_wall = over the wall clock of the "
                        "vxh_run call incl. batch assembly and upload; tail_efficiency = value / steady_rate, steady_rate = the same engine on "
                        "the same population over %d steps in which every robot still steps" % steady}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def shard_of_population(engine, paths, device, parts, steps, init_time=INIT_CM_TIME):
    """BASELINE configs[2] AS STATED -- a population of 512 sharded across 8 GPUs -- measured on ONE GPU: shard 0 of the cost-balanced (greedy
    LPT on voxels, evosoro_amd/parallel.py shard_by_cost: what rank 0 of an 8-rank run steps) of the bench population, with the engine's
    default kernels and with the option tile_small; the 8-GPU strong rate is then a projection from a measurement, not from a model
    (round-5 review, task 2).  Every shard holds the same number of voxels to within a robot, so the projection is `parts` x this shard."""
    from evosoro_amd import parallel
    costs = [engine.inspect_vxa(p).nvox for p in paths]
    shards = parallel.shard_by_cost(costs, parts)
    loads = [sum(costs[i] for i in sh) for sh in shards]
    mine = shards[0]
    out = {"workload": "configs[2] as stated: shard 0 of %d (greedy LPT by voxels) of the population of %d random 10x10x10 robots = what ONE of %d GPUs steps"
                       % (parts, len(paths), parts),
           "robots": len(mine), "shard_voxels_min_max": [min(loads), max(loads)], "steps": steps}
    for label, options in (("default_kernels", {}), ("tile_small", {"tile_small": 1})):
        with engine.Engine(engine.VOXCAD, device) as eng:
            for k, val in options.items():
                eng.set_option(k, val)
            for i in mine:
                eng.add_vxa_file(paths[i])
            dims = [eng.dims(i) for i in range(len(mine))]
            nvox = sum(d["nvox"] for d in dims)
            eng.step(int(max(init_time / d["dt"] for d in dims)) + 32 + 200)
            elapsed, host = timed_steps(eng, steps)
            c1 = eng.counters()
            out[label] = {"value": nvox * steps / elapsed, "unit": "voxel-timesteps/s", "us_per_step": elapsed / steps * 1e6,
                          "us_per_step_host_clock": host / steps * 1e6, "kernel": kernel_name(c1.dominant_block),
                          "projected_%d_gpu_strong_value" % parts: sum(loads) * steps / elapsed,
                          "binding": binding("shard8" if not options else "shard8_tiles", nvox * steps / elapsed)}
    out["voxels"] = nvox
    best = max(("default_kernels", "tile_small"), key=lambda k: out[k]["value"])
    out["faster"] = best
    out["projection_note"] = ("projected strong value = voxels of the WHOLE population x steps / this shard's time (all %d shards are LPT-balanced to within "
                              "%.2f %% of voxels; no data-path collective; the fitness gather is outside the timed steps)" % (parts, 100.0 * (max(loads) - min(loads)) / max(loads)))
    return out


def side_workload(engine, key, device):
    """the other_configs entries by key (also what scripts/unit_workload.py runs under the FP64 instruction counters)"""
    from evosoro_amd.base import Env
    if key == "cfg1":
        return side_config(engine, "configs[1]: batch of 64 random 6x6x6 robots", engine.VOXCAD, 64, (6, 6, 6), Env(), device, 1024, key=key)
    if key == "cfg3":
        env_w = Env()
        env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
        env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
        return side_config(engine, "configs[3]: 64 random 8x8x8 swimmers (_voxcad_land_water, fluid drag)", engine.VOXCAD_LAND_WATER,
                           64, (8, 8, 8), env_w, device, 1024, phase=True, init_time=0.005, key=key)
    if key == "cfg4":
        return side_config(engine, "configs[4]: one full 20x20x20 lattice, self-collision on", engine.VOXCAD, 1, (20, 20, 20), Env(),
                           device, 2048, full=True, init_time=0.005, key=key)
    if key == "dense":
        # the resident kernel's largest variant: a real 10^3 population (fill 30-100 %) puts its fuller robots here
        return side_config(engine, "512 dense 10x10x10 robots (1000 voxels each), self-collision on", engine.VOXCAD, 512, (10, 10, 10), Env(),
                           device, 512, full=True, init_time=0.01, key=key)
    if key == "mixed":
        return mixed_generation(engine, device)
    raise KeyError(key)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--robots-per-gpu", type=int, default=512)
    ap.add_argument("--lattice", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--shard-of", type=int, default=0, metavar="K",
                    help="time shard 0 of K (greedy LPT by voxels) of the population instead of the whole of it: BASELINE configs[2] as one of K GPUs sees it")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from evosoro_amd import engine, parallel
    from evosoro_amd.base import Env

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_launcher(args.gpus)       # (does not return)
    # ONE JSON line on stdout, nothing else: libraries that write to the C stdout of their own accord (RCCL prints a five-line version
    # banner per rank on this image, buffered until the process ends -- i.e. BEHIND the JSON line) are pointed at stderr; the line itself
    # goes to the descriptor stdout was (the ranks a launcher started inherit the launcher's).
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or os.environ.get("VXH_FORCE_DIST") == "1"   # (the latter: 1-rank RCCL smoke test)
    if args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE %d\n" % (args.gpus, world))
        raise SystemExit(2)
    # VXH_BENCH_SHARE_GPU=1: every rank on device 0 and gloo instead of RCCL -- only to walk the N > 1 code path on a 1-GPU box
    share_gpu = os.environ.get("VXH_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    # CONTROL PLANE vs DATA PLANE.  Everything that merely coordinates the ranks -- the barriers that bracket the timed regions, the
    # max / sum of the per-rank clocks, the file list of the strong run, the waits around the multi_handle run -- goes over a
    # HOST-side (gloo) group, `ctl`.  An RCCL collective is a kernel plus a stream wait on every rank: an RCCL barrier in front of
    # a timed region delayed the engine's first launch by ~70 us (1-rank RCCL run on this image: 0.77 against 0.69 ms for the
    # 20-step region), and an RCCL barrier other ranks sit in while rank 0 times the multi_handle route SPINS ON THEIR GPUs -- the
    # very devices rank 0 is measuring (round-3 review).  The default (RCCL) group is used for exactly one thing, the path's own
    # collective: the fitness gather, outside every timed region.  With VXH_BENCH_HOST_BARRIER=0, or when no host interface for
    # gloo exists, `ctl` falls back to the default group and the line says so (`control_plane`).
    ctl = None                                # process group of the control plane (None: the default group)
    control_plane = "single process"
    if distributed:
        control_plane = "the default group (%s)" % ("gloo" if share_gpu else "RCCL: no host-side group")
        if os.environ.get("VXH_BENCH_HOST_BARRIER", "1") == "1":
            try:
                ctl = dist.new_group(backend="gloo")
                control_plane = "gloo host group; the default group (%s) carries the fitness gather only" % ("gloo" if share_gpu else "RCCL")
            except Exception as exc:             # (no usable host interface for gloo: the default group it is)
                sys.stderr.write("bench.py: host-side control group unavailable (%s); using the default group\n" % exc)
        trace_collectives(dist, ctl, rank)
    ctl_on_host = distributed and (share_gpu or ctl is not None)
    barrier = (lambda: dist.barrier(group=ctl)) if distributed else None

    shape = (args.lattice,) * 3
    n_local = args.robots_per_gpu

    def run_population(paths, options=None):
        """pre-advance past InitCmTime + warmup (untimed), then the timed steps; returns everything the line needs"""
        eng = engine.Engine(engine.VOXCAD, local_rank)
        for key, val in (options or {}).items():
            eng.set_option(key, val)
        for p in paths:
            eng.add_vxa_file(p)
        n = len(paths)
        dims = [eng.dims(i) for i in range(n)]
        nvox, nbond = sum(d["nvox"] for d in dims), sum(d["nbond"] for d in dims)
        pre = (int(max(INIT_CM_TIME / d["dt"] for d in dims)) + 32) if n else 0
        if n:
            eng.step(pre + max(args.warmup, 1))             # upload, past InitCmTime, warmup: all untimed
        c0 = eng.counters()
        elapsed, host = timed_steps(eng, args.steps, barrier)
        c1 = eng.counters()
        local_vs = c1.voxel_steps - c0.voxel_steps
        assert abs(local_vs - float(nvox) * args.steps) < 0.5, "a robot stopped inside the timed region"
        return eng, (elapsed, host), local_vs, nvox, nbond, c0, c1, pre

    def reduce_stats(clocks, local_vs):
        """(device seconds, host seconds) of this rank -> MAX over ranks of each; voxel-steps -> SUM"""
        if not distributed:
            return clocks[0], clocks[1], local_vs
        stats = torch.tensor([clocks[0], clocks[1], local_vs], dtype=torch.float64, device="cpu" if ctl_on_host else "cuda")
        tmax = stats.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=ctl)
        tsum = stats.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM, group=ctl)
        return float(tmax[0]), float(tmax[1]), float(tsum[2])

    def gather_ranks(ms_per_step):
        """proof that `world` ranks on `world` devices took part: one all_gather over the DEFAULT group (RCCL on a GPU node) of
        (rank, device index, PCI domain / bus / device of that device, this rank's ms per step)"""
        try:
            props = torch.cuda.get_device_properties(local_rank)
            ident = [float(torch.cuda.current_device()), float(getattr(props, "pci_domain_id", -1)), float(getattr(props, "pci_bus_id", -1)),
                     float(getattr(props, "pci_device_id", -1))]
        except Exception:                      # (the CPU walk-through of tests/test_bench_cli.py: no device behind the rank)
            ident = [float(local_rank), -1.0, -1.0, -1.0]
        mine = [float(rank)] + ident + [float(ms_per_step)]
        if not distributed:
            rows = [mine]
        else:
            dev = torch.device("cpu") if share_gpu else torch.device("cuda", local_rank)
            t = torch.tensor(mine, dtype=torch.float64, device=dev)
            got = torch.zeros(world * len(mine), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(got, t)
            rows = got.cpu().reshape(world, len(mine)).tolist()
        return {"world": world, "backend": (dist.get_backend() if distributed else "none"),
                "device_of_rank": [int(r[1]) for r in rows], "pci_of_rank": ["%04x:%02x:%02x" % (int(r[2]) & 0xffff, int(r[3]) & 0xff, int(r[4]) & 0xff) for r in rows],
                "distinct_devices": len({(r[2], r[3], r[4]) for r in rows}), "ms_per_step_of_rank": [r[5] for r in rows]}

    tmp = tempfile.mkdtemp(prefix="vxbench_r%d_" % rank)
    try:
        # long enough that no robot reaches its stop condition inside pre-advance + warmup + timed steps
        # (sized for the largest time step a robot can have, ten times the usual one: nothing is run to its stop condition)
        sim_time = max(0.5, INIT_CM_TIME + (args.steps + args.warmup + 1100) * 7.2e-4)
        paths = make_population(tmp, n_local, rank * n_local, shape, sim_time, INIT_CM_TIME)
        all_local_paths = paths
        if args.shard_of > 1:
            # (--shard-of K: this rank steps shard 0 of K of its population -- what one of K GPUs sees of BASELINE configs[2])
            picked = parallel.shard_by_cost([engine.inspect_vxa(p).nvox for p in paths], args.shard_of)[0]
            paths = [paths[i] for i in picked]
            n_local = len(paths)
        eng, clocks, local_vs, nvox, nbond, c0, c1, pre = run_population(paths)
        elapsed_max, host_max, total_vs = reduce_stats(clocks, local_vs)
        ranks = gather_ranks(clocks[0] / args.steps * 1e3)          # (outside every timed region)
        large, total_b = eng.bond_modes()
        # the path's collective: fitness records of every rank to every rank (untimed region, reported separately)
        tg = time.perf_counter()
        records = np.stack([parallel.result_to_record(eng.result(i)) for i in range(n_local)])
        table = parallel.gather_records(records, list(range(rank * n_local, (rank + 1) * n_local)), world * n_local,
                                        torch.device("cuda", local_rank) if (distributed and not share_gpu) else None)
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert table.shape[0] == world * n_local and (table[:, 2] > 0).all()
        eng.close()

        strong = None
        if world > 1:
            # BASELINE configs[2] as stated: ONE population of 512, partitioned over the ranks by cost (LPT on voxels x steps)
            shared = [make_population(os.path.join(tmp, "strong"), n_local, 0, shape, sim_time, INIT_CM_TIME) if rank == 0 else None]
            dist.broadcast_object_list(shared, src=0, group=ctl)   # (one node: rank 0's files are visible to every rank)
            all_paths = shared[0]
            costs = [engine.inspect_vxa(p).nvox for p in all_paths]          # (every robot takes the same number of steps here)
            mine = parallel.shard_by_cost(costs, world)[rank]
            seng, s_clocks, s_vs, s_nvox, _, _, sc1, _ = run_population([all_paths[i] for i in mine])
            s_elapsed_max, s_host_max, s_total_vs = reduce_stats(s_clocks, s_vs)
            strong = {"scaling": "strong", "workload": "ONE population of %d random %dx%dx%d robots sharded %d-way by cost (%d on this rank)"
                                 % (n_local, shape[0], shape[1], shape[2], world, len(mine)),
                      "value": s_total_vs / s_elapsed_max, "unit": "voxel-timesteps/s", "ms_per_step": s_elapsed_max / args.steps * 1e3,
                      "ms_per_step_host_clock": s_host_max / args.steps * 1e3, "kernel": kernel_name(sc1.dominant_block)}
            seng.close()
            # ... and once more with the option tile_small: a shard of 64 robots occupies a quarter of a GPU's CUs, where cutting the large
            # robots into tiles was measured 6-14 % faster (DESIGN.md section 4 "Tiled path"); the default keeps the kernel choice a function
            # of the robot alone, so both are printed
            teng, t_clocks, t_vs, _, _, _, tc1, _ = run_population([all_paths[i] for i in mine], {"tile_small": 1})
            t_elapsed_max, t_host_max, t_total_vs = reduce_stats(t_clocks, t_vs)
            strong["tile_small"] = {"value": t_total_vs / t_elapsed_max, "ms_per_step": t_elapsed_max / args.steps * 1e3,
                                    "ms_per_step_host_clock": t_host_max / args.steps * 1e3, "kernel": kernel_name(tc1.dominant_block)}
            teng.close()

        handle = None
        if world > 1:
            # the other N-GPU route: ONE process, one handle over the N devices (vxh_create_multi: the C ABI partitions the population
            # by cost, one host thread per device, results stay in host memory -- no collective).  Rank 0 steps N x 512 robots through
            # it while the other ranks wait at the HOST-side barrier below (their engines are closed, their GPUs idle: a rank
            # waiting in an RCCL barrier would keep a spinning kernel on the device being measured); same timing protocol.
            barrier()                          # every rank has closed its engines
            if rank == 0:
                heng = None
                try:                       # (an extra leg: whatever goes wrong in it, the line above all is still printed)
                    hp = []
                    for k in range(world):
                        hp += make_population(os.path.join(tmp, "handle%d" % k), n_local, k * n_local, shape, sim_time, INIT_CM_TIME)
                    devices = [0] * world if share_gpu else list(range(world))
                    heng = engine.Engine(engine.VOXCAD, devices)
                    heng.add_vxa_files(hp)
                    hd = [heng.dims(i) for i in range(len(hp))]
                    h_nvox = sum(d["nvox"] for d in hd)
                    heng.step((int(max(INIT_CM_TIME / d["dt"] for d in hd)) + 32) + max(args.warmup, 1))
                    h0 = heng.counters()
                    hw0 = time.time()
                    h_dev, h_elapsed = timed_steps(heng, args.steps)
                    trace_mark("multi_handle_region", hw0, time.time(), rank)
                    h1 = heng.counters()
                    assert abs((h1.voxel_steps - h0.voxel_steps) - float(h_nvox) * args.steps) < 0.5
                    # (host clock: this route's host side -- one thread per device launching and waiting -- is part of it)
                    handle = {"scaling": "weak", "route": "one process, vxh_create_multi over devices %s" % devices, "value": h_nvox * args.steps / h_elapsed,
                              "unit": "voxel-timesteps/s", "ms_per_step": h_elapsed / args.steps * 1e3, "ms_per_step_device": h_dev / args.steps * 1e3,
                              "timing": "host clock between the synchronisations", "robots": len(hp)}
                except Exception as exc:
                    handle = {"route": "one process, vxh_create_multi", "error": "%s: %s" % (type(exc).__name__, exc)}
                finally:
                    if heng is not None:
                        heng.close()
            barrier()

        if rank == 0:
            dom_seconds = c1.dominant_seconds
            roof_bw = (c1.dominant_alg_bytes / dom_seconds / 1e9) if dom_seconds > 0 else 0.0
            out = {
                "metric": "voxel_timesteps_per_sec", "value": total_vs / elapsed_max, "unit": "voxel-timesteps/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "clock": "hip_event",           # (what `value` and `ms_per_step` are priced on; `host_clock` = the contract's bracket, beside it)
                "timing": "HIP events on the engine's own stream around the timed steps (first event in front of the call's first launch, last "
                          "behind its last), max over ranks; host_clock = perf_counter between the barrier + torch.cuda.synchronize() pairs",
                "host_clock": {"ms_per_step": host_max / args.steps * 1e3, "value": total_vs / host_max},
                "ranks": ranks,
                "config": {"workload": ("population of %d random %dx%dx%d soft robots per GPU (BASELINE configs[2], "
                                        "pop-512 of 10x10x10), self-collision on, DtFrac 0.9, evosoro default materials"
                                        % (n_local, shape[0], shape[1], shape[2]))
                                       + ((" -- shard 0 of %d (greedy LPT by voxels) of a population of %d" % (args.shard_of, len(all_local_paths))) if args.shard_of > 1 else ""),
                           "robots_per_gpu": n_local, "voxels_per_gpu": nvox, "bonds_per_gpu": nbond,
                           "pre_advanced_steps": pre + max(args.warmup, 1),
                           "large_angle_bonds": large / max(1, total_b),
                           "parallelism": "population sharded %d-way, no data-path collective" % world},
                "roofline": {"bound": "hbm", "achieved": roof_bw, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": roof_bw / HBM_PEAK_GBS, "traffic": None,
                             "kernel": kernel_name(c1.dominant_block),
                             "launches": int(c1.dominant_launches),
                             "avg_launch_ms": dom_seconds / max(1, c1.dominant_launches) * 1e3,
                             "alg_bytes_per_voxel_step": c1.dominant_alg_bytes / max(1.0, c1.dominant_voxel_steps),
                             "note": "achieved = (224*Nvox + 144*Nbond) bytes per voxel-step x steps / HIP-event time of "
                                     "the dominant kernel on its stream; achieved and traffic are both rates (GB/s); traffic << "
                                     "achieved because the resident robots' bond history is served by the L2 (hit rate 0.97): the "
                                     "kernel is FP64-issue-bound, not HBM-bound"},
                "timed_region": {
                    "launches": int(c1.dominant_launches),          # (of the timed call: this counter is per call, not cumulative)
                    "mean_steps_per_launch": args.steps / max(1, int(c1.dominant_launches)),
                    "note": "a launch of the resident kernel ends with its slowest workgroup and carries a fixed cost: ~0.02 ms for a "
                            "population without self-collision, ~0.058 ms for this one (prologue/epilogue of two robots per CU ~0.03 ms; the "
                            "rest is waiting for the CUs whose robots ran a collision broad-phase, ~0.04 ms per run, ~50 of 512 robots in "
                            "any 20-step launch; 0.27 / 0.17 ms until round 3, 0.069 until round 5's dispatch order); a call costs ~0.04 ms "
                            "on the host.  Per step WITHOUT those: ~23.5 us.  --steps 20 times ONE 20-step launch (25.0-26.0 us per step by "
                            "events, 1.37-1.43e10 voxel-steps/s; host clock 27.1-27.9 us); the default --steps 2000 times two launches of up "
                            "to 1024 steps (23.7-24.4 us, 1.46-1.51e10).  DESIGN.md section 5 'The cost of a launch'; HISTORY.md"},
                "kernel_seconds": c1.kernel_seconds - c0.kernel_seconds,
                "fitness_gather_ms": gather_ms,
                "control_plane": control_plane,
            }
            par = measured_parity()
            if par:
                out["parity"] = par
            out["roofline"].update(measured_traffic(n_local, args.lattice))
            if strong:
                out["strong"] = strong
            if handle:
                out["multi_handle"] = handle
            comp = measured_compute(kernel_name(c1.dominant_block)) if (n_local == 512 and args.lattice == 10) else None
            if comp:
                # the bound that binds, next to the contract's algorithmic-HBM fraction
                out["roofline"]["compute"] = comp
                out["roofline"]["bound"] = comp.get("bound", "hbm")
            bind = binding("headline", total_vs / elapsed_max / world) if (n_local == 512 and args.lattice == 10) else None
            if bind:
                out["roofline"]["binding"] = bind          # (per GPU: this rank's population over the slowest rank's time)
                out["roofline"]["bound"] = "fp64-issue"
            if world == 1 and not args.no_other_configs:
                out["other_configs"] = [
                    side_workload(engine, "cfg1", local_rank), side_workload(engine, "cfg3", local_rank), side_workload(engine, "cfg4", local_rank),
                    side_workload(engine, "dense", local_rank), side_workload(engine, "mixed", local_rank),
                ]
                if n_local == 512 and args.lattice == 10 and args.shard_of <= 1:
                    out["other_configs"].append(shard_of_population(engine, all_local_paths, local_rank, 8, 1024))
                for entry in out["other_configs"]:
                    entry.pop("_voxel_steps_process", None)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(shape, local_rank)
                check = out["cpu_baseline"].pop("parity_check", None)
                if check:
                    out.setdefault("parity", {})["live_check"] = check
            os.write(json_fd, (json.dumps(out) + "\n").encode())
    finally:
        if distributed:
            barrier()
        shutil.rmtree(tmp, ignore_errors=True)
        if distributed:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
