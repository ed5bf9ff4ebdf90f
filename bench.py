#!/usr/bin/env python3
"""Benchmark of the hot path: voxel-timesteps/sec of a batch of random 10x10x10 soft robots (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[2], "population 512 of 10x10x10 robots": it fits one MI355X, so a single GPU
steps the whole population of 512; with N GPUs every rank steps its own 512 robots (weak scaling, seeds
offset by rank, no data-path collective).  A "step" is one Voxelyze TimeStep of every robot of the batch.
Robots: material per voxel uniform on {0..4} with P(empty)=0.3, largest connected component, evosoro's default
materials/environment, self-collision ON, DtFrac 0.9.  After the timed region the
fitness records of all ranks are gathered with one RCCL all_gather (the path's only collective).

The printed JSON line carries `roofline` (HBM: algorithmic bytes (224*Nvox + 144*Nbond) per voxel-step of
SURVEY.md 8(d) over the HIP-event time of the dominant kernel) and `cpu_baseline` (the reference C++
voxelyze, oracle/_ref/voxelyze_ref built from the reference sources, timed on this box's host cores on a
bounded sample of the same robots; rank 0, N=1 only).
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


def make_population(tmp, count, first_seed, shape, sim_time, init_time, selfcol=True):
    from evosoro_amd import workloads
    from evosoro_amd.base import Sim, Env
    from evosoro_amd.tools.read_write_voxelyze import write_voxelyze_file
    for d in ("voxelyzeFiles", "fitnessFiles", "tempFiles"):
        os.makedirs(os.path.join(tmp, d), exist_ok=True)
    sim = Sim(self_collisions_enabled=selfcol, dt_frac=0.9, simulation_time=sim_time, fitness_eval_init_time=init_time)
    env = Env()
    paths = []
    for i in range(count):
        ind = workloads.random_robot(first_seed + i, shape, first_seed + i)
        write_voxelyze_file(sim, env, ind, tmp, "bench")
        paths.append(os.path.join(tmp, "voxelyzeFiles", "bench--id_%05i.vxa" % ind.id))
    return paths


def cpu_baseline(shape, sim_time=0.25):
    """Reference voxelyze (oracle/_ref) on the host cores: 4 robots per core (at most 128) of the bench workload,
    0.25 s of simulated time each (about 3900 steps), launched concurrently like evaluation.py:89 does; sized for
    roughly 10-30 s of CPU work."""
    from evosoro_amd import engine
    ref = os.path.join(REPO, "oracle", "_ref", "voxelyze_ref")
    cores = os.cpu_count() or 1
    budget_robots = min(128, 4 * cores)
    tmp = tempfile.mkdtemp(prefix="vxbench_cpu_")
    try:
        paths = make_population(tmp, budget_robots, 0, shape, sim_time, 0.05)
        work = 0.0
        with engine.Engine(engine.VOXCAD, 0) as eng:      # only to get voxel counts and planned step counts
            for p in paths:
                eng.add_vxa_file(p)
            for i in range(len(paths)):
                d = eng.dims(i)
                work += d["nvox"] * d["planned_steps"]
        if os.path.exists(ref):
            kind, used = "reference", min(cores, len(paths))
            t0 = time.time()
            running, queue = [], list(paths)
            while queue or running:
                while queue and len(running) < used:
                    running.append(subprocess.Popen(["timeout", "600", ref, "-f", queue.pop(0)], cwd=tmp,
                                                    stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
                running = [p for p in running if p.poll() is None]
                time.sleep(0.005)
            wall = time.time() - t0
            done = len([f for f in os.listdir(os.path.join(tmp, "fitnessFiles")) if f.endswith(".xml")])
            if done != len(paths):
                raise RuntimeError("reference finished %d of %d robots" % (done, len(paths)))
        else:
            from oracle import vxoracle
            kind, used = "port", 1
            t0 = time.time()
            for p in paths:
                sim = vxoracle.OracleSim.from_vxa(p)
                sim.step(-1)
            wall = time.time() - t0
        return {"value": work / wall, "unit": "voxel-timesteps/s", "cores": used, "kind": kind,
                "sample": "%d random %dx%dx%d robots of the bench population, %.2f s simulated each (%.3g voxel-steps), "
                          "%d concurrent processes, %.1f s wall" % (len(paths), shape[0], shape[1], shape[2], sim_time, work, used, wall)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measured_traffic(robots_per_gpu, lattice):
    """HBM bytes the dominant kernel really moved, from the newest rocprofv3 counter summary under profiles/
    (scripts/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate passes over this same command; FETCH_SIZE is
    doubled, as the 8-byte-per-lane calibration kernels of the same passes show it counts half the bytes).  PMC
    counters cannot be read from inside this process, so the figure is only reported for the profiled workload."""
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
    return {"traffic": nbytes / t["avg_launch_ns"],      # bytes / ns = GB/s, per launch like `achieved`
            "traffic_bytes_per_launch": nbytes,
            "traffic_source": os.path.relpath(files[-1], REPO) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              "command; FETCH x%.2f, WRITE x%.2f from the calibration kernels)" % (fetch_scale, write_scale)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--robots-per-gpu", type=int, default=512)
    ap.add_argument("--lattice", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from evosoro_amd import engine, parallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or os.environ.get("VXH_FORCE_DIST") == "1"   # (the latter: 1-rank RCCL smoke test)
    if args.gpus != world and distributed:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    shape = (args.lattice,) * 3
    n_local = args.robots_per_gpu
    tmp = tempfile.mkdtemp(prefix="vxbench_r%d_" % rank)
    try:
        # long enough that no robot reaches its stop condition inside warmup + timed steps
        sim_time = max(0.5, (args.steps + args.warmup + 64) * 7.2e-4)
        paths = make_population(tmp, n_local, rank * n_local, shape, sim_time, 0.05)
        eng = engine.Engine(engine.VOXCAD, local_rank)
        for p in paths:
            eng.add_vxa_file(p)
        nvox = sum(eng.dims(i)["nvox"] for i in range(n_local))
        nbond = sum(eng.dims(i)["nbond"] for i in range(n_local))
        eng.step(max(args.warmup, 1))                       # upload + warmup (untimed)
        c0 = eng.counters()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step(args.steps)                                # EXACTLY K time steps of every robot
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        c1 = eng.counters()
        local_vs = c1.voxel_steps - c0.voxel_steps
        assert abs(local_vs - float(nvox) * args.steps) < 0.5, "a robot stopped inside the timed region"
        stats = torch.tensor([elapsed, local_vs, c1.kernel_seconds - c0.kernel_seconds], dtype=torch.float64, device="cuda")
        if distributed:
            tmax = stats.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = stats.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            elapsed_max, total_vs = float(tmax[0]), float(tsum[1])
        else:
            elapsed_max, total_vs = elapsed, local_vs
        # the path's collective: fitness records of every rank to every rank (untimed region, reported separately)
        tg = time.perf_counter()
        records = np.stack([parallel.result_to_record(eng.result(i)) for i in range(n_local)])
        table = parallel.gather_records(records, list(range(rank * n_local, (rank + 1) * n_local)), world * n_local,
                                        torch.device("cuda", local_rank) if distributed else None)
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert table.shape[0] == world * n_local and (table[:, 2] > 0).all()

        if rank == 0:
            dom_seconds = c1.dominant_seconds
            roof_bw = (c1.dominant_alg_bytes / dom_seconds / 1e9) if dom_seconds > 0 else 0.0
            out = {
                "metric": "voxel_timesteps_per_sec", "value": total_vs / elapsed_max, "unit": "voxel-timesteps/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "population of %d random %dx%dx%d soft robots per GPU (BASELINE configs[2], "
                                       "pop-512 of 10x10x10), self-collision on, DtFrac 0.9, evosoro default materials"
                                       % (n_local, shape[0], shape[1], shape[2]),
                           "robots_per_gpu": n_local, "voxels_per_gpu": nvox, "bonds_per_gpu": nbond,
                           "parallelism": "population sharded %d-way, no data-path collective" % world},
                "roofline": {"bound": "hbm", "achieved": roof_bw, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": roof_bw / HBM_PEAK_GBS, "traffic": None,
                             "kernel": ("k_robot_steps<%d,...>" % c1.dominant_block) if c1.dominant_block else "k_bonds+k_voxels",
                             "launches": int(c1.dominant_launches),
                             "avg_launch_ms": dom_seconds / max(1, c1.dominant_launches) * 1e3,
                             "alg_bytes_per_launch": c1.dominant_alg_bytes / max(1, c1.dominant_launches),
                             "note": "achieved = (224*Nvox + 144*Nbond) bytes per voxel-step x steps / HIP-event time of "
                                     "the dominant size class on its stream; other size classes run concurrently"},
                "kernel_seconds": c1.kernel_seconds - c0.kernel_seconds,
                "fitness_gather_ms": gather_ms,
            }
            out["roofline"].update(measured_traffic(n_local, args.lattice))
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(shape)
            print(json.dumps(out))
        eng.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        if distributed:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
