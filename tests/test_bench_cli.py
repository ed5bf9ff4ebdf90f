"""bench.py's command line: `--gpus N` must either run N ranks or fail loudly -- never print a one-GPU number under an N-GPU label."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "6", "--warmup", "2", "--robots-per-gpu", "3", "--lattice", "4", "--no-cpu-baseline", "--no-other-configs"]


def _run(cmd, env_extra, timeout=600):
    env = dict(os.environ, PYTHONPATH=REPO, **env_extra)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(key, None)
    return subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def _line(proc):
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (proc.stdout.decode()[-2000:], proc.stderr.decode()[-3000:])
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks_on_the_stub():
    """plain `python bench.py --gpus 2`: bench.py re-launches itself under torch.distributed.run (here: the stub wrapper, gloo,
    every rank on 'device 0') and the line says n_gpus 2, with the strong run and the one-handle route beside the weak one"""
    proc = _run([sys.executable, os.path.join(REPO, "tests", "bench_stub_runner.py"), "--gpus", "2"] + SMALL,
                {"VXH_BENCH_SHARE_GPU": "1"})
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    out = _line(proc)
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["scaling"] == "weak"
    assert out["config"]["robots_per_gpu"] == 3 and out["value"] > 0
    assert out["strong"]["scaling"] == "strong" and out["strong"]["value"] > 0
    assert out["strong"]["tile_small"]["value"] > 0                      # (the strong case twice: default kernels and tile_small)
    assert out["multi_handle"]["robots"] == 6 and out["multi_handle"]["value"] > 0
    # the line proves that two ranks took part: gathered over the default group, one row per rank
    assert out["ranks"]["world"] == 2 and out["ranks"]["backend"] == "gloo"
    assert len(out["ranks"]["device_of_rank"]) == 2 and len(out["ranks"]["ms_per_step_of_rank"]) == 2
    assert all(ms > 0 for ms in out["ranks"]["ms_per_step_of_rank"])
    # both clocks: `value` from the engine's event time, the host clock beside it (never faster than the device time it contains)
    assert out["host_clock"]["ms_per_step"] >= out["ms_per_step"] * 0.999 and out["host_clock"]["value"] > 0


def test_shard_of_times_one_cost_balanced_shard_of_the_population_on_the_stub():
    """`--shard-of K` (round 6): BASELINE configs[2] is stated as a population sharded over 8 GPUs; one GPU measures what one of them
    steps -- shard 0 of K by greedy LPT on voxels (evosoro_amd/parallel.py shard_by_cost) -- and the line says so."""
    small = ["--steps", "6", "--warmup", "2", "--robots-per-gpu", "8", "--lattice", "4", "--no-cpu-baseline", "--no-other-configs"]
    proc = _run([sys.executable, os.path.join(REPO, "tests", "bench_stub_runner.py"), "--gpus", "1", "--shard-of", "4"] + small, {})
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    out = _line(proc)
    assert out["n_gpus"] == 1 and out["config"]["robots_per_gpu"] == 2 and "shard 0 of 4" in out["config"]["workload"]
    assert out["value"] > 0 and out["clock"] == "hip_event"
    whole = _line(_run([sys.executable, os.path.join(REPO, "tests", "bench_stub_runner.py"), "--gpus", "1"] + small, {}))
    assert whole["config"]["robots_per_gpu"] == 8 and whole["config"]["voxels_per_gpu"] > 3 * out["config"]["voxels_per_gpu"]


def test_no_default_group_collective_overlaps_a_timed_region(tmp_path):
    """The N > 1 run, traced (VXH_BENCH_TRACE): on a GPU node the default group is RCCL, whose collectives are kernels that spin on
    the device until every rank has arrived -- so no collective of the DEFAULT group may be in flight, on any rank, while any rank
    (or rank 0's handle over all the devices) is inside a timed region; whatever coordinates the ranks around those regions must go
    over the host-side control group.  And the default group must carry the fitness gather and nothing else."""
    prefix = str(tmp_path / "trace")
    proc = _run([sys.executable, os.path.join(REPO, "tests", "bench_stub_runner.py"), "--gpus", "2"] + SMALL,
                {"VXH_BENCH_SHARE_GPU": "1", "VXH_BENCH_TRACE": prefix})
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    out = _line(proc)
    assert "gloo host group" in out["control_plane"]
    events = []
    for rank in (0, 1):
        with open("%s.rank%d.jsonl" % (prefix, rank)) as f:
            events += [json.loads(ln) for ln in f if ln.strip()]
    regions = [e for e in events if e["op"] in ("timed_region", "multi_handle_region")]
    assert len([e for e in regions if e["op"] == "timed_region"]) >= 5      # weak + strong on both ranks, the handle's on rank 0
    assert len([e for e in regions if e["op"] == "multi_handle_region"]) == 1
    default_ops = [e for e in events if e["group"] == "default"]
    assert default_ops and set(e["op"] for e in default_ops) == {"all_gather_into_tensor"}, sorted(set(e["op"] for e in default_ops))
    for r in regions:
        for e in default_ops:
            assert e["t1"] <= r["t0"] or e["t0"] >= r["t1"], ("a default-group collective overlaps a timed region", e, r)
    # ... and the ranks that wait while rank 0 times the handle wait in a control-group barrier that spans it
    handle = [e for e in regions if e["op"] == "multi_handle_region"][0]
    waits = [e for e in events if e["rank"] == 1 and e["op"] == "barrier" and e["group"] == "ctl" and e["t0"] <= handle["t0"] and e["t1"] >= handle["t1"]]
    assert waits, "rank 1 did not sit in a host-side barrier during the multi_handle measurement"


def test_gpus_more_than_visible_fails_loudly():
    proc = _run([sys.executable, os.path.join(REPO, "tests", "bench_stub_runner.py"), "--gpus", "4"] + SMALL, {"VXH_STUB_GPUS": "1"})
    assert proc.returncode != 0
    assert "--gpus 4 but 1 GPU(s) visible" in proc.stderr.decode()
    assert not [ln for ln in proc.stdout.decode().splitlines() if ln.startswith("{")]


def test_world_size_mismatch_fails_loudly():
    proc = _run([sys.executable, os.path.join(REPO, "tests", "bench_stub_runner.py"), "--gpus", "1"] + SMALL, {})
    assert proc.returncode == 0 and _line(proc)["n_gpus"] == 1
    env = dict(os.environ, PYTHONPATH=REPO, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, os.path.join(REPO, "tests", "bench_stub_runner.py"), "--gpus", "1"] + SMALL, env=env, cwd=REPO,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert proc.returncode != 0 and "WORLD_SIZE 2" in proc.stderr.decode()


@pytest.mark.gpu
def test_gpus_2_on_one_device_with_the_real_engine():
    """the same on the GPU box: two ranks sharing device 0 (gloo), the real library"""
    proc = _run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--robots-per-gpu", "16",
                 "--lattice", "6", "--no-cpu-baseline", "--no-other-configs"], {"VXH_BENCH_SHARE_GPU": "1"})
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    out = _line(proc)
    assert out["n_gpus"] == 2 and out["config"]["robots_per_gpu"] == 16 and out["value"] > 0
    assert out["strong"]["value"] > 0 and out["multi_handle"]["robots"] == 32


@pytest.mark.gpu
def test_one_rank_over_rccl_with_the_host_side_control_group():
    """The closest a 1-GPU box gets to the N > 1 launch the round-end driver makes: ONE rank, but through the distributed branch with the
    real backends -- the default group is RCCL (`init_process_group("nccl", device_id=...)`), the control plane a gloo group beside it,
    the fitness gather an RCCL collective (VXH_FORCE_DIST=1, rendezvous on 127.0.0.1 as the driver's launcher sets it up)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=REPO, VXH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--robots-per-gpu", "16",
                           "--lattice", "6", "--no-cpu-baseline", "--no-other-configs"], env=env, cwd=REPO, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=900)
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    out = _line(proc)
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert "gloo host group" in out["control_plane"] and "RCCL" in out["control_plane"], out["control_plane"]
    assert out["fitness_gather_ms"] is not None and out["fitness_gather_ms"] >= 0


@pytest.mark.gpu
def test_the_bench_lines_live_parity_check(tmp_path):
    """bench.py's cpu_baseline leg compares the reference binary's result files for its sample with the engine's (parity.live_check):
    here on six robots of the bench population, 0.06 s simulated -- every file byte-identical."""
    ref = os.path.join(REPO, "oracle", "_ref", "voxelyze_ref")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref is not built")
    sys.path.insert(0, REPO)
    try:
        import bench
    finally:
        sys.path.remove(REPO)
    paths = bench.make_population(str(tmp_path), 6, 40, (6, 6, 6), 0.06, 0.02)
    bench.run_reference(ref, paths, str(tmp_path), 6)
    check = bench.live_parity(paths, str(tmp_path), 0)
    assert check["live"] and check["robots"] == 6 and check["numeric_tags_per_file"] >= 10
    assert check["result_files_byte_identical"] == 6 and check["worst_tag_difference"] == 0.0, check
