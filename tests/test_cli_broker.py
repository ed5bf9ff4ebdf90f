"""The `voxelyze` command line under the reference's launch pattern: evosoro starts ONE process per robot, all of a generation at once
(evosoro/tools/evaluation.py:59-90), and reads result files as they appear (:101-211).

CPU part (no GPU here): the plumbing of the per-user broker (evosoro_amd/csrc/voxelyze_main.cpp) -- concurrent one-file invocations
start exactly one broker, are coalesced into one batch, each gets ITS verdict, and the broker leaves when idle; without a GPU that
verdict is the loud failure of the product path (no CPU fallback), through the broker and with --direct alike.

GPU part (-m gpu): 16 concurrent `voxelyze -f` processes, one golden robot each, one of them the 8000-voxel lattice of BASELINE
configs[4] (the tiled kernel), every result XML checked against the reference binary's -- once with every process stepping its own robot
(VXH_BROKER=0: 16 HIP contexts on one GPU, the tiles of the large lattice competing with 15 other processes for the CUs) and once
through the broker; wall clocks written to gpurun_out/r04_concurrent_cli.json."""
import json
import os
import subprocess
import sys
import time

import pytest

from evosoro_amd import engine as eng_mod
from oracle import vxoracle as vo

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(tmp_path, **extra):
    env = dict(os.environ, VXH_BROKER_SOCKET=str(tmp_path / "broker.sock"), VXH_BROKER_IDLE_S="3", VXH_BROKER_LOG=str(tmp_path / "broker.log"))
    env.pop("VXH_BROKER", None)
    env.update(extra)
    return env


def _stat(env):
    out = subprocess.run([eng_mod.CLI_PATH, "--broker-stat"], env=env, stdout=subprocess.PIPE, timeout=30).stdout.decode()
    return out.strip()


def test_broker_plumbing_without_a_gpu(golden_dir, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU test below covers the broker")
    eng_mod.build()
    env = _env(tmp_path)
    vxa = os.path.join(golden_dir, "vxa", "soft5_init0.vxa")
    procs = [subprocess.Popen([eng_mod.CLI_PATH, "-f", vxa], env=env, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE) for _ in range(6)]
    outs = [p.communicate(timeout=60) for p in procs]
    assert [p.returncode for p in procs] == [0] * 6                  # the reference's "did not complete" code: there is no device
    for _, err in outs:
        assert b"no usable HIP device" in err and b"broker" in err, err
    stat = _stat(env)
    assert stat.startswith("batches ") and " largest " in stat, stat
    assert int(stat.split()[1]) >= 1 and int(stat.split()[-1]) >= 2, stat        # requests were coalesced (one broker, one queue)
    # a file that does not exist is refused by the client itself, like a failed LoadVXAFile
    p = subprocess.run([eng_mod.CLI_PATH, "-f", str(tmp_path / "nothing.vxa")], env=env, cwd=tmp_path, stderr=subprocess.PIPE, timeout=30)
    assert p.returncode == 0 and b"cannot read the file" in p.stderr
    # ... and --direct / VXH_BROKER=0 never talk to a broker
    for how in (["--direct"], []):
        e2 = dict(env, VXH_BROKER="0") if not how else env
        p = subprocess.run([eng_mod.CLI_PATH, "-f", vxa] + how, env=e2, cwd=tmp_path, stderr=subprocess.PIPE, timeout=30)
        assert p.returncode == 0 and b"no usable HIP device" in p.stderr and b"broker" not in p.stderr
    # the broker leaves when nobody has asked for VXH_BROKER_IDLE_S seconds, and takes its socket with it
    deadline = time.time() + 20
    while os.path.exists(env["VXH_BROKER_SOCKET"]) and time.time() < deadline:
        time.sleep(0.2)
    assert not os.path.exists(env["VXH_BROKER_SOCKET"])
    assert _stat(env).startswith("no broker")


CASES = ["probe6", "rand6_nocol", "rand6_col", "soft5_init0", "stiff5", "grow5", "devo4", "bench10_0", "bench10_1", "cfg1_00", "cfg1_21",
         "cfg1_42", "cfg1_63", "stop1_5", "stop3_5", "cfg4_full20"]          # 16 _voxcad robots, the last one an 8000-voxel lattice


def _generation(golden_dir, work, env, cases=CASES):
    """start one process per robot the way evaluation.py:89-90 does (all at once, no waiting in between), wait for all of them"""
    os.makedirs(os.path.join(work, "golden_run", "fitnessFiles"))
    t0 = time.time()
    procs = [subprocess.Popen([eng_mod.CLI_PATH, "-f", os.path.join(golden_dir, "vxa", n + ".vxa")], env=env, cwd=work,
                              stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for n in cases]
    errs = [p.communicate(timeout=900)[1] for p in procs]
    wall = time.time() - t0
    return wall, [p.returncode for p in procs], errs


def _check_results(golden_dir, work, cases=CASES):
    files = sorted(os.listdir(os.path.join(work, "golden_run", "fitnessFiles")))
    assert len(files) == len(cases), files
    for n in cases:
        want_path = os.path.join(golden_dir, "expected", n + ".xml")
        want = vo.read_result_xml(want_path)
        ident = [ln for ln in open(os.path.join(golden_dir, "vxa", n + ".vxa")) if "<FitnessFileName>" in ln][0].split("--id_")[1].split(".xml")[0]
        got_path = os.path.join(work, "golden_run", "fitnessFiles", "softbotsOutput--id_%s.xml" % ident)
        got = vo.read_result_xml(got_path)
        assert list(got) == list(want), n
        for tag in want:
            assert abs(got[tag] - want[tag]) <= 1e-5 * max(1.0, abs(want[tag])), (n, tag, got[tag], want[tag])
        fit = lambda p: [ln for ln in open(p).read().splitlines() if "NormFinalDist" in ln]
        assert fit(got_path) == fit(want_path), n                  # the fitness line byte for byte


@pytest.mark.gpu
def test_sixteen_concurrent_voxelyze_processes(golden_dir, tmp_path, kernel_path):
    if kernel_path != "auto":
        pytest.skip("the command line with the engine's own kernel choice (the lattice above 1024 voxels needs its own tile count)")
    record = {"cases": CASES, "note": "16 concurrent `voxelyze -f x.vxa` processes started the way evosoro/tools/evaluation.py:89-90 starts them; "
                                      "wall clock from the first Popen to the last exit; every result XML checked against the reference binary's"}
    # (a) every process steps its own robot: 16 HIP contexts on one GPU
    wall, codes, errs = _generation(golden_dir, str(tmp_path / "direct"), _env(tmp_path, VXH_BROKER="0"))
    assert codes == [1] * len(CASES), [e.decode()[-300:] for e in errs if e]
    _check_results(golden_dir, str(tmp_path / "direct"))
    record["direct"] = {"wall_s": wall, "tile_timeouts_recovered": sum(1 for e in errs if b"stepped again without the tiled kernel" in e)}
    # (b) through the broker: the first client starts it; a second generation right behind finds it running (HIP runtime warm)
    env = _env(tmp_path)
    wall1, codes, errs = _generation(golden_dir, str(tmp_path / "broker1"), env)
    assert codes == [1] * len(CASES), [e.decode()[-300:] for e in errs if e]
    _check_results(golden_dir, str(tmp_path / "broker1"))
    stat1 = _stat(env)
    wall2, codes, errs = _generation(golden_dir, str(tmp_path / "broker2"), env)
    assert codes == [1] * len(CASES), [e.decode()[-300:] for e in errs if e]
    _check_results(golden_dir, str(tmp_path / "broker2"))
    stat2 = _stat(env)
    subprocess.run([eng_mod.CLI_PATH, "--broker-quit"], env=env, stdout=subprocess.DEVNULL, timeout=30)
    record["broker"] = {"wall_s_first_generation_incl_broker_start": wall1, "wall_s_second_generation": wall2, "stat_after_first": stat1, "stat_after_second": stat2}
    assert int(stat1.split()[-1]) >= 8, stat1          # the requests of a generation were coalesced (largest batch)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "r04_concurrent_cli.json"), "w") as f:
        json.dump(record, f, indent=1)
    print("16 concurrent voxelyze processes: direct %.2f s; broker %.2f s (first generation, incl. its start), %.2f s (second)" % (wall, wall1, wall2))
    assert wall2 < wall, (wall, wall2)
