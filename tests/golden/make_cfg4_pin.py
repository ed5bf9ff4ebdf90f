#!/usr/bin/env python3
"""Pins the C restatement (oracle/vx_oracle.c) on BASELINE configs[4] at its full size -- one full 20x20x20 lattice, self-collision
on, the whole 781-step evaluation -- against the REFERENCE: runs oracle/_ref/vxprobe (the unmodified reference sources + our
state-dumping main, built by `make -C oracle ref`) on the .vxa that tests/test_gpu_tiled.py steps on the GPU, and commits, instead of
the 0.9 MB final state, its SHA-256 together with the few numbers a failing test wants to show (tests/golden/expected/cfg4_full20.json).
tests/test_oracle_vs_reference.py::test_oracle_on_the_full_20_cube_equals_the_reference_binary recomputes the hash from the oracle's
final state: bit-exact or it fails.  Needs /root/reference (build container only); the fixture is data.
    python tests/golden/make_cfg4_pin.py"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def cfg4_vxa(tmp, run_directory=None, ident=0):
    """the configs[4] robot exactly as the GPU tests and bench.py build it (workloads.full_material(20, 1), 0.05 s, InitCmTime 0.01);
    run_directory: what the file names inside the .vxa start with (default: tmp itself)"""
    from evosoro_amd import workloads
    from evosoro_amd.base import Sim, Env
    from evosoro_amd.tools.read_write_voxelyze import write_voxelyze_file
    sim = Sim(dt_frac=0.9, simulation_time=0.05, fitness_eval_init_time=0.01)
    ind = workloads.make_individual(ident, workloads.full_material(20, 1))
    if run_directory is None:
        os.makedirs(os.path.join(tmp, "voxelyzeFiles"), exist_ok=True)
        write_voxelyze_file(sim, Env(), ind, tmp, "c4")
        return os.path.join(tmp, "voxelyzeFiles", "c4--id_%05i.vxa" % ident)
    here = os.getcwd()
    os.chdir(tmp)
    try:
        for d in ("voxelyzeFiles", "fitnessFiles"):
            os.makedirs(os.path.join(run_directory, d), exist_ok=True)
        write_voxelyze_file(sim, Env(), ind, run_directory, "c4")
    finally:
        os.chdir(here)
    return os.path.join(tmp, run_directory, "voxelyzeFiles", "c4--id_%05i.vxa" % ident)


def vxa_digest(path):
    """SHA-256 of the .vxa text from <Environment> on (environment, materials, structure: everything but the <Simulator> block, whose
    file names carry the run directory)"""
    text = open(path).read()
    return hashlib.sha256(text[text.index("<Environment>"):].encode()).hexdigest()


def state_digest(state14):
    """SHA-256 over the little-endian doubles of the [nvox, 14] final state (pos3, quat wxyz, scale, vel3, angvel3)"""
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(state14, dtype="<f8").tobytes()).hexdigest()


def main():
    from oracle import vxoracle as vo
    probe = os.path.join(REPO, "oracle", "_ref", "vxprobe")
    assert os.path.exists(probe), "build it first: make -C oracle ref"
    with tempfile.TemporaryDirectory() as tmp:
        vxa = cfg4_vxa(tmp)
        out = os.path.join(tmp, "final.bin")
        subprocess.run(["timeout", "1800", probe, "-f", vxa, "-o", out, "-every", "100000000", "-noresult"], check=True, cwd=tmp,
                       stdout=subprocess.DEVNULL)
        tr = vo.read_trace(out)
        last = tr["records"][-1]
        pin = {"robot": "workloads.full_material(20, 1), Sim(dt_frac=0.9, simulation_time=0.05, fitness_eval_init_time=0.01), Env() defaults",
               "vxa_sha256": vxa_digest(vxa),
               "nvox": int(tr["nvox"]), "nbond": int(tr["nbond"]), "total_steps": int(tr["total_steps"]), "ncol_at_end": int(last["ncol"]),
               "final_state_sha256": state_digest(last["state"]),
               "ini_cm_hex": [float(x).hex() for x in tr["ini_cm"]], "cur_cm_hex": [float(x).hex() for x in tr["cur_cm"]],
               "source": "oracle/_ref/vxprobe (reference sources under /root/reference compiled by oracle/Makefile), this script"}
    with open(os.path.join(HERE, "expected", "cfg4_full20.json"), "w") as f:
        json.dump(pin, f, indent=1)
    print(json.dumps(pin, indent=1))
    # the same robot as a golden CASE for the command-line tests (tests/test_gpu_cli.py: a lattice above 1024 voxels among the
    # concurrent `voxelyze -f` processes): the .vxa with the goldens' relative run directory and the result XML the reference
    # binary writes for it
    import shutil
    ref = os.path.join(REPO, "oracle", "_ref", "voxelyze_ref")
    with tempfile.TemporaryDirectory() as tmp:
        vxa = cfg4_vxa(tmp, run_directory="golden_run", ident=900)
        subprocess.run(["timeout", "1800", ref, "-f", vxa], check=False, cwd=tmp, stdout=subprocess.DEVNULL)
        shutil.copy(vxa, os.path.join(HERE, "vxa", "cfg4_full20.vxa"))
        shutil.copy(os.path.join(tmp, "golden_run", "fitnessFiles", "softbotsOutput--id_00900.xml"), os.path.join(HERE, "expected", "cfg4_full20.xml"))


if __name__ == "__main__":
    main()
