"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and the reference's golden outputs.

Tolerances (FP64 on both sides; differences come only from device libm ulps and FMA contraction):
  * first 200 steps: positions within 1e-9 voxel, quaternions within 1e-9
  * whole runs (0.2-0.5 s of simulated time, thousands of steps, friction/contact discontinuities amplify
    ulp noise): CoM displacement within 2e-3 voxel of the reference, result tags to that accuracy
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = ["probe6", "rand6_nocol", "rand6_col", "soft5_init0", "phase4"]


@pytest.fixture(scope="module")
def eng_mod():
    from evosoro_amd import engine
    return engine


def test_early_steps_match_oracle(eng_mod, golden_dir):
    from oracle import vxoracle as vo
    with eng_mod.Engine(eng_mod.VOXCAD, 0) as eng:
        for name in CASES:
            eng.add_vxa_file(os.path.join(golden_dir, "vxa", name + ".vxa"))
        sims = [vo.OracleSim.from_vxa(os.path.join(golden_dir, "vxa", name + ".vxa")) for name in CASES]
        done = 0
        for upto in (1, 2, 10, 50, 200):
            eng.step(upto - done)
            done = upto
            for i, (name, sim) in enumerate(zip(CASES, sims)):
                sim.step(upto - sim.info().steps)
                want, got = sim.state(), eng.state(i)
                lat = sim.model["lattice_dim"]
                assert np.abs(got[:, 0:3] - want[:, 0:3]).max() / lat < 1e-9, (name, upto)
                assert np.abs(got[:, 3:7] - want[:, 3:7]).max() < 1e-9, (name, upto)
                assert np.abs(got[:, 7] - want[:, 7]).max() / lat < 1e-12, (name, upto)


def test_full_runs_match_reference(eng_mod, golden_dir):
    from oracle import vxoracle as vo
    with eng_mod.Engine(eng_mod.VOXCAD, 0) as eng:
        for name in CASES:
            eng.add_vxa_file(os.path.join(golden_dir, "vxa", name + ".vxa"))
        eng.run()
        for i, name in enumerate(CASES):
            trace = vo.read_trace(os.path.join(golden_dir, "expected", name + ".final.bin"))
            want = vo.read_result_xml(os.path.join(golden_dir, "expected", name + ".xml"))
            res = eng.result(i)
            lat = vo.parse_vxa(os.path.join(golden_dir, "vxa", name + ".vxa"))["lattice_dim"]
            assert res.status == eng_mod.ROBOT_FINISHED
            assert res.steps == trace["total_steps"], name
            assert np.abs(np.array(res.cur_cm) - trace["cur_cm"]).max() / lat < 2e-3, name
            assert np.abs(np.array(res.ini_cm) - trace["ini_cm"]).max() / lat < 2e-3, name
            assert abs(res.norm_final_dist - want["NormFinalDist"]) < 2e-3, name
            assert abs(res.final_dist_y - want["finalDistY"]) < 2e-3, name
            assert abs(res.lifetime - want["Lifetime"]) < 1e-5, name
