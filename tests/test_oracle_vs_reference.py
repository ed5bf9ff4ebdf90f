"""Pins the CPU restatement (oracle/vx_oracle.c) on the reference itself.

Expected values are outputs of the UNMODIFIED reference C++ (built from /root/reference by oracle/Makefile)
captured by tests/golden/make_golden.py: full-state traces of the first 200 steps, the final state of the
whole run, and the result XML.  The restatement follows the reference's operation order, and both sides use
the same glibc libm without FMA contraction, so the bar is BIT-EXACT state (not a tolerance) and
identical 6-significant-digit result fields.
"""
import os

import numpy as np
import pytest

from oracle import vxoracle as vo

LAND_CASES = ["probe6", "rand6_nocol", "rand6_col", "soft5_init0", "phase4", "stiff5", "grow5", "devo4",
              "stop1_5", "stop3_5"]           # (the last two: StopConditionType 1 = time steps, 3 = actuation periods)
SHIPPED = ["example_1", "example_phaseoffset"]
# _voxcad_land_water: generated (land + fluid swimmer with facet drag) and two sample files shipped with the reference
LW_CASES = ["lw_land6", "lw_swim6", "lw_stiff5", "lw_stop3_5"]
LW_SHIPPED = ["lw_hexapus", "lw_quadruped_land"]
LW_TAGS = [("normAbsoluteDisplacement", "norm_abs_disp"), ("normDistX", "norm_dist_x"), ("normDistY", "norm_dist_y"),
           ("normDistZ", "norm_dist_z"), ("VoxelNumber", "nvox")]

RESULT_TAGS = [("NormFinalDist", "norm_final_dist"), ("NormRegimeDist", "norm_regime_dist"),
               ("NormFrozenDist", "norm_frozen_dist"), ("FinalDist", "final_dist"), ("finalDistY", "final_dist_y"),
               ("AnteriorDist", "anterior_dist"), ("PosteriorDist", "posterior_dist"), ("AnteriorY", "anterior_y"),
               ("PosteriorY", "posterior_y"), ("EndOfLifePosteriorY", "end_of_life_posterior_y"),
               ("FallAdjPostY", "fall_adj_post_y"), ("NumNonFeetTouchingFloor", "num_non_feet_touching_floor"),
               ("NumTouchingFloor", "num_touching_floor"), ("Lifetime", "lifetime")]


def _sim(golden_dir, name):
    return vo.OracleSim.from_vxa(os.path.join(golden_dir, "vxa", name + ".vxa"), variant=1 if name.startswith(("lw_", "cfg3_")) else 0)


@pytest.mark.parametrize("name", LAND_CASES + SHIPPED + LW_CASES + LW_SHIPPED)
def test_early_trace_bit_exact(golden_dir, name):
    trace = vo.read_trace(os.path.join(golden_dir, "expected", name + ".early.bin"))
    sim = _sim(golden_dir, name)
    info = sim.info()
    assert (info.nvox, info.nbond) == (trace["nvox"], trace["nbond"])
    assert info.opt_dt == trace["opt_dt"]          # CalcMaxDt, bitwise
    for rec in trace["records"]:
        todo = rec["step"] - sim.info().steps
        if todo > 0:
            assert sim.step(todo) == todo
        got = sim.info()
        assert got.ncol == rec["ncol"], "collision bond count at step %d" % rec["step"]
        assert got.cur_time == rec["time"]
        assert np.array_equal(sim.state(), rec["state"]), "state differs at step %d" % rec["step"]


# BASELINE configs[2] size: two robots of the bench population, whole 0.5 s evaluation (7806 steps, ~700 voxels, collisions)
# ... and BASELINE configs[1]: four robots of the batch of 64 random 6x6x6 walkers, whole 0.5 s evaluation
BIG_CASES = ["bench10_0", "bench10_1", "cfg1_00", "cfg1_21", "cfg1_42", "cfg1_63"]


@pytest.mark.parametrize("name", LAND_CASES + BIG_CASES)
def test_full_run_final_state_and_result(golden_dir, name):
    trace = vo.read_trace(os.path.join(golden_dir, "expected", name + ".final.bin"))
    sim = _sim(golden_dir, name)
    sim.step(-1)
    info = sim.info()
    assert info.status == 1
    assert info.steps == trace["total_steps"]
    assert np.array_equal(sim.state(), trace["records"][-1]["state"])
    assert np.array_equal(np.array(info.ini_cm), trace["ini_cm"])    # IniCM latch (one step late, App. A.9)
    assert np.array_equal(np.array(info.cur_cm), trace["cur_cm"])
    expected = vo.read_result_xml(os.path.join(golden_dir, "expected", name + ".xml"))
    result = sim.result()
    for tag, field in RESULT_TAGS:
        assert "%.6g" % getattr(result, field) == "%.6g" % expected[tag], tag


# BASELINE configs[2] size in _voxcad_land_water: a ~700-voxel swimmer and a full 10x10x10 lattice on land (final state + XML)
# ... and BASELINE configs[3]: four swimmers of the batch of 64 random 8x8x8 ones, whole 0.5 s evaluation
LW_BIG_CASES = ["lw_swim10", "lw_land10", "cfg3_00", "cfg3_21", "cfg3_42", "cfg3_63"]


@pytest.mark.parametrize("name", LW_CASES + LW_BIG_CASES)
def test_land_water_full_run(golden_dir, name):
    trace = vo.read_trace(os.path.join(golden_dir, "expected", name + ".final.bin"))
    sim = _sim(golden_dir, name)
    sim.step(-1)
    info = sim.info()
    assert info.status == 1 and info.steps == trace["total_steps"]
    assert np.array_equal(sim.state(), trace["records"][-1]["state"])
    assert np.array_equal(np.array(info.ini_cm), trace["ini_cm"]) and np.array_equal(np.array(info.cur_cm), trace["cur_cm"])
    expected = vo.read_result_xml(os.path.join(golden_dir, "expected", name + ".xml"))
    result = sim.result()
    for tag, field in LW_TAGS:
        assert "%.6g" % getattr(result, field) == "%.6g" % expected[tag], tag


def test_shipped_example_result(golden_dir):
    # input file shipped with the reference simulator (5x5x5, per-voxel <PhaseOffset>, 3.24 s, collisions on)
    name = "example_phaseoffset"
    sim = _sim(golden_dir, name)
    sim.step(-1)
    expected = vo.read_result_xml(os.path.join(golden_dir, "expected", name + ".xml"))
    result = sim.result()
    for tag, field in RESULT_TAGS:
        assert "%.6g" % getattr(result, field) == "%.6g" % expected[tag], tag


def test_empty_and_single_voxel():
    base = vo.parse_vxa(os.path.join(os.path.dirname(__file__), "golden", "vxa", "phase4.vxa"))
    empty = dict(base)
    empty["structure"] = np.zeros_like(base["structure"])
    empty["phase_offset"] = None
    sim = vo.OracleSim(empty)
    assert sim.info().status == 3 and sim.step(-1) == 0       # the reference would loop forever
    single = dict(empty)
    cells = np.zeros_like(base["structure"])
    cells[0] = 3
    single["structure"] = cells
    sim = vo.OracleSim(single)
    info = sim.info()
    assert (info.nvox, info.nbond) == (1, 0)
    sim.step(-1)
    assert sim.info().status == 1 and np.isfinite(sim.state()).all()


def test_cm_trace_equals_the_reference_xml(golden_dir):
    """<TimeBetweenTraces> + <SaveTraces>: the points of SS.CMTrace (VX_Sim.cpp:1537-1547) as the reference binary printed them"""
    import re
    sim = _sim(golden_dir, "trace4")
    sim.step(-1)
    trace = sim.cm_trace()
    xml = open(os.path.join(golden_dir, "expected", "trace4.xml")).read()
    want = [re.findall(r"<%s>(.*?)</%s>" % (tag, tag), xml) for tag in ("Time", "TraceX", "TraceY", "TraceZ")]
    assert len(trace) == len(want[0]) == 13
    for k in range(4):
        assert ["%g" % v for v in trace[:, k]] == want[k]
    # without the tag: no trace
    plain = _sim(golden_dir, "phase4")
    plain.step(-1)
    assert len(plain.cm_trace()) == 0


def test_the_test_instruments_leave_the_algorithm_alone(golden_dir):
    """vxo_set_state / vxo_jitter are instruments of the GPU tests (tests/test_gpu_parity.py test_one_step_from_the_same_state, the
    ledger's spreads), not part of the restatement: an oracle put on ITS OWN state before every step must walk the trajectory of the
    untouched one (bit for bit in the poses; the momenta are rebuilt from the velocities, an ulp), and one-ulp noise must stay
    one-ulp-sized after one step -- the yardstick the one-step test reads the engine against."""
    import numpy as np
    from oracle import vxoracle as vo
    for name, variant in (("bench10_0", 0), ("lw_swim6", 1)):
        model = vo.parse_vxa(os.path.join(golden_dir, "vxa", name + ".vxa"), variant)
        lat = model["lattice_dim"]
        a, b, c = vo.OracleSim(model), vo.OracleSim(model), vo.OracleSim(model)
        worst_self = 0.0
        noise = []
        for step in range(120):
            before = a.state()
            b.set_state(before)
            c.set_state(before)
            a.step(1)
            b.step(1)
            c.step_jittered(1, seed=step + 1)
            sa, sb, sc = a.state(), b.state(), c.state()
            worst_self = max(worst_self, np.abs(sa[:, :3] - sb[:, :3]).max() / lat)
            noise.append(np.abs(sa[:, :3] - sc[:, :3]).max() / lat)
            assert a.info().steps == b.info().steps == c.info().steps == step + 1
        assert worst_self <= 1e-15, (name, worst_self)
        # (in the first steps of a robot at rest many bonds sit exactly on the small- / large-angle thresholds: there an ulp flips a
        # mode on one side, a 1e-10-voxel event; everywhere else the noise stays noise)
        noise = np.array(noise)
        assert 0 < np.median(noise) <= 1e-14 and (noise > 5e-14).sum() <= 12, (name, np.median(noise), noise.max(), (noise > 5e-14).sum())


def test_oracle_on_the_full_20_cube_equals_the_reference_binary(golden_dir, tmp_path):
    """BASELINE configs[4] at its full size (8000 voxels, 22 800 bonds, self-collision on, the whole 781-step evaluation): the final state
    of the C restatement hashes to what the reference binary's final state hashed to (tests/golden/make_cfg4_pin.py ->
    expected/cfg4_full20.json: SHA-256 of the [8000, 14] doubles, IniCM and CurCM as hex floats) -- bit-exact, like every other pin of
    the oracle.  The GPU tests of the tiled kernel (tests/test_gpu_tiled.py) compare with the oracle at this size; this closes the
    chain to the reference."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_cfg4_pin", os.path.join(golden_dir, "make_cfg4_pin.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(os.path.join(golden_dir, "expected", "cfg4_full20.json")) as f:
        pin = json.load(f)
    vxa = gen.cfg4_vxa(str(tmp_path))
    assert gen.vxa_digest(vxa) == pin["vxa_sha256"]      # the same input (environment, materials, structure)
    sim = vo.OracleSim.from_vxa(vxa)
    sim.step(-1)
    info = sim.info()
    assert (info.nvox, info.nbond, info.steps) == (pin["nvox"], pin["nbond"], pin["total_steps"])
    assert [float(x).hex() for x in info.ini_cm] == pin["ini_cm_hex"]
    assert [float(x).hex() for x in info.cur_cm] == pin["cur_cm_hex"]
    assert gen.state_digest(sim.state()) == pin["final_state_sha256"]
