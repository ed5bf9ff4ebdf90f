"""The file formats on either side of the simulator: .vxa writer and result reader.

Golden .vxa texts were produced by IMPORTING the reference writer (tests/golden/make_golden.py); ours must
reproduce them byte for byte, consume the `random` stream identically and return the same md5 cache key.
"""
import hashlib
import os
import random
from collections import OrderedDict

import numpy as np
import pytest

from evosoro_amd.base import Sim, Env, ObjectiveDict
from evosoro_amd.tools.read_write_voxelyze import write_voxelyze_file, read_voxlyze_results
from evosoro_amd.tools.utils import py2_str, xml_format
from evosoro_amd import workloads


def _cases():
    out = OrderedDict()
    out["probe6"] = (Sim(dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0.1), Env(),
                     workloads.make_individual(0, workloads.probe_material()))
    out["rand6_nocol"] = (Sim(self_collisions_enabled=False, dt_frac=0.9, simulation_time=0.2,
                              fitness_eval_init_time=0.05), Env(), workloads.random_robot(1, (6, 6, 6), 3))
    out["rand6_col"] = (Sim(dt_frac=0.9, simulation_time=0.25, fitness_eval_init_time=0.1), Env(),
                        workloads.random_robot(2, (6, 6, 6), 7))
    env4 = Env(frequency=5.0, temp_amp=35)
    env4.add_param("growth_amplitude", 0.3, "<GrowthAmplitude>")
    phase = np.round(np.random.RandomState(99).uniform(-1, 1, size=(4, 4, 4)), 3)
    out["phase4"] = (Sim(dt_frac=0.8, simulation_time=0.3, fitness_eval_init_time=0.05), env4,
                     workloads.make_individual(4, workloads.random_material((4, 4, 4), 5, 0.1),
                                               OrderedDict([("<PhaseOffset>", phase)])))
    return out


@pytest.mark.parametrize("name", list(_cases()))
def test_writer_matches_reference_bytes(tmp_path, golden_dir, manifest, name):
    sim, env, ind = _cases()[name]
    os.makedirs(tmp_path / "golden_run" / "voxelyzeFiles")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        random.seed(12345)
        md5 = write_voxelyze_file(sim, env, ind, "golden_run", "golden")
        after = random.random()
    finally:
        os.chdir(cwd)
    ours = (tmp_path / "golden_run" / "voxelyzeFiles" / ("golden--id_%05i.vxa" % ind.id)).read_text()
    with open(os.path.join(golden_dir, "vxa", name + ".vxa")) as handle:
        assert ours == handle.read()
    assert md5 == manifest[name]["md5"]
    # three random.uniform draws per file (materials 3, 4, 6), even with zero actuation variance
    random.seed(12345)
    for _ in range(3):
        random.uniform(0, 0)
    assert after == random.random()


def test_md5_is_over_cell_strings():
    ind = workloads.make_individual(7, np.array([[[1, 0], [3, 4]]]))
    sim, env = Sim(), Env()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "voxelyzeFiles"))
        md5 = write_voxelyze_file(sim, env, ind, d, "x")
    # x fastest, then y, then z: state[x, y, z]
    assert md5 == hashlib.md5("1304".encode()).hexdigest()


def test_reader_matches_reference_reader(golden_dir, manifest):
    class Pop(object):
        pass
    pop = Pop()
    pop.objective_dict = ObjectiveDict()
    pop.objective_dict.add_objective(name="fitness", maximize=True, tag="<NormFinalDist>")
    pop.objective_dict.add_objective(name="age", maximize=False, tag=None)
    pop.objective_dict.add_objective(name="y", maximize=True, tag="<finalDistY>")
    pop.objective_dict.add_objective(name="touch", maximize=True, tag="<NumTouchingFloor>")
    for name in ("probe6", "rand6_col", "phase4"):
        got = read_voxlyze_results(pop, None, os.path.join(golden_dir, "expected", name + ".xml"))
        want = manifest[name]["read_results"]
        assert {str(k): v for k, v in got.items()} == want


def test_objective_dict_rank_rules():
    od = ObjectiveDict()
    od.add_objective("age", False, None)
    od.add_objective("fitness", True, "NormFinalDist")
    assert od[0]["name"] == "fitness" and od[0]["tag"] == "<NormFinalDist>" and od[0]["worst_value"] == -10e6
    assert od[1]["name"] == "age" and od[1]["worst_value"] == 10e6


def test_py2_float_formatting():
    assert py2_str(1.0 / 3.0) == "0.333333333333"     # python2 str(): 12 significant digits
    assert py2_str(0.25) == "0.25" and py2_str(5e6) == "5000000.0" and py2_str(10) == "10"
    assert py2_str(1e-05) == "1e-05" and py2_str(np.float64(0.5)) == "0.5" and py2_str(1e22) == "1e+22"
    assert xml_format("Tag") == "<Tag>" and xml_format("<Tag>") == "<Tag>"
