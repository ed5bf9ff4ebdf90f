"""GPU test (-m gpu) of the one-handle-over-several-devices route (vxh_create_multi, include/vxhip.h): the robots of a batch are
partitioned over the devices by cost and stepped from one host thread per device; nothing about a robot's result may depend on
which device it landed on or on who else shared it.  The GPU box has ONE device, so the handle is opened on {0, 0}: two engines,
two streams, two host threads -- everything of the route except the second piece of silicon."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ["probe6", "rand6_nocol", "rand6_col", "soft5_init0", "stiff5", "grow5", "devo4"]
SKIP = ("reserved",)


def _val(res, f):
    v = getattr(res, f)
    return tuple(v) if hasattr(v, "__len__") else v


def _same(a, b):
    return all(_val(a, f) == _val(b, f) for f, _ in a._fields_ if f not in SKIP)


def test_two_engines_behind_one_handle(golden_dir):
    from evosoro_amd import engine as eng_mod
    paths = [os.path.join(golden_dir, "vxa", n + ".vxa") for n in NAMES]
    with eng_mod.Engine(eng_mod.VOXCAD, 0) as one:
        one.set_option("tiled", 0)
        for p in paths:
            one.add_vxa_file(p)
        one.run()
        want = [one.result(i) for i in range(len(paths))]
        want_state = [one.state(i) for i in range(len(paths))]
        want_counters = one.counters()
    with eng_mod.Engine(eng_mod.VOXCAD, (0, 0)) as two:
        for p in paths[:3]:
            two.add_vxa_file(p)
        assert two.add_vxa_files(paths[3:]) == 3         # index of the first robot added
        assert two.num_robots() == len(paths)
        two.run()
        for i, n in enumerate(NAMES):
            got = two.result(i)
            assert _same(got, want[i]), (n, got.as_dict(), want[i].as_dict())
            assert np.array_equal(two.state(i), want_state[i]), n
        c = two.counters()
        assert c.voxel_steps == want_counters.voxel_steps and c.bond_steps == want_counters.bond_steps
        # an addition after a run brings the robots back together; the next run starts everything afresh
        two.add_vxa_file(paths[0])
        two.run()
        assert _val(two.result(len(paths)), "cur_cm") == _val(want[0], "cur_cm")
        assert _val(two.result(2), "cur_cm") == _val(want[2], "cur_cm")


def test_cli_over_a_device_list(golden_dir, tmp_path):
    import subprocess
    from evosoro_amd import engine as eng_mod
    args = []
    for n in NAMES[:4]:
        args += ["-f", os.path.join(golden_dir, "vxa", n + ".vxa")]
    outs = {}
    for devs in ("0", "0,0"):
        d = tmp_path / ("out" + devs.replace(",", "_"))
        os.makedirs(d / "golden_run" / "fitnessFiles")
        proc = subprocess.run([eng_mod.CLI_PATH] + args + ["--devices", devs], cwd=d, timeout=600)
        assert proc.returncode == 1                      # the reference's "success" code
        fit = d / "golden_run" / "fitnessFiles"
        outs[devs] = {f: (fit / f).read_text() for f in sorted(os.listdir(fit))}
    assert len(outs["0"]) == 4 and outs["0"] == outs["0,0"]


def test_more_engines_than_robots_and_reuse_of_the_handle(golden_dir):
    """Edge cases of the several-devices handle: three engines for two robots (one engine holds nothing), an empty handle, and the
    handle used again after vxh_reset and after vxh_clear -- every time the results of the one-device engine, bit for bit."""
    from evosoro_amd import engine as eng_mod
    paths = [os.path.join(golden_dir, "vxa", n + ".vxa") for n in ("probe6", "rand6_col")]
    with eng_mod.Engine(eng_mod.VOXCAD, 0) as one:
        one.set_option("tiled", 0)
        for p in paths:
            one.add_vxa_file(p)
        one.run()
        want = [one.result(i) for i in range(2)]
        one.reset()
        one.step(300)
        want_300 = [one.state(i) for i in range(2)]
    with eng_mod.Engine(eng_mod.VOXCAD, (0, 0, 0)) as many:
        assert many.num_robots() == 0
        many.run()                                        # nothing to do is not an error
        for p in paths:
            many.add_vxa_file(p)
        many.run()
        for i in range(2):
            assert _same(many.result(i), want[i]), i
        large, total = many.bond_modes()
        assert 0 <= large <= total and total > 0
        many.reset()                                      # same robots, from the start
        many.step(300)
        for i in range(2):
            assert np.array_equal(many.state(i), want_300[i]), i
        many.run()
        for i in range(2):
            assert _same(many.result(i), want[i]), i
        many.clear()
        assert many.num_robots() == 0
        many.add_vxa_file(paths[1])                       # a different population afterwards
        many.run()
        assert _same(many.result(0), want[1])
