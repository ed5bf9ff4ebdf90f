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


def test_eight_engines_behind_one_handle(golden_dir):
    """The shape of the first real 8-GPU contact (round-5 review, task 7): a handle over EIGHT engines -- here eight times the one device of
    the box: eight streams, eight host threads, the population partitioned eight ways by cost -- more robots than engines, and fewer
    in a second batch; every result equal to the single engine's, bit for bit."""
    from evosoro_amd import engine as eng_mod
    names = NAMES + ["rand6_col", "probe6", "soft5_init0", "rand6_nocol"]      # 11 robots on 8 engines
    paths = [os.path.join(golden_dir, "vxa", n + ".vxa") for n in names]
    with eng_mod.Engine(eng_mod.VOXCAD, 0) as one:
        one.set_option("tiled", 0)             # (engines that share a device do not tile: their tiles could not all be co-resident)
        one.add_vxa_files(paths)
        one.run()
        want = [one.result(i) for i in range(len(paths))]
        want_state = [one.state(i) for i in range(len(paths))]
    with eng_mod.Engine(eng_mod.VOXCAD, (0,) * 8) as eight:
        eight.add_vxa_files(paths)
        eight.run()
        for i, n in enumerate(names):
            assert _same(eight.result(i), want[i]), (i, n)
            assert np.array_equal(eight.state(i), want_state[i]), (i, n)
        eight.clear()
        eight.add_vxa_files(paths[:3])                   # fewer robots than engines: five of them stay idle
        eight.run()
        for i in range(3):
            assert _same(eight.result(i), want[i]), i


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
        many.set_option("tiled", 0)                       # (as `one`: a batch that lands on ONE engine may tile, whatever the handle)
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


def test_a_generation_pipelined_over_two_engines_of_one_device(tmp_path):
    """vxh_create_multi with a repeated device id + vxh_add_robots: the additions are only checked and copied, vxh_run builds, uploads
    and launches chunk after chunk before it waits for the first.  Nothing about a robot's result may
    depend on that: bit for bit the records of the one-engine route; a refused addition leaves nothing behind; readers and later
    additions see the robots whether or not they have been built yet."""
    from evosoro_amd import engine as eng_mod, workloads
    from evosoro_amd.base import Sim, Env
    from evosoro_amd.tools.read_write_voxelyze import write_voxelyze_file, phenotype_arrays
    os.makedirs(tmp_path / "voxelyzeFiles")
    sim = Sim(dt_frac=0.9, simulation_time=0.06, fitness_eval_init_time=0.02)
    pop = [workloads.random_robot(i, (6, 6, 6), 20 + i, phase_offset=(i % 2 == 0)) for i in range(21)] + \
          [workloads.random_robot(21 + i, (9, 9, 9), 70 + i) for i in range(4)]
    template = write_voxelyze_file(sim, Env(), pop[0], str(tmp_path), "t", write=False, want_text=True)[1]
    robots = []
    for ind in pop:
        material, layers = phenotype_arrays(ind)
        robots.append((material, layers, None))

    def run(device):
        with eng_mod.Engine(eng_mod.VOXCAD, device) as eng:
            eng.set_option("tiled", 0)                       # (engines that share a device do not tile: the same kernels on both sides,
            first = eng.add_robots(template, robots[:10])    # whatever VXH_ENGINE_OPTIONS says)
            second = eng.add_robots(template, robots[10:])
            assert (first, second, eng.num_robots()) == (0, 10, len(robots))
            eng.run()
            return [eng.result(i) for i in range(len(robots))], [eng.state(i) for i in (0, 9, 10, 24)]

    want, want_states = run(0)
    got, got_states = run((0, 0))
    for i in range(len(robots)):
        assert got[i].status == eng_mod.ROBOT_FINISHED and _same(got[i], want[i]), i
    for a, b in zip(got_states, want_states):
        assert np.array_equal(a, b)
    with eng_mod.Engine(eng_mod.VOXCAD, (0, 0)) as eng:
        eng.set_option("tiled", 0)
        eng.add_robots(template, robots[:5])
        bad = np.array(robots[5][0]).copy()
        bad[0, 0, 0] = 9                                     # a material index outside the palette: refused at the addition
        with pytest.raises(eng_mod.VxhError):
            eng.add_robots(template, [(bad, robots[5][1], None)])
        assert eng.num_robots() == 5
        assert eng.dims(3)["nvox"] == want[3].nvox           # a reader before the run: the models are built on demand
        eng.add_robots(template, robots[5:7])
        eng.run()
        for i in range(7):
            assert _same(eng.result(i), want[i]), i


def test_a_repeated_device_handle_tiles_again_when_one_engine_holds_the_batch(golden_dir, tmp_path):
    """Engines that share a device do not launch the multi-workgroup kernel while they step side by side; the veto is per batch, not for
    the life of the handle: after a small generation spread over both engines, one lattice above 1024 voxels (file route: it lands on
    ONE engine) is stepped by k_tile_steps again, with the record of a one-device engine."""
    from evosoro_amd import engine as eng_mod, workloads
    from evosoro_amd.base import Sim, Env
    from evosoro_amd.tools.read_write_voxelyze import write_voxelyze_file
    if os.environ.get("VXH_ENGINE_OPTIONS"):
        pytest.skip("kernel path forced by VXH_ENGINE_OPTIONS")
    os.makedirs(tmp_path / "voxelyzeFiles")
    sim = Sim(dt_frac=0.9, simulation_time=0.02, fitness_eval_init_time=0.005)
    write_voxelyze_file(sim, Env(), workloads.make_individual(0, workloads.full_material(11, 1)), str(tmp_path), "big")
    big = str(tmp_path / "voxelyzeFiles" / "big--id_00000.vxa")
    small = [os.path.join(golden_dir, "vxa", n + ".vxa") for n in ("probe6", "rand6_col")]
    with eng_mod.Engine(eng_mod.VOXCAD, 0) as one:
        one.add_vxa_file(big)
        one.run()
        want, want_state = one.result(0), one.state(0)
        assert one.counters().dominant_block == 1
    with eng_mod.Engine(eng_mod.VOXCAD, (0, 0)) as two:
        for p in small:
            two.add_vxa_file(p)
        two.run()                                             # both engines busy: no tiling in this batch
        two.clear()
        two.add_vxa_file(big)
        two.run()
        assert two.counters().dominant_block == 1             # k_tile_steps, not the streaming kernels
        assert _same(two.result(0), want) and np.array_equal(two.state(0), want_state)
        # ... and a mixed batch afterwards (spread again) steps the lattice on the streaming kernels: same state to the last bit is NOT
        # promised across kernels, the record's step count and status are
        two.add_vxa_file(small[0])
        two.run()
        assert two.result(0).status == eng_mod.ROBOT_FINISHED and two.result(0).steps == want.steps
