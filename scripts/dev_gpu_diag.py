#!/usr/bin/env python3
"""Developer diagnostics for the GPU box (not collected by pytest, not part of the product): per-step error of the HIP
engine against the CPU oracle on the golden robots, timings of synthetic batches, per-phase cycle shares (library
built by `make -C evosoro_amd/csrc prof`), whole-generation wall clock.  Uses oracle/ as the checker, like the tests do."""
import os
import sys
import time
import shutil
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from evosoro_amd import engine, workloads  # noqa: E402
from evosoro_amd.base import Sim, Env  # noqa: E402
from evosoro_amd.tools.read_write_voxelyze import write_voxelyze_file  # noqa: E402
from oracle import vxoracle as vo  # noqa: E402

CASES = ["probe6", "rand6_nocol", "rand6_col", "soft5_init0", "phase4"]
G = os.path.join(REPO, "tests", "golden")


def errors():
    with engine.Engine(engine.VOXCAD, 0) as eng:
        for n in CASES:
            eng.add_vxa_file(os.path.join(G, "vxa", n + ".vxa"))
        sims = [vo.OracleSim.from_vxa(os.path.join(G, "vxa", n + ".vxa")) for n in CASES]
        done = 0
        for upto in (1, 2, 5, 10, 50, 200, 1000):
            eng.step(upto - done)
            done = upto
            for i, (n, sim) in enumerate(zip(CASES, sims)):
                sim.step(upto - sim.info().steps)
                w, g = sim.state(), eng.state(i)
                lat = sim.model["lattice_dim"]
                print("step %5d %-12s pos %.3e vox  quat %.3e  scale %.3e  vel %.3e" % (
                    upto, n, np.abs(g[:, :3] - w[:, :3]).max() / lat, np.abs(g[:, 3:7] - w[:, 3:7]).max(),
                    np.abs(g[:, 7] - w[:, 7]).max() / lat, np.abs(g[:, 8:11] - w[:, 8:11]).max()))
        eng.run()
        for i, n in enumerate(CASES):
            tr = vo.read_trace(os.path.join(G, "expected", n + ".final.bin"))
            x = vo.read_result_xml(os.path.join(G, "expected", n + ".xml"))
            r = eng.result(i)
            lat = sims[i].model["lattice_dim"]
            print("final %-12s status %d steps %d/%d  dCoM %.3e vox  NormFinalDist %.6g (ref %.6g) finalDistY %.6g (ref %.6g)" % (
                n, r.status, r.steps, tr["total_steps"], np.abs(np.array(r.cur_cm) - tr["cur_cm"]).max() / lat,
                r.norm_final_dist, x["NormFinalDist"], r.final_dist_y, x["finalDistY"]))


def timing(count, shape, sim_time, selfcol=True, graph_steps=32):
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    sim = Sim(self_collisions_enabled=selfcol, dt_frac=0.9, simulation_time=sim_time, fitness_eval_init_time=min(0.1, sim_time / 5))
    env = Env()
    with engine.Engine(engine.VOXCAD, 0) as eng:
        eng.set_option("graph_steps", graph_steps)
        for ind in workloads.population(count, shape):
            write_voxelyze_file(sim, env, ind, tmp, "t")
            eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "t--id_%05i.vxa" % ind.id))
        t0 = time.time()
        eng.run()
        wall = time.time() - t0
        c = eng.counters()
        st = [eng.result(i).status for i in range(count)]
        print("batch %d x %s sim %.3fs col=%d graph=%d: voxel_steps %.3e max_steps %d kernel %.3fs wall %.3fs -> %.3e vox-steps/s (kernel), alg GB/s %.1f, statuses %s" % (
            count, shape, sim_time, selfcol, graph_steps, c.voxel_steps, c.max_steps, c.kernel_seconds, wall,
            c.voxel_steps / c.kernel_seconds, c.algorithmic_bytes / c.kernel_seconds / 1e9, sorted(set(st))))


if __name__ == "__main__":
    what = sys.argv[1:] or ["errors", "timing"]
    if "errors" in what:
        errors()
    if "timing" in what:
        timing(64, (6, 6, 6), 0.05)
        timing(64, (10, 10, 10), 0.02)
        timing(512, (10, 10, 10), 0.01)
        timing(512, (10, 10, 10), 0.01, graph_steps=0)


def bisect(name, nocol):
    text = open(os.path.join(G, "vxa", name + ".vxa")).read()
    if nocol:
        text = text.replace("<SelfColEnabled>1</SelfColEnabled>", "<SelfColEnabled>0</SelfColEnabled>")
    sim = vo.OracleSim(vo.parse_vxa(text))
    with engine.Engine(engine.VOXCAD, 0) as eng:
        eng.add_vxa_text(text)
        for upto in list(range(1, 60)) + [80, 100, 150, 200]:
            eng.step(upto - sim.info().steps)
            sim.step(upto - sim.info().steps)
            w, g = sim.state(), eng.state(0)
            lat = sim.model["lattice_dim"]
            i = sim.info()
            e = np.abs(g[:, :3] - w[:, :3]).max(axis=1) / lat
            print("%s nocol=%d step %4d pos %.3e vox (voxel %d) ncol_oracle %d rebuilds oracle %d" % (
                name, nocol, upto, e.max(), int(e.argmax()), i.ncol, i.col_rebuilds))
        eng.run()
        print("gpu rebuilds", eng.result(0).col_rebuilds)


if __name__ == "__main__" and "bisect" in sys.argv[1:]:
    bisect("phase4", False)
    bisect("phase4", True)


if __name__ == "__main__" and "prof" in sys.argv[1:]:
    timing(512, (10, 10, 10), 0.01)


if __name__ == "__main__" and "prof_nocol" in sys.argv[1:]:
    timing(512, (10, 10, 10), 0.004, selfcol=False)


PHASES = "phases" in sys.argv[1:]
if PHASES:   # library built by `make -C evosoro_amd/csrc prof`
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libvxhip_prof.so")


def timing2(count, shape, sim_time, selfcol, opts):
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    sim = Sim(self_collisions_enabled=selfcol, dt_frac=0.9, simulation_time=sim_time, fitness_eval_init_time=min(0.1, sim_time / 5))
    env = Env()
    with engine.Engine(engine.VOXCAD, 0) as eng:
        for k, val in opts.items():
            eng.set_option(k, val)
        for ind in workloads.population(count, shape):
            write_voxelyze_file(sim, env, ind, tmp, "t")
            eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "t--id_%05i.vxa" % ind.id))
        eng.run()
        c = eng.counters()
        st = [eng.result(i).status for i in range(count)]
        cm = np.array([eng.result(i).cur_cm for i in range(count)])
        if PHASES:
            nres = [(eng.result(i).col_rebuilds, eng.result(i).nvox) for i in range(count)]
            eng.clear()          # the developer build prints its per-wave phase shares here
            return
        print("   mean rebuilds per robot %.1f, mean nvox %.0f" % (np.mean([eng.result(i).col_rebuilds for i in range(count)]), np.mean([eng.result(i).nvox for i in range(count)])))
        print("batch %d x %s sim %.3fs col=%d %s: max_steps %d kernel %.4fs -> %.3e vox-steps/s, %.1f us/step, alg GB/s %.1f, statuses %s cmsum %.12g" % (
            count, shape, sim_time, selfcol, opts, c.max_steps, c.kernel_seconds,
            c.voxel_steps / c.kernel_seconds, 1e6 * c.kernel_seconds / c.max_steps, c.algorithmic_bytes / c.kernel_seconds / 1e9, sorted(set(st)), cm.sum()))


if __name__ == "__main__" and "fused" in sys.argv[1:]:
    for col in (False, True):
        for opts in ({"fused": 0}, {"fused": 1, "steps_per_launch": 1}, {"fused": 1, "steps_per_launch": 64}):
            timing2(512, (10, 10, 10), 0.01, col, opts)
    for opts in ({"fused": 0}, {"fused": 1, "steps_per_launch": 64}):
        timing2(64, (10, 10, 10), 0.02, True, opts)
        timing2(64, (6, 6, 6), 0.05, True, opts)
        timing2(2048, (6, 6, 6), 0.01, True, opts)


if __name__ == "__main__" and "col" in sys.argv[1:]:
    timing2(512, (10, 10, 10), 0.01, True, {"fused": 1, "steps_per_launch": 64})
    timing2(512, (10, 10, 10), 0.05, True, {"fused": 1, "steps_per_launch": 64})
    timing2(512, (10, 10, 10), 0.05, False, {"fused": 1, "steps_per_launch": 64})


if __name__ == "__main__" and "dbg" in sys.argv[1:]:
    for d in (0, 1, 2, 3):
        timing2(512, (10, 10, 10), 0.03, True, {"fused": 1, "steps_per_launch": 64, "dbg": d})


def bisect_lw(name):
    path = os.path.join(G, "vxa", name + ".vxa")
    sim = vo.OracleSim.from_vxa(path, 1)
    with engine.Engine(engine.VOXCAD_LAND_WATER, 0) as eng:
        eng.add_vxa_file(path)
        for upto in list(range(1, 12)) + [20, 50]:
            eng.step(upto - sim.info().steps)
            sim.step(upto - sim.info().steps)
            w, g = sim.state(), eng.state(0)
            lat = sim.model["lattice_dim"]
            e = np.abs(g[:, :3] - w[:, :3]).max(axis=1) / lat
            ev = np.abs(g[:, 8:11] - w[:, 8:11]).max(axis=1)
            print("%s step %4d pos %.3e vox (voxel %d) vel %.3e (voxel %d) scale %.3e ncol %d" % (
                name, upto, e.max(), int(e.argmax()), ev.max(), int(ev.argmax()), np.abs(g[:, 7] - w[:, 7]).max() / lat, sim.info().ncol))


if __name__ == "__main__" and "bisect_lw" in sys.argv[1:]:
    bisect_lw("lw_hexapus")
    bisect_lw("lw_swim6")


if __name__ == "__main__" and PHASES:
    for col, dbg in ((True, 0), (True, 1), (True, 3), (False, 0)):
        print("col", col, "dbg", dbg, flush=True)
        timing2(512, (10, 10, 10), 0.02, col, {"fused": 1, "steps_per_launch": 64, "dbg": dbg})


if __name__ == "__main__" and "spl" in sys.argv[1:]:
    for spl in (64, 128, 256, 512):
        timing2(512, (10, 10, 10), 0.05, True, {"fused": 1, "steps_per_launch": spl})


_WARM = set()


def _warm_up(variant, opts):
    """the first launch of a kernel in a process pays for loading its code object (milliseconds): keep that out of the timings"""
    key = (variant, tuple(sorted((k, v) for k, v in opts.items() if k in ("tiled", "fused"))))
    if key in _WARM:
        return
    _WARM.add(key)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    sim = Sim(dt_frac=0.9, simulation_time=0.002, fitness_eval_init_time=0.001)
    with engine.Engine(variant, 0) as eng:
        for k, val in opts.items():
            if k in ("tiled", "fused", "tiles_per_robot"):
                eng.set_option(k, min(val, 2) if k == "tiles_per_robot" else val)
        for i, shape in enumerate(((4, 4, 4), (7, 7, 7), (9, 9, 9), (10, 10, 10))):
            ind = workloads.make_individual(i, workloads.full_material(shape[0], 1 + i))
            write_voxelyze_file(sim, Env(), ind, tmp, "w")
            eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "w--id_%05i.vxa" % i))
        eng.run()


KEEP_STATES = False
LAST_STATES = None


def timing_cfg(variant, count, shape, sim_time, env, opts, full=False, per_voxel_phase=False, phases=False, stiffness=False, selfcol=True):
    """BASELINE configs[3]/[4]: throughput of other workloads (not the bench line)"""
    from collections import OrderedDict
    _warm_up(variant, opts)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    sim = Sim(dt_frac=0.9, simulation_time=sim_time, fitness_eval_init_time=min(0.05, sim_time / 5), self_collisions_enabled=selfcol)
    with engine.Engine(variant, 0) as eng:
        for k, val in opts.items():
            eng.set_option(k, val)
        for i in range(count):
            mat = workloads.full_material(shape[0], 1 + i) if full else workloads.random_material(shape, i)
            extra = None
            if per_voxel_phase:
                extra = OrderedDict([("<PhaseOffset>", np.round(np.random.RandomState(i).uniform(-1, 1, size=shape), 3))])
            if stiffness:      # per-voxel evolved stiffness: nearly every bond its own class (TABG kernel variants from ~300 voxels on)
                extra = extra or OrderedDict()
                extra["<Stiffness>"] = np.round(10 ** np.random.RandomState(1000 + i).uniform(6.0, 7.7, size=shape), 0)
            ind = workloads.make_individual(i, mat, extra)
            write_voxelyze_file(sim, env, ind, tmp, "t")
            eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "t--id_%05i.vxa" % i))
        eng.run()
        c = eng.counters()
        st = sorted(set(eng.result(i).status for i in range(count)))
        print("   broad-phase runs per robot: max %d" % max(eng.result(i).col_rebuilds for i in range(count)), flush=True)
        global LAST_STATES
        LAST_STATES = [eng.state(i) for i in range(count)] if KEEP_STATES else None
        if phases:
            eng.clear()
            return
        print("variant %d: %d x %s full=%s sim %.3fs %s: max_steps %d kernel %.4fs -> %.3e vox-steps/s, %.1f us/step, alg GB/s %.1f, statuses %s" % (
            variant, count, shape, full, sim_time, opts, c.max_steps, c.kernel_seconds, c.voxel_steps / c.kernel_seconds,
            1e6 * c.kernel_seconds / c.max_steps, c.algorithmic_bytes / c.kernel_seconds / 1e9, st), flush=True)


if __name__ == "__main__" and "lwcfgs" in sys.argv[1:]:
    # swimmers through the four mesh variants of the fused kernel (256, 512, 768, 1024 threads); VXH_LIB=<path> to compare builds
    if os.environ.get("VXH_LIB"):
        engine.LIB_PATH = os.environ["VXH_LIB"]
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    timing_cfg(engine.VOXCAD_LAND_WATER, 1024, (6, 6, 6), 0.05, env_w, {}, per_voxel_phase=True)
    timing_cfg(engine.VOXCAD_LAND_WATER, 512, (8, 8, 8), 0.05, env_w, {}, per_voxel_phase=True)
    timing_cfg(engine.VOXCAD_LAND_WATER, 512, (10, 10, 10), 0.03, env_w, {}, per_voxel_phase=True)
    timing_cfg(engine.VOXCAD_LAND_WATER, 256, (10, 10, 10), 0.02, env_w, {}, full=True, per_voxel_phase=True)
    # the same simulator on land (no drag; the MESH variants still record the strains for the RobotVolumeEnd tag)
    timing_cfg(engine.VOXCAD_LAND_WATER, 512, (8, 8, 8), 0.05, Env(), {}, per_voxel_phase=True)
    timing_cfg(engine.VOXCAD_LAND_WATER, 512, (10, 10, 10), 0.03, Env(), {}, per_voxel_phase=True)


if __name__ == "__main__" and "stiffcfgs" in sys.argv[1:]:
    timing_cfg(engine.VOXCAD, 512, (6, 6, 6), 0.03, Env(), {}, stiffness=True)
    timing_cfg(engine.VOXCAD, 512, (8, 8, 8), 0.03, Env(), {}, stiffness=True)
    timing_cfg(engine.VOXCAD, 512, (10, 10, 10), 0.02, Env(), {}, stiffness=True)
    timing_cfg(engine.VOXCAD, 512, (10, 10, 10), 0.02, Env(), {})


if __name__ == "__main__" and "cfgs" in sys.argv[1:]:
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    timing_cfg(engine.VOXCAD_LAND_WATER, 64, (8, 8, 8), 0.05, env_w, {}, per_voxel_phase=True)      # configs[3]
    timing_cfg(engine.VOXCAD_LAND_WATER, 512, (8, 8, 8), 0.05, env_w, {}, per_voxel_phase=True)
    timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.05, Env(), {})                                        # configs[1]
    timing_cfg(engine.VOXCAD, 2048, (6, 6, 6), 0.05, Env(), {})
    timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.01, Env(), {}, full=True)                           # configs[4]
    timing_cfg(engine.VOXCAD, 512, (10, 10, 10), 0.02, Env(), {}, full=True)                         # dense 10^3 (1024-thread variant)


if __name__ == "__main__" and "e2e" in sys.argv[1:]:
    # whole-generation wall clock through the file boundary: write 512 .vxa, parse + upload, 0.5 s of simulated time, result XMLs
    import time
    tmp = tempfile.mkdtemp()
    for d in ("voxelyzeFiles", "fitnessFiles"):
        os.makedirs(os.path.join(tmp, d))
    sim = Sim(dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0.1)
    env = Env()
    pop = list(workloads.population(512, (10, 10, 10)))
    t0 = time.perf_counter()
    paths = []
    for ind in pop:
        write_voxelyze_file(sim, env, ind, tmp, "t")
        paths.append(os.path.join(tmp, "voxelyzeFiles", "t--id_%05i.vxa" % ind.id))
    t1 = time.perf_counter()
    with engine.Engine(engine.VOXCAD, 0) as eng:
        tc = time.perf_counter()
        eng.add_vxa_files(paths)
        t2 = time.perf_counter()
        eng.run()
        t3 = time.perf_counter()
        for i in range(len(paths)):
            eng.write_result_xml(i, os.path.join(tmp, "fitnessFiles", "o%05i.xml" % i))
        t4 = time.perf_counter()
        c = eng.counters()
        print("e2e 512 x 10^3, 0.5 s simulated: python .vxa writer %.2f s | HIP runtime + engine creation (once per process) %.2f s | parse + build %.3f s | "
              "upload+run %.2f s (kernel %.2f s, %d max steps) | download+XML %.2f s | generation total %.2f s -> %.3e vox-steps/s end to end vs %.3e in-kernel" % (
                  t1 - t0, tc - t1, t2 - tc, t3 - t2, c.kernel_seconds, c.max_steps, t4 - t3, t4 - tc, c.voxel_steps / (t4 - tc), c.voxel_steps / c.kernel_seconds))


if __name__ == "__main__" and "e2emem" in sys.argv[1:]:
    # the same generation through the in-memory hand-off (vxh_add_robots): the template's text once, then arrays; result values from
    # vxh_get_result (device reductions) instead of result XMLs.  VERDICT item 7: below the 0.38 s of the file route?
    import time
    from evosoro_amd.tools.read_write_voxelyze import phenotype_arrays
    tmp = tempfile.mkdtemp()
    for d in ("voxelyzeFiles", "fitnessFiles"):
        os.makedirs(os.path.join(tmp, d))
    sim = Sim(dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0.1)
    env = Env()
    pop = list(workloads.population(512, (10, 10, 10)))
    n_eng = int(os.environ.get("VXH_E2E_ENGINES", "1"))      # > 1: one handle, several engines (host threads, streams) on the ONE device
    with engine.Engine(engine.VOXCAD, [0] * n_eng if n_eng > 1 else 0) as eng:
        eng.add_vxa_text(write_voxelyze_file(sim, env, pop[0], tmp, "w", write=False, want_text=True)[1]); eng.step(1); eng.clear()   # (HIP runtime, code objects: once per process)
        for rep in range(2):
            t0 = time.perf_counter()
            template = write_voxelyze_file(sim, env, pop[0], tmp, "t", write=False, want_text=True)[1]      # (returns (md5, text))
            robots = []
            for ind in pop:
                material, layers = phenotype_arrays(ind)
                robots.append((material, layers, None))
            t1 = time.perf_counter()
            eng.add_robots(template, robots)
            t2 = time.perf_counter()
            eng.run()
            t3 = time.perf_counter()
            fit = [eng.result(i).norm_final_dist for i in range(len(pop))]
            t4 = time.perf_counter()
            c = eng.counters()
            print("e2e in memory, 512 x 10^3, 0.5 s simulated (rep %d): arrays from the genotypes %.3f s | vxh_add_robots (build) %.3f s | upload+run %.3f s (kernel %.3f s, "
                  "%d max steps) | results %.3f s | generation total %.3f s -> %.3e vox-steps/s end to end vs %.3e in-kernel" % (
                      rep, t1 - t0, t2 - t1, t3 - t2, c.kernel_seconds, c.max_steps, t4 - t3, t4 - t0, c.voxel_steps / (t4 - t0), c.voxel_steps / c.kernel_seconds), flush=True)
            eng.clear()


if __name__ == "__main__" and "small" in sys.argv[1:]:
    timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.05, Env(), {})
    timing_cfg(engine.VOXCAD, 2048, (6, 6, 6), 0.05, Env(), {})
    timing_cfg(engine.VOXCAD, 4096, (6, 6, 6), 0.05, Env(), {})


if __name__ == "__main__" and "lwphases" in sys.argv[1:]:
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libvxhip_prof.so")
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    timing_cfg(engine.VOXCAD_LAND_WATER, 512, (8, 8, 8), 0.03, env_w, {}, per_voxel_phase=True, phases=True)
    if "10" in sys.argv[1:]:      # the 768-thread variant (swimmers of up to 768 voxels)
        timing_cfg(engine.VOXCAD_LAND_WATER, 512, (10, 10, 10), 0.03, env_w, {}, per_voxel_phase=True, phases=True)


if __name__ == "__main__" and "crosscheck" in sys.argv[1:]:
    # the two kernel families (fused resident / streaming) on the whole bench population: max state difference per robot
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    sim = Sim(dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0.05)
    paths = []
    for ind in workloads.population(512, (10, 10, 10)):
        write_voxelyze_file(sim, Env(), ind, tmp, "t")
        paths.append(os.path.join(tmp, "voxelyzeFiles", "t--id_%05i.vxa" % ind.id))
    states = {}
    for fused in (1, 0):
        with engine.Engine(engine.VOXCAD, 0) as eng:
            eng.set_option("fused", fused)
            eng.add_vxa_files(paths)
            eng.step(400)
            states[fused] = [eng.state(i) for i in range(len(paths))]
    diff = np.array([np.abs(a[:, :3] - b[:, :3]).max() / 0.01 for a, b in zip(states[1], states[0])])
    print("fused vs streaming after 400 steps, position difference in voxels: median %.2e, 90%% %.2e, 99%% %.2e, max %.2e" % (
        np.median(diff), np.percentile(diff, 90), np.percentile(diff, 99), diff.max()))


if __name__ == "__main__" and "tilecfgs" in sys.argv[1:]:
    # the multi-workgroup kernel against the round-1 paths on the populations that cannot fill the chip one robot per CU
    for opts in ({"tiled": 0}, {"tiled": 1}):
        timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.01, Env(), opts, full=True)                     # configs[4]
        timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.05, Env(), opts)                                  # configs[1]
        timing_cfg(engine.VOXCAD, 64, (10, 10, 10), 0.03, Env(), opts)                               # configs[2] at 8 GPUs: 64 per GPU
        timing_cfg(engine.VOXCAD, 16, (10, 10, 10), 0.03, Env(), opts)
    for k in (27, 64, 125, 216):
        timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.01, Env(), {"tiled": 2, "tiles_per_robot": k}, full=True)
    for k in (1, 2, 4, 8):
        timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.05, Env(), {"tiled": 2, "tiles_per_robot": k})


if __name__ == "__main__" and "tilephases" in sys.argv[1:]:
    # per-phase cycle shares of the tiled kernel (library built by `make -C evosoro_amd/csrc prof`)
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libvxhip_prof.so")
    engine._lib = None
    print("20^3, 125 tiles", flush=True)
    timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.01, Env(), {"tiled": 2, "tiles_per_robot": 125}, full=True, phases=True)
    print("64 x 6^3, 4 tiles each", flush=True)
    timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.05, Env(), {"tiled": 2, "tiles_per_robot": 4}, phases=True)
    print("64 x 6^3, 1 tile each", flush=True)
    timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.05, Env(), {"tiled": 2, "tiles_per_robot": 1}, phases=True)


if __name__ == "__main__" and "tilelong" in sys.argv[1:]:
    if "prof" in sys.argv[1:]:
        engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libvxhip_prof.so")
        engine._lib = None
    timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.08, Env(), {"tiled": 2, "tiles_per_robot": 125}, full=True, phases="prof" in sys.argv[1:])
    timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.08, Env(), {"tiled": 2, "tiles_per_robot": 64}, full=True, phases="prof" in sys.argv[1:])


if __name__ == "__main__" and "tilefence" in sys.argv[1:]:
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libvxhip_prof.so")
    engine._lib = None
    for dbg in (0, 4):
        print("dbg", dbg, flush=True)
        timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.08, Env(), {"tiled": 2, "tiles_per_robot": 125, "dbg": dbg}, full=True)


if __name__ == "__main__" and "tilenocol" in sys.argv[1:]:
    for t in (0.04, 0.16):
        timing_cfg(engine.VOXCAD, 1, (20, 20, 20), t, Env(), {"tiled": 2, "tiles_per_robot": 125}, full=True, selfcol=False)
        timing_cfg(engine.VOXCAD, 1, (20, 20, 20), t, Env(), {"tiled": 2, "tiles_per_robot": 125}, full=True, selfcol=True)


if __name__ == "__main__" and "tileprof" in sys.argv[1:]:
    # the other BASELINE configs at their stated sizes, engine defaults (what scripts/profile_bench.sh traces)
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.16, Env(), {}, full=True)                                   # configs[4]
    timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.1, Env(), {})                                                 # configs[1]
    timing_cfg(engine.VOXCAD, 64, (10, 10, 10), 0.06, Env(), {})                                             # configs[2] as sharded over 8 GPUs
    timing_cfg(engine.VOXCAD, 64, (10, 10, 10), 0.06, Env(), {"tile_small": 1})                              # ... with the option under which they are tiled
    timing_cfg(engine.VOXCAD_LAND_WATER, 64, (8, 8, 8), 0.1, env_w, {}, per_voxel_phase=True)                # configs[3]


if __name__ == "__main__" and "ab" in sys.argv[1:]:      # A/B of the resident kernel on the bench population: time, or (phases) shares
    timing2(8, (10, 10, 10), 0.002, True, {"tiled": 0})          # (the first launch of a process loads the code object: ~3 ms)
    for _ in range(2):
        timing2(512, (10, 10, 10), 0.03, True, {"tiled": 0})
        timing2(512, (10, 10, 10), 0.03, False, {"tiled": 0})


if __name__ == "__main__" and "contactcheck" in sys.argv[1:]:
    # Resident kernel, developer library: (1) two identical runs of a colliding 10x10x10 robot must agree bit for bit at every checkpoint;
    # (2) with dbg = 8 every lane whose contact row is in LDS recomputes its contact sum through the rows in memory -- the library prints
    # how many lanes got different bits (must be 0), and whether the wavefronts that did not fit were the same ones in both runs.
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "vxa", "bench10_0.vxa")
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libvxhip_prof.so")
    for dbg in (8, 0):
        runs = []
        for _ in range(2):
            states = []
            with engine.Engine(engine.VOXCAD, 0) as eng:
                eng.set_option("tiled", 0)
                eng.set_option("dbg", dbg)
                eng.add_vxa_file(golden)
                for k in (1, 332, 2000, 100000):
                    eng.step(k)
                    states.append(eng.state(0))
                eng.clear()
            runs.append(states)
        same = [bool(np.array_equal(a, b)) for a, b in zip(*runs)]
        print("dbg", dbg, "two runs bit-identical at the checkpoints:", same, flush=True)


if __name__ == "__main__" and "l2pop" in sys.argv[1:]:      # the bench population for scripts/profile_l2.sh: `l2pop 1` / `l2pop 0` = self-collision on / off
    timing2(512, (10, 10, 10), 0.03, sys.argv[-1] == "1", {"tiled": 0})


if __name__ == "__main__" and "dense" in sys.argv[1:]:
    # the 1024-thread resident variant: 512 FULL 10x10x10 lattices (1000 voxels each), with and without self-collision, warm, twice
    for _ in range(2):
        timing_cfg(engine.VOXCAD, 512, (10, 10, 10), 0.03, Env(), {"tiled": 0}, full=True, selfcol=True)
        timing_cfg(engine.VOXCAD, 512, (10, 10, 10), 0.03, Env(), {"tiled": 0}, full=True, selfcol=False)


if __name__ == "__main__" and "launchcost" in sys.argv[1:]:
    # Fixed cost of a launch of the resident kernel: the bench population stepped 1000 steps past a 900-step pre-advance in launches of
    # L steps each; kernel time = launches x (fixed + L x per-step).  The driver's bench command times ONE launch of 20 steps.
    # VXH_LC_DBG=<n>: developer library with the what-if switch n (1: contact rows ignored, 4: no LDS copy of the rows), collisions only
    lc_dbg = int(os.environ.get("VXH_LC_DBG", "0"))
    if lc_dbg:
        engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libvxhip_prof.so")
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    for col in ((True,) if lc_dbg else (True, False)):
        sim = Sim(self_collisions_enabled=col, dt_frac=0.9, simulation_time=2.0, fitness_eval_init_time=0.4)
        paths = []
        pop = ([workloads.make_individual(i, workloads.full_material(10, 1 + i)) for i in range(512)] if "dense" in sys.argv[1:]      # (the 1024-thread variant)
               else workloads.population(512, (10, 10, 10)))
        for ind in pop:
            write_voxelyze_file(sim, Env(), ind, tmp, "c%d" % col)
            paths.append(os.path.join(tmp, "voxelyzeFiles", "c%d--id_%05i.vxa" % (col, ind.id)))
        rows = []
        for L in ((250, 20) if lc_dbg else (250, 100, 50, 20, 10, 5)):
            with engine.Engine(engine.VOXCAD, 0) as eng:
                eng.set_option("tiled", 0)
                eng.set_option("steps_per_launch", L)
                if lc_dbg:
                    eng.set_option("dbg", lc_dbg)
                eng.add_vxa_files(paths)
                eng.step(900)
                c0 = eng.counters()
                t0 = time.perf_counter()
                eng.step(1000)
                wall = time.perf_counter() - t0
                c1 = eng.counters()
                ks, nl = c1.kernel_seconds - c0.kernel_seconds, c1.launches - c0.launches
                rows.append((L, nl, ks, wall))
                if col:
                    reb = [eng.result(i).col_rebuilds for i in range(len(paths))]
                    print("        broad-phase runs per robot over the 1900 steps: mean %.1f, max %d" % (np.mean(reb), max(reb)), flush=True)
                print("col=%d  %3d steps per launch: %4d launches, kernel %.4f s = %.1f us per step, %.1f us per launch; wall %.4f s" % (col, L, nl, ks, 1e3 * ks, 1e6 * ks / nl, wall), flush=True)
        (La, na, ka, _), (Lb, nb, kb, _) = rows[0], rows[1 if lc_dbg else 3]
        fixed = (kb - ka) / (nb - na)
        print("col=%d  -> fixed cost per launch %.0f us, per step %.1f us (from the 250- and 20-step rows)" % (col, 1e6 * fixed, 1e6 * (ka - na * fixed) / 1000), flush=True)


if __name__ == "__main__" and "proepi" in sys.argv[1:]:
    # developer library: cycles a wavefront of the resident kernel spends before the step loop and after it, per launch, colliding vs not
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), os.environ.get("VXH_PROF_LIB", "libvxhip_prof.so"))
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    for col in (True, False):
        sim = Sim(self_collisions_enabled=col, dt_frac=0.9, simulation_time=2.0, fitness_eval_init_time=0.4)
        paths = []
        # VXH_LC_FULL=1: full lattices (1000 voxels: the 1024-thread variant) instead of the random robots (768-thread variant)
        full = os.environ.get("VXH_LC_FULL") == "1"
        inds = [workloads.make_individual(k, workloads.full_material(10, 1 + k)) for k in range(512)] if full else workloads.population(512, (10, 10, 10))
        for ind in inds:
            write_voxelyze_file(sim, Env(), ind, tmp, "p%d" % col)
            paths.append(os.path.join(tmp, "voxelyzeFiles", "p%d--id_%05i.vxa" % (col, ind.id)))
        with engine.Engine(engine.VOXCAD, 0) as eng:
            eng.set_option("tiled", 0)
            eng.set_option("steps_per_launch", 20)
            if os.environ.get("VXH_LC_DBG"):  # e.g. 16: every broad-phase run also executes the scan it replaced and compares the rows
                eng.set_option("dbg", int(os.environ["VXH_LC_DBG"]))
            eng.add_vxa_files(paths)
            eng.step(1000)                   # 50 launches of 20 steps
            print("self-collision %d: 50 launches x 512 robots; the two last numbers of a wave's line / (50 x 512) = cycles per launch" % col, flush=True)
            eng.clear()                      # prints the per-wave lines


if __name__ == "__main__" and "tilesweep" in sys.argv[1:]:
    # BASELINE configs[1] (64 random 6x6x6 robots): the resident kernel against the tiled kernel with 1, 2, 3, 4 tiles per robot
    for opts in ({"tiled": 0}, {"tiled": 2, "tiles_per_robot": 1}, {"tiled": 2, "tiles_per_robot": 2}, {"tiled": 2, "tiles_per_robot": 3}, {"tiled": 2, "tiles_per_robot": 4}, {}):
        timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.1, Env(), opts)
    for opts in ({"tiled": 0}, {"tiled": 2, "tiles_per_robot": 1}, {"tiled": 2, "tiles_per_robot": 2}, {"tiled": 2, "tiles_per_robot": 4}, {}):
        timing_cfg(engine.VOXCAD, 64, (8, 8, 8), 0.06, Env(), opts)


if __name__ == "__main__" and "tilepolicy" in sys.argv[1:]:
    # where the tiled kernel pays for robots the resident kernel could take: populations too small to fill the chip, by robot size
    for count, n, sim_time in ((64, 6, 0.1), (64, 7, 0.08), (64, 8, 0.06), (64, 9, 0.05), (64, 10, 0.04), (16, 10, 0.04), (128, 8, 0.06), (128, 10, 0.04)):
        for opts in ({"tiled": 0}, {}):
            timing_cfg(engine.VOXCAD, count, (n, n, n), sim_time, Env(), opts)


if __name__ == "__main__" and "launchlen" in sys.argv[1:]:
    # how long should a launch of the resident kernel be?  A whole evaluation (0.5 s simulated, 7806 steps) of the bench population with
    # launches of L steps: kernel time and wall time of vxh_run
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    sim = Sim(self_collisions_enabled=True, dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0.1)
    paths = []
    for ind in workloads.population(512, (10, 10, 10)):
        write_voxelyze_file(sim, Env(), ind, tmp, "L")
        paths.append(os.path.join(tmp, "voxelyzeFiles", "L--id_%05i.vxa" % ind.id))
    _warm_up(engine.VOXCAD, {"tiled": 0})
    for L in (256, 512, 1024, 2048, 4096, 8192, 256):
        with engine.Engine(engine.VOXCAD, 0) as eng:
            eng.set_option("steps_per_launch", L)
            eng.add_vxa_files(paths)
            t0 = time.perf_counter()
            eng.run()
            wall = time.perf_counter() - t0
            c = eng.counters()
            print("steps per launch %5d: %3d launches, kernel %.4f s = %.2f us per step, vxh_run wall %.4f s; all finished: %s" % (
                L, c.launches, c.kernel_seconds, 1e6 * c.kernel_seconds / c.max_steps, wall, all(eng.result(i).status == 1 for i in range(len(paths)))), flush=True)


if __name__ == "__main__" and "smallphases" in sys.argv[1:]:
    # round 3: where a step of the latency-bound configs goes (resident kernel; developer library), and their plain timings
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    if "prof" in sys.argv[1:]:
        engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libvxhip_prof.so")
        engine._lib = None
    ph = "prof" in sys.argv[1:]
    print("64 x 6^3 walkers", flush=True)
    timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.05, Env(), {}, phases=ph)
    print("64 x 8^3 swimmers", flush=True)
    timing_cfg(engine.VOXCAD_LAND_WATER, 64, (8, 8, 8), 0.05, env_w, {}, per_voxel_phase=True, phases=ph)
    print("64 x 8^3 walkers", flush=True)
    timing_cfg(engine.VOXCAD, 64, (8, 8, 8), 0.05, Env(), {}, phases=ph)
    print("64 x 10^3 walkers", flush=True)
    timing_cfg(engine.VOXCAD, 64, (10, 10, 10), 0.03, Env(), {}, phases=ph)


if __name__ == "__main__" and "lc2" in sys.argv[1:]:
    # round 3: the bench population (512 random 10^3 robots) with self-collision in launches of 250 and of 20 steps (fixed cost of a
    # launch from the two), and without self-collision (the step alone).  Run through scripts/ab_lib.py to compare two libraries.
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    for col in ((True,) if "colonly" in sys.argv[1:] else (True, False)):
        sim = Sim(self_collisions_enabled=col, dt_frac=0.9, simulation_time=2.0, fitness_eval_init_time=0.4)
        paths = []
        pop = ([workloads.make_individual(i, workloads.full_material(10, 1 + i)) for i in range(512)] if "dense" in sys.argv[1:]      # (the 1024-thread variant)
               else workloads.population(512, (10, 10, 10)))
        for ind in pop:
            write_voxelyze_file(sim, Env(), ind, tmp, "c%d" % col)
            paths.append(os.path.join(tmp, "voxelyzeFiles", "c%d--id_%05i.vxa" % (col, ind.id)))
        rows = []
        for L in ((250, 20) if col else (250,)):
            with engine.Engine(engine.VOXCAD, 0) as eng:
                eng.set_option("tiled", 0)
                eng.set_option("steps_per_launch", L)
                eng.add_vxa_files(paths)
                eng.step(900)
                c0 = eng.counters()
                eng.step(1000)
                c1 = eng.counters()
                ks, nl = c1.kernel_seconds - c0.kernel_seconds, c1.launches - c0.launches
                rows.append((L, nl, ks))
                print("%s col=%d  %3d steps per launch: %4d launches, %.2f us per step" % (os.path.basename(engine.LIB_PATH), col, L, nl, 1e3 * ks), flush=True)
        if col:
            (La, na, ka), (Lb, nb, kb) = rows
            fixed = (kb - ka) / (nb - na)
            print("%s col=1  -> fixed cost per launch %.0f us, per step without it %.2f us" % (os.path.basename(engine.LIB_PATH), 1e6 * fixed, 1e6 * (ka - na * fixed) / 1000), flush=True)


if __name__ == "__main__" and "drift" in sys.argv[1:]:
    # round 3: error growth of a long run against the oracle, per kernel path (the shipped land_water examples: 26 k and 68 k steps)
    from oracle import vxoracle as vo
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "vxa")
    for name, variant in (("lw_hexapus", 1), ("lw_quadruped_land", 1), ("example_1", 0)):
        path = os.path.join(golden, name + ".vxa")
        model = vo.parse_vxa(path, variant)
        lat = model["lattice_dim"]
        checkpoints = (100, 1000, 4000, 16000, 25000, 60000)
        sim = vo.OracleSim(model)
        want = []
        for upto in checkpoints:
            sim.step(upto - sim.info().steps)
            want.append((sim.info().steps, sim.state()))
        for label, opts in (("wide", {}), ("resident", {"wide": 0, "tiled": 0}), ("streaming", {"fused": 0, "tiled": 0})):
            with engine.Engine(variant, 0) as eng:
                for k, v in opts.items():
                    eng.set_option(k, v)
                eng.add_vxa_file(path)
                done, line = 0, []
                for (steps, st) in want:
                    eng.step(steps - done)
                    done = steps
                    got = eng.state(0)
                    line.append("%d: %.1e" % (steps, np.abs(got[:, :3] - st[:, :3]).max() / lat))
                print("%-18s %-10s max position error in voxels at step  %s" % (name, label, "  ".join(line)), flush=True)


if __name__ == "__main__" and "drift2" in sys.argv[1:]:
    # round 3: which ingredient of the shipped land_water example makes the engine leave the oracle by 1e-10 voxel within 100 steps
    import re
    from oracle import vxoracle as vo
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "vxa")
    base = open(os.path.join(golden, "lw_hexapus.vxa")).read()
    swim = open(os.path.join(golden, "lw_swim6.vxa")).read()

    def sub(text, tag, val):
        out, n = re.subn(r"<%s>[^<]*</%s>" % (tag, tag), "<%s>%s</%s>" % (tag, val, tag), text)
        assert n >= 1, tag
        return out
    variants = [("hexapus as shipped", base),
                ("hexapus, TempPeriod 0.25", sub(base, "TempPeriod", "0.25")),
                ("hexapus, DtFrac 0.9", sub(base, "DtFrac", "0.9")),
                ("hexapus, Lattice_Dim 0.01", sub(base, "Lattice_Dim", "0.01")),
                ("hexapus, no actuation", sub(base, "TempEnabled", "0")),
                ("hexapus, no fluid", sub(base, "FluidEnvironment", "0")),
                ("swim6 as generated", swim),
                ("swim6, Lattice_Dim 0.05", sub(swim, "Lattice_Dim", "0.05")),
                ("swim6, TempPeriod 0.263662540539", sub(swim, "TempPeriod", "0.263662540539")),
                ("swim6, DtFrac 0.7", sub(swim, "DtFrac", "0.7"))]
    tmp = tempfile.mkdtemp()
    for label, text in variants:
        path = os.path.join(tmp, "v.vxa")
        open(path, "w").write(text)
        model = vo.parse_vxa(path, 1)
        lat = model["lattice_dim"]
        sim = vo.OracleSim(model)
        with engine.Engine(1, 0) as eng:
            eng.add_vxa_file(path)
            done, line = 0, []
            for upto in (1, 2, 3, 5, 10, 100, 1000):
                eng.step(upto - done); done = upto
                sim.step(upto - sim.info().steps)
                got, want = eng.state(0), sim.state()
                line.append("%d: %.1e/%.0e/%.0e" % (upto, np.abs(got[:, :3] - want[:, :3]).max() / lat, np.abs(got[:, 3:7] - want[:, 3:7]).max(), np.abs(got[:, 7] - want[:, 7]).max() / lat))
            print("%-36s pos/quat/scale error at step  %s" % (label, "  ".join(line)), flush=True)


if __name__ == "__main__" and "drift3" in sys.argv[1:]:
    # round 3: step by step through the start of the shipped land_water example: where does the first non-rounding deviation appear
    from oracle import vxoracle as vo
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "vxa")
    path = os.path.join(golden, "lw_hexapus.vxa")
    model = vo.parse_vxa(path, 1)
    lat = model["lattice_dim"]
    sim = vo.OracleSim(model)
    with engine.Engine(1, 0) as eng:
        eng.add_vxa_file(path)
        for step in range(1, 16):
            eng.step(1)
            sim.step(1)
            got, want = eng.state(0), sim.state()
            d = np.abs(got - want)
            worst = int(d[:, :3].max(axis=1).argmax())
            print("step %2d: pos %.1e (voxel %d, z %.4f)  quat %.1e  scale %.1e  vel %.1e (rel %.1e)  angvel %.1e (rel %.1e)   oracle: max |vel| %.2e max |angvel| %.2e  min quat w - 1: %.2e" % (
                step, d[:, :3].max() / lat, worst, want[worst, 2] / lat, d[:, 3:7].max(), d[:, 7].max() / lat, d[:, 8:11].max(), d[:, 8:11].max() / max(1e-300, np.abs(want[:, 8:11]).max()),
                d[:, 11:14].max(), d[:, 11:14].max() / max(1e-300, np.abs(want[:, 11:14]).max()), np.abs(want[:, 8:11]).max(), np.abs(want[:, 11:14]).max(), (want[:, 3] - 1).min()), flush=True)


if __name__ == "__main__" and "drift6" in sys.argv[1:]:
    # round 3: do engine and oracle ever disagree on a bond's small-/large-angle mode?  (count of large-angle bonds after every step)
    from oracle import vxoracle as vo
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "vxa")
    for name, variant, nsteps in (("lw_hexapus", 1, 3000), ("lw_quadruped_land", 1, 3000), ("lw_swim6", 1, 1500)):
        path = os.path.join(golden, name + ".vxa")
        model = vo.parse_vxa(path, variant)
        sim = vo.OracleSim(model)
        bad, flips, last = [], 0, 0
        with engine.Engine(variant, 0) as eng:
            eng.set_option("steps_per_launch", 1)
            eng.add_vxa_file(path)
            for step in range(1, nsteps + 1):
                eng.step(1); sim.step(1)
                info = sim.info()
                want = info.nbond - info.n_small_angle
                got = eng.bond_modes()[0]
                flips += abs(want - last); last = want
                if got != want:
                    bad.append((step, got, want))
        print("%s: %d steps, net mode changes %d, steps where the counts of large-angle bonds differ: %d %s" % (name, nsteps, flips, len(bad), bad[:10]), flush=True)


if __name__ == "__main__" and "drift7" in sys.argv[1:]:
    # round 3: what ONE step of the engine and of the oracle differ by, from the same state (the oracle is put on the engine's state
    # before every step: oracle instrument vxo_set_state): size, which voxels, and whether the differences have a direction
    from oracle import vxoracle as vo
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "vxa")
    for name, variant, nsteps in (("lw_hexapus", 1, 4000), ("bench10_0", 0, 2000)):
        path = os.path.join(golden, name + ".vxa")
        model = vo.parse_vxa(path, variant)
        lat = model["lattice_dim"]
        sim = vo.OracleSim(model, half_angle="ha" in sys.argv[1:])      # ("ha": the oracle build with the engine's half-angle form of FromAngleToPosX)
        with engine.Engine(variant, 0) as eng:
            eng.set_option("steps_per_launch", 1)
            eng.add_vxa_file(path)
            prev = sim.state()
            cm_bias = np.zeros(3)
            big = []
            for step in range(1, nsteps + 1):
                eng.step(1)
                E = eng.state(0)
                sim.set_state(prev)
                sim.step(1)
                O = sim.state()
                d = E - O
                dpos = np.abs(d[:, :3]).max() / lat
                cm_bias += d[:, :3].mean(axis=0) / lat
                vmax = max(1e-300, np.abs(O[:, 8:11]).max())
                wmax = max(1e-300, np.abs(O[:, 11:14]).max())
                dv, dw = np.abs(d[:, 8:11]).max() / vmax, np.abs(d[:, 11:14]).max() / wmax
                if step <= 12 or step % 250 == 0:
                    wv = int(np.abs(d[:, 11:14]).max(axis=1).argmax())
                    print("%s step %5d: one-step difference pos %.1e voxel, vel %.1e, angvel %.1e (rel. to the largest; worst voxel %d)   sum of the CoM differences so far %s voxel" % (
                        name, step, dpos, dv, dw, wv, np.array2string(cm_bias, precision=2)), flush=True)
                big.append((dw, dv, dpos))
                prev = E
            a = np.array(big)
            print("%s: over %d steps: one-step angvel difference median %.1e max %.1e; vel median %.1e max %.1e; pos median %.1e max %.1e" % (
                name, nsteps, np.median(a[:, 0]), a[:, 0].max(), np.median(a[:, 1]), a[:, 1].max(), np.median(a[:, 2]), a[:, 2].max()), flush=True)


if __name__ == "__main__" and "twotiles" in sys.argv[1:]:
    # round 3: the wide kernel with and without its second pose tile (option wide_two_tiles): time, and every voxel of every robot bit for bit
    KEEP_STATES = True
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    for label, args, kw in (("64 x 6^3 walkers", (engine.VOXCAD, 64, (6, 6, 6), 0.1, Env()), {}),
                            ("64 x 8^3 walkers", (engine.VOXCAD, 64, (8, 8, 8), 0.05, Env()), {}),
                            ("64 x 8^3 swimmers", (engine.VOXCAD_LAND_WATER, 64, (8, 8, 8), 0.05, env_w), {"per_voxel_phase": True}),
                            ("512 x 8^3 walkers", (engine.VOXCAD, 512, (8, 8, 8), 0.05, Env()), {}),
                            ("64 x 5^3 stiffness layers", (engine.VOXCAD, 64, (5, 5, 5), 0.05, Env()), {"stiffness": True})):
        print(label, flush=True)
        got = []
        for two in (1, 0, 1):
            timing_cfg(*args, {"wide_two_tiles": two}, **kw)
            got.append(LAST_STATES)
        same10 = all(np.array_equal(a, b) for a, b in zip(got[0], got[1]))
        same11 = all(np.array_equal(a, b) for a, b in zip(got[0], got[2]))
        print("   two tiles against one: %s;  two runs with two tiles: %s" % ("bit-identical" if same10 else "DIFFERENT", "bit-identical" if same11 else "DIFFERENT"), flush=True)


if __name__ == "__main__" and sys.argv[1:2] and sys.argv[1] == "cfg4tiles":    # configs[4] by number of tiles
    for k in (0, 27, 64, 125, 216, 250):
        timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.13, Env(), {"tiled": 2, "tiles_per_robot": k} if k else {}, full=True)


if __name__ == "__main__" and sys.argv[1:2] and sys.argv[1] == "cfg4l":      # configs[4] over ~2000 steps, like the bench line's other_configs
    timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.13, Env(), {}, full=True)


if __name__ == "__main__" and sys.argv[1:2] and sys.argv[1] in ("cfg1", "cfg3", "cfg4"):
    # one BASELINE config each, for the counter passes of scripts/profile_bench.sh
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    if sys.argv[1] == "cfg1":
        timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.1, Env(), {})
    elif sys.argv[1] == "cfg3":
        timing_cfg(engine.VOXCAD_LAND_WATER, 64, (8, 8, 8), 0.1, env_w, {}, per_voxel_phase=True)
    else:
        timing_cfg(engine.VOXCAD, 1, (20, 20, 20), 0.02, Env(), {}, full=True)


if __name__ == "__main__" and "satcheck" in sys.argv[1:]:
    # round 3: the wide kernel against the resident one on populations that fill the chip (is the latency layout slower there?)
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    for rep in range(2):
        for opts in ({}, {"wide": 0}):
            timing_cfg(engine.VOXCAD, 2048, (6, 6, 6), 0.05, Env(), opts)
            timing_cfg(engine.VOXCAD, 512, (8, 8, 8), 0.05, Env(), opts)
            timing_cfg(engine.VOXCAD_LAND_WATER, 512, (8, 8, 8), 0.05, env_w, opts, per_voxel_phase=True)


if __name__ == "__main__" and "cli30" in sys.argv[1:]:
    # Round 4: a generation of evosoro's basic.py shape (pop 30, 6x6x6, 0.5 s simulated... here the bench robots' defaults) through the
    # COMMAND LINE the way evosoro/tools/evaluation.py:59-90 drives it -- one `voxelyze -f` process per robot, all at once -- wall clock
    # from the first Popen to the last exit: the reference binary on the host cores, our binary with every process stepping its own
    # robot (VXH_BROKER=0), and our binary through the broker (first generation = incl. starting the broker and the HIP runtime).
    import subprocess
    import json as _json
    pop = int(os.environ.get("VXH_CLI_POP", "30"))
    out = {"pop": pop, "shape": [6, 6, 6]}
    ref = os.path.join(REPO, "oracle", "_ref", "voxelyze_ref")

    def generation(binary, env_extra, tag, rep):
        work = tempfile.mkdtemp(prefix="cli30_%s%d_" % (tag, rep))
        for d in ("voxelyzeFiles", "fitnessFiles", "tempFiles"):
            os.makedirs(os.path.join(work, "run", d))
        sim = Sim(self_collisions_enabled=True, dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0.1)
        here = os.getcwd()
        os.chdir(work)
        files = []
        for i in range(pop):
            ind = workloads.random_robot(i, (6, 6, 6), 1000 * rep + i)
            write_voxelyze_file(sim, Env(), ind, "run", "g")
            files.append(os.path.join("run", "voxelyzeFiles", "g--id_%05i.vxa" % i))
        os.chdir(here)
        env = dict(os.environ, **env_extra)
        t0 = time.time()
        procs = [subprocess.Popen([binary, "-f", f], cwd=work, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for f in files]
        for p in procs:
            p.wait()
        wall = time.time() - t0
        done = len([f for f in os.listdir(os.path.join(work, "run", "fitnessFiles")) if f.endswith(".xml")])
        return wall, done, [p.returncode for p in procs].count(1)

    sock = os.path.join(tempfile.mkdtemp(), "b.sock")
    for tag, binary, env_extra, reps in (("reference_cpu", ref, {}, 2), ("direct", engine.CLI_PATH, {"VXH_BROKER": "0"}, 2),
                                         ("broker", engine.CLI_PATH, {"VXH_BROKER_SOCKET": sock, "VXH_BROKER_IDLE_S": "20"}, 3)):
        if not os.path.exists(binary):
            continue
        rows = []
        for rep in range(reps):
            wall, done, ok = generation(binary, env_extra, tag, rep)
            rows.append({"wall_s": wall, "result_files": done, "exit_1": ok})
            print("%-14s generation %d: %.3f s, %d result files, %d processes returned 1" % (tag, rep, wall, done, ok), flush=True)
        out[tag] = rows
    subprocess.run([engine.CLI_PATH, "--broker-quit"], env=dict(os.environ, VXH_BROKER_SOCKET=sock), stdout=subprocess.DEVNULL)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "r04_cli_generation_pop%d.json" % pop), "w") as f:
        _json.dump(out, f, indent=1)


if __name__ == "__main__" and "statehash" in sys.argv[1:]:
    # a digest of every voxel's state of 48 bench robots (10^3, self-collision) + 16 dense 9^3 + 8 dense 10^3 + 32 small ones after 700 steps: two builds of the
    # library that claim the same arithmetic in another schedule must print the same line (scripts/ab_lib.py <lib> statehash)
    import hashlib
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    sim = Sim(self_collisions_enabled=True, dt_frac=0.9, simulation_time=0.2, fitness_eval_init_time=0.01)
    with engine.Engine(engine.VOXCAD, 0) as eng:
        eng.set_option("tiled", 0)
        inds = ([workloads.random_robot(i, (10, 10, 10), i) for i in range(48)] + [workloads.make_individual(100 + i, workloads.full_material(9, 1 + i)) for i in range(16)]
                + [workloads.make_individual(200 + i, workloads.full_material(10, 1 + i)) for i in range(8)]      # (the 1024-thread variant)
                + [workloads.random_robot(300 + i, (6, 6, 6), 300 + i) for i in range(16)]                             # (the wide kernel, few bonds)
                + [workloads.random_robot(400 + i, (8, 8, 8), 400 + i) for i in range(16)])                            # (the wide kernel, most lanes busy)
        for ind in inds:
            write_voxelyze_file(sim, Env(), ind, tmp, "h")
            eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "h--id_%05i.vxa" % ind.id))
        eng.step(300); eng.step(400)
        h = hashlib.sha256()
        for i in range(len(inds)):
            h.update(np.ascontiguousarray(eng.state(i)).tobytes())
        print("%s statehash %s (%d robots, 700 steps, kernel of most voxel-steps %d)" % (os.path.basename(engine.LIB_PATH), h.hexdigest()[:24], len(inds), eng.counters().dominant_block), flush=True)


if __name__ == "__main__" and "mixedlaunch" in sys.argv[1:]:
    # the mixed generation of bench.py (three size classes, two step counts) run to completion with launches of different lengths:
    # how much of its time is CUs idling behind the slowest robot of a launch group?
    sys.path.insert(0, REPO)
    import torch
    torch.cuda.init()          # (torch's HIP runtime first, as in bench.py: initialised after the engine's it finds no device)
    import bench
    for L in (128, 256, 512, 1024, 4096):
        r = bench.mixed_generation(engine, 0, options={"steps_per_launch": L})
        print("steps_per_launch %5d: %.3e voxel-steps/s over the GPU's time (%.1f ms, %d launches), wall %.1f ms" % (
            L, r["value"], 1e3 * r["gpu_seconds"], r["launches"], 1e3 * r["wall_seconds"]), flush=True)


if __name__ == "__main__" and "widephases" in sys.argv[1:]:
    # developer timers of the wide kernel on BASELINE configs[1] and [3] (library built by `make -C evosoro_amd/csrc prof`)
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libvxhip_prof.so")
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    print("64 x 6^3", flush=True)
    timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.1, Env(), {}, phases=True)
    print("64 x 8^3 swimmers", flush=True)
    timing_cfg(engine.VOXCAD_LAND_WATER, 64, (8, 8, 8), 0.1, env_w, {}, per_voxel_phase=True, phases=True)


if __name__ == "__main__" and "benchsides" in sys.argv[1:]:
    # bench.py's own side configurations (the dense 10^3 population, the mixed generation), for A/B runs of two libraries
    sys.path.insert(0, REPO)
    import torch
    torch.cuda.init()
    import bench
    for _ in range(2):
        r = bench.side_config(engine, "dense", engine.VOXCAD, 512, (10, 10, 10), Env(), 0, 512, full=True, init_time=0.01)
        print("%s dense 10^3: %.2f us per step" % (os.path.basename(engine.LIB_PATH), r["us_per_step"]), flush=True)
        r = bench.mixed_generation(engine, 0)
        print("%s mixed generation: %.3e voxel-steps/s" % (os.path.basename(engine.LIB_PATH), r["value"]), flush=True)


if __name__ == "__main__" and "smallcfgs" in sys.argv[1:]:
    # the small populations of BASELINE configs[1] and [3] (wide kernel) and a saturated one, for A/B runs of two libraries
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    for _ in range(2):
        timing_cfg(engine.VOXCAD, 64, (6, 6, 6), 0.1, Env(), {})
        timing_cfg(engine.VOXCAD_LAND_WATER, 64, (8, 8, 8), 0.1, env_w, {}, per_voxel_phase=True)
        timing_cfg(engine.VOXCAD, 512, (8, 8, 8), 0.04, Env(), {})


def _pair_population(kind, count):
    """robots for the pair-kernel A/B: 'dense' = full 10^3 lattices, 'p15' = random 10^3 with P(empty) 0.15 (769-1000 voxels mostly), 'bench' = the bench population"""
    if kind == "dense":
        return [workloads.make_individual(i, workloads.full_material(10, 1 + i)) for i in range(count)]
    if kind == "p15":
        return [workloads.random_robot(i, (10, 10, 10), 5000 + i, p_empty=0.15) for i in range(count)]
    return workloads.population(count, (10, 10, 10))


if __name__ == "__main__" and "pairab" in sys.argv[1:]:
    # round 5: k_robot_pair (512 threads, two voxels per lane) against k_robot_steps<1024> / <768>, same library, option `pair`
    #   pairab check   states after 40 / 400 / 1200 steps of 24 robots per population kind with pair = 0 and pair = 1 (2 for the bench kind)
    #   pairab time    us per population step of 512 robots, launches of 250 steps past a 600-step pre-advance, each option twice
    sel = 1 if "sel" in sys.argv[1:] else 0
    if "check" in sys.argv[1:]:
        for kind in ("dense", "p15", "bench"):
            tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
            sim = Sim(self_collisions_enabled=True, dt_frac=0.9, simulation_time=2.0, fitness_eval_init_time=0.02)
            paths = []
            for ind in _pair_population(kind, 24):
                write_voxelyze_file(sim, Env(), ind, tmp, "p"); paths.append(os.path.join(tmp, "voxelyzeFiles", "p--id_%05i.vxa" % ind.id))
            runs = {}
            for pair in (0, 2 if kind == "bench" else 1):
                with engine.Engine(engine.VOXCAD, 0) as eng:
                    eng.set_option("tiled", 0); eng.set_option("pair", pair); eng.set_option("pair_sel", sel)
                    eng.add_vxa_files(paths)
                    out, done = [], 0
                    for upto in (40, 400, 1200):
                        eng.step(upto - done); done = upto
                        out.append([eng.state(i) for i in range(len(paths))])
                    runs[pair] = (out, eng.counters().dominant_block, [eng.result(i).col_rebuilds for i in range(len(paths))])
            (a, ka, ra), (b, kb, rb) = runs[0], runs[2 if kind == "bench" else 1]
            for k, upto in enumerate((40, 400, 1200)):
                err = max(np.abs(x[:, :3] - y[:, :3]).max() for x, y in zip(a[k], b[k])) / 0.01
                same = sum(bool(np.array_equal(x, y)) for x, y in zip(a[k], b[k]))
                print("pairab check %-5s step %4d: kernels %d vs %d, max |dpos| %.3e voxel, %d of %d robots bit-identical, rebuilds equal %s" % (
                    kind, upto, ka, kb, err, same, len(paths), ra == rb), flush=True)
    if "time" in sys.argv[1:]:
        kinds = [k for k in ("dense", "p15", "bench") if k in sys.argv[1:]] or ["dense", "p15", "bench"]
        for kind in kinds:
            tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
            sim = Sim(self_collisions_enabled=True, dt_frac=0.9, simulation_time=2.0, fitness_eval_init_time=0.3)
            paths, nvox = [], 0
            for ind in _pair_population(kind, 512):
                write_voxelyze_file(sim, Env(), ind, tmp, "p"); paths.append(os.path.join(tmp, "voxelyzeFiles", "p--id_%05i.vxa" % ind.id))
            for rep in range(2):
                for pair in ((0, 2) if kind == "bench" else (0, 1)):
                    with engine.Engine(engine.VOXCAD, 0) as eng:
                        eng.set_option("tiled", 0); eng.set_option("pair", pair); eng.set_option("pair_sel", sel); eng.set_option("steps_per_launch", 250)
                        eng.add_vxa_files(paths)
                        eng.step(600)
                        c0 = eng.counters(); eng.step(1000); c1 = eng.counters()
                        print("pairab time %-5s pair=%d sel=%d: %.2f us per population step (kernel %d, %.3e voxel-steps/s)" % (
                            kind, pair, sel, 1e3 * (c1.kernel_seconds - c0.kernel_seconds), c1.dominant_block,
                            (c1.voxel_steps - c0.voxel_steps) / (c1.kernel_seconds - c0.kernel_seconds)), flush=True)


if __name__ == "__main__" and "onestep" in sys.argv[1:]:
    # round 5: which voxel differs, and by how much, on the steps where ONE step of engine and oracle from the same state differ by more than
    # 5e-14 voxel (tests/test_gpu_parity.py test_one_step_from_the_same_state): onestep <golden case> <variant 0/1> [steps]
    name, variant = sys.argv[2], int(sys.argv[3])
    nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
    path = os.path.join(G, "vxa", name + ".vxa")
    model = vo.parse_vxa(path, variant)
    lat = model["lattice_dim"]
    sim = vo.OracleSim(model)
    twin = vo.OracleSim(model)          # the oracle once more, its inputs moved by one ulp before every step: the reference algorithm's OWN one-step noise
    with engine.Engine(variant, 0) as eng:
        eng.add_vxa_file(path)
        prev = sim.state()
        for step in range(1, nsteps + 1):
            eng.step(1)
            got = eng.state(0)
            sim.set_state(prev)
            sim.step(1)
            want = sim.state()
            twin.set_state(prev)
            twin.step_jittered(1, seed=step)
            own = np.abs(twin.state() - want)
            d = np.abs(got - want)
            dp = d[:, :3].max() / lat
            if dp > 5e-14 or own[:, :3].max() / lat > 5e-14:
                print("step %d: engine - oracle %.3e voxel; oracle(one-ulp jitter) - oracle %.3e voxel (at voxel %d)" % (
                    step, dp, own[:, :3].max() / lat, int(np.argmax(own[:, :3].max(axis=1)))), flush=True)
            modes = sim.bond_modes()
            large_e, total_e = eng.bond_modes()
            if step <= 40:
                print("step %d: large-angle bonds engine %d oracle %d of %d%s" % (step, large_e, int((modes == 0).sum()), total_e,
                      "" if step == 1 else "  oracle flips this step: %s" % np.nonzero(modes != last_modes)[0].tolist()), flush=True)
            last_modes = modes.copy()
            if dp > 5e-14:
                v = int(np.argmax(d[:, :3].max(axis=1)))
                print("step %d: dpos %.3e voxel at voxel %d (axis %d); its z/lat %.6f, scale/lat %.6f, |vel xy| %.3e, vel z %.3e; quat diff %.2e; voxels over the bar: %d" % (
                    step, dp, v, int(np.argmax(d[v, :3])), prev[v, 2] / lat, prev[v, 7] / lat, float(np.hypot(prev[v, 8], prev[v, 9])), prev[v, 10],
                    d[v, 3:7].max(), int((d[:, :3].max(axis=1) / lat > 5e-14).sum())), flush=True)
                print("   got  pos %s vel %s\n   want pos %s vel %s" % (got[v, :3], got[v, 8:11], want[v, :3], want[v, 8:11]), flush=True)
            prev = got


if __name__ == "__main__" and "swimhash" in sys.argv[1:]:
    # a digest of every voxel's state of 24 swimmers of 8^3 and 8 of 6^3 (wide MESH kernel) + 4 of 10^3 (768-thread MESH variant) after 500 steps: two builds that
    # claim the same arithmetic must print the same line (scripts/ab_lib.py <lib> swimhash)
    import hashlib
    from collections import OrderedDict
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    sim = Sim(self_collisions_enabled=True, dt_frac=0.9, simulation_time=0.3, fitness_eval_init_time=0.005)
    with engine.Engine(engine.VOXCAD_LAND_WATER, 0) as eng:
        n = 0
        for shape, cnt in (((8, 8, 8), 24), ((6, 6, 6), 8), ((10, 10, 10), 4)):
            for i in range(cnt):
                ind = workloads.random_robot(n, shape, 900 + n, phase_offset=True)
                write_voxelyze_file(sim, env_w, ind, tmp, "sw")
                eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "sw--id_%05i.vxa" % n))
                n += 1
        eng.step(200); eng.step(300)
        h = hashlib.sha256()
        for i in range(n):
            h.update(np.ascontiguousarray(eng.state(i)).tobytes())
        print("%s swimhash %s (%d swimmers, 500 steps, kernel of most voxel-steps %d)" % (os.path.basename(engine.LIB_PATH), h.hexdigest()[:24], n, eng.counters().dominant_block), flush=True)


def _window(variant, mats, env, opts, pre, steps, extra=None, tag="", label="tileab"):
    """us per step of a population over `steps` steps past a pre-advance of `pre` (kernel time by the engine's events), for same-box A/B"""
    tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    sim = Sim(dt_frac=0.9, simulation_time=1.0, fitness_eval_init_time=0.005, self_collisions_enabled=True)
    with engine.Engine(variant, 0) as eng:
        for k, v in opts.items():
            eng.set_option(k, v)
        for i, m in enumerate(mats):
            ind = workloads.make_individual(i, m, extra(i) if extra else None)
            write_voxelyze_file(sim, env, ind, tmp, "w")
            eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "w--id_%05i.vxa" % i))
        eng.step(pre)
        c0 = eng.counters(); eng.step(steps); c1 = eng.counters()
        print("%s %-34s %s: %.3f us per step (kernel %d)" % (label, tag, opts, 1e6 * (c1.kernel_seconds - c0.kernel_seconds) / steps, c1.dominant_block), flush=True)
    shutil.rmtree(tmp, ignore_errors=True)


def _water():
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    return env_w


def _phase_layer(shape, seed0):
    from collections import OrderedDict
    return lambda i: OrderedDict([("<PhaseOffset>", np.round(np.random.RandomState(seed0 + i).uniform(-1, 1, size=shape), 3))])


if __name__ == "__main__" and "tileab" in sys.argv[1:]:
    # round 6: the tiled kernel's workloads in one process, for same-box A/B of two libraries (scripts/ab_lib.py <lib> tileab; scripts/r6_ab.sh):
    # configs[4], one 64-robot shard of configs[2] with and without tile_small, an 11^3 swimmer; us per step over a window past a pre-advance
    shard = [workloads.random_material((10, 10, 10), i) for i in range(64)]
    for rep in range(2):
        _window(engine.VOXCAD, [workloads.full_material(20, 1)], Env(), {}, 300, 2000, tag="1 x 20^3 (configs[4])")
        _window(engine.VOXCAD, shard, Env(), {"tile_small": 1}, 300, 600, tag="64 x 10^3 random")
        _window(engine.VOXCAD, shard, Env(), {}, 300, 600, tag="64 x 10^3 random")
        _window(engine.VOXCAD_LAND_WATER, [workloads.full_material(11, 1)], _water(), {}, 200, 600, tag="1 x 11^3 swimmer", extra=_phase_layer((11, 11, 11), 70))


if __name__ == "__main__" and "kab" in sys.argv[1:]:
    # round 6: every kernel family's main workload in one process, for same-box A/B of a change to shared device code (bond_compute, voxel_update):
    # headline (512 random 10^3, k_robot_steps<768>), dense 10^3 (<1024>), configs[1] and [3] (k_robot_wide), configs[4] (k_tile_steps)
    bench = [workloads.random_material((10, 10, 10), i) for i in range(512)]
    for rep in range(2):
        _window(engine.VOXCAD, bench, Env(), {}, 900, 1000, tag="512 x 10^3 random (headline)", label="kab")
        _window(engine.VOXCAD, [workloads.full_material(10, 1 + i) for i in range(512)], Env(), {}, 300, 500, tag="512 x 10^3 dense", label="kab")
        _window(engine.VOXCAD, [workloads.random_material((6, 6, 6), i) for i in range(64)], Env(), {}, 300, 1500, tag="64 x 6^3 (configs[1])", label="kab")
        _window(engine.VOXCAD_LAND_WATER, [workloads.random_material((8, 8, 8), i) for i in range(64)], _water(), {}, 300, 1000, tag="64 x 8^3 swimmers (configs[3])",
                extra=_phase_layer((8, 8, 8), 0), label="kab")
        _window(engine.VOXCAD, [workloads.full_material(20, 1)], Env(), {}, 300, 2000, tag="1 x 20^3 (configs[4])", label="kab")


if __name__ == "__main__" and "bigswim" in sys.argv[1:]:
    # round 5: swimmers above 1024 voxels (full 11^3 and 14^3 lattices in a fluid) on the tiled kernel (fluid tiles) against the streaming kernels
    from collections import OrderedDict
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    for n, count in ((11, 1), (14, 1), (11, 16)):
        for opts in ({}, {"tiled": 0}):
            tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
            sim = Sim(dt_frac=0.9, simulation_time=0.06, fitness_eval_init_time=0.005, self_collisions_enabled=True)
            with engine.Engine(engine.VOXCAD_LAND_WATER, 0) as eng:
                for k, v in opts.items():
                    eng.set_option(k, v)
                for i in range(count):
                    ind = workloads.make_individual(i, workloads.full_material(n, 1 + i), OrderedDict([("<PhaseOffset>", np.round(np.random.RandomState(70 + i).uniform(-1, 1, size=(n, n, n)), 3))]))
                    write_voxelyze_file(sim, env_w, ind, tmp, "b")
                    eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "b--id_%05i.vxa" % i))
                eng.step(200)
                c0 = eng.counters(); eng.step(600); c1 = eng.counters()
                print("bigswim %d x %d^3 swimmers %s: %.2f us per step (kernel %d)" % (count, n, opts, 1e6 * (c1.kernel_seconds - c0.kernel_seconds) / 600, c1.dominant_block), flush=True)
    # the same lattices as land_water robots ON LAND (tiles with the strain tile, no drag mesh), for what the drag costs a tile
    for n in (11, 14):
        tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
        sim = Sim(dt_frac=0.9, simulation_time=0.06, fitness_eval_init_time=0.005, self_collisions_enabled=True)
        with engine.Engine(engine.VOXCAD_LAND_WATER, 0) as eng:
            ind = workloads.make_individual(0, workloads.full_material(n, 1), OrderedDict([("<PhaseOffset>", np.round(np.random.RandomState(70).uniform(-1, 1, size=(n, n, n)), 3))]))
            write_voxelyze_file(sim, Env(), ind, tmp, "b")
            eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "b--id_%05i.vxa" % 0))
            eng.step(200)
            c0 = eng.counters(); eng.step(600); c1 = eng.counters()
            print("bigswim 1 x %d^3 ON LAND: %.2f us per step (kernel %d)" % (n, 1e6 * (c1.kernel_seconds - c0.kernel_seconds) / 600, c1.dominant_block), flush=True)


if __name__ == "__main__" and "swimtimeline" in sys.argv[1:]:
    # per-tile timeline (VXH_PROF_TILES=1, developer library) of a full 11^3 land_water lattice in a fluid and on land
    from collections import OrderedDict
    engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), "libvxhip_prof.so")
    for fluid in (1, 0):
        env_w = Env()
        if fluid:
            env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
            env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
        tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
        sim = Sim(dt_frac=0.9, simulation_time=0.06, fitness_eval_init_time=0.005, self_collisions_enabled=True)
        with engine.Engine(engine.VOXCAD_LAND_WATER, 0) as eng:
            ind = workloads.make_individual(0, workloads.full_material(11, 1), OrderedDict([("<PhaseOffset>", np.round(np.random.RandomState(70).uniform(-1, 1, size=(11, 11, 11)), 3))]))
            write_voxelyze_file(sim, env_w, ind, tmp, "b")
            eng.add_vxa_file(os.path.join(tmp, "voxelyzeFiles", "b--id_%05i.vxa" % 0))
            eng.step(600)
            print("swimtimeline fluid=%d" % fluid, flush=True)
            eng.clear()


if __name__ == "__main__" and "cfg3tiles" in sys.argv[1:]:
    # BASELINE configs[3] (64 random 8^3 swimmers) on the tiled kernel by tiles per robot (fluid tiles, round 5) against the wide kernel
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    for opts in ({}, {"tiled": 2, "tiles_per_robot": 2}, {"tiled": 2, "tiles_per_robot": 3}, {"tiled": 2, "tiles_per_robot": 4}, {"tiled": 2, "tiles_per_robot": 6}):
        timing_cfg(engine.VOXCAD_LAND_WATER, 64, (8, 8, 8), 0.1, env_w, opts, per_voxel_phase=True)


if __name__ == "__main__" and "tilecap" in sys.argv[1:]:
    # more tiles in one launch than the chip keeps resident: populations of large robots (VXH_PROF_TILES=1 prints what a CU holds);
    # VXH_LIB=<path> to compare builds
    if os.environ.get("VXH_LIB"):
        engine.LIB_PATH = os.environ["VXH_LIB"]
    for count, n in ((40, 11), (96, 11), (200, 11), (24, 16)):
        timing_cfg(engine.VOXCAD, count, (n, n, n), 0.02, Env(), {}, full=True)


if __name__ == "__main__" and "sweepcase" in sys.argv[1:]:
    # one robot of a wider campaign of tests/test_gpu_parity.py test_land_water_parameter_sweep_vs_oracle (VXH_SWEEP_SEED / _COUNT / _MAXDIM) that
    # missed its bar: sweepcase <seed> <count> <maxdim> <index> [steps] -- regenerated with the test's generator, then ONE step of engine and
    # oracle from the same state, step after step (what differs first, and how much the oracle itself moves under a one-ulp jitter), and the
    # free-running difference at the end
    from collections import OrderedDict
    seed, count, maxdim, index = (int(a) for a in sys.argv[2:6])
    nsteps = int(sys.argv[6]) if len(sys.argv) > 6 else 150
    rng = np.random.RandomState(seed)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "voxelyzeFiles"))
    path = None
    for k in range(count):
        shape = tuple(int(n) for n in rng.randint(2, maxdim, size=3))
        sim_p = Sim(dt_frac=float(np.round(rng.uniform(0.3, 0.95), 2)), simulation_time=0.05,
                    fitness_eval_init_time=float(np.round(rng.uniform(0.0, 0.01), 3)), self_collisions_enabled=bool(rng.randint(2)))
        env_p = Env(frequency=float(np.round(rng.uniform(2, 8), 1)), gravity_enabled=int(rng.randint(2)), temp_enabled=int(rng.randint(2)),
                    floor_enabled=int(rng.randint(2)), temp_amp=float(np.round(rng.uniform(26, 45), 0)))
        if rng.randint(3) > 0:
            env_p.add_param("fluid_environment", 1, "<FluidEnvironment>")
            env_p.add_param("aggregate_drag_coefficient", float(rng.choice([50.0, 750.0, 3000.0])), "<AggregateDragCoefficient>")
        layers = OrderedDict()
        if rng.randint(2):
            layers["<PhaseOffset>"] = np.round(rng.uniform(-1, 1, size=shape), 3)
        if rng.randint(2):
            layers["<Stiffness>"] = np.round(10 ** rng.uniform(6.0, 8.0, size=shape), 0)
        ind = workloads.make_individual(k, workloads.random_material(shape, 300 + k, 0.2), layers or None)
        if k == index:
            write_voxelyze_file(sim_p, env_p, ind, tmp, "w")
            path = os.path.join(tmp, "voxelyzeFiles", "w--id_%05i.vxa" % k)
            print("robot %d: shape %s, dt_frac %.2f, selfcol %s, gravity %d temp %d floor %d, extra %s, layers %s" % (
                k, shape, sim_p.dt_frac, sim_p.self_collisions_enabled, env_p.gravity_enabled, env_p.temp_enabled, env_p.floor_enabled,
                [(p, v) for p, v in getattr(env_p, "new_param_tag_dict", {}).items()] if hasattr(env_p, "new_param_tag_dict") else "?", list(layers)), flush=True)
    model = vo.parse_vxa(path, 1)
    lat = model["lattice_dim"]
    sim, twin, free = vo.OracleSim(model), vo.OracleSim(model), vo.OracleSim(model)
    with engine.Engine(engine.VOXCAD_LAND_WATER, 0) as eng:
        eng.add_vxa_file(path)
        print("nvox %d nbond %d, kernel %s" % (eng.dims(0)["nvox"], eng.dims(0)["nbond"], eng.counters().dominant_block), flush=True)
        prev = sim.state()
        for step in range(1, nsteps + 1):
            eng.step(1)
            got = eng.state(0)
            sim.set_state(prev); sim.step(1)
            want = sim.state()
            twin.set_state(prev); twin.step_jittered(1, seed=step)
            free.step(1)
            own = np.abs(twin.state() - want)[:, :3].max() / lat
            dp = np.abs(got - want)[:, :3].max() / lat
            fr = np.abs(got - free.state())[:, :3].max() / lat
            if dp > 5e-14 or own > 5e-14 or step in (1, 3, 20, 60, 100, 150) or step == nsteps:
                v = int(np.argmax(np.abs(got - want)[:, :3].max(axis=1)))
                print("step %3d: one step engine - oracle %.3e voxel (voxel %d, z/lat %.4f); oracle under a one-ulp jitter %.3e; free-running difference %.3e" % (
                    step, dp, v, prev[v, 2] / lat, own, fr), flush=True)
            prev = got


if __name__ == "__main__" and "refcampaign" in sys.argv[1:]:
    # whole evaluations through the boundary against the REFERENCE BINARY (oracle/_ref, built where the reference lies): `count` random robots with
    # random switches (the generator of the parameter sweeps), every result file compared tag by tag.  refcampaign <variant 0/1> <seed> <count> <maxdim>
    import re, subprocess
    from collections import OrderedDict
    variant, seed, count, maxdim = (int(a) for a in sys.argv[2:6])
    ref = os.path.join(REPO, "oracle", "_ref", "voxelyze_lw_ref" if variant else "voxelyze_ref")
    rng = np.random.RandomState(seed)
    tmp = tempfile.mkdtemp()
    for d in ("voxelyzeFiles", "fitnessFiles", "tempFiles", "engineFitness"):
        os.makedirs(os.path.join(tmp, d))
    paths = []
    for k in range(count):
        shape = tuple(int(n) for n in rng.randint(9 if "big" in sys.argv[1:] else 2, maxdim, size=3))      # big: 9 .. maxdim - 1 per axis (the 768- / 1024-thread variants, tiles)
        sim_p = Sim(dt_frac=float(np.round(rng.uniform(0.3, 0.95), 2)), simulation_time=float(np.round(rng.uniform(0.05, 0.25), 2)),
                    fitness_eval_init_time=float(np.round(rng.uniform(0.0, 0.02), 3)), self_collisions_enabled=bool(rng.randint(2)))
        env_p = Env(frequency=float(np.round(rng.uniform(2, 8), 1)), gravity_enabled=int(rng.randint(2)), temp_enabled=int(rng.randint(2)),
                    floor_enabled=int(rng.randint(2)), temp_amp=float(np.round(rng.uniform(26, 45), 0)))
        if variant and rng.randint(3) > 0:
            env_p.add_param("fluid_environment", 1, "<FluidEnvironment>")
            env_p.add_param("aggregate_drag_coefficient", float(rng.choice([50.0, 750.0, 3000.0])), "<AggregateDragCoefficient>")
        layers = OrderedDict()
        if rng.randint(2):
            layers["<PhaseOffset>"] = np.round(rng.uniform(-1, 1, size=shape), 3)
        if rng.randint(2):
            layers["<Stiffness>"] = np.round(10 ** rng.uniform(6.0, 8.0, size=shape), 0)
        rich = "rich" in sys.argv[1:] and variant == 0          # more of the _voxcad writer's switches: development layers, sticky floor, growth amplitude, hard-wired constants
        if "stops" in sys.argv[1:]:          # the other stop conditions (a step count; a number of actuation periods), afterlife, mid-life freeze, floor slope
            kind = int(rng.randint(3))
            if kind == 1:
                sim_p.stop_condition = 1; sim_p.simulation_time = int(rng.randint(50, 400))
            elif kind == 2:
                sim_p.stop_condition = 3; sim_p.simulation_time = float(np.round(rng.uniform(0.3, 1.5), 2))
            if variant == 0 and rng.randint(3) == 0:
                sim_p.afterlife_time = float(np.round(rng.uniform(0.01, 0.05), 3))
            if variant == 0 and rng.randint(4) == 0:
                sim_p.mid_life_freeze_time = float(np.round(rng.uniform(0.01, 0.04), 3))
            if rng.randint(3) == 0:
                env_p.floor_slope = float(np.round(rng.uniform(0.0, 20.0), 1))
        if rich:
            env_p.sticky_floor = int(rng.randint(2))
            sim_p.min_temp_fact = float(np.round(rng.uniform(0.1, 0.6), 2))
            if rng.randint(2):
                env_p.add_param("growth_amplitude", float(np.round(rng.uniform(0.05, 0.5), 2)), "<GrowthAmplitude>")
            if rng.randint(3) == 0:
                if rng.randint(2):
                    env_p.add_param("min_growth_time", 0.01, "<MinGrowthTime>")
                for tname, lo, hi in (("<FinalPhaseOffset>", -1, 1), ("<TempAmpDamp>", 0.2, 1), ("<FinalTempAmpDamp>", 0.2, 1), ("<InitialVoxelSize>", -1, 1),
                                      ("<FinalVoxelSize>", -1, 1), ("<GrowthTime>", 0, 1), ("<StartGrowthTime>", 0, 1)):
                    if rng.randint(2):
                        layers[tname] = np.round(rng.uniform(lo, hi, size=shape), 3)
            if rng.randint(3) == 0:
                env_p.time_between_traces = float(rng.choice([0.005, 0.02]))
                env_p.add_param("save_traces", 1, "<SaveTraces>")
        ind = workloads.make_individual(k, workloads.random_material(shape, 300 + k, 0.2), layers or None)
        write_voxelyze_file(sim_p, env_p, ind, tmp, "c")
        paths.append(os.path.join(tmp, "voxelyzeFiles", "c--id_%05i.vxa" % k))
        if rich:
            text = open(paths[-1]).read()
            for tname, choices in (("BondDampingZ", ["1", "0.5", "0.1"]), ("ColDampingZ", ["0.8", "0.2"]), ("SlowDampingZ", ["0.01", "0.001", "0"]),
                                   ("ColSystem", ["3", "1"]), ("CollisionHorizon", ["2", "3"])):
                if "<" + tname + ">" in text:
                    old_t = text[text.index("<" + tname + ">"):text.index("</" + tname + ">")]
                    text = text.replace(old_t, "<" + tname + ">" + str(rng.choice(choices)), 1)
            open(paths[-1], "w").write(text)
    if "rich" in sys.argv[1:] or "stops" in sys.argv[1:]:
        # the engine refuses what it does not support (velocity-adjusted development: the reference reads past its trace there): leave those out
        keep = []
        for pth in paths:
            try:
                engine.inspect_vxa(pth)
                with engine.Engine(variant, 0) as probe:
                    probe.add_vxa_file(pth)
                keep.append(pth)
            except Exception as exc:
                print("   left out: %s (%s)" % (os.path.basename(pth), str(exc)[:80]), flush=True)
        paths = keep
        count = len(paths)
    running, queue = [], list(paths)
    while queue or running:
        while queue and len(running) < (os.cpu_count() or 8):
            running.append(subprocess.Popen(["timeout", "600", ref, "-f", queue.pop(0)], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        running = [p for p in running if p.poll() is None]
        time.sleep(0.005)
    tag = re.compile(r"<(\w+)>\s*([-+0-9.eE]+|nan|inf|-inf|-nan)\s*</\1>")
    identical, worst, worst_of, missing, steps, saved, differing, significant, sig_files = 0, 0.0, None, 0, 0, 0, {}, {}, set()
    hull_checked, hull_worst = 0, 0.0
    with engine.Engine(variant, 0) as eng:
        eng.add_vxa_files(paths)
        names = [eng.fitness_file_name(i) for i in range(count)]
        eng.run()
        for i in range(count):
            steps += eng.result(i).steps
            rp = names[i] if os.path.isabs(names[i]) else os.path.join(tmp, names[i])
            if not os.path.exists(rp):
                missing += 1          # (the reference wrote nothing: an empty lattice, or it ran into its own time-out)
                continue
            ref_text = open(rp).read()
            mine = os.path.join(tmp, "engineFitness", "%05d.xml" % i)
            eng.write_result_xml(i, mine)
            text = open(mine).read()
            identical += text == ref_text
            if variant == 1:
                # <ConvexHullVolumeStart> (the reference prints -1 without qhull): against scipy's qhull over the corners of the lattice cells the
                # robot fills, each coordinate printed with six significant digits as the reference hands them over (LW/VX_MeshUtil.cpp:806)
                from scipy.spatial import ConvexHull
                m = vo.parse_vxa(paths[i], 1)
                cells = np.argwhere(np.asarray(m["structure"]).reshape(m["nz"], m["ny"], m["nx"]) > 0)
                if cells is not None and len(cells) == eng.dims(i)["nvox"]:
                    lat_d = m["lattice_dim"]
                    corners = set()
                    for z, y, x in cells:
                        for dz in (0, 1):
                            for dy in (0, 1):
                                for dx in (0, 1):
                                    corners.add((x + dx, y + dy, z + dz))
                    pts = np.array([[float("%g" % (c * lat_d)) for c in p3] for p3 in corners])
                    want_h = ConvexHull(pts).volume
                    got_h = eng.result(i).hull_volume_start
                    hull_checked += 1
                    hull_worst = max(hull_worst, abs(got_h - want_h) / want_h)
            ta, tb = dict(tag.findall(ref_text)), dict(tag.findall(text))
            forced = str(i) in os.environ.get("VXH_CAMPAIGN_SAVE", "").split(",")                   # (robots to keep whatever the comparison says)
            if forced or (any(ta[k3] != tb.get(k3) for k3 in ta if not k3.startswith("ConvexHull")) and saved < 4):     # keep a few differing pairs for inspection (hull volumes aside: -1 from a reference without qhull)
                saved += 0 if forced else 1
                out = os.path.join(REPO, "gpurun_out", "refcampaign")
                os.makedirs(out, exist_ok=True)
                open(os.path.join(out, "v%d_s%d_r%d_reference.xml" % (variant, seed, i)), "w").write(ref_text)
                open(os.path.join(out, "v%d_s%d_r%d_engine.xml" % (variant, seed, i)), "w").write(text)
                import shutil
                shutil.copy(paths[i], os.path.join(out, "v%d_s%d_r%d.vxa" % (variant, seed, i)))
            a, b = dict(tag.findall(ref_text)), dict(tag.findall(text))
            for k2 in a:
                if a[k2] != b[k2]:
                    differing[k2] = differing.get(k2, 0) + 1
                    fa, fb = float(a[k2]), float(b[k2])
                    # not worth a look: NaN against NaN (x86 prints the sign of an invalid operation's NaN, "-nan"), and values that are rounding
                    # noise around zero on both sides (a robot that does not move: the difference of two centres of mass an ulp apart)
                    if not ((fa != fa and fb != fb) or max(abs(fa), abs(fb)) < 1e-9 or k2.startswith("ConvexHull")):
                        significant[k2] = significant.get(k2, 0) + 1
                        sig_files.add(i)
            assert set(a) == set(b), (i, sorted(set(a) ^ set(b)))
            for k2 in a:
                d = abs(float(a[k2]) - float(b[k2])) / max(1e-12, abs(float(a[k2])))
                if d > worst:
                    worst, worst_of = d, (i, k2, a[k2], b[k2])
    print("variant %d seed %d: %d robots, %d steps in all; reference files missing %d; byte-identical %d of %d; worst relative tag difference %.3e %s" % (
        variant, seed, count, steps, missing, identical, count - missing, worst, worst_of), flush=True)
    print("   files in which a tag's TEXT differs, by tag: %s" % differing, flush=True)
    if variant == 1:
        print("   ConvexHullVolumeStart against scipy's qhull over the filled cells' corners: %d robots, worst relative difference %.3e" % (hull_checked, hull_worst), flush=True)
    print("   ... of those, differences that are neither NaN against NaN nor noise around zero (< 1e-9 on both sides) nor a hull volume: %s in files %s" % (
        significant, sorted(sig_files)), flush=True)


if __name__ == "__main__" and "refusedloop" in sys.argv[1:]:
    # hunt for the flake of tests/test_gpu_tiled.py::test_a_tile_kernel_that_needs_scratch_is_refused...: refusedloop <count> [noprev] [norefuse] [noreset]
    count = int(sys.argv[2])
    path = os.path.join(REPO, "tests", "golden", "vxa", "rand6_col.vxa")
    with engine.Engine(engine.VOXCAD, 0) as eng:
        eng.set_option("tiled", 0); eng.set_option("wide", 0)
        eng.add_vxa_file(path)
        rest = eng.state(0).copy()
        eng.step(10)
        want = eng.state(0).copy()
    bad = 0
    for k in range(count):
        if "noprev" not in sys.argv[1:]:
            with engine.Engine(engine.VOXCAD, 0) as prev:     # (what ran before in the test file: tiles, several launches)
                prev.set_option("tiled", 2); prev.set_option("tiles_per_robot", 3); prev.set_option("steps_per_launch", 7)
                prev.add_vxa_file(path)
                prev.step(40)
        with engine.Engine(engine.VOXCAD, 0) as eng:
            eng.set_option("tiled", 2); eng.set_option("tiles_per_robot", 3)
            eng.add_vxa_file(path)
            if "norefuse" not in sys.argv[1:]:
                os.environ["VXH_TILE_SCRATCH_LIMIT"] = "-1"
                try:
                    eng.step(10)
                    print("not refused?!")
                except engine.VxhError:
                    pass
                del os.environ["VXH_TILE_SCRATCH_LIMIT"]
            if "noreset" not in sys.argv[1:]:
                eng.reset()
            eng.step(10)
            got = eng.state(0)
            d_want, d_rest = np.abs(got[:, :8] - want[:, :8]).max(), np.abs(got[:, :8] - rest[:, :8]).max()
            if d_want > 1e-12:
                bad += 1
                r, c = eng.result(0), eng.counters()
                print("iteration %d: |got - want| %.3e, |got - rest| %.3e; result steps %d status %d; counters launches %d max_steps %d dominant_block %d voxel_steps %.0f" % (
                    k, d_want, d_rest, r.steps, r.status, c.launches, c.max_steps, c.dominant_block, c.voxel_steps), flush=True)
                eng.step(10)
                got2 = eng.state(0)
                print("   ten more steps: result steps %d; |got2 - want| %.3e |got2 - got| %.3e" % (eng.result(0).steps, np.abs(got2[:, :8] - want[:, :8]).max(), np.abs(got2[:, :8] - got[:, :8]).max()), flush=True)
    print("refusedloop %s: %d of %d wrong" % (" ".join(sys.argv[3:]), bad, count), flush=True)


if __name__ == "__main__" and "xmlof" in sys.argv[1:]:
    # the result file of every .vxa given, as this library writes it: xmlof <variant> <path>...   (A/B of two libraries on one robot: scripts/ab_lib.py)
    variant = int(sys.argv[2])
    for path in sys.argv[3:]:
        with engine.Engine(variant, 0) as eng:
            eng.add_vxa_file(path)
            eng.run()
            out = os.path.join(tempfile.mkdtemp(), "r.xml")
            eng.write_result_xml(0, out)
            print("== %s: %d steps (%s)" % (os.path.basename(path), eng.result(0).steps, os.path.basename(engine.LIB_PATH)), flush=True)
            print(open(out).read(), flush=True)


if __name__ == "__main__" and "onestepfile" in sys.argv[1:]:
    # as sweepcase, for a .vxa file on disk: onestepfile <path> <variant> [steps]
    path, variant = sys.argv[2], int(sys.argv[3])
    model = vo.parse_vxa(path, variant)
    lat = model["lattice_dim"]
    sim, twin, free = vo.OracleSim(model), vo.OracleSim(model), vo.OracleSim(model)
    with engine.Engine(variant, 0) as eng:
        eng.add_vxa_file(path)
        planned = eng.dims(0)["planned_steps"]
        nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else planned
        print("nvox %d nbond %d planned steps %d" % (eng.dims(0)["nvox"], eng.dims(0)["nbond"], planned), flush=True)
        prev = sim.state()
        worst_one, worst_own = 0.0, 0.0
        for step in range(1, nsteps + 1):
            eng.step(1)
            got = eng.state(0)
            sim.set_state(prev); sim.step(1)
            want = sim.state()
            twin.set_state(prev); twin.step_jittered(1, seed=step)
            free.step(1)
            own = np.abs(twin.state() - want)[:, :3].max() / lat
            dp = np.abs(got - want)[:, :3].max() / lat
            worst_one, worst_own = max(worst_one, dp), max(worst_own, own)
            if dp > max(5e-14, 4 * own) or step % max(1, nsteps // 12) == 0 or step == nsteps:
                print("step %4d: one step engine - oracle %.3e voxel (worst so far %.3e); oracle under a one-ulp jitter %.3e (worst %.3e); free-running difference %.3e" % (
                    step, dp, worst_one, own, worst_own, np.abs(got - free.state())[:, :3].max() / lat), flush=True)
            prev = got
        r, o = eng.result(0), free.result()
        print("engine: steps %d num_touching_floor %s; oracle: steps %d" % (r.steps, getattr(r, "num_touching_floor", "?"), free.info().steps), flush=True)


if __name__ == "__main__" and "swimtiles" in sys.argv[1:]:
    # round 6: BASELINE configs[3] (64 random 8^3 swimmers) on the wide kernel against the same robots cut into k tiles each (k_tile_steps FLUID)
    swimmers = [workloads.random_material((8, 8, 8), i) for i in range(64)]
    for rep in range(2):
        _window(engine.VOXCAD_LAND_WATER, swimmers, _water(), {}, 300, 1000, tag="64 x 8^3 swimmers", extra=_phase_layer((8, 8, 8), 0), label="swimtiles")
        for k in (2, 3, 4):
            _window(engine.VOXCAD_LAND_WATER, swimmers, _water(), {"tiled": 2, "tiles_per_robot": k}, 300, 1000, tag="64 x 8^3 swimmers", extra=_phase_layer((8, 8, 8), 0), label="swimtiles")
    walkers = [workloads.random_material((8, 8, 8), i) for i in range(64)]
    _window(engine.VOXCAD, walkers, Env(), {}, 300, 1000, tag="64 x 8^3 walkers", label="swimtiles")
    for k in (2, 4):
        _window(engine.VOXCAD, walkers, Env(), {"tiled": 2, "tiles_per_robot": k}, 300, 1000, tag="64 x 8^3 walkers", label="swimtiles")
