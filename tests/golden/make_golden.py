#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REFERENCE itself.  Runs only in the build container (it needs
/root/reference and oracle/_ref/*, built by `make -C oracle ref`); the fixtures it writes are data
(inputs + expected outputs) and are what travels to the GPU box.

 1. .vxa text pinned by IMPORTING the reference writer (evosoro/tools/read_write_voxelyze.py runs under
    python3 up to its final md5 line, which raises TypeError after the file is complete) on duck-typed
    inputs; our writer must produce identical bytes (asserted here and again in tests/test_vxa_io.py).
 2. Expected physics from the reference C++ built from its own sources: the result XML written by
    oracle/_ref/voxelyze_ref, and binary state traces written by oracle/_ref/vxprobe (first 200 steps,
    every 25th step, full voxel state; plus initial/final state of the whole run).
 3. read_voxlyze_results pinned by running the reference reader on those result XMLs.
"""
import json
import os
import shutil
import subprocess
import sys
import random
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, "evosoro", "tools"))

import read_write_voxelyze as ref_rw  # noqa: E402  (the reference module)
from evosoro_amd.base import Sim, Env, ObjectiveDict  # noqa: E402
from evosoro_amd.tools import read_write_voxelyze as our_rw  # noqa: E402
from evosoro_amd import workloads  # noqa: E402

RUN_DIR = "golden_run"   # relative on purpose: it is embedded in <FitnessFileName>
RUN_NAME = "golden"


def cases():
    out = OrderedDict()
    # config 1 "plumbing": SURVEY App. C probe robot
    out["probe6"] = dict(variant="land", sim=Sim(dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0.1),
                         env=Env(), ind=workloads.make_individual(0, workloads.probe_material()))
    out["rand6_nocol"] = dict(variant="land",
                              sim=Sim(self_collisions_enabled=False, dt_frac=0.9, simulation_time=0.2,
                                      fitness_eval_init_time=0.05),
                              env=Env(), ind=workloads.random_robot(1, (6, 6, 6), 3))
    out["rand6_col"] = dict(variant="land", sim=Sim(dt_frac=0.9, simulation_time=0.25, fitness_eval_init_time=0.1),
                            env=Env(), ind=workloads.random_robot(2, (6, 6, 6), 7))
    # soft-only robot (no bone): 10x larger dt, big deformations, sticky floor off, InitCmTime = 0 quirk
    soft = workloads.random_material((5, 5, 5), 11)
    soft[soft == 2] = 1
    out["soft5_init0"] = dict(variant="land", sim=Sim(dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0),
                              env=Env(), ind=workloads.make_individual(3, soft))
    # per-voxel phase offsets + extra env tag through add_param
    env4 = Env(frequency=5.0, temp_amp=35)
    env4.add_param("growth_amplitude", 0.3, "<GrowthAmplitude>")
    # (phases rounded to 3 decimals: python2 str() prints 12 significant digits, python3 the shortest repr; the
    #  reference is python2 code but can only be imported under python3 here, so pin on values where both agree)
    phase = np.round(np.random.RandomState(99).uniform(-1, 1, size=(4, 4, 4)), 3)
    out["phase4"] = dict(variant="land", sim=Sim(dt_frac=0.8, simulation_time=0.3, fitness_eval_init_time=0.05),
                         env=env4, ind=workloads.make_individual(4, workloads.random_material((4, 4, 4), 5, 0.1),
                                                                OrderedDict([("<PhaseOffset>", phase)])))
    # ---- _voxcad_land_water semantics (two-sided actuation, float-typed stress modulus, other stop rule/result tags)
    phase6 = np.round(np.random.RandomState(7).uniform(-1, 1, size=(6, 6, 6)), 3)
    out["lw_land6"] = dict(variant="lw", sim=Sim(dt_frac=0.9, simulation_time=0.3, fitness_eval_init_time=0.05),
                           env=Env(), ind=workloads.make_individual(5, workloads.random_material((6, 6, 6), 21),
                                                                    OrderedDict([("<PhaseOffset>", phase6)])))
    # BASELINE configs[3] in small: swimmer in fluid (facet drag, no gravity/floor), as evosoro/examples/swimming_basic.py:137-138
    env_w = Env()
    env_w.add_param("fluid_environment", 1, "<FluidEnvironment>")
    env_w.add_param("aggregate_drag_coefficient", 750.0, "<AggregateDragCoefficient>")
    out["lw_swim6"] = dict(variant="lw", sim=Sim(dt_frac=0.9, simulation_time=0.3, fitness_eval_init_time=0.0),
                           env=env_w, ind=workloads.make_individual(6, workloads.random_material((6, 6, 6), 22),
                                                                    OrderedDict([("<PhaseOffset>", phase6)])))
    # ---- per-voxel evolved stiffness (evosoro/examples/land_continuous.py:109): the two variants apply it at different
    # points of the import, so one case each.  Whole numbers: python2 and python3 print them alike.
    stiff = np.round(10 ** np.random.RandomState(31).uniform(6.0, 8.0, size=(5, 5, 5)), 0)
    out["stiff5"] = dict(variant="land", sim=Sim(dt_frac=0.9, simulation_time=0.1, fitness_eval_init_time=0.02),
                         env=Env(), ind=workloads.make_individual(7, workloads.random_material((5, 5, 5), 41),
                                                                  OrderedDict([("<Stiffness>", stiff)])))
    out["lw_stiff5"] = dict(variant="lw", sim=Sim(dt_frac=0.9, simulation_time=0.1, fitness_eval_init_time=0.02),
                            env=Env(), ind=workloads.make_individual(8, workloads.random_material((5, 5, 5), 41),
                                                                     OrderedDict([("<Stiffness>", stiff)])))
    # ---- development (evosoro/examples/growth.py:69-75,91-95: IND_SIZE (5,5,4), GrowthAmplitude 0.5, MinTempFact 0.4,
    # DtFrac 0.5): per-voxel initial and final size; and a second case with every development layer present
    env_g = Env()
    env_g.add_param("growth_amplitude", 0.5, "<GrowthAmplitude>")
    rs = np.random.RandomState(51)
    ini = np.round(rs.uniform(-1, 1, size=(5, 5, 4)), 3)
    fin = np.round(rs.uniform(-1, 1, size=(5, 5, 4)), 3)
    out["grow5"] = dict(variant="land",
                        sim=Sim(dt_frac=0.5, simulation_time=0.4, fitness_eval_init_time=0.05, min_temp_fact=0.4),
                        env=env_g, ind=workloads.make_individual(9, workloads.random_material((5, 5, 4), 52, 0.15),
                                                                 OrderedDict([("<InitialVoxelSize>", ini), ("<FinalVoxelSize>", fin)])))
    env_g2 = Env()
    env_g2.add_param("growth_amplitude", 0.3, "<GrowthAmplitude>")
    env_g2.add_param("min_growth_time", 0.01, "<MinGrowthTime>")
    rs = np.random.RandomState(53)
    layers = OrderedDict()
    for tag, lo, hi in (("<PhaseOffset>", -1, 1), ("<FinalPhaseOffset>", -1, 1), ("<TempAmpDamp>", 0.2, 1), ("<FinalTempAmpDamp>", 0.2, 1),
                        ("<InitialVoxelSize>", -1, 1), ("<FinalVoxelSize>", -1, 1), ("<GrowthTime>", 0, 1), ("<StartGrowthTime>", 0, 1)):
        layers[tag] = np.round(rs.uniform(lo, hi, size=(4, 4, 4)), 3)
    out["devo4"] = dict(variant="land",
                        sim=Sim(dt_frac=0.7, simulation_time=0.3, fitness_eval_init_time=0.04, min_temp_fact=0.5),
                        env=env_g2, ind=workloads.make_individual(10, workloads.random_material((4, 4, 4), 54, 0.1), layers))
    # ---- BASELINE configs[2] size: the first two robots of the bench population (bench.py: random 10x10x10, seeds 0 and 1,
    # self-collision on), the whole 0.5 s evaluation; final state and result XML only
    for k in (0, 1):
        out["bench10_%d" % k] = dict(variant="land", early=False,
                                     sim=Sim(dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0.05),
                                     env=Env(), ind=workloads.random_robot(20 + k, (10, 10, 10), k))
    # ---- _voxcad_land_water at BASELINE configs[2] size: a random 10x10x10 swimmer (~700 voxels: the fused kernel's
    # 768-thread MESH variant) and a full 10x10x10 lattice walking on land (1000 voxels: the 1024-thread one); final
    # state and result XML (RobotVolumeEnd from the strains of the last step) only
    phase10 = np.round(np.random.RandomState(17).uniform(-1, 1, size=(10, 10, 10)), 3)
    out["lw_swim10"] = dict(variant="lw", early=False, sim=Sim(dt_frac=0.9, simulation_time=0.1, fitness_eval_init_time=0.0),
                            env=env_w, ind=workloads.make_individual(30, workloads.random_material((10, 10, 10), 61),
                                                                     OrderedDict([("<PhaseOffset>", phase10)])))
    out["lw_land10"] = dict(variant="lw", early=False, sim=Sim(dt_frac=0.9, simulation_time=0.1, fitness_eval_init_time=0.02),
                            env=Env(), ind=workloads.make_individual(31, workloads.full_material(10, 62),
                                                                     OrderedDict([("<PhaseOffset>", phase10)])))
    # ---- BASELINE configs[1] and [3] at their stated sizes: four robots of each batch (the GPU tests step the whole batches and
    # compare these four with the reference), the whole 0.5 s evaluation; final state and result XML only
    for k in (0, 21, 42, 63):
        out["cfg1_%02d" % k] = dict(variant="land", early=False,
                                    sim=Sim(dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0.1),
                                    env=Env(), ind=workloads.random_robot(100 + k, (6, 6, 6), k))
        out["cfg3_%02d" % k] = dict(variant="lw", early=False,
                                    sim=Sim(dt_frac=0.9, simulation_time=0.5, fitness_eval_init_time=0.05),
                                    env=env_w, ind=workloads.swimmer(200 + k, (8, 8, 8), k))
    # ---- the other stop rules (VX_Sim.cpp:1398-1423): type 1 = a number of time steps, type 3 = a number of actuation periods
    soft4 = workloads.random_material((5, 5, 4), 71, 0.2)
    soft4[soft4 == 2] = 1
    out["stop1_5"] = dict(variant="land", sim=Sim(dt_frac=0.9, stop_condition=1, simulation_time=700, fitness_eval_init_time=0.05),
                          env=Env(), ind=workloads.make_individual(40, soft4))
    out["stop3_5"] = dict(variant="land", sim=Sim(dt_frac=0.9, stop_condition=3, simulation_time=1.5, fitness_eval_init_time=0.05),
                          env=Env(), ind=workloads.make_individual(41, soft4))
    # ---- centre-of-mass trace in the result file (VX_Sim.cpp:1537-1547, VX_SimGA.cpp:170-184): <TimeBetweenTraces> + <SaveTraces>
    env_t = Env(frequency=5.0, temp_amp=35, time_between_traces=0.02)
    env_t.add_param("growth_amplitude", 0.3, "<GrowthAmplitude>")
    env_t.add_param("save_traces", 1, "<SaveTraces>")
    out["trace4"] = dict(variant="land", sim=Sim(dt_frac=0.8, simulation_time=0.3, fitness_eval_init_time=0.05),
                         env=env_t, ind=workloads.make_individual(43, workloads.random_material((4, 4, 4), 5, 0.1),
                                                                  OrderedDict([("<PhaseOffset>", phase)])))
    out["lw_stop3_5"] = dict(variant="lw", sim=Sim(dt_frac=0.9, stop_condition=3, simulation_time=1.5, fitness_eval_init_time=0.05),
                             env=Env(), ind=workloads.make_individual(42, soft4))
    return out


class _Pop(object):
    pass


def main():
    # `make_golden.py --only name,name`: (re)generate just these generated cases and keep everything else as it is
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    work = os.path.join("/tmp", "vx_golden_work")
    shutil.rmtree(work, ignore_errors=True)
    # the land_water simulator shells out to `qhull` for <ConvexHullVolumeStart/End> (LW/VX_MeshUtil.cpp:821-900) and prints -1 when
    # there is none: give it the binary the reference vendors (a temporary executable copy, on PATH for the child processes only)
    qdir = os.path.join("/tmp", "vx_golden_qhull")
    os.makedirs(qdir, exist_ok=True)
    if not os.path.exists(os.path.join(qdir, "qhull")):
        shutil.copy(os.path.join(REF, "evosoro", "_voxcad", "qhull"), os.path.join(qdir, "qhull"))
        os.chmod(os.path.join(qdir, "qhull"), 0o755)
    os.environ["PATH"] = qdir + os.pathsep + os.environ["PATH"]
    for sub in ("ref", "ours"):
        for d in ("voxelyzeFiles", "fitnessFiles", "tempFiles"):
            os.makedirs(os.path.join(work, sub, RUN_DIR, d))
    vxa_dir = os.path.join(HERE, "vxa")
    exp_dir = os.path.join(HERE, "expected")
    manifest = OrderedDict()
    if only is None:
        shutil.rmtree(vxa_dir, ignore_errors=True)
        shutil.rmtree(exp_dir, ignore_errors=True)
        os.makedirs(vxa_dir)
        os.makedirs(exp_dir)
    else:
        with open(os.path.join(HERE, "manifest.json")) as f:
            manifest = json.load(f, object_pairs_hook=OrderedDict)
    probe = {"land": os.path.join(REPO, "oracle/_ref/vxprobe"), "lw": os.path.join(REPO, "oracle/_ref/vxprobe_lw")}
    refbin = {"land": os.path.join(REPO, "oracle/_ref/voxelyze_ref"),
              "lw": os.path.join(REPO, "oracle/_ref/voxelyze_lw_ref")}

    for name, case in cases().items():
        if only is not None and name not in only:
            continue
        ind = case["ind"]
        fname = RUN_NAME + "--id_%05i.vxa" % ind.id
        # reference writer (py3: file complete, then TypeError at the md5 update)
        os.chdir(os.path.join(work, "ref"))
        random.seed(12345)
        try:
            ref_rw.write_voxelyze_file(case["sim"], case["env"], ind, RUN_DIR, RUN_NAME)
        except TypeError:
            pass
        ref_text = open(os.path.join(RUN_DIR, "voxelyzeFiles", fname)).read()
        state_after_ref = random.getstate()
        # ours
        os.chdir(os.path.join(work, "ours"))
        random.seed(12345)
        md5 = our_rw.write_voxelyze_file(case["sim"], case["env"], ind, RUN_DIR, RUN_NAME)
        our_text = open(os.path.join(RUN_DIR, "voxelyzeFiles", fname)).read()
        assert our_text == ref_text, "writer mismatch for %s" % name
        assert random.getstate() == state_after_ref, "random stream consumption differs for %s" % name
        with open(os.path.join(vxa_dir, name + ".vxa"), "w") as f:
            f.write(ref_text)

        # reference physics
        os.chdir(os.path.join(work, "ref"))
        vxa = os.path.join(RUN_DIR, "voxelyzeFiles", fname)
        subprocess.run(["timeout", "900", refbin[case["variant"]], "-f", vxa], check=False)
        xml = os.path.join(RUN_DIR, "fitnessFiles", "softbotsOutput--id_%05i.xml" % ind.id)
        shutil.copy(xml, os.path.join(exp_dir, name + ".xml"))
        if name == "soft5_init0":       # what the reference prints with -p (voxelyzeMain/main.cpp:60-63,92-104,128): the CLI's only diagnostic
            with open(os.path.join(exp_dir, name + ".p.txt"), "wb") as f:
                subprocess.run(["timeout", "900", refbin[case["variant"]], "-f", vxa, "-p"], check=False, stdout=f, stderr=subprocess.DEVNULL)
            # ... and with -p --computeShapeDescriptors (main.cpp:65-88,113-126: mesh size, volumes, shape complexity, printAllMeshInfo before and after the run)
            with open(os.path.join(exp_dir, name + ".pcsd.txt"), "wb") as f:
                subprocess.run(["timeout", "900", refbin[case["variant"]], "-f", vxa, "-p", "--computeShapeDescriptors"], check=False, stdout=f, stderr=subprocess.DEVNULL)
        if case.get("early", True):     # (skipped for the 10^3 robots: 700 voxels x 9 snapshots would be 0.7 MB each)
            subprocess.run(["timeout", "900", probe[case["variant"]], "-f", vxa, "-o",
                            os.path.join(exp_dir, name + ".early.bin"), "-max", "200", "-every", "25", "-noresult"],
                           check=True)
        subprocess.run(["timeout", "900", probe[case["variant"]], "-f", vxa, "-o",
                        os.path.join(exp_dir, name + ".final.bin"), "-every", "100000000", "-noresult"], check=True)

        # reference reader on the reference XML
        pop = _Pop()
        pop.objective_dict = ObjectiveDict()
        if case["variant"] == "land":
            pop.objective_dict.add_objective(name="fitness", maximize=True, tag="<NormFinalDist>")
            pop.objective_dict.add_objective(name="age", maximize=False, tag=None)
            pop.objective_dict.add_objective(name="y", maximize=True, tag="<finalDistY>")
            pop.objective_dict.add_objective(name="touch", maximize=True, tag="<NumTouchingFloor>")
        else:   # objectives of evosoro/examples/swimming_basic.py:147 and land_continuous.py:159
            pop.objective_dict.add_objective(name="fitness", maximize=True, tag="<normAbsoluteDisplacement>")
            pop.objective_dict.add_objective(name="age", maximize=False, tag=None)
            pop.objective_dict.add_objective(name="z", maximize=True, tag="<normDistZ>")
            pop.objective_dict.add_objective(name="n", maximize=True, tag="<VoxelNumber>")
        values = ref_rw.read_voxlyze_results(pop, None, xml)
        manifest[name] = {"variant": case["variant"], "id": ind.id, "md5": md5,
                          "read_results": {str(k): v for k, v in values.items()},
                          "nvox": int((ind.genotype.to_phenotype_mapping["material"]["state"] > 0).sum())}
        print(name, manifest[name])

    if only is not None:
        with open(os.path.join(HERE, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1)
        return
    # input files the reference ships next to its simulator (data, not code); expected values from the reference
    shipped = [("land", "evosoro/_voxcad/voxelyzeMain/Example_withPhaseOffset.vxa", "example_phaseoffset"),
               ("land", "evosoro/_voxcad/voxelyzeMain/Example_1.vxa", "example_1"),
               ("lw", "evosoro/_voxcad_land_water/sample_vxa/hexapus.vxa", "lw_hexapus"),
               ("lw", "evosoro/_voxcad_land_water/sample_vxa/quadruped_land.vxa", "lw_quadruped_land")]
    for variant, rel, name in shipped:
        wd = os.path.join(work, "shipped_" + name)
        os.makedirs(os.path.join(wd, "fitnessFiles"))
        shutil.copy(os.path.join(REF, rel), os.path.join(wd, name + ".vxa"))
        os.chdir(wd)
        subprocess.run(["timeout", "900", refbin[variant], "-f", name + ".vxa"], check=False)
        outs = [p for p in os.listdir(wd) + [os.path.join("fitnessFiles", q) for q in os.listdir("fitnessFiles")]
                if p.endswith(".xml")]
        assert len(outs) == 1, outs
        shutil.copy(os.path.join(wd, name + ".vxa"), os.path.join(vxa_dir, name + ".vxa"))
        shutil.copy(outs[0], os.path.join(exp_dir, name + ".xml"))
        subprocess.run(["timeout", "900", probe[variant], "-f", name + ".vxa", "-o",
                        os.path.join(exp_dir, name + ".early.bin"), "-max", "800" if variant == "lw" else "200",
                        "-every", "200" if variant == "lw" else "50", "-noresult"], check=True)
        manifest[name] = {"variant": variant, "shipped": rel, "result_path": outs[0]}

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
