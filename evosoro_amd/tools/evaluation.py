"""evaluate_all: evaluate every individual of a population — one batched GPU run instead of one OS process each.

Drop-in for evosoro/tools/evaluation.py:18-219 (same signature, same side effects):
  * every individual is serialised with write_voxelyze_file and gets `ind.md5`                 (reference :62)
  * invalid phenotypes get the worst value of every objective except "age"                     (:65-69)
  * with zero actuation variance an md5 already evaluated reuses the cached objective values   (:72-81)
  * the rest are simulated; `pop.total_evaluations`, `pop.all_evaluated_individuals_ids`,
    `pop.already_evaluated[md5]` and `pop.best_fit_so_far` are maintained                      (:84-86,179-190)
  * objectives with a tag are read back from the result XML (6 significant digits, through
    read_voxlyze_results exactly as in the reference), objectives without a tag are computed
    from the phenotype with their node_func                                                    (:160-177)
  * .vxa housekeeping: champions copied to bestSoFar/fitOnly, lineages to ancestors/, the
    generation's files moved to Gen_%04i/ or deleted                                           (:185-203)
  * individuals whose simulation did not finish (diverged, empty, ...) keep the worst value, which is
    what the reference leaves after its timeout                                                (:107-119,213-215)
What changes is the transport: instead of `sub.Popen("./voxelyze -f ...")` per robot and polling
fitnessFiles/ (:89-90,128-158), all pending robots go through libvxhip in ONE call, which writes the same
result XMLs.  In a multi-GPU job (torch.distributed initialised: one process per GPU, all running the same
seeded EA on one shared run directory) the robots are sharded over the ranks, the run directory belongs to
rank 0, and every rank reads the objective values from the gathered fitness table, rounded to the six
significant digits the XML would carry: all ranks end the generation with identical populations.
`max_eval_time` and `time_to_try_again` are accepted for compatibility; there is nothing to time out.
"""
import os
import shutil
import time

import numpy as np

import random

from evosoro_amd.tools.read_write_voxelyze import phenotype_arrays, read_voxlyze_results, write_voxelyze_file


def _vxa_path(run_directory, run_name, ident):
    return run_directory + "/voxelyzeFiles/" + run_name + "--id_%05i.vxa" % ident


def _estimated_cost(ind):
    """voxels x relative step count (a stiff 'bone' material, id 2, shrinks dt about tenfold)."""
    for _, details in ind.genotype.to_phenotype_mapping.items():
        if details["tag"] == "<Data>":
            state = np.asarray(details["state"])
            return float(np.count_nonzero(state)) * (10.0 if (state == 2).any() else 1.0)
    return float(np.prod(ind.genotype.orig_size_xyz))


# result-XML tag of an objective -> field of the fitness record every rank receives (parallel.RECORD_FIELDS); the tags are those of
# CVX_SimGA::WriteResultFile (VX_SimGA.cpp:145-168, LW/VX_SimGA.cpp:58-68)
_TAG_FIELDS = {"NormFinalDist": "norm_final_dist", "NormRegimeDist": "norm_regime_dist", "NormFrozenDist": "norm_frozen_dist",
               "FinalDist": "final_dist", "finalDistY": "final_dist_y", "AnteriorDist": "anterior_dist",
               "PosteriorDist": "posterior_dist", "AnteriorY": "anterior_y", "PosteriorY": "posterior_y",
               "EndOfLifePosteriorY": "end_of_life_posterior_y", "FallAdjPostY": "fall_adj_post_y",
               "NumNonFeetTouchingFloor": "num_non_feet_touching_floor", "NumTouchingFloor": "num_touching_floor",
               "Lifetime": "lifetime", "normAbsoluteDisplacement": "norm_abs_disp", "normDistX": "norm_dist_x",
               "normDistY": "norm_dist_y", "normDistZ": "norm_dist_z", "RobotVolumeStart": "robot_volume_start",
               "RobotVolumeEnd": "robot_volume_end", "ConvexHullVolumeStart": "hull_volume_start",
               "ConvexHullVolumeEnd": "hull_volume_end"}


def _values_from_record(pop, record, parallel):
    """What read_voxlyze_results would parse out of the robot's result XML, taken from its fitness record instead: the XML
    prints 6 significant digits (C++ stream default), so the record's doubles go through the same rounding."""
    values = {}
    for rank, details in pop.objective_dict.items():
        tag = details["tag"]
        if tag is None:
            values[rank] = None
            continue
        field = _TAG_FIELDS.get(tag.strip("<>"))
        if field is None:
            raise KeyError("objective tag %s is not a tag of the result file" % tag)
        values[rank] = float("%.6g" % record[parallel.RECORD_FIELDS.index(field)])
    return values


def evaluate_all(sim, env, pop, print_log, save_vxa_every, run_directory, run_name, max_eval_time=60,
                 time_to_try_again=10, save_lineages=False, variant=0, engine_module=None, engine_options=None, in_memory=None):
    """in_memory: hand the generation to the engine as arrays (vxh_add_robots) instead of through .vxa files; the files are then
    only written for the individuals the bookkeeping keeps (champions, lineages, saved generations).  Default (None): whenever that is
    exact -- no actuation variance (the file would carry per-robot random CTEs), no genotype output driving the environment -- and the
    engine offers it.  The robots are bit-identical to those the file route builds (tests/test_gpu_parity.py)."""
    start_time = time.time()
    if engine_module is None:
        from evosoro_amd import engine as engine_module   # fails loudly when libvxhip.so / a GPU is missing
    from evosoro_amd import parallel

    # Multi-GPU job (torch.distributed initialised, one process per GPU, every rank runs the same seeded EA on the same
    # shared run directory): the files belong to rank 0 -- it alone writes, copies, moves and removes them -- and every rank
    # takes the objective values from the gathered fitness table, so that all ranks end the generation with identical populations.
    rank, world = parallel.rank_and_world()
    distributed = world > 1
    owner = rank == 0

    arrays = {}
    if in_memory is None or in_memory:
        possible = env.actuation_variance == 0 and hasattr(engine_module.Engine, "add_robots")
        if possible:
            for ind in pop:
                arrays[ind.id] = phenotype_arrays(ind)
            possible = all(a is not None for a in arrays.values())
        if in_memory and not possible:
            raise ValueError("in_memory=True, but this population needs its .vxa files (actuation variance, environment driven by the genotype)")
        in_memory = possible
    template = None

    def ensure_file(ind):
        """the individual's .vxa, written now if the in-memory route skipped it (without touching the global random stream again)"""
        if in_memory and not os.path.exists(_vxa_path(run_directory, run_name, ind.id)):
            state = random.getstate()
            write_voxelyze_file(sim, env, ind, run_directory, run_name)
            random.setstate(state)

    pending = []
    for ind in pop:
        if in_memory:
            ind.md5, text = write_voxelyze_file(sim, env, ind, run_directory, run_name, write=False, want_text=True)
            template = template or text
        else:
            ind.md5 = write_voxelyze_file(sim, env, ind, run_directory, run_name, write=owner)
        if not ind.phenotype.is_valid():
            for _, goal in pop.objective_dict.items():
                if goal["name"] != "age":
                    setattr(ind, goal["name"], goal["worst_value"])
            print_log.message("Skipping invalid individual")
        elif env.actuation_variance == 0 and ind.md5 in pop.already_evaluated:
            for obj_rank, goal in pop.objective_dict.items():
                if goal["tag"] is not None:
                    setattr(ind, goal["name"], pop.already_evaluated[ind.md5][obj_rank])
            if owner and save_vxa_every > 0 and pop.gen % save_vxa_every == 0:
                ensure_file(ind)
                shutil.copy(_vxa_path(run_directory, run_name, ind.id),
                            run_directory + "/Gen_%04i/" % pop.gen + run_name +
                            "--Gen_%04i--fit_%.08f--id_%05i.vxa" % (pop.gen, ind.fitness, ind.id))
        else:
            pop.total_evaluations += 1
            pending.append(ind)

    print_log.message("Launched {0} voxelyze calls, out of {1} individuals".format(len(pending), len(pop)))

    table = None
    if pending:
        if owner:
            os.makedirs(run_directory + "/fitnessFiles", exist_ok=True)
        if distributed and not in_memory:
            parallel.barrier()                 # the files of the generation are complete before any rank opens them
        paths = [_vxa_path(run_directory, run_name, ind.id) for ind in pending]

        def make_loader(indices):
            def loader(eng):
                eng.add_robots(template, [(arrays[pending[i].id][0], arrays[pending[i].id][1],
                                           run_directory + "/fitnessFiles/softbotsOutput--id_%05i.xml" % pending[i].id) for i in indices])
            return loader
        table = parallel.run_population(engine_module, paths, variant=variant,
                                        costs=[_estimated_cost(ind) for ind in pending], options=engine_options,
                                        write_xml=not distributed, make_loader=make_loader if in_memory else None)
    num_finished = 0
    for k, ind in enumerate(pending):
        status = int(table[k, 0])
        xml = run_directory + "/fitnessFiles/softbotsOutput--id_%05i.xml" % ind.id
        if status != engine_module.ROBOT_FINISHED or (not distributed and not os.path.exists(xml)):
            print_log.message("WARNING: simulation of id {0} did not finish (status {1}); "
                              "the min fitness was assigned".format(ind.id, status))
            continue
        num_finished += 1
        if distributed:
            values = _values_from_record(pop, table[k], parallel)
        else:
            values = read_voxlyze_results(pop, print_log, xml)      # the reference's own path: the numbers as the file prints them
            os.remove(xml)
        print_log.message("{0} fit = {1} ({2} / {3})".format(os.path.basename(xml), values[0], num_finished,
                                                             len(pending)))
        for obj_rank, details in pop.objective_dict.items():
            if values[obj_rank] is not None:
                setattr(ind, details["name"], values[obj_rank])
            else:
                for name, details_phenotype in ind.genotype.to_phenotype_mapping.items():
                    if name == details["output_node_name"]:
                        setattr(ind, details["name"], details["node_func"](details_phenotype["state"]))
        pop.already_evaluated[ind.md5] = [getattr(ind, details["name"]) for _, details in pop.objective_dict.items()]
        pop.all_evaluated_individuals_ids += [ind.id]

        vxa = _vxa_path(run_directory, run_name, ind.id)
        stamped = run_name + "--Gen_%04i--fit_%.08f--id_%05i.vxa" % (pop.gen, ind.fitness, ind.id)
        champion = ind.fitness > pop.best_fit_so_far
        if champion:
            pop.best_fit_so_far = ind.fitness
        if owner:
            keep = save_vxa_every > 0 and pop.gen % save_vxa_every == 0
            if champion or save_lineages or keep:
                ensure_file(ind)
            if champion:
                shutil.copy(vxa, run_directory + "/bestSoFar/fitOnly/" + stamped)
            if save_lineages:
                shutil.copy(vxa, run_directory + "/ancestors/")
            if keep:
                shutil.move(vxa, run_directory + "/Gen_%04i/" % pop.gen + stamped)
            elif os.path.exists(vxa):
                os.remove(vxa)
    if owner:
        # robots that did not finish leave their .vxa behind in the reference too (it times out and moves on); nothing else to do
        pass
    if distributed and pending and not in_memory:
        parallel.barrier()                     # rank 0 has finished with the files before any rank starts the next generation

    if num_finished < len(pending):
        print_log.message("WARNING: Couldn't get a fitness value in time for some individuals. "
                          "The min fitness was assigned for these individuals")
    print_log.message("\nAll Voxelyze evals finished in {} seconds".format(time.time() - start_time))
    print_log.message("num_evaluated_this_gen: {0}".format(len(pending)))
    print_log.message("total_evaluations: {}".format(pop.total_evaluations))
