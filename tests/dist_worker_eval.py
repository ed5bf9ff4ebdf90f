"""Worker of tests/test_evaluation.py::test_evaluate_all_on_two_ranks: one rank of a gloo job on CPU running evaluate_all
(oracle-backed stub engine) on a run directory shared with the other rank."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from evosoro_amd import workloads  # noqa: E402
from evosoro_amd.base import Sim, Env  # noqa: E402
from evosoro_amd.tools.evaluation import evaluate_all  # noqa: E402
import stub_engine  # noqa: E402
from test_evaluation import Log, make_pop  # noqa: E402


def main():
    run, in_memory = sys.argv[1], sys.argv[2] == "memory"
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    sim = Sim(dt_frac=0.9, simulation_time=0.06, fitness_eval_init_time=0.02)
    env = Env()
    inds = [workloads.random_robot(i, (4, 4, 4), 50 + i, 0.2) for i in range(5)]
    inds.append(workloads.make_individual(5, np.zeros((4, 4, 4), dtype=int)))        # invalid phenotype
    pop = make_pop(inds)
    log = Log()
    evaluate_all(sim, env, pop, log, save_vxa_every=1, run_directory=run, run_name="D", engine_module=stub_engine, in_memory=in_memory)
    # second generation: two clones of evaluated robots (md5 cache) and a new one
    inds2 = [workloads.random_robot(10, (4, 4, 4), 50, 0.2), workloads.random_robot(11, (4, 4, 4), 99, 0.2)]
    pop2 = make_pop(inds2)
    pop2.gen = 1
    pop2.already_evaluated, pop2.best_fit_so_far = pop.already_evaluated, pop.best_fit_so_far
    pop2.total_evaluations, pop2.all_evaluated_individuals_ids = pop.total_evaluations, pop.all_evaluated_individuals_ids
    evaluate_all(sim, env, pop2, log, save_vxa_every=0, run_directory=run, run_name="D", engine_module=stub_engine, in_memory=in_memory)
    out = {"fitness": [ind.fitness for ind in list(pop) + list(pop2)], "md5": [ind.md5 for ind in list(pop) + list(pop2)],
           "best": pop2.best_fit_so_far, "total": pop2.total_evaluations, "ids": pop2.all_evaluated_individuals_ids,
           "cache": sorted(pop2.already_evaluated), "warnings": [l for l in log.lines if "WARNING" in l]}
    with open(os.path.join(run, "rank%d.json" % rank), "w") as f:
        json.dump(out, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
