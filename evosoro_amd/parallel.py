"""Population sharding across GPUs and the fitness gather.

The reference evaluates a generation by launching one OS process per robot and letting the kernel schedule them
over the host cores (evosoro/tools/evaluation.py:59-90); results come back through files.  Here a generation is
a batch: robots are independent, so the batch is partitioned over the ranks of a `torch.distributed` job (one
process per GPU, backend "nccl" = RCCL over xGMI on MI355X, "gloo" on CPU for tests), every rank steps its shard
on its own GPU with no data-path communication, and ONE collective returns the fixed-size result records
(<= 256 B per robot) to every rank.
"""
import os
import sys

import numpy as np

RECORD_FIELDS = ["status", "steps", "nvox", "nbond", "dt", "cur_time", "lifetime",
                 "ini_cm_x", "ini_cm_y", "ini_cm_z", "cur_cm_x", "cur_cm_y", "cur_cm_z",
                 "norm_final_dist", "norm_regime_dist", "norm_frozen_dist", "final_dist", "final_dist_y",
                 "anterior_dist", "posterior_dist", "anterior_y", "posterior_y", "end_of_life_posterior_y",
                 "fall_adj_post_y", "num_non_feet_touching_floor", "num_touching_floor",
                 "norm_abs_disp", "norm_dist_x", "norm_dist_y", "norm_dist_z", "robot_volume_start", "robot_volume_end",
                 "col_rebuilds", "hull_volume_start", "hull_volume_end"]
RECORD_LEN = len(RECORD_FIELDS)


def result_to_record(res):
    """vxh_result (ctypes struct from evosoro_amd.engine) -> float64 vector in RECORD_FIELDS order."""
    rec = np.empty(RECORD_LEN, dtype=np.float64)
    rec[0:7] = (res.status, res.steps, res.nvox, res.nbond, res.dt, res.cur_time, res.lifetime)
    rec[7:10] = list(res.ini_cm)
    rec[10:13] = list(res.cur_cm)
    for k, name in enumerate(RECORD_FIELDS[13:]):
        rec[13 + k] = getattr(res, name)
    return rec


def shard_by_cost(costs, world_size):
    """Greedy longest-processing-time partition: returns, per rank, the sorted list of item indices.

    cost of a robot = voxels x time steps (the step count varies 10x with the stiffest material present,
    SURVEY.md section 7).  Deterministic: ties broken by index.
    """
    costs = np.asarray(costs, dtype=np.float64)
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


def shard_round_robin(count, world_size):
    return [list(range(r, count, world_size)) for r in range(world_size)]


def _dist():
    """torch.distributed if importable and initialised with more than one rank, else None (single GPU: no torch needed)"""
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


def rank_and_world():
    dist = _dist()
    return (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)


def barrier():
    dist = _dist()
    if dist:
        dist.barrier()


def gather_records(local_records, local_indices, total, device=None, per_rank=None):
    """All ranks receive the full [total, RECORD_LEN] table.  ONE all_gather of a padded per-rank block: row 0 of a rank's block
    carries its record count, rows 1.. carry (robot index, record).  `per_rank`: rows every rank reserves for its records -- all
    ranks must pass the same value; callers that know the partition pass its largest shard (run_population), the default is `total`
    (always enough: 288 B per robot and rank, a few hundred KB for an evosoro population).

    Without an initialised process group (single GPU) this is a plain scatter into the table.
    """
    table = np.zeros((total, RECORD_LEN), dtype=np.float64)
    local_records = np.asarray(local_records, dtype=np.float64).reshape(-1, RECORD_LEN)
    dist = _dist()
    if dist is None:
        table[list(local_indices)] = local_records
        return table
    import torch
    world = dist.get_world_size()
    per_rank = int(total if per_rank is None else per_rank)
    n_mine = len(local_indices)
    if n_mine > per_rank:
        raise ValueError("gather_records: %d records on this rank, %d rows reserved per rank" % (n_mine, per_rank))
    block = np.zeros((per_rank + 1, RECORD_LEN + 1), dtype=np.float64)
    block[0, 0] = n_mine
    if n_mine:
        block[1:n_mine + 1, 0] = list(local_indices)
        block[1:n_mine + 1, 1:] = local_records
    mine = torch.from_numpy(block)
    if device is not None:
        mine = mine.to(device)
    gathered = torch.zeros((world * (per_rank + 1), RECORD_LEN + 1), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(gathered, mine)
    gathered = gathered.cpu().numpy().reshape(world, per_rank + 1, RECORD_LEN + 1)
    for r in range(world):
        n = int(gathered[r, 0, 0])
        for k in range(1, n + 1):
            table[int(gathered[r, k, 0])] = gathered[r, k, 1:]
    return table


PIPELINE_ENGINES = 2      # in-memory hand-off: engines of the one handle a generation is pipelined over (chunk k + 1 is built on the host
                          # cores while chunk k steps: include/vxhip.h vxh_create_multi with a repeated device id); 1 = no pipeline


def run_shard(engine_module, paths, variant, device_index, options=None, write_xml=True, loader=None):
    """Step the given .vxa files as one batch on one GPU; returns (records [n, RECORD_LEN], counters).  `loader(eng)`, when given,
    adds the robots instead (the in-memory hand-off); `paths` then only says how many there are."""
    records = np.zeros((len(paths), RECORD_LEN), dtype=np.float64)
    if not paths:
        return records, None
    device = [device_index] * PIPELINE_ENGINES if (loader is not None and PIPELINE_ENGINES > 1) else device_index
    with engine_module.Engine(variant, device) as eng:
        for key, val in (options or {}).items():
            eng.set_option(key, val)
        if loader is not None:
            loader(eng)
        elif hasattr(eng, "add_vxa_files"):
            eng.add_vxa_files(list(paths))       # parsed and built on all host cores
        else:
            for path in paths:
                eng.add_vxa_file(path)
        eng.run()
        for i in range(len(paths)):
            res = eng.result(i)
            records[i] = result_to_record(res)
            if write_xml and res.status == engine_module.ROBOT_FINISHED and eng.fitness_file_name(i):
                eng.write_result_xml(i)
        counters = eng.counters()
    return records, counters


def run_population(engine_module, paths, variant=0, costs=None, options=None, write_xml=True, device=None, make_loader=None):
    """Evaluate every .vxa of a generation; with an initialised process group the files are sharded over ranks.

    Returns the full record table (identical on every rank).  `device`: HIP device index of this process; by default the
    launcher's LOCAL_RANK in a multi-GPU job, else torch's current device when torch is loaded and CUDA-initialised, else 0
    (the single-GPU path needs no torch at all).  `make_loader(indices)` -> loader(eng): the in-memory hand-off of this rank's robots.
    """
    dist = _dist()
    if dist:
        world, rank = dist.get_world_size(), dist.get_rank()
        shards = shard_by_cost(costs, world) if costs is not None else shard_round_robin(len(paths), world)
        mine = shards[rank]
        largest = max(len(sh) for sh in shards)
    else:
        mine = list(range(len(paths)))
        largest = len(paths)
    torch = sys.modules.get("torch")
    cuda_ready = torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized()
    if device is None:
        if dist and "LOCAL_RANK" in os.environ and not (cuda_ready and torch.cuda.current_device() != 0):
            # one process per GPU: a launcher (torch.distributed.run) exports LOCAL_RANK; honour it unless the caller has
            # selected a device itself, instead of piling every rank onto device 0
            # (the device count from the engine's own library: a CPU-only torch with gloo on a multi-GPU node reports none)
            n_dev = engine_module.device_count() if hasattr(engine_module, "device_count") else 0
            if n_dev <= 0:
                n_dev = torch.cuda.device_count() if (torch is not None and torch.cuda.is_available()) else 0
            device = int(os.environ["LOCAL_RANK"]) % n_dev if n_dev > 0 else 0
        else:
            device = torch.cuda.current_device() if cuda_ready else 0
    records, _ = run_shard(engine_module, [paths[i] for i in mine], variant, device, options, write_xml,
                           make_loader(mine) if make_loader else None)
    gather_device = torch.device("cuda", device) if (dist and dist.get_backend() == "nccl") else None
    return gather_records(records, mine, len(paths), gather_device, per_rank=largest)
