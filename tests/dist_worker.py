"""Worker of tests/test_parallel.py: one rank of a gloo job on CPU."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from evosoro_amd import parallel  # noqa: E402
import stub_engine  # noqa: E402


def main():
    out_dir, vxa_dir = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    names = ["phase4", "soft5_init0", "rand6_nocol", "phase4", "soft5_init0"]
    paths = [os.path.join(vxa_dir, n + ".vxa") for n in names]
    costs = [57 * 742, 96 * 781, 143 * 3123, 57 * 742, 96 * 781]
    table = parallel.run_population(stub_engine, paths, variant=0, costs=costs, write_xml=False)
    np.save(os.path.join(out_dir, "table_rank%d.npy" % rank), table)
    # uneven synthetic shards through the bare collective
    mine = [i for i in range(7) if (i % world == rank) or (rank == 0 and i >= 5)]
    mine = sorted(set(mine)) if rank == 0 else [i for i in mine if i < 5]
    recs = np.array([[float(i)] * parallel.RECORD_LEN for i in mine]).reshape(-1, parallel.RECORD_LEN)
    full = parallel.gather_records(recs, mine, 7)
    assert [int(x) for x in full[:, 0]] == list(range(7)), full[:, 0]
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
