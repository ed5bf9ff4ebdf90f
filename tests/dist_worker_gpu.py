"""Worker of tests/test_gpu_parity.py::test_two_ranks_share_the_gpu: one rank of a 2-rank gloo job, both ranks stepping
their shard with the REAL engine on cuda:0 (RCCL refuses two ranks on one device, gloo does not care)."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evosoro_amd import engine, parallel  # noqa: E402


def main():
    out_dir, vxa_dir = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    names = ["phase4", "soft5_init0", "rand6_nocol", "probe6", "stiff5", "grow5", "rand6_col"]
    paths = [os.path.join(vxa_dir, n + ".vxa") for n in names]
    costs = [57 * 742, 96 * 781, 143 * 3123, 150 * 7806, 81 * 300, 88 * 1500, 150 * 3900]
    table = parallel.run_population(engine, paths, variant=0, costs=costs, write_xml=False)
    np.save(os.path.join(out_dir, "table_rank%d.npy" % dist.get_rank()), table)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
