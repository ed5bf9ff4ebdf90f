"""Runs bench.py with the engine replaced by the CPU test double (tests/stub_engine.py) and torch's CUDA calls made no-ops:
the N > 1 code paths of bench.py -- self-launch under torch.distributed.run, weak + strong populations, the fitness gather,
the one-handle route -- walked on a box without a GPU (tests/test_bench_cli.py).  Test infrastructure only."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import torch  # noqa: E402
import stub_engine  # noqa: E402
import evosoro_amd  # noqa: E402
from evosoro_amd import engine as real_engine  # noqa: E402

stub_engine._real = real_engine
sys.modules["evosoro_amd.engine"] = stub_engine
evosoro_amd.engine = stub_engine
torch.cuda.set_device = lambda *a, **k: None
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.device_count = lambda: int(os.environ.get("VXH_STUB_GPUS", "0"))
torch.cuda.is_available = lambda: int(os.environ.get("VXH_STUB_GPUS", "0")) > 0
os.environ["VXH_BENCH_ENTRY"] = os.path.abspath(__file__)
sys.argv[0] = os.path.join(REPO, "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
