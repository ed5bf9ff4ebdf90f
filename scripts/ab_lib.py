"""Same-box A/B of two builds of the library: run a mode of scripts/dev_gpu_diag.py against another library file.
    python scripts/ab_lib.py <library file name under evosoro_amd/> <dev_gpu_diag mode>
Build the other version under a second name (e.g. `git stash; make; cp evosoro_amd/libvxhip.so evosoro_amd/libvxhip_head.so; git stash
pop; make`), then in ONE gpurun call alternate the two, twice.  Timings taken in different gpurun calls come from different boxes and
differ by 1 % on the bench population -- and a 'baseline' from another call once sent an afternoon after a regression's wrong cause."""
import os, sys, runpy
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from evosoro_amd import engine
engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), sys.argv[1])
sys.argv = ["dev_gpu_diag.py"] + sys.argv[2:]
runpy.run_path(os.path.join(HERE, "dev_gpu_diag.py"), run_name="__main__")
