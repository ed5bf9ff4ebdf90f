"""evosoro_amd — MI355X-native batched Voxelyze time-stepper behind evosoro's evaluation boundary.

Layout (only what the hot path needs):
  base.py                      Sim / Env / ObjectiveDict parameter containers (reference: evosoro/base.py)
  tools/read_write_voxelyze.py .vxa writer + results reader            (reference: evosoro/tools/read_write_voxelyze.py)
  tools/evaluation.py          evaluate_all(): one batched engine call  (reference: evosoro/tools/evaluation.py)
  engine.py                    ctypes binding of the C ABI in include/vxhip.h (libvxhip.so)
  parallel.py                  population sharding + RCCL fitness gather (one process per GPU)
  csrc/                        C++ host code (.vxa parser, model builder, results writer, CLI) + HIP kernels
"""
__version__ = "0.1.0"
