import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from evosoro_amd import engine as E
E.LIB_PATH = os.path.join(os.path.dirname(E.LIB_PATH), "libvxhip_prof.so")
DBG = int(sys.argv[2])
g = "/root/repo/tests/golden"
names = sys.argv[1].split(",")
def run():
    out = []
    with E.Engine(E.VOXCAD, 0) as eng:
        eng.set_option("tiled", 0)
        eng.set_option("dbg", DBG)
        for n in names: eng.add_vxa_file(os.path.join(g, "vxa", n + ".vxa"))
        for k in (1, 332, 2000, 100000):
            eng.step(k)
            out.append([eng.state(i) for i in range(len(names))])
    return out
a, b = run(), run()
for c in range(len(a)):
    for i, n in enumerate(names):
        if not np.array_equal(a[c][i], b[c][i]): print("DIFF checkpoint", c, n, np.abs(a[c][i] - b[c][i]).max())
print("done")
