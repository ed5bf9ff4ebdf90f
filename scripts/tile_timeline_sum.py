"""Medians of the per-tile step timeline printed by `VXH_PROF_TILES=1 python scripts/dev_gpu_diag.py tilelong prof` (developer library):
    python scripts/tile_timeline_sum.py < output
columns of a `tile N:` line: top, halo-done, bond-done, voxel-done, C-passed, svc barrier-resolved, next top, mv-published (us)."""
import sys, statistics
runs, cur = [], []
for line in sys.stdin:
    if line.startswith("tile: top"):
        if cur: runs.append(cur)
        cur = []
    elif line.startswith("tile ") and ":" in line:
        v = [float(x) for x in line.split(":")[1].split()]
        if len(v) == 8: cur.append(v)
if cur: runs.append(cur)
for r in runs:
    med = lambda f: statistics.median(f(v) for v in r)
    print("%3d tiles | top->halo %.2f | bond %.2f | voxel %.2f | voxel-done->C %.2f | C->next top %.2f | period %.2f | svc resolved - voxel-done %.2f | mv out - voxel-done %.2f | spread of tops %.2f" % (
        len(r), med(lambda v: v[1] - v[0]), med(lambda v: v[2] - v[1]), med(lambda v: v[3] - v[2]), med(lambda v: v[4] - v[3]), med(lambda v: v[6] - v[4]),
        med(lambda v: v[6] - v[0]), med(lambda v: v[5] - v[3]), med(lambda v: v[7] - v[3]), max(v[0] for v in r) - min(v[0] for v in r)))
