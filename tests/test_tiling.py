"""Host-side checks of the tile planner behind the multi-workgroup kernel (vxh_plan_tiles_buffer; no GPU needed):
every voxel owned by exactly one tile, balanced tiles, and the known answer for BASELINE configs[4] (a full 20x20x20
lattice cut into 125 cubes of 4x4x4)."""
import os

import numpy as np

from evosoro_amd import engine


def _vxa(tmp_path, material, name="t"):
    from evosoro_amd import workloads
    from evosoro_amd.base import Sim, Env
    from evosoro_amd.tools.read_write_voxelyze import write_voxelyze_file
    os.makedirs(tmp_path / "voxelyzeFiles", exist_ok=True)
    ind = workloads.make_individual(0, material)
    write_voxelyze_file(Sim(dt_frac=0.9, simulation_time=0.01, fitness_eval_init_time=0.002), Env(), ind, str(tmp_path), name)
    return str(tmp_path / "voxelyzeFiles" / (name + "--id_00000.vxa"))


def test_full_lattice_is_cut_into_cubes(tmp_path):
    from evosoro_amd import workloads
    path = _vxa(tmp_path, workloads.full_material(20, 1))
    info, owner = engine.plan_tiles(path, 125)
    assert (info.k, info.kx, info.ky, info.kz) == (125, 5, 5, 5)
    assert info.max_own == 64 and np.bincount(owner, minlength=125).tolist() == [64] * 125
    # an interior 4x4x4 cube: 144 bonds inside, 96 leaving it (listed by both sides), 96 mirrored voxels
    assert info.max_bonds == 240 and info.max_local == 160
    # every bond once, plus once more for each one that crosses a boundary: 3 * 20^2 * 19 + 3 * 4 * 400
    assert info.total_bonds == 22800 + 4800
    # the owner of a voxel is the cube it lies in (voxel order: x fastest)
    v = np.arange(8000)
    x, y, z = v % 20, (v // 20) % 20, v // 400
    assert np.array_equal(owner, ((x // 4) * 5 + (y // 4)) * 5 + (z // 4))


def test_random_robots_every_voxel_owned_once_and_balanced(tmp_path):
    from evosoro_amd import workloads
    for seed, shape in ((3, (6, 6, 6)), (4, (10, 10, 10)), (5, (8, 8, 8))):
        path = _vxa(tmp_path, workloads.random_material(shape, seed), "r%d" % seed)
        nvox = engine.inspect_vxa(path).nvox
        for k in (1, 2, 3, 4, 7, 8, 16):
            info, owner = engine.plan_tiles(path, k)
            assert 1 <= info.k <= k and info.k >= (4 * k) // 5         # k or somewhat fewer (a grid that factors well)
            assert info.kx * info.ky * info.kz == info.k
            assert owner.shape == (nvox,) and owner.min() >= 0 and owner.max() < info.k
            counts = np.bincount(owner, minlength=info.k)
            assert counts.sum() == nvox and counts.max() == info.max_own
            # cuts at equal counts along x, then y, then z: tiles differ by at most one voxel per cut level
            assert counts.max() - counts.min() <= 3, (seed, k, counts)
            assert info.max_local >= info.max_own and (info.k == 1) == (info.max_local == info.max_own)
    # one tile = the whole robot, every bond once
    info, owner = engine.plan_tiles(path, 1)
    assert info.k == 1 and info.total_bonds == engine.inspect_vxa(path).nbond and not owner.any()
