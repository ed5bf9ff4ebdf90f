"""Synthetic robots for benchmarks and parity tests (BASELINE.json configs).

The evolutionary algorithm of the reference (CPPN genotypes, softbot.py) is out of scope; what the
evaluation boundary consumes is only the duck-typed surface used by write_voxelyze_file/evaluate_all
(evosoro/tools/read_write_voxelyze.py:45-49,345-399; evosoro/tools/evaluation.py:59-69,166-177):
  individual.id, individual.genotype.orig_size_xyz, individual.genotype.to_phenotype_mapping
  (name -> {"tag","env_kws","params","param_tags","output_type","state"}), individual.phenotype.is_valid().
This module builds such individuals from plain numpy arrays.
"""
from collections import OrderedDict, deque

import numpy as np


class _Bag(object):
    pass


def make_individual(ident, material, per_voxel=None):
    """Wrap a material array [x, y, z] (ints 0..7) and optional per-voxel float layers into an individual.

    per_voxel: OrderedDict tag -> float array [x, y, z], e.g. {"<PhaseOffset>": phase}.
    """
    material = np.asarray(material)
    ind = _Bag()
    ind.id = int(ident)
    ind.genotype = _Bag()
    ind.genotype.orig_size_xyz = tuple(int(n) for n in material.shape)
    mapping = OrderedDict()
    mapping["material"] = {"tag": "<Data>", "env_kws": None, "params": None, "param_tags": None,
                           "output_type": int, "state": material}
    for tag, state in (per_voxel or {}).items():
        mapping[tag.strip("<>")] = {"tag": tag, "env_kws": None, "params": None, "param_tags": None,
                                    "output_type": float, "state": np.asarray(state, dtype=np.float64)}
    ind.genotype.to_phenotype_mapping = mapping
    ind.phenotype = _Bag()
    ind.phenotype.is_valid = lambda: bool((material > 0).any())
    ind.fitness = None
    ind.md5 = None
    return ind


def largest_component(material):
    """Zero every occupied cell that is not in the largest face-connected component.

    Mirrors the intent of the reference's make_one_shape_only (evosoro/tools/utils.py:199-240): one robot
    per file, no floating debris.  Ties are broken towards the component found first in x-fastest scan order.
    """
    occ = material > 0
    labels = np.zeros(material.shape, dtype=np.int32)
    sizes = [0]
    nx, ny, nz = material.shape
    for z in range(nz):
        for y in range(ny):
            for x in range(nx):
                if not occ[x, y, z] or labels[x, y, z]:
                    continue
                lab = len(sizes)
                labels[x, y, z] = lab
                todo, count = deque([(x, y, z)]), 0
                while todo:
                    cx, cy, cz = todo.popleft()
                    count += 1
                    for dx, dy, dz in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
                        px, py, pz = cx + dx, cy + dy, cz + dz
                        if 0 <= px < nx and 0 <= py < ny and 0 <= pz < nz and occ[px, py, pz] \
                                and not labels[px, py, pz]:
                            labels[px, py, pz] = lab
                            todo.append((px, py, pz))
                sizes.append(count)
    if len(sizes) == 1:
        return material.copy()
    keep = int(np.argmax(sizes[1:])) + 1
    return np.where(labels == keep, material, 0)


def random_material(shape, seed, p_empty=0.3):
    """Material per voxel: 0 with probability p_empty, else uniform on {1,2,3,4}; largest component kept."""
    rng = np.random.RandomState(seed)
    p_mat = (1.0 - p_empty) / 4.0
    mat = rng.choice(5, size=shape, p=[p_empty, p_mat, p_mat, p_mat, p_mat]).astype(np.int64)
    return largest_component(mat)


def random_robot(ident, shape, seed, p_empty=0.3, phase_offset=False):
    mat = random_material(shape, seed, p_empty)
    extra = None
    if phase_offset:
        rng = np.random.RandomState(seed + 100003)
        extra = OrderedDict([("<PhaseOffset>", rng.uniform(-1.0, 1.0, size=shape))])
    return make_individual(ident, mat, extra)


def swimmer(ident, shape, seed, p_empty=0.3):
    """Random robot with a per-voxel <PhaseOffset> layer, as the _voxcad_land_water examples evolve them (BASELINE configs[3]).
    Phases are rounded to 3 decimals: Python 2 (the reference) and Python 3 print such numbers alike, so the .vxa text of the
    golden cases is the same whichever interpreter wrote it."""
    phase = np.round(np.random.RandomState(seed + 100003).uniform(-1.0, 1.0, size=shape), 3)
    return make_individual(ident, random_material(shape, seed, p_empty), OrderedDict([("<PhaseOffset>", phase)]))


def probe_material():
    """The 6x6x6 plumbing robot of SURVEY.md Appendix C: RandomState(1).randint(0,5), floor layer all muscle."""
    mat = np.random.RandomState(1).randint(0, 5, size=(6, 6, 6)).astype(np.int64)
    mat[:, :, 0] = 3
    layer1 = mat[:, :, 1]
    layer1[layer1 == 0] = 1
    return mat


def folded_material(plates=4, side=10):
    """A sheet folded back and forth: `plates` full side x side layers at z = 0, 2, 4, ..., joined along alternating edges -- so that a
    voxel has dozens of others a gap away in space but many bonds away along the sheet, i.e. long collision lists
    (CalcL1Bonds pairs what is within CollisionHorizon voxel sizes and more than 1.5 x CollisionHorizon bonds away, VX_Sim.cpp:2357-2413):
    with <CollisionHorizon> 5 the rows of the inner plates hold well over 64 partners.  Muscle plates (materials 3 / 4 in stripes: they
    actuate in antiphase, the sheet flaps and the plates touch), soft joints."""
    mat = np.zeros((side, side, 2 * plates - 1), dtype=np.int64)
    for k in range(plates):
        mat[:, :, 2 * k] = 3
        mat[::2, :, 2 * k] = 4
        if k + 1 < plates:
            mat[0 if k % 2 == 0 else side - 1, :, 2 * k + 1] = 1
    return mat


def full_material(n, seed=1):
    """Full n^3 lattice with materials {1..4} (the 'large' config uses n=20)."""
    return np.random.RandomState(seed).randint(1, 5, size=(n, n, n)).astype(np.int64)


def population(count, shape, first_seed=0, p_empty=0.3, phase_offset=False):
    return [random_robot(i, shape, first_seed + i, p_empty, phase_offset) for i in range(count)]
