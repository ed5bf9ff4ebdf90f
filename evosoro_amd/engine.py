"""ctypes binding of the C ABI in include/vxhip.h (libvxhip.so, built in-tree by evosoro_amd/csrc/Makefile).

There is no fallback: if the shared library is missing or no HIP device is usable this module raises —
the product path never routes through a CPU implementation.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvxhip.so")
CLI_PATH = os.path.join(_HERE, "voxelyze")

VOXCAD, VOXCAD_LAND_WATER = 0, 1
ROBOT_PENDING, ROBOT_FINISHED, ROBOT_DIVERGED, ROBOT_EMPTY, ROBOT_COL_OVERFLOW = 0, 1, 2, 3, 4

EXPORTS = ["vxh_inspect_vxa_buffer", "vxh_inspect_constants", "vxh_inspect_angle_excess", "vxh_get_angle_excess", "vxh_get_mesh", "vxh_get_shape_descriptors", "vxh_plan_tiles_buffer", "vxh_convex_hull_volume", "vxh_create", "vxh_create_multi", "vxh_destroy", "vxh_add_vxa_file", "vxh_add_vxa_buffer", "vxh_add_vxa_files", "vxh_add_robots", "vxh_num_robots", "vxh_robot_dims", "vxh_voxel_actuation",
           "vxh_run", "vxh_step", "vxh_reset", "vxh_clear", "vxh_get_result", "vxh_write_result_xml",
           "vxh_fitness_file_name", "vxh_get_state", "vxh_get_cm_trace", "vxh_get_counters", "vxh_count_bond_modes", "vxh_set_option", "vxh_strerror",
           "vxh_last_error", "vxh_version", "vxh_device_count"]


class VxhResult(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int), ("steps", ctypes.c_int), ("nvox", ctypes.c_int), ("nbond", ctypes.c_int),
                ("dt", ctypes.c_double), ("cur_time", ctypes.c_double), ("lifetime", ctypes.c_double),
                ("ini_cm", ctypes.c_double * 3), ("cur_cm", ctypes.c_double * 3),
                ("norm_final_dist", ctypes.c_double), ("norm_regime_dist", ctypes.c_double),
                ("norm_frozen_dist", ctypes.c_double), ("final_dist", ctypes.c_double),
                ("final_dist_y", ctypes.c_double), ("anterior_dist", ctypes.c_double),
                ("posterior_dist", ctypes.c_double), ("anterior_y", ctypes.c_double),
                ("posterior_y", ctypes.c_double), ("end_of_life_posterior_y", ctypes.c_double),
                ("fall_adj_post_y", ctypes.c_double), ("num_non_feet_touching_floor", ctypes.c_double),
                ("num_touching_floor", ctypes.c_double), ("norm_abs_disp", ctypes.c_double),
                ("norm_dist_x", ctypes.c_double), ("norm_dist_y", ctypes.c_double), ("norm_dist_z", ctypes.c_double),
                ("robot_volume_start", ctypes.c_double), ("robot_volume_end", ctypes.c_double),
                ("col_rebuilds", ctypes.c_int), ("reserved", ctypes.c_int),
                ("hull_volume_start", ctypes.c_double), ("hull_volume_end", ctypes.c_double)]

    def as_dict(self):
        out = {}
        for name, ctype in self._fields_:
            val = getattr(self, name)
            out[name] = list(val) if hasattr(val, "__len__") else val
        return out


class VxhCounters(ctypes.Structure):
    _fields_ = [("voxel_steps", ctypes.c_double), ("bond_steps", ctypes.c_double),
                ("algorithmic_bytes", ctypes.c_double), ("kernel_seconds", ctypes.c_double),
                ("run_seconds", ctypes.c_double), ("launches", ctypes.c_longlong), ("max_steps", ctypes.c_longlong),
                ("dominant_block", ctypes.c_int), ("dominant_robots", ctypes.c_int),
                ("dominant_launches", ctypes.c_longlong), ("dominant_seconds", ctypes.c_double),
                ("dominant_alg_bytes", ctypes.c_double), ("dominant_voxel_steps", ctypes.c_double)]


class VxhModelInfo(ctypes.Structure):
    _fields_ = [("nvox", ctypes.c_int), ("nbond", ctypes.c_int), ("nsurf", ctypes.c_int),
                ("n_vox_classes", ctypes.c_int), ("n_bond_classes", ctypes.c_int), ("reserved", ctypes.c_int),
                ("opt_dt", ctypes.c_double), ("dt", ctypes.c_double), ("planned_steps", ctypes.c_longlong),
                ("alg_bytes_per_step", ctypes.c_double)]


class VxhTilingInfo(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int), ("kx", ctypes.c_int), ("ky", ctypes.c_int), ("kz", ctypes.c_int),
                ("max_own", ctypes.c_int), ("max_local", ctypes.c_int), ("max_bonds", ctypes.c_int),
                ("total_bonds", ctypes.c_int)]


class VxhRobotArrays(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int), ("ny", ctypes.c_int), ("nz", ctypes.c_int),
                ("material", ctypes.POINTER(ctypes.c_ubyte)), ("n_layers", ctypes.c_int),
                ("layer_tags", ctypes.POINTER(ctypes.c_char_p)), ("layers", ctypes.POINTER(ctypes.POINTER(ctypes.c_double))),
                ("fitness_file_name", ctypes.c_char_p)]


class VxhError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, "libvxhip status %d: %s" % (status, message))
        self.status = status


def build(force=False):
    """Compile libvxhip.so + voxelyze with hipcc for gfx950 (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src))
    newest = max(newest, os.path.getmtime(os.path.join(_HERE, "..", "include", "vxhip.h")))
    dev_lib = os.path.join(_HERE, "libvxhip_prof.so")     # developer library (phase timers, the opt-in pair kernel): what the A/B tests and scripts load
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(CLI_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", src, "-j2", "all", "prof"], stdout=subprocess.DEVNULL)
    elif not os.path.exists(dev_lib) or os.path.getmtime(dev_lib) < newest:
        subprocess.check_call(["make", "-C", src, "prof"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load_library():
    """dlopen libvxhip.so and declare the prototypes; raises OSError if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libvxhip.so not found at %s: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    P, I, D, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_longlong
    lib.vxh_inspect_vxa_buffer.argtypes = [ctypes.c_char_p, ctypes.c_size_t, I, ctypes.POINTER(VxhModelInfo),
                                           ctypes.c_char_p, ctypes.c_size_t]
    lib.vxh_plan_tiles_buffer.argtypes = [ctypes.c_char_p, ctypes.c_size_t, I, I, ctypes.POINTER(VxhTilingInfo),
                                          ctypes.POINTER(I), I, ctypes.c_char_p, ctypes.c_size_t]
    if hasattr(lib, "vxh_inspect_constants"):     # (absent from libraries of earlier rounds, which scripts/ab_lib.py loads for same-box comparisons)
        lib.vxh_inspect_constants.argtypes = [ctypes.c_char_p, ctypes.c_size_t, I, P, I, P, I, ctypes.c_char_p, ctypes.c_size_t]
        lib.vxh_inspect_angle_excess.argtypes = [ctypes.c_char_p, ctypes.c_size_t, I, P, I, ctypes.POINTER(I), ctypes.c_char_p, ctypes.c_size_t]
        lib.vxh_get_angle_excess.argtypes = [P, I, I, P, I, ctypes.POINTER(I)]
    if hasattr(lib, "vxh_get_mesh"):
        lib.vxh_get_mesh.argtypes = [P, I, I, P, I, ctypes.POINTER(I), P, I, ctypes.POINTER(I)]
        lib.vxh_get_shape_descriptors.argtypes = [P, I, I, ctypes.POINTER(D), ctypes.POINTER(D), ctypes.POINTER(D)]
    lib.vxh_convex_hull_volume.argtypes = [ctypes.POINTER(ctypes.c_double), I]
    lib.vxh_convex_hull_volume.restype = D
    lib.vxh_create.argtypes = [ctypes.POINTER(P), I, I]
    lib.vxh_create_multi.argtypes = [ctypes.POINTER(P), I, ctypes.POINTER(I), I]
    lib.vxh_destroy.argtypes = [P]
    lib.vxh_destroy.restype = None
    lib.vxh_add_vxa_file.argtypes = [P, ctypes.c_char_p, ctypes.POINTER(I)]
    lib.vxh_add_vxa_buffer.argtypes = [P, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(I)]
    lib.vxh_add_vxa_files.argtypes = [P, ctypes.POINTER(ctypes.c_char_p), I, ctypes.POINTER(I)]
    lib.vxh_add_robots.argtypes = [P, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(VxhRobotArrays), I, I, ctypes.POINTER(I)]
    lib.vxh_num_robots.argtypes = [P]
    lib.vxh_robot_dims.argtypes = [P, I, ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(D), ctypes.POINTER(LL)]
    lib.vxh_voxel_actuation.argtypes = [P, I, I, ctypes.POINTER(D), ctypes.POINTER(D), ctypes.POINTER(D)]
    lib.vxh_run.argtypes = [P]
    lib.vxh_step.argtypes = [P, LL]
    lib.vxh_reset.argtypes = [P]
    lib.vxh_clear.argtypes = [P]
    lib.vxh_get_result.argtypes = [P, I, ctypes.POINTER(VxhResult)]
    lib.vxh_write_result_xml.argtypes = [P, I, ctypes.c_char_p]
    lib.vxh_fitness_file_name.argtypes = [P, I, ctypes.c_char_p, ctypes.c_size_t]
    lib.vxh_get_state.argtypes = [P, I, P, I]
    lib.vxh_get_counters.argtypes = [P, ctypes.POINTER(VxhCounters)]
    lib.vxh_get_cm_trace.argtypes = [P, I, P, I, ctypes.POINTER(I)]
    lib.vxh_count_bond_modes.argtypes = [P, ctypes.POINTER(LL), ctypes.POINTER(LL)]
    lib.vxh_set_option.argtypes = [P, ctypes.c_char_p, D]
    lib.vxh_strerror.argtypes = [I]
    lib.vxh_strerror.restype = ctypes.c_char_p
    lib.vxh_last_error.argtypes = [P]
    lib.vxh_last_error.restype = ctypes.c_char_p
    lib.vxh_version.restype = ctypes.c_char_p
    _lib = lib
    return lib


def device_count():
    """HIP devices the library can use (0 without a GPU)"""
    return int(load_library().vxh_device_count())


def inspect_vxa(text_or_path, variant=VOXCAD):
    """Host-only model summary of a .vxa (works without a GPU): counts, dt, planned step count."""
    lib = load_library()
    if os.path.exists(text_or_path):
        with open(text_or_path, "rb") as handle:
            raw = handle.read()
    else:
        raw = text_or_path.encode("latin-1") if isinstance(text_or_path, str) else text_or_path
    info, err = VxhModelInfo(), ctypes.create_string_buffer(512)
    rc = lib.vxh_inspect_vxa_buffer(raw, len(raw), variant, ctypes.byref(info), err, len(err))
    if rc != 0:
        raise VxhError(rc, "%s (%s)" % (lib.vxh_strerror(rc).decode(), err.value.decode()))
    return info


def inspect_constants(text_or_path, variant=VOXCAD):
    """Host-only: ([nvox, 12], [nbond, 23]) constants of every voxel and bond as the import computes them (include/vxhip.h)."""
    lib = load_library()
    if os.path.exists(text_or_path):
        with open(text_or_path, "rb") as handle:
            raw = handle.read()
    else:
        raw = text_or_path.encode("latin-1") if isinstance(text_or_path, str) else text_or_path
    info = inspect_vxa(raw, variant)
    vox, bond = np.zeros((max(info.nvox, 1), 12)), np.zeros((max(info.nbond, 1), 23))
    err = ctypes.create_string_buffer(512)
    rc = lib.vxh_inspect_constants(raw, len(raw), variant, vox.ctypes.data, info.nvox, bond.ctypes.data, info.nbond, err, len(err))
    if rc != 0:
        raise VxhError(rc, "%s (%s)" % (lib.vxh_strerror(rc).decode(), err.value.decode()))
    return vox[:info.nvox], bond[:info.nbond]


def inspect_angle_excess(text_or_path, variant=VOXCAD_LAND_WATER):
    """Host-only: per-vertex angle excess of the rest-state surface mesh of a land_water .vxa (include/vxhip.h)."""
    lib = load_library()
    if os.path.exists(text_or_path):
        with open(text_or_path, "rb") as handle:
            raw = handle.read()
    else:
        raw = text_or_path.encode("latin-1") if isinstance(text_or_path, str) else text_or_path
    count, err = ctypes.c_int(), ctypes.create_string_buffer(512)
    rc = lib.vxh_inspect_angle_excess(raw, len(raw), variant, None, 0, ctypes.byref(count), err, len(err))
    if rc != 0:
        raise VxhError(rc, "%s (%s)" % (lib.vxh_strerror(rc).decode(), err.value.decode()))
    out = np.zeros(max(count.value, 1))
    lib.vxh_inspect_angle_excess(raw, len(raw), variant, out.ctypes.data, count.value, ctypes.byref(count), err, len(err))
    return out[:count.value]


def convex_hull_volume(points):
    """Host-only: volume of the convex hull of an [n, 3] point set (what <ConvexHullVolume*> is computed with)."""
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
    return load_library().vxh_convex_hull_volume(pts.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(pts))


def plan_tiles(text_or_path, k_request, variant=VOXCAD):
    """Host-only: how the engine would cut the robot into tiles; returns (VxhTilingInfo, owner tile of every voxel)."""
    lib = load_library()
    if os.path.exists(text_or_path):
        with open(text_or_path, "rb") as handle:
            raw = handle.read()
    else:
        raw = text_or_path.encode("latin-1") if isinstance(text_or_path, str) else text_or_path
    nvox = inspect_vxa(raw, variant).nvox
    info, err = VxhTilingInfo(), ctypes.create_string_buffer(512)
    owner = (ctypes.c_int * max(nvox, 1))()
    rc = lib.vxh_plan_tiles_buffer(raw, len(raw), variant, k_request, ctypes.byref(info), owner, nvox, err, len(err))
    if rc != 0:
        raise VxhError(rc, "%s (%s)" % (lib.vxh_strerror(rc).decode(), err.value.decode()))
    return info, np.array(owner[:nvox], dtype=np.int64)


class Engine(object):
    """One population shard on one GPU: add .vxa robots, run them all at once, read results."""

    def __init__(self, variant=VOXCAD, device=0):
        """device: one HIP device index, or a sequence of them (one handle over several GPUs: the batch is partitioned by cost)"""
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        devices = [int(device)] if np.isscalar(device) else [int(d) for d in device]
        rc = self._lib.vxh_create_multi(ctypes.byref(self._h), variant, (ctypes.c_int * len(devices))(*devices), len(devices))
        if rc != 0:
            self._h = ctypes.c_void_p()
            raise VxhError(rc, self._lib.vxh_strerror(rc).decode())
        self.variant = variant

    def _check(self, rc):
        if rc != 0:
            raise VxhError(rc, "%s (%s)" % (self._lib.vxh_strerror(rc).decode(),
                                            self._lib.vxh_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.vxh_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def add_vxa_file(self, path):
        idx = ctypes.c_int(-1)
        self._check(self._lib.vxh_add_vxa_file(self._h, os.fsencode(path), ctypes.byref(idx)))
        return idx.value

    def add_vxa_files(self, paths):
        """a whole generation: parsed and built on all host cores, appended in order; returns the first robot index"""
        arr = (ctypes.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        idx = ctypes.c_int(-1)
        self._check(self._lib.vxh_add_vxa_files(self._h, arr, len(paths), ctypes.byref(idx)))
        return idx.value

    def add_robots(self, template_text, robots, round_like_text=True):
        """A generation from arrays (vxh_add_robots).  `robots`: sequence of (material, layers, fitness_file_name) with material an
        integer array [x, y, z] as the genotype holds it, layers an ordered mapping tag -> float array [x, y, z] (tags with or without
        angle brackets), fitness_file_name a str or None.  Returns the first robot index."""
        raw = template_text.encode("latin-1") if isinstance(template_text, str) else template_text
        n = len(robots)
        descs = (VxhRobotArrays * max(n, 1))()
        keep = []                                         # buffers must outlive the call
        for i, (material, layers, name) in enumerate(robots):
            mat = np.ascontiguousarray(np.asarray(material).transpose(2, 1, 0), dtype=np.uint8)     # -> z slowest, x fastest
            tags = (ctypes.c_char_p * max(len(layers), 1))(*[t.strip("<>").encode() for t in layers])
            arrs = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).transpose(2, 1, 0)) for a in layers.values()]
            ptrs = (ctypes.POINTER(ctypes.c_double) * max(len(arrs), 1))(*[a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) for a in arrs])
            keep += [mat, tags, arrs, ptrs]
            d = descs[i]
            d.nx, d.ny, d.nz = (int(v) for v in np.asarray(material).shape)
            d.material = mat.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))
            d.n_layers, d.layer_tags, d.layers = len(arrs), tags, ptrs
            d.fitness_file_name = None if name is None else os.fsencode(name)
        idx = ctypes.c_int(-1)
        self._check(self._lib.vxh_add_robots(self._h, raw, len(raw), descs, n, 1 if round_like_text else 0, ctypes.byref(idx)))
        return idx.value

    def add_vxa_text(self, text):
        raw = text.encode("latin-1") if isinstance(text, str) else text
        idx = ctypes.c_int(-1)
        self._check(self._lib.vxh_add_vxa_buffer(self._h, raw, len(raw), ctypes.byref(idx)))
        return idx.value

    def num_robots(self):
        return self._lib.vxh_num_robots(self._h)

    def dims(self, robot):
        nvox, nbond, dt, steps = ctypes.c_int(), ctypes.c_int(), ctypes.c_double(), ctypes.c_longlong()
        self._check(self._lib.vxh_robot_dims(self._h, robot, ctypes.byref(nvox), ctypes.byref(nbond),
                                             ctypes.byref(dt), ctypes.byref(steps)))
        return {"nvox": nvox.value, "nbond": nbond.value, "dt": dt.value, "planned_steps": steps.value}

    def voxel_actuation(self, robot, voxel):
        """(TempAmplitude, TempPeriod, phaseOffset) of one voxel as the reference's float members hold them (what `voxelyze -p` prints)"""
        a, p, ph = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._check(self._lib.vxh_voxel_actuation(self._h, robot, voxel, ctypes.byref(a), ctypes.byref(p), ctypes.byref(ph)))
        return a.value, p.value, ph.value

    def run(self):
        self._check(self._lib.vxh_run(self._h))

    def step(self, n):
        self._check(self._lib.vxh_step(self._h, n))

    def reset(self):
        self._check(self._lib.vxh_reset(self._h))

    def clear(self):
        self._check(self._lib.vxh_clear(self._h))

    def result(self, robot):
        out = VxhResult()
        self._check(self._lib.vxh_get_result(self._h, robot, ctypes.byref(out)))
        return out

    def write_result_xml(self, robot, path=None):
        self._check(self._lib.vxh_write_result_xml(self._h, robot, None if path is None else os.fsencode(path)))

    def fitness_file_name(self, robot):
        buf = ctypes.create_string_buffer(4096)
        self._check(self._lib.vxh_fitness_file_name(self._h, robot, buf, len(buf)))
        return buf.value.decode()

    def state(self, robot):
        n = self.dims(robot)["nvox"]
        out = np.zeros((n, 14), dtype=np.float64)
        self._check(self._lib.vxh_get_state(self._h, robot, out.ctypes.data, n))
        return out

    def counters(self):
        out = VxhCounters()
        self._check(self._lib.vxh_get_counters(self._h, ctypes.byref(out)))
        return out

    def cm_trace(self, robot):
        """[n, 4] (time, x, y, z): the centre-of-mass trace of a robot with <TimeBetweenTraces> > 0"""
        count = ctypes.c_int()
        self._check(self._lib.vxh_get_cm_trace(self._h, robot, None, 0, ctypes.byref(count)))
        out = np.zeros((max(count.value, 1), 4), dtype=np.float64)
        self._check(self._lib.vxh_get_cm_trace(self._h, robot, out.ctypes.data, count.value, ctypes.byref(count)))
        return out[:count.value]

    def angle_excess(self, robot, at_end=True):
        """per-vertex angle excess of a land_water robot's surface mesh: rest state (at_end False) or current state"""
        count = ctypes.c_int()
        self._check(self._lib.vxh_get_angle_excess(self._h, robot, 1 if at_end else 0, None, 0, ctypes.byref(count)))
        out = np.zeros(max(count.value, 1))
        self._check(self._lib.vxh_get_angle_excess(self._h, robot, 1 if at_end else 0, out.ctypes.data, count.value, ctypes.byref(count)))
        return out[:count.value]

    def mesh(self, robot, at_end=True):
        """(vertices [n, 3], facets [m, 3]) of the robot's deformable surface mesh (land_water robots; _voxcad robots added under the option
        shape_descriptors): rest state or current state, in the reference's vertex and facet order"""
        nv, nf = ctypes.c_int(), ctypes.c_int()
        self._check(self._lib.vxh_get_mesh(self._h, robot, 1 if at_end else 0, None, 0, ctypes.byref(nv), None, 0, ctypes.byref(nf)))
        verts, facets = np.zeros((max(nv.value, 1), 3)), np.zeros((max(nf.value, 1), 3), dtype=np.int32)
        self._check(self._lib.vxh_get_mesh(self._h, robot, 1 if at_end else 0, verts.ctypes.data, nv.value, ctypes.byref(nv), facets.ctypes.data, nf.value, ctypes.byref(nf)))
        return verts[:nv.value], facets[:nf.value]

    def shape_descriptors(self, robot, at_end=True):
        """(robot volume, convex-hull volume, shape complexity as the reference binary prints it) of that mesh; -1 each without a mesh"""
        vol, hull, cplx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._check(self._lib.vxh_get_shape_descriptors(self._h, robot, 1 if at_end else 0, ctypes.byref(vol), ctypes.byref(hull), ctypes.byref(cplx)))
        return vol.value, hull.value, cplx.value

    def bond_modes(self):
        """(bonds in the large-angle branch, bonds) of the whole batch right now"""
        large, total = ctypes.c_longlong(), ctypes.c_longlong()
        self._check(self._lib.vxh_count_bond_modes(self._h, ctypes.byref(large), ctypes.byref(total)))
        return large.value, total.value

    def set_option(self, key, value):
        self._check(self._lib.vxh_set_option(self._h, key.encode(), float(value)))

    def version(self):
        return self._lib.vxh_version().decode()
