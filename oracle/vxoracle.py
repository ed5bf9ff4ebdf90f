"""TEST INFRASTRUCTURE ONLY (oracle/): Python binding of the CPU restatement (libvxoracle.so) plus an
independent .vxa reader built on xml.etree, and readers for the reference probe traces.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
parses .vxa files with its own C++ reader (evosoro_amd/csrc/vxa_reader.cpp); keeping this reader separate
(different language, different XML library) makes oracle-vs-product comparisons a check of the parsing
too.  Tag names and defaults follow the reference readers:
  Simulator   evosoro/_voxcad/Voxelyze/VX_Sim.cpp:263-354, VX_SimGA.cpp:216-230
  Environment VX_Environment.cpp:123-234 (LW/VX_Environment.cpp:123-200 for the fluid tags)
  VXC         VX_Object.cpp:1064-1073 (lattice), 1344-1441 (materials), 1733-1900 (structure, per-voxel layers)
"""
import ctypes
import os
import struct
import subprocess
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvxoracle.so")


class VxoModel(ctypes.Structure):
    _fields_ = [
        ("variant", ctypes.c_int), ("nx", ctypes.c_int), ("ny", ctypes.c_int), ("nz", ctypes.c_int),
        ("lattice_dim", ctypes.c_double), ("structure", ctypes.POINTER(ctypes.c_ubyte)), ("nmat", ctypes.c_int),
        ("mat_E", ctypes.POINTER(ctypes.c_double)), ("mat_rho", ctypes.POINTER(ctypes.c_double)),
        ("mat_nu", ctypes.POINTER(ctypes.c_double)), ("mat_cte", ctypes.POINTER(ctypes.c_double)),
        ("mat_us", ctypes.POINTER(ctypes.c_double)), ("mat_ud", ctypes.POINTER(ctypes.c_double)),
        ("phase_offset", ctypes.POINTER(ctypes.c_double)), ("temp_amp_damp", ctypes.POINTER(ctypes.c_double)),
        ("stiffness", ctypes.POINTER(ctypes.c_double)),
        ("final_phase_offset", ctypes.POINTER(ctypes.c_double)), ("final_temp_amp_damp", ctypes.POINTER(ctypes.c_double)),
        ("initial_voxel_size", ctypes.POINTER(ctypes.c_double)), ("final_voxel_size", ctypes.POINTER(ctypes.c_double)),
        ("growth_time", ctypes.POINTER(ctypes.c_double)), ("start_growth_time", ctypes.POINTER(ctypes.c_double)),
        ("dt_frac", ctypes.c_double), ("bond_damping_z", ctypes.c_double), ("col_damping_z", ctypes.c_double),
        ("slow_damping_z", ctypes.c_double), ("self_col_enabled", ctypes.c_int), ("col_system", ctypes.c_int),
        ("collision_horizon", ctypes.c_double), ("stop_type", ctypes.c_int), ("stop_value", ctypes.c_double),
        ("afterlife_time", ctypes.c_double), ("midlife_freeze_time", ctypes.c_double),
        ("init_cm_time", ctypes.c_double), ("min_temp_fact", ctypes.c_double),
        ("grav_enabled", ctypes.c_int), ("grav_acc", ctypes.c_double), ("floor_enabled", ctypes.c_int),
        ("temp_enabled", ctypes.c_int), ("temp_amplitude", ctypes.c_double), ("temp_base", ctypes.c_double),
        ("temp_period", ctypes.c_double), ("vary_temp_enabled", ctypes.c_int),
        ("growth_amplitude", ctypes.c_double), ("min_growth_time", ctypes.c_double), ("sticky_floor", ctypes.c_int),
        ("fluid_env", ctypes.c_int), ("aggregate_drag_coef", ctypes.c_double),
        ("time_between_traces", ctypes.c_double),
    ]


class VxoInfo(ctypes.Structure):
    _fields_ = [("nvox", ctypes.c_int), ("nbond", ctypes.c_int), ("nsurf", ctypes.c_int), ("ncol", ctypes.c_int),
                ("steps", ctypes.c_int), ("status", ctypes.c_int), ("cm_initialized", ctypes.c_int),
                ("n_small_angle", ctypes.c_int), ("col_rebuilds", ctypes.c_int), ("reserved", ctypes.c_int), ("opt_dt", ctypes.c_double), ("dt", ctypes.c_double),
                ("cur_time", ctypes.c_double), ("max_vox_vel", ctypes.c_double),
                ("cur_cm", ctypes.c_double * 3), ("ini_cm", ctypes.c_double * 3)]


class VxoResult(ctypes.Structure):
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
                ("norm_dist_x", ctypes.c_double), ("norm_dist_y", ctypes.c_double), ("norm_dist_z", ctypes.c_double)]


def build(force=False):
    """Compile libvxoracle.so (gcc) if missing; building the checker is not using it."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "vx_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None
_libs = {}


def lib(half_angle=False):
    """the restatement; half_angle=True: the instrument build with the engine's half-angle rewrite (oracle/Makefile)"""
    global _lib
    if half_angle not in _libs:
        build()
        _lib = ctypes.CDLL(LIB_PATH.replace("libvxoracle.so", "libvxoracle_ha.so") if half_angle else LIB_PATH)
        _lib.vxo_create.restype = ctypes.c_void_p
        _lib.vxo_create.argtypes = [ctypes.POINTER(VxoModel)]
        _lib.vxo_destroy.argtypes = [ctypes.c_void_p]
        _lib.vxo_step.restype = ctypes.c_long
        _lib.vxo_step.argtypes = [ctypes.c_void_p, ctypes.c_long]
        _lib.vxo_get_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(VxoInfo)]
        _lib.vxo_get_state.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.vxo_get_bond_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _lib.vxo_get_bond_modes.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.vxo_get_constants.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _lib.vxo_jitter.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        if hasattr(_lib, "vxo_set_state"):
            _lib.vxo_set_state.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.vxo_get_result.argtypes = [ctypes.c_void_p, ctypes.POINTER(VxoResult)]
        _lib.vxo_get_cm_trace.restype = ctypes.c_int
        _lib.vxo_get_cm_trace.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
        _lib.vxo_alg_bytes_per_step.restype = ctypes.c_double
        _lib.vxo_alg_bytes_per_step.argtypes = [ctypes.c_void_p]
        _libs[half_angle] = _lib
    return _libs[half_angle]


# ---------------------------------------------------------------------------------------------- .vxa reader
def _atof(text):
    """C atof(): longest numeric prefix, 0.0 if none."""
    text = text.strip()
    for end in range(len(text), 0, -1):
        try:
            return float(text[:end])
        except ValueError:
            continue
    return 0.0


def _atoi(text):
    text = text.strip()
    sign, i = 1, 0
    if text[:1] in "+-":
        sign = -1 if text[0] == "-" else 1
        i = 1
    j = i
    while j < len(text) and text[j].isdigit():
        j += 1
    return sign * int(text[i:j]) if j > i else 0


def _find(parent, tag):
    return None if parent is None else parent.find(tag)


def _num(parent, tag, default, conv=_atof):
    el = _find(parent, tag)
    if el is None or el.text is None:
        return default
    return conv(el.text)


def _flag(parent, tag, default):
    el = _find(parent, tag)
    if el is None or el.text is None:
        return default
    return 1 if _atoi(el.text) != 0 else 0


DEV_LAYERS = (("FinalPhaseOffset", "final_phase_offset"), ("FinalTempAmpDamp", "final_temp_amp_damp"),
              ("InitialVoxelSize", "initial_voxel_size"), ("FinalVoxelSize", "final_voxel_size"),
              ("GrowthTime", "growth_time"), ("StartGrowthTime", "start_growth_time"))


def parse_vxa(path_or_text, variant=0):
    """Parse a .vxa into a plain dict (numpy arrays for the lattice and per-voxel layers)."""
    if os.path.exists(path_or_text):
        with open(path_or_text, "rb") as handle:
            raw = handle.read()
    else:
        raw = path_or_text.encode("latin-1") if isinstance(path_or_text, str) else path_or_text
    root = ET.fromstring(raw)
    sim, env, vxc = root.find("Simulator"), root.find("Environment"), root.find("VXC")
    d = {"variant": variant}

    # Simulator: a present block with an absent tag gets the reader's fallback, an absent block keeps the
    # constructor value (VX_Sim.cpp:18-134)
    integ, damp, col = _find(sim, "Integration"), _find(sim, "Damping"), _find(sim, "Collisions")
    d["dt_frac"] = _num(integ, "DtFrac", 0.9)
    if damp is not None:
        d["bond_damping_z"] = _num(damp, "BondDampingZ", 0.1)
        d["col_damping_z"] = _num(damp, "ColDampingZ", 1.0)
        d["slow_damping_z"] = _num(damp, "SlowDampingZ", 1.0)
    else:
        d["bond_damping_z"], d["col_damping_z"], d["slow_damping_z"] = 0.1, 1.0, 0.001
    if col is not None:
        d["self_col_enabled"] = _flag(col, "SelfColEnabled", 0)
        d["col_system"] = int(_num(col, "ColSystem", 3, _atoi))
        d["collision_horizon"] = _num(col, "CollisionHorizon", 2.0)
    else:
        d["self_col_enabled"], d["col_system"], d["collision_horizon"] = 0, 3, 3.0
    stop = _find(sim, "StopCondition")
    d["stop_type"] = int(_num(stop, "StopConditionType", 0, _atoi))
    d["stop_value"] = _num(stop, "StopConditionValue", 0.0)
    d["afterlife_time"] = _num(stop, "AfterlifeTime", 0.0)
    d["midlife_freeze_time"] = _num(stop, "MidLifeFreezeTime", 0.0)
    d["init_cm_time"] = _num(stop, "InitCmTime", 0.0)
    d["min_temp_fact"] = _num(sim, "MinTempFact", 0.1)
    ga = _find(sim, "GA")
    d["fitness_file_name"] = (ga.findtext("FitnessFileName") or "") if ga is not None else ""

    # Environment
    grav, therm = _find(env, "Gravity"), _find(env, "Thermal")
    d["grav_enabled"] = _flag(grav, "GravEnabled", 0)
    d["grav_acc"] = _num(grav, "GravAcc", -9.81)
    d["floor_enabled"] = _flag(grav, "FloorEnabled", 0)
    d["temp_enabled"] = _flag(therm, "TempEnabled", 0)
    d["temp_base"] = _num(therm, "TempBase", 25.0)
    if _find(therm, "TempAmplitude") is not None:
        d["temp_amplitude"] = _num(therm, "TempAmplitude", 0.0)
    elif _find(therm, "TempAmp") is not None:
        d["temp_amplitude"] = _num(therm, "TempAmp", 0.0) - d["temp_base"]
    else:
        d["temp_amplitude"] = 0.0
    d["vary_temp_enabled"] = _flag(therm, "VaryTempEnabled", 0)
    d["temp_period"] = _num(therm, "TempPeriod", 0.1)
    d["growth_amplitude"] = _num(env, "GrowthAmplitude", 0.0)
    d["min_growth_time"] = _num(env, "MinGrowthTime", 0.0)
    d["sticky_floor"] = _flag(env, "StickyFloor", 0)
    d["fluid_env"] = _flag(env, "FluidEnvironment", 0)
    d["aggregate_drag_coef"] = _num(env, "AggregateDragCoefficient", 0.0)
    d["time_between_traces"] = _num(env, "TimeBetweenTraces", 0.0)       # VX_Environment.cpp:214-215
    d["save_traces"] = _flag(env, "SaveTraces", 0)

    # VXC
    lattice = _find(vxc, "Lattice")
    d["lattice_dim"] = _num(lattice, "Lattice_Dim", 0.001)
    mats = {"E": [0.0], "rho": [0.0], "nu": [0.0], "cte": [0.0], "us": [0.0], "ud": [0.0]}  # index 0 = "Erase"
    palette = _find(vxc, "Palette")
    for mat in ([] if palette is None else palette.findall("Material")):
        mech = mat.find("Mechanical")
        mats["E"].append(_num(mech, "Elastic_Mod", 0.0))
        mats["rho"].append(_num(mech, "Density", 0.0))
        mats["nu"].append(_num(mech, "Poissons_Ratio", 0.0))
        mats["cte"].append(_num(mech, "CTE", 0.0))
        mats["us"].append(_num(mech, "uStatic", 0.0))
        mats["ud"].append(_num(mech, "uDynamic", 0.0))
    d["materials"] = {k: np.array(v, dtype=np.float64) for k, v in mats.items()}
    st = _find(vxc, "Structure")
    if st.get("Compression") != "ASCII_READABLE":
        raise ValueError("only ASCII_READABLE structures are supported (the reference's headless build too)")
    nx, ny, nz = (int(_num(st, t, 1, _atoi)) for t in ("X_Voxels", "Y_Voxels", "Z_Voxels"))
    d["nx"], d["ny"], d["nz"] = nx, ny, nz
    layers = st.find("Data").findall("Layer")
    cells = np.zeros(nx * ny * nz, dtype=np.uint8)
    for z in range(nz):
        text = layers[z].text or ""
        if len(text) != nx * ny:
            raise ValueError("layer size mismatch")
        cells[z * nx * ny:(z + 1) * nx * ny] = np.frombuffer(text.encode("latin-1"), dtype=np.uint8) - 48
    d["structure"] = cells
    occupied = cells > 0
    nvox = int(occupied.sum())
    for tag, key in (("PhaseOffset", "phase_offset"), ("TempAmpDamp", "temp_amp_damp"), ("Stiffness", "stiffness")) + DEV_LAYERS:
        block = st.find(tag)
        if block is None:
            d[key] = None
            continue
        values = np.zeros(nvox, dtype=np.float64)
        counter = 0
        for z, layer in enumerate(block.findall("Layer")[:nz]):
            items = (layer.text or "").split(",")
            for k in range(nx * ny):
                if occupied[z * nx * ny + k]:
                    values[counter] = _atof(items[k])
                    counter += 1
        d[key] = values
    d["nvox"] = nvox
    return d


def _dptr(arr):
    return None if arr is None else arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class OracleSim(object):
    """One robot stepped by the CPU restatement."""

    def __init__(self, model, half_angle=False):
        self.model = model  # keeps the numpy buffers alive
        self._lib = lib(half_angle)     # (half_angle: the instrument build, see lib())
        m = VxoModel()
        for name, _ in VxoModel._fields_:
            if name in ("structure", "nmat", "mat_E", "mat_rho", "mat_nu", "mat_cte", "mat_us", "mat_ud",
                        "phase_offset", "temp_amp_damp", "stiffness") + tuple(k for _, k in DEV_LAYERS):
                continue
            setattr(m, name, model[name])
        m.structure = model["structure"].ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))
        mats = model["materials"]
        m.nmat = len(mats["E"])
        m.mat_E, m.mat_rho, m.mat_nu = _dptr(mats["E"]), _dptr(mats["rho"]), _dptr(mats["nu"])
        m.mat_cte, m.mat_us, m.mat_ud = _dptr(mats["cte"]), _dptr(mats["us"]), _dptr(mats["ud"])
        m.phase_offset, m.temp_amp_damp = _dptr(model["phase_offset"]), _dptr(model["temp_amp_damp"])
        m.stiffness = _dptr(model["stiffness"])
        for _, key in DEV_LAYERS:
            setattr(m, key, _dptr(model.get(key)))
        self._cmodel = m
        self._h = self._lib.vxo_create(ctypes.byref(m))

    @classmethod
    def from_vxa(cls, path, variant=0):
        return cls(parse_vxa(path, variant))

    def close(self):
        if self._h:
            self._lib.vxo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, n=-1):
        return self._lib.vxo_step(self._h, n)

    def info(self):
        out = VxoInfo()
        self._lib.vxo_get_info(self._h, ctypes.byref(out))
        return out

    def state(self):
        n = self.info().nvox
        out = np.zeros((n, 14), dtype=np.float64)
        self._lib.vxo_get_state(self._h, out.ctypes.data)
        return out

    def set_state(self, state14):
        """test instrument: overwrite the voxels' state (the layout of state())"""
        a = np.ascontiguousarray(state14, dtype=np.float64)
        assert a.shape == (self.info().nvox, 14)
        self._lib.vxo_set_state(self._h, a.ctypes.data)

    def bonds(self):
        n = self.info().nbond
        v1, v2, ax = (np.zeros(n, dtype=np.int32) for _ in range(3))
        self._lib.vxo_get_bond_table(self._h, v1.ctypes.data, v2.ctypes.data, ax.ctypes.data)
        return v1, v2, ax

    def bond_modes(self):
        """instrument: 1 where the bond is in the small-angle branch, in bond-table order"""
        out = np.zeros(self.info().nbond, dtype=np.int32)
        self._lib.vxo_get_bond_modes(self._h, out.ctypes.data)
        return out

    def step_jittered(self, n, seed=1):
        """n steps, every position component moved by one ulp (up or down, pseudo-random) before each of them: the twin run that
        measures the reference algorithm's own drift under rounding-size noise (tests/test_gpu_ledger.py)"""
        L = self._lib
        done = 0
        for k in range(n):
            L.vxo_jitter(self._h, ctypes.c_uint(seed * 1000003 + self.info().steps if False else seed + k))
            if L.vxo_step(self._h, 1) != 1:
                break
            done += 1
        return done

    def constants(self):
        """([nvox, 12], [nbond, 23]): the constants Import leaves on voxels and bonds (vx_oracle.c vxo_get_constants)"""
        info = self.info()
        vox, bond = np.zeros((info.nvox, 12)), np.zeros((max(info.nbond, 1), 23))
        self._lib.vxo_get_constants(self._h, vox.ctypes.data, bond.ctypes.data)
        return vox, bond[:info.nbond]

    def result(self):
        out = VxoResult()
        self._lib.vxo_get_result(self._h, ctypes.byref(out))
        return out

    def cm_trace(self):
        """[n, 4] (time, x, y, z): SS.CMTraceTime / SS.CMTrace"""
        n = self._lib.vxo_get_cm_trace(self._h, None, 0)
        out = np.zeros((max(n, 1), 4), dtype=np.float64)
        self._lib.vxo_get_cm_trace(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n)
        return out[:n]

    def alg_bytes_per_step(self):
        return self._lib.vxo_alg_bytes_per_step(self._h)


# ------------------------------------------------------------------------------------------- probe traces
def read_trace(path):
    """Read a trace written by oracle/_ref/vxprobe (format: oracle/ref_probe_main.cpp header comment)."""
    with open(path, "rb") as handle:
        buf = handle.read()
    magic, nvox, nbond, opt_dt, dt_frac = struct.unpack_from("<iiidd", buf, 0)
    assert magic == 0x56585452
    off = struct.calcsize("<iiidd")
    records = []
    while True:
        (step,) = struct.unpack_from("<i", buf, off)
        if step == -1:
            off += 4
            break
        step, ncol, cur_time, dt = struct.unpack_from("<iidd", buf, off)
        off += struct.calcsize("<iidd")
        cm = np.frombuffer(buf, dtype="<f8", count=3, offset=off).copy()
        off += 24
        state = np.frombuffer(buf, dtype="<f8", count=nvox * 14, offset=off).reshape(nvox, 14).copy()
        off += nvox * 14 * 8
        records.append({"step": step, "ncol": ncol, "time": cur_time, "dt": dt, "cm": cm, "state": state})
    ini_cm = np.frombuffer(buf, dtype="<f8", count=3, offset=off).copy()
    cur_cm = np.frombuffer(buf, dtype="<f8", count=3, offset=off + 24).copy()
    (total_steps,) = struct.unpack_from("<i", buf, off + 48)
    return {"nvox": nvox, "nbond": nbond, "opt_dt": opt_dt, "dt_frac": dt_frac, "records": records,
            "ini_cm": ini_cm, "cur_cm": cur_cm, "total_steps": total_steps}


def read_result_xml(path):
    """All <tag>number</tag> pairs of a result XML as floats."""
    out = {}
    for el in ET.parse(path).getroot().iter():
        if el.text and el.text.strip():
            try:
                out[el.tag] = float(el.text)
            except ValueError:
                pass
    return out
