"""The C-ABI shared library: loads, exports every symbol include/vxhip.h declares, fails loudly without a GPU,
and its host-side model builder agrees with the oracle and the reference (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from evosoro_amd import engine
from oracle import vxoracle as vo

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["probe6", "rand6_nocol", "rand6_col", "soft5_init0", "phase4", "example_1", "example_phaseoffset",
         "lw_land6", "lw_swim6", "lw_hexapus", "lw_quadruped_land"]


def _declared_functions():
    text = open(os.path.join(REPO, "include", "vxhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vxh_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = engine.load_library()
    declared = _declared_functions()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), "libvxhip.so does not export %s" % name
    assert sorted(engine.EXPORTS) == declared
    assert b"gfx950" in lib.vxh_version()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(engine.VxhError) as err:
        engine.Engine(engine.VOXCAD, 0)
    assert err.value.status == -2          # VXH_ERR_NO_DEVICE: the product path has no CPU implementation


def test_result_struct_layout_matches_header():
    # field order/types of the ctypes mirrors vs the C declarations (catches silent ABI drift)
    text = open(os.path.join(REPO, "include", "vxhip.h")).read()
    body = text[text.index("typedef struct vxh_result {"):text.index("} vxh_result;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"(?:int|double|long long)\s+([^;]+);", body):
        for item in decl.split(","):
            names.append(item.strip().split("[")[0])
    assert names == [n for n, _ in engine.VxhResult._fields_]


@pytest.mark.parametrize("name", CASES)
def test_host_model_matches_oracle_and_reference(golden_dir, name):
    path = os.path.join(golden_dir, "vxa", name + ".vxa")
    variant = 1 if name.startswith("lw_") else 0
    info = engine.inspect_vxa(path, variant)
    sim = vo.OracleSim.from_vxa(path, variant)
    oi = sim.info()
    assert (info.nvox, info.nbond, info.nsurf) == (oi.nvox, oi.nbond, oi.nsurf)
    assert info.opt_dt == oi.opt_dt                      # CalcMaxDt bitwise
    trace = vo.read_trace(os.path.join(golden_dir, "expected", name + ".early.bin"))
    assert info.opt_dt == trace["opt_dt"] and info.nbond == trace["nbond"]
    final = os.path.join(golden_dir, "expected", name + ".final.bin")
    if os.path.exists(final):                            # the planner replays CurTime += dt exactly like the reference
        assert info.planned_steps == vo.read_trace(final)["total_steps"]
    assert info.alg_bytes_per_step == 224.0 * info.nvox + 144.0 * info.nbond


@pytest.mark.parametrize("name", CASES)
def test_import_constants_equal_the_oracles_bit_for_bit(golden_dir, name):
    """every constant the import leaves on a voxel (SetMaterial) and on a bond (UpdateConstants:
    stiffnesses a1 a2 b1 b2 b3 and the 2 sqrt(k m) damping terms), compared with the oracle's -- which is pinned on the reference --
    as bit patterns, bond by bond in the reference's creation order.  (The kernels read per-class tables derived from these.)"""
    import numpy as np
    path = os.path.join(golden_dir, "vxa", name + ".vxa")
    variant = 1 if name.startswith("lw_") else 0
    vox, bond = engine.inspect_constants(path, variant)
    ovox, obond = vo.OracleSim.from_vxa(path, variant).constants()
    assert vox.shape == ovox.shape and bond.shape == obond.shape
    assert np.array_equal(vox.view(np.int64), ovox.view(np.int64)), np.argwhere(vox != ovox)[:5]
    assert np.array_equal(bond.view(np.int64), obond.view(np.int64)), np.argwhere(bond != obond)[:5]


def test_reader_rejects_bad_input():
    with pytest.raises(engine.VxhError) as err:
        engine.inspect_vxa("<VXA><Simulator></Simulator>")
    assert err.value.status == -3
    good = open(os.path.join(REPO, "tests", "golden", "vxa", "phase4.vxa")).read()
    with pytest.raises(engine.VxhError) as err:      # features outside the supported scope are refused, not ignored
        engine.inspect_vxa(good.replace("<NumFixed>0</NumFixed>", "<NumFixed>1</NumFixed>"))
    assert err.value.status == -7
    with pytest.raises(engine.VxhError):
        engine.inspect_vxa(good.replace('Compression="ASCII_READABLE"', 'Compression="ZLIB"'))
    empty = re.sub(r"<!\[CDATA\[[0-9]+\]\]>", lambda m: "<![CDATA[" + "0" * (len(m.group(0)) - 12) + "]]>", good, count=4)
    info = engine.inspect_vxa(re.sub(r"<PhaseOffset>.*?</PhaseOffset>", "", empty, flags=re.S))
    assert info.nvox == 0 and info.nbond == 0        # an empty robot is representable (status EMPTY at run time)


def test_convex_hull_volume_known_answers():
    """the computation behind <ConvexHullVolumeStart/End> (the reference shells out to qhull, LW/VX_MeshUtil.cpp:821-900): closed forms,
    the degenerate inputs a voxel lattice produces (thousands of coplanar / collinear points), and volumes `qhull FS` printed for
    seeded point sets (recorded from the binary the reference vendors)"""
    import numpy as np
    hull = engine.convex_hull_volume
    assert abs(hull([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [.1, .1, .1]]) - 1.0 / 6) < 1e-15
    grid = np.array([[x, y, z] for x in range(5) for y in range(5) for z in range(5)], float) * 0.01
    assert abs(hull(grid) - 6.4e-05) < 1e-18                                  # a 4 x 4 x 4 cm cube
    ell = np.array([[x, y, z] for x in range(7) for y in range(7) for z in range(4) if not (x > 3 and y > 3)], float) * 0.01
    assert abs(hull(ell) - 9.45e-05) < 1e-18                                  # hull of an L: the square minus a corner triangle
    assert hull(grid[:25]) == 0.0 and hull(grid[:5]) == 0.0 and hull(grid[:3]) == 0.0    # a plane, a line, too few points
    rs = np.random.RandomState(1)
    assert abs(hull(rs.uniform(-1, 1, (500, 3))) - 7.13974032262) < 1e-10     # qhull FS: 7.13974032262
    sphere = rs.normal(size=(800, 3))
    sphere /= np.linalg.norm(sphere, axis=1)[:, None]
    assert abs(hull(sphere) - 4.12888859762) < 1e-10                          # qhull FS: 4.12888859762
    # order of the points does not matter
    assert abs(hull(sphere[::-1]) - hull(sphere)) < 1e-13


def test_every_engine_option_is_documented_in_the_header():
    """include/vxhip.h is where a caller learns the keys of vxh_set_option and their defaults.  Its list went stale once (round 2: it
    still described a tiling policy and a launch length that had been replaced), so: every key Engine::set_option accepts must be
    named there -- except "dbg", which only exists in the developer library -- and the default launch length it states must be the
    engine's."""
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    src = open(os.path.join(root, "evosoro_amd", "csrc", "engine.hip")).read()
    body = src[src.index("void Engine::check_option("):]          # (set_option validates through it: every key appears there)
    body = body[:body.index("\n}\n")]
    keys = set(re.findall(r'key == "([a-z_]+)"', body))
    assert {"tiled", "tile_small", "steps_per_launch", "fused"} <= keys          # (the parse found the function)
    header = open(os.path.join(root, "include", "vxhip.h")).read()
    doc = header[header.index("/* Options (all have working defaults):"):header.index("int  vxh_set_option(")]
    missing = sorted(k for k in keys - {"dbg"} if '"%s"' % k not in doc)
    assert not missing, "options accepted by the engine but not documented in include/vxhip.h: %s" % missing
    default = re.search(r"int steps_per_launch_ = (\d+);", open(os.path.join(root, "evosoro_amd", "csrc", "engine.hpp")).read()).group(1)
    assert re.search(r'"steps_per_launch".*\(default %s\)' % default, doc), "the header's default of steps_per_launch is not %s" % default
