"""GPU parity ledger (-m gpu): every golden case of tests/golden/manifest.json run to its stop condition on the engine, once per
kernel path, and the outcome WRITTEN DOWN -- steps, the spread the reference algorithm itself shows under a one-ulp perturbation,
the tolerance that follows from it, the error actually achieved against the reference binary's final state, and whether the strict
1e-9-voxel bar held over the whole run -- as gpurun_out/r06_parity_<kernel path>.json (copied to profiles/ when a round is closed).
Asserts the tolerance for every case and a floor on the number of cases that meet the strict bar.

Stated tolerance: max(1e-9 voxel, 20 x spread) -- tests/test_gpu_parity.py's.  For most of round 3 it carried a third term, 3e-13 voxel
x steps, for a creep of the engine against the reference that grew with the length of the run (8e-9 voxel after the 68 319 steps of
the reference's own hexapus.vxa) and whose source had not been found.  It was found late in the round (scripts/dev_gpu_diag.py drift7:
the oracle put on the ENGINE's state before every step shows what one step of the two differs by -- 1e-12 voxel where a one-ulp
change of the inputs moves the oracle by 2e-15): the back-rotation of a bond's forces through a rotation MATRIX that assumed a unit
quaternion, where the reference's RotateVec3DInv carries |q|^2 -- and FromAngleToPosX's small-angle branch returns a quaternion of
norm^2 1 + (y^2 + z^2)^2 / 4 (kernels.hpp RotInv).  With the diagonal corrected one step differs by 2e-15 voxel, the hexapus run ends
4e-14 voxel from the reference, and the term is gone.  Ruled out on the way, each as an instrument build of the oracle or a what-if
build of the engine (drift .. drift6): FMA contraction, the half-angle form of FromAngleToPosX, the angle-addition form of the
actuation sine, the damping constants folded with 1 / dt, a bond in different angle modes on the two sides; one other contributor
was real and is pinned: 1 - w * w in ToRotationVector contracted into one FMA (kernels.hpp one_minus_square)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOOR_VOX = 1e-9
_SPREADS = {}


def _whole_run_spread(vo, model, planned):
    """largest deviation, at four points of the run, between the oracle and the oracle fed a gravity constant (in a fluid: a drag
    coefficient) changed by an ulp or two"""
    twin = dict(model)
    if model.get("fluid_env"):
        twin["aggregate_drag_coef"] = model["aggregate_drag_coef"] * (1 + 4e-16)
    else:
        twin["grav_acc"] = model["grav_acc"] * (1 + 4e-16)
    a, b = vo.OracleSim(model), vo.OracleSim(twin)
    worst = 0.0
    for upto in sorted(set([planned // 4, planned // 2, 3 * planned // 4, planned])):
        a.step(upto - a.info().steps)
        b.step(upto - b.info().steps)
        worst = max(worst, np.abs(a.state()[:, :3] - b.state()[:, :3]).max() / model["lattice_dim"])
    return worst


def test_parity_ledger(golden_dir, manifest, kernel_path):
    from evosoro_amd import engine as eng_mod
    from oracle import vxoracle as vo
    rows = []
    for variant in (0, 1):
        names = [n for n, e in manifest.items() if e["variant"] == ("lw" if variant else "land")]
        with eng_mod.Engine(variant, 0) as eng:
            for n in names:
                eng.add_vxa_file(os.path.join(golden_dir, "vxa", n + ".vxa"))
            eng.run()
            for i, n in enumerate(names):
                model = vo.parse_vxa(os.path.join(golden_dir, "vxa", n + ".vxa"), variant)
                lat = model["lattice_dim"]
                planned = eng.dims(i)["planned_steps"]
                if n not in _SPREADS:
                    _SPREADS[n] = _whole_run_spread(vo, model, planned)
                spread = _SPREADS[n]
                tol = max(FLOOR_VOX, 20 * spread)
                res = eng.result(i)
                final = os.path.join(golden_dir, "expected", n + ".final.bin")
                row = {"case": n, "variant": variant, "kernel_path": kernel_path, "nvox": res.nvox, "nbond": res.nbond, "steps": res.steps,
                       "spread_vox": spread, "tolerance_vox": tol}
                if os.path.exists(final):                  # final state as the reference binary left it (oracle/_ref/vxprobe)
                    trace = vo.read_trace(final)
                    row["steps_reference"] = int(trace["total_steps"])
                    row["err_cur_cm_vox"] = float(np.abs(np.array(res.cur_cm) - trace["cur_cm"]).max() / lat)
                    row["err_ini_cm_vox"] = float(np.abs(np.array(res.ini_cm) - trace["ini_cm"]).max() / lat)
                    row["against"] = "reference binary, final state"
                else:                                      # (the .vxa files shipped with the reference: result XML only) -> the oracle's final state
                    sim = vo.OracleSim(model)
                    sim.step(-1)
                    info = sim.info()
                    row["steps_reference"] = int(info.steps)
                    row["err_cur_cm_vox"] = float(np.abs(np.array(res.cur_cm) - np.array(info.cur_cm)).max() / lat)
                    row["err_ini_cm_vox"] = float(np.abs(np.array(res.ini_cm) - np.array(info.ini_cm)).max() / lat)
                    row["against"] = "oracle (pinned on the reference), final state"
                err = max(row["err_cur_cm_vox"], row["err_ini_cm_vox"])
                row["strict_1e-9"] = bool(err <= FLOOR_VOX and res.steps == row["steps_reference"])
                row["within_1e-12"] = bool(err <= 1e-12 and res.steps == row["steps_reference"])
                row["within_tolerance"] = bool(err <= tol and res.steps == row["steps_reference"] and res.status == eng_mod.ROBOT_FINISHED)
                rows.append(row)
    strict = sum(1 for r in rows if r["strict_1e-9"])
    tight = sum(1 for r in rows if r["within_1e-12"])
    ledger = {"kernel_path": kernel_path, "cases": len(rows), "strict_1e-9": strict, "within_1e-12": tight,
              "within_tolerance": sum(1 for r in rows if r["within_tolerance"]),
              "note": "error = max over x, y, z of |centre of mass - reference| at the end of the whole evaluation, in voxels (also IniCM); "
                      "tolerance = max(1e-9, 20 x spread of the reference algorithm under a 1-ulp change of one input); "
                      "strict_1e-9 = within 1e-9 voxel whatever the length of the run",
              "rows": rows}
    for out_dir in (os.path.join(REPO, "gpurun_out"),):
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "r06_parity_%s.json" % kernel_path), "w") as f:
            json.dump(ledger, f, indent=1)
    print("parity ledger [%s]: %d cases, %d within the strict 1e-9 voxel bar over the whole run (%d within 1e-12), %d within tolerance" % (
        kernel_path, len(rows), strict, tight, ledger["within_tolerance"]))
    assert len(rows) == len(manifest)
    bad = [r["case"] for r in rows if not r["within_tolerance"]]
    assert not bad, bad
    assert strict >= len(rows) - 3, "only %d of %d cases meet the strict bar" % (strict, len(rows))
    assert tight >= len(rows) - 6, "only %d of %d cases end within 1e-12 voxel of the reference" % (tight, len(rows))
