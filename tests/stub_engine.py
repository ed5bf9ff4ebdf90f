"""Test double for evosoro_amd.engine used by the CPU suite (-m "not gpu"): same Python surface, but robots are
stepped by the CPU oracle.  Lives under tests/ (test infrastructure); the product never imports it."""
import os

from oracle import vxoracle as vo

VOXCAD, VOXCAD_LAND_WATER = 0, 1
ROBOT_PENDING, ROBOT_FINISHED, ROBOT_DIVERGED, ROBOT_EMPTY, ROBOT_COL_OVERFLOW = 0, 1, 2, 3, 4

_TAGS = [("NormFinalDist", "norm_final_dist"), ("NormRegimeDist", "norm_regime_dist"),
         ("NormFrozenDist", "norm_frozen_dist"), ("FinalDist", "final_dist"), ("finalDistY", "final_dist_y"),
         ("AnteriorDist", "anterior_dist"), ("PosteriorDist", "posterior_dist"), ("AnteriorY", "anterior_y"),
         ("PosteriorY", "posterior_y"), ("EndOfLifePosteriorY", "end_of_life_posterior_y"),
         ("FallAdjPostY", "fall_adj_post_y"), ("NumNonFeetTouchingFloor", "num_non_feet_touching_floor"),
         ("NumTouchingFloor", "num_touching_floor"), ("Lifetime", "lifetime")]


class _Result(object):
    def __init__(self, res, rebuilds):
        for name, _ in res._fields_:
            val = getattr(res, name)
            setattr(self, name, list(val) if hasattr(val, "__len__") else val)
        self.col_rebuilds = rebuilds
        self.robot_volume_start = self.robot_volume_end = -1.0      # (the oracle does not restate the mesh-volume tags)
        self.hull_volume_start = self.hull_volume_end = -1.0


class _Counters(object):
    """the fields of vxh_counters that bench.py reads"""
    def __init__(self):
        self.voxel_steps = self.kernel_seconds = self.dominant_seconds = self.dominant_alg_bytes = self.dominant_voxel_steps = 0.0
        self.launches = self.dominant_launches = self.dominant_block = 0


_real = None          # the real evosoro_amd.engine, for its host-only calls (bench_stub_runner.py sets it before putting this module in its place)


def inspect_vxa(path, variant=VOXCAD):
    global _real
    if _real is None:
        from evosoro_amd import engine as _real_engine
        _real = _real_engine
    return _real.inspect_vxa(path, variant)


class Engine(object):
    def __init__(self, variant=VOXCAD, device=0):
        self.variant, self.models, self.sims, self.paths = variant, [], [], []
        self._c = _Counters()

    def add_vxa_files(self, paths):
        first = len(self.sims)
        for p in paths:
            self.add_vxa_file(p)
        return first

    def dims(self, i):
        d = inspect_vxa(self.paths[i], self.variant)
        return {"nvox": d.nvox, "nbond": d.nbond, "dt": d.dt, "planned_steps": d.planned_steps}

    def step(self, n):
        import time
        t0 = time.perf_counter()
        for sim in self.sims:
            before = sim.info().steps
            sim.step(n)
            info = sim.info()
            self._c.voxel_steps += float(info.nvox) * (info.steps - before)
            self._c.dominant_alg_bytes += (224.0 * info.nvox + 144.0 * info.nbond) * (info.steps - before)
        self._c.dominant_voxel_steps = self._c.voxel_steps
        self._c.kernel_seconds += time.perf_counter() - t0
        self._c.dominant_seconds = self._c.kernel_seconds
        self._c.launches += 1
        self._c.dominant_launches = 1

    def counters(self):
        import copy
        return copy.copy(self._c)

    def bond_modes(self):
        total = sum(sim.info().nbond for sim in self.sims)
        return total - sum(sim.info().n_small_angle for sim in self.sims), total

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        self.sims = []

    def set_option(self, key, value):
        pass

    def add_vxa_file(self, path):
        self.models.append(vo.parse_vxa(path, self.variant))
        self.sims.append(vo.OracleSim(self.models[-1]))
        self.paths.append(path)
        return len(self.sims) - 1

    def add_robots(self, template_text, robots, round_like_text=True):
        """the in-memory hand-off (evosoro_amd.engine.Engine.add_robots) on the oracle's data model: simulator / environment / palette
        from the template, lattice and layers from the arrays, values through the writer's decimal text"""
        import tempfile
        import numpy as np
        with tempfile.NamedTemporaryFile("w", suffix=".vxa", delete=False) as f:
            f.write(template_text)
        base = vo.parse_vxa(f.name, self.variant)
        os.remove(f.name)
        keys = dict([("PhaseOffset", "phase_offset"), ("TempAmpDamp", "temp_amp_damp"), ("Stiffness", "stiffness")] + list(vo.DEV_LAYERS))
        first = len(self.sims)
        for material, layers, name in robots:
            model = dict(base)
            cells = np.ascontiguousarray(np.asarray(material).transpose(2, 1, 0)).reshape(-1).astype(np.uint8)
            model["nx"], model["ny"], model["nz"] = (int(v) for v in np.asarray(material).shape)
            model["structure"], model["nvox"] = cells, int((cells > 0).sum())
            for key in keys.values():
                model[key] = None
            for tag, arr in layers.items():
                flat = np.asarray(arr, dtype=np.float64).transpose(2, 1, 0).reshape(-1)[cells > 0]
                model[keys[tag.strip("<>")]] = np.array([float("%.12g" % v) for v in flat]) if round_like_text else flat.copy()
            if name is not None:
                model["fitness_file_name"] = name
            self.models.append(model)
            self.sims.append(vo.OracleSim(model))
        return first

    def run(self):
        for sim in self.sims:
            sim.step(-1)

    def result(self, i):
        return _Result(self.sims[i].result(), self.sims[i].info().col_rebuilds)

    def fitness_file_name(self, i):
        return self.models[i]["fitness_file_name"]

    def write_result_xml(self, i, path=None):
        res = self.result(i)
        path = path or self.fitness_file_name(i)
        with open(path, "w") as f:
            f.write("<?xml version=\"1.0\" ?>\n<Voxelyze_Sim_Result Version=\"1.0\">\n    <Fitness>\n")
            for tag, field in _TAGS:
                f.write("        <%s>%g</%s>\n" % (tag, getattr(res, field), tag))
            f.write("    </Fitness>\n</Voxelyze_Sim_Result>\n")

