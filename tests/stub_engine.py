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


class Engine(object):
    def __init__(self, variant=VOXCAD, device=0):
        self.variant, self.models, self.sims = variant, [], []

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

    def counters(self):
        return None
