"""Parameter containers whose attributes are stringified into the .vxa file.

Mirrors the attribute names and defaults of the reference containers so that experiment scripts written
for evosoro (`Sim(dt_frac=0.9, ...)`, `Env(frequency=4.0, ...)`, `ObjectiveDict().add_objective(...)`)
keep working unchanged:
  VoxCadParams.add_param   evosoro/base.py:9-18
  Sim                      evosoro/base.py:21-47
  Env                      evosoro/base.py:50-73
  ObjectiveDict            evosoro/base.py:95-154
"""
from collections import OrderedDict

from evosoro_amd.tools.utils import xml_format

_SIM_DEFAULTS = OrderedDict([
    ("self_collisions_enabled", True), ("simulation_time", 10), ("dt_frac", 0.7), ("stop_condition", 2),
    ("fitness_eval_init_time", 2), ("equilibrium_mode", 0), ("min_temp_fact", 0.1),
    ("max_temp_fact_change", 0.00001), ("max_stiffness_change", 10000), ("min_elastic_mod", 5e006),
    ("max_elastic_mod", 5e008), ("afterlife_time", 0), ("mid_life_freeze_time", 0),
])

_ENV_DEFAULTS = OrderedDict([
    ("frequency", 4.0), ("gravity_enabled", 1), ("temp_enabled", 1), ("floor_enabled", 1), ("floor_slope", 0.0),
    ("lattice_dimension", 0.01), ("fat_stiffness", 5e+006), ("bone_stiffness", 5e+008),
    ("muscle_stiffness", 5e+006), ("sticky_floor", 0), ("time_between_traces", 0), ("actuation_variance", 0),
    ("temp_amp", 39),
])


class VoxCadParams(object):
    """Attribute bag; `add_param` registers an extra `<Tag>value</Tag>` line for the .vxa writer."""

    def __init__(self):
        self.sub_groups = []
        self.new_param_tag_dict = OrderedDict()

    def add_param(self, name, val, tag):
        setattr(self, name, val)
        self.new_param_tag_dict[name] = xml_format(tag)

    def _assign(self, defaults, positional, keywords):
        names = list(defaults)
        if len(positional) > len(names):
            raise TypeError("too many positional arguments")
        values = OrderedDict(defaults)
        for name, val in zip(names, positional):
            values[name] = val
        for name, val in keywords.items():
            if name not in defaults:
                raise TypeError("unexpected keyword argument %r" % name)
            values[name] = val
        for name, val in values.items():
            setattr(self, name, val)


class Sim(VoxCadParams):
    """Simulator block of the .vxa (`<Simulator>`): integration, collisions, stop condition."""

    def __init__(self, *args, **kwargs):
        VoxCadParams.__init__(self)
        self.sub_groups = ["Integration", "Damping", "Collisions", "Features", "StopCondition", "EquilibriumMode",
                           "GA"]
        self._assign(_SIM_DEFAULTS, args, kwargs)


class Env(VoxCadParams):
    """Environment block of the .vxa (`<Environment>`) plus lattice/material stiffness used in `<VXC>`."""

    def __init__(self, *args, **kwargs):
        VoxCadParams.__init__(self)
        self.sub_groups = ["Fixed_Regions", "Forced_Regions", "Gravity", "Thermal"]
        self._assign(_ENV_DEFAULTS, args, kwargs)


class ObjectiveDict(dict):
    """rank -> objective description; rank order = order of `add_objective` calls, "fitness" forced to rank 0."""

    def __init__(self):
        super(ObjectiveDict, self).__init__()
        self.max_rank = 0

    def add_objective(self, name, maximize, tag, node_func=None, output_node_name=None, logging_only=False):
        rank = self.max_rank
        if name == "fitness" and self.max_rank > 0:
            rank = 0
            for old in reversed(range(len(self))):
                self[old + 1] = self[old]
        dict.__setitem__(self, rank, {
            "name": name,
            "maximize": maximize,
            "tag": None if tag is None else xml_format(tag),
            "worst_value": -10e6 if maximize else 10e6,
            "node_func": node_func,
            "output_node_name": output_node_name,
            "logging_only": logging_only,
        })
        self.max_rank += 1
