"""Small helpers shared by the boundary modules (reference: evosoro/tools/utils.py)."""


def xml_format(tag):
    """Return `tag` wrapped in angle brackets if it is not already (reference: evosoro/tools/utils.py:72-78)."""
    if not tag.startswith("<"):
        tag = "<" + tag
    if not tag.endswith(">"):
        tag = tag + ">"
    return tag


def py2_str(value):
    """str() with Python-2.7 float semantics.

    The reference is Python-2 code and stringifies every number it writes into the .vxa with str()
    (evosoro/tools/read_write_voxelyze.py:60,65,...).  Python 2 prints floats with 12 significant digits
    ('%.12g', plus '.0' when the result looks like an integer) while Python 3 prints the shortest
    round-tripping repr; the two differ for values such as 1/3.0.  A drop-in writer must emit what the
    reference emitted, so floats go through this function.  Non-floats are str()'d unchanged.
    """
    try:
        import numpy as _np
        is_float = isinstance(value, (float, _np.floating))
    except ImportError:  # pragma: no cover
        is_float = isinstance(value, float)
    if not is_float:
        return str(value)
    value = float(value)
    if value != value:
        return "nan"
    if value in (float("inf"), float("-inf")):
        return "inf" if value > 0 else "-inf"
    text = "%.12g" % value
    if "." not in text and "e" not in text and "n" not in text:
        text += ".0"
    return text
