"""dwave_networkx: the names the reference's UI helper touches (/root/reference/src/utils/callback_helpers.py:24,366-374):
``pegasus_graph`` / ``zephyr_graph`` generators and ``drawing.pegasus_layout`` / ``drawing.zephyr_layout``, served by
this package's own topology code (image_generation_amd.graphs).  Chimera is not a topology of any local solver."""
from image_generation_amd import graphs as _graphs

from . import drawing  # noqa: F401


def pegasus_graph(m, **_unused):
    g = _graphs.pegasus_graph(int(m))
    g.graph.update(rows=int(m), family="pegasus")
    return g


def zephyr_graph(m, t=4, **_unused):
    g = _graphs.zephyr_graph(int(m), int(t))
    g.graph.update(rows=int(m), tile=int(t), family="zephyr")
    return g


def chimera_graph(*_a, **_k):
    raise NotImplementedError("no local solver has the Chimera topology")
