"""``dwave_networkx.drawing`` layouts: node -> (x, y) dictionaries for the graphs of this shim's generators."""
from image_generation_amd import graphs as _graphs


def pegasus_layout(G, scale=1.0, center=None, dim=2, crosses=False):
    pos = _graphs.pegasus_layout(int(G.graph.get("rows", 16)), crosses=bool(crosses))
    return {v: (scale * x, scale * y) for v, (x, y) in pos.items() if v in G}


def zephyr_layout(G, scale=1.0, center=None, dim=2):
    pos = _graphs.zephyr_layout(int(G.graph.get("rows", 12)), int(G.graph.get("tile", 4)))
    return {v: (scale * x, scale * y) for v, (x, y) in pos.items() if v in G}


def chimera_layout(*_a, **_k):
    raise NotImplementedError("no local solver has the Chimera topology")
