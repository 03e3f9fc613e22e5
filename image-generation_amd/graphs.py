"""Host-side graph logic for the GRBM prior: QPU-topology generators, the
reference's sub-graph selection, colouring and the CSR "plan" the Gibbs kernel
consumes.

Replaces, for the local sampler, what the reference obtains from a live QPU:
``DWaveSampler(solver=qpu).to_networkx_graph()`` + ``greedy_get_subgraph`` +
``get_graph_mapping`` (/root/reference/src/utils/common.py:22-100, :123-126).
``dwave_networkx`` is not available offline, so the Pegasus / Zephyr
generators are written here from the published topology definitions
(Boothby et al., "Next-Generation Topology of D-Wave Quantum Processors",
2020; "Zephyr Topology of D-Wave Quantum Processors", 2021).
"""
from __future__ import annotations

import random
from dataclasses import dataclass
from itertools import product
from typing import Iterable, Optional, Sequence

import networkx as nx
import numpy as np

# ----------------------------------------------------------------------------
# topology generators
# ----------------------------------------------------------------------------

_PEGASUS_OFFSETS = (
    (2, 2, 2, 2, 10, 10, 10, 10, 6, 6, 6, 6),
    (6, 6, 6, 6, 2, 2, 2, 2, 10, 10, 10, 10),
)


def pegasus_graph(m: int = 16, fabric_only: bool = True) -> nx.Graph:
    """Pegasus P_m with linear integer labels (P16 = Advantage: 5640 qubits, degree 15)."""
    m1 = m - 1

    def label(u, w, k, z):
        return ((u * m + w) * 12 + k) * m1 + z

    g = nx.Graph()
    g.add_nodes_from(range(2 * m * 12 * m1))
    # external couplers: same wire, consecutive segments
    g.add_edges_from(
        (label(u, w, k, z), label(u, w, k, z + 1))
        for u, w, k, z in product((0, 1), range(m), range(12), range(m1 - 1))
    )
    # odd couplers: paired parallel qubits
    g.add_edges_from(
        (label(u, w, 2 * k, z), label(u, w, 2 * k + 1, z))
        for u, w, k, z in product((0, 1), range(m), range(6), range(m1))
    )
    # internal couplers: vertical (u=0) x horizontal (u=1) crossings
    off0, off1 = _PEGASUS_OFFSETS
    for w, kk, k, z in product(range(m), range(12), range(12), range(m1)):
        w2 = z + (1 if kk < off0[k] else 0)
        z2 = w - (1 if k < off1[kk] else 0)
        if 0 <= w2 < m and 0 <= z2 < m1:
            g.add_edge(label(0, w, k, z), label(1, w2, kk, z2))
    if fabric_only:
        keep = max(nx.connected_components(g), key=len)
        # keep the generator's node order (ascending labels)
        h = nx.Graph()
        h.add_nodes_from(v for v in g.nodes if v in keep)
        h.add_edges_from((a, b) for a, b in g.edges if a in keep and b in keep)
        g = h
    g.graph["family"] = "pegasus"
    return g


def zephyr_graph(m: int = 12, t: int = 4) -> nx.Graph:
    """Zephyr Z_{m,t} with linear integer labels (Z12 = Advantage2: 4800 qubits, degree 20)."""
    M = 2 * m + 1

    def label(u, w, k, j, z):
        return (((u * M + w) * t + k) * 2 + j) * m + z

    g = nx.Graph()
    g.add_nodes_from(range(2 * M * t * 2 * m))
    g.add_edges_from(
        (label(u, w, k, j, z), label(u, w, k, j, z + 1))
        for u, w, k, j, z in product((0, 1), range(M), range(t), (0, 1), range(m - 1))
    )
    g.add_edges_from(
        (label(u, w, k, 0, z), label(u, w, k, 1, z - a))
        for u, w, k, a in product((0, 1), range(M), range(t), (0, 1))
        for z in range(a, m)
    )
    g.add_edges_from(
        (label(0, 2 * w + 1 + a * (2 * i - 1), k, j, z), label(1, 2 * z + 1 + b * (2 * j - 1), h, i, w))
        for w, z, h, k, i, j, a, b in product(
            range(m), range(m), range(t), range(t), (0, 1), (0, 1), (0, 1), (0, 1)
        )
    )
    g.graph["family"] = "zephyr"
    return g


# ----------------------------------------------------------------------------
# layouts (for the UI's topology picture: /root/reference/src/utils/callback_helpers.py:344-381)
# ----------------------------------------------------------------------------
# ``dwave_networkx.drawing.{pegasus,zephyr}_layout`` are not available offline; these place every qubit at the midpoint
# of its wire segment in the lattice the topology papers draw (unit square, y pointing down), which is what those
# layouts depict.  Coordinates are a picture, not arithmetic: nothing on the training path reads them.


def pegasus_layout(m: int = 16, crosses: bool = True) -> dict:
    """node -> (x, y) for Pegasus P_m with the linear labels of :func:`pegasus_graph`.  A vertical qubit (u = 0) with
    tile offset w, wire k and segment z spans rows 12 z + shift .. 12 z + shift + 12 at column 12 w + k; horizontal
    qubits the same with the axes swapped.  ``crosses``: draw the two qubits of an odd-coupler pair slightly apart so
    that K4,4 crossings show as crosses (the reference asks for that look)."""
    m1 = m - 1
    off0, off1 = _PEGASUS_OFFSETS
    span = 12.0 * m
    pos = {}
    for u, w, k, z in product((0, 1), range(m), range(12), range(m1)):
        shift = (off0 if u == 0 else off1)[k]
        across = 12 * w + k + (0.25 * (1 if k % 2 else -1) if crosses else 0.0) + 0.5
        along = 12 * z + shift + 6.0
        x, y = (across, along) if u == 0 else (along, across)
        pos[((u * m + w) * 12 + k) * m1 + z] = (x / span, 1.0 - y / span)
    return pos


def zephyr_layout(m: int = 12, t: int = 4) -> dict:
    """node -> (x, y) for Zephyr Z_{m,t} with the linear labels of :func:`zephyr_graph`: a vertical qubit
    (u, w, k, j, z) = (0, ...) sits at column (2 t) w / 2-ish + k and spans two unit cells starting at 2 z + j."""
    M = 2 * m + 1
    span = float(M * t)
    pos = {}
    for u, w, k, j, z in product((0, 1), range(M), range(t), (0, 1), range(m)):
        across = w * t + k + 0.5
        along = (2 * z + j + 1) * t
        x, y = (across, along) if u == 0 else (along, across)
        pos[(((u * M + w) * t + k) * 2 + j) * m + z] = (x / span, 1.0 - y / span)
    return pos


# solver name -> (generator, h_range, j_range); the ranges are the published
# QPU properties the reference reads at /root/reference/src/utils/common.py:129
LOCAL_SOLVERS = {
    "Advantage_system4": (lambda: pegasus_graph(16), (-4.0, 4.0), (-1.0, 1.0)),
    "Advantage_system6": (lambda: pegasus_graph(16), (-4.0, 4.0), (-1.0, 1.0)),
    "Advantage2_system1": (lambda: zephyr_graph(12), (-4.0, 4.0), (-2.0, 1.0)),
    "MI355X_gibbs_pegasus": (lambda: pegasus_graph(16), (-4.0, 4.0), (-1.0, 1.0)),
    "MI355X_gibbs_zephyr": (lambda: zephyr_graph(12), (-4.0, 4.0), (-2.0, 1.0)),
}


# ----------------------------------------------------------------------------
# sub-graph selection (restates /root/reference/src/utils/common.py:22-100)
# ----------------------------------------------------------------------------


def greedy_get_subgraph(n_nodes: int, random_seed: Optional[int], graph: nx.Graph) -> nx.Graph:
    """Seeded greedy dense-subgraph pick; same draws, same result as the reference.

    The reference (/root/reference/src/utils/common.py:22-84) keeps the picked
    qubits in a list and tests membership by scanning it; here the list (whose
    order drives the shuffles and therefore the RNG stream) is shadowed by a
    set for the membership / intersection tests.
    """
    rng = random.Random(random_seed)
    adj = {v: list(graph.neighbors(v)) for v in graph.nodes()}
    picked = [rng.choice(list(graph.nodes()))]
    picked_set = set(picked)
    max_degree = max(len(a) for a in adj.values())

    while len(picked) < n_nodes:
        best_conn = 0
        want = min(max_degree, len(picked))
        candidate = None
        done = False
        rng.shuffle(picked)
        for v in picked:
            nbrs = list(adj[v])
            rng.shuffle(nbrs)
            for w in nbrs:
                if w in picked_set:
                    continue
                conn = sum(1 for x in set(adj[w]) if x in picked_set)
                if conn >= want:
                    candidate = w
                    done = True
                    break
                if conn > best_conn:
                    best_conn = conn
                    candidate = w
            if done:
                break
        picked.append(candidate)
        picked_set.add(candidate)
    return graph.subgraph(picked)


def get_graph_mapping(graph: nx.Graph):
    """Relabel to 0..n-1 in node-iteration order (/root/reference/src/utils/common.py:86-100)."""
    mapping = {phys: logical for logical, phys in enumerate(graph.nodes())}
    return nx.relabel_nodes(graph, mapping), mapping


# ----------------------------------------------------------------------------
# colouring + CSR plan for the block-Gibbs kernel
# ----------------------------------------------------------------------------


def dsatur_colouring(n: int, nbrs: Sequence[Sequence[int]]) -> np.ndarray:
    """Deterministic DSATUR (ties: higher degree, then lower index). Returns colour per node."""
    colour = -np.ones(n, dtype=np.int64)
    sat = [set() for _ in range(n)]
    deg = [len(a) for a in nbrs]
    for _ in range(n):
        best, key = -1, None
        for v in range(n):
            if colour[v] >= 0:
                continue
            kv = (len(sat[v]), deg[v], -v)
            if key is None or kv > key:
                best, key = v, kv
        c = 0
        while c in sat[best]:
            c += 1
        colour[best] = c
        for w in nbrs[best]:
            sat[w].add(c)
    return colour


@dataclass
class GibbsPlan:
    """Everything the sampler needs about the graph, as flat arrays.

    ``order``/``class_ptr``: spins grouped by colour class (the update order,
    part of the sampler's definition).  ``adj_*``: CSR adjacency in edge-list
    order; ``adj_eid`` indexes the GRBM's ``_quadratic`` vector.
    """

    n: int
    n_edges: int
    edge_i: np.ndarray
    edge_j: np.ndarray
    order: np.ndarray
    class_ptr: np.ndarray
    adj_ptr: np.ndarray
    adj_idx: np.ndarray
    adj_eid: np.ndarray

    @property
    def n_colours(self) -> int:
        return len(self.class_ptr) - 1

    @property
    def max_class(self) -> int:
        return int(np.max(np.diff(self.class_ptr)))

    @property
    def max_degree(self) -> int:
        return int(np.max(np.diff(self.adj_ptr))) if self.n else 0


def build_plan(n: int, edge_i: Iterable[int], edge_j: Iterable[int]) -> GibbsPlan:
    edge_i = np.asarray(list(edge_i), dtype=np.int64)
    edge_j = np.asarray(list(edge_j), dtype=np.int64)
    if edge_i.shape != edge_j.shape:
        raise ValueError("edge_i and edge_j must have the same length")
    if len(edge_i) and (edge_i.min() < 0 or edge_j.min() < 0 or max(edge_i.max(), edge_j.max()) >= n):
        raise ValueError("edge endpoint out of range")
    if np.any(edge_i == edge_j):
        raise ValueError("self-loops are not allowed in a GRBM graph")
    nbrs = [[] for _ in range(n)]
    eids = [[] for _ in range(n)]
    for e, (a, b) in enumerate(zip(edge_i.tolist(), edge_j.tolist())):
        nbrs[a].append(b)
        eids[a].append(e)
        nbrs[b].append(a)
        eids[b].append(e)
    colour = dsatur_colouring(n, nbrs)
    ncol = int(colour.max()) + 1 if n else 0
    order = np.concatenate([np.nonzero(colour == c)[0] for c in range(ncol)]).astype(np.int32) if n else np.zeros(0, np.int32)
    class_ptr = np.zeros(ncol + 1, dtype=np.int32)
    for c in range(ncol):
        class_ptr[c + 1] = class_ptr[c] + int(np.sum(colour == c))
    adj_ptr = np.zeros(n + 1, dtype=np.int32)
    for i in range(n):
        adj_ptr[i + 1] = adj_ptr[i] + len(nbrs[i])
    adj_idx = np.asarray([b for a in nbrs for b in a], dtype=np.int32)
    adj_eid = np.asarray([e for a in eids for e in a], dtype=np.int32)
    return GibbsPlan(
        n=n,
        n_edges=len(edge_i),
        edge_i=edge_i,
        edge_j=edge_j,
        order=order,
        class_ptr=class_ptr,
        adj_ptr=adj_ptr,
        adj_idx=adj_idx,
        adj_eid=adj_eid,
    )


def edges_of(graph: nx.Graph):
    """(nodes, edge_i, edge_j) in the iteration order the GRBM uses (i < j)."""
    nodes = list(graph.nodes())
    index = {v: k for k, v in enumerate(nodes)}
    ei, ej = [], []
    for a, b in graph.edges():
        ia, ib = index[a], index[b]
        ei.append(min(ia, ib))
        ej.append(max(ia, ib))
    return nodes, np.asarray(ei, dtype=np.int64), np.asarray(ej, dtype=np.int64)
