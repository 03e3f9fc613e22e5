"""dwave.system: a named LOCAL solver with the surface the reference touches on its QPU objects
(/root/reference/src/utils/common.py:41-42,123-129; /root/reference/src/utils/callback_helpers.py:359-374).

``DWaveSampler(solver=name)``: ``.to_networkx_graph()`` (ideal Pegasus P16 / Zephyr Z12 of ``graphs.LOCAL_SOLVERS``),
``.properties`` (``h_range``, ``j_range``, ``topology``, ``chip_id``), ``.solver.name``.
``FixedEmbeddingComposite(qpu, {logical: [physical]})``: ``.sample_ising(h, J, num_reads=..., ...)`` draws with the
MI355X block-Gibbs sampler on the induced sub-graph, variables labelled by the logical names.
"""
from types import SimpleNamespace

import networkx as nx

from image_generation_amd import graphs
from image_generation_amd.sampler import GibbsSampler

# settings of the local draw (the QPU has none of these; beta = 1 / PREFACTOR of training_parameters.yaml)
LOCAL_SOLVER = dict(sweeps=50, beta=20.0, seed=0, persistent=True, device="cuda")



def _topology(graph: nx.Graph):
    # Pegasus P16 has 5640 fabric qubits of degree <= 15, Zephyr Z12 4800 of degree <= 20
    return ("zephyr", [12, 4]) if max(d for _, d in graph.degree) > 15 else ("pegasus", [16])


class DWaveSampler:
    def __init__(self, solver=None, **_unused):
        name = solver if isinstance(solver, str) else (solver or {}).get("name") if isinstance(solver, dict) else None
        if name is None:
            name = sorted(graphs.LOCAL_SOLVERS)[0]
        if name not in graphs.LOCAL_SOLVERS:
            raise ValueError(f"unknown local solver {name!r}; known: {sorted(graphs.LOCAL_SOLVERS)}")
        self._make, h_range, j_range = graphs.LOCAL_SOLVERS[name]
        self._graph = self._make()
        kind, shape = _topology(self._graph)
        self.solver = SimpleNamespace(name=name)
        self.properties = {"h_range": list(h_range), "j_range": list(j_range), "chip_id": name,
                           "topology": {"type": kind, "shape": shape}, "category": "local-gibbs"}

    def to_networkx_graph(self) -> nx.Graph:
        return self._graph

    @property
    def nodelist(self):
        return sorted(self.to_networkx_graph().nodes)

    @property
    def edgelist(self):
        return sorted(tuple(sorted(e)) for e in self.to_networkx_graph().edges)


class FixedEmbeddingComposite:
    def __init__(self, child_sampler: DWaveSampler, embedding: dict):
        self.child = child_sampler
        self.embedding = {k: list(v) for k, v in embedding.items()}
        if any(len(v) != 1 for v in self.embedding.values()):
            raise ValueError("the local solver takes one-to-one embeddings only (chains of length 1), as the reference builds")
        self.properties = dict(child_sampler.properties)
        self._sampler = None

    def _build(self) -> GibbsSampler:
        if self._sampler is None:
            phys_to_log = {v[0]: k for k, v in self.embedding.items()}
            sub = self.child.to_networkx_graph().subgraph(phys_to_log.keys())
            nodes = sorted(self.embedding)  # logical labels; the reference's are 0..n-1
            pos = {lab: k for k, lab in enumerate(nodes)}
            ei, ej = [], []
            for a, b in sub.edges:
                ia, ib = pos[phys_to_log[a]], pos[phys_to_log[b]]
                ei.append(min(ia, ib)); ej.append(max(ia, ib))
            order = sorted(range(len(ei)), key=lambda e: (ei[e], ej[e]))
            plan = graphs.build_plan(len(nodes), [ei[e] for e in order], [ej[e] for e in order])
            cfg = LOCAL_SOLVER
            self._sampler = GibbsSampler(plan, nodes, beta=cfg["beta"], sweeps=cfg["sweeps"], seed=cfg["seed"],
                                         persistent=cfg["persistent"], device=cfg["device"],
                                         h_range=tuple(self.properties["h_range"]), j_range=tuple(self.properties["j_range"]))
        return self._sampler

    def sample_ising(self, h, J, **kwargs):
        return self._build().sample_ising(h, J, **kwargs)
