"""Local sampler: the on-GPU block-Gibbs stand-in for the QPU.

Mirrors the sampler side of the reference:
``get_sampler_and_sampler_kwargs`` (/root/reference/src/utils/common.py:103-140)
returns ``(sampler, sampler_kwargs, graph, linear_range, quadratic_range)`` where
``sampler.sample_ising(h, J, **sampler_kwargs)`` is a blocking QPU call.  Here
``sampler`` is a :class:`GibbsSampler` whose draw is ``dvg_gibbs_sample``
(include/dvg.h): many persistent chains, graph-coloured block-Gibbs sweeps.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence, Tuple

import networkx as nx
import numpy as np
import torch

from . import _lib
from .graphs import LOCAL_SOLVERS, GibbsPlan, build_plan, edges_of, get_graph_mapping, greedy_get_subgraph


class GraphHandle:
    """Owns a ``dvg_graph_t`` (device copy of the plan) for one device."""

    def __init__(self, plan: GibbsPlan, device: torch.device):
        self.plan = plan
        self.device = torch.device(device)
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        arrs = [i32(plan.edge_i), i32(plan.edge_j), i32(plan.order), i32(plan.class_ptr), i32(plan.adj_ptr),
                i32(plan.adj_idx), i32(plan.adj_eid)]
        p = [a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) for a in arrs]
        with torch.cuda.device(self.device):
            _lib.check(
                L.dvg_graph_create(plan.n, plan.n_edges, p[0], p[1], p[2], p[3], plan.n_colours, p[4], p[5], p[6],
                                   ctypes.byref(self._h)),
                "dvg_graph_create",
            )

    @property
    def ptr(self):
        return self._h

    def __del__(self):
        try:
            if self._h:
                _lib.lib().dvg_graph_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass


class _Record:
    def __init__(self, sample, energy):
        self.sample = sample
        self.energy = energy


class SampleSet:
    """The slice of ``dimod.SampleSet`` the hot path touches
    (/root/reference/src/losses.py:59, /root/reference/src/utils/persistent_qpu_sampler.py:90-95):
    ``.record.sample`` (reads x variables, int8), ``.record.energy``, ``.variables``, ``.vartype``.
    ``device_tensor`` keeps the float32 samples on the GPU so ``sampleset_to_tensor``
    needs no host round trip."""

    def __init__(self, samples: np.ndarray, variables: Sequence, energy=None, vartype="SPIN", device_tensor=None):
        self._samples = samples
        self._energy = energy
        self.variables = list(variables)
        self.vartype = vartype
        self.device_tensor = device_tensor

    @property
    def record(self):
        if self._samples is None:  # lazily materialise on the host
            self._samples = self.device_tensor.to(torch.int8).cpu().numpy()
        return _Record(self._samples, self._energy)

    def __len__(self):
        return int(self.device_tensor.shape[0]) if self.device_tensor is not None else len(self._samples)


class GibbsSampler:
    """``sample_ising``-compatible local solver running on the MI355X.

    Args:
        plan: graph plan (colour classes + CSR) of the relabelled sub-graph.
        nodes: variable labels in GRBM order (0..n-1 after get_graph_mapping).
        beta: inverse temperature applied to the (h, J) it is handed.  The
            reference sends ``prefactor * (h, J)`` to the QPU
            (/root/reference/src/model_wrapper.py:311); ``beta = 1/prefactor`` makes the
            stationary distribution the GRBM's own ``exp(-E)``.
        sweeps: Gibbs sweeps per draw.   persistent: keep chains between draws (PCD).
        chain_offset: global id of this sampler's first chain (rank sharding).
    """

    def __init__(self, plan: GibbsPlan, nodes: Sequence, *, beta: float, sweeps: int, seed: int,
                 persistent: bool = True, device="cuda", chain_offset: int = 0,
                 h_range: Tuple[float, float] = (-4.0, 4.0), j_range: Tuple[float, float] = (-1.0, 1.0)):
        self.plan = plan
        self.nodes = list(nodes)
        self.beta = float(beta)
        self.sweeps = int(sweeps)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.persistent = persistent
        self.device = torch.device(device)
        self.chain_offset = int(chain_offset)
        self.properties = {"h_range": list(h_range), "j_range": list(j_range)}
        self._graph: Optional[GraphHandle] = None
        self._state: Optional[torch.Tensor] = None
        self.sweep_count = 0
        self.calls = 0

    # -- native fast path ------------------------------------------------------------------
    @property
    def graph(self) -> GraphHandle:
        if self._graph is None:
            self._graph = GraphHandle(self.plan, self.device)
        return self._graph

    def reset(self):
        self._state = None
        self.sweep_count = 0

    def launch_info(self, num_reads: int) -> dict:
        """Launch geometry of one draw of ``num_reads`` chains (``dvg_gibbs_launch_info``: nothing is launched):
        ``workgroups``, ``threads`` per workgroup, ``lds_bytes`` per workgroup."""
        import ctypes

        wg, th, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        _lib.check(_lib.lib().dvg_gibbs_launch_info(self.graph.ptr, int(num_reads), ctypes.byref(wg), ctypes.byref(th),
                                                    ctypes.byref(lds)), "dvg_gibbs_launch_info")
        return {"workgroups": wg.value, "threads": th.value, "lds_bytes": lds.value}

    def sample_native(self, linear: torch.Tensor, quadratic: torch.Tensor, prefactor: float,
                      linear_range=None, quadratic_range=None, num_reads: int = 256) -> torch.Tensor:
        """Device tensors in, (num_reads, n) float32 +-1 device tensor out; no host sync."""
        L = _lib.lib()
        lin = _lib.require_cuda(linear.detach(), "linear")
        quad = _lib.require_cuda(quadratic.detach(), "quadratic")
        n = self.plan.n
        if lin.numel() != n or quad.numel() != self.plan.n_edges:
            raise _lib.DvgError("linear/quadratic do not match the sampler's graph")
        init = 0
        if self._state is None or not self.persistent or self._state.shape[0] != num_reads:
            self._state = torch.empty((num_reads, n), dtype=torch.int8, device=lin.device)
            init = 1
            if not self.persistent:
                pass  # fresh chains every call; the sweep counter keeps the random stream moving
        out = torch.empty((num_reads, n), dtype=torch.float32, device=lin.device)
        inf = float("inf")
        hl, hh = linear_range if linear_range is not None else (-inf, inf)
        jl, jh = quadratic_range if quadratic_range is not None else (-inf, inf)
        with torch.cuda.device(lin.device):
            _lib.check(
                L.dvg_gibbs_sample(self.graph.ptr, lin.data_ptr(), quad.data_ptr(), float(prefactor), float(hl),
                                   float(hh), float(jl), float(jh), self.beta, self._state.data_ptr(), num_reads,
                                   self.chain_offset & 0xFFFFFFFF, self.seed, self.sweep_count & 0xFFFFFFFF,
                                   self.sweeps, init, out.data_ptr(), _lib.DYN, _lib.stream_ptr(lin.device)),
                "dvg_gibbs_sample",
            )
        self.sweep_count += self.sweeps
        self.calls += 1
        return out

    # -- dimod-style entry point -----------------------------------------------------------
    def sample_ising(self, h: Dict, J: Dict, num_reads: int = 1, answer_mode: str = "raw", auto_scale: bool = False,
                     annealing_time: float = 1.0, label: str = "", **_unused) -> SampleSet:
        """Same call the reference makes on its QPU sampler (kwargs built at
        /root/reference/src/utils/common.py:130-138).  ``h``/``J`` are already
        scaled by the prefactor and clamped by the caller (plugin ``to_ising``)."""
        if auto_scale:
            raise ValueError("auto_scale=True is not supported: the sampler draws from the distribution it is given")
        idx = {v: k for k, v in enumerate(self.nodes)}
        lin = torch.zeros(self.plan.n, dtype=torch.float32)
        for v, val in h.items():
            lin[idx[v]] = float(val)
        edge_pos = {(int(a), int(b)): e for e, (a, b) in enumerate(zip(self.plan.edge_i, self.plan.edge_j))}
        quad = torch.zeros(self.plan.n_edges, dtype=torch.float32)
        for (a, b), val in J.items():
            ia, ib = idx[a], idx[b]
            key = (min(ia, ib), max(ia, ib))
            if key not in edge_pos:
                raise ValueError(f"coupler {(a, b)} is not an edge of the sampler's graph")
            quad[edge_pos[key]] = float(val)
        dev = self.device
        t = self.sample_native(lin.to(dev), quad.to(dev), 1.0, None, None, num_reads)
        return SampleSet(None, self.nodes, energy=None, device_tensor=t)


def get_sampler_and_sampler_kwargs(num_reads: int, annealing_time: float, n_latents: int, random_seed: int, qpu: str,
                                   *, sweeps: int = 50, beta: Optional[float] = None, prefactor: float = 0.05,
                                   persistent: bool = True, device="cuda", chain_offset: int = 0,
                                   graph: Optional[nx.Graph] = None):
    """Local-solver counterpart of /root/reference/src/utils/common.py:103-140.

    Returns the same 5-tuple ``(sampler, sampler_kwargs, mapped_graph, linear_range, quadratic_range)``.
    ``qpu`` names a topology in ``graphs.LOCAL_SOLVERS`` (the three QPU names of the shipped
    checkpoints are accepted and map to the ideal Pegasus P16 / Zephyr Z12 graph).
    """
    if qpu not in LOCAL_SOLVERS:
        raise ValueError(f"unknown local solver {qpu!r}; known: {sorted(LOCAL_SOLVERS)}")
    make, h_range, j_range = LOCAL_SOLVERS[qpu]
    full = graph if graph is not None else make()
    sub = greedy_get_subgraph(n_nodes=n_latents, random_seed=random_seed, graph=full)
    mapped, _mapping = get_graph_mapping(sub)
    nodes, ei, ej = edges_of(mapped)
    plan = build_plan(len(nodes), ei, ej)
    sampler = GibbsSampler(plan, nodes, beta=(1.0 / prefactor) if beta is None else beta, sweeps=sweeps,
                           seed=random_seed, persistent=persistent, device=device, chain_offset=chain_offset,
                           h_range=h_range, j_range=j_range)
    sampler_kwargs = dict(num_reads=num_reads, answer_mode="raw", auto_scale=False, annealing_time=annealing_time,
                          label="Examples - ML MNIST Image Gen")
    return sampler, sampler_kwargs, mapped, tuple(h_range), tuple(j_range)
