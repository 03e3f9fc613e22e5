"""MI355X-native counterparts of the ``dwave.plugins.torch`` classes the reference uses
(/root/reference/src/model_wrapper.py:25-30): same names, constructor arguments,
attributes, ``state_dict`` keys and call signatures, with the arithmetic behind them
running in libdvg.so.

``dwave-pytorch-plugin`` is absent from the reference tree (requirements.txt:4); the
semantics follow the published API as restated (and documented, switch by switch) in
oracle/plugin.py and DESIGN.md.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import numpy as np
import torch

from . import functional as F
from .graphs import GibbsPlan, build_plan
from .sampler import GibbsSampler, GraphHandle, SampleSet


class DiscreteVariationalAutoencoder(torch.nn.Module):
    """``(latents, discretes, reconstructed) = dvae(x, n_samples)``
    (/root/reference/src/model_wrapper.py:184-188, :297)."""

    def __init__(self, encoder: torch.nn.Module, decoder: torch.nn.Module,
                 latent_to_discrete: Optional[Callable[[torch.Tensor, int], torch.Tensor]] = None):
        super().__init__()
        self._encoder = encoder
        self._decoder = decoder
        self._custom_l2d = latent_to_discrete
        self.gumbel_seed = 0          # device-RNG stream of the default latent_to_discrete
        self._gumbel_calls = 0
        self._injected_gumbels: Optional[torch.Tensor] = None

    @property
    def encoder(self):
        return self._encoder

    @property
    def decoder(self):
        return self._decoder

    def inject_gumbels(self, gumbels: Optional[torch.Tensor]):
        """Parity hook: Gumbel(0,1) noise (B,R,n,2) for the next default latent_to_discrete call."""
        self._injected_gumbels = gumbels

    def _default_l2d(self, logits: torch.Tensor, n_samples: int) -> torch.Tensor:
        g, self._injected_gumbels = self._injected_gumbels, None
        offset = self._gumbel_calls
        self._gumbel_calls += 1
        return F.gumbel_latent_to_discrete(logits, n_samples, gumbels=g, seed=self.gumbel_seed, offset=offset)

    def default_l2d_raw(self, logits: torch.Tensor, n_samples: int):
        """The default latent_to_discrete outside autograd: ``(spins, dspin)`` (same noise stream / injected noise as
        :meth:`_default_l2d`); ``None`` when a custom latent_to_discrete is installed."""
        if self._custom_l2d is not None:
            return None
        g, self._injected_gumbels = self._injected_gumbels, None
        offset = self._gumbel_calls
        self._gumbel_calls += 1
        return F.gumbel_forward_raw(logits, n_samples, gumbels=g, seed=self.gumbel_seed, offset=offset)

    @property
    def latent_to_discrete(self):
        return self._custom_l2d if self._custom_l2d is not None else self._default_l2d

    def forward(self, x: torch.Tensor, n_samples: int = 1):
        latents = self._encoder(x)
        discretes = self.latent_to_discrete(latents, n_samples)
        return latents, discretes, self._decoder(discretes)


class GraphRestrictedBoltzmannMachine(torch.nn.Module):
    """Fully-visible GRBM: parameters ``_linear`` (n), ``_quadratic`` (|E|) and the index
    buffers of the shipped checkpoints.  ``grbm(x)`` is the energy."""

    def __init__(self, nodes: Iterable, edges: Iterable):
        super().__init__()
        self._nodes = list(nodes)
        idx = {v: k for k, v in enumerate(self._nodes)}
        ei, ej = [], []
        for a, b in edges:
            ia, ib = idx[a], idx[b]
            ei.append(min(ia, ib))
            ej.append(max(ia, ib))
        n, ne = len(self._nodes), len(ei)
        self._linear = torch.nn.Parameter(0.05 * (2 * torch.rand(n) - 1))
        self._quadratic = torch.nn.Parameter(5.0 * (2 * torch.rand(ne) - 1))
        self.register_buffer("_edge_idx_i", torch.tensor(ei, dtype=torch.int64))
        self.register_buffer("_edge_idx_j", torch.tensor(ej, dtype=torch.int64))
        self.register_buffer("_visible_idx", torch.arange(n, dtype=torch.int64))
        for name in ("_hidden_idx", "_flat_adj", "_flat_j_idx", "_bin_idx"):
            self.register_buffer(name, torch.zeros(0, dtype=torch.int64))
        self._plan: Optional[GibbsPlan] = None
        self._handles = {}

    # -- graph plumbing --------------------------------------------------------------------
    @property
    def n_nodes(self) -> int:
        return len(self._nodes)

    @property
    def nodes(self):
        return list(self._nodes)

    @property
    def plan(self) -> GibbsPlan:
        if self._plan is None or self._plan.n_edges != self._edge_idx_i.numel():
            self._plan = build_plan(self.n_nodes, self._edge_idx_i.cpu().numpy(), self._edge_idx_j.cpu().numpy())
        return self._plan

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._plan = None  # the checkpoint's edge list is authoritative
        self._handles = {}

    def graph_handle(self, device) -> GraphHandle:
        key = str(torch.device(device))
        if key not in self._handles:
            self._handles[key] = GraphHandle(self.plan, device)
        return self._handles[key]

    # -- energy ---------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.grbm_energy(x, self._linear, self._quadratic, self.graph_handle(x.device))

    # -- sampling -------------------------------------------------------------------------
    def to_ising(self, prefactor: float, linear_range=None, quadratic_range=None):
        """(h, J) dicts handed to a generic sampler: prefactor-scaled, clamped to the solver ranges."""
        h = prefactor * self._linear.detach()
        J = prefactor * self._quadratic.detach()
        if linear_range is not None:
            h = h.clamp(*linear_range)
        if quadratic_range is not None:
            J = J.clamp(*quadratic_range)
        hd = dict(zip(self._nodes, h.tolist()))
        Jd = {(self._nodes[i], self._nodes[j]): v
              for i, j, v in zip(self._edge_idx_i.tolist(), self._edge_idx_j.tolist(), J.tolist())}
        return hd, Jd

    def sampleset_to_tensor(self, sample_set, device=None) -> torch.Tensor:
        dt = getattr(sample_set, "device_tensor", None)
        if dt is not None and list(sample_set.variables) == self._nodes:
            return dt if device is None else dt.to(device)
        col = {v: k for k, v in enumerate(sample_set.variables)}
        perm = [col[v] for v in self._nodes]
        t = torch.from_numpy(np.asarray(sample_set.record.sample)[:, perm].astype(np.float32))
        return t if device is None else t.to(device)

    def sample(self, sampler, *, prefactor: float, linear_range=None, quadratic_range=None, device=None,
               sample_params: Optional[dict] = None, as_tensor: bool = True):
        """Draw spin strings from the model (/root/reference/src/model_wrapper.py:309-316).

        With the local :class:`GibbsSampler` the GRBM parameters go to the kernel as device
        tensors (scaling and clamping happen in its prologue): no host round trip.  Any other
        ``sample_ising``-style sampler gets the (h, J) dicts like the plugin would send."""
        sample_params = dict(sample_params or {})
        if isinstance(sampler, GibbsSampler) and list(sampler.nodes) == self._nodes:
            num_reads = int(sample_params.get("num_reads", 1))
            t = sampler.sample_native(self._linear, self._quadratic, prefactor, linear_range, quadratic_range, num_reads)
            if as_tensor:
                return t if device is None else t.to(device)
            return SampleSet(None, self._nodes, device_tensor=t)
        h, J = self.to_ising(prefactor, linear_range, quadratic_range)
        ss = sampler.sample_ising(h, J, **sample_params)
        return self.sampleset_to_tensor(ss, device) if as_tensor else ss


class GaussianKernel(torch.nn.Module):
    """Multi-bandwidth RBF kernel configuration (/root/reference/src/model_wrapper.py:273).

    The hot path never builds the kernel matrix: ``maximum_mean_discrepancy_loss`` hands this
    module's settings to the fused ``dvg_mmd_fwd_bwd``.  ``forward(x, y)`` (the explicit matrix)
    is kept for API compatibility as plain tensor algebra on the inputs' device."""

    def __init__(self, n_kernels: int, factor: float = 2.0, bandwidth: Optional[float] = None,
                 squared: bool = False, reduce: str = "sum"):
        super().__init__()
        self.register_buffer("factors", factor ** (torch.arange(n_kernels) - n_kernels // 2).to(torch.float32))
        self.n_kernels, self.factor, self.bandwidth = int(n_kernels), float(factor), bandwidth
        self.squared, self.reduce = bool(squared), reduce

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        d2 = ((x * x).sum(-1, keepdim=True) + (y * y).sum(-1, keepdim=True).T - 2.0 * (x @ y.T)).clamp_min(0.0)
        D = d2 if self.squared else torch.where(d2 > 0, torch.where(d2 > 0, d2, torch.ones_like(d2)).sqrt(), d2)
        N = D.shape[0]
        bw = D.detach().sum() / (N * N - N) if self.bandwidth is None else torch.as_tensor(self.bandwidth, device=D.device)
        K = torch.exp(-D.unsqueeze(0) / (bw * self.factors.to(D.device)).reshape(-1, 1, 1))
        return K.sum(0) if self.reduce == "sum" else K.mean(0)


def maximum_mean_discrepancy_loss(x: torch.Tensor, y: torch.Tensor, kernel: GaussianKernel, biased: bool = False):
    """MMD^2(x, y) under ``kernel`` with gradient wrt x (/root/reference/src/model_wrapper.py:320)."""
    return F.mmd_loss(x, y, n_kernels=kernel.n_kernels, factor=kernel.factor, bandwidth=kernel.bandwidth,
                      squared=kernel.squared, reduce_mean=(kernel.reduce == "mean"), biased=biased)


def maximum_mean_discrepancy_loss_and_grad(x: torch.Tensor, y: torch.Tensor, kernel: GaussianKernel, biased: bool = False):
    """Same value as :func:`maximum_mean_discrepancy_loss` plus its gradient wrt ``x``, both detached (the fused
    kernel computes them together anyway)."""
    return F.mmd_loss_and_grad(x, y, n_kernels=kernel.n_kernels, factor=kernel.factor, bandwidth=kernel.bandwidth,
                               squared=kernel.squared, reduce_mean=(kernel.reduce == "mean"), biased=biased)
