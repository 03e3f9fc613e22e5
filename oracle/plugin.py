"""CPU restatement of the ``dwave-pytorch-plugin`` functions on the hot path.

Test infrastructure (see oracle/__init__.py).

**PARITY UNPINNED.**  ``dwave-pytorch-plugin~=0.3`` is a third-party
dependency (/root/reference/requirements.txt:4) that is neither vendored in
/root/reference nor installed in this image, and the reference has no tests or
golden vectors at this boundary (/root/reference/tests/ is empty).  What is
restated here is the published algorithm, anchored on the reference's call
sites and README:

* ``DiscreteVariationalAutoencoder``: call sites
  /root/reference/src/model_wrapper.py:184-188, :297, :465; 3-tuple order
  ``(latents, discretes, reconstructed)``.
* default ``latent_to_discrete``: Gumbel-softmax (tau = 1/7, hard, two classes
  with logits [l, 0]), replicated n_samples times, mapped to +-1.
* ``GraphRestrictedBoltzmannMachine.__call__``: energy
  ``x.h + sum_e J_e x_i x_j`` (/root/reference/README.md, static/eq6.png;
  used at /root/reference/src/losses.py:61).
* ``GaussianKernel(n_kernels=7)`` + ``maximum_mean_discrepancy_loss``
  (/root/reference/src/model_wrapper.py:273, :320; README eq3/eq4).
  Defaults: plain Euclidean distance, data-driven bandwidth
  ``sum(D)/(N^2-N)`` (no grad), bandwidths ``bw * 2^(k - K//2)``, kernels
  summed, unbiased estimator.  Alternatives are switches.
"""
from __future__ import annotations

from typing import Optional

import torch

GUMBEL_TAU = 1.0 / 7.0


# --------------------------------------------------------------------------
# latent -> discrete
# --------------------------------------------------------------------------


def gumbel_latent_to_discrete(
    logits: torch.Tensor, n_samples: int, gumbels: Optional[torch.Tensor] = None, tau: float = GUMBEL_TAU
) -> torch.Tensor:
    """(B,n) logits -> (B,R,n) spins in {-1,+1} with straight-through gradient.

    ``gumbels`` (B,R,n,2): the Gumbel(0,1) noise ``-log(Exp(1))`` that
    ``torch.nn.functional.gumbel_softmax`` would draw; injected so GPU and CPU
    consume identical noise.
    """
    B, n = logits.shape
    two = torch.stack([logits, torch.zeros_like(logits)], dim=-1)  # (B,n,2): class 0 = "+1"
    two = two.unsqueeze(1).repeat(1, n_samples, 1, 1)  # (B,R,n,2)
    if gumbels is None:
        gumbels = -torch.empty_like(two).exponential_().log()
    y = (two + gumbels) / tau
    y_soft = y.softmax(-1)
    index = y_soft.max(-1, keepdim=True)[1]
    y_hard = torch.zeros_like(two).scatter_(-1, index, 1.0)
    one_hot = y_hard - y_soft.detach() + y_soft
    return one_hot[..., 0] * 2 - 1


def heaviside_latent_to_discrete(logits: torch.Tensor, n_samples: int) -> torch.Tensor:
    """/root/reference/src/utils/common.py:160-173 (H(0) = 0 -> -1; identity gradient)."""
    with torch.no_grad():
        hard = torch.heaviside(logits, values=torch.tensor(0, dtype=logits.dtype)) * 2 - 1
    return (hard - logits.detach() + logits).unsqueeze(1)


# --------------------------------------------------------------------------
# GRBM energy
# --------------------------------------------------------------------------


def grbm_energy(x: torch.Tensor, linear: torch.Tensor, quadratic: torch.Tensor, edge_i: torch.Tensor, edge_j: torch.Tensor) -> torch.Tensor:
    """E(x) = x @ h + (x_i * x_j) @ J over the last dim."""
    return x @ linear + (x[..., edge_i] * x[..., edge_j]) @ quadratic


def grbm_sufficient_statistics(x: torch.Tensor, edge_i: torch.Tensor, edge_j: torch.Tensor):
    """(mean_b x_i, mean_b x_i x_j): the gradients of mean-energy wrt (h, J)."""
    x = x.reshape(-1, x.shape[-1])
    return x.mean(0), (x[:, edge_i] * x[:, edge_j]).mean(0)


# --------------------------------------------------------------------------
# Gaussian kernel + MMD
# --------------------------------------------------------------------------


def kernel_factors(n_kernels: int, factor: float = 2.0) -> torch.Tensor:
    return factor ** (torch.arange(n_kernels) - n_kernels // 2).to(torch.float32)


def pairwise_distance(x: torch.Tensor, y: torch.Tensor, squared: bool) -> torch.Tensor:
    # matmul form (what torch.cdist uses for > 25 rows): clamp(|x|^2 + |y|^2 - 2 x.y, 0)
    x2 = (x * x).sum(-1, keepdim=True)
    y2 = (y * y).sum(-1, keepdim=True)
    d2 = (x2 + y2.T - 2.0 * (x @ y.T)).clamp_min(0.0)
    if squared:
        return d2
    # sqrt with a zero sub-gradient at d2 == 0 (torch.cdist's backward does the same)
    safe = torch.where(d2 > 0, d2, torch.ones_like(d2))
    return torch.where(d2 > 0, safe.sqrt(), torch.zeros_like(d2))


def gaussian_kernel_matrix(
    x: torch.Tensor,
    y: torch.Tensor,
    n_kernels: int = 7,
    factor: float = 2.0,
    bandwidth: Optional[float] = None,
    squared: bool = False,
    reduce: str = "sum",
) -> torch.Tensor:
    D = pairwise_distance(x, y, squared)
    if bandwidth is None:
        N = D.shape[0]
        bw = D.detach().sum() / (N * N - N)
    else:
        bw = torch.as_tensor(bandwidth, dtype=D.dtype, device=D.device)
    bws = bw * kernel_factors(n_kernels, factor).to(device=D.device, dtype=D.dtype)
    K = torch.exp(-D.unsqueeze(0) / bws.reshape(-1, 1, 1))
    return K.sum(0) if reduce == "sum" else K.mean(0)


def mmd_loss(
    x: torch.Tensor,
    y: torch.Tensor,
    n_kernels: int = 7,
    factor: float = 2.0,
    bandwidth: Optional[float] = None,
    squared: bool = False,
    reduce: str = "sum",
    biased: bool = False,
) -> torch.Tensor:
    nx, ny = x.shape[0], y.shape[0]
    xy = torch.cat([x, y], dim=0)
    K = gaussian_kernel_matrix(xy, xy, n_kernels, factor, bandwidth, squared, reduce)
    kxx, kyy, kxy = K[:nx, :nx], K[nx:, nx:], K[:nx, nx:]
    if biased:
        return kxx.mean() + kyy.mean() - 2.0 * kxy.mean()
    xx = (kxx.sum() - kxx.trace()) / (nx * (nx - 1))
    yy = (kyy.sum() - kyy.trace()) / (ny * (ny - 1))
    return xx + yy - 2.0 * kxy.mean()


# --------------------------------------------------------------------------
# nn.Module shells with the plugin's names (so the reference's verbatim
# ModelWrapper.step can be driven over this restatement: tests/golden/make_golden.py)
# --------------------------------------------------------------------------


class GaussianKernel(torch.nn.Module):
    def __init__(self, n_kernels: int, factor: float = 2.0, bandwidth: Optional[float] = None,
                 squared: bool = False, reduce: str = "sum"):
        super().__init__()
        self.register_buffer("factors", kernel_factors(n_kernels, factor))
        self.n_kernels, self.factor, self.bandwidth = n_kernels, factor, bandwidth
        self.squared, self.reduce = squared, reduce

    def forward(self, x, y):
        return gaussian_kernel_matrix(x, y, self.n_kernels, self.factor, self.bandwidth, self.squared, self.reduce)


def maximum_mean_discrepancy_loss(x, y, kernel: GaussianKernel, biased: bool = False):
    return mmd_loss(x, y, kernel.n_kernels, kernel.factor, kernel.bandwidth, kernel.squared, kernel.reduce, biased)


class DiscreteVariationalAutoencoder(torch.nn.Module):
    def __init__(self, encoder, decoder, latent_to_discrete=None):
        super().__init__()
        self._encoder = encoder
        self._decoder = decoder
        self._latent_to_discrete = latent_to_discrete or gumbel_latent_to_discrete

    @property
    def encoder(self):
        return self._encoder

    @property
    def decoder(self):
        return self._decoder

    @property
    def latent_to_discrete(self):
        return self._latent_to_discrete

    def forward(self, x, n_samples: int = 1):
        latents = self._encoder(x)
        discretes = self._latent_to_discrete(latents, n_samples)
        return latents, discretes, self._decoder(discretes)


class SampleSetShim:
    """The three things the hot path reads off a dimod.SampleSet."""

    class _Record:
        def __init__(self, sample, energy):
            self.sample = sample
            self.energy = energy

    def __init__(self, samples, variables, energy=None, vartype="SPIN"):
        import numpy as np

        self.record = SampleSetShim._Record(np.asarray(samples), energy)
        self.variables = list(variables)
        self.vartype = vartype


class GraphRestrictedBoltzmannMachine(torch.nn.Module):
    """Fully-visible GRBM with the checkpoint schema of SURVEY.md App. B."""

    def __init__(self, nodes, edges):
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

    def forward(self, x):
        return grbm_energy(x, self._linear, self._quadratic, self._edge_idx_i, self._edge_idx_j)

    def to_ising(self, prefactor, linear_range=None, quadratic_range=None):
        h = prefactor * self._linear.detach()
        J = prefactor * self._quadratic.detach()
        if linear_range is not None:
            h = h.clamp(*linear_range)
        if quadratic_range is not None:
            J = J.clamp(*quadratic_range)
        hd = dict(zip(self._nodes, h.tolist()))
        Jd = {
            (self._nodes[i], self._nodes[j]): v
            for i, j, v in zip(self._edge_idx_i.tolist(), self._edge_idx_j.tolist(), J.tolist())
        }
        return hd, Jd

    def sampleset_to_tensor(self, sample_set, device=None):
        import numpy as np

        col = {v: k for k, v in enumerate(sample_set.variables)}
        perm = [col[v] for v in self._nodes]
        return torch.from_numpy(np.asarray(sample_set.record.sample)[:, perm].astype("float32")).to(device)

    def sample(self, sampler, *, prefactor, linear_range=None, quadratic_range=None, device=None,
               sample_params=None, as_tensor=True):
        h, J = self.to_ising(prefactor, linear_range, quadratic_range)
        ss = sampler.sample_ising(h, J, **(sample_params or {}))
        return self.sampleset_to_tensor(ss, device) if as_tensor else ss
