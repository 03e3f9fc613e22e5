"""Test infrastructure: one training step of ``ModelWrapper`` restated with the oracle (oracle/nets.py,
oracle/plugin.py, the C restatement of the sampler) from the model's CURRENT parameters, in any dtype on any device --
float32 on the CPU for small shapes, float64 on the GPU through stock PyTorch-ROCm for BASELINE.json's full sizes
(where the reference's own N x N kernel matrix would not fit: the MMD is evaluated in row chunks, same estimator).

Follows /root/reference/src/model_wrapper.py:297-344 (forward, MSE, draw, MMD, backward; then the GRBM branch:
second draw, ``mean E(spins) - mean E(samples)``).  Checker only: never imported by the product.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from oracle import gibbs, nets, plugin
from oracle.sampler import OracleGibbsSampler


def snapshot(model) -> Dict[str, torch.Tensor]:
    """A detached copy of everything the step reads: both state_dicts and the sampler's position."""
    snap = {"dvae." + k: v.detach().clone() for k, v in model._dvae.state_dict().items()}
    snap.update({"grbm." + k: v.detach().clone() for k, v in model._grbm.state_dict().items()})
    return snap


def oracle_sampler_like(model) -> OracleGibbsSampler:
    """The C restatement positioned where the model's sampler stands (same plan, seed, chain numbering, sweep index
    and -- for persistent chains -- the same state)."""
    s = model.sampler
    o = OracleGibbsSampler(s.plan, beta=s.beta, sweeps=s.sweeps, seed=s.seed, persistent=s.persistent,
                           chain_offset=s.chain_offset)
    o.sweep_count = s.sweep_count
    if s.persistent and s._state is not None:
        o.state = s._state.cpu().numpy().astype(np.int8)
    return o


def sampler_position(o: OracleGibbsSampler) -> dict:
    """Where an oracle sampler stands (to run the same draws a second time)."""
    return dict(plan=o.plan, beta=o.beta, sweeps=o.sweeps, seed=o.seed, persistent=o.persistent,
                chain_offset=o.chain_offset, sweep_count=o.sweep_count, calls=o.calls,
                state=None if o.state is None else o.state.copy())


def oracle_sampler_like_snapshot(pos: dict) -> OracleGibbsSampler:
    o = OracleGibbsSampler(pos["plan"], beta=pos["beta"], sweeps=pos["sweeps"], seed=pos["seed"],
                           persistent=pos["persistent"], chain_offset=pos["chain_offset"])
    o.sweep_count, o.calls = pos["sweep_count"], pos["calls"]
    o.state = None if pos["state"] is None else pos["state"].copy()
    return o


def meta_of(model) -> dict:
    """The step's configuration as plain values (so the model itself can be released before the oracle runs)."""
    return dict(n=int(model.n_latents), R=int(model.N_REPLICAS), C=int(model.NUM_READS), prefactor=float(model.PREFACTOR),
                h_range=tuple(model.linear_range), j_range=tuple(model.quadratic_range))


def _draw(osampler: OracleGibbsSampler, meta: dict, linear: torch.Tensor, quadratic: torch.Tensor) -> np.ndarray:
    num_reads = meta["C"]
    hs, Js = gibbs.scaled_fields(linear.detach().float().cpu().numpy(), quadratic.detach().float().cpu().numpy(),
                                 meta["prefactor"], meta["h_range"], meta["j_range"])
    p = osampler.plan
    nodes = list(range(p.n))
    h = {v: float(hs[v]) for v in nodes}
    J = {(int(a), int(b)): float(Js[e]) for e, (a, b) in enumerate(zip(p.edge_i, p.edge_j))}
    return osampler.sample_ising(h, J, num_reads=num_reads).record.sample.astype(np.float32)


def chunked_mmd(x: torch.Tensor, y: torch.Tensor, chunk: int = 1024, n_kernels: int = 7, factor: float = 2.0):
    """oracle/plugin.py's ``mmd_loss`` defaults (plain distance, data-driven detached bandwidth, kernels summed,
    unbiased estimator) evaluated ``chunk`` rows at a time: returns ``(loss, d loss / d x)`` without ever holding
    the (nx + ny)^2 matrix.  Equal to ``plugin.mmd_loss`` (asserted by tests/test_oracle_pinned.py on small shapes)."""
    nx, ny = x.shape[0], y.shape[0]
    N = nx + ny
    xs = x.detach().clone().requires_grad_(True)
    yd = y.detach()
    with torch.no_grad():
        z = torch.cat([xs.detach(), yd])
        dsum = torch.zeros((), dtype=x.dtype, device=x.device)
        for r0 in range(0, N, chunk):
            dsum += plugin.pairwise_distance(z[r0:r0 + chunk], z, False).sum()
        bws = (dsum / (N * N - N)) * plugin.kernel_factors(n_kernels, factor).to(device=x.device, dtype=x.dtype)

    def kern(a, b):
        return torch.exp(-plugin.pairwise_distance(a, b, False).unsqueeze(0) / bws.reshape(-1, 1, 1)).sum(0)

    total = torch.zeros((), dtype=x.dtype, device=x.device)
    for r0 in range(0, nx, chunk):
        rows = xs[r0:r0 + chunk]
        kxx = kern(rows, xs)
        kxy = kern(rows, yd)
        m = rows.shape[0]
        part = (kxx.sum() - kxx[:, r0:r0 + m].diagonal().sum()) / (nx * (nx - 1)) - 2.0 * kxy.sum() / (nx * ny)
        part.backward()
        total = total + part.detach()
    with torch.no_grad():
        kyy = kern(yd, yd)
        total = total + (kyy.sum() - kyy.trace()) / (ny * (ny - 1))
    return total, xs.grad


def oracle_step(meta: dict, snap: Dict[str, torch.Tensor], images: torch.Tensor, gumbels: torch.Tensor,
                masks: List[torch.Tensor], osampler: OracleGibbsSampler, *, dtype=torch.float64, device="cuda",
                grbm_branch: bool = True, mmd_chunk: Optional[int] = None) -> dict:
    """Losses and every parameter gradient of one step, from the snapshot ``snap`` (and ``meta_of(model)``,
    ``oracle_sampler_like(model)``) taken BEFORE the model stepped."""
    dev = torch.device(device)
    n, R = meta["n"], meta["R"]

    def cast(v, grad):
        if v.dtype.is_floating_point:
            return v.to(dev, dtype).requires_grad_(grad)
        return v.to(dev)

    enc = {k[len("dvae._encoder."):]: cast(v, "running" not in k) for k, v in snap.items() if k.startswith("dvae._encoder.")}
    dec = {k[len("dvae._decoder."):]: cast(v, "running" not in k) for k, v in snap.items() if k.startswith("dvae._decoder.")}
    lin = cast(snap["grbm._linear"], True)
    quad = cast(snap["grbm._quadratic"], True)
    ei, ej = snap["grbm._edge_idx_i"].to(dev), snap["grbm._edge_idx_j"].to(dev)
    x = images.to(dev, dtype)
    logits = nets.encoder_forward(enc, x, training=True)
    spins = plugin.gumbel_latent_to_discrete(logits, R, gumbels=gumbels.to(dev, dtype))
    recon = nets.decoder_forward(dec, spins, training=True, dropout_masks=[m.to(dev, dtype) for m in masks])
    # replicated target without the repeat: mean over (B, R, 1, 32, 32)
    mse = ((recon - x.unsqueeze(1)) ** 2).mean()
    samples = torch.from_numpy(_draw(osampler, meta, snap["grbm._linear"], snap["grbm._quadratic"])).to(dev, dtype)
    flat = spins.reshape(-1, n)
    if mmd_chunk is None:
        mmd = plugin.mmd_loss(flat, samples)
        torch.autograd.backward([mse + mmd])
        mmd = mmd.detach()
    else:
        mmd, g_flat = chunked_mmd(flat, samples, chunk=mmd_chunk)
        torch.autograd.backward([mse, flat], [torch.ones_like(mse), g_flat])
    out = {"mse": float(mse), "mmd": float(mmd), "spins": flat.detach(), "samples": samples,
           "grads": {**{"_encoder." + k: v.grad for k, v in enc.items() if v.dtype.is_floating_point and v.requires_grad},
                     **{"_decoder." + k: v.grad for k, v in dec.items() if v.dtype.is_floating_point and v.requires_grad}}}
    if grbm_branch:
        samples2 = torch.from_numpy(_draw(osampler, meta, snap["grbm._linear"], snap["grbm._quadratic"])).to(dev, dtype)
        nll = plugin.grbm_energy(flat.detach(), lin, quad, ei, ej).mean() - plugin.grbm_energy(samples2, lin, quad, ei, ej).mean()
        nll.backward()
        out.update(nll=float(nll), grad_linear=lin.grad, grad_quadratic=quad.grad)
    return out


def zero_true_gradient(name: str) -> bool:
    """Biases of the convolutions that sit in front of a BatchNorm: their true gradient is zero (the batch mean removes
    them), what either implementation returns is rounding noise."""
    parts = name.split(".")
    if not name.endswith("bias") or len(parts) < 3:
        return False
    if parts[0] == "_encoder" and parts[1] == "conv":
        return int(parts[2]) % 4 == 0
    if parts[0] == "_decoder" and parts[1] == "convtrans":
        return parts[2] in ("0", "5", "10", "15")
    return False
