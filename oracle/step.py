"""CPU restatement of one full training step (the work `ModelWrapper.step` does,
/root/reference/src/model_wrapper.py:279-353), on stock PyTorch CPU ops plus the oracle
Gibbs sampler in place of the QPU.  Test infrastructure: used by tests as the checker and by
bench.py's ``cpu_baseline`` leg (kind "port"); never by the product.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import nets, plugin
from .sampler import OracleGibbsSampler


def init_params(n: int, seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic parameters with the reference's shapes (not its init: baseline timing only)."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}

    def w(shape, fan_in):
        return (torch.randn(shape, generator=g) / np.sqrt(fan_in)).requires_grad_(True)

    ch = nets.encoder_channels(n)
    for l in range(4):
        ci = 4 * l
        p[f"enc.conv.{ci}.weight"] = w((ch[l + 1], ch[l], 3, 3), ch[l] * 9)
        p[f"enc.conv.{ci}.bias"] = torch.zeros(ch[l + 1], requires_grad=True)
        p[f"enc.conv.{ci + 1}.weight"] = torch.ones(ch[l + 1], requires_grad=True)
        p[f"enc.conv.{ci + 1}.bias"] = torch.zeros(ch[l + 1], requires_grad=True)
        p[f"enc.conv.{ci + 1}.running_mean"] = torch.zeros(ch[l + 1])
        p[f"enc.conv.{ci + 1}.running_var"] = torch.ones(ch[l + 1])
        p[f"enc.conv.{ci + 1}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    p["enc.projection.weight"] = w((1, 4), 4)
    p["enc.projection.bias"] = torch.zeros(1, requires_grad=True)
    ch = nets.decoder_channels(n)
    p["dec.increase_latent_dim.weight"] = w((4 * n, n), n)
    p["dec.increase_latent_dim.bias"] = torch.zeros(4 * n, requires_grad=True)
    for l in range(4):
        ci = 5 * l
        p[f"dec.convtrans.{ci}.weight"] = w((ch[l], ch[l + 1], 3, 3), ch[l] * 9)
        p[f"dec.convtrans.{ci}.bias"] = torch.zeros(ch[l + 1], requires_grad=True)
        p[f"dec.convtrans.{ci + 1}.weight"] = torch.ones(ch[l + 1], requires_grad=True)
        p[f"dec.convtrans.{ci + 1}.bias"] = torch.zeros(ch[l + 1], requires_grad=True)
        p[f"dec.convtrans.{ci + 1}.running_mean"] = torch.zeros(ch[l + 1])
        p[f"dec.convtrans.{ci + 1}.running_var"] = torch.ones(ch[l + 1])
        p[f"dec.convtrans.{ci + 1}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    p["dec.convtrans.20.weight"] = w((1, 1, 3, 3), 9)
    p["dec.convtrans.20.bias"] = torch.zeros(1, requires_grad=True)
    return p


class OracleTrainer:
    """enc fwd/bwd, Gumbel discretisation, R decoder replicas fwd/bwd, MSE, one sampler draw,
    MMD fwd/bwd, Adam; every 10th step the GRBM quasi-NLL branch with its second draw."""

    def __init__(self, plan, n: int, replicas: int, num_reads: int, sweeps: int, prefactor: float, seed: int,
                 lr=(1e-4, 1e-3), weight_decay=0.01, h_range=(-4.0, 4.0), j_range=(-1.0, 1.0)):
        self.n, self.R, self.C, self.prefactor = n, replicas, num_reads, prefactor
        self.h_range, self.j_range = h_range, j_range
        self.p = init_params(n, seed)
        g = torch.Generator().manual_seed(seed + 1)
        self.linear = (0.05 * (2 * torch.rand(plan.n, generator=g) - 1)).requires_grad_(True)
        self.quadratic = (5.0 * (2 * torch.rand(plan.n_edges, generator=g) - 1)).requires_grad_(True)
        self.ei = torch.from_numpy(np.asarray(plan.edge_i, dtype=np.int64))
        self.ej = torch.from_numpy(np.asarray(plan.edge_j, dtype=np.int64))
        self.plan = plan
        self.sampler = OracleGibbsSampler(plan, beta=1.0 / prefactor, sweeps=sweeps, seed=seed, persistent=True)
        self.opt = torch.optim.Adam([t for t in self.p.values() if t.requires_grad], lr=lr[0], weight_decay=weight_decay)
        self.gopt = torch.optim.Adam([self.linear, self.quadratic], lr=lr[1], weight_decay=weight_decay)
        self.opt_step = 0

    def _draw(self) -> torch.Tensor:
        hs = (self.prefactor * self.linear.detach()).clamp(*self.h_range).numpy()
        Js = (self.prefactor * self.quadratic.detach()).clamp(*self.j_range).numpy()
        nodes = list(range(self.plan.n))
        h = {v: float(hs[v]) for v in nodes}
        J = {(int(a), int(b)): float(Js[e]) for e, (a, b) in enumerate(zip(self.plan.edge_i, self.plan.edge_j))}
        ss = self.sampler.sample_ising(h, J, num_reads=self.C)
        return torch.from_numpy(ss.record.sample.astype(np.float32))

    def step(self, images: torch.Tensor, gumbels: Optional[torch.Tensor] = None,
             dropout_masks: Optional[List[torch.Tensor]] = None, force_grbm: Optional[bool] = None):
        enc = {k[4:]: v for k, v in self.p.items() if k.startswith("enc.")}
        dec = {k[4:]: v for k, v in self.p.items() if k.startswith("dec.")}
        logits = nets.encoder_forward(enc, images, training=True)
        spins = plugin.gumbel_latent_to_discrete(logits, self.R, gumbels=gumbels)
        recon = nets.decoder_forward(dec, spins, training=True, dropout_masks=dropout_masks)
        self.opt.zero_grad()
        mse = torch.nn.functional.mse_loss(recon, images.unsqueeze(1).repeat(1, self.R, 1, 1, 1))
        with torch.no_grad():
            samples = self._draw()
        flat = spins.reshape(-1, self.n)
        mmd = plugin.mmd_loss(flat, samples)
        (mse + mmd).backward()
        self.opt.step()
        out = {"mse": float(mse), "mmd": float(mmd)}
        do_grbm = (self.opt_step % 10 == 0) if force_grbm is None else force_grbm
        if do_grbm:
            self.gopt.zero_grad()
            model_samples = self._draw()
            nll = plugin.grbm_energy(flat.detach(), self.linear, self.quadratic, self.ei, self.ej).mean() - \
                plugin.grbm_energy(model_samples, self.linear, self.quadratic, self.ei, self.ej).mean()
            nll.backward()
            self.gopt.step()
            out["nll"] = float(nll)
        self.opt_step += 1
        return out
