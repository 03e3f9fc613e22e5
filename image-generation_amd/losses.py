"""Counterpart of /root/reference/src/losses.py."""
from __future__ import annotations

import torch


def nll_loss(spins, grbm, sampler, sampler_kwargs, linear_range, quadratic_range, prefactor,
             persistent_qpu_sample_helper, sample_set=None):
    """Quasi-objective whose gradient is the NLL gradient of the data under the GRBM
    (/root/reference/src/losses.py:38-63): mean E(spins) - mean E(model samples).
    Returns ``(nll, sample_set)`` like the reference does."""
    sample_set = persistent_qpu_sample_helper.sample(prefactor, grbm, sampler, sampler_kwargs, linear_range,
                                                     quadratic_range)
    samples = grbm.sampleset_to_tensor(sample_set, device=spins.device)
    spins = spins.reshape(-1, spins.shape[-1])
    nll = torch.mean(grbm(spins)) - torch.mean(grbm(samples))
    return nll, sample_set
