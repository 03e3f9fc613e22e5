"""Counterpart of /root/reference/src/utils/persistent_qpu_sampler.py.

The reference's helper is meant to cache samples in a FIFO, but it resets its own state at
the top of every ``sample`` call (:61-63), so the resampling branch always runs and the deque
branch (:79-88) is dead.  The observable behaviour -- one fresh ``grbm.sample(..., as_tensor=False)``
per call, ``deque`` holding the latest draw -- is what is reproduced here; ``push_to_deque`` is kept
because it is part of the module's surface.
"""
from __future__ import annotations

from typing import Optional

import torch


def push_to_deque(deque: torch.Tensor, x: torch.Tensor, deque_size: Optional[int] = None, dim: int = 0) -> torch.Tensor:
    """FIFO push along ``dim``: keep the newest ``deque_size`` entries of ``cat(deque, x)``
    (same results as /root/reference/src/utils/persistent_qpu_sampler.py:12-38)."""
    if deque_size is None:
        deque_size = deque.shape[dim]
    n_in, n_old = x.shape[dim], deque.shape[dim]
    # room left for old entries once the newest `deque_size` inputs are in
    keep_old = min(max(deque_size - n_in, 0), n_old)
    old = deque.narrow(dim, n_old - keep_old, keep_old)
    keep_new = min(n_in, deque_size)
    new = x.narrow(dim, n_in - keep_new, keep_new)
    return torch.cat((old, new), dim=dim)


class PersistentQPUSampleHelper:
    """Draws the model samples of the GRBM step (always resamples, like the reference)."""

    def __init__(self, max_deque_size: int, iterations_before_resampling: int):
        self.current_deque_size = 0
        self.max_deque_size = max_deque_size
        self.iterations_before_resampling = iterations_before_resampling
        self.iterations_since_last_resampling = 0
        self.deque = None
        self.sample_set = None

    def sample(self, prefactor, grbm, sampler, sampler_kwargs, linear_range, quadratic_range):
        with torch.no_grad():
            self.sample_set = grbm.sample(sampler, prefactor=prefactor, linear_range=linear_range,
                                          quadratic_range=quadratic_range, sample_params=sampler_kwargs, as_tensor=False)
        self.deque = grbm.sampleset_to_tensor(self.sample_set)
        self.current_deque_size = self.deque.shape[0]
        self.iterations_since_last_resampling = 0
        return self.sample_set
