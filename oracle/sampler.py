"""Oracle-side local sampler object: ``sample_ising(h, J, **kwargs)`` over the
block-Gibbs definition (oracle/gibbs.py / gibbs_ref.c).  Test infrastructure.

Mirrors the call the reference makes on its QPU sampler
(/root/reference/src/utils/common.py:128-138 builds it;
the plugin's ``GraphRestrictedBoltzmannMachine.sample`` calls
``sampler.sample_ising(h, J, num_reads=..., answer_mode="raw", auto_scale=False,
annealing_time=..., label=...)``).
"""
import numpy as np

from . import cref
from .gibbs import build_csr
from .plugin import SampleSetShim


class OracleGibbsSampler:
    def __init__(self, plan, beta, sweeps, seed, persistent=True, chain_offset=0):
        self.plan, self.beta, self.sweeps, self.seed = plan, float(beta), int(sweeps), int(seed)
        self.persistent = persistent
        self.chain_offset = chain_offset
        self.state = None
        self.sweep_count = 0
        self.calls = 0

    def sample_ising(self, h, J, num_reads=1, **_ignored):
        p = self.plan
        nodes = list(h.keys())
        hs = np.asarray([h[v] for v in nodes], dtype=np.float32)
        Js = np.asarray([J[(nodes[i], nodes[j])] for i, j in zip(p.edge_i, p.edge_j)], dtype=np.float32)
        chain_ids = np.arange(num_reads, dtype=np.uint32) + np.uint32(self.chain_offset)
        if self.state is None or not self.persistent or self.state.shape[0] != num_reads:
            self.state = cref.init_state(chain_ids, p.n, self.seed, self.sweep_count)
        self.state = cref.gibbs_sweeps(
            self.state, chain_ids, hs, Js, self.beta, p.order, p.class_ptr, p.adj_ptr, p.adj_idx, p.adj_eid,
            self.seed, self.sweep_count, self.sweeps,
        )
        self.sweep_count += self.sweeps
        self.calls += 1
        return SampleSetShim(self.state.copy(), nodes)
