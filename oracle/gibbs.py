"""Block-Gibbs sampler for a graph-restricted Boltzmann machine: THE DEFINITION.

Test infrastructure (see oracle/__init__.py).  The reference draws its
negative-phase samples from a D-Wave QPU
(/root/reference/src/utils/common.py:123-138, call sites
/root/reference/src/model_wrapper.py:309-316 and
/root/reference/src/utils/persistent_qpu_sampler.py:71-78); it contains no
offline sampler, so there is nothing to restate.  This file defines the
stand-in ("DVG block-Gibbs v1") and the HIP kernel
(image-generation_amd/csrc/gibbs.hip) must reproduce it bit for bit.

Target distribution  p(s) ∝ exp(-beta * (sum_i hs_i s_i + sum_e Js_e s_i s_j)),
with hs = clamp(prefactor*h, h_range), Js = clamp(prefactor*J, j_range)  (what
`GraphRestrictedBoltzmannMachine.to_ising` hands the QPU).

One sweep = for each colour class k (in order): every spin i of the class is
redrawn from its conditional  P(s_i=+1 | rest) = sigmoid(-2 beta f_i),
f_i = hs_i + sum_{j in adj(i)} Js_ij s_j  (neighbours summed in CSR order).

Arithmetic spec (float32, every operation individually rounded, no FMA):
    f   = hs[i];  for each CSR neighbour: f = f + (s_j > 0 ? Js : -Js)
    z   = clamp((2*beta) * f, -87, 87)              # = -logit of P(+1)
    t   = spec_exp(z)
    u   = (philox(ctr=(i, chain, sweep>>2, STREAM_GIBBS), key=seed)[sweep&3] >> 8) * 2^-24
    s_i = +1  if  u * (1 + t) < 1  else  -1
spec_exp(z): k = rint(z*log2e); r = z - k*LN2_HI - k*LN2_LO;
    p = Horner degree-6 Taylor in r;  result = p * 2^k (2^k built from bits).
"""
import numpy as np

from .philox import (
    STREAM_GIBBS,
    STREAM_INIT,
    philox4x32_10,
    u32_to_unit_float,
)

f32 = np.float32

LOG2E = f32(1.4426950408889634)
LN2_HI = f32(0.693359375)  # 0x3f318000: 9 significant bits -> k*LN2_HI exact for |k| < 2^15
LN2_LO = f32(-2.12194440e-4)
EXP_C = [f32(1.0 / 720.0), f32(1.0 / 120.0), f32(1.0 / 24.0), f32(1.0 / 6.0), f32(0.5), f32(1.0), f32(1.0)]
Z_CLAMP = f32(87.0)


def spec_exp(z):
    """Bit-specified exp for float32 arrays with |z| <= 87."""
    z = np.asarray(z, dtype=f32)
    k = np.rint(z * LOG2E).astype(f32)
    r = (z - k * LN2_HI).astype(f32)
    r = (r - k * LN2_LO).astype(f32)
    p = np.full_like(r, EXP_C[0])
    for c in EXP_C[1:]:
        p = (p * r).astype(f32)
        p = (p + c).astype(f32)
    two_k = ((k.astype(np.int32) + 127) << 23).astype(np.uint32).view(f32)
    return (p * two_k).astype(f32)


def scaled_fields(h, J, prefactor, h_range=None, j_range=None):
    """hs, Js = clamp(prefactor * (h, J)) in float32 (one rounding each)."""
    hs = (np.asarray(h, dtype=f32) * f32(prefactor)).astype(f32)
    Js = (np.asarray(J, dtype=f32) * f32(prefactor)).astype(f32)
    if h_range is not None:
        hs = np.minimum(np.maximum(hs, f32(h_range[0])), f32(h_range[1]))
    if j_range is not None:
        Js = np.minimum(np.maximum(Js, f32(j_range[0])), f32(j_range[1]))
    return hs, Js


def init_state(chain_ids, n, seed, sweep0=0):
    """Random +-1 start: bit 31 of philox(ctr=(i, chain, sweep0, STREAM_INIT))[0].  ``sweep0`` is the index of the
    first sweep the chains will run (0 for chains started at the beginning of a run), so a sampler that restarts its
    chains on every draw (non-persistent mode) starts each draw from a different configuration."""
    chain_ids = np.asarray(chain_ids, dtype=np.uint32)
    i = np.arange(n, dtype=np.uint32)[None, :]
    r0, _, _, _ = philox4x32_10(
        i, chain_ids[:, None], np.uint32(int(sweep0) & 0xFFFFFFFF), np.uint32(STREAM_INIT), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    )
    return np.where((r0 >> np.uint32(31)) != 0, 1, -1).astype(np.int8)


def gibbs_sweeps(
    state,
    chain_ids,
    hs,
    Js,
    beta,
    order,
    class_ptr,
    adj_ptr,
    adj_idx,
    adj_eid,
    seed,
    sweep0,
    nsweeps,
):
    """Run `nsweeps` sweeps in place on `state` (C, n) int8; vectorised over chains.

    `sweep0` is the global index of the first sweep (persistent chains keep
    counting so the random stream never repeats).
    """
    state = np.ascontiguousarray(state, dtype=np.int8)
    C, n = state.shape
    chain_ids = np.asarray(chain_ids, dtype=np.uint32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    two_beta = f32(f32(2.0) * f32(beta))
    ncol = len(class_ptr) - 1
    for t in range(sweep0, sweep0 + nsweeps):
        for k in range(ncol):
            spins = order[class_ptr[k] : class_ptr[k + 1]]
            new_vals = np.empty((C, len(spins)), dtype=np.int8)
            for a, i in enumerate(spins):
                f = np.full(C, hs[i], dtype=f32)
                for q in range(adj_ptr[i], adj_ptr[i + 1]):
                    j = adj_idx[q]
                    w = Js[adj_eid[q]]
                    f = (f + np.where(state[:, j] > 0, w, -w).astype(f32)).astype(f32)
                z = (two_beta * f).astype(f32)
                z = np.minimum(np.maximum(z, -Z_CLAMP), Z_CLAMP)
                tt = spec_exp(z)
                r = philox4x32_10(np.uint32(i), chain_ids, np.uint32(t >> 2), np.uint32(STREAM_GIBBS), k0, k1)[t & 3]
                u = u32_to_unit_float(r)
                acc = (u * (f32(1.0) + tt).astype(f32)).astype(f32) < f32(1.0)
                new_vals[:, a] = np.where(acc, 1, -1)
            # a colour class is an independent set: simultaneous update == sequential update
            state[:, spins] = new_vals
    return state


def build_csr(n, edge_i, edge_j):
    """CSR adjacency in edge-list order (neighbour order = order of appearance)."""
    edge_i = np.asarray(edge_i, dtype=np.int64)
    edge_j = np.asarray(edge_j, dtype=np.int64)
    nbrs = [[] for _ in range(n)]
    for e, (a, b) in enumerate(zip(edge_i.tolist(), edge_j.tolist())):
        nbrs[a].append((b, e))
        nbrs[b].append((a, e))
    adj_ptr = np.zeros(n + 1, dtype=np.int32)
    adj_idx, adj_eid = [], []
    for i in range(n):
        adj_ptr[i + 1] = adj_ptr[i] + len(nbrs[i])
        for b, e in nbrs[i]:
            adj_idx.append(b)
            adj_eid.append(e)
    return adj_ptr, np.asarray(adj_idx, dtype=np.int32), np.asarray(adj_eid, dtype=np.int32)


def energy(state, h, J, edge_i, edge_j):
    """E(s) = sum h_i s_i + sum_e J_e s_i s_j in float64 (for statistics tests)."""
    s = np.asarray(state, dtype=np.float64)
    return s @ np.asarray(h, dtype=np.float64) + (s[:, edge_i] * s[:, edge_j]) @ np.asarray(J, dtype=np.float64)


__all__ = [
    "spec_exp",
    "scaled_fields",
    "init_state",
    "gibbs_sweeps",
    "build_csr",
    "energy",
    "STREAM_GIBBS",
    "STREAM_INIT",
]
