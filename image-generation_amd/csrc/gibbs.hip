// Many-chain graph-coloured block-Gibbs sampler for the GRBM prior (stands in for
// the QPU draw of /root/reference/src/model_wrapper.py:309-316).  Bit-exact
// against oracle/gibbs.py ("DVG block-Gibbs v1").
//
// Mapping: a workgroup stages the graph once into LDS (h, CSR neighbours with
// their couplings; coalesced HBM reads of the J/h tensors) and each group of LPC
// lanes of a wavefront owns one chain whose +-1 state lives in LDS for all sweeps.
// Within a colour class spins are conditionally independent, so the lanes update
// them simultaneously; classes are separated by a wavefront-level fence only (no
// workgroup barrier: chains never interact).
#include <cstdlib>

#include "common.h"
#include "graph.h"
#include "philox.h"

namespace dvg {

struct GibbsArgs {
  const int32_t *order, *class_ptr, *adj_ptr, *adj_idx, *adj_eid;
  const float *linear, *quadratic;
  int n, n_adj, n_colours;
  float prefactor, h_lo, h_hi, j_lo, j_hi, two_beta;
  int8_t* state;
  float* samples_out;
  int n_chains;
  uint32_t chain_id0, k0, k1, sweep0;
  const uint32_t* sweep0_dev;  // non-null: read the first-sweep index from device memory (graph replay)
  int n_sweeps, init;
  int passes;  // fast kernel: ceil(max_class / lanes per chain)
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// LPC = lanes per chain (16, 32 or 64); WAVES waves per workgroup.
template <int LPC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gibbs_kernel(GibbsArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int n = a.n, n_adj = a.n_adj;
  // LDS carve-up (all 4-byte aligned first, then 2-byte, then bytes)
  float* s_hs = reinterpret_cast<float*>(smem);                    // [n]
  float* s_adjJ = s_hs + n;                                        // [n_adj]
  int32_t* s_cls = reinterpret_cast<int32_t*>(s_adjJ + n_adj);     // [n_colours + 1]
  uint16_t* s_adjptr = reinterpret_cast<uint16_t*>(s_cls + a.n_colours + 1);  // [n + 1] (+pad)
  uint16_t* s_order = s_adjptr + ((n + 2) & ~1);                   // [n] (+pad)
  uint16_t* s_adjidx = s_order + ((n + 1) & ~1);                   // [n_adj] (+pad)
  int8_t* s_state = reinterpret_cast<int8_t*>(s_adjidx + ((n_adj + 1) & ~1));
  const int n_pad = (n + 15) & ~15;

  const int tid = threadIdx.x;
  constexpr int NT = WAVES * 64;
  for (int i = tid; i < n; i += NT) {
    s_hs[i] = clampf(__fmul_rn(a.prefactor, a.linear[i]), a.h_lo, a.h_hi);
    s_order[i] = (uint16_t)a.order[i];
  }
  for (int i = tid; i <= n; i += NT) s_adjptr[i] = (uint16_t)a.adj_ptr[i];
  for (int q = tid; q < n_adj; q += NT) {
    s_adjJ[q] = clampf(__fmul_rn(a.prefactor, a.quadratic[a.adj_eid[q]]), a.j_lo, a.j_hi);
    s_adjidx[q] = (uint16_t)a.adj_idx[q];
  }
  for (int i = tid; i <= a.n_colours; i += NT) s_cls[i] = a.class_ptr[i];
  __syncthreads();

  constexpr int CPW = 64 / LPC;  // chains per wave
  const int wave = tid >> 6, lane = tid & 63;
  const int sub = lane / LPC, l = lane % LPC;
  const int chain = (blockIdx.x * WAVES + wave) * CPW + sub;
  const bool valid = chain < a.n_chains;
  int8_t* st = s_state + (size_t)(wave * CPW + sub) * n_pad;
  const uint32_t cid = a.chain_id0 + (uint32_t)chain;

  // (read before the chains start: a fresh chain's start configuration is keyed by the sweep index it starts at, so
  // non-persistent draws do not all restart from one configuration)
  const uint32_t sweep0 = a.sweep0_dev ? *a.sweep0_dev : a.sweep0;
  if (valid) {
    if (a.init) {
      for (int i = l; i < n; i += LPC) {
        u32x4 r = philox4x32_10((uint32_t)i, cid, sweep0, STREAM_INIT, a.k0, a.k1);
        st[i] = (r.x >> 31) ? 1 : -1;
      }
    } else {
      const int8_t* src = a.state + (size_t)chain * n;
      for (int i = l; i < n; i += LPC) st[i] = src[i];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();

  if (valid) {
    for (uint32_t t = sweep0; t < sweep0 + (uint32_t)a.n_sweeps; ++t) {
      const uint32_t tq = t >> 2, tw = t & 3u;
      for (int k = 0; k < a.n_colours; ++k) {
        const int lo = s_cls[k], hi = s_cls[k + 1];
        for (int p = lo + l; p < hi; p += LPC) {
          const int i = s_order[p];
          float f = s_hs[i];
          const int q1 = s_adjptr[i + 1];
          for (int q = s_adjptr[i]; q < q1; ++q) {
            const float w = s_adjJ[q];
            f = __fadd_rn(f, st[s_adjidx[q]] > 0 ? w : -w);
          }
          float z = clampf(__fmul_rn(a.two_beta, f), -87.0f, 87.0f);
          const float tt = spec_exp(z);
          const u32x4 r = philox4x32_10((uint32_t)i, cid, tq, STREAM_GIBBS, a.k0, a.k1);
          const float u = u32_to_unit(pick(r, tw));
          const float b = __fmul_rn(u, __fadd_rn(1.0f, tt));
          st[i] = (b < 1.0f) ? 1 : -1;
        }
        // the next class reads what this one wrote (same wave): order LDS traffic
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
    }
    int8_t* dst = a.state + (size_t)chain * n;
    for (int i = l; i < n; i += LPC) {
      const int8_t v = st[i];
      dst[i] = v;
      if (a.samples_out) a.samples_out[(size_t)chain * n + i] = (float)v;
    }
  }
}

// Fast path for graphs with at most GIBBS_MAXS (colour class, pass) slots per lane, a pass being LPC spins of a class
// -- every graph the shipped solvers produce up to 512 spins.  Same arithmetic, same order, same random stream as
// gibbs_kernel (bit-exact); what changes is the schedule:
//   * a lane owns the same spin of every slot in every sweep, so its CSR range, local field offset and spin index
//     are read from LDS once, before the sweep loop, and live in registers;
//   * Philox words are drawn once per (spin, sweep >> 2) and serve four sweeps, as the counter layout intends,
//     instead of being recomputed every sweep;
//   * the neighbour loop issues its LDS reads eight at a time (indices and couplings, then the eight states) and only
//     the adds stay sequential -- the rolled loop paid two dependent LDS latencies per neighbour.
constexpr int GIBBS_MAXS = 12;       // slots of the common instantiations (every shipped graph up to 512 spins)
//   * (round 3) WPC = 2: TWO waves per chain (LPC = 64): a colour class of up to 128 spins is one pass of 128 lanes
//     instead of two passes of 64, which halves the dependent slot-steps of a sweep.  With few chains (c3: 256) the draw is
//     one instruction stream per chain on a quarter of the chip's SIMDs, issue-bound at one wave per SIMD (~4000
//     instructions per sweep); it runs beside the encoder forward and the MMD cannot start before it ends.  The two waves
//     of a chain meet at a workgroup barrier per colour class (every wave of the workgroup runs the same classes).
template <int LPC, int WAVES, int MAXS = GIBBS_MAXS, int WPC = 1>
__global__ __launch_bounds__(WAVES * 64) void gibbs_fast_kernel(GibbsArgs a) {
  static_assert(WPC == 1 || (LPC == 64 && WAVES % WPC == 0 && MAXS <= GIBBS_MAXS), "two waves per chain: 64-lane chains only");
  extern __shared__ __align__(16) unsigned char smem[];
  const int n = a.n, n_adj = a.n_adj;
  float* s_hs = reinterpret_cast<float*>(smem);
  float* s_adjJ = s_hs + n;
  int32_t* s_cls = reinterpret_cast<int32_t*>(s_adjJ + n_adj);
  uint16_t* s_adjptr = reinterpret_cast<uint16_t*>(s_cls + a.n_colours + 1);
  uint16_t* s_order = s_adjptr + ((n + 2) & ~1);
  uint16_t* s_adjidx = s_order + ((n + 1) & ~1);
  int8_t* s_state = reinterpret_cast<int8_t*>(s_adjidx + ((n_adj + 1) & ~1));
  const int n_pad = (n + 15) & ~15;

  const int tid = threadIdx.x;
  constexpr int NT = WAVES * 64;
  for (int i = tid; i < n; i += NT) {
    s_hs[i] = clampf(__fmul_rn(a.prefactor, a.linear[i]), a.h_lo, a.h_hi);
    s_order[i] = (uint16_t)a.order[i];
  }
  for (int i = tid; i <= n; i += NT) s_adjptr[i] = (uint16_t)a.adj_ptr[i];
  for (int q = tid; q < n_adj; q += NT) {
    s_adjJ[q] = clampf(__fmul_rn(a.prefactor, a.quadratic[a.adj_eid[q]]), a.j_lo, a.j_hi);
    s_adjidx[q] = (uint16_t)a.adj_idx[q];
  }
  for (int i = tid; i <= a.n_colours; i += NT) s_cls[i] = a.class_ptr[i];
  __syncthreads();

  constexpr int CPW = 64 / LPC;
  constexpr int LPCE = LPC * WPC;  // lanes that work on one chain
  const int wave = tid >> 6, lane = tid & 63;
  const int sub = lane / LPC, l = WPC == 1 ? lane % LPC : lane + 64 * (wave % WPC);
  const int chain = WPC == 1 ? (blockIdx.x * WAVES + wave) * CPW + sub : blockIdx.x * (WAVES / WPC) + wave / WPC;
  const bool valid = chain < a.n_chains;
  int8_t* st = s_state + (size_t)(WPC == 1 ? wave * CPW + sub : wave / WPC) * n_pad;
  const uint32_t cid = a.chain_id0 + (uint32_t)chain;

  // (read before the chains start: a fresh chain's start configuration is keyed by the sweep index it starts at, so
  // non-persistent draws do not all restart from one configuration)
  const uint32_t sweep0 = a.sweep0_dev ? *a.sweep0_dev : a.sweep0;
  if (valid) {
    if (a.init) {
      for (int i = l; i < n; i += LPCE) {
        u32x4 r = philox4x32_10((uint32_t)i, cid, sweep0, STREAM_INIT, a.k0, a.k1);
        st[i] = (r.x >> 31) ? 1 : -1;
      }
    } else {
      const int8_t* src = a.state + (size_t)chain * n;
      for (int i = l; i < n; i += LPCE) st[i] = src[i];
    }
  }
  if constexpr (WPC > 1) {
    __syncthreads();  // (both waves of a chain wrote its start state; invalid chains keep the barriers company below)
  } else {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (!valid) return;
  }

  const int passes = a.passes, n_slots = a.n_colours * passes;
  {
  // this lane's spin in each (colour, pass) slot (-1: none), with its CSR range and clamped field offset
  int sp[MAXS], q0[MAXS], q1[MAXS];
  float hs[MAXS];
#pragma unroll
  for (int k = 0; k < MAXS; ++k) {
    sp[k] = -1; q0[k] = 0; q1[k] = 0; hs[k] = 0.f;
    if (k < n_slots) {
      const int col = k / passes, pass = k - col * passes;
      const int p = s_cls[col] + pass * LPCE + l;
      if (valid && p < s_cls[col + 1]) {
        const int i = s_order[p];
        sp[k] = i; q0[k] = s_adjptr[i]; q1[k] = s_adjptr[i + 1]; hs[k] = s_hs[i];
      }
    }
  }
  u32x4 rr[MAXS];
  for (uint32_t t = sweep0; t < sweep0 + (uint32_t)a.n_sweeps; ++t) {
    const uint32_t tq = t >> 2, tw = t & 3u;
    if (t == sweep0 || tw == 0u) {
#pragma unroll
      for (int k = 0; k < MAXS; ++k)
        if (sp[k] >= 0) rr[k] = philox4x32_10((uint32_t)sp[k], cid, tq, STREAM_GIBBS, a.k0, a.k1);
    }
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
      if (k < n_slots) {
        if (sp[k] >= 0) {
          float f = hs[k];
          const int qe = q1[k];
          for (int q = q0[k]; q < qe; q += 8) {
            int idx[8];
            float w[8];
            int8_t sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int qq = q + u < qe ? q + u : qe - 1;  // clamped: the batch is one straight line of LDS reads
              idx[u] = s_adjidx[qq];
              w[u] = s_adjJ[qq];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) sv[u] = st[idx[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (q + u < qe) f = __fadd_rn(f, sv[u] > 0 ? w[u] : -w[u]);
          }
          const float z = clampf(__fmul_rn(a.two_beta, f), -87.0f, 87.0f);
          const float tt = spec_exp(z);
          const float u01 = u32_to_unit(pick(rr[k], tw));
          const float b = __fmul_rn(u01, __fadd_rn(1.0f, tt));
          st[sp[k]] = (b < 1.0f) ? 1 : -1;
        }
        // the next class reads what this one wrote (same wave, or the chain's two waves): order LDS traffic.  (Passes of
        // one class are independent of each other, so a fence between them is harmless.)
        if constexpr (WPC > 1) {
          __syncthreads();
        } else {
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
  }
  if (!valid) return;
  int8_t* dst = a.state + (size_t)chain * n;
  for (int i = l; i < n; i += LPCE) {
    const int8_t v = st[i];
    dst[i] = v;
    if (a.samples_out) a.samples_out[(size_t)chain * n + i] = (float)v;
  }
}

static size_t gibbs_lds_bytes(int n, int n_adj, int n_colours, int chains_per_block) {
  size_t b = 0;
  b += sizeof(float) * (size_t)(n + n_adj);
  b += sizeof(int32_t) * (size_t)(n_colours + 1);
  b += sizeof(uint16_t) * (size_t)(((n + 2) & ~1) + ((n + 1) & ~1) + ((n_adj + 1) & ~1));
  b += (size_t)chains_per_block * ((n + 15) & ~15);
  return b;
}

// dvg_gibbs_launch_info: the dispatch below runs with a probe set and reports its launch geometry instead of launching
struct GibbsProbe { int workgroups; int threads; size_t lds; };
static thread_local GibbsProbe* g_gibbs_probe = nullptr;

// two waves per chain (fast kernel only): 4-wave workgroups of two chains
static int launch_gibbs_wpc2(GibbsArgs a, hipStream_t s, int max_class) {
  constexpr int WAVES = 4, CPB = 2;
  a.passes = (max_class + 127) / 128;
  const size_t lds = gibbs_lds_bytes(a.n, a.n_adj, a.n_colours, CPB);
  auto kern = gibbs_fast_kernel<64, WAVES, GIBBS_MAXS, 2>;
  if (lds > 64 * 1024)
    DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (int)ceil_div(a.n_chains, CPB);
  if (g_gibbs_probe) { *g_gibbs_probe = GibbsProbe{grid, WAVES * 64, lds}; return DVG_OK; }
  DVG_LAUNCH_WORK(K_GIBBS, (double)a.n_chains * a.n * a.n_sweeps, kern, dim3(grid), dim3(WAVES * 64), lds, s, a);
  return DVG_OK;
}

template <int LPC, int WAVES, int MAXS = GIBBS_MAXS>
static int launch_gibbs(GibbsArgs a, hipStream_t s, bool fast, int max_class) {
  a.passes = (max_class + LPC - 1) / LPC;
  fast = fast && a.n_colours * a.passes <= MAXS;
  constexpr int CPB = WAVES * (64 / LPC);
  const size_t lds = gibbs_lds_bytes(a.n, a.n_adj, a.n_colours, CPB);
  if (lds > 160 * 1024) {
    set_error("gibbs: graph (n=%d, 2|E|=%d) needs %zu B of LDS > 160 KiB", a.n, a.n_adj, lds);
    return DVG_E_UNSUPPORTED;
  }
  const int grid = (int)ceil_div(a.n_chains, CPB);
  if (g_gibbs_probe) { *g_gibbs_probe = GibbsProbe{grid, WAVES * 64, lds}; return DVG_OK; }
  auto kern = fast ? gibbs_fast_kernel<LPC, WAVES, MAXS> : gibbs_kernel<LPC, WAVES>;
  if (lds > 64 * 1024)
    DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // work = spin updates of the draw (bench.py's sampler roofline)
  DVG_LAUNCH_WORK(K_GIBBS, (double)a.n_chains * a.n * a.n_sweeps, kern, dim3(grid), dim3(WAVES * 64), lds, s, a);
  return DVG_OK;
}

}  // namespace dvg

using namespace dvg;

static int gibbs_dispatch(const dvg_graph_t* g, GibbsArgs& a, int n_chains, hipStream_t s);

extern "C" int dvg_gibbs_sample(const dvg_graph_t* g, const float* linear, const float* quadratic,
                                float prefactor, float h_lo, float h_hi, float j_lo, float j_hi,
                                float beta, int8_t* state, int n_chains, uint32_t chain_id0,
                                uint64_t seed, uint32_t sweep0, int n_sweeps, int init,
                                float* samples_out, const dvg_step_state_t* dyn, dvg_stream_t stream) {
  DVG_REQUIRE(g && linear && quadratic && state, "gibbs: null argument");
  DVG_REQUIRE(n_chains > 0 && n_sweeps >= 0, "gibbs: n_chains=%d n_sweeps=%d", n_chains, n_sweeps);
  DVG_REQUIRE(g->n <= 65535 && g->n_adj <= 65535, "gibbs: graph too large (n=%d, 2|E|=%d)", g->n, g->n_adj);
  GibbsArgs a;
  a.order = g->order; a.class_ptr = g->class_ptr; a.adj_ptr = g->adj_ptr;
  a.adj_idx = g->adj_idx; a.adj_eid = g->adj_eid;
  a.linear = linear; a.quadratic = quadratic;
  a.n = g->n; a.n_adj = g->n_adj; a.n_colours = g->n_colours;
  a.prefactor = prefactor; a.h_lo = h_lo; a.h_hi = h_hi; a.j_lo = j_lo; a.j_hi = j_hi;
  a.two_beta = 2.0f * beta;
  a.state = state; a.samples_out = samples_out; a.n_chains = n_chains;
  a.chain_id0 = chain_id0; a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32);
  a.sweep0 = sweep0; a.n_sweeps = n_sweeps; a.init = init;
  a.sweep0_dev = dyn ? &dyn->sweep0 : nullptr;
  return gibbs_dispatch(g, a, n_chains, (hipStream_t)stream);
}

// The launch geometry dvg_gibbs_sample would use for `n_chains` chains on this graph (nothing is launched): workgroups,
// threads per workgroup, LDS bytes per workgroup.  The host side sizes what runs BESIDE the draw with it (the encoder
// forward's Winograd launches take whole CUs: ModelWrapper leaves the draw's CUs out of their persistent grid).
extern "C" int dvg_gibbs_launch_info(const dvg_graph_t* g, int n_chains, int* workgroups, int* threads, size_t* lds_bytes) {
  DVG_REQUIRE(g && n_chains > 0, "gibbs_launch_info: null graph / no chains");
  GibbsArgs a{};
  a.n = g->n; a.n_adj = g->n_adj; a.n_colours = g->n_colours; a.n_chains = n_chains;
  GibbsProbe pr{0, 0, 0};
  g_gibbs_probe = &pr;
  const int rc = gibbs_dispatch(g, a, n_chains, nullptr);
  g_gibbs_probe = nullptr;
  DVG_TRY(rc);
  if (workgroups) *workgroups = pr.workgroups;
  if (threads) *threads = pr.threads;
  if (lds_bytes) *lds_bytes = pr.lds;
  return DVG_OK;
}

static int gibbs_dispatch(const dvg_graph_t* g, GibbsArgs& a, int n_chains, hipStream_t s) {
  // lanes per chain: the smallest of 16/32/64 that covers the largest colour class in one pass
  const int mc = g->max_class;
  const bool big = gibbs_lds_bytes(g->n, g->n_adj, g->n_colours, 4) > 72 * 1024;
  // (option gibbs_generic = 1 forces the rolled reference schedule: A/B runs and the bit-exactness test of the fast one)
  const bool force_generic = opt(OPT_GIBBS_GENERIC) != 0;
  const bool fast = !force_generic;
  // Waves per workgroup (option gibbs_waves = 1, 2, 4 or 8 overrides for tuning runs).  Measured on the c2 step with the
  // draw overlapped with the encoder forward: 4 -> 1.237 ms, 8 -> 1.273, 16 -> 1.391: the sweep loop does contend for
  // issue slots, fatter workgroups do not pay for the CUs they free.
  const int waves_env = (int)opt(OPT_GIBBS_WAVES);
  // Small graphs with few chains (c2: 128 spins, 256 chains -> 32 four-wave workgroups on 256 CUs): one wave per
  // workgroup spreads the draw over four times as many CUs at a few KB of tables each (c2 step 1.082 -> 1.060 ms).  Larger
  // graphs keep four waves: every extra workgroup stages its own ~50 KB copy of the tables and takes that LDS from the
  // encoder's convolutions that run beside the draw (c3: 19.15 ms with four waves, 19.4 with two, 19.7 with one).
  const bool small = gibbs_lds_bytes(g->n, g->n_adj, g->n_colours, 0) <= 16 * 1024 && n_chains <= 1024;
  const int waves = waves_env ? waves_env : (small ? 1 : 4);
  // Large graphs (c5: 1024 spins, 2|E| = 16 K -> ~105 KB of tables): one workgroup per CU fits, so the workgroup must
  // carry the CU's whole latency-hiding: 16 waves = 16 chains share one LDS copy of the graph (2 waves left 7/8 of
  // the issue slots empty: 7.3 ms per 2048-chain, 50-sweep draw)
  if (big) {
    const int wv = waves_env ? waves_env : 16;
    if (wv <= 2) return launch_gibbs<64, 2>(a, s, fast, mc);
    if (wv <= 4) return launch_gibbs<64, 4>(a, s, fast, mc);
    if (wv <= 8) return launch_gibbs<64, 8>(a, s, fast, mc);
    // (An 8-wave register-resident form of the fast schedule for these graphs -- 24 slots, `gibbs_bigfast` -- existed in
    // rounds 2-3: the faster draw alone, 1.49 against 2.20 ms at the c5 slice, but its ~110 KB LDS footprint on EVERY CU
    // starved the convolutions beside it: c5 step 5.0 against 4.0 ms.  Retired in round 4.)
    return launch_gibbs<64, 16>(a, s, fast, mc);
  }
#define DVG_GIBBS_DISPATCH(LPC)                                              \
  switch (waves) {                                                           \
    case 1: return launch_gibbs<LPC, 1>(a, s, fast, mc);                     \
    case 2: return launch_gibbs<LPC, 2>(a, s, fast, mc);                     \
    case 8: return launch_gibbs<LPC, 8>(a, s, fast, mc);                     \
    default: return launch_gibbs<LPC, 4>(a, s, fast, mc);                    \
  }
  // classes of 65..128 spins with few chains (c3: 512 spins, 256 chains): two waves per chain, one pass per class --
  // option gibbs_waves_per_chain = 2.  Alone it is the faster draw (1.79 -> 1.38 ms); inside a training step it runs
  // beside the encoder forward on twice the workgroups and the STEP does not move (c3 11.74 vs 11.76 ms, the encoder's
  // GEMMs 433 -> 450-500 us each in-situ), so the default stays one wave per chain.
  if (fast && !waves_env && mc > 64 && mc <= 128 && g->n_colours <= GIBBS_MAXS && n_chains <= 1024 &&
      gibbs_lds_bytes(g->n, g->n_adj, g->n_colours, 2) <= 80 * 1024 && opt(OPT_GIBBS_WAVES_PER_CHAIN) == 2)
    return launch_gibbs_wpc2(a, s, mc);
  if (mc <= 16) { DVG_GIBBS_DISPATCH(16) }
  if (mc <= 32) { DVG_GIBBS_DISPATCH(32) }
  DVG_GIBBS_DISPATCH(64)
#undef DVG_GIBBS_DISPATCH
}
