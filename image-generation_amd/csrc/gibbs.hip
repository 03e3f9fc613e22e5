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
#include <utility>

#include "common.h"
#include "graph.h"
#include "philox.h"

namespace dvg {

struct GibbsArgs {
  const int32_t *order, *class_ptr, *adj_idx, *adj_eid;
  const int32_t *adj_row, *adj_src4;  // padded-row image (graph.h)
  const int32_t *lane_spin, *lane_eid;  // lane-major image (graph.h); null: none for this graph
  const uint16_t* lane_off;
  int n_rows;                           // fast kernel: rows of that image
  const float *linear, *quadratic;
  int n, n_batches, max_batches, n_colours;
  float prefactor, h_lo, h_hi, j_lo, j_hi, two_beta;
  int8_t* state;
  float* samples_out;
  int n_chains;
  uint32_t chain_id0, k0, k1, sweep0;
  const uint32_t* sweep0_dev;  // non-null: read the first-sweep index from device memory (graph replay)
  int n_sweeps, init;
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// ---------------------------------------------------------------- the LDS image and the neighbour sum (round 4)
// The local field of a spin is  f = h_i + sum over its CSR row, IN ROW ORDER, of (s_j > 0 ? +J : -J)  in float32, each add
// rounded (oracle/gibbs.py, "DVG block-Gibbs v1").  Rounds 1-3 spent ~12 VALU and 3.4 LDS instructions per neighbour on it
// (index clamping, three address computations, compare / select / add); this form spends 3 and 1.5, for the same bits:
//   * the state of a chain is kept as float16 +-1 in LDS, and the add is ONE v_fma_mix_f32:  fma(s, J, f) = rn(s J + f)
//     with s J = +-J exact, i.e. exactly rn(f +- J) -- the same single rounding as the add it replaces;
//   * rows are padded to whole batches of 4 entries (J = 0, any valid state: rn(f +- 0) = f; only the LAST partial sum
//     can see a pad, and a -0 turned +0 there is the same exp(0)), so a batch is one ds_read_b128 of couplings and one
//     ds_read_b64 of four 16-bit state offsets at immediate offsets from one address -- no per-neighbour index math;
//   * every lane walks the same NB batches per round (NB = 5: 20 neighbours, the degree of the Zephyr graphs); a lane
//     whose row is shorter reads the image's all-zero batch instead (one compare / select per BATCH), so the loads of a
//     round are one straight line: 2 NB reads, then 4 NB state reads, then 4 NB dependent fmas.
typedef float gf32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t gu32x2 __attribute__((ext_vector_type(2)));
constexpr int GIBBS_NB = 5;

struct GibbsLds {
  float* hs;         // [n]  clamped prefactor * h
  float* w;          // [4 (n_batches + 1)]  clamped prefactor * J per padded entry; the last batch is the zero batch
  uint32_t* row;     // [n]  (first batch << 8) | batches
  int32_t* cls;      // [n_colours + 1]
  uint16_t* off;     // [4 (n_batches + 1)]  byte offset of the neighbour's state inside a chain's state row
  uint16_t* order;   // [n]
  _Float16* state;   // [chains][n_pad]
};
__host__ __device__ __forceinline__ size_t gibbs_lds_layout(int n, int n_batches, int n_colours, size_t off[7]) {
  size_t b = 0;
  off[0] = b; b += sizeof(float) * (size_t)((n + 3) & ~3);
  off[1] = b; b += sizeof(float) * 4 * (size_t)(n_batches + 1);
  off[2] = b; b += sizeof(uint32_t) * (size_t)n;
  off[3] = b; b += sizeof(int32_t) * (size_t)(n_colours + 1);
  b = (b + 7) & ~(size_t)7;
  off[4] = b; b += sizeof(uint16_t) * 4 * (size_t)(n_batches + 1);
  off[5] = b; b += sizeof(uint16_t) * (size_t)((n + 1) & ~1);
  off[6] = b;
  return b;
}
__device__ __forceinline__ GibbsLds gibbs_carve(unsigned char* smem, int n, int n_batches, int n_colours) {
  size_t o[7];
  gibbs_lds_layout(n, n_batches, n_colours, o);
  GibbsLds L;
  L.hs = reinterpret_cast<float*>(smem + o[0]);
  L.w = reinterpret_cast<float*>(smem + o[1]);
  L.row = reinterpret_cast<uint32_t*>(smem + o[2]);
  L.cls = reinterpret_cast<int32_t*>(smem + o[3]);
  L.off = reinterpret_cast<uint16_t*>(smem + o[4]);
  L.order = reinterpret_cast<uint16_t*>(smem + o[5]);
  L.state = reinterpret_cast<_Float16*>(smem + o[6]);
  return L;
}

// stage the graph (coalesced reads of h, J and the padded-row image), all threads of the workgroup; ends with a barrier
template <int NT, int OFF_SCALE = 2>  // OFF_SCALE: bytes between two spins of a chain's state (2: a row per chain; 32: [spin][16 chains])
__device__ __forceinline__ void gibbs_stage(const GibbsArgs& a, const GibbsLds& L, int tid) {
  const int n = a.n;
  for (int i = tid; i < n; i += NT) {
    L.hs[i] = clampf(__fmul_rn(a.prefactor, a.linear[i]), a.h_lo, a.h_hi);
    L.order[i] = (uint16_t)a.order[i];
    L.row[i] = (uint32_t)a.adj_row[i];
  }
  const int n_ent = 4 * a.n_batches;
  for (int q = tid; q < n_ent + 4; q += NT) {
    const int src = q < n_ent ? a.adj_src4[q] : -1;
    L.w[q] = src >= 0 ? clampf(__fmul_rn(a.prefactor, a.quadratic[a.adj_eid[src]]), a.j_lo, a.j_hi) : 0.0f;
    L.off[q] = src >= 0 ? (uint16_t)(OFF_SCALE * a.adj_idx[src]) : (uint16_t)0;
  }
  for (int i = tid; i <= a.n_colours; i += NT) L.cls[i] = a.class_ptr[i];
  __syncthreads();
}

__device__ __forceinline__ float gibbs_signed_add(_Float16 s, float w, float f) {
  // rn(s w + f), s = +-1: v_fma_mix_f32 with a float16 first operand
  return __builtin_fmaf((float)s, w, f);
}

// f + (the row's signed couplings, in row order); max_batches is uniform (the graph's longest row)
__device__ __forceinline__ float gibbs_field(float f, uint32_t rowinfo, const GibbsLds& L, const _Float16* st,
                                             int zero_batch, int max_batches) {
  const int first = (int)(rowinfo >> 8), nb = (int)(rowinfo & 255u);
  const unsigned char* stb = reinterpret_cast<const unsigned char*>(st);
  for (int b0 = 0; b0 < max_batches; b0 += GIBBS_NB) {
    gf32x4 w[GIBBS_NB];
    gu32x2 o[GIBBS_NB];
#pragma unroll
    for (int j = 0; j < GIBBS_NB; ++j) {
      const int bi = b0 + j < nb ? first + b0 + j : zero_batch;
      w[j] = *reinterpret_cast<const gf32x4*>(L.w + 4 * bi);
      o[j] = *reinterpret_cast<const gu32x2*>(L.off + 4 * bi);
    }
    _Float16 h[GIBBS_NB][4];
#pragma unroll
    for (int j = 0; j < GIBBS_NB; ++j) {
      h[j][0] = *reinterpret_cast<const _Float16*>(stb + (o[j][0] & 0xffffu));
      h[j][1] = *reinterpret_cast<const _Float16*>(stb + (o[j][0] >> 16));
      h[j][2] = *reinterpret_cast<const _Float16*>(stb + (o[j][1] & 0xffffu));
      h[j][3] = *reinterpret_cast<const _Float16*>(stb + (o[j][1] >> 16));
    }
#pragma unroll
    for (int j = 0; j < GIBBS_NB; ++j) {
      f = gibbs_signed_add(h[j][0], w[j][0], f);
      f = gibbs_signed_add(h[j][1], w[j][1], f);
      f = gibbs_signed_add(h[j][2], w[j][2], f);
      f = gibbs_signed_add(h[j][3], w[j][3], f);
    }
  }
  return f;
}

__device__ __forceinline__ _Float16 gibbs_decide(float f, float two_beta, uint32_t word) {
  const float z = clampf(__fmul_rn(two_beta, f), -87.0f, 87.0f);
  const float tt = spec_exp(z);
  const float b = __fmul_rn(u32_to_unit(word), __fadd_rn(1.0f, tt));
  return (b < 1.0f) ? (_Float16)1.0f : (_Float16)-1.0f;
}

// LPC = lanes per chain (16, 32 or 64); WAVES waves per workgroup.  The rolled schedule: any graph; Philox per update.
template <int LPC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gibbs_kernel(GibbsArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int n = a.n;
  const GibbsLds L = gibbs_carve(smem, n, a.n_batches, a.n_colours);
  const int n_pad = (n + 15) & ~15;
  const int tid = threadIdx.x;
  gibbs_stage<WAVES * 64>(a, L, tid);

  constexpr int CPW = 64 / LPC;  // chains per wave
  const int wave = tid >> 6, lane = tid & 63;
  const int sub = lane / LPC, l = lane % LPC;
  const int chain = (blockIdx.x * WAVES + wave) * CPW + sub;
  const bool valid = chain < a.n_chains;
  _Float16* st = L.state + (size_t)(wave * CPW + sub) * n_pad;
  const uint32_t cid = a.chain_id0 + (uint32_t)chain;

  // (read before the chains start: a fresh chain's start configuration is keyed by the sweep index it starts at, so
  // non-persistent draws do not all restart from one configuration)
  const uint32_t sweep0 = a.sweep0_dev ? *a.sweep0_dev : a.sweep0;
  if (valid) {
    if (a.init) {
      for (int i = l; i < n; i += LPC) {
        u32x4 r = philox4x32_10((uint32_t)i, cid, sweep0, STREAM_INIT, a.k0, a.k1);
        st[i] = (r.x >> 31) ? (_Float16)1.0f : (_Float16)-1.0f;
      }
    } else {
      const int8_t* src = a.state + (size_t)chain * n;
      for (int i = l; i < n; i += LPC) st[i] = (_Float16)(float)src[i];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();

  if (valid) {
    for (uint32_t t = sweep0; t < sweep0 + (uint32_t)a.n_sweeps; ++t) {
      const uint32_t tq = t >> 2, tw = t & 3u;
      for (int k = 0; k < a.n_colours; ++k) {
        const int lo = L.cls[k], hi = L.cls[k + 1];
        for (int p = lo + l; p < hi; p += LPC) {
          const int i = L.order[p];
          const float f = gibbs_field(L.hs[i], L.row[i], L, st, a.n_batches, a.max_batches);
          const u32x4 r = philox4x32_10((uint32_t)i, cid, tq, STREAM_GIBBS, a.k0, a.k1);
          st[i] = gibbs_decide(f, a.two_beta, pick(r, tw));
        }
        // the next class reads what this one wrote (same wave): order LDS traffic
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
    }
    int8_t* dst = a.state + (size_t)chain * n;
    for (int i = l; i < n; i += LPC) {
      const float v = (float)st[i];
      dst[i] = (int8_t)v;
      if (a.samples_out) a.samples_out[(size_t)chain * n + i] = v;
    }
  }
}

// The rolled schedule with the 16 chains of a workgroup side by side in LDS (round 6), for the large graphs that keep it
// (c5: 1024 spins, one 16-wave workgroup per CU, four waves per SIMD).  Same image, same arithmetic, same order, same random
// stream as gibbs_kernel -- what changes is who does what.  gibbs_kernel gives a wave one chain, its 64 lanes 64 spins of a
// colour class: every neighbour read is a gather of 64 random halfwords of the chain's state row -- ~11 cycles of the CU's
// one LDS pipe per instruction (64 lanes over 64 banks at random: 4-5 deep), 20 per update, 16 waves: the draw is bound by
// bank conflicts, which is why neither a quarter of the Philox calls nor two fewer dependent table reads per update moved
// it (both built and measured this round: 1203 / 1193 against 1207 us), and why the lane-major form on twice the CUs takes
// half the time.  Here a wave's lanes are 4 spins x 16 chains, and the state is stored [spin][chain]: a neighbour read is
// four runs of sixteen consecutive halfwords (8 banks each), the table reads four addresses broadcast to sixteen lanes.
// All 16 waves now work on the same 16 chains, so a colour class ends in a workgroup barrier instead of a wave fence
// (5 per sweep).  A lane's (class, pass) slots are static (the host lists them: GibbsSlots), the slot loop unrolled.
struct GibbsSlots { int n; uint32_t last_mask; int16_t lo[20], hi[20]; };

// f + (the row's signed couplings, in row order) with the state stored [spin][16 chains]: `stc` = the lane's chain column
// (FIX: the image holds exactly GIBBS_NB batches per spin, spin i's at GIBBS_NB i -- rowinfo = that index; no per-batch select)
template <bool FIX>
__device__ __forceinline__ float gibbs_field_cm(float f, uint32_t rowinfo, const GibbsLds& L, const unsigned char* stc,
                                                int zero_batch, int max_batches) {
  const int first = FIX ? (int)rowinfo : (int)(rowinfo >> 8), nb = FIX ? GIBBS_NB : (int)(rowinfo & 255u);
  for (int b0 = 0; b0 < (FIX ? GIBBS_NB : max_batches); b0 += GIBBS_NB) {
    gf32x4 w[GIBBS_NB];
    gu32x2 o[GIBBS_NB];
#pragma unroll
    for (int j = 0; j < GIBBS_NB; ++j) {
      const int bi = (FIX || b0 + j < nb) ? first + b0 + j : zero_batch;
      w[j] = *reinterpret_cast<const gf32x4*>(L.w + 4 * bi);
      o[j] = *reinterpret_cast<const gu32x2*>(L.off + 4 * bi);
    }
    _Float16 h[GIBBS_NB][4];
#pragma unroll
    for (int j = 0; j < GIBBS_NB; ++j) {  // (this kernel stages the offsets as 32 x spin: gibbs_stage<.., 32>)
      h[j][0] = *reinterpret_cast<const _Float16*>(stc + (o[j][0] & 0xffffu));
      h[j][1] = *reinterpret_cast<const _Float16*>(stc + (o[j][0] >> 16));
      h[j][2] = *reinterpret_cast<const _Float16*>(stc + (o[j][1] & 0xffffu));
      h[j][3] = *reinterpret_cast<const _Float16*>(stc + (o[j][1] >> 16));
    }
#pragma unroll
    for (int j = 0; j < GIBBS_NB; ++j) {
      f = gibbs_signed_add(h[j][0], w[j][0], f);
      f = gibbs_signed_add(h[j][1], w[j][1], f);
      f = gibbs_signed_add(h[j][2], w[j][2], f);
      f = gibbs_signed_add(h[j][3], w[j][3], f);
    }
  }
  return f;
}

template <int K, int MAXS, int SB, bool FIX>  // SB: bytes between two spins in the state image (2 x chains per workgroup)
struct GibbsSlotLoop {
  static __device__ __forceinline__ void run(const GibbsArgs& a, const GibbsSlots& sl, const GibbsLds& L, unsigned char* stc,
                                             bool valid, uint32_t cid, uint32_t tq, uint32_t tw, bool fresh,
                                             const uint32_t (&ir)[MAXS], uint32_t (&cy)[MAXS], uint32_t (&cz)[MAXS],
                                             uint32_t (&cw)[MAXS]) {
    if (K >= sl.n) return;
    // (FIX: the descriptor is the spin alone, two slots per register -- the word cache leaves no room for one each)
    const uint32_t desc = FIX ? (ir[K >> 1] >> (16 * (K & 1))) & 0xffffu : ir[K];
    if (valid && desc != (FIX ? 0xffffu : 0xffffffffu)) {
      // (opaque copy: everything derived from the descriptor -- batch indices, table addresses, the state address -- is
      // invariant over the sweeps, and hoisted out of the sweep loop it costs ~12 registers per slot: spills)
      uint32_t irk = desc;
      asm volatile("" : "+v"(irk));
      const int i = FIX ? (int)irk : (int)(irk >> 21);
      const float f = gibbs_field_cm<FIX>(L.hs[i], FIX ? (uint32_t)(GIBBS_NB * i) : (irk & 0x1fffffu), L, stc, a.n_batches, a.max_batches);
      // the four words of a (spin, sweep >> 2) counter serve four sweeps: drawn once, three kept (with the LDS gathers out
      // of the way the kernel is bound by its vector instructions, and a Philox call is 100 of an update's 190)
      uint32_t word;
      if (fresh) {
        const u32x4 r = philox4x32_10((uint32_t)i, cid, tq, STREAM_GIBBS, a.k0, a.k1);
        word = pick(r, tw);
        cy[K] = r.y; cz[K] = r.z; cw[K] = r.w;
      } else {
        word = tw == 1u ? cy[K] : (tw == 2u ? cz[K] : cw[K]);
      }
      *reinterpret_cast<_Float16*>(stc + SB * i) = gibbs_decide(f, a.two_beta, word);
    }
    if ((sl.last_mask >> K) & 1u) __syncthreads();  // the next class reads what this one wrote -- all 16 waves, the same chains
    GibbsSlotLoop<K + 1, MAXS, SB, FIX>::run(a, sl, L, stc, valid, cid, tq, tw, fresh, ir, cy, cz, cw);
  }
};
template <int MAXS, int SB, bool FIX>
struct GibbsSlotLoop<MAXS, MAXS, SB, FIX> {
  static __device__ __forceinline__ void run(const GibbsArgs&, const GibbsSlots&, const GibbsLds&, unsigned char*, bool, uint32_t,
                                             uint32_t, uint32_t, bool, const uint32_t (&)[MAXS], uint32_t (&)[MAXS],
                                             uint32_t (&)[MAXS], uint32_t (&)[MAXS]) {}
};

// LDS of the fixed-row image (FIX): h [n] | couplings [n][20] | state offsets [n][20] | order [n] | state [n_pad][CH]
__host__ __device__ __forceinline__ size_t gibbs_fix_layout(int n, size_t off[5]) {
  size_t b = 0;
  off[0] = b; b += sizeof(float) * (size_t)((n + 3) & ~3);
  off[1] = b; b += sizeof(float) * 20 * (size_t)n;
  off[2] = b; b += sizeof(uint16_t) * 20 * (size_t)n;
  off[3] = b; b += sizeof(uint16_t) * (size_t)((n + 7) & ~7);
  off[4] = b;
  return b;
}

// CH chains per workgroup (16; 8: the draws of few chains -- twice the workgroups, runs of 4 banks).  FIX: graphs of at most
// 20 neighbours per spin whose image fits with every spin's row padded to five batches (c5's do, with 2 KB to spare): the
// batch tables are then addressed by the spin alone -- 5 compares, 5 selects and 10 address computations less per update.
template <int MAXS, int CH, bool FIX>
__global__ __launch_bounds__(1024) void gibbs_slot_kernel(GibbsArgs a, GibbsSlots sl) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int WAVES = 16, SP = 64 / CH, SB = 2 * CH;  // spins per wave and pass; bytes between two spins of the state image
  const int n = a.n;
  const int tid = threadIdx.x;
  GibbsLds L;
  if constexpr (FIX) {
    size_t o[5];
    gibbs_fix_layout(n, o);
    L.hs = reinterpret_cast<float*>(smem + o[0]);
    L.w = reinterpret_cast<float*>(smem + o[1]);
    L.off = reinterpret_cast<uint16_t*>(smem + o[2]);
    L.order = reinterpret_cast<uint16_t*>(smem + o[3]);
    L.state = reinterpret_cast<_Float16*>(smem + o[4]);
    L.row = nullptr; L.cls = nullptr;
    for (int i = tid; i < n; i += WAVES * 64) {
      L.hs[i] = clampf(__fmul_rn(a.prefactor, a.linear[i]), a.h_lo, a.h_hi);
      L.order[i] = (uint16_t)a.order[i];
    }
    for (int q = tid; q < 20 * n; q += WAVES * 64) {
      const int i = q / 20, e = q - 20 * i;
      const uint32_t ri = (uint32_t)a.adj_row[i];
      const int src = (e >> 2) < (int)(ri & 255u) ? a.adj_src4[4 * (int)(ri >> 8) + e] : -1;
      L.w[q] = src >= 0 ? clampf(__fmul_rn(a.prefactor, a.quadratic[a.adj_eid[src]]), a.j_lo, a.j_hi) : 0.0f;
      L.off[q] = src >= 0 ? (uint16_t)(SB * a.adj_idx[src]) : (uint16_t)0;
    }
    __syncthreads();
  } else {
    L = gibbs_carve(smem, n, a.n_batches, a.n_colours);
    gibbs_stage<WAVES * 64, SB>(a, L, tid);
  }

  const int wave = tid >> 6, lane = tid & 63;
  const int g = lane / CH, c = lane % CH;  // spin of the wave's SP, chain of the workgroup's CH
  const int chain = blockIdx.x * CH + c;
  const bool valid = chain < a.n_chains;
  unsigned char* stc = reinterpret_cast<unsigned char*>(L.state) + 2 * c;  // state[spin][chain]: spin i at stc + SB i
  const uint32_t cid = a.chain_id0 + (uint32_t)chain;
  const uint32_t sweep0 = a.sweep0_dev ? *a.sweep0_dev : a.sweep0;
  if (valid) {
    // (64 (wave, spin-of-four) pairs walk the chain's spins)
    if (a.init) {
      for (int i = wave * SP + g; i < n; i += 16 * SP) {
        u32x4 r = philox4x32_10((uint32_t)i, cid, sweep0, STREAM_INIT, a.k0, a.k1);
        *reinterpret_cast<_Float16*>(stc + SB * i) = (r.x >> 31) ? (_Float16)1.0f : (_Float16)-1.0f;
      }
    } else {
      const int8_t* src = a.state + (size_t)chain * n;
      for (int i = wave * SP + g; i < n; i += 16 * SP) *reinterpret_cast<_Float16*>(stc + SB * i) = (_Float16)(float)src[i];
    }
  }
  // this lane's spin of every slot and its row descriptor (spin << 21 | first batch << 8 | batches), read once (11 bits
  // of spin, 13 of first batch: the launcher checks); words 1..3 of the slot's current Philox counter
  uint32_t ir[MAXS], cy[MAXS], cz[MAXS], cw[MAXS];
#pragma unroll
  for (int k = 0; k < MAXS; ++k) {
    const int p = k < sl.n ? sl.lo[k] + wave * SP + g : 0;
    const bool has = k < sl.n && p < sl.hi[k];
    const int i = has ? L.order[p] : 0;
    if (FIX) { if (!(k & 1)) ir[k >> 1] = 0u; ir[k >> 1] |= (has ? (uint32_t)i : 0xffffu) << (16 * (k & 1)); }
    else ir[k] = has ? ((uint32_t)i << 21) | L.row[i] : 0xffffffffu;
    cy[k] = 0u; cz[k] = 0u; cw[k] = 0u;
  }
  __syncthreads();  // every chain's start state is in place
  for (uint32_t t = sweep0; t < sweep0 + (uint32_t)a.n_sweeps; ++t) {
    // (static slot indices: the slot registers must never be indexed at run time.  A plain unrolled loop with the `k >= n`
    // exit was left rolled by the compiler, the arrays in scratch; a generic lambda per slot lost the LDS address space of
    // the image -- flat loads -- and put the batch arrays in scratch: a recursive template it is.)
    GibbsSlotLoop<0, MAXS, SB, FIX>::run(a, sl, L, stc, valid, cid, t >> 2, t & 3u, t == sweep0 || (t & 3u) == 0u, ir, cy, cz, cw);
  }
  if (valid) {
    int8_t* dst = a.state + (size_t)chain * n;
    for (int i = wave * SP + g; i < n; i += 16 * SP) {
      const float v = (float)*reinterpret_cast<const _Float16*>(stc + SB * i);
      dst[i] = (int8_t)v;
      if (a.samples_out) a.samples_out[(size_t)chain * n + i] = v;
    }
  }
}

// Fast path for graphs of at most 20 rows per lane -- a row being one pass of LPC spins of a colour class -- and
// at most 4 MB <= 20 neighbours per spin: every graph the shipped solvers produce up to 1024 spins.  Same arithmetic,
// same order, same random stream as gibbs_kernel (bit-exact); what changes is the schedule and the LDS image:
//   * a lane owns the same spin of every row in every sweep, so its spin index and clamped field offset are read once,
//     before the sweep loop, and live in registers;
//   * the neighbour tables are stored LANE-MAJOR (graph.h, `lane_eid` / `lane_off`): batch j of row k of lane l sits at
//     ((k MB + j) LPC + l), so a batch read is one ds_read_b128 / ds_read_b64 at an IMMEDIATE offset from a per-lane base
//     -- consecutive lanes read consecutive 16-byte words (no bank conflicts, where the per-row image scattered them) and
//     the ~25 compare / select / address instructions per row of the per-row walk are gone; a lane without a spin in a
//     row, or a spin of fewer than MB batches, finds J = 0 there;
//   * the passes of a colour class are independent of each other and are computed two at a time (NR = 2: classes of more
//     than LPC spins -- c3, c5), side by side: at one wave per SIMD a single row leaves the wave waiting out its LDS
//     round trips and a chain of 20 dependent adds with nothing else to issue;
//   * the state offsets of a step are read one step AHEAD, the couplings first, the stores are unconditional (a sink
//     for lanes without a spin): a step is one straight line of code with one LDS round trip waited out;
//   * Philox words are drawn once per (spin, sweep >> 2) and serve four sweeps, as the counter layout intends.
// (Rounds 3-5 history, c3 draw alone: the per-row image with register-resident row descriptors 1.00 ms; lane-major
// 0.73; two passes side by side 0.68; offsets one step ahead + straight-line steps 0.54.  Forms that lost: two waves
// per chain with 4-wave workgroups (faster alone, twice the workgroups beside the encoder: slower step); a wave PAIR per
// chain in 8-wave workgroups meeting at a barrier per class (0.61 ms); 8 or 2 waves per workgroup (step +0.2 ms).)
__host__ __device__ __forceinline__ size_t gibbs_lane_lds_bytes(int slots, int mb, int lpc, int n, int chains) {
  // tables | one state row per chain | a 64-entry sink per wave (where lanes without a spin in a slot store)
  return (size_t)slots * mb * lpc * 24 + sizeof(_Float16) * (size_t)chains * ((n + 15) & ~15) +
         sizeof(_Float16) * 64 * (size_t)((chains * lpc + 63) / 64);
}

// One colour-class step of the fast schedule: NRUN rows (slots k, k + 1) side by side -- per row the same reads and
// the same adds, in row order.  The rows' state offsets `o` were read one step AHEAD (they do not depend on the state);
// this step issues its state reads and coupling reads, then reads the NEXT step's offsets (`ol_next`) while those are in
// flight, so that of the two dependent LDS round trips per step (tables, then state) only the second one is waited out.
template <int MB, int LPC, int NR, int NRUN>
__device__ __forceinline__ void gibbs_lane_class(float (&f)[NR], gu32x2 (&o)[NR][MB], const unsigned char* wlk,
                                                 const unsigned char* ol_next, const unsigned char* stb) {
  _Float16 h[NRUN][MB][4];
  gf32x4 w[NRUN][MB];
  // (the couplings first: nothing they wait for, and the LDS serves them while the state addresses are computed)
#pragma unroll
  for (int j = 0; j < MB; ++j) {
#pragma unroll
    for (int r = 0; r < NRUN; ++r) w[r][j] = *reinterpret_cast<const gf32x4*>(wlk + (size_t)(r * MB + j) * LPC * 16);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < MB; ++j) {
#pragma unroll
    for (int r = 0; r < NRUN; ++r) {
      h[r][j][0] = *reinterpret_cast<const _Float16*>(stb + (o[r][j][0] & 0xffffu));
      h[r][j][1] = *reinterpret_cast<const _Float16*>(stb + (o[r][j][0] >> 16));
      h[r][j][2] = *reinterpret_cast<const _Float16*>(stb + (o[r][j][1] & 0xffffu));
      h[r][j][3] = *reinterpret_cast<const _Float16*>(stb + (o[r][j][1] >> 16));
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
#pragma unroll
    for (int j = 0; j < MB; ++j) o[r][j] = *reinterpret_cast<const gu32x2*>(ol_next + (size_t)(r * MB + j) * LPC * 8);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < MB; ++j) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int r = 0; r < NRUN; ++r) f[r] = gibbs_signed_add(h[r][j][e], w[r][j][e], f[r]);
    }
  }
}

template <int LPC, int WAVES, int MB, int NR, int MAXS>
__global__ __launch_bounds__(WAVES * 64) void gibbs_fast_kernel(GibbsArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int n = a.n, n_pad = (n + 15) & ~15;
  const int n_rows = a.n_rows;
  const int tid = threadIdx.x;
  // LDS: couplings [n_rows MB][LPC] x 16 B | state offsets [n_rows MB][LPC] x 8 B | state [chains][n_pad] float16 | sinks
  gf32x4* wimg = reinterpret_cast<gf32x4*>(smem);
  gu32x2* oimg = reinterpret_cast<gu32x2*>(smem + (size_t)n_rows * MB * LPC * 16);
  _Float16* state = reinterpret_cast<_Float16*>(smem + (size_t)n_rows * MB * LPC * 24);
  {
    // one (row, batch, lane) cell per thread and trip: four edge ids and four offsets in two coalesced loads, four
    // gathered couplings; two cells in flight
    const int n_cell = n_rows * MB * LPC;
    const int4* eid4 = reinterpret_cast<const int4*>(a.lane_eid);
    const gu32x2* off4 = reinterpret_cast<const gu32x2*>(a.lane_off);
    auto coupling = [&](int e) {
      return e >= 0 ? clampf(__fmul_rn(a.prefactor, a.quadratic[e]), a.j_lo, a.j_hi) : 0.0f;
    };
    for (int q = tid; q < n_cell; q += 2 * WAVES * 64) {
      const int q1 = q + WAVES * 64;
      const bool two = q1 < n_cell;
      const int4 e0 = eid4[q], e1 = two ? eid4[q1] : int4{-1, -1, -1, -1};
      const gu32x2 o0 = off4[q], o1 = two ? off4[q1] : gu32x2{0u, 0u};
      const gf32x4 w0 = {coupling(e0.x), coupling(e0.y), coupling(e0.z), coupling(e0.w)};
      const gf32x4 w1 = {coupling(e1.x), coupling(e1.y), coupling(e1.z), coupling(e1.w)};
      wimg[q] = w0; oimg[q] = o0;
      if (two) { wimg[q1] = w1; oimg[q1] = o1; }
    }
  }

  constexpr int CPW = 64 / LPC;
  const int wave = tid >> 6, lane = tid & 63;
  const int sub = lane / LPC, l = lane % LPC;
  const int chain = (blockIdx.x * WAVES + wave) * CPW + sub;
  const bool valid = chain < a.n_chains;
  _Float16* st = state + (size_t)(wave * CPW + sub) * n_pad;
  const uint32_t cid = a.chain_id0 + (uint32_t)chain;

  // (read before the chains start: a fresh chain's start configuration is keyed by the sweep index it starts at, so
  // non-persistent draws do not all restart from one configuration)
  const uint32_t sweep0 = a.sweep0_dev ? *a.sweep0_dev : a.sweep0;
  if (valid) {
    if (a.init) {
      for (int i = l; i < n; i += LPC) {
        u32x4 r = philox4x32_10((uint32_t)i, cid, sweep0, STREAM_INIT, a.k0, a.k1);
        st[i] = (r.x >> 31) ? (_Float16)1.0f : (_Float16)-1.0f;
      }
    } else {
      const int8_t* src = a.state + (size_t)chain * n;
      for (int i = l; i < n; i += LPC) st[i] = (_Float16)(float)src[i];
    }
  }
  // this lane's spin in each row (-1: none) and its clamped field offset
  int sp[MAXS];
  float hs[MAXS];
#pragma unroll
  for (int k = 0; k < MAXS; ++k) {
    sp[k] = (valid && k < n_rows) ? a.lane_spin[k * LPC + l] : -1;
    hs[k] = sp[k] >= 0 ? clampf(__fmul_rn(a.prefactor, a.linear[sp[k]]), a.h_lo, a.h_hi) : 0.f;
  }
  __syncthreads();  // the tables are staged
  if (!valid) return;

  // the byte a lane stores its slot-k decision to: its spin's, or its own entry of the wave's sink when it has none
  // (an unconditional store keeps a class step one straight line of code)
  _Float16* sink = state + (size_t)WAVES * CPW * n_pad + (size_t)wave * 64 + lane;
  const unsigned char* wl = smem + (size_t)l * 16;
  const unsigned char* ol = smem + (size_t)n_rows * MB * LPC * 16 + (size_t)l * 8;
  const unsigned char* stb = reinterpret_cast<const unsigned char*>(st);
  u32x4 rr[MAXS];
#pragma unroll
  for (int k = 0; k < MAXS; ++k) rr[k] = u32x4{0u, 0u, 0u, 0u};
  gu32x2 o[NR][MB];                  // the coming step's state offsets
#pragma unroll
  for (int r = 0; r < NR; ++r) {
#pragma unroll
    for (int j = 0; j < MB; ++j) o[r][j] = *reinterpret_cast<const gu32x2*>(ol + (size_t)(r * MB + j) * LPC * 8);
  }
  for (uint32_t t = sweep0; t < sweep0 + (uint32_t)a.n_sweeps; ++t) {
    const uint32_t tq = t >> 2, tw = t & 3u;
    if (t == sweep0 || tw == 0u) {
#pragma unroll
      for (int k = 0; k < MAXS; ++k)
        if (sp[k] >= 0) rr[k] = philox4x32_10((uint32_t)sp[k], cid, tq, STREAM_GIBBS, a.k0, a.k1);
    }
    // word tw of a slot's four, as two bit-selects under wave-uniform masks (a uniform `?:` becomes scalar branches,
    // four per slot and sweep)
    const uint32_t m1 = 0u - (tw & 1u), m2 = 0u - (tw >> 1);
#define DVG_GIBBS_WORD(K) (((((rr[K].x & ~m1) | (rr[K].y & m1)) & ~m2)) | (((rr[K].z & ~m1) | (rr[K].w & m1)) & m2))
#pragma unroll
    for (int c = 0; c < MAXS / NR; ++c) {
      const int k0 = NR * c, k1 = k0 + NR - 1;
      if (k0 >= n_rows) break;
      const unsigned char* wlk = wl + (size_t)k0 * MB * LPC * 16;
      const unsigned char* ol_next = ol + (size_t)(k0 + NR < n_rows ? k0 + NR : 0) * MB * LPC * 8;
      float f[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) f[r] = hs[k0 + r];
      bool both = false;
      if constexpr (NR == 2) both = __builtin_amdgcn_ballot_w64(sp[k1] >= 0) != 0;
      if (both) {
        gibbs_lane_class<MB, LPC, NR, NR>(f, o, wlk, ol_next, stb);
        const _Float16 s0 = gibbs_decide(f[0], a.two_beta, DVG_GIBBS_WORD(k0));
        const _Float16 s1 = gibbs_decide(f[NR - 1], a.two_beta, DVG_GIBBS_WORD(k1));
        *(sp[k0] >= 0 ? st + sp[k0] : sink) = s0;
        *(sp[k1] >= 0 ? st + sp[k1] : sink) = s1;
      } else {  // one row (NR = 2: the step's second row is empty for the whole wave)
        gibbs_lane_class<MB, LPC, NR, 1>(f, o, wlk, ol_next, stb);
        const _Float16 s0 = gibbs_decide(f[0], a.two_beta, DVG_GIBBS_WORD(k0));
        *(sp[k0] >= 0 ? st + sp[k0] : sink) = s0;
      }
      // the next class reads what this one wrote (same wave): order LDS traffic.  (Passes of one class are independent
      // of each other, so a fence between them is harmless.)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
#undef DVG_GIBBS_WORD
  int8_t* dst = a.state + (size_t)chain * n;
  for (int i = l; i < n; i += LPC) {
    const float v = (float)st[i];
    dst[i] = (int8_t)v;
    if (a.samples_out) a.samples_out[(size_t)chain * n + i] = v;
  }
}

static size_t gibbs_lds_bytes(int n, int n_batches, int n_colours, int chains_per_block) {
  size_t o[7];
  return gibbs_lds_layout(n, n_batches, n_colours, o) + sizeof(_Float16) * (size_t)chains_per_block * ((n + 15) & ~15);
}

// dvg_gibbs_launch_info: the dispatch below runs with a probe set and reports its launch geometry instead of launching
struct GibbsProbe { int workgroups; int threads; size_t lds; };
static thread_local GibbsProbe* g_gibbs_probe = nullptr;

template <int LPC, int WAVES>
static int launch_gibbs(GibbsArgs a, hipStream_t s) {
  constexpr int CPB = WAVES * (64 / LPC);
  const size_t lds = gibbs_lds_bytes(a.n, a.n_batches, a.n_colours, CPB);
  if (lds > 160 * 1024) {
    set_error("gibbs: graph (n=%d, %d neighbour batches) needs %zu B of LDS > 160 KiB", a.n, a.n_batches, lds);
    return DVG_E_UNSUPPORTED;
  }
  const int grid = (int)ceil_div(a.n_chains, CPB);
  if (g_gibbs_probe) { *g_gibbs_probe = GibbsProbe{grid, WAVES * 64, lds}; return DVG_OK; }
  auto kern = gibbs_kernel<LPC, WAVES>;
  if (lds > 64 * 1024)
    DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // work = spin updates of the draw (bench.py's sampler roofline)
  DVG_LAUNCH_WORK(K_GIBBS, (double)a.n_chains * a.n * a.n_sweeps, kern, dim3(grid), dim3(WAVES * 64), lds, s, a);
  return DVG_OK;
}

// pass = 16 waves x (64 / ch) spins; false: more than 20 (class, pass) slots, or a graph beyond the slot registers' bit fields
static bool gibbs_make_slots(const dvg_graph_t* g, int ch, GibbsSlots* sl) {
  *sl = GibbsSlots{};
  if (g->n_colours > 64 || g->n >= 2048 || g->n_batches >= 8192) return false;
  const int pass = 16 * (64 / ch);
  for (int k = 0; k < g->n_colours; ++k) {
    const int lo = g->h_class_ptr[k], hi = g->h_class_ptr[k + 1];
    for (int p = lo; p < hi; p += pass) {
      if (sl->n >= 20) return false;
      sl->lo[sl->n] = (int16_t)p; sl->hi[sl->n] = (int16_t)hi;
      if (p + pass >= hi) sl->last_mask |= 1u << sl->n;
      ++sl->n;
    }
  }
  return sl->n > 0;
}

template <int CH>
static int launch_gibbs_slots(GibbsArgs a, const GibbsSlots& sl, hipStream_t s) {
  size_t fo[5];
  const size_t lds_fix = gibbs_fix_layout(a.n, fo) + sizeof(_Float16) * (size_t)CH * ((a.n + 15) & ~15);
  const bool fix = a.max_batches <= GIBBS_NB && lds_fix <= 160 * 1024;
  const size_t lds = fix ? lds_fix : gibbs_lds_bytes(a.n, a.n_batches, a.n_colours, CH);
  if (lds > 160 * 1024) {
    set_error("gibbs: graph (n=%d, %d neighbour batches) needs %zu B of LDS > 160 KiB", a.n, a.n_batches, lds);
    return DVG_E_UNSUPPORTED;
  }
  const int grid = (int)ceil_div(a.n_chains, CH);
  if (g_gibbs_probe) { *g_gibbs_probe = GibbsProbe{grid, 1024, lds}; return DVG_OK; }
  // (18 slots: c5's graphs; the word registers of two more slots are what spills at 128 registers per wave)
  auto kern = fix ? (sl.n <= 18 ? gibbs_slot_kernel<18, CH, true> : gibbs_slot_kernel<20, CH, true>)
                  : (sl.n <= 18 ? gibbs_slot_kernel<18, CH, false> : gibbs_slot_kernel<20, CH, false>);
  if (lds > 64 * 1024)
    DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  DVG_LAUNCH_WORK(K_GIBBS, (double)a.n_chains * a.n * a.n_sweeps, kern, dim3(grid), dim3(1024), lds, s, a, sl);
  return DVG_OK;
}

template <int LPC, int WAVES, int MB, int NR, int MAXS>
static int launch_gibbs_fast(GibbsArgs a, hipStream_t s) {
  constexpr int CPB = WAVES * (64 / LPC);
  const size_t lds = gibbs_lane_lds_bytes(a.n_rows, MB, LPC, a.n, CPB);
  if (lds > 160 * 1024) {
    set_error("gibbs: graph (n=%d, %d rows per lane) needs %zu B of LDS > 160 KiB", a.n, a.n_rows, lds);
    return DVG_E_UNSUPPORTED;
  }
  const int grid = (int)ceil_div(a.n_chains, CPB);
  if (g_gibbs_probe) { *g_gibbs_probe = GibbsProbe{grid, WAVES * 64, lds}; return DVG_OK; }
  auto kern = gibbs_fast_kernel<LPC, WAVES, MB, NR, MAXS>;
  if (lds > 64 * 1024)
    DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
  DVG_REQUIRE(g->n <= 32767, "gibbs: graph too large (n=%d: 16-bit state offsets)", g->n);
  GibbsArgs a;
  a.order = g->order; a.class_ptr = g->class_ptr;
  a.adj_idx = g->adj_idx; a.adj_eid = g->adj_eid; a.adj_row = g->adj_row; a.adj_src4 = g->adj_src4;
  a.linear = linear; a.quadratic = quadratic;
  a.n = g->n; a.n_batches = g->n_batches; a.max_batches = g->max_batches; a.n_colours = g->n_colours;
  a.prefactor = prefactor; a.h_lo = h_lo; a.h_hi = h_hi; a.j_lo = j_lo; a.j_hi = j_hi;
  a.two_beta = 2.0f * beta;
  a.state = state; a.samples_out = samples_out; a.n_chains = n_chains;
  a.chain_id0 = chain_id0; a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32);
  a.sweep0 = sweep0; a.n_sweeps = n_sweeps; a.init = init;
  a.sweep0_dev = dyn ? &dyn->sweep0 : nullptr;
  a.lane_spin = nullptr; a.lane_eid = nullptr; a.lane_off = nullptr; a.n_rows = 0;
  return gibbs_dispatch(g, a, n_chains, (hipStream_t)stream);
}

// The launch geometry dvg_gibbs_sample would use for `n_chains` chains on this graph (nothing is launched): workgroups,
// threads per workgroup, LDS bytes per workgroup.  The host side sizes what runs BESIDE the draw with it (the encoder
// forward's Winograd launches take whole CUs: ModelWrapper leaves the draw's CUs out of their persistent grid).
extern "C" int dvg_gibbs_launch_info(const dvg_graph_t* g, int n_chains, int* workgroups, int* threads, size_t* lds_bytes) {
  DVG_REQUIRE(g && n_chains > 0, "gibbs_launch_info: null graph / no chains");
  GibbsArgs a{};
  a.n = g->n; a.n_batches = g->n_batches; a.max_batches = g->max_batches; a.n_colours = g->n_colours; a.n_chains = n_chains;
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
  // (option gibbs_generic = 1 forces the rolled reference schedule: A/B runs and the bit-exactness test of the fast one;
  // 2 = the fast schedule without its two-passes-side-by-side form)
  const int64_t form = opt(OPT_GIBBS_GENERIC);
  // Graphs of more than 12 rows per lane (1024 spins: ~140 KB of tables, one 4-chain workgroup per CU) take the fast
  // schedule only for draws of at most 512 chains: it is the LATENCY form (c5-size graph, 2048 chains x 50 sweeps: 0.59 ms
  // alone against the rolled 16-wave form's 1.20) and costs the same CU-time -- but it takes every CU's LDS while it
  // runs, and the c5 step, whose encoder AND decoder forward hide behind a 1.2 ms draw on half the chip, got slower
  // with it (2.88 -> 3.01 ms: `profiles/r05_gibbs_c5_forms.txt`).
  if (form == 4) {  // (4: the chains-side-by-side schedule in 8-chain workgroups whatever the size: A/B runs)
    GibbsSlots sl8;
    if (gibbs_make_slots(g, 8, &sl8)) return launch_gibbs_slots<8>(a, sl8, s);
  }
  bool lane_ok = g->lane_eid && (g->lane_rows <= 12 || n_chains <= 512 || form == 3);  // (3: the fast schedule whatever the size: A/B runs)
  if (lane_ok) {
    // (a lane image the fast form cannot serve -- more than 160 KiB of LDS: 20 rows x 5 batches at n_pad > 1216;
    // or a shape without an instantiation: two rows per step with a single small class -- falls through to the rolled
    // schedule, which serves every graph, instead of failing the draw)
    const int lpc = g->lane_lpc, mb = g->lane_mb, rows = g->lane_rows, nr = form == 2 ? 1 : g->lane_nr;
    const bool small = gibbs_lane_lds_bytes(rows, mb, lpc, g->n, 0) <= 16 * 1024 && n_chains <= 1024 && rows <= 12;
    const int waves = small ? 1 : 4;
    const bool has_kernel = (lpc == 16 || lpc == 32 || lpc == 64) && (mb == 4 || mb == 5) && rows <= 20 &&
                            (nr == 1 ? (rows <= 12 || (lpc == 64 && !small)) : (nr == 2 && lpc == 64 && !small));
    if (!has_kernel || gibbs_lane_lds_bytes(rows, mb, lpc, g->n, waves * (64 / lpc)) > 160 * 1024) lane_ok = false;
  }
  if (lane_ok && form != 1) {
    // Lanes per chain: the smallest of 16/32/64 that covers the largest colour class in one pass, else 64 (graph.cpp).
    // Waves per workgroup.  Measured on the c2 step with the draw overlapped with the encoder forward: 4 -> 1.237 ms,
    // 8 -> 1.273, 16 -> 1.391: the sweep loop does contend for issue slots, fatter workgroups do not pay for the CUs they
    // free (round 5, c3: 8 or 2 waves instead of 4: step +0.2 ms).  Small graphs with few chains (c2: 128 spins, 256
    // chains -> 32 four-wave workgroups on 256 CUs): one wave per workgroup spreads the draw over four times as many
    // CUs at a few KB of tables each (c2 step 1.082 -> 1.060 ms).  Larger graphs keep four waves: every extra workgroup
    // stages its own copy of the tables.
    const int lpc = g->lane_lpc, mb = g->lane_mb, rows = g->lane_rows;
    a.lane_spin = g->lane_spin; a.lane_eid = g->lane_eid; a.lane_off = g->lane_off; a.n_rows = rows;
    const int nr = form == 2 ? 1 : g->lane_nr;  // (2: a two-row image walked one row at a time -- empty rows and all)
    const bool small = gibbs_lane_lds_bytes(rows, mb, lpc, g->n, 0) <= 16 * 1024 && n_chains <= 1024 && rows <= 12;
#define DVG_GIBBS_FAST(LPC, W, MB, NR, MAXS) \
  if (lpc == LPC && mb == MB && nr == NR && rows <= MAXS && (W == 1) == small) return launch_gibbs_fast<LPC, W, MB, NR, MAXS>(a, s);
    DVG_GIBBS_FAST(16, 1, 4, 1, 12) DVG_GIBBS_FAST(16, 4, 4, 1, 12) DVG_GIBBS_FAST(16, 1, 5, 1, 12) DVG_GIBBS_FAST(16, 4, 5, 1, 12)
    DVG_GIBBS_FAST(32, 1, 4, 1, 12) DVG_GIBBS_FAST(32, 4, 4, 1, 12) DVG_GIBBS_FAST(32, 1, 5, 1, 12) DVG_GIBBS_FAST(32, 4, 5, 1, 12)
    DVG_GIBBS_FAST(64, 1, 4, 1, 12) DVG_GIBBS_FAST(64, 4, 4, 1, 12) DVG_GIBBS_FAST(64, 1, 5, 1, 12) DVG_GIBBS_FAST(64, 4, 5, 1, 12)
    DVG_GIBBS_FAST(64, 4, 4, 1, 20) DVG_GIBBS_FAST(64, 4, 5, 1, 20)
    DVG_GIBBS_FAST(64, 4, 4, 2, 12) DVG_GIBBS_FAST(64, 4, 5, 2, 12) DVG_GIBBS_FAST(64, 4, 4, 2, 20) DVG_GIBBS_FAST(64, 4, 5, 2, 20)
#undef DVG_GIBBS_FAST
    set_error("gibbs: lane image (%d lanes per chain, %d batches, %d rows in steps of %d) has no kernel", lpc, mb, rows, nr);
    return DVG_E_UNSUPPORTED;
  }
  // The rolled schedule: graphs without a lane image (more than 20 rows per lane, more than 20 neighbours per spin).
  const int mc = g->max_class;
  // Large graphs (c5: 1024 spins, 2|E| = 16 K -> ~105 KB of tables): one workgroup per CU fits, so the workgroup must
  // carry the CU's whole latency-hiding: 16 waves = 16 chains share one LDS copy of the graph (2 waves left 7/8 of
  // the issue slots empty: 7.3 ms per 2048-chain, 50-sweep draw).
  if (gibbs_lds_bytes(g->n, g->n_batches, g->n_colours, 4) > 72 * 1024) {
    // ... with static (class, pass) slots when the classes make at most 20 of them (gibbs_slot_kernel; option
    // gibbs_generic = 1 keeps the plain rolled kernel: the reference of the bit-exactness tests)
    GibbsSlots sl;
    if (form != 1 && gibbs_make_slots(g, 16, &sl)) return launch_gibbs_slots<16>(a, sl, s);
    return launch_gibbs<64, 16>(a, s);
  }
  const bool small = gibbs_lds_bytes(g->n, g->n_batches, g->n_colours, 0) <= 16 * 1024 && n_chains <= 1024;
#define DVG_GIBBS_DISPATCH(LPC) return small ? launch_gibbs<LPC, 1>(a, s) : launch_gibbs<LPC, 4>(a, s);
  if (mc <= 16) { DVG_GIBBS_DISPATCH(16) }
  if (mc <= 32) { DVG_GIBBS_DISPATCH(32) }
  DVG_GIBBS_DISPATCH(64)
#undef DVG_GIBBS_DISPATCH
}
