// Implicit-GEMM 3x3 convolution on the f32 MFMA (v_mfma_f32_32x32x2_f32: exact fmaf chain),
// used for every GEMM-shaped layer of the encoder / decoder, forward, data-gradient and
// weight-gradient (/root/reference/src/encoder.py:28-30, /root/reference/src/decoder.py:28, :34-38).
//
// Forward / dgrad kernel: block tile BM pixels x BN output channels, K = 9*Cin walked in
// 32-channel chunks per tap; A (im2col rows, gathered with the Morton neighbour map, optional
// on-the-fly nearest upsample) and B (packed weights) staged through LDS, register prefetch of
// the next chunk under the MFMA loop.  Epilogues: +bias, per-channel (sum, sum^2) partials for
// BatchNorm, or the 2x2 quad-sum that is the adjoint of Upsample(x2).
// (weight-gradient kernels, slab reduction and weight packing: conv_wgrad.hip)
#include <cstdlib>

#include "conv_tile.h"

namespace dvg {

// ------------------------------------------------------------------------------------------
// WK = wave groups along K inside the block (each works on its own 64-deep half of a 128-deep K slab, partial
// accumulators combined through LDS at the end).  Only WK = 1 is launched: the idea was to give small launches the
// latency hiding that co-resident blocks provide (standalone 128->128 3x3 layer: 70 TFLOP/s at 1 block/CU, 83 at 2, 94
// at 4, 100 from 8 up), but groups that share the block's barriers stay IN PHASE -- all waves issue loads together and
// run MFMAs together -- and WK = 2 measured 0 % on that layer and -3.5 % on the c2 step.  (The other route, pairs of
// half-K blocks combined in-kernel through an arrival counter, is correct and deterministic but 2x slower: its
// agent-scope release fence writes back the XCD's L2.)
// PM = 0: float32 operands staged through registers (the first form of this kernel; kept as the A/B reference the LDS-DMA
// form is tested bit-identical against: option igemm_dma = 0).
// PM = 3: the float32 form with LDS-DMA staging (global_load_lds_dwordx4: no staging registers, no ds_write pass, no
// mask multiply).  One 32-channel chunk per iteration, two LDS stages, ONE barrier per iteration; both operands sit in
// LDS as [row][32 floats] (weights packed K-major, PackJob.bf16t = 3), 16-byte slots XOR-swizzled on the per-lane SOURCE
// address, and the MFMA's two k-lanes take k = s and k = 16 + s at step s, so a lane's 16 operands of a chunk are
// four ds_read_b128 instead of sixteen ds_read_b32.  The transfers are raw buffer loads: padding rows carry an
// out-of-range offset, for which the hardware writes zeros.
// PM = 4, 5: the SAME staging, LDS image, tiles, position-major / folded / composed forms and epilogues as PM = 3 -- the
// operands arrive in LDS as float32 -- with the arithmetic on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, f32 accumulators),
// the conversion happening at OPERAND-READ time, in registers, between the ds_read_b128 and the MFMA:
//   PM = 4 ("f32x3"): every float32 operand is split into three bf16 pieces by truncation (hi + mid + lo == x bit for
//     bit: 3 x 8 significand bits) and each 16-deep k-step issues the six piece products down to 2^-16 of a product,
//     smallest first, into ONE accumulator (the three dropped ones are below 2^-23): float32-class results at 6/16 of
//     the f32 MFMA's matrix time.  ~5 VALU instructions per operand element, issued under the MFMAs.
//   PM = 5 ("bf16 inputs"): every operand is rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32): the product of
//     the bf16-rounded operands, accumulated in float32.
// A lane's 16 floats of a chunk (k = 16 hh .. 16 hh + 15) feed two k-steps of 8; A and B are converted by the same code
// in the same element order, so the pairing of k indices is the f32 form's.  (Rounds 1-2 staged these two modes through
// registers, converting on the way INTO LDS; they could not use the LDS-DMA path, the position-major tiles or the composed
// first decoder layers, and by the end of round 2 the f32x3 mode was slower than strict float32.)

template <int BM, int BN, int WM, int WN, int WK, int PM = 0>
__global__ __launch_bounds__(WM* WN* WK * 64) void conv_igemm_kernel(ConvArgs a) {
  static_assert(PM == 0 || PM == 3 || PM == 4 || PM == 5, "operand forms: 0 register-staged f32, 3 / 4 / 5 LDS-DMA");
  constexpr bool DMA = PM >= 3;
  static_assert(!DMA || WK == 1, "LDS-DMA form: no K wave groups");
  // (A dedicated loader wave per block -- every buffer_load ... lds of the block issued by a fifth wave -- was built and
  // measured in round 3: 1.3x SLOWER for float32 and 2.3x for the bf16 forms.  One wave's LDS-DMA issue rate, ~25 GB/s,
  // is a fraction of what four waves issuing their own pieces reach, and with two stages its issue and its wait for
  // landing serialise.  The pieces stay with the compute waves; what pays is WHERE in the instruction stream they sit.)
  constexpr int NTC = WM * WN * WK * 64, NT = NTC, NTG = WM * WN * 64;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int BK = DMA ? 32 : 64;  // K per wave group and iteration: 32-channel chunks (possibly of different taps)
  constexpr int KH = (BK / 32) * WK, BKT = BK * WK;  // chunks / K extent staged per iteration by the whole block
  // A rows are 16-byte aligned so the staging stores are ds_write_b128; the MFMA A-operand reads (one float per lane,
  // row stride AP) then see a 2-way bank conflict, which costs less than the 4x ds_write_b32 a 65-float pitch needs
  constexpr int AP = BKT + 4, BP = BN + 4;
  constexpr int RA = BM * 8 / NTC;       // float4 loads of A per thread per 32-chunk (register-staged form)
  constexpr int NB16 = 8 * BN;                // 16-byte loads of B per 32-chunk
  constexpr int RB = (NB16 + NTC - 1) / NTC;  // ... per thread
  constexpr bool BPART = RB * NTC != NB16;    // fewer B loads than threads (32-column tile, 512 threads)
  static_assert(RA * NTC == 8 * BM && (!BPART || RB == 1), "tile loaders must divide evenly");
  // taps per row in the neighbour table: 16 for the folded data gradient, else 9 (a 128-row table is 4.5 KB instead of
  // 8: with it three blocks of the 128x64 LDS-DMA form fit a CU's 160 KB)
  const int NBS = a.ntaps > 9 ? 16 : 9;
  extern __shared__ __align__(16) unsigned char igemm_smem[];  // conv_igemm_lds_bytes<...>() bytes
  float* As = reinterpret_cast<float*>(igemm_smem);            // [BM][AP]
  float* Bs = As + BM * AP;                                    // [BKT][BP]
  constexpr int STAGE = (BM + BN) * 128;  // LDS-DMA form: one stage = A [BM][32 f32] then B [BN][32 f32]
  // Stages of the LDS ring.  Three (two chunks in flight per block) were measured for the bf16-MFMA forms, whose chunk of
  // matrix work is shorter than the LDS-DMA round trip: SLOWER (128 -> 128 layer: f32x3 938 -> 1173 us, bf16 inputs
  // 405 -> 616 us) because the third stage costs the second resident block per CU; co-resident blocks hide that round
  // trip better than a deeper ring does.
  constexpr int NS = 2;
#ifndef IGEMM_LATE_ISSUE
#define IGEMM_LATE_ISSUE 1
#endif
  // [WM][BN][2] BatchNorm partials of the epilogue (LDS-DMA form: on top of the then-dead stage 0, behind a barrier)
  float* red = DMA ? reinterpret_cast<float*>(igemm_smem)
               : Bs + BKT * BP;
  // [BM][NBS] source row of every (tile row, tap), -1 = padding
  int* nbr = DMA ? reinterpret_cast<int*>(igemm_smem + NS * STAGE) : reinterpret_cast<int*>(red + WM * BN * 2);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = wave / (WM * WN), wv = wave - kg * (WM * WN);  // K group, wave within the (WM x WN) tile grid
  const int wm = wv / WN, wn = wv % WN, hh = lane >> 5, c = lane & 31;
  const bool active = kg == 0;           // after the cross-group reduction only group 0 stores
  // fold = 1: grid.x = 4 output parity classes x row blocks of source pixels
  const int mblocks = (int)((a.M + BM - 1) / BM);
  const int cls = a.fold == 1 ? (int)blockIdx.x / mblocks : 0;
  const int64_t m0 = (int64_t)(a.fold == 1 ? (int)blockIdx.x - cls * mblocks : (int)blockIdx.x) * BM;
  const int n0 = blockIdx.y * BN;
  const int L = a.L, H = 1 << L, logHW = 2 * L;
  const int64_t HWin = a.ups ? ((int64_t)1 << logHW) >> 2 : ((int64_t)1 << logHW);
  // position-major tiles (ConvArgs.posmajor): block = (pixel position `pos`, BM consecutive images from pm_img0)
  const bool PMJ = DMA && a.posmajor;
  // The blocks of one image group (all positions of the same BM images) read the same input rows: the hardware deals
  // consecutive workgroups round-robin to the 8 XCDs, so logical block (b % 8) * (n / 8) + b / 8 puts a group's blocks on
  // ONE XCD, next to each other in time, and the group's rows are fetched into that XCD's L2 once instead of 8 times.
  // (fold = 1: the grid spans 4 output classes x row blocks; the mapping works inside a class)
  const uint32_t pm_nb = a.fold == 1 ? (uint32_t)mblocks : gridDim.x;
  const uint32_t pm_bx = a.fold == 1 ? blockIdx.x - (uint32_t)cls * (uint32_t)mblocks : blockIdx.x;
  const uint32_t pm_b = (PMJ && pm_nb % 8 == 0) ? (pm_bx % 8) * (pm_nb / 8) + pm_bx / 8 : pm_bx;
  const int pm_pos = PMJ ? (int)(pm_b & ((1u << logHW) - 1u)) : 0;
  const int64_t pm_img0 = PMJ ? (int64_t)(pm_b >> logHW) * BM : 0;
  auto grow = [&](int row) -> int64_t { return PMJ ? ((pm_img0 + row) << logHW) + pm_pos : m0 + row; };  // GEMM row of a tile row
  // the taps of this block's K loop: all of them, or (position-major) those inside the image, 4 bits each
  // displacement (and, for the folded data gradient, output class) behind tap index `tap` of this launch
  auto tap_delta = [&](int tap, int& dy, int& dx, int& cq) {
    cq = 0;
    if (a.fold == 1) {         // source pixel (i-1+pa+dr, j-1+pb+dc) of output class (pa, pb)
      dy = (cls >> 1) - 1 + (tap >> 1); dx = (cls & 1) - 1 + (tap & 1);
    } else if (a.fold == 2) {  // adjoint: the output pixel of class cq whose tap (dr, dc) read this source pixel
      cq = tap >> 2;
      dy = -((cq >> 1) - 1 + ((tap >> 1) & 1)); dx = -((cq & 1) - 1 + (tap & 1));
    } else {
      dy = a.ntaps == 9 ? tap / 3 - 1 : 0; dx = a.ntaps == 9 ? tap % 3 - 1 : 0;
    }
  };
  uint64_t tap_list = 0;
  int ntv = a.ntaps;
  if (PMJ) {
    ntv = 0;
    const int py = (int)morton_y((uint32_t)pm_pos), px = (int)morton_x((uint32_t)pm_pos);
    for (int t = 0; t < a.ntaps; ++t) {
      int dy, dx, cq;
      tap_delta(t, dy, dx, cq);
      const int yy = py + dy, xx = px + dx;
      if (yy >= 0 && yy < H && xx >= 0 && xx < H) { tap_list |= (uint64_t)t << (4 * ntv); ++ntv; }
    }
  }

  // neighbour table: the Morton decode / re-encode happens once per (row, tap), not once per K-chunk
  for (int e = tid; e < BM * a.ntaps; e += NT) {
    const int row = e / a.ntaps, tap = e - row * a.ntaps;
    const int64_t m = grow(row);
    const uint32_t p = (uint32_t)(m & (((int64_t)1 << logHW) - 1));
    int dy, dx, cq;
    tap_delta(tap, dy, dx, cq);
    const int yy = (int)morton_y(p) + dy, xx = (int)morton_x(p) + dx;
    const bool ok = m < a.M && yy >= 0 && yy < H && xx >= 0 && xx < H;
    uint32_t src = morton((uint32_t)yy, (uint32_t)xx);
    if (a.ups) src >>= 2;
    const int64_t img = m >> logHW;
    const int srow = a.fold == 2 ? (int)(img * (HWin << 2) + 4 * src + cq) : (int)(img * HWin + src);
    // (LDS-DMA form: byte offsets; padding = an offset past the buffer descriptor's range, which loads zeros)
    if constexpr (DMA) nbr[row * NBS + tap] = !ok ? (int)0xFFFF0000u : (int)((uint32_t)srow * ((uint32_t)a.Cin * 4u));
    else nbr[row * NBS + tap] = !ok ? -1 : srow;
  }
  __syncthreads();

  const int ac4 = tid & 7;
  const char* in_bytes = reinterpret_cast<const char*>(a.in);
  const char* wp_bytes = reinterpret_cast<const char*>(a.wp);
  const uint32_t row_bytes = (uint32_t)a.Cin * 4u, wrow_bytes = (uint32_t)a.Cout * 4u;
  const int nci = a.Cin >> 5, nchunk = ntv * nci, niter = (nchunk + KH - 1) / KH;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x16){0};

  // Software pipeline with ONE load site: phase `it` issues the global loads of K-chunk pair `it` into registers, runs
  // the MFMAs of pair it-1 out of LDS while they fly, then (barrier) parks pair `it` in LDS.
  // Tried and measured on MI355X (A/B inside one box, c2 and c3 steps), all no better than this form:
  //   * a second register stage (loads get two MFMA phases to land): +-0 at c3, -3 % at c2 (registers -> occupancy);
  //   * issuing the loads one by one between the MFMA k-steps (sched_group_barrier VMEM groups): +-0 at c2, the
  //     128-row tiles 17-20 % slower at c3.
  //   * two LDS stages with ONE barrier per iteration (park pair it+1 in the other stage right behind the MFMAs of
  //     pair it; 75 KB per 64x64 block): +-0 at c2 for every launch-size cut-off tried (300 / 600 / all blocks).
  // split-K (small-M layers): grid.z slices the K iterations; each slice writes a raw partial slab
  const int it_per = (niter + a.ksplit - 1) / a.ksplit;
  const int it_beg = (int)blockIdx.z * it_per, it_end = it_beg + it_per < niter ? it_beg + it_per : niter;

#define IGEMM_LOAD(AREG, BREG, AMASK, IT)                                                                             \
  do {                                                                                                                \
    _Pragma("unroll") for (int h = 0; h < KH; ++h) {                                                                  \
      const int kc = KH * (IT) + h;                                                                                   \
      const bool live = kc < nchunk;                                                                                  \
      const int tap = live ? kc / nci : 0, cc = live ? kc - tap * nci : 0;                                            \
      /* 32-bit byte offsets from the (wave-uniform) tensor bases: every tensor here is < 4 GiB */                    \
      const uint32_t a_col = (uint32_t)(cc * 128 + ac4 * 16);                                                         \
      _Pragma("unroll") for (int q = 0; q < RA; ++q) {                                                                \
        const int src = nbr[((tid >> 3) + (NTC >> 3) * q) * NBS + tap];                                                \
        const bool ok = live && src >= 0;                                                                             \
        /* branch-free zero padding: always load (from a valid address); the 0/1 mask is applied when the */         \
        /* registers are parked in LDS (a select or multiply here would make the in-order issue wait)      */         \
        const uint32_t off = ok ? (uint32_t)src * row_bytes + a_col : 0u;                                             \
        AREG[h][q] = *reinterpret_cast<const f32x4*>(in_bytes + off);                                                 \
        AMASK[h][q] = ok ? 1.0f : 0.0f;                                                                               \
      }                                                                                                               \
      const uint32_t b_row0 =                                                                                         \
          (uint32_t)((cls * 4 * (a.fold == 1) + tap) * a.Cin + cc * 32) * wrow_bytes + (uint32_t)n0 * 4u;             \
      _Pragma("unroll") for (int q = 0; q < RB; ++q) {                                                                \
        const int idx = BPART ? (tid < NB16 ? tid : NB16 - 1) : tid + NTC * q;  /* clamped: spare threads re-read */   \
        {                                                                                                             \
          const int krow = idx / (BN / 4), c4 = idx % (BN / 4);                                                       \
          BREG[h][q] = *reinterpret_cast<const f32x4*>(wp_bytes + (b_row0 + (uint32_t)krow * wrow_bytes + (uint32_t)c4 * 16u)); \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
  } while (0)

  // (the second half of an odd last pair multiplies A-zeros: its B rows only have to be finite -- a re-read of tap 0)
#define IGEMM_STORE(AREG, BREG, AMASK)                                                                                \
  do {                                                                                                                \
    _Pragma("unroll") for (int h = 0; h < KH; ++h) {                                                                  \
      _Pragma("unroll") for (int q = 0; q < RA; ++q) {                                                                \
        const int row = (tid + NTC * q) >> 3;                                                                          \
        const float mk = AMASK[h][q];                                                                                 \
        const f32x4 v = AREG[h][q] * mk;                                                                              \
        *reinterpret_cast<f32x4*>(As + row * AP + h * 32 + ac4 * 4) = v;                                              \
      }                                                                                                               \
      _Pragma("unroll") for (int q = 0; q < RB; ++q) {                                                                \
        const int idx = tid + NTC * q;                                                                                \
        {                                                                                                             \
          const int krow = idx / (BN / 4), c4 = idx % (BN / 4);                                                       \
          if (!BPART || idx < NB16) *reinterpret_cast<f32x4*>(Bs + (h * 32 + krow) * BP + c4 * 4) = BREG[h][q];      \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
  } while (0)

  // operand reads run one k-step ahead of the MFMAs (register double buffer): the LDS latency of step s+1 hides
  // behind the matrix pipe working on step s.  The sched_group_barriers pin the interleave [LDS reads of step s+1]
  // then [MFMAs of step s] (hipcc otherwise re-serialises read -> wait -> MFMA; mask 0x100 = DS read, 0x008 = MFMA)
#define IGEMM_MFMA()                                                                                                  \
  do {                                                                                                                \
    const float* ap = As + (wm * TM * 32 + c) * AP + kg * BK + hh;                                                    \
    const float* bp = Bs + (kg * BK + hh) * BP + wn * TN * 32 + c;                                                    \
    float av[2][TM], bv[2][TN];                                                                                       \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) av[0][i] = ap[i * 32 * AP];                                        \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) bv[0][j] = bp[j * 32];                                             \
    _Pragma("unroll") for (int s = 0; s < BK / 2; ++s) {                                                              \
      const int cur = s & 1, nxt = cur ^ 1;                                                                           \
      if (s + 1 < BK / 2) {                                                                                           \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) av[nxt][i] = ap[i * 32 * AP + 2 * (s + 1)];                    \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) bv[nxt][j] = bp[2 * (s + 1) * BP + j * 32];                    \
      }                                                                                                               \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);               \
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);                                                        \
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);                                                        \
    }                                                                                                                 \
  } while (0)

  if constexpr (DMA) {
    constexpr int NW = NT / 64, PAW = BM / 8 / NW, PBW = BN / 8 / NW;  // 1 KiB pieces (8 rows) per wave and stage
    static_assert(PAW * NW * 8 == BM && PBW * NW * 8 == BN, "LDS-DMA pieces must divide the tile");
    typedef __attribute__((address_space(3))) void lds_void;
    unsigned char* stg = igemm_smem;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // buffer descriptors (raw, 32-bit byte offsets): a lane whose offset is past num_records writes ZEROS to LDS --
    // that is the zero padding of the im2col rows (checked on the MI355X: tools/buflds_oob_test.hip)
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)0xFFFF0000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, (int)0xFFFF0000u, 0x00020000);
    // per-lane pieces, constant over the kernel: lane l of piece e writes LDS slot (row = e*8 + l/8, slot l%8) and
    // therefore fetches the logical slot (l%8) ^ f(row), f(row) = (row >> 1) & 7 (128-byte rows: conflict-free b128 reads)
    int arow[PAW];
    uint32_t acol[PAW], bcol[PBW];
#pragma unroll
    for (int q = 0; q < PAW; ++q) {
      const int e = (wave * PAW + q) * 64 + lane, row = e >> 3, ps = e & 7;
      arow[q] = row * NBS;
      acol[q] = (uint32_t)((ps ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int q = 0; q < PBW; ++q) {
      const int e = (wave * PBW + q) * 64 + lane, n = e >> 3, ps = e & 7;
      bcol[q] = (uint32_t)(n0 + n) * row_bytes + (uint32_t)((ps ^ ((n >> 1) & 7)) << 4);  // Wk[tap][col][ci] f32
    }
    const uint32_t tap_bytes = (uint32_t)a.Cout * row_bytes;
    const uint32_t cls_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)((a.fold == 1 ? (uint32_t)cls * 4u : 0u) * tap_bytes));
    // chunk counters: `f*` runs over the chunk whose neighbour rows are fetched next, `i*` over the chunk issued next
    int ftap = it_beg / nci, fcc = it_beg - ftap * nci, itap = ftap, icc = fcc;
    uint32_t srcn[PAW];  // per-lane byte offsets of the next chunk's A pieces (without the channel-chunk offset)
    auto fetch_nbr = [&]() {
      const int tf = PMJ ? (int)((tap_list >> (4 * ftap)) & 15) : ftap;
#pragma unroll
      for (int q = 0; q < PAW; ++q) srcn[q] = (uint32_t)nbr[arow[q] + tf];  // (used a whole chunk later: no wait here)
      if (++fcc == nci) { fcc = 0; ++ftap; }
    };
    // LDS byte address of the stage ring as a SCALAR: the destination of a buffer_load ... lds travels in M0, and a
    // destination the compiler cannot prove wave-uniform costs a readfirstlane waterfall loop (~10 instructions and a
    // branch) per 1 KiB piece -- eight of them per chunk and wave in rounds 2-3a (visible in the ISA; the "100-185 cycles
    // per piece" of the issue-cost table).
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t stg_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_byte*)stg);
    auto issue = [&](int buf) {
      const uint32_t dstA = (uint32_t)__builtin_amdgcn_readfirstlane((int)(stg_lds + (uint32_t)(buf * STAGE + wave_u * PAW * 1024)));
      // (the instruction's scalar offset: forced into an SGPR -- the chunk counters are wave-uniform, but the compiler
      // keeps them in vector registers and then wraps every buffer_load in a readfirstlane waterfall loop)
      const int ccb = __builtin_amdgcn_readfirstlane(icc * 128);
#pragma unroll
      for (int q = 0; q < PAW; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(uintptr_t)(dstA + q * 1024), 16, (int)(srcn[q] + acol[q]), ccb, 0, 0);
      const uint32_t dstB = (uint32_t)__builtin_amdgcn_readfirstlane((int)(stg_lds + (uint32_t)(buf * STAGE + BM * 128 + wave_u * PBW * 1024)));
      const int ti = PMJ ? (int)((tap_list >> (4 * itap)) & 15) : itap;
      const int boff = __builtin_amdgcn_readfirstlane((int)(cls_off + (uint32_t)ti * tap_bytes) + ccb);
#pragma unroll
      for (int q = 0; q < PBW; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(uintptr_t)(dstB + q * 1024), 16, (int)bcol[q], boff, 0, 0);
      if (++icc == nci) { icc = 0; ++itap; }
    };
    // operand reads: lane (c, hh) owns floats 16 hh .. 16 hh + 15 of row / column c: logical slots 4 hh + j
    int offl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) offl[j] = c * 128 + (((4 * hh + j) ^ ((c >> 1) & 7)) << 4);
    if (it_beg < it_end) {
      fetch_nbr();
      issue(0);
      if (it_beg + 1 < it_end) fetch_nbr();
    }
    constexpr int NPC = PM == 4 ? 3 : 1;  // bf16 pieces per operand (forms 4 / 5)
    bf16x8v pa[NPC][TM], pb[NPC][TN];     // converted operands of the pending k-step (forms 4 / 5)
    if constexpr (PM == 4 || PM == 5) {
      const bf16x8v zero8 = __builtin_bit_cast(bf16x8v, (u32x4v){0u, 0u, 0u, 0u});
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) {
#pragma unroll
        for (int i = 0; i < TM; ++i) pa[pc][i] = zero8;
#pragma unroll
        for (int j = 0; j < TN; ++j) pb[pc][j] = zero8;
      }
    }
    for (int it = it_beg; it < it_end; ++it) {
      const int buf = (it - it_beg) & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of chunk `it` have landed ...
      __syncthreads();                                    // ... everyone's have, and stage buf^1 is free again
      if constexpr (PM != 3 || !IGEMM_LATE_ISSUE) {
        if (it + 1 < it_end) issue(buf ^ 1);
        if (it + 2 < it_end) fetch_nbr();
      }
      const unsigned char* Ab = stg + buf * STAGE + wm * TM * 32 * 128;
      const unsigned char* Bb = stg + buf * STAGE + BM * 128 + wn * TN * 32 * 128;
      if constexpr (PM == 4 || PM == 5) {
        // bf16 MFMA on the float32 LDS image: two k-steps of 16 per chunk; step st takes this lane's floats
        // 16 hh + 8 st .. + 7 (logical slots 4 hh + 2 st, + 1) of every row / column and converts them in registers.
        // Software pipeline INSIDE the wave, one k-step deep and across the chunk barrier: the MFMAs of k-step t run
        // while the VALU converts the operands of k-step t+1 (the waves of a block share its barriers and stay in phase,
        // so VALU work of one wave does not land under the MFMAs of another by itself: measured 37 % matrix-pipe use
        // with convert -> MFMA in sequence).  `pa / pb` carry the converted operands of the pending k-step into the
        // next iteration; they start as zeros (one k-step of MFMAs on zeros per block instead of a peeled first pass).
        constexpr int NMF = TM * TN * (PM == 4 ? 6 : 1);  // MFMAs per k-step
        // VALU instructions of one k-step's conversions (measured on the ISA: 11 per pair of elements of the split, the
        // round-to-nearest form one v_cvt_pk per pair) spread over the gaps between that phase's MFMAs
        constexpr int NVT = (PM == 4 ? 44 : 4) * (TM + TN);
        constexpr int NVA = (NVT + NMF - 1) / NMF;
        constexpr int LEAD = NMF >= 8 ? 3 : 1;  // MFMAs issued ahead of the first conversion (they cover the LDS read latency)
        auto mfma_step = [&](const bf16x8v (&xa)[NPC][TM], const bf16x8v (&xb)[NPC][TN]) {
          if constexpr (PM == 4) {
            // six piece products, smallest first (lo hi, hi lo, mid mid, mid hi, hi mid, hi hi), into ONE accumulator
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            constexpr int NT6 = 6;
#pragma unroll
            for (int t = 0; t < NT6; ++t)
#pragma unroll
              for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[PA[t]][i], xb[PB[t]][j], acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0][i], xb[0][j], acc[i][j], 0, 0, 0);
          }
        };
        auto convert_step = [&](const f32x4 (&xa)[TM][2], const f32x4 (&xb)[TN][2], bf16x8v (&oa)[NPC][TM], bf16x8v (&ob)[NPC][TN]) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            if constexpr (PM == 4) f32x8_split3(xa[i][0], xa[i][1], oa[0][i], oa[1][i], oa[2][i]);
            else oa[0][i] = f32x8_to_bf16(xa[i][0], xa[i][1]);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (PM == 4) f32x8_split3(xb[j][0], xb[j][1], ob[0][j], ob[1][j], ob[2][j]);
            else ob[0][j] = f32x8_to_bf16(xb[j][0], xb[j][1]);
          }
        };
        // both k-steps' raw operands of this chunk: 4 (TM + TN) ds_read_b128, issued at once
        f32x4 ra[2][TM][2], rb[2][TN][2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            ra[st][i][0] = *reinterpret_cast<const f32x4*>(Ab + i * 4096 + offl[2 * st]);
            ra[st][i][1] = *reinterpret_cast<const f32x4*>(Ab + i * 4096 + offl[2 * st + 1]);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            rb[st][j][0] = *reinterpret_cast<const f32x4*>(Bb + j * 4096 + offl[2 * st]);
            rb[st][j][1] = *reinterpret_cast<const f32x4*>(Bb + j * 4096 + offl[2 * st + 1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // (the reads and their address arithmetic: a scheduling region of their own)
        bf16x8v na[NPC][TM], nb[NPC][TN];
        // phase A: MFMAs of the pending k-step (the previous chunk's second) over the conversion of this chunk's first
        mfma_step(pa, pb);
        convert_step(ra[0], rb[0], na, nb);
        __builtin_amdgcn_sched_group_barrier(0x008, LEAD, 0);
#pragma unroll
        for (int g = LEAD; g < NMF; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x002, NVA + 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // phase B: MFMAs of this chunk's first k-step over the conversion of its second (pending into the next iteration)
        mfma_step(na, nb);
        convert_step(ra[1], rb[1], pa, pb);
#pragma unroll
        for (int g = 0; g < NMF; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, NVA, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      f32x4 av[2][TM], bv[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[0][i] = *reinterpret_cast<const f32x4*>(Ab + i * 4096 + offl[0]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[0][j] = *reinterpret_cast<const f32x4*>(Bb + j * 4096 + offl[0]);
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      // [reads of quarter j4+1] then [MFMAs of quarter j4]: the LDS latency elapses under the matrix pipe
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const int cur = j4 & 1, nxt = cur ^ 1;
        if (j4 < 3) {
#pragma unroll
          for (int i = 0; i < TM; ++i) av[nxt][i] = *reinterpret_cast<const f32x4*>(Ab + i * 4096 + offl[j4 + 1]);
#pragma unroll
          for (int j = 0; j < TN; ++j) bv[nxt][j] = *reinterpret_cast<const f32x4*>(Bb + j * 4096 + offl[j4 + 1]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i][e], bv[cur][j][e], acc[i][j], 0, 0, 0);
        if (j4 < 3) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
        if constexpr (IGEMM_LATE_ISSUE) {
          if (j4 == 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < it_end) issue(buf ^ 1);
            if (it + 2 < it_end) fetch_nbr();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    if constexpr (PM == 4 || PM == 5) {  // the last k-step's MFMAs
      if constexpr (PM == 4) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[PA[t]][i], pb[PB[t]][j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0][i], pb[0][j], acc[i][j], 0, 0, 0);
      }
    }
  } else {
    f32x4 aA[KH][RA], bA[KH][RB];
    float mA[KH][RA];
    for (int it = it_beg; it <= it_end; ++it) {
      if (it < it_end) IGEMM_LOAD(aA, bA, mA, it);
      if (it > it_beg) {
        IGEMM_MFMA();
      }
      __syncthreads();
      if (it < it_end) IGEMM_STORE(aA, bA, mA);
      __syncthreads();
    }
  }
#undef IGEMM_LOAD
#undef IGEMM_STORE
#undef IGEMM_MFMA

  if constexpr (WK > 1) {
    // combine the K groups through LDS (the staging tiles are dead: the loop ended on a barrier)
    float* xch = As;  // [TM][TN][16][NTG]
    for (int g = 1; g < WK; ++g) {
      if (kg == g) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[((i * TN + j) * 16 + r) * NTG + (tid - g * NTG)] = acc[i][j][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += xch[((i * TN + j) * 16 + r) * NTG + tid];
      }
      __syncthreads();
    }
  }

  // ---------------- epilogue (K groups other than 0 only keep the barriers company)
  if (a.ksplit > 1) {  // raw partial sums (quad-summed if asked); bias and BN partials happen in splitk_reduce_kernel
    const int64_t rows_out = a.poolsum ? a.M >> 2 : (a.fold == 1 ? a.M << 2 : a.M);
    float* slab = a.splitk_ws + (size_t)blockIdx.z * rows_out * a.Cout;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + c;
        if (a.poolsum) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float v = (acc[i][j][4 * g] + acc[i][j][4 * g + 1]) + (acc[i][j][4 * g + 2] + acc[i][j][4 * g + 3]);
            const int64_t m = m0 + wm * TM * 32 + i * 32 + 8 * g + 4 * hh;
            if (active && m < a.M) slab[(m >> 2) * a.Cout + col] = v;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * TM * 32 + i * 32 + crow16(r, hh);
            if (active && m < a.M) slab[(a.fold == 1 ? 4 * m + cls : m) * a.Cout + col] = acc[i][j][r];
          }
        }
      }
    return;
  }
  if (a.poolsum) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + c;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v = (acc[i][j][4 * g] + acc[i][j][4 * g + 1]) + (acc[i][j][4 * g + 2] + acc[i][j][4 * g + 3]);
          const int64_t m = m0 + wm * TM * 32 + i * 32 + 8 * g + 4 * hh;
          if (active && m < a.M) a.out[(m >> 2) * a.Cout + col] = v;
        }
      }
    return;
  }
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + c;
      const float bias = a.bias ? a.bias[a.bias_perm ? (col % a.bias_perm) * 4 + col / a.bias_perm
                                                : a.bias_mod ? col % a.bias_mod : col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = grow(wm * TM * 32 + i * 32 + crow16(r, hh));
        if (active && m < a.M) {
          const float v = acc[i][j][r] + bias;
          a.out[(a.fold == 1 ? 4 * m + cls : m) * a.Cout + col] = v;
          s1[j] += v;
          s2[j] = fmaf(v, v, s2[j]);
        }
      }
    }
  if (a.stats) {
    if constexpr (DMA) __syncthreads();  // every wave is done reading the stages `red` overlays
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 32, 64);
      s2[j] += __shfl_xor(s2[j], 32, 64);
      if (active && hh == 0) {
        red[(wm * BN + wn * TN * 32 + j * 32 + c) * 2] = s1[j];
        red[(wm * BN + wn * TN * 32 + j * 32 + c) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (tid < BN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
      float* dst = a.stats + ((size_t)blockIdx.x * a.Cout + n0 + tid) * 2;
      dst[0] = t1; dst[1] = t2;
    }
  }
}

// Sums the split-K slabs in order, adds the bias, writes the output and (optionally) the per-row-block BatchNorm
// partials in the layout of the unsplit kernel: stats[row_block][Cout][2], row blocks of `bm` OUTPUT pixels.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, int64_t rows, int C,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            float* __restrict__ stats, int bm, int bias_perm, int bias_mod) {
  __shared__ float red[2 * 256];
  // block = (row block of `bm` rows) x (32 channels); thread = (channel, one of 8 row lanes)
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int64_t r0 = (int64_t)blockIdx.x * bm;
  const int64_t r1 = r0 + bm < rows ? r0 + bm : rows;
  const int c = blockIdx.y * 32 + cl;
  const float b = bias ? bias[bias_perm ? (c % bias_perm) * 4 + c / bias_perm : bias_mod ? c % bias_mod : c] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t r = r0 + rl; r < r1; r += 8) {
    float v = b;
    for (int k = 0; k < ksplit; ++k) v += ws[((size_t)k * rows + r) * C + c];
    out[r * C + c] = v;
    s1 += v;
    s2 = fmaf(v, v, s2);
  }
  if (stats) {
    red[threadIdx.x] = s1;
    red[256 + threadIdx.x] = s2;
    __syncthreads();
    if (rl == 0) {
      float t1 = 0.f, t2 = 0.f;
      for (int j = 0; j < 8; ++j) { t1 += red[j * 32 + cl]; t2 += red[256 + j * 32 + cl]; }
      stats[((size_t)blockIdx.x * C + c) * 2] = t1;
      stats[((size_t)blockIdx.x * C + c) * 2 + 1] = t2;
    }
  }
}

// Tile configuration: 0 = 128x64, 4 = 128x128 (large launches), 1 = 64x64, 2 = 128x32 (Cout = 32), 3 = 32x64 (two waves) for launches whose 64-row
// tiling would leave most CUs without a block (small-M layers: a finer tiling fills the chip without the split-K slabs
// and their reduce pass).
static int igemm_cfg(int64_t M, int Cout) {
  if (Cout % 64) return 2;
  // 4 = 128x128 (wave tile 64x64: half the LDS operand reads and half the L2 -> LDS bytes per FLOP of 128x64) once
  // the launch still has >= 512 blocks of that size (two per CU are resident: 79 KB of LDS, 231 VGPRs).  c3: the
  // launches that qualify run at 82 instead of 59 TFLOP/s in-situ, step 22.9 -> 22.55 ms; 256 / 128 measured no
  // better, c2 has no such launch.  (option igemm_thr128 overrides the 512: the tests send small fixtures through the tile)
  const int64_t thr128 = opt(OPT_IGEMM_THR128);
  if (Cout % 128 == 0 && ceil_div(M, 128) * (Cout / 128) >= thr128) return 4;
  if (ceil_div(M, 128) * (Cout / 64) >= 512) return 0;
  if (ceil_div(M, 64) * (Cout / 64) >= 96) return 1;  // (>= 192: unsplit; 96..191: split-K beats finer tiles at c2)
  return 3;
}
template <int BM, int BN, int WM, int WN, int WK, int PM = 0>
static constexpr size_t conv_igemm_lds_bytes() {
  if (PM >= 3) return (size_t)(2 * (BM + BN) * 128);  // + the neighbour table (launch_igemm_cfg); the BN partials overlay a stage
  return sizeof(float) * (size_t)(BM * (64 * WK + 4) + 64 * WK * (BN + 4) + WM * BN * 2) + sizeof(int) * (size_t)(BM * 16);
}

template <int BM, int BN, int WM, int WN, int WK, int PM = 0>
static int launch_igemm_cfg(int id, double flops, dim3 grid, const ConvArgs& a, hipStream_t s) {
  // (LDS-DMA form: the neighbour table is sized by the launch's tap count; the other forms always carry 16 per row)
  const size_t lds = conv_igemm_lds_bytes<BM, BN, WM, WN, WK, PM>() + (PM >= 3 ? sizeof(int) * (size_t)BM * (a.ntaps > 9 ? 16 : 9) : 0);
  auto kern = conv_igemm_kernel<BM, BN, WM, WN, WK, PM>;
  static std::atomic<uint64_t> attr_done{0};  // one instantiation = one static
  constexpr size_t lds_max = conv_igemm_lds_bytes<BM, BN, WM, WN, WK, PM>() + (PM >= 3 ? sizeof(int) * (size_t)BM * 16 : 0);
  if (lds_max > 64 * 1024) DVG_TRY(raise_dynamic_lds(attr_done, (const void*)kern, (int)lds_max));
  DVG_LAUNCH_WORK(id, flops, kern, grid, dim3(WM * WN * WK * 64), lds, s, a);
  return DVG_OK;
}

// option igemm_dma = 0: the register-staged form of the float32 kernel instead of the LDS-DMA form (tests, A/B runs)
static int igemm_dma_env() { return opt(OPT_IGEMM_DMA) != 0 ? 1 : 0; }
// Operand form of one forward / data-gradient launch (the PM argument of conv_igemm_kernel): the process-wide mode
// (dvg_set_conv_precision) mapped onto the LDS-DMA kernels.  All three modes read the SAME float32 K-major weight pack.
int conv_launch_mode(int64_t gemm_rows, int Cout) {
  (void)gemm_rows; (void)Cout;
  const int mode = conv_precision_mode();
  if (mode == 0) return igemm_dma_env() != 0 ? 3 : 0;  // (measured faster for every tile configuration and launch size: tools/igemm_ab.py)
  return mode == 2 ? 4 : 5;
}
bool conv_pack_is_f32_kmajor(int launch_mode) { return launch_mode >= 3; }

static int igemm_bm(int cfg) { return cfg == 1 ? 64 : cfg == 3 ? 32 : 128; }
static int igemm_bn(int cfg) { return cfg == 2 ? 32 : cfg == 4 ? 128 : 64; }

int conv_stats_blocks(int64_t M, int Cout) { return (int)ceil_div(M, igemm_bm(igemm_cfg(M, Cout))); }

// fold = 1: tile config chosen on the OUTPUT pixel count (4 per source pixel); row blocks never straddle a class
bool conv_fold_ok(int64_t Msrc) { return Msrc > 0 && Msrc % 128 == 0; }
int conv_stats_blocks_fold(int64_t Msrc, int Cout) { return 4 * (int)ceil_div(Msrc, igemm_bm(igemm_cfg(4 * Msrc, Cout))); }

// K-split of the forward / data-gradient kernel: only for launches that would leave most CUs idle
// (fold = 1 callers pass M = 4 * source pixels, ntaps = 4: same block count and K depth as the launch)
int conv_igemm_ksplit(int64_t M, int Cin, int Cout, int ntaps) {
  const int cfg = igemm_cfg(M, Cout);
  const int64_t blocks = ceil_div(M, igemm_bm(cfg)) * (Cout / igemm_bn(cfg));
  const int niter = (ntaps * (Cin / 32) + 1) / 2;
  if (blocks >= 192 || niter < 6) return 1;  // measured: splitting a launch with >= 256 blocks loses to its reduce pass
  int64_t k = ceil_div(512, blocks);
  if (k > niter / 3) k = niter / 3;
  return (int)(k < 1 ? 1 : k);
}

size_t conv_splitk_floats(int64_t M, int Cin, int Cout, int ntaps, int poolsum) {
  const int k = conv_igemm_ksplit(M, Cin, Cout, ntaps);
  return k > 1 ? (size_t)k * (size_t)(poolsum ? M / 4 : M) * Cout : 0;
}

int launch_conv_igemm(const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  // (the packs were written in the matching format: launch_weight_pack_multi makes the same decision per job)
  a.bf16 = conv_launch_mode(a.fold == 1 ? a.M * 4 : a.M, a.Cout);
  if (a.force_f32 && a.bf16 > 3) a.bf16 = 3;
  const bool taps_ok = a.fold == 1 ? a.ntaps == 4 : a.fold == 2 ? a.ntaps == 16 : (a.ntaps == 9 || a.ntaps == 1);
  if (a.fold && (a.ups || a.poolsum || !conv_fold_ok(a.M))) {
    set_error("conv_igemm: fold=%d needs ups=poolsum=0 and whole 128-row blocks (M=%lld)", a.fold, (long long)a.M);
    return DVG_E_INVALID;
  }
  if (a.Cin % 32 || a.Cout % 32 || a.M <= 0 || !taps_ok) {
    set_error("conv_igemm: unsupported shape Cin=%d Cout=%d M=%lld ntaps=%d", a.Cin, a.Cout, (long long)a.M, a.ntaps);
    return DVG_E_INVALID;
  }
  // the kernel addresses its input and weights with 32-bit byte offsets
  const double in_bytes = (double)(a.ups ? a.M / 4 : (a.fold == 2 ? a.M * 4 : a.M)) * a.Cin * 4.0;
  if (in_bytes >= 4294901760.0 || a.M >= 2147483647LL) {  // (0xFFFF0000: offsets from there up mean "padding")
    set_error("conv_igemm: input tensor of %.0f bytes exceeds the 4 GiB addressing range of this kernel", in_bytes);
    return DVG_E_UNSUPPORTED;
  }
  // FLOPs executed (the folded forms do 4/9 of the 9-tap count): what the roofline is priced on
  const int64_t Mg = a.fold == 1 ? a.M * 4 : a.M;  // GEMM rows over all classes
  const double flops = 2.0 * (double)Mg * a.Cin * a.Cout * a.ntaps;
  a.ksplit = a.splitk_ws ? conv_igemm_ksplit(Mg, a.Cin, a.Cout, a.ntaps) : 1;
  double flops_exec = flops;
  {
    // position-major tiles: plain 3x3 launches of the LDS-DMA form whose images fill whole row blocks (then the grid and
    // the BatchNorm partial rows are what they were).  option igemm_posmajor = 0: pixel-major tiles everywhere (A/B runs, tests)
    const int bm_ = igemm_bm(igemm_cfg(Mg, a.Cout));
    const int64_t nimg = a.L >= 1 ? a.M >> (2 * a.L) : 0;
    const bool taps9 = !a.fold && a.ntaps == 9;
    a.posmajor = opt(OPT_IGEMM_POSMAJOR) != 0 && a.bf16 >= 3 && (taps9 || a.fold) && !a.ups && !a.poolsum && a.ksplit == 1 &&
                 a.L >= 1 && a.L <= 5 && nimg > 0 && (nimg << (2 * a.L)) == a.M && nimg % bm_ == 0;
    if (a.posmajor) {
      // executed FLOPs: the (pixel, [class,] tap) combinations whose displacement stays inside the image
      const int H = 1 << a.L;
      int64_t pairs = 0;
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < H; ++x)
          for (int c4 = 0; c4 < (a.fold == 1 ? 4 : 1); ++c4)
            for (int t = 0; t < a.ntaps; ++t) {
              int dy, dx;
              if (a.fold == 1) { dy = (c4 >> 1) - 1 + (t >> 1); dx = (c4 & 1) - 1 + (t & 1); }
              else if (a.fold == 2) { const int cq = t >> 2; dy = -((cq >> 1) - 1 + ((t >> 1) & 1)); dx = -((cq & 1) - 1 + (t & 1)); }
              else { dy = t / 3 - 1; dx = t % 3 - 1; }
              pairs += (y + dy >= 0 && y + dy < H && x + dx >= 0 && x + dx < H) ? 1 : 0;
            }
      flops_exec = 2.0 * (double)nimg * (double)pairs * a.Cin * a.Cout;
    }
  }
  const unsigned kz = (unsigned)a.ksplit;
  const int cfg = igemm_cfg(Mg, a.Cout);
  const unsigned cm = a.fold == 1 ? 4u : 1u;
  const int bm = igemm_bm(cfg), bn = igemm_bn(cfg);
  const dim3 grid(cm * (unsigned)ceil_div(a.M, bm), (unsigned)(a.Cout / bn), kz);
  int rc;
  // (weight-space products of the composed decoder layers: their own profiler id, always float32)
  const bool wspace = a.force_f32 != 0;
#define IGEMM_LAUNCH_PM(PMV, FL)                                                                                        \
  switch (cfg) {                                                                                                        \
    case 0: rc = launch_igemm_cfg<128, 64, 2, 2, 1, PMV>(wspace ? K_IGEMM_WSPACE : K_IGEMM_128x64, FL, grid, a, s); break;   \
    case 1: rc = launch_igemm_cfg<64, 64, 2, 2, 1, PMV>(wspace ? K_IGEMM_WSPACE : K_IGEMM_64x64, FL, grid, a, s); break;     \
    case 3: rc = launch_igemm_cfg<32, 64, 1, 2, 1, PMV>(wspace ? K_IGEMM_WSPACE : K_IGEMM_32x64, FL, grid, a, s); break;     \
    case 4: rc = launch_igemm_cfg<128, 128, 2, 2, 1, PMV>(wspace ? K_IGEMM_WSPACE : K_IGEMM_128x128, FL, grid, a, s); break; \
    default: rc = launch_igemm_cfg<128, 32, 4, 1, 1, PMV>(wspace ? K_IGEMM_WSPACE : K_IGEMM_128x32, FL, grid, a, s); break;  \
  }
  if (a.bf16 == 3) { IGEMM_LAUNCH_PM(3, flops_exec) }
  else if (a.bf16 == 4) { IGEMM_LAUNCH_PM(4, flops_exec) }
  else if (a.bf16 == 5) { IGEMM_LAUNCH_PM(5, flops_exec) }
  else { IGEMM_LAUNCH_PM(0, flops) }
#undef IGEMM_LAUNCH_PM
  DVG_TRY(rc);
  if (a.ksplit > 1) {
    // row blocks in units of OUTPUT rows; for the BN partials they coincide with the unsplit kernel's blocks
    const int64_t rows_out = a.poolsum ? a.M / 4 : Mg;
    const int bm = igemm_bm(cfg) / (a.poolsum ? 4 : 1);
    DVG_LAUNCH(K_MISC, splitk_reduce_kernel, dim3((unsigned)ceil_div(rows_out, bm), a.Cout / 32), dim3(256), 0, s, a.splitk_ws, a.ksplit,
               rows_out, a.Cout, a.bias, a.out, a.poolsum ? nullptr : a.stats, bm, a.bias_perm, a.bias_mod);
  }
  return DVG_OK;
}

}  // namespace dvg
