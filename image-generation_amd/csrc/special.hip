// Layers whose GEMM shape is degenerate (1 input or 1 output channel, or a 4-element dot):
// direct VALU kernels.  encoder conv0 1->32 (/root/reference/src/encoder.py:28-30 with
// channels[0]=1), encoder Linear(4,1) (:41), decoder ConvTranspose 32->1 and 1->1
// (/root/reference/src/decoder.py:34-38 last iteration, :49-51).
#include <cstdlib>
#include "kernels.h"
#include "conv.h"

namespace dvg {

// ------------------------------------------------------------------------------ encoder conv0
// images (B,32,32) row-major {0,1}; W (32,1,3,3); Y [B*1024 (Morton)][32]
// As a GEMM: out[32 pixels][32 channels] = A[32 pixels][K = 10] x B[10][32], A = the 9 shifted input pixels and a
// column of ones, B = the 9 taps and the bias: five f32 MFMAs (k = 2 each) per 32-pixel tile, operands straight from
// global memory / registers (no LDS), two tiles in flight per wave.  The kernel is bound by its 128 B/pixel of output.
typedef float f32x16c __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void enc_conv0_fwd_kernel(const float* __restrict__ img, int64_t B,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ Y, float* __restrict__ stats_part) {
  __shared__ double red[2 * 4 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, c = lane & 31;
  // B operand of MFMA j: B[k = 2j + hh][co = c]
  float bw[5];
  int dy[5], dx[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int t = 2 * j + hh;
    bw[j] = t < 9 ? w[c * 9 + t] : bias[c];
    dy[j] = t < 9 ? t / 3 - 1 : 0;
    dx[j] = t < 9 ? t % 3 - 1 : 0;
  }
  const int64_t tiles = B * 32;  // 32 pixels each (B * 1024 pixels, Morton order within an image)
  // BatchNorm partials in double: a lane adds up to ~700 values at c3, and the variance is E[y^2] - E[y]^2 of a map that is
  // mostly one constant (the background of a binary image): float32 running sums put 2e-5 on 1/sigma, enough to re-route
  // dozens of near-tie pooling windows further down (measured against float64).  The kernel is bound by its stores.
  double s1 = 0.0, s2 = 0.0;
  constexpr int TU = 2;
  for (int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * TU; t0 < tiles; t0 += (int64_t)gridDim.x * 4 * TU) {
    float av[TU][5];
    bool ok[TU][5];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int64_t tile = t0 + u < tiles ? t0 + u : tiles - 1;
      const int64_t m = tile * 32 + c;  // A[row = c][k = 2j + hh]
      const uint32_t p = (uint32_t)(m & 1023);
      const int y = (int)morton_y(p), x = (int)morton_x(p);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int yy = y + dy[j], xx = x + dx[j];
        ok[u][j] = yy >= 0 && yy < 32 && xx >= 0 && xx < 32;
        av[u][j] = img[(m >> 10) * 1024 + (ok[u][j] ? yy * 32 + xx : 0)];
      }
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      if (t0 + u < tiles) {  // wave-uniform
        f32x16c acc = {0};
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const float a = (2 * j + hh == 9) ? 1.0f : (ok[u][j] ? av[u][j] : 0.f);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[j], acc, 0, 0, 0);
        }
        float* dst = Y + ((t0 + u) * 32) * 32 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[r];
          dst[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32] = v;
          s1 += (double)v;
          s2 = fma((double)v, (double)v, s2);
        }
      }
    }
  }
  // per-block BatchNorm partials: lanes c and c + 32 hold the same channel
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if (hh == 0) { red[wave * 32 + c] = s1; red[128 + wave * 32 + c] = s2; }
  __syncthreads();
  if (tid < 32) {
    stats_part[((size_t)blockIdx.x * 32 + tid) * 2] = (float)((red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]));
    stats_part[((size_t)blockIdx.x * 32 + tid) * 2 + 1] =
        (float)((red[128 + tid] + red[160 + tid]) + (red[192 + tid] + red[224 + tid]));
  }
}

// Grid cap of the kernels that emit BatchNorm partials per block: below the 1024 rows at which launch_bn_finalize
// inserts its fold pass (768 instead of 1024: c2 1.088 -> 1.080 ms, two launches fewer on the chain; c3 unchanged).
static int special_block_cap() { return 768; }
int enc_conv0_blocks(int64_t B) { const int64_t b = ceil_div(B * 32, 8); return (int)(b > special_block_cap() ? special_block_cap() : b); }

int launch_enc_conv0_fwd(const float* images, int64_t B, const float* w, const float* b, float* Y, float* stats_part,
                         hipStream_t s) {
  DVG_LAUNCH(K_ENC_CONV0_FWD, enc_conv0_fwd_kernel, dim3((unsigned)enc_conv0_blocks(B)), dim3(256), 0, s, images, B, w, b, Y,
             stats_part);
  return DVG_OK;
}

// ------------------------------------------------------------------------------ encoder layer 0, recomputed
// The layer's output Y0 is the largest tensor of the network (c3: 537 MB) and the cheapest to make (nine taps of ONE input
// channel: five MFMAs per 32 pixels from a 4 KB image that sits in cache).  Round 3: nothing stores or reads it any more.
// Every pass that needed it -- the BatchNorm statistics, BN -> MaxPool -> LeakyReLU, the two backward passes of that stage,
// the weight gradient -- recomputes the 32-pixel tile with the same five MFMAs (bit-identical every time) and works on
// the accumulator registers: a lane holds channel c of pixels crow(r, hh), i.e. FOUR WHOLE pooling quads (r = 4g .. 4g+3),
// so pooling, arg-max and the BatchNorm backward are lane-local.  HBM traffic of the stage at c3: 2.9 GB -> 0.45 GB.
//   moments  (enc_l0_moments_kernel, below) the BatchNorm statistics of the layer WITHOUT making its output: y = w . patch + b
//           is linear in the nine shifted copies in_t of the one input channel, so sum y and sum y^2 over the batch are
//           w^T S + M b and quadratic forms w^T P w in the first and second moments S[t] = sum in_t, P[t][t'] = sum in_t in_t'
//           of the 3x3 patches: 54 numbers that do not depend on the channel (round 5; MODE 0 of rounds 3-5 redid the five
//           MFMAs per tile and summed the 32 x 32 accumulator in double: 63 us alone at c3, 123 us beside the draw)
//   MODE 1  Xp = LeakyReLU(MaxPool(BN(y)))                                    (replaces enc_bn_pool_fwd's read of Y0)
//   (MODE 2 / 3 of round 3 -- the backward in two passes: per-block (sum dz, sum dz zhat), then the weight gradient with
//   dY formed in registers -- were the A/B reference of MODE 4 and are gone since round 5)
//   MODE 4  the whole backward of the stage in ONE pass.  dY = gi (delta dz - m1 - zhat m2) is linear in the two batch means m1, m2, so the
//           weight gradient is gi (S - m1 T1 - m2 T2) with S = sum delta dz (x) in_t, T2 = sum zhat (x) in_t and T1 = sum in_t.
//           T1 and T2 need no pass over the pixels at all: T1 is the first moment and, zhat being (y - mu) invstd with y linear
//           in the patch, T2[co][t] = invstd (sum_t' w[co][t'] P[t'][t] + (b - mu) S[t]) -- both come from the 54 moments the
//           forward call left in the workspace (enc_l0_combine_kernel, in double).  The pass accumulates S (one MFMA
//           accumulation) and sum dz zhat.  part [blocks][ENC_L0_ROW]: S (320), sum dz zhat (32).

constexpr int ENC_L0_ROW = ENC_L0_ROW_FLOATS;  // floats per block of MODE 4's partials: S [0, 320), sum dz zhat [320, 352)
// MODE 4's weight-space accumulation runs on v_mfma_f32_16x16x4_f32 (round 5): the product is [32 channels] x [10
// tap columns] over the pixels, and a 32x32x2 MFMA spends 64 cycles on 32 columns of which 10 are used; the 16 x 16 tile
// spends 32 on 16.  Its A operand wants 16 channels x 4 pixels per instruction where the recomputed tile (a 32x32
// accumulator) holds 32 channels x 2 pixels per register: ONE v_permlane16_swap of two registers (pixel rows r, r + 1)
// gives the low-channel and the high-channel operand of the 4 pixels (r, r + 1) x (hh = 0, 1) -- which are x = 0..3 of
// one row of the 8 x 4 tile, so a lane's B value (tap t = lane & 15 of pixel slot k = lane >> 4) is a fixed offset from
// one per-lane base.  16 MFMAs of 32 cycles per tile (32 while T2 was accumulated here, 32 of 64 cycles before that).
typedef float f32x4c __attribute__((ext_vector_type(4)));
// PAD (round 5): the images are read from the zero-padded copy the moments kernel leaves in the workspace ([34][40] floats
// per image, interior at row 1 / column 4, ones in eight pad cells, zeros in eight others): a tap outside the image reads
// a zero, the bias column of the operands reads a one, and a load's address is a wave-uniform image base + a per-lane
// constant + a compile-time offset -- no bounds tests, no selects (they were a third of this kernel's 450 vector
// instructions per tile).  Evaluation-mode forwards (no moments pass) keep the bounds-tested reads of the plain image.
constexpr int ENC_L0_ONES = 37, ENC_L0_ZEROS = 38;  // cells c, c + 40 r (r < 4) and those + 4 hold the constant
template <int MODE, bool PAD>
__global__ __launch_bounds__(256) void enc_l0_kernel(EncL0Args a) {
  static_assert(MODE == 1 || MODE == 4, "BN -> pool -> LeakyReLU, the backward in one pass");
  __shared__ float redf[MODE == 4 ? 4 * 32 * 33 : 1];
  __shared__ float redh[MODE == 4 ? 4 * 32 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, c = lane & 31;
  float bw[5];
  int dy[5], dx[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int t = 2 * j + hh;
    bw[j] = t < 9 ? a.w[c * 9 + t] : a.bias[c];
    dy[j] = t < 9 ? t / 3 - 1 : 0;
    dx[j] = t < 9 ? t % 3 - 1 : 0;
  }
  float mu = 0.f, is = 0.f, gm = 0.f, bt = 0.f;
  if (MODE >= 1) { mu = a.mean[c]; is = a.invstd[c]; gm = a.gamma[c]; bt = a.beta[c]; }
  // MODE 4: this lane's column of the 16x16x4 B operand: tap tq of pixel slot kq (tq = 9: the column of ones that
  // carries the bias gradient, 10..15: zero)
  const int tq = lane & 15, kq = lane >> 4;
  const int tdy = tq < 9 ? tq / 3 - 1 : 0, tdx = tq < 9 ? tq % 3 - 1 : 0;
  f32x4c s_lo = {0}, s_hi = {0};  // S, channels 0-15 / 16-31: rows 4 kq + reg, column tq
  const int cy = (int)morton_y((uint32_t)c), cx = (int)morton_x((uint32_t)c);  // pixel c of a tile, inside the tile
  const int64_t tiles = a.B * 32;
  float r2 = 0.f;
  // PAD: this lane's cell of the padded image for tap 2 j + hh of its pixel (tap 9 = the bias column: a one), and for the
  // B operand's tap tq of pixel slot kq (tq = 9: ones, above: zeros), both relative to the tile's corner
  int aoff[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) aoff[j] = (cy + dy[j] + 1) * 40 + cx + dx[j] + 4;
  const int boff = (tdy + 1) * 40 + kq + tdx + 4, bconst = tq == 9 ? ENC_L0_ONES : ENC_L0_ZEROS;
  constexpr int TU = 2;
  for (int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * TU; t0 < tiles; t0 += (int64_t)gridDim.x * 4 * TU) {
    float av[TU][5], gov[TU][4], bq[TU][8];
    bool ok[TU][5];
    // every load of the TU tiles goes out before the first MFMA needs one (clamped, always-valid addresses)
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int64_t tile = t0 + u < tiles ? t0 + u : tiles - 1;
      // a tile = 32 consecutive Morton indices = an 8 (x) by 4 (y) block of its image: corner from the tile index
      // (wave-uniform scalar work), the pixel's place inside from the lane (constants)
      const int ti = (int)(tile & 31);
      const int ty0 = 4 * ((ti & 1) + 2 * ((ti >> 2) & 1) + 4 * ((ti >> 4) & 1)), tx0 = 8 * (((ti >> 1) & 1) + 2 * ((ti >> 3) & 1));
      const int y = ty0 + cy, x = tx0 + cx;
      if constexpr (PAD) {
        const float* im = a.pimg + (int64_t)__builtin_amdgcn_readfirstlane((int)(tile >> 5)) * ENC_L0_PIMG;
        const int toff = ty0 * 40 + tx0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          ok[u][j] = true;
          av[u][j] = im[(j == 4 && hh == 1) ? ENC_L0_ONES : toff + aoff[j]];
        }
        if (MODE == 4) {
          const int bidx = tq < 9 ? toff + boff : bconst;
#pragma unroll
          for (int q = 0; q < 8; ++q) bq[u][q] = im[bidx + (q & 3) * 40 + 4 * (q >> 2)];
        }
      } else {
        const float* im = a.img + (tile >> 5) * 1024;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int yy = y + dy[j], xx = x + dx[j];
          ok[u][j] = yy >= 0 && yy < 32 && xx >= 0 && xx < 32;
          av[u][j] = im[ok[u][j] ? yy * 32 + xx : 0];
        }
        if (MODE == 4) {
          // B operand of pixel-row pair q: the pixel at (row q & 3, x = 4 (q >> 2) + kq) of the tile, shifted by tap tq
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int yy = ty0 + (q & 3) + tdy, xx = tx0 + 4 * (q >> 2) + kq + tdx;
            const bool inb = yy >= 0 && yy < 32 && xx >= 0 && xx < 32;
            const float v = im[inb ? yy * 32 + xx : 0];
            bq[u][q] = tq == 9 ? 1.0f : ((tq < 9 && inb) ? v : 0.f);
          }
        }
      }
      if (MODE == 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gov[u][g] = a.dXp[(tile * 8 + 2 * g + hh) * 32 + c];
      }
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      if (t0 + u >= tiles) continue;  // wave-uniform
      const int64_t tile = t0 + u;
      f32x16c acc = {0};
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const float av_ = PAD ? av[u][j] : ((2 * j + hh == 9) ? 1.0f : (ok[u][j] ? av[u][j] : 0.f));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av_, bw[j], acc, 0, 0, 0);
      }
      float a1[16];  // MODE 4: per pixel row r of the accumulator: delta dz (dz at the window's arg-max, else 0)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // the window of enc_bn_pool_*: zhat, arg-max (first maximum wins), LeakyReLU slope at the pooled value
        float zh[4], best = 0.f;
        int arg = 0;
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) {
          zh[sq] = (acc[4 * g + sq] - mu) * is;
          const float z = fmaf(zh[sq], gm, bt);
          if (sq == 0 || z > best) { best = z; arg = sq; }
        }
        if (MODE == 1) {
          a.Xp[(tile * 8 + 2 * g + hh) * 32 + c] = best < 0.f ? best * LRELU_SLOPE : best;
        } else {
          const float slope = !(best > 0.f) ? LRELU_SLOPE : 1.0f;
          const float dz = gov[u][g] * slope;
          r2 = fmaf(dz, zh[arg], r2);
#pragma unroll
          for (int sq = 0; sq < 4; ++sq) a1[4 * g + sq] = (sq == arg) ? dz : 0.f;
        }
      }
      if (MODE == 4) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const auto sa = __builtin_amdgcn_permlane16_swap(__float_as_uint(a1[2 * q]), __float_as_uint(a1[2 * q + 1]), false, false);
          const float b = bq[u][q];
          s_lo = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(sa[0]), b, s_lo, 0, 0, 0);
          s_hi = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(sa[1]), b, s_hi, 0, 0, 0);
        }
      }
    }
  }
  if (MODE == 4) {
    // D[row = co][col = t]: a lane holds column tq, rows 4 kq + reg (channels 0-15 in s_lo, 16-31 in s_hi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      redf[(wave * 32 + 4 * kq + r) * 33 + tq] = s_lo[r];
      redf[(wave * 32 + 16 + 4 * kq + r) * 33 + tq] = s_hi[r];
    }
    r2 += __shfl_xor(r2, 32, 64);
    if (hh == 0) redh[wave * 32 + c] = r2;
    __syncthreads();
    for (int w = tid; w < 320; w += 256) {
      const int co = w < 288 ? w / 9 : w - 288, t = w < 288 ? w % 9 : 9;
      a.part[(size_t)blockIdx.x * ENC_L0_ROW + w] = (redf[(0 * 32 + co) * 33 + t] + redf[(1 * 32 + co) * 33 + t]) +
                                                    (redf[(2 * 32 + co) * 33 + t] + redf[(3 * 32 + co) * 33 + t]);
    }
    if (tid < 32) a.part[(size_t)blockIdx.x * ENC_L0_ROW + 320 + tid] = (redh[tid] + redh[32 + tid]) + (redh[64 + tid] + redh[96 + tid]);
  }
}

// blocks: MODE 4: enc_l0_blocks(B) rows of `part` (2048 rows: the passes are bound by load latency, and 512 blocks are
// two per CU)
int enc_l0_blocks(int64_t B) { const int64_t b = ceil_div(B * 32, 8); return (int)(b > STREAM_BLOCKS ? STREAM_BLOCKS : b); }

int launch_enc_l0(int mode, const EncL0Args& a, hipStream_t s) {
  const dim3 g((unsigned)enc_l0_blocks(a.B)), g1(2048 < a.B * 4 ? 2048 : (unsigned)(a.B * 4));
  const bool pad = a.pimg != nullptr;  // (the padded copy exists: a training-mode forward's moments pass wrote it)
  switch (mode) {
    case 1:
      if (pad) DVG_LAUNCH(K_ENC_BN_POOL_FWD, (enc_l0_kernel<1, true>), g1, dim3(256), 0, s, a);
      else DVG_LAUNCH(K_ENC_BN_POOL_FWD, (enc_l0_kernel<1, false>), g1, dim3(256), 0, s, a);
      break;
    case 4:
      if (pad) DVG_LAUNCH(K_ENC_CONV0_WGRAD, (enc_l0_kernel<4, true>), g, dim3(256), 0, s, a);
      else DVG_LAUNCH(K_ENC_CONV0_WGRAD, (enc_l0_kernel<4, false>), g, dim3(256), 0, s, a);
      break;
    default: set_error("launch_enc_l0: mode %d", mode); return DVG_E_INVALID;
  }
  return DVG_OK;
}

// ---- first and second moments of the 3x3 patches of the batch (the statistics pass of the recomputed layer 0)
// Term k of the ENC_L0_MOM = 54: k < 9: S[t = k] = sum in_t;  k >= 9: P[t][t'] = sum in_t in_t' for the k-th pair t <= t'
// in row order ((0,0), (0,1), .., (0,8), (1,1), ..).  in_t of a pixel = the image at (y + t / 3 - 1, x + t % 3 - 1), zero
// outside (the zero padding of the convolution).  Double sums: a product of two floats is exact in double, so the
// moments carry nothing but the summation error of 10^6..10^7 terms in double -- the statistics that come out of them
// are those of the EXACT layer output (the float32 accumulator of the MFMA form rounds each y first).
// A block stages an image in LDS (zero halo, rows of 40 floats with the interior at column 4: 16-byte rows), one float4
// per thread, the next image's load in flight meanwhile.  Wave w owns the terms k = w (mod 4) -- 14 or 13 double
// accumulators per lane instead of 54 -- for ALL pixels of the image: a lane walks four 1 x 4 strips (three 6-wide rows
// each = 3 x (b32, b128, b32) LDS reads).  part [blocks][ENC_L0_MOM_ROW] doubles.
__host__ __device__ constexpr int enc_l0_pair_index(int t, int u) { return 9 + t * 9 - t * (t - 1) / 2 + (u - t); }  // t <= u

template <int ROLE>
__device__ __forceinline__ void enc_l0_moments_strip(const float* __restrict__ T, int y, int x0, double (&acc)[14]) {
  double d[3][6];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float* row = T + (y + r) * 40 + 3 + x0;
    const f32x4c mid = *reinterpret_cast<const f32x4c*>(row + 1);
    d[r][0] = (double)row[0]; d[r][1] = (double)mid[0]; d[r][2] = (double)mid[1]; d[r][3] = (double)mid[2];
    d[r][4] = (double)mid[3]; d[r][5] = (double)row[5];
  }
#pragma unroll
  for (int px = 0; px < 4; ++px) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if ((t & 3) == ROLE) acc[t >> 2] += d[t / 3][px + t % 3];
#pragma unroll
      for (int u = t; u < 9; ++u) {
        const int k = enc_l0_pair_index(t, u);
        if ((k & 3) == ROLE) acc[k >> 2] = fma(d[t / 3][px + t % 3], d[u / 3][px + u % 3], acc[k >> 2]);
      }
    }
  }
}

template <int ROLE>
__device__ __forceinline__ void enc_l0_moments_wave(const float* __restrict__ T, int lane, double (&acc)[14]) {
#pragma unroll
  for (int sp = 0; sp < 4; ++sp) {
    const int strip = lane + 64 * sp;  // 256 strips of an image: row strip >> 3, columns 4 (strip & 7) ..
    enc_l0_moments_strip<ROLE>(T, strip >> 3, 4 * (strip & 7), acc);
  }
}

__global__ __launch_bounds__(256) void enc_l0_moments_kernel(const float* __restrict__ img, int64_t B, double* __restrict__ part,
                                                             float* __restrict__ pimg) {
  __shared__ __align__(16) float tile[2][34 * 40];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 34 * 40; i += 256) (&tile[0][0])[i] = 0.f;
  __syncthreads();
  // the constant cells enc_l0_kernel<., PAD> reads for its bias column (pad columns: the strips below never touch them)
  if (tid < 16) {
    const int bf = tid >> 3, r = tid & 3, second = (tid >> 2) & 1;
    tile[bf][r * 40 + ENC_L0_ONES + 4 * second] = 1.0f;
  }
  double acc[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) acc[i] = 0.0;
  int64_t b = blockIdx.x;
  f32x4c cur = {0.f, 0.f, 0.f, 0.f};
  if (b < B) cur = *reinterpret_cast<const f32x4c*>(img + b * 1024 + tid * 4);
  __syncthreads();
  for (int it = 0; b < B; b += gridDim.x, ++it) {
    float* T = tile[it & 1];
    // (the buffer was last read two images ago: every thread has passed the barrier of the image in between since)
    *reinterpret_cast<f32x4c*>(T + ((tid >> 3) + 1) * 40 + 4 + 4 * (tid & 7)) = cur;
    __syncthreads();
    if (b + gridDim.x < B) cur = *reinterpret_cast<const f32x4c*>(img + (b + gridDim.x) * 1024 + tid * 4);
    // the padded image as it stands in LDS goes to the workspace: the two passes that recompute the layer read it from there
    for (int i = tid; i < ENC_L0_PIMG / 4; i += 256)
      reinterpret_cast<f32x4c*>(pimg + b * ENC_L0_PIMG)[i] = reinterpret_cast<const f32x4c*>(T)[i];
    if (wave == 0) enc_l0_moments_wave<0>(T, lane, acc);
    else if (wave == 1) enc_l0_moments_wave<1>(T, lane, acc);
    else if (wave == 2) enc_l0_moments_wave<2>(T, lane, acc);
    else enc_l0_moments_wave<3>(T, lane, acc);
  }
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    double v = acc[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0 && 4 * i + wave < ENC_L0_MOM_ROW) part[(size_t)blockIdx.x * ENC_L0_MOM_ROW + 4 * i + wave] = v;
  }
}

int enc_l0_moment_blocks(int64_t B) { return (int)(B < 512 ? B : 512); }

// One block: the moments summed over the blocks (fixed order, double) -> mom [ENC_L0_MOM_ROW] (kept for the backward call's
// combine kernel), then per channel the batch mean and variance of y = w . patch + b:
//   mean = b + sum_t w_t S_t / M,   var = sum_{t,t'} w_t w_t' (P_tt' / M - S_t S_t' / M^2)   (biased, as BatchNorm normalises)
// and the running statistics as bn_finalize_kernel updates them.
__global__ __launch_bounds__(1024) void enc_l0_moments_finalize_kernel(const double* __restrict__ part, int nblk, double M,
                                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                                       double* __restrict__ mom, float* __restrict__ mean,
                                                                       float* __restrict__ invstd, float* __restrict__ rm,
                                                                       float* __restrict__ rv, int64_t* __restrict__ nbt) {
  __shared__ double red[16][ENC_L0_MOM_ROW];
  __shared__ double em[ENC_L0_MOM_ROW];  // the moments over M: E[in_t], E[in_t in_t']
  const int tid = threadIdx.x, j = tid & 63, g = tid >> 6;  // column j, row group g of 16
  double sum = 0.0;
  for (int r0 = g; r0 < nblk; r0 += 32 * 16) {  // (enc_l0_moment_blocks: at most 512 rows = one round, 32 loads in flight)
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = r0 + 16 * u < nblk ? part[(size_t)(r0 + 16 * u) * ENC_L0_MOM_ROW + j] : 0.0;
#pragma unroll
    for (int u = 0; u < 32; ++u) sum += v[u];
  }
  red[g][j] = sum;
  __syncthreads();
  if (tid < ENC_L0_MOM_ROW) {
    double t = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += red[u][tid];
    if (tid >= ENC_L0_MOM) t = 0.0;
    mom[tid] = t;
    em[tid] = t / M;
  }
  __syncthreads();
  if (tid == 0 && nbt) *nbt += 1;
  if (tid >= 32) return;
  const int c = tid;
  double wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = (double)w[c * 9 + t];
  double lin = 0.0, quad = 0.0;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    lin += wt[t] * em[t];
#pragma unroll
    for (int u = t; u < 9; ++u) {
      const double cov = em[enc_l0_pair_index(t, u)] - em[t] * em[u];
      quad += (u == t ? 1.0 : 2.0) * wt[t] * wt[u] * cov;
    }
  }
  const double mu = lin + (double)bias[c];
  const double var = quad < 0.0 ? 0.0 : quad;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
  if (rm) {  // torch: running = (1 - momentum) * running + momentum * batch, unbiased variance
    const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
    rm[c] = (float)((1.0 - (double)BN_MOMENTUM) * (double)rm[c] + (double)BN_MOMENTUM * mu);
    rv[c] = (float)((1.0 - (double)BN_MOMENTUM) * (double)rv[c] + (double)BN_MOMENTUM * unbiased);
  }
}

int launch_enc_l0_moments(const float* images, int64_t B, const float* w, const float* bias, double* part, double* mom,
                          float* pimg, float* mean, float* invstd, float* rm, float* rv, int64_t* nbt, hipStream_t s) {
  const int nblk = enc_l0_moment_blocks(B);
  DVG_LAUNCH(K_ENC_CONV0_FWD, enc_l0_moments_kernel, dim3((unsigned)nblk), dim3(256), 0, s, images, B, part, pimg);
  DVG_LAUNCH(K_BN_FINALIZE, enc_l0_moments_finalize_kernel, dim3(1), dim3(1024), 0, s, (const double*)part, nblk,
             (double)B * 1024.0, w, bias, mom, mean, invstd, rm, rv, nbt);
  return DVG_OK;
}

// tot [ENC_L0_ROW]: the column sums of MODE 4's partials; mom: the patch moments of the forward call.  One block: every
// gradient of the stage.  T1[t] = S[t] (t = 9, the bias column of ones: M);
// T2[co][t] = sum zhat[co] in_t = invstd (sum_t' w[co][t'] P[t'][t] + (b - mu) S[t])  (t = 9: invstd (w . S + (b - mu) M))
__global__ __launch_bounds__(320) void enc_l0_combine_kernel(const float* __restrict__ tot, const double* __restrict__ mom,
                                                             const float* __restrict__ wgt, const float* __restrict__ bias,
                                                             const float* __restrict__ mean, const float* __restrict__ gamma,
                                                             const float* __restrict__ invstd, double M,
                                                             float* __restrict__ gw, float* __restrict__ gb,
                                                             float* __restrict__ g_bn_b, float* __restrict__ g_bn_g) {
  const int w = threadIdx.x;  // [0, 288): weight (co, t) = (w / 9, w % 9); [288, 320): bias of co = w - 288 (column 9)
  const int co = w < 288 ? w / 9 : w - 288, t = w < 288 ? w % 9 : 9;
  const float sum_dz = tot[288 + co], sum_dzzh = tot[320 + co];
  const double m1 = (double)sum_dz / M, m2 = (double)sum_dzzh / M;
  const double shift = (double)bias[co] - (double)mean[co];
  double t1, t2 = 0.0;
  if (t < 9) {
    t1 = mom[t];
    for (int u = 0; u < 9; ++u) t2 += (double)wgt[co * 9 + u] * mom[u <= t ? enc_l0_pair_index(u, t) : enc_l0_pair_index(t, u)];
    t2 += shift * mom[t];
  } else {
    t1 = M;
    for (int u = 0; u < 9; ++u) t2 += (double)wgt[co * 9 + u] * mom[u];
    t2 += shift * M;
  }
  t2 *= (double)invstd[co];
  const float v = (float)((double)gamma[co] * (double)invstd[co] * ((double)tot[w] - m1 * t1 - m2 * t2));
  if (w < 288) gw[w] = v;
  else { gb[co] = v; g_bn_b[co] = sum_dz; g_bn_g[co] = sum_dzzh; }
}

int launch_enc_l0_combine(const float* tot, const double* mom, const float* w, const float* bias, const float* mean,
                          const float* gamma, const float* invstd, int64_t B, float* gw, float* gb, float* g_bn_b,
                          float* g_bn_g, hipStream_t s) {
  DVG_LAUNCH(K_MISC, enc_l0_combine_kernel, dim3(1), dim3(320), 0, s, tot, mom, w, bias, mean, gamma, invstd, (double)B * 1024.0,
             gw, gb, g_bn_b, g_bn_g);
  return DVG_OK;
}

// dW[co][t] = sum_m dY[m][co] * in_t[m];  db[co] = sum_m dY[m][co];  part [stream_blocks(B)][320]
// A (32 x K) x (K x 10) GEMM with K = B*1024 pixels: one f32 MFMA per two pixels.  The A operand (dY rows, 32
// channels = 128 B) and the B operand (the 9 shifted input pixels + a column of ones for the bias, rest zero) go
// straight from global memory to the MFMA: no LDS, no staging.
typedef float f32x16s __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void enc_conv0_wgrad_kernel(const float* __restrict__ img, int64_t B,
                                                              const float* __restrict__ dY, float* __restrict__ part) {
  __shared__ float red[4 * 32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, c = lane & 31;
  const int64_t P = B * 1024;
  int64_t per = (P + gridDim.x * 4 - 1) / (gridDim.x * 4);
  per = (per + 1) & ~(int64_t)1;  // pixels per wave, even
  const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * per;
  const int64_t p1 = p0 + per < P ? p0 + per : P;
  const int dy = c < 9 ? c / 3 - 1 : 0, dx = c < 9 ? c % 3 - 1 : 0;
  f32x16s acc = {0};
  // 8 pixel pairs per iteration: all 16 loads are issued (from clamped, always-valid addresses) before the first
  // MFMA needs one -- the one-pair loop paid a full global-load latency per MFMA.
  constexpr int UN = 8;
  for (int64_t p = p0; p < p1; p += 2 * UN) {
    float av[UN], bv[UN];
    bool al[UN], bl[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t m = p + 2 * u + hh;
      const bool live = m < p1;
      const int64_t mc = live ? m : p0;
      av[u] = dY[mc * 32 + c];  // A[co = c][k = hh]
      const uint32_t q = (uint32_t)(mc & 1023);
      const int yy = (int)morton_y(q) + dy, xx = (int)morton_x(q) + dx;
      const bool inb = yy >= 0 && yy < 32 && xx >= 0 && xx < 32;
      bv[u] = img[(mc >> 10) * 1024 + (inb ? yy * 32 + xx : 0)];  // B[k = hh][col = c]
      al[u] = live;
      bl[u] = live && c < 9 && inb;
    }
    // masks are applied here, after every load has been issued (a select next to its load would make the in-order
    // issue wait for it before the next load goes out)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float a = al[u] ? av[u] : 0.f;
      const float b = (c == 9) ? (al[u] ? 1.0f : 0.0f) : (bl[u] ? bv[u] : 0.f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  // D[row = co][col = t]: lane holds column c, rows (r&3) + 8*(r>>2) + 4*hh
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + c] = acc[r];
  __syncthreads();
  for (int w = tid; w < 320; w += 256) {
    const int co = w < 288 ? w / 9 : w - 288, t = w < 288 ? w % 9 : 9;
    part[(size_t)blockIdx.x * 320 + w] =
        (red[(0 * 32 + co) * 33 + t] + red[(1 * 32 + co) * 33 + t]) + (red[(2 * 32 + co) * 33 + t] + red[(3 * 32 + co) * 33 + t]);
  }
}

int launch_enc_conv0_wgrad(const float* images, int64_t B, const float* dY, float* part, hipStream_t s) {
  DVG_LAUNCH(K_ENC_CONV0_WGRAD, enc_conv0_wgrad_kernel, dim3((unsigned)stream_blocks(8 * B)), dim3(256), 0, s, images, B, dY, part);  // (pixel ranges, not images)
  return DVG_OK;
}

// ------------------------------------------------------------------------------ encoder projection
__global__ __launch_bounds__(256) void enc_proj_fwd_kernel(const float* __restrict__ P, int64_t B, int n,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           float* __restrict__ logits) {
  const int64_t total = B * n;
  const float w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], bb = b[0];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % n);
    const float* p = P + (e / n) * 4 * n + c;
    // flatten order of the (2,2) map is h*2+w == Morton index of a 2x2 image
    logits[e] = fmaf(p[3 * (size_t)n], w3, fmaf(p[2 * (size_t)n], w2, fmaf(p[n], w1, p[0] * w0))) + bb;
  }
}

int launch_enc_proj_fwd(const float* P, int64_t B, int n, const float* w, const float* b, float* logits, hipStream_t s) {
  const int64_t g = ceil_div(B * n, 256);
  DVG_LAUNCH(K_ENC_PROJ_FWD, enc_proj_fwd_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, P, B, n, w, b,
             logits);
  return DVG_OK;
}

__global__ __launch_bounds__(256) void enc_proj_bwd_kernel(const float* __restrict__ P, int64_t B, int n,
                                                           const float* __restrict__ w, const float* __restrict__ dl,
                                                           float* __restrict__ dP, float* __restrict__ part) {
  // four latent units per thread (n % 32 == 0): five 16-byte loads and four 16-byte stores in flight per item instead of
  // five 4-byte loads -- the one-float form ran at 0.6 TB/s on the head of the encoder's backward chain (c3: 137 us for
  // 75 MB); the partial sums keep their two-stage fixed order (per thread, then per block)
  __shared__ float red[5 * 256];
  const int n4 = n >> 2;
  const int64_t total = B * n4;
  const float wv[4] = {w[0], w[1], w[2], w[3]};
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(e % n4);
    const int64_t b = e / n4;
    const float4 g = *reinterpret_cast<const float4*>(dl + b * n + 4 * c4);
    const float* p = P + b * 4 * n + 4 * c4;
    float* dp = dP + b * 4 * n + 4 * c4;
    float4 pv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pv[k] = *reinterpret_cast<const float4*>(p + (size_t)k * n);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[k] = fmaf(g.x, pv[k].x, acc[k]);
      acc[k] = fmaf(g.y, pv[k].y, acc[k]);
      acc[k] = fmaf(g.z, pv[k].z, acc[k]);
      acc[k] = fmaf(g.w, pv[k].w, acc[k]);
      *reinterpret_cast<float4*>(dp + (size_t)k * n) = make_float4(g.x * wv[k], g.y * wv[k], g.z * wv[k], g.w * wv[k]);
    }
    acc[4] += g.x; acc[4] += g.y; acc[4] += g.z; acc[4] += g.w;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) red[k * 256 + threadIdx.x] = acc[k];
  __syncthreads();
  if (threadIdx.x < 5) {
    float t = 0.f;
    for (int j = 0; j < 256; ++j) t += red[threadIdx.x * 256 + j];
    part[(size_t)blockIdx.x * 5 + threadIdx.x] = t;
  }
}

int launch_enc_proj_bwd(const float* P, int64_t B, int n, const float* w, const float* dlogits, float* dP, float* part,
                        hipStream_t s) {
  DVG_LAUNCH(K_ENC_PROJ_BWD, enc_proj_bwd_kernel, dim3(EW_BLOCKS), dim3(256), 0, s, P, B, n, w, dlogits, dP, part);
  return DVG_OK;
}

// ------------------------------------------------------------------------------ decoder conv3 (32 -> 1)
// X [N*64 (8x8 Morton)][32], upsampled x2 on the fly; Wt (32,1,3,3); Y [N*256 (16x16 Morton)]
// ConvTranspose (stride 1, pad 1):  out(y,x) = sum_{ci,kh,kw} in(y+1-kh, x+1-kw) Wt[ci][0][kh][kw]
// Folded form (conv.h: fold_src): an output pixel of parity class (pa, pb) reads only the 2x2 source pixels
// (i-1+pa+dr, j-1+pb+dc) with the 9 kernel taps pre-summed onto them.  One image per block iteration: its 8x8x32 source
// map is staged in LDS once (the old gather re-read every source row 36 times through L1); wave w computes the 64
// outputs of class w with its 4x32 folded weights in registers; the 256 outputs leave through LDS as one contiguous row.
template <bool ACT>  // ACT: X is the previous layer's pre-BatchNorm output, activated on the way into LDS (DecActIn)
__global__ __launch_bounds__(256) void dec_conv3_fwd_kernel(const float* __restrict__ X, DecActIn in, int64_t N,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ Y, float* __restrict__ stats_part) {
  constexpr int XP = 36;  // row pitch in floats: 16-byte aligned rows, conflict-free b128 reads across source rows
  __shared__ __align__(16) float xs[64 * XP];
  __shared__ float w9[288];    // w[ci*9 + kh*3 + kw]
  __shared__ __align__(16) float wfs[512];
  __shared__ float outs[256];
  __shared__ double red[2 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int cls = __builtin_amdgcn_readfirstlane(tid >> 6), pa = cls >> 1, pb = cls & 1;
  for (int i = tid; i < 288; i += 256) w9[i] = w[i];
  __syncthreads();
  // folded weights wfs[class][t][ci] = sum of the forward-GEMM taps (r, s) that land on source (dr, dc); forward tap
  // (r, s) of the ConvTranspose is checkpoint tap (kh, kw) = (2 - r, 2 - s).  Built once per block, two entries per
  // thread, then each wave keeps its class's 4 x 32 in registers.
  for (int e = tid; e < 512; e += 256) {
    const int ci = e & 31, t = (e >> 5) & 3, cc = e >> 7;
    const int dr = t >> 1, dc = t & 1, qa = cc >> 1, qb = cc & 1;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int sx = 0; sx < 3; ++sx)
        if (fold_src(qa, r) == dr && fold_src(qb, sx) == dc) acc += w9[ci * 9 + (8 - (r * 3 + sx))];
    wfs[e] = acc;
  }
  __syncthreads();
  // The class's 4 x 32 folded weights stay in LDS and are read as wave-uniform (broadcast) float4s: in registers they
  // made this a 168-VGPR kernel, 8 more than an MMD pair-kernel wave (352 of a SIMD's 512) leaves -- at c3 the kernel
  // then sat behind the pair kernel for 1.4 ms of the step's critical chain instead of running beside it.
  const float* wcls = wfs + cls * 128;
  const float b0 = bias[0];
  const int ys = (int)morton_y((uint32_t)lane), xq = (int)morton_x((uint32_t)lane);
  int src[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int yy = ys - 1 + pa + (t >> 1), xx = xq - 1 + pb + (t & 1);
    src[t] = (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) ? (int)morton((uint32_t)yy, (uint32_t)xx) : -1;
  }
  double s1 = 0.0, s2 = 0.0;  // (BatchNorm partials in double: see enc_conv0_fwd_kernel)
  const int cg4 = (tid & 7) * 4;  // the channel quad this thread stages (the same for both of its float4s)
  float4 amu = {0.f, 0.f, 0.f, 0.f}, ais = amu, agm = amu, abt = amu;
  if (ACT) {
    amu = *reinterpret_cast<const float4*>(in.mean + cg4); ais = *reinterpret_cast<const float4*>(in.invstd + cg4);
    agm = *reinterpret_cast<const float4*>(in.gamma + cg4); abt = *reinterpret_cast<const float4*>(in.beta + cg4);
  }
  // the next image's rows are requested before this image is computed (registers): one block iteration is three barriers
  // long and would otherwise start with a full global-load latency
  float4 nv[2], nmk = {1.f, 1.f, 1.f, 1.f};
  auto fetch = [&](int64_t img) {
    const int64_t im = img < N ? img : N - 1;
    const float4* g = reinterpret_cast<const float4*>((ACT ? in.y : X) + im * 64 * 32);
    nv[0] = g[tid]; nv[1] = g[tid + 256];
    if (ACT && in.mask) nmk = *reinterpret_cast<const float4*>(in.mask + im * 32 + cg4);
  };
  fetch(blockIdx.x);
  for (int64_t img = blockIdx.x; img < N; img += gridDim.x) {
    __syncthreads();  // previous image's xs / outs readers are done
    {  // 64 rows x 32 floats = 512 float4: two per thread, coalesced
      float4 mk = nmk;
      const float4 cv[2] = {nv[0], nv[1]};
      if (ACT && in.mask) {
        const float ks = 1.0f / DROPOUT_KEEP;
        mk.x *= ks; mk.y *= ks; mk.z *= ks; mk.w *= ks;
      }
      fetch(img + gridDim.x);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + 256 * k;
        float4 v = cv[k];
        if (ACT) {  // dec_bn_act_fwd's arithmetic, element for element
          float z;
          z = fmaf((v.x - amu.x) * ais.x, agm.x, abt.x); if (in.mask) z *= mk.x; v.x = z < 0.f ? z * LRELU_SLOPE : z;
          z = fmaf((v.y - amu.y) * ais.y, agm.y, abt.y); if (in.mask) z *= mk.y; v.y = z < 0.f ? z * LRELU_SLOPE : z;
          z = fmaf((v.z - amu.z) * ais.z, agm.z, abt.z); if (in.mask) z *= mk.z; v.z = z < 0.f ? z * LRELU_SLOPE : z;
          z = fmaf((v.w - amu.w) * ais.w, agm.w, abt.w); if (in.mask) z *= mk.w; v.w = z < 0.f ? z * LRELU_SLOPE : z;
        }
        *reinterpret_cast<float4*>(xs + (e >> 3) * XP + (e & 7) * 4) = v;
      }
    }
    __syncthreads();
    float acc = b0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (src[t] >= 0) {
        const float4* row = reinterpret_cast<const float4*>(xs + src[t] * XP);
        const float4* wr = reinterpret_cast<const float4*>(wcls + t * 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 a = row[k], wv = wr[k];
          acc = fmaf(a.x, wv.x, acc); acc = fmaf(a.y, wv.y, acc);
          acc = fmaf(a.z, wv.z, acc); acc = fmaf(a.w, wv.w, acc);
        }
      }
    }
    outs[4 * lane + cls] = acc;  // Morton: output pixel = 4 * source pixel + class
    s1 += (double)acc;
    s2 = fma((double)acc, (double)acc, s2);
    __syncthreads();
    Y[img * 256 + tid] = outs[tid];
  }
  // per-block BatchNorm partials (sum, sum of squares) over every image this block produced
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
  if (lane == 0) { red[cls] = s1; red[4 + cls] = s2; }
  __syncthreads();
  if (tid == 0) {
    stats_part[(size_t)blockIdx.x * 2] = (float)((red[0] + red[1]) + (red[2] + red[3]));
    stats_part[(size_t)blockIdx.x * 2 + 1] = (float)((red[4] + red[5]) + (red[6] + red[7]));
  }
}

int dec_conv3_blocks(int64_t N) { return (int)(N < special_block_cap() ? N : special_block_cap()); }

int launch_dec_conv3_fwd(const float* X, int64_t N, const float* w, const float* b, float* Y, float* stats_part,
                         hipStream_t s) {
  DVG_LAUNCH(K_DEC_CONV3_FWD, dec_conv3_fwd_kernel<false>, dim3((unsigned)dec_conv3_blocks(N)), dim3(256), 0, s, X, DecActIn{}, N,
             w, b, Y, stats_part);
  return DVG_OK;
}

int launch_dec_conv3_fwd_act(const DecActIn& in, int64_t N, const float* w, const float* b, float* Y, float* stats_part,
                             hipStream_t s) {
  DVG_LAUNCH(K_DEC_CONV3_FWD, dec_conv3_fwd_kernel<true>, dim3((unsigned)dec_conv3_blocks(N)), dim3(256), 0, s, nullptr, in, N,
             w, b, Y, stats_part);
  return DVG_OK;
}

// (the separate data-gradient and weight-gradient kernels of this layer were retired in round 3: dec_conv3_bwd_kernel below
// produces both from one pass over the images)

// Data gradient AND weight gradient in one pass over the images.  Both are products with the same small matrix
//   G[q][tap] = sum of dY over the (<= 4) output pixels whose tap reads source pixel q        (64 x 9 per image):
//   dX[q][ci] = sum_tap G[q][tap] Wt[ci][tap]          dWt[ci][tap] += sum_q G[q][tap] X[q][ci]
// so an image costs one read of its dY row (1 KB) and X tile (8 KB), one write of dX (8 KB) and 2 x 18 k FMAs -- the
// separate kernels gathered dY 36 times per source pixel (74 k FMAs) and read X / wrote dX in passes of their own
// (0.47 + 0.44 ms alone at c3 for 0.57 GB of traffic).  288 threads: (tap, ci) for the weight gradient; the first 256 are
// (source pixel, 8-channel group) for the data gradient.  part[blk][tap*32 + ci] as dec_conv3_wgrad_kernel.
__global__ __launch_bounds__(288) void dec_conv3_bwd_kernel(const float* __restrict__ X, int64_t N,
                                                            const float* __restrict__ dY, const float* __restrict__ w,
                                                            float* __restrict__ dX, float* __restrict__ part) {
  __shared__ float dys[256];
  __shared__ float G[64 * 9];
  __shared__ __align__(16) float xs[64 * 32];
  __shared__ float ws[9 * 32];  // [tap][ci]
  const int tid = threadIdx.x;
  for (int i = tid; i < 288; i += 288) ws[(i % 9) * 32 + i / 9] = w[i];
  // the (<= 4) dY elements behind this thread's two G entries: fixed for the whole kernel
  int gsrc[2][4];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = tid + 288 * k, q = e / 9, t = e % 9, kh = t / 3, kw = t % 3;
    const int ys = (int)morton_y((uint32_t)q), xq = (int)morton_x((uint32_t)q);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int y = 2 * ys + (u >> 1) - 1 + kh, x = 2 * xq + (u & 1) - 1 + kw;
      gsrc[k][u] = (y >= 0 && y < 16 && x >= 0 && x < 16) ? (int)morton((uint32_t)y, (uint32_t)x) : -1;
    }
  }
  const int tap = tid >> 5, ci = tid & 31;      // weight-gradient role
  const int q = tid >> 2, cg = (tid & 3) * 8;   // data-gradient role (tid < 256)
  float acc = 0.f;
  __syncthreads();
  for (int64_t img = blockIdx.x; img < N; img += gridDim.x) {
    __syncthreads();  // the previous image's readers of dys / G / xs are done
    if (tid < 256) dys[tid] = dY[img * 256 + tid];
    {
      const float4* g4 = reinterpret_cast<const float4*>(X + img * 64 * 32);
      for (int e = tid; e < 512; e += 288) reinterpret_cast<float4*>(xs)[e] = g4[e];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float g = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) g += gsrc[k][u] >= 0 ? dys[gsrc[k][u]] : 0.f;
      G[tid + 288 * k] = g;
    }
    __syncthreads();
    if (tid < 256) {
      float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float g = G[q * 9 + t];
        const float* wr = ws + t * 32 + cg;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaf(g, wr[k], o[k]);
      }
      float4* dst = reinterpret_cast<float4*>(dX + (img * 64 + q) * 32 + cg);
      dst[0] = make_float4(o[0], o[1], o[2], o[3]);
      dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
#pragma unroll 8
    for (int qq = 0; qq < 64; ++qq) acc = fmaf(G[qq * 9 + tap], xs[qq * 32 + ci], acc);
  }
  part[(size_t)blockIdx.x * 288 + tid] = acc;
}

int launch_dec_conv3_bwd(const float* X, int64_t N, const float* dY, const float* w, float* dX, float* part, hipStream_t s) {
  DVG_LAUNCH(K_DEC_CONV3_BWD, dec_conv3_bwd_kernel, dim3((unsigned)stream_blocks(N)), dim3(288), 0, s, X, N, dY, w, dX, part);
  return DVG_OK;
}


// The same pass with the layer in front of conv3 folded in (round 3).  The image's source map arrives as that layer's
// pre-BatchNorm output y (DecActIn): zhat and the activated value x are formed while it is staged; the gradient wrt x
// (what dec_conv3_bwd_kernel writes as dX) stays in registers and goes straight through the LeakyReLU / Dropout2d backward:
//   APPLY = false: (sum dz, sum dz zhat) partials of that layer + conv3's weight-gradient partials   (x, dX never stored)
//   APPLY = true:  that layer's dY = gamma invstd (dz - mean(dz) - zhat mean(dz zhat)) and its column-sum partials
// Together they replace dec_conv3_bwd + dec_bn_act_bwd reduce + apply of the 8x8x32 stage: 2.4 GB -> 0.9 GB at c3.
template <bool APPLY>
__global__ __launch_bounds__(256) void dec_conv3_bwd_fused_kernel(DecActIn in, int64_t N, const float* __restrict__ dY3,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ sum_dz,
                                                                  const float* __restrict__ sum_dzzh, float inv_m,
                                                                  float* __restrict__ dY2, float* __restrict__ part_w,
                                                                  float* __restrict__ part_bn) {
  __shared__ float dys[256];
  __shared__ float G[64 * 9];
  __shared__ __align__(16) float xs[APPLY ? 4 : 64 * 32];  // activated input, for the weight-gradient role
  __shared__ float ws[9 * 32];  // [tap][ci]
  __shared__ float red[8 * 32 * (APPLY ? 1 : 2)];
  const int tid = threadIdx.x;
  for (int i = tid; i < 288; i += 256) ws[(i % 9) * 32 + i / 9] = w[i];
  // (256 threads since round 5 -- the weight-gradient role no longer needs one thread per (tap, channel): four waves, three
  // blocks per CU at this kernel's register count where the 288-thread form fitted two)
  int gsrc[3][4];  // G[e], e = tid + 256 k < 576 = 64 source pixels x 9 taps
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = tid + 256 * k < 576 ? tid + 256 * k : 0, q = e / 9, t = e % 9, kh = t / 3, kw = t % 3;
    const int ys = (int)morton_y((uint32_t)q), xq = (int)morton_x((uint32_t)q);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int y = 2 * ys + (u >> 1) - 1 + kh, x = 2 * xq + (u & 1) - 1 + kw;
      gsrc[k][u] = (y >= 0 && y < 16 && x >= 0 && x < 16) ? (int)morton((uint32_t)y, (uint32_t)x) : -1;
    }
  }
  const int q = tid >> 2, cg = (tid & 3) * 8;   // data-gradient role (tid < 256): source pixel q, channels cg .. cg + 7
  // weight-gradient role (APPLY = false), round 5: dW3[tap][ci] += sum_q G[q][tap] x[q][ci] on v_mfma_f32_16x16x4_f32 --
  // rows = taps (9 of 16), columns = 16 channels, k = source pixels: wave w < 4 owns pixels 16 w .. 16 w + 15 (four
  // k-steps x two channel halves = 8 MFMAs and 12 LDS reads per image where every one of the 288 threads ran 64 FMAs
  // off 128 LDS reads: the kernel was bound by those reads and by its vector instructions, 200 us at c3 for 295 MB)
  const int lane = tid & 63, wv = tid >> 6, mt = lane & 15, mq = lane >> 4;
  f32x4c wacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  constexpr bool dg = true;  // (every thread has the data-gradient role)
  // per-channel constants in LDS (read as broadcast float4s where they are used: in registers they made this a 170-VGPR
  // kernel, two waves per SIMD, bound by the latency of its own loads)
  __shared__ __align__(16) float cst[7][32];  // mean, invstd, gamma, beta, gamma invstd, mean(dz), mean(dz zhat)
  if (tid < 32) {
    cst[0][tid] = in.mean[tid]; cst[1][tid] = in.invstd[tid]; cst[2][tid] = in.gamma[tid]; cst[3][tid] = in.beta[tid];
    cst[4][tid] = in.gamma[tid] * in.invstd[tid];
    cst[5][tid] = APPLY ? sum_dz[tid] * inv_m : 0.f;
    cst[6][tid] = APPLY ? sum_dzzh[tid] * inv_m : 0.f;
  }
  float a1[8], a2[8];  // APPLY: a1 = column sums of dY; else (sum dz, sum dz zhat)
#pragma unroll
  for (int k = 0; k < 8; ++k) { a1[k] = 0.f; a2[k] = 0.f; }
  const float ks = 1.0f / DROPOUT_KEEP;
  __syncthreads();
  // (the next image's values are requested before this image is computed, as in the forward kernel)
  float4 ny0 = {0.f, 0.f, 0.f, 0.f}, ny1 = ny0, nk0 = {1.f, 1.f, 1.f, 1.f}, nk1 = nk0;
  float ndy = 0.f;
  auto fetch = [&](int64_t img) {
    if (!dg) return;
    const int64_t im = img < N ? img : N - 1;
    const float4* yp = reinterpret_cast<const float4*>(in.y + (im * 64 + q) * 32 + cg);
    ny0 = yp[0]; ny1 = yp[1];
    if (in.mask) {
      const float4* mp = reinterpret_cast<const float4*>(in.mask + im * 32 + cg);
      nk0 = mp[0]; nk1 = mp[1];
    }
    ndy = dY3[im * 256 + tid];
  };
  fetch(blockIdx.x);
  for (int64_t img = blockIdx.x; img < N; img += gridDim.x) {
    __syncthreads();  // the previous image's readers of dys / G / xs are done
    if (dg) dys[tid] = ndy;
    // this thread's 8 values of the layer's pre-BatchNorm output: zhat, the slope of the LeakyReLU at the activated value
    // and the keep-mask stay in registers; the activated values go to LDS for the weight-gradient role
    float zh[8], sl[8];
    const float4 y0 = ny0, y1 = ny1, k0 = nk0, k1 = nk1;
    fetch(img + gridDim.x);
    if (dg) {
      const float yv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
      const float kv[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
      float xv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        zh[k] = (yv[k] - cst[0][cg + k]) * cst[1][cg + k];
        float z = fmaf(zh[k], cst[2][cg + k], cst[3][cg + k]);
        const float mk = in.mask ? kv[k] * ks : 1.0f;
        if (in.mask) z *= mk;
        xv[k] = z < 0.f ? z * LRELU_SLOPE : z;
        sl[k] = ((xv[k] > 0.f) ? 1.0f : LRELU_SLOPE) * mk;  // dec_dz of elementwise.hip: LeakyReLU'(x) * keep-mask
      }
      if (!APPLY) {
        float4* xd = reinterpret_cast<float4*>(xs + q * 32 + cg);
        xd[0] = make_float4(xv[0], xv[1], xv[2], xv[3]);
        xd[1] = make_float4(xv[4], xv[5], xv[6], xv[7]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float g = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) g += gsrc[k][u] >= 0 ? dys[gsrc[k][u]] : 0.f;
      if (tid + 256 * k < 576) G[tid + 256 * k] = g;
    }
    __syncthreads();
    if (dg) {
      float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float g = G[q * 9 + t];
        const float* wr = ws + t * 32 + cg;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaf(g, wr[k], o[k]);
      }
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dz = o[k] * sl[k];
        if (APPLY) {
          v[k] = cst[4][cg + k] * (dz - cst[5][cg + k] - zh[k] * cst[6][cg + k]);
          a1[k] += v[k];
        } else {
          a1[k] += dz;
          a2[k] = fmaf(dz, zh[k], a2[k]);
        }
      }
      if (APPLY) {
        float4* dst = reinterpret_cast<float4*>(dY2 + (img * 64 + q) * 32 + cg);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    if constexpr (!APPLY) {
      float av[4], b0[4], b1[4];  // every operand read goes out before the first MFMA waits for one
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int qq = 16 * wv + 4 * s4 + mq;
        av[s4] = G[qq * 9 + (mt < 9 ? mt : 0)];
        b0[s4] = xs[qq * 32 + mt];
        b1[s4] = xs[qq * 32 + 16 + mt];
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float a_ = mt < 9 ? av[s4] : 0.f;
        wacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b0[s4], wacc[0], 0, 0, 0);
        wacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b1[s4], wacc[1], 0, 0, 0);
      }
    }
  }
  if constexpr (!APPLY) {
    // D[row = tap][col = channel]: a lane holds rows 4 mq + r of column mt; the four waves' partial sums meet in LDS
    __syncthreads();
    {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * mq + r;
        if (row < 9) {
          xs[(wv * 9 + row) * 32 + mt] = wacc[0][r];
          xs[(wv * 9 + row) * 32 + 16 + mt] = wacc[1][r];
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < 288; e += 256)  // e = tap * 32 + channel
      part_w[(size_t)blockIdx.x * 288 + e] = (xs[0 * 288 + e] + xs[1 * 288 + e]) + (xs[2 * 288 + e] + xs[3 * 288 + e]);
  }
  // per-block column sums over the 64 source-pixel threads of each 8-channel group: waves 0-3 hold 16 pixels x 4 groups
  // each; lanes with the same (tid & 3) share a group
  if (dg) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int off = 4; off < 64; off <<= 1) {
        a1[k] += __shfl_xor(a1[k], off, 64);
        if (!APPLY) a2[k] += __shfl_xor(a2[k], off, 64);
      }
    }
    if ((tid & 63) < 4) {
      const int wv = tid >> 6;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[(wv * 32 + cg + k) * (APPLY ? 1 : 2)] = a1[k];
        if (!APPLY) red[(wv * 32 + cg + k) * 2 + 1] = a2[k];
      }
    }
  }
  __syncthreads();
  if (APPLY) {
    if (tid < 32) part_bn[(size_t)blockIdx.x * 32 + tid] = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
  } else if (tid < 64) {
    const int k = tid >> 5, cc = tid & 31;  // [2][32]: sums of dz, then sums of dz zhat
    part_bn[(size_t)blockIdx.x * 64 + tid] =
        (red[(0 * 32 + cc) * 2 + k] + red[(1 * 32 + cc) * 2 + k]) + (red[(2 * 32 + cc) * 2 + k] + red[(3 * 32 + cc) * 2 + k]);
  }
}

int dec_tail_blocks(int64_t N) { return stream_blocks(N); }

int launch_dec_conv3_bwd_reduce(const DecActIn& in, int64_t N, const float* dY3, const float* w, float* part_w, float* part_bn,
                                hipStream_t s) {
  DVG_LAUNCH(K_DEC_CONV3_BWD, dec_conv3_bwd_fused_kernel<false>, dim3((unsigned)dec_tail_blocks(N)), dim3(256), 0, s, in, N, dY3,
             w, nullptr, nullptr, 0.f, nullptr, part_w, part_bn);
  return DVG_OK;
}

int launch_dec_conv3_bwd_apply(const DecActIn& in, int64_t N, const float* dY3, const float* w, const float* sum_dz,
                               const float* sum_dzzh, float* dY2, float* part_db, hipStream_t s) {
  DVG_LAUNCH(K_DEC_BN_ACT_BWD_APPLY, dec_conv3_bwd_fused_kernel<true>, dim3((unsigned)dec_tail_blocks(N)), dim3(256), 0, s, in, N,
             dY3, w, sum_dz, sum_dzzh, (float)(1.0 / ((double)N * 64.0)), dY2, nullptr, part_db);
  return DVG_OK;
}

// ------------------------------------------------------------------------------ decoder final conv (1 -> 1)
// X [N*256 (16x16 Morton)] upsampled to 32x32; out (N,32,32) row-major
// One image per block iteration, its 16x16 source map in LDS with a zero halo (row-major 18x18): a thread produces 4
// consecutive output pixels of a row from 3 x 4 source values (the nearest upsample makes neighbouring outputs share
// them) and stores one float4 -- instead of 36 bounds-checked global gathers.
// ACT (round 5): the source map arrives as the 1-channel stage's pre-BatchNorm output (DecActIn, C = 1) and is activated
// while it is staged -- dec_bn_act_fwd's arithmetic, element for element -- so that stage's activated map is never written.
__device__ __forceinline__ float dec_act1(float y, float mu, float is, float gm, float bt, const float* mask, int64_t img) {
  float z = fmaf((y - mu) * is, gm, bt);
  if (mask) z *= mask[img] * (1.0f / DROPOUT_KEEP);
  return z < 0.f ? z * LRELU_SLOPE : z;
}

template <bool ACT>
__global__ __launch_bounds__(256) void dec_final_fwd_kernel(const float* __restrict__ X, DecActIn in, int64_t N,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ out) {
  __shared__ float xs[18 * 18];
  const float mu = ACT ? in.mean[0] : 0.f, is = ACT ? in.invstd[0] : 0.f, gm = ACT ? in.gamma[0] : 0.f, bt = ACT ? in.beta[0] : 0.f;
  const int tid = threadIdx.x;
  float wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = w[t];
  const float bb = bias[0];
  for (int e = tid; e < 18 * 18; e += 256) xs[e] = 0.f;  // the halo stays zero for every image
  const int sy_ = (int)morton_y((uint32_t)tid), sx_ = (int)morton_x((uint32_t)tid);  // staging: source pixel `tid`
  const int y = tid >> 3, x0 = (tid & 7) * 4;
  for (int64_t img = blockIdx.x; img < N; img += gridDim.x) {
    __syncthreads();
    xs[(sy_ + 1) * 18 + sx_ + 1] = ACT ? dec_act1(in.y[img * 256 + tid], mu, is, gm, bt, in.mask, img) : X[img * 256 + tid];
    __syncthreads();
    float acc[4] = {bb, bb, bb, bb};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const float* row = xs + (((y + 1 - kh) >> 1) + 1) * 18 + (x0 >> 1);  // source columns x0/2-1 .. x0/2+2 (+1 halo)
      const float v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3];
      // output x0+j, tap kw reads upsampled column x0+j+1-kw, i.e. source column (x0+j+1-kw) >> 1
      const float src[6] = {v0, v1, v1, v2, v2, v3};  // upsampled columns x0-1 .. x0+4
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[j] = fmaf(src[j + 2 - kw], wv[kh * 3 + kw], acc[j]);
    }
    *reinterpret_cast<float4*>(out + img * 1024 + y * 32 + x0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

int launch_dec_final_fwd(const float* X, int64_t N, const float* w, const float* b, float* out, hipStream_t s) {
  DVG_LAUNCH(K_DEC_FINAL_FWD, dec_final_fwd_kernel<false>, dim3((unsigned)(N > 2048 ? 2048 : N)), dim3(256), 0, s, X, DecActIn{},
             N, w, b, out);
  return DVG_OK;
}

int launch_dec_final_fwd_act(const DecActIn& in, int64_t N, const float* w, const float* b, float* out, hipStream_t s) {
  DVG_LAUNCH(K_DEC_FINAL_FWD, dec_final_fwd_kernel<true>, dim3((unsigned)(N > 2048 ? 2048 : N)), dim3(256), 0, s, nullptr, in, N,
             w, b, out);
  return DVG_OK;
}

// dX[q] = sum over the quad's 4 pixels and taps of dOut(y+kh-1, x+kw-1) w[kh][kw]: a 4x4 stencil on dOut around the
// quad with the taps pre-summed per offset (rows {w0}, {w0+w1}, {w1+w2}, {w2}, same for columns).  The image's dOut sits
// in LDS with a zero halo (34x34); thread = source pixel.
// MODE 1 (round 3): the thread that forms dX[img][q] also takes it through the 1-channel stage's Dropout / LeakyReLU
// backward and adds its (dz, dz zhat) to the block's partials -- dec_bn_act_bwd_reduce's pass over Y, X and dX is gone
// (the slope from the sign of the recomputed z, as in elementwise.hip).  part [blocks][2].
// MODE 2 (round 5): the same pass again once the two sums are known: dX is formed a second time (the image's dOut is
// 4 KB in LDS; the stencil is 16 FMAs) and leaves as the stage's dY = gamma invstd (dz - mean(dz) - zhat mean(dz zhat)),
// with the block's sum of dY in part [blocks][1] -- dX itself is written by neither pass (MODE 1 stopped storing it), and
// dec_bn_act_bwd_apply's scalar pass over Y, X and dX (68 us at c3) is gone.  MODE 0: the plain data gradient.
template <int MODE>
__global__ __launch_bounds__(256) void dec_final_dgrad_kernel(const float* __restrict__ dOut, int64_t N,
                                                              const float* __restrict__ w, float* __restrict__ dX,
                                                              DecActIn in, const float* __restrict__ sum_dz,
                                                              const float* __restrict__ sum_dzzh, float inv_m,
                                                              float* __restrict__ part) {
  constexpr bool BN = MODE != 0;
  __shared__ float gs[34 * 34];
  __shared__ float red[2 * 4];
  float r1 = 0.f, r2 = 0.f;
  const float mu = BN ? in.mean[0] : 0.f, is = BN ? in.invstd[0] : 0.f, gm = BN ? in.gamma[0] : 0.f, bt = BN ? in.beta[0] : 0.f;
  const float m1 = MODE == 2 ? sum_dz[0] * inv_m : 0.f, m2 = MODE == 2 ? sum_dzzh[0] * inv_m : 0.f, gi = gm * is;
  const int tid = threadIdx.x;
  float wf[4][4];  // wf[u+1][v+1], u, v in -1..2: offset of the dOut pixel from the quad's top-left pixel
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float acc = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int sy = (u - 1) - (kh - 1), sx = (v - 1) - (kw - 1);  // which pixel of the quad this tap belongs to
          if (sy >= 0 && sy < 2 && sx >= 0 && sx < 2) acc += w[kh * 3 + kw];
        }
      wf[u][v] = acc;
    }
  for (int e = tid; e < 34 * 34; e += 256) gs[e] = 0.f;
  const int ys = (int)morton_y((uint32_t)tid), xq = (int)morton_x((uint32_t)tid);
  for (int64_t img = blockIdx.x; img < N; img += gridDim.x) {
    __syncthreads();
    {  // 1024 floats: one float4 per thread, row-major rows of 32
      const float4 v = reinterpret_cast<const float4*>(dOut + img * 1024)[tid];
      float* d = gs + ((tid >> 3) + 1) * 34 + (tid & 7) * 4 + 1;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* row = gs + (2 * ys + u) * 34 + 2 * xq;  // (2ys + u - 1) + 1 halo, (2xq - 1) + 1 halo
#pragma unroll
      for (int v = 0; v < 4; ++v) acc = fmaf(row[v], wf[u][v], acc);
    }
    if (MODE == 0) dX[img * 256 + tid] = acc;
    if (BN) {
      const float zh = (in.y[img * 256 + tid] - mu) * is;
      const float mk = in.mask ? in.mask[img] * (1.0f / DROPOUT_KEEP) : 1.0f;
      const float dz = acc * ((fmaf(zh, gm, bt) > 0.f) ? 1.0f : LRELU_SLOPE) * mk;
      if (MODE == 1) {
        r1 += dz;
        r2 = fmaf(dz, zh, r2);
      } else {
        const float v = gi * (dz - m1 - zh * m2);
        dX[img * 256 + tid] = v;  // (MODE 2: `dX` is the stage's dY)
        r1 += v;
      }
    }
  }
  if (BN) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { r1 += __shfl_xor(r1, off, 64); r2 += __shfl_xor(r2, off, 64); }
    if ((tid & 63) == 0) { red[tid >> 6] = r1; red[4 + (tid >> 6)] = r2; }
    __syncthreads();
    if (MODE == 1 && tid < 2) part[(size_t)blockIdx.x * 2 + tid] = (red[4 * tid] + red[4 * tid + 1]) + (red[4 * tid + 2] + red[4 * tid + 3]);
    if (MODE == 2 && tid == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

int dec_final_dgrad_blocks(int64_t N) { return (int)(N > 2048 ? 2048 : N); }

int launch_dec_final_dgrad(const float* dOut, int64_t N, const float* w, float* dX, hipStream_t s) {
  DVG_LAUNCH(K_DEC_FINAL_BWD, dec_final_dgrad_kernel<0>, dim3((unsigned)dec_final_dgrad_blocks(N)), dim3(256), 0, s, dOut, N, w,
             dX, DecActIn{}, nullptr, nullptr, 0.f, nullptr);
  return DVG_OK;
}

int launch_dec_final_dgrad_bn(const float* dOut, int64_t N, const float* w, const DecActIn& in, float* part, hipStream_t s) {
  DVG_LAUNCH(K_DEC_FINAL_BWD, dec_final_dgrad_kernel<1>, dim3((unsigned)dec_final_dgrad_blocks(N)), dim3(256), 0, s, dOut, N, w,
             nullptr, in, nullptr, nullptr, 0.f, part);
  return DVG_OK;
}

int launch_dec_final_dgrad_apply(const float* dOut, int64_t N, const float* w, const DecActIn& in, const float* sum_dz,
                                 const float* sum_dzzh, float* dY, float* part_db, hipStream_t s) {
  DVG_LAUNCH(K_DEC_FINAL_BWD, dec_final_dgrad_kernel<2>, dim3((unsigned)dec_final_dgrad_blocks(N)), dim3(256), 0, s, dOut,
             N, w, dY, in, sum_dz, sum_dzzh, (float)(1.0 / ((double)N * 256.0)), part_db);
  return DVG_OK;
}

// ---- the training step's tail in one pass per image (round 5): final ConvTranspose forward -> MSE (value + gradient) ->
// the final layer's data gradient -> the 1-channel stage's LeakyReLU / Dropout / BatchNorm backward.  The reconstruction
// (N x 1024 floats: 134 MB at c3) and its gradient never reach memory: the separate kernels wrote and read them five
// times between them (dec_final_fwd, mse_partial, dec_final_dgrad<1>, <2>, dec_final_wgrad: 1.1 GB -> 0.23 GB, 207 us of
// the critical chain).  Every value is formed by the arithmetic of the kernel it replaces, in its order: the gradients
// are those of the separate path bit for bit (the loss's double partial sums are grouped by image instead of by
// stride: equal to rounding).
//   MODE 1: (sum dz, sum dz zhat) partials part [blocks][2] + the block's sum of squared differences mse_part [blocks]
//   MODE 2: the stage's dY [N*256] once the two sums are known (the pass again) + the block sums of dY part [blocks][1]
// images [N / R][1024] (replica r of image b is row b R + r of the decoder's batch); gscale = 2 grad_scale / numel.
template <int MODE>
__global__ __launch_bounds__(256) void dec_tail_mse_kernel(DecActIn in, int64_t N, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ images,
                                                           int R, float gscale, const float* __restrict__ sum_dz,
                                                           const float* __restrict__ sum_dzzh, float inv_m,
                                                           float* __restrict__ dY, float* __restrict__ part,
                                                           double* __restrict__ mse_part) {
  __shared__ float xs[18 * 18];
  __shared__ float gs[34 * 34];
  __shared__ float red[2 * 4];
  __shared__ double redd[4];
  const int tid = threadIdx.x;
  const float mu = in.mean[0], is = in.invstd[0], gm = in.gamma[0], bt = in.beta[0];
  const float m1 = MODE == 2 ? sum_dz[0] * inv_m : 0.f, m2 = MODE == 2 ? sum_dzzh[0] * inv_m : 0.f, gi = gm * is;
  float wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = w[t];
  const float bb = bias[0];
  float wf[4][4];  // the data gradient's pre-summed taps (dec_final_dgrad_kernel)
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float acc = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int sy = (u - 1) - (kh - 1), sx = (v - 1) - (kw - 1);
          if (sy >= 0 && sy < 2 && sx >= 0 && sx < 2) acc += w[kh * 3 + kw];
        }
      wf[u][v] = acc;
    }
  for (int e = tid; e < 18 * 18; e += 256) xs[e] = 0.f;
  for (int e = tid; e < 34 * 34; e += 256) gs[e] = 0.f;
  const int ys = (int)morton_y((uint32_t)tid), xq = (int)morton_x((uint32_t)tid);  // source pixel `tid` (Morton)
  const int y = tid >> 3, x0 = (tid & 7) * 4;                                        // output pixels (y, x0 .. x0 + 3)
  float r1 = 0.f, r2 = 0.f;
  double msum = 0.0;
  // (the next image's values are requested before this image is computed: an iteration is three barriers long and would
  // otherwise open with a full global-load latency)
  float nyv = 0.f, nmk = 1.0f;
  float4 ntv = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int64_t img) {
    const int64_t im = img < N ? img : N - 1;
    nyv = in.y[im * 256 + tid];
    ntv = *reinterpret_cast<const float4*>(images + (im / R) * 1024 + y * 32 + x0);
    if (in.mask) nmk = in.mask[im];
  };
  fetch(blockIdx.x);
  for (int64_t img = blockIdx.x; img < N; img += gridDim.x) {
    __syncthreads();  // the previous image's readers of xs / gs are done
    const float yv = nyv, mkv = nmk;
    const float4 tv = ntv;
    fetch(img + gridDim.x);
    {  // dec_act1 with the mask value already in hand
      float z = fmaf((yv - mu) * is, gm, bt);
      if (in.mask) z *= mkv * (1.0f / DROPOUT_KEEP);
      xs[(ys + 1) * 18 + xq + 1] = z < 0.f ? z * LRELU_SLOPE : z;
    }
    __syncthreads();
    float acc[4] = {bb, bb, bb, bb};  // dec_final_fwd_kernel
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const float* row = xs + (((y + 1 - kh) >> 1) + 1) * 18 + (x0 >> 1);
      const float v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3];
      const float src[6] = {v0, v1, v1, v2, v2, v3};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[j] = fmaf(src[j + 2 - kw], wv[kh * 3 + kw], acc[j]);
    }
    const float d0 = acc[0] - tv.x, d1 = acc[1] - tv.y, d2 = acc[2] - tv.z, d3 = acc[3] - tv.w;  // mse_partial_kernel
    if (MODE == 1) msum += (double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2) + (double)(d3 * d3);
    float* gd = gs + (y + 1) * 34 + x0 + 1;
    gd[0] = gscale * d0; gd[1] = gscale * d1; gd[2] = gscale * d2; gd[3] = gscale * d3;
    __syncthreads();
    float dx = 0.f;  // dec_final_dgrad_kernel
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* row = gs + (2 * ys + u) * 34 + 2 * xq;
#pragma unroll
      for (int v = 0; v < 4; ++v) dx = fmaf(row[v], wf[u][v], dx);
    }
    const float zh = (yv - mu) * is;
    const float mk = in.mask ? mkv * (1.0f / DROPOUT_KEEP) : 1.0f;
    const float dz = dx * ((fmaf(zh, gm, bt) > 0.f) ? 1.0f : LRELU_SLOPE) * mk;
    if (MODE == 1) {
      r1 += dz;
      r2 = fmaf(dz, zh, r2);
    } else {
      const float v = gi * (dz - m1 - zh * m2);
      dY[img * 256 + tid] = v;
      r1 += v;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { r1 += __shfl_xor(r1, off, 64); r2 += __shfl_xor(r2, off, 64); msum += __shfl_xor(msum, off, 64); }
  if ((tid & 63) == 0) { red[tid >> 6] = r1; red[4 + (tid >> 6)] = r2; redd[tid >> 6] = msum; }
  __syncthreads();
  if (MODE == 1 && tid < 2) part[(size_t)blockIdx.x * 2 + tid] = (red[4 * tid] + red[4 * tid + 1]) + (red[4 * tid + 2] + red[4 * tid + 3]);
  if (MODE == 2 && tid == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  if (MODE == 1 && tid == 0) mse_part[blockIdx.x] = (redd[0] + redd[1]) + (redd[2] + redd[3]);
}

int launch_dec_tail_mse_sums(const DecActIn& in, int64_t N, const float* w, const float* bias, const float* images, int R,
                             float gscale, float* part, double* mse_part, hipStream_t s) {
  DVG_LAUNCH(K_DEC_FINAL_FWD, dec_tail_mse_kernel<1>, dim3((unsigned)dec_final_dgrad_blocks(N)), dim3(256), 0, s, in, N, w, bias,
             images, R, gscale, nullptr, nullptr, 0.f, nullptr, part, mse_part);
  return DVG_OK;
}

int launch_dec_tail_mse_apply(const DecActIn& in, int64_t N, const float* w, const float* bias, const float* images, int R,
                              float gscale, const float* sum_dz, const float* sum_dzzh, float* dY, float* part_db,
                              hipStream_t s) {
  DVG_LAUNCH(K_DEC_FINAL_BWD, dec_tail_mse_kernel<2>, dim3((unsigned)dec_final_dgrad_blocks(N)), dim3(256), 0, s, in, N, w, bias,
             images, R, gscale, sum_dz, sum_dzzh, (float)(1.0 / ((double)N * 256.0)), dY, part_db, nullptr);
  return DVG_OK;
}

// dw[kh][kw] = sum dOut(y,x) Xup(y+1-kh, x+1-kw); db = sum dOut;  part [EW_BLOCKS][10]
// One image per block pass: the 16x16 source image goes to LDS (row-major, zero border, double-buffered: one barrier
// per image); thread (i, j) owns source pixel (i, j), i.e. the 2x2 output quad that upsamples it, whose 4 x 9 taps
// all fall in that pixel's 3x3 neighbourhood: 9 LDS reads + 36 FMAs per thread and image, the next image's global
// loads in flight meanwhile.  (The per-output-pixel form -- 9 guarded, Morton-addressed global gathers per element --
// took 1.1 ms at c3, 17x its HBM time.)
// MSE (with ACT): dOut does not exist (dec_tail_mse_kernel): the thread forms its quad of it again -- the final layer's
// forward over the 3 x 3 neighbourhood it holds anyway, in dec_final_fwd_kernel's order, minus the image, times gscale.
template <bool ACT, bool MSE>
__global__ __launch_bounds__(256) void dec_final_wgrad_kernel(const float* __restrict__ X, DecActIn in, int64_t N,
                                                              const float* __restrict__ dOut, float* __restrict__ part,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ images, int R, float gscale) {
  __shared__ float Xs[2][18 * 18];
  float wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = MSE ? w[t] : 0.f;
  const float bb = MSE ? bias[0] : 0.f;
  const float mu = ACT ? in.mean[0] : 0.f, is = ACT ? in.invstd[0] : 0.f, gm = ACT ? in.gamma[0] : 0.f, bt = ACT ? in.beta[0] : 0.f;
  __shared__ float red[10 * 256];
  const int tid = threadIdx.x;
  for (int e = tid; e < 2 * 18 * 18; e += 256) (&Xs[0][0])[e] = 0.f;  // borders stay zero
  __syncthreads();
  const int spos = ((int)morton_y((uint32_t)tid) + 1) * 18 + (int)morton_x((uint32_t)tid) + 1;  // where X[img][tid] lives
  const int i = tid >> 4, j = tid & 15;
  float acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.f;
  int64_t img = blockIdx.x;
  float xv = 0.f;
  float2 g0 = {0.f, 0.f}, g1 = {0.f, 0.f};
  const float* gsrc = MSE ? images : dOut;  // (MSE: g0 / g1 carry the IMAGE's quad until the neighbourhood is read)
  if (img < N) {
    xv = ACT ? dec_act1(in.y[img * 256 + tid], mu, is, gm, bt, in.mask, img) : X[img * 256 + tid];
    const int64_t gi_ = MSE ? img / R : img;
    g0 = *reinterpret_cast<const float2*>(gsrc + gi_ * 1024 + (2 * i) * 32 + 2 * j);
    g1 = *reinterpret_cast<const float2*>(gsrc + gi_ * 1024 + (2 * i + 1) * 32 + 2 * j);
  }
  for (int buf = 0; img < N; img += gridDim.x, buf ^= 1) {
    Xs[buf][spos] = xv;
    float g[2][2] = {{g0.x, g0.y}, {g1.x, g1.y}};
    const int64_t nimg = img + gridDim.x;
    if (nimg < N) {
      xv = ACT ? dec_act1(in.y[nimg * 256 + tid], mu, is, gm, bt, in.mask, nimg) : X[nimg * 256 + tid];
      const int64_t gi_ = MSE ? nimg / R : nimg;
      g0 = *reinterpret_cast<const float2*>(gsrc + gi_ * 1024 + (2 * i) * 32 + 2 * j);
      g1 = *reinterpret_cast<const float2*>(gsrc + gi_ * 1024 + (2 * i + 1) * 32 + 2 * j);
    }
    __syncthreads();
    float sn[3][3];  // source pixels (i-1 .. i+1, j-1 .. j+1)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) sn[r][c] = Xs[buf][(i + r) * 18 + j + c];
    // output (2i+a, 2j+b), tap (kh, kw) reads upsampled (2i+a+1-kh, 2j+b+1-kw) = source ((2i+a+1-kh)>>1, ...):
    // neighbourhood row 1,1,0 for a = 0 and 2,1,1 for a = 1 (kh = 0,1,2); same along the columns
    if (MSE) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float rec = bb;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const int r = a == 0 ? (kh == 2 ? 0 : 1) : (kh == 0 ? 2 : 1);
              const int c = b == 0 ? (kw == 2 ? 0 : 1) : (kw == 0 ? 2 : 1);
              rec = fmaf(sn[r][c], wv[kh * 3 + kw], rec);
            }
          g[a][b] = gscale * (rec - g[a][b]);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int r = a == 0 ? (kh == 2 ? 0 : 1) : (kh == 0 ? 2 : 1);
            const int c = b == 0 ? (kw == 2 ? 0 : 1) : (kw == 0 ? 2 : 1);
            acc[kh * 3 + kw] = fmaf(g[a][b], sn[r][c], acc[kh * 3 + kw]);
          }
        acc[9] += g[a][b];
      }
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) red[k * 256 + tid] = acc[k];
  __syncthreads();
  if (tid < 10) {
    float t = 0.f;
    for (int q = 0; q < 256; ++q) t += red[tid * 256 + q];
    part[(size_t)blockIdx.x * 10 + tid] = t;
  }
}

int launch_dec_final_wgrad(const float* X, int64_t N, const float* dOut, float* part, hipStream_t s) {
  DVG_LAUNCH(K_DEC_FINAL_BWD, (dec_final_wgrad_kernel<false, false>), dim3(EW_BLOCKS), dim3(256), 0, s, X, DecActIn{}, N, dOut, part,
             nullptr, nullptr, nullptr, 1, 0.f);
  return DVG_OK;
}

int launch_dec_final_wgrad_act(const DecActIn& in, int64_t N, const float* dOut, float* part, hipStream_t s) {
  DVG_LAUNCH(K_DEC_FINAL_BWD, (dec_final_wgrad_kernel<true, false>), dim3(EW_BLOCKS), dim3(256), 0, s, nullptr, in, N, dOut, part,
             nullptr, nullptr, nullptr, 1, 0.f);
  return DVG_OK;
}

int launch_dec_final_wgrad_mse(const DecActIn& in, int64_t N, const float* w, const float* bias, const float* images, int R,
                               float gscale, float* part, hipStream_t s) {
  DVG_LAUNCH(K_DEC_FINAL_BWD, (dec_final_wgrad_kernel<true, true>), dim3(EW_BLOCKS), dim3(256), 0, s, nullptr, in, N, nullptr, part,
             w, bias, images, R, gscale);
  return DVG_OK;
}

}  // namespace dvg
