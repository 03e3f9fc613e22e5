// Fused multi-bandwidth RBF MMD loss + gradient wrt x.
//
// Replaces GaussianKernel(n_kernels=7) + maximum_mean_discrepancy_loss(x, y, kernel)
// (/root/reference/src/model_wrapper.py:273, :320; plugin code absent, restated in
// oracle/plugin.py).  The reference materialises the (nx+ny)^2 kernel matrix; here a
// flash-style kernel walks 32-row x 128-column tiles:
//     GEMM1 (MFMA f32 32x32x2):  T[j][i] = z_j . x_i          (Gram tile, transposed)
//     VALU:                      D = sqrt(|z_j|^2+|x_i|^2-2T),  K = sum_k exp(c_k D),
//                                w = a_ij K'(D)/D               (a_ij: estimator weights)
//     GEMM2 (MFMA f32 32x32x2):  G^T[f][i] += z_j[f] * w[j][i]  (T's accumulator registers are
//                                the B operand as they stand: no LDS transpose)
//     grad_i = x_i * sum_j w_ij - G_i
// f32 MFMA is an exact fmaf chain, so for +-1 spins the Gram is exact.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace dvg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

struct MmdArgs {
  const float* x; const float* y;
  int64_t nx, ny;
  int d;
  const float* sq;      // [nx + ny] squared row norms
  const float* coef;    // [8]: c_k = -1/(bw * mult_k); coef[7+1]... see MmdCoef
  int n_kernels, squared, reduce_mean, biased;
  int pow2;             // factor == 2: exp(c_k D) by repeated squaring
  double* loss_part;    // [nblocks][3]  (xx, xy, yy)
  int vgx = 0, vgy = 0, vgz = 0;  // logical grid of the gated float32 launch (mmd_main_kernel)
  float* grad_part;     // [S][nx][d] (or grad_x itself when S == 1)
  int S;                // column splits
  double* dist_part;    // distsum mode: [nblocks]
  const int8_t* zi8;    // [nx + ny][d] int8 copy of (x ; y), valid when *not_pm1 == 0
  const int* not_pm1;   // device flag written by mmd_prep_kernel: 0 <=> every entry of x and y is exactly +-1
  int pm1_ok;           // host: shape is served by the +-1 kernels (so the f32 kernels may stand down on the flag)
  int gate_main;        // host: mmd_main_kernel runs behind a spin-only kernel and stands down when *not_pm1 == 0
  const uint16_t* zt;   // bf16 transposed copy [32-row block][d][32] of (x ; y), each padded to whole 128-row tiles
  int64_t ztb_y;        // first 32-row block of y inside zt
  const uint4* tab;     // [2][d + 1] pair table for +-1 rows (see mmd_table_kernel)
};

constexpr int MMD_BI = 32;    // rows (i) per block
constexpr int MMD_BJ = 128;   // columns (j) per tile: 4 waves x 32
constexpr int MMD_PITCH = 33;

__device__ __forceinline__ int crow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// Gram tile for this wave: T[jj][ii], jj in [0,32) (wave's j-block), ii in [0,32).
// src_j/cnt_j/base_j describe the column set; rows are src_i/cnt_i/base_i.
__device__ __forceinline__ f32x16 gram_tile(const float* __restrict__ src_i, int64_t cnt_i, int64_t base_i,
                                            const float* __restrict__ src_j, int64_t cnt_j, int64_t base_j, int d,
                                            float* Zs, float* Xs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, c = lane & 31;
  f32x16 acc = {0};
  const int nchunk = d / 32;
  f32x4 zreg[4], xreg;
  float zmask[4], xmask;
  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx >> 3, c4 = idx & 7;
      const int64_t gr = base_j + row;
      const bool okz = gr < cnt_j;
      // unconditional load; the 0/1 mask is applied when parking the registers in LDS (see conv_igemm.hip)
      zreg[q] = *reinterpret_cast<const f32x4*>(src_j + (okz ? gr * d + ch * 32 + c4 * 4 : 0));
      zmask[q] = okz ? 1.0f : 0.0f;
    }
    const int row = tid >> 3, c4 = tid & 7;
    const int64_t gr = base_i + row;
    const bool okx = gr < cnt_i;
    xreg = *reinterpret_cast<const f32x4*>(src_i + (okx ? gr * d + ch * 32 + c4 * 4 : 0));
    xmask = okx ? 1.0f : 0.0f;
  };
  load_chunk(0);
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();  // previous chunk's MFMA reads are done
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx >> 3, c4 = idx & 7;
      float* p = Zs + row * MMD_PITCH + c4 * 4;
      p[0] = zreg[q].x * zmask[q]; p[1] = zreg[q].y * zmask[q]; p[2] = zreg[q].z * zmask[q]; p[3] = zreg[q].w * zmask[q];
    }
    {
      const int row = tid >> 3, c4 = tid & 7;
      float* p = Xs + row * MMD_PITCH + c4 * 4;
      p[0] = xreg.x * xmask; p[1] = xreg.y * xmask; p[2] = xreg.z * xmask; p[3] = xreg.w * xmask;
    }
    __syncthreads();
    if (ch + 1 < nchunk) load_chunk(ch + 1);
    const float* za = Zs + (wave * 32 + c) * MMD_PITCH + hh;
    const float* xb = Xs + c * MMD_PITCH + hh;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[2 * s], xb[2 * s], acc, 0, 0, 0);
  }
  return acc;
}

// Exact Gram tile for +-1 inputs on the int8 MFMA (v_mfma_i32_32x32x32_i8: K = 32 per instruction, 32x the K-rate
// of the f32 MFMA).  The A and B fragments are both "16 consecutive bytes of my row at the same k offset", so any
// internal k ordering of the instruction cancels out of the dot product.  Whole 512-feature panels are staged at once.
constexpr int MMD_I8_PANEL = 512;

// Two-step staging of ROWS x (16 n16) bytes of int8 rows (n16 <= 32) so that a panel's global loads can be in flight
// while the block computes: load() issues them all (8 threads cover one 128-byte row segment), store() writes LDS.
// Addresses are clamped instead of guarded (rows past the end re-read the last row; callers mask those pairs), which
// keeps the loads one straight batch -- guarded loads become branches the compiler waits on one by one.
template <int ROWS>
struct I8Stage {
  i32x4 v[ROWS / 32][4];
  __device__ __forceinline__ void load(const int8_t* __restrict__ src, int d, int64_t base, int64_t cnt, int n16) {
    const int tr = threadIdx.x >> 3, tc = threadIdx.x & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q * 8 < n16) {
        const int c16 = q * 8 + tc < n16 ? q * 8 + tc : n16 - 1;
#pragma unroll
        for (int p = 0; p < ROWS / 32; ++p) {
          int64_t gr = base + tr + 32 * p;
          gr = gr < cnt ? gr : cnt - 1;
          v[p][q] = *reinterpret_cast<const i32x4*>(src + gr * d + c16 * 16);
        }
      }
    }
  }
  __device__ __forceinline__ void store(int8_t* dst, int pitch, int n16) const {
    const int tr = threadIdx.x >> 3, tc = threadIdx.x & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q * 8 < n16) {
        const int c16 = q * 8 + tc < n16 ? q * 8 + tc : n16 - 1;
#pragma unroll
        for (int p = 0; p < ROWS / 32; ++p)
          *reinterpret_cast<i32x4*>(dst + (tr + 32 * p) * pitch + c16 * 16) = v[p][q];
      }
    }
  }
};

__device__ __forceinline__ f32x16 gram_tile_i8(const int8_t* __restrict__ zi8, int64_t goff_i, int64_t cnt_i,
                                               int64_t base_i, int64_t goff_j, int64_t cnt_j, int64_t base_j, int d,
                                               int8_t* Zs8, int8_t* Xs8) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, c = lane & 31;
  const int pw = d < MMD_I8_PANEL ? d : MMD_I8_PANEL;
  const int pitch = pw + 16;
  i32x16 acc = {0};
  for (int c0 = 0; c0 < d; c0 += MMD_I8_PANEL) {
    const int cw = d - c0 < MMD_I8_PANEL ? d - c0 : MMD_I8_PANEL;
    const int n16 = cw >> 4;
    I8Stage<MMD_BJ> zst;
    I8Stage<MMD_BI> xst;
    zst.load(zi8 + goff_j * d + c0, d, base_j, cnt_j, n16);
    xst.load(zi8 + goff_i * d + c0, d, base_i, cnt_i, n16);
    __syncthreads();
    zst.store(Zs8, pitch, n16);
    xst.store(Xs8, pitch, n16);
    __syncthreads();
    const int8_t* za = Zs8 + (wave * 32 + c) * pitch + hh * 16;
    const int8_t* xb = Xs8 + c * pitch + hh * 16;
    for (int s = 0; s < (cw >> 5); ++s)
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(za + s * 32),
                                                  *reinterpret_cast<const i32x4*>(xb + s * 32), acc, 0, 0, 0);
  }
  f32x16 out;
#pragma unroll
  for (int r = 0; r < 16; ++r) out[r] = (float)acc[r];
  return out;
}

// Kernel-sum terms of one pair at distance D: ks = sum_k exp(c_k D), kp = sum_k c_k exp(c_k D).
__device__ __forceinline__ void mmd_pair_terms(float D, const float (&ck)[8], int nk, bool pow2, float& ks, float& kp) {
  ks = 0.f; kp = 0.f;
  if (pow2) {
    // bandwidth multipliers are powers of two: c_k = 2 c_{k+1}, so exp(c_k D) = exp(c_{k+1} D)^2.
    // One exp for the widest kernel, the others by repeated squaring (VALU-bound phase: 7 exps -> 1).
    float e = expf(ck[nk - 1] * D);
#pragma unroll
    for (int k = 7; k >= 0; --k) {
      if (k < nk) {
        ks += e;
        kp = fmaf(ck[k], e, kp);
        e *= e;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < nk) {
        const float e = expf(ck[k] * D);
        ks += e;
        kp = fmaf(ck[k], e, kp);
      }
    }
  }
}

__device__ __forceinline__ double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// ------------------------------------------------------------------ pass 0: row norms
__global__ __launch_bounds__(256) void mmd_prep_kernel(const float* __restrict__ x, int64_t nx,
                                                       const float* __restrict__ y, int64_t ny, int d,
                                                       float* __restrict__ sq, int8_t* __restrict__ zi8,
                                                       int* __restrict__ not_pm1) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= nx + ny) return;
  const float* p = row < nx ? x + row * d : y + (row - nx) * d;
  float acc = 0.f;
  bool bad = false;
  // four entries per lane and trip (d % 32 == 0): one 16-byte load, one 4-byte store of the int8 copy -- the one-float
  // form stored single bytes and ran at 1.1 TB/s (c3: 77 us at the head of the MMD chain)
  uint32_t* z4 = reinterpret_cast<uint32_t*>(zi8 + row * d);
  for (int k = lane; k < d / 4; k += 64) {
    const float4 v = *reinterpret_cast<const float4*>(p + 4 * k);
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    bad |= !(v.x == 1.0f || v.x == -1.0f) | !(v.y == 1.0f || v.y == -1.0f) | !(v.z == 1.0f || v.z == -1.0f) | !(v.w == 1.0f || v.w == -1.0f);
    z4[k] = (v.x > 0.f ? 0x01u : 0xffu) | (v.y > 0.f ? 0x0100u : 0xff00u) | (v.z > 0.f ? 0x010000u : 0xff0000u) |
            (v.w > 0.f ? 0x01000000u : 0xff000000u);
  }
  if (__any(bad) && lane == 0) atomicOr(not_pm1, 1);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) sq[row] = acc;
}

// ------------------------------------------------------------------ pass 1: sum of all distances
// Spin rows, d <= 1024: 128 rows per block pass.  Each wave owns 32 of them with their int8 values resident in REGISTERS
// as MFMA B fragments; the 128-row column panel is staged once in LDS and serves all four waves (the 32-row form
// streamed every column row 4x as often: at c3 17 GB of L2 traffic per call).  The pair distance is symmetric, so over
// the concatenated row set (x then y) row block g walks only the column tiles g .. T-1 (x2 for the tiles strictly right
// of its own).  That is a triangle: block `b` of the grid takes row blocks b AND T-1-b, T+1 tiles in all for every
// block (the unfolded walk made block 0 the critical path at twice the mean length).  The distance of a pair is a
// function of its Hamming distance h = (d - <a,b>) / 2 alone, so it comes from a (d+1)-entry LDS table built by the
// block (same correctly-rounded sqrtf values as computing them per pair, which cost more VALU time than the MFMAs);
// neighbouring h sit in neighbouring banks, so the gather is conflict-free in practice.  NS <= 16 fits two blocks per
// CU (registers and LDS): one block's lookups run under the other's MFMAs.
template <int NS>  // 32-feature steps held in registers: d <= 32 NS
__device__ __forceinline__ void mmd_distsum_spin128(const MmdArgs& a, unsigned char* dsm) {
  double* red = reinterpret_cast<double*>(dsm);                 // [256]
  float* Dtab = reinterpret_cast<float*>(dsm + 2048);           // [d+1]
  const int d = a.d, pw = d < MMD_I8_PANEL ? d : MMD_I8_PANEL, zp = pw + 16;
  int8_t* Zs8 = reinterpret_cast<int8_t*>(dsm + 2048 + ((d + 1) * 4 + 15) / 16 * 16);  // [128][pw+16]
  for (int h = threadIdx.x; h <= d; h += 256) {
    const float d2 = (float)(4 * h);  // |a|^2 + |b|^2 - 2ab with |.|^2 = d: exact
    Dtab[h] = a.squared ? d2 : sqrtf(d2);
  }
  const int64_t tx = (a.nx + MMD_BJ - 1) / MMD_BJ, ty = (a.ny + MMD_BJ - 1) / MMD_BJ, T = tx + ty;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5, c = lane & 31;
  const int nsteps = d >> 5;
  double total = 0.0;
  for (int seg = 0; seg < 2; ++seg) {
    const int64_t g = seg == 0 ? (int64_t)blockIdx.x : T - 1 - (int64_t)blockIdx.x;
    if (g >= T || (seg == 1 && g <= (int64_t)blockIdx.x)) break;  // (block-uniform)
    const bool rows_x = g < tx;
    const int64_t cnt_i = rows_x ? a.nx : a.ny, base_i = (rows_x ? g : g - tx) * 128, goff_i = rows_x ? 0 : a.nx;
    const int64_t gi = base_i + wave * 32 + c;
    const bool vi = gi < cnt_i;
    i32x4 xb[NS];
    {
      const int8_t* xrow = a.zi8 + (goff_i + (vi ? gi : cnt_i - 1)) * d + hh * 16;
#pragma unroll
      for (int s = 0; s < NS; ++s) xb[s] = s < nsteps ? *reinterpret_cast<const i32x4*>(xrow + s * 32) : (i32x4){0, 0, 0, 0};
    }
    I8Stage<MMD_BJ> zst;
    auto tile_src = [&](int64_t t, const int8_t*& src, int64_t& base_j, int64_t& cnt_j) {
      const bool cx = t < tx;
      src = a.zi8 + (cx ? 0 : a.nx) * d;
      base_j = (cx ? t : t - tx) * MMD_BJ;
      cnt_j = cx ? a.nx : a.ny;
    };
    int64_t t = g + blockIdx.y;
    if (t < T) {
      const int8_t* src; int64_t bj, cj;
      tile_src(t, src, bj, cj);
      zst.load(src, d, bj, cj, pw >> 4);
    }
    for (; t < T; t += gridDim.y) {
      const int8_t* src; int64_t base_j, cnt_j;
      tile_src(t, src, base_j, cnt_j);
      i32x16 acc[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) acc[jt] = (i32x16){0};
      for (int c0 = 0; c0 < d; c0 += MMD_I8_PANEL) {
        const int cw = d - c0 < MMD_I8_PANEL ? d - c0 : MMD_I8_PANEL;
        if (c0 > 0) zst.load(src + c0, d, base_j, cnt_j, cw >> 4);
        __syncthreads();
        zst.store(Zs8, zp, cw >> 4);
        __syncthreads();
        const int s0 = c0 >> 5;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (s >= s0 && s < s0 + (cw >> 5)) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
              const i32x4 za = *reinterpret_cast<const i32x4*>(Zs8 + (jt * 32 + c) * zp + (s - s0) * 32 + hh * 16);
              acc[jt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(za, xb[s], acc[jt], 0, 0, 0);
            }
          }
        }
      }
      {  // prefetch the next tile's first panel under the distance sums
        const int64_t tn = t + gridDim.y;
        if (tn < T) {
          const int8_t* nsrc; int64_t nbj, ncj;
          tile_src(tn, nsrc, nbj, ncj);
          zst.load(nsrc, d, nbj, ncj, pw >> 4);
        }
      }
      const float wgt = t == g ? 1.0f : 2.0f;  // the own (diagonal) tile holds both orders of its pairs
      float part = 0.f;
      if (base_i + 128 <= cnt_i && base_j + MMD_BJ <= cnt_j) {  // whole tile valid (block-uniform): no masks
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int r = 0; r < 16; ++r) part += Dtab[(d - acc[jt][r]) >> 1];
      } else {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t gj = base_j + jt * 32 + crow(r, hh);
            const float D = Dtab[(d - acc[jt][r]) >> 1];
            part += (vi && gj < cnt_j) ? D : 0.f;
          }
      }
      total += (double)(part * wgt);
    }
  }
  const double sum = block_sum(total, red);
  if (threadIdx.x == 0) a.dist_part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = sum;
}

// General rows (f32 Gram), or spin rows wider than the register-resident form takes (int8 Gram through LDS): 32-row
// blocks, grid-stride over them so the launch geometry can be the spin form's.
__device__ __forceinline__ void mmd_distsum_generic(const MmdArgs& a, unsigned char* dsm, bool I8) {
  double* red = reinterpret_cast<double*>(dsm);                 // [256]
  float* Zs = reinterpret_cast<float*>(dsm + 2048);             // f32 path: [128][33], then Xs [32][33]
  float* Xs = Zs + MMD_BJ * MMD_PITCH;
  int8_t* Zs8 = reinterpret_cast<int8_t*>(dsm + 2048);          // int8 path: [128][pw+16], then Xs8 [32][pw+16]
  int8_t* Xs8 = Zs8 + MMD_BJ * ((a.d < MMD_I8_PANEL ? a.d : MMD_I8_PANEL) + 16);
  const int64_t rbx = (a.nx + MMD_BI - 1) / MMD_BI, rby = (a.ny + MMD_BI - 1) / MMD_BI;
  const int64_t tx = (a.nx + MMD_BJ - 1) / MMD_BJ, ty = (a.ny + MMD_BJ - 1) / MMD_BJ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5, c = lane & 31;
  double total = 0.0;
  for (int64_t rb = blockIdx.x; rb < rbx + rby; rb += gridDim.x) {
    const bool rows_x = rb < rbx;
    const float* src_i = rows_x ? a.x : a.y;
    const int64_t cnt_i = rows_x ? a.nx : a.ny, base_i = (rows_x ? rb : rb - rbx) * MMD_BI;
    const float* sq_i = rows_x ? a.sq : a.sq + a.nx;
    const int64_t gi = base_i + c;
    const float sqi = gi < cnt_i ? sq_i[gi] : 0.f;
    for (int64_t t = blockIdx.y; t < tx + ty; t += gridDim.y) {
      const bool cols_x = t < tx;
      const float* src_j = cols_x ? a.x : a.y;
      const int64_t cnt_j = cols_x ? a.nx : a.ny, base_j = (cols_x ? t : t - tx) * MMD_BJ;
      const float* sq_j = cols_x ? a.sq : a.sq + a.nx;
      f32x16 Tt;
      if (I8) Tt = gram_tile_i8(a.zi8, rows_x ? 0 : a.nx, cnt_i, base_i, cols_x ? 0 : a.nx, cnt_j, base_j, a.d, Zs8, Xs8);
      else Tt = gram_tile(src_i, cnt_i, base_i, src_j, cnt_j, base_j, a.d, Zs, Xs);
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gj = base_j + wave * 32 + crow(r, hh);
        if (gi < cnt_i && gj < cnt_j) {
          const float d2 = fmaxf(sqi + sq_j[gj] - 2.0f * Tt[r], 0.f);
          part += a.squared ? d2 : sqrtf(d2);
        }
      }
      total += (double)part;
      __syncthreads();  // the next tile restages Zs / Xs
    }
  }
  const double s = block_sum(total, red);
  if (threadIdx.x == 0) a.dist_part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// Spin rows, d = 128 .. 512 in steps of 128, many rows (c3): 256 rows per block, 64 per wave, resident in registers as
// int8 MFMA B fragments; the rows of x and y are ONE contiguous int8 array (x then y), walked as a triangle of
// 256-row blocks x 32-row column chunks (the chunks of a row block are dealt to gridDim.y streams; the blocks of a
// stream share its row blocks' chunks evenly, see the walk below).  The kernel is bound by the L2 -> LDS traffic of the column rows, which this shape halves against
// the 128-row form; the chunks [32][d] arrive by LDS-DMA (buffer_load ... lds, double-buffered, 16-byte slots swizzled
// on the source address, rows past the end read zeros and are masked), one barrier per chunk.  A pair's distance is
// symmetric and zero on the diagonal: chunks inside the row block's own range count once (the square holds both
// orders), chunks to its right twice.
template <int NST>
__device__ __forceinline__ void mmd_distsum_spin256(const MmdArgs& a, unsigned char* dsm) {
  constexpr int D = 32 * NST, CH = 32 * D, NPW = NST / 4;  // chunk bytes; 1 KiB DMA pieces per wave and chunk
  typedef __attribute__((address_space(3))) void lds_void;
  double* red = reinterpret_cast<double*>(dsm);                  // [256]
  float* Dtab = reinterpret_cast<float*>(dsm + 2048);            // [D + 1]
  unsigned char* zbuf = dsm + 2048 + ((D + 1) * 4 + 1023) / 1024 * 1024;  // [2][32][D]
  for (int h = threadIdx.x; h <= D; h += 256) {
    const float d2 = (float)(4 * h);  // |a|^2 + |b|^2 - 2ab with |.|^2 = d: exact
    Dtab[h] = a.squared ? d2 : sqrtf(d2);
  }
  const int N = (int)(a.nx + a.ny);
  const int T = (N + 255) >> 8, TC = (N + 31) >> 5;  // 256-row blocks, 32-row chunks
  // XCD-aware block -> (row-block pair bx, chunk stream by) map.  The array (c3: 17 MB) does not fit one XCD's 4 MB L2,
  // and with the plain grid map the blocks that walk the same chunk stream sit on different XCDs (workgroups go
  // round-robin to the 8 XCDs in linear order): 82 % of the chunk bytes missed L2 (PMC, round 3: 0.89 GB fetched per
  // launch for 1.09 GB of chunks) and the fabric, not the MFMA, set the pace.  Here every stream lives on ONE XCD
  // (stream by on XCD by % 8) with its row blocks side by side: block bx reads at step k the chunk its neighbour
  // bx + 1 read at step k - 1, so a chunk is fetched once per XCD and hit by the other blocks.
  int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const int lin = (int)(blockIdx.x + blockIdx.y * gridDim.x), per = (int)gridDim.y >> 3;
    const int idx = lin >> 3;
    by = (lin & 7) + 8 * (idx % per);
    bx = idx / per;
  }
  const int lane = threadIdx.x & 63, hh = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.zi8), 0, N * D, 0x00020000);
  // DMA pieces: piece q of this wave covers bytes 1024 (NPW wave + q) + 16 lane of the chunk image: row r, physical slot
  // sl; it fetches logical slot sl ^ f(r)  (a pre-swizzled copy that makes the pieces contiguous measured no faster)
  int poff[NPW];
#pragma unroll
  for (int q = 0; q < NPW; ++q) {
    const int byte = (wave * NPW + q) * 1024 + lane * 16;
    const int r = byte / D, sl = (byte % D) >> 4;
    const int f = (D % 256 == 0) ? (r & 15) : ((r >> 1) & 7);
    poff[q] = r * D + ((sl ^ f) << 4);
  }
  auto issue = [&](int t, int buf) {
    unsigned char* dst = zbuf + buf * CH + wave * NPW * 1024;
#pragma unroll
    for (int q = 0; q < NPW; ++q)  // (rows past N: offset past num_records: zeros)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + q * 1024), 16, poff[q], t * CH, 0, 0);
  };
  const int fc = (D % 256 == 0) ? (c & 15) : ((c >> 1) & 7);
  // LDS byte address of Dtab[h] for Gram value S: tb2 - 2 S  (h = (D - S) / 2).  The lookups address LDS absolutely
  // (accumulator value + immediate offset, no VALU in between): the kernel's dynamic LDS block must start at address 0.
  constexpr int tb2 = 2048 + 2 * D;
  typedef __attribute__((address_space(3))) const float lds_cfloat;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  typedef __attribute__((address_space(3))) const i32x4 lds_ci32x4;
  if ((uint32_t)(uintptr_t)(lds_byte*)dsm != 0u) __builtin_trap();
  uint32_t zaddr[8];  // absolute LDS addresses of this lane's operand slots in stage 0 (see zread below)
#pragma unroll
  for (int j = 0; j < 8; ++j)
    zaddr[j] = (uint32_t)(2048 + ((D + 1) * 4 + 1023) / 1024 * 1024) + (uint32_t)(c * D + ((((hh ^ fc) ^ (2 * j)) & 15) << 4));
  double total = 0.0;
  __syncthreads();
  // The walk of stream `by`: row blocks in the order 0, T-1, 1, T-2, ... (long and short alternate), row block g holding
  // the chunks t = 8 g + by + S k, k < seglen(g).  The gridDim.x blocks of the stream cut that walk into EQUAL runs of
  // chunks -- a run covers parts of two to four row blocks -- so that exactly 2 x 256 blocks of equal work fill the chip
  // (round 2 dealt whole row-block pairs: c3's 65 pairs x 8 streams = 520 blocks for 512 slots, i.e. a second round of 8
  // blocks that doubled the kernel's time at 41 % MFMA busy).  Everything here is block-uniform scalar work.
  const int S = (int)gridDim.y, NB = (int)gridDim.x;
  auto seglen = [&](int g) { const int r = TC - g * 8 - by; return r > 0 ? (r + S - 1) / S : 0; };
  auto walk_g = [&](int q) { return (q & 1) ? T - 1 - (q >> 1) : (q >> 1); };
  int walk_total = 0;
  for (int q = 0; q < T; ++q) walk_total += seglen(walk_g(q));
  const int per = (walk_total + NB - 1) / NB;
  const int lo = bx * per, hi = lo + per < walk_total ? lo + per : walk_total;
  int base = 0;
  for (int q = 0; q < T && base < hi; ++q) {
    const int g = walk_g(q), len = seglen(g);
    const int k0 = lo > base ? lo - base : 0, k1 = hi - base < len ? hi - base : len;
    base += len;
    if (k1 <= k0) continue;  // (block-uniform)
    i32x4 xb[2][NST];
    bool vi[2];
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      const int gi = g * 256 + wave * 64 + rs * 32 + c;
      vi[rs] = gi < N;
      const int8_t* xrow = a.zi8 + (int64_t)(vi[rs] ? gi : N - 1) * D + hh * 16;
#pragma unroll
      for (int s = 0; s < NST; ++s) {
        // B operand = -2 x (bytes +1 / -1 -> 0xFE / 0x02): the Gram accumulator then holds -2 S, which IS the byte offset
        // of the pair's distance in the table (tb2 - 2 S) -- no address arithmetic between the MFMA and the lookup
        const i32x4 w = *reinterpret_cast<const i32x4*>(xrow + s * 32);
#pragma unroll
        for (int u = 0; u < 4; ++u) xb[rs][s][u] = (int)((((uint32_t)w[u] << 1) & 0xFEFEFEFEu) ^ 0xFCFCFCFCu);
      }
    }
    const int t0 = g * 8 + by + S * k0, t_end = g * 8 + by + S * k1, t_own_end = g * 8 + 8;
    float part[2] = {0.f, 0.f};  // this segment's sums of the two row sets, flushed to double below
    double segsum[2] = {0.0, 0.0};
    // distances of one chunk's two Gram tiles (`masked`: the last chunk of the array may hold rows past the end)
    auto lookups = [&](const i32x16& g0, const i32x16& g1, int t, bool masked) {
      float p0 = 0.f, p1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d0 = *reinterpret_cast<lds_cfloat*>((uintptr_t)(uint32_t)(g0[r] + tb2));
        const float d1 = *reinterpret_cast<lds_cfloat*>((uintptr_t)(uint32_t)(g1[r] + tb2));
        const bool vj = !masked || t * 32 + crow(r, hh) < N;
        p0 += vj ? d0 : 0.f;
        p1 += vj ? d1 : 0.f;
      }
      const float wgt = t < t_own_end ? 1.0f : 2.0f;
      part[0] = fmaf(p0, wgt, part[0]);
      part[1] = fmaf(p1, wgt, part[1]);
    };
    issue(t0, 0);
    int tp = -1, since_flush = 0;
    // One straight-line stream per chunk, pinned step by step (left alone the compiler waits on every operand read with
    // one read in flight and puts the 32 table lookups BEHIND the 2 NST MFMAs: round-3 PMC, 0.41 MFMA busy with nothing
    // else busy either).  Step s: the lookups of rows [rb(s), rb(s+1)) of the PREVIOUS chunk's two Gram tiles go out,
    // then the operand read of step s + PF; the step's two MFMAs; the adds of the lookups issued PF steps ago (LDS
    // returns in order: they have landed once this step's operand has).  A chunk that is looked up here is never the
    // array's last one (that one ends its row block's run of chunks), so there are no masks; the first chunk of a run
    // looks up a zero tile with weight 0.  The two tile pairs swap roles from chunk to chunk (no register copies).
    constexpr int PF = NST >= 16 ? 4 : (NST >= 8 ? 2 : 1);
    auto chunk = [&](int t, auto bufc, const i32x16 (&prev)[2], i32x16 (&cur)[2]) {
      constexpr int buf = decltype(bufc)::value;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + S < t_end) issue(t + S, buf ^ 1);
      // operand read of step s: slot (2 s + hh) ^ fc = ((hh ^ fc) ^ (2 s & 15)) | (2 s & 16): one of 8 per-lane addresses
      // plus compile-time offsets (stage, upper half of the row) -- 8 address registers instead of 2 NST
      auto zread = [&](int st) -> i32x4 {
        return *reinterpret_cast<lds_ci32x4*>((uintptr_t)(zaddr[st & 7] + (uint32_t)(buf * CH + (st >> 3) * 256)));
      };
      auto rb = [](int st) { return (16 * st) / NST; };
      i32x4 zr[NST];
      float lk0[16], lk1[16], p0 = 0.f, p1 = 0.f;
#pragma unroll
      for (int s = 0; s < PF; ++s) zr[s] = zread(s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NST; ++s) {
#pragma unroll
        for (int r = rb(s); r < rb(s + 1); ++r) {
          lk0[r] = *reinterpret_cast<lds_cfloat*>((uintptr_t)(uint32_t)(prev[0][r] + tb2));
          lk1[r] = *reinterpret_cast<lds_cfloat*>((uintptr_t)(uint32_t)(prev[1][r] + tb2));
        }
        if (s + PF < NST) zr[s + PF] = zread(s + PF);
        __builtin_amdgcn_sched_barrier(0);
        cur[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(zr[s], xb[0][s], s == 0 ? (i32x16){0} : cur[0], 0, 0, 0);
        cur[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(zr[s], xb[1][s], s == 0 ? (i32x16){0} : cur[1], 0, 0, 0);
        if (s >= PF) {
#pragma unroll
          for (int r = rb(s - PF); r < rb(s - PF + 1); ++r) { p0 += lk0[r]; p1 += lk1[r]; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = rb(NST - PF); r < 16; ++r) { p0 += lk0[r]; p1 += lk1[r]; }
      const float wgt = tp < 0 ? 0.0f : (tp < t_own_end ? 1.0f : 2.0f);
      part[0] = fmaf(p0, wgt, part[0]);
      part[1] = fmaf(p1, wgt, part[1]);
      tp = t;
      if (++since_flush == 16) {  // keep the float32 running sums short
        segsum[0] += (double)part[0]; segsum[1] += (double)part[1];
        part[0] = 0.f; part[1] = 0.f;
        since_flush = 0;
      }
    };
    i32x16 accp[2], accq[2];  // Gram tiles of the previous / the current chunk: their lookups run under the next MFMAs
    accp[0] = (i32x16){0}; accp[1] = (i32x16){0};
    bool last_in_q = false;
    for (int t = t0; t < t_end; t += 2 * S) {
      chunk(t, std::integral_constant<int, 0>{}, accp, accq);
      last_in_q = true;
      if (t + S >= t_end) break;
      chunk(t + S, std::integral_constant<int, 1>{}, accq, accp);
      last_in_q = false;
    }
    if (last_in_q) { accp[0] = accq[0]; accp[1] = accq[1]; }  // (block-uniform)
    if (tp >= 0) lookups(accp[0], accp[1], tp, tp * 32 + 32 > N);
    segsum[0] += (double)part[0]; segsum[1] += (double)part[1];
    total += (vi[0] ? segsum[0] : 0.0) + (vi[1] ? segsum[1] : 0.0);
    __syncthreads();  // (the next segment's first DMA must not overtake a straggler's reads of stage 0)
  }
  const double sum = block_sum(total, red);
  if (threadIdx.x == 0) a.dist_part[(size_t)by * gridDim.x + bx] = sum;
}

template <int NST>
__global__ __launch_bounds__(256, 2) void mmd_distsum_spin256_kernel(MmdArgs a) {
  extern __shared__ __align__(16) unsigned char dsm[];
  if (*a.not_pm1 == 0) mmd_distsum_spin256<NST>(a, dsm);
  else mmd_distsum_generic(a, dsm, false);
}

// One launch serves both kinds of input (device flag written by mmd_prep_kernel: exact int8 Gram for +-1 rows).
template <int NS>
__global__ __launch_bounds__(256, NS <= 16 ? 2 : 1) void mmd_distsum_spin_kernel(MmdArgs a) {
  extern __shared__ __align__(16) unsigned char dsm[];
  if (*a.not_pm1 == 0) mmd_distsum_spin128<NS>(a, dsm);
  else mmd_distsum_generic(a, dsm, false);
}

__global__ __launch_bounds__(256) void mmd_distsum_kernel(MmdArgs a) {
  extern __shared__ __align__(16) unsigned char dsm[];
  mmd_distsum_generic(a, dsm, *a.not_pm1 == 0);
}

__device__ __forceinline__ double mmd_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ------------------------------------------------------------------ pass 2: loss sums + gradient
// (vbx, vby, vbz, vgx, vgy): the block's place in the LOGICAL grid (rbx + rby, column splits, feature slices) -- the
// hardware grid, or a virtual block of the capped 1-D grid of the gated launch behind the 128-row spin kernel
template <int NFB>
__device__ __forceinline__ void mmd_main_body(const MmdArgs& a, float* smem, int vbx, int vby, int vbz, int vgx, int vgy) {
  float* Zs = smem;                           // [128][33]
  float* Xs = Zs + MMD_BJ * MMD_PITCH;        // [32][33]
  float* Gs = Xs + MMD_BI * MMD_PITCH;        // [NFB*32][33] cross-wave reduction of G^T
  float* rs_s = Gs + NFB * 32 * MMD_PITCH;    // [4][32] row sums per wave
  double* red = reinterpret_cast<double*>(rs_s + 4 * 32 + 2);  // [256] (8-byte aligned: offsets are even)

  const int64_t rbx = (a.nx + MMD_BI - 1) / MMD_BI;
  const int64_t rb = vbx;
  const bool rows_x = rb < rbx;
  const int zslice = vbz;
  if (!rows_x && zslice > 0) return;  // y-row blocks only feed the loss; count them once
  const float* src_i = rows_x ? a.x : a.y;
  const int64_t cnt_i = rows_x ? a.nx : a.ny, base_i = (rows_x ? rb : rb - rbx) * MMD_BI;
  const float* sq_i = rows_x ? a.sq : a.sq + a.nx;
  const int64_t tx = (a.nx + MMD_BJ - 1) / MMD_BJ, ty = (a.ny + MMD_BJ - 1) / MMD_BJ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, c = lane & 31;
  const int64_t gi = base_i + c;
  const bool vi = gi < cnt_i;
  const float sqi = vi ? sq_i[gi] : 0.f;
  const int f0 = zslice * NFB * 32;

  float ck[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ck[k] = a.coef[k];
  const float kscale = a.reduce_mean ? 1.0f / (float)a.n_kernels : 1.0f;
  const int nk = a.n_kernels;
  const double dnx = (double)a.nx, dny = (double)a.ny;
  const float a_xx = (float)(2.0 / (a.biased ? dnx * dnx : dnx * (dnx - 1.0)));
  const float a_xy = (float)(-2.0 / (dnx * dny));

  f32x16 G[NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) G[fb] = (f32x16){0};
  float rowsum = 0.f;
  double l_xx = 0.0, l_xy = 0.0, l_yy = 0.0;

  // x-row blocks visit x- and y-column tiles; y-row blocks only y-column tiles (the yy term)
  const int64_t t_begin = rows_x ? 0 : tx;
  for (int64_t t = t_begin + vby; t < tx + ty; t += vgy) {
    const bool cols_x = t < tx;
    const float* src_j = cols_x ? a.x : a.y;
    const int64_t cnt_j = cols_x ? a.nx : a.ny, base_j = (cols_x ? t : t - tx) * MMD_BJ;
    const float* sq_j = cols_x ? a.sq : a.sq + a.nx;
    f32x16 T = gram_tile(src_i, cnt_i, base_i, src_j, cnt_j, base_j, a.d, Zs, Xs);

    const bool same = (rows_x == cols_x);
    const float aw = cols_x ? a_xx : a_xy;
    float w[16];
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t gj = base_j + wave * 32 + crow(r, hh);
      float wv = 0.f;
      if (vi && gj < cnt_j) {
        const bool diag = same && (gi == gj);
        // a row's distance to itself is exactly 0 (the |a|^2+|b|^2-2ab form only says so up to rounding)
        const float d2 = diag ? 0.f : fmaxf(sqi + sq_j[gj] - 2.0f * T[r], 0.f);
        const float D = a.squared ? d2 : sqrtf(d2);
        float ks, kp;
        mmd_pair_terms(D, ck, nk, a.pow2, ks, kp);
        if (a.biased || !diag) lsum += ks * kscale;
        if (rows_x) {
          const float dD = a.squared ? 2.0f : (D > 0.f ? 1.0f / D : 0.f);  // zero sub-gradient at D = 0
          wv = diag ? 0.f : aw * kscale * kp * dD;
        }
      }
      w[r] = wv;
      rowsum += wv;
    }
    if (rows_x) { if (cols_x) l_xx += (double)lsum; else l_xy += (double)lsum; }
    else l_yy += (double)lsum;

    if (rows_x && a.grad_part) {
      // GEMM2: G^T[f][i] += sum_j z_j[f] * w[j][i]; A operand straight from global (L2-hot: the
      // same rows were just staged for GEMM1), B operand = w as it sits in T's registers.
      const int f = f0 + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gj = base_j + wave * 32 + crow(r, hh);
        const bool vj = gj < cnt_j;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
          const int ff = f + fb * 32;
          const bool oka = vj && ff < a.d;
          const float av = src_j[oka ? gj * a.d + ff : 0] * (oka ? 1.0f : 0.0f);
          G[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w[r], G[fb], 0, 0, 0);
        }
      }
    }
  }

  // ---- loss partial sums (feature slice 0 only, so each pair is counted once)
  const double sxx = block_sum(l_xx, red), sxy = block_sum(l_xy, red), syy = block_sum(l_yy, red);
  if (tid == 0 && zslice == 0) {
    double* lp = a.loss_part + ((size_t)vby * vgx + vbx) * 3;
    lp[0] = sxx; lp[1] = sxy; lp[2] = syy;
  }
  if (!rows_x || !a.grad_part) return;

  // ---- combine the 4 waves (each saw a different 32-column slice of every tile), deterministic order
  rowsum += __shfl_xor(rowsum, 32, 64);
  if (hh == 0) rs_s[wave * 32 + c] = rowsum;
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* p = Gs + (fb * 32 + crow(r, hh)) * MMD_PITCH + c;
          *p = (wv == 0) ? G[fb][r] : *p + G[fb][r];
        }
    }
    __syncthreads();
  }
  // grad[i][f] = x[i][f] * rowsum_i - G^T[f][i]
  float* out = a.grad_part + (size_t)vby * a.nx * a.d;
  for (int e = tid; e < MMD_BI * NFB * 32; e += 256) {
    const int i = e / (NFB * 32), fl = e % (NFB * 32);
    const int64_t gr = base_i + i;
    const int ff = f0 + fl;
    if (gr < a.nx && ff < a.d) {
      const float rsum = (rs_s[i] + rs_s[32 + i]) + (rs_s[64 + i] + rs_s[96 + i]);
      out[gr * a.d + ff] = a.x[gr * a.d + ff] * rsum - Gs[fl * MMD_PITCH + i];
    }
  }
}


// ================================================================== +-1 ("spin") fast path
// For rows in {-1,+1}^d the squared distance is 4h with h the Hamming distance, an integer in [0, d]:
//   * the Gram tile is exact on the int8 MFMA (gram_tile_i8),
//   * every per-pair quantity (kernel sum, gradient weight) depends on h only, so the exp/sqrt/divide phase becomes
//     one 16-byte LDS table lookup per pair; the table is built on the device once the bandwidth is known,
//   * rows are exactly representable in bf16, so the gradient GEMM runs on the bf16 MFMA (32x32x16) with the f32
//     weight split EXACTLY into three bf16 terms (truncation split: hi + mid + lo == w bit for bit); products with
//     +-1 are exact and accumulation is f32, i.e. the same arithmetic as the f32 path at 16x the MFMA rate.
// The weights never leave registers: the Gram accumulator layout (row j = (r&3) + 8(r>>2) + 4h, col i = lane&31) is
// reused as the B operand of the gradient GEMM under the k-permutation j = 16s + 8(e>>2) + 4h + (e&3); the A operand
// comes from a bf16 copy of the rows written transposed and in that same k order (mmd_prep_fused_kernel).

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Table entry for Hamming distance h (16 bytes): .x = kernel sum * kscale, .y = gradient weight w (f32),
// .z = bf16 hi (upper half) | bf16 mid (lower half), .w = bf16 lo (lower half); hi + mid + lo == w exactly.
__device__ __forceinline__ void mmd_table_entry(const MmdArgs& a, const float (&ck)[8], int h, uint4* __restrict__ tab) {
  const float kscale = a.reduce_mean ? 1.0f / (float)a.n_kernels : 1.0f;
  const double dnx = (double)a.nx, dny = (double)a.ny;
  const float a_xx = (float)(2.0 / (a.biased ? dnx * dnx : dnx * (dnx - 1.0)));
  const float a_xy = (float)(-2.0 / (dnx * dny));
  const float d2 = 4.0f * (float)h;
  const float D = a.squared ? d2 : sqrtf(d2);
  float ks, kp;
  mmd_pair_terms(D, ck, a.n_kernels, a.pow2, ks, kp);
  const float dD = a.squared ? 2.0f : (D > 0.f ? 1.0f / D : 0.f);
  for (int which = 0; which < 2; ++which) {
    const float aw = which == 0 ? a_xx : a_xy;
    const float w = aw * kscale * kp * dD;
    const uint32_t wb = __float_as_uint(w);
    const uint32_t hi = wb & 0xffff0000u;
    const float r1 = w - __uint_as_float(hi);
    const uint32_t mid = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mid);
    const uint32_t lo = __float_as_uint(r2) & 0xffff0000u;
    uint4 e;
    e.x = __float_as_uint(ks * kscale);
    e.y = wb;
    e.z = hi | (mid >> 16);
    e.w = lo >> 16;
    tab[(size_t)which * (a.d + 1) + h] = e;
  }
}

// Bandwidth, kernel coefficients and (for +-1 inputs) the pair table in ONE block: wave 0 reduces the distance
// partials and publishes the coefficients, then all 256 threads fill the d + 1 table rows.
// coef layout in workspace: [0..7] c_k, [8] bandwidth.
__global__ __launch_bounds__(256) void mmd_bandwidth_table_kernel(MmdArgs a, const double* __restrict__ part, int nparts,
                                                                  double n_total, float fixed_bw, float factor,
                                                                  float* __restrict__ coef, uint4* __restrict__ tab) {
  __shared__ float ck_s[8];
  if (threadIdx.x < 64) {
    double s = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 64) s += part[k];
    s = mmd_wave_sum(s);
    if (threadIdx.x == 0) {
      const double bw = fixed_bw > 0.f ? (double)fixed_bw : s / (n_total * n_total - n_total);
      const float bwf = (float)bw;
      coef[8] = bwf;
      for (int k = 0; k < 8; ++k) {
        float cv = 0.f;
        if (k < a.n_kernels) cv = -1.0f / (bwf * powf(factor, (float)(k - a.n_kernels / 2)));
        coef[k] = cv;
        ck_s[k] = cv;
      }
    }
  }
  __syncthreads();
  if (!tab || *a.not_pm1 != 0) return;
  float ck[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ck[k] = ck_s[k];
  for (int h = threadIdx.x; h <= a.d; h += 256) mmd_table_entry(a, ck, h, tab);
}

// Both preparation passes in one (plans with the +-1 path, d <= 1024): block jb owns the 32 rows of zt block jb.  Wave w
// takes rows w, w + 4, ..: the loads of its eight rows go out together (the one-row-per-wave form of mmd_prep_kernel had
// two 16-byte loads in flight per lane: 1.2 TB/s, and the transposed copy then re-read the int8 rows byte by byte:
// 54 + 71 us at the head of c3's MMD chain, which IS the step's critical chain).  Row norms: the arithmetic and the
// order of mmd_prep_kernel (same bits).  The int8 signs go to global memory and to LDS, from where thread f gathers
// its 32 rows for the bf16 block.  (The transposed copy is written whatever the rows hold: it is only read when the
// flag says +-1.)
__global__ __launch_bounds__(256) void mmd_prep_fused_kernel(MmdArgs a, float* __restrict__ sq, int8_t* __restrict__ zi8,
                                                             int* __restrict__ not_pm1, uint16_t* __restrict__ zt,
                                                             int zero_loss_parts) {
  extern __shared__ __align__(16) unsigned char prep_smem[];  // int8 signs [32][d + 16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = a.d, pitch = d + 16;
  for (int64_t e = (int64_t)blockIdx.x * 256 + tid; e < 3 * (int64_t)zero_loss_parts; e += (int64_t)gridDim.x * 256)
    a.loss_part[e] = 0.0;
  const int64_t jb = blockIdx.x;
  const bool is_x = jb < a.ztb_y;
  const int64_t cnt = is_x ? a.nx : a.ny, row0 = (is_x ? jb : jb - a.ztb_y) * 32, goff = is_x ? 0 : a.nx;
  const float* src = is_x ? a.x : a.y;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  bool bad = false;
  for (int k = lane; k < d / 4; k += 64) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t row = row0 + wave + 4 * u;
      v[u] = row < cnt ? *reinterpret_cast<const float4*>(src + row * d + 4 * k) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t row = row0 + wave + 4 * u;
      const float4 w = v[u];
      acc[u] = fmaf(w.x, w.x, acc[u]); acc[u] = fmaf(w.y, w.y, acc[u]); acc[u] = fmaf(w.z, w.z, acc[u]); acc[u] = fmaf(w.w, w.w, acc[u]);
      bad |= !(w.x == 1.0f || w.x == -1.0f) | !(w.y == 1.0f || w.y == -1.0f) | !(w.z == 1.0f || w.z == -1.0f) | !(w.w == 1.0f || w.w == -1.0f);
      const uint32_t z = (w.x > 0.f ? 0x01u : 0xffu) | (w.y > 0.f ? 0x0100u : 0xff00u) | (w.z > 0.f ? 0x010000u : 0xff0000u) |
                         (w.w > 0.f ? 0x01000000u : 0xff000000u);
      *reinterpret_cast<uint32_t*>(prep_smem + (wave + 4 * u) * pitch + 4 * k) = z;
      if (row < cnt) reinterpret_cast<uint32_t*>(zi8 + (goff + row) * d)[k] = z;
    }
  }
  if (__any(bad) && lane == 0) atomicOr(not_pm1, 1);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    float t = acc[u];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    const int64_t row = row0 + wave + 4 * u;
    if (lane == 0 && row < cnt) sq[goff + row] = t;
  }
  __syncthreads();
  // zt[(jb * d + f) * 32 + 16 s + 8 h + e] = row[32 jb + 16 s + 8 (e >> 2) + 4 h + (e & 3)][f]  (zero beyond the last row)
  for (int f = tid; f < d; f += 256) {
    uint32_t packed[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      uint32_t v = 0;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int k = 2 * q + half, sgrp = k >> 4, hh = (k >> 3) & 1, e = k & 7;
        const int r = 16 * sgrp + 8 * (e >> 2) + 4 * hh + (e & 3);
        const uint32_t b = row0 + r < cnt ? ((int8_t)prep_smem[r * pitch + f] > 0 ? 0x3f80u : 0xbf80u) : 0u;
        v |= b << (16 * half);
      }
      packed[q] = v;
    }
    uint4* dst = reinterpret_cast<uint4*>(zt + (jb * d + f) * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_uint4(packed[4 * q], packed[4 * q + 1], packed[4 * q + 2], packed[4 * q + 3]);
  }
}

template <int NFB>
__device__ __forceinline__ void mmd_pm1_body(const MmdArgs& a, unsigned char* pm1_smem) {
  const int d = a.d;
  const int pw = d < MMD_I8_PANEL ? d : MMD_I8_PANEL;
  const int zp = pw + 16, xp = d + 16;
  // LDS: pair table | row sums | reduction scratch | X rows (all features) | Z panel (aliased by the G^T reduction)
  uint4* tab_s = reinterpret_cast<uint4*>(pm1_smem);                       // [2][d+1]
  float* rs_s = reinterpret_cast<float*>(tab_s + 2 * (d + 1));             // [4][32]
  double* red = reinterpret_cast<double*>(rs_s + 128);                     // [256]
  int8_t* Xs8 = reinterpret_cast<int8_t*>(red + 256);                      // [32][d+16]
  int8_t* Zs8 = Xs8 + MMD_BI * xp;                                         // [128][pw+16]
  float* Gs = reinterpret_cast<float*>(Zs8);                               // [NFB*32][33] (after the tile loop)

  const int64_t rbx = (a.nx + MMD_BI - 1) / MMD_BI;
  const int64_t rb = blockIdx.x;
  const bool rows_x = rb < rbx;
  const int zslice = blockIdx.z;
  if (!rows_x && zslice > 0) return;  // y-row blocks only feed the loss; count them once
  const int64_t cnt_i = rows_x ? a.nx : a.ny, base_i = (rows_x ? rb : rb - rbx) * MMD_BI;
  const int64_t goff_i = rows_x ? 0 : a.nx;
  const int64_t tx = (a.nx + MMD_BJ - 1) / MMD_BJ, ty = (a.ny + MMD_BJ - 1) / MMD_BJ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, c = lane & 31;
  const int64_t gi = base_i + c;
  const bool vi = gi < cnt_i;
  const int f0 = zslice * NFB * 32;
  const bool want_grad = rows_x && a.grad_part != nullptr;

  // ---- one-time staging: pair table and this block's 32 rows (int8, every feature)
  for (int e = tid; e < 2 * (d + 1); e += 256) tab_s[e] = a.tab[e];
  for (int c0 = 0; c0 < d; c0 += MMD_I8_PANEL) {
    const int cw = d - c0 < MMD_I8_PANEL ? d - c0 : MMD_I8_PANEL;
    I8Stage<MMD_BI> xst;
    xst.load(a.zi8 + goff_i * d + c0, d, base_i, cnt_i, cw >> 4);
    xst.store(Xs8 + c0, xp, cw >> 4);
  }

  f32x16 G[NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) G[fb] = (f32x16){0};
  float rowsum = 0.f;
  double l_xx = 0.0, l_xy = 0.0, l_yy = 0.0;

  const int64_t t_begin = rows_x ? 0 : tx;
  I8Stage<MMD_BJ> zst;
  {
    const int64_t t0 = t_begin + blockIdx.y;
    if (t0 < tx + ty) {
      const bool x0 = t0 < tx;
      zst.load(a.zi8 + (x0 ? 0 : a.nx) * d, d, (x0 ? t0 : t0 - tx) * MMD_BJ, x0 ? a.nx : a.ny, pw >> 4);
    }
  }
  for (int64_t t = t_begin + blockIdx.y; t < tx + ty; t += gridDim.y) {
    const bool cols_x = t < tx;
    const int64_t cnt_j = cols_x ? a.nx : a.ny, base_j = (cols_x ? t : t - tx) * MMD_BJ;
    const int64_t goff_j = cols_x ? 0 : a.nx;

    // ---- Gram tile on the int8 MFMA, 512-feature panels of the 128 column rows through LDS
    i32x16 acc = {0};
    for (int c0 = 0; c0 < d; c0 += MMD_I8_PANEL) {
      const int cw = d - c0 < MMD_I8_PANEL ? d - c0 : MMD_I8_PANEL;
      const int n16 = cw >> 4;
      if (c0 > 0) zst.load(a.zi8 + goff_j * d + c0, d, base_j, cnt_j, n16);  // panel 0 was prefetched
      __syncthreads();
      zst.store(Zs8, zp, n16);
      __syncthreads();
      const int8_t* za = Zs8 + (wave * 32 + c) * zp + hh * 16;
      const int8_t* xb = Xs8 + c * xp + c0 + hh * 16;
      for (int s = 0; s < (cw >> 5); ++s)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(za + s * 32),
                                                    *reinterpret_cast<const i32x4*>(xb + s * 32), acc, 0, 0, 0);
    }
    {  // prefetch the next tile's first panel; it lands while this tile's lookups and gradient GEMM run
      const int64_t tn = t + gridDim.y;
      if (tn < tx + ty) {
        const bool nx_ = tn < tx;
        zst.load(a.zi8 + (nx_ ? 0 : a.nx) * d, d, (nx_ ? tn : tn - tx) * MMD_BJ, nx_ ? a.nx : a.ny, pw >> 4);
      }
    }

    // ---- per-pair phase: one table lookup
    const bool same = (rows_x == cols_x);
    const uint4* tb = tab_s + (cols_x ? 0 : d + 1);
    // (branch-free: masks instead of selects, so the 16 lookups stay one straight line of ds_read_b128)
    uint32_t himid[16], lo[16];
    float lsum = 0.f;
    const int nj = (int)(cnt_j - base_j < MMD_BJ ? cnt_j - base_j : MMD_BJ);
    const int64_t dd = gi - base_j;
    const int dloc = (same && !a.biased && dd >= 0 && dd < MMD_BJ) ? (int)dd : -1;  // column excluded from the loss
    const i32x4* tb4 = reinterpret_cast<const i32x4*>(tb);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = wave * 32 + crow(r, hh);
      const int h = (d - acc[r]) >> 1;
      const i32x4 e = tb4[h];
      const uint32_t vm = (vi && jl < nj) ? 0xffffffffu : 0u;
      const uint32_t lm = jl != dloc ? vm : 0u;
      const uint32_t wm = rows_x ? vm : 0u;  // a row against itself has h = 0, whose table weight is already 0
      lsum += __uint_as_float((uint32_t)e[0] & lm);
      rowsum += __uint_as_float((uint32_t)e[1] & wm);
      himid[r] = (uint32_t)e[2] & wm;
      lo[r] = (uint32_t)e[3] & wm;
    }
    if (rows_x) { if (cols_x) l_xx += (double)lsum; else l_xy += (double)lsum; }
    else l_yy += (double)lsum;

    if (want_grad) {
      // ---- gradient GEMM on the bf16 MFMA: G^T[f][i] += sum_j z_j[f] * (hi + mid + lo)[j][i]
      i32x4 Bw[2][3];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 8 * s + 2 * q;
          Bw[s][0][q] = (int)__builtin_amdgcn_perm(himid[r + 1], himid[r], 0x07060302u);
          Bw[s][1][q] = (int)__builtin_amdgcn_perm(himid[r + 1], himid[r], 0x05040100u);
          Bw[s][2][q] = (int)__builtin_amdgcn_perm(lo[r + 1], lo[r], 0x05040100u);
        }
      const int64_t jb = (cols_x ? 0 : a.ztb_y) + base_j / 32 + wave;
      const uint16_t* zrow = a.zt + (jb * d + f0 + c) * 32 + 8 * hh;
      // A operand: groups of PF feature blocks, the next group's loads issued before this group's MFMAs
      constexpr int PF = NFB < 4 ? NFB : 4;
      i32x4 av[2][PF][2];
      auto load_group = [&](int g, i32x4 (&dst)[PF][2]) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int fb = g * PF + u;
            dst[u][s] = *reinterpret_cast<const i32x4*>(zrow + (size_t)fb * 32 * 32 + 16 * s);  // d % (32 NFB) == 0
          }
      };
      load_group(0, av[0]);
#pragma unroll
      for (int g = 0; g < NFB / PF; ++g) {
        if (g + 1 < NFB / PF) load_group(g + 1, av[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);  // keep the compiler from hoisting every group's loads (spills at NFB = 16)
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int term = 0; term < 3; ++term)
              G[g * PF + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[g & 1][u][s]),
                                                                      __builtin_bit_cast(bf16x8, Bw[s][term]),
                                                                      G[g * PF + u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- loss partial sums (feature slice 0 only, so each pair is counted once)
  const double sxx = block_sum(l_xx, red), sxy = block_sum(l_xy, red), syy = block_sum(l_yy, red);
  if (tid == 0 && zslice == 0) {
    double* lp = a.loss_part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
    lp[0] = sxx; lp[1] = sxy; lp[2] = syy;
  }
  if (!want_grad) return;

  // ---- combine the 4 waves (each saw a different 32-column slice of every tile), deterministic order
  rowsum += __shfl_xor(rowsum, 32, 64);
  if (hh == 0) rs_s[wave * 32 + c] = rowsum;
  __syncthreads();  // Gs aliases the Z panel: every wave is done reading it
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* p = Gs + (fb * 32 + crow(r, hh)) * MMD_PITCH + c;
          *p = (wv == 0) ? G[fb][r] : *p + G[fb][r];
        }
    }
    __syncthreads();
  }
  // grad[i][f] = x[i][f] * rowsum_i - G^T[f][i]
  float* out = a.grad_part + (size_t)blockIdx.y * a.nx * d;
  for (int e = tid; e < MMD_BI * NFB * 32; e += 256) {
    const int i = e / (NFB * 32), fl = e % (NFB * 32);
    const int64_t gr = base_i + i;
    const int ff = f0 + fl;
    if (gr < a.nx && ff < d) {
      const float rsum = (rs_s[i] + rs_s[32 + i]) + (rs_s[64 + i] + rs_s[96 + i]);
      out[gr * d + ff] = a.x[gr * d + ff] * rsum - Gs[fl * MMD_PITCH + i];
    }
  }
}

// Spin path, d = 128..512 (multiples of 128): feature-quarter form.  The four waves still split a column tile's 128 rows
// for the Gram and the table lookups, but for the gradient GEMM each wave owns a QUARTER OF THE FEATURES over ALL 128
// rows: the waves publish their packed bf16 weight fragments (already in B-operand layout: 6 x 16 bytes per lane) to
// LDS, one barrier, and every wave multiplies all 24 of them with its own rows of the transposed copy.  Per tile the
// staging, Gram and lookups then run ONCE instead of once per 256-feature slice (they were 2/3 of the kernel at
// d = 512 by ablation: 684 us -> 444 without the GEMM, 516 without the Gram MFMAs), the accumulators shrink from 128
// to 16 NFBW registers, and the end-of-kernel cross-wave reduction of G^T disappears.
template <int NFBW>  // feature blocks per wave = d / 128
__device__ __forceinline__ void mmd_pm1_fq_body(const MmdArgs& a, unsigned char* pm1_smem) {
  constexpr int NFB = 4 * NFBW;
  const int d = a.d;
  const int pw = d < MMD_I8_PANEL ? d : MMD_I8_PANEL;
  const int zp = pw + 16, xp = d + 16;
  // LDS: pair table | row sums | reduction scratch | X rows (all features) | Z panel (aliased by the G^T reduction)
  uint4* tab_s = reinterpret_cast<uint4*>(pm1_smem);                       // [2][d+1]
  float* rs_s = reinterpret_cast<float*>(tab_s + 2 * (d + 1));             // [4][32]
  double* red = reinterpret_cast<double*>(rs_s + 128);                     // [256]
  int8_t* Xs8 = reinterpret_cast<int8_t*>(red + 256);                      // [32][d+16]
  int8_t* Zs8 = Xs8 + MMD_BI * xp;                                         // [128][pw+16]
  float* Gs = reinterpret_cast<float*>(Zs8);                               // [NFB*32][33] (after the tile loop)
  i32x4* wx = reinterpret_cast<i32x4*>(Zs8 + (((size_t)MMD_BJ * zp > sizeof(float) * NFB * 32 * MMD_PITCH ? (size_t)MMD_BJ * zp : sizeof(float) * NFB * 32 * MMD_PITCH) + 15) / 16 * 16);  // [4 jb][2 s][3 terms][64 lanes]

  const int64_t rbx = (a.nx + MMD_BI - 1) / MMD_BI;
  const int64_t rb = blockIdx.x;
  const bool rows_x = rb < rbx;
  if (blockIdx.z > 0) return;  // the z extent of the grid only serves the f32 body of general inputs
  constexpr int zslice = 0;
  const int64_t cnt_i = rows_x ? a.nx : a.ny, base_i = (rows_x ? rb : rb - rbx) * MMD_BI;
  const int64_t goff_i = rows_x ? 0 : a.nx;
  const int64_t tx = (a.nx + MMD_BJ - 1) / MMD_BJ, ty = (a.ny + MMD_BJ - 1) / MMD_BJ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, c = lane & 31;
  const int64_t gi = base_i + c;
  const bool vi = gi < cnt_i;
  const int f0 = (tid >> 6) * NFBW * 32;  // this wave's feature quarter
  const bool want_grad = rows_x && a.grad_part != nullptr;

  // ---- one-time staging: pair table and this block's 32 rows (int8, every feature)
  for (int e = tid; e < 2 * (d + 1); e += 256) tab_s[e] = a.tab[e];
  for (int c0 = 0; c0 < d; c0 += MMD_I8_PANEL) {
    const int cw = d - c0 < MMD_I8_PANEL ? d - c0 : MMD_I8_PANEL;
    I8Stage<MMD_BI> xst;
    xst.load(a.zi8 + goff_i * d + c0, d, base_i, cnt_i, cw >> 4);
    xst.store(Xs8 + c0, xp, cw >> 4);
  }

  f32x16 G[NFBW];
#pragma unroll
  for (int fb = 0; fb < NFBW; ++fb) G[fb] = (f32x16){0};
  float rowsum = 0.f;
  double l_xx = 0.0, l_xy = 0.0, l_yy = 0.0;

  const int64_t t_begin = rows_x ? 0 : tx;
  I8Stage<MMD_BJ> zst;
  {
    const int64_t t0 = t_begin + blockIdx.y;
    if (t0 < tx + ty) {
      const bool x0 = t0 < tx;
      zst.load(a.zi8 + (x0 ? 0 : a.nx) * d, d, (x0 ? t0 : t0 - tx) * MMD_BJ, x0 ? a.nx : a.ny, pw >> 4);
    }
  }
  for (int64_t t = t_begin + blockIdx.y; t < tx + ty; t += gridDim.y) {
    const bool cols_x = t < tx;
    const int64_t cnt_j = cols_x ? a.nx : a.ny, base_j = (cols_x ? t : t - tx) * MMD_BJ;
    const int64_t goff_j = cols_x ? 0 : a.nx;

    // ---- Gram tile on the int8 MFMA, 512-feature panels of the 128 column rows through LDS
    i32x16 acc = {0};
    for (int c0 = 0; c0 < d; c0 += MMD_I8_PANEL) {
      const int cw = d - c0 < MMD_I8_PANEL ? d - c0 : MMD_I8_PANEL;
      const int n16 = cw >> 4;
      if (c0 > 0) zst.load(a.zi8 + goff_j * d + c0, d, base_j, cnt_j, n16);  // panel 0 was prefetched
      __syncthreads();
      zst.store(Zs8, zp, n16);
      __syncthreads();
      const int8_t* za = Zs8 + (wave * 32 + c) * zp + hh * 16;
      const int8_t* xb = Xs8 + c * xp + c0 + hh * 16;
      for (int s = 0; s < (cw >> 5); ++s)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(za + s * 32),
                                                    *reinterpret_cast<const i32x4*>(xb + s * 32), acc, 0, 0, 0);
    }
    {  // prefetch the next tile's first panel; it lands while this tile's lookups and gradient GEMM run
      const int64_t tn = t + gridDim.y;
      if (tn < tx + ty) {
        const bool nx_ = tn < tx;
        zst.load(a.zi8 + (nx_ ? 0 : a.nx) * d, d, (nx_ ? tn : tn - tx) * MMD_BJ, nx_ ? a.nx : a.ny, pw >> 4);
      }
    }

    // ---- per-pair phase: one table lookup
    const bool same = (rows_x == cols_x);
    const uint4* tb = tab_s + (cols_x ? 0 : d + 1);
    // (branch-free: masks instead of selects, so the 16 lookups stay one straight line of ds_read_b128)
    uint32_t himid[16], lo[16];
    float lsum = 0.f;
    const int nj = (int)(cnt_j - base_j < MMD_BJ ? cnt_j - base_j : MMD_BJ);
    const int64_t dd = gi - base_j;
    const int dloc = (same && !a.biased && dd >= 0 && dd < MMD_BJ) ? (int)dd : -1;  // column excluded from the loss
    const i32x4* tb4 = reinterpret_cast<const i32x4*>(tb);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = wave * 32 + crow(r, hh);
      const int h = (d - acc[r]) >> 1;
      const i32x4 e = tb4[h];
      const uint32_t vm = (vi && jl < nj) ? 0xffffffffu : 0u;
      const uint32_t lm = jl != dloc ? vm : 0u;
      const uint32_t wm = rows_x ? vm : 0u;  // a row against itself has h = 0, whose table weight is already 0
      lsum += __uint_as_float((uint32_t)e[0] & lm);
      rowsum += __uint_as_float((uint32_t)e[1] & wm);
      himid[r] = (uint32_t)e[2] & wm;
      lo[r] = (uint32_t)e[3] & wm;
    }
    if (rows_x) { if (cols_x) l_xx += (double)lsum; else l_xy += (double)lsum; }
    else l_yy += (double)lsum;

    if (want_grad) {
      // ---- gradient GEMM on the bf16 MFMA: G^T[f][i] += sum_j z_j[f] * (hi + mid + lo)[j][i]
      i32x4 Bw[2][3];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 8 * s + 2 * q;
          Bw[s][0][q] = (int)__builtin_amdgcn_perm(himid[r + 1], himid[r], 0x07060302u);
          Bw[s][1][q] = (int)__builtin_amdgcn_perm(himid[r + 1], himid[r], 0x05040100u);
          Bw[s][2][q] = (int)__builtin_amdgcn_perm(lo[r + 1], lo[r], 0x05040100u);
        }
      // publish this wave's 6 B-operand fragments, then every wave reads all 24 (conflict-free: lane-linear)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int term = 0; term < 3; ++term) wx[((wave * 2 + s) * 3 + term) * 64 + lane] = Bw[s][term];
      __syncthreads();
      const uint16_t* zbase = a.zt + (((cols_x ? 0 : a.ztb_y) + base_j / 32) * d + f0 + c) * 32 + 8 * hh;
      i32x4 av[2][2][NFBW];
      auto load_jb = [&](int jb, i32x4 (&dst)[2][NFBW]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int fb = 0; fb < NFBW; ++fb)
            dst[s][fb] = *reinterpret_cast<const i32x4*>(zbase + ((size_t)jb * d + fb * 32) * 32 + 16 * s);
      };
      load_jb(0, av[0]);
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        if (jb + 1 < 4) load_jb(jb + 1, av[(jb + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int term = 0; term < 3; ++term) {
            const i32x4 bw = wx[((jb * 2 + s) * 3 + term) * 64 + lane];
#pragma unroll
            for (int fb = 0; fb < NFBW; ++fb)
              G[fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[jb & 1][s][fb]),
                                                              __builtin_bit_cast(bf16x8, bw), G[fb], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- loss partial sums (feature slice 0 only, so each pair is counted once)
  const double sxx = block_sum(l_xx, red), sxy = block_sum(l_xy, red), syy = block_sum(l_yy, red);
  if (tid == 0 && zslice == 0) {
    double* lp = a.loss_part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
    lp[0] = sxx; lp[1] = sxy; lp[2] = syy;
  }
  if (!want_grad) return;

  // ---- combine the 4 waves (each saw a different 32-column slice of every tile), deterministic order
  rowsum += __shfl_xor(rowsum, 32, 64);
  if (hh == 0) rs_s[wave * 32 + c] = rowsum;
  __syncthreads();  // Gs aliases the Z panel: every wave is done reading it
#pragma unroll
  for (int fb = 0; fb < NFBW; ++fb)
#pragma unroll
    for (int r = 0; r < 16; ++r) Gs[(wave * NFBW * 32 + fb * 32 + crow(r, hh)) * MMD_PITCH + c] = G[fb][r];
  __syncthreads();
  // grad[i][f] = x[i][f] * rowsum_i - G^T[f][i]
  float* out = a.grad_part + (size_t)blockIdx.y * a.nx * d;
  for (int e = tid; e < MMD_BI * NFB * 32; e += 256) {
    const int i = e / (NFB * 32), fl = e % (NFB * 32);
    const int64_t gr = base_i + i;
    const int ff = fl;
    if (gr < a.nx && ff < d) {
      const float rsum = (rs_s[i] + rs_s[32 + i]) + (rs_s[64 + i] + rs_s[96 + i]);
      out[gr * d + ff] = a.x[gr * d + ff] * rsum - Gs[fl * MMD_PITCH + i];
    }
  }
}

// ================================================================== spin path, large row counts: 128-row blocks
// One workgroup owns 128 rows of x (each of its four waves 32 of them, for the whole kernel) and streams the column
// rows past them in chunks of 32.  Per chunk and wave:
//     Gram   S[j][i]   (32 x 32, K = d)   int8 MFMA, A = the chunk's int8 rows from LDS, B = the wave's own rows, which
//                                          stay in REGISTERS as B fragments for the whole kernel (d / 8 VGPRs);
//     lookup w(h(S))                       one 16-byte LDS table read per pair (kernel sum, weight as 3 bf16 terms);
//     G^T[f][i] += Z^T[f][j] W[j][i]       (d x 32, K = 32) bf16 MFMA, A = the chunk's transposed bf16 copy from LDS,
//                                          B = the weights as they sit in the Gram accumulator's layout.
// Against the 32-row forms above this reads every column row once per 128 (not 32) rows of x -- they were bound by
// the L2 -> LDS traffic of their panels as much as by the matrix pipe -- keeps all of G^T (d x 32 per
// wave = 16 d / 32 accumulator registers) in the wave that needs it, so there is no exchange of weight fragments and no
// cross-wave reduction, and needs ONE workgroup barrier per chunk.  16 NFT accumulator registers hold G^T: d = 512 would
// need all 256 AGPRs and leave none for the Gram tile (the compiler then shuffles hundreds of registers per chunk through
// VGPRs and scratch), so d > 256 runs as two feature slices (grid z), each recomputing the Gram and the lookups: 128
// instead of 112 MFMAs per chunk and slice pair, everything register-resident.  Both chunk images come in by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass), double buffered, issued a whole chunk ahead; the
// bank-conflict-avoiding swizzles are applied to the per-lane SOURCE address (the DMA's LDS destination is
// lane-linear).  The wave software-pipelines across chunks: the Gram of chunk t+1 is issued ahead of the gradient GEMM
// of chunk t, and the table lookups of chunk t+1 are interleaved with that GEMM's MFMAs, so the matrix pipe has
// independent work while the lookups' LDS latencies elapse.
// ONE: only the pair table of the block's own column kind is resident (x columns or y columns: the plan guarantees that no
// block's chunk range spans both) -- the 8 KB that lets d = 1024 run with 256-feature slices of G^T (NFT = 8)
template <int NST, int NFT, bool ONE = false>  // d = 32 NST features; a block accumulates 32 NFT of them (grid z covers the rest)
struct W128 {
  static constexpr int D = 32 * NST;
  static constexpr int TABN = D + 3;  // entries per table: h = 0 .. D, the all-zero entry D+1, the diagonal entry D+2
  // 8-byte entries {kernel sum (f32), weight as two bf16 terms hi | lo}: one ds_read_b64 per pair (round 3: 16 bytes with
  // the float32 weight beside its two terms; the row sums now add the two terms themselves, which is also what G^T holds)
  static constexpr int TAB_BYTES = ((ONE ? 1 : 2) * TABN * 8 + 1023) / 1024 * 1024;
  static constexpr int RED_BYTES = 2048;
  static constexpr int Z8_BYTES = 32 * D;        // one chunk of int8 rows [32][D]
  static constexpr int ZT_BYTES = 64 * 32 * NFT; // one chunk of this block's slice of the transposed bf16 copy [32 NFT][32]
  static constexpr int NSTG = 3;                 // ring depth of both images (see the loop)
  static constexpr int OFF_RED = TAB_BYTES, OFF_Z8 = OFF_RED + RED_BYTES, OFF_ZT = OFF_Z8 + NSTG * Z8_BYTES;
  static constexpr int LDS_BYTES = OFF_ZT + NSTG * ZT_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "w128: LDS image over the CU's 160 KiB");
};

__device__ __forceinline__ void dma16(const void* gsrc, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

template <int NST, int NFT, bool ONE = false>
__device__ __forceinline__ void mmd_pm1_w128_body(const MmdArgs& a, unsigned char* smem) {
  using L = W128<NST, NFT, ONE>;
  constexpr int D = L::D;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  double* red = reinterpret_cast<double*>(smem + L::OFF_RED);
  unsigned char* z8buf = smem + L::OFF_Z8;
  unsigned char* ztbuf = smem + L::OFF_ZT;

  // (wave: uniform by construction; readfirstlane tells the compiler so -- the LDS-DMA destinations derived from it are
  // then scalar, no readfirstlane per piece)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hh = lane >> 5, c = lane & 31;
  const int64_t rbx = (a.nx + 127) / 128;
  // Block -> (row block, column split, feature slice).  One feature slice: the grid's own order (x fastest: the few y-row
  // blocks are the last to start).  Sliced form: the gradient blocks take the lowest linear ids -- the chip's first round
  // of workgroups is then gradient blocks only (the plan sizes them to the CU count) and the short loss-only blocks fill
  // in behind; in grid order, 16 of c5's 256 gradient blocks started a whole block's time late (in-kernel timestamps).
  int rb_i = blockIdx.x, sp_i = blockIdx.y, zc_i = blockIdx.z;
  if (gridDim.z > 1) {
    const int GX = gridDim.x, S = gridDim.y, nbx = (int)rbx, nby = GX - nbx;
    int lin = blockIdx.x + GX * (blockIdx.y + S * blockIdx.z);
    const int ngrad = nbx * S * gridDim.z;
    const bool gx = lin < ngrad;
    lin = gx ? lin : lin - ngrad;
    const int nb = gx ? nbx : nby;
    rb_i = (gx ? 0 : nbx) + lin % nb; sp_i = (lin / nb) % S; zc_i = lin / (nb * S);
  }
  const int64_t rb = rb_i;
  const bool rows_x = rb < rbx;
  const int sp = sp_i;
  if (!rows_x && sp > 0) return;  // y-row blocks only feed the loss: one split walks all of their (few) chunks
  const int64_t cnt_i = rows_x ? a.nx : a.ny, base_i = (rows_x ? rb : rb - rbx) * 128, goff_i = rows_x ? 0 : a.nx;
  const int64_t gi = base_i + wave * 32 + c;  // this lane's row (column of the Gram / weight tiles)
  const bool vi = gi < cnt_i;
  const int f0 = zc_i * 32 * NFT;  // first feature of this block's slice of G^T
  const bool want_grad = rows_x && a.grad_part != nullptr;
  // Feature slices beyond the first only add gradient columns; the loss-only blocks (y rows; x rows when no gradient is
  // asked for) exist once per slice too: they deal their chunks over those copies (d = 1024: 16 y-row blocks of 64 chunks
  // beside 256 gradient blocks would be a second round of workgroups as long as the first; 128 blocks of 8 chunks are not)
  const int zc = zc_i, nzc = gridDim.z;
  // chunk range of this block: x-row blocks see the x chunks then the y chunks, split evenly over gridDim.y
  const int64_t ncx = (a.nx + 31) / 32, ncy = (a.ny + 31) / 32;
  int64_t t0, t1;
  if (rows_x) {
    const int64_t nc = ncx + ncy, per = (nc + gridDim.y - 1) / gridDim.y;
    t0 = sp * per; t1 = t0 + per < nc ? t0 + per : nc;
  } else {
    t0 = ncx; t1 = ncx + ncy;
  }
  if (!want_grad && nzc > 1) {
    const int64_t per = (t1 - t0 + nzc - 1) / nzc, u0 = t0 + zc * per;
    t1 = u0 + per < t1 ? u0 + per : t1;
    t0 = u0;
  }
  const int T = t1 > t0 ? (int)(t1 - t0) : 0;
  // (32-bit copies for the block-uniform per-chunk arithmetic of the loop: chunk and row counts are far below 2^31)
  const int ncx_i = (int)ncx, nx_i = (int)a.nx, ny_i = (int)a.ny, gi_i = (int)gi, base_i_i = (int)base_i;

  // ---- one-time staging: pair table (+ the two masking entries per table) -> LDS, this wave's 32 rows -> B fragments
  constexpr int TABN = L::TABN;
  const int one_which = (rows_x && t0 < (int64_t)((a.nx + 31) / 32)) ? 0 : 1;  // (ONE: the block's columns are x chunks / y chunks)
  for (int e = tid; e < (ONE ? 1 : 2) * TABN; e += 256) {
    const int which = ONE ? one_which : e / TABN, h = ONE ? e : e - which * TABN;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (h <= D) v = a.tab[which * (D + 1) + h];
    else if (h == D + 2) { v = a.tab[which * (D + 1)]; v.x = 0u; }
    // This kernel multiplies with TWO bf16 terms, both rounded to nearest: hi = bf16(w), lo = bf16(w - hi), so
    // |w - (hi + lo)| <= 2^-18 |w|, unbiased.  (The 32-row kernels' third, truncation-split term is below half an ulp
    // of a float32 accumulator that already holds a few hundred weights, i.e. it is rounded away add by add -- a
    // systematic loss of ~2^-17 of the sum -- and costs a third of the gradient GEMM)
    const float w = __uint_as_float(v.y);
    auto rn_bf16 = [](float f) { const uint32_t u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; };
    const uint32_t hi = rn_bf16(w);
    const uint32_t lo = rn_bf16(w - __uint_as_float(hi));
    reinterpret_cast<uint2*>(smem)[e] = make_uint2(v.x, hi | (lo >> 16));
  }
  i32x4 xb[NST];
  {
    const int8_t* xrow = a.zi8 + (goff_i + (vi ? gi : cnt_i - 1)) * D + hh * 16;
#pragma unroll
    for (int s = 0; s < NST; ++s) xb[s] = *reinterpret_cast<const i32x4*>(xrow + s * 32);
  }

  // ---- LDS-DMA of one chunk's images.  Destination = wave-uniform base + 16 lane (1 KiB per instruction); the source
  // address carries the swizzle: 16-byte slot `sl` of row `r` is stored at slot sl ^ f(r).
  //   int8 rows [32][D]: f(r) = r & 15 when a row is a whole number of 256-byte bank rows, else (r >> 1) & 7;
  //   transposed bf16 [D][32] (64-byte rows): f(r) = (r >> 2) & 3.
  // Every piece is ONE instruction, buffer_load_dwordx4 ... lds: the per-lane offset inside the chunk (z8off / ztoff:
  // constant over the kernel) is the vector offset, the chunk's position the SCALAR offset, the LDS destination a scalar
  // too -- no per-piece vector arithmetic (the global_load_lds form of round 3 cost 6 vector instructions per piece for
  // its 64-bit per-lane pointer, its row clamp and the readfirstlane of its destination).  Rows past the end of x read
  // the first rows of y, rows past the end of y are out of the buffer's range and arrive as zeros: either way those
  // pairs are masked in the lookups.
  constexpr int NZ8 = NST / 4, NZT = NFT / 2;
  int z8off[NZ8], ztoff0;  // (transposed copy: piece q of a wave = piece 0 + q KiB, the swizzle class does not change)
#pragma unroll
  for (int q = 0; q < NZ8; ++q) {
    const int byte = (wave * NZ8 + q) * 1024 + lane * 16;  // position in the [32][D] image
    const int r = byte / D, sl = (byte % D) >> 4;
    const int fz = (D % 256 == 0) ? (r & 15) : ((r >> 1) & 7);
    z8off[q] = r * D + ((sl ^ fz) << 4);
  }
  {
    const int f = wave * NZT * 16 + (lane >> 2), sl = lane & 3;  // 1 KiB piece = 16 feature rows of 64 bytes
    ztoff0 = f * 64 + ((sl ^ ((f >> 2) & 3)) << 4);
  }
  const __amdgpu_buffer_rsrc_t rsrc8 = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(a.zi8), 0, (int)((a.nx + a.ny) * D), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrct = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.zt), 0, (int)((a.ztb_y + (a.ny + 127) / 128 * 4) * (int64_t)D * 64), 0x00020000);
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_byte*)smem);
  // scalar source offsets of chunk t's images (chunks past the block's last one re-fetch the last: harmless, branch-free)
  const int tl_i = (int)t1 - 1;
  auto z8_soff = [&](int t) -> int {
    t = t < tl_i ? t : tl_i;
    const bool cx = t < ncx_i;
    return ((cx ? 0 : nx_i) + (cx ? t : t - ncx_i) * 32) * D;
  };
  auto zt_soff = [&](int t) -> int {
    t = t < tl_i ? t : tl_i;
    const bool cx = t < ncx_i;
    return (((cx ? 0 : (int)a.ztb_y) + (cx ? t : t - ncx_i)) * D + f0) * 64;
  };
  auto issue_z8 = [&](int t, int stg) {
    const int so = __builtin_amdgcn_readfirstlane(z8_soff(t));
    const uint32_t dst = lds0 + (uint32_t)(L::OFF_Z8 + stg * L::Z8_BYTES + wave * NZ8 * 1024);
#pragma unroll
    for (int q = 0; q < NZ8; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc8, (lds_void*)(uintptr_t)(dst + q * 1024), 16, z8off[q], so, 0, 0);
  };
  auto issue_zt = [&](int t, int stg) {
    const int so = __builtin_amdgcn_readfirstlane(zt_soff(t));
    const uint32_t dst = lds0 + (uint32_t)(L::OFF_ZT + stg * L::ZT_BYTES + wave * NZT * 1024);
#pragma unroll
    for (int q = 0; q < NZT; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrct, (lds_void*)(uintptr_t)(dst + q * 1024), 16, ztoff0, so + q * 1024, 0, 0);
  };

  // per-lane LDS read offsets (constant over the kernel)
  const int fz_c = (D % 256 == 0) ? (c & 15) : ((c >> 1) & 7);
  const int sw_c = (c >> 2) & 3;
  const int aoff0 = c * 64 + ((hh ^ sw_c) << 4), aoff1 = c * 64 + (((2 + hh) ^ sw_c) << 4);
  // Byte offset of k-step s inside this lane's LDS row: ((2 s + hh) ^ fz_c) << 4.  The swizzle touches the low ZSW slots
  // only, so steps s and s + ZSW / 2 differ by a CONSTANT ZSW * 16 bytes: ZSW / 2 per-lane registers + immediates
  // instead of one register per k-step.
  constexpr int ZSW = (D % 256 == 0) ? 16 : 8;
  constexpr int NZB = ZSW / 2 < NST ? ZSW / 2 : NST;
  int zlow[NZB];
#pragma unroll
  for (int m = 0; m < NZB; ++m) zlow[m] = c * D + (((2 * m + hh) ^ fz_c) << 4);

  auto gram = [&](int stg) -> i32x16 {
    const unsigned char* z = z8buf + stg * L::Z8_BYTES;
    i32x16 acc = {0};
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      const i32x4 za = *reinterpret_cast<const i32x4*>(z + zlow[s % NZB] + (s / NZB) * (ZSW * 16));
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(za, xb[s], acc, 0, 0, 0);
    }
    return acc;
  };

  f32x16 G[NFT];
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft) G[ft] = (f32x16){0};

  // Masking.  The steady-state iteration is ONE straight-line instruction stream with no per-pair masks (variants of
  // the loop body make the register allocator copy all of G^T between them every iteration; a branch inside the body
  // stops the scheduler from interleaving lookups with MFMAs; per-pair compare / select pairs made the loop
  // issue-bound).  Instead:
  //   * a row of x past the end: its lane's kernel sums are dropped when they are flushed (one select per chunk); its
  //     weights only reach columns of G^T that are never stored;
  //   * the rare chunks with invalid COLUMNS (the ragged last chunk of x or y, the dummy chunks that pad the pipeline)
  //     or with the diagonal pair of an unbiased estimate get their Gram tile rewritten before the lookups, in a
  //     block-uniform branch outside the MFMA stream: an invalid pair becomes S = -(D+2), i.e. "Hamming distance" D+1,
  //     whose table entry is all zero; the diagonal pair becomes S = -(D+4), entry D+2 = entry 0 with a zero kernel sum.
  struct ChunkMeta { bool cols_x, fix; int nj, djrow; };  // djrow: first column row of a chunk that crosses the diagonal, else INT_MIN / 2
  const bool diag_kind = !a.biased;
  auto chunk_meta = [&](int t, bool exists) -> ChunkMeta {
    ChunkMeta m;
    m.cols_x = t < ncx_i;
    const int jrow0 = (m.cols_x ? t : t - ncx_i) * 32, cnt_j = m.cols_x ? nx_i : ny_i;
    const int left = cnt_j - jrow0;
    m.nj = exists ? (left < 32 ? left : 32) : 0;
    const bool diag_here = exists && diag_kind && rows_x == m.cols_x && jrow0 + 32 > base_i_i && jrow0 < base_i_i + 128;
    m.djrow = diag_here ? jrow0 : -(1 << 30);  // (gi - djrow is then outside [0, 32) for every lane)
    m.fix = m.nj < 32 || diag_here;
    return m;
  };
  auto fixup = [&](i32x16& S, const ChunkMeta& m) {
    // (rare and block-uniform: a real branch.  The empty asm keeps the compiler from if-converting it into 32 selects per
    // chunk, and from hoisting the 16 row indices into registers that would be live across the whole loop.)
    int hho = hh;
    asm volatile("" : "+v"(hho));
    const int dloc = gi_i - m.djrow;  // this lane's diagonal pair, if the chunk holds it
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = crow(r, hho);
      int v = S[r];
      v = jl == dloc ? -(D + 4) : v;
      v = jl < m.nj ? v : -(D + 2);
      S[r] = v;
    }
  };

  // One pair-row r of the lookup phase: table read (byte offset 4 (D - S) = 8 h), then loss / row sums and -- every
  // second row -- the pair's weight pieces packed straight into the next chunk's B-operand registers (k-permuted layout
  // of the Gram accumulator: see the comment on the spin path above).  Split in two so that a row's table read is issued
  // a feature tile ahead of its use.  The row sum adds the two bf16 terms themselves (two running sums): exactly the
  // weights the gradient GEMM multiplies with.
  float lsum = 0.f, rs_hi = 0.f, rs_lo = 0.f;
  uint32_t hm_prev = 0;
  constexpr int NTERM = 2;  // bf16 terms of a weight
  auto look_issue = [&](const i32x16& S, int r, const unsigned char* tb) -> i32x2 {
    return *reinterpret_cast<const i32x2*>(tb + __mul24(S[r], -4));
  };
  // (the empty asm statements pin each row's arithmetic to the tile it is written in: its results are only needed by the
  // NEXT iteration, and left alone the compiler sinks all of it into one VALU-only stretch during which the matrix pipe
  // idles -- with one wave per SIMD nothing else would fill it)
  auto look_use = [&](const i32x2& e, int r, i32x4 (&Bn)[2][NTERM]) {
    lsum += __int_as_float(e[0]);
    const uint32_t hm = (uint32_t)e[1];
    rs_hi += __uint_as_float(hm & 0xffff0000u);
    rs_lo += __uint_as_float(hm << 16);
    asm volatile("" ::"v"(lsum), "v"(rs_hi), "v"(rs_lo));
    if (r & 1) {
      const int s2 = r >> 3, q = (r & 7) >> 1;
      Bn[s2][0][q] = (int)__builtin_amdgcn_perm(hm, hm_prev, 0x07060302u);
      Bn[s2][1][q] = (int)__builtin_amdgcn_perm(hm, hm_prev, 0x05040100u);
      asm volatile("" ::"v"(Bn[s2][0][q]), "v"(Bn[s2][1][q]));
    } else {
      hm_prev = hm;
    }
  };
  // (tb = table base + 4 D: the byte offset of Gram value S is then -4 S)
  auto table_base = [&](bool cols_x) -> const unsigned char* { return smem + ((ONE || cols_x) ? 0 : TABN * 8) + 4 * D; };
  double l_cx = 0.0, l_cy = 0.0;  // kernel sums against x columns / y columns
  // Per chunk the float32 running sums of a lane (16 pairs each) move into double accumulators.  For the row sums
  // this is an accuracy matter, not a nicety: a lane adds ~N/2 table values -- a few dozen DISTINCT values, so the
  // rounding errors of a float32 chain do not average out -- and x_i * rowsum_i - G_i subtracts two nearly equal
  // numbers (measured at c3's size against float64: 1.3e-5 of the parts with a float32 chain of 16 k terms per lane).
  double rowsum_d = 0.0;
  auto flush_lsum = [&](bool cols_x) {
    const double dl = vi ? (double)lsum : 0.0;
    l_cx += cols_x ? dl : 0.0;
    l_cy += cols_x ? 0.0 : dl;
    lsum = 0.f;
    rowsum_d += (double)rs_hi + (double)rs_lo;
    rs_hi = 0.f; rs_lo = 0.f;
  };

  i32x4 Bw[2][NTERM];
  if (T > 0 && want_grad) {
    // ================================================================================================ gradient blocks
    // Three-stage rings for both images; chunk u (counted from t0) lives in stage u % 3.  Iteration k (chunk t = t0 + k)
    // runs   G^T += Z^T(t) W(t)           transposed copy of chunk t      from stage  k      % 3
    //        lookups of chunk t+1          its Gram tile (registers) -> the next weights
    //        Gram of chunk t+2             int8 rows of chunk t+2          from stage (k + 2) % 3
    //        LDS-DMA: int8 rows of chunk t+4 -> stage (k+1) % 3,  transposed copy of chunk t+2 -> stage (k+2) % 3
    // with ONE workgroup barrier, at the TOP of the iteration and in front of the DMA issue: the stages the DMA
    // overwrites were last read in iteration k-1 (every wave is past it), and what it fetches is first read in
    // iteration k+2, behind the next barrier, in front of which each wave has waited for its own pieces (a whole
    // iteration after issuing them: that wait does not stall).  With two stages (round 3) the barrier had to sit
    // BETWEEN an iteration's last reads and the next one's first, i.e. every chunk ended in
    // `wait - barrier - first operand reads - wait`, a serial stretch with the matrix pipe empty; here the operand reads
    // of the next chunk's first tile go out beside the last tile's MFMAs and the tile pipeline never drains.  The
    // block-uniform bookkeeping of the NEXT iteration (chunk meta, DMA offsets, stage addresses: ~60 scalar and a dozen
    // vector instructions) is computed in the middle of the current one, between MFMAs, for the same reason.
    issue_z8((int)t0, 0);
    issue_z8((int)t0 + 1, 1);
    issue_z8((int)t0 + 2, 2);
    issue_zt((int)t0, 0);
    issue_zt((int)t0 + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // also publishes the pair table
    {
      const ChunkMeta m = chunk_meta((int)t0, true);
      i32x16 S = gram(0);
      if (m.fix) fixup(S, m);
      const unsigned char* tb = table_base(m.cols_x);
#pragma unroll
      for (int r = 0; r < 16; ++r) look_use(look_issue(S, r, tb), r, Bw);
      flush_lsum(m.cols_x);
    }
    i32x16 Sa = gram(1), Sb = {0};  // Sa: Gram tile of chunk t0+1 (a repeat of chunk t0's rows if T == 1: masked by its fixup)
    __builtin_amdgcn_s_barrier();   // every wave is done with stage 0 of the int8 ring: chunk t0+3 may land there
    issue_z8((int)t0 + 3, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(Sa), "+v"(Sb));  // both Gram tiles live in vector registers from here on (see the tile loop)

    // pair-rows [rb(ft), rb(ft+1)) are requested beside feature tile ft < NFT - 1 and consumed beside tile ft + 1
    auto rb = [](int ft) { return ft >= NFT - 1 ? 16 : (16 * ft + NFT - 2) / (NFT - 1); };
    constexpr int RMAX = (16 + NFT - 2) / (NFT - 1);
    constexpr int GPT = NST / NFT;  // Gram k-steps (MFMAs) beside each feature tile
    // what an iteration needs from the one before it (block-uniform unless said otherwise)
    struct Carry {
      int z8so, ztso;          // scalar source offsets of this iteration's DMA (int8 rows of chunk t+4, transposed copy of t+2)
      uint32_t z8dst, ztdst;   // ... and their LDS destinations
      bool fix, cols_x;        // lookups of this iteration (chunk t+1): Gram tile needs the fixup / which table
      int nj, djrow;
      bool prev_cols_x;        // the chunk whose sums this iteration flushes
    };
    auto carry_for = [&](int k, int stg) -> Carry {  // stg = k % 3
      Carry cy;
      const int t = (int)t0 + k;
      const int s1 = stg == 2 ? 0 : stg + 1, s2 = stg == 0 ? 2 : stg - 1;  // (k+1) % 3, (k+2) % 3
      cy.z8so = __builtin_amdgcn_readfirstlane(z8_soff(t + 4));
      cy.ztso = __builtin_amdgcn_readfirstlane(zt_soff(t + 2));
      cy.z8dst = lds0 + (uint32_t)(L::OFF_Z8 + s1 * L::Z8_BYTES + wave * NZ8 * 1024);
      cy.ztdst = lds0 + (uint32_t)(L::OFF_ZT + s2 * L::ZT_BYTES + wave * NZT * 1024);
      const bool more = k + 1 < T;
      const ChunkMeta m = chunk_meta(more ? t + 1 : t, more);
      cy.fix = m.fix; cy.cols_x = m.cols_x; cy.nj = m.nj; cy.djrow = m.djrow;
      cy.prev_cols_x = t < ncx_i;  // chunk t's lookups ran in iteration k-1 (or the prologue)
      return cy;
    };
    // per-lane LDS addresses of a stage's operands: [0], [1] the two halves of a transposed-copy row (+ 2048 per feature
    // tile), [2 ...] the swizzle classes of the int8 row (+ ZSW * 16 per class repeat)
    struct Addr { const unsigned char* a0; const unsigned char* a1; const unsigned char* z[NZB]; };
    auto addr_for = [&](int stg_zt, int stg_z8) -> Addr {
      Addr ad;
      ad.a0 = ztbuf + stg_zt * L::ZT_BYTES + aoff0;
      ad.a1 = ztbuf + stg_zt * L::ZT_BYTES + aoff1;
#pragma unroll
      for (int m = 0; m < NZB; ++m) ad.z[m] = z8buf + stg_z8 * L::Z8_BYTES + zlow[m];
      return ad;
    };
    auto zread = [&](const Addr& ad, int s_) -> i32x4 {
      return *reinterpret_cast<const i32x4*>(ad.z[s_ % NZB] + (s_ / NZB) * (ZSW * 16));
    };

    int stg = 0;  // k % 3
    Carry cy = carry_for(0, 0);
    Addr ad = addr_for(0, 2);  // ONE set of per-lane addresses, moved to the next stages in place at an iteration's last tile
    // operands of the first tile of iteration 0 (from here on they are requested a tile ahead, across iterations too)
    i32x4 ac[2], zc[GPT];
    ac[0] = *reinterpret_cast<const i32x4*>(ad.a0);
    ac[1] = *reinterpret_cast<const i32x4*>(ad.a1);
#pragma unroll
    for (int u = 0; u < GPT; ++u) zc[u] = zread(ad, u);
    i32x2 ent[RMAX];

    auto iteration = [&](int k, auto parity, i32x16& Scur, i32x16& Snext) {
      (void)parity;  // (the two instantiations differ in which accumulator set is current: Scur / Snext)
      // ---- top: the one barrier (behind the wait for this wave's pieces of the previous iteration), the rare fixup
      // (sliced form, d = 1024: an iteration is 48 MFMAs, 0.6 us -- less than the LDS-DMA round trip under 256 blocks' load,
      // and the wait was the chunk's time, 3.5 us.  What iteration k fetches is first read in iteration k + 2: the wait
      // leaves the newest iteration's pieces in flight; the barrier then still publishes everything iteration k + 1 reads.)
      if constexpr (GPT > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NZ8 + NZT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" : "+v"(Scur));
      if (cy.fix) {
        asm volatile("" ::: "memory");
        ChunkMeta m;
        m.cols_x = cy.cols_x; m.fix = true; m.nj = cy.nj; m.djrow = cy.djrow;
        fixup(Scur, m);
      }
      const unsigned char* tb = table_base(cy.cols_x);
      const Carry cur = cy;
      const int stg_n = stg == 2 ? 0 : stg + 1;  // (k+1) % 3: the next iteration's transposed-copy stage; its int8 stage is stg
      i32x4 Bn[2][NTERM];
      i32x4 an[2], zn[GPT];
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft) {
        // Everything this tile consumes was requested a tile ago: wait for it HERE, before the next tile's requests go
        // out (the compiler waits with lgkmcnt(0) at the first use; placed behind fresh requests that wait would sit
        // out their whole latency every tile).  The empty asm is that first use.
#pragma unroll
        for (int u = 0; u < GPT; ++u) asm volatile("" ::"v"(zc[u]));
        asm volatile("" ::"v"(ac[0]), "v"(ac[1]));
        if (ft > 0) {
#pragma unroll
          for (int u = 0; u < RMAX; ++u)
            if (rb(ft - 1) + u < rb(ft)) asm volatile("" ::"v"(ent[u]));
        }
        __builtin_amdgcn_sched_barrier(0);
        // the Gram k-steps go FIRST in the tile: the tile's four gradient MFMAs then separate them from the next vector
        // instruction that could read the Gram tile (inline asm: the compiler keeps no hazard book for it; the tile is
        // read a whole iteration later anyway).  The Gram tile lives in VECTOR registers ("v" operands): left to the
        // compiler it goes to the accumulator file, which G^T fills completely at d = 512 -- round 3's build moved one
        // G^T tile out to vector registers and back around this tile every chunk.  The first k-step takes C = 0.
#pragma unroll
        for (int u = 0; u < GPT; ++u) {
          if constexpr (GPT > 1) {
            if (ft == 0 && u == 0) Snext = (i32x16){0};
            Snext = __builtin_amdgcn_mfma_i32_32x32x32_i8(zc[u], xb[ft * GPT + u], Snext, 0, 0, 0);
          } else
          if (ft == 0 && u == 0)
            asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(Snext) : "v"(zc[u]), "v"(xb[ft * GPT + u]));
          else
            asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(Snext) : "v"(zc[u]), "v"(xb[ft * GPT + u]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ft + 1 < NFT) {
          an[0] = *reinterpret_cast<const i32x4*>(ad.a0 + (ft + 1) * 2048);
          an[1] = *reinterpret_cast<const i32x4*>(ad.a1 + (ft + 1) * 2048);
#pragma unroll
          for (int u = 0; u < GPT; ++u) zn[u] = zread(ad, (ft + 1) * GPT + u);
        } else {
          // the NEXT iteration's first tile (its stages were published by this iteration's barrier or earlier): every read
          // of this iteration's stages has been issued, so the addresses move on in place (stage + 1, or back by two)
          const int dzt = stg == 2 ? -2 * L::ZT_BYTES : L::ZT_BYTES;  // transposed copy: stage k % 3 -> (k + 1) % 3
          const int dz8 = stg == 0 ? -2 * L::Z8_BYTES : L::Z8_BYTES;  // int8 rows: stage (k + 2) % 3 -> k % 3
          ad.a0 += dzt; ad.a1 += dzt;
#pragma unroll
          for (int m = 0; m < NZB; ++m) ad.z[m] += dz8;
          an[0] = *reinterpret_cast<const i32x4*>(ad.a0);
          an[1] = *reinterpret_cast<const i32x4*>(ad.a1);
#pragma unroll
          for (int u = 0; u < GPT; ++u) zn[u] = zread(ad, u);
        }
        i32x2 enew[RMAX];
#pragma unroll
        for (int u = 0; u < RMAX; ++u)
          if (rb(ft) + u < rb(ft + 1)) enew[u] = look_issue(Scur, rb(ft) + u, tb);
        {  // this tile's share of the DMA pieces (all of them go out in the first half of the tiles)
          constexpr int PPT = (NZ8 + NZT + NFT / 2 - 1) / (NFT / 2);
#pragma unroll
          for (int u = 0; u < PPT; ++u) {
            const int i = ft * PPT + u;
            if (i < NZ8)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc8, (lds_void*)(uintptr_t)(cur.z8dst + i * 1024), 16, z8off[i], cur.z8so, 0, 0);
            else if (i < NZ8 + NZT)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrct, (lds_void*)(uintptr_t)(cur.ztdst + (i - NZ8) * 1024), 16, ztoff0, cur.ztso + (i - NZ8) * 1024, 0, 0);
          }
        }
        // the loads above serve the NEXT tile: they must be issued early in this tile, not after its MFMAs (left to
        // itself the scheduler sinks them to the end of the tile, where the next tile waits out their full latency) --
        // but not all in ONE MFMA gap either (round 5): three 16-byte reads per gap and wave saturate the LDS array and
        // stretch the gap from 32 to 48 cycles (MI355X_MICROARCH.md).  The group barriers at the tile's end ask for two
        // reads behind the Gram MFMA and one behind each of the first two gradient MFMAs: 1872 -> 1837 us alone at c3
        // (one read per gap, or all of the rest behind the first gradient MFMA: slower again).
#pragma unroll
        for (int term = 0; term < NTERM; ++term) {
          G[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ac[0]),
                                                          __builtin_bit_cast(bf16x8, Bw[0][term]), G[ft], 0, 0, 0);
          G[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ac[1]),
                                                          __builtin_bit_cast(bf16x8, Bw[1][term]), G[ft], 0, 0, 0);
        }
        if (ft > 0) {
#pragma unroll
          for (int u = 0; u < RMAX; ++u)
            if (rb(ft - 1) + u < rb(ft)) look_use(ent[u], rb(ft - 1) + u, Bn);
        }
        if (ft == 0) flush_lsum(cur.prev_cols_x);  // the sums of the previous iteration's lookups (complete since its last tile)
        if (ft == NFT / 2) {  // the next iteration's bookkeeping, between this tile's MFMAs
          cy = carry_for(k + 1, stg_n);
        }
#pragma unroll
        for (int u = 0; u < RMAX; ++u) ent[u] = enew[u];
        ac[0] = an[0]; ac[1] = an[1];
#pragma unroll
        for (int u = 0; u < GPT; ++u) zc[u] = zn[u];
        // wanted order inside the tile: two of the next tile's reads and the DMA pieces, an MFMA, a read and a few vector
        // instructions of the lookups, an MFMA, ...
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#pragma unroll
        for (int i = 0; i < 2 * NTERM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // pin the tile order: unpinned, the scheduler hoists every tile's loads (spills)
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int term = 0; term < NTERM; ++term) Bw[s2][term] = Bn[s2][term];
      stg = stg_n;
    };
    // two chunks per trip, so that the two Gram tiles (and the two address sets) swap roles without register copies; an
    // odd T runs one dummy chunk
    for (int k = 0; k < T; k += 2) {
      iteration(k, std::integral_constant<int, 0>{}, Sa, Sb);
      iteration(k + 1, std::integral_constant<int, 1>{}, Sb, Sa);
    }
    flush_lsum(cy.prev_cols_x);  // the last iteration's lookups belong to a dummy chunk (all masked: zeros)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (T > 0) {
    // ============================================================== loss only (y-row blocks, or no gradient asked for)
    issue_z8((int)t0, 0);
    if (T > 1) issue_z8((int)t0 + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // also publishes the pair table
    for (int k = 0; k < T; ++k) {
      const int t = (int)t0 + k;
      if (k + 2 < T) issue_z8(t + 2, (k + 2) % 3);
      const ChunkMeta m = chunk_meta(t, true);
      i32x16 S = gram(k % 3);
      if (m.fix) fixup(S, m);
      const unsigned char* tb = table_base(m.cols_x);
#pragma unroll
      for (int r = 0; r < 16; ++r) look_use(look_issue(S, r, tb), r, Bw);
      flush_lsum(m.cols_x);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    __syncthreads();
  }
  double l_xx = 0.0, l_xy = 0.0, l_yy = 0.0;
  if (rows_x) { l_xx = l_cx; l_xy = l_cy; } else { l_yy = l_cy; }

  // ---- loss partial sums
  const double sxx = block_sum(l_xx, red), sxy = block_sum(l_xy, red), syy = block_sum(l_yy, red);
  if (tid == 0) {
    // (a gradient block's kernel sums are the same in every feature slice: slice 0's count)
    const bool counts = !want_grad || zc == 0;
    double* lp = a.loss_part + (((size_t)zc * gridDim.y + sp) * gridDim.x + rb) * 3;
    lp[0] = counts ? sxx : 0.0; lp[1] = counts ? sxy : 0.0; lp[2] = counts ? syy : 0.0;
  }
  if (!want_grad) return;

  // ---- grad[i][f] = x[i][f] * rowsum_i - G^T[f][i]; a lane holds features 32 ft + 8 q + 4 hh + (0..3) of its row in
  // accumulator registers 4 q .. 4 q + 3: one 16-byte store each.  x is +-1: its sign comes from the int8 copy.
  rowsum_d += __shfl_xor(rowsum_d, 32, 64);
  const float rowsum = (float)rowsum_d;
  if (vi) {
    float* out = a.grad_part + (size_t)sp * a.nx * D + gi * D + f0;
    const int8_t* xs = a.zi8 + gi * D + f0;
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = 32 * ft + 8 * q + 4 * hh;
        const uint32_t sg = *reinterpret_cast<const uint32_t*>(xs + f);
        f32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool neg = (sg >> (8 * u + 7)) & 1u;
          o[u] = (neg ? -rowsum : rowsum) - G[ft][4 * q + u];
        }
        *reinterpret_cast<f32x4*>(out + f) = o;
      }
  }
}

template <int NST, int NFT, bool ONE = false>
__global__ __launch_bounds__(256, 1) void mmd_pair_w128_kernel(MmdArgs a) {
  extern __shared__ __align__(16) unsigned char w128_smem[];
  if (*a.not_pm1 != 0) return;  // general rows: the f32 kernel launched behind this one serves them
  mmd_pm1_w128_body<NST, NFT, ONE>(a, w128_smem);
}

template <int NFB>
__global__ __launch_bounds__(256, 1) void mmd_main_kernel(MmdArgs a) {
  extern __shared__ __align__(16) unsigned char main_smem[];
  if (a.gate_main) {
    // Launched behind the 128-row spin kernel for the rows that are not +-1 (known on the device only): a capped 1-D grid
    // whose blocks walk the logical grid, so that the usual case -- nothing to do -- is a few hundred empty workgroups,
    // not the thousands of the logical grid queueing for LDS (0.35 ms at c3) behind the kernels that do have work.
    if (*a.not_pm1 == 0) return;
    const int gx = a.vgx, gy = a.vgy, total = gx * gy * a.vgz;
    for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
      mmd_main_body<NFB>(a, reinterpret_cast<float*>(main_smem), vb % gx, (vb / gx) % gy, vb / (gx * gy), gx, gy);
      __syncthreads();
    }
    return;
  }
  mmd_main_body<NFB>(a, reinterpret_cast<float*>(main_smem), blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

template <int NFBW>
__global__ __launch_bounds__(256, 1) void mmd_pair_fq_kernel(MmdArgs a) {
  extern __shared__ __align__(16) unsigned char fq_smem[];
  if (*a.not_pm1 == 0) mmd_pm1_fq_body<NFBW>(a, fq_smem);
  else mmd_main_body<(NFBW * 4 > 8 ? 8 : NFBW * 4)>(a, reinterpret_cast<float*>(fq_smem), blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// One launch serves both kinds of input: the prep kernel's device flag picks the body (no host synchronisation, and
// no second "twin" launch whose blocks would queue behind the LDS they never use).
template <int NFB>
__global__ __launch_bounds__(256, 1) void mmd_pair_kernel(MmdArgs a) {
  extern __shared__ __align__(16) unsigned char pair_smem[];
  if (a.pm1_ok && *a.not_pm1 == 0) mmd_pm1_body<NFB>(a, pair_smem);
  else mmd_main_body<NFB>(a, reinterpret_cast<float*>(pair_smem), blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// ------------------------------------------------------------------ finalize
__global__ __launch_bounds__(256) void mmd_final_kernel(const double* __restrict__ loss_part, int nparts, int64_t nx,
                                                        int64_t ny, int biased, float* __restrict__ loss_out,
                                                        const float* __restrict__ grad_part, int S, int64_t numel,
                                                        float* __restrict__ grad_x) {
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    double sxx = 0, sxy = 0, syy = 0;
    for (int k = threadIdx.x; k < nparts; k += 64) { sxx += loss_part[3 * k]; sxy += loss_part[3 * k + 1]; syy += loss_part[3 * k + 2]; }
    sxx = mmd_wave_sum(sxx); sxy = mmd_wave_sum(sxy); syy = mmd_wave_sum(syy);
    if (threadIdx.x == 0) {
      const double dnx = (double)nx, dny = (double)ny;
      const double xx = sxx / (biased ? dnx * dnx : dnx * (dnx - 1.0));
      const double yy = syy / (biased ? dny * dny : dny * (dny - 1.0));
      *loss_out = (float)(xx + yy - 2.0 * sxy / (dnx * dny));
    }
  }
  if (grad_x && S > 1) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < numel; e += (int64_t)gridDim.x * 256) {
      float s = 0.f;
      for (int k = 0; k < S; ++k) s += grad_part[(size_t)k * numel + e];
      grad_x[e] = s;
    }
  }
}

struct MmdPlan {
  int nfb, zslices, S, S1;
  bool d256;  // pass 1 by the 256-row-block spin kernel
  int64_t rbx, rby, GX1;
  size_t off_sq, off_coef, off_dist, off_loss, off_grad, off_flag, off_zi8, off_zt, off_tab, total;
  int pm1_ok;
  int64_t ztb_x, ztb_y;  // 32-row blocks of the transposed copy (whole 128-row tiles)
  int w128, S2;          // 128-row-block pair kernel (large spin problems) and its column splits
  int z128;              // ... its feature slices (grid z): 1; d = 1024: 4 (one resident table, W128<32, 8, true>) or 8 (W128<32, 4>)
  int64_t rb128x, rb128y;
};

// option mmd_w128 = 1 / 0 forces the 128-row-block pair kernel on (wherever the shape allows it) / off: tests, A/B runs
static int mmd_w128_env() {  // (read per call: the tests flip it inside one process)
  const int64_t v = opt(OPT_MMD_W128);
  return v < 0 ? -1 : (v ? 1 : 0);
}

// column splits of the pair kernel aim for this many blocks
static int64_t mmd_target_blocks() { return 256; }
// feature slices (grid z) of the 128-row-block pair kernel: NST / NFT of the instantiation dvg_mmd_fwd_bwd launches
static int mmd_w128_slices(int d) { return d == 1024 ? 8 : 1; }

static MmdPlan mmd_plan(int64_t nx, int64_t ny, int d) {
  MmdPlan p;
  const int fbt = d / 32;
  p.nfb = fbt <= 2 ? 2 : (fbt <= 4 ? 4 : 8);  // 16 feature blocks (512 accumulator registers) would spill
  p.zslices = (fbt + p.nfb - 1) / p.nfb;
  p.rbx = ceil_div(nx, MMD_BI);
  p.rby = ceil_div(ny, MMD_BI);
  const int64_t tiles = ceil_div(nx, MMD_BJ) + ceil_div(ny, MMD_BJ);
  // column splits: aim for mmd_target_blocks() blocks, never more splits than tiles
  int64_t S = ceil_div(mmd_target_blocks(), p.rbx * p.zslices);
  if (S > tiles) S = tiles;
  if (S < 1) S = 1;
  if (S > 16) S = 16;
  p.S = (int)S;
  p.pm1_ok = d <= 1024;  // LDS: table + resident X rows + one Z panel
  // 128-row-block pair kernel: d = 128 .. 512 in steps of 128 (16 d / 32 accumulator registers per lane), and enough
  // rows of x that 128-row blocks fill the chip with at most 4 column splits of >= 16 chunks each
  // (d = 1024, round 6: the rows still fit the registers -- 128 of them hold a lane's 1024 int8 features -- and the LDS
  // image fits with 128-feature slices of G^T, W128<32, 4>: eight feature slices per row block, each redoing the int8
  // Gram (32 of its 48 MFMAs per chunk).  c5's per-GPU slice, 2048 + 2048 rows: 16 row blocks x 8 slices x 2 splits;
  // the 32-row kernel it replaces there staged 192 KB through LDS per 32 x 128 pairs and held every CU's register file
  // with one workgroup at 0.05 of the matrix peak.)
  p.rb128x = ceil_div(nx, 128);
  p.rb128y = ceil_div(ny, 128);
  {
    const int64_t chunks = ceil_div(nx, 32) + ceil_div(ny, 32);
    const int env = mmd_w128_env();
    auto splits_for = [&](int64_t z) {
      int64_t S = ceil_div(mmd_target_blocks(), p.rb128x * z);
      if (S > chunks / 16) S = chunks / 16;
      if (S < 1) S = 1;
      if (env == 1) { if (S > 16) S = 16; } else if (S > 4) S = 4;
      return S;
    };
    // d = 1024: 256-feature slices (four of them: the int8 Gram redone 4 instead of 8 times, 48 instead of 40 KB staged for
    // 64 instead of 48 MFMAs per chunk) need the LDS of one pair table: possible when no block's chunk range spans the x / y
    // boundary, i.e. when the x chunks are a whole number of column splits (c5: 64 + 64 chunks in 4 splits of 32)
    int64_t z128 = mmd_w128_slices(d);
    if (d == 1024) {
      const int64_t S4 = splits_for(4), per4 = ceil_div(chunks, S4);
      if (ceil_div(nx, 32) % per4 == 0) z128 = 4;
    }
    p.z128 = (int)z128;
    int64_t S2 = ceil_div(mmd_target_blocks(), p.rb128x * z128);
    if (S2 > chunks / 16) S2 = chunks / 16;
    if (S2 < 1) S2 = 1;
    // (the 128-row-block kernel addresses both chunk images through 32-bit buffer offsets: the int8 rows and the bf16
    // transposed copy, 2 (nx + ny) d bytes, must stay below 2^31 -- larger problems keep the 32-row kernels' 64-bit pointers)
    const bool shape_ok = d % 128 == 0 && (d <= 512 || d == 1024) && (int64_t)(nx + ny) * (int64_t)d * 2 < 2147483647LL;
    p.w128 = shape_ok && (env == 1 || (env != 0 && S2 <= 4 && p.rb128x * S2 * z128 >= 128));
    if (env == 1 && shape_ok) { S2 = S2 > 16 ? 16 : S2; }
    else if (S2 > 4) S2 = 4;
    p.S2 = (int)S2;
  }
  // pass 1 (distance sum): spin-capable shapes launch the folded 128-row form, ceil(T/2) x S1 blocks of about
  // (T+1)/S1 tiles each, two blocks per CU wanted; other shapes one block per 32-row block
  int64_t S1;
  {
    const int64_t ov = opt(OPT_MMD_D256);  // 0: never, 1: whenever the shape allows (tests); read per call
    const int env = ov < 0 ? -1 : (ov ? 1 : 0);
    const int64_t t256 = ceil_div(nx + ny, 256);
    p.d256 = p.pm1_ok && env != 0 && d % 128 == 0 && d <= 512 && (t256 >= 32 || env == 1) && (nx + ny) * (int64_t)d < 2147483647LL;
  }
  if (p.d256) {  // 256-row blocks
    p.GX1 = 64;  // 8 chunk streams (one per XCD) x 64 blocks of equal work: exactly two resident blocks per CU
    S1 = 8;
  } else if (p.pm1_ok) {
    p.GX1 = (tiles + 1) / 2;
    S1 = ceil_div(512, p.GX1);
    if (S1 > tiles + 1) S1 = tiles + 1;
    if (S1 > 32) S1 = 32;
  } else {
    p.GX1 = p.rbx + p.rby;
    S1 = ceil_div(1024, p.GX1);
    if (S1 > tiles) S1 = tiles;
    if (S1 > 16) S1 = 16;
  }
  if (S1 < 1) S1 = 1;
  p.S1 = (int)S1;
  size_t o = 0;
  p.off_sq = o; o = align_up(o + sizeof(float) * (size_t)(nx + ny), 256);
  p.off_coef = o; o = align_up(o + sizeof(float) * 16, 256);
  p.off_dist = o; o = align_up(o + sizeof(double) * (size_t)(p.S1 * p.GX1), 256);
  {
    size_t nparts = (size_t)(p.S * (p.rbx + p.rby));
    if (p.w128 && (size_t)(p.S2 * p.z128 * (p.rb128x + p.rb128y)) > nparts) nparts = (size_t)(p.S2 * p.z128 * (p.rb128x + p.rb128y));
    if (p.w128 && (size_t)(p.S2 * (p.rbx + p.rby)) > nparts) nparts = (size_t)(p.S2 * (p.rbx + p.rby));
    p.off_loss = o; o = align_up(o + sizeof(double) * 3 * nparts, 256);
  }
  {
    const int smax = p.w128 && p.S2 > p.S ? p.S2 : p.S;
    p.off_grad = o; o = align_up(o + (smax > 1 ? sizeof(float) * (size_t)smax * (size_t)nx * (size_t)d : 0), 256);
  }
  p.ztb_x = ceil_div(nx, MMD_BJ) * 4;
  p.ztb_y = ceil_div(ny, MMD_BJ) * 4;
  p.off_flag = o; o = align_up(o + sizeof(int), 256);
  p.off_zi8 = o; o = align_up(o + (size_t)(nx + ny) * (size_t)d, 256);
  p.off_zt = o; o = align_up(o + (p.pm1_ok ? (size_t)(p.ztb_x + p.ztb_y) * (size_t)d * 32 * sizeof(uint16_t) : 0), 256);
  p.off_tab = o; o = align_up(o + (p.pm1_ok ? 2 * (size_t)(d + 1) * sizeof(uint4) : 0), 256);
  p.total = o;
  return p;
}

template <int NFB>
static int launch_main(const MmdArgs& a, const MmdPlan& p, hipStream_t s) {
  const size_t lds = sizeof(float) * (size_t)(MMD_BJ * MMD_PITCH + MMD_BI * MMD_PITCH + NFB * 32 * MMD_PITCH + 4 * 32 + 2) +
                     sizeof(double) * 256;
  auto kern = mmd_main_kernel<NFB>;  // d beyond the spin path's LDS budget: the f32 body alone
  if (lds > 64 * 1024)
    DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double N = (double)(a.nx + a.ny);
  const double flops = 2.0 * N * N * a.d + 2.0 * (double)a.nx * N * a.d;  // Gram + gradient GEMM
  if (a.gate_main) {
    MmdArgs g = a;
    g.vgx = (int)(p.rbx + p.rby); g.vgy = p.S; g.vgz = p.zslices;
    const int64_t total = (int64_t)g.vgx * g.vgy * g.vgz;
    DVG_LAUNCH_WORK(K_MMD_MAIN, flops, kern, dim3((unsigned)(total < 512 ? total : 512)), dim3(256), lds, s, g);
    return DVG_OK;
  }
  DVG_LAUNCH_WORK(K_MMD_MAIN, flops, kern, dim3((unsigned)(p.rbx + p.rby), (unsigned)p.S, (unsigned)p.zslices), dim3(256), lds, s, a);
  return DVG_OK;
}

// Executed matrix work of a spin-path pair launch in bf16-MFMA-equivalent FLOPs (what bench.py prices against the
// 2.5 PFLOP/s dense bf16 peak): the int8 Gram of every visited pair (x rows against all rows, y rows against y rows)
// runs at twice the bf16 rate, hence the 0.5; the gradient GEMM runs `terms` bf16 products per pair and feature.
// `gram_passes`: how often the x-row Gram is computed (the 128-row-block kernel recomputes it per feature slice).
static void mmd_pm1_flops(int64_t nx_, int64_t ny_, int d_, bool grad, int terms, int gram_passes, double* i8, double* b16) {
  const double nx = (double)nx_, ny = (double)ny_, d = (double)d_;
  *i8 = 2.0 * (gram_passes * nx * (nx + ny) + ny * ny) * d;
  *b16 = grad ? terms * 2.0 * nx * (nx + ny) * d : 0.0;
}
static double mmd_pm1_work(const MmdArgs& a, int terms, int gram_passes = 1) {
  double i8, b16;
  mmd_pm1_flops(a.nx, a.ny, a.d, a.grad_part != nullptr, terms, gram_passes, &i8, &b16);
  return 0.5 * i8 + b16;
}

template <int NFBW>
static int launch_pair_fq(const MmdArgs& a, const MmdPlan& p, hipStream_t s) {
  constexpr int NFB = 4 * NFBW, NFBM = NFB > 8 ? 8 : NFB;
  const int d = a.d, pw = d < MMD_I8_PANEL ? d : MMD_I8_PANEL;
  const size_t zbytes = (size_t)MMD_BJ * (pw + 16), gbytes = sizeof(float) * NFB * 32 * MMD_PITCH;
  const size_t zg = ((zbytes > gbytes ? zbytes : gbytes) + 15) / 16 * 16;
  const size_t lds_s = 2 * (size_t)(d + 1) * sizeof(uint4) + sizeof(float) * 128 + sizeof(double) * 256 +
                       (size_t)MMD_BI * (d + 16) + zg + sizeof(i32x4) * 24 * 64;
  const size_t lds_m = sizeof(float) * (size_t)(MMD_BJ * MMD_PITCH + MMD_BI * MMD_PITCH + NFBM * 32 * MMD_PITCH + 4 * 32 + 2) +
                       sizeof(double) * 256;
  const size_t lds = lds_s > lds_m ? lds_s : lds_m;
  auto kern = mmd_pair_fq_kernel<NFBW>;
  if (lds > 64 * 1024)
    DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int zs = (d / 32 + NFBM - 1) / NFBM;  // slices the f32 body needs
  DVG_LAUNCH_WORK(K_MMD_PM1, mmd_pm1_work(a, 3), kern, dim3((unsigned)(p.rbx + p.rby), (unsigned)p.S, (unsigned)zs), dim3(256), lds, s, a);
  return DVG_OK;
}

template <int NST, int NFT, bool ONE = false>
static int launch_pair_w128(const MmdArgs& a, const MmdPlan& p, hipStream_t s) {
  auto kern = mmd_pair_w128_kernel<NST, NFT, ONE>;
  constexpr int lds = W128<NST, NFT, ONE>::LDS_BYTES;
  if (lds > 64 * 1024)
    DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  DVG_LAUNCH_WORK(K_MMD_PM1, mmd_pm1_work(a, 2, NST / NFT), kern, dim3((unsigned)(p.rb128x + p.rb128y), (unsigned)p.S2, NST / NFT),
                  dim3(256), (size_t)lds, s, a);
  return DVG_OK;
}

template <int NFB>
static int launch_pair(const MmdArgs& a, const MmdPlan& p, hipStream_t s) {
  const int d = a.d, pw = d < MMD_I8_PANEL ? d : MMD_I8_PANEL;
  const size_t zbytes = (size_t)MMD_BJ * (pw + 16), gbytes = sizeof(float) * NFB * 32 * MMD_PITCH;
  const size_t lds_s = 2 * (size_t)(d + 1) * sizeof(uint4) + sizeof(float) * 128 + sizeof(double) * 256 +
                       (size_t)MMD_BI * (d + 16) + (zbytes > gbytes ? zbytes : gbytes);
  const size_t lds_m = sizeof(float) * (size_t)(MMD_BJ * MMD_PITCH + MMD_BI * MMD_PITCH + NFB * 32 * MMD_PITCH + 4 * 32 + 2) +
                       sizeof(double) * 256;
  const size_t lds = lds_s > lds_m ? lds_s : lds_m;
  auto kern = mmd_pair_kernel<NFB>;
  if (lds > 64 * 1024)
    DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int zs = d / (32 * NFB);
  DVG_LAUNCH_WORK(K_MMD_PM1, mmd_pm1_work(a, 3), kern, dim3((unsigned)(p.rbx + p.rby), (unsigned)p.S, (unsigned)zs), dim3(256), lds, s, a);
  return DVG_OK;
}

// (the two-feature-slice form of the 128-row-block pair kernel at d > 256 -- the Gram and the lookups computed once per
// slice -- was retired in round 3: the single-slice form replaced it at 2.70 -> 2.10 ms)
static constexpr bool mmd_two_slices() { return false; }

}  // namespace dvg

using namespace dvg;

extern "C" int dvg_mmd_spin_flops(int64_t nx, int64_t ny, int dim, double* int8_flops, double* bf16_flops, int* bf16_terms) {
  DVG_REQUIRE(nx >= 2 && ny >= 2 && dim >= 32 && dim % 32 == 0 && int8_flops && bf16_flops && bf16_terms, "mmd_spin_flops: bad argument");
  const MmdPlan p = mmd_plan(nx, ny, dim);
  if (!p.pm1_ok) { *int8_flops = 0.0; *bf16_flops = 0.0; *bf16_terms = 0; return DVG_OK; }  // f32 kernels only
  // the form dvg_mmd_fwd_bwd launches: ONE feature slice per block (the Gram and the lookups computed once) unless the
  // two-slice form is asked for (mmd_two_slices(), d > 256 only)
  const int passes = p.w128 ? p.z128 * ((dim > 256 && dim <= 512 && mmd_two_slices()) ? 2 : 1) : 1;
  *bf16_terms = p.w128 ? 2 : 3;
  mmd_pm1_flops(nx, ny, dim, true, *bf16_terms, passes, int8_flops, bf16_flops);
  return DVG_OK;
}

extern "C" size_t dvg_mmd_workspace_bytes(int64_t nx, int64_t ny, int dim) {
  if (nx <= 0 || ny <= 0 || dim <= 0 || dim % 32) return 0;
  return mmd_plan(nx, ny, dim).total;
}

extern "C" int dvg_mmd_fwd_bwd(const float* x, int64_t nx, const float* y, int64_t ny, int dim,
                               const dvg_mmd_cfg_t* cfg, float* loss_out, float* grad_x, void* ws, size_t ws_bytes,
                               dvg_stream_t stream) {
  DVG_REQUIRE(x && y && cfg && loss_out && ws, "mmd: null argument");
  DVG_REQUIRE(nx >= 2 && ny >= 2, "mmd: need at least 2 rows in x and y (nx=%lld ny=%lld)", (long long)nx, (long long)ny);
  DVG_REQUIRE(dim >= 32 && dim % 32 == 0 && dim <= 16384, "mmd: dim=%d must be a multiple of 32", dim);
  DVG_REQUIRE(cfg->n_kernels >= 1 && cfg->n_kernels <= 8, "mmd: n_kernels=%d not in [1,8]", cfg->n_kernels);
  const MmdPlan p = mmd_plan(nx, ny, dim);
  if (ws_bytes < p.total) { set_error("mmd: workspace %zu < %zu", ws_bytes, p.total); return DVG_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  MmdArgs a;
  a.x = x; a.y = y; a.nx = nx; a.ny = ny; a.d = dim;
  a.sq = (float*)(w + p.off_sq);
  a.coef = (float*)(w + p.off_coef);
  a.n_kernels = cfg->n_kernels; a.squared = cfg->squared; a.reduce_mean = cfg->reduce_mean; a.biased = cfg->biased;
  a.pow2 = cfg->factor == 2.0f;
  a.loss_part = (double*)(w + p.off_loss);
  const int S_eff = p.w128 ? p.S2 : p.S;  // column splits of the pair kernel that will run for spin rows
  a.grad_part = grad_x ? (S_eff > 1 ? (float*)(w + p.off_grad) : grad_x) : nullptr;
  a.S = S_eff;
  a.dist_part = (double*)(w + p.off_dist);
  a.zi8 = (const int8_t*)(w + p.off_zi8);
  a.not_pm1 = (const int*)(w + p.off_flag);
  a.pm1_ok = p.pm1_ok;
  a.gate_main = 0;
  a.zt = (const uint16_t*)(w + p.off_zt);
  a.ztb_y = p.ztb_x;
  a.tab = (const uint4*)(w + p.off_tab);

  // Two implementations of the pair kernels are enqueued back to back and gate themselves on a device flag the prep
  // kernel writes (are all entries exactly +-1?), so no host synchronisation is needed to choose between them.
  DVG_CHECK_HIP(hipMemsetAsync(w + p.off_flag, 0, sizeof(int), s));
  int loss_parts = (int)(p.S * (p.rbx + p.rby));
  if (p.w128) {
    // (the 128-row-block pair kernel and the float32 kernel behind it number their loss partials by their own grids and
    // exactly one of them runs: the slots the other layout would have filled must read as zero in the final sum)
    if ((int)(p.S2 * p.z128 * (p.rb128x + p.rb128y)) > loss_parts) loss_parts = (int)(p.S2 * p.z128 * (p.rb128x + p.rb128y));
    if ((int)(p.S2 * (p.rbx + p.rby)) > loss_parts) loss_parts = (int)(p.S2 * (p.rbx + p.rby));
  }
  if (p.pm1_ok) {  // row norms, int8 copy, +-1 flag and the transposed bf16 copy in one pass over the rows
    DVG_LAUNCH(K_MMD_PREP, mmd_prep_fused_kernel, dim3((unsigned)(p.ztb_x + p.ztb_y)), dim3(256), (size_t)32 * (dim + 16), s, a,
               (float*)(w + p.off_sq), (int8_t*)(w + p.off_zi8), (int*)(w + p.off_flag), (uint16_t*)(w + p.off_zt),
               p.w128 ? loss_parts : 0);
  } else {
    DVG_LAUNCH(K_MMD_PREP, mmd_prep_kernel, dim3((unsigned)ceil_div(nx + ny, 4)), dim3(256), 0, s, x, nx, y, ny, dim,
               (float*)(w + p.off_sq), (int8_t*)(w + p.off_zi8), (int*)(w + p.off_flag));
  }
  int ndist = 0;
  if (!(cfg->bandwidth > 0.f)) {
    ndist = (int)(p.S1 * p.GX1);
    const int pw = dim < MMD_I8_PANEL ? dim : MMD_I8_PANEL;
    const size_t lds_f = 2048 + sizeof(float) * (MMD_BJ + MMD_BI) * MMD_PITCH;
    const dim3 grid((unsigned)p.GX1, (unsigned)p.S1);
    if (p.pm1_ok) {
      const size_t lds_s = 2048 + ((size_t)(dim + 1) * 4 + 15) / 16 * 16 + (size_t)MMD_BJ * (pw + 16);
      const size_t lds_d = lds_f > lds_s ? lds_f : lds_s;
      auto launch = [&](auto kern) -> int {
        if (lds_d > 64 * 1024)
          DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d));
        DVG_LAUNCH(K_MMD_DISTSUM, kern, grid, dim3(256), lds_d, s, a);
        return DVG_OK;
      };
      if (p.d256) {
        const size_t lds_2 = 2048 + ((size_t)(dim + 1) * 4 + 1023) / 1024 * 1024 + 2 * (size_t)32 * dim;
        const size_t lds_256 = lds_f > lds_2 ? lds_f : lds_2;
        auto launch256 = [&](auto kern) -> int {
          DVG_LAUNCH(K_MMD_DISTSUM, kern, grid, dim3(256), lds_256, s, a);
          return DVG_OK;
        };
        if (dim == 128) DVG_TRY(launch256(mmd_distsum_spin256_kernel<4>));
        else if (dim == 256) DVG_TRY(launch256(mmd_distsum_spin256_kernel<8>));
        else if (dim == 384) DVG_TRY(launch256(mmd_distsum_spin256_kernel<12>));
        else DVG_TRY(launch256(mmd_distsum_spin256_kernel<16>));
      } else
      if (dim <= 128) DVG_TRY(launch(mmd_distsum_spin_kernel<4>));
      else if (dim <= 512) DVG_TRY(launch(mmd_distsum_spin_kernel<16>));
      else DVG_TRY(launch(mmd_distsum_spin_kernel<32>));
    } else {
      const size_t lds_i = 2048 + (size_t)(MMD_BJ + MMD_BI) * (pw + 16);
      const size_t lds_d = lds_f > lds_i ? lds_f : lds_i;
      if (lds_d > 64 * 1024)
        DVG_CHECK_HIP(hipFuncSetAttribute((const void*)mmd_distsum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d));
      DVG_LAUNCH(K_MMD_DISTSUM, mmd_distsum_kernel, grid, dim3(256), lds_d, s, a);
    }
  }
  DVG_LAUNCH(K_MMD_FINAL, mmd_bandwidth_table_kernel, dim3(1), dim3(256), 0, s, a, (const double*)(w + p.off_dist), ndist,
             (double)(nx + ny), cfg->bandwidth, cfg->factor, (float*)(w + p.off_coef),
             p.pm1_ok ? (uint4*)(w + p.off_tab) : (uint4*)nullptr);
  int rc;
  if (p.w128) {
    // Large spin problems: 128-row blocks.  General (not +-1) rows cannot be known on the host without a sync, so the
    // f32 kernel is launched behind it with the same split count and stands down on the device flag (its blocks exit at
    // once; it writes the same loss_part / grad_part slots when it does run).
    switch (dim / 128) {
      case 1: rc = launch_pair_w128<4, 4>(a, p, s); break;
      case 2: rc = launch_pair_w128<8, 8>(a, p, s); break;
      // One feature slice per block wherever the accumulators fit the 512-register file (d = 512: 256 of them hold G^T,
      // the compiler places the Gram tile and the operands in the other half without spills): the Gram and the lookups
      // are then computed once, 80 instead of 96 MFMAs per chunk: 2.70 -> 2.10 ms at c3 (the two-slice form is retired).
      case 3: rc = launch_pair_w128<12, 12>(a, p, s); break;
      case 8: rc = p.z128 == 4 ? launch_pair_w128<32, 8, true>(a, p, s) : launch_pair_w128<32, 4>(a, p, s); break;  // (four / eight feature slices: mmd_plan)
      default: rc = launch_pair_w128<16, 16>(a, p, s); break;
    }
    DVG_TRY(rc);
    MmdArgs g = a;
    g.gate_main = 1;
    MmdPlan pg = p;
    pg.S = p.S2;
    switch (p.nfb) {
      case 2: rc = launch_main<2>(g, pg, s); break;
      case 4: rc = launch_main<4>(g, pg, s); break;
      default: rc = launch_main<8>(g, pg, s); break;
    }
  } else if (p.pm1_ok) {
    // feature blocks per launch slice: the largest of 8/4/2/1 that divides d/32 (no feature guards in the spin body;
    // 16 blocks = 256 accumulator registers makes the compiler shuffle accumulators through scratch)
    const int fbt = dim / 32;
    // d = 128, 256, 384, 512: feature-quarter form; other widths: the sliced form
    if (dim % 128 == 0 && dim <= 512) {
      switch (dim / 128) {
        case 1: rc = launch_pair_fq<1>(a, p, s); break;
        case 2: rc = launch_pair_fq<2>(a, p, s); break;
        case 3: rc = launch_pair_fq<3>(a, p, s); break;
        default: rc = launch_pair_fq<4>(a, p, s); break;
      }
    } else if (fbt % 8 == 0) rc = launch_pair<8>(a, p, s);
    else if (fbt % 4 == 0) rc = launch_pair<4>(a, p, s);
    else if (fbt % 2 == 0) rc = launch_pair<2>(a, p, s);
    else rc = launch_pair<1>(a, p, s);
  } else {
    switch (p.nfb) {
      case 2: rc = launch_main<2>(a, p, s); break;
      case 4: rc = launch_main<4>(a, p, s); break;
      default: rc = launch_main<8>(a, p, s); break;
    }
  }
  DVG_TRY(rc);
  const int64_t numel = nx * (int64_t)dim;
  const unsigned fgrid = (grad_x && S_eff > 1) ? (unsigned)(ceil_div(numel, 256) > 2048 ? 2048 : ceil_div(numel, 256)) : 1u;
  DVG_LAUNCH(K_MMD_FINAL, mmd_final_kernel, dim3(fgrid), dim3(256), 0, s, (const double*)(w + p.off_loss),
             loss_parts, nx, ny, cfg->biased, loss_out, (const float*)(w + p.off_grad), S_eff, numel, grad_x);
  return DVG_OK;
}
