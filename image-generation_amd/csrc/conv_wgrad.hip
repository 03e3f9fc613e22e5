// Weight-gradient GEMMs of the convolution layers (dW[tap] = X_tap^T dY, split-K over pixels into slabs), the
// fixed-order slab reduction back into the checkpoint layout, and the weight packing kernels
// (/root/reference/src/encoder.py:28-30, /root/reference/src/decoder.py:28, :34-38).
// Separate translation unit from conv_igemm.hip so the two can carry different code-generation options (Makefile).
#include <cstdlib>

#include "conv_tile.h"

namespace dvg {

// ------------------------------------------------------------------------------------------ wgrad
template <int WA, int WB>
__global__ __launch_bounds__(WA* WB * 64) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int NT = WA * WB * 64, BA = 32 * WA, BB = 32 * WB;
  constexpr int XP = BA + 4, YP = BB + 4;
  constexpr int RX = 8 * BA / NT, RY = 8 * BB / NT;
  __shared__ __align__(16) float Xs[32 * XP];
  __shared__ __align__(16) float Ys[32 * YP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave / WB, wb = wave % WB, hh = lane >> 5, c = lane & 31;
  const int tiles_b = a.Cout / BB;
  const int a0 = (blockIdx.x / tiles_b) * BA, b0 = (blockIdx.x % tiles_b) * BB;
  const int tap = blockIdx.y, z = blockIdx.z;
  const int L = a.L, H = 1 << L, logHW = 2 * L;
  const int64_t HWin = a.ups ? ((int64_t)1 << logHW) >> 2 : ((int64_t)1 << logHW);
  const int dy = a.ntaps == 9 ? tap / 3 - 1 : 0, dx = a.ntaps == 9 ? tap % 3 - 1 : 0;
  int64_t per = (a.M + a.ksplit - 1) / a.ksplit;
  per = (per + 31) & ~(int64_t)31;
  const int64_t mbeg = (int64_t)z * per;
  const int64_t mend = mbeg + per < a.M ? mbeg + per : a.M;

  f32x4 xreg[RX], yreg[RY];
  float xmask[RX], ymask[RY];
  auto load = [&](int64_t m1) {
#pragma unroll
    for (int q = 0; q < RX; ++q) {
      const int idx = tid + NT * q;
      const int px = idx / (BA / 4), c4 = idx % (BA / 4);
      const int64_t m = m1 + px;
      const uint32_t p = (uint32_t)(m & (((int64_t)1 << logHW) - 1));
      const int yy = (int)morton_y(p) + dy, xx = (int)morton_x(p) + dx;
      const bool ok = m < mend && yy >= 0 && yy < H && xx >= 0 && xx < H;
      uint32_t src = morton((uint32_t)yy, (uint32_t)xx);
      if (a.ups) src >>= 2;
      const float* ptr = a.in + (ok ? ((m >> logHW) * HWin + src) * a.Cin + a0 + c4 * 4 : 0);
      xreg[q] = *reinterpret_cast<const f32x4*>(ptr);  // unconditional load; 0/1 mask applied at the LDS store
      xmask[q] = ok ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < RY; ++q) {
      const int idx = tid + NT * q;
      const int px = idx / (BB / 4), c4 = idx % (BB / 4);
      const int64_t m = m1 + px;
      const bool oky = m < mend;
      yreg[q] = *reinterpret_cast<const f32x4*>(a.dy + (oky ? m * a.Cout + b0 + c4 * 4 : 0));
      ymask[q] = oky ? 1.0f : 0.0f;
    }
  };
  f32x16 acc = {0};
  if (mbeg < mend) load(mbeg);
  for (int64_t m1 = mbeg; m1 < mend; m1 += 32) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RX; ++q) {
      const int idx = tid + NT * q;
      *reinterpret_cast<f32x4*>(Xs + (idx / (BA / 4)) * XP + (idx % (BA / 4)) * 4) = xreg[q] * xmask[q];
    }
#pragma unroll
    for (int q = 0; q < RY; ++q) {
      const int idx = tid + NT * q;
      *reinterpret_cast<f32x4*>(Ys + (idx / (BB / 4)) * YP + (idx % (BB / 4)) * 4) = yreg[q] * ymask[q];
    }
    __syncthreads();
    if (m1 + 32 < mend) load(m1 + 32);
    const float* xp = Xs + hh * XP + wa * 32 + c;
    const float* yp = Ys + hh * YP + wb * 32 + c;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[2 * s * XP], yp[2 * s * YP], acc, 0, 0, 0);
  }
  float* dst = a.slabs + (((size_t)z * a.ntaps + tap) * a.Cin + a0 + wa * 32) * a.Cout + b0 + wb * 32 + c;
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[(size_t)crow16(r, hh) * a.Cout] = acc[r];
}

// ------------------------------------------------------------------------------------------ wgrad, 3x3
// All 9 taps in one block.  A 32-pixel chunk of the output (Morton-aligned: an 8x4 patch, or whole small images)
// needs input pixels only from that patch plus a one-pixel halo, so the patch is staged ONCE per chunk
// ("slots": (ph+2) x (pw+2) rows per image) and every tap reads its shifted rows from LDS:
//     dW[tap][ci][co] += sum_k Xs[slot(k) + shift(tap)][ci] * dYs[k][co]
// dY is loaded once per chunk (not once per tap), the input once (not 9 gathers), and there are 9x fewer
// split-K slabs to reduce.  9 accumulator tiles per wave (144 registers).
template <int WA, int WB, int WT>
__global__ __launch_bounds__(WA* WB* WT * 64) void conv_wgrad9_kernel(WgradArgs a) {
  // waves: WA x WB sub-tiles of the (BA x BB) channel tile, times WT tap groups (wave wt owns taps wt, wt+WT, ...)
  constexpr int NT = WA * WB * WT * 64, BA = 32 * WA, BB = 32 * WB;
  constexpr int NACC = (9 + WT - 1) / WT;
  constexpr int XP = BA + 4, YP = BB + 4;
  constexpr int SMAX = 128;  // slots per chunk: 60 (H >= 8), 72 (H = 4), 128 (H = 2)
  __shared__ __align__(16) float Xs[SMAX * XP];
  __shared__ __align__(16) float Ys[32 * YP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // wave-uniform by construction; readfirstlane tells the compiler so (otherwise `if (t < 9)` below becomes an
  // exec-masked waterfall around every MFMA -- seen in the ISA)
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int wt = wv / (WA * WB), wa = (wv / WB) % WA, wb = wv % WB, hh = lane >> 5, c = lane & 31;
  const int tiles_b = a.Cout / BB;
  const int a0 = (blockIdx.x / tiles_b) * BA, b0 = (blockIdx.x % tiles_b) * BB;
  const int z = blockIdx.y;
  const int L = a.L, H = 1 << L, logHW = 2 * L, HW = 1 << logHW;
  const int64_t HWin = a.ups ? (int64_t)(HW >> 2) : (int64_t)HW;
  // chunk geometry
  const int ph = H < 4 ? H : 4, pw = H < 8 ? H : 8;       // patch of one image inside a 32-pixel chunk
  const int SW = pw + 2, SP = (ph + 2) * SW;              // slots per image (with halo)
  const int pix_per_img = HW < 32 ? HW : 32;
  const int nimg = 32 / pix_per_img;
  const int S = nimg * SP;
  int64_t per = (a.M + a.ksplit - 1) / a.ksplit;
  per = (per + 31) & ~(int64_t)31;
  const int64_t mbeg = (int64_t)z * per;
  const int64_t mend = mbeg + per < a.M ? mbeg + per : a.M;

  // slot of pixel k of a chunk: identical for every chunk (Morton-aligned), so computed once per lane
  int base_slot[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = 2 * s + hh;
    const uint32_t p = (uint32_t)(k & (pix_per_img - 1));
    base_slot[s] = (k / pix_per_img) * SP + ((int)morton_y(p) + 1) * SW + (int)morton_x(p) + 1;
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = (f32x16){0};

  // Same one-load-site software pipeline as the forward kernel: iteration `it` issues the global loads of chunk
  // `it` into registers, runs the 144 MFMAs of chunk it-1 out of LDS while they fly, then parks chunk `it` in LDS.
  constexpr int RX = (SMAX * (BA / 4) + NT - 1) / NT;  // 8: worst case (2x2 images: 128 slots)
  constexpr int RY = 32 * (BB / 4) / NT;               // 2
  const int nx = S * (BA / 4);
  const int64_t nchunks = mbeg < mend ? (mend - mbeg + 31) / 32 : 0;
  // which slot / channel group each of this thread's staging loads serves: fixed for the whole kernel
  int q_il[RX], q_dy[RX], q_dx[RX];
  uint32_t q_col[RX];
  bool q_live[RX];
#pragma unroll
  for (int q = 0; q < RX; ++q) {
    const int e = tid + NT * q;
    const int slot = e / (BA / 4), c4 = e % (BA / 4);
    const int il = slot / SP, r = slot - il * SP;
    q_il[q] = il; q_dy[q] = r / SW - 1; q_dx[q] = r % SW - 1;
    q_col[q] = (uint32_t)(a0 + c4 * 4);
    q_live[q] = e < nx;
  }
#ifdef DVG_STAMP
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, phs[5] = {0, 0, 0, 0, 0};
#endif
  for (int64_t it = 0; it <= nchunks; ++it) {
    f32x4 xreg[RX], yreg[RY];
    float xmask[RX], ymask[RY];
    STAMP(t0);
    if (it < nchunks) {
      const int64_t m1 = mbeg + it * 32;
      const int64_t img0 = m1 >> logHW;
      const uint32_t p0 = (uint32_t)(m1 & (HW - 1));
      const int y0 = (int)morton_y(p0), x0 = (int)morton_x(p0);
#pragma unroll
      for (int q = 0; q < RX; ++q) {
        const int y = y0 + q_dy[q], x = x0 + q_dx[q];
        const int64_t img = img0 + q_il[q];
        const bool ok = q_live[q] && y >= 0 && y < H && x >= 0 && x < H && img * HW < a.M;
        uint32_t src = morton((uint32_t)y, (uint32_t)x);
        if (a.ups) src >>= 2;
        // unconditional load from a valid address; the 0/1 mask is applied at the LDS store (see the forward kernel)
        xreg[q] = *reinterpret_cast<const f32x4*>(a.in + (ok ? (img * HWin + src) * a.Cin + q_col[q] : 0));
        xmask[q] = ok ? 1.0f : 0.0f;
      }
#pragma unroll
      for (int q = 0; q < RY; ++q) {
        const int e = tid + NT * q;
        const int px = e / (BB / 4), c4 = e % (BB / 4);
        const int64_t m = m1 + px;
        const bool ok = m < mend;
        yreg[q] = *reinterpret_cast<const f32x4*>(a.dy + (ok ? m * a.Cout + b0 + c4 * 4 : 0));
        ymask[q] = ok ? 1.0f : 0.0f;
      }
    }
    STAMP(t1);
    if (it > 0) {
      const float* xp = Xs + wa * 32 + c;
      const float* yp = Ys + hh * YP + wb * 32 + c;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float bv = yp[2 * s * YP];
        const float* xr = xp + base_slot[s] * XP;
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
          const int t = wt + WT * j;
          if (j < NACC - 1 || t < 9) {  // only the last tap of a group can be out of range (wave-uniform test)
            const int shift = (t / 3 - 1) * SW + (t % 3 - 1);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[shift * XP], bv, acc[j], 0, 0, 0);
          }
        }
      }
    }
    STAMP(t2);
    __syncthreads();
    STAMP(t3);
    if (it < nchunks) {
#pragma unroll
      for (int q = 0; q < RX; ++q) {
        const int e = tid + NT * q;
        if (e < nx) *reinterpret_cast<f32x4*>(Xs + (e / (BA / 4)) * XP + (e % (BA / 4)) * 4) = xreg[q] * xmask[q];
      }
#pragma unroll
      for (int q = 0; q < RY; ++q) {
        const int e = tid + NT * q;
        *reinterpret_cast<f32x4*>(Ys + (e / (BB / 4)) * YP + (e % (BB / 4)) * 4) = yreg[q] * ymask[q];
      }
    }
    STAMP(t4);
    __syncthreads();
    STAMP(t5);
#ifdef DVG_STAMP
    phs[0] += t1 - t0; phs[1] += t2 - t1; phs[2] += t3 - t2; phs[3] += t4 - t3; phs[4] += t5 - t4;
#endif
  }
#ifdef DVG_STAMP
  if (tid == 0) {  // debug words live just past the slabs (the diagnostic harness allocates the room)
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.slabs + (size_t)a.ksplit * 9 * a.Cin * a.Cout) +
                              ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    for (int k = 0; k < 5; ++k) dbg[k] = phs[k];
    dbg[5] = (unsigned long long)nchunks;
  }
#endif
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    const int t = wt + WT * j;
    if (t < 9) {
      float* dst = a.slabs + (((size_t)z * 9 + t) * a.Cin + a0 + wa * 32) * a.Cout + b0 + wb * 32 + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(size_t)crow16(r, hh) * a.Cout] = acc[j][r];
    }
  }
}

// ------------------------------------------------------------------------------------------ wgrad, folded upsample
// Weight gradient of Upsample(x2) + ConvTranspose 3x3 in the folded form (conv.h: ConvArgs.fold):
//     dWf[cls][t][ci][co] = sum_q X[q + off(cls, t)][ci] * dY[4q + cls][co],   off = (pa-1+dr, pb-1+dc)
// over SOURCE pixels q: 16 (class, tap) pairs x 1/4 of the pixels = 4/9 of the 9-tap FLOPs.  A chunk is 32 source
// pixels: their haloed patch of X is staged once (as in conv_wgrad9_kernel) together with the 128 CONTIGUOUS dY rows
// of their 4 output pixels each (Morton order).  One wave per parity class: it needs one dY row and 4 shifted X rows
// per pixel pair, 4 accumulator tiles.
__global__ __launch_bounds__(256) void conv_wgrad_fold_kernel(WgradArgs a) {
  constexpr int NT = 256, BA = 32, BB = 32, XP = BA + 4, YP = BB + 4, SMAX = 128;
  __shared__ __align__(16) float Xs[SMAX * XP];
  __shared__ __align__(16) float Ys[128 * YP];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, c = lane & 31;
  const int cls = __builtin_amdgcn_readfirstlane(tid >> 6), pa = cls >> 1, pb = cls & 1;
  const int tiles_b = a.Cout / BB;
  const int a0 = (blockIdx.x / tiles_b) * BA, b0 = (blockIdx.x % tiles_b) * BB;
  const int z = blockIdx.y;
  const int L = a.L, H = 1 << L, logHW = 2 * L, HW = 1 << logHW;
  const int ph = H < 4 ? H : 4, pw = H < 8 ? H : 8;
  const int SW = pw + 2, SP = (ph + 2) * SW;
  const int pix_per_img = HW < 32 ? HW : 32;
  const int nimg = 32 / pix_per_img;
  const int S = nimg * SP;
  int64_t per = (a.M + a.ksplit - 1) / a.ksplit;
  per = (per + 31) & ~(int64_t)31;
  const int64_t mbeg = (int64_t)z * per;
  const int64_t mend = mbeg + per < a.M ? mbeg + per : a.M;

  int base_slot[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = 2 * s + hh;
    const uint32_t p = (uint32_t)(k & (pix_per_img - 1));
    base_slot[s] = (k / pix_per_img) * SP + ((int)morton_y(p) + 1) * SW + (int)morton_x(p) + 1;
  }
  int shift[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) shift[t] = ((pa - 1 + (t >> 1)) * SW + (pb - 1 + (t & 1))) * XP;
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x16){0};

  constexpr int RX = SMAX * (BA / 4) / NT;  // 4
  constexpr int RY = 128 * (BB / 4) / NT;   // 4
  const int nx = S * (BA / 4);
  const int64_t nchunks = mbeg < mend ? (mend - mbeg + 31) / 32 : 0;
  int q_il[RX], q_dy[RX], q_dx[RX];
  uint32_t q_col[RX];
  bool q_live[RX];
#pragma unroll
  for (int q = 0; q < RX; ++q) {
    const int e = tid + NT * q;
    const int slot = e / (BA / 4), c4 = e % (BA / 4);
    const int il = slot / SP, r = slot - il * SP;
    q_il[q] = il; q_dy[q] = r / SW - 1; q_dx[q] = r % SW - 1;
    q_col[q] = (uint32_t)(a0 + c4 * 4);
    q_live[q] = e < nx;
  }
  for (int64_t it = 0; it <= nchunks; ++it) {
    f32x4 xreg[RX], yreg[RY];
    float xmask[RX], ymask[RY];
    if (it < nchunks) {
      const int64_t m1 = mbeg + it * 32;
      const int64_t img0 = m1 >> logHW;
      const uint32_t p0 = (uint32_t)(m1 & (HW - 1));
      const int y0 = (int)morton_y(p0), x0 = (int)morton_x(p0);
#pragma unroll
      for (int q = 0; q < RX; ++q) {
        const int y = y0 + q_dy[q], x = x0 + q_dx[q];
        const int64_t img = img0 + q_il[q];
        const bool ok = q_live[q] && y >= 0 && y < H && x >= 0 && x < H && img * HW < a.M;
        const uint32_t src = morton((uint32_t)y, (uint32_t)x);
        xreg[q] = *reinterpret_cast<const f32x4*>(a.in + (ok ? (img * HW + src) * a.Cin + q_col[q] : 0));
        xmask[q] = ok ? 1.0f : 0.0f;
      }
#pragma unroll
      for (int q = 0; q < RY; ++q) {
        const int e = tid + NT * q;
        const int row = e / (BB / 4), c4 = e % (BB / 4);
        const int64_t m = 4 * m1 + row;  // output pixel: 4 * source pixel + class
        const bool ok = m < 4 * mend;
        yreg[q] = *reinterpret_cast<const f32x4*>(a.dy + (ok ? m * a.Cout + b0 + c4 * 4 : 0));
        ymask[q] = ok ? 1.0f : 0.0f;
      }
    }
    if (it > 0) {
      const float* xp = Xs + c;
      const float* yp = Ys + cls * YP + c;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float bv = yp[4 * (2 * s + hh) * YP];
        const float* xr = xp + base_slot[s] * XP;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[shift[t]], bv, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
    if (it < nchunks) {
#pragma unroll
      for (int q = 0; q < RX; ++q) {
        const int e = tid + NT * q;
        if (e < nx) *reinterpret_cast<f32x4*>(Xs + (e / (BA / 4)) * XP + (e % (BA / 4)) * 4) = xreg[q] * xmask[q];
      }
#pragma unroll
      for (int q = 0; q < RY; ++q) {
        const int e = tid + NT * q;
        *reinterpret_cast<f32x4*>(Ys + (e / (BB / 4)) * YP + (e % (BB / 4)) * 4) = yreg[q] * ymask[q];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float* dst = a.slabs + (((size_t)z * 16 + cls * 4 + t) * a.Cin + a0) * a.Cout + b0 + c;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(size_t)crow16(r, hh) * a.Cout] = acc[t][r];
  }
}

int wgrad_fold_ksplit(int64_t Msrc, int Cin, int Cout) {
  const int64_t tiles = (int64_t)(Cin / 32) * (Cout / 32);
  int64_t k = 512 / tiles;  // two 4-wave blocks are resident per CU (4 accumulator tiles per wave)
  const int64_t kmax = ceil_div(Msrc, 64);
  if (k > kmax) k = kmax;
  if (k > 256) k = 256;
  return (int)(k < 1 ? 1 : k);
}

// Sums the slabs in order AND the 4 parity classes back onto the 9 taps; scatters into the ConvTranspose2d layout.
__global__ __launch_bounds__(256) void wgrad_fold_reduce_kernel(const float* __restrict__ slabs, int ksplit, int Cin, int Cout,
                                                                float* __restrict__ grad_w) {
  const int64_t plane = (int64_t)Cin * Cout, total = 9 * plane;
  const int sub = threadIdx.x & 7;
  for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; e < total; e += ((int64_t)gridDim.x * 256) >> 3) {
    const int tap = (int)(e / plane);
    const int64_t ab = e - (int64_t)tap * plane;  // ci * Cout + co
    const int r = tap / 3, sx = tap % 3;
    float s = 0.f;
    for (int k = sub; k < ksplit; k += 8) {
      const float* sl = slabs + (size_t)k * 16 * plane + ab;
#pragma unroll
      for (int cls = 0; cls < 4; ++cls)
        s += sl[(size_t)(cls * 4 + fold_src(cls >> 1, r) * 2 + fold_src(cls & 1, sx)) * plane];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0) grad_w[ab * 9 + (8 - tap)] = s;
  }
}

int wgrad_ksplit(int64_t M, int Cin, int Cout, int ntaps) {
  const int ba = (Cin % 64 == 0) ? 64 : 32, bb = (Cout % 64 == 0) ? 64 : 32;
  const int64_t tiles = (int64_t)(Cin / ba) * (Cout / bb);
  int64_t k;
  if (ntaps == 9) {
    // conv_wgrad9_kernel keeps 9 accumulator tiles per wave (> 256 registers): one wave per SIMD, i.e.
    // 4 / (waves per block) blocks per CU.  Size the split so that ALL blocks are resident at once (a second,
    // partial round of blocks would double the kernel's time), and keep >= 64 pixels per slab.
    // blocks are always 4 waves (channel sub-tiles x tap groups); with 9 or 5 accumulator tiles per wave only
    // one block is resident per CU, with 3 (32x32 channel tile) two are
    const int64_t resident = (ba == 32 && bb == 32) ? 512 : 256;
    k = resident / tiles;
    const int64_t kmax = ceil_div(M, 64);
    if (k > kmax) k = kmax;
  } else {
    k = ceil_div(1024, tiles * ntaps);
    const int64_t kmax = ceil_div(M, 256);
    if (k > kmax) k = kmax;
  }
  if (k > 256) k = 256;
  if (k < 1) k = 1;
  return (int)k;
}

int launch_conv_wgrad(const WgradArgs& a, hipStream_t s) {
  if (a.Cin % 32 || a.Cout % 32 || a.M <= 0 || a.ksplit < 1) {
    set_error("conv_wgrad: unsupported shape Cin=%d Cout=%d M=%lld", a.Cin, a.Cout, (long long)a.M);
    return DVG_E_INVALID;
  }
  if (a.fold) {
    if (a.ntaps != 16 || a.ups) { set_error("conv_wgrad: fold needs ntaps=16, ups=0"); return DVG_E_INVALID; }
    const double fl = 2.0 * (double)a.M * a.Cin * a.Cout * 16;  // executed FLOPs (4/9 of the 9-tap form)
    DVG_LAUNCH_WORK(K_WGRAD_FOLD, fl, conv_wgrad_fold_kernel, dim3((unsigned)((a.Cin / 32) * (a.Cout / 32)), (unsigned)a.ksplit),
                    dim3(256), 0, s, a);
    return DVG_OK;
  }
  const bool a64 = a.Cin % 64 == 0, b64 = a.Cout % 64 == 0;
  const int ba = a64 ? 64 : 32, bb = b64 ? 64 : 32;
  const double flops = 2.0 * (double)a.M * a.Cin * a.Cout * a.ntaps;
  if (a.ntaps == 9) {
    const dim3 g9((unsigned)((a.Cin / ba) * (a.Cout / bb)), (unsigned)a.ksplit);
    // 64x64 channel tile: two tap groups (8 waves, 5 / 4 accumulator tiles per wave, two waves per SIMD) instead of one
    // (4 waves x 9 tiles, one per SIMD): +10 % on the kernel in-situ (c3 53 -> 59, c2 41 -> 45 TFLOP/s), steps neutral
    // to -1 %.  DVG_WGRAD9_WT2=0 restores the 4-wave form (A/B runs).
    static const bool wt2 = [] { const char* e = getenv("DVG_WGRAD9_WT2"); return !e || e[0] != '0'; }();
    if (a64 && b64 && wt2) DVG_LAUNCH_WORK(K_WGRAD_2x2, flops, (conv_wgrad9_kernel<2, 2, 2>), g9, dim3(512), 0, s, a);
    else if (a64 && b64) DVG_LAUNCH_WORK(K_WGRAD_2x2, flops, (conv_wgrad9_kernel<2, 2, 1>), g9, dim3(256), 0, s, a);
    else if (a64) DVG_LAUNCH_WORK(K_WGRAD_2x1, flops, (conv_wgrad9_kernel<2, 1, 2>), g9, dim3(256), 0, s, a);
    else if (b64) DVG_LAUNCH_WORK(K_WGRAD_1x2, flops, (conv_wgrad9_kernel<1, 2, 2>), g9, dim3(256), 0, s, a);
    else DVG_LAUNCH_WORK(K_WGRAD_1x1, flops, (conv_wgrad9_kernel<1, 1, 4>), g9, dim3(256), 0, s, a);
    return DVG_OK;
  }
  const dim3 grid((unsigned)((a.Cin / ba) * (a.Cout / bb)), (unsigned)a.ntaps, (unsigned)a.ksplit);
  if (a64 && b64) DVG_LAUNCH_WORK(K_WGRAD_2x2, flops, (conv_wgrad_kernel<2, 2>), grid, dim3(256), 0, s, a);
  else if (a64) DVG_LAUNCH_WORK(K_WGRAD_2x1, flops, (conv_wgrad_kernel<2, 1>), grid, dim3(128), 0, s, a);
  else if (b64) DVG_LAUNCH_WORK(K_WGRAD_1x2, flops, (conv_wgrad_kernel<1, 2>), grid, dim3(128), 0, s, a);
  else DVG_LAUNCH_WORK(K_WGRAD_1x1, flops, (conv_wgrad_kernel<1, 1>), grid, dim3(64), 0, s, a);
  return DVG_OK;
}

// 8 lanes cooperate on one element (strided over the slabs, fixed-shape shuffle tree: deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int ksplit, WeightMap map,
                                                           float* __restrict__ grad_w) {
  const int64_t total = (int64_t)map.ntaps * map.Ca * map.Cb;
  const int sub = threadIdx.x & 7;
  for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; e < total; e += ((int64_t)gridDim.x * 256) >> 3) {
    float s = 0.f;
    for (int k = sub; k < ksplit; k += 8) s += slabs[(size_t)k * total + e];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0) {
      const int b = (int)(e % map.Cb);
      const int a = (int)((e / map.Cb) % map.Ca);
      const int tap = (int)(e / ((int64_t)map.Cb * map.Ca));
      grad_w[torch_weight_offset(map, tap, a, b)] = s;
    }
  }
}

// one packed entry in the format `fmt`: 0 = f32 [tap][a][b] (index e); 1 = bf16 K-major [tap][b][a] (index kmaj);
// 2 = three bf16 K-major planes, the exact pieces hi + mid + lo of the weight; 3 = f32 K-major
__device__ __forceinline__ void pack_store(float* wp, int64_t total, int64_t kmaj, int64_t e, float v, int fmt) {
  if (fmt == 0) { wp[e] = v; return; }
  if (fmt == 3) { wp[kmaj] = v; return; }  // f32 K-major [tap][b][a]: the LDS-DMA form of the float32 GEMM kernel
  uint16_t* w16 = reinterpret_cast<uint16_t*>(wp);
  const uint16_t hi = f32_to_bf16_rne(v);
  w16[kmaj] = hi;
  if (fmt == 2) {
    const float r1 = v - bf16_to_f32(hi);
    const uint16_t mid = f32_to_bf16_rne(r1);
    w16[total + kmaj] = mid;
    w16[2 * total + kmaj] = f32_to_bf16_rne(r1 - bf16_to_f32(mid));
  }
}

__global__ __launch_bounds__(256) void weight_pack_kernel(const float* __restrict__ w, WeightMap map, float* __restrict__ wp,
                                                          int bf16t) {
  const int64_t total = (int64_t)map.ntaps * map.Ca * map.Cb;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e % map.Cb);
    const int a = (int)((e / map.Cb) % map.Ca);
    const int tap = (int)(e / ((int64_t)map.Cb * map.Ca));
    const float v = packed_weight(w, map, tap, a, b);
    pack_store(wp, total, ((int64_t)tap * map.Cb + b) * map.Ca + a, e, v, bf16t);
  }
}

struct PackJobs { PackJob job[MAX_PACK_JOBS]; };

__global__ __launch_bounds__(256) void weight_pack_multi_kernel(PackJobs jobs) {
  const PackJob& j = jobs.job[blockIdx.y];
  const WeightMap map = j.map;
  const int64_t total = (int64_t)map.ntaps * map.Ca * map.Cb;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e % map.Cb);
    const int a = (int)((e / map.Cb) % map.Ca);
    const int tap = (int)(e / ((int64_t)map.Cb * map.Ca));
    const float v = packed_weight(j.w, map, tap, a, b);
    pack_store(j.wp, total, ((int64_t)tap * map.Cb + b) * map.Ca + a, e, v, j.bf16t);
  }
}

int launch_weight_pack_multi(const PackJob* jobs, int njobs, hipStream_t s) {
  if (njobs < 1 || njobs > MAX_PACK_JOBS) { set_error("weight_pack_multi: %d jobs", njobs); return DVG_E_INVALID; }
  PackJobs pj;
  int64_t biggest = 0;
  for (int k = 0; k < njobs; ++k) {
    pj.job[k] = jobs[k];
    // every pack feeds launch_conv_igemm: same format decision as that launch will make
    pj.job[k].bf16t = jobs[k].rows > 0 ? conv_launch_mode(jobs[k].rows, jobs[k].map.Cb) : conv_precision_mode();
    const int64_t t = (int64_t)jobs[k].map.ntaps * jobs[k].map.Ca * jobs[k].map.Cb;
    if (t > biggest) biggest = t;
  }
  int64_t gx = ceil_div(biggest, 256 * 4);
  if (gx > 256) gx = 256;
  if (gx < 1) gx = 1;
  DVG_LAUNCH(K_WEIGHT_PACK, weight_pack_multi_kernel, dim3((unsigned)gx, (unsigned)njobs), dim3(256), 0, s, pj);
  return DVG_OK;
}

static unsigned ew_grid(int64_t n) {
  const int64_t b = ceil_div(n, 256);
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int launch_wgrad_reduce(const float* slabs, int ksplit, const WeightMap& map, float* grad_w, hipStream_t s) {
  const int64_t total = (int64_t)map.ntaps * map.Ca * map.Cb;
  DVG_LAUNCH(K_WGRAD_REDUCE, wgrad_reduce_kernel, dim3(ew_grid(total * 8)), dim3(256), 0, s, slabs, ksplit, map, grad_w);
  return DVG_OK;
}

int launch_wgrad_fold_reduce(const float* slabs, int ksplit, int Cin, int Cout, float* grad_w, hipStream_t s) {
  DVG_LAUNCH(K_WGRAD_REDUCE, wgrad_fold_reduce_kernel, dim3(ew_grid((int64_t)9 * Cin * Cout * 8)), dim3(256), 0, s, slabs, ksplit,
             Cin, Cout, grad_w);
  return DVG_OK;
}

int launch_weight_pack(const float* w, const WeightMap& map, float* wp, hipStream_t s, int bf16t) {
  if (bf16t < 0) bf16t = conv_precision_mode();
  const int64_t total = (int64_t)map.ntaps * map.Ca * map.Cb;
  DVG_LAUNCH(K_WEIGHT_PACK, weight_pack_kernel, dim3(ew_grid(total)), dim3(256), 0, s, w, map, wp, bf16t);
  return DVG_OK;
}

}  // namespace dvg
