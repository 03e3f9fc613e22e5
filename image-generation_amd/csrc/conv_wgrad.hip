// Weight-gradient GEMMs of the convolution layers (dW[tap] = X_tap^T dY, split-K over pixels into slabs), the
// fixed-order slab reduction back into the checkpoint layout, and the weight packing kernels
// (/root/reference/src/encoder.py:28-30, /root/reference/src/decoder.py:28, :34-38).
// Separate translation unit from conv_igemm.hip so the two can carry different code-generation options (Makefile).
#include <cstdlib>
#include <type_traits>

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

// ------------------------------------------------------------------------------------------ wgrad, 3x3, LDS-DMA
// conv_wgrad9_kernel with the staging done by `buffer_load ... lds` (as the LDS-DMA form of the forward kernel,
// conv_igemm.hip): no staging registers, no ds_write pass, no mask multiply (slots outside the image and rows past the
// slab carry an out-of-range offset: the hardware writes zeros), two LDS stages and ONE barrier per 32-pixel chunk.  The
// per-lane source offsets of chunk t+2 are computed in Morton space (masked adds from the chunk's origin, no
// decode / re-encode) between the MFMAs of chunk t.  Needs both tensors below 0xFFFF0000 bytes (32-bit offsets).
// FOLD = true: the folded Upsample(x2) + 3x3 layer (conv_wgrad_fold_kernel's job: <1, 1, 4>, one wave per output
// parity class, 4 shifted rows of the SOURCE map per class, the chunk's 128 contiguous dY rows staged with it).
template <int WA, int WB, int WT, bool FOLD = false>
__global__ __launch_bounds__(WA* WB* WT * 64) void conv_wgrad9_dma_kernel(WgradArgs a) {
  static_assert(!FOLD || (WA == 1 && WB == 1 && WT == 4), "folded form: one 32x32 channel tile, one wave per class");
  constexpr int NT = WA * WB * WT * 64, NW = NT / 64, BA = 32 * WA, BB = 32 * WB;
  constexpr int NACC = FOLD ? 4 : (9 + WT - 1) / WT;
  constexpr int SMAX = 128;                              // slots per chunk: 60 (H >= 8), 72 (H = 4), 128 (H = 2)
  constexpr int YROWS = FOLD ? 128 : 32;                 // dY rows per chunk (folded: 4 output pixels per source pixel)
  constexpr int XB = SMAX * BA * 4, YB = YROWS * BB * 4, STAGE = XB + YB;
  constexpr int RXD = SMAX * (BA / 4) / 64 / NW;         // 1 KiB pieces of the patch per wave (worst case)
  constexpr int NYI = YROWS * (BB / 4) / 64;             // 1 KiB pieces of the dY rows
  constexpr int RYD = (NYI + NW - 1) / NW;               // ... per wave (piece wv + NW q, if < NYI)
  static_assert(RXD * NW * 64 == SMAX * (BA / 4), "LDS-DMA pieces must divide the stage");
  typedef __attribute__((address_space(3))) void lds_void;
  extern __shared__ __align__(16) unsigned char wg_smem[];  // 2 * STAGE bytes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wt = wv / (WA * WB), wa = (wv / WB) % WA, wb = wv % WB, hh = lane >> 5, c = lane & 31;
  const int tiles_b = a.Cout / BB;
  const int a0 = (blockIdx.x / tiles_b) * BA, b0 = (blockIdx.x % tiles_b) * BB;
  const int z = blockIdx.y;
  const int L = a.L, H = 1 << L, logHW = 2 * L, HW = 1 << logHW;
  const uint32_t HWin = a.ups ? (uint32_t)(HW >> 2) : (uint32_t)HW;
  const int ph = H < 4 ? H : 4, pw = H < 8 ? H : 8;
  const int SW = pw + 2, SP = (ph + 2) * SW;
  const int pix_per_img = HW < 32 ? HW : 32;
  const int nimg = 32 / pix_per_img;
  const int S = nimg * SP;
  int64_t per = (a.M + a.ksplit - 1) / a.ksplit;
  per = (per + 31) & ~(int64_t)31;
  const int64_t mbeg = (int64_t)z * per;
  const int64_t mend = mbeg + per < a.M ? mbeg + per : a.M;
  const int nchunks = mbeg < mend ? (int)((mend - mbeg + 31) / 32) : 0;

  int base_slot[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = 2 * s + hh;
    const uint32_t p = (uint32_t)(k & (pix_per_img - 1));
    base_slot[s] = ((k / pix_per_img) * SP + ((int)morton_y(p) + 1) * SW + (int)morton_x(p) + 1) * BA;
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = (f32x16){0};

  const uint32_t row_bytes = (uint32_t)a.Cin * 4u, yrow_bytes = (uint32_t)a.Cout * 4u;
  const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)0xFFFF0000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)0xFFFF0000u, 0x00020000);
  constexpr uint32_t XM = 0x55555555u, YM = 0xAAAAAAAAu, PAD = 0xFFFF0000u;
  // per-lane piece constants: piece q of this wave is 1 KiB piece j = wv + NW q of the patch image [slot][BA floats];
  // its lane covers 16-byte element e = 64 j + lane: slot e / (BA/4), channels 4 (e % (BA/4)) ...
  const int nxi = (S * (BA / 4) + 63) / 64;  // pieces in use (wave-uniform)
  uint32_t q_dxm[RXD], q_dym[RXD], q_const[RXD];  // Morton-space deltas from (y0-1, x0-1); image + channel byte offset
  int q_il[RXD];
  bool q_live[RXD];
#pragma unroll
  for (int q = 0; q < RXD; ++q) {
    const int e = (wv + NW * q) * 64 + lane;
    const int slot = e / (BA / 4), c4 = e % (BA / 4);
    const int il = slot / SP, r = slot - il * SP;
    q_il[q] = il;
    q_dym[q] = part1by1((uint32_t)(r / SW)) << 1;
    q_dxm[q] = part1by1((uint32_t)(r % SW));
    q_const[q] = (uint32_t)il * HWin * row_bytes + (uint32_t)(a0 + c4 * 4) * 4u;
    q_live[q] = slot < S;
  }
  int ypx[RYD];       // dY pieces: row and channel bytes of this lane
  uint32_t ycol[RYD];
#pragma unroll
  for (int q = 0; q < RYD; ++q) {
    const int e = (wv + NW * q) * 64 + lane;
    ypx[q] = e / (BB / 4);
    ycol[q] = (uint32_t)(b0 + (e % (BB / 4)) * 4) * 4u;
  }

  uint32_t xoff[RXD], yoff[RYD];  // offsets of the chunk to be issued next
  // scalar part of a chunk's addressing
  const int n_img = (int)((a.M + HW - 1) >> logHW);  // images (whole or partial) in the tensor
  struct Org { uint32_t bx, by, img_bytes; int64_t img0; };
  auto origin = [&](int t) -> Org {
    const int64_t m1 = mbeg + (int64_t)t * 32;
    const uint32_t p0 = (uint32_t)(m1 & (HW - 1));
    Org o;
    o.img0 = m1 >> logHW;
    // Morton images of x0 - 1 and y0 - 1 in two's complement over the masked bit lanes (-1 = all lanes set)
    o.bx = ((p0 & XM) - 1u) & XM;
    o.by = ((p0 & YM) - 2u) & YM;
    o.img_bytes = (uint32_t)o.img0 * HWin * row_bytes;
    return o;
  };
  auto calc_x = [&](const Org& o, int q) {
    const uint32_t nx = ((o.bx | YM) + q_dxm[q]) & XM, ny = ((o.by | XM) + q_dym[q]) & YM;
    const uint32_t pm = nx | ny;  // Morton index of the slot's pixel; bits at or above HW: outside the image
    // (bitwise, not short-circuit: no branches inside the MFMA stream)
    const bool ok = q_live[q] & ((pm >> logHW) == 0u) & ((int)o.img0 + q_il[q] < n_img);
    const uint32_t src = a.ups ? pm >> 2 : pm;
    xoff[q] = ok ? o.img_bytes + q_const[q] + src * row_bytes : PAD;
  };
  auto calc_y = [&](int t) {
#pragma unroll
    for (int q = 0; q < RYD; ++q) {
      const int64_t m = (mbeg + (int64_t)t * 32) * (FOLD ? 4 : 1) + ypx[q];  // (folded: output pixel 4 q + class)
      yoff[q] = m < mend * (FOLD ? 4 : 1) ? (uint32_t)m * yrow_bytes + ycol[q] : PAD;
    }
  };
  auto issue = [&](int buf) {
    unsigned char* dx = wg_smem + buf * STAGE;
#pragma unroll
    for (int q = 0; q < RXD; ++q)
      if (wv + NW * q < nxi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void*)(dx + (wv + NW * q) * 1024), 16, (int)xoff[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < RYD; ++q)
      if (wv + NW * q < NYI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (lds_void*)(dx + XB + (wv + NW * q) * 1024), 16, (int)yoff[q], 0, 0, 0);
  };
  auto calc_all = [&](int t) {
    const Org o = origin(t);
#pragma unroll
    for (int q = 0; q < RXD; ++q) calc_x(o, q);
    calc_y(t);
  };

  if (nchunks > 0) {
    calc_all(0);
    issue(0);
    if (nchunks > 1) calc_all(1);
  }
  int shiftw[NACC];  // LDS float offset of tap t's shifted row (wave-uniform)
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    const int t = wt + WT * j;
    // folded: class (pa, pb) = wave, tap j = (dr, dc) reads source pixel (i - 1 + pa + dr, j - 1 + pb + dc)
    shiftw[j] = FOLD ? (((wt >> 1) - 1 + (j >> 1)) * SW + ((wt & 1) - 1 + (j & 1))) * BA : ((t / 3 - 1) * SW + (t % 3 - 1)) * BA;
  }
  // The chunk loop, instantiated for the two tap counts a wave can have (its last tap may not exist: t = 9): a
  // wave-uniform choice made ONCE, so that the 16 k-steps are one straight line the reads can be pipelined through
  auto run = [&](auto nc) {
    constexpr int NA = decltype(nc)::value;
    for (int it = 0; it < nchunks; ++it) {
      const int buf = it & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (it + 1 < nchunks) issue(buf ^ 1);
      const Org o2 = origin(it + 2 < nchunks ? it + 2 : it);
      const float* xp = reinterpret_cast<const float*>(wg_smem + buf * STAGE) + wa * 32 + c;
      constexpr int YS = FOLD ? 8 * BB : 2 * BB;  // floats between the dY rows of consecutive k-steps
      const float* yp = reinterpret_cast<const float*>(wg_smem + buf * STAGE + XB) +
                        (FOLD ? (4 * hh + wt) * BB : hh * BB + wb * 32) + c;
      float bv[2], xv[2][NA];
      bv[0] = yp[0];
#pragma unroll
      for (int j = 0; j < NA; ++j) xv[0][j] = xp[base_slot[0] + shiftw[j]];
      __builtin_amdgcn_sched_group_barrier(0x100, NA + 1, 0);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < 16) {  // operands of k-step s+1: their LDS latency elapses under the MFMAs of step s
          bv[nxt] = yp[(s + 1) * YS];
#pragma unroll
          for (int j = 0; j < NA; ++j) xv[nxt][j] = xp[base_slot[s + 1] + shiftw[j]];
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[cur][j], bv[cur], acc[j], 0, 0, 0);
        if (s + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, NA + 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NA, 0);
        // the address arithmetic of chunk it+2 rides between the k-steps (the matrix pipe is busy for NA x 64 cycles)
        if (s >= 2 && s < 2 + RXD) calc_x(o2, s - 2);
        if (s == 2 + RXD) calc_y(it + 2);
      }
    }
  };
  if (FOLD || wt + WT * (NACC - 1) < 9) run(std::integral_constant<int, NACC>{});
  else run(std::integral_constant<int, NACC - 1>{});
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    const int t = FOLD ? j : wt + WT * j;
    if (t < 9) {
      float* dst = FOLD ? a.slabs + (((size_t)z * 16 + wt * 4 + j) * a.Cin + a0) * a.Cout + b0 + c
                        : a.slabs + (((size_t)z * 9 + t) * a.Cin + a0 + wa * 32) * a.Cout + b0 + wb * 32 + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(size_t)crow16(r, hh) * a.Cout] = acc[j][r];
    }
  }
}

// ------------------------------------------------------------------------------------------ wgrad, 1 tap, LDS-DMA
// dW[ci][co] = sum_m X[m][ci] dY[m][co] (the decoder's Linear layer): a plain TN GEMM over the rows.  128 x 128 channel
// tile per block, 4 waves with a 64 x 64 wave tile (4 accumulator tiles), 32 rows per stage by LDS-DMA, two stages, one
// barrier per stage; rows past the slab carry an out-of-range offset (zeros).
__global__ __launch_bounds__(256) void conv_wgrad1_dma_kernel(WgradArgs a) {
  constexpr int BA = 128, BB = 128, XB = 32 * BA * 4, STAGE = 2 * XB;  // X rows then dY rows, 16 KB each
  typedef __attribute__((address_space(3))) void lds_void;
  extern __shared__ __align__(16) unsigned char wg_smem[];  // 2 * STAGE bytes
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, c = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wa = wv >> 1, wb = wv & 1;
  const int tiles_b = a.Cout / BB;
  const int a0 = (blockIdx.x / tiles_b) * BA, b0 = (blockIdx.x % tiles_b) * BB;
  const int z = blockIdx.y;
  int64_t per = (a.M + a.ksplit - 1) / a.ksplit;
  per = (per + 31) & ~(int64_t)31;
  const int64_t mbeg = (int64_t)z * per;
  const int64_t mend = mbeg + per < a.M ? mbeg + per : a.M;
  const int nchunks = mbeg < mend ? (int)((mend - mbeg + 31) / 32) : 0;
  const uint32_t xrow_bytes = (uint32_t)a.Cin * 4u, yrow_bytes = (uint32_t)a.Cout * 4u;
  const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)0xFFFF0000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)0xFFFF0000u, 0x00020000);
  constexpr uint32_t PAD = 0xFFFF0000u;
  // stage image [32 rows][128 floats]: 16 pieces of 1 KiB (2 rows each), 4 per wave and operand
  int prow[4];
  uint32_t pcol[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = (wv * 4 + q) * 64 + lane;
    prow[q] = e >> 5;
    pcol[q] = (uint32_t)(e & 31) * 16u;
  }
  auto issue = [&](int t, int buf) {
    unsigned char* dst = wg_smem + buf * STAGE + wv * 4096;
    const int64_t m1 = mbeg + (int64_t)t * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t m = m1 + prow[q];
      const bool ok = m < mend;
      const uint32_t xo = ok ? (uint32_t)m * xrow_bytes + (uint32_t)a0 * 4u + pcol[q] : PAD;
      const uint32_t yo = ok ? (uint32_t)m * yrow_bytes + (uint32_t)b0 * 4u + pcol[q] : PAD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void*)(dst + q * 1024), 16, (int)xo, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (lds_void*)(dst + XB + q * 1024), 16, (int)yo, 0, 0, 0);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
  if (nchunks > 0) issue(0, 0);
  for (int it = 0; it < nchunks; ++it) {
    const int buf = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (it + 1 < nchunks) issue(it + 1, buf ^ 1);
    const float* xp = reinterpret_cast<const float*>(wg_smem + buf * STAGE) + hh * BA + wa * 64 + c;
    const float* yp = reinterpret_cast<const float*>(wg_smem + buf * STAGE + XB) + hh * BB + wb * 64 + c;
    float av[2][2], bv[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { av[0][i] = xp[i * 32]; bv[0][i] = yp[i * 32]; }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < 16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { av[nxt][i] = xp[2 * (s + 1) * BA + i * 32]; bv[nxt][i] = yp[2 * (s + 1) * BB + i * 32]; }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
      if (s + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* dst = a.slabs + ((size_t)z * a.Cin + a0 + wa * 64 + i * 32) * a.Cout + b0 + wb * 64 + j * 32 + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(size_t)crow16(r, hh) * a.Cout] = acc[i][j][r];
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

static bool wgrad_dma_on() { return opt(OPT_WGRAD_DMA) != 0; }  // (read per call: the tests flip it inside one process)
// the 1-tap form: 128 x 128 channel tiles (DMA kernel) when both channel counts allow and the tensors fit 32-bit offsets
static bool wgrad1_dma_ok(int64_t M, int Cin, int Cout) {
  // (and enough rows that the 128 x 128 tiles still fill the chip: c2's Linear layer has 2048 rows -- 32 such blocks)
  return wgrad_dma_on() && Cin % 128 == 0 && Cout % 128 == 0 && (double)M * Cin * 4.0 < 4294901760.0 &&
         (double)M * Cout * 4.0 < 4294901760.0 && (int64_t)(Cin / 128) * (Cout / 128) * ceil_div(M, 256) >= 256;
}

// Dense 2x2 form: slabs [ksplit][4 Cin][4 Cout]; grad_w (Cin,Cout,3,3)[ci][co][8 - t] = sum over the slabs (in order, 8
// lanes per element, fixed shuffle tree) of the blocks (p_in, p_out) whose displacement is forward tap t.
__global__ __launch_bounds__(256) void wgrad_d22_reduce_kernel(const float* __restrict__ slabs, int ksplit, int Cin, int Cout,
                                                              float* __restrict__ grad_w) {
  const int64_t plane = (int64_t)Cin * Cout, total = 9 * plane, srow = 4 * (int64_t)Cout, slab = 4 * (int64_t)Cin * srow;
  const int sub = threadIdx.x & 7;
  for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; e < total; e += ((int64_t)gridDim.x * 256) >> 3) {
    const int t = (int)(e / plane);
    const int64_t ab = e - (int64_t)t * plane;
    const int ci = (int)(ab / Cout), co = (int)(ab - (int64_t)ci * Cout);
    const int dy = t / 3 - 1, dx = t % 3 - 1;  // p_in = p_out + (dy, dx)
    float s = 0.f;
    for (int k = sub; k < ksplit; k += 8) {
      const float* sl = slabs + (size_t)k * slab;
#pragma unroll
      for (int po = 0; po < 4; ++po) {
        const int yi = (po >> 1) + dy, xi = (po & 1) + dx;
        if (yi >= 0 && yi < 2 && xi >= 0 && xi < 2) s += sl[((int64_t)((yi << 1 | xi) * Cin + ci)) * srow + po * Cout + co];
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0) grad_w[ab * 9 + (8 - t)] = s;
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
  } else if (ntaps == 1 && wgrad1_dma_ok(M, Cin, Cout)) {
    k = ceil_div(512, (int64_t)(Cin / 128) * (Cout / 128));  // two 4-wave blocks of 128 x 128 are resident per CU
    const int64_t kmax = ceil_div(M, 256);
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

template <int WA, int WB, int WT, bool FOLD = false>
static int launch_wgrad9_dma(int id, double flops, dim3 grid, const WgradArgs& a, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)(128 * 32 * WA * 4 + (FOLD ? 128 : 32) * 32 * WB * 4);
  auto kern = conv_wgrad9_dma_kernel<WA, WB, WT, FOLD>;
  static std::atomic<uint64_t> attr_done{0};  // one instantiation = one static
  if (lds > 64 * 1024) DVG_TRY(raise_dynamic_lds(attr_done, (const void*)kern, (int)lds));
  DVG_LAUNCH_WORK(id, flops, kern, grid, dim3(WA * WB * WT * 64), lds, s, a);
  return DVG_OK;
}

int launch_conv_wgrad(const WgradArgs& a, hipStream_t s) {
  if (a.Cin % 32 || a.Cout % 32 || a.M <= 0 || a.ksplit < 1) {
    set_error("conv_wgrad: unsupported shape Cin=%d Cout=%d M=%lld", a.Cin, a.Cout, (long long)a.M);
    return DVG_E_INVALID;
  }
  if (a.fold) {
    if (a.ntaps != 16 || a.ups) { set_error("conv_wgrad: fold needs ntaps=16, ups=0"); return DVG_E_INVALID; }
    const double fl = 2.0 * (double)a.M * a.Cin * a.Cout * 16;  // executed FLOPs (4/9 of the 9-tap form)
    if (wgrad_dma_on() && (double)a.M * a.Cin * 4.0 < 4294901760.0 && (double)a.M * 4.0 * a.Cout * 4.0 < 4294901760.0)
      return launch_wgrad9_dma<1, 1, 4, true>(K_WGRAD_FOLD, fl, dim3((unsigned)((a.Cin / 32) * (a.Cout / 32)), (unsigned)a.ksplit), a, s);
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
    // to -1 %.  (The 4-wave form is still instantiated for the shapes that need it.)
    constexpr bool wt2 = true;
    // LDS-DMA form (32-bit offsets: both tensors below 0xFFFF0000 bytes); option wgrad_dma = 0: the register-staged form
    const double xbytes = (double)(a.ups ? a.M / 4 : a.M) * a.Cin * 4.0, ybytes = (double)a.M * a.Cout * 4.0;
    if (wgrad_dma_on() && xbytes < 4294901760.0 && ybytes < 4294901760.0) {
      if (a64 && b64) return launch_wgrad9_dma<2, 2, 2>(K_WGRAD_2x2, flops, g9, a, s);
      if (a64) return launch_wgrad9_dma<2, 1, 2>(K_WGRAD_2x1, flops, g9, a, s);
      if (b64) return launch_wgrad9_dma<1, 2, 2>(K_WGRAD_1x2, flops, g9, a, s);
      return launch_wgrad9_dma<1, 1, 4>(K_WGRAD_1x1, flops, g9, a, s);
    }
    if (a64 && b64 && wt2) DVG_LAUNCH_WORK(K_WGRAD_2x2, flops, (conv_wgrad9_kernel<2, 2, 2>), g9, dim3(512), 0, s, a);
    else if (a64 && b64) DVG_LAUNCH_WORK(K_WGRAD_2x2, flops, (conv_wgrad9_kernel<2, 2, 1>), g9, dim3(256), 0, s, a);
    else if (a64) DVG_LAUNCH_WORK(K_WGRAD_2x1, flops, (conv_wgrad9_kernel<2, 1, 2>), g9, dim3(256), 0, s, a);
    else if (b64) DVG_LAUNCH_WORK(K_WGRAD_1x2, flops, (conv_wgrad9_kernel<1, 2, 2>), g9, dim3(256), 0, s, a);
    else DVG_LAUNCH_WORK(K_WGRAD_1x1, flops, (conv_wgrad9_kernel<1, 1, 4>), g9, dim3(256), 0, s, a);
    return DVG_OK;
  }
  if (a.ntaps == 1 && wgrad1_dma_ok(a.M, a.Cin, a.Cout)) {
    auto kern = conv_wgrad1_dma_kernel;
    static std::atomic<uint64_t> attr_done{0};
    DVG_TRY(raise_dynamic_lds(attr_done, (const void*)kern, 65536));
    DVG_LAUNCH_WORK(K_WGRAD_2x2, flops, kern, dim3((unsigned)((a.Cin / 128) * (a.Cout / 128)), (unsigned)a.ksplit), dim3(256), 65536, s, a);
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
    for (int k0 = sub; k0 < ksplit; k0 += 64) {  // eight loads in flight, summed in the same order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = k0 + 8 * u < ksplit ? slabs[(size_t)(k0 + 8 * u) * total + e] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) if (k0 + 8 * u < ksplit) s += v[u];
    }
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

// one packed entry in the format `fmt` (the operand form of the launch the pack feeds): 0 = f32 [tap][a][b] (index e);
// 3 / 4 / 5 = f32 K-major [tap][b][a] (index kmaj): the LDS-DMA forms of the GEMM kernel
__device__ __forceinline__ void pack_store(float* wp, int64_t total, int64_t kmaj, int64_t e, float v, int fmt) {
  (void)total;
  wp[fmt == 0 ? e : kmaj] = v;
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

// (32-bit index arithmetic: a pack has < 2^31 entries, and the 64-bit divisions of the one-job kernel above were most of
// this launch's time at small sizes -- it sits at the head of every forward call; one entry per thread up to 2048 x 256)
__global__ __launch_bounds__(256) void weight_pack_multi_kernel(PackJobs jobs) {
  const PackJob& j = jobs.job[blockIdx.y];
  const WeightMap map = j.map;
  const uint32_t Ca = (uint32_t)map.Ca, Cb = (uint32_t)map.Cb, total = (uint32_t)map.ntaps * Ca * Cb;
  if (j.wino >= 2) {
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < Ca * Cb; e += gridDim.x * 256u) wino4_pack_entry(j.w, map, e, j.wp, j.wino == 3);
    return;
  }
  if (j.wino) {
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < Ca * Cb; e += gridDim.x * 256u) wino_pack_entry(j.w, map, e, j.wp);
    return;
  }
  for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    const uint32_t q = e / Cb, b = e - q * Cb, tap = q / Ca, a = q - tap * Ca;
    const float v = packed_weight(j.w, map, (int)tap, (int)a, (int)b);
    pack_store(j.wp, total, (int64_t)((tap * Cb + b) * Ca + a), e, v, j.bf16t);
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
  if (biggest >= (int64_t)1 << 31) { set_error("weight_pack_multi: a pack of %lld entries", (long long)biggest); return DVG_E_INVALID; }
  int64_t gx = ceil_div(biggest, 256);
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  DVG_LAUNCH(K_WEIGHT_PACK, weight_pack_multi_kernel, dim3((unsigned)gx, (unsigned)njobs), dim3(256), 0, s, pj);
  return DVG_OK;
}

static unsigned ew_grid(int64_t n) {
  const int64_t b = ceil_div(n, 256);
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// The sum the 8-lanes-per-element kernels make of at most eight slabs, by one thread: there lane k holds slab k alone
// (0.f + v_k) and the shuffle tree adds ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)), absent slabs +0 -- the same bits.
__device__ __forceinline__ float slab_tree8(const float (&v)[8]) {
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// Few slabs of a LARGE weight (c5: the 1024-latent Linear layer, 4 M entries in 2 slabs; the 128 -> 1024 convolution): with
// eight lanes per element most lanes idle, a wave reads 32-byte pieces of eight slabs and scatters single floats into the
// checkpoint layout (207 us for the Linear layer's gradient at c5).  Here a block owns a TA x 32 (a, b) tile with all its
// taps: reads with lanes along b (the slabs' fastest index), one thread per element; writes through LDS with lanes along
// whatever is fastest in the checkpoint layout (a for Conv2d / Linear weights, b for ConvTranspose2d: 128 / 1152 contiguous
// bytes per row of the tile).
template <int NT, int TA>  // taps; a rows of a tile (32 b columns): 9 taps: 8 rows -- 72 independent loads per thread, 4 x the blocks
__global__ __launch_bounds__(256) void wgrad_reduce_tile_kernel(const float* __restrict__ slabs, int ksplit, WeightMap map,
                                                                float* __restrict__ grad_w) {
  __shared__ float tile[NT][TA][33];
  const int tb = map.Cb / 32;
  const int a0 = ((int)blockIdx.x / tb) * TA, b0 = ((int)blockIdx.x % tb) * 32;
  const int64_t plane = (int64_t)map.Ca * map.Cb, total = NT * plane;
  const int bl = threadIdx.x & 31, ar = threadIdx.x >> 5;
  constexpr int NA = TA / 8;
  float v[NT][NA][8];
#pragma unroll
  for (int tap = 0; tap < NT; ++tap)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int64_t e = tap * plane + (int64_t)(a0 + ar + 8 * i) * map.Cb + b0 + bl;
#pragma unroll
      for (int u = 0; u < 8; ++u) v[tap][i][u] = u < ksplit ? slabs[(size_t)u * total + e] : 0.f;
    }
#pragma unroll
  for (int tap = 0; tap < NT; ++tap)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[tap][i][u] = 0.f + v[tap][i][u];
      tile[tap][ar + 8 * i][bl] = slab_tree8(v[tap][i]);
    }
  __syncthreads();
  const bool a_fast = map.mode == WM_CONV_FWD || map.mode == WM_CONVT_DGRAD || map.mode == WM_LIN_FWD;
  for (int idx = threadIdx.x; idx < NT * TA * 32; idx += 256) {
    const int tap = idx % NT, r = idx / NT;
    const int al = a_fast ? r % TA : r / 32, bb = a_fast ? r / TA : r % 32;
    grad_w[torch_weight_offset(map, tap, a0 + al, b0 + bb)] = tile[tap][al][bb];
  }
}

// dev option wgrad_reduce_tiled (1): the tiled form where it applies -- at most 8 slabs and at least 256 tiles
static bool reduce_tiled(int ksplit, int64_t tiles) { return opt(OPT_WGRAD_REDUCE_TILED) != 0 && ksplit <= 8 && tiles >= 256; }

int launch_wgrad_reduce(const float* slabs, int ksplit, const WeightMap& map, float* grad_w, hipStream_t s) {
  const int64_t total = (int64_t)map.ntaps * map.Ca * map.Cb;
  const int ta = map.ntaps == 9 ? 8 : 32;
  const int64_t tiles = (int64_t)(map.Ca / ta) * (map.Cb / 32);
  if (map.Ca % 32 == 0 && map.Cb % 32 == 0 && (map.ntaps == 1 || map.ntaps == 9) && map.mode != WM_CONVT_D22_FWD &&
      map.mode != WM_CONVT_D22_DGRAD && reduce_tiled(ksplit, tiles)) {
    if (map.ntaps == 9) DVG_LAUNCH(K_WGRAD_REDUCE, (wgrad_reduce_tile_kernel<9, 8>), dim3((unsigned)tiles), dim3(256), 0, s, slabs, ksplit, map, grad_w);
    else DVG_LAUNCH(K_WGRAD_REDUCE, (wgrad_reduce_tile_kernel<1, 32>), dim3((unsigned)tiles), dim3(256), 0, s, slabs, ksplit, map, grad_w);
    return DVG_OK;
  }
  DVG_LAUNCH(K_WGRAD_REDUCE, wgrad_reduce_kernel, dim3(ew_grid(total * 8)), dim3(256), 0, s, slabs, ksplit, map, grad_w);
  return DVG_OK;
}

int launch_wgrad_fold_reduce(const float* slabs, int ksplit, int Cin, int Cout, float* grad_w, hipStream_t s) {
  DVG_LAUNCH(K_WGRAD_REDUCE, wgrad_fold_reduce_kernel, dim3(ew_grid((int64_t)9 * Cin * Cout * 8)), dim3(256), 0, s, slabs, ksplit,
             Cin, Cout, grad_w);
  return DVG_OK;
}

__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, int ksplit, int rows, int cols,
                                                      float* __restrict__ out, float* __restrict__ outT) {
  const int64_t total = (int64_t)rows * cols;
  const int sub = threadIdx.x & 7;
  for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; e < total; e += ((int64_t)gridDim.x * 256) >> 3) {
    float s = 0.f;
    for (int k0 = sub; k0 < ksplit; k0 += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = k0 + 8 * u < ksplit ? slabs[(size_t)(k0 + 8 * u) * total + e] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) if (k0 + 8 * u < ksplit) s += v[u];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0) {
      out[e] = s;
      if (outT) outT[(e % cols) * rows + e / cols] = s;
    }
  }
}

int launch_slab_sum(const float* slabs, int ksplit, int rows, int cols, float* out, float* outT, hipStream_t s) {
  DVG_LAUNCH(K_WGRAD_REDUCE, slab_sum_kernel, dim3(ew_grid((int64_t)rows * cols * 8)), dim3(256), 0, s, slabs, ksplit, rows, cols,
             out, outT);
  return DVG_OK;
}

// one wavefront per output: a dot product of length `len` in double, fixed lane partition and shuffle tree
__device__ __forceinline__ double lc0_wave_dot(const float* __restrict__ a_vec, const float* __restrict__ row, int len, int lane,
                                               int perm_n) {
  double s = 0.0;
  for (int j = lane; j < len; j += 64) {
    // perm_n > 0: element j = p*n + c of the vector lives at a_vec[c*4 + p] (the Linear bias in checkpoint order)
    const float av = perm_n ? a_vec[(j % perm_n) * 4 + j / perm_n] : a_vec[j];
    s += (double)av * (double)row[j];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  return s;
}

__global__ __launch_bounds__(256) void lc0_bias_kernel(const float* __restrict__ lin_b, const float* __restrict__ wk_eff,
                                                      const float* __restrict__ conv_b, int n, int C, float* __restrict__ bc) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= 4 * C) return;
  const double s = lc0_wave_dot(lin_b, wk_eff + (size_t)o * 4 * n, 4 * n, lane, n);
  if (lane == 0) bc[o] = (float)(s + (double)conv_b[o % C]);
}

__global__ __launch_bounds__(256) void lc0_lin_bias_grad_kernel(const float* __restrict__ dbc, const float* __restrict__ wk_d,
                                                               int n, int C, float* __restrict__ grad_lin_b) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= 4 * n) return;
  const double s = lc0_wave_dot(dbc, wk_d + (size_t)j * 4 * C, 4 * C, lane, 0);
  if (lane == 0) grad_lin_b[(j % n) * 4 + j / n] = (float)s;
}

__global__ __launch_bounds__(256) void lc0_rows_to_linear_kernel(const float* __restrict__ t, int n, float* __restrict__ grad_lin_w) {
  const int64_t total = (int64_t)4 * n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e / n), i = (int)(e - (int64_t)j * n);
    grad_lin_w[(int64_t)((j % n) * 4 + j / n) * n + i] = t[e];
  }
}

// dWeff[j][o] += lin_b[c*4 + p] * dbc[o]  (j = p*n + c): the Linear bias reaches the dense map's weight gradient too
__global__ __launch_bounds__(256) void lc0_rank1_add_kernel(float* __restrict__ dweff, const float* __restrict__ lin_b,
                                                           const float* __restrict__ dbc, int n, int C4) {
  const int64_t total = (int64_t)4 * n * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e / C4), o = (int)(e - (int64_t)j * C4);
    dweff[e] = fmaf(lin_b[(j % n) * 4 + j / n], dbc[o], dweff[e]);
  }
}
int launch_lc0_rank1_add(float* dweff, const float* lin_b, const float* dbc, int n, int C, hipStream_t s) {
  DVG_LAUNCH(K_MISC, lc0_rank1_add_kernel, dim3(ew_grid((int64_t)4 * n * 4 * C)), dim3(256), 0, s, dweff, lin_b, dbc, n, 4 * C);
  return DVG_OK;
}

int launch_lc0_bias(const float* lin_b, const float* wk_eff, const float* conv_b, int n, int C, float* bc, hipStream_t s) {
  DVG_LAUNCH(K_MISC, lc0_bias_kernel, dim3((unsigned)C), dim3(256), 0, s, lin_b, wk_eff, conv_b, n, C, bc);
  return DVG_OK;
}
int launch_lc0_lin_bias_grad(const float* dbc, const float* wk_d, int n, int C, float* grad_lin_b, hipStream_t s) {
  DVG_LAUNCH(K_MISC, lc0_lin_bias_grad_kernel, dim3((unsigned)n), dim3(256), 0, s, dbc, wk_d, n, C, grad_lin_b);
  return DVG_OK;
}
int launch_lc0_rows_to_linear(const float* t, int n, float* grad_lin_w, hipStream_t s) {
  DVG_LAUNCH(K_MISC, lc0_rows_to_linear_kernel, dim3(ew_grid((int64_t)4 * n * n)), dim3(256), 0, s, t, n, grad_lin_w);
  return DVG_OK;
}

// The same for few slabs of a large weight (wgrad_reduce_tile_kernel's reasoning): one thread per (ci, co) pair makes all
// nine taps -- the sixteen (p_in, p_out) blocks of each slab read once, lanes along co -- and a block writes its 256 pairs'
// 2304 contiguous floats through LDS.  Same sums: per slab the blocks in p_out order, then slab_tree8.
__global__ __launch_bounds__(256) void wgrad_d22_reduce_pair_kernel(const float* __restrict__ slabs, int ksplit, int Cin, int Cout,
                                                                   float* __restrict__ grad_w) {
  __shared__ float o[256 * 9];
  const int64_t srow = 4 * (int64_t)Cout, slab = 4 * (int64_t)Cin * srow;
  const int64_t ab = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (Cin Cout is a multiple of 256: the launcher checks)
  const int ci = (int)(ab / Cout), co = (int)(ab - (int64_t)ci * Cout);
  float v[9][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t][k] = 0.f;
    if (k < ksplit) {
      const float* sl = slabs + (size_t)k * slab;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
        for (int po = 0; po < 4; ++po) {
          const int yi = (po >> 1) + dy, xi = (po & 1) + dx;
          if (yi >= 0 && yi < 2 && xi >= 0 && xi < 2) v[t][k] += sl[((int64_t)((yi << 1 | xi) * Cin + ci)) * srow + po * Cout + co];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) o[threadIdx.x * 9 + (8 - t)] = slab_tree8(v[t]);
  __syncthreads();
  float* dst = grad_w + (int64_t)blockIdx.x * 256 * 9;
  for (int i = threadIdx.x; i < 256 * 9; i += 256) dst[i] = o[i];
}

int launch_wgrad_d22_reduce(const float* slabs, int ksplit, int Cin, int Cout, float* grad_w, hipStream_t s) {
  if (((int64_t)Cin * Cout) % 256 == 0 && reduce_tiled(ksplit, (int64_t)Cin * Cout / 256)) {
    DVG_LAUNCH(K_WGRAD_REDUCE, wgrad_d22_reduce_pair_kernel, dim3((unsigned)((int64_t)Cin * Cout / 256)), dim3(256), 0, s, slabs,
               ksplit, Cin, Cout, grad_w);
    return DVG_OK;
  }
  DVG_LAUNCH(K_WGRAD_REDUCE, wgrad_d22_reduce_kernel, dim3(ew_grid((int64_t)9 * Cin * Cout * 8)), dim3(256), 0, s, slabs, ksplit,
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
