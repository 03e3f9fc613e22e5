// Weight gradient of the encoder's stride-1 3x3 layers (/root/reference/src/encoder.py:28-36) in the Winograd F(4x4, 3x3)
// domain, float32 on v_mfma_f32_16x16x4_f32:
//
//   dU[xi][nu][ci][co] = sum over the 4x4 output tiles of  V[xi][nu][tile][ci] * Z[xi][nu][tile][co]
//   V = B^T d B   (the forward's input transform of the tile's 6x6 input patch d, zero padded)
//   Z = A dY A^T  (the ADJOINT of the forward's output transform: 4x4 gradient tile -> 6x6)
//   dg = G^T dU G (once per launch, wino4_wgrad_reduce_kernel)
//
// 36 position GEMMs over the tiles per 16 output pixels where F(2x2, 3x3) (conv_wino_wgrad.hip) runs 16 per 4: 0.5625 of
// the matrix work.  The tiles are the GEMMs' reduction dimension: one MFMA k-step = 4 tiles = one chunk.
//
// One workgroup = 8 waves, two per SIMD, a channel tile of CI x CO = 64 x 32 or 32 x 64 and a slab of the tiles; a wave
// owns ALL 36 positions of a 16 x 16 channel sub-tile (36 accumulator tiles of 4 registers), so nothing is exchanged
// and the slab stores are lane-local.  Both operands are transformed by the workgroup: a thread owns one patch per
// chunk -- (tile, input channel): 36 (16 for 4x4 images: the halo is known) coalesced global loads, lanes = channels, no
// LDS staging of raw pixels, no LDS-DMA -- or one gradient tile (tile, output channel): 16 loads; the transformed entries
// [tile-of-chunk][channel][36 floats] go to LDS (two stages, one barrier per chunk), where the MFMA operands are one
// ds_read_b128 per four positions (entries 144 bytes apart: conflict-free as they are, for the reads and for the writes).
// The loads of a chunk are issued behind the transform of the chunk before it: a chunk and a half ahead of their use.
#include <atomic>

#include "conv.h"
#include "conv_tile.h"
#include "conv_wino4.h"

namespace dvg {

typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

struct Wino4WgradArgs {
  const float* in;   // [16 tiles][Cin]  (Morton pixel order)
  const float* dy;   // [16 tiles][Cout]
  float* slabs;      // [nsplit][36][Cin][Cout]
  int Cin, Cout;
  int tiles;         // 4x4 output tiles (M / 16)
  int per;           // chunks (4 tiles) per slab: slab z covers chunks [z per, min((z + 1) per, tiles / 4))
};

// WA x WB waves = 16 WA input channels x 16 WB output channels
template <int L, int WA, int WB>
struct Wino4WgradCfg {
  static constexpr int CI = 16 * WA, CO = 16 * WB;
  static constexpr int NV = CI / 16, NZ = CO / 16;  // waves that transform input patches (4 tiles x CI) / gradient tiles (4 tiles x CO)
  static constexpr int V_B = 4 * CI * 144, Z_B = 4 * CO * 144;
  static constexpr int OFF_V = 0, OFF_Z = 2 * V_B, LDS_BYTES = OFF_Z + 2 * Z_B;
  static_assert(WA * WB == 8 && (WA == 4 || WA == 2) && NV + NZ <= 8 && L >= 2 && L <= 4, "unsupported tile");
};

template <int L, int WA, int WB>
__global__ __launch_bounds__(512) void conv_wino4_wgrad_kernel(Wino4WgradArgs a) {
  using C = Wino4WgradCfg<L, WA, WB>;
  constexpr int H = 1 << L, HW = H * H, TPI = HW / 16;  // (tiles per image)
  extern __shared__ __align__(16) unsigned char wsm[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte_t*)wsm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave % WA, wb = wave / WA;  // the wave's 16 input channels and 16 output channels
  const int kq = lane >> 4, r16 = lane & 15;
  const int ci0 = blockIdx.x * C::CI, co0 = blockIdx.y * C::CO;
  // the slab's chunks (4 tiles each)
  const int chunk0 = (int)blockIdx.z * a.per;
  const int nchunks = a.tiles / 4 - chunk0 < a.per ? a.tiles / 4 - chunk0 : a.per;

  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in), 0, (int)((int64_t)a.tiles * 16 * a.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.dy), 0, (int)((int64_t)a.tiles * 16 * a.Cout * 4), 0x00020000);

  // MFMA operands: A = Z (rows: output channels), B = V (columns: input channels); lane (kq, r16) reads tile kq of the chunk
  uint32_t zaddr[2], vaddr[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    zaddr[st] = lds0 + C::OFF_Z + st * C::Z_B + (uint32_t)((kq * C::CO + 16 * wb + r16) * 144);
    vaddr[st] = lds0 + C::OFF_V + st * C::V_B + (uint32_t)((kq * C::CI + 16 * wa + r16) * 144);
    asm volatile("" : "+v"(zaddr[st]), "+v"(vaddr[st]));
  }

  f32x4 acc[36];  // acc[6 xi + nu][v] = dU[xi][nu] of output channel 16 wb + 4 kq + v, input channel 16 wa + r16
#pragma unroll
  for (int p = 0; p < 36; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the MFMA stream of one chunk: 9 operand quads x 4 MFMAs, operands two quads ahead; `piece(q)` is the role's work beside
  // quad q
  auto mfma_chunk = [&](auto stc, auto&& piece) {
    constexpr int st = decltype(stc)::value;
    f32x4 qa[3], qb[3];
    auto load_quad = [&](int q) {
      qa[q % 3] = *reinterpret_cast<lds_cf32x4*>((uintptr_t)(zaddr[st] + (uint32_t)(q * 16)));
      qb[q % 3] = *reinterpret_cast<lds_cf32x4*>((uintptr_t)(vaddr[st] + (uint32_t)(q * 16)));
    };
    load_quad(0);
    load_quad(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      if (q + 2 < 9) load_quad(q + 2);
      piece(q);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        acc[4 * q + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[q % 3][m], qb[q % 3][m], acc[4 * q + m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  if (wave < C::NV) {
    // ================= input-patch role: tile (wave's quarter of the chunk's tiles ... ) =================
    // 4 tiles x CI channels = 4 CI patches over NV waves of 64 lanes: CI = 64: wave w = tile w, lane = channel;
    // CI = 32: wave w = tiles 2 w, 2 w + 1 (lanes 0-31 / 32-63), lane & 31 = channel
    constexpr int TPW = 4 / C::NV;                     // tiles per wave
    const int tsub = TPW == 1 ? 0 : lane >> 5;         // the lane's tile among the wave's
    const int tch = wave * TPW + tsub;                 // ... among the chunk's four
    const int ch = ci0 + (TPW == 1 ? lane : (lane & 31));
    uint32_t vst[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      vst[st] = lds0 + C::OFF_V + st * C::V_B + (uint32_t)((tch * C::CI + (ch - ci0)) * 144);
      asm volatile("" : "+v"(vst[st]));
    }
    const int lo = ch * 4;
    float d[36];
    // the patch of tile `t` (wave-uniform unless two tiles share the wave: then per lane half)
    auto load_patch = [&](int chunk) {
      const int t = chunk * 4 + tch;
      if constexpr (L == 2) {
        // one tile per image: pixel (i - 1, j - 1) at its Morton index in the image
        const int base = t * 16 * a.Cin * 4;
#pragma unroll
        for (int i = 1; i < 5; ++i)
#pragma unroll
          for (int j = 1; j < 5; ++j) {
            const int px = (int)morton((uint32_t)(i - 1), (uint32_t)(j - 1));
            if constexpr (TPW == 1)  // (the tile is the wave's: its offset travels in the scalar operand)
              d[i * 6 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, lo, __builtin_amdgcn_readfirstlane(base) + px * a.Cin * 4, 0));
            else
              d[i * 6 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, lo + base, px * a.Cin * 4, 0));
          }
      } else {
        const int img = t / TPI, tp = t - img * TPI;
        const int ty = (int)morton_y((uint32_t)tp), tx = (int)morton_x((uint32_t)tp);
        // Morton index of pixel (4 ty - 1 + i, 4 tx - 1 + j) = row part + column part (disjoint bits); a tap outside the image
        // gets an offset past the buffer: the hardware returns zero
        int rp[6], cp[6];
        bool rv[6], cv[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int y = 4 * ty - 1 + i, x = 4 * tx - 1 + i;
          rv[i] = y >= 0 && y < H; cv[i] = x >= 0 && x < H;
          rp[i] = (int)(part1by1((uint32_t)(rv[i] ? y : 0)) << 1);
          cp[i] = (int)part1by1((uint32_t)(cv[i] ? x : 0));
        }
        const int base = img * HW;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const bool ok = rv[i] && cv[j];
            if constexpr (TPW == 1) {
              const int soff = __builtin_amdgcn_readfirstlane((base + rp[i] + cp[j]) * a.Cin * 4);
              d[i * 6 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, ok ? lo : (int)0x80000000u, soff, 0));
            } else {
              const int off = ok ? lo + (base + rp[i] + cp[j]) * a.Cin * 4 : (int)0x80000000u;
              d[i * 6 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, off, 0, 0));
            }
          }
      }
    };
    auto zero_ring = [&]() {
      if constexpr (L == 2) {
#pragma unroll
        for (int e = 0; e < 36; ++e)
          if (e / 6 == 0 || e / 6 == 5 || e % 6 == 0 || e % 6 == 5) d[e] = 0.f;
      }
    };
    auto xform_col = [&](int j) {
      if (L == 2 && (j == 0 || j == 5)) return;
      wino4_in6<L == 2>(d[j], d[6 + j], d[12 + j], d[18 + j], d[24 + j], d[30 + j]);
    };
    auto xform_row = [&](int i) { wino4_in6<L == 2>(d[6 * i], d[6 * i + 1], d[6 * i + 2], d[6 * i + 3], d[6 * i + 4], d[6 * i + 5]); };
    auto store_quads = [&](int vs, int q0) {
#pragma unroll
      for (int q = q0; q < q0 + 3; ++q) {
        const f32x4 o = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
        *reinterpret_cast<lds_f32x4*>((uintptr_t)(vst[vs] + (uint32_t)(q * 16))) = o;
      }
    };
    auto xform_all = [&](int vs) {
      zero_ring();
#pragma unroll
      for (int j = 0; j < 6; ++j) xform_col(j);
#pragma unroll
      for (int i = 0; i < 6; ++i) xform_row(i);
      store_quads(vs, 0); store_quads(vs, 3); store_quads(vs, 6);
    };
    // prologue: chunk 0 transformed, chunk 1 loaded
    load_patch(chunk0);
    xform_all(0);
    if (nchunks > 1) load_patch(chunk0 + 1);
    __syncthreads();
    auto step = [&](int c, auto stc) {
      constexpr int st = decltype(stc)::value;
      const bool more = c + 1 < nchunks;
      mfma_chunk(stc, [&](int q) {
        // the NEXT chunk's patch (loaded behind the previous transform): columns beside quads 3-5, rows and stores 6-8
        if (!more) return;
        if (q == 3) zero_ring();
        if (q >= 3 && q <= 5) { xform_col(2 * q - 6); xform_col(2 * q - 5); }
        if (q >= 6) { xform_row(2 * q - 12); xform_row(2 * q - 11); store_quads(st ^ 1, 3 * (q - 6)); }
      });
      if (c + 2 < nchunks) load_patch(chunk0 + c + 2);
      __syncthreads();
    };
    for (int c = 0; c < nchunks; c += 2) {
      step(c, I0{});
      if (c + 1 < nchunks) step(c + 1, I1{});
    }
  } else if (wave < C::NV + C::NZ) {
    // ================= gradient-tile role: 4 tiles x CO channels over NZ waves: lanes 0-31 / 32-63 two tiles (CO = 32: wave
    // = tile pair; CO = 64: wave = (tile pair, channel half)) =================
    const int zw = wave - C::NV;
    const int tch = C::CO == 32 ? 2 * zw + (lane >> 5) : 2 * (zw >> 1) + (lane >> 5);
    const int chl = C::CO == 32 ? (lane & 31) : 32 * (zw & 1) + (lane & 31);
    uint32_t zst[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      zst[st] = lds0 + C::OFF_Z + st * C::Z_B + (uint32_t)((tch * C::CO + chl) * 144);
      asm volatile("" : "+v"(zst[st]));
    }
    const int lo = ((tch * 16) * a.Cout + co0 + chl) * 4;
    float z[36];  // rows 0..3 hold the 4x4 gradient tile (row-major 4 x 6: columns 0..3), transformed in place to 6 x 6
    auto load_tile = [&](int chunk) {
      const int soff = __builtin_amdgcn_readfirstlane(chunk * 64 * a.Cout * 4);
#pragma unroll
      for (int aa = 0; aa < 4; ++aa)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          z[aa * 6 + b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
              rsrc_dy, lo, soff + (int)morton((uint32_t)aa, (uint32_t)b) * a.Cout * 4, 0));
    };
    // columns: (A dY)[xi][b] over the tile's rows a; rows: (.) A^T over b
    auto xf_col = [&](int b) { wino4_dy6(z[b], z[6 + b], z[12 + b], z[18 + b], z[24 + b], z[30 + b]); };
    auto xf_row = [&](int i) { wino4_dy6(z[6 * i], z[6 * i + 1], z[6 * i + 2], z[6 * i + 3], z[6 * i + 4], z[6 * i + 5]); };
    auto store_quads = [&](int vs, int q0) {
#pragma unroll
      for (int q = q0; q < q0 + 3; ++q) {
        const f32x4 o = {z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]};
        *reinterpret_cast<lds_f32x4*>((uintptr_t)(zst[vs] + (uint32_t)(q * 16))) = o;
      }
    };
    load_tile(chunk0);
#pragma unroll
    for (int b = 0; b < 4; ++b) xf_col(b);
#pragma unroll
    for (int i = 0; i < 6; ++i) xf_row(i);
    store_quads(0, 0); store_quads(0, 3); store_quads(0, 6);
    if (nchunks > 1) load_tile(chunk0 + 1);
    __syncthreads();
    auto step = [&](int c, auto stc) {
      constexpr int st = decltype(stc)::value;
      const bool more = c + 1 < nchunks;
      mfma_chunk(stc, [&](int q) {
        if (!more) return;
        if (q >= 3 && q <= 4) { xf_col(2 * q - 6); xf_col(2 * q - 5); }
        if (q >= 6) { xf_row(2 * q - 12); xf_row(2 * q - 11); store_quads(st ^ 1, 3 * (q - 6)); }
      });
      if (c + 2 < nchunks) load_tile(chunk0 + c + 2);
      __syncthreads();
    };
    for (int c = 0; c < nchunks; c += 2) {
      step(c, I0{});
      if (c + 1 < nchunks) step(c + 1, I1{});
    }
  } else {
    // ================= no transform role =================
    __syncthreads();
    auto step = [&](int c, auto stc) {
      mfma_chunk(stc, [&](int) {});
      __syncthreads();
    };
    for (int c = 0; c < nchunks; c += 2) {
      step(c, I0{});
      if (c + 1 < nchunks) step(c + 1, I1{});
    }
  }

  // the slab: [36][Cin][Cout], the lane's four consecutive output channels of input channel 16 wa + r16
  float* dst = a.slabs + (size_t)blockIdx.z * 36 * a.Cin * a.Cout + (size_t)(ci0 + 16 * wa + r16) * a.Cout + co0 + 16 * wb + 4 * kq;
#pragma unroll
  for (int p = 0; p < 36; ++p) *reinterpret_cast<f32x4*>(dst + (size_t)p * a.Cin * a.Cout) = acc[p];
}

// (1) every (position, ci, co) element summed over the slabs by 8 lanes (strided over the slabs, fixed-shape shuffle tree:
// deterministic), in place into slab 0 (conv_wino_wgrad.hip: the same two-pass form)
__global__ __launch_bounds__(256) void wino4_wgrad_slabsum_kernel(float* __restrict__ slabs, int nslabs, int64_t total) {
  const int sub = threadIdx.x & 7;
  for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; e < total; e += ((int64_t)gridDim.x * 256) >> 3) {
    float s = 0.f;
    for (int k0 = sub; k0 < nslabs; k0 += 64) {  // eight loads in flight, summed in the same order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = k0 + 8 * u < nslabs ? slabs[(size_t)(k0 + 8 * u) * total + e] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) if (k0 + 8 * u < nslabs) s += v[u];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0) slabs[e] = s;
  }
}

// (2) dg = G^T dU G per channel pair from slab 0, written in the checkpoint layout
__global__ __launch_bounds__(256) void wino4_wgrad_reduce_kernel(const float* __restrict__ du, WeightMap map, float* __restrict__ grad_w) {
  const int64_t pairs = (int64_t)map.Ca * map.Cb;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < pairs; e += (int64_t)gridDim.x * 256) {
    float u[36];
#pragma unroll
    for (int p = 0; p < 36; ++p) u[p] = du[(size_t)p * pairs + e];
    // G^T x for a line of six: [x0/4 - (x1+x2)/6 + (x3+x4)/24,  (x2-x1)/6 + (x3-x4)/12,  x5 - (x1+x2)/6 + (x3+x4)/6]
    float t[3][6];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) {
      const float s12 = u[6 + nu] + u[12 + nu], d21 = u[12 + nu] - u[6 + nu], s34 = u[18 + nu] + u[24 + nu], d34 = u[18 + nu] - u[24 + nu];
      t[0][nu] = (0.25f * u[nu] - s12 * (1.f / 6.f)) + s34 * (1.f / 24.f);
      t[1][nu] = d21 * (1.f / 6.f) + d34 * (1.f / 12.f);
      t[2][nu] = (u[30 + nu] - s12 * (1.f / 6.f)) + s34 * (1.f / 6.f);
    }
    const int b = (int)(e % map.Cb), av = (int)(e / map.Cb);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float s12 = t[r][1] + t[r][2], d21 = t[r][2] - t[r][1], s34 = t[r][3] + t[r][4], d34 = t[r][3] - t[r][4];
      grad_w[torch_weight_offset(map, 3 * r + 0, av, b)] = (0.25f * t[r][0] - s12 * (1.f / 6.f)) + s34 * (1.f / 24.f);
      grad_w[torch_weight_offset(map, 3 * r + 1, av, b)] = d21 * (1.f / 6.f) + d34 * (1.f / 12.f);
      grad_w[torch_weight_offset(map, 3 * r + 2, av, b)] = (t[r][5] - s12 * (1.f / 6.f)) + s34 * (1.f / 6.f);
    }
  }
}

bool wino4_wgrad_shape_ok(int64_t M, int Cin, int Cout, int L) {
  if (L < 2 || L > 4 || M <= 0 || M % 64) return false;  // whole chunks of four tiles (of whole images: M % 4^L below)
  if (M & (((int64_t)1 << (2 * L)) - 1)) return false;
  const bool t42 = Cin % 64 == 0 && Cout % 32 == 0, t24 = Cin % 32 == 0 && Cout % 64 == 0;
  if (!(t42 || t24)) return false;
  if (M * (int64_t)(Cin > Cout ? Cin : Cout) * 4 >= 2147483647LL) return false;  // (32-bit buffer offsets)
  return true;
}

struct Wino4WgradGeom { int cfg, nx, ny, nsplit, per; };
Wino4WgradGeom wino4_wgrad_geom(int64_t M, int Cin, int Cout, int cus) {
  Wino4WgradGeom g;
  g.cfg = Cin % 64 == 0 ? 0 : 1;  // 0: 64 x 32 channel tile, 1: 32 x 64
  g.nx = Cin / (g.cfg == 0 ? 64 : 32);
  g.ny = Cout / (g.cfg == 0 ? 32 : 64);
  const int chunks = (int)(M / 64);
  // one workgroup per CU of the budget, every workgroup at least 8 chunks deep; the last slab may be shorter
  int want = cus / (g.nx * g.ny);
  if (want < 1) want = 1;
  if (want > chunks / 8) want = chunks / 8 > 0 ? chunks / 8 : 1;
  g.per = (chunks + want - 1) / want;
  g.nsplit = (chunks + g.per - 1) / g.per;
  return g;
}

template <int L, int WA, int WB>
int launch_wino4_wgrad_cfg(const Wino4WgradArgs& a, double flops, dim3 grid, hipStream_t s) {
  using C = Wino4WgradCfg<L, WA, WB>;
  auto kern = conv_wino4_wgrad_kernel<L, WA, WB>;
  static std::atomic<uint64_t> attr_done{0};
  DVG_TRY(raise_dynamic_lds(attr_done, (const void*)kern, C::LDS_BYTES));
  const unsigned wgs = grid.x * grid.y * grid.z;
  DVG_LAUNCH_WORK_SHARE(K_WGRAD_WINO4, flops, (float)(wgs > 256u ? 256u : wgs) / 256.0f, kern, grid, dim3(512), C::LDS_BYTES, s, a);
  return DVG_OK;
}

bool conv_wino4_wgrad_shape(int64_t M, int Cin, int Cout, int L) { return wino4_wgrad_shape_ok(M, Cin, Cout, L); }

// option enc_wino4: 1 (default) the weight gradients of the layers on 4x4 images (alone on 128 CUs at c3 0.76 of the F(2x2)
// kernel's time; on 8x8 and 16x16 images the two forms are level), 2 every layer the shape allows, 0 never
bool conv_wino4_wgrad_ok(int64_t M, int Cin, int Cout, int L) {
  const int64_t o = opt(OPT_ENC_WINO4);
  return o != 0 && wino4_wgrad_shape_ok(M, Cin, Cout, L);  // (which layers: option enc_wino4_mask, encoder.cpp)
}

size_t conv_wino4_wgrad_slab_floats(int64_t M, int Cin, int Cout) {
  const Wino4WgradGeom g = wino4_wgrad_geom(M, Cin, Cout, 256);  // (the largest split any CU budget gives)
  return (size_t)g.nsplit * 36 * Cin * Cout;
}

int launch_conv_wino4_wgrad(const float* in, const float* dy, int64_t M, int Cin, int Cout, int L, float* slabs,
                            const WeightMap& map, float* grad_w, hipStream_t s, int cus) {
  DVG_REQUIRE(wino4_wgrad_shape_ok(M, Cin, Cout, L), "conv_wino4_wgrad: unsupported launch (M=%lld Cin=%d Cout=%d L=%d)",
              (long long)M, Cin, Cout, L);
  if (cus <= 0) cus = WINO_CUS_ENC_WGRAD;
  if (cus > 256) cus = 256;
  const Wino4WgradGeom g = wino4_wgrad_geom(M, Cin, Cout, cus);
  Wino4WgradArgs a;
  a.in = in; a.dy = dy; a.slabs = slabs; a.Cin = Cin; a.Cout = Cout; a.tiles = (int)(M / 16); a.per = g.per;
  const double flops = 2.0 * (double)(M / 16) * 36.0 * Cin * Cout;  // executed position GEMMs
  const dim3 grid((unsigned)g.nx, (unsigned)g.ny, (unsigned)g.nsplit);
  int rc;
  if (g.cfg == 0) rc = L == 2 ? launch_wino4_wgrad_cfg<2, 4, 2>(a, flops, grid, s) : L == 3 ? launch_wino4_wgrad_cfg<3, 4, 2>(a, flops, grid, s) : launch_wino4_wgrad_cfg<4, 4, 2>(a, flops, grid, s);
  else rc = L == 2 ? launch_wino4_wgrad_cfg<2, 2, 4>(a, flops, grid, s) : L == 3 ? launch_wino4_wgrad_cfg<3, 2, 4>(a, flops, grid, s) : launch_wino4_wgrad_cfg<4, 2, 4>(a, flops, grid, s);
  DVG_TRY(rc);
  const int64_t pairs = (int64_t)Cin * Cout;
  if (g.nsplit > 1) {
    int64_t blocks = (36 * pairs * 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    DVG_LAUNCH(K_WGRAD_REDUCE, wino4_wgrad_slabsum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slabs, g.nsplit, 36 * pairs);
  }
  DVG_LAUNCH(K_WGRAD_REDUCE, wino4_wgrad_reduce_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, (const float*)slabs, map, grad_w);
  return DVG_OK;
}

}  // namespace dvg
