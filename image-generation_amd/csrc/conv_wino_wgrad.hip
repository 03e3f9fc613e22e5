// Winograd F(2x2, 3x3) form of the WEIGHT gradient of the encoder's stride-1 3x3 layers
// (/root/reference/src/encoder.py:28-36: what autograd computes for nn.Conv2d(3x3, padding 1) behind those lines).
//
// The forward form (conv_wino.hip) computes  Y = A^T [ (G g G^T) .* (B^T d B) ] A  per 2x2 output quad; its adjoint in the
// weights is again sixteen independent GEMMs, now with the QUADS as the reduction dimension:
//     V[p][t][ci] = (B^T d_t B)[p]         the transformed 4x4 input patch of quad t   (p = 4 xi + nu: 16 positions)
//     Z[p][t][co] = (A dY_t A^T)[p]        the transformed 2x2 output-gradient quad
//     dU[p][ci][co] = sum_t V[p][t][ci] Z[p][t][co]                      <- the MFMA work: 16 x 2 T Cin Cout FLOPs
//     dg = G^T dU G                        (3x3 per channel pair; the slab-reduce pass, wino_wgrad_reduce_kernel)
// i.e. 8 M Cin Cout FLOPs where the direct form (conv_wgrad.hip) multiplies 18 M Cin Cout, padding taps included (31 % of
// them at 4x4 images).  One workgroup = 8 waves, two per SIMD: a wave PAIR owns a 32 x 32 channel tile and splits its 16
// positions (conv_wino_wgrad8_kernel below).  The reduction runs over chunks of NQ quads whose transformed operands sit
// in LDS as [quad][channel][16 floats] (64-byte entries, 16-byte slots XOR-swizzled by bits 2-3 of the channel); each
// thread transforms one (quad, 2 channels) item per chunk -- 16 (input) or 4 (gradient) buffer_load_dwordx2 straight
// from global memory, a padding tap being an out-of-range offset that loads zeros -- and writes it into the other stage
// while the matrix pipe works on this one.
// UPS = true: the decoder's Upsample(x2) + ConvTranspose2d 3x3 layers (/root/reference/src/decoder.py:34-46).  The layer is
// a 3x3 convolution of the nearest-upsampled map, so the 4x4 input patch of the output quad of source pixel (i, j) is
// d[u][v] = s[i + r(u)][j + r(v)], r = (-1, 0, 0, +1): along each axis B^T maps (x-, x0, x0, x+) to (x- - x0, 2 x0, 0,
// x0 - x+) -- transform position 2 VANISHES in both directions, 9 of the 16 position GEMMs remain (xi, nu in {0, 1, 3}).
// Nine MFMAs per k-step instead of the folded form's sixteen (class, tap) pairs, from a 3x3 SOURCE patch.
// K order: a thread keeps ONE quad position (its 16 patch offsets are constants of the kernel) and walks the images, so
// the split over workgroups is (position group) x (image range); slabs are reduced in fixed order (no float atomics).
#include <type_traits>

#include "conv.h"
#include "conv_tile.h"

namespace dvg {

struct WinoWgradArgs {
  const float* in;   // [M][Cin]   layer input, Morton pixel order (quad t = rows 4t .. 4t+3); UPS: the SOURCE map [M / 4][Cin]
  const float* dy;   // [M][Cout]  gradient of the layer's output
  float* slabs;      // [nsplit][16][Cin][Cout]
  int64_t M;
  int Cin, Cout, L;
  int isplit;        // image-range splits (grid.z = position groups x isplit [x K groups inside the block])
};

// Eight waves per workgroup, two per SIMD (round 5; rounds 3-4 ran four waves with all 16 positions = 256 accumulator
// registers each): the 16 (9) transform positions of a 32 x 32 channel sub-tile are split over a wave PAIR (waves w and
// w + 4 share a SIMD): rows xi in {0, 1} and {2, 3} of dU, 8 accumulator tiles = 128 registers each, so the pair fits the
// register file side by side and one wave's operand waits and transform arithmetic fall under the other's MFMAs.  The
// positions are independent in the weight gradient: no exchange, each wave stores its own eight slab planes.  Transform items are (quad, TWO channels) -- 16 (4) buffer_load_dwordx2 per thread
// and chunk, 32 (8) registers in flight -- so that all eight waves carry a role (64 x 64 tile: waves 0-3 the input
// transform, waves 4-7 the gradient transform: one of each per SIMD).
template <int WA, int WB>
struct WinoWgrad8Cfg {
  static constexpr int WK = 4 / (WA * WB);          // K groups inside the block (wave pairs beyond the WA x WB channel tile)
  static constexpr int CIB = 32 * WA, COB = 32 * WB;
  static constexpr int QC = WK == 1 ? 8 : 4;         // quads per K group and chunk
  static constexpr int NQ = QC * WK;                 // quads per chunk of the whole block
  static constexpr int V_B = NQ * CIB * 64, Z_B = NQ * COB * 64, STAGE = V_B + Z_B;
  static constexpr int LDS_BYTES = 2 * STAGE;
  static constexpr int NV = NQ * CIB / 2, NZ = NQ * COB / 2;  // (quad, 2-channel) items per chunk
  static_assert(WA * WB * WK == 4 && NV + NZ <= 512 && NV % 64 == 0 && NZ % 64 == 0 && LDS_BYTES <= 160 * 1024, "unsupported tile");
};

template <int WA, int WB, bool UPS = false>
__global__ __launch_bounds__(512) void conv_wino_wgrad8_kernel(WinoWgradArgs a) {
  using C = WinoWgrad8Cfg<WA, WB>;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int NE = UPS ? 9 : 16;  // raw rows of an input item: the 3x3 source patch, or the 4x4 patch
  constexpr int CIB = C::CIB, COB = C::COB, QC = C::QC, NQ = C::NQ, NV = C::NV, NZ = C::NZ;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
  typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
  extern __shared__ __align__(16) unsigned char wwg_smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte*)wwg_smem;
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ps = wave >> 2, wr = wave & 3;  // position set (transform rows 2 ps, 2 ps + 1); place of the pair in the tile
  const int kg = wr / (WA * WB), wa = (wr / WB) % WA, wb = wr % WB;
  const int tiles_b = a.Cout / COB;
  const int a0 = ((int)blockIdx.x / tiles_b) * CIB, b0 = ((int)blockIdx.x % tiles_b) * COB;
  const int H = 1 << a.L, HW = H * H, QI = HW / 4;
  const int PGN = QI >= NQ ? QI / NQ : 1, IPC = QI >= NQ ? 1 : NQ / QI;
  const int n_img = (int)(a.M >> (2 * a.L));
  const int pg = (int)blockIdx.z % PGN, isp = (int)blockIdx.z / PGN;
  const int ichunks = (n_img + IPC - 1) / IPC;
  const int per = (ichunks + a.isplit - 1) / a.isplit;
  const int c_beg = isp * per, c_end = c_beg + per < ichunks ? c_beg + per : ichunks;
  const int nchunks = c_end > c_beg ? c_end - c_beg : 0;

  // ---- this thread's transform item: (quad ql of the chunk, channels 2 c2, 2 c2 + 1) of the input (tid < NV) or of dY
  const bool is_v = wave < NV / 64, is_z = !is_v && wave < (NV + NZ) / 64;
  const int item = is_v ? tid : tid - NV;
  const int ql = is_v ? item / (CIB / 2) : (is_z ? item / (COB / 2) : 0);
  const int c2 = is_v ? item % (CIB / 2) : (is_z ? item % (COB / 2) : 0);
  const int qpos = QI >= NQ ? pg * NQ + ql : ql % QI;
  const int isub = QI >= NQ ? 0 : ql / QI;
  constexpr uint32_t PAD = 0xFFFF0000u;
  uint32_t voff[16];
  {
    const int ty = (int)morton_y((uint32_t)qpos), tx = (int)morton_x((uint32_t)qpos);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (is_v && UPS) {
        const int y = ty - 1 + e / 3, x = tx - 1 + e % 3, Hs = H / 2;
        const bool ok = e < 9 && y >= 0 && y < Hs && x >= 0 && x < Hs;
        voff[e] = ok ? (uint32_t)((isub * (HW / 4) + (int)morton((uint32_t)y, (uint32_t)x)) * a.Cin + a0 + 2 * c2) * 4u : PAD;
      } else if (is_v) {
        const int y = 2 * ty - 1 + (e >> 2), x = 2 * tx - 1 + (e & 3);
        const bool ok = y >= 0 && y < H && x >= 0 && x < H;
        voff[e] = ok ? (uint32_t)((isub * HW + (int)morton((uint32_t)y, (uint32_t)x)) * a.Cin + a0 + 2 * c2) * 4u : PAD;
      } else {
        voff[e] = (e < 4 && is_z) ? (uint32_t)((isub * HW + 4 * qpos + e) * a.Cout + b0 + 2 * c2) * 4u : PAD;
      }
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)0xFFFF0000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)0xFFFF0000u, 0x00020000);
  const int chan0 = 2 * c2;
  const uint32_t wst = lds0 + (is_v ? 0u : (uint32_t)C::V_B) + (uint32_t)((ql * (is_v ? CIB : COB) + chan0) * 64);
  const int wsw = (chan0 >> 2) & 3;  // (the same for the item's two channels)
  // MFMA operand addresses (stage 0): k-step s reads quads kg QC + 2 s + hh; this wave reads the slots of rows 2 ps, 2 ps + 1
  const int rowA = wa * 32 + c, colB = wb * 32 + c;
  uint32_t aaddr[2], baddr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    aaddr[i] = lds0 + (uint32_t)(((kg * QC + hh) * CIB + rowA) * 64 + (((2 * ps + i) ^ ((rowA >> 2) & 3)) << 4));
    baddr[i] = lds0 + (uint32_t)C::V_B + (uint32_t)(((kg * QC + hh) * COB + colB) * 64 + (((2 * ps + i) ^ ((colB >> 2) & 3)) << 4));
  }

  f32x2 raw[16];
  auto issue_loads = [&](int ch, auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;  // 0: input item, 1: gradient item, 2: none
    const int img = (c_beg + ch) * IPC;
    const int soff_x = __builtin_amdgcn_readfirstlane(img * (UPS ? HW / 4 : HW) * a.Cin * 4), soff_y = __builtin_amdgcn_readfirstlane(img * HW * a.Cout * 4);
    const bool live = ch < nchunks && img + isub < n_img;
    if constexpr (ROLE == 0) {
#pragma unroll
      for (int e = 0; e < NE; ++e)
        raw[e] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_x, (int)(live ? voff[e] : PAD), soff_x, 0));
    } else if constexpr (ROLE == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        raw[e] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_y, (int)(live ? voff[e] : PAD), soff_y, 0));
    }
  };
  auto store_entry = [&](int st, int k, const float (&v)[16]) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      if (UPS && x == 2) continue;
      const f32x4 o = {v[x * 4], v[x * 4 + 1], v[x * 4 + 2], v[x * 4 + 3]};
      *reinterpret_cast<lds_f32x4*>((uintptr_t)(wst + st * C::STAGE + k * 64 + (uint32_t)((x ^ wsw) << 4))) = o;
    }
  };
  // The transform in two phases, so that the raw registers are free EARLY: phase 1 consumes every raw value (the first
  // matrix product: B^T d, T s, or the row combinations of the gradient quad) into tt, phase 2 forms the second product
  // and stores the entries.  Between them the loads of the chunk after next are issued: a quarter into the chunk, where
  // they have three quarters of a chunk and the barrier to land (issued behind the whole transform they had a quarter,
  // and every chunk opened with the matrix pipe waiting for memory).
  float tt[2][16];
  auto xform1 = [&](auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if constexpr (ROLE == 0 && UPS) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {  // T s (columns q), T = [1 -1 0; 0 2 0; 0 1 -1]
          tt[k][0 * 3 + q] = raw[0 * 3 + q][k] - raw[1 * 3 + q][k];
          tt[k][1 * 3 + q] = raw[1 * 3 + q][k] + raw[1 * 3 + q][k];
          tt[k][2 * 3 + q] = raw[1 * 3 + q][k] - raw[2 * 3 + q][k];
        }
      } else if constexpr (ROLE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // B^T d
          tt[k][0 * 4 + q] = raw[0 * 4 + q][k] - raw[2 * 4 + q][k];
          tt[k][1 * 4 + q] = raw[1 * 4 + q][k] + raw[2 * 4 + q][k];
          tt[k][2 * 4 + q] = raw[2 * 4 + q][k] - raw[1 * 4 + q][k];
          tt[k][3 * 4 + q] = raw[1 * 4 + q][k] - raw[3 * 4 + q][k];
        }
      } else if constexpr (ROLE == 1) {
        // A dY with A = [1 0; 1 1; 1 -1; 0 -1]: dY = [a b; c d] (rows 4t .. 4t+3 of the quad in Morton order)
        const float qa = raw[0][k], qb = raw[1][k], qc = raw[2][k], qd = raw[3][k];
        tt[k][0] = qa; tt[k][1] = qb; tt[k][2] = qa + qc; tt[k][3] = qb + qd;
        tt[k][4] = qa - qc; tt[k][5] = qb - qd; tt[k][6] = -qc; tt[k][7] = -qd;
      }
    }
  };
  auto xform2 = [&](int st, auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float v[16];
      if constexpr (ROLE == 0 && UPS) {
#pragma unroll
        for (int x = 0; x < 3; ++x) {  // (.) T^T: row x -> transform row (0, 1, 3)[x]
          const int xr = x == 2 ? 3 : x;
          v[xr * 4 + 0] = tt[k][x * 3 + 0] - tt[k][x * 3 + 1];
          v[xr * 4 + 1] = tt[k][x * 3 + 1] + tt[k][x * 3 + 1];
          v[xr * 4 + 2] = 0.f;
          v[xr * 4 + 3] = tt[k][x * 3 + 1] - tt[k][x * 3 + 2];
        }
        v[8] = v[9] = v[10] = v[11] = 0.f;
        store_entry(st, k, v);
      } else if constexpr (ROLE == 0) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {  // (.) B
          v[x * 4 + 0] = tt[k][x * 4 + 0] - tt[k][x * 4 + 2];
          v[x * 4 + 1] = tt[k][x * 4 + 1] + tt[k][x * 4 + 2];
          v[x * 4 + 2] = tt[k][x * 4 + 2] - tt[k][x * 4 + 1];
          v[x * 4 + 3] = tt[k][x * 4 + 1] - tt[k][x * 4 + 3];
        }
        store_entry(st, k, v);
      } else if constexpr (ROLE == 1) {  // (.) A^T
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float pp = tt[k][2 * x], qq = tt[k][2 * x + 1];
          v[4 * x] = pp; v[4 * x + 1] = pp + qq; v[4 * x + 2] = pp - qq; v[4 * x + 3] = -qq;
        }
        store_entry(st, k, v);
      }
    }
  };

  f32x16 acc[8];  // acc[4 i + nu] = dU[2 ps + i][nu] of the wave's 32 x 32 channel pairs
#pragma unroll
  for (int p = 0; p < 8; ++p) acc[p] = (f32x16){0};

  // The chunk loop per (transform role, position set): both are wave-uniform choices made ONCE (see the one-wave form)
  auto run = [&](auto role_c, auto psc) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr int PSC = decltype(psc)::value;  // -1: either set (plain layers: the sets differ in addresses only)
    if (nchunks <= 0) return;
    issue_loads(0, role_c);
    xform1(role_c);
    issue_loads(1, role_c);
    xform2(0, role_c);
    __syncthreads();
    // groups of a chunk: g = 2 ks + i (k-step ks, transform row 2 ps + i); behind the upsample row 2 is identically zero:
    // set 1 runs the odd groups only, and position nu = 2 is skipped everywhere
    constexpr int G0 = (UPS && PSC == 1) ? 1 : 0, GSTEP = (UPS && PSC == 1) ? 2 : 1;
    constexpr int NGRP = (QC - G0 + GSTEP - 1) / GSTEP, NH = NGRP / 2;  // executed groups of a chunk, of its first half
    constexpr int MPG = UPS ? 3 : 4;                                     // MFMAs per group
    auto chunk = [&](int ch, auto stc) {
      constexpr int st = decltype(stc)::value;
      f32x4 ca, cb, na, nb;
      auto load_grp = [&](int g, f32x4& x, f32x4& y) {
        x = *reinterpret_cast<lds_cf32x4*>((uintptr_t)(aaddr[g & 1] + st * C::STAGE + (g >> 1) * 2 * CIB * 64));
        y = *reinterpret_cast<lds_cf32x4*>((uintptr_t)(baddr[g & 1] + st * C::STAGE + (g >> 1) * 2 * COB * 64));
      };
      load_grp(G0, ca, cb);
      na = ca; nb = cb;
      __builtin_amdgcn_sched_barrier(0);
      // first half of the chunk's MFMAs in two quarters: quarter 1 || phase 1 of the transform of the NEXT chunk's raw
      // rows (they were loaded a quarter into the previous chunk), then the loads of the chunk after next, then quarter
      // 2 || phase 2 and its stores into the other stage
      constexpr int NQ1 = NH / 2;
      constexpr int NV1 = ROLE == 0 ? (UPS ? 18 : 32) : (ROLE == 1 ? 12 : 0);   // vector instructions of phase 1 (about)
      constexpr int NV2 = ROLE == 0 ? (UPS ? 18 : 32) : (ROLE == 1 ? 28 : 0);   // ... of phase 2
      constexpr int NST = ROLE == 2 ? 0 : (UPS && ROLE == 0 ? 6 : 8);           // its LDS stores
      auto mfma_groups = [&](int e0, int e1) {
#pragma unroll
        for (int e = e0; e < e1; ++e) {
          const int g = G0 + e * GSTEP;
          if (e + 1 < NGRP) load_grp(g + GSTEP, na, nb);
#pragma unroll
          for (int m = 0; m < 4; ++m)
            if (!UPS || m != 2)
              acc[4 * (g & 1) + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[m], cb[m], acc[4 * (g & 1) + m], 0, 0, 0);
          ca = na; cb = nb;
        }
      };
      mfma_groups(0, NQ1);
      xform1(role_c);
      {
        constexpr int NM = NQ1 * MPG, VPM = NM > 0 ? (NV1 + NM - 1) / NM : 0;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          if (i % MPG == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // the next group's operand reads
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      issue_loads(ch + 2, role_c);
      __builtin_amdgcn_sched_barrier(0);
      mfma_groups(NQ1, NH);
      xform2(st ^ 1, role_c);
      {
        constexpr int NM = (NH - NQ1) * MPG, VPM = (NV2 + NM - 1) / NM;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          if (i % MPG == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
          if (NST > 0 && ((i + 1) * NST) / NM - (i * NST) / NM >= 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // transform stores, spread evenly
          if (NST > 0 && ((i + 1) * NST) / NM - (i * NST) / NM >= 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          if (NST > 0 && ((i + 1) * NST) / NM - (i * NST) / NM >= 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = NH; e < NGRP; ++e) {
        const int g = G0 + e * GSTEP;
        if (e + 1 < NGRP) load_grp(g + GSTEP, na, nb);
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (!UPS || m != 2)
            acc[4 * (g & 1) + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[m], cb[m], acc[4 * (g & 1) + m], 0, 0, 0);
        ca = na; cb = nb;
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    };
    for (int ch = 0; ch < nchunks; ch += 2) {
      chunk(ch, std::integral_constant<int, 0>{});
      chunk(ch + 1, std::integral_constant<int, 1>{});
    }
  };
  using R0 = std::integral_constant<int, 0>;
  using R1 = std::integral_constant<int, 1>;
  using R2 = std::integral_constant<int, 2>;
  using PX = std::integral_constant<int, -1>;
  if constexpr (UPS) {
    if (ps == 0) { if (is_v) run(R0{}, R0{}); else if (is_z) run(R1{}, R0{}); else run(R2{}, R0{}); }
    else { if (is_v) run(R0{}, R1{}); else if (is_z) run(R1{}, R1{}); else run(R2{}, R1{}); }
  } else {
    if (is_v) run(R0{}, PX{}); else if (is_z) run(R1{}, PX{}); else run(R2{}, PX{});
  }

  // ---- raw slab of this (tile, split, K group): planes 8 ps .. 8 ps + 7 of [16][Cin][Cout]
  const size_t slab = ((size_t)blockIdx.z * C::WK + kg) * 16 * (size_t)a.Cin * a.Cout;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (UPS && (j & 3) == 2) continue;               // (vanishing positions: the reduce pass skips them too)
    const int p = 8 * ps + j;
    if (UPS && (p >> 2) == 2) continue;              // (wave-uniform: set 1's first row)
    float* dst = a.slabs + slab + ((size_t)p * a.Cin + a0 + wa * 32) * a.Cout + b0 + wb * 32 + c;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(size_t)crow16(r, hh) * a.Cout] = acc[j][r];
  }
}

// Slab reduction in two passes.  (1) every (position, ci, co) element summed over the slabs by 8 lanes (strided over the
// slabs, fixed-shape shuffle tree: deterministic), IN PLACE into slab 0 -- 16 Cin Cout work items: a one-pass form with one
// item per channel pair had 64 workgroups for the decoder's 64 -> 32 layer, whose 512 slabs it then read for 405 us
// in-situ, with the next layer's weight gradient waiting for the slab buffer behind it.  (2) dg = G^T dU G per channel
// pair from slab 0, written in the checkpoint layout.  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].
__global__ __launch_bounds__(256) void wino_wgrad_slabsum_kernel(float* __restrict__ slabs, int nslabs, int64_t pairs, int ups) {
  const int64_t total = 16 * pairs;
  const int sub = threadIdx.x & 7;
  for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; e < total; e += ((int64_t)gridDim.x * 256) >> 3) {
    const int p = (int)(e / pairs);
    if (ups && ((p >> 2) == 2 || (p & 3) == 2)) continue;  // (upsampled layers: identically zero, never written)
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

__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ du, WeightMap map, float* __restrict__ grad_w, int ups) {
  const int64_t pairs = (int64_t)map.Ca * map.Cb;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < pairs; e += (int64_t)gridDim.x * 256) {
    float u[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) u[p] = (ups && ((p >> 2) == 2 || (p & 3) == 2)) ? 0.f : du[(size_t)p * pairs + e];
    float t[3][4];  // G^T dU: rows r = 0..2 over xi
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      t[0][nu] = u[0 * 4 + nu] + 0.5f * (u[1 * 4 + nu] + u[2 * 4 + nu]);
      t[1][nu] = 0.5f * (u[1 * 4 + nu] - u[2 * 4 + nu]);
      t[2][nu] = 0.5f * (u[1 * 4 + nu] + u[2 * 4 + nu]) + u[3 * 4 + nu];
    }
    const int b = (int)(e % map.Cb), av = (int)(e / map.Cb);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float g0 = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
      const float g1 = 0.5f * (t[r][1] - t[r][2]);
      const float g2 = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
      grad_w[torch_weight_offset(map, 3 * r + 0, av, b)] = g0;
      grad_w[torch_weight_offset(map, 3 * r + 1, av, b)] = g1;
      grad_w[torch_weight_offset(map, 3 * r + 2, av, b)] = g2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- host
// channel tiles: 64 x 64 (one K group), or 32 x 64 / 64 x 32 with two K groups inside the block
static bool wino_wgrad_shape_ok(int64_t M, int Cin, int Cout, int L) {
  if (L < 1 || L > 5 || M <= 0 || (M & (((int64_t)1 << (2 * L)) - 1))) return false;  // whole images
  const bool t22 = Cin % 64 == 0 && Cout % 64 == 0, t12 = Cin == 32 && Cout % 64 == 0, t21 = Cin % 64 == 0 && Cout == 32;
  if (!(t22 || t12 || t21)) return false;
  if (M * (int64_t)(Cin > Cout ? Cin : Cout) * 4 >= 2147483647LL) return false;          // (32-bit buffer offsets)
  return true;
}
bool conv_wino_wgrad_shape(int64_t M, int Cin, int Cout, int L) { return wino_wgrad_shape_ok(M, Cin, Cout, L); }

struct WinoWgradGeom { int tiles, pgn, isplit, wk, nslabs; };
static WinoWgradGeom wino_wgrad_geom(int64_t M, int Cin, int Cout, int L, int cus) {
  WinoWgradGeom g;
  const bool full = Cin % 64 == 0 && Cout % 64 == 0;
  const int cib = Cin % 64 == 0 ? 64 : 32, cob = Cout % 64 == 0 ? 64 : 32, nq = 8;
  g.wk = full ? 1 : 2;
  g.tiles = (Cin / cib) * (Cout / cob);
  const int qi = 1 << (2 * L - 2);
  g.pgn = qi >= nq ? qi / nq : 1;
  const int ipc = qi >= nq ? 1 : nq / qi;
  const int n_img = (int)(M >> (2 * L));
  const int ichunks = (n_img + ipc - 1) / ipc;
  // one workgroup per CU of the budget; every workgroup at least 8 chunks deep
  int want = cus / (g.tiles * g.pgn);
  if (want < 1) want = 1;
  if (want > ichunks / 8) want = ichunks / 8 > 0 ? ichunks / 8 : 1;
  g.isplit = want;
  g.nslabs = g.pgn * g.isplit * g.wk;
  return g;
}

// the weight gradient of a training call takes the Winograd form with the layer's other training launches (option enc_wino:
// -1 from WINO_MIN_BLOCKS workgroups' worth of tile blocks up -- in every operand mode: weight gradients are float32 in
// all of them; 0 never; 1 whenever the shape allows)
bool conv_wino_wgrad_ok(int64_t M, int Cin, int Cout, int L) {
  const int64_t o = opt(OPT_ENC_WINO);
  if (o == 0 || !wino_wgrad_shape_ok(M, Cin, Cout, L)) return false;
  if (o >= 1) return true;
  return conv_wino_ok(M, Cin, Cout, L, 3);
}

size_t conv_wino_wgrad_slab_floats(int64_t M, int Cin, int Cout, int L) {
  const WinoWgradGeom g = wino_wgrad_geom(M, Cin, Cout, L, 256);  // (the largest split any CU budget gives)
  return (size_t)g.nslabs * 16 * Cin * Cout;
}

template <int WA, int WB, bool UPS>
static int launch_wino_wgrad8_cfg(const WinoWgradArgs& a, double flops, dim3 grid, hipStream_t s) {
  using C = WinoWgrad8Cfg<WA, WB>;
  auto kern = conv_wino_wgrad8_kernel<WA, WB, UPS>;
  DVG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
  const unsigned wgs = grid.x * grid.y * grid.z;  // (one workgroup per CU)
  DVG_LAUNCH_WORK_SHARE(K_WGRAD_WINO, flops, (float)(wgs > 256u ? 256u : wgs) / 256.0f, kern, grid, dim3(512), C::LDS_BYTES, s, a);
  return DVG_OK;
}

// ups = 1: `in` is the SOURCE map of an Upsample(x2) + 3x3 layer ([M / 4][Cin]); M, L describe the layer's OUTPUT grid
int launch_conv_wino_wgrad(const float* in, const float* dy, int64_t M, int Cin, int Cout, int L, float* slabs,
                           const WeightMap& map, float* grad_w, hipStream_t s, int ups, int cus) {
  DVG_REQUIRE(wino_wgrad_shape_ok(M, Cin, Cout, L), "conv_wino_wgrad: unsupported launch (M=%lld Cin=%d Cout=%d L=%d)",
              (long long)M, Cin, Cout, L);
  if (cus <= 0) cus = WINO_CUS_ENC_WGRAD;
  if (cus < 1) cus = 1;
  if (cus > 256) cus = 256;
  const WinoWgradGeom g = wino_wgrad_geom(M, Cin, Cout, L, cus);
  WinoWgradArgs a;
  a.in = in; a.dy = dy; a.slabs = slabs; a.M = M; a.Cin = Cin; a.Cout = Cout; a.L = L; a.isplit = g.isplit;
  const double flops = 2.0 * (double)(M / 4) * (ups ? 9.0 : 16.0) * Cin * Cout;  // executed position GEMMs
  const dim3 grid((unsigned)g.tiles, 1, (unsigned)(g.pgn * g.isplit));
  const int cfg = (Cin % 64 == 0 ? 2 : 0) + (Cout % 64 == 0 ? 1 : 0);  // 3: 64x64, 1: 32x64, 2: 64x32
  int rc;
  if (ups) {
    rc = cfg == 3 ? launch_wino_wgrad8_cfg<2, 2, true>(a, flops, grid, s)
       : cfg == 1 ? launch_wino_wgrad8_cfg<1, 2, true>(a, flops, grid, s) : launch_wino_wgrad8_cfg<2, 1, true>(a, flops, grid, s);
  } else {
    rc = cfg == 3 ? launch_wino_wgrad8_cfg<2, 2, false>(a, flops, grid, s)
       : cfg == 1 ? launch_wino_wgrad8_cfg<1, 2, false>(a, flops, grid, s) : launch_wino_wgrad8_cfg<2, 1, false>(a, flops, grid, s);
  }
  DVG_TRY(rc);
  const int64_t pairs = (int64_t)Cin * Cout;
  int64_t blocks = (16 * pairs * 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DVG_LAUNCH(K_WGRAD_REDUCE, wino_wgrad_slabsum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slabs, g.nslabs, pairs, ups);
  DVG_LAUNCH(K_WGRAD_REDUCE, wino_wgrad_reduce_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, (const float*)slabs, map, grad_w, ups);
  return DVG_OK;
}

}  // namespace dvg
