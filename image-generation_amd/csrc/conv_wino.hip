// Winograd F(2x2, 3x3) form of the stride-1 3x3 convolutions of the encoder (forward of layers 1-3 and their data
// gradients: /root/reference/src/encoder.py:28-36), float32 on v_mfma_f32_32x32x2_f32.
//
// Why: the float32 MFMA is the slow matrix instruction of the chip (157 TFLOP/s: 1/16 of the bf16 rate), the strict-f32
// step spends its matrix time in these layers, and the minimal-filtering form needs 16 multiplies per 2x2 output quad and
// channel pair where the direct form needs 36: 2.25x fewer MFMAs for the same float32 convolution (the transforms only
// add and halve).  Morton-ordered activations (conv.h) make it cheap: a 2x2 output quad IS four consecutive rows, a run
// of 64 or 128 quads is a whole number of images (no halo between workgroups), and a lane's MFMA accumulator registers
// hold all 16 transform-domain values of a (quad, output channel) pair, so the output transform is lane-local.
//
//   U[xi][nu][ci][co] = (G g G^T)          weights, once per step (wino_weight_pack_kernel), G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//   V[xi][nu][quad][ci] = (B^T d B)        4x4 input patch d of the quad (zero padded), B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
//   M[xi][nu][quad][co] = sum_ci V U       16 independent GEMMs: the MFMA work
//   Y[quad][2x2][co]    = A^T M A (+bias)  A^T = [1 1 1 0; 0 1 -1 -1]
//
// One workgroup = 8 waves on one CU (145 KB of LDS), two per SIMD; a wave PAIR shares a sub-tile of 32 quads x 32
// channels and splits its 16 transform positions (conv_wino8_body below).  Channels are walked in chunks of KC; per
// chunk three LDS images: the raw input pixels of the block's images [pixel][KC] and the transformed weights
// [k][co][16] arrive by LDS-DMA, the transformed input [k][quad][16] is written by the block itself (each thread
// transforms one patch per chunk, under the previous chunk's MFMAs).  Two stages of each, one barrier per chunk, and
// the chunk sequence runs on across the tile blocks a workgroup owns (persistent grid), so the staging pipeline is
// filled once per workgroup, not once per tile.  A 64-byte entry (16 transform positions of one (k, row)) is read as
// ds_read_b128 whose 16-byte slots are XOR-ed with bits 2-3 of the row: conflict-free for the 16 lanes of a read pass.
// (Rounds 3-4 ran one wave per SIMD with all 16 positions -- 256 accumulator registers -- and left the matrix pipe idle
// for every exposed wait: deleted in round 5, see DESIGN.md.)
#include <atomic>

#include "conv.h"
#include "conv_tile.h"

namespace dvg {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) const float lds_cf32;

struct WinoArgs {
  const float* in;    // [4 tiles][Cin]  (Morton pixel order: quad t = rows 4t .. 4t+3)
  const float* u;     // [Cin][Cout][16] transformed weights, 16-byte slots swizzled by (co >> 2) & 3
  const float* bias;  // [Cout] or null
  float* out;         // [4 tiles][Cout]
  float* stats;       // [nblk][Cout][2] per tile block (sum, sum of squares) of the output, or null
  int cus = 0;        // CUs the persistent grid is sized for (0 = 256)
  int* dyn = nullptr; // [32] zeroed tile counters (dyn[y]: tile blocks handed out beyond the first round; dyn[16 + y]:
                      // workgroups done): tile blocks are then dealt dynamically (see the kernel), else round-robin
  // (UM = 1: `in` is the SOURCE map [tiles][Cin] of an Upsample(x2) + 3x3 layer -- template argument of the kernel)
  int Cin, Cout, L;   // L = log2 of the image side
  int nblk;           // tile blocks (of 32 WM quads)
};

// UM = 1: the decoder's Upsample(x2) + ConvTranspose2d 3x3 layers (/root/reference/src/decoder.py:34-46), forward.  The
// layer is a 3x3 convolution of the nearest-upsampled map, so the 4x4 input patch of the output quad of source pixel
// (i, j) is d[u][v] = s[i + r(u)][j + r(v)], r = (-1, 0, 0, +1), and B^T maps (x-, x0, x0, x+) to (x- - x0, 2 x0, 0,
// x0 - x+): transform position 2 vanishes in both directions, NINE of the sixteen position GEMMs remain (the folded
// direct form multiplies sixteen (class, tap) pairs per source pixel).
//
// Two waves per SIMD: EIGHT waves per workgroup, the 16 (9) transform positions of a 32 quad x 32 channel sub-tile
// split over a wave PAIR -- wave w and wave w + 4 land on the same SIMD, read the same transformed operands and hold
// the transform rows xi in {0, 1} and {2, 3}: 8 accumulator tiles = 128 registers each, so the two fit the register
// file side by side and one wave's operand waits, LDS-DMA issue and transform arithmetic fall under the other's MFMAs
// (the one-wave form of rounds 3-4 left the matrix pipe idle for every exposed ds_read: PMC busy 0.21-0.37 in a
// training step).  Every thread transforms ONE patch per chunk (512 threads), and the output transform Y = A^T M A needs rows of M from
// both waves of a pair, so each wave first applies the column half (C_xi = M[xi][.] A, lane-local as before), the pair
// swaps ONE row of C through LDS (two rounds of 4 KB per wave in the transformed-input stage the block's last chunk has
// just finished with), and each wave then owns one output row of the quad: wave set 0 the pixels 4 q, 4 q + 1, set 1 the
// pixels 4 q + 2, 4 q + 3.  Behind the upsample (UM = 1: xi, nu in {0, 1, 3}) set 0 holds xi = 0, 1 (6 MFMAs per k-step)
// and set 1 xi = 3 (3 MFMAs): unequal, but they share one matrix pipe.
template <int WM, int WN, int KC, int UM = 0>
struct Wino8Cfg {
  static constexpr int TBLK = 32 * WM, CB = 32 * WN, KS = KC / 2;
  static constexpr int PPQ = UM == 1 ? 1 : 4;                // raw pixels per quad (UM = 1: one source pixel)
  static constexpr int RAW_PIX = TBLK * PPQ * KC * 4;        // bytes of raw pixels per stage
  static constexpr int RAW_PIECES = RAW_PIX / 1024;          // 1 KiB DMA pieces (one wave instruction each)
  static constexpr int RAW_B = 64 + RAW_PIX;                 // + the zero entry padding taps read
  static constexpr int V_B = KC * TBLK * 64, U_B = KC * CB * 64;
  static constexpr int U_PIECES = U_B / 1024;
  static constexpr int OFF_RAW = 0, OFF_V = OFF_RAW + 2 * RAW_B, OFF_U = OFF_V + 2 * V_B, OFF_RED = OFF_U + 2 * U_B;
  static constexpr int OFF_NEXT = OFF_RED + 2 * WM * CB * 2 * 4;
  static constexpr int LDS_BYTES = OFF_NEXT + 16;
  static_assert(WM * WN == 4 && TBLK * KC == 512 && RAW_PIX % 1024 == 0 && U_PIECES % 8 == 0 && V_B >= 8 * 4096, "unsupported shape");
};

template <int WM, int WN, int KC, int UM, int TAIL>
__device__ __forceinline__ void conv_wino8_body(const WinoArgs& a, unsigned char* wsm) {
  using C = Wino8Cfg<WM, WN, KC, UM>;
  static_assert(UM >= 0 && UM <= 2, "plain 3x3 layer (forward / data gradient); forward / data gradient behind an upsample");
  constexpr bool SK = UM != 0;  // upsampled forms: transform row / column 2 is never multiplied (9 of 16 positions)
  constexpr int NP = UM == 1 ? 9 : 16;  // raw pixels of a patch
  constexpr int TBLK = C::TBLK, CB = C::CB, KS = C::KS;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte_t*)wsm;
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ps = wave >> 2, pw = wave & 3;  // position set (transform rows 2 ps, 2 ps + 1) and sub-tile of the pair
  const int wm = pw / WN, wn = pw % WN;
  const int n0 = blockIdx.y * CB;
  const int H = 1 << a.L, HW = H * H;

  // ---- the thread's patch: quad tl of the tile block, channel k0 of the chunk (rotated by the quad so that the 32 lanes
  // of a read spread over the banks); a padding tap reads the stage's zero entry
  const int tl = tid % TBLK, k0 = ((tid / TBLK) + tl / (64 / KC)) % KC;
  int poff[16];
  if constexpr (UM == 1) {
    const int img = (tl * 4) / HW, tq = tl - img * (HW / 4), Hs = H / 2;
    const int ty = (int)morton_y((uint32_t)tq), tx = (int)morton_x((uint32_t)tq);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int y = ty - 1 + e / 3, x = tx - 1 + e % 3;
      const bool ok = e < 9 && y >= 0 && y < Hs && x >= 0 && x < Hs;
      const int pos = img * (HW / 4) + (int)morton((uint32_t)(ok ? y : 0), (uint32_t)(ok ? x : 0));
      poff[e] = ok ? 64 + (pos * KC + k0) * 4 : 0;
    }
  } else {
    const int img = (tl * 4) / HW, tq = tl - img * (HW / 4);
    const int ty = (int)morton_y((uint32_t)tq), tx = (int)morton_x((uint32_t)tq);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = 2 * ty - 1 + i, x = 2 * tx - 1 + j;
        const bool ok = y >= 0 && y < H && x >= 0 && x < H;
        const int px = img * HW + (int)morton((uint32_t)(ok ? y : 0), (uint32_t)(ok ? x : 0));
        const int pos = (px & 3) * TBLK + (px >> 2);
        poff[i * 4 + j] = ok ? 64 + (pos * KC + k0) * 4 : 0;
      }
  }
  const uint32_t vst = lds0 + C::OFF_V + (uint32_t)((k0 * TBLK + tl) * 64);
  const int swt = (tl >> 2) & 3;
  // MFMA operands: entry (k, row), slot xi at (xi ^ ((row >> 2) & 3)) << 4; this wave reads xi = 2 ps and 2 ps + 1
  const int rowA = wm * 32 + c, colB = wn * 32 + c;
  uint32_t aaddr[2], baddr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    aaddr[i] = lds0 + C::OFF_V + (uint32_t)((hh * TBLK + rowA) * 64 + (((2 * ps + i) ^ ((rowA >> 2) & 3)) << 4));
    baddr[i] = lds0 + C::OFF_U + (uint32_t)((hh * CB + colB) * 64 + (((2 * ps + i) ^ ((colB >> 2) & 3)) << 4));
  }
  if (tid < 32) *reinterpret_cast<float*>(wsm + C::OFF_RAW + (tid >> 4) * C::RAW_B + (tid & 15) * 4) = 0.f;

  // ---- DMA of one chunk's images: 1 KiB pieces, piece p by wave p % 8; scalar offsets carry the chunk
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in), 0, (int)((int64_t)a.nblk * TBLK * C::PPQ * a.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.u), 0, (int)((int64_t)a.Cin * a.Cout * 64), 0x00020000);
  constexpr int RPW = (C::RAW_PIECES + 7) / 8, UPW = C::U_PIECES / 8;  // pieces per wave
  int rvoff[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int byte = ((wave + 8 * q) * 64 + lane) * 16, pos = byte / (KC * 4);
    if constexpr (UM == 1) rvoff[q] = pos * a.Cin * 4 + byte % (KC * 4);
    else rvoff[q] = (4 * (pos % TBLK) + pos / TBLK) * a.Cin * 4 + byte % (KC * 4);
  }
  int usoff[UPW];  // scalar part of a weight piece's source: row kk of the chunk, bytes `within` of that row's CB entries
#pragma unroll
  for (int q = 0; q < UPW; ++q) {
    const int off = (wave + 8 * q) * 1024, kk = off / (CB * 64), within = off % (CB * 64);
    usoff[q] = __builtin_amdgcn_readfirstlane((kk * a.Cout + n0) * 64 + within);
  }
  const int nch = a.Cin / KC;
  auto issue_raw = [&](int blk, int ch, int st) {
    const int soff = __builtin_amdgcn_readfirstlane((blk * TBLK * C::PPQ * a.Cin + ch * KC) * 4);
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      if (wave + 8 * q < C::RAW_PIECES) {
        const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + C::OFF_RAW + st * C::RAW_B + 64 + (wave + 8 * q) * 1024));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_void_t*)(uintptr_t)dst, 16, rvoff[q], soff, 0, 0);
      }
    }
  };
  auto issue_u_piece = [&](int ch, int st, int q) {
    const int sbase = __builtin_amdgcn_readfirstlane(ch * KC * a.Cout * 64);
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + C::OFF_U + st * C::U_B + (wave + 8 * q) * 1024));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_u, (lds_void_t*)(uintptr_t)dst, 16, lane * 16, sbase + usoff[q], 0, 0);
  };
  auto issue_u = [&](int ch, int st) {
#pragma unroll
    for (int q = 0; q < UPW; ++q) issue_u_piece(ch, st, q);
  };

  // ---- input transform of the thread's patch: raw stage `rs` -> transformed stage `vs`
  auto load_patch = [&](int rs, float (&d)[16]) {
#pragma unroll
    for (int e = 0; e < NP; ++e)
      d[e] = *reinterpret_cast<lds_cf32*>((uintptr_t)(lds0 + C::OFF_RAW + rs * C::RAW_B + (uint32_t)poff[e]));
  };
  // B^T d, column m (UM = 1: T s, column m of 3)
  auto xform_col = [&](int m, const float (&d)[16], float (&t)[16]) {
    if constexpr (UM == 2) {  // rows 0, 1, 3 of B^T d; piece m is column (0, 1, 3)[m], piece 0 also column 2 (an input of every row)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        if (cc == 1 && m != 0) continue;
        const int q = cc == 1 ? 2 : (m == 2 ? 3 : m);
        t[0 * 4 + q] = d[0 * 4 + q] - d[2 * 4 + q];
        t[1 * 4 + q] = d[1 * 4 + q] + d[2 * 4 + q];
        t[3 * 4 + q] = d[1 * 4 + q] - d[3 * 4 + q];
      }
    } else if constexpr (UM == 1) {
      t[0 * 3 + m] = d[0 * 3 + m] - d[1 * 3 + m];
      t[1 * 3 + m] = d[1 * 3 + m] + d[1 * 3 + m];
      t[2 * 3 + m] = d[1 * 3 + m] - d[2 * 3 + m];
    } else {
      t[0 * 4 + m] = d[0 * 4 + m] - d[2 * 4 + m];
      t[1 * 4 + m] = d[1 * 4 + m] + d[2 * 4 + m];
      t[2 * 4 + m] = d[2 * 4 + m] - d[1 * 4 + m];
      t[3 * 4 + m] = d[1 * 4 + m] - d[3 * 4 + m];
    }
  };
  // (.) B, row m, and its store (UM = 1: row m of 3 -> transform row (0, 1, 3)[m])
  auto xform_row_store = [&](int vs, int m, const float (&t)[16]) {
    if constexpr (UM == 2) {  // (.) B on row xr = (0, 1, 3)[m]: columns 0, 1, 3
      const int xr = m == 2 ? 3 : m;
      const f32x4 o = {t[xr * 4 + 0] - t[xr * 4 + 2], t[xr * 4 + 1] + t[xr * 4 + 2], 0.f, t[xr * 4 + 1] - t[xr * 4 + 3]};
      *reinterpret_cast<lds_f32x4*>((uintptr_t)(vst + vs * C::V_B + (uint32_t)((xr ^ swt) << 4))) = o;
    } else if constexpr (UM == 1) {
      const f32x4 o = {t[m * 3 + 0] - t[m * 3 + 1], t[m * 3 + 1] + t[m * 3 + 1], 0.f, t[m * 3 + 1] - t[m * 3 + 2]};
      const int xr = m == 2 ? 3 : m;
      *reinterpret_cast<lds_f32x4*>((uintptr_t)(vst + vs * C::V_B + (uint32_t)((xr ^ swt) << 4))) = o;
    } else {
      const f32x4 o = {t[m * 4 + 0] - t[m * 4 + 2], t[m * 4 + 1] + t[m * 4 + 2], t[m * 4 + 2] - t[m * 4 + 1], t[m * 4 + 1] - t[m * 4 + 3]};
      *reinterpret_cast<lds_f32x4*>((uintptr_t)(vst + vs * C::V_B + (uint32_t)((m ^ swt) << 4))) = o;
    }
  };
  constexpr int NPIECE = SK ? 3 : 4;  // pieces of each transform half (columns, then rows)

  f32x16 acc[8];  // acc[4 i + nu] = M[2 ps + i][nu] of the wave's 32 quads x 32 channels

  const bool dynq = a.dyn != nullptr;
  volatile int* nslot = reinterpret_cast<volatile int*>(wsm + C::OFF_NEXT);
  auto finish = [&]() {
    if (dynq && tid == 0 && atomicAdd(a.dyn + 16 + blockIdx.y, 1) == (int)gridDim.x - 1) {
      atomicExch(a.dyn + blockIdx.y, 0);
      atomicExch(a.dyn + 16 + blockIdx.y, 0);
    }
  };
  int blk_cur = (int)blockIdx.x, blk_nxt = blk_cur + (int)gridDim.x;
  if (dynq) {
    if (tid == 0) { nslot[0] = atomicAdd(a.dyn + blockIdx.y, 1); nslot[1] = atomicAdd(a.dyn + blockIdx.y, 1); }
    __syncthreads();
    blk_cur = __builtin_amdgcn_readfirstlane(nslot[0]);
    blk_nxt = __builtin_amdgcn_readfirstlane(nslot[1]);
    __syncthreads();
  }
  bool has_next = blk_nxt < a.nblk;
  if (blk_cur >= a.nblk) { finish(); return; }
  issue_raw(blk_cur, 0, 0);
  issue_u(0, 0);
  issue_raw(blk_cur, 1 % nch, 1);
  __syncthreads();  // (the workgroup fence waits for the LDS-DMA pieces: see the chunk's end)
  {
    float d[16], t[16];
    load_patch(0, d);
#pragma unroll
    for (int m = 0; m < NPIECE; ++m) xform_col(m, d, t);
#pragma unroll
    for (int m = 0; m < NPIECE; ++m) xform_row_store(0, m, t);
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(
      a.out, 0, (int)((int64_t)a.nblk * TBLK * (UM == 2 ? 1 : 4) * a.Cout * 4), 0x00020000);
  const int ovoff = (16 * hh * a.Cout + n0 + wn * 32 + c) * 4;  // quad row 4 hh of the lane's first quad: 4 pixels per quad
  const int ovoff1 = (4 * hh * a.Cout + n0 + wn * 32 + c) * 4;  // (UM = 2: one row per quad)
  (void)ovoff1;
  const float bias = a.bias ? a.bias[n0 + wn * 32 + c] : 0.f;   // (the lane's output channel is the same for every tile block)
  const uint32_t xaddr = lds0 + C::OFF_V + C::V_B + (uint32_t)(lane * 16);  // exchange slots: stage 1 of the transformed input

  // outputs of the tile block just finished (pixels 2 ps, 2 ps + 1 of the wave's 32 quads x 32 channels), stored beside the
  // next block's first MFMAs (or behind the last block): store idx = 2 r + q is accumulator row r of pixel 2 ps + q
  f32x16 py0, py1;
  int pblk = 0;
  constexpr int NSTORE = UM == 2 ? 16 : 32;  // (UM = 2: ONE output row per quad, held by wave set 0)
  auto store_pending = [&](int idx) {
    // (the row index is kept opaque: distributed over the sum, its constant parts x Cout would be hoisted out of the
    // tile-block loop into 32 scalar registers, which then spill)
    if constexpr (UM == 2) {  // (wave set 0 only: the callers know their set at compile time)
      const int rq = (idx & 3) + 8 * (idx >> 2);
      int row = __builtin_amdgcn_readfirstlane(pblk * TBLK + wm * 32);
      asm volatile("" : "+s"(row));
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(py0[idx]), rsrc_out, ovoff1, (row + rq) * a.Cout * 4, 0);
    } else {
      const int r = idx >> 1, q = idx & 1;
      const int rq = 4 * ((r & 3) + 8 * (r >> 2));
      int row = __builtin_amdgcn_readfirstlane((pblk * TBLK + wm * 32) * 4 + 2 * ps);
      asm volatile("" : "+s"(row));
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(q == 0 ? py0[r] : py1[r]), rsrc_out, ovoff, (row + rq + q) * a.Cout * 4, 0);
    }
  };

  // the epilogue of wave set PSC (compile-time copy of ps)
  auto epilogue = [&](int blk, auto psc) {
    constexpr int PSC = decltype(psc)::value;
    if constexpr (UM == 2) {
      // data gradient behind the upsample: the gradient of the source pixel is the sum of its quad's four fine-grid
      // gradients = sum_xi,nu w_xi w_nu M[xi][nu], w = A 1 = (1, 2, 0, -1).  Set 0 holds rows 0, 1, set 1 row 3: set 1
      // gives its row sum r3, set 0 forms (r0 + 2 r1) - r3 and keeps the ONE output row per quad.
      const f32x16 rr = (acc[4] + (acc[5] + acc[5])) - acc[7];  // row 2 ps + 1: xi = 1 (set 0) or 3 (set 1)
      const uint32_t mine = xaddr + (uint32_t)(wave * 4096), theirs = xaddr + (uint32_t)((wave ^ 4) * 4096);
      if constexpr (PSC == 1) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 o = {rr[4 * r4], rr[4 * r4 + 1], rr[4 * r4 + 2], rr[4 * r4 + 3]};
          *reinterpret_cast<lds_f32x4*>((uintptr_t)(mine + r4 * 1024)) = o;
        }
      }
      __syncthreads();
      if constexpr (PSC == 0) {
        const f32x16 r0 = (acc[0] + (acc[1] + acc[1])) - acc[3];
        f32x16 got;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 o = *reinterpret_cast<lds_cf32x4*>((uintptr_t)(theirs + r4 * 1024));
          got[4 * r4] = o[0]; got[4 * r4 + 1] = o[1]; got[4 * r4 + 2] = o[2]; got[4 * r4 + 3] = o[3];
        }
        const f32x16 y = (r0 + (rr + rr)) - got;
#pragma unroll
        for (int r = 0; r < 16; ++r) py0[r] = y[r] + bias;
      }
      pblk = blk;
      return;  // (the barrier behind the read is the caller's)
    }
    // column half of the output transform on the wave's own rows: C_i[b] = sum_nu M[2 ps + i][nu] A[nu][b]
    f32x16 c00 = {0}, c01 = {0}, c10, c11;  // c<i><b>
    if constexpr (UM == 1) {
      if constexpr (PSC == 0) { c00 = acc[0] + acc[1]; c01 = acc[1] - acc[3]; }
      c10 = acc[4] + acc[5]; c11 = acc[5] - acc[7];
    } else {
      c00 = (acc[0] + acc[1]) + acc[2]; c01 = (acc[1] - acc[2]) - acc[3];
      c10 = (acc[4] + acc[5]) + acc[6]; c11 = (acc[5] - acc[6]) - acc[7];
    }
    // the pair swaps one row: set 0 gives C_1 (its i = 1) and takes C_2; set 1 gives C_2 (its i = 0) and takes C_1.
    // (UM = 1: C_2 = 0 -- set 1 gives nothing.)  Both halves b of the row in ONE round where the free LDS holds 8 KB per
    // wave (the 64 x 64 tile: the transformed-input and the weight stage the last chunk finished with, 32 KB each), else
    // in two rounds of 4 KB.  The barrier behind the LAST read is the caller's (run: shared with the block hand-over).
    f32x16 y0, y1;  // the wave's output row a = PSC: pixels 2 a + 0, 2 a + 1 of every quad
    constexpr bool ONE_ROUND = C::V_B >= 4 * 8192 && C::U_B >= 4 * 8192;
    auto slot_of = [&](int w) -> uint32_t {  // exchange slot of wave w (its lane's 16 bytes of piece 0)
      if constexpr (ONE_ROUND) return lds0 + (uint32_t)((w < 4 ? C::OFF_V + C::V_B : C::OFF_U + C::U_B) + (w & 3) * 8192 + lane * 16);
      else return xaddr + (uint32_t)(w * 4096);
    };
    const uint32_t mine = slot_of(wave), theirs = slot_of(wave ^ 4);
    auto put = [&](const f32x16& give, int off) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 o = {give[4 * r4], give[4 * r4 + 1], give[4 * r4 + 2], give[4 * r4 + 3]};
        *reinterpret_cast<lds_f32x4*>((uintptr_t)(mine + off + r4 * 1024)) = o;
      }
    };
    auto take = [&](int off) -> f32x16 {
      f32x16 got;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 o = *reinterpret_cast<lds_cf32x4*>((uintptr_t)(theirs + off + r4 * 1024));
        got[4 * r4] = o[0]; got[4 * r4 + 1] = o[1]; got[4 * r4 + 2] = o[2]; got[4 * r4 + 3] = o[3];
      }
      return got;
    };
    constexpr bool GIVES = !(UM == 1 && PSC == 1), TAKES = !(UM == 1 && PSC == 0);
    if constexpr (ONE_ROUND) {
      if constexpr (GIVES) { put(PSC == 0 ? c10 : c00, 0); put(PSC == 0 ? c11 : c01, 4096); }
      __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if constexpr (!ONE_ROUND) {
        if (b == 1) __syncthreads();  // (round 0's slots are read: they may be rewritten)
        if constexpr (GIVES) put(PSC == 0 ? (b == 0 ? c10 : c11) : (b == 0 ? c00 : c01), 0);
        __syncthreads();
      }
      f32x16 got;
      if constexpr (TAKES) got = take(ONE_ROUND ? b * 4096 : 0);
      f32x16 yb;
      if constexpr (UM == 1) {
        if constexpr (PSC == 0) yb = (b == 0 ? c00 : c01) + (b == 0 ? c10 : c11);  // Y[0][b] = C_0 + C_1
        else yb = got - (b == 0 ? c10 : c11);                                      // Y[1][b] = C_1 - C_3
      } else {
        if constexpr (PSC == 0) yb = ((b == 0 ? c00 : c01) + (b == 0 ? c10 : c11)) + got;  // Y[0][b] = (C_0 + C_1) + C_2
        else yb = (got - (b == 0 ? c00 : c01)) - (b == 0 ? c10 : c11);                      // Y[1][b] = (C_1 - C_2) - C_3
      }
      if (b == 0) y0 = yb; else y1 = yb;
    }
    // (+ bias; the stores wait: they are issued beside the MFMAs of the next tile block's first chunk -- store_pending --
    // all workgroups reach their epilogues together, and 64 KB per workgroup stored at once is a burst the memory
    // system serialises: 1.8 us of an 8-chunk block's 20 us, measured by switching the stores off)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v0 = y0[r] + bias, v1 = y1[r] + bias;
      py0[r] = v0; py1[r] = v1;
      s1 += v0; s2 = fmaf(v0, v0, s2);
      s1 += v1; s2 = fmaf(v1, v1, s2);
    }
    pblk = blk;
    if (a.stats) {  // (the per-wave column sums; summed over the block behind the caller's barrier: stats_finish)
      float* red = reinterpret_cast<float*>(wsm + C::OFF_RED);
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      const int slot = ((PSC * WM + wm) * CB + wn * 32 + c) * 2;
      if (hh == 0) { red[slot] = s1; red[slot + 1] = s2; }
    }
  };
  auto stats_finish = [&](int blk) {
    if (a.stats && tid < CB) {  // (waves 0 .. CB / 64 - 1: set 0)
      const float* red = reinterpret_cast<const float*>(wsm + C::OFF_RED);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 2 * WM; ++w) { t1 += red[(w * CB + tid) * 2]; t2 += red[(w * CB + tid) * 2 + 1]; }
      float* dst = a.stats + ((size_t)blk * a.Cout + n0 + tid) * 2;
      dst[0] = t1; dst[1] = t2;
    }
  };

  // One chunk of wave set PSC: KS k-steps x 2 transform rows (groups) x 4 (3) MFMAs as one pinned stream -- the operands
  // of the next group are requested before this group's MFMAs, the thread's patch of the NEXT chunk is read at the start
  // and transformed beside the MFMAs of the second and third executed groups, a piece per MFMA.
  // TAIL = 1: the chunk's LAST group of MFMAs is issued BEHIND the chunk barrier -- its operands are in registers when
  // the wave arrives there -- between the next chunk's first operand reads and its LDS-DMA pieces, so the matrix pipe
  // has 4 (3) MFMAs to run while the barrier releases, the DMA instructions issue and the first ds_reads of the new
  // chunk travel; the first chunk of a tile block has no such tail in front (FIRST), the last one flushes its own behind
  // its barrier (LAST), in front of the epilogue.
  f32x4 ta = {0.f, 0.f, 0.f, 0.f}, tb = {0.f, 0.f, 0.f, 0.f};
  auto chunk = [&](int cix, auto stc, auto psc, auto firstc, auto lastc, auto pendc) {  // chunk cix of tile block blk_cur
    constexpr int st = decltype(stc)::value;
    constexpr int PSC = decltype(psc)::value;
    constexpr bool FIRST = decltype(firstc)::value != 0, LAST = decltype(lastc)::value != 0, PEND = decltype(pendc)::value != 0;
    // what the chunk prefetches: the weights of the next chunk (past the block's end: chunk 0 again), the raw pixels of
    // the chunk after next (past the end: the next block's, or -- no next block -- this block's last chunk once more,
    // into a stage nobody reads)
    const int u_ch = cix + 1 < nch ? cix + 1 : 0;
    const bool over = cix + 2 >= nch;
    const int r_ch = !over ? cix + 2 : (has_next ? cix + 2 - nch : nch - 1);
    const int r_blk = over && has_next ? blk_nxt : blk_cur;
    f32x4 ca, cb, na, nb;
    auto load_grp = [&](int g, f32x4& x, f32x4& y) {  // group g = 2 ks + i: transform row 2 ps + i of k-step ks
      x = *reinterpret_cast<lds_cf32x4*>((uintptr_t)(aaddr[g & 1] + st * C::V_B + (g >> 1) * 2 * TBLK * 64));
      y = *reinterpret_cast<lds_cf32x4*>((uintptr_t)(baddr[g & 1] + st * C::U_B + (g >> 1) * 2 * CB * 64));
    };
    // (UM = 1, set 1: transform row 2 is identically zero -- only the odd groups exist)
    constexpr int G0 = (SK && PSC == 1) ? 1 : 0, GSTEP = (SK && PSC == 1) ? 2 : 1;
    constexpr int NG = (2 * KS - G0 + GSTEP - 1) / GSTEP;  // executed groups of the chunk
    constexpr int NOWN = TAIL ? NG - 1 : NG;               // ... in front of the chunk barrier
    constexpr int EC = NOWN >= 3 ? 1 : 0;                  // the group that carries the column half of the transform
    static_assert(NOWN >= 2, "the transform needs two executed groups per chunk");
    float d[16], t[16];
    if constexpr (!TAIL) {
      issue_u(u_ch, st ^ 1);
      issue_raw(r_blk, r_ch, st);
      load_grp(G0, ca, cb);
      load_patch(st ^ 1, d);
    } else {
      load_grp(G0, ca, cb);
      load_patch(st ^ 1, d);
      __builtin_amdgcn_sched_barrier(0);
      int q = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (SK && m == 2) continue;
        if constexpr (!FIRST) acc[4 + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[m], tb[m], acc[4 + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q < UPW) issue_u_piece(u_ch, st ^ 1, q);
        ++q;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (; q < UPW; ++q) issue_u_piece(u_ch, st ^ 1, q);
      issue_raw(r_blk, r_ch, st);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < NOWN; ++e) {  // e: ordinal of the group among the executed ones
      const int g = G0 + e * GSTEP;
      if (g + GSTEP < 2 * KS) load_grp(g + GSTEP, na, nb);
      __builtin_amdgcn_sched_barrier(0);
      int piece = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (SK && m == 2) continue;
        if (e == EC) xform_col(piece, d, t);
        if (e == EC + 1) xform_row_store(st ^ 1, piece, t);
        __builtin_amdgcn_sched_barrier(0);
        // (the first chunk's first product into a tile starts from the constant 0: no pass that zeroes 128 registers)
        if (TAIL && FIRST && e < 2 / GSTEP) acc[4 * (g & 1) + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[m], cb[m], (f32x16){0}, 0, 0, 0);
        else acc[4 * (g & 1) + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[m], cb[m], acc[4 * (g & 1) + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PEND) {
          // the previous tile block's 32 output stores beside the MFMAs of the first two groups -- the ones that start
          // from the constant 0: a store's register is free again before the accumulator tiles come alive, and the
          // stores have the rest of the chunk to complete
          constexpr int MPG = SK ? 3 : 4, NMS = 2 * MPG, SPM = (NSTORE + NMS - 1) / NMS;
          const int mi = e * MPG + piece;
          if (mi < NMS && !(UM == 2 && PSC == 1)) {
#pragma unroll
            for (int idx = mi * SPM; idx < (mi + 1) * SPM && idx < NSTORE; ++idx) store_pending(idx);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        ++piece;
      }
      ca = na; cb = nb;
    }
    if constexpr (TAIL) { ta = ca; tb = cb; }  // (the last group is always an odd one: transform row 2 ps + 1)
    // The chunk barrier.  hipcc counts the LDS-DMA pieces (buffer_load ... lds builtins) on vmcnt and the workgroup fence
    // of __syncthreads() waits for exactly them: `s_waitcnt vmcnt(N)` with N = the vector-memory operations issued BEHIND
    // the last piece -- the output stores of the previous tile block, which may stay in flight across the barrier.  (An
    // inline-asm vmcnt(0) here, as in the one-wave form, drains those stores too and hides the wait from the compiler,
    // which then repeats it in front of the next chunk's first ds_read.)
    __syncthreads();
    if constexpr (TAIL && LAST) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (SK && m == 2) continue;
        acc[4 + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[m], tb[m], acc[4 + m], 0, 0, 0);
      }
    }
  };

  auto run = [&](auto psc) {
    constexpr int PSC = decltype(psc)::value;
    int par = 1;
    bool go = true, pend = false;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    while (go) {
      int fetched = 0;
      if (dynq && tid == 0 && has_next) fetched = atomicAdd(a.dyn + blockIdx.y, 1);
      if constexpr (TAIL) {
        // (first, middle pairs, last: the middle chunks run stage 1, stage 0 in turn -- an even chunk count)
        if (pend) chunk(0, I0{}, psc, I1{}, I0{}, I1{});
        else chunk(0, I0{}, psc, I1{}, I0{}, I0{});
        for (int ch = 2; ch < nch; ch += 2) {
          chunk(ch - 1, I1{}, psc, I0{}, I0{}, I0{});
          chunk(ch, I0{}, psc, I0{}, I0{}, I0{});
        }
        chunk(nch - 1, I1{}, psc, I0{}, I1{}, I0{});
      } else {
        if (pend && !(UM == 2 && PSC == 1)) {
#pragma unroll
          for (int idx = 0; idx < NSTORE; ++idx) store_pending(idx);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[p] = (f32x16){0};
        for (int ch = 0; ch < nch; ch += 2) {
          chunk(ch, I0{}, psc, I0{}, I0{}, I0{});
          chunk(ch + 1, I1{}, psc, I0{}, I0{}, I0{});
        }
      }
      epilogue(blk_cur, psc);
      pend = true;
      // ONE barrier per tile block behind the epilogue: the exchange slots have been read (the next chunk's transform and
      // weight DMA overwrite them), the per-wave BatchNorm partials are in LDS, the next block's id is handed round
      // (dynamic deal: two slots in turn, a slot is rewritten two blocks later)
      if (dynq) {
        par ^= 1;
        if (tid == 0) nslot[par] = fetched;
      }
      __syncthreads();
      stats_finish(blk_cur);
      go = has_next;
      blk_cur = blk_nxt;
      if (dynq) blk_nxt = __builtin_amdgcn_readfirstlane(nslot[par]);
      else blk_nxt = blk_cur + (int)gridDim.x;
      has_next = blk_nxt < a.nblk;
    }
    if constexpr (!(UM == 2 && PSC == 1)) {
#pragma unroll
      for (int idx = 0; idx < NSTORE; ++idx) store_pending(idx);  // (the last block's)
    }
  };
  // (one instantiation per wave set: which rows of M a wave holds decides its MFMA pattern behind the upsample and its
  // half of the epilogue; the sets meet at every workgroup barrier -- both paths execute the same number of them)
  if (ps == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
  finish();
}

template <int WM, int WN, int KC, int UM = 0, int TAIL = 1>
__global__ __launch_bounds__(512) void conv_wino8_kernel(WinoArgs a) {
  extern __shared__ __align__(16) unsigned char wino_smem[];
  conv_wino8_body<WM, WN, KC, UM, TAIL>(a, wino_smem);
}

// U = G g G^T of every (reduction channel a, output channel b) pair, in the layout the kernel's DMA copies verbatim
__global__ __launch_bounds__(256) void wino_weight_pack_kernel(const float* __restrict__ w, WeightMap map, float* __restrict__ u) {
  const uint32_t total = (uint32_t)map.Ca * (uint32_t)map.Cb;
  for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) wino_pack_entry(w, map, e, u);
}

int launch_wino_weight_pack(const float* w, const WeightMap& map, float* u, hipStream_t s) {
  int64_t gx = ceil_div((int64_t)map.Ca * map.Cb, 256);
  if (gx > 1024) gx = 1024;
  DVG_LAUNCH(K_WEIGHT_PACK, wino_weight_pack_kernel, dim3((unsigned)gx), dim3(256), 0, s, w, map, u);
  return DVG_OK;
}

// Which tile shape serves a layer: 64 quads x 64 channels (8-channel chunks) or 128 quads x 32 channels (4-channel chunks)
static int wino_cfg(int Cout) { return Cout % 64 == 0 ? 0 : 1; }
static int wino_tblk(int cfg) { return cfg == 0 ? 64 : 128; }

static bool wino_shape_ok(int64_t M, int Cin, int Cout, int L) {
  if (L < 1 || L > 4 || Cout % 32 || Cin % (Cout % 64 == 0 ? 16 : 8) || M % 4) return false;  // (an even number of chunks)
  const int cfg = wino_cfg(Cout);
  const int64_t tiles = M / 4;
  if (tiles % wino_tblk(cfg)) return false;                      // whole tile blocks ...
  if ((wino_tblk(cfg) * 4) % (1 << (2 * L))) return false;       // ... of whole images
  if (M * (int64_t)(Cin > Cout ? Cin : Cout) * 4 >= 2147483647LL) return false;  // (32-bit buffer offsets)
  return true;
}

bool conv_wino_shape(int64_t M, int Cin, int Cout, int L) { return wino_shape_ok(M, Cin, Cout, L); }

// kind: 0 forward launch of a training call, 1 data-gradient launch, 2 forward launch of an evaluation call, 3 weight
// gradient.  option enc_wino: -1 (default) by size -- evaluation-mode forward launches of 256 workgroups' worth of tile
// blocks or more, training launches of WINO_MIN_BLOCKS or more (small launches keep the direct form's finer tiles) -- 0
// never, 1 every launch the shape allows.  (Round 3 ran the form in evaluation calls only: inside a training step its
// whole-CU workgroups measured neutral to slower.  The cause was the GRID, not the form: 256 persistent workgroups
// beside the sampler's 64 resident workgroups leave 64 of them waiting for a CU for a whole round; sized to the CUs the
// launch can actually get -- ConvArgs.wino_cus, the WINO_CUS_* constants of conv.h -- the form pays inside the step.)
bool conv_wino_ok(int64_t M, int Cin, int Cout, int L, int kind) {
  const int64_t o = opt(OPT_ENC_WINO);
  if (o == 0 || !wino_shape_ok(M, Cin, Cout, L)) return false;
  // (the bf16-input mode means bf16-rounded operands for every forward / data-gradient launch: the float32 Winograd form
  // would silently compute those layers in float32, forced or not; weight gradients -- kind 3 -- are float32 in every mode)
  if (kind != 3 && conv_precision_mode() == 1) return false;
  if (o >= 1) return true;
  const int cfg = wino_cfg(Cout);
  const int64_t blocks = M / 4 / wino_tblk(cfg) * (Cout / (cfg == 0 ? 64 : 32));
  // evaluation-mode forward calls: from 256 workgroups' worth up (round 3).  Training calls: the float32 and f32x3 operand
  // modes (the form is strict float32 arithmetic on 4/9 of the multiplications: faster than the split direct kernel
  // too; in the bf16-input mode the direct kernel on the bf16 MFMA is the faster one) and from WINO_MIN_BLOCKS
  // (512) up -- all of c3's launches, the larger ones of mid-size batches; a c2 step with its one 256-block launch
  // switched measured 3.5 % slower (the draw's 128 one-wave workgroups leave the persistent grid 128 CUs there)
  // (kind 3: a weight gradient -- float32 in every operand mode; forward / data gradient: not in the bf16-input mode)
  if (kind == 3) return blocks >= WINO_MIN_BLOCKS;
  return kind == 2 ? blocks >= 256 : blocks >= WINO_MIN_BLOCKS;
}

int conv_wino_stats_blocks(int64_t M, int Cout) { return (int)(M / 4 / wino_tblk(wino_cfg(Cout))); }

template <int WM, int WN, int KC, int UM = 0, int TAIL = 1>
static int launch_wino8_cfg(const WinoArgs& a, double flops, hipStream_t s) {
  using C = Wino8Cfg<WM, WN, KC, UM>;
  auto kern = conv_wino8_kernel<WM, WN, KC, UM, TAIL>;
  static std::atomic<uint64_t> attr_done{0};
  DVG_TRY(raise_dynamic_lds(attr_done, (const void*)kern, C::LDS_BYTES));
  const int ny = a.Cout / C::CB;
  int cus = a.cus > 0 ? a.cus : 256;
  if (cus < ny) cus = ny;
  if (cus > 256) cus = 256;
  int gx = cus / ny;
  if (gx < 1) gx = 1;
  if (gx > a.nblk) gx = a.nblk;
  WinoArgs ad = a;
  // (the dynamic deal hands every workgroup TWO blocks up front: a launch of fewer than about three rounds keeps the
  // static deal, where the grid is shrunk to its round count -- an evaluation forward at the 256-block threshold would
  // otherwise be done by half the grid in two rounds.  Dynamic launches of one device must be ordered on ONE stream:
  // the counters are a per-device pool that the launch's last workgroup re-zeroes.)
  ad.dyn = (opt(OPT_WINO_DYNAMIC) != 0 && a.Cin / KC >= 2 && ny <= 16 && a.nblk >= 3 * gx) ? dyn_tile_counters() : nullptr;
  if (!ad.dyn) gx = (a.nblk + (a.nblk + gx - 1) / gx - 1) / ((a.nblk + gx - 1) / gx);
  DVG_LAUNCH_WORK_SHARE(K_IGEMM_WINO, flops, (float)(gx * ny > 256 ? 256 : gx * ny) / 256.0f, kern, dim3((unsigned)gx, (unsigned)ny), dim3(512), C::LDS_BYTES, s, ad);
  return DVG_OK;
}

// a.wp must be the transformed pack of launch_wino_weight_pack; a.M, Cin, Cout, L, bias, out, stats as for launch_conv_igemm
int launch_conv_wino(const ConvArgs& a, hipStream_t s) {
  // wino_um = 1: Upsample(x2) + 3x3 forward (a.in = the SOURCE map, a.M / a.L of the output grid); 2: its data gradient
  // (a.in = the fine-grid gradient, a.out = the source map's gradient, one row per quad)
  DVG_REQUIRE(wino_shape_ok(a.M, a.Cin, a.Cout, a.L) && a.ntaps == 9 && !a.ups && !a.poolsum && !a.fold && a.wino_um >= 0 && a.wino_um <= 2,
              "conv_wino: unsupported launch (M=%lld Cin=%d Cout=%d L=%d)", (long long)a.M, a.Cin, a.Cout, a.L);
  WinoArgs w;
  w.in = a.in; w.u = a.wp; w.bias = a.bias; w.out = a.out; w.stats = a.stats;
  w.Cin = a.Cin; w.Cout = a.Cout; w.L = a.L; w.cus = a.wino_cus;
  const int cfg = wino_cfg(a.Cout);
  w.nblk = (int)(a.M / 4 / wino_tblk(cfg));
  // EXECUTED matrix FLOPs: 16 transform-domain GEMMs over the M / 4 quads (4/9 of the direct form's 2 M Cin Cout 9;
  // the roofline prices what the matrix pipe does -- a rate in direct-form FLOPs would pass the f32 MFMA peak)
  const double flops = 2.0 * (double)(a.M / 4) * (a.wino_um ? 9.0 : 16.0) * a.Cin * a.Cout;
  // (the 32-channel tile behind the upsample has two groups per chunk in wave set 1: no tail to hold back)
  if (a.wino_um == 1) return cfg == 0 ? launch_wino8_cfg<2, 2, 8, 1>(w, flops, s) : launch_wino8_cfg<4, 1, 4, 1, 0>(w, flops, s);
  if (a.wino_um == 2) return cfg == 0 ? launch_wino8_cfg<2, 2, 8, 2>(w, flops, s) : launch_wino8_cfg<4, 1, 4, 2, 0>(w, flops, s);
  return cfg == 0 ? launch_wino8_cfg<2, 2, 8>(w, flops, s) : launch_wino8_cfg<4, 1, 4>(w, flops, s);
}

}  // namespace dvg
