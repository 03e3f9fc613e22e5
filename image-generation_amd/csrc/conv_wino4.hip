// Winograd F(4x4, 3x3) form of the stride-1 3x3 convolutions of the encoder (forward of layers 1-3 and their data
// gradients: /root/reference/src/encoder.py:28-36), float32 on v_mfma_f32_16x16x4_f32.
//
// Why: F(2x2, 3x3) (conv_wino.hip) issues 16 multiplies per 2x2 output quad and channel pair -- 4.0 per output; the
// 4x4 output tile needs 36 per 16 outputs -- 2.25 per output: 0.5625 of the matrix work for the same float32 convolution,
// paid for with transform constants 1/24 .. 8 instead of +-1, 1/2 (the measured error) and 36 accumulator
// tiles per (tile, channel) sub-tile where the 2x2 form has 16.
//
//   U[xi][nu][ci][co] = (G g G^T)            weights, once per step (wino4_pack_entry), G = 6x3
//   V[xi][nu][tile][ci] = (B^T d B)          6x6 input patch d of the 4x4 output tile (zero padded), B^T = 6x6
//   M[xi][nu][tile][co] = sum_ci V U         36 independent GEMMs: the MFMA work
//   Y[tile][4x4][co]    = A^T M A (+bias)    A^T = 4x6
//
// Morton-ordered activations (conv.h): a 4x4 output tile IS sixteen consecutive rows, 64 tiles are whole images.
//
// One workgroup = 8 waves on one CU, two per SIMD, a tile block of 64 tiles (1024 pixels) x 32 output channels.  The 36
// position GEMMs of a 16 tile x 16 channel sub-tile live in ONE wave (36 accumulator tiles of 4 registers on the 16x16x4
// MFMA = 144 registers), so the output transform is lane-local: no exchange between waves, no barrier in the epilogue
// but the one that hands the block on.  (The 32x32x2 MFMA would need 576 accumulator registers per sub-tile, i.e. the 36
// positions split over four waves and a reduce-scatter of 10 accumulator tiles per wave through LDS per tile block.)
// The price is operand traffic: a 16x16x4 MFMA takes one A and one B value per lane for 32 cycles of matrix work, twice
// the LDS read rate per FLOP of the 32x32x2 form -- one ds_read_b128 per operand and FOUR MFMAs (a 144-byte entry holds
// the 36 positions of one (k, row); entries 36 words apart are conflict-free for the 16 lanes of a read pass as they
// are, no swizzle).
//
// Channels are walked in chunks of 4 (one MFMA k-step); per chunk three LDS images: the raw input pixels of the block's
// images as zero-haloed row-major pictures [image][y][x][4] (LDS-DMA with a per-lane source: the Morton -> row-major
// permutation and the halo -- an out-of-range offset loads zeros -- cost nothing, and every tap of a patch is then ONE
// base register + an immediate), the transformed weights [k][co][36] (LDS-DMA of a contiguous 18 KiB piece of the
// pack), the transformed input [k][tile][36], written by waves 0-3 (a patch per thread and chunk, under the MFMAs).
// Two stages of each, one barrier per chunk, the chunk sequence runs on across the tile blocks a workgroup owns
// (persistent grid, dynamic deal: conv_wino.hip).
#include <atomic>

#include "conv.h"
#include "conv_tile.h"
#include "conv_wino4.h"

namespace dvg {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) const float lds_cf32;

struct Wino4Args {
  const float* in;    // [16 tiles][Cin]  (Morton pixel order: tile t = rows 16t .. 16t+15)
  const float* u;     // [Cin / 4][Cout / 32][4][32][36] transformed weights (wino4_pack_entry)
  const float* bias;  // [Cout] or null
  float* out;         // [16 tiles][Cout]
  float* stats;       // [nblk][Cout][2] per tile block (sum, sum of squares) of the output, or null
  int cus = 0;        // CUs the persistent grid is sized for (0 = 256)
  int* dyn = nullptr; // tile counters of the dynamic deal (conv_wino.hip), or null: round-robin
  int Cin, Cout;
  int nblk;           // tile blocks (of 64 tiles)
};

#ifndef WINO4_STORE_AUX
#define WINO4_STORE_AUX 0   // cache policy of the output stores (2 = nt)
#endif
#ifndef WINO4_RX
#define WINO4_RX 0   // raw pieces per wave of the transform role (-1: an eighth of them)
#endif

// UM = 1: the decoder's Upsample(x2) + 3x3 forward (/root/reference/src/decoder.py:34-46): `in` is the SOURCE map (half the
// output's side).  The 6x6 patch of the upsampled map repeats a 4x4 patch s of source pixels, d = P s P^T, so V = T s T^T
// with T = B^T P = [4 -5 1 0; 0 -8 2 0; 0 0 0 0; 0 -3 3 0; 0 1 -1 0; 0 4 -5 1]: transform row / column 2 vanishes
// identically and 25 of the 36 position GEMMs remain -- 1.56 multiplies per output where the F(2x2) form behind the
// upsample (conv_wino.hip, UM = 1: 9 of 16 positions) has 2.25.  Entries hold the 25 positions (xi', nu' over {0,1,3,4,5})
// in 28 floats.
// UM = 2: its adjoint, the data gradient onto the source map: `in` is the fine-grid gradient (the plain form's patches and
// pictures), the input transform keeps rows / columns {0,1,3,4,5}, and the output transform sums each 2x2 quad of the 4x4
// tile, (Q A^T) M (Q A^T)^T with Q A^T = [1 2 0 3 -1 0; 0 2 0 12 -4 1] -- column 2 is zero there too: the same 25 positions;
// `out` is the source map's gradient, four rows per tile.
template <int L, int UM = 0>
struct Wino4Cfg {
  static constexpr int H = 1 << L, HW = H * H, TPI = HW / 16, IPB = 64 / TPI;  // tiles per image, images per tile block
  static constexpr int NPOS = UM ? 25 : 36, EF = UM ? 28 : 36, EB = 4 * EF, NQ = EF / 4;  // positions, floats / bytes / quads per entry
  // The raw pictures, in 16-byte cells (a pixel's 4 channels).  4x4 images: one tile per image, the halo is known at
  // compile time -- 16 cells in Morton order + 1 (an ODD image stride: the 8 images of a 32-lane read pass start in 8
  // different bank groups).  8x8: a zero-haloed 10 x 10 picture per image + 1.  16x16: rows of 23 cells -- left halo, a
  // dead cell, then FIVE cells per 4 pixels (the fifth dead), right halo: a tile's x-step is 5 cells, so the 8 tiles a
  // read pass spans (x bit 0, x bit 1, y bit 0) start in 8 different bank groups -- and 17 rows per image (the bottom
  // halo row is the next image's top one).
  // UM = 1 (source pictures): 4x4 outputs <- 2x2 sources: 4 cells + 1 (the halo is known); 8x8 <- 4x4: a zero-haloed 6 x 6
  // picture + 1.
  static constexpr bool SRC = UM == 1;  // the input is the half-resolution source map (UM = 2: the fine-grid gradient, as the plain form)
  static constexpr int ROWC = SRC ? 6 : (L == 3 ? 10 : 23), ROWS_IMG = L == 3 ? 10 : 17;
  static constexpr int CELLS_IMG = SRC ? (L == 2 ? 5 : 37) : (L == 2 ? 17 : (L == 3 ? 101 : ROWS_IMG * ROWC));
  static constexpr int NCELL = (L == 4 && !SRC) ? (IPB * ROWS_IMG + 1) * ROWC : IPB * CELLS_IMG;
  static constexpr int SRC_PIX = SRC ? 256 : 1024;  // input pixels of a tile block
  static constexpr int OUT_PIX = UM == 2 ? 256 : 1024;  // output pixels of a tile block (UM = 2: the source map's gradient)
  static constexpr int RAW_B = NCELL * 16;
  static constexpr int NRAWP = (NCELL + 63) / 64;   // 1 KiB DMA pieces (the last one is moved back to end at NCELL)
  // transformed input: [k][entry][36 floats], 32 bytes between the k planes (see the bank notes in the kernel)
  static constexpr int VPL = 64 * EB + 32, V_B = 4 * VPL;
  static constexpr int U_B = 4 * 32 * EB, U_PIECES = U_B / 1024;
  static constexpr int OFF_RAW = 0, OFF_V = 2 * RAW_B, OFF_U = OFF_V + 2 * V_B, OFF_RED = OFF_U + 2 * U_B, OFF_NEXT = OFF_RED + 1024;
  static constexpr int LDS_BYTES = OFF_NEXT + 16;
  static_assert(L >= 2 && L <= 4 && (!UM || L <= 3) && NCELL >= 64 && U_B % 1024 == 0 && LDS_BYTES <= 160 * 1024, "unsupported shape");
  // cell offset of patch element (i, j) from the thread's base cell
  static constexpr int poff(int i, int j) {
    if (SRC) return L == 2 ? (int)(((unsigned)(j - 1) & 1u) | (((unsigned)(i - 1) & 1u) << 1)) : i * ROWC + j;  // (4x4 source patch)
    if (L == 2) return (int)(((unsigned)(j - 1) & 1u) | (((unsigned)(i - 1) & 1u) << 1) | (((unsigned)(j - 1) & 2u) << 1) | (((unsigned)(i - 1) & 2u) << 2));
    if (L == 3) return i * ROWC + j;
    return i * ROWC + (j == 0 ? 0 : (j == 5 ? 7 : j + 1));  // (from the row's cell 5 tx: the left neighbour of the tile)
  }
};

// Entry of MFMA row r (0..15) inside its group of sixteen: rows {0-3, 12-15} on the even entries, rows {4-11} on the odd
// ones.  A ds_read_b128 is served in lane groups that pair the rows {0-3, 12-15} of one k plane with the rows {4-11} of
// the next; with the planes 32 bytes apart (what makes the transform's ds_write_b128 conflict-free: its 8-lane groups
// hold 4 planes x 2 entries) the two halves of a group then fall on the even and the odd 16-byte bank groups.
__device__ __forceinline__ constexpr int wino4_entry(int r) { return r < 4 ? 2 * r : (r < 12 ? 2 * (r - 4) + 1 : 2 * (r - 8)); }

template <int L, int UM>
__device__ __forceinline__ void conv_wino4_body(const Wino4Args& a, unsigned char* wsm) {
  using C = Wino4Cfg<L, UM>;
  constexpr int EB = C::EB, NQ = C::NQ, NPOS = C::NPOS;
  constexpr int H = C::H, HW = C::HW;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte_t*)wsm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wt = wave & 3, wc = wave >> 2;  // the wave's 16 tiles and 16 channels of the block's 64 x 32
  const int kq = lane >> 4, r16 = lane & 15;
  const int n0 = blockIdx.y * 32;
  const int nch = a.Cin / 4;

  // ---- the patch a thread of waves 0-3 transforms per chunk: channel t_k of tile tl.  Lane bits: k (2), then three tile
  // bits p, a, b, then c; wave bits w.  (1) The 32 lanes of a raw read pass (k, p, a, b) hold 4 channels x 8 cells that
  // start in 8 different bank groups; (2) the 8 lanes of a ds_write_b128 group (k, p) hold 4 planes x 2 entries of
  // different parity (p flips bit 2 of the tile's MFMA row: wino4_entry), 8 different bank groups.
  const int t_k = lane & 3, bp = (lane >> 2) & 1, ba = (lane >> 3) & 1, bb = (lane >> 4) & 1, bc = lane >> 5;
  int t_img, t_ty = 0, t_tx = 0, cell0;
  if constexpr (UM == 1 && L == 3) {  // (source cells: tile x-step 2, y-step 12, image stride 37: p, a = image bits, b = y)
    t_img = bp | (ba << 1) | ((wt & 1) << 2) | ((wt >> 1) << 3); t_ty = bb; t_tx = bc;
    cell0 = t_img * C::CELLS_IMG + 2 * t_ty * C::ROWC + 2 * t_tx;
  } else if constexpr (L == 2) {
    t_img = ba | (bb << 1) | (bp << 2) | (bc << 3) | (wt << 4);
    cell0 = t_img * C::CELLS_IMG;
  } else if constexpr (L == 3) {
    t_img = bp | (ba << 1) | (bc << 2) | ((wt & 1) << 3); t_tx = bb; t_ty = wt >> 1;
    cell0 = t_img * C::CELLS_IMG + 4 * t_ty * C::ROWC + 4 * t_tx;
  } else {
    t_img = wt; t_tx = ba | (bp << 1); t_ty = bb | (bc << 1);
    cell0 = (t_img * C::ROWS_IMG + 4 * t_ty) * C::ROWC + 5 * t_tx;
  }
  const int tl = t_img * C::TPI + (int)morton((uint32_t)t_ty, (uint32_t)t_tx);
  const int t_entry = (tl & ~15) + wino4_entry(tl & 15);
  // (one base register per stage: the LDS image spans 160 KB and a DS instruction's immediate offset 64 KB -- with ONE
  // base the compiler adds the stage's offset in front of every access)
  uint32_t rbase[2], vst[2], aaddr[2], baddr[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    rbase[st] = lds0 + C::OFF_RAW + st * C::RAW_B + (uint32_t)((cell0 * 4 + t_k) * 4);
    vst[st] = lds0 + C::OFF_V + st * C::V_B + (uint32_t)(t_k * C::VPL + t_entry * EB);
    // MFMA operands: entry (k, row) of 144 bytes; lane (kq, r16) reads k = kq of its row / column
    aaddr[st] = lds0 + C::OFF_V + st * C::V_B + (uint32_t)(kq * C::VPL + (16 * wt + wino4_entry(r16)) * EB);
    baddr[st] = lds0 + C::OFF_U + st * C::U_B + (uint32_t)((kq * 32 + 16 * wc + r16) * EB);
    asm volatile("" : "+v"(rbase[st]), "+v"(vst[st]), "+v"(aaddr[st]), "+v"(baddr[st]));  // (kept apart: not base + constant again)
  }

  // ---- DMA of one chunk's images: 1 KiB pieces; raw piece p by wave p % 8, weight piece q by wave 4 + q % 4
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in), 0, (int)((int64_t)a.nblk * C::SRC_PIX * a.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.u), 0, (int)((int64_t)a.Cin * a.Cout * EB), 0x00020000);
  // Every wave of a role issues the SAME number of pieces, unconditionally (a piece index past the end repeats the last
  // piece: same bytes, same place): with a conditional piece the compiler's wait-count pass loses the order of the
  // vector-memory operations at the join and puts vmcnt(0) in front of every barrier -- which drains the output stores.
  // Waves 0-3 (the transform role): RX raw pieces each; waves 4-7: RN raw pieces and the UPW weight pieces.
  constexpr int RX = WINO4_RX < 0 ? C::NRAWP / 8 : WINO4_RX, RN = (C::NRAWP - 4 * RX + 3) / 4, RPW = RN, UPW = (C::U_PIECES + 3) / 4;
  static_assert(RN >= RX && 4 * (RX + RN) >= C::NRAWP, "raw pieces");
  int rvoff[RPW];
  uint32_t rdst[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    int p = wave < 4 ? wave + 4 * q : 4 * RX + (wave - 4) + 4 * q;
    if (p > C::NRAWP - 1) p = C::NRAWP - 1;
    // (the last piece ends at the image's last cell: it rewrites a few cells of the piece before it with the same bytes
    // instead of running past the stage)
    const int c0 = p == C::NRAWP - 1 ? C::NCELL - 64 : p * 64;
    const int cell = c0 + lane, img = cell / C::CELLS_IMG, rem = cell - img * C::CELLS_IMG;
    bool inside;
    int px;
    if constexpr (UM == 1 && L == 2) {
      inside = rem < 4;
      px = img * 4 + rem;
    } else if constexpr (UM == 1) {  // (L = 3: 4x4 source images in zero-haloed 6 x 6 pictures)
      const int yy = rem / C::ROWC, xx = rem - yy * C::ROWC;
      inside = rem < C::ROWC * C::ROWC && yy >= 1 && yy <= H / 2 && xx >= 1 && xx <= H / 2;
      px = img * (HW / 4) + (int)morton((uint32_t)(inside ? yy - 1 : 0), (uint32_t)(inside ? xx - 1 : 0));
    } else if constexpr (L == 2) {
      inside = rem < 16;
      px = img * 16 + rem;
    } else if constexpr (L == 3) {
      const int yy = rem / C::ROWC, xx = rem - yy * C::ROWC;
      inside = rem < C::ROWC * C::ROWC && yy >= 1 && yy <= H && xx >= 1 && xx <= H;
      px = img * HW + (int)morton((uint32_t)(inside ? yy - 1 : 0), (uint32_t)(inside ? xx - 1 : 0));
    } else {
      const int row = cell / C::ROWC, xx = cell - row * C::ROWC, im = row / C::ROWS_IMG, yy = row - im * C::ROWS_IMG;
      const int g = xx - 2, grp = g / 5, w5 = g - grp * 5;  // cells 0 / 22: the halo; 1 and every fifth: dead
      inside = yy >= 1 && im < C::IPB && g >= 0 && xx < C::ROWC - 1 && w5 < 4;
      px = im * HW + (int)morton((uint32_t)(inside ? yy - 1 : 0), (uint32_t)(inside ? 4 * grp + w5 : 0));
    }
    rvoff[q] = inside ? px * a.Cin * 4 : (int)0x80000000u;  // (past num_records: the hardware writes zeros -- the halo)
    rdst[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + C::OFF_RAW + (uint32_t)c0 * 16));
  }
  // (one piece per call, so that a chunk can place them between its MFMA groups; q-th piece of this wave)
  auto raw_soff = [&](int blk, int ch) { return __builtin_amdgcn_readfirstlane((blk * C::SRC_PIX * a.Cin + ch * 4) * 4); };
  auto issue_raw_piece = [&](int soff, int st, int q) {
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(rdst[q] + (uint32_t)(st * C::RAW_B)));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_void_t*)(uintptr_t)dst, 16, rvoff[q], soff, 0, 0);
  };
  const int ustride = a.Cout / 32 * C::U_B;  // bytes between two chunks of the pack
  auto u_sbase = [&](int ch) { return __builtin_amdgcn_readfirstlane(ch * ustride + (int)blockIdx.y * C::U_B); };
  auto issue_u_piece = [&](int sbase, int st, int q) {  // (waves 4-7; pieces 18, 19 repeat pieces 0, 1)
    int p = ((wave - 4) & 3) + 4 * q;
    if (p >= C::U_PIECES) p -= C::U_PIECES;
    p = __builtin_amdgcn_readfirstlane(p);
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + C::OFF_U + st * C::U_B + p * 1024));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_u, (lds_void_t*)(uintptr_t)dst, 16, lane * 16, sbase + p * 1024, 0, 0);
  };
  // all of a chunk's pieces at once (the prologue): XFW = the wave's role
  auto issue_all = [&](auto xfc, int blk, int ch, int st, bool with_u) {
    constexpr bool XFW = decltype(xfc)::value != 0;
    const int soff = raw_soff(blk, ch);
#pragma unroll
    for (int q = 0; q < (XFW ? RX : RN); ++q) issue_raw_piece(soff, st, q);
    if constexpr (!XFW) {
      if (with_u) {
        const int sbase = u_sbase(ch);
#pragma unroll
        for (int q = 0; q < UPW; ++q) issue_u_piece(sbase, st, q);
      }
    }
  };

  // ---- input transform of the thread's patch: raw stage `rs` -> transformed stage `vs`, in place in d[36]
  constexpr int I0 = L == 2 ? 1 : 0, I1 = L == 2 ? 5 : 6;  // (4x4 images: ring 0 / 5 of the patch is the zero padding)
  auto load_patch = [&](int rs, float (&d)[36]) {
#pragma unroll
    for (int i = I0; i < I1; ++i)
#pragma unroll
      for (int j = I0; j < I1; ++j) {
        d[i * 6 + j] = *reinterpret_cast<lds_cf32*>((uintptr_t)(rbase[rs] + (uint32_t)(C::poff(i, j) * 16)));
      }
    if constexpr (L == 2) {
#pragma unroll
      for (int e = 0; e < 36; ++e)
        if (e / 6 == 0 || e / 6 == 5 || e % 6 == 0 || e % 6 == 5) d[e] = 0.f;
    }
  };
  auto xform_col = [&](int j, float (&d)[36]) {  // B^T d, column j
    if (L == 2 && (j == 0 || j == 5)) return;    // (a zero column stays zero)
    wino4_in6<L == 2>(d[j], d[6 + j], d[12 + j], d[18 + j], d[24 + j], d[30 + j]);
  };
  auto xform_row = [&](int i, float (&d)[36]) {  // (.) B, row i
    wino4_in6<L == 2>(d[6 * i], d[6 * i + 1], d[6 * i + 2], d[6 * i + 3], d[6 * i + 4], d[6 * i + 5]);
  };
  auto store_quads = [&](int vs, int q0, const float (&d)[36]) {  // entries 4 q0 .. 4 q0 + 11 (two transformed rows)
#pragma unroll
    for (int q = q0; q < q0 + 3; ++q) {
      const f32x4 o = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
      *reinterpret_cast<lds_f32x4*>((uintptr_t)(vst[vs] + (uint32_t)(q * 16))) = o;
    }
  };

  // UM = 1: the 4x4 source patch (4x4 outputs <- 2x2 sources: its ring is the zero padding) -> V = T s T^T, 25 values in
  // 28 floats (positions xi' 5 + nu' over {0,1,3,4,5}; three zero floats complete the last quad)
  auto load_patch_u = [&](int rs, float (&sv)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ring = i == 0 || i == 3 || j == 0 || j == 3;
        if (L == 2 && ring) sv[i * 4 + j] = 0.f;
        else sv[i * 4 + j] = *reinterpret_cast<lds_cf32*>((uintptr_t)(rbase[rs] + (uint32_t)(C::poff(i, j) * 16)));
      }
  };
  auto xform_u_cols = [&](const float (&sv)[16], float (&t)[30]) {  // T s: t[xi' 4 + j]
#pragma unroll
    for (int j = 0; j < 4; ++j) wino4_ups5<L == 2>(sv[j], sv[4 + j], sv[8 + j], sv[12 + j], t[j], t[4 + j], t[8 + j], t[12 + j], t[16 + j]);
  };
  auto xform_u_row = [&](int i, const float (&t)[30], float (&v)[28]) {  // (.) T^T, row xi' = i
    wino4_ups5<L == 2>(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3], v[5 * i], v[5 * i + 1], v[5 * i + 2], v[5 * i + 3], v[5 * i + 4]);
  };
  auto store_quads_u = [&](int vs, int q0, int q1, const float (&v)[28]) {
#pragma unroll
    for (int q = q0; q < q1; ++q) {
      const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      *reinterpret_cast<lds_f32x4*>((uintptr_t)(vst[vs] + (uint32_t)(q * 16))) = o;
    }
  };

  // UM = 2: the plain 6x6 patch, rows / columns {0,1,3,4,5} of B^T d B
  auto xform_g_cols = [&](const float (&d)[36], float (&t)[30], int j0, int j1) {  // t[xi' 6 + j]
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      if (L == 2 && (j == 0 || j == 5)) { t[j] = t[6 + j] = t[12 + j] = t[18 + j] = t[24 + j] = 0.f; continue; }
      wino4_in5<L == 2>(d[j], d[6 + j], d[12 + j], d[18 + j], d[24 + j], d[30 + j], t[j], t[6 + j], t[12 + j], t[18 + j], t[24 + j]);
    }
  };
  auto xform_g_row = [&](int i, const float (&t)[30], float (&v)[28]) {
    wino4_in5<L == 2>(t[6 * i], t[6 * i + 1], t[6 * i + 2], t[6 * i + 3], t[6 * i + 4], t[6 * i + 5],
                      v[5 * i], v[5 * i + 1], v[5 * i + 2], v[5 * i + 3], v[5 * i + 4]);
  };

  f32x4 acc[4 * NQ];  // acc[position][i] = M[xi][nu] of tile 16 wt + 4 kq + i, channel 16 wc + r16 (position 6 xi + nu, or 5 xi' + nu')

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
  if (wave < 4) { issue_all(std::integral_constant<int, 1>{}, blk_cur, 0, 0, true); issue_all(std::integral_constant<int, 1>{}, blk_cur, 1, 1, false); }
  else { issue_all(std::integral_constant<int, 0>{}, blk_cur, 0, 0, true); issue_all(std::integral_constant<int, 0>{}, blk_cur, 1, 1, false); }
  __syncthreads();  // (the workgroup fence waits for the LDS-DMA pieces)
  if (wave < 4) {
    if constexpr (UM == 1) {
      float sv[16], t[30], v[28] = {};
      load_patch_u(0, sv);
      xform_u_cols(sv, t);
#pragma unroll
      for (int i = 0; i < 5; ++i) xform_u_row(i, t, v);
      store_quads_u(0, 0, NQ, v);
    } else if constexpr (UM == 2) {
      float d[36], t[30], v[28] = {};
      load_patch(0, d);
      xform_g_cols(d, t, 0, 6);
#pragma unroll
      for (int i = 0; i < 5; ++i) xform_g_row(i, t, v);
      store_quads_u(0, 0, NQ, v);
    } else {
      float d[36];
      load_patch(0, d);
#pragma unroll
      for (int j = 0; j < 6; ++j) xform_col(j, d);
#pragma unroll
      for (int i = 0; i < 6; ++i) xform_row(i, d);
      store_quads(0, 0, d); store_quads(0, 3, d); store_quads(0, 6, d);
    }
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(
      a.out, 0, (int)((int64_t)a.nblk * C::OUT_PIX * a.Cout * 4), 0x00020000);
  const int ch_out = n0 + 16 * wc + r16;
  constexpr int PPT = UM == 2 ? 4 : 16;  // output pixels per tile
  const int ovoff = (4 * kq * PPT * a.Cout + ch_out) * 4;  // tile 4 kq of the wave's sixteen, pixel 0, the lane's channel
  const float bias = a.bias ? a.bias[ch_out] : 0.f;

  // outputs of the tile block just finished (64 per lane: 4 tiles x 16 pixels of one channel), stored beside the MFMAs of
  // the NEXT block's first chunk (or behind the last block), BEHIND that chunk's LDS-DMA pieces: stores count on vmcnt like
  // the pieces, in issue order, so they stay in flight across that chunk's barrier (the next one waits them out: all
  // workgroups reach their epilogues together and 128 KB per workgroup stored at once is a burst the memory system
  // takes microseconds to drain); a store's register is free again before the accumulator tiles come alive.
  constexpr int NSTORE = 4 * PPT;
  float yp[NSTORE];  // yp[(b 4 + a) 4 + i]: pixel (a, b) of tile 4 kq + i  (UM = 2: (b 2 + a) 4 + i, source pixel (a, b))
  int pblk = 0;
  auto store_pending = [&](int idx) {
    const int i = idx & 3;
    const int b = UM == 2 ? idx >> 3 : idx >> 4, aa = UM == 2 ? (idx >> 2) & 1 : (idx >> 2) & 3;
    // (the row index is kept opaque: its products with the 64 constant row offsets would each take a scalar register)
    int row0 = __builtin_amdgcn_readfirstlane((pblk * 64 + 16 * wt) * PPT);
    asm volatile("" : "+s"(row0));
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yp[idx]), rsrc_out, ovoff,
                                          (row0 + i * PPT + (int)morton((uint32_t)aa, (uint32_t)b)) * a.Cout * 4, WINO4_STORE_AUX);
  };
  // the epilogue: Y = A^T M A per (tile, channel), lane-local; + bias, BatchNorm partials
  auto epilogue = [&](int blk) {
    f32x4 c[24];  // c[4 xi + b] = sum_nu M[xi][nu] A[nu][b]  (UM: rows xi' over {0,1,3,4,5}; row / column 2 of M is zero)
    if constexpr (UM == 2) {
    } else if constexpr (UM) {
#pragma unroll
      for (int xi = 0; xi < 5; ++xi)
        wino4_out5(acc[5 * xi], acc[5 * xi + 1], acc[5 * xi + 2], acc[5 * xi + 3], acc[5 * xi + 4],
                   c[4 * xi], c[4 * xi + 1], c[4 * xi + 2], c[4 * xi + 3]);
    } else {
#pragma unroll
      for (int xi = 0; xi < 6; ++xi)
        wino4_out6(acc[6 * xi], acc[6 * xi + 1], acc[6 * xi + 2], acc[6 * xi + 3], acc[6 * xi + 4], acc[6 * xi + 5],
                   c[4 * xi], c[4 * xi + 1], c[4 * xi + 2], c[4 * xi + 3]);
    }
    if constexpr (UM == 2) {
      // the source map's gradient: (Q A^T) M (Q A^T)^T, two source pixels per line
      f32x4 cq[10];  // cq[2 xi' + b]
#pragma unroll
      for (int xi = 0; xi < 5; ++xi)
        wino4_out5q(acc[5 * xi], acc[5 * xi + 1], acc[5 * xi + 2], acc[5 * xi + 3], acc[5 * xi + 4], cq[2 * xi], cq[2 * xi + 1]);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f32x4 y[2];
        wino4_out5q(cq[b], cq[2 + b], cq[4 + b], cq[6 + b], cq[8 + b], y[0], y[1]);
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
          for (int i = 0; i < 4; ++i) yp[(b * 2 + aa) * 4 + i] = y[aa][i] + bias;
      }
      pblk = blk;
      return;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f32x4 y[4];
      if constexpr (UM) wino4_out5(c[b], c[4 + b], c[8 + b], c[12 + b], c[16 + b], y[0], y[1], y[2], y[3]);
      else wino4_out6(c[b], c[4 + b], c[8 + b], c[12 + b], c[16 + b], c[20 + b], y[0], y[1], y[2], y[3]);
#pragma unroll
      for (int aa = 0; aa < 4; ++aa)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = y[aa][i] + bias;
          s1 += v; s2 = __builtin_fmaf(v, v, s2);
          yp[(b * 4 + aa) * 4 + i] = v;
        }
    }
    pblk = blk;
    if (a.stats) {  // per-wave column sums; summed over the block's four tile groups behind the caller's barrier
      float* red = reinterpret_cast<float*>(wsm + C::OFF_RED);
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (kq == 0) { red[(wt * 32 + 16 * wc + r16) * 2] = s1; red[(wt * 32 + 16 * wc + r16) * 2 + 1] = s2; }
    }
  };
  auto stats_finish = [&](int blk) {
    if (a.stats && tid < 32) {
      const float* red = reinterpret_cast<const float*>(wsm + C::OFF_RED);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += red[(w * 32 + tid) * 2]; t2 += red[(w * 32 + tid) * 2 + 1]; }
      float* dst = a.stats + ((size_t)blk * a.Cout + n0 + tid) * 2;
      dst[0] = t1; dst[1] = t2;
    }
  };

  // One chunk: 9 operand quads (a ds_read_b128 per operand, positions 4 q .. 4 q + 3) x 4 MFMAs as one pinned stream,
  // operands two quads ahead; waves 0-3 (XF) read their patch of the NEXT chunk at the start and transform it in pieces
  // beside the MFMAs: the six columns, then the six rows, the entries stored as they complete.
  auto chunk = [&](int cix, auto stc, auto xfc, auto firstc, auto pendc) {
    constexpr int st = decltype(stc)::value;
    constexpr bool XF = decltype(xfc)::value != 0, FIRST = decltype(firstc)::value != 0, PEND = decltype(pendc)::value != 0;
    const int u_ch = cix + 1 < nch ? cix + 1 : 0;
    const bool over = cix + 2 >= nch;
    const int r_ch = !over ? cix + 2 : (has_next ? cix + 2 - nch : nch - 1);
    const int r_blk = over && has_next ? blk_nxt : blk_cur;
    // the chunk's LDS-DMA pieces first (they have the whole chunk to land): the next chunk's weights (waves 4-7), then the
    // raw pixels of the chunk after next
    {
      const int usb = u_sbase(u_ch), rso = raw_soff(r_blk, r_ch);
      if constexpr (!XF) {
#pragma unroll
        for (int q = 0; q < UPW; ++q) issue_u_piece(usb, st ^ 1, q);
      }
#pragma unroll
      for (int q = 0; q < (XF ? RX : RN); ++q) issue_raw_piece(rso, st, q);
    }
    f32x4 qa[3], qb[3];
    auto load_quad = [&](int q) {
      qa[q % 3] = *reinterpret_cast<lds_cf32x4*>((uintptr_t)(aaddr[st] + (uint32_t)(q * 16)));
      qb[q % 3] = *reinterpret_cast<lds_cf32x4*>((uintptr_t)(baddr[st] + (uint32_t)(q * 16)));
    };
    float d[36], su[16], tu[30], vu[28];  // (the plain form's patch; behind the upsample: source patch, T s / B^T d, V)
    load_quad(0);
    load_quad(1);
    if constexpr (XF) {
      if constexpr (UM == 1) load_patch_u(st ^ 1, su);
      else load_patch(st ^ 1, d);
      if constexpr (UM) {
#pragma unroll
        for (int e = 25; e < 28; ++e) vu[e] = 0.f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q + 2 < NQ) load_quad(q + 2);
      if constexpr (XF && UM == 2) {
        // pieces: q = 1, 2 three columns each, q = 3 rows 0-2 and entry quads 0-2, q = 4 rows 3-4 and quads 3-6
        if (q == 1) xform_g_cols(d, tu, 0, 3);
        if (q == 2) xform_g_cols(d, tu, 3, 6);
        if (q == 3) { xform_g_row(0, tu, vu); xform_g_row(1, tu, vu); xform_g_row(2, tu, vu); store_quads_u(st ^ 1, 0, 3, vu); }
        if (q == 4) { xform_g_row(3, tu, vu); xform_g_row(4, tu, vu); store_quads_u(st ^ 1, 3, NQ, vu); }
      } else if constexpr (XF && UM == 1) {
        // pieces: q = 1 the four columns, q = 2 rows 0-2 and entry quads 0-2, q = 3 rows 3-4 and quads 3-6
        if (q == 1) xform_u_cols(su, tu);
        if (q == 2) { xform_u_row(0, tu, vu); xform_u_row(1, tu, vu); xform_u_row(2, tu, vu); store_quads_u(st ^ 1, 0, 3, vu); }
        if (q == 3) { xform_u_row(3, tu, vu); xform_u_row(4, tu, vu); store_quads_u(st ^ 1, 3, NQ, vu); }
      } else if constexpr (XF) {
        // pieces: q = 1..3 two columns each, q = 4..6 two rows each and their three entry quads
        if (q >= 1 && q <= 3) { xform_col(2 * q - 2, d); xform_col(2 * q - 1, d); }
        if (q >= 4 && q <= 6) { xform_row(2 * q - 8, d); xform_row(2 * q - 7, d); store_quads(st ^ 1, 3 * (q - 4), d); }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (4 * q + m >= NPOS) continue;
        if constexpr (FIRST) acc[4 * q + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[q % 3][m], qb[q % 3][m], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else acc[4 * q + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[q % 3][m], qb[q % 3][m], acc[4 * q + m], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PEND) {
        constexpr int SPQ = (NSTORE + NQ - 2) / (NQ - 1);  // stores per quad: all of them beside the first NQ - 1 quads
        if (q < NQ - 1) {
#pragma unroll
          for (int idx = SPQ * q; idx < SPQ * q + SPQ && idx < NSTORE; ++idx) store_pending(idx);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // The chunk barrier.  hipcc counts the LDS-DMA pieces on vmcnt and the workgroup fence waits for exactly them: the
    // output stores issued behind this chunk's pieces stay in flight across it (conv_wino.hip).
    __syncthreads();
  };

  auto run = [&](auto xfc) {
    int par = 1;
    bool go = true, pend = false;
    using I0c = std::integral_constant<int, 0>;
    using I1c = std::integral_constant<int, 1>;
    while (go) {
      int fetched = 0;
      if (dynq && tid == 0 && has_next) fetched = atomicAdd(a.dyn + blockIdx.y, 1);
      if (pend) chunk(0, I0c{}, xfc, I1c{}, I1c{});
      else chunk(0, I0c{}, xfc, I1c{}, I0c{});
      for (int ch = 2; ch < nch; ch += 2) {
        chunk(ch - 1, I1c{}, xfc, I0c{}, I0c{});
        chunk(ch, I0c{}, xfc, I0c{}, I0c{});
      }
      chunk(nch - 1, I1c{}, xfc, I0c{}, I0c{});
      epilogue(blk_cur);
      pend = true;
      // ONE barrier per tile block behind the epilogue: the per-wave BatchNorm partials are in LDS, the next block's id is
      // handed round (dynamic deal: two slots in turn)
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
#pragma unroll
    for (int idx = 0; idx < NSTORE; ++idx) store_pending(idx);  // (the last block's)
  };
  // (one instantiation per role: waves 0-3 carry the input transform)
  if (wave < 4) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
  finish();
}

template <int L, int UM = 0>
__global__ __launch_bounds__(512) void conv_wino4_kernel(Wino4Args a) {
  extern __shared__ __align__(16) unsigned char wino4_smem[];
  conv_wino4_body<L, UM>(a, wino4_smem);
}

bool wino4_shape_ok(int64_t M, int Cin, int Cout, int L) {
  if (L < 2 || L > 4 || Cout % 32 || Cin % 8 || M % 1024) return false;  // whole tile blocks (of whole images), an even chunk count
  if (Cout / 32 > 16) return false;                                        // (the dynamic deal's counters: one per column block)
  if (M * (int64_t)(Cin > Cout ? Cin : Cout) * 4 >= 2147483647LL || (int64_t)Cin * Cout * 144 >= 2147483647LL) return false;
  return true;
}

template <int L, int UM = 0>
int launch_wino4_cfg(const Wino4Args& a, double flops, hipStream_t s) {
  using C = Wino4Cfg<L, UM>;
  auto kern = conv_wino4_kernel<L, UM>;
  static std::atomic<uint64_t> attr_done{0};
  DVG_TRY(raise_dynamic_lds(attr_done, (const void*)kern, C::LDS_BYTES));
  const int ny = a.Cout / 32;
  int cus = a.cus > 0 ? a.cus : 256;
  if (cus < ny) cus = ny;
  if (cus > 256) cus = 256;
  int gx = cus / ny;
  if (gx < 1) gx = 1;
  if (gx > a.nblk) gx = a.nblk;
  Wino4Args ad = a;
  ad.dyn = (opt(OPT_WINO_DYNAMIC) != 0 && a.nblk >= 3 * gx) ? dyn_tile_counters() : nullptr;
  if (!ad.dyn) gx = (a.nblk + (a.nblk + gx - 1) / gx - 1) / ((a.nblk + gx - 1) / gx);
  DVG_LAUNCH_WORK_SHARE(K_IGEMM_WINO4, flops, (float)(gx * ny > 256 ? 256 : gx * ny) / 256.0f, kern, dim3((unsigned)gx, (unsigned)ny), dim3(512), C::LDS_BYTES, s, ad);
  return DVG_OK;
}

bool conv_wino4_shape(int64_t M, int Cin, int Cout, int L) { return wino4_shape_ok(M, Cin, Cout, L); }

// option enc_wino4: 1 (default) the encoder launches of dev option enc_wino4_mask that run in the Winograd domain and whose
// shape qualifies, 0 never (F(2x2,3x3) everywhere).  Measured at c3, alternating runs in one process: the step 7.46-7.54 ms
// with F(2x2) everywhere, 7.24-7.34 with layer 3's forward, every data gradient and layer 3's weight gradient in this form
// (the default mask), the same with the forwards of layers 1-2 added (alone on the chip those are 0.85 / 1.0 of the F(2x2)
// kernel's time -- both forms wait for their output stores there -- and they stay on F(2x2): encoder.cpp, options.cpp).
bool conv_wino4_ok(int64_t M, int Cin, int Cout, int L) {
  return opt(OPT_ENC_WINO4) != 0 && wino4_shape_ok(M, Cin, Cout, L);
}

int conv_wino4_stats_blocks(int64_t M) { return (int)(M / 1024); }

// a.wp must be the F(4x4, 3x3) pack of a PackJob with wino = 2 ([Cin / 4][Cout / 32][4][32][36] floats: wino4_pack_entry)
int launch_conv_wino4(const ConvArgs& a, hipStream_t s) {
  DVG_REQUIRE(wino4_shape_ok(a.M, a.Cin, a.Cout, a.L) && a.ntaps == 9 && !a.ups && !a.poolsum && !a.fold && a.wino_um >= 0 &&
                  a.wino_um <= 2 && (a.wino_um == 0 || a.L <= 3),
              "conv_wino4: unsupported launch (M=%lld Cin=%d Cout=%d L=%d um=%d)", (long long)a.M, a.Cin, a.Cout, a.L, a.wino_um);
  Wino4Args w;
  w.in = a.in; w.u = a.wp; w.bias = a.bias; w.out = a.out; w.stats = a.stats;
  w.Cin = a.Cin; w.Cout = a.Cout; w.cus = a.wino_cus;
  w.nblk = (int)(a.M / 1024);
  // EXECUTED matrix FLOPs: 36 transform-domain GEMMs over the M / 16 tiles (1/4 of the direct form's 2 M Cin Cout 9)
  // (wino_um = 1: Upsample(x2) + 3x3 forward, a.in = the SOURCE map, a.M / a.L of the output grid, a.wp = the pack of a
  // PackJob with wino = 3: 25 of the 36 position GEMMs)
  // (wino_um = 2: its data gradient: a.in = the fine-grid gradient, a.out = the source map's gradient, four rows per tile)
  if (a.wino_um) {
    const double flops_u = 2.0 * (double)(a.M / 16) * 25.0 * a.Cin * a.Cout;
    if (a.wino_um == 1) return a.L == 2 ? launch_wino4_cfg<2, 1>(w, flops_u, s) : launch_wino4_cfg<3, 1>(w, flops_u, s);
    return a.L == 2 ? launch_wino4_cfg<2, 2>(w, flops_u, s) : launch_wino4_cfg<3, 2>(w, flops_u, s);
  }
  const double flops = 2.0 * (double)(a.M / 16) * 36.0 * a.Cin * a.Cout;
  switch (a.L) {
    case 2: return launch_wino4_cfg<2>(w, flops, s);
    case 3: return launch_wino4_cfg<3>(w, flops, s);
    default: return launch_wino4_cfg<4>(w, flops, s);
  }
}

__global__ __launch_bounds__(256) void wino4_weight_pack_kernel(const float* __restrict__ w, WeightMap map, float* __restrict__ u, int ups) {
  const uint32_t total = (uint32_t)map.Ca * (uint32_t)map.Cb;
  for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) wino4_pack_entry(w, map, e, u, ups != 0);
}

int launch_wino4_weight_pack(const float* w, const WeightMap& map, float* u, hipStream_t s, int ups) {
  int64_t gx = ceil_div((int64_t)map.Ca * map.Cb, 256);
  if (gx > 1024) gx = 1024;
  DVG_LAUNCH(K_WEIGHT_PACK, wino4_weight_pack_kernel, dim3((unsigned)gx), dim3(256), 0, s, w, map, u, ups);
  return DVG_OK;
}

}  // namespace dvg
