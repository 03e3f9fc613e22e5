// Internal interfaces of the convolution machinery (not part of the C ABI).
//
// Activation layout inside the library: NHWC float32 with the pixels of each image
// in Morton (Z-curve) order.  Morton order makes every 2x2 window contiguous at
// every resolution: MaxPool2d(2) reads rows 4q..4q+3, nearest Upsample(x2) is
// `index >> 2`, and the 2x2-sum that is the adjoint of the upsample falls on four
// consecutive rows of an MFMA accumulator (one lane's registers).
#pragma once
#include "common.h"

namespace dvg {

__host__ __device__ __forceinline__ uint32_t compact1by1(uint32_t v) {
  v &= 0x55555555u;
  v = (v | (v >> 1)) & 0x33333333u;
  v = (v | (v >> 2)) & 0x0f0f0f0fu;
  v = (v | (v >> 4)) & 0x00ff00ffu;
  return v;
}
__host__ __device__ __forceinline__ uint32_t part1by1(uint32_t v) {
  v &= 0xffu;
  v = (v | (v << 4)) & 0x0f0fu;
  v = (v | (v << 2)) & 0x3333u;
  v = (v | (v << 1)) & 0x5555u;
  return v;
}
__host__ __device__ __forceinline__ uint32_t morton(uint32_t y, uint32_t x) { return part1by1(x) | (part1by1(y) << 1); }
__host__ __device__ __forceinline__ uint32_t morton_y(uint32_t p) { return compact1by1(p >> 1); }
__host__ __device__ __forceinline__ uint32_t morton_x(uint32_t p) { return compact1by1(p); }

// How a packed GEMM weight Wp[tap][a][b] (a: reduction channel, b: output channel) maps to the
// checkpoint (torch) layout.
enum WeightMode : int {
  WM_CONV_FWD = 0,    // Conv2d weight (Cout,Cin,3,3): a = ci, b = co
  WM_CONV_DGRAD,      // a = co, b = ci, taps flipped
  WM_CONVT_FWD,       // ConvTranspose2d weight (Cin,Cout,3,3): a = ci, b = co, taps flipped
  WM_CONVT_DGRAD,     // a = co, b = ci
  WM_LIN_FWD,         // Linear (4n, n): a = ci, b = p*n + c  <->  row c*4 + p   (NHWC output)
  WM_LIN_DGRAD,       // a = p*n + c, b = ci
  // ConvTranspose2d behind a nearest Upsample(x2), folded (see ConvArgs.fold): 16 "taps" = 4 output parity classes
  // x 2x2 source pixels; each packed entry is the SUM of the 1, 2 or 4 original taps that read that source pixel
  WM_CONVT_FOLD_FWD,  // tap = cls*4 + t: a = ci, b = co
  WM_CONVT_FOLD_DGRAD,// tap = cls*4 + t: a = co, b = ci
  // ConvTranspose2d 3x3 (padding 1) on 2x2 images as ONE dense map per image: of the 9 taps of an output pixel only the
  // 4 that land inside the image multiply anything but padding, and input pixel p_in reaches output pixel p_out through
  // exactly one tap.  1-tap GEMM over images: Ca = 4 Cin, Cb = 4 Cout, ntaps = 1; 16/36 of the 9-tap form's FLOPs.
  WM_CONVT_D22_FWD,   // a = p_in*Cin + ci, b = p_out*Cout + co   (p = Morton index of the pixel: x | y << 1)
  WM_CONVT_D22_DGRAD  // a = p_out*Cout + co, b = p_in*Cin + ci
};

// Folding a 3x3 kernel over a x2-upsampled input: output pixel (2i+pa, 2j+pb) reads source rows i-1+pa+dr, dr in
// {0,1}; kernel row r (0..2, forward-GEMM tap order) lands on dr = fold_src(pa, r).
__host__ __device__ __forceinline__ int fold_src(int parity, int r) { return parity == 0 ? (r >= 1) : (r >= 2); }

struct WeightMap {
  int mode, Ca, Cb, ntaps;
};

__host__ __device__ __forceinline__ int64_t torch_weight_offset(const WeightMap& w, int tap, int a, int b) {
  switch (w.mode) {
    case WM_CONV_FWD: return ((int64_t)b * w.Ca + a) * 9 + tap;
    case WM_CONV_DGRAD: return ((int64_t)a * w.Cb + b) * 9 + (8 - tap);
    case WM_CONVT_FWD: return ((int64_t)a * w.Cb + b) * 9 + (8 - tap);
    case WM_CONVT_DGRAD: return ((int64_t)b * w.Ca + a) * 9 + tap;
    case WM_LIN_FWD: { const int c = b % w.Ca, p = b / w.Ca; return (int64_t)(c * 4 + p) * w.Ca + a; }
    case WM_CONVT_D22_FWD:
    case WM_CONVT_D22_DGRAD: {
      const bool fwd = w.mode == WM_CONVT_D22_FWD;
      const int Cin = (fwd ? w.Ca : w.Cb) / 4, Cout = (fwd ? w.Cb : w.Ca) / 4;
      const int ai = fwd ? a : b, bo = fwd ? b : a;  // (p_in, ci) index, (p_out, co) index
      const int p_in = ai / Cin, ci = ai % Cin, p_out = bo / Cout, co = bo % Cout;
      const int t = ((p_in >> 1) - (p_out >> 1) + 1) * 3 + ((p_in & 1) - (p_out & 1) + 1);  // forward-GEMM tap
      return ((int64_t)ci * Cout + co) * 9 + (8 - t);  // ConvTranspose2d (Cin,Cout,3,3), taps flipped
    }
    default: { const int c = a % w.Cb, p = a / w.Cb; return (int64_t)(c * 4 + p) * w.Cb + b; }
  }
}

// Value of packed entry (tap, a, b): one checkpoint element, or for the folded layouts the sum (fixed order) of the
// original taps that share a source pixel.
__host__ __device__ __forceinline__ float packed_weight(const float* w, const WeightMap& m, int tap, int a, int b) {
  if (m.mode != WM_CONVT_FOLD_FWD && m.mode != WM_CONVT_FOLD_DGRAD) return w[torch_weight_offset(m, tap, a, b)];
  const int cls = tap >> 2, t = tap & 3, pa = cls >> 1, pb = cls & 1, dr = t >> 1, dc = t & 1;
  const int ci = m.mode == WM_CONVT_FOLD_FWD ? a : b, co = m.mode == WM_CONVT_FOLD_FWD ? b : a;
  const int Cout = m.mode == WM_CONVT_FOLD_FWD ? m.Cb : m.Ca;
  float acc = 0.f;
  for (int r = 0; r < 3; ++r)
    for (int s = 0; s < 3; ++s)
      if (fold_src(pa, r) == dr && fold_src(pb, s) == dc)
        acc += w[((int64_t)ci * Cout + co) * 9 + (8 - (r * 3 + s))];  // ConvTranspose2d (Cin,Cout,3,3), taps flipped
  return acc;
}

// One (a, b) entry of the Winograd weight pack U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1] (conv_wino.hip):
// u[e][16] with e = a Cb + b, the four 16-byte slots (one per row of U) XOR-swizzled by (b >> 2) & 3.
__device__ __forceinline__ void wino_pack_entry(const float* __restrict__ w, const WeightMap& map, uint32_t e,
                                                float* __restrict__ u) {
  typedef float pack_f32x4 __attribute__((ext_vector_type(4)));
  const uint32_t av = e / (uint32_t)map.Cb, b = e - av * (uint32_t)map.Cb;
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = packed_weight(w, map, t, (int)av, (int)b);
  float tg[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    tg[0][s] = g[s];
    tg[1][s] = 0.5f * ((g[s] + g[3 + s]) + g[6 + s]);
    tg[2][s] = 0.5f * ((g[s] - g[3 + s]) + g[6 + s]);
    tg[3][s] = g[6 + s];
  }
  const int sw = (int)(b >> 2) & 3;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    pack_f32x4 o;
    o[0] = tg[x][0];
    o[1] = 0.5f * ((tg[x][0] + tg[x][1]) + tg[x][2]);
    o[2] = 0.5f * ((tg[x][0] - tg[x][1]) + tg[x][2]);
    o[3] = tg[x][2];
    *reinterpret_cast<pack_f32x4*>(u + (size_t)e * 16 + ((x ^ sw) << 2)) = o;
  }
}

// One (a, b) entry of the F(4x4, 3x3) weight pack U = G g G^T (conv_wino4.hip), G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6;
// 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]: 36 floats (position 6 xi + nu) at [a / 4][b / 32][a % 4][b % 32] -- the 18 KiB a
// workgroup's chunk reads (4 reduction channels x 32 output channels) are one contiguous piece.
__device__ __forceinline__ void wino4_pack_entry(const float* __restrict__ w, const WeightMap& map, uint32_t e,
                                                 float* __restrict__ u, bool ups = false) {
  typedef float pack_f32x4 __attribute__((ext_vector_type(4)));
  const uint32_t av = e / (uint32_t)map.Cb, b = e - av * (uint32_t)map.Cb;
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = packed_weight(w, map, t, (int)av, (int)b);
  float tg[6][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float g0 = g[s], g1 = g[3 + s], g2 = g[6 + s];
    tg[0][s] = 0.25f * g0;
    tg[1][s] = (-1.f / 6.f) * ((g0 + g1) + g2);
    tg[2][s] = (-1.f / 6.f) * ((g0 - g1) + g2);
    tg[3][s] = (g0 * (1.f / 24.f) + g1 * (1.f / 12.f)) + g2 * (1.f / 6.f);
    tg[4][s] = (g0 * (1.f / 24.f) - g1 * (1.f / 12.f)) + g2 * (1.f / 6.f);
    tg[5][s] = g2;
  }
  float o[36];
#pragma unroll
  for (int x = 0; x < 6; ++x) {
    const float g0 = tg[x][0], g1 = tg[x][1], g2 = tg[x][2];
    o[6 * x + 0] = 0.25f * g0;
    o[6 * x + 1] = (-1.f / 6.f) * ((g0 + g1) + g2);
    o[6 * x + 2] = (-1.f / 6.f) * ((g0 - g1) + g2);
    o[6 * x + 3] = (g0 * (1.f / 24.f) + g1 * (1.f / 12.f)) + g2 * (1.f / 6.f);
    o[6 * x + 4] = (g0 * (1.f / 24.f) - g1 * (1.f / 12.f)) + g2 * (1.f / 6.f);
    o[6 * x + 5] = g2;
  }
  const size_t cell = (size_t)((av >> 2) * ((uint32_t)map.Cb >> 5) + (b >> 5)) * 128 + (size_t)((av & 3) * 32 + (b & 31));
  if (ups) {
    // behind the x2 upsample: the 25 positions whose transformed input does not vanish (xi, nu over {0, 1, 3, 4, 5}), 28 floats
    float o5[28];
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
      for (int n = 0; n < 5; ++n) o5[5 * x + n] = o[6 * (x < 2 ? x : x + 1) + (n < 2 ? n : n + 1)];
    o5[25] = o5[26] = o5[27] = 0.f;
    float* dst = u + cell * 28;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const pack_f32x4 v = {o5[4 * q], o5[4 * q + 1], o5[4 * q + 2], o5[4 * q + 3]};
      *reinterpret_cast<pack_f32x4*>(dst + 4 * q) = v;
    }
    return;
  }
  float* dst = u + cell * 36;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const pack_f32x4 v = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
    *reinterpret_cast<pack_f32x4*>(dst + 4 * q) = v;
  }
}

// Implicit-GEMM 3x3 (or 1-tap) convolution:  out[m][co] = sum_{tap,ci} in[nbr(m,tap)][ci] Wp[tap][ci][co] (+ bias)
struct ConvArgs {
  const float* in;    // [(images * HW_in), Cin]; HW_in = HW/4 when `ups` (nearest-upsampled on the fly)
  const float* wp;    // [ntaps][Cin][Cout]
  const float* bias;  // [Cout] or null
  float* out;         // [M, Cout], or [M/4, Cout] when `poolsum`
  float* stats;       // [m_blocks][Cout][2] per-block (sum, sum of squares) of the output, or null
  int64_t M;          // images * HW (output-resolution pixels)
  int Cin, Cout;
  int L;              // log2(H) = log2(W) at the output resolution (0 for the 1-tap linear layer)
  int ntaps;          // 9 or 1
  int ups;            // input is stored at half resolution
  int poolsum;        // sum each 2x2 output quad (adjoint of the upsample) before storing
  // Folded upsample (x2 nearest) + 3x3: the 9 taps of an output pixel read only 2x2 distinct source pixels, so the
  // layer is 4 GEMMs (one per output parity class) with 4 pre-summed taps each: 4/9 of the FLOPs.
  //   fold = 1 (forward):  rows = SOURCE pixels (M = their count, L = log2 of the source side), ntaps = 4,
  //            wp = [16][Cin][Cout] (WM_CONVT_FOLD_FWD); grid.x also spans the 4 classes; out row = 4*m + class.
  //   fold = 2 (data gradient wrt the source map): rows = SOURCE pixels, ntaps = 16, `in` = dY at the output
  //            resolution, wp = [16][Cout_fwd][Cin_fwd] (WM_CONVT_FOLD_DGRAD); out row = m.
  int fold = 0;
  float* splitk_ws;   // scratch of conv_splitk_floats() floats, or null: never split K
  int ksplit;         // set by launch_conv_igemm
  // operand form of the launch (set by launch_conv_igemm from conv_launch_mode; the PM argument of conv_igemm_kernel):
  // 0 float32 staged through registers (`wp` = [tap][Cin][Cout]); 3 float32 by LDS-DMA; 4 float32 operands as three bf16
  // pieces, split at operand-read time ("f32x3"); 5 operands rounded to bf16 at operand-read time ("bf16 inputs").
  // 3 / 4 / 5 read the same float32 K-major pack `wp` = [tap][Cout][Cin].
  int bf16 = 0;
  // force_f32 = 1: this launch runs in float32 (form 3 or 0) whatever the process-wide mode: the weight-space products of
  // the composed decoder layers (their operands are weights, not activations of the network)
  int force_f32 = 0;
  // bias_perm = n > 0: column j = p*n + c takes bias[c*4 + p] (the decoder's Linear(n, 4n) bias in checkpoint order)
  int bias_perm = 0;
  // bias_mod = C > 0: column j = p*C + c takes bias[c] (the dense 2x2 form: one bias per channel, four pixels per row)
  int bias_mod = 0;
  // posmajor (set by launch_conv_igemm): the rows of a block tile are ONE pixel position of BM consecutive images instead
  // of BM consecutive pixels, so every row of the tile has the same taps inside the image and the K loop runs over those
  // only: the padding taps of the image border ((3H-2)^2 of 9 H^2 (pixel, tap) pairs are real: 69 % at 4x4, 84 % at 8x8)
  // are never staged or multiplied.  Same outputs, same number of BatchNorm partial rows.
  int posmajor = 0;
  // Winograd launches (launch_conv_wino): CUs the persistent grid is sized for, 0 = all 256 (encoder.cpp: a training
  // call's forward launches leave the sampler's CUs out, its data gradients share the chip with the weight-gradient chain)
  int wino_cus = 0;
  // Winograd launches of the decoder's Upsample(x2) + 3x3 layers: 1 = forward (`in` is the source map, M / L of the output
  // grid: 9 of 16 transform positions), 2 = data gradient (`out` = the source map's gradient, one row per output quad)
  int wino_um = 0;
};
// CU budgets of the Winograd launches of a TRAINING call (whole-CU persistent grids: a launch is sized to the CUs its
// neighbours leave).  Measured at c3, both wave forms: the encoder's data gradient and weight gradient side
// by side on half the chip each -- (128, 128) 8.41 ms, (144, 112) 8.65, (112, 144) 8.64, (160, 96) 8.7-8.8, (96, 160)
// 8.86 (round 5) -- the decoder's launches at the whole chip (its weight gradient at 128 / 192: 8.49 / 8.46 against 8.41).
constexpr int WINO_CUS_ENC_DGRAD = 128, WINO_CUS_ENC_WGRAD = 128, WINO_CUS_DEC = 256;
// encoder Winograd launches of a training call: from this many workgroups' worth of tile blocks up (measured, n = 512
// model: 1024 / 512 / 256 -> B = 512: 2.16 / 2.13 / 2.06 ms, B = 1024: 3.03 / 2.82 / 2.82, B = 2048: 4.67 / 4.60 / 4.53;
// c2: 0.920 / 0.924 / 0.953 -- 512 is the lowest value that costs c2 nothing)
constexpr int WINO_MIN_BLOCKS = 512;
// process-wide precision of the forward / data-gradient GEMMs (set through the ABI only: dvg_set_conv_precision; the library reads no environment variable)
bool conv_precision_bf16();
int conv_precision_mode();  // 0 f32, 1 bf16 inputs, 2 f32 as three bf16 pieces
// packed-weight buffers: one float32 per entry (every operand form reads a float32 pack)
static inline size_t conv_pack_floats(size_t entries) { return entries; }
void plan_note_forward(const void* ws, uint32_t signature);      // forward calls: what shaped the workspace's contents
bool plan_forward_flag(const void* ws, uint32_t bit);   // the forward noted on `ws` had this signature bit (or none was noted)
bool plan_matches_forward(const void* ws, uint32_t signature);   // backward calls: the same plan as the forward's?
void conv_precision_note_forward(const void* ws);      // forward calls: remember the mode that wrote the packs
bool conv_precision_matches_forward(const void* ws);   // backward calls: same mode as the forward on this workspace?
int launch_conv_igemm(const ConvArgs& a, hipStream_t s);
// Winograd F(2x2,3x3) form of a stride-1 3x3 layer (conv_wino.hip): same ConvArgs, `wp` = the transformed pack of
// launch_wino_weight_pack ([Cin][Cout][16] floats), stats rows = conv_wino_stats_blocks
bool conv_wino_shape(int64_t M, int Cin, int Cout, int L);  // shape only
bool conv_wino_ok(int64_t M, int Cin, int Cout, int L, int kind = 0);  // kind: 0 training forward, 1 data gradient, 2 evaluation forward
int conv_wino_stats_blocks(int64_t M, int Cout);
int launch_conv_wino(const ConvArgs& a, hipStream_t s);
int launch_wino_weight_pack(const float* w, const WeightMap& map, float* u, hipStream_t s);
// Winograd F(4x4,3x3) form of the same layers (conv_wino4.hip): `wp` = the pack of a PackJob with wino = 2 (36 Cin Cout
// floats: wino4_pack_entry), stats rows = conv_wino4_stats_blocks (tile blocks of 1024 pixels)
bool conv_wino4_shape(int64_t M, int Cin, int Cout, int L);  // shape only
bool conv_wino4_ok(int64_t M, int Cin, int Cout, int L);     // policy (option enc_wino4) + shape: asked for launches conv_wino_ok accepted
int conv_wino4_stats_blocks(int64_t M);
int launch_conv_wino4(const ConvArgs& a, hipStream_t s);
int launch_wino4_weight_pack(const float* w, const WeightMap& map, float* u, hipStream_t s, int ups = 0);  // ups: the 25-position pack
// ... and of their weight gradients (conv_wino4_wgrad.hip): slabs [nsplit][36][Cin][Cout] (at most
// conv_wino4_wgrad_slab_floats), summed and transformed back (G^T dU G) into the checkpoint layout `map` by the same call;
// cus = CUs the launch is sized for (0: WINO_CUS_ENC_WGRAD)
bool conv_wino4_wgrad_ok(int64_t M, int Cin, int Cout, int L);      // policy (option enc_wino4) + shape
bool conv_wino4_wgrad_shape(int64_t M, int Cin, int Cout, int L);   // shape only
size_t conv_wino4_wgrad_slab_floats(int64_t M, int Cin, int Cout);
int launch_conv_wino4_wgrad(const float* in, const float* dy, int64_t M, int Cin, int Cout, int L, float* slabs,
                            const WeightMap& map, float* grad_w, hipStream_t s, int cus = 0);
// Winograd form of a stride-1 3x3 layer's WEIGHT gradient (conv_wino_wgrad.hip): slabs [nslabs][16][Cin][Cout] (at most
// conv_wino_wgrad_slab_floats), reduced and transformed back (G^T dU G) into the checkpoint layout `map` by the same call
bool conv_wino_wgrad_ok(int64_t M, int Cin, int Cout, int L);      // policy (options) + shape
bool conv_wino_wgrad_shape(int64_t M, int Cin, int Cout, int L);   // shape only
size_t conv_wino_wgrad_slab_floats(int64_t M, int Cin, int Cout, int L);
// ups = 1: the decoder's Upsample(x2) + 3x3 layers (`in` = the source map, M / L of the output grid: 9 of the 16 position
// GEMMs remain); cus = CUs the launch is sized for (0: WINO_CUS_ENC_WGRAD)
int launch_conv_wino_wgrad(const float* in, const float* dy, int64_t M, int Cin, int Cout, int L, float* slabs,
                           const WeightMap& map, float* grad_w, hipStream_t s, int ups = 0, int cus = 0);
int conv_stats_blocks(int64_t M, int Cout);
// fold = 1 launches: M source pixels; usable when conv_fold_ok (whole row blocks per class)
bool conv_fold_ok(int64_t Msrc);
int conv_stats_blocks_fold(int64_t Msrc, int Cout);
// K-split chosen for a launch (1 = none) and the scratch it needs (0 = none)
int conv_igemm_ksplit(int64_t M, int Cin, int Cout, int ntaps);
size_t conv_splitk_floats(int64_t M, int Cin, int Cout, int ntaps, int poolsum);  // number of m_blocks launch_conv_igemm will use (for `stats`)

// Weight-gradient GEMM:  dWp[tap][a][b] = sum_m in[nbr(m,tap)][a] * dy[m][b], split over `ksplit` slabs.
struct WgradArgs {
  const float* in;   // as ConvArgs.in
  const float* dy;   // [M, Cout]
  float* slabs;      // [ksplit][ntaps][Cin][Cout]
  int64_t M;
  int Cin, Cout, L, ntaps, ups, ksplit;
  // fold = 1: weight gradient of the folded Upsample(x2)+3x3 layer (ConvArgs.fold): `in` is the SOURCE map (M source
  // pixels, side 2^L), `dy` has 4*M rows, ntaps = 16 and the slabs hold the 16 (class, tap) pairs; the reduce pass
  // (launch_wgrad_reduce with fold = true) sums the 4 classes back onto the 9 checkpoint taps.
  int fold = 0;
};
int wgrad_ksplit(int64_t M, int Cin, int Cout, int ntaps);
int wgrad_fold_ksplit(int64_t Msrc, int Cin, int Cout);
int launch_wgrad_fold_reduce(const float* slabs, int ksplit, int Cin, int Cout, float* grad_w, hipStream_t s);
// dense 2x2 form (WM_CONVT_D22_FWD): slabs [ksplit][4 Cin][4 Cout] -> ConvTranspose2d gradient (Cin,Cout,3,3): every tap
// sums the (p_in, p_out) blocks that use it (4 for the centre tap, 2 for an edge, 1 for a corner), slabs in order
int launch_wgrad_d22_reduce(const float* slabs, int ksplit, int Cin, int Cout, float* grad_w, hipStream_t s);
// Helpers of the composed Linear + dense-2x2 form of the decoder's first two layers (decoder.cpp):
// out[r][c] = sum of the slabs [ksplit][rows][cols] in order (8 lanes per element), and its transpose outT[c][r]
int launch_slab_sum(const float* slabs, int ksplit, int rows, int cols, float* out, float* outT, hipStream_t s);
// bc[o] = sum_j lin_b[c*4 + p] * WkEff[o][j] + conv_b[o % C]   (j = p*n + c; WkEff = the K-major dense-2x2 forward pack)
int launch_lc0_bias(const float* lin_b, const float* wk_eff, const float* conv_b, int n, int C, float* bc, hipStream_t s);
// grad_lin_b[c*4 + p] = sum_o dbc[o] * WkD[j][o]   (WkD = the K-major dense-2x2 data-gradient pack, rows j = p*n + c)
int launch_lc0_lin_bias_grad(const float* dbc, const float* wk_d, int n, int C, float* grad_lin_b, hipStream_t s);
// dWeff[(p*n + c)][o] += lin_b[c*4 + p] * dbc[o]   (dWeff: [4n][4C])
int launch_lc0_rank1_add(float* dweff, const float* lin_b, const float* dbc, int n, int C, hipStream_t s);
// grad_lin_w[(c*4 + p)][i] = t[(p*n + c)][i]   (rows of the [4n][n] product back into the Linear weight's row order)
int launch_lc0_rows_to_linear(const float* t, int n, float* grad_lin_w, hipStream_t s);
int launch_conv_wgrad(const WgradArgs& a, hipStream_t s);
// sums the slabs in order and scatters into the checkpoint layout (grad_w is overwritten)
int launch_wgrad_reduce(const float* slabs, int ksplit, const WeightMap& map, float* grad_w, hipStream_t s);
// Wp[tap][a][b] <- checkpoint-layout weight
int launch_weight_pack(const float* w, const WeightMap& map, float* wp, hipStream_t s, int bf16t = -1);  // -1: the process-wide mode
// several packs in ONE launch (forward and data-gradient packs of a whole network)
// bf16t: the operand form of the launch the pack feeds (conv_launch_mode): 0 -> [tap][a][b], 3 / 4 / 5 -> K-major [tap][b][a]
// rows: GEMM rows of the launch this pack feeds (launch_conv_igemm's M over all classes): together with map.Cb it
// decides the operand format that launch will use (conv_launch_mode); 0 = unknown: the process-wide mode as it stands
// wino = 1: the job writes the Winograd pack [Ca][Cb][16] of a 9-tap map instead (wino_pack_entry); 2: the F(4x4,3x3) pack (wino4_pack_entry); 3: its 25-position form behind the x2 upsample
struct PackJob { const float* w; float* wp; WeightMap map; int bf16t = 0; int64_t rows = 0; int wino = 0; };
// operand form of one forward / data-gradient launch (ConvArgs.bf16): the process-wide mode mapped onto the kernels
int conv_launch_mode(int64_t gemm_rows, int Cout);
bool conv_pack_is_f32_kmajor(int launch_mode);  // the pack such a launch reads is the float32 K-major one
constexpr int MAX_PACK_JOBS = 12;
int launch_weight_pack_multi(const PackJob* jobs, int njobs, hipStream_t s);

}  // namespace dvg
