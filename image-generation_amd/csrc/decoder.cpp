// dvg_decoder_fwd / dvg_decoder_bwd: Decoder.forward of /root/reference/src/decoder.py:54-62 and its backward.
//   Linear(n,4n) as a 1-tap MFMA GEMM writing NHWC (N,2x2,n) directly;
//   layers 0-2 (n->128 @2x2, 128->64 @4x4, 64->32 @8x8): MFMA implicit GEMM with the x2 nearest upsample of the
//   previous stage fused into the gather (never materialised); layer 3 (32->1 @16x16) and the final 1->1 @32x32
//   ConvTranspose: VALU kernels.  Each of layers 0-3 is followed by BN(batch stats) -> Dropout2d -> LeakyReLU.
//   Backward data-gradients of the upsample-fused layers sum each 2x2 quad in the MFMA epilogue.
#include "conv.h"
#include "kernels.h"

using namespace dvg;

namespace {

struct DecPlan {
  int64_t N;
  int n;
  int ch[5];
  int64_t M[4];
  int L[4];
  int nblk[4];
  size_t X0, Y[4], Xs[4], mean[4], invstd[4], stats[4], mask[4];
  size_t wp_lin, wpd_lin, bias_lin, wp[3], wpd[3];
  size_t dXbuf, dYl[4], slabs, partA, partB[4], partL, partW, partF, partT, msep, splitk;
  int ksplit_lin, ksplit[3];
  bool fold[4];  // layer runs in the folded-upsample form (conv.h: ConvArgs.fold)
  bool wino_w[4];  // layer's weight gradient runs in the Winograd form (conv_wino_wgrad.hip, 9 of 16 positions)
  bool wino_f[4], wino_d[4];  // ... its forward / data-gradient GEMM (conv_wino.hip, wino_um = 1 / 2)
  bool wino4_f[4], wino4_d[4];  // ... the forward / data gradient in the F(4x4,3x3) form with 25 of 36 positions (conv_wino4.hip, wino_um = 1 / 2)
  bool d22;      // layer 0 (3x3 on 2x2 images) runs as the dense per-image map (conv.h: WM_CONVT_D22_FWD)
  // Large batches: Linear(n, 4n) and layer 0 have nothing but a reshape between them, so the pair is ONE linear map per
  // image, spins (n) -> the 4 x 128 pre-BatchNorm values, with the composed weight Wc = Wlin . Weff (a 2 n x 4n x 512
  // weight-space product per step).  Forward, data gradient and weight gradient then cost 3 x 2 N n 512 FLOPs instead of
  // 3 x 2 N (n 4n + 4n 512); the gradients of BOTH original parameter tensors follow by the chain rule in weight space
  // (dWlin = dWc . Weff^T, dWeff = Wlin^T . dWc + blin (x) dbc).  X0 is never materialised.
  bool lc0;
  bool tail;  // layer 2's BN/Dropout/LeakyReLU folded into layer 3's kernels (special.hip: DecActIn)
  size_t WcT, Wc, bc, dbc, dWc, dWcT, T1, dWeff, partC;
  int ksplit_c;
  size_t total_floats;
};

// option dec_fold = 0 keeps the upsample-fused 9-tap form everywhere (A/B runs)
bool fold_enabled() { return opt(OPT_DEC_FOLD) != 0; }

// option dec_d22 = 0 keeps the 9-tap form for the first ConvTranspose layer (A/B runs, tests)
bool d22_enabled() { return opt(OPT_DEC_D22) != 0; }

// option dec_lc0 = 0: never compose the Linear layer with layer 0 (A/B runs, tests); 1: also for small batches (tests);
// -1 (default): from 4096 rows up, or N * n_latents >= 2^20 (dec_plan)
int lc0_env() {
  const int64_t v = opt(OPT_DEC_LC0);
  return v < 0 ? -1 : (v ? 1 : 0);
}

size_t bump(size_t& o, size_t count) {
  const size_t r = o;
  o += (count + 63) & ~(size_t)63;
  return r;
}

DecPlan dec_plan(int64_t N, int n) {
  DecPlan p;
  p.N = N; p.n = n;
  const int ch[5] = {n, 128, 64, 32, 1};
  for (int i = 0; i < 5; ++i) p.ch[i] = ch[i];
  size_t o = 0;
  p.X0 = bump(o, (size_t)N * 4 * n);
  p.wp_lin = bump(o, conv_pack_floats((size_t)n * 4 * n));
  p.wpd_lin = bump(o, conv_pack_floats((size_t)n * 4 * n));
  p.bias_lin = bump(o, (size_t)4 * n);
  p.ksplit_lin = wgrad_ksplit(N, n, 4 * n, 1);
  size_t max_slab = (size_t)p.ksplit_lin * n * 4 * n;
  size_t max_dx = (size_t)N * 4 * n;
  size_t max_split = conv_splitk_floats(N, n, 4 * n, 1, 0);
  { const size_t sd = conv_splitk_floats(N, 4 * n, n, 1, 0); if (sd > max_split) max_split = sd; }
  int cmax = 4 * n;
  p.d22 = d22_enabled();
  for (int l = 0; l < 4; ++l) {
    p.L[l] = l + 1;
    p.M[l] = N * ((int64_t)4 << (2 * l));
    const int C = ch[l + 1];
    p.fold[l] = (l == 1 || l == 2) && fold_enabled() && conv_fold_ok(p.M[l] / 4);
    // (weight gradients are float32 in every operand mode, so this form serves all three)
    p.wino_w[l] = (l == 1 || l == 2) && opt(OPT_DEC_WINO) != 0 &&
                  conv_wino_wgrad_shape(p.M[l], ch[l], C, p.L[l]) && (opt(OPT_DEC_WINO) > 0 || N >= 8192);
    {
      // option dec_wino: -1 (default) forward, data gradient and weight gradient from 8192 decoder rows up; 1 whenever
      // the shape allows, 0 never.  (Round 4's one-wave-per-SIMD data gradient measured neutral inside the c3 step and
      // stayed off; the two-waves-per-SIMD kernel of round 5 pays: 8.48 -> 8.32 ms, same box.)
      const int64_t o = opt(OPT_DEC_WINO);
      // (strict float32 arithmetic: also the faster form in the f32x3 mode -- 4/9 of the multiplications at f32 rate
      // against 6/16 of the matrix time plus the split arithmetic -- and at least as exact; the bf16-input mode keeps
      // the direct kernel on the bf16 MFMA)
      const bool on = (l == 1 || l == 2) && o != 0 && conv_precision_mode() != 1 && (o > 0 || N >= 8192);
      p.wino_f[l] = on && conv_wino_shape(p.M[l], ch[l], C, p.L[l]);
      p.wino_d[l] = on && conv_wino_shape(p.M[l], C, ch[l], p.L[l]);
      p.wino4_f[l] = p.wino_f[l] && opt(OPT_ENC_WINO4) != 0 && ((opt(OPT_DEC_WINO4_MASK) >> (l - 1)) & 1) && p.L[l] <= 3 &&
                     conv_wino4_shape(p.M[l], ch[l], C, p.L[l]);
      p.wino4_d[l] = p.wino_d[l] && opt(OPT_ENC_WINO4) != 0 && ((opt(OPT_DEC_WINO4_MASK) >> (l + 1)) & 1) && p.L[l] <= 3 &&
                     conv_wino4_shape(p.M[l], C, ch[l], p.L[l]);
    }
    p.nblk[l] = l == 3 ? dec_conv3_blocks(N) : p.wino4_f[l] ? conv_wino4_stats_blocks(p.M[l]) : p.wino_f[l] ? conv_wino_stats_blocks(p.M[l], C) : p.fold[l] ? conv_stats_blocks_fold(p.M[l] / 4, C) : conv_stats_blocks(p.M[l], C);
    // dense 2x2 form: the GEMM has N rows of 4 C columns; a row block's partials [4 C][2] read as 4 rows of [C][2]
    if (l == 0 && p.d22) p.nblk[l] = 4 * conv_stats_blocks(N, 4 * C);
    p.Y[l] = bump(o, (size_t)p.M[l] * C);
    p.Xs[l] = bump(o, (size_t)p.M[l] * C);
    p.mean[l] = bump(o, C);
    p.invstd[l] = bump(o, C);
    p.stats[l] = bump(o, (size_t)(p.nblk[l] + BN_FOLD_ROWS) * C * 2);  // + scratch rows of launch_bn_finalize
    p.mask[l] = bump(o, (size_t)N * C);
    if (l < 3) {
      p.wp[l] = bump(o, conv_pack_floats((size_t)28 * ch[l] * C));   // 9 taps, 16 folded (class, tap) pairs, or the 25-position F(4x4) pack
      p.wpd[l] = bump(o, conv_pack_floats((size_t)28 * ch[l] * C));
      const bool d22 = l == 0 && p.d22;
      p.ksplit[l] = d22 ? wgrad_ksplit(N, 4 * ch[l], 4 * C, 1)
                        : p.fold[l] ? wgrad_fold_ksplit(p.M[l] / 4, ch[l], C) : wgrad_ksplit(p.M[l], ch[l], C, 9);
      const size_t slab = (size_t)p.ksplit[l] * (p.fold[l] || d22 ? 16 : 9) * ch[l] * C;
      if (slab > max_slab) max_slab = slab;
      if (p.wino_w[l]) {
        const size_t sw = conv_wino_wgrad_slab_floats(p.M[l], ch[l], C, p.L[l]);
        if (sw > max_slab) max_slab = sw;
      }
      const size_t sk_f = d22 ? conv_splitk_floats(N, 4 * ch[l], 4 * C, 1, 0)
                              : p.fold[l] ? conv_splitk_floats(p.M[l], ch[l], C, 4, 0) : conv_splitk_floats(p.M[l], ch[l], C, 9, 0);
      const size_t sk_d = d22 ? conv_splitk_floats(N, 4 * C, 4 * ch[l], 1, 0)
                              : p.fold[l] ? conv_splitk_floats(p.M[l] / 4, C, ch[l], 16, 0) : conv_splitk_floats(p.M[l], C, ch[l], 9, l > 0);
      if (sk_f > max_split) max_split = sk_f;
      if (sk_d > max_split) max_split = sk_d;
    }
    const size_t act = (size_t)p.M[l] * C;
    if (act > max_dx) max_dx = act;  // dXs[l] has the shape of Xs[l]
  }
  {
    const int C = ch[1], env = lc0_env();
    // composed form: the LDS-DMA GEMMs only (their K-major float32 packs double as plain row-major matrices); every
    // operand mode has them (conv_launch_mode: 3 float32, 4 f32x3, 5 bf16 inputs), the register-staged A/B form does not
    // (from 4096 rows up at c3's 256 latent spins -- measured there in round 4 -- and from N n >= 2^20 in general: the two
    // layers cost 2 (4 n^2 + 16 n C) FLOPs per row and pass, the composed map 2 (4 n C), forming it ~128 C n^2 per step; at
    // c5's n = 1024 the crossover measured between 512 and 1024 rows, and at its 2048 rows the step went 2.57 -> 2.18 ms)
    p.lc0 = p.d22 && env != 0 && (N >= 4096 || N * (int64_t)n >= (1 << 20) || env == 1) && conv_pack_is_f32_kmajor(conv_launch_mode(N, 4 * C));
    p.tail = opt(OPT_DEC_TAIL_FUSED) != 0;
    if (p.lc0) {
      p.WcT = bump(o, (size_t)4 * C * n);
      p.Wc = bump(o, (size_t)n * 4 * C);
      p.bc = bump(o, (size_t)4 * C);
      p.dbc = bump(o, (size_t)4 * C);
      p.dWc = bump(o, (size_t)n * 4 * C);
      p.dWcT = bump(o, (size_t)4 * C * n);
      p.T1 = bump(o, (size_t)4 * n * n);
      p.dWeff = bump(o, (size_t)4 * n * 4 * C);
      p.partC = bump(o, (size_t)EW_BLOCKS * 4 * C);
      p.ksplit_c = wgrad_ksplit(N, n, 4 * C, 1);
      const size_t slab = (size_t)p.ksplit_c * n * 4 * C;
      if (slab > max_slab) max_slab = slab;
      // (the composed forward GEMM: N rows, n -> 4 C; its BatchNorm partials as in the dense form)
      p.nblk[0] = 4 * conv_stats_blocks(N, 4 * C);
      const size_t sk = conv_splitk_floats(N, n, 4 * C, 1, 0), sk2 = conv_splitk_floats(N, 4 * C, n, 1, 0);
      if (sk > max_split) max_split = sk;
      if (sk2 > max_split) max_split = sk2;
    }
  }
  p.dXbuf = bump(o, max_dx);
  // one dY buffer per layer: a layer's weight gradient (side stream) may still be reading its dY when the data-gradient
  // chain reaches the next layers, and a shared buffer would make the caller's stream wait on the side stream
  for (int l = 0; l < 4; ++l) p.dYl[l] = bump(o, (size_t)p.M[l] * ch[l + 1]);
  p.slabs = bump(o, max_slab);
  p.partA = bump(o, (size_t)EW_BLOCKS * 2 * cmax);
  // one bias-gradient partial buffer per layer: their column sums run on the side stream, behind the main chain
  for (int l = 0; l < 4; ++l) p.partB[l] = bump(o, (size_t)(l >= 2 ? STREAM_BLOCKS : EW_BLOCKS) * ch[l + 1]);  // (l = 2, 3: rows of the fused tail)
  p.partL = bump(o, (size_t)EW_BLOCKS * cmax);
  p.partF = bump(o, (size_t)EW_BLOCKS * 10);
  p.partW = bump(o, (size_t)STREAM_BLOCKS * 288);
  // the fused tail (dvg_decoder_fwd_mse_ex): the 1-channel stage's (sum dz, sum dz zhat) partials, written by the FORWARD call
  // and read by the backward call (a buffer of their own: partA is every other stage's scratch), and the MSE partials (doubles)
  p.partT = bump(o, (size_t)STREAM_BLOCKS * 2);
  p.msep = bump(o, (size_t)STREAM_BLOCKS * 2);
  p.splitk = bump(o, max_split);
  p.total_floats = o;
  return p;
}

// what of the plan the backward relies on the forward having done (which buffers hold what, in which pack format)
uint32_t plan_signature(const DecPlan& pl) {
  return (uint32_t)pl.d22 | (uint32_t)pl.lc0 << 1 | (uint32_t)pl.fold[1] << 2 | (uint32_t)pl.fold[2] << 3 |
         (uint32_t)conv_launch_mode(pl.N, 128) << 4 | (uint32_t)pl.tail << 12 | (uint32_t)pl.wino_f[1] << 13 |
         (uint32_t)pl.wino_f[2] << 14 | (uint32_t)pl.wino_d[1] << 15 | (uint32_t)pl.wino_d[2] << 16 |
         (uint32_t)pl.wino4_f[1] << 17 | (uint32_t)pl.wino4_f[2] << 18 | (uint32_t)pl.wino4_d[1] << 19 | (uint32_t)pl.wino4_d[2] << 20;
}

int check_common(const dvg_decoder_params_t* p, int n, int64_t N, const void* ws, size_t ws_bytes, const DecPlan& pl) {
  DVG_REQUIRE(p && ws, "decoder: null params/workspace");
  DVG_REQUIRE(n >= 32 && n % 32 == 0 && n <= 4096, "decoder: n_latents=%d must be a multiple of 32", n);
  DVG_REQUIRE(N >= 1 && N <= (1 << 22), "decoder: batch*replicas %lld out of range", (long long)N);
  DVG_REQUIRE(p->lin_w && p->lin_b, "decoder: null linear parameter");
  for (int l = 0; l < 5; ++l) DVG_REQUIRE(p->conv_w[l] && p->conv_b[l], "decoder: null conv parameter %d", l);
  for (int l = 0; l < 4; ++l)
    DVG_REQUIRE(p->bn_g[l] && p->bn_b[l] && p->bn_rm[l] && p->bn_rv[l], "decoder: null BN parameter %d", l);
  if (ws_bytes < pl.total_floats * sizeof(float)) {
    set_error("decoder: workspace %zu < %zu bytes", ws_bytes, pl.total_floats * sizeof(float));
    return DVG_E_WORKSPACE;
  }
  return DVG_OK;
}

}  // namespace

extern "C" size_t dvg_decoder_workspace_bytes(int64_t N, int n_latents) {
  dvg::side_stream_warm();  // the backward's fork/join context exists before any step is captured
  if (N < 1 || n_latents < 32 || n_latents % 32) return 0;
  return dec_plan(N, n_latents).total_floats * sizeof(float);
}

namespace {
// Everything of a forward call that depends on the parameters (and the dropout stream) alone: the weight packs, the
// composed Linear o ConvTranspose weights and bias, the Dropout2d keep-masks.  dvg_decoder_fwd runs it at its head;
// dvg_decoder_prepare runs it ahead of time, beside whatever produces the spins.
int dec_prologue(const dvg_decoder_params_t* p, int n, int64_t N, int training, const float* const dropout_keep[4],
                 uint64_t seed, uint64_t offset, float* W, const DecPlan& pl, const dvg_step_state_t* dyn, hipStream_t s) {
  {  // all weight packs of the network (forward AND data-gradient layouts) in one launch
    PackJob jobs[8];
    jobs[0] = PackJob{p->lin_w, W + pl.wp_lin, WeightMap{WM_LIN_FWD, n, 4 * n, 1}, 0, N};
    jobs[1] = PackJob{p->lin_w, W + pl.wpd_lin, WeightMap{WM_LIN_DGRAD, 4 * n, n, 1}, 0, N};
    for (int l = 0; l < 3; ++l) {
      if (l == 0 && pl.d22) {  // dense 2x2 form: 1-tap GEMMs over the N images
        jobs[2] = PackJob{p->conv_w[0], W + pl.wp[0], WeightMap{WM_CONVT_D22_FWD, 4 * pl.ch[0], 4 * pl.ch[1], 1}, 0, N};
        jobs[3] = PackJob{p->conv_w[0], W + pl.wpd[0], WeightMap{WM_CONVT_D22_DGRAD, 4 * pl.ch[1], 4 * pl.ch[0], 1}, 0, N};
        continue;
      }
      if (pl.fold[l]) {
        // (forward: 4 classes x M/4 source rows = M GEMM rows; data gradient: M/4 source rows)
        jobs[2 + 2 * l] = PackJob{p->conv_w[l], W + pl.wp[l], WeightMap{WM_CONVT_FOLD_FWD, pl.ch[l], pl.ch[l + 1], 16}, 0, pl.M[l]};
        jobs[3 + 2 * l] = PackJob{p->conv_w[l], W + pl.wpd[l], WeightMap{WM_CONVT_FOLD_DGRAD, pl.ch[l + 1], pl.ch[l], 16}, 0, pl.M[l] / 4};
        continue;
      }
      jobs[2 + 2 * l] = PackJob{p->conv_w[l], W + pl.wp[l], WeightMap{WM_CONVT_FWD, pl.ch[l], pl.ch[l + 1], 9}, 0, pl.M[l]};
      jobs[3 + 2 * l] = PackJob{p->conv_w[l], W + pl.wpd[l], WeightMap{WM_CONVT_DGRAD, pl.ch[l + 1], pl.ch[l], 9}, 0, pl.M[l]};
    }
    // Winograd launches (conv_wino.hip, wino_um = 1 / 2) read the transformed pack U = G g G^T instead
    for (int l = 1; l < 3; ++l) {
      if (pl.wino_f[l]) jobs[2 + 2 * l] = PackJob{p->conv_w[l], W + pl.wp[l], WeightMap{WM_CONVT_FWD, pl.ch[l], pl.ch[l + 1], 9}, 0, pl.M[l], pl.wino4_f[l] ? 3 : 1};
      if (pl.wino_d[l]) jobs[3 + 2 * l] = PackJob{p->conv_w[l], W + pl.wpd[l], WeightMap{WM_CONVT_DGRAD, pl.ch[l + 1], pl.ch[l], 9}, 0, pl.M[l], pl.wino4_d[l] ? 3 : 1};
    }
    DVG_TRY(launch_weight_pack_multi(jobs, 8, s));
  }
  if (pl.lc0) {
    // composed weight, both orientations (K-major operands of the forward and of the data-gradient GEMM), and bias:
    //   WcT[o][i] = sum_j Weff[j][o] Wlin[i][j]   rows of the dense-2x2 forward pack [o][j]  x  the Linear dgrad pack [i][j]
    //   Wc[i][o]                                    rows of the Linear dgrad pack [i][j]       x  the dense-2x2 forward pack
    const int C = pl.ch[1];
    ConvArgs a;
    a.bias = nullptr; a.stats = nullptr; a.L = 0; a.ntaps = 1; a.ups = 0; a.poolsum = 0; a.splitk_ws = nullptr;
    a.force_f32 = 1;  // products of WEIGHTS: float32 in every operand mode (the modes are about the network's activations)
    a.in = W + pl.wp[0]; a.wp = W + pl.wpd_lin; a.out = W + pl.WcT; a.M = 4 * C; a.Cin = 4 * n; a.Cout = n;
    DVG_TRY(launch_conv_igemm(a, s));
    a.in = W + pl.wpd_lin; a.wp = W + pl.wp[0]; a.out = W + pl.Wc; a.M = n; a.Cin = 4 * n; a.Cout = 4 * C;
    DVG_TRY(launch_conv_igemm(a, s));
    DVG_TRY(launch_lc0_bias(p->lin_b, W + pl.wp[0], p->conv_b[0], n, C, W + pl.bc, s));
  }
  if (training) {  // Dropout2d keep-masks of all four stages: copied in (parity runs) or drawn on device, one launch
    bool given = dropout_keep != nullptr;
    for (int l = 0; l < 4 && given; ++l) given = dropout_keep[l] != nullptr;
    if (given) {
      for (int l = 0; l < 4; ++l)
        DVG_CHECK_HIP(hipMemcpyAsync(W + pl.mask[l], dropout_keep[l], sizeof(float) * (size_t)N * pl.ch[l + 1],
                                     hipMemcpyDeviceToDevice, s));
    } else {
      DVG_REQUIRE(dropout_keep == nullptr || (!dropout_keep[0] && !dropout_keep[1] && !dropout_keep[2] && !dropout_keep[3]),
                  "decoder_fwd: give all four dropout masks or none");
      float* masks[4] = {W + pl.mask[0], W + pl.mask[1], W + pl.mask[2], W + pl.mask[3]};
      DVG_TRY(launch_dropout_masks(N, &pl.ch[1], masks, seed, offset, dyn ? &dyn->dropout_offset : nullptr, s));
    }
  }
  return DVG_OK;
}

uint64_t prologue_signature(const dvg_decoder_params_t* p, int n, int64_t N, int training, uint64_t seed, uint64_t offset,
                            const dvg_step_state_t* dyn, const DecPlan& pl) {
  uint64_t h = 0xcbf29ce484222325ull;
  auto mix = [&](uint64_t v) { h = (h ^ v) * 0x100000001b3ull; };
  mix((uint64_t)n); mix((uint64_t)N); mix((uint64_t)training); mix(seed); mix(dyn ? ~0ull : offset); mix((uint64_t)(uintptr_t)dyn);
  mix(plan_signature(pl)); mix((uint64_t)conv_precision_mode()); mix((uint64_t)(uintptr_t)p->lin_w); mix((uint64_t)(uintptr_t)p->conv_w[0]);
  return h;
}
}  // namespace

// The weight-only part of the NEXT dvg_decoder_fwd_ex(prepared = 1) on this workspace, enqueued on `stream` now: a
// stream of the caller's that is forked off its main stream where the step starts, so that the packs, weight-space
// products and mask draws (~190 us at B R = 32768) run beside the encoder instead of between the spins and the
// decoder's first GEMM; the forward call joins them (an event recorded here, waited for there).  The parameters must not
// change between the two calls; device-drawn dropout masks only (injected masks: call dvg_decoder_fwd, which copies them).
extern "C" int dvg_decoder_prepare(const dvg_decoder_params_t* p, int n, int64_t N, int training, uint64_t seed,
                                   uint64_t offset, void* ws, size_t ws_bytes, const dvg_step_state_t* dyn,
                                   dvg_stream_t stream) {
  const DecPlan pl = dec_plan(N > 0 ? N : 1, (n >= 32 && n % 32 == 0) ? n : 32);
  DVG_TRY(check_common(p, n, N, ws, ws_bytes, pl));
  hipStream_t s = (hipStream_t)stream;
  DVG_TRY(dec_prologue(p, n, N, training, nullptr, seed, offset, (float*)ws, pl, dyn, s));
  return prep_arm(ws, prologue_signature(p, n, N, training, seed, offset, dyn, pl), s);
}

namespace {
// The MSE against `images` ([N / R][1024]; replica r of image b = decoder row b R + r) fused behind the decoder
// (dvg_decoder_fwd_mse_ex / dvg_decoder_bwd_mse_ex): the reconstruction and its gradient are never written.
struct MseTail { const float* images; int R; float grad_scale; float* loss_out; };
float mse_gscale(const MseTail& mt, int64_t N) { return (float)(2.0 * (double)mt.grad_scale / ((double)N * 1024.0)); }  // (dvg_mse_fwd_bwd's)

int decoder_fwd_impl(const dvg_decoder_params_t* p, int n, const float* spins, int64_t N, int training,
                     const float* const dropout_keep[4], uint64_t seed, uint64_t offset, float* out, const MseTail* mt,
                     void* ws, size_t ws_bytes, const dvg_step_state_t* dyn, int prepared, dvg_stream_t stream) {
  const DecPlan pl = dec_plan(N > 0 ? N : 1, (n >= 32 && n % 32 == 0) ? n : 32);
  DVG_TRY(check_common(p, n, N, ws, ws_bytes, pl));
  conv_precision_note_forward(ws);
  plan_note_forward(ws, plan_signature(pl) | (mt ? 1u << 30 : 0u));
  DVG_REQUIRE(spins && (out || mt), "decoder_fwd: null spins/out");
  if (mt) {
    DVG_REQUIRE(training, "decoder_fwd_mse: training-mode calls only (the fused tail produces the backward's sums)");
    DVG_REQUIRE(pl.tail, "decoder_fwd_mse: needs option dec_tail_fused != 0");
    DVG_REQUIRE(mt->images && mt->loss_out && mt->R >= 1 && N % mt->R == 0, "decoder_fwd_mse: images / loss_out / R (N=%lld R=%d)", (long long)N, mt->R);
  }
  hipStream_t s = (hipStream_t)stream;
  float* W = (float*)ws;
  {
    // a prologue in flight on this workspace is joined whatever it was prepared for (it writes what this call reads
    // or rewrites); it is USED only when the caller says so and it was prepared for exactly this call
    bool armed = false, matched = false;
    DVG_TRY(prep_join(ws, prologue_signature(p, n, N, training, seed, offset, dyn, pl), s, &armed, &matched));
    DVG_REQUIRE(!prepared || (armed && matched && dropout_keep == nullptr),
                "decoder_fwd: prepared = 1 without a matching dvg_decoder_prepare on this workspace (same parameters, N, "
                "training flag, dropout seed / offset, kernel-form options; device-drawn masks)");
    if (!prepared) DVG_TRY(dec_prologue(p, n, N, training, dropout_keep, seed, offset, W, pl, dyn, s));
  }
  if (!pl.lc0) {  // Linear(n, 4n) -> X0 (N, 2x2 Morton, n)
    ConvArgs a;
    a.in = spins; a.wp = W + pl.wp_lin; a.bias = p->lin_b; a.bias_perm = n; a.out = W + pl.X0; a.stats = nullptr;  // bias'[p*n + c] = b[c*4 + p]
    a.M = N; a.Cin = n; a.Cout = 4 * n; a.L = 0; a.ntaps = 1; a.ups = 0; a.poolsum = 0;
    a.splitk_ws = W + pl.splitk;
    DVG_TRY(launch_conv_igemm(a, s));
  }
  const float* x = W + pl.X0;
  for (int l = 0; l < 4; ++l) {
    const int Cin = pl.ch[l], C = pl.ch[l + 1];
    if (l < 3) {
      ConvArgs a;
      a.in = x; a.wp = W + pl.wp[l]; a.bias = p->conv_b[l]; a.out = W + pl.Y[l];
      a.stats = training ? W + pl.stats[l] : nullptr;
      a.M = pl.M[l]; a.Cin = Cin; a.Cout = C; a.L = pl.L[l]; a.ntaps = 9; a.ups = l > 0; a.poolsum = 0;
      if (pl.fold[l]) { a.M = pl.M[l] / 4; a.L = pl.L[l] - 1; a.ntaps = 4; a.ups = 0; a.fold = 1; }
      if (l == 0 && pl.d22) {  // rows = images, columns = (pixel, channel): the same memory as [4 N][C]
        a.M = N; a.Cin = 4 * Cin; a.Cout = 4 * C; a.L = 0; a.ntaps = 1; a.bias_mod = C;
      }
      if (l == 0 && pl.lc0) {  // ... straight from the spins through the composed weight
        a.in = spins; a.wp = W + pl.WcT; a.bias = W + pl.bc; a.bias_mod = 0; a.Cin = n;
      }
      a.splitk_ws = W + pl.splitk;
      if (pl.wino_f[l]) {  // Winograd on the upsampled map: 9 of 16 transform positions (conv_wino.hip, UM = 1)
        a.M = pl.M[l]; a.L = pl.L[l]; a.ntaps = 9; a.ups = 0; a.fold = 0; a.wino_um = 1; a.wino_cus = WINO_CUS_DEC;
        if (pl.wino4_f[l]) DVG_TRY(launch_conv_wino4(a, s));  // (F(4x4,3x3): 25 of 36 positions)
        else DVG_TRY(launch_conv_wino(a, s));
      } else {
        DVG_TRY(launch_conv_igemm(a, s));
      }
    } else if (pl.tail) {  // layer 2's activation happens while layer 3 stages its input (Xs[2] is never written)
      const DecActIn in{W + pl.Y[2], W + pl.mean[2], W + pl.invstd[2], p->bn_g[2], p->bn_b[2], training ? W + pl.mask[2] : nullptr};
      DVG_TRY(launch_dec_conv3_fwd_act(in, N, p->conv_w[3], p->conv_b[3], W + pl.Y[3], W + pl.stats[3], s));
    } else {
      DVG_TRY(launch_dec_conv3_fwd(x, N, p->conv_w[3], p->conv_b[3], W + pl.Y[3], W + pl.stats[3], s));
    }
    DVG_TRY(launch_bn_finalize(W + pl.stats[l], pl.nblk[l], C, pl.M[l], training, W + pl.mean[l], W + pl.invstd[l],
                               p->bn_rm[l], p->bn_rv[l], p->bn_nbt[l], s));
    const float* mask = training ? W + pl.mask[l] : nullptr;
    if ((l == 2 || l == 3) && pl.tail) continue;  // (activated by the next layer's kernels while they stage their input)
    DVG_TRY(launch_dec_bn_act_fwd(W + pl.Y[l], pl.M[l], C, 2 * pl.L[l], W + pl.mean[l], W + pl.invstd[l], p->bn_g[l],
                                  p->bn_b[l], mask, W + pl.Xs[l], s));
    x = W + pl.Xs[l];
  }
  if (mt) {
    // final layer -> MSE -> the final layer's data gradient -> the 1-channel stage's backward sums, one pass per image
    const DecActIn in3{W + pl.Y[3], W + pl.mean[3], W + pl.invstd[3], p->bn_g[3], p->bn_b[3], W + pl.mask[3]};
    DVG_TRY(launch_dec_tail_mse_sums(in3, N, p->conv_w[4], p->conv_b[4], mt->images, mt->R, mse_gscale(*mt, N), W + pl.partT,
                                     (double*)(W + pl.msep), s));
    DVG_TRY(launch_mse_final((const double*)(W + pl.msep), dec_final_dgrad_blocks(N), 1.0 / ((double)N * 1024.0), mt->loss_out, s));
  } else if (pl.tail) {
    const DecActIn in3{W + pl.Y[3], W + pl.mean[3], W + pl.invstd[3], p->bn_g[3], p->bn_b[3], training ? W + pl.mask[3] : nullptr};
    DVG_TRY(launch_dec_final_fwd_act(in3, N, p->conv_w[4], p->conv_b[4], out, s));
  } else {
    DVG_TRY(launch_dec_final_fwd(x, N, p->conv_w[4], p->conv_b[4], out, s));
  }
  return DVG_OK;
}
}  // namespace

extern "C" int dvg_decoder_fwd_ex(const dvg_decoder_params_t* p, int n, const float* spins, int64_t N, int training,
                                  const float* const dropout_keep[4], uint64_t seed, uint64_t offset, float* out, void* ws,
                                  size_t ws_bytes, const dvg_step_state_t* dyn, int prepared, dvg_stream_t stream) {
  return decoder_fwd_impl(p, n, spins, N, training, dropout_keep, seed, offset, out, nullptr, ws, ws_bytes, dyn, prepared, stream);
}

extern "C" int dvg_decoder_fwd_mse_ex(const dvg_decoder_params_t* p, int n, const float* spins, int64_t N,
                                      const float* const dropout_keep[4], uint64_t seed, uint64_t offset, const float* images,
                                      int R, float grad_scale, float* loss_out, void* ws, size_t ws_bytes,
                                      const dvg_step_state_t* dyn, int prepared, dvg_stream_t stream) {
  const MseTail mt{images, R, grad_scale, loss_out};
  return decoder_fwd_impl(p, n, spins, N, 1, dropout_keep, seed, offset, nullptr, &mt, ws, ws_bytes, dyn, prepared, stream);
}

extern "C" int dvg_decoder_fwd(const dvg_decoder_params_t* p, int n, const float* spins, int64_t N, int training,
                               const float* const dropout_keep[4], uint64_t seed, uint64_t offset, float* out, void* ws,
                               size_t ws_bytes, const dvg_step_state_t* dyn, dvg_stream_t stream) {
  return dvg_decoder_fwd_ex(p, n, spins, N, training, dropout_keep, seed, offset, out, ws, ws_bytes, dyn, 0, stream);
}

extern "C" int dvg_decoder_bwd(const dvg_decoder_params_t* p, int n, const float* spins, int64_t N,
                               const float* grad_out, const dvg_decoder_grads_t* g, float* grad_spins, void* ws,
                               size_t ws_bytes, dvg_stream_t stream) {
  return dvg_decoder_bwd_ex(p, n, spins, N, grad_out, g, grad_spins, ws, ws_bytes, 0, stream);
}

extern "C" int dvg_stream_join_side(dvg_stream_t stream) {
  hipStream_t s = (hipStream_t)stream, s2 = side_stream(s);
  if (s2 != s) DVG_TRY(stream_order_after(s, s2));
  return DVG_OK;
}

namespace {
int decoder_bwd_impl(const dvg_decoder_params_t* p, int n, const float* spins, int64_t N, const float* grad_out,
                     const MseTail* mt, const dvg_decoder_grads_t* g, float* grad_spins, void* ws, size_t ws_bytes,
                     int defer_join, dvg_stream_t stream) {
  const DecPlan pl = dec_plan(N > 0 ? N : 1, (n >= 32 && n % 32 == 0) ? n : 32);
  DVG_TRY(check_common(p, n, N, ws, ws_bytes, pl));
  DVG_REQUIRE(conv_precision_matches_forward(ws), "decoder_bwd: the GEMM operand mode (dvg_set_conv_precision) changed since the forward call on this workspace");
  DVG_REQUIRE(plan_matches_forward(ws, plan_signature(pl) | (mt ? 1u << 30 : 0u)), "decoder_bwd: a kernel-form option (dvg_set_option: dec_fold / dec_d22 / dec_lc0 / igemm_dma) changed since the forward call on this workspace, or the forward call was not of the same kind (dvg_decoder_fwd_ex / dvg_decoder_fwd_mse_ex)");
  DVG_REQUIRE(spins && (grad_out || mt) && g, "decoder_bwd: null argument");
  if (mt) DVG_REQUIRE(pl.tail && mt->images && mt->R >= 1 && N % mt->R == 0, "decoder_bwd_mse: images / R (N=%lld R=%d), option dec_tail_fused", (long long)N, mt->R);
  DVG_REQUIRE(g->lin_w && g->lin_b, "decoder_bwd: null linear gradient buffer");
  for (int l = 0; l < 5; ++l) DVG_REQUIRE(g->conv_w[l] && g->conv_b[l], "decoder_bwd: null conv gradient buffer %d", l);
  for (int l = 0; l < 4; ++l) DVG_REQUIRE(g->bn_g[l] && g->bn_b[l], "decoder_bwd: null BN gradient buffer %d", l);
  hipStream_t s = (hipStream_t)stream;
  hipStream_t s2 = side_stream(s);  // weight-gradient chain (streams.cpp); the data-gradient chain stays on `s`
  ColsumBatch sums;  // the bias / small-weight column sums of the whole call: ONE launch at the end of the side chain
  float* W = (float*)ws;
  float* dX = W + pl.dXbuf;
  float* partA = W + pl.partA;
  float* partW = W + pl.partW;

  // final ConvTranspose2d(1,1): gradient wrt Xs[3] (16x16, quad-summed) on the caller's stream; its weight/bias
  // gradient on the side stream (main-chain kernel first: see the fork note below)
  {
    hipEvent_t start = nullptr;
    if (s2 != s) DVG_TRY(stream_mark(s, &start));
    const DecActIn in3{W + pl.Y[3], W + pl.mean[3], W + pl.invstd[3], p->bn_g[3], p->bn_b[3], W + pl.mask[3]};
    if (mt) {
      // (the sums are in partT since the forward call: dvg_decoder_fwd_mse_ex)
    } else if (pl.tail) {  // ... as the 1-channel stage's BatchNorm-backward sums (no reduce pass below; dX itself is not stored)
      DVG_TRY(launch_dec_final_dgrad_bn(grad_out, N, p->conv_w[4], in3, partA, s));
    } else {
      DVG_TRY(launch_dec_final_dgrad(grad_out, N, p->conv_w[4], dX, s));
    }
    if (s2 != s) DVG_TRY(stream_wait_mark(s2, start));
    if (mt) DVG_TRY(launch_dec_final_wgrad_mse(in3, N, p->conv_w[4], p->conv_b[4], mt->images, mt->R, mse_gscale(*mt, N), W + pl.partF, s2));
    else if (pl.tail) DVG_TRY(launch_dec_final_wgrad_act(in3, N, grad_out, W + pl.partF, s2));  // (Xs[3] does not exist)
    else DVG_TRY(launch_dec_final_wgrad(W + pl.Xs[3], N, grad_out, W + pl.partF, s2));
    DVG_REQUIRE(sums.add2(W + pl.partF, EW_BLOCKS, 10, 9, g->conv_w[4], 1, g->conv_b[4]), "decoder_bwd: column-sum batch full");
  }

  // (the fused conv3 backward leaves its two column sums to the batch below)
  for (int l = 3; l >= 0; --l) {
    const int Cin = pl.ch[l], C = pl.ch[l + 1];
    const float* Y = W + pl.Y[l];
    const float* Xs = W + pl.Xs[l];
    const float* mask = W + pl.mask[l];  // backward only exists for a training-mode forward
    float* dY = W + pl.dYl[l];
    const DecActIn in2{W + pl.Y[2], W + pl.mean[2], W + pl.invstd[2], p->bn_g[2], p->bn_b[2], W + pl.mask[2]};
    if (l == 2 && pl.tail) {
      // the (sum dz, sum dz zhat) partials came out of layer 3's fused backward below; dY2 = its second pass
      DVG_TRY(launch_colsum2(partA, dec_tail_blocks(N), 2 * C, C, g->bn_b[l], C, g->bn_g[l], s));
      DVG_TRY(launch_dec_conv3_bwd_apply(in2, N, W + pl.dYl[3], p->conv_w[3], g->bn_b[l], g->bn_g[l], dY, W + pl.partB[l], s));
    } else if (l == 3 && pl.tail) {
      // the 1-channel stage: its sums came out of the final layer's data-gradient pass above; dY3 = that pass again
      const DecActIn in3{W + pl.Y[3], W + pl.mean[3], W + pl.invstd[3], p->bn_g[3], p->bn_b[3], W + pl.mask[3]};
      DVG_TRY(launch_colsum2(mt ? W + pl.partT : partA, dec_final_dgrad_blocks(N), 2 * C, C, g->bn_b[l], C, g->bn_g[l], s));
      if (mt)
        DVG_TRY(launch_dec_tail_mse_apply(in3, N, p->conv_w[4], p->conv_b[4], mt->images, mt->R, mse_gscale(*mt, N), g->bn_b[l],
                                          g->bn_g[l], dY, W + pl.partB[l], s));
      else
        DVG_TRY(launch_dec_final_dgrad_apply(grad_out, N, p->conv_w[4], in3, g->bn_b[l], g->bn_g[l], dY, W + pl.partB[l], s));
    } else {
      DVG_TRY(launch_dec_bn_act_bwd_reduce(Y, Xs, pl.M[l], C, 2 * pl.L[l], W + pl.mean[l], W + pl.invstd[l], p->bn_g[l],
                                           p->bn_b[l], mask, dX, partA, s));
      DVG_TRY(launch_colsum2(partA, EW_BLOCKS, 2 * C, C, g->bn_b[l], C, g->bn_g[l], s));
      DVG_TRY(launch_dec_bn_act_bwd_apply(Y, Xs, pl.M[l], C, 2 * pl.L[l], W + pl.mean[l], W + pl.invstd[l], p->bn_g[l],
                                          p->bn_b[l], mask, dX, g->bn_b[l], g->bn_g[l], dY, W + pl.partB[l], s));
    }
    // Fork: dY is ready.  The caller's-stream kernel is enqueued BEFORE the side-stream ones: when the call is being
    // captured into a hipGraph, the first child captured after a fork inherits the parent's hardware queue, and a
    // data-gradient chain that changes queue at every layer pays a cross-queue signal (~10-15 us) per hop.
    hipEvent_t dy_ready = nullptr;
    if (s2 != s) DVG_TRY(stream_mark(s, &dy_ready));
    const float* xin = l == 0 ? W + pl.X0 : W + pl.Xs[l - 1];
    if (l == 0 && pl.lc0) {
      // Composed Linear + layer 0: one data-gradient GEMM straight to the spins, one weight-gradient GEMM for dWc, and the
      // gradients of the two original weights / biases by the chain rule in weight space (side stream).
      const int n4 = 4 * n, C4 = 4 * C;
      if (grad_spins) {
        ConvArgs a;
        a.in = dY; a.wp = W + pl.Wc; a.bias = nullptr; a.out = grad_spins; a.stats = nullptr;
        a.M = N; a.Cin = C4; a.Cout = n; a.L = 0; a.ntaps = 1; a.ups = 0; a.poolsum = 0;
        a.splitk_ws = W + pl.splitk;
        DVG_TRY(launch_conv_igemm(a, s));
      }
      if (s2 != s) DVG_TRY(stream_wait_mark(s2, dy_ready));
      DVG_REQUIRE(sums.add(W + pl.partB[l], EW_BLOCKS, C, C, 1.0f, g->conv_b[l], 0, 0), "decoder_bwd: column-sum batch full");
      // dbc[o] = column sums of dY seen as [N][4 C]
      DVG_TRY(launch_rowsum_partial(dY, N, C4, W + pl.partC, s2));
      DVG_TRY(launch_colsum(W + pl.partC, EW_BLOCKS, C4, C4, 1.0f, W + pl.dbc, 0, 0, s2));
      DVG_TRY(launch_lc0_lin_bias_grad(W + pl.dbc, W + pl.wpd[0], n, C, g->lin_b, s2));
      // dWc[i][o] = sum_images spins[i] dY[o]  (and its transpose)
      WgradArgs wa;
      wa.in = spins; wa.dy = dY; wa.slabs = W + pl.slabs;
      wa.M = N; wa.Cin = n; wa.Cout = C4; wa.L = 0; wa.ntaps = 1; wa.ups = 0; wa.ksplit = pl.ksplit_c;
      DVG_TRY(launch_conv_wgrad(wa, s2));
      DVG_TRY(launch_slab_sum(W + pl.slabs, pl.ksplit_c, n, C4, W + pl.dWc, W + pl.dWcT, s2));
      ConvArgs b;
      b.bias = nullptr; b.stats = nullptr; b.L = 0; b.ntaps = 1; b.ups = 0; b.poolsum = 0; b.splitk_ws = nullptr;
      b.force_f32 = 1;  // weight-space products (see the forward call)
      // T1[j][i] = sum_o Weff[j][o] dWc[i][o]: rows of the dense-2x2 data-gradient pack [j][o] x dWc as the K-major operand
      b.in = W + pl.wpd[0]; b.wp = W + pl.dWc; b.out = W + pl.T1; b.M = n4; b.Cin = C4; b.Cout = n;
      DVG_TRY(launch_conv_igemm(b, s2));
      DVG_TRY(launch_lc0_rows_to_linear(W + pl.T1, n, g->lin_w, s2));
      // dWeff[j][o] = sum_i Wlin[i][j] dWc[i][o]: rows of the Linear forward pack [j][i] x dWc^T [o][i]
      b.in = W + pl.wp_lin; b.wp = W + pl.dWcT; b.out = W + pl.dWeff; b.M = n4; b.Cin = n; b.Cout = C4;
      DVG_TRY(launch_conv_igemm(b, s2));
      DVG_TRY(launch_lc0_rank1_add(W + pl.dWeff, p->lin_b, W + pl.dbc, n, C, s2));  // (+ the Linear bias' share of X0)
      DVG_TRY(launch_wgrad_d22_reduce(W + pl.dWeff, 1, Cin, C, g->conv_w[l], s2));
      continue;
    }
    if (l == 3) {
      // data gradient and weight-gradient partials in ONE pass over the images on the caller's stream (special.hip); only
      // the column sums go to the side stream (no fork here: they are in the batch at the end of the side chain).  (The two
      // separate kernels this replaced were retired in round 3.)
      const int w3_blocks = stream_blocks(N);  // partial rows of the weight gradient
      if (pl.tail) DVG_TRY(launch_dec_conv3_bwd_reduce(in2, N, dY, p->conv_w[3], partW, partA, s));
      else DVG_TRY(launch_dec_conv3_bwd(xin, N, dY, p->conv_w[3], dX, partW, s));
      DVG_REQUIRE(sums.add(W + pl.partB[l], pl.tail ? dec_final_dgrad_blocks(N) : EW_BLOCKS, C, C, 1.0f, g->conv_b[l], 0, 0) &&
                  sums.add(partW, w3_blocks, 288, 288, 1.0f, g->conv_w[3], 32, 9),  // [tap][ci] -> [ci][tap]
                  "decoder_bwd: column-sum batch full");
      continue;
    }
    ConvArgs a;
    a.in = dY; a.wp = W + pl.wpd[l]; a.bias = nullptr; a.out = dX; a.stats = nullptr;
    a.M = pl.M[l]; a.Cin = C; a.Cout = Cin; a.L = pl.L[l]; a.ntaps = 9; a.ups = 0; a.poolsum = l > 0;
    if (pl.fold[l]) { a.M = pl.M[l] / 4; a.L = pl.L[l] - 1; a.ntaps = 16; a.poolsum = 0; a.fold = 2; }
    const bool d22 = l == 0 && pl.d22;
    if (d22) { a.M = N; a.Cin = 4 * C; a.Cout = 4 * Cin; a.L = 0; a.ntaps = 1; a.poolsum = 0; }
    a.splitk_ws = W + pl.splitk;
    if (pl.wino_d[l]) {  // fine-grid Winograd data gradient with the 2x2 sum folded into its output transform (UM = 2)
      a.M = pl.M[l]; a.L = pl.L[l]; a.ntaps = 9; a.poolsum = 0; a.fold = 0; a.wino_um = 2; a.wino_cus = WINO_CUS_DEC;
      if (pl.wino4_d[l]) DVG_TRY(launch_conv_wino4(a, s));  // (F(4x4,3x3): 25 of 36 positions)
      else DVG_TRY(launch_conv_wino(a, s));
    } else {
      DVG_TRY(launch_conv_igemm(a, s));
    }
    if (s2 != s) DVG_TRY(stream_wait_mark(s2, dy_ready));
    DVG_REQUIRE(sums.add(W + pl.partB[l], (l == 2 && pl.tail) ? dec_tail_blocks(N) : EW_BLOCKS, C, C, 1.0f, g->conv_b[l], 0, 0),
                "decoder_bwd: column-sum batch full");
    WgradArgs wa;
    wa.in = xin; wa.dy = dY; wa.slabs = W + pl.slabs;
    wa.M = pl.M[l]; wa.Cin = Cin; wa.Cout = C; wa.L = pl.L[l]; wa.ntaps = 9; wa.ups = l > 0; wa.ksplit = pl.ksplit[l];
    if (d22) {
      wa.M = N; wa.Cin = 4 * Cin; wa.Cout = 4 * C; wa.L = 0; wa.ntaps = 1; wa.ups = 0;
      DVG_TRY(launch_conv_wgrad(wa, s2));
      DVG_TRY(launch_wgrad_d22_reduce(W + pl.slabs, pl.ksplit[l], Cin, C, g->conv_w[l], s2));
    } else if (pl.wino_w[l]) {
      DVG_TRY(launch_conv_wino_wgrad(xin, dY, pl.M[l], Cin, C, pl.L[l], W + pl.slabs, WeightMap{WM_CONVT_FWD, Cin, C, 9},
                                     g->conv_w[l], s2, 1, WINO_CUS_DEC));
    } else if (pl.fold[l]) {
      wa.M = pl.M[l] / 4; wa.L = pl.L[l] - 1; wa.ntaps = 16; wa.ups = 0; wa.fold = 1;
      DVG_TRY(launch_conv_wgrad(wa, s2));
      DVG_TRY(launch_wgrad_fold_reduce(W + pl.slabs, pl.ksplit[l], Cin, C, g->conv_w[l], s2));
    } else {
      DVG_TRY(launch_conv_wgrad(wa, s2));
      DVG_TRY(launch_wgrad_reduce(W + pl.slabs, pl.ksplit[l], WeightMap{WM_CONVT_FWD, Cin, C, 9}, g->conv_w[l], s2));
    }
  }
  // dX now holds the gradient wrt X0 (N, 4n) in (p, c) order.  Linear backward (composed form: done with layer 0 above):
  if (pl.lc0) {
    DVG_TRY(launch_colsum_batch(sums, s2));
    if (!defer_join) DVG_TRY(stream_order_after(s, s2));
  } else {
    hipEvent_t dx_ready = nullptr;  // dX is final
    if (s2 != s) DVG_TRY(stream_mark(s, &dx_ready));
    if (grad_spins) {
      ConvArgs a;
      a.in = dX; a.wp = W + pl.wpd_lin; a.bias = nullptr; a.out = grad_spins; a.stats = nullptr;
      a.M = N; a.Cin = 4 * n; a.Cout = n; a.L = 0; a.ntaps = 1; a.ups = 0; a.poolsum = 0;
      a.splitk_ws = W + pl.splitk;
      DVG_TRY(launch_conv_igemm(a, s));
    }
    if (s2 != s) DVG_TRY(stream_wait_mark(s2, dx_ready));
    DVG_TRY(launch_rowsum_partial(dX, N, 4 * n, W + pl.partL, s2));
    DVG_REQUIRE(sums.add(W + pl.partL, EW_BLOCKS, 4 * n, 4 * n, 1.0f, g->lin_b, n, 4), "decoder_bwd: column-sum batch full");  // j' = p*n+c -> c*4+p
    WgradArgs wa;
    wa.in = spins; wa.dy = dX; wa.slabs = W + pl.slabs;
    wa.M = N; wa.Cin = n; wa.Cout = 4 * n; wa.L = 0; wa.ntaps = 1; wa.ups = 0; wa.ksplit = pl.ksplit_lin;
    DVG_TRY(launch_conv_wgrad(wa, s2));
    DVG_TRY(launch_wgrad_reduce(W + pl.slabs, pl.ksplit_lin, WeightMap{WM_LIN_FWD, n, 4 * n, 1}, g->lin_w, s2));
    DVG_TRY(launch_colsum_batch(sums, s2));
    if (!defer_join) DVG_TRY(stream_order_after(s, s2));  // join (deferred: dvg_stream_join_side / the next backward call)
  }
  return DVG_OK;
}
}  // namespace

extern "C" int dvg_decoder_bwd_ex(const dvg_decoder_params_t* p, int n, const float* spins, int64_t N,
                                  const float* grad_out, const dvg_decoder_grads_t* g, float* grad_spins, void* ws,
                                  size_t ws_bytes, int defer_join, dvg_stream_t stream) {
  return decoder_bwd_impl(p, n, spins, N, grad_out, nullptr, g, grad_spins, ws, ws_bytes, defer_join, stream);
}

extern "C" int dvg_decoder_bwd_mse_ex(const dvg_decoder_params_t* p, int n, const float* spins, int64_t N,
                                      const float* images, int R, float grad_scale, const dvg_decoder_grads_t* g,
                                      float* grad_spins, void* ws, size_t ws_bytes, int defer_join, dvg_stream_t stream) {
  const MseTail mt{images, R, grad_scale, nullptr};
  return decoder_bwd_impl(p, n, spins, N, nullptr, &mt, g, grad_spins, ws, ws_bytes, defer_join, stream);
}
