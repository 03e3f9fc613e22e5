// dvg_encoder_fwd / dvg_encoder_bwd: Encoder.forward of /root/reference/src/encoder.py:44-49 and its
// backward, as a fixed sequence of kernels on the caller's stream.
//   layer 0 (1->32 @32x32): VALU conv;  layers 1-3 (32->64 @16, 64->128 @8, 128->n @4): MFMA implicit GEMM
//   each followed by BN(batch stats) -> MaxPool2d(2) -> LeakyReLU (none after the last); then Linear(4,1).
#include "conv.h"
#include "kernels.h"
#include "../../include/dvg_dev.h"

using namespace dvg;

namespace {

struct EncPlan {
  int64_t B;
  int n;
  int ch[5];
  int64_t M[4], Q[4];
  int L[4];
  int nblk[4];
  // offsets in floats
  size_t Y[4], Xp[4], mean[4], invstd[4], stats[4], wp[4], wpd[4];
  bool wino_f[4], wino_d[4];  // layer's forward / data-gradient GEMM runs in the Winograd form (conv_wino.hip)
  bool wino_w[4];             // ... its weight gradient too (conv_wino_wgrad.hip)
  bool wino4_f[4], wino4_d[4], wino4_w[4]; // ... in the F(4x4,3x3) form (conv_wino4.hip, conv_wino4_wgrad.hip) instead of F(2x2,3x3)
  size_t dXbuf, dYl[4], slabs, partA, partB[4], partP, part320, mom_part, mom, pimg, splitk;
  int ksplit[4];
  size_t total_floats;
};

size_t bump(size_t& o, size_t count) {
  const size_t r = o;
  o += (count + 63) & ~(size_t)63;  // 256-byte granules
  return r;
}

// `training` = 0 plans an evaluation-mode forward call (no backward follows; nothing else runs beside it): by default
// only those calls use the Winograd form (conv_wino_ok)
EncPlan enc_plan(int64_t B, int n, int training = 1) {
  EncPlan p;
  p.B = B; p.n = n;
  const int ch[5] = {1, 32, 64, 128, n};
  for (int i = 0; i < 5; ++i) p.ch[i] = ch[i];
  size_t o = 0;
  size_t max_dx = 0, max_slab = 0, max_split = 0;
  int cmax = 0;
  for (int l = 0; l < 4; ++l) {
    p.L[l] = 5 - l;
    p.M[l] = B * (1024 >> (2 * l));
    p.Q[l] = p.M[l] / 4;
    const int C = ch[l + 1];
    if (C > cmax) cmax = C;
    p.wino_f[l] = l > 0 && conv_wino_ok(p.M[l], ch[l], C, p.L[l], training ? 0 : 2);
    p.wino_d[l] = l > 0 && training && conv_wino_ok(p.M[l], C, ch[l], p.L[l], 1);
    p.wino_w[l] = l > 0 && training && conv_wino_wgrad_ok(p.M[l], ch[l], C, p.L[l]);
    const int64_t m4 = opt(OPT_ENC_WINO4_MASK);  // (which launches take the F(4x4,3x3) form: measured per launch inside the step)
    p.wino4_f[l] = l > 0 && p.wino_f[l] && ((m4 >> (l - 1)) & 1) && conv_wino4_ok(p.M[l], ch[l], C, p.L[l]);
    p.wino4_d[l] = l > 0 && p.wino_d[l] && ((m4 >> (2 + l)) & 1) && conv_wino4_ok(p.M[l], C, ch[l], p.L[l]);
    p.wino4_w[l] = l > 0 && p.wino_w[l] && ((m4 >> (5 + l)) & 1) && conv_wino4_wgrad_ok(p.M[l], ch[l], C, p.L[l]);
    p.nblk[l] = l == 0 ? enc_conv0_blocks(B)
                       : (p.wino4_f[l] ? conv_wino4_stats_blocks(p.M[l]) : (p.wino_f[l] ? conv_wino_stats_blocks(p.M[l], C) : conv_stats_blocks(p.M[l], C)));
    p.Y[l] = bump(o, (size_t)p.M[l] * C);
    p.Xp[l] = bump(o, (size_t)p.Q[l] * C);
    p.mean[l] = bump(o, C);
    p.invstd[l] = bump(o, C);
    p.stats[l] = bump(o, (size_t)(p.nblk[l] + BN_FOLD_ROWS) * C * 2);  // + scratch rows of launch_bn_finalize
    p.wp[l] = p.wpd[l] = 0;
    p.ksplit[l] = 0;
    if (l > 0) {
      p.wp[l] = bump(o, conv_pack_floats((size_t)(p.wino4_f[l] ? 36 : (p.wino_f[l] ? 16 : 9)) * ch[l] * C));
      p.wpd[l] = bump(o, conv_pack_floats((size_t)(p.wino4_d[l] ? 36 : (p.wino_d[l] ? 16 : 9)) * ch[l] * C));
      p.ksplit[l] = wgrad_ksplit(p.M[l], ch[l], C, 9);
      const size_t slab = (size_t)p.ksplit[l] * 9 * ch[l] * C;
      if (slab > max_slab) max_slab = slab;
      if (p.wino_w[l]) {
        const size_t sw = p.wino4_w[l] ? conv_wino4_wgrad_slab_floats(p.M[l], ch[l], C) : conv_wino_wgrad_slab_floats(p.M[l], ch[l], C, p.L[l]);
        if (sw > max_slab) max_slab = sw;
      }
      const size_t sk_f = conv_splitk_floats(p.M[l], ch[l], C, 9, 0), sk_d = conv_splitk_floats(p.M[l], C, ch[l], 9, 0);
      if (sk_f > max_split) max_split = sk_f;
      if (sk_d > max_split) max_split = sk_d;
      const size_t dx = (size_t)p.M[l] * ch[l];  // gradient wrt the layer's input (same resolution)
      if (dx > max_dx) max_dx = dx;
    }
  }
  const size_t dp = (size_t)p.Q[3] * n;  // gradient wrt the pooled map feeding the projection
  if (dp > max_dx) max_dx = dp;
  p.dXbuf = bump(o, max_dx);
  // one dY buffer per layer: a layer's weight gradient (side stream) may still be reading its dY when the data-gradient
  // chain reaches the next layers, and a shared buffer would make the caller's stream wait on the side stream
  for (int l = 0; l < 4; ++l) p.dYl[l] = bump(o, (size_t)p.M[l] * ch[l + 1]);
  p.slabs = bump(o, max_slab);
  p.partA = bump(o, (size_t)EW_BLOCKS * 2 * cmax);
  for (int l = 0; l < 4; ++l) p.partB[l] = bump(o, (size_t)EW_BLOCKS * ch[l + 1]);  // per layer: reduced on the side stream
  p.partP = bump(o, (size_t)EW_BLOCKS * 8);
  p.part320 = bump(o, (size_t)STREAM_BLOCKS * ENC_L0_ROW_FLOATS);  // (rows of the one-pass layer-0 backward)
  // layer 0's patch moments (doubles: two floats each; 256-byte granules keep them aligned): per-block rows, then the sums
  p.mom_part = bump(o, (size_t)enc_l0_moment_blocks(B) * ENC_L0_MOM_ROW * 2);
  p.mom = bump(o, (size_t)ENC_L0_MOM_ROW * 2);
  p.pimg = bump(o, (size_t)B * ENC_L0_PIMG);  // the zero-padded images the passes that recompute layer 0 read
  p.splitk = bump(o, max_split);
  p.total_floats = o;
  return p;
}

// what shapes the workspace's contents: the pack format of the direct GEMMs and which layers run in the Winograd form
// (bit 31: the forward ran in training mode -- the only kind a backward call may follow)
uint32_t enc_plan_signature(const EncPlan& pl, int training) {
  uint32_t sig = (uint32_t)conv_launch_mode(pl.B, 64) | (opt(OPT_ENC_L0_FUSED) != 0 ? 1u << 30 : 0u) | (training ? 1u << 31 : 0u);
  for (int l = 1; l < 4; ++l) sig |= (pl.wino_f[l] ? 1u : 0u) << (8 + 2 * l) | (pl.wino_d[l] ? 1u : 0u) << (9 + 2 * l);
  for (int l = 1; l < 4; ++l) sig |= (pl.wino4_f[l] ? 1u : 0u) << (16 + 2 * l) | (pl.wino4_d[l] ? 1u : 0u) << (17 + 2 * l);
  return sig;
}

int check_common(const dvg_encoder_params_t* p, int n, int64_t B, const void* ws, size_t ws_bytes, const EncPlan& pl) {
  DVG_REQUIRE(p && ws, "encoder: null params/workspace");
  DVG_REQUIRE(n >= 32 && n % 32 == 0 && n <= 4096, "encoder: n_latents=%d must be a multiple of 32", n);
  DVG_REQUIRE(B >= 1 && B <= (1 << 20), "encoder: batch %lld out of range", (long long)B);
  for (int l = 0; l < 4; ++l)
    DVG_REQUIRE(p->conv_w[l] && p->conv_b[l] && p->bn_g[l] && p->bn_b[l] && p->bn_rm[l] && p->bn_rv[l],
                "encoder: null parameter in layer %d", l);
  DVG_REQUIRE(p->proj_w && p->proj_b, "encoder: null projection parameter");
  if (ws_bytes < pl.total_floats * sizeof(float)) {
    set_error("encoder: workspace %zu < %zu bytes", ws_bytes, pl.total_floats * sizeof(float));
    return DVG_E_WORKSPACE;
  }
  return DVG_OK;
}

}  // namespace

extern "C" int dvg_dev_encoder_layout(int64_t B, int n_latents, size_t out[16]) {
  DVG_REQUIRE(out && B >= 1 && n_latents >= 32 && n_latents % 32 == 0, "dev_encoder_layout: bad argument");
  const EncPlan pl = enc_plan(B, n_latents);
  for (int l = 0; l < 4; ++l) { out[l] = pl.Y[l]; out[4 + l] = pl.Xp[l]; out[8 + l] = pl.mean[l]; out[12 + l] = pl.invstd[l]; }
  return DVG_OK;
}

extern "C" size_t dvg_encoder_workspace_bytes(int64_t B, int n_latents) {
  dvg::side_stream_warm();  // the backward's fork/join context exists before any step is captured
  if (B < 1 || n_latents < 32 || n_latents % 32) return 0;
  const size_t t1 = enc_plan(B, n_latents, 1).total_floats, t0 = enc_plan(B, n_latents, 0).total_floats;
  return (t1 > t0 ? t1 : t0) * sizeof(float);
}

extern "C" int dvg_encoder_fwd(const dvg_encoder_params_t* p, int n, const float* images, int64_t B, int training,
                               float* logits, void* ws, size_t ws_bytes, dvg_stream_t stream) {
  const EncPlan pl = enc_plan(B > 0 ? B : 1, (n >= 32 && n % 32 == 0) ? n : 32, training ? 1 : 0);
  DVG_TRY(check_common(p, n, B, ws, ws_bytes, pl));
  conv_precision_note_forward(ws);
  plan_note_forward(ws, enc_plan_signature(pl, training));  // (the pack formats the backward will read)
  DVG_REQUIRE(images && logits, "encoder_fwd: null images/logits");
  hipStream_t s = (hipStream_t)stream;
  float* W = (float*)ws;
  const float* x = images;
  {  // all weight packs of the network (forward AND data-gradient layouts) in one launch
    PackJob jobs[6];
    int nj = 0;
    for (int l = 1; l < 4; ++l) {
      const WeightMap mf{WM_CONV_FWD, pl.ch[l], pl.ch[l + 1], 9}, md{WM_CONV_DGRAD, pl.ch[l + 1], pl.ch[l], 9};
      jobs[nj++] = PackJob{p->conv_w[l], W + pl.wp[l], mf, 0, pl.M[l], pl.wino4_f[l] ? 2 : (pl.wino_f[l] ? 1 : 0)};  // (Winograd launches read U = G g G^T)
      jobs[nj++] = PackJob{p->conv_w[l], W + pl.wpd[l], md, 0, pl.M[l], pl.wino4_d[l] ? 2 : (pl.wino_d[l] ? 1 : 0)};
    }
    DVG_TRY(launch_weight_pack_multi(jobs, nj, s));
  }
  for (int l = 0; l < 4; ++l) {
    const int Cin = pl.ch[l], C = pl.ch[l + 1];
    if (l == 0 && opt(OPT_ENC_L0_FUSED) != 0) {
      // layer 0 recomputed (special.hip): BatchNorm statistics from the moments of the input patches, then BN -> pool ->
      // LeakyReLU from the images (enc_l0_kernel<1>); Y0 is never written
      EncL0Args a0{};
      a0.img = images; a0.B = B; a0.w = p->conv_w[0]; a0.bias = p->conv_b[0];
      a0.mean = W + pl.mean[0]; a0.invstd = W + pl.invstd[0]; a0.gamma = p->bn_g[0]; a0.beta = p->bn_b[0];
      a0.Xp = W + pl.Xp[0];
      a0.pimg = training ? W + pl.pimg : nullptr;
      if (training)
        DVG_TRY(launch_enc_l0_moments(images, B, p->conv_w[0], p->conv_b[0], (double*)(W + pl.mom_part), (double*)(W + pl.mom),
                                      W + pl.pimg, W + pl.mean[0], W + pl.invstd[0], p->bn_rm[0], p->bn_rv[0], p->bn_nbt[0], s));
      else
        DVG_TRY(launch_bn_finalize(W + pl.stats[0], pl.nblk[0], C, pl.M[0], 0, W + pl.mean[0], W + pl.invstd[0],
                                   p->bn_rm[0], p->bn_rv[0], p->bn_nbt[0], s));
      DVG_TRY(launch_enc_l0(1, a0, s));
      x = W + pl.Xp[0];
      continue;
    }
    if (l == 0) {
      DVG_TRY(launch_enc_conv0_fwd(images, B, p->conv_w[0], p->conv_b[0], W + pl.Y[0], W + pl.stats[0], s));
    } else {
      ConvArgs a;
      a.in = x; a.wp = W + pl.wp[l]; a.bias = p->conv_b[l]; a.out = W + pl.Y[l];
      a.stats = training ? W + pl.stats[l] : nullptr;
      a.M = pl.M[l]; a.Cin = Cin; a.Cout = C; a.L = pl.L[l]; a.ntaps = 9; a.ups = 0; a.poolsum = 0;
      a.splitk_ws = W + pl.splitk;
      // (a training call's forward runs beside the step's sampler draw, which holds its CUs: the tile blocks are dealt
      // dynamically, so the grid is sized to the chip and the workgroups that get their CU late, when the draw ends,
      // take what is left; under option wino_dynamic = 0 -- the static deal, A/B -- a quarter of the chip is left out)
      a.wino_cus = (training && opt(OPT_WINO_DYNAMIC) == 0) ? 192 : 0;
      if (pl.wino4_f[l]) DVG_TRY(launch_conv_wino4(a, s));
      else if (pl.wino_f[l]) DVG_TRY(launch_conv_wino(a, s));
      else DVG_TRY(launch_conv_igemm(a, s));
    }
    DVG_TRY(launch_bn_finalize(W + pl.stats[l], pl.nblk[l], C, pl.M[l], training, W + pl.mean[l], W + pl.invstd[l],
                               p->bn_rm[l], p->bn_rv[l], p->bn_nbt[l], s));
    DVG_TRY(launch_enc_bn_pool_fwd(W + pl.Y[l], pl.Q[l], C, W + pl.mean[l], W + pl.invstd[l], p->bn_g[l], p->bn_b[l],
                                   l < 3, W + pl.Xp[l], s));
    x = W + pl.Xp[l];
  }
  DVG_TRY(launch_enc_proj_fwd(W + pl.Xp[3], B, n, p->proj_w, p->proj_b, logits, s));
  return DVG_OK;
}

extern "C" int dvg_encoder_bwd(const dvg_encoder_params_t* p, int n, const float* images, int64_t B,
                               const float* grad_logits, const dvg_encoder_grads_t* g, void* ws, size_t ws_bytes,
                               dvg_stream_t stream) {
  const EncPlan pl = enc_plan(B > 0 ? B : 1, (n >= 32 && n % 32 == 0) ? n : 32);
  DVG_TRY(check_common(p, n, B, ws, ws_bytes, pl));
  DVG_REQUIRE(conv_precision_matches_forward(ws), "encoder_bwd: the GEMM operand mode (dvg_set_conv_precision) changed since the forward call on this workspace");
  DVG_REQUIRE(plan_forward_flag(ws, 1u << 31), "encoder_bwd: backward requires a training-mode forward on this workspace (the last forward call here ran in evaluation mode: running statistics, no saved batch statistics)");
  DVG_REQUIRE(plan_matches_forward(ws, enc_plan_signature(pl, 1)), "encoder_bwd: a kernel-form option (dvg_set_option: igemm_dma / enc_wino / enc_wino4 / enc_l0_fused) changed since the forward call on this workspace");
  DVG_REQUIRE(images && grad_logits && g, "encoder_bwd: null argument");
  for (int l = 0; l < 4; ++l)
    DVG_REQUIRE(g->conv_w[l] && g->conv_b[l] && g->bn_g[l] && g->bn_b[l], "encoder_bwd: null gradient buffer, layer %d", l);
  DVG_REQUIRE(g->proj_w && g->proj_b, "encoder_bwd: null projection gradient buffer");
  hipStream_t s = (hipStream_t)stream;
  hipStream_t s2 = side_stream(s);  // weight-gradient chain (streams.cpp); the data-gradient chain stays on `s`
  ColsumBatch sums;  // the bias / projection column sums of the whole call: ONE launch at the end of the side chain
  float* W = (float*)ws;
  float* dX = W + pl.dXbuf;
  float* partA = W + pl.partA;

  // projection: dP (B,4,n), d proj_w (4), d proj_b (1)
  DVG_TRY(launch_enc_proj_bwd(W + pl.Xp[3], B, n, p->proj_w, grad_logits, dX, W + pl.partP, s));
  bool proj_pending = true;  // its column sums go to the side stream at the first fork

  for (int l = 3; l >= 0; --l) {
    const int Cin = pl.ch[l], C = pl.ch[l + 1];
    const float* Y = W + pl.Y[l];
    float* dY = W + pl.dYl[l];
    if (l == 0 && opt(OPT_ENC_L0_FUSED) != 0) {
      // layer 0 recomputed: (sum dz, sum dz zhat) from the images and the pooled gradient, then the weight gradient with dY0
      // formed in registers -- neither Y0 nor dY0 exists
      EncL0Args a0{};
      a0.img = images; a0.B = B; a0.w = p->conv_w[0]; a0.bias = p->conv_b[0];
      a0.mean = W + pl.mean[0]; a0.invstd = W + pl.invstd[0]; a0.gamma = p->bn_g[0]; a0.beta = p->bn_b[0];
      a0.dXp = dX;
      a0.pimg = W + pl.pimg;  // (written by the forward call's moments pass: the backward follows a training-mode forward)
      // one pass: S and sum dz zhat per block, their column sums, then every gradient of the stage from the sums and the
      // patch moments the forward call left in the workspace
      a0.part = W + pl.part320;
      DVG_TRY(launch_enc_l0(4, a0, s));
      DVG_TRY(launch_colsum(W + pl.part320, enc_l0_blocks(B), ENC_L0_ROW_FLOATS, ENC_L0_ROW_FLOATS, 1.0f, partA, 0, 0, s));
      DVG_TRY(launch_enc_l0_combine(partA, (const double*)(W + pl.mom), p->conv_w[0], p->conv_b[0], W + pl.mean[0], p->bn_g[0],
                                    W + pl.invstd[0], B, g->conv_w[0], g->conv_b[0], g->bn_b[0], g->bn_g[0], s));
      break;
    }
    // BN + pool + lrelu backward: (sum dz -> d beta, sum dz*zhat -> d gamma), then dY
    DVG_TRY(launch_enc_bn_pool_bwd_reduce(Y, pl.Q[l], C, W + pl.mean[l], W + pl.invstd[l], p->bn_g[l], p->bn_b[l], l < 3,
                                          dX, partA, s, opt(OPT_ENC_BN_REDUCE_POOLED) != 0 ? W + pl.Xp[l] : nullptr));
    DVG_TRY(launch_colsum2(partA, EW_BLOCKS, 2 * C, C, g->bn_b[l], C, g->bn_g[l], s));
    DVG_TRY(launch_enc_bn_pool_bwd_apply(Y, pl.Q[l], C, W + pl.mean[l], W + pl.invstd[l], p->bn_g[l], p->bn_b[l], l < 3,
                                         dX, g->bn_b[l], g->bn_g[l], dY, W + pl.partB[l], s));
    if (l == 0) {
      DVG_TRY(launch_enc_conv0_wgrad(images, B, dY, W + pl.part320, s));
      DVG_TRY(launch_colsum2(W + pl.part320, stream_blocks(8 * B), 320, 288, g->conv_w[0], 32, g->conv_b[0], s));
      break;
    }
    // fork: dY is ready; the caller's-stream kernel goes first (see decoder.cpp: queue inheritance under capture)
    hipEvent_t dy_ready = nullptr;
    if (s2 != s) DVG_TRY(stream_mark(s, &dy_ready));
    // data gradient -> dX (gradient wrt this layer's input = previous stage's output)
    // (the data-gradient weight layout was packed by the forward call: same weights)
    ConvArgs a;
    a.in = dY; a.wp = W + pl.wpd[l]; a.bias = nullptr; a.out = dX; a.stats = nullptr;
    a.M = pl.M[l]; a.Cin = C; a.Cout = Cin; a.L = pl.L[l]; a.ntaps = 9; a.ups = 0; a.poolsum = 0;
    a.splitk_ws = W + pl.splitk;
    a.wino_cus = opt(OPT_ENC_DGRAD_CUS) > 0 ? (int)opt(OPT_ENC_DGRAD_CUS) : WINO_CUS_ENC_DGRAD;  // (the layer's weight-gradient chain runs beside it on the side stream)
    if (pl.wino4_d[l]) DVG_TRY(launch_conv_wino4(a, s));
    else if (pl.wino_d[l]) DVG_TRY(launch_conv_wino(a, s));
    else DVG_TRY(launch_conv_igemm(a, s));
    if (s2 != s) DVG_TRY(stream_wait_mark(s2, dy_ready));
    if (proj_pending) {
      DVG_REQUIRE(sums.add2(W + pl.partP, EW_BLOCKS, 5, 4, g->proj_w, 1, g->proj_b), "encoder_bwd: column-sum batch full");
      proj_pending = false;
    }
    DVG_REQUIRE(sums.add(W + pl.partB[l], EW_BLOCKS, C, C, 1.0f, g->conv_b[l], 0, 0), "encoder_bwd: column-sum batch full");
    // weight gradient
    WgradArgs wa;
    wa.in = W + pl.Xp[l - 1]; wa.dy = dY; wa.slabs = W + pl.slabs;
    wa.M = pl.M[l]; wa.Cin = Cin; wa.Cout = C; wa.L = pl.L[l]; wa.ntaps = 9; wa.ups = 0; wa.ksplit = pl.ksplit[l];
    if (pl.wino4_w[l]) {
      DVG_TRY(launch_conv_wino4_wgrad(W + pl.Xp[l - 1], dY, pl.M[l], Cin, C, pl.L[l], W + pl.slabs, WeightMap{WM_CONV_FWD, Cin, C, 9},
                                      g->conv_w[l], s2, (int)opt(OPT_ENC_WGRAD_CUS)));
    } else if (pl.wino_w[l]) {
      DVG_TRY(launch_conv_wino_wgrad(W + pl.Xp[l - 1], dY, pl.M[l], Cin, C, pl.L[l], W + pl.slabs, WeightMap{WM_CONV_FWD, Cin, C, 9},
                                     g->conv_w[l], s2, 0, (int)opt(OPT_ENC_WGRAD_CUS)));
    } else {
      DVG_TRY(launch_conv_wgrad(wa, s2));
      DVG_TRY(launch_wgrad_reduce(W + pl.slabs, pl.ksplit[l], WeightMap{WM_CONV_FWD, Cin, C, 9}, g->conv_w[l], s2));
    }
  }
  DVG_TRY(launch_colsum_batch(sums, s2));
  DVG_TRY(stream_order_after(s, s2));  // join
  return DVG_OK;
}
