// Small fused kernels: latent->discrete (Gumbel-softmax / heaviside), MSE(+grad), Adam.
#include "common.h"
#include "philox.h"

namespace dvg {

// ---------------------------------------------------------------- Gumbel-softmax, 2 classes
// Plugin default latent_to_discrete (call site /root/reference/src/model_wrapper.py:184-188, :297).
// Device-drawn Gumbel noise of element e from its 32-bit words w0, w1.  One Philox call serves TWO elements (e even:
// words x, y; e odd: z, w): the counter is e >> 1.
// u = (k + 1/2) 2^-23 with k the top 23 bits: every value is exact in float32 and strictly inside (0, 1), so the
// Gumbel noise is finite (|g| < 17.4).  A 24-bit k does NOT work: 16777215 + 0.5 rounds to 2^24, u = 1, g = +inf,
// and (inf - inf) in the softmax below poisons the whole encoder gradient -- a 2^-24 event per draw, i.e. a few
// per cent per training step at B R n = 2.6e5 (found by a soak run; regression test in tests/test_gpu_losses.py).
// The noise is a SAMPLE (no counterpart in the reference to agree with bit for bit): the hardware logarithm
// (v_log_f32, 1 ulp) instead of logf's longer form.  -log u >= 2^-24 for every u above; the clamp keeps a rounding
// of the hardware result towards zero from reaching log(0).
__device__ __forceinline__ float gumbel_from_word(uint32_t w) {
  const float u = __fmul_rn(__uint2float_rn(w >> 9) + 0.5f, 1.1920928955078125e-07f);
  return -__logf(fmaxf(-__logf(u), 2.98023224e-08f));
}
__device__ __forceinline__ u32x4 gumbel_words(int64_t pair, uint32_t k0, uint32_t k1, uint32_t off_lo, uint32_t off_hi) {
  return philox4x32_10((uint32_t)pair, off_lo, off_hi ^ (uint32_t)(pair >> 32), STREAM_GUMBEL, k0, k1);
}

// one element of the Gumbel-softmax (2 classes): spin = argmax, dspin = d p0 / d logit * 2
__device__ __forceinline__ void gumbel_softmax2(float l, float g0, float g1, float tau, float& spin, float& ds) {
  const float y0 = __fdiv_rn(__fadd_rn(l, g0), tau);
  const float y1 = __fdiv_rn(g1, tau);
  const float m = fmaxf(y0, y1);
  const float e0 = expf(y0 - m), e1 = expf(y1 - m);
  const float p0 = e0 / (e0 + e1);
  spin = (y0 >= y1) ? 1.0f : -1.0f;  // argmax, ties -> class 0 (+1)
  ds = 2.0f * p0 * (1.0f - p0) / tau;
}

// four consecutive latent units per thread (n % 4 == 0): one 16-byte load of the logits, two 16-byte stores, two Philox
// calls (or two 16-byte loads of injected noise)
__global__ __launch_bounds__(256) void gumbel_fwd_kernel(const float* __restrict__ logits, int64_t B, int n, int R,
                                                         float tau, const float* __restrict__ gumbels,
                                                         uint32_t k0, uint32_t k1, uint32_t off_lo, uint32_t off_hi,
                                                         const uint64_t* __restrict__ off_dev,
                                                         float* __restrict__ spins, float* __restrict__ dspin) {
  if (off_dev) { const uint64_t o = *off_dev; off_lo = (uint32_t)o; off_hi = (uint32_t)(o >> 32); }
  const int n4 = n >> 2;
  const int64_t total4 = B * R * (int64_t)n4;
  const bool small = total4 < (int64_t)1 << 31;  // (uniform: 32-bit index arithmetic -- two 64-bit divisions by run-time
                                                 // values were a fifth of this kernel's instructions)
  const uint32_t n4u = (uint32_t)n4, n4r = (uint32_t)n4 * (uint32_t)R;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total4; q += (int64_t)gridDim.x * 256) {
    const int i4 = small ? (int)((uint32_t)q % n4u) : (int)(q % n4);
    const int64_t b = small ? (int64_t)((uint32_t)q / n4r) : q / ((int64_t)n4 * R);
    const float4 l = *reinterpret_cast<const float4*>(logits + b * n + 4 * i4);
    const int64_t e = 4 * q;
    float g[8];
    if (gumbels) {
      const float4 a = *reinterpret_cast<const float4*>(gumbels + 2 * e), c = *reinterpret_cast<const float4*>(gumbels + 2 * e + 4);
      g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = c.x; g[5] = c.y; g[6] = c.z; g[7] = c.w;
    } else {
      const u32x4 r0 = gumbel_words(e >> 1, k0, k1, off_lo, off_hi), r1 = gumbel_words((e >> 1) + 1, k0, k1, off_lo, off_hi);
      g[0] = gumbel_from_word(r0.x); g[1] = gumbel_from_word(r0.y); g[2] = gumbel_from_word(r0.z); g[3] = gumbel_from_word(r0.w);
      g[4] = gumbel_from_word(r1.x); g[5] = gumbel_from_word(r1.y); g[6] = gumbel_from_word(r1.z); g[7] = gumbel_from_word(r1.w);
    }
    float4 sp, ds;
    gumbel_softmax2(l.x, g[0], g[1], tau, sp.x, ds.x);
    gumbel_softmax2(l.y, g[2], g[3], tau, sp.y, ds.y);
    gumbel_softmax2(l.z, g[4], g[5], tau, sp.z, ds.z);
    gumbel_softmax2(l.w, g[6], g[7], tau, sp.w, ds.w);
    *reinterpret_cast<float4*>(spins + e) = sp;
    *reinterpret_cast<float4*>(dspin + e) = ds;
  }
}

// one latent unit per thread: any n, any alignment (the same per-element arithmetic and Philox counters: same bits)
__global__ __launch_bounds__(256) void gumbel_fwd_scalar_kernel(const float* __restrict__ logits, int64_t B, int n, int R,
                                                                float tau, const float* __restrict__ gumbels,
                                                                uint32_t k0, uint32_t k1, uint32_t off_lo, uint32_t off_hi,
                                                                const uint64_t* __restrict__ off_dev,
                                                                float* __restrict__ spins, float* __restrict__ dspin) {
  if (off_dev) { const uint64_t o = *off_dev; off_lo = (uint32_t)o; off_hi = (uint32_t)(o >> 32); }
  const int64_t total = B * R * (int64_t)n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e % n);
    const int64_t b = e / ((int64_t)n * R);
    float sp, ds, g0, g1;
    if (gumbels) {
      g0 = gumbels[2 * e]; g1 = gumbels[2 * e + 1];
    } else {
      const u32x4 r = gumbel_words(e >> 1, k0, k1, off_lo, off_hi);
      g0 = gumbel_from_word((e & 1) ? r.z : r.x); g1 = gumbel_from_word((e & 1) ? r.w : r.y);
    }
    gumbel_softmax2(logits[b * n + i], g0, g1, tau, sp, ds);
    spins[e] = sp;
    dspin[e] = ds;
  }
}

// gs2 (optional): a second gradient wrt the spins, added to the first on the way in (the training step has two: the
// decoder's and the MMD's; summing them here saves the separate add pass over (B, R, n))
__global__ __launch_bounds__(256) void gumbel_bwd_kernel(const float* __restrict__ gs, const float* __restrict__ gs2,
                                                         const float* __restrict__ dspin,
                                                         int64_t B, int n, int R, float* __restrict__ gl) {
  const int64_t total = B * (int64_t)n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e % n);
    const int64_t b = e / n;
    float acc = 0.f;
    for (int r = 0; r < R; ++r) {
      const int64_t k = (b * R + r) * n + i;
      const float g = gs2 ? __fadd_rn(gs[k], gs2[k]) : gs[k];
      acc += g * dspin[k];
    }
    gl[e] = acc;
  }
}

// The same over four adjacent latent units per lane (n % 4 == 0): 16-byte loads, and the loads of four repeats in flight
// before the first use -- the scalar form above waits for each repeat's three loads in turn and runs at a quarter of the
// HBM rate.  Per component the arithmetic and its order over r are the scalar form's, so the results are bit-identical.
template <bool HAS2>
__global__ __launch_bounds__(256) void gumbel_bwd_v4_kernel(const float4* __restrict__ gs, const float4* __restrict__ gs2,
                                                            const float4* __restrict__ dspin, int64_t B, int n4, int R,
                                                            float4* __restrict__ gl) {
  const int64_t total = B * (int64_t)n4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e % n4);
    const int64_t b = e / n4;
    const int64_t base = b * R * n4 + i;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto use = [&](const float4& a, const float4& c, const float4& d) {
      float4 g = a;
      if (HAS2) { g.x = __fadd_rn(a.x, c.x); g.y = __fadd_rn(a.y, c.y); g.z = __fadd_rn(a.z, c.z); g.w = __fadd_rn(a.w, c.w); }
      acc.x += g.x * d.x; acc.y += g.y * d.y; acc.z += g.z * d.z; acc.w += g.w * d.w;
    };
    int r = 0;
    for (; r + 4 <= R; r += 4) {
      float4 a[4], c[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = base + (int64_t)(r + u) * n4;
        a[u] = gs[k];
        if (HAS2) c[u] = gs2[k];
        d[u] = dspin[k];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) use(a[u], c[u], d[u]);
    }
    for (; r < R; ++r) {
      const int64_t k = base + (int64_t)r * n4;
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
      if (HAS2) c = gs2[k];
      use(gs[k], c, dspin[k]);
    }
    gl[e] = acc;
  }
}

__global__ void scalar_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
  if (threadIdx.x == 0) out[0] = __fadd_rn(a[0], b[0]);
}

__global__ __launch_bounds__(256) void heaviside_kernel(const float* __restrict__ l, int64_t numel, float* __restrict__ s) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < numel; e += (int64_t)gridDim.x * 256)
    s[e] = l[e] > 0.f ? 1.0f : -1.0f;  // H(0) = 0 -> -1 (/root/reference/src/utils/common.py:164-171)
}

// ---------------------------------------------------------------- MSE + gradient
constexpr int MSE_BLOCKS = 512;

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ recon, const float* __restrict__ img,
                                                          int64_t B, int R, float gscale, double* __restrict__ partial,
                                                          float* __restrict__ grad) {
  const int64_t total4 = B * R * 256;  // float4 units (1024 pixels per image)
  const float4* r4 = reinterpret_cast<const float4*>(recon);
  const float4* i4 = reinterpret_cast<const float4*>(img);
  float4* g4 = reinterpret_cast<float4*>(grad);
  double acc = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / ((int64_t)R * 256);
    const int px = (int)(e % 256);
    const float4 a = r4[e], t = i4[b * 256 + px];
    const float d0 = a.x - t.x, d1 = a.y - t.y, d2 = a.z - t.z, d3 = a.w - t.w;
    acc += (double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2) + (double)(d3 * d3);
    if (grad) g4[e] = make_float4(gscale * d0, gscale * d1, gscale * d2, gscale * d3);
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(64) void mse_final_kernel(const double* __restrict__ partial, int nb, double inv_numel,
                                                       float* __restrict__ loss) {
  double s = 0.0;
  int k = threadIdx.x;
  for (; k + 7 * 64 < nb; k += 8 * 64) {  // eight loads in flight (the same order of additions as one by one)
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[k + 64 * u];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < nb; k += 64) s += partial[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (threadIdx.x == 0) *loss = (float)(s * inv_numel);
}

// ---------------------------------------------------------------- Adam (coupled L2), flat buffers
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t numel,
                                                   float step_size, float b1, float b2, float eps, float wd,
                                                   float bc2_sqrt, float gscale, const float* __restrict__ dyn_step,
                                                   const float* __restrict__ dyn_bc2) {
  if (dyn_step) { step_size = *dyn_step; bc2_sqrt = *dyn_bc2; }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < numel; e += (int64_t)gridDim.x * 256) {
    const float pe = p[e];
    float ge = g[e] * gscale;
    if (wd != 0.f) ge = ge + wd * pe;          // grad = grad.add(param, alpha=weight_decay)
    const float me = m[e] + (1.0f - b1) * (ge - m[e]);  // exp_avg.lerp_(grad, 1 - beta1)
    const float ve = b2 * v[e] + (1.0f - b2) * ge * ge;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[e] = me;
    v[e] = ve;
    const float denom = sqrtf(ve) / bc2_sqrt + eps;
    p[e] = pe - step_size * (me / denom);
  }
}

static inline unsigned grid_for(int64_t n) {
  int64_t b = ceil_div(n, 256);
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace dvg

using namespace dvg;

extern "C" int dvg_gumbel_fwd(const float* logits, int64_t B, int n, int R, float tau, const float* gumbels,
                              uint64_t seed, uint64_t offset, float* spins, float* dspin, const dvg_step_state_t* dyn,
                              dvg_stream_t stream) {
  DVG_REQUIRE(logits && spins && dspin, "gumbel_fwd: null argument");
  DVG_REQUIRE(B > 0 && n > 0 && R > 0 && tau > 0.f, "gumbel_fwd: B=%lld n=%d R=%d tau=%g", (long long)B, n, R, tau);
  if (n % 4 != 0 || ((((uintptr_t)logits | (uintptr_t)spins | (uintptr_t)dspin | (uintptr_t)gumbels) & 15) != 0)) {  // (e.g. a slice view)
    DVG_LAUNCH(K_GUMBEL_FWD, gumbel_fwd_scalar_kernel, dim3(grid_for(B * R * (int64_t)n)), dim3(256), 0, (hipStream_t)stream,
               logits, B, n, R, tau, gumbels, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)offset,
               (uint32_t)(offset >> 32), dyn ? &dyn->gumbel_offset : nullptr, spins, dspin);
    return DVG_OK;
  }
  DVG_LAUNCH(K_GUMBEL_FWD, gumbel_fwd_kernel, dim3(grid_for(B * R * (int64_t)(n / 4))), dim3(256), 0, (hipStream_t)stream,
             logits, B, n, R, tau, gumbels, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)offset,
             (uint32_t)(offset >> 32), dyn ? &dyn->gumbel_offset : nullptr, spins, dspin);
  return DVG_OK;
}

extern "C" int dvg_gumbel_bwd2(const float* grad_spins, const float* grad_spins2, const float* dspin, int64_t B, int n,
                               int R, float* grad_logits, dvg_stream_t stream) {
  DVG_REQUIRE(grad_spins && dspin && grad_logits, "gumbel_bwd: null argument");
  DVG_REQUIRE(B > 0 && n > 0 && R > 0, "gumbel_bwd: bad shape");
  const bool al16 = (((uintptr_t)grad_spins | (uintptr_t)grad_spins2 | (uintptr_t)dspin | (uintptr_t)grad_logits) & 15) == 0;
  if (n % 4 == 0 && al16) {
    const int n4 = n / 4;
    if (grad_spins2)
      DVG_LAUNCH(K_GUMBEL_BWD, gumbel_bwd_v4_kernel<true>, dim3(grid_for(B * (int64_t)n4)), dim3(256), 0,
                 (hipStream_t)stream, (const float4*)grad_spins, (const float4*)grad_spins2, (const float4*)dspin, B, n4,
                 R, (float4*)grad_logits);
    else
      DVG_LAUNCH(K_GUMBEL_BWD, gumbel_bwd_v4_kernel<false>, dim3(grid_for(B * (int64_t)n4)), dim3(256), 0,
                 (hipStream_t)stream, (const float4*)grad_spins, (const float4*)nullptr, (const float4*)dspin, B, n4, R,
                 (float4*)grad_logits);
    return DVG_OK;
  }
  DVG_LAUNCH(K_GUMBEL_BWD, gumbel_bwd_kernel, dim3(grid_for(B * (int64_t)n)), dim3(256), 0, (hipStream_t)stream,
             grad_spins, grad_spins2, dspin, B, n, R, grad_logits);
  return DVG_OK;
}

extern "C" int dvg_gumbel_bwd(const float* grad_spins, const float* dspin, int64_t B, int n, int R,
                              float* grad_logits, dvg_stream_t stream) {
  return dvg_gumbel_bwd2(grad_spins, nullptr, dspin, B, n, R, grad_logits, stream);
}

extern "C" int dvg_scalar_add(const float* a, const float* b, float* out, dvg_stream_t stream) {
  DVG_REQUIRE(a && b && out, "scalar_add: null argument");
  DVG_LAUNCH(K_MISC, scalar_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, out);
  return DVG_OK;
}

extern "C" int dvg_heaviside_fwd(const float* logits, int64_t numel, float* spins, dvg_stream_t stream) {
  DVG_REQUIRE(logits && spins && numel > 0, "heaviside_fwd: bad argument");
  DVG_LAUNCH(K_GUMBEL_FWD, heaviside_kernel, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, logits, numel,
             spins);
  return DVG_OK;
}

namespace dvg {
int launch_mse_final(const double* partial, int nb, double inv_numel, float* loss, hipStream_t s) {
  DVG_LAUNCH(K_MSE, mse_final_kernel, dim3(1), dim3(64), 0, s, partial, nb, inv_numel, loss);
  return DVG_OK;
}
}  // namespace dvg

extern "C" size_t dvg_mse_workspace_bytes(void) { return sizeof(double) * MSE_BLOCKS; }

extern "C" int dvg_mse_fwd_bwd(const float* recon, const float* images, int64_t B, int R, float grad_scale,
                               float* loss_out, float* grad_recon, void* ws, size_t ws_bytes, dvg_stream_t stream) {
  DVG_REQUIRE(recon && images && loss_out && ws, "mse: null argument");
  DVG_REQUIRE(B > 0 && R > 0, "mse: B=%lld R=%d", (long long)B, R);
  if (ws_bytes < dvg_mse_workspace_bytes()) { set_error("mse: workspace too small"); return DVG_E_WORKSPACE; }
  const double numel = (double)B * R * 1024.0;
  const float gs = (float)(2.0 * (double)grad_scale / numel);
  hipStream_t s = (hipStream_t)stream;
  DVG_LAUNCH(K_MSE, mse_partial_kernel, dim3(MSE_BLOCKS), dim3(256), 0, s, recon, images, B, R, gs, (double*)ws,
             grad_recon);
  DVG_LAUNCH(K_MSE, mse_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, MSE_BLOCKS, 1.0 / numel, loss_out);
  return DVG_OK;
}

extern "C" int dvg_adam_step(float* p, const float* g, float* m, float* v, int64_t numel, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                             const dvg_step_state_t* dyn, int dyn_slot, dvg_stream_t stream) {
  DVG_REQUIRE(dyn_slot == 0 || dyn_slot == 1, "adam: dyn_slot=%d", dyn_slot);
  DVG_REQUIRE(p && g && m && v, "adam: null argument");
  DVG_REQUIRE(numel > 0 && step >= 1, "adam: numel=%lld step=%lld", (long long)numel, (long long)step);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  DVG_LAUNCH(K_ADAM, adam_kernel, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, numel,
             (float)((double)lr / bc1), beta1, beta2, eps, weight_decay, (float)sqrt(bc2), grad_scale,
             dyn ? &dyn->adam_step_size[dyn_slot] : nullptr, dyn ? &dyn->adam_bc2_sqrt[dyn_slot] : nullptr);
  return DVG_OK;
}
