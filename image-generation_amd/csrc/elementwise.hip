// BatchNorm (batch statistics) fused with the layer's pointwise tail, forward and backward.
//   encoder stage (/root/reference/src/encoder.py:32-36): BN -> MaxPool2d(2,2) -> LeakyReLU
//   decoder stage (/root/reference/src/decoder.py:41-46): BN -> Dropout2d(0.2) -> [Upsample x2] -> LeakyReLU
// (LeakyReLU commutes with the nearest upsample, which is fused into the consuming convolution.)
// Reductions are two-stage and deterministic: fixed grid of EW_BLOCKS blocks writes per-block
// partials, a finalize kernel sums them in order in double.
#include "kernels.h"
#include "philox.h"

namespace dvg {

// channel-lane mapping of the reducing kernels: C is 1 or a multiple of 32
struct ChanMap {
  int CW, RL, cl, rl;
  __device__ ChanMap(int C) {
    CW = C >= 32 ? 32 : C;
    RL = 256 / CW;
    cl = threadIdx.x % CW;
    rl = threadIdx.x / CW;
  }
};

template <int K>
__device__ __forceinline__ void block_reduce_store(float (&acc)[K], const ChanMap& cm, float* red, float* dst_base,
                                                   int C, int c0) {
#pragma unroll
  for (int k = 0; k < K; ++k) red[k * 256 + threadIdx.x] = acc[k];
  __syncthreads();
  if (cm.rl == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float t = 0.f;
      for (int j = 0; j < cm.RL; ++j) t += red[k * 256 + j * cm.CW + cm.cl];
      dst_base[((size_t)blockIdx.x * K + k) * C + c0 + cm.cl] = t;  // planar [block][k][C]
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------- float4-of-channels variants (C % 32 == 0)
// The BN tails are pure streaming passes; with one float per thread they were latency-bound (one 4-byte load per
// tensor per iteration, ~1-1.7 TB/s on L2/MALL-resident tensors).  Here a thread owns 4 consecutive channels
// (16-byte loads, a wave covers whole rows) and walks two rows per iteration.
typedef float f32x4v __attribute__((ext_vector_type(4)));
struct Chan4 {
  int CQ, RL, cq, rl;  // channel quads per pass (<= 256), rows per pass, this thread's quad and row lane
  __device__ Chan4(int C) {
    CQ = (C >> 2) < 256 ? (C >> 2) : 256;
    RL = 256 / CQ;
    cq = threadIdx.x % CQ;
    rl = threadIdx.x / CQ;
  }
};
__device__ __forceinline__ f32x4v ld4(const float* p) { return *reinterpret_cast<const f32x4v*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4v v) { *reinterpret_cast<f32x4v*>(p) = v; }

// red: [K][4][256]; partials land planar as [block][k][C] like block_reduce_store
template <int K>
__device__ __forceinline__ void block_reduce_store4(f32x4v (&acc)[K], const Chan4& cm, float* red, float* dst_base,
                                                    int C, int c0) {
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(k * 4 + j) * 256 + threadIdx.x] = acc[k][j];
  __syncthreads();
  if (cm.rl == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      f32x4v t = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < cm.RL; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] += red[(k * 4 + j) * 256 + r * cm.CQ + cm.cq];
      st4(dst_base + ((size_t)blockIdx.x * K + k) * C + c0 + 4 * cm.cq, t);
    }
  }
  __syncthreads();
}

__global__ void enc_bn_pool_fwd_v4_kernel(const float* __restrict__ Y, int64_t Q, int C, const float* __restrict__ mean,
                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, int lrelu, float* __restrict__ out);
__global__ void dec_bn_act_fwd_v4_kernel(const float* __restrict__ Y, int64_t M, int C, int logHW,
                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const float* __restrict__ mask, float* __restrict__ X);

// ---------------------------------------------------------------- generic column sums of partials
// one wavefront per output element: lanes stride over the G partials, fixed-shape shuffle tree in
// double -> deterministic and ~G/64 dependent adds deep instead of G
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum of part[g * stride] over g = lane, lane + 64, ... < G in that order (double), with eight loads in flight: the
// partials were written by another kernel's blocks on other XCDs, so every dependent load is a trip past the L2
// (~1 us each; the finalizer kernels are a few dependent trips long and sit on the step's critical chain at c2).
__device__ __forceinline__ double strided_sum8(const float* __restrict__ p, int lane, int G, size_t stride) {
  double s = 0.0;
  for (int g0 = lane; g0 < G; g0 += 512) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = g0 + 64 * u < G ? p[(size_t)(g0 + 64 * u) * stride] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (g0 + 64 * u < G) s += (double)v[u];
  }
  return s;
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int G, int stride, int count,
                                                     float scale, float* __restrict__ out, int permA, int permB) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= count) return;
  double s = strided_sum8(part + w, lane, G, (size_t)stride);
  s = wave_sum(s);
  if (lane == 0) {
    const int o = permA > 0 ? (w % permA) * permB + w / permA : w;
    out[o] = (float)(s * (double)scale);
  }
}

int launch_colsum(const float* part, int G, int stride, int count, float scale, float* out, int permA, int permB,
                  hipStream_t s) {
  DVG_LAUNCH(K_MISC, colsum_kernel, dim3((unsigned)ceil_div(count, 4)), dim3(256), 0, s, part, G, stride, count, scale,
             out, permA, permB);
  return DVG_OK;
}

__global__ __launch_bounds__(256) void colsum_batch_kernel(ColsumBatch b) {
  const ColsumJob& j = b.job[blockIdx.y];
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= j.count) return;
  double s = strided_sum8(j.part + w, lane, j.G, (size_t)j.stride);
  s = wave_sum(s);
  if (lane == 0) {
    const int o = j.permA > 0 ? (w % j.permA) * j.permB + w / j.permA : w;
    j.out[o] = (float)(s * (double)j.scale);
  }
}

int launch_colsum_batch(const ColsumBatch& b, hipStream_t s) {
  if (b.n == 0) return DVG_OK;
  int most = 1;
  for (int k = 0; k < b.n; ++k) most = b.job[k].count > most ? b.job[k].count : most;
  DVG_LAUNCH(K_MISC, colsum_batch_kernel, dim3((unsigned)ceil_div(most, 4), (unsigned)b.n), dim3(256), 0, s, b);
  return DVG_OK;
}

__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ part, int G, int stride, int count_a,
                                                      float* __restrict__ out_a, int count_b, float* __restrict__ out_b) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= count_a + count_b) return;
  double s = strided_sum8(part + w, lane, G, (size_t)stride);
  s = wave_sum(s);
  if (lane == 0) {
    if (w < count_a) out_a[w] = (float)s;
    else out_b[w - count_a] = (float)s;
  }
}

int launch_colsum2(const float* part, int G, int stride, int count_a, float* out_a, int count_b, float* out_b,
                   hipStream_t s) {
  DVG_LAUNCH(K_MISC, colsum2_kernel, dim3((unsigned)ceil_div(count_a + count_b, 4)), dim3(256), 0, s, part, G, stride,
             count_a, out_a, count_b, out_b);
  return DVG_OK;
}

// part[block][c] = sum over the block's rows of mat[row][c]   (cols is a multiple of 32)
__global__ __launch_bounds__(256) void rowsum_partial_kernel(const float* __restrict__ mat, int64_t rows, int cols,
                                                             float* __restrict__ part) {
  __shared__ float red[256];
  const ChanMap cm(cols);
  for (int c0 = 0; c0 < cols; c0 += cm.CW) {
    float acc[1] = {0.f};
    for (int64_t r = (int64_t)blockIdx.x * cm.RL + cm.rl; r < rows; r += (int64_t)gridDim.x * cm.RL)
      acc[0] += mat[r * cols + c0 + cm.cl];
    block_reduce_store<1>(acc, cm, red, part, cols, c0);
  }
}

// The same with a thread owning four consecutive columns (16-byte loads, a wavefront covers 1 KiB of a row) and four
// rows in flight per thread: the one-float form above walked a 2048-column matrix in 64 passes of 128-byte row pieces
// (c3's decoder Linear bias gradient: 268 MB in 454 us = 0.6 TB/s, on the tail of the decoder's backward).
__global__ __launch_bounds__(256) void rowsum_partial_v4_kernel(const float* __restrict__ mat, int64_t rows, int cols,
                                                                float* __restrict__ part) {
  __shared__ float red[4 * 256];
  const Chan4 cm(cols);
  for (int c0 = 0; c0 < cols; c0 += 4 * cm.CQ) {
    f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    const int64_t stride = (int64_t)gridDim.x * cm.RL;
    int64_t r = (int64_t)blockIdx.x * cm.RL + cm.rl;
    const float* p = mat + c0 + 4 * cm.cq;
    for (; r + 3 * stride < rows; r += 4 * stride) {
      const f32x4v v0 = ld4(p + r * cols), v1 = ld4(p + (r + stride) * cols), v2 = ld4(p + (r + 2 * stride) * cols),
                   v3 = ld4(p + (r + 3 * stride) * cols);
      acc[0] += (v0 + v1) + (v2 + v3);
    }
    for (; r < rows; r += stride) acc[0] += ld4(p + r * cols);
    block_reduce_store4<1>(acc, cm, red, part, cols, c0);
  }
}

int launch_rowsum_partial(const float* mat, int64_t rows, int cols, float* part, hipStream_t s) {
  if (cols % 4 == 0 && ((cols >> 2) <= 256 ? 256 % (cols >> 2) == 0 : (cols >> 2) % 256 == 0)) {
    DVG_LAUNCH(K_MISC, rowsum_partial_v4_kernel, dim3(EW_BLOCKS), dim3(256), 0, s, mat, rows, cols, part);
    return DVG_OK;
  }
  DVG_LAUNCH(K_MISC, rowsum_partial_kernel, dim3(EW_BLOCKS), dim3(256), 0, s, mat, rows, cols, part);
  return DVG_OK;
}

// out[w] = in[(w % A) * B + w / A]
__global__ __launch_bounds__(256) void permute_vec_kernel(const float* __restrict__ in, int count, int A, int B,
                                                          float* __restrict__ out) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w < count) out[w] = in[(w % A) * B + w / A];
}

int launch_permute_vec(const float* in, int count, int A, int B, float* out, hipStream_t s) {
  DVG_LAUNCH(K_MISC, permute_vec_kernel, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, s, in, count, A, B, out);
  return DVG_OK;
}

// ---------------------------------------------------------------- BN statistics
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblk, int C, double M,
                                                          int training, float* __restrict__ mean,
                                                          float* __restrict__ invstd, float* __restrict__ rm,
                                                          float* __restrict__ rv, int64_t* __restrict__ nbt) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wavefront per channel
  const int lane = threadIdx.x & 63;
  if (c == 0 && lane == 0 && training && nbt) *nbt += 1;
  if (c >= C) return;
  if (!training) {
    if (lane == 0) {
      mean[c] = rm[c];
      invstd[c] = 1.0f / sqrtf(rv[c] + BN_EPS);
    }
    return;
  }
  double s1 = strided_sum8(part + (size_t)c * 2, lane, nblk, (size_t)C * 2);
  double s2 = strided_sum8(part + (size_t)c * 2 + 1, lane, nblk, (size_t)C * 2);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane != 0) return;
  const double mu = s1 / M;
  double var = s2 / M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
  if (rm) {  // torch: running = (1 - momentum) * running + momentum * batch, unbiased variance
    const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
    rm[c] = (float)((1.0 - (double)BN_MOMENTUM) * (double)rm[c] + (double)BN_MOMENTUM * mu);
    rv[c] = (float)((1.0 - (double)BN_MOMENTUM) * (double)rv[c] + (double)BN_MOMENTUM * unbiased);
  }
}

// Long partial lists (large batches: 10^4..10^5 row blocks): fold them to BN_FOLD_ROWS rows first.  Block g sums rows
// [g R, (g+1) R) of the [nblk][2C] matrix with the 2C values of a row spread over consecutive lanes (coalesced), in
// double, fixed order; the rows land behind the list and bn_finalize_kernel then runs on those.
__global__ __launch_bounds__(256) void bn_fold_partials_kernel(const float* __restrict__ part, int nblk, int C2,
                                                               float* __restrict__ out) {
  __shared__ double red[256];
  const int rows_per = (nblk + (int)gridDim.x - 1) / (int)gridDim.x;
  const int r0 = blockIdx.x * rows_per, r1 = r0 + rows_per < nblk ? r0 + rows_per : nblk;
  const int CW = C2 < 256 ? C2 : 256, RL = 256 / CW, cl = threadIdx.x % CW, rl = threadIdx.x / CW;
  for (int c0 = 0; c0 < C2; c0 += CW) {
    double acc = 0.0;
    if (rl < RL && c0 + cl < C2) {
      const float* col = part + c0 + cl;
      int k = r0 + rl;
      for (; k + 3 * RL < r1; k += 4 * RL) {  // four loads in flight per thread
        const float v0 = col[(size_t)k * C2], v1 = col[(size_t)(k + RL) * C2], v2 = col[(size_t)(k + 2 * RL) * C2],
                    v3 = col[(size_t)(k + 3 * RL) * C2];
        acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
      }
      for (; k < r1; k += RL) acc += (double)col[(size_t)k * C2];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && c0 + cl < C2) {
      double t = 0.0;
      for (int j = 0; j < RL; ++j) t += red[j * CW + cl];
      out[(size_t)blockIdx.x * C2 + c0 + cl] = (float)t;
    }
    __syncthreads();
  }
}

int launch_bn_finalize(const float* stats_part, int nblk, int C, int64_t M, int training, float* mean, float* invstd,
                       float* running_mean, float* running_var, int64_t* nbt, hipStream_t s) {
  if (training && nblk >= 1024) {
    float* folded = const_cast<float*>(stats_part) + (size_t)nblk * C * 2;
    DVG_LAUNCH(K_BN_FINALIZE, bn_fold_partials_kernel, dim3(BN_FOLD_ROWS), dim3(256), 0, s, stats_part, nblk, 2 * C, folded);
    DVG_LAUNCH(K_BN_FINALIZE, bn_finalize_kernel, dim3((unsigned)ceil_div(C, 4)), dim3(256), 0, s, folded, BN_FOLD_ROWS, C,
               (double)M, training, mean, invstd, running_mean, running_var, nbt);
    return DVG_OK;
  }
  DVG_LAUNCH(K_BN_FINALIZE, bn_finalize_kernel, dim3((unsigned)ceil_div(C, 4)), dim3(256), 0, s, stats_part, nblk, C,
             (double)M, training, mean, invstd, running_mean, running_var, nbt);
  return DVG_OK;
}

// ---------------------------------------------------------------- encoder: BN -> maxpool -> lrelu
// Y: [4Q][C] (Morton: rows 4q..4q+3 are one 2x2 window, in scan order), out: [Q][C]
__global__ __launch_bounds__(256) void enc_bn_pool_fwd_kernel(const float* __restrict__ Y, int64_t Q, int C,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int lrelu,
                                                              float* __restrict__ out) {
  const int64_t total = Q * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t q = e / C;
    const float mu = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
    const float* y = Y + (q * 4) * C + c;
    float best = fmaf((y[0] - mu) * is, g, b);
#pragma unroll
    for (int s = 1; s < 4; ++s) {
      const float z = fmaf((y[(size_t)s * C] - mu) * is, g, b);
      best = z > best ? z : best;  // strict: the first maximum wins, as in torch's max_pool2d
    }
    out[e] = (lrelu && best < 0.f) ? best * LRELU_SLOPE : best;
  }
}

int launch_enc_bn_pool_fwd(const float* Y, int64_t Q, int C, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, int lrelu, float* out, hipStream_t s) {
  if (C % 32 == 0) {
    const int64_t b4 = ceil_div(Q * (C / 4), 256);
    DVG_LAUNCH(K_ENC_BN_POOL_FWD, enc_bn_pool_fwd_v4_kernel, dim3((unsigned)(b4 > 2048 ? 2048 : b4)), dim3(256), 0, s, Y, Q,
               C, mean, invstd, gamma, beta, lrelu, out);
    return DVG_OK;
  }
  const int64_t b = ceil_div(Q * C, 256);
  DVG_LAUNCH(K_ENC_BN_POOL_FWD, enc_bn_pool_fwd_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, s, Y, Q, C,
             mean, invstd, gamma, beta, lrelu, out);
  return DVG_OK;
}

// recompute the window: returns argmax index, its zhat, and the LeakyReLU slope at the pooled value
__device__ __forceinline__ int enc_window(const float* y, int C, float mu, float is, float g, float b, int lrelu,
                                          float (&zh)[4], float& slope) {
  int arg = 0;
  float best = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    zh[s] = (y[(size_t)s * C] - mu) * is;
    const float z = fmaf(zh[s], g, b);
    if (s == 0 || z > best) { best = z; arg = s; }
  }
  slope = (lrelu && !(best > 0.f)) ? LRELU_SLOPE : 1.0f;
  return arg;
}

__global__ __launch_bounds__(256) void enc_bn_pool_bwd_reduce_kernel(const float* __restrict__ Y, int64_t Q, int C,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, int lrelu,
                                                                     const float* __restrict__ dOut,
                                                                     float* __restrict__ part) {
  __shared__ float red[2 * 256];
  const ChanMap cm(C);
  for (int c0 = 0; c0 < C; c0 += cm.CW) {
    const int c = c0 + cm.cl;
    const float mu = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
    float acc[2] = {0.f, 0.f};
    for (int64_t q = (int64_t)blockIdx.x * cm.RL + cm.rl; q < Q; q += (int64_t)gridDim.x * cm.RL) {
      float zh[4], slope;
      const int arg = enc_window(Y + (q * 4) * C + c, C, mu, is, g, b, lrelu, zh, slope);
      const float dz = dOut[q * C + c] * slope;
      acc[0] += dz;
      acc[1] = fmaf(dz, zh[arg], acc[1]);
    }
    block_reduce_store<2>(acc, cm, red, part, C, c0);
  }
}

__global__ __launch_bounds__(256) void enc_bn_pool_bwd_apply_kernel(const float* __restrict__ Y, int64_t Q, int C,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int lrelu,
                                                                    const float* __restrict__ dOut,
                                                                    const float* __restrict__ sum_dz,
                                                                    const float* __restrict__ sum_dzzh, float inv_m,
                                                                    float* __restrict__ dY, float* __restrict__ part_db) {
  __shared__ float red[256];
  const ChanMap cm(C);
  for (int c0 = 0; c0 < C; c0 += cm.CW) {
    const int c = c0 + cm.cl;
    const float mu = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
    const float m1 = sum_dz[c] * inv_m, m2 = sum_dzzh[c] * inv_m, gi = g * is;
    float acc[1] = {0.f};
    for (int64_t q = (int64_t)blockIdx.x * cm.RL + cm.rl; q < Q; q += (int64_t)gridDim.x * cm.RL) {
      float zh[4], slope;
      const int arg = enc_window(Y + (q * 4) * C + c, C, mu, is, g, b, lrelu, zh, slope);
      const float dz = dOut[q * C + c] * slope;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float v = gi * (((s == arg) ? dz : 0.f) - m1 - zh[s] * m2);
        dY[(q * 4 + s) * C + c] = v;
        acc[0] += v;
      }
    }
    block_reduce_store<1>(acc, cm, red, part_db, C, c0);
  }
}

// ---- encoder stage, float4-of-channels variants
__global__ __launch_bounds__(256) void enc_bn_pool_fwd_v4_kernel(const float* __restrict__ Y, int64_t Q, int C,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int lrelu,
                                                                 float* __restrict__ out) {
  const Chan4 cm(C);
  for (int c0 = 0; c0 < C; c0 += 4 * cm.CQ) {
    const int c = c0 + 4 * cm.cq;
    const f32x4v mu = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
    for (int64_t q = (int64_t)blockIdx.x * cm.RL + cm.rl; q < Q; q += (int64_t)gridDim.x * cm.RL) {
      const float* y = Y + (q * 4) * C + c;
      f32x4v w[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) w[s] = ld4(y + (size_t)s * C);
      f32x4v o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float best = fmaf((w[0][j] - mu[j]) * is[j], g[j], b[j]);
#pragma unroll
        for (int s = 1; s < 4; ++s) {
          const float z = fmaf((w[s][j] - mu[j]) * is[j], g[j], b[j]);
          best = z > best ? z : best;  // strict: the first maximum wins, as in torch's max_pool2d
        }
        o[j] = (lrelu && best < 0.f) ? best * LRELU_SLOPE : best;
      }
      st4(out + q * C + c, o);
    }
  }
}

template <bool APPLY>
__global__ __launch_bounds__(256) void enc_bn_pool_bwd_v4_kernel(const float* __restrict__ Y, int64_t Q, int C,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int lrelu,
                                                                 const float* __restrict__ dOut,
                                                                 const float* __restrict__ sum_dz,
                                                                 const float* __restrict__ sum_dzzh, float inv_m,
                                                                 float* __restrict__ dY, float* __restrict__ part) {
  __shared__ float red[2 * 4 * 256];
  const Chan4 cm(C);
  for (int c0 = 0; c0 < C; c0 += 4 * cm.CQ) {
    const int c = c0 + 4 * cm.cq;
    const f32x4v mu = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
    f32x4v m1 = {0.f, 0.f, 0.f, 0.f}, m2 = m1;
    if (APPLY) { m1 = ld4(sum_dz + c) * inv_m; m2 = ld4(sum_dzzh + c) * inv_m; }
    f32x4v acc[APPLY ? 1 : 2];
#pragma unroll
    for (int k = 0; k < (APPLY ? 1 : 2); ++k) acc[k] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    for (int64_t q = (int64_t)blockIdx.x * cm.RL + cm.rl; q < Q; q += (int64_t)gridDim.x * cm.RL) {
      const float* y = Y + (q * 4) * C + c;
      f32x4v w[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) w[s] = ld4(y + (size_t)s * C);
      const f32x4v go = ld4(dOut + q * C + c);
      f32x4v v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // recompute the window: argmax, its zhat, and the LeakyReLU slope at the pooled value
        float zh[4], best = 0.f;
        int arg = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          zh[s] = (w[s][j] - mu[j]) * is[j];
          const float z = fmaf(zh[s], g[j], b[j]);
          if (s == 0 || z > best) { best = z; arg = s; }
        }
        const float slope = (lrelu && !(best > 0.f)) ? LRELU_SLOPE : 1.0f;
        const float dz = go[j] * slope;
        if (APPLY) {
          const float gi = g[j] * is[j];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float val = gi * (((s == arg) ? dz : 0.f) - m1[j] - zh[s] * m2[j]);
            v[s][j] = val;
            acc[0][j] += val;
          }
        } else {
          acc[0][j] += dz;
          acc[APPLY ? 0 : 1][j] = fmaf(dz, zh[arg], acc[APPLY ? 0 : 1][j]);
        }
      }
      if (APPLY) {
#pragma unroll
        for (int s = 0; s < 4; ++s) st4(dY + (q * 4 + s) * C + c, v[s]);
      }
    }
    block_reduce_store4<APPLY ? 1 : 2>(acc, cm, red, part, C, c0);
  }
}

// The reduce pass from the POOLED activations the forward call kept (the next layer's input): the pooled value is
// lrelu(best) with best = gamma zhat_max + beta, both maps invertible, so (sum dz, sum dz zhat_max) need two values per pooled
// pixel and channel -- the gradient and the activation -- instead of the gradient and the four pre-BatchNorm outputs of the
// window (5 -> 2 floats read per element: 167 -> ~70 us over c3's three layers, on the data-gradient chain).  zhat_max =
// (best - beta) / gamma carries the rounding of best (6e-8 max(|zhat|, |beta / gamma|)); a channel whose |gamma| is too small
// for that to be harmless -- or zero: the window's maximum is then its first element, whatever zhat -- takes the window itself.
__global__ __launch_bounds__(256) void enc_bn_pool_bwd_reduce_p_kernel(const float* __restrict__ Y, const float* __restrict__ P,
                                                                       int64_t Q, int C, const float* __restrict__ mean,
                                                                       const float* __restrict__ invstd,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, int lrelu,
                                                                       const float* __restrict__ dOut, float* __restrict__ part) {
  __shared__ float red[2 * 4 * 256];
  const Chan4 cm(C);
  for (int c0 = 0; c0 < C; c0 += 4 * cm.CQ) {
    const int c = c0 + 4 * cm.cq;
    const f32x4v mu = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
    bool inv_ok = true;
    f32x4v ginv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      inv_ok = inv_ok && fabsf(g[j]) >= 1e-3f * (1.0f + fabsf(b[j]));
      ginv[j] = 1.0f / g[j];
    }
    f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (inv_ok) {
      const int64_t stride = (int64_t)gridDim.x * cm.RL;
      int64_t q = (int64_t)blockIdx.x * cm.RL + cm.rl;
      auto one = [&](const f32x4v& pv, const f32x4v& go) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool neg = lrelu && !(pv[j] > 0.f);
          const float best = neg ? pv[j] * (1.0f / LRELU_SLOPE) : pv[j];
          const float dz = go[j] * (neg ? LRELU_SLOPE : 1.0f);
          acc[0][j] += dz;
          acc[1][j] = fmaf(dz, (best - b[j]) * ginv[j], acc[1][j]);
        }
      };
      for (; q + stride < Q; q += 2 * stride) {  // (two rows in flight)
        const f32x4v p0 = ld4(P + q * C + c), g0 = ld4(dOut + q * C + c);
        const f32x4v p1 = ld4(P + (q + stride) * C + c), g1 = ld4(dOut + (q + stride) * C + c);
        one(p0, g0);
        one(p1, g1);
      }
      for (; q < Q; q += stride) one(ld4(P + q * C + c), ld4(dOut + q * C + c));
    } else {
      for (int64_t q = (int64_t)blockIdx.x * cm.RL + cm.rl; q < Q; q += (int64_t)gridDim.x * cm.RL) {
        const float* y = Y + (q * 4) * C + c;
        f32x4v w[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) w[s] = ld4(y + (size_t)s * C);
        const f32x4v go = ld4(dOut + q * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float zh[4], best = 0.f;
          int arg = 0;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            zh[s] = (w[s][j] - mu[j]) * is[j];
            const float z = fmaf(zh[s], g[j], b[j]);
            if (s == 0 || z > best) { best = z; arg = s; }
          }
          const float dz = go[j] * ((lrelu && !(best > 0.f)) ? LRELU_SLOPE : 1.0f);
          acc[0][j] += dz;
          acc[1][j] = fmaf(dz, zh[arg], acc[1][j]);
        }
      }
    }
    block_reduce_store4<2>(acc, cm, red, part, C, c0);
  }
}

int launch_enc_bn_pool_bwd_reduce(const float* Y, int64_t Q, int C, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, int lrelu, const float* dOut, float* part,
                                  hipStream_t s, const float* pooled) {
  if (C % 32 == 0 && pooled) {
    DVG_LAUNCH(K_ENC_BN_POOL_BWD_REDUCE, enc_bn_pool_bwd_reduce_p_kernel, dim3(EW_BLOCKS), dim3(256), 0, s, Y, pooled, Q, C, mean,
               invstd, gamma, beta, lrelu, dOut, part);
    return DVG_OK;
  }
  if (C % 32 == 0) {
    DVG_LAUNCH(K_ENC_BN_POOL_BWD_REDUCE, enc_bn_pool_bwd_v4_kernel<false>, dim3(EW_BLOCKS), dim3(256), 0, s, Y, Q, C, mean,
               invstd, gamma, beta, lrelu, dOut, nullptr, nullptr, 0.f, nullptr, part);
    return DVG_OK;
  }
  DVG_LAUNCH(K_ENC_BN_POOL_BWD_REDUCE, enc_bn_pool_bwd_reduce_kernel, dim3(EW_BLOCKS), dim3(256), 0, s, Y, Q, C, mean,
             invstd, gamma, beta, lrelu, dOut, part);
  return DVG_OK;
}

int launch_enc_bn_pool_bwd_apply(const float* Y, int64_t Q, int C, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int lrelu, const float* dOut, const float* sum_dz,
                                 const float* sum_dzzh, float* dY, float* part_db, hipStream_t s) {
  if (C % 32 == 0) {
    DVG_LAUNCH(K_ENC_BN_POOL_BWD_APPLY, enc_bn_pool_bwd_v4_kernel<true>, dim3(EW_BLOCKS), dim3(256), 0, s, Y, Q, C, mean,
               invstd, gamma, beta, lrelu, dOut, sum_dz, sum_dzzh, (float)(1.0 / (4.0 * (double)Q)), dY, part_db);
    return DVG_OK;
  }
  DVG_LAUNCH(K_ENC_BN_POOL_BWD_APPLY, enc_bn_pool_bwd_apply_kernel, dim3(EW_BLOCKS), dim3(256), 0, s, Y, Q, C, mean,
             invstd, gamma, beta, lrelu, dOut, sum_dz, sum_dzzh, (float)(1.0 / (4.0 * (double)Q)), dY, part_db);
  return DVG_OK;
}

// ---------------------------------------------------------------- decoder: BN -> dropout2d -> lrelu
// all four Dropout2d keep-masks of a decoder forward in one launch (blockIdx.y = layer)
struct DropoutJobs { float* mask[4]; int C[4]; };
__global__ __launch_bounds__(256) void dropout_mask_kernel(int64_t N, DropoutJobs jobs, uint32_t k0, uint32_t k1,
                                                           uint32_t off_lo, uint32_t off_hi,
                                                           const uint64_t* __restrict__ off_dev) {
  if (off_dev) { const uint64_t o = *off_dev; off_lo = (uint32_t)o; off_hi = (uint32_t)(o >> 32); }
  const uint32_t layer = blockIdx.y;
  float* mask = jobs.mask[layer];
  const int64_t total = N * jobs.C[layer];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const u32x4 r = philox4x32_10((uint32_t)e, off_lo ^ (layer << 28), off_hi ^ (uint32_t)(e >> 32), STREAM_DROPOUT, k0, k1);
    mask[e] = u32_to_unit(r.x) < DROPOUT_KEEP ? 1.0f : 0.0f;
  }
}

int launch_dropout_masks(int64_t N, const int C[4], float* const mask[4], uint64_t seed, uint64_t offset,
                         const uint64_t* offset_dev, hipStream_t s) {
  DropoutJobs jobs;
  int cmax = 1;
  for (int l = 0; l < 4; ++l) { jobs.mask[l] = mask[l]; jobs.C[l] = C[l]; if (C[l] > cmax) cmax = C[l]; }
  const int64_t b = ceil_div(N * cmax, 256);
  DVG_LAUNCH(K_MISC, dropout_mask_kernel, dim3((unsigned)(b > 1024 ? 1024 : b), 4), dim3(256), 0, s, N, jobs, (uint32_t)seed,
             (uint32_t)(seed >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), offset_dev);
  return DVG_OK;
}

__global__ __launch_bounds__(256) void dec_bn_act_fwd_kernel(const float* __restrict__ Y, int64_t M, int C, int logHW,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mask, float* __restrict__ X) {
  const int64_t total = M * C;
  const float keep_scale = 1.0f / DROPOUT_KEEP;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t m = e / C;
    float z = fmaf((Y[e] - mean[c]) * invstd[c], gamma[c], beta[c]);
    if (mask) z *= mask[(m >> logHW) * C + c] * keep_scale;
    X[e] = z < 0.f ? z * LRELU_SLOPE : z;
  }
}

int launch_dec_bn_act_fwd(const float* Y, int64_t M, int C, int logHW, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, const float* mask, float* X, hipStream_t s) {
  if (C % 32 == 0) {
    const int64_t b4 = ceil_div(M * (C / 4), 256);
    DVG_LAUNCH(K_DEC_BN_ACT_FWD, dec_bn_act_fwd_v4_kernel, dim3((unsigned)(b4 > 2048 ? 2048 : b4)), dim3(256), 0, s, Y, M, C,
               logHW, mean, invstd, gamma, beta, mask, X);
    return DVG_OK;
  }
  const int64_t b = ceil_div(M * C, 256);
  DVG_LAUNCH(K_DEC_BN_ACT_FWD, dec_bn_act_fwd_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, s, Y, M, C,
             logHW, mean, invstd, gamma, beta, mask, X);
  return DVG_OK;
}

__device__ __forceinline__ float dec_dz(float dx, float x, float mk) {
  return dx * ((x > 0.f) ? 1.0f : LRELU_SLOPE) * mk;
}

__global__ __launch_bounds__(256) void dec_bn_act_bwd_reduce_kernel(const float* __restrict__ Y,
                                                                    const float* __restrict__ X, int64_t M, int C,
                                                                    int logHW, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    const float* __restrict__ mask,
                                                                    const float* __restrict__ dX,
                                                                    float* __restrict__ part) {
  __shared__ float red[2 * 256];
  const ChanMap cm(C);
  const float keep_scale = 1.0f / DROPOUT_KEEP;
  for (int c0 = 0; c0 < C; c0 += cm.CW) {
    const int c = c0 + cm.cl;
    const float mu = mean[c], is = invstd[c];
    float acc[2] = {0.f, 0.f};
    for (int64_t m = (int64_t)blockIdx.x * cm.RL + cm.rl; m < M; m += (int64_t)gridDim.x * cm.RL) {
      const int64_t e = m * C + c;
      const float mk = mask ? mask[(m >> logHW) * C + c] * keep_scale : 1.0f;
      const float dz = dec_dz(dX[e], X[e], mk);
      acc[0] += dz;
      acc[1] = fmaf(dz, (Y[e] - mu) * is, acc[1]);
    }
    block_reduce_store<2>(acc, cm, red, part, C, c0);
  }
}

__global__ __launch_bounds__(256) void dec_bn_act_bwd_apply_kernel(const float* __restrict__ Y,
                                                                   const float* __restrict__ X, int64_t M, int C,
                                                                   int logHW, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ mask,
                                                                   const float* __restrict__ dX,
                                                                   const float* __restrict__ sum_dz,
                                                                   const float* __restrict__ sum_dzzh, float inv_m,
                                                                   float* __restrict__ dY, float* __restrict__ part_db) {
  __shared__ float red[256];
  const ChanMap cm(C);
  const float keep_scale = 1.0f / DROPOUT_KEEP;
  for (int c0 = 0; c0 < C; c0 += cm.CW) {
    const int c = c0 + cm.cl;
    const float mu = mean[c], is = invstd[c], gi = gamma[c] * is;
    const float m1 = sum_dz[c] * inv_m, m2 = sum_dzzh[c] * inv_m;
    float acc[1] = {0.f};
    for (int64_t m = (int64_t)blockIdx.x * cm.RL + cm.rl; m < M; m += (int64_t)gridDim.x * cm.RL) {
      const int64_t e = m * C + c;
      const float mk = mask ? mask[(m >> logHW) * C + c] * keep_scale : 1.0f;
      const float dz = dec_dz(dX[e], X[e], mk);
      const float v = gi * (dz - m1 - (Y[e] - mu) * is * m2);
      dY[e] = v;
      acc[0] += v;
    }
    block_reduce_store<1>(acc, cm, red, part_db, C, c0);
  }
}

// ---- decoder stage, float4-of-channels variants
__global__ __launch_bounds__(256) void dec_bn_act_fwd_v4_kernel(const float* __restrict__ Y, int64_t M, int C, int logHW,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ mask, float* __restrict__ X) {
  const Chan4 cm(C);
  const float keep_scale = 1.0f / DROPOUT_KEEP;
  for (int c0 = 0; c0 < C; c0 += 4 * cm.CQ) {
    const int c = c0 + 4 * cm.cq;
    const f32x4v mu = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
    for (int64_t m = (int64_t)blockIdx.x * cm.RL + cm.rl; m < M; m += (int64_t)gridDim.x * cm.RL) {
      const f32x4v y = ld4(Y + m * C + c);
      f32x4v mk = {1.f, 1.f, 1.f, 1.f};
      if (mask) mk = ld4(mask + (m >> logHW) * C + c) * keep_scale;
      f32x4v o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z = fmaf((y[j] - mu[j]) * is[j], g[j], b[j]);
        if (mask) z *= mk[j];
        o[j] = z < 0.f ? z * LRELU_SLOPE : z;
      }
      st4(X + m * C + c, o);
    }
  }
}

template <bool APPLY>
__global__ __launch_bounds__(256) void dec_bn_act_bwd_v4_kernel(const float* __restrict__ Y, const float* __restrict__ X,
                                                                int64_t M, int C, int logHW,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ mask,
                                                                const float* __restrict__ dX,
                                                                const float* __restrict__ sum_dz,
                                                                const float* __restrict__ sum_dzzh, float inv_m,
                                                                float* __restrict__ dY, float* __restrict__ part) {
  __shared__ float red[2 * 4 * 256];
  const Chan4 cm(C);
  const float keep_scale = 1.0f / DROPOUT_KEEP;
  for (int c0 = 0; c0 < C; c0 += 4 * cm.CQ) {
    const int c = c0 + 4 * cm.cq;
    const f32x4v mu = ld4(mean + c), is = ld4(invstd + c), gmv = ld4(gamma + c), btv = ld4(beta + c);
    f32x4v gi = {0.f, 0.f, 0.f, 0.f}, m1 = gi, m2 = gi;
    if (APPLY) { gi = gmv * is; m1 = ld4(sum_dz + c) * inv_m; m2 = ld4(sum_dzzh + c) * inv_m; }
    f32x4v acc[APPLY ? 1 : 2];
#pragma unroll
    for (int k = 0; k < (APPLY ? 1 : 2); ++k) acc[k] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    for (int64_t m = (int64_t)blockIdx.x * cm.RL + cm.rl; m < M; m += (int64_t)gridDim.x * cm.RL) {
      const int64_t e = m * C + c;
      // (the activated map X is not read: x > 0 exactly when z = fma(zhat, gamma, beta) > 0 and the element is kept, and a
      // dropped element's dz is zero whatever its slope -- 134 MB fewer per pass at c3's 8x8 stage)
      const f32x4v dx = ld4(dX + e), y = ld4(Y + e);
      f32x4v mk = {1.f, 1.f, 1.f, 1.f};
      if (mask) mk = ld4(mask + (m >> logHW) * C + c) * keep_scale;
      f32x4v v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float zj = fmaf((y[j] - mu[j]) * is[j], gmv[j], btv[j]);
        const float dz = dx[j] * ((zj > 0.f) ? 1.0f : LRELU_SLOPE) * mk[j];
        if (APPLY) {
          v[j] = gi[j] * (dz - m1[j] - (y[j] - mu[j]) * is[j] * m2[j]);
          acc[0][j] += v[j];
        } else {
          acc[0][j] += dz;
          acc[APPLY ? 0 : 1][j] = fmaf(dz, (y[j] - mu[j]) * is[j], acc[APPLY ? 0 : 1][j]);
        }
      }
      if (APPLY) st4(dY + e, v);
    }
    block_reduce_store4<APPLY ? 1 : 2>(acc, cm, red, part, C, c0);
  }
}

int launch_dec_bn_act_bwd_reduce(const float* Y, const float* X, int64_t M, int C, int logHW, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, const float* mask,
                                 const float* dX, float* part, hipStream_t s) {
  if (C % 32 == 0) {
    DVG_LAUNCH(K_DEC_BN_ACT_BWD_REDUCE, dec_bn_act_bwd_v4_kernel<false>, dim3(EW_BLOCKS), dim3(256), 0, s, Y, X, M, C, logHW,
               mean, invstd, gamma, beta, mask, dX, nullptr, nullptr, 0.f, nullptr, part);
    return DVG_OK;
  }
  DVG_LAUNCH(K_DEC_BN_ACT_BWD_REDUCE, dec_bn_act_bwd_reduce_kernel, dim3(EW_BLOCKS), dim3(256), 0, s, Y, X, M, C, logHW,
             mean, invstd, mask, dX, part);
  return DVG_OK;
}

int launch_dec_bn_act_bwd_apply(const float* Y, const float* X, int64_t M, int C, int logHW, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, const float* mask, const float* dX,
                                const float* sum_dz, const float* sum_dzzh, float* dY, float* part_db, hipStream_t s) {
  if (C % 32 == 0) {
    DVG_LAUNCH(K_DEC_BN_ACT_BWD_APPLY, dec_bn_act_bwd_v4_kernel<true>, dim3(EW_BLOCKS), dim3(256), 0, s, Y, X, M, C, logHW,
               mean, invstd, gamma, beta, mask, dX, sum_dz, sum_dzzh, (float)(1.0 / (double)M), dY, part_db);
    return DVG_OK;
  }
  DVG_LAUNCH(K_DEC_BN_ACT_BWD_APPLY, dec_bn_act_bwd_apply_kernel, dim3(EW_BLOCKS), dim3(256), 0, s, Y, X, M, C, logHW,
             mean, invstd, gamma, mask, dX, sum_dz, sum_dzzh, (float)(1.0 / (double)M), dY, part_db);
  return DVG_OK;
}

}  // namespace dvg
