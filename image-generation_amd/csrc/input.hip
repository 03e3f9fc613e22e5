// Input pipeline of the training path on the device:
//   * dvg_resize_binarise: the reference's per-image transform, Resize((S, S)) -> ToTensor() -> round
//     (/root/reference/src/model_wrapper.py:70-77), on a whole uint8 data set at once.  torchvision resizes a PIL image
//     with PIL's BILINEAR resampling: two separable passes (horizontal, then vertical) with 8-bit intermediates and
//     fixed-point coefficients (22 fractional bits); this restates that arithmetic bit for bit (oracle/resize.py is the
//     numpy restatement, pinned against PIL itself by tests/test_oracle_resize.py).  ToTensor divides by 255 and round
//     binarises: the output is 1.0f where the resized byte is >= 128, else 0.0f.
//   * dvg_gather_rows: batch = table[idx] (the DataLoader's shuffled batch of a device-resident data set).
// Both are single-pass HBM-bound kernels: a row of the output is written once, with 16-byte stores.
#include <cmath>

#include "common.h"

namespace dvg {

constexpr int RS_MAX_OUT = 64;   // output side
constexpr int RS_MAX_K = 8;      // taps per output pixel (3 when enlarging)
constexpr int RS_PREC = 22;      // PIL: PRECISION_BITS = 32 - 8 - 2

struct ResizeCoefs {
  int ksize;
  int xmin[RS_MAX_OUT], xcnt[RS_MAX_OUT];
  int k[RS_MAX_OUT][RS_MAX_K];
};

// PIL's precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter, in double as PIL computes them.
static bool resize_coefs(int in_size, int out_size, ResizeCoefs* rc) {
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  if (ksize > RS_MAX_K || out_size > RS_MAX_OUT) return false;
  rc->ksize = ksize;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double w[RS_MAX_K], ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double v = (x + xmin - center + 0.5) * ss;
      v = v < 0.0 ? -v : v;
      w[x] = v < 1.0 ? 1.0 - v : 0.0;
      ww += w[x];
    }
    for (int x = 0; x < ksize; ++x) {
      double v = x < xmax ? w[x] : 0.0;
      if (x < xmax && ww != 0.0) v /= ww;
      rc->k[xx][x] = v < 0 ? (int)(-0.5 + v * (double)(1 << RS_PREC)) : (int)(0.5 + v * (double)(1 << RS_PREC));
    }
    rc->xmin[xx] = xmin;
    rc->xcnt[xx] = xmax;
  }
  return true;
}

__device__ __forceinline__ int rs_clip8(int v) {
  v >>= RS_PREC;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One block per image: the horizontal pass of the whole image into LDS (in_size x out_size bytes), then the vertical
// pass + binarisation, four output pixels (one 16-byte store) per thread.
__global__ __launch_bounds__(256) void resize_binarise_kernel(const uint8_t* __restrict__ src, int64_t N, int in_size,
                                                              int out_size, ResizeCoefs rc, float* __restrict__ out) {
  __shared__ uint8_t img[RS_MAX_OUT * RS_MAX_OUT];
  __shared__ uint8_t hor[RS_MAX_OUT * RS_MAX_OUT];
  for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
    const uint8_t* s = src + n * in_size * in_size;
    for (int e = threadIdx.x; e < in_size * in_size; e += 256) img[e] = s[e];
    __syncthreads();
    for (int e = threadIdx.x; e < in_size * out_size; e += 256) {
      const int y = e / out_size, x = e - y * out_size;
      int acc = 1 << (RS_PREC - 1);
      for (int t = 0; t < rc.xcnt[x]; ++t) acc += (int)img[y * in_size + rc.xmin[x] + t] * rc.k[x][t];
      hor[e] = (uint8_t)rs_clip8(acc);
    }
    __syncthreads();
    float* o = out + n * out_size * out_size;
    for (int e = threadIdx.x; e < out_size * out_size; e += 256) {
      const int y = e / out_size, x = e - y * out_size;
      int acc = 1 << (RS_PREC - 1);
      for (int t = 0; t < rc.xcnt[y]; ++t) acc += (int)hor[(rc.xmin[y] + t) * out_size + x] * rc.k[y][t];
      o[e] = rs_clip8(acc) >= 128 ? 1.0f : 0.0f;  // round(v / 255): v / 255 is never exactly 1/2
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int64_t row_floats,
                                                          const int64_t* __restrict__ idx, int64_t n, int64_t rows,
                                                          float* __restrict__ out, int* __restrict__ bad) {
  const int64_t r = blockIdx.x;
  if (r >= n) return;
  int64_t src = idx[r];
  if (src < 0 || src >= rows) {  // an out-of-range index is reported, not dereferenced
    if (threadIdx.x == 0 && bad) atomicOr(bad, 1);
    src = 0;
  }
  const float* s = table + src * row_floats;
  float* o = out + r * row_floats;
  if ((row_floats & 3) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(s);
    float4* o4 = reinterpret_cast<float4*>(o);
    for (int64_t e = threadIdx.x; e < row_floats / 4; e += 256) o4[e] = s4[e];
  } else {
    for (int64_t e = threadIdx.x; e < row_floats; e += 256) o[e] = s[e];
  }
}

}  // namespace dvg

using namespace dvg;

extern "C" int dvg_resize_binarise(const uint8_t* src, int64_t n_images, int in_size, int out_size, float* out,
                                   dvg_stream_t stream) {
  DVG_REQUIRE(src && out, "resize_binarise: null argument");
  DVG_REQUIRE(n_images >= 0 && in_size >= 2 && in_size <= RS_MAX_OUT && out_size >= 2 && out_size <= RS_MAX_OUT,
              "resize_binarise: n=%lld in=%d out=%d (sides up to %d)", (long long)n_images, in_size, out_size, RS_MAX_OUT);
  if (n_images == 0) return DVG_OK;
  ResizeCoefs rc;
  if (!resize_coefs(in_size, out_size, &rc)) {
    set_error("resize_binarise: %d -> %d needs more than %d taps", in_size, out_size, RS_MAX_K);
    return DVG_E_UNSUPPORTED;
  }
  const unsigned grid = (unsigned)(n_images < 4096 ? n_images : 4096);
  DVG_LAUNCH(K_MISC, resize_binarise_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, n_images, in_size,
             out_size, rc, out);
  return DVG_OK;
}

extern "C" int dvg_gather_rows(const float* table, int64_t rows, int64_t row_floats, const int64_t* idx, int64_t n,
                               float* out, int* bad_index_flag, dvg_stream_t stream) {
  DVG_REQUIRE(table && idx && out, "gather_rows: null argument");
  DVG_REQUIRE(rows > 0 && row_floats > 0 && n >= 0 && n < (1ll << 31), "gather_rows: rows=%lld row_floats=%lld n=%lld",
              (long long)rows, (long long)row_floats, (long long)n);
  if (n == 0) return DVG_OK;
  DVG_LAUNCH(K_MISC, gather_rows_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, table, row_floats, idx, n,
             rows, out, bad_index_flag);
  return DVG_OK;
}
