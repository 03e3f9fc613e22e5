// GRBM energy and its backward (weighted sufficient statistics).
// Energy: plugin GraphRestrictedBoltzmannMachine.__call__ as used at
// /root/reference/src/losses.py:61;  E(x) = x.h + sum_e J_e x_i x_j.
#include "common.h"
#include "graph.h"

namespace dvg {

// one wavefront per row; double accumulation (cheap here, removes summation-order noise
// from the difference of two means that the quasi-NLL takes)
__global__ __launch_bounds__(256) void grbm_energy_kernel(const float* __restrict__ x, int64_t rows, int n,
                                                          int n_edges, const int32_t* __restrict__ ei,
                                                          const int32_t* __restrict__ ej,
                                                          const float* __restrict__ h,
                                                          const float* __restrict__ J,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * n;
  double acc = 0.0;
  for (int i = lane; i < n; i += 64) acc += (double)xr[i] * (double)h[i];
  for (int e = lane; e < n_edges; e += 64) acc += (double)(xr[ei[e]] * xr[ej[e]]) * (double)J[e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) out[row] = (float)acc;
}

constexpr int SS_SLABS = 64;

// partial[slab][idx] = sum over the slab's rows of w_r * x_ri (idx < n) or w_r * x_ri x_rj
__global__ __launch_bounds__(256) void grbm_suffstats_partial(const float* __restrict__ x, int64_t rows, int n,
                                                              int n_edges, const int32_t* __restrict__ ei,
                                                              const int32_t* __restrict__ ej,
                                                              const float* __restrict__ w,
                                                              double* __restrict__ partial) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int total = n + n_edges;
  if (idx >= total) return;
  const int slab = blockIdx.y;
  const int64_t per = (rows + SS_SLABS - 1) / SS_SLABS;
  const int64_t r0 = slab * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  int a, b;
  if (idx < n) { a = idx; b = -1; } else { a = ei[idx - n]; b = ej[idx - n]; }
  double acc = 0.0;
  for (int64_t r = r0; r < r1; ++r) {
    const float* xr = x + r * n;
    float v = xr[a];
    if (b >= 0) v *= xr[b];
    acc += (double)(w ? w[r] * v : v);
  }
  partial[(size_t)slab * total + idx] = acc;
}

__global__ __launch_bounds__(256) void grbm_suffstats_reduce(const double* __restrict__ partial, int n, int n_edges,
                                                             float scale, float* __restrict__ acc_lin,
                                                             float* __restrict__ acc_quad, int accumulate) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int total = n + n_edges;
  if (idx >= total) return;
  double s = 0.0;
  for (int k = 0; k < SS_SLABS; ++k) s += partial[(size_t)k * total + idx];
  float v = (float)(s * (double)scale);
  float* dst = idx < n ? acc_lin + idx : acc_quad + (idx - n);
  *dst = accumulate ? *dst + v : v;
}

}  // namespace dvg

using namespace dvg;

extern "C" int dvg_grbm_energy(const dvg_graph_t* g, const float* x, int64_t rows, const float* linear,
                               const float* quadratic, float* energy_out, dvg_stream_t stream) {
  DVG_REQUIRE(g && x && linear && quadratic && energy_out, "grbm_energy: null argument");
  DVG_REQUIRE(rows >= 0, "grbm_energy: rows=%lld", (long long)rows);
  if (rows == 0) return DVG_OK;
  hipStream_t s = (hipStream_t)stream;
  DVG_LAUNCH(K_GRBM_ENERGY, grbm_energy_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, x, rows,
             g->n, g->n_edges, g->edge_i, g->edge_j, linear, quadratic, energy_out);
  return DVG_OK;
}

extern "C" size_t dvg_grbm_suffstats_workspace_bytes(const dvg_graph_t* g) {
  return g ? sizeof(double) * (size_t)SS_SLABS * (size_t)(g->n + g->n_edges) : 0;
}

extern "C" int dvg_grbm_suffstats(const dvg_graph_t* g, const float* x, int64_t rows, const float* row_weight,
                                  float scale, float* acc_linear, float* acc_quadratic, int accumulate,
                                  void* ws, size_t ws_bytes, dvg_stream_t stream) {
  DVG_REQUIRE(g && x && acc_linear && acc_quadratic && ws, "grbm_suffstats: null argument");
  DVG_REQUIRE(rows > 0, "grbm_suffstats: rows=%lld", (long long)rows);
  if (ws_bytes < dvg_grbm_suffstats_workspace_bytes(g)) {
    set_error("grbm_suffstats: workspace %zu < %zu", ws_bytes, dvg_grbm_suffstats_workspace_bytes(g));
    return DVG_E_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int total = g->n + g->n_edges;
  const unsigned nb = (unsigned)ceil_div(total, 256);
  DVG_LAUNCH(K_GRBM_SUFFSTATS, grbm_suffstats_partial, dim3(nb, SS_SLABS), dim3(256), 0, s, x, rows, g->n,
             g->n_edges, g->edge_i, g->edge_j, row_weight, (double*)ws);
  DVG_LAUNCH(K_GRBM_SUFFSTATS, grbm_suffstats_reduce, dim3(nb), dim3(256), 0, s, (const double*)ws, g->n,
             g->n_edges, scale, acc_linear, acc_quadratic, accumulate);
  return DVG_OK;
}
