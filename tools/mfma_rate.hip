// Micro-benchmark: issue rate of the MFMA forms the kernels use, as a function of how many INDEPENDENT accumulator
// chains a wave keeps in flight and of how many waves share a SIMD.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Prints ns per MFMA per SIMD and the implied dense rate of the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NCH>
__global__ __launch_bounds__(512) void rate_kernel(int iters, int* out) {
  i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
  i32x16 ci[NCH];
  f32x16 cf[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { ci[c] = (i32x16){0}; cf[c] = (f32x16){0}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (KIND == 0) ci[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, ci[c], 0, 0, 0);
        if (KIND == 1) cf[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), cf[c], 0, 0, 0);
        if (KIND == 2) cf[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a[0]), __int_as_float(b[0]), cf[c], 0, 0, 0);
      }
  }
  int acc = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc += ci[c][0] + (int)cf[c][0];
  if (acc == 0x7fffffff) out[0] = acc;
}

template <int KIND, int NCH>
void run(const char* name, double ops, int threads, int blocks_per_cu) {
  int* out; hipMalloc(&out, 4);
  const int iters = 2000, cus = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<KIND, NCH><<<cus * blocks_per_cu, threads>>>(10, out);
  hipEventRecord(e0);
  rate_kernel<KIND, NCH><<<cus * blocks_per_cu, threads>>>(iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = threads / 64.0 / 4.0 * blocks_per_cu;
  const double per_simd = (double)iters * 8 * NCH * waves_per_simd;
  printf("%-22s chains %d  waves/SIMD %.0f : %6.2f ns per MFMA per SIMD, chip %7.1f T(FL)OP/s\n", name, NCH, waves_per_simd,
         ms * 1e6 / per_simd, per_simd * 1024 * ops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<0, 1>("i32_32x32x32_i8", 65536, 256, 1); run<0, 2>("i32_32x32x32_i8", 65536, 256, 1);
  run<0, 4>("i32_32x32x32_i8", 65536, 256, 1); run<0, 1>("i32_32x32x32_i8", 65536, 512, 1);
  run<0, 2>("i32_32x32x32_i8", 65536, 512, 1); run<0, 4>("i32_32x32x32_i8", 65536, 512, 1);
  run<1, 1>("f32_32x32x16_bf16", 32768, 256, 1); run<1, 2>("f32_32x32x16_bf16", 32768, 256, 1);
  run<1, 4>("f32_32x32x16_bf16", 32768, 256, 1); run<1, 2>("f32_32x32x16_bf16", 32768, 512, 1);
  run<1, 4>("f32_32x32x16_bf16", 32768, 512, 1);
  run<2, 1>("f32_32x32x2_f32", 4096, 256, 1); run<2, 2>("f32_32x32x2_f32", 4096, 256, 1);
  run<2, 4>("f32_32x32x2_f32", 4096, 256, 1); run<2, 2>("f32_32x32x2_f32", 4096, 512, 1);
  return 0;
}
