// Micro-benchmark: what keeps a real kernel's MFMA stream below the issue rate of tools/mfma_rate.hip?  One wave per SIMD
// (512 threads = 2 waves per SIMD optional), bf16 32x32x16 MFMAs with 4 accumulator chains, plus per MFMA:
//   V independent VALU instructions, and / or operands that come from ds_read_b128 (fresh registers every MFMA, LEAD MFMAs ahead).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VALU, int LDS, int LEAD, int SALU = 0>
__global__ __launch_bounds__(512) void mix_kernel(int iters, int* out) {
  __shared__ i32x4 buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = (i32x4){i, i + 1, i + 2, i + 3};
  __syncthreads();
  const int lane = threadIdx.x & 63;
  i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
  f32x16 c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = (f32x16){0};
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  int sreg = iters;
  i32x4 ring[LEAD + 1];
#pragma unroll
  for (int k = 0; k <= LEAD; ++k) ring[k] = a;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (LDS) ring[(u + LEAD) % (LEAD + 1)] = buf[(lane + 16 * ((u + it) & 31)) & 1023];
      __builtin_amdgcn_sched_barrier(0);
      const i32x4 av = LDS ? ring[u % (LEAD + 1)] : a;
      c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b), c[u & 3], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < VALU; ++k) v[k & 3] = v[k & 3] * 1.0001f + 0.5f;
#pragma unroll
      for (int k = 0; k < SALU; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float acc = v[0] + v[1] + v[2] + v[3] + (float)sreg;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc += c[k][0];
  if (acc == 12345.678f) out[0] = 1;
}

template <int VALU, int LDS, int LEAD, int SALU = 0>
void run(int threads) {
  int* out; hipMalloc(&out, 4);
  const int iters = 1000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mix_kernel<VALU, LDS, LEAD, SALU><<<256, threads>>>(10, out);
  hipEventRecord(e0);
  mix_kernel<VALU, LDS, LEAD, SALU><<<256, threads>>>(iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wps = threads / 256.0;
  printf("VALU/MFMA %d  SALU/MFMA %d  LDS operand %d (lead %d)  waves/SIMD %.0f : %6.2f ns per MFMA per SIMD\n", VALU, SALU, LDS, LEAD, wps,
         ms * 1e6 / (iters * 16.0 * wps));
  hipFree(out);
}

int main() {
  run<0, 0, 1>(256); run<2, 0, 1>(256); run<4, 0, 1>(256); run<6, 0, 1>(256); run<8, 0, 1>(256);
  run<0, 1, 1>(256); run<0, 1, 2>(256); run<0, 1, 4>(256); run<0, 1, 8>(256);
  run<4, 1, 4>(256); run<4, 1, 8>(256);
  run<0, 1, 2>(512); run<4, 1, 4>(512); run<6, 0, 1>(512);
  run<0, 0, 1, 2>(256); run<0, 0, 1, 4>(256); run<0, 0, 1, 8>(256); run<2, 1, 4, 4>(256);
  return 0;
}
