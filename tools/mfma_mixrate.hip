// Micro-benchmark: the MFMA mix of the MMD pair kernel's tile loop -- per feature tile four v_mfma_f32_32x32x16_bf16 on ONE
// accumulator tile (16 tiles = the whole accumulator file) and one v_mfma_i32_32x32x32_i8 on a Gram tile -- with nothing
// else in the stream.  Is a bf16 <-> int8 switch, or the 4-deep dependent chain, slower than the homogeneous streams of
// tools/mfma_rate.hip?   hipcc --offload-arch=gfx950 -O3 tools/mfma_mixrate.hip -o /tmp/mixrate && /tmp/mixrate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: bf16 only (4 per tile, dependent)   1: + one int8 MFMA per tile (builtin)   2: + one int8 MFMA per tile into VGPRs (asm)
// 3: int8 only, one per "tile"
template <int MODE>
__global__ __launch_bounds__(256, 1) void mix_kernel(int iters, int* out) {
  i32x4 a = {(int)threadIdx.x * 0x01010101, 0x7f01ff01, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
  i32x4 a2 = {(int)threadIdx.x, 17, 0x3f803f80, 0x3f80bf80}, b2 = {0x3f803f80, 0x3f80bf80, 0x3c003f80, 0x3f803f80};
  f32x16 G[16];
  i32x16 S = {0};
#pragma unroll
  for (int t = 0; t < 16; ++t) G[t] = (f32x16){0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (MODE == 2) {
        if (t == 0) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(S) : "v"(a), "v"(b));
        else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(S) : "v"(a), "v"(b));
      }
      if (MODE == 1 || MODE == 3) S = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, S, 0, 0, 0);
      if (MODE != 3) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          G[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, u & 1 ? a2 : b2), __builtin_bit_cast(bf16x8, u & 2 ? b2 : a2), G[t], 0, 0, 0);
      }
    }
    if (MODE == 2) { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); a[1] ^= S[3] & 1; }
  }
  float acc = 0;
#pragma unroll
  for (int t = 0; t < 16; ++t) acc += G[t][0];
  if (acc == 1234.5f || S[0] == 0x7fffffff) out[0] = (int)acc + S[1];
}

template <int MODE>
void run(const char* name, int per_tile) {
  int* out; hipMalloc(&out, 4);
  const int iters = 1000, cus = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mix_kernel<MODE><<<cus, 256>>>(10, out);
  hipEventRecord(e0);
  mix_kernel<MODE><<<cus, 256>>>(iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s %6.2f ns per MFMA per SIMD\n", name, ms * 1e6 / ((double)iters * 16 * per_tile));
  hipFree(out);
}

int main() {
  run<0>("bf16 x4 per tile, 16 tiles", 4);
  run<3>("int8 x1 per tile (dependent chain)", 1);
  run<1>("bf16 x4 + int8 x1 per tile (builtin, AGPR)", 5);
  run<2>("bf16 x4 + int8 x1 per tile (asm, VGPR dst)", 5);
  return 0;
}
