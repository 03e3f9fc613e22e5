// Does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros to LDS (or leave it untouched)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, float* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 777.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
  int off = threadIdx.x * 16; if (threadIdx.x & 1) off = 0xFFFF0000 + threadIdx.x * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 1024);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1.f + i;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(in, out, 1024);
  float o[256]; hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]);
  int bad = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { float want = (l & 1) ? 0.f : 1.f + 4*l + j; if (o[4*l+j] != want) ++bad; }
  printf("mismatches vs (odd lanes zero): %d\n", bad);
  return 0;
}
