// Philox4x32-10 and the bit-specified exp shared by the sampler / noise kernels.
// Device restatement of oracle/philox.py and oracle/gibbs.py::spec_exp: every
// float op is an explicit round-to-nearest intrinsic so no contraction can occur.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dvg {

enum : uint32_t { STREAM_GIBBS = 0, STREAM_INIT = 1, STREAM_GUMBEL = 2, STREAM_DROPOUT = 3 };

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}

__device__ __forceinline__ uint32_t pick(const u32x4& v, uint32_t i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// top 24 bits -> [0,1): exact
__device__ __forceinline__ float u32_to_unit(uint32_t r) {
  return __fmul_rn(__uint2float_rn(r >> 8), 5.9604644775390625e-08f);
}

// exp(z) for |z| <= 87, bit-identical to oracle/gibbs.py::spec_exp
__device__ __forceinline__ float spec_exp(float z) {
  float k = __builtin_rintf(__fmul_rn(z, 1.4426950408889634f));
  float r = __fsub_rn(z, __fmul_rn(k, 0.693359375f));
  r = __fsub_rn(r, __fmul_rn(k, -2.12194440e-4f));
  float p = 1.0f / 720.0f;
  p = __fadd_rn(__fmul_rn(p, r), 1.0f / 120.0f);
  p = __fadd_rn(__fmul_rn(p, r), 1.0f / 24.0f);
  p = __fadd_rn(__fmul_rn(p, r), 1.0f / 6.0f);
  p = __fadd_rn(__fmul_rn(p, r), 0.5f);
  p = __fadd_rn(__fmul_rn(p, r), 1.0f);
  p = __fadd_rn(__fmul_rn(p, r), 1.0f);
  float two_k = __uint_as_float((uint32_t)((int)k + 127) << 23);
  return __fmul_rn(p, two_k);
}

}  // namespace dvg
