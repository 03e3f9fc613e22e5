// MFMA tile helpers shared by the convolution translation units (conv_igemm.hip, conv_wgrad.hip).
#pragma once
#include "conv.h"

namespace dvg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));  // operand of v_mfma_f32_32x32x16_bf16  // register-resident (a float4 array can end up in scratch)

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to LDS at wave-uniform base + 16 lane (global_load_lds_dwordx4)
__device__ __forceinline__ void dma16(const void* gsrc, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int crow16(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// float32 -> bfloat16, round to nearest even (NaN stays NaN): the bf16-input mode of the forward / data-gradient GEMMs
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float v) {
  const uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// x = hi + mid + lo with three bf16 pieces (round to nearest even each): exact for every finite float32 whose low pieces
// do not underflow (3 x 8 significand bits).  The operand format of the split mode (conv_precision_mode() == 2).
__device__ __forceinline__ void f32_split3(float x, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
  hi = f32_to_bf16_rne(x);
  const float r1 = x - bf16_to_f32(hi);
  mid = f32_to_bf16_rne(r1);
  lo = f32_to_bf16_rne(r1 - bf16_to_f32(mid));
}

// The same decomposition by truncation, on the float's bits (hi = top 8 significand bits, mid = next 8, lo = last 8: hi + mid
// + lo == x bit for bit, all three of x's sign): five VALU instructions per element, what the GEMM kernels can afford for
// every activation they stage.  Returned as float32 bit patterns whose low 16 bits are zero.
__device__ __forceinline__ void f32_split3_trunc(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  lo = __float_as_uint(r1 - __uint_as_float(mid)) & 0xffff0000u;  // (at most 8 significant bits are left: exact)
}
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// Eight float32 operands (two ds_read_b128 of a row of the LDS image) -> the operand registers of ONE bf16 MFMA k-step.
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
// ... as three bf16 pieces each, by truncation (f32_split3_trunc): hi + mid + lo == x bit for bit
__device__ __forceinline__ void f32x8_split3(const f32x4& q0, const f32x4& q1, bf16x8v& hi, bf16x8v& mid, bf16x8v& lo) {
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  u32x4v h, m, l;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float x0 = u < 2 ? q0[2 * u] : q1[2 * u - 4], x1 = u < 2 ? q0[2 * u + 1] : q1[2 * u - 3];
    const float h0 = __uint_as_float(__float_as_uint(x0) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
    const float r0 = x0 - h0, r1 = x1 - h1;
    const float m0 = __uint_as_float(__float_as_uint(r0) & 0xffff0000u), m1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float t0 = r0 - m0, t1 = r1 - m1;  // (<= 8 significant bits left: exact in bf16)
    // the pieces have zero low halves, so the round-to-nearest pack (v_cvt_pk_bf16_f32: one instruction per pair) is exact
    h[u] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){h0, h1}, bf16x2v));
    m[u] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){m0, m1}, bf16x2v));
    l[u] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){t0, t1}, bf16x2v));
  }
  hi = __builtin_bit_cast(bf16x8v, h);
  mid = __builtin_bit_cast(bf16x8v, m);
  lo = __builtin_bit_cast(bf16x8v, l);
}
// ... rounded to bf16, round to nearest even (v_cvt_pk_bf16_f32): the "bf16 GEMM inputs" mode
typedef float f32x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8v f32x8_to_bf16(const f32x4& q0, const f32x4& q1) {
  const f32x8v x = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
  return __builtin_convertvector(x, bf16x8v);
}

// Diagnostic build only (-DDVG_STAMP): per-phase cycle sums of the main loop, written to ConvArgs.stats
// (which the diagnostic harness points at a debug buffer: 8 uint64 per block).  Never in the product build.
#ifdef DVG_STAMP
#define STAMP(var)                                                                 \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");     \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

}  // namespace dvg
