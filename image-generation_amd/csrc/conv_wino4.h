// Transforms of the Winograd F(4x4, 3x3) kernels (conv_wino4.hip: forward / data gradient; conv_wino4_wgrad.hip: weight
// gradient), one line of a 6x6 / 4x4 tile at a time, in registers.
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]   (conv.h: wino4_pack_entry)
#pragma once
#include "conv_tile.h"

namespace dvg {

// B^T x for one line of six (12 operations); ZE: x0 = x5 = 0 (the halo of a 4x4 image)
template <bool ZE>
__device__ __forceinline__ void wino4_in6(float& x0, float& x1, float& x2, float& x3, float& x4, float& x5) {
  const float p = __builtin_fmaf(-4.f, x2, x4), q = __builtin_fmaf(-4.f, x1, x3);
  const float r = x4 - x2, s = x3 - x1;
  float t0, t5;
  if constexpr (ZE) {
    t0 = __builtin_fmaf(-5.f, x2, x4);
    t5 = __builtin_fmaf(-5.f, x3, 4.f * x1);
  } else {
    t0 = __builtin_fmaf(4.f, x0, __builtin_fmaf(-5.f, x2, x4));
    t5 = __builtin_fmaf(4.f, x1, __builtin_fmaf(-5.f, x3, x5));
  }
  x0 = t0; x1 = p + q; x2 = p - q; x3 = __builtin_fmaf(2.f, s, r); x4 = __builtin_fmaf(-2.f, s, r); x5 = t5;
}

__device__ __forceinline__ f32x4 vfma(float k, const f32x4& a, const f32x4& b) {
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = __builtin_fmaf(k, a[i], b[i]);
  return o;
}

// A^T m for one line of six accumulator tiles -> four (10 operations per component)
__device__ __forceinline__ void wino4_out6(const f32x4& m0, const f32x4& m1, const f32x4& m2, const f32x4& m3, const f32x4& m4,
                                           const f32x4& m5, f32x4& y0, f32x4& y1, f32x4& y2, f32x4& y3) {
  const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  y0 = (m0 + s1) + s2;
  y1 = vfma(2.f, d2, d1);
  y2 = vfma(4.f, s2, s1);
  y3 = vfma(8.f, d2, d1) + m5;
}

// T s for one line of four source values behind the x2 upsample, T = B^T P (rows 0, 1, 3, 4, 5; row 2 vanishes): five outputs
// (8 operations); ZE: s0 = s3 = 0 (the halo of a 2x2 source image)
template <bool ZE>
__device__ __forceinline__ void wino4_ups5(float s0, float s1, float s2, float s3, float& t0, float& t1, float& t3, float& t4, float& t5) {
  const float dd = s1 - s2;
  if constexpr (ZE) {
    t0 = __builtin_fmaf(-5.f, s1, s2);
    t5 = __builtin_fmaf(-5.f, s2, 4.f * s1);
  } else {
    t0 = __builtin_fmaf(4.f, s0, __builtin_fmaf(-5.f, s1, s2));
    t5 = __builtin_fmaf(4.f, s1, __builtin_fmaf(-5.f, s2, s3));
  }
  t1 = __builtin_fmaf(-8.f, s1, 2.f * s2);
  t3 = -3.f * dd;
  t4 = dd;
}

// B^T x for one line of six WITHOUT its element 2 (the data gradient behind the upsample multiplies rows 0, 1, 3, 4, 5 only):
// five outputs (11 operations); ZE: x0 = x5 = 0
template <bool ZE>
__device__ __forceinline__ void wino4_in5(float x0, float x1, float x2, float x3, float x4, float x5,
                                          float& t0, float& t1, float& t3, float& t4, float& t5) {
  const float p = __builtin_fmaf(-4.f, x2, x4), q = __builtin_fmaf(-4.f, x1, x3);
  const float r = x4 - x2, s = x3 - x1;
  if constexpr (ZE) {
    t0 = __builtin_fmaf(-5.f, x2, x4);
    t5 = __builtin_fmaf(-5.f, x3, 4.f * x1);
  } else {
    t0 = __builtin_fmaf(4.f, x0, __builtin_fmaf(-5.f, x2, x4));
    t5 = __builtin_fmaf(4.f, x1, __builtin_fmaf(-5.f, x3, x5));
  }
  t1 = p + q; t3 = __builtin_fmaf(2.f, s, r); t4 = __builtin_fmaf(-2.f, s, r);
}

// (Q A^T) m for a line whose element 2 is zero, Q = [1 1 0 0; 0 0 1 1] (the 2x2 sum that is the adjoint of the upsample):
// Q A^T = [1 2 0 3 -1 0; 0 2 0 12 -4 1]: five accumulator tiles -> the two source pixels of the line
__device__ __forceinline__ void wino4_out5q(const f32x4& m0, const f32x4& m1, const f32x4& m3, const f32x4& m4, const f32x4& m5,
                                            f32x4& y0, f32x4& y1) {
  const f32x4 a = m1 + m1;
  y0 = (m0 + a) + vfma(3.f, m3, -m4);
  y1 = (a + m5) + vfma(12.f, m3, -4.f * m4);
}

// A^T m for a line whose element 2 is zero (behind the upsample): five accumulator tiles m0, m1, m3, m4, m5 -> four
__device__ __forceinline__ void wino4_out5(const f32x4& m0, const f32x4& m1, const f32x4& m3, const f32x4& m4, const f32x4& m5,
                                           f32x4& y0, f32x4& y1, f32x4& y2, f32x4& y3) {
  const f32x4 s2 = m3 + m4, d2 = m3 - m4;
  y0 = (m0 + m1) + s2;
  y1 = vfma(2.f, d2, m1);
  y2 = vfma(4.f, s2, m1);
  y3 = vfma(8.f, d2, m1) + m5;
}

// A y for one line of four (the ADJOINT of the output transform: 4x4 gradient tile -> 6x6; 8 operations), in place:
// x0..x3 in, x0..x5 out
__device__ __forceinline__ void wino4_dy6(float& x0, float& x1, float& x2, float& x3, float& x4, float& x5) {
  const float e = x0 + x2, o = x1 + x3;
  const float e2 = __builtin_fmaf(4.f, x2, x0), t = __builtin_fmaf(4.f, x3, x1);
  const float y3 = x3;
  x1 = e + o; x2 = e - o; x3 = __builtin_fmaf(2.f, t, e2); x4 = __builtin_fmaf(-2.f, t, e2); x5 = y3;
}

}  // namespace dvg
