// MFMA tile helpers shared by the convolution translation units (conv_igemm.hip, conv_wgrad.hip).
#pragma once
#include "conv.h"

namespace dvg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // register-resident (a float4 array can end up in scratch)

__device__ __forceinline__ int crow16(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

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
