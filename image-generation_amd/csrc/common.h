// Shared plumbing for libdvg.so: error reporting, launch + optional event profiling.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dvg.h"

namespace dvg {

void set_error(const char* fmt, ...);

#define DVG_CHECK_HIP(expr)                                                            \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      dvg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                     __LINE__);                                                        \
      return DVG_E_HIP;                                                                \
    }                                                                                  \
  } while (0)

#define DVG_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      dvg::set_error(__VA_ARGS__);        \
      return DVG_E_INVALID;               \
    }                                     \
  } while (0)

#define DVG_TRY(expr)          \
  do {                         \
    int _rc = (expr);          \
    if (_rc != DVG_OK) return _rc; \
  } while (0)

// ---- kernel ids for the profiler (names in prof.cpp must stay in sync) ----
enum KernelId : int {
  K_GIBBS = 0,
  K_GRBM_ENERGY,
  K_GRBM_SUFFSTATS,
  K_GUMBEL_FWD,
  K_GUMBEL_BWD,
  K_MMD_PREP,
  K_MMD_DISTSUM,
  K_MMD_MAIN,
  K_MMD_PM1,
  K_MMD_FINAL,
  K_IGEMM_128x64,   // conv_igemm_kernel<128,64,2,2>
  K_IGEMM_64x64,    // conv_igemm_kernel<64,64,2,2>
  K_IGEMM_128x32,   // conv_igemm_kernel<128,32,4,1>
  K_IGEMM_32x64,    // conv_igemm_kernel<32,64,1,2>
  K_IGEMM_128x128,  // conv_igemm_kernel<128,128,2,2>
  K_WGRAD_2x2,      // conv_wgrad_kernel<2,2>
  K_WGRAD_2x1,
  K_WGRAD_1x2,
  K_WGRAD_1x1,
  K_WGRAD_FOLD,     // conv_wgrad_fold_kernel
  K_WGRAD_REDUCE,
  K_WEIGHT_PACK,
  K_BN_FINALIZE,
  K_ENC_CONV0_FWD,
  K_ENC_CONV0_WGRAD,
  K_ENC_BN_POOL_FWD,
  K_ENC_BN_POOL_BWD_REDUCE,
  K_ENC_BN_POOL_BWD_APPLY,
  K_ENC_PROJ_FWD,
  K_ENC_PROJ_BWD,
  K_DEC_BN_ACT_FWD,
  K_DEC_BN_ACT_BWD_REDUCE,
  K_DEC_BN_ACT_BWD_APPLY,
  K_DEC_CONV3_FWD,
  K_DEC_CONV3_BWD,
  K_DEC_FINAL_FWD,
  K_DEC_FINAL_BWD,
  K_MSE,
  K_ADAM,
  K_MISC,
  K_IGEMM_WSPACE,   // conv_igemm_kernel launches on weights only (the composed decoder layers' weight-space products)
  K_IGEMM_WINO,     // conv_wino8_kernel (profiler name "conv_wino_kernel"): Winograd F(2x2,3x3) form of the encoder's 3x3 layers (work = executed FLOPs: 4/9 of the direct form's)
  K_WGRAD_WINO,     // conv_wino_wgrad8_kernel (profiler name "conv_wino_wgrad_kernel"): Winograd form of the encoder's 3x3 weight gradients (work = executed FLOPs)
  K_IGEMM_WINO4,    // conv_wino4_kernel: Winograd F(4x4,3x3) form of the encoder's 3x3 layers (work = executed FLOPs: 1/4 of the direct form's)
  K_WGRAD_WINO4,    // conv_wino4_wgrad_kernel: F(4x4,3x3) form of the encoder's 3x3 weight gradients (work = executed FLOPs)
  K_COUNT
};

// Kernel-form options (options.cpp): set through dvg_set_option, read per call.
enum Opt : int {
  OPT_IGEMM_DMA = 0, OPT_IGEMM_POSMAJOR, OPT_IGEMM_THR128, OPT_WGRAD_DMA, OPT_DEC_FOLD, OPT_DEC_D22, OPT_DEC_LC0, OPT_MMD_W128,
  OPT_MMD_D256, OPT_GIBBS_GENERIC, OPT_SIDE_STREAM, OPT_ENC_WINO, OPT_DEC_WINO, OPT_ENC_L0_FUSED, OPT_DEC_TAIL_FUSED,
  OPT_WINO_DYNAMIC, OPT_ENC_WINO4, OPT_ENC_WINO4_MASK, OPT_ENC_DGRAD_CUS, OPT_ENC_WGRAD_CUS, OPT_DEC_WINO4_MASK, OPT_WGRAD_REDUCE_TILED, OPT_ENC_BN_REDUCE_POOLED, OPT_COUNT
};
int64_t opt(Opt id);

// Fork/join helpers (streams.cpp).  side_stream(s) is the library's side stream of the current device (or `s` itself
// when option side_stream = 0: dvg_set_option("side_stream", 0)); stream_order_after(w, p) makes everything enqueued on `w` from now on wait for what
// has been enqueued on `p` so far (no-op when w == p).  Capture-safe.
bool side_enabled();
void side_stream_warm();  // create the current device's side stream + event ring now (outside any capture)
int* dyn_tile_counters();  // 32 zeroed ints for one dynamically scheduled launch (streams.cpp), or nullptr
hipStream_t side_stream(hipStream_t fallback);
int stream_order_after(hipStream_t waiter, hipStream_t producer);
// Weight-only prologues (streams.cpp): the per-workspace mark between dvg_decoder_prepare and the forward call that
// consumes it.
int prep_arm(const void* ws, uint64_t sig, hipStream_t producer);
int prep_join(const void* ws, uint64_t sig, hipStream_t waiter, bool* armed, bool* matched);
// The same in two halves: mark a point of `producer` now, make `waiter` wait for exactly that point later.
int stream_mark(hipStream_t producer, hipEvent_t* mark);
int stream_wait_mark(hipStream_t waiter, hipEvent_t mark);

// Profiler hooks (prof.cpp).  begin/end record a hipEvent pair on `s` when enabled.
bool prof_on(int id);
void prof_begin(int id, hipStream_t s);
void prof_end(int id, hipStream_t s, double work, float share);

// `work`: algorithmic work of the launch (FLOPs for the GEMM kernels), summed per kernel id.  `share`: the fraction of the
// chip's CUs the launch's grid was SIZED for (persistent whole-CU grids that are given a CU budget because another
// kernel holds the rest: conv_wino*.hip); the profiler sums duration x share beside the plain duration
// (dvg_prof_query_share), so a kernel that is handed half the chip is also priced against half the chip's peak.
struct ProfScope {
  int id; hipStream_t s; bool on; double work; float share;
  ProfScope(int id_, hipStream_t s_, double work_ = 0.0, float share_ = 1.0f) : id(id_), s(s_), on(prof_on(id_)), work(work_), share(share_) {
    if (on) prof_begin(id, s);
  }
  ~ProfScope() { if (on) prof_end(id, s, work, share); }
};

// Launch helper: kernel<<<grid, block, shmem, stream>>>(args...) wrapped in a profiler scope,
// returning DVG_E_HIP from the enclosing function on a launch error.
#define DVG_LAUNCH(id, kernel, grid, block, shmem, stream, ...) \
  DVG_LAUNCH_WORK(id, 0.0, kernel, grid, block, shmem, stream, __VA_ARGS__)

#define DVG_LAUNCH_WORK(id, work, kernel, grid, block, shmem, stream, ...) \
  DVG_LAUNCH_WORK_SHARE(id, work, 1.0f, kernel, grid, block, shmem, stream, __VA_ARGS__)

#define DVG_LAUNCH_WORK_SHARE(id, work, share, kernel, grid, block, shmem, stream, ...) \
  do {                                                                                  \
    dvg::ProfScope _ps((id), (stream), (work), (share));                                \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                \
    hipError_t _le = hipGetLastError();                                                 \
    if (_le != hipSuccess) {                                                            \
      dvg::set_error("launch of %s failed: %s (%s:%d)", #kernel, hipGetErrorString(_le), \
                     __FILE__, __LINE__);                                               \
      return DVG_E_HIP;                                                                 \
    }                                                                                   \
  } while (0)

// The raised dynamic-LDS limit of a kernel (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-DEVICE attribute: set it
// once per device and call site (`done`: a static bit set per device ordinal), not once per process.
static inline int raise_dynamic_lds(std::atomic<uint64_t>& done, const void* kern, int bytes) {
  int dev = 0;
  DVG_CHECK_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    DVG_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(bit, std::memory_order_release);
  }
  return DVG_OK;
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

}  // namespace dvg
