// Process-wide arithmetic mode of the forward / data-gradient convolution GEMMs.
//   0 (default)  float32 operands on the f32 MFMA: the <= 1e-5 relative loss parity of the north star.
//   1            "bf16 GEMM inputs, f32 accumulate" (BASELINE.json configs[1]): activations and weights
//                are rounded to bf16 (round to nearest even) on their way into LDS, products and sums stay f32
//                (v_mfma_f32_32x32x16_bf16).  Weight gradients, BatchNorm, losses, Adam stay float32.
//   2            float32 operands carried as THREE bf16 pieces each (x = hi + mid + lo exactly: 3 x 8 significand bits),
//                multiplied as the six piece products down to 2^-16 (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid) on
//                v_mfma_f32_32x32x16_bf16 with f32 accumulation: every piece product is exact, what is dropped is below
//                2^-23 of a product -- float32-class arithmetic at 6/16 of the f32 MFMA's matrix time.
// A forward call and its backward call must run in the same mode (the backward reuses the forward's weight packs).
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "conv.h"

namespace dvg {
bool conv_precision_bf16();
int conv_precision_mode();
namespace {
std::atomic<int> g_mode{-1};
std::mutex g_ws_mutex;
std::unordered_map<const void*, int> g_ws_mode;  // mode of the last forward call per workspace
std::unordered_map<const void*, uint32_t> g_ws_plan;  // plan signature of the last forward call per workspace
}

// The same bookkeeping for everything else that shapes what a forward call leaves in its workspace (pack formats, which
// of the decoder's algebraic forms ran): options may be flipped between calls (dvg_set_option), and a backward call that
// would read buffers its forward never wrote must fail loudly instead.
void plan_note_forward(const void* ws, uint32_t signature) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  if (g_ws_plan.size() > 4096) g_ws_plan.clear();
  g_ws_plan[ws] = signature;
}
bool plan_matches_forward(const void* ws, uint32_t signature) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  auto it = g_ws_plan.find(ws);
  return it == g_ws_plan.end() || it->second == signature;
}

// true when no forward was noted for `ws` (nothing to contradict) or the noted signature has `bit` set
bool plan_forward_flag(const void* ws, uint32_t bit) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  auto it = g_ws_plan.find(ws);
  return it == g_ws_plan.end() || (it->second & bit) != 0;
}

// A backward call reuses the weight packs its forward call left in the workspace, so it must run in the mode that wrote
// them: the forward notes its mode per workspace pointer, the backward checks it (host-side bookkeeping only).
void conv_precision_note_forward(const void* ws) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  if (g_ws_mode.size() > 4096) g_ws_mode.clear();
  g_ws_mode[ws] = conv_precision_mode();
}
bool conv_precision_matches_forward(const void* ws) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  auto it = g_ws_mode.find(ws);
  return it == g_ws_mode.end() || it->second == conv_precision_mode();
}

int conv_precision_mode() {
  int m = g_mode.load(std::memory_order_relaxed);
  if (m < 0) {  // never set: float32 (the mode is chosen through dvg_set_conv_precision, not the environment)
    m = 0;
    g_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}
bool conv_precision_bf16() { return conv_precision_mode() == 1; }
}  // namespace dvg

extern "C" int dvg_set_conv_precision(int mode) {
  if (mode != DVG_PRECISION_F32 && mode != DVG_PRECISION_BF16_INPUTS && mode != DVG_PRECISION_F32_SPLIT3) {
    dvg::set_error("dvg_set_conv_precision: unknown mode %d", mode);
    return DVG_E_INVALID;
  }
  dvg::g_mode.store(mode, std::memory_order_relaxed);
  return DVG_OK;
}

extern "C" int dvg_get_conv_precision(void) { return dvg::conv_precision_mode(); }
