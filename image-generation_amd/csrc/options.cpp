// Kernel-form options of libdvg.so: ONE registry, set through the C ABI (dvg_set_option / dvg_get_option, include/dvg.h).
// Rounds 1-2 selected kernel forms through ~26 environment variables read at scattered getenv() sites, some once per
// process, some per call; the boundary, not the environment, selects them now.  Every option has a default that is the
// product's path; the others exist for A/B measurements and for the tests that compare two forms of the same function.
// Values are read per call (relaxed atomics): an option may be flipped between calls, and the forward / backward pair of
// a network call records the plan-shaping ones in its workspace (decoder.cpp) so that a flip in between is an error,
// not silent garbage.
#include <atomic>
#include <cstring>
#include <mutex>

#include "common.h"

namespace dvg {
namespace {
struct OptDef { const char* name; int64_t def; const char* doc; };
// (order = enum Opt in common.h)
const OptDef kDefs[OPT_COUNT] = {
    {"igemm_dma", 1, "float32 forward / data-gradient GEMM: 1 LDS-DMA staging (default), 0 register staging (A/B reference)"},
    {"igemm_posmajor", 1, "position-major tiles (padding taps of small images never multiplied): 1 on, 0 pixel-major tiles"},
    {"igemm_thr128", 512, "blocks a launch must have for the 128x128 tile"},
    {"igemm_thr64", 512, "blocks a launch must have for the 128x64 tile"},
    {"igemm_thr32", 96, "blocks below which the 32x64 tile replaces the 64x64 tile"},
    {"igemm_no32", 0, "1: never the 32x64 tile"},
    {"wgrad_dma", 1, "weight-gradient GEMMs: 1 LDS-DMA staging (default), 0 register staging"},
    {"dec_fold", 1, "decoder: Upsample(x2) + ConvTranspose as 4 class GEMMs with pre-summed taps (4/9 of the FLOPs)"},
    {"dec_d22", 1, "decoder: first ConvTranspose layer on 2x2 images as one dense map per image (16/36 of the FLOPs)"},
    {"dec_lc0", -1, "decoder: Linear composed with that map: -1 from 4096 rows up (default), 0 never, 1 always"},
    {"mmd_w128", -1, "MMD: 128-row-block pair kernel: -1 by problem size (default), 0 never, 1 whenever the shape allows"},
    {"mmd_d256", -1, "MMD: 256-row distance-sum kernel: -1 by problem size (default), 0 never, 1 whenever the shape allows"},
    {"mmd_blocks", 256, "MMD: target workgroup count of the pair kernels' column split"},
    {"gibbs_generic", 0, "sampler: 1 forces the rolled reference schedule (bit-identical; A/B reference)"},
    {"gibbs_waves", 0, "sampler: waves per workgroup (0 = chosen by graph size)"},
    {"gibbs_waves_per_chain", 1, "sampler: 2 = two waves per chain where colour classes hold 65..128 spins and chains are few (the faster draw ALONE: generation; neutral inside a training step), 1 = one (default)"},
    {"side_stream", 1, "weight-gradient chains on the library's side stream (0 serialises everything on the caller's)"},
    {"enc_wino", -1, "encoder 3x3 layers in the Winograd F(2x2,3x3) form: -1 by size (default: evaluation-mode forward launches of 256 workgroups' worth of tiles or more, training launches of wino_min_blocks or more), 0 never, 1 every launch the shape allows, 2 / 3 forward / data-gradient launches only; never in the bf16-input mode"},
    {"enc_wino_mask", 0, "A/B: when non-zero, picks the Winograd form per launch instead of enc_wino: bit l-1 = forward of layer l (1..3), bit 2+l = its data gradient"},
    {"enc_l0_fused", 1, "encoder layer 0: 1 = its output is recomputed by every pass that needs it (BatchNorm statistics, BN/pool/LeakyReLU, both backward passes, the weight gradient) and never stored (default; 2 = the same with the backward in two passes), 0 = stored and re-read (rounds 1-2)"},
    {"dec_tail_fused", 1, "decoder: 1 = the 8x8x32 stage's BatchNorm / Dropout / LeakyReLU (forward and backward) run inside the 32 -> 1 layer's kernels, its activated map and that map's gradient are never stored (default), 0 = separate passes (rounds 1-2)"},
    {"enc_wino_cus", 256, "Winograd FORWARD launches of a training call: CUs the persistent grid is sized for -- a workgroup needs a whole CU, and the step's sampler draw, enqueued first, keeps its own (ModelWrapper sets 256 - the draw's workgroups, dvg_gibbs_launch_info); read only under wino_dynamic = 0"},
    {"enc_wino_cus_d", 128, "the same for the data-gradient launches, which share the chip with the weight-gradient chain on the side stream (measured at c3: 128 -> 10.16 ms, 192 -> 10.23, 256 -> 10.6)"},
    {"enc_wino_wgrad", -1, "encoder 3x3 WEIGHT gradients in the Winograd form (conv_wino_wgrad.hip): -1 with the other training launches (default), 0 never, 1 whenever the shape allows"},
    {"enc_wino_cus_w", 128, "CUs the Winograd weight-gradient launches are sized for (whole-CU workgroups; the data-gradient chain runs beside them; measured at c3 with enc_wino_cus_d: (128, 128) 9.40 ms, (160, 96) 9.64, (192, 64) 10.09, (192, 128) 9.41, (256, 128) 9.45; a budget of its own for layer 1's launch, the last of the step: neutral)"},
    {"dec_wino_wgrad", -1, "decoder Upsample(x2) + 3x3 layers: weight gradient in the Winograd form (9 of 16 transform positions; conv_wino_wgrad.hip): -1 from 8192 decoder rows up, 0 never, 1 whenever the shape allows"},
    {"dec_wino_cus_w", 256, "CUs those launches are sized for"},
    {"dec_wino", -1, "decoder Upsample(x2) + 3x3 layers in the Winograd form (9 of 16 transform positions; conv_wino.hip): -1 forward and data-gradient launches from 8192 decoder rows up (default), 0 never, 1 whenever the shape allows, 2 / 3 forward / data gradient only"},
    {"dec_wino_cus", 256, "CUs the decoder's Winograd forward launches are sized for"},
    {"dec_wino_cus_d", 256, "... its data-gradient launches"},
    {"wino_min_blocks", 512, "encoder Winograd launches of a training call (enc_wino = -1): from this many workgroups' worth of tiles up (measured, n = 512 model: 1024 / 512 / 256 -> B = 512: 2.16 / 2.13 / 2.06 ms, B = 1024: 3.03 / 2.82 / 2.82, B = 2048: 4.67 / 4.60 / 4.53; c2: 0.920 / 0.924 / 0.953 -- 512 is the lowest value that costs c2 nothing)"},
    {"wino_dynamic", 1, "Winograd forward / data-gradient launches deal their tile blocks dynamically (an atomic counter per grid row) instead of round-robin: a workgroup that gets its CU late takes fewer blocks (1 default, 0 = the static deal of rounds 3-4)"},
};
std::atomic<int64_t> g_val[OPT_COUNT];
std::once_flag g_once;
void init_once() {  // a real once: a racing dvg_set_option can no longer be overwritten by a late default store
  std::call_once(g_once, [] {
    for (int i = 0; i < OPT_COUNT; ++i) g_val[i].store(kDefs[i].def, std::memory_order_relaxed);
  });
}
}  // namespace

int64_t opt(Opt id) {
  init_once();
  return g_val[id].load(std::memory_order_relaxed);
}
}  // namespace dvg

using namespace dvg;

extern "C" int dvg_option_count(void) { return OPT_COUNT; }

extern "C" const char* dvg_option_name(int index) { return index >= 0 && index < OPT_COUNT ? kDefs[index].name : nullptr; }

extern "C" const char* dvg_option_doc(int index) { return index >= 0 && index < OPT_COUNT ? kDefs[index].doc : nullptr; }

extern "C" int dvg_set_option(const char* name, int64_t value) {
  DVG_REQUIRE(name, "dvg_set_option: null name");
  init_once();
  for (int i = 0; i < OPT_COUNT; ++i)
    if (!strcmp(name, kDefs[i].name)) { g_val[i].store(value, std::memory_order_relaxed); return DVG_OK; }
  set_error("dvg_set_option: unknown option '%s'", name);
  return DVG_E_INVALID;
}

extern "C" int dvg_get_option(const char* name, int64_t* value) {
  DVG_REQUIRE(name && value, "dvg_get_option: null argument");
  init_once();
  for (int i = 0; i < OPT_COUNT; ++i)
    if (!strcmp(name, kDefs[i].name)) { *value = g_val[i].load(std::memory_order_relaxed); return DVG_OK; }
  set_error("dvg_get_option: unknown option '%s'", name);
  return DVG_E_INVALID;
}

extern "C" int dvg_reset_options(void) {
  init_once();
  for (int i = 0; i < OPT_COUNT; ++i) g_val[i].store(kDefs[i].def, std::memory_order_relaxed);
  return DVG_OK;
}
