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
struct OptDef { const char* name; int64_t def; const char* doc; bool dev = false; };  // dev: reached through include/dvg_dev.h only
// (order = enum Opt in common.h)
const OptDef kDefs[OPT_COUNT] = {
    {"igemm_dma", 1, "float32 forward / data-gradient GEMM: 1 LDS-DMA staging (default), 0 register staging (tensors of 4 GiB and more always; A/B reference)", true},
    {"igemm_posmajor", 1, "position-major tiles (padding taps of small images never multiplied): 1 on, 0 pixel-major tiles"},
    {"igemm_thr128", 512, "blocks a launch must have for the 128x128 tile (the tests lower it to send small fixtures through that tile)", true},
    {"wgrad_dma", 1, "weight-gradient GEMMs: 1 LDS-DMA staging (default), 0 register staging (tensors of 4 GiB and more always)", true},
    {"dec_fold", 1, "decoder: Upsample(x2) + ConvTranspose as 4 class GEMMs with pre-summed taps (4/9 of the FLOPs)"},
    {"dec_d22", 1, "decoder: first ConvTranspose layer on 2x2 images as one dense map per image (16/36 of the FLOPs)"},
    {"dec_lc0", -1, "decoder: Linear composed with that map: -1 from 4096 rows up or rows x n_latents >= 2^20 (default), 0 never, 1 always"},
    {"mmd_w128", -1, "MMD: 128-row-block pair kernel: -1 by problem size (default), 0 never, 1 whenever the shape allows", true},
    {"mmd_d256", -1, "MMD: 256-row distance-sum kernel: -1 by problem size (default), 0 never, 1 whenever the shape allows", true},
    {"gibbs_generic", 0, "sampler: 1 forces the plain rolled reference schedule (no chains-side-by-side form for large graphs), 2 the lane-major schedule one row at a time instead of two passes of a class side by side, 3 the lane-major schedule whatever the size, 4 the chains-side-by-side schedule in 8-chain workgroups whatever the size (all bit-identical; A/B references)", true},
    {"side_stream", 1, "weight-gradient chains on the library's side stream (0 serialises everything on the caller's)"},
    {"enc_wino", -1, "encoder 3x3 layers in the Winograd F(2x2,3x3) form, forward, data gradient and weight gradient: -1 by size (default: evaluation-mode forward launches of 256 workgroups' worth of tiles or more, training launches of 512 or more), 0 never, 1 every launch the shape allows; forward / data gradient never in the bf16-input mode"},
    {"dec_wino", -1, "decoder Upsample(x2) + 3x3 layers in the Winograd form (9 of 16 transform positions), forward, data gradient and weight gradient: -1 from 8192 decoder rows up (default), 0 never, 1 whenever the shape allows"},
    {"enc_l0_fused", 1, "encoder layer 0: 1 = its output is recomputed by every pass that needs it (BatchNorm statistics, BN/pool/LeakyReLU, the backward sums, the weight gradient) and never stored (default), 0 = stored and re-read (rounds 1-2)", true},
    {"dec_tail_fused", 1, "decoder: 1 = the 8x8x32 stage's BatchNorm / Dropout / LeakyReLU (forward and backward) run inside the 32 -> 1 layer's kernels, its activated map and that map's gradient are never stored (default), 0 = separate passes (rounds 1-2)", true},
    {"wino_dynamic", 1, "Winograd forward / data-gradient launches deal their tile blocks dynamically (an atomic counter per grid row) instead of round-robin: a workgroup that gets its CU late takes fewer blocks (1 default, 0 = the static deal of rounds 3-4)", true},
    {"enc_wino4", 1, "encoder 3x3 layers that run in the Winograd domain: 1 (default) forward and data gradient in the F(4x4,3x3) form wherever the shape qualifies (36 position GEMMs per 4x4 output tile: 2.25 multiplies per output instead of 4.0), 0 the F(2x2,3x3) form everywhere"},
    {"enc_wino4_mask", 0x13c, "which encoder launches option enc_wino4 covers: bit l-1 the forward of layer l (1..3), bit 2+l its data gradient, bit 5+l its weight gradient (default 0x13c: layer 3's forward, every data gradient, layer 3's weight gradient -- the forwards of layers 1-2 gain nothing inside the step and their 7x float32 noise in front of two more max-pool stages moved the B = 1024 full-size gradient test from 4.6e-3 to 5.4e-3 of its 5e-3 bar; the weight gradients of layers 1-2 are level with the F(2x2,3x3) kernel)", true},
    {"enc_dgrad_cus", 0, "CUs the encoder's Winograd data-gradient launches are sized for (0: the measured constant of conv.h)", true},
    {"enc_wgrad_cus", 0, "CUs the encoder's Winograd weight-gradient launches are sized for (0: the measured constant of conv.h)", true},
    {"dec_wino4_mask", 0x2, "which of the decoder's Winograd launches behind the upsample take the F(4x4,3x3) form with 25 of 36 positions (option enc_wino4 != 0): bit 0 / 1 the forward of the 128 -> 64 / 64 -> 32 layer (bit 0 off: with it the full-size decoder's reconstruction is 2.97e-6 from float64, the bar is 2e-6), bit 2 / 3 their data gradients", true},
    {"wgrad_reduce_tiled", 1, "weight-gradient slab sums of at most 8 slabs and at least 256 tiles by the tiled one-thread-per-element kernels (same bits; 0: the 8-lanes-per-element kernels everywhere)", true},
    {"enc_bn_reduce_pooled", 1, "encoder BatchNorm/pool backward: the (sum dz, sum dz zhat) pass reads the pooled activations the forward kept (2 floats per element) instead of the four pre-BatchNorm outputs of every window (5); 0: the window form", true},
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

namespace {
// index of the i-th option of one kind (public: include/dvg.h; dev: include/dvg_dev.h), or -1
int nth(bool dev, int i) {
  for (int k = 0; k < OPT_COUNT; ++k)
    if (kDefs[k].dev == dev && i-- == 0) return k;
  return -1;
}
int count(bool dev) {
  int n = 0;
  for (int k = 0; k < OPT_COUNT; ++k) n += kDefs[k].dev == dev;
  return n;
}
int set_opt(bool dev, const char* what, const char* name, int64_t value) {
  DVG_REQUIRE(name, "%s: null name", what);
  init_once();
  for (int i = 0; i < OPT_COUNT; ++i)
    if (kDefs[i].dev == dev && !strcmp(name, kDefs[i].name)) { g_val[i].store(value, std::memory_order_relaxed); return DVG_OK; }
  set_error("%s: unknown option '%s'", what, name);
  return DVG_E_INVALID;
}
int get_opt(bool dev, const char* what, const char* name, int64_t* value) {
  DVG_REQUIRE(name && value, "%s: null argument", what);
  init_once();
  for (int i = 0; i < OPT_COUNT; ++i)
    if (kDefs[i].dev == dev && !strcmp(name, kDefs[i].name)) { *value = g_val[i].load(std::memory_order_relaxed); return DVG_OK; }
  set_error("%s: unknown option '%s'", what, name);
  return DVG_E_INVALID;
}
}  // namespace

// The product's switches (include/dvg.h): eight.
extern "C" int dvg_option_count(void) { return count(false); }
extern "C" const char* dvg_option_name(int index) { const int k = nth(false, index); return k >= 0 ? kDefs[k].name : nullptr; }
extern "C" const char* dvg_option_doc(int index) { const int k = nth(false, index); return k >= 0 ? kDefs[k].doc : nullptr; }
extern "C" int dvg_set_option(const char* name, int64_t value) { return set_opt(false, "dvg_set_option", name, value); }
extern "C" int dvg_get_option(const char* name, int64_t* value) { return get_opt(false, "dvg_get_option", name, value); }

// A/B references and test knobs (include/dvg_dev.h): the forms a default replaced, kept so that the tests can hold the
// product's form to them bit for bit.  Not part of the drop-in boundary.
extern "C" int dvg_dev_option_count(void) { return count(true); }
extern "C" const char* dvg_dev_option_name(int index) { const int k = nth(true, index); return k >= 0 ? kDefs[k].name : nullptr; }
extern "C" const char* dvg_dev_option_doc(int index) { const int k = nth(true, index); return k >= 0 ? kDefs[k].doc : nullptr; }
extern "C" int dvg_dev_set_option(const char* name, int64_t value) { return set_opt(true, "dvg_dev_set_option", name, value); }
extern "C" int dvg_dev_get_option(const char* name, int64_t* value) { return get_opt(true, "dvg_dev_get_option", name, value); }

extern "C" int dvg_reset_options(void) {  // (both kinds)
  init_once();
  for (int i = 0; i < OPT_COUNT; ++i) g_val[i].store(kDefs[i].def, std::memory_order_relaxed);
  return DVG_OK;
}
