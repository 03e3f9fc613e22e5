// Error string + optional per-kernel HIP-event profiler of libdvg.so.
#include "common.h"
#include <mutex>
#include <string>
#include <vector>

namespace dvg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static const char* kNames[K_COUNT] = {
    "gibbs_sweeps", "grbm_energy", "grbm_suffstats", "gumbel_fwd", "gumbel_bwd", "mmd_prep",
    "mmd_distsum", "mmd_main", "mmd_pm1", "mmd_final", "conv_igemm_kernel<128,64,2,2,1>", "conv_igemm_kernel<64,64,2,2,1>",
    "conv_igemm_kernel<128,32,4,1,1>", "conv_igemm_kernel<32,64,1,2,1>", "conv_igemm_kernel<128,128,2,2,1>", "conv_wgrad_kernel<2,2>", "conv_wgrad_kernel<2,1>", "conv_wgrad_kernel<1,2>",
    "conv_wgrad_kernel<1,1>", "conv_wgrad_fold_kernel",
    "wgrad_reduce", "weight_pack", "bn_finalize", "enc_conv0_fwd", "enc_conv0_wgrad",
    "enc_bn_pool_fwd", "enc_bn_pool_bwd_reduce", "enc_bn_pool_bwd_apply", "enc_proj_fwd",
    "enc_proj_bwd", "dec_bn_act_fwd", "dec_bn_act_bwd_reduce",
    "dec_bn_act_bwd_apply", "dec_conv3_fwd", "dec_conv3_bwd", "dec_final_fwd", "dec_final_bwd",
    "mse", "adam", "misc", "conv_igemm_weight_space", "conv_wino_kernel", "conv_wino_wgrad_kernel",
    "conv_wino4_kernel", "conv_wino4_wgrad_kernel"};

struct EvPair { hipEvent_t a, b; float share = 1.0f; };
static uint64_t g_mask = 0;
static std::mutex g_mu;
static std::vector<EvPair> g_pairs[K_COUNT];
static std::vector<EvPair> g_free;
static double g_ms[K_COUNT];
static int64_t g_n[K_COUNT];
static double g_work[K_COUNT];
static double g_ms_share[K_COUNT];  // sum of duration x the share of the chip's CUs the launch's grid was sized for
static thread_local EvPair g_open[K_COUNT];

bool prof_on(int id) { return (g_mask >> id) & 1ull; }

void prof_begin(int id, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  EvPair p;
  if (!g_free.empty()) { p = g_free.back(); g_free.pop_back(); }
  else { hipEventCreate(&p.a); hipEventCreate(&p.b); }
  hipEventRecord(p.a, s);
  g_open[id] = p;
}

void prof_end(int id, hipStream_t s, double work, float share) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_work[id] += work;
  g_open[id].share = share;
  hipEventRecord(g_open[id].b, s);
  g_pairs[id].push_back(g_open[id]);
}

static void drain(int id) {
  for (auto& p : g_pairs[id]) {
    hipEventSynchronize(p.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { g_ms[id] += ms; g_ms_share[id] += (double)ms * p.share; g_n[id] += 1; }
    g_free.push_back(p);
  }
  g_pairs[id].clear();
}

}  // namespace dvg

using namespace dvg;

extern "C" {

int dvg_version(void) { return 110; /* 0.1.1 */ }
#ifndef DVG_SRC_HASH
#error "DVG_SRC_HASH must be defined by the build (image-generation_amd/Makefile)"
#endif
const char* dvg_source_hash(void) { return DVG_SRC_HASH; }
const char* dvg_last_error(void) { return g_err; }

int dvg_prof_enable(uint64_t kernel_mask) { g_mask = kernel_mask; return DVG_OK; }
int dvg_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < K_COUNT; ++i) { drain(i); g_ms[i] = 0; g_ms_share[i] = 0; g_n[i] = 0; g_work[i] = 0; }
  return DVG_OK;
}
int dvg_prof_num_kernels(void) { return K_COUNT; }
const char* dvg_prof_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kNames[id] : ""; }
int dvg_prof_query(int id, double* total_ms, int64_t* launches) {
  if (id < 0 || id >= K_COUNT) { set_error("bad kernel id %d", id); return DVG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g_mu);
  drain(id);
  if (total_ms) *total_ms = g_ms[id];
  if (launches) *launches = g_n[id];
  return DVG_OK;
}
int dvg_prof_query_share(int id, double* share_ms) {
  if (id < 0 || id >= K_COUNT || !share_ms) { set_error("bad kernel id %d", id); return DVG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g_mu);
  drain(id);
  *share_ms = g_ms_share[id];
  return DVG_OK;
}
int dvg_prof_query_work(int id, double* work) {
  if (id < 0 || id >= K_COUNT || !work) { set_error("bad kernel id %d", id); return DVG_E_INVALID; }
  std::lock_guard<std::mutex> lk(g_mu);
  *work = g_work[id];
  return DVG_OK;
}

}  // extern "C"
