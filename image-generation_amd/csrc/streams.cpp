// Library-owned side stream for fork/join concurrency inside one C-ABI call.
//
// The backward passes have two independent chains per layer: the data gradient (critical path: it feeds the next
// layer's BN backward) and the weight gradient (only the optimizer reads it).  The weight-gradient chain is enqueued
// on a side stream behind an event of the caller's stream and joined back before the call returns, so to the caller
// the call is still "a sequence of work on `stream`".  The pattern is capture-safe: under hipStreamBeginCapture the
// event wait pulls the side stream into the capture and the join closes the fork, so the captured graph simply gets
// two parallel branches.  Option side_stream = 0 keeps everything on the caller's stream (A/B measurements, debugging).
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace dvg {

namespace {
constexpr int kMaxDevices = 16;
constexpr int kEvents = 64;
// The library is entered from the caller's thread (forward) and from the autograd engine's worker thread (backward):
// creation is serialised by a mutex and published through an acquire/release flag, the event-ring cursor is atomic.
constexpr int kDynSets = 32, kDynInts = 32;
struct SideCtx {
  hipStream_t side = nullptr;
  int* dyn = nullptr;  // [kDynSets][kDynInts] zero-initialised tile counters of the dynamically scheduled persistent grids
  std::atomic<unsigned> dyn_next{0};
  hipEvent_t ev[kEvents] = {};
  std::atomic<unsigned> next{0};
  std::atomic<bool> ready{false};
  std::mutex mu;
};
SideCtx g_ctx[kMaxDevices];

SideCtx* ctx() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  SideCtx& c = g_ctx[dev];
  if (c.ready.load(std::memory_order_acquire)) return &c;
  std::lock_guard<std::mutex> lock(c.mu);
  if (c.ready.load(std::memory_order_relaxed)) return &c;
  // what an earlier, partially failed attempt created is reused, not leaked
  if (!c.side && hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess) { c.side = nullptr; return nullptr; }
  for (int i = 0; i < kEvents; ++i)
    if (!c.ev[i] && hipEventCreateWithFlags(&c.ev[i], hipEventDisableTiming) != hipSuccess) { c.ev[i] = nullptr; return nullptr; }
  if (!c.dyn) {
    if (hipMalloc((void**)&c.dyn, sizeof(int) * kDynSets * kDynInts) != hipSuccess) { c.dyn = nullptr; return nullptr; }
    if (hipMemset(c.dyn, 0, sizeof(int) * kDynSets * kDynInts) != hipSuccess) { (void)hipFree(c.dyn); c.dyn = nullptr; return nullptr; }
  }
  c.ready.store(true, std::memory_order_release);
  return &c;
}
}  // namespace

bool side_enabled();

// Creates the device's side stream and event ring now (never under a stream capture: the workspace-size queries and
// dvg_graph_create call this, and they always precede the first captured step).
void side_stream_warm() { (void)ctx(); }

// A set of zeroed tile counters for ONE launch of a dynamically scheduled persistent grid (conv_wino.hip): the kernel
// leaves the set zeroed again (its last workgroup resets it), sets are handed out round-robin (32 of them) so that launches in
// flight on different streams do not share one.  nullptr before side_stream_warm() has run on this device (never under a capture).
int* dyn_tile_counters() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  SideCtx& c = g_ctx[dev];
  if (!c.ready.load(std::memory_order_acquire) || !c.dyn) return nullptr;
  return c.dyn + (c.dyn_next.fetch_add(1, std::memory_order_relaxed) % kDynSets) * kDynInts;
}

bool side_enabled() { return opt(OPT_SIDE_STREAM) != 0; }

hipStream_t side_stream(hipStream_t fallback) {
  if (!side_enabled()) return fallback;
  SideCtx* c = ctx();
  return c ? c->side : fallback;
}

// Per-workspace mark of a prologue in flight (dvg_decoder_prepare -> dvg_decoder_fwd_ex): its own event (the ring above is
// reused after kEvents marks, and a whole encoder pass lies between the two calls) and the signature of what was prepared.
namespace {
struct PrepMark { hipEvent_t ev = nullptr; uint64_t sig = 0; bool armed = false; };
std::mutex g_prep_mu;
std::unordered_map<const void*, PrepMark> g_prep;
}

int prep_arm(const void* ws, uint64_t sig, hipStream_t producer) {
  std::lock_guard<std::mutex> lock(g_prep_mu);
  if (g_prep.size() > 256 && g_prep.find(ws) == g_prep.end()) {  // bounded: drop the marks nobody is waiting for
    for (auto it = g_prep.begin(); it != g_prep.end();) {
      if (!it->second.armed) { if (it->second.ev) (void)hipEventDestroy(it->second.ev); it = g_prep.erase(it); }
      else ++it;
    }
  }
  PrepMark& m = g_prep[ws];
  if (!m.ev) DVG_CHECK_HIP(hipEventCreateWithFlags(&m.ev, hipEventDisableTiming));
  DVG_CHECK_HIP(hipEventRecord(m.ev, producer));
  m.sig = sig;
  m.armed = true;
  return DVG_OK;
}

// If a prologue is in flight for `ws`: `waiter` waits for it and the mark is consumed.  *armed / *matched report whether
// there was one and whether it was prepared for `sig`.
int prep_join(const void* ws, uint64_t sig, hipStream_t waiter, bool* armed, bool* matched) {
  std::lock_guard<std::mutex> lock(g_prep_mu);
  *armed = *matched = false;
  auto it = g_prep.find(ws);
  if (it == g_prep.end() || !it->second.armed) return DVG_OK;
  *armed = true;
  *matched = it->second.sig == sig;
  it->second.armed = false;
  DVG_CHECK_HIP(hipStreamWaitEvent(waiter, it->second.ev, 0));
  return DVG_OK;
}

int stream_mark(hipStream_t producer, hipEvent_t* mark) {
  SideCtx* c = ctx();
  DVG_REQUIRE(c, "side stream: no context for this device");
  hipEvent_t e = c->ev[c->next.fetch_add(1, std::memory_order_relaxed) % kEvents];
  DVG_CHECK_HIP(hipEventRecord(e, producer));
  *mark = e;
  return DVG_OK;
}

int stream_wait_mark(hipStream_t waiter, hipEvent_t mark) {
  DVG_CHECK_HIP(hipStreamWaitEvent(waiter, mark, 0));
  return DVG_OK;
}

int stream_order_after(hipStream_t waiter, hipStream_t producer) {
  if (waiter == producer) return DVG_OK;
  hipEvent_t e;
  DVG_TRY(stream_mark(producer, &e));
  return stream_wait_mark(waiter, e);
}

__global__ void anchor_kernel() {}

}  // namespace dvg

extern "C" int dvg_stream_anchor(dvg_stream_t stream) {
  hipLaunchKernelGGL(dvg::anchor_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
  DVG_CHECK_HIP(hipGetLastError());
  return DVG_OK;
}
