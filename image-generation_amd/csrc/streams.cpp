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

#include "common.h"

namespace dvg {

namespace {
constexpr int kMaxDevices = 16;
constexpr int kEvents = 64;
// The library is entered from the caller's thread (forward) and from the autograd engine's worker thread (backward):
// creation is serialised by a mutex and published through an acquire/release flag, the event-ring cursor is atomic.
struct SideCtx {
  hipStream_t side = nullptr;
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
  c.ready.store(true, std::memory_order_release);
  return &c;
}
}  // namespace

bool side_enabled();

// Creates the device's side stream and event ring now (never under a stream capture: the workspace-size queries and
// dvg_graph_create call this, and they always precede the first captured step).
void side_stream_warm() { if (side_enabled()) (void)ctx(); }

bool side_enabled() { return opt(OPT_SIDE_STREAM) != 0; }

hipStream_t side_stream(hipStream_t fallback) {
  if (!side_enabled()) return fallback;
  SideCtx* c = ctx();
  return c ? c->side : fallback;
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
