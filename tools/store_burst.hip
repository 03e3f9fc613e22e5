// How fast does a burst of output stores drain?  Every workgroup (512 threads, one per CU) stores a 1024-row x 32-channel float32
// block (128 KB; row stride = Cout floats) in one of three per-instruction shapes, nothing else; time per launch = the burst.
//   0: b32, lane (g = lane >> 4, c = lane & 15): four rows x 16 channels per instruction (four 64-byte segments) -- the F(4x4) kernel's
//   1: b32, lane (h = lane >> 5, c = lane & 31): two rows x 32 channels (two 128-byte segments) -- the F(2x2) kernel's
//   2: b128, lane (r = lane >> 3, q = lane & 7): eight rows x 32 channels (eight whole 128-byte lines)
// build: hipcc --offload-arch=gfx950 -O3 tools/store_burst.hip -o /tmp/store_burst ; run: /tmp/store_burst [Cout]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int P>
__global__ __launch_bounds__(512) void burst(float* out, int Cout, int nblk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    float* base = out + (size_t)(blk >> 1) * 1024 * Cout + (blk & 1) * 32;  // two column blocks share the rows (Cout = 64)
    if (P == 0) {
      const int g = lane >> 4, c = lane & 15, wt = wave & 3, wc = wave >> 2;
      for (int idx = 0; idx < 64; ++idx) {
        const int i = idx & 3, p = idx >> 2;
        base[(size_t)((16 * wt + 4 * g + i) * 16 + p) * Cout + 16 * wc + c] = (float)idx;
      }
    } else if (P == 1) {
      const int h = lane >> 5, c = lane & 31;
      for (int idx = 0; idx < 64; ++idx) base[(size_t)(wave * 128 + idx * 2 + h) * Cout + c] = (float)idx;
    } else {
      const int r = lane >> 3, q = lane & 7;
      for (int idx = 0; idx < 16; ++idx) {
        const f4 v = {(float)idx, 1.f, 2.f, 3.f};
        *reinterpret_cast<f4*>(base + (size_t)(wave * 128 + idx * 8 + r) * Cout + 4 * q) = v;
      }
    }
  }
}
int main(int argc, char** argv) {
  const int Cout = argc > 1 ? atoi(argv[1]) : 64;
  const int nblk = 256 * (Cout / 32) / (Cout / 32);  // one block per CU
  float* out; hipMalloc(&out, (size_t)1024 * 1024 * Cout * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int P = 0; P < 3; ++P)
    for (int rounds : {1, 4}) {
      float best = 1e9f;
      for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0);
        if (P == 0) burst<0><<<256, 512>>>(out, Cout, 256 * rounds);
        else if (P == 1) burst<1><<<256, 512>>>(out, Cout, 256 * rounds);
        else burst<2><<<256, 512>>>(out, Cout, 256 * rounds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
      }
      printf("shape %d, %d tile block(s) per workgroup, Cout %d: %.1f us per launch = %.2f TB/s (%.1f us per 128 KB block)\n", P, rounds, Cout,
             best * 1e3, 256.0 * rounds * 131072 / (best * 1e-3) / 1e12, best * 1e3 / rounds);
    }
  return 0;
}
