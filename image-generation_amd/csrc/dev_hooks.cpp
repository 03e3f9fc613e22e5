// Development hooks (include/dvg_dev.h): direct launches of the MFMA GEMM kernels for unit tests / micro-benchmarks.
#include "conv.h"
#include "../../include/dvg_dev.h"
using namespace dvg;

extern "C" int dvg_dev_conv_igemm(const float* in, const float* w, int mode, float* wp, const float* bias, float* out,
                                  float* stats, int64_t M, int Cin, int Cout, int L, int ntaps, int ups, int poolsum,
                                  int repack, float* splitk_ws, dvg_stream_t stream) {
  DVG_REQUIRE(in && w && wp && out, "dev_conv_igemm: null argument");
  hipStream_t s = (hipStream_t)stream;
  if (repack) DVG_TRY(launch_weight_pack(w, WeightMap{mode, Cin, Cout, ntaps}, wp, s, conv_launch_mode(M, Cout)));
  ConvArgs a;
  a.in = in; a.wp = wp; a.bias = bias; a.out = out; a.stats = stats;
  a.M = M; a.Cin = Cin; a.Cout = Cout; a.L = L; a.ntaps = ntaps; a.ups = ups; a.poolsum = poolsum;
  a.splitk_ws = splitk_ws;
  return launch_conv_igemm(a, s);
}

// The same layer in the Winograd F(2x2,3x3) form (conv_wino.hip): `u` takes the 16 Cin Cout transformed weights; stats rows =
// dvg_dev_conv_wino_stats_blocks.  DVG_E_INVALID when the shape does not qualify (dvg_dev_conv_wino_ok).
extern "C" int dvg_dev_conv_wino(const float* in, const float* w, int mode, float* u, const float* bias, float* out,
                                 float* stats, int64_t M, int Cin, int Cout, int L, dvg_stream_t stream) {
  DVG_REQUIRE(in && w && u && out, "dev_conv_wino: null argument");
  hipStream_t s = (hipStream_t)stream;
  DVG_TRY(launch_wino_weight_pack(w, WeightMap{mode, Cin, Cout, 9}, u, s));
  ConvArgs a;
  a.in = in; a.wp = u; a.bias = bias; a.out = out; a.stats = stats;
  a.M = M; a.Cin = Cin; a.Cout = Cout; a.L = L; a.ntaps = 9; a.ups = 0; a.poolsum = 0;
  a.splitk_ws = nullptr;
  return launch_conv_wino(a, s);
}
// The same layer in the Winograd F(4x4,3x3) form (conv_wino4.hip): `u` takes the 36 Cin Cout transformed weights; stats rows =
// M / 1024.  DVG_E_INVALID when the shape does not qualify (dvg_dev_conv_wino4_shape).
extern "C" int dvg_dev_conv_wino4(const float* in, const float* w, int mode, float* u, const float* bias, float* out,
                                  float* stats, int64_t M, int Cin, int Cout, int L, int cus, int um, dvg_stream_t stream) {
  DVG_REQUIRE(in && w && u && out, "dev_conv_wino4: null argument");
  hipStream_t s = (hipStream_t)stream;
  DVG_TRY(launch_wino4_weight_pack(w, WeightMap{mode, Cin, Cout, 9}, u, s, um != 0));
  ConvArgs a;
  a.in = in; a.wp = u; a.bias = bias; a.out = out; a.stats = stats;
  a.M = M; a.Cin = Cin; a.Cout = Cout; a.L = L; a.ntaps = 9; a.ups = 0; a.poolsum = 0;
  a.splitk_ws = nullptr; a.wino_cus = cus; a.wino_um = um;
  return launch_conv_wino4(a, s);
}
extern "C" int dvg_dev_conv_wino4_shape(int64_t M, int Cin, int Cout, int L) { return conv_wino4_shape(M, Cin, Cout, L) ? 1 : 0; }
extern "C" int dvg_dev_conv_wino_ok(int64_t M, int Cin, int Cout, int L) { return conv_wino_ok(M, Cin, Cout, L) ? 1 : 0; }
extern "C" int dvg_dev_conv_wino_stats_blocks(int64_t M, int Cout) { return conv_wino_stats_blocks(M, Cout); }

extern "C" size_t dvg_dev_conv_splitk_floats(int64_t M, int Cin, int Cout, int ntaps, int poolsum) {
  return conv_splitk_floats(M, Cin, Cout, ntaps, poolsum);
}

extern "C" int dvg_dev_conv_stats_blocks(int64_t M, int Cout) { return conv_stats_blocks(M, Cout); }

extern "C" size_t dvg_dev_wgrad_slab_floats(int64_t M, int Cin, int Cout, int ntaps) {
  return (size_t)wgrad_ksplit(M, Cin, Cout, ntaps) * ntaps * Cin * Cout;
}

extern "C" int dvg_dev_conv_wgrad(const float* in, const float* dy, float* slabs, float* grad_w, int mode, int64_t M,
                                  int Cin, int Cout, int L, int ntaps, int ups, dvg_stream_t stream) {
  DVG_REQUIRE(in && dy && slabs && grad_w, "dev_conv_wgrad: null argument");
  hipStream_t s = (hipStream_t)stream;
  WgradArgs wa;
  wa.in = in; wa.dy = dy; wa.slabs = slabs; wa.M = M; wa.Cin = Cin; wa.Cout = Cout; wa.L = L; wa.ntaps = ntaps;
  wa.ups = ups; wa.ksplit = wgrad_ksplit(M, Cin, Cout, ntaps);
  DVG_TRY(launch_conv_wgrad(wa, s));
  return launch_wgrad_reduce(slabs, wa.ksplit, WeightMap{mode, Cin, Cout, ntaps}, grad_w, s);
}

extern "C" size_t dvg_dev_wino_wgrad_slab_floats(int64_t M, int Cin, int Cout, int L) {
  return conv_wino_wgrad_shape(M, Cin, Cout, L) ? conv_wino_wgrad_slab_floats(M, Cin, Cout, L) : 0;
}

extern "C" int dvg_dev_conv_wino_wgrad(const float* in, const float* dy, float* slabs, float* grad_w, int mode, int64_t M,
                                       int Cin, int Cout, int L, int ups, int cus, dvg_stream_t stream) {
  DVG_REQUIRE(in && dy && slabs && grad_w, "dev_conv_wino_wgrad: null argument");
  return launch_conv_wino_wgrad(in, dy, M, Cin, Cout, L, slabs, WeightMap{mode, Cin, Cout, 9}, grad_w, (hipStream_t)stream, ups, cus);
}

// The same weight gradient in the Winograd F(4x4,3x3) form (conv_wino4_wgrad.hip): slabs of dvg_dev_wino4_wgrad_slab_floats()
// floats (0 = the shape does not qualify)
extern "C" size_t dvg_dev_wino4_wgrad_slab_floats(int64_t M, int Cin, int Cout, int L) {
  return conv_wino4_wgrad_shape(M, Cin, Cout, L) ? conv_wino4_wgrad_slab_floats(M, Cin, Cout) : 0;
}

extern "C" int dvg_dev_conv_wino4_wgrad(const float* in, const float* dy, float* slabs, float* grad_w, int mode, int64_t M,
                                        int Cin, int Cout, int L, int cus, dvg_stream_t stream) {
  DVG_REQUIRE(in && dy && slabs && grad_w, "dev_conv_wino4_wgrad: null argument");
  return launch_conv_wino4_wgrad(in, dy, M, Cin, Cout, L, slabs, WeightMap{mode, Cin, Cout, 9}, grad_w, (hipStream_t)stream, cus);
}
