/* dvg_dev.h — development hooks of libdvg.so: direct access to the two MFMA GEMM kernels, used by
 * the kernel-level unit tests and micro-benchmarks (tests/test_gpu_kernels.py, scratch/).  NOT part of
 * the drop-in boundary (include/dvg.h); signatures may change between rounds.
 * Tensors are in the library's internal layout: NHWC float32, pixels of each image in Morton order. */
#ifndef DVG_DEV_H
#define DVG_DEV_H
#include "dvg.h"
#ifdef __cplusplus
extern "C" {
#endif
/* mode: 0 Conv2d fwd, 1 Conv2d dgrad, 2 ConvTranspose2d fwd, 3 ConvTranspose2d dgrad, 4 Linear fwd, 5 Linear dgrad
 * (csrc/conv.h WeightMode).  w: checkpoint-layout weight; wp: scratch for the packed copy (ntaps*Cin*Cout floats).
 * out[m][co] = sum_{tap,ci} in[nbr(m,tap)][ci] * Wp[tap][ci][co] (+bias); stats: [blocks][Cout][2] or NULL. */
int dvg_dev_conv_igemm(const float *in, const float *w, int mode, float *wp, const float *bias, float *out,
                       float *stats, int64_t M, int Cin, int Cout, int L, int ntaps, int ups, int poolsum,
                       int repack, float *splitk_ws, dvg_stream_t stream);
size_t dvg_dev_conv_splitk_floats(int64_t M, int Cin, int Cout, int ntaps, int poolsum);
/* The same stride-1 3x3 layer in the Winograd F(2x2,3x3) form the encoder uses from 256 workgroups up (option enc_wino):
 * `u` = scratch of 16*Cin*Cout floats (transformed weights), `stats` rows = dvg_dev_conv_wino_stats_blocks(M, Cout).
 * Replaces (with the implicit GEMM above) /root/reference/src/encoder.py:28-36's nn.Conv2d calls. */
int dvg_dev_conv_wino(const float *in, const float *w, int mode, float *u, const float *bias, float *out, float *stats,
                      int64_t M, int Cin, int Cout, int L, dvg_stream_t stream);
int dvg_dev_conv_wino_ok(int64_t M, int Cin, int Cout, int L);
/* The same layer in the Winograd F(4x4,3x3) form (csrc/conv_wino4.hip: 36 position GEMMs per 4x4 output tile, 2.25 multiplies
 * per output where F(2x2,3x3) issues 4.0): `u` = scratch of 36*Cin*Cout floats, `stats` rows = M / 1024 (tile blocks of 64
 * tiles), cus = CUs the persistent grid is sized for (0 = 256).  Shapes: whole 1024-pixel blocks of 4x4 / 8x8 / 16x16 images. */
int dvg_dev_conv_wino4(const float *in, const float *w, int mode, float *u, const float *bias, float *out, float *stats,
                       int64_t M, int Cin, int Cout, int L, int cus, int um, dvg_stream_t stream);
/* (um = 1: Upsample(x2) + 3x3 forward, `in` = the source map ([M / 4][Cin]), 25 of 36 positions; um = 2: its data gradient,
 * `in` = the fine-grid gradient, `out` = the source map's gradient ([M / 4][Cout])) */
int dvg_dev_conv_wino4_shape(int64_t M, int Cin, int Cout, int L);
/* ... and the layer's weight gradient in that form (csrc/conv_wino4_wgrad.hip): slabs of dvg_dev_wino4_wgrad_slab_floats()
 * floats (0 = the shape does not qualify); cus: CUs the grid is sized for (0 = the training step's budget). */
size_t dvg_dev_wino4_wgrad_slab_floats(int64_t M, int Cin, int Cout, int L);
int dvg_dev_conv_wino4_wgrad(const float *in, const float *dy, float *slabs, float *grad_w, int mode, int64_t M, int Cin,
                             int Cout, int L, int cus, dvg_stream_t stream);
int dvg_dev_conv_wino_stats_blocks(int64_t M, int Cout);
int dvg_dev_conv_stats_blocks(int64_t M, int Cout);
/* grad_w (checkpoint layout, `mode` = the layer's FORWARD mode) = sum_m in[nbr(m,tap)] (x) dy[m]; slabs: scratch of
 * dvg_dev_wgrad_slab_floats() floats. */
size_t dvg_dev_wgrad_slab_floats(int64_t M, int Cin, int Cout, int ntaps);
int dvg_dev_conv_wgrad(const float *in, const float *dy, float *slabs, float *grad_w, int mode, int64_t M, int Cin,
                       int Cout, int L, int ntaps, int ups, dvg_stream_t stream);
/* The same weight gradient in the Winograd F(2x2,3x3) form (conv_wino_wgrad.hip; ups = 1: `in` is the source map of an Upsample(x2) + 3x3 layer): slabs of
 * dvg_dev_wino_wgrad_slab_floats() floats (0 = the shape does not qualify). */
size_t dvg_dev_wino_wgrad_slab_floats(int64_t M, int Cin, int Cout, int L);
int dvg_dev_conv_wino_wgrad(const float *in, const float *dy, float *slabs, float *grad_w, int mode, int64_t M, int Cin,
                            int Cout, int L, int ups, int cus, dvg_stream_t stream); /* cus: CUs the grid is sized for (0 = the training step's budget) */
/* Where the encoder's workspace (dvg_encoder_workspace_bytes) keeps what a forward call saved, as FLOAT offsets:
 * out[0..3] = pre-BatchNorm convolution outputs Y[l] ([B * HW_l][C_l], Morton NHWC), out[4..7] = pooled stage outputs
 * Xp[l], out[8..11] = batch means, out[12..15] = batch inverse standard deviations.  Diagnostics only.  (Y[0] is only
 * written under option enc_l0_fused = 0: the default recomputes layer 0 wherever its output is needed.) */
int dvg_dev_encoder_layout(int64_t B, int n_latents, size_t out[16]);
/* A/B references and test knobs: the kernel forms a default of the library replaced (register-staged GEMMs, the stored
 * encoder layer 0, the unfused decoder tail, the static Winograd tile deal, the rolled sampler schedule, forced MMD
 * kernels) and the tile threshold the small fixtures lower.  Same semantics as dvg_set_option / dvg_get_option
 * (include/dvg.h), a separate name space: the product's boundary carries eight switches, these are not among them.
 * dvg_reset_options restores both kinds. */
int dvg_dev_option_count(void);
const char *dvg_dev_option_name(int index);
const char *dvg_dev_option_doc(int index);
int dvg_dev_set_option(const char *name, int64_t value);
int dvg_dev_get_option(const char *name, int64_t *value);
#ifdef __cplusplus
}
#endif
#endif
