// Internal launch interfaces of the elementwise / special-case kernels (not part of the C ABI).
#pragma once
#include "common.h"

namespace dvg {

constexpr int EW_BLOCKS = 512;       // fixed grid of the reducing elementwise kernels (partials per block)
// Streaming kernels that keep one image or pixel range per iteration in flight want more resident waves than 2 blocks per
// CU give them (enc_conv0_wgrad ran at 2.3 TB/s, dec_conv3_bwd at 2.6 TB/s with EW_BLOCKS blocks): their own, larger
// grids for large batches, partials per block as before
constexpr int STREAM_BLOCKS = 2048;
static inline int stream_blocks(int64_t images) {  // >= 8 images per block, between EW_BLOCKS and STREAM_BLOCKS blocks
  const int64_t b = images / 8;
  return (int)(b < EW_BLOCKS ? EW_BLOCKS : (b > STREAM_BLOCKS ? STREAM_BLOCKS : b));
}
constexpr int BN_FOLD_ROWS = 256;    // scratch rows behind the BN partials: launch_bn_finalize folds long lists first
constexpr float BN_EPS = 1e-5f;      // torch.nn.BatchNorm2d defaults (/root/reference/src/encoder.py:32)
constexpr float BN_MOMENTUM = 0.1f;
constexpr float LRELU_SLOPE = 0.01f;  // torch.nn.LeakyReLU() default (/root/reference/src/encoder.py:36)
constexpr float DROPOUT_KEEP = 0.8f;  // Dropout2d(0.2) (/root/reference/src/decoder.py:42)

// out[perm(w)] = scale * sum_g part[g*stride + w], w < count, accumulated in double in fixed order.
// permA > 0: perm(w) = (w % permA) * permB + w / permA, else identity.
// Several column sums in ONE launch (the bias / small-weight gradient finalisers of a whole backward call: each is a
// ~5 us kernel, and a dozen of them strung along the side stream were a tenth of c2's step).  Same arithmetic per job as
// launch_colsum; launch_colsum2's two outputs are two jobs over the same partials.
struct ColsumJob { const float* part; int G, stride, count; float scale; float* out; int permA, permB; };
constexpr int MAX_COLSUM_JOBS = 12;
struct ColsumBatch {
  ColsumJob job[MAX_COLSUM_JOBS];
  int n = 0;
  bool add(const float* part, int G, int stride, int count, float scale, float* out, int permA, int permB) {
    if (n >= MAX_COLSUM_JOBS) return false;
    job[n++] = ColsumJob{part, G, stride, count, scale, out, permA, permB};
    return true;
  }
  bool add2(const float* part, int G, int stride, int count_a, float* out_a, int count_b, float* out_b) {  // as launch_colsum2
    return add(part, G, stride, count_a, 1.0f, out_a, 0, 0) && add(part + count_a, G, stride, count_b, 1.0f, out_b, 0, 0);
  }
};
int launch_colsum_batch(const ColsumBatch& b, hipStream_t s);
int launch_colsum(const float* part, int G, int stride, int count, float scale, float* out, int permA, int permB,
                  hipStream_t s);

// two destinations in one launch: out_a[w] for w < count_a, out_b[w - count_a] for the next count_b columns
int launch_colsum2(const float* part, int G, int stride, int count_a, float* out_a, int count_b, float* out_b,
                   hipStream_t s);

// part[EW_BLOCKS][cols] column sums of a (rows, cols) matrix; out[w] = in[(w % A) * B + w / A]
int launch_rowsum_partial(const float* mat, int64_t rows, int cols, float* part, hipStream_t s);
int launch_permute_vec(const float* in, int count, int A, int B, float* out, hipStream_t s);

// BatchNorm statistics from per-block (sum, sum^2) partials; updates running stats when training.
// `stats_part` must have BN_FOLD_ROWS spare rows behind its nblk rows (used when nblk is large).
int launch_bn_finalize(const float* stats_part, int nblk, int C, int64_t M, int training, float* mean, float* invstd,
                       float* running_mean, float* running_var, int64_t* nbt, hipStream_t s);

// ---- encoder stage: BN -> MaxPool2d(2) -> (LeakyReLU)
int launch_enc_bn_pool_fwd(const float* Y, int64_t Q, int C, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, int lrelu, float* out, hipStream_t s);
// part [EW_BLOCKS][2][C]: per-block (sum dz, sum dz*zhat)
int launch_enc_bn_pool_bwd_reduce(const float* Y, int64_t Q, int C, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, int lrelu, const float* dOut, float* part,
                                  hipStream_t s, const float* pooled = nullptr);  // pooled: the forward's pooled activations [Q][C] (the sums from them: elementwise.hip)
int launch_enc_bn_pool_bwd_apply(const float* Y, int64_t Q, int C, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int lrelu, const float* dOut, const float* sum_dz,
                                 const float* sum_dzzh, float* dY, float* part_db, hipStream_t s);

// ---- decoder stage: BN -> Dropout2d mask -> LeakyReLU (the x2 upsample is fused into the consumer)
int launch_dropout_masks(int64_t N, const int C[4], float* const mask[4], uint64_t seed, uint64_t offset,
                         const uint64_t* offset_dev, hipStream_t s);
int launch_dec_bn_act_fwd(const float* Y, int64_t M, int C, int logHW, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, const float* mask, float* X, hipStream_t s);
// (gamma / beta: the float4-of-channels kernels take the LeakyReLU slope from the sign of z = fma(zhat, gamma, beta) and do
// not read X; the scalar kernels of the 1-channel stage read X)
int launch_dec_bn_act_bwd_reduce(const float* Y, const float* X, int64_t M, int C, int logHW, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, const float* mask,
                                 const float* dX, float* part, hipStream_t s);
int launch_dec_bn_act_bwd_apply(const float* Y, const float* X, int64_t M, int C, int logHW, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, const float* mask, const float* dX,
                                const float* sum_dz, const float* sum_dzzh, float* dY, float* part_db, hipStream_t s);

// ---- special-case layers
// encoder conv0 (1 -> 32 channels, 32x32): images row-major in, Morton NHWC out (+ BN partials [blocks][32][2])
int enc_conv0_blocks(int64_t B);
int launch_enc_conv0_fwd(const float* images, int64_t B, const float* w, const float* b, float* Y, float* stats_part,
                         hipStream_t s);
// encoder layer 0 with its output recomputed instead of stored (special.hip: enc_l0_kernel<MODE>)
struct EncL0Args {
  const float* img; int64_t B;
  const float* w; const float* bias;                       // conv weight (32,1,3,3), bias (32)
  const float* mean; const float* invstd; const float* gamma; const float* beta;
  const float* pimg;                                       // zero-padded copy of the images [B][ENC_L0_PIMG] written by the moments
                                                           // pass of a training-mode forward, or null (bounds-tested reads of img)
  const float* dXp;                                        // gradient wrt the pooled map [B*256][32]          (MODE 4)
  float* Xp;                                               // [B*256][32]                                     (MODE 1)
  float* part;                                             // MODE 4: [enc_l0_blocks(B)][ENC_L0_ROW_FLOATS]
};
int enc_l0_blocks(int64_t B);
int launch_enc_l0(int mode, const EncL0Args& a, hipStream_t s);  // mode 1 or 4
// The layer's BatchNorm statistics from the first and second moments of the 3x3 input patches (no pass over the layer's
// output): part [enc_l0_moment_blocks(B)][ENC_L0_MOM_ROW] doubles of scratch, mom [ENC_L0_MOM_ROW] doubles = the summed
// moments, which the backward call's launch_enc_l0_combine reads.  Also updates the running statistics and the batch counter.
constexpr int ENC_L0_MOM = 54, ENC_L0_MOM_ROW = 64;
constexpr int ENC_L0_PIMG = 34 * 40;  // floats per padded image: 34 rows of 40, the 32 x 32 interior at row 1 / column 4
int enc_l0_moment_blocks(int64_t B);
int launch_enc_l0_moments(const float* images, int64_t B, const float* w, const float* bias, double* part, double* mom,
                          float* pimg, float* mean, float* invstd, float* rm, float* rv, int64_t* nbt, hipStream_t s);
// mode 4 (both backward passes in one): partial rows of ENC_L0_ROW_FLOATS floats (S: 320, sum dz zhat: 32); their column
// sums `tot` go through launch_enc_l0_combine, which writes the four gradients of the stage
constexpr int ENC_L0_ROW_FLOATS = 352;
int launch_enc_l0_combine(const float* tot, const double* mom, const float* w, const float* bias, const float* mean,
                          const float* gamma, const float* invstd, int64_t B, float* gw, float* gb, float* g_bn_b,
                          float* g_bn_g, hipStream_t s);
// part: [EW_BLOCKS][320]: 288 weight-gradient entries in checkpoint order, then 32 bias-gradient entries
int launch_enc_conv0_wgrad(const float* images, int64_t B, const float* dY, float* part, hipStream_t s);
// Linear(4,1) over the 2x2 pooled map: P (B,4,n) -> logits (B,n)
int launch_enc_proj_fwd(const float* P, int64_t B, int n, const float* w, const float* b, float* logits, hipStream_t s);
// dP (B,4,n); part [EW_BLOCKS][5]: d w[0..3], d b
int launch_enc_proj_bwd(const float* P, int64_t B, int n, const float* w, const float* dlogits, float* dP, float* part,
                        hipStream_t s);
// decoder conv3 (32 -> 1 channel, 16x16, input upsampled from 8x8)
// The layer's input given as the PREVIOUS layer's pre-BatchNorm output: the kernels apply BN -> Dropout2d -> LeakyReLU
// while they stage an image (decoder layer 2's activated map, 268 MB at c3, is then never written or read).  `mask`:
// Dropout2d keep-mask [N][32] or null (evaluation).  Null `y`: the input is the activated map itself.
struct DecActIn {
  const float* y; const float* mean; const float* invstd; const float* gamma; const float* beta; const float* mask;
};
int launch_dec_conv3_fwd_act(const DecActIn& in, int64_t N, const float* w, const float* b, float* Y, float* stats_part,
                             hipStream_t s);
// backward of conv3 fused with the BatchNorm/Dropout/LeakyReLU backward of the layer in front of it: pass 1 = conv3's weight
// gradient partials part_w [blocks][288] + that layer's (sum dz, sum dz zhat) partials part_bn [blocks][64] (no dX);
// pass 2 = that layer's dY [N*64][32] + its column-sum partials part_db [blocks][32]
int dec_tail_blocks(int64_t N);
// the final layer's data gradient with the 1-channel stage's (sum dz, sum dz zhat) partials [dec_final_dgrad_blocks][2]
int dec_final_dgrad_blocks(int64_t N);
int launch_dec_final_dgrad_bn(const float* dOut, int64_t N, const float* w, const DecActIn& in, float* part, hipStream_t s);
// ... and the stage's dY [N*256] once the sums are known (the data gradient is formed again; it is never stored), with the
// block sums of dY in part_db [dec_final_dgrad_blocks][1]
int launch_dec_final_dgrad_apply(const float* dOut, int64_t N, const float* w, const DecActIn& in, const float* sum_dz,
                                 const float* sum_dzzh, float* dY, float* part_db, hipStream_t s);
int launch_dec_conv3_bwd_reduce(const DecActIn& in, int64_t N, const float* dY3, const float* w, float* part_w, float* part_bn,
                                hipStream_t s);
int launch_dec_conv3_bwd_apply(const DecActIn& in, int64_t N, const float* dY3, const float* w, const float* sum_dz,
                               const float* sum_dzzh, float* dY2, float* part_db, hipStream_t s);
int launch_dec_conv3_fwd(const float* X, int64_t N, const float* w, const float* b, float* Y, float* stats_part,
                         hipStream_t s);
int dec_conv3_blocks(int64_t N);
// part [EW_BLOCKS][288] indexed tap*32 + ci (tap = kh*3+kw of the checkpoint weight)
// data gradient + weight-gradient partials in one pass over the images
int launch_dec_conv3_bwd(const float* X, int64_t N, const float* dY, const float* w, float* dX, float* part, hipStream_t s);
// decoder final ConvTranspose2d(1,1) at 32x32 from the upsampled 16x16 map; row-major output
int launch_dec_final_fwd(const float* X, int64_t N, const float* w, const float* b, float* out, hipStream_t s);
// ... with its input given as the 1-channel stage's pre-BatchNorm output (C = 1 DecActIn; mask [N] or null): activated
// while it is staged (forward and weight gradient), so that stage's activated map is never written
int launch_dec_final_fwd_act(const DecActIn& in, int64_t N, const float* w, const float* b, float* out, hipStream_t s);
int launch_dec_final_wgrad_act(const DecActIn& in, int64_t N, const float* dOut, float* part, hipStream_t s);
// The training step's tail in one pass per image (special.hip, dec_tail_mse_kernel): final ConvTranspose forward -> MSE
// against images [N / R][1024] -> the final layer's data gradient -> the 1-channel stage's backward; the reconstruction
// and its gradient are never written.  gscale = 2 grad_scale / (N 1024).  _sums: part [dec_final_dgrad_blocks][2] and the
// squared-error partials mse_part [dec_final_dgrad_blocks] (launch_mse_final sums them); _apply: the stage's dY and
// part_db [dec_final_dgrad_blocks][1]; _wgrad_mse: the final layer's weight / bias gradient partials [EW_BLOCKS][10].
int launch_dec_tail_mse_sums(const DecActIn& in, int64_t N, const float* w, const float* bias, const float* images, int R,
                             float gscale, float* part, double* mse_part, hipStream_t s);
int launch_dec_tail_mse_apply(const DecActIn& in, int64_t N, const float* w, const float* bias, const float* images, int R,
                              float gscale, const float* sum_dz, const float* sum_dzzh, float* dY, float* part_db,
                              hipStream_t s);
int launch_dec_final_wgrad_mse(const DecActIn& in, int64_t N, const float* w, const float* bias, const float* images, int R,
                               float gscale, float* part, hipStream_t s);
int launch_mse_final(const double* partial, int nb, double inv_numel, float* loss, hipStream_t s);  // misc.hip
int launch_dec_final_dgrad(const float* dOut, int64_t N, const float* w, float* dX, hipStream_t s);
// part [EW_BLOCKS][10]: d w[0..8], d b
int launch_dec_final_wgrad(const float* X, int64_t N, const float* dOut, float* part, hipStream_t s);

}  // namespace dvg
