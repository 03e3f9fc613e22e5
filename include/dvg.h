/* dvg.h — C ABI of libdvg.so: the MI355X (gfx950) DVAE + GRBM training path.
 *
 * This is the drop-in boundary.  The reference is pure Python
 * over PyTorch and the un-vendored `dwave-pytorch-plugin`; each entry point
 * below replaces the arithmetic behind one reference call site (cited per
 * function).  The reference-side binding is a ctypes stub, shown in
 * INTEGRATION.md and implemented in image-generation_amd/_lib.py.
 *
 * Conventions
 *   - plain pointers and sizes only; every `float*`/`int8_t*`/... argument not
 *     marked [host] is a DEVICE pointer owned by the caller (torch owns memory);
 *   - every function enqueues its work on `stream` (a hipStream_t passed as
 *     void*) and returns without synchronising; no hidden allocation except
 *     inside dvg_graph_create;
 *   - return value: 0 (DVG_OK) or a negative DVG_E_* code; the message is
 *     available from dvg_last_error() (thread-local); nothing throws or exits;
 *   - process-wide state, all of it listed here: (1) the optional per-kernel
 *     profiler (dvg_prof_*), (2) the GEMM-operand mode set by
 *     dvg_set_conv_precision (one atomic int; a forward call and its backward
 *     call must run in the same mode, which the library checks per workspace),
 *     (3) one library-owned side stream + event ring per device, created under
 *     a mutex by the first *_workspace_bytes query or backward call on that
 *     device and used for the fork/join inside dvg_encoder_bwd /
 *     dvg_decoder_bwd, (4) eight kernel-form switches (dvg_set_option: named
 *     integers, relaxed atomics, read per call), each selecting between two
 *     TESTED forms of the same function; the product path is every switch's
 *     default and no caller in this repository sets one outside tests and A/B
 *     measurements (round 4 had 32, among them grid-sizing knobs that
 *     ModelWrapper set per model: those are compile-time constants now; round
 *     5 had 16: the A/B references of forms a default replaced and the test
 *     knobs live behind include/dvg_dev.h, dvg_dev_set_option, since round 6).  A
 *     switch is process-wide: flipping one between a forward call and its
 *     backward call is caught per workspace (the forward records the plan, a
 *     backward under another plan fails with DVG_E_INVALID instead of reading
 *     what its forward never wrote),
 *     (5) host-side bookkeeping keyed by workspace pointer: the mode and plan a
 *     forward call ran under (checked by its backward call) and the mark of a
 *     dvg_decoder_prepare in flight (one event per workspace), (6) per device, a
 *     pool of 32 sets of tile counters in device memory for the dynamically
 *     scheduled Winograd grids (option wino_dynamic): a launch takes the next
 *     set and leaves it zeroed; more than 32 such launches in flight at once
 *     on one device are not supported.
 *     Nothing else persists between calls; entry points are re-entrant per
 *     (stream, workspace).  HIP is initialised lazily by the
 *     first call in each process (the Dash app runs training in a spawned
 *     worker: /root/reference/app.py:37-43).
 */
#ifndef DVG_H
#define DVG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVG_OK 0
#define DVG_E_INVALID (-1)     /* bad argument (shape, null pointer, unsupported size) */
#define DVG_E_HIP (-2)         /* a HIP runtime call failed */
#define DVG_E_WORKSPACE (-3)   /* caller workspace too small */
#define DVG_E_UNSUPPORTED (-4) /* valid request this build cannot serve */

typedef void *dvg_stream_t; /* hipStream_t */

int dvg_version(void);
/* sha256 (first 16 hex digits) of the kernel sources this library was built from (csrc, include/dvg.h): profiles
 * under profiles/ carry the hash of the library they were measured on, bench.py drops the ones that do not match. */
const char *dvg_source_hash(void);
const char *dvg_last_error(void);

/* Device-resident per-step values, for steps replayed from a captured hipGraph.  A function handed a
 * non-NULL `dyn` (DEVICE pointer) reads these fields at run time instead of its by-value arguments of the
 * same meaning; the host rewrites the struct (one small async copy) before each replay.  NULL = by-value. */
typedef struct {
  uint32_t sweep0;         /* dvg_gibbs_sample: global index of the first sweep */
  uint32_t reserved;
  uint64_t gumbel_offset;  /* dvg_gumbel_fwd: Philox offset */
  uint64_t dropout_offset; /* dvg_decoder_fwd: Philox offset */
  float adam_step_size[2]; /* dvg_adam_step (slot 0 / 1): lr / (1 - beta1^t) */
  float adam_bc2_sqrt[2];  /*                              sqrt(1 - beta2^t) */
} dvg_step_state_t;

/* ------------------------------------------------------------------ graph
 * The GRBM graph as the sampler and the energy kernels need it.  Built on the
 * host by image-generation_amd/graphs.py::build_plan from what the reference
 * derives at /root/reference/src/utils/common.py:123-126 and hands to
 * GraphRestrictedBoltzmannMachine(nodes, edges)
 * (/root/reference/src/model_wrapper.py:202-206).  All arrays [host], int32.
 * The handle owns small device copies (the only allocation in this library).
 */
typedef struct dvg_graph dvg_graph_t;
int dvg_graph_create(int n, int n_edges, const int32_t *edge_i, const int32_t *edge_j,
                     const int32_t *order, const int32_t *class_ptr, int n_colours,
                     const int32_t *adj_ptr, const int32_t *adj_idx, const int32_t *adj_eid,
                     dvg_graph_t **out);
int dvg_graph_destroy(dvg_graph_t *g);

/* ------------------------------------------------------------------ sampler
 * Replaces the QPU draw `sampler.sample_ising(h, J, num_reads=...)` issued by
 * GraphRestrictedBoltzmannMachine.sample (/root/reference/src/model_wrapper.py:309-316,
 * :369-376; /root/reference/src/utils/persistent_qpu_sampler.py:71-78) with
 * `n_sweeps` sweeps of the graph-coloured block-Gibbs sampler defined in
 * oracle/gibbs.py, on `n_chains` chains with global ids chain_id0 ... .
 *   linear (n), quadratic (n_edges): raw GRBM parameters; the kernel applies
 *     hs = clamp(prefactor*h, h_lo, h_hi), Js = clamp(prefactor*J, j_lo, j_hi)
 *     (the plugin's to_ising) and samples p(s) ~ exp(-beta (hs.s + s.Js.s)).
 *   state (n_chains, n) int8 +-1: in/out (persistent chains).  init != 0 draws
 *     the start state from the INIT stream first, keyed by (spin, chain id, sweep0):
 *     chains restarted on every draw (non-persistent mode) start each draw elsewhere.
 *   samples_out (n_chains, n) float32 +-1, or NULL.
 *   sweep0: global index of the first sweep (keeps the random stream moving).
 */
int dvg_gibbs_sample(const dvg_graph_t *g, const float *linear, const float *quadratic,
                     float prefactor, float h_lo, float h_hi, float j_lo, float j_hi, float beta,
                     int8_t *state, int n_chains, uint32_t chain_id0, uint64_t seed,
                     uint32_t sweep0, int n_sweeps, int init, float *samples_out,
                     const dvg_step_state_t *dyn, dvg_stream_t stream);
/* The launch geometry dvg_gibbs_sample uses for n_chains chains on this graph (nothing is launched; any out pointer may
 * be NULL): workgroups, threads per workgroup, LDS bytes per workgroup.  For callers that size what runs BESIDE the draw
 * (tools and A/B measurements; since round 5 the Winograd grids beside the draw deal their tile blocks dynamically and
 * no caller needs it on the product path; the reference has no counterpart -- its draw is a QPU call,
 * /root/reference/src/model_wrapper.py:309-316). */
int dvg_gibbs_launch_info(const dvg_graph_t *g, int n_chains, int *workgroups, int *threads, size_t *lds_bytes);

/* ------------------------------------------------------------------ GRBM
 * Energy  E(x) = x.h + sum_e J_e x_i x_j  per row: the plugin's
 * GraphRestrictedBoltzmannMachine.__call__ as used by nll_loss
 * (/root/reference/src/losses.py:61).  x (rows, n) float32.
 */
int dvg_grbm_energy(const dvg_graph_t *g, const float *x, int64_t rows, const float *linear,
                    const float *quadratic, float *energy_out, dvg_stream_t stream);
/* Backward of the energy = weighted sufficient statistics:
 *   acc_linear[i]    (+)= scale * sum_r w_r x_ri ;
 *   acc_quadratic[e] (+)= scale * sum_r w_r x_ri x_rj      (w_r = 1 when row_weight is NULL).
 * accumulate == 0 overwrites.  With w = 1, scale = 1/rows this is the data (or, negated, the
 * model) half of the quasi-NLL gradient of /root/reference/src/losses.py:61.  Deterministic
 * (fixed-order two-stage reduction in double).  ws: dvg_grbm_suffstats_workspace_bytes(g). */
size_t dvg_grbm_suffstats_workspace_bytes(const dvg_graph_t *g);
int dvg_grbm_suffstats(const dvg_graph_t *g, const float *x, int64_t rows, const float *row_weight,
                       float scale, float *acc_linear, float *acc_quadratic, int accumulate,
                       void *ws, size_t ws_bytes, dvg_stream_t stream);

/* ------------------------------------------------------------------ input pipeline
 * The reference's per-image transform Resize((S,S)) -> ToTensor() -> round
 * (/root/reference/src/model_wrapper.py:70-77) over a whole data set: src (n_images, in, in) uint8 ->
 * out (n_images, 1, out, out) float32 in {0, 1}.  Bit-exact restatement of PIL's two-pass BILINEAR
 * resampling (8-bit intermediate, 22-bit fixed-point coefficients), which is what torchvision's Resize
 * applies to the PIL images MNIST yields; sides up to 64. */
int dvg_resize_binarise(const uint8_t *src, int64_t n_images, int in_size, int out_size, float *out,
                        dvg_stream_t stream);
/* out[r] = table[idx[r]], rows of row_floats floats (the shuffled mini-batch of a device-resident data
 * set: DataLoader(shuffle=True), /root/reference/src/model_wrapper.py:103).  idx: DEVICE int64.  An index
 * outside [0, rows) is not dereferenced; *bad_index_flag (DEVICE int, may be NULL) is set to 1. */
int dvg_gather_rows(const float *table, int64_t rows, int64_t row_floats, const int64_t *idx, int64_t n,
                    float *out, int *bad_index_flag, dvg_stream_t stream);

/* ------------------------------------------------------------------ latent -> discrete
 * Default latent_to_discrete of DiscreteVariationalAutoencoder
 * (/root/reference/src/model_wrapper.py:184-188 passes None -> plugin default):
 * Gumbel-softmax over logits [l, 0], temperature tau, hard, R replicas, mapped
 * to spins +-1 with the straight-through gradient.
 *   logits (B, n); gumbels (B, R, n, 2) Gumbel(0,1) noise or NULL (then drawn
 *   on device from Philox stream GUMBEL with (seed, offset));
 *   spins (B, R, n) out; dspin (B, R, n) out: d spin / d logit, kept for bwd.
 */
int dvg_gumbel_fwd(const float *logits, int64_t B, int n, int R, float tau, const float *gumbels,
                   uint64_t seed, uint64_t offset, float *spins, float *dspin,
                   const dvg_step_state_t *dyn, dvg_stream_t stream);
/* grad_logits (B, n) = sum_r grad_spins[b,r,:] * dspin[b,r,:] */
int dvg_gumbel_bwd(const float *grad_spins, const float *dspin, int64_t B, int n, int R,
                   float *grad_logits, dvg_stream_t stream);
/* the same with TWO gradients wrt the spins (grad_spins2 may be NULL): grad_logits = sum_r (g1 + g2) * dspin.  The step
 * has two -- through the decoder (MSE) and from the MMD, /root/reference/src/model_wrapper.py:322-326 backpropagates
 * their sum -- and adding them here saves a pass over (B, R, n). */
int dvg_gumbel_bwd2(const float *grad_spins, const float *grad_spins2, const float *dspin, int64_t B, int n, int R,
                    float *grad_logits, dvg_stream_t stream);
/* out[0] = a[0] + b[0] on the device (dvae_loss = mse + mmd, /root/reference/src/model_wrapper.py:322, without a host
 * synchronisation and without a framework kernel in the captured step) */
int dvg_scalar_add(const float *a, const float *b, float *out, dvg_stream_t stream);
/* "heaviside" mode (/root/reference/src/utils/common.py:160-173): spins (B,1,n) = 2*H(l)-1 with
 * H(0)=0; the backward is the identity and needs no kernel. */
int dvg_heaviside_fwd(const float *logits, int64_t numel, float *spins, dvg_stream_t stream);

/* ------------------------------------------------------------------ MMD
 * GaussianKernel(n_kernels) + maximum_mean_discrepancy_loss(x, y, kernel)
 * (/root/reference/src/model_wrapper.py:273, :320).  One fused pass computes the
 * loss and d loss / d x (y carries no gradient in the reference: it is drawn
 * under torch.no_grad(), model_wrapper.py:308).  Never materialises the
 * (nx+ny)^2 kernel matrix.
 */
typedef struct {
  int32_t n_kernels;   /* 7 in the reference */
  float factor;        /* bandwidth multiplier base, 2.0 */
  float bandwidth;     /* > 0: fixed base bandwidth; <= 0: data-driven sum(D)/(N^2-N) */
  int32_t squared;     /* 0: Euclidean distance (default), 1: squared distance */
  int32_t reduce_mean; /* 0: kernels summed (default), 1: averaged */
  int32_t biased;      /* 0: unbiased estimator (default), 1: biased (plain means) */
} dvg_mmd_cfg_t;
size_t dvg_mmd_workspace_bytes(int64_t nx, int64_t ny, int dim);
/* Executed matrix work of one dvg_mmd_fwd_bwd call with gradient on +-1 (spin) rows of this shape, by MFMA type: int8
 * FLOPs of the Gram tiles, bf16 FLOPs of the gradient GEMM and the number of bf16 terms a weight is split into (0 / 0 / 0
 * when the shape is served by the f32 kernels).  For roofline pricing (bench.py); launches nothing. */
int dvg_mmd_spin_flops(int64_t nx, int64_t ny, int dim, double *int8_flops, double *bf16_flops, int *bf16_terms);
int dvg_mmd_fwd_bwd(const float *x, int64_t nx, const float *y, int64_t ny, int dim,
                    const dvg_mmd_cfg_t *cfg, float *loss_out /* device scalar */,
                    float *grad_x /* (nx, dim) or NULL */, void *ws, size_t ws_bytes,
                    dvg_stream_t stream);

/* ------------------------------------------------------------------ encoder
 * Encoder.forward / backward (/root/reference/src/encoder.py:18-49): 4 x
 * [conv3x3 -> BatchNorm2d -> MaxPool2d(2) -> LeakyReLU] (no LeakyReLU after the
 * last) with channels [1,32,64,128,n], then Linear(4,1) per channel.
 * Parameter tensors are in the checkpoint layout (conv weight (Cout,Cin,3,3)).
 * images (B,1,32,32) -> logits (B,n).  `ws` carries the saved activations from
 * fwd to bwd (size from dvg_encoder_workspace_bytes) and must stay untouched in
 * between.  training != 0: batch statistics, running stats updated in place
 * (momentum 0.1, unbiased variance), num_batches_tracked += 1.
 */
typedef struct {
  const float *conv_w[4], *conv_b[4]; /* Conv2d weight / bias */
  const float *bn_g[4], *bn_b[4];     /* BatchNorm2d weight / bias */
  float *bn_rm[4], *bn_rv[4];         /* running_mean / running_var (updated when training) */
  int64_t *bn_nbt[4];                 /* num_batches_tracked (may be NULL) */
  const float *proj_w, *proj_b;       /* Linear(4,1) */
} dvg_encoder_params_t;
typedef struct {
  float *conv_w[4], *conv_b[4], *bn_g[4], *bn_b[4], *proj_w, *proj_b;
} dvg_encoder_grads_t;
size_t dvg_encoder_workspace_bytes(int64_t B, int n_latents);
int dvg_encoder_fwd(const dvg_encoder_params_t *p, int n_latents, const float *images, int64_t B,
                    int training, float *logits, void *ws, size_t ws_bytes, dvg_stream_t stream);
int dvg_encoder_bwd(const dvg_encoder_params_t *p, int n_latents, const float *images, int64_t B,
                    const float *grad_logits, const dvg_encoder_grads_t *grads, void *ws,
                    size_t ws_bytes, dvg_stream_t stream);

/* ------------------------------------------------------------------ decoder
 * Decoder.forward / backward (/root/reference/src/decoder.py:18-62):
 * Linear(n,4n) -> (n,2,2) -> 4 x [ConvTranspose2d 3x3 -> BatchNorm2d ->
 * Dropout2d(0.2) -> Upsample x2 -> LeakyReLU] with channels [n,128,64,32,1] ->
 * ConvTranspose2d(1,1).  spins (N = B*R, n) -> out (N,1,32,32).
 * ConvTranspose weights in checkpoint layout (Cin,Cout,3,3).
 * dropout_keep[l]: (N, C_l) float {0,1} keep-masks, or NULL to draw them on
 * device (Philox stream DROPOUT, (seed, offset)); ignored unless training.
 */
typedef struct {
  const float *lin_w, *lin_b;         /* increase_latent_dim (4n,n), (4n) */
  const float *conv_w[5], *conv_b[5]; /* convtrans.{0,5,10,15,20} */
  const float *bn_g[4], *bn_b[4];
  float *bn_rm[4], *bn_rv[4];
  int64_t *bn_nbt[4];
} dvg_decoder_params_t;
typedef struct {
  float *lin_w, *lin_b, *conv_w[5], *conv_b[5], *bn_g[4], *bn_b[4];
} dvg_decoder_grads_t;
size_t dvg_decoder_workspace_bytes(int64_t N, int n_latents);
int dvg_decoder_fwd(const dvg_decoder_params_t *p, int n_latents, const float *spins, int64_t N,
                    int training, const float *const dropout_keep[4], uint64_t seed,
                    uint64_t offset, float *out, void *ws, size_t ws_bytes, const dvg_step_state_t *dyn,
                    dvg_stream_t stream);
/* The part of a forward call that depends on the parameters and the dropout stream alone (weight packs, the composed
 * Linear o ConvTranspose weights, the Dropout2d keep-masks), enqueued on `stream` ahead of time: a training step hands
 * in a stream forked off its main stream where the step starts, so that these launches (~190 us at B R = 32768) run
 * beside the encoder instead of between the spins and the decoder's first GEMM.  The next
 * dvg_decoder_fwd_ex(..., prepared = 1, ...) on the same workspace waits for them (an event recorded here) and skips
 * its own prologue; it fails unless it is called with the same parameters, N, training flag, seed / offset / dyn and
 * kernel-form options, which must not change in between.  A forward call with prepared = 0 (dvg_decoder_fwd) joins a
 * prologue in flight on its workspace and then ignores it.  Device-drawn dropout masks only. */
int dvg_decoder_prepare(const dvg_decoder_params_t *p, int n_latents, int64_t N, int training, uint64_t seed,
                        uint64_t offset, void *ws, size_t ws_bytes, const dvg_step_state_t *dyn,
                        dvg_stream_t stream);
int dvg_decoder_fwd_ex(const dvg_decoder_params_t *p, int n_latents, const float *spins, int64_t N,
                       int training, const float *const dropout_keep[4], uint64_t seed, uint64_t offset,
                       float *out, void *ws, size_t ws_bytes, const dvg_step_state_t *dyn, int prepared,
                       dvg_stream_t stream);
int dvg_decoder_bwd(const dvg_decoder_params_t *p, int n_latents, const float *spins, int64_t N,
                    const float *grad_out, const dvg_decoder_grads_t *grads,
                    float *grad_spins /* (N,n) or NULL */, void *ws, size_t ws_bytes,
                    dvg_stream_t stream);
/* The same with the join of the weight-gradient chain left to the caller: grad_spins is complete in `stream` order when
 * the call returns, the PARAMETER gradients only after dvg_stream_join_side(stream) or after the next dvg_encoder_bwd /
 * dvg_decoder_bwd call on `stream` has returned (their weight-gradient chains run on the same library side stream, in
 * order, and their own join covers what was queued before).  A training step whose decoder backward is followed by the
 * encoder backward overlaps the tail of the decoder's weight-gradient chain (the Linear layer's, ~1 ms at B R = 32768)
 * with the head of the encoder's data-gradient chain this way.  defer_join = 0 is dvg_decoder_bwd. */
int dvg_decoder_bwd_ex(const dvg_decoder_params_t *p, int n_latents, const float *spins, int64_t N,
                       const float *grad_out, const dvg_decoder_grads_t *grads, float *grad_spins, void *ws,
                       size_t ws_bytes, int defer_join, dvg_stream_t stream);
/* Orders `stream` behind everything queued so far on the library's side stream of the current device (no-op when the
 * side stream is disabled). */
int dvg_stream_join_side(dvg_stream_t stream);
/* The decoder with the reconstruction loss fused behind it -- Decoder.forward + mse_loss of the training step
 * (/root/reference/src/model_wrapper.py:297-305) as ONE pair of calls.  An addition to the surface above, not a
 * replacement: dvg_decoder_fwd_ex + dvg_mse_fwd_bwd + dvg_decoder_bwd_ex compute the same loss (to rounding of its double
 * partial sums) and the same gradients BIT FOR BIT (tests/test_gpu_nets.py); what the pair saves is memory traffic.
 * The reconstruction (N,1,32,32) and its gradient are never written: the final ConvTranspose2d(1,1), the squared error
 * against images (N / R, 1024) -- replica r of image b is decoder row b R + r --, the final layer's data gradient and the
 * 1-channel stage's BatchNorm / Dropout2d / LeakyReLU backward run in one pass per image, and the passes of the backward
 * call that need the loss gradient form it again from the stage's saved pre-BatchNorm output (1.1 GB -> 0.23 GB of HBM
 * traffic at N = 32768).  Training mode only; needs option dec_tail_fused != 0 (the default).
 * loss_out = mean((recon - image)^2); the gradient seeded into the backward is grad_scale * d loss / d recon, as in
 * dvg_mse_fwd_bwd, and dvg_decoder_bwd_mse_ex must be given the same images, R and grad_scale as the forward call on
 * this workspace.  Every other argument as in dvg_decoder_fwd_ex / dvg_decoder_bwd_ex. */
int dvg_decoder_fwd_mse_ex(const dvg_decoder_params_t *p, int n_latents, const float *spins, int64_t N,
                           const float *const dropout_keep[4], uint64_t seed, uint64_t offset, const float *images,
                           int R, float grad_scale, float *loss_out, void *ws, size_t ws_bytes,
                           const dvg_step_state_t *dyn, int prepared, dvg_stream_t stream);
int dvg_decoder_bwd_mse_ex(const dvg_decoder_params_t *p, int n_latents, const float *spins, int64_t N,
                           const float *images, int R, float grad_scale, const dvg_decoder_grads_t *grads,
                           float *grad_spins, void *ws, size_t ws_bytes, int defer_join, dvg_stream_t stream);

/* ------------------------------------------------------------------ MSE
 * torch.nn.functional.mse_loss(reconstructed, images.unsqueeze(1).repeat(1,R,...))
 * (/root/reference/src/model_wrapper.py:302-305), fused with its gradient:
 * loss_out = mean((recon - image)^2); grad_recon = grad_scale * 2 (recon - image) / numel.
 * recon (B,R,1024), images (B,1024).
 */
size_t dvg_mse_workspace_bytes(void);
int dvg_mse_fwd_bwd(const float *recon, const float *images, int64_t B, int R, float grad_scale,
                    float *loss_out, float *grad_recon /* or NULL */, void *ws, size_t ws_bytes,
                    dvg_stream_t stream);

/* ------------------------------------------------------------------ Adam
 * torch.optim.Adam with coupled L2 weight decay, as configured at
 * /root/reference/src/model_wrapper.py:208-217, on one flat parameter buffer.
 * g <- grad_scale*g + wd*p;  m,v updated; p -= lr/(1-b1^t) * m / (sqrt(v/(1-b2^t)) + eps).
 */
int dvg_adam_step(float *p, const float *g, float *m, float *v, int64_t numel, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int64_t step,
                  float grad_scale, const dvg_step_state_t *dyn, int dyn_slot, dvg_stream_t stream);

/* ------------------------------------------------------------------ stream anchor
 * Enqueues one empty kernel on `stream`.  For callers that capture a step into a hipGraph with a side stream forked
 * off `stream`: the HIP graph executor keeps the first KERNEL node captured after a fork on the parent's hardware
 * queue and replays nodes in capture order, so "record event; dvg_stream_anchor(stream); enqueue the side work;
 * enqueue the rest of the main work" keeps the main chain on its queue (a queue change costs ~10-15 us per hop, and
 * far more when it lands behind a long kernel) while the side work is still submitted early.
 */
int dvg_stream_anchor(dvg_stream_t stream);

/* ------------------------------------------------------------------ arithmetic mode of the convolution GEMMs
 * Process-wide switch read by dvg_encoder_fwd/bwd and dvg_decoder_fwd/bwd (a forward call and its backward call must
 * run in the same mode).  DVG_PRECISION_F32 (default): float32 operands, the <= 1e-5 relative loss parity against the
 * reference's CPU path.  DVG_PRECISION_BF16_INPUTS: "bf16 GEMM inputs, f32 accumulate" for the forward and
 * data-gradient GEMMs of encoder and decoder -- what a `torch.autocast(bfloat16)` run of src/encoder.py:28-30 /
 * src/decoder.py:28,34-38 would feed its convolutions; weight gradients, BatchNorm, losses and Adam stay float32.
 * DVG_PRECISION_F32_SPLIT3: float32 operands of the same GEMMs carried as three bf16 pieces each (x = hi + mid + lo,
 * exact) and multiplied as the six piece products down to 2^-16 on the bf16 MFMA with float32 accumulation: what is
 * dropped is below 2^-23 of a product, i.e. float32-class results (same parity bars as DVG_PRECISION_F32) at 6/16 of
 * the f32 MFMA's matrix time.
 * (The library reads no environment variable: the host side -- `CONV_PRECISION` in the YAML, bench.py --precision --
 * calls dvg_set_conv_precision.)
 */
#define DVG_PRECISION_F32 0
#define DVG_PRECISION_BF16_INPUTS 1
#define DVG_PRECISION_F32_SPLIT3 2
int dvg_set_conv_precision(int mode);
int dvg_get_conv_precision(void);

/* ------------------------------------------------------------------ kernel-form options
 * The library has ONE path per operation by default; a handful of named integer options select alternative kernel forms
 * of the same function (A/B measurements; tests that compare two forms).  They are part of this boundary -- nothing in
 * the library reads the environment.  Names and meanings: dvg_option_name(i) / dvg_option_doc(i), i < dvg_option_count();
 * INTEGRATION.md lists them.  Options may be changed between calls; a backward call whose forward ran under another plan
 * fails with DVG_E_INVALID.  dvg_reset_options restores every default. */
int dvg_option_count(void);
const char *dvg_option_name(int index);
const char *dvg_option_doc(int index);
int dvg_set_option(const char *name, int64_t value);
int dvg_get_option(const char *name, int64_t *value);
int dvg_reset_options(void);

/* ------------------------------------------------------------------ profiler
 * Optional per-kernel HIP-event timing inside the library (used by bench.py for
 * the `roofline` object).  Off by default.  `kernel_mask` bit i enables kernel id i
 * (ids/names: dvg_prof_num_kernels / dvg_prof_kernel_name); an enabled kernel gets
 * one event pair recorded on the caller's stream around each of its launches.
 */
int dvg_prof_enable(uint64_t kernel_mask);
int dvg_prof_reset(void);
int dvg_prof_num_kernels(void);
const char *dvg_prof_kernel_name(int id);
/* synchronises the recorded events; returns total milliseconds and launch count */
int dvg_prof_query(int id, double *total_ms, int64_t *launches);
/* algorithmic work (FLOPs for the GEMM-shaped kernels, 0 otherwise) summed over the timed launches */
int dvg_prof_query_work(int id, double *work);
/* sum over the kernel's launches of duration x the share of the chip's CUs the launch's grid was sized for (1 for every
 * kernel but the CU-budgeted persistent Winograd grids): the time base of roofline fractions "of the CUs it was given" */
int dvg_prof_query_share(int id, double *share_ms);

#ifdef __cplusplus
}
#endif
#endif /* DVG_H */
