/*
 * alphapig_hip.h -- C ABI of libalphapig_hip.so: the batched policy-value evaluator on
 * hand-written gfx950 (MI355X / CDNA4) HIP kernels.
 *
 * Plain C: pointers and sizes only.  "dev" pointers are device (HBM) addresses, "host"
 * pointers ordinary or pinned host memory.  One engine = one device + one HIP stream; an
 * engine is not thread-safe.  int-returning calls return >= 0 on success, a negative
 * APZ_E_* code on failure (text: apz_last_error()).  No exceptions cross the boundary.
 *
 * Reference interfaces replaced (file:line in the reference tree):
 *   PolicyValueNet.__init__ / create_policy_value_predict / set_params
 *                                   policy_value_net_mxnet.py:21-38, :214-230
 *   network graph (stem, residual blocks, heads)      policy_value_net_mxnet.py:70-102
 *                                                     policy_value_net_mxnet_simple.py:68-92
 *   PolicyValueNet.policy_value (batched forward)     policy_value_net_mxnet.py:232-242
 *   PolicyValueNet.policy_value_fn (H2D, forward, D2H) policy_value_net_mxnet.py:261-280
 *   Board.current_state (plane encoding, done on device from compact codes) game.py:68-94
 *   TrainPipeline.get_equi_data (8-fold dihedral augmentation) train_mxnet.py:115-135
 */
#ifndef ALPHAPIG_HIP_H
#define ALPHAPIG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APZ_OK 0
#define APZ_E_ARG (-1)
#define APZ_E_HIP (-2)       /* a HIP runtime call failed                        */
#define APZ_E_STATE (-3)     /* e.g. forward before weights were loaded           */
#define APZ_E_UNSUPPORTED (-4)

#define APZ_NET_RESNET 0     /* policy_value_net_mxnet.py        */
#define APZ_NET_SIMPLE 1     /* policy_value_net_mxnet_simple.py */

/* kernel classes for apz_kernel_time_ms */
#define APZ_K_STEM 0
#define APZ_K_TRUNK 1
#define APZ_K_HEAD_CONV 2
#define APZ_K_HEAD_FC 3
#define APZ_K_ENCODE 4
#define APZ_K_FORWARD 5   /* one whole forward (stem .. value head): EVERY forward while profiling is on, not only the sampled ones */
#define APZ_K_COUNT 6

typedef struct apz_engine apz_engine;

typedef struct apz_config {
    int32_t height;     /* board_height                                                 */
    int32_t width;      /* board_width                                                  */
    int32_t c_in;       /* input planes: 9 (channelnum, policy_value_net_mxnet.py:24) or 4 */
    int32_t n_filter;   /* trunk width (resnet), policy_value_net_mxnet.py:21           */
    int32_t n_blocks;   /* residual blocks (resnet)                                     */
    int32_t net_kind;   /* APZ_NET_*                                                    */
    int32_t max_batch;  /* largest n accepted by forward calls                          */
    int32_t device;     /* HIP device ordinal                                           */
} apz_config;

const char *apz_last_error(void);
int apz_version(void);
int apz_device_count(void);

apz_engine *apz_create(const apz_config *cfg);
void apz_destroy(apz_engine *e);

/* Parameter table (MXNet names, policy_value_loss.json): name and element count. */
int apz_param_count(apz_engine *e);
const char *apz_param_name(apz_engine *e, int i);
int64_t apz_param_size(apz_engine *e, int i);
/* Load RAW parameters (conv weight/bias, BN gamma/beta/moving stats, FC weight/bias) from
 * host float32 arrays; every table entry must be supplied.  BatchNorm (eps 1e-3,
 * fix_gamma semantics per layer) is folded into conv weights/biases inside. */
int apz_load_weights(apz_engine *e, const char *const *names, const float *const *host_ptrs,
                     const int64_t *sizes, int n);

/* Batched forward on the engine's stream, asynchronous.  planes_dev [n][c_in][H][W] f32.
 * probs_dev [n][H*W], values_dev [n]; logits_dev / vlogits_dev (pre-softmax / pre-tanh)
 * may be NULL. */
int apz_forward(apz_engine *e, const void *planes_dev, int n, void *probs_dev, void *values_dev,
                void *logits_dev, void *vlogits_dev);
/* Host-buffer convenience (H2D, forward, D2H, sync): policy_value / policy_value_fn path. */
int apz_forward_host(apz_engine *e, const float *planes_host, int n, float *probs_host,
                     float *values_host);
/* Self-play path: compact position codes (include/alphapig_host.h, apzh_code_stride bytes
 * per leaf) H2D, planes encoded on device, forward, D2H, sync. */
int apz_forward_codes_host(apz_engine *e, const uint8_t *codes_host, int n, float *probs_host,
                           float *values_host);
/* Asynchronous halves of the same, for overlapping host tree work with the GPU: buffers
 * must be pinned (apz_host_alloc) and stay valid until apz_sync() returns. */
int apz_forward_codes_async(apz_engine *e, const uint8_t *codes_pinned, int n, float *probs_pinned,
                            float *values_pinned);
/* Stream-ordered queue of up to APZ_MAX_SLOTS batches on the engine's ONE stream: submit copies
 * the codes into the slot's pinned buffer and enqueues H2D, encode, forward, D2H and an event;
 * wait blocks on that event and copies the slot's results out.  A second batch submitted while
 * the first runs starts the moment the first one's last kernel ends (no host round trip in
 * between).  submit calls are serialised by an internal lock; different slots may be submitted
 * and waited from different host threads. */
#define APZ_MAX_SLOTS 4
int apz_submit_codes(apz_engine *e, int slot, const uint8_t *codes_host, int n);
int apz_wait(apz_engine *e, int slot, float *probs_host, float *values_host);
void *apz_host_alloc(int64_t bytes);   /* pinned host memory */
void apz_host_free(void *p);

/* codes_dev [n][stride] u8 -> planes_dev [n][n_planes][H][W] f32 (n_planes 9 or 4). */
int apz_encode_planes(apz_engine *e, const void *codes_dev, int n, int n_planes, void *planes_dev);
/* 8-fold dihedral augmentation, reference order r1, r1f, r2, r2f, r3, r3f, id, idf:
 * planes_dev [n][c][H][W], pi_dev [n][H*W] -> planes_out_dev [n*8][c][H][W], pi_out_dev [n*8][H*W]. */
int apz_augment8(apz_engine *e, const void *planes_dev, const void *pi_dev, int n, int c,
                 void *planes_out_dev, void *pi_out_dev);

/* Root move sampling on the GPU (opt-in; NOT bit-compatible with the reference's NumPy stream):
 * visits_host [g][H*W] int32 with -1 where the root has no child; pi_host [g][H*W] gets
 * softmax(log(visits+1e-10)/temp) (mcts_alphaZero.py:152-155), moves_host[g] a draw from
 * (1-eps)*pi + eps*Dirichlet(alpha) (:198-201).  Deterministic in (seed, step, game index). */
int apz_sample_moves_host(apz_engine *e, const int32_t *visits_host, int g, float temp, float alpha,
                          float eps, uint64_t seed, uint64_t step, float *pi_host, int32_t *moves_host);
/* The same with one caller-chosen 64-bit key per row instead of (step, row): the draw of row i depends on
 * (seed, keys_host[i]) only.  The self-play engine passes (global game index << 20 | ply), so a game's noise is
 * reproducible whatever batch, rank or step it is sampled in (the reference's per-game reproducibility comes from
 * np.random.seed before a game, mcts_alphaZero.py:198-201).  keys_host == NULL: as apz_sample_moves_host. */
int apz_sample_moves_keyed_host(apz_engine *e, const int32_t *visits_host, int g, float temp, float alpha,
                                float eps, uint64_t seed, uint64_t step, const uint64_t *keys_host,
                                float *pi_host, int32_t *moves_host);

/* ---- training-side convolution primitives (hand-written forward, data- and
 * weight-gradient of the 3x3 convolutions, callable on caller-owned dense NCHW float32 device
 * tensors, e.g. from a torch.autograd.Function).  `stream` = the hipStream_t to launch on (NULL is
 * the null stream, which is what torch.cuda.current_stream().cuda_stream reports by default);
 * APZ_ENGINE_STREAM selects the engine's own stream.  The training entry points share per-engine scratch buffers: calls on
 * one stream are ordered by it, and a call on a DIFFERENT stream than the previous one is ordered (event + stream wait)
 * behind everything that was queued on the previous stream -- one engine never works on two streams at once.
 * Supported: boards 15x15 and 8x8; packed C_out in {64, 128, 256}.
 *   apz_conv3x3_pack   w_dev [cout][cin][3][3] -> wpk_dev (apz_conv3x3_packed_size floats);
 *                      transpose_flip=1 packs the weights of the data-gradient convolution
 *                      dX = conv(dY, W'), W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]
 *   apz_conv3x3_fwd    y = conv(x, wpk) (+ bias_dev if not NULL) (ReLU if relu); channels are
 *                      those of the PACKED convolution (cin_p inputs, cout_p outputs)
 *   apz_conv3x3_wgrad  dw_dev [cout][cin][3][3] = sum_b x_b (*) dy_b (overwritten) */
#define APZ_ENGINE_STREAM ((void *)(intptr_t)-1)
int64_t apz_conv3x3_packed_size(int cin_p, int cout_p);
int apz_conv3x3_pack(apz_engine *e, const void *w_dev, int cin, int cout, int transpose_flip,
                     void *wpk_dev, void *stream);
int apz_conv3x3_fwd(apz_engine *e, const void *x_dev, const void *wpk_dev, const void *bias_dev,
                    void *y_dev, int n, int cin_p, int cout_p, int relu, void *stream);
/* activation layouts of the training primitives: dense NCHW, or the trunk's padded rows [n][C][15][16]
 * (15x15 boards; pad column zero) in which the self-play kernels keep their activations */
#define APZ_LAYOUT_DENSE 0
#define APZ_LAYOUT_ROWS16 1
int apz_conv3x3_wgrad(apz_engine *e, const void *x_dev, const void *dy_dev, void *dw_dev, int n,
                      int cin, int cout, int layout, void *stream);
/* The same forward / data-gradient convolution for the trunk shape (128 -> 128 channels, 15x15) on the
 * fused Winograd F(4x4,3x3) kernel of the self-play path (csrc/trunk15_wino3.h, csrc/wino_common.h):
 *   apz_wino_pack   w_dev [128][128][3][3] -> upk_dev (apz_wino_packed_size floats), U = G g G^T in fp32
 *                   on the device; transpose_flip as in apz_conv3x3_pack
 *   apz_wino_conv   y = conv(x, upk) + bias_dev (NULL: none) (ReLU if relu), x / y dense [n][128][15][15]
 *                   (copied to / from the kernel's padded-row layout in engine-owned scratch) */
int64_t apz_wino_packed_size(void);
int apz_wino_pack(apz_engine *e, const void *w_dev, int transpose_flip, void *upk_dev, void *stream);
/* ... of `count` layers whose weights lie back to back (w_dev [count][128][128][3][3]) in both orientations, one launch:
 * upk_dev [count][2][apz_wino_packed_size()] (forward, data gradient).  A training step packs every trunk layer twice
 * (policy_value_net_mxnet.py:282-299: the weights change with every optimiser step). */
int apz_wino_pack_many(apz_engine *e, const void *w_dev, int count, void *upk_dev, void *stream);
int apz_wino_conv(apz_engine *e, const void *x_dev, const void *upk_dev, const void *bias_dev,
                  void *y_dev, int n, int relu, int layout, void *stream);
/* ... + resid_dev (padded-row layout only; NULL: none) before the ReLU: the data gradient of a residual block's first
 * convolution meets the skip gradient inside the kernel's epilogue (trunk15_wino3.h) */
int apz_wino_conv_add(apz_engine *e, const void *x_dev, const void *upk_dev, const void *bias_dev,
                      const void *resid_dev, void *y_dev, int n, int relu, int layout, void *stream);
/* The training forward of a trunk layer (policy_value_net_mxnet.py:41-56: Convolution, then BatchNorm): y = conv(x, upk)
 * + bias in the padded-row layout, and stats_dev [128][n][2] doubles = per (channel, board) the sum and the sum of
 * squares of the board's 225 outputs, taken in the kernel's epilogue.  apz_bn_fwd_stats = apz_bn_fwd that takes its
 * batch statistics from such a buffer (stats_dev NULL: computes them itself) instead of a pass over x.  mask_dev (NULL:
 * not wanted; padded-row layout): n * C * 60 bytes, bit e of byte i = [element 4 i + e of y > 0] -- apz_bn_bwd reads the
 * ReLU decisions from it instead of from the 16-times-larger y. */
int apz_wino_conv_stats(apz_engine *e, const void *x_dev, const void *upk_dev, const void *bias_dev, void *y_dev,
                        void *stats_dev, int n, void *stream);
int apz_bn_fwd_stats(apz_engine *e, const void *x_dev, const void *resid_dev, const void *gamma_dev,
                     const void *beta_dev, void *run_mean_dev, void *run_var_dev, void *y_dev, void *mean_dev,
                     void *invstd_dev, const void *stats_dev, void *mask_dev, int n, int C, int layout, int relu,
                     float momentum, float eps, void *stream);
/* Training-mode BatchNorm (+ residual) (+ ReLU), the reference's BatchNorm(eps = 1e-3) between the trunk's
 * convolutions (policy_value_net_mxnet.py:41-102), over n x C planes in `layout`:
 *   apz_bn_fwd  y = act((x - mean_c) * invstd_c * gamma_c + beta_c (+ resid)); batch statistics (biased
 *               variance) written to mean_dev / invstd_dev [C] for the backward pass; run_mean / run_var
 *               (may be NULL) updated as run = (1 - momentum) * run + momentum * batch (unbiased variance);
 *               gamma_dev NULL = 1 (the reference's fix_gamma layers)
 *   apz_bn_bwd  dz = dy (* [out > 0] with relu: from out_dev, or from mask_dev when given -- then out_dev may be NULL);
 *               dbeta = sum dz; dgamma = sum dz * xhat;
 *               dx = gamma * invstd * (dz - dbeta / M - xhat * dgamma / M); dres = dz (NULL: not wanted);
 *               dxsum_dev (NULL: not wanted): [apz_bn_bwd_splits(n, C, layout)][dxsum_ld] floats, dxsum_ld >= C;
 *               row s, column c gets the sum of channel c's dx over the boards of batch split s.  The column
 *               sums (apz_colsum) are the bias gradient of the convolution that produced x; a caller with
 *               several layers gives each its own C columns of one matrix and adds them all in one launch.
 * Channel sums: double precision, per (channel, batch split) partials (no atomics) that every consumer workgroup
 * adds in a fixed order (csrc/conv_train.h) -- two launches per call, the same bits on every run.  The training
 * entry points share per-engine scratch: calls on one engine must be ordered by ONE stream at a time.
 *   apz_colsum  out[j] = scale * sum_i in[i][j] over rows x cols floats, fixed order (double accumulation) */
int apz_bn_fwd(apz_engine *e, const void *x_dev, const void *resid_dev, const void *gamma_dev,
               const void *beta_dev, void *run_mean_dev, void *run_var_dev, void *y_dev, void *mean_dev,
               void *invstd_dev, int n, int C, int layout, int relu, float momentum, float eps, void *stream);
/* Weight gradient of the trunk shape (128 -> 128, 15x15) through the Winograd domain (csrc/wgrad_wino3.h; 3.6x fewer
 * MFMAs than apz_conv3x3_wgrad): x_dev / dy_dev in the padded-row layout [n][128][15][16], dw_dev [128][128][3][3]
 * overwritten. */
int apz_wgrad_wino(apz_engine *e, const void *x_dev, const void *dy_dev, void *dw_dev, int n, void *stream);
/* One optimiser step of the reference's Adam (policy_value_net_mxnet.py:198-205: rescale_grad = 1/batch_size,
 * wd on *_weight / *_gamma) over ntensors device tensors in ONE launch.  table_host: ntensors entries
 * { float *w; const float *g; float *m; float *v; int64 n; float wd; int32 pad; } (48 bytes) in HOST memory;
 * g = g * rescale + wd * w; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g*g; w -= lr_t * m / (sqrt(v) + eps),
 * lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) computed by the caller. */
int apz_adam_step(apz_engine *e, const void *table_host, int ntensors, float lr_t, float b1, float b2,
                  float eps, float rescale, void *stream);
int apz_bn_bwd(apz_engine *e, const void *dy_dev, const void *x_dev, const void *out_dev, const void *mask_dev,
               const void *gamma_dev, const void *mean_dev, const void *invstd_dev, void *dx_dev,
               void *dres_dev, void *dgamma_dev, void *dbeta_dev, void *dxsum_dev, int dxsum_ld, int n, int C,
               int layout, int relu, void *stream);
int apz_bn_bwd_splits(apz_engine *e, int n, int C, int layout);
int apz_colsum(apz_engine *e, const void *in_dev, void *out_dev, int rows, int cols, float scale, void *stream);

/* ---- heads and loss of the training graph (policy_value_net_mxnet.py:85-102, :173-193), csrc/heads_train.h.
 * All tensors float32 on the device; `layout` as above for the trunk-side tensors (x, dx); y / dy dense.
 *   apz_conv1x1_fwd / _bwd   the 1x1 head convolutions conv3_1_1 (C -> 4) and conv3_2_1 (C -> 2): y = W x + b;
 *                            dx (NULL: not wanted), dW [CO][C], db [CO] (NULL: not wanted); sums over the batch are
 *                            taken in index order (no float atomics)
 *   apz_fc_fwd / _bwd        FullyConnected: y[n][N] = x[n][K] W[N][K]^T + b; dx, dW, db (each may be NULL): one
 *                            fp32-MFMA GEMM kernel with operand strides
 *   apz_dropout              y = x * mask / keep, mask = [hash(seed, step, index) < keep]: the same call with dy gives
 *                            the backward pass (the mask is regenerated, not stored)
 *   apz_pv_loss              p = softmax(logits), v = tanh(vlogit); loss3 = (mean (z - v)^2, mean -sum pi log p,
 *                            mean -sum p log p); dlogits = (p sum(pi) - pi) / n, dvlogit = 2 (v - z)(1 - v^2) / n;
 *                            probs / values = p, v.  Every output pointer may be NULL.
 *   apz_layout_convert       dense [planes][15][15] <-> padded rows [planes][15][16]
 *   apz_bias_grad            db[c] = sum over boards and cells of dy[n][c][.] (the bias gradient of a convolution;
 *                            pad cells of a padded-row gradient are zero), fixed summation order
 *   apz_add                  y += x (count floats): the meeting point of the trunk and skip gradients
 * conv1x1_bwd: dx gets zero pad cells; accumulate_dx != 0 adds to what dx already holds (second head). */
/* apz_load_weights for tensors that already live in device memory (policy_value_net_mxnet.py:295-297: after an
 * optimiser step the new parameters go into the predict modules): BatchNorm folding and every packing of
 * apz_load_weights as kernels on `stream`; forwards queued afterwards use the new weights.  Same name / size table as
 * apz_load_weights; the engine must have been loaded once from the host. */
int apz_load_weights_dev(apz_engine *e, const char *const *names, const void *const *dev_ptrs, const int64_t *sizes,
                         int n, void *stream);
int apz_conv1x1_fwd(apz_engine *e, const void *x_dev, const void *w_dev, const void *bias_dev, void *y_dev, int n,
                    int C, int CO, int layout, void *stream);
int apz_conv1x1_bwd(apz_engine *e, const void *x_dev, const void *w_dev, const void *dy_dev, void *dx_dev,
                    void *dw_dev, void *db_dev, int n, int C, int CO, int layout, int accumulate_dx, void *stream);
/* ... of two 1x1 convolutions that share their input (the reference's two heads, policy_value_net_mxnet.py:85-96):
 * dx = W1^T dy1 + W2^T dy2 in one pass over x and dx; dw_dev [CO1 + CO2][C]: the first head's rows, then the second's. */
int apz_conv1x1_bwd2(apz_engine *e, const void *x_dev, const void *w1_dev, const void *dy1_dev, int CO1,
                     const void *w2_dev, const void *dy2_dev, int CO2, void *dx_dev, void *dw_dev, int n, int C,
                     int layout, int accumulate_dx, void *stream);
int apz_bias_grad(apz_engine *e, const void *dy_dev, void *db_dev, int n, int C, int layout, void *stream);
int apz_add(apz_engine *e, void *y_dev, const void *x_dev, int64_t count, void *stream);
int apz_fc_fwd(apz_engine *e, const void *x_dev, const void *w_dev, const void *bias_dev, void *y_dev, int n, int K,
               int N, void *stream);
int apz_fc_bwd(apz_engine *e, const void *x_dev, const void *w_dev, const void *dy_dev, void *dx_dev, void *dw_dev,
               void *db_dev, int n, int K, int N, void *stream);
int apz_dropout(apz_engine *e, const void *x_dev, void *y_dev, int64_t count, float keep, uint64_t seed,
                uint64_t step, void *stream);
int apz_pv_loss(apz_engine *e, const void *logits_dev, const void *vlogit_dev, const void *pi_dev, const void *z_dev,
                int n, void *loss3_dev, void *dlogits_dev, void *dvlogit_dev, void *probs_dev, void *values_dev,
                void *stream);
int apz_layout_convert(apz_engine *e, const void *src_dev, void *dst_dev, int64_t planes, int to_rows16, void *stream);

int apz_sync(apz_engine *e);
void *apz_stream(apz_engine *e);
void *apz_device_alloc(apz_engine *e, int64_t bytes);
void apz_device_free(apz_engine *e, void *p);
int apz_memcpy_h2d(apz_engine *e, void *dst_dev, const void *src_host, int64_t bytes);
int apz_memcpy_d2h(apz_engine *e, void *dst_host, const void *src_dev, int64_t bytes);

/* ---- measurement / test hooks ---------------------------------------------------------- */
/* Time conv layer `layer` (0 = stem, 1.. = trunk convs in graph order) alone at batch n on
 * synthetic resident inputs: `warmup` untimed + `iters` timed launches bracketed by HIP
 * events on the engine stream; ms_out[0] = average milliseconds per launch. */
int apz_conv3x3_bench(apz_engine *e, int layer, int n, int iters, int warmup, float *ms_out);
/* After a forward of batch n: copy the output activation of conv layer `layer`
 * ([n][C_out][H][W]) to host (per-layer parity tests). */
int apz_layer_io(apz_engine *e, int layer, float *host_out, int64_t count);
/* Per-kernel-class HIP-event timing over subsequent forwards: on = 1 times every forward,
 * on = k > 1 every k-th forward (the event records cost a few us per kernel, so sampled timing
 * keeps the measured run undisturbed); then read out[0] = total ms, out[1] = timed launches
 * (resolved at apz_sync). */
/* on != 0: apz_submit_codes replays the launch sequence of a small batch (<= 64 boards) as a HIP graph from the third
 * submission of a (slot, batch size) pair on (results unchanged: the same kernels with the same arguments).  Default: off
 * -- measured slower than the plain launches on ROCm 7.2 (profiles/r04_graph_ab.log). */
int apz_set_forward_graphs(apz_engine *e, int on);
int apz_set_profiling(apz_engine *e, int on);
/* TEST HOOK.  The 15x15 / 128-filter residual net has two trunk convolution kernels: the fused F(4x4,3x3) Winograd kernel
 * (default, csrc/trunk15_wino3.h) and the direct convolution (csrc/trunk15_ring.h) that the tests use as the in-tree
 * cross-check of the former (exact fp32 FMA chains, no transform).  Takes effect from the next forward. */
/* Arithmetic of the 128 -> 128 trunk convolutions of the 15x15 residual net (policy_value_net_mxnet.py:77-83), to be
 * chosen BEFORE apz_load_weights.  APZ_ARITH_F32 (default): exact fp32 products on the fp32 matrix pipe -- the bits the
 * parity tests rest on.  APZ_ARITH_BF16X3: batches of more than 32 boards run csrc/trunk15_wino3b.h -- every fp32 operand
 * as three bf16 terms, six bf16 products per fp32 product, fp32 accumulation: fp32-accurate (tests/
 * test_gpu_winograd_numerics.py), different low-order bits.  APZ_ARITH_F16X2 (round 6): batches of more than 32 boards run
 * csrc/trunk15_wino3h.h -- every fp32 operand as two fp16 terms (weights times a per-channel power of two), three products
 * per fp32 product on the fp16 matrix pipe, fp32 accumulation: the same accuracy class, 1.5x the exact kernel's speed.
 * An activation beyond the fp16 range (|x| > ~655) shows as a non-finite output; the kernel raises a word and the entry
 * point that collects the forward (apz_wait, apz_forward_host, apz_forward_codes_host, apz_forward, apz_forward_codes_async
 * -- the last two then return with the stream drained) repeats it on the exact-fp32 kernel: results are always finite-
 * checked, never silently wrong.  apz_trunk_overflows: how many forwards were repeated.
 * 8x8 boards (policy_value_net_mxnet_simple.py:68-92 and the 8x8 residual nets): APZ_ARITH_F16X2 runs every 3x3 convolution
 * with a multiple of 64 input channels on csrc/conv8_split.h (the same two-term split, no Winograd transform: activations up
 * to 65 504 are in range) for EVERY batch size, the first layer on conv8_kernel; same overflow word, same repeat.
 * APZ_ARITH_BF16X3 exists for the 15x15 / 128-filter net only (APZ_E_UNSUPPORTED elsewhere). */
#define APZ_ARITH_F32 0
#define APZ_ARITH_BF16X3 1
#define APZ_ARITH_F16X2 2
int apz_set_trunk_arith(apz_engine *e, int arith);
long apz_trunk_overflows(apz_engine *e);
#define APZ_TRUNK_DIRECT 0
#define APZ_TRUNK_WINOGRAD 3            /* default: batches of <= 32 boards take the small-batch form (csrc/trunk15_wino3s.h) */
#define APZ_TRUNK_WINOGRAD_BATCHED 4    /* ... the batched form for every batch size (the tests hold the two forms to bit equality) */
#define APZ_TRUNK_WINOGRAD_NO_QUARTER 5 /* ... the batched form with 64-channel work items only (no 32-channel items for few-pair batches) */
int apz_test_select_trunk(apz_engine *e, int kind);
int apz_kernel_time_ms(apz_engine *e, int kernel_class, float *out2);
/* Enqueue `iters` forwards of n empty boards on the engine's stream and return without waiting (apz_sync waits).
 * GPU-only warm-up for measurements: clocks, the runtime's event / signal pools, instruction caches.  The results go
 * to the engine's device buffers and are never read; pending submissions are not disturbed (same stream, in order). */
int apz_prewarm(apz_engine *e, int n, int iters);

#ifdef __cplusplus
}
#endif
#endif
