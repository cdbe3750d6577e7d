/* libecgmm_hip.so -- C ABI of the MI355X (gfx950) multimodal ECG hot path.
 *
 * The reference (hyeeiin/ECG-Multimodal-Model) is pure PyTorch and has no FFI; the boundary this
 * library sits behind is the torch.nn.Module / autograd contract of its model classes.  Each entry
 * point below therefore names the reference torch call it replaces (paths relative to the reference
 * tree; "PMB" = multimodal_paper_modal_balance.py).  The Python host (ecg-multimodal-model_amd/)
 * binds these with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; the caller (PyTorch) owns ALL memory.  No entry point
 *     allocates, frees, synchronises or retains a pointer after it returns.
 *   - every function enqueues on the hipStream_t passed as `stream` (void*; 0 = default stream).
 *   - return 0 on success, an ECGMM_ERR_* code otherwise; ecgmm_last_error() gives the message
 *     (thread-local).  The Python wrapper raises RuntimeError.
 *   - activations are channels-last ([N,H,W,C] / [N,L,C]) in the compute dtype (ECGMM_BF16 or
 *     ECGMM_F32); parameters, statistics, gradients and the dense tails are fp32.
 */
#ifndef ECGMM_H
#define ECGMM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECGMM_VERSION 100

enum { ECGMM_F32 = 0, ECGMM_BF16 = 1 };
enum { ECGMM_ACT_NONE = 0, ECGMM_ACT_RELU = 1, ECGMM_ACT_SIGMOID = 2 };
enum {
  ECGMM_OK = 0,
  ECGMM_ERR_SHAPE = 1,
  ECGMM_ERR_DTYPE = 2,
  ECGMM_ERR_WORKSPACE = 3,
  ECGMM_ERR_LAUNCH = 4
};

int ecgmm_version(void);
const char* ecgmm_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Encoder plans (the product path: one call = one encoder forward / backward)
 * ------------------------------------------------------------------------------------------- */
#define ECGMM_RESNET18_NPARAMS 62
#define ECGMM_RESNET18_NBUFFERS 60
typedef struct {
  int N, H, W;      /* image batch, NCHW fp32 */
  int out_dim;      /* fc out_features (PMB:221 -> image_dim; train_image_only.py:96 -> num_classes) */
  int dtype;        /* trunk compute dtype */
  int training;     /* BatchNorm batch statistics + running-stat update (model.train(), train.py:57) */
  float bn_momentum, bn_eps;
} ecgmm_resnet18_desc;

/* replaces self.image_encoder(image) -- torchvision resnet18 instantiated at PMB:210, called PMB:325;
 * ImageOnlyClassifier.forward, train_image_only.py:98 */
size_t ecgmm_resnet18_fwd_workspace(const ecgmm_resnet18_desc* d);
size_t ecgmm_resnet18_bwd_workspace(const ecgmm_resnet18_desc* d);
int ecgmm_resnet18_forward(const ecgmm_resnet18_desc* d, const float* image, const void* const* params,
                           void* const* buffers, float* feat_out, void* ws, size_t ws_bytes, void* stream);
/* replaces the image-encoder part of total_loss.backward() (train.py:80, train_image_only.py:131).
 * grads[i] == NULL skips that parameter (requires_grad=False, train.py:35-40). Gradients are WRITTEN
 * (not accumulated).  stages [begin,end): 0 = fc+avgpool, 1..8 = blocks 7..0, 9 = stem -- the split
 * lets the host launch a gradient all-reduce bucket between stages. */
int ecgmm_resnet18_backward(const ecgmm_resnet18_desc* d, const float* image, const float* dfeat,
                            const void* const* params, void* const* grads, void* ws_fwd, void* ws_bwd,
                            size_t ws_bwd_bytes, int stage_begin, int stage_end, void* stream);

/* The ResNet18 backward runs its weight-gradient kernels on a library-owned side stream (forked from /
 * joined to `stream` with events inside each call).  0 = keep everything on the caller's stream. */
int ecgmm_side_wgrad(int on);
/* Data-parallel overlap (parallel.py): defer = 1 makes ecgmm_resnet18_backward return without joining the side
 * stream; ecgmm_side_wait(stream) orders `stream` (the all-reduce stream) after the weight gradients issued so
 * far.  The last stage group of a backward must run with defer = 0. */
int ecgmm_side_defer_join(int defer);
/* the same switch for the ResNet1D_SE plan alone (ecgmm_side_wgrad sets both): a host that already runs the signal encoder
 * on a side stream of its own beside the image encoder turns it off -- more than four busy HIP streams share hardware queues */
int ecgmm_resnet1d_side_wgrad(int on);
int ecgmm_side_wait(void* stream);
/* The side stream itself (a hipStream_t, NULL when switched off) and "fork" (work enqueued on `stream` so far happens-before
 * later work on the side stream): a data-parallel host can issue its overlapped all-reduces from the side stream's
 * context instead of a stream of its own -- a fifth busy stream oversubscribes the four hardware queues. */
void* ecgmm_side_stream(void);
int ecgmm_side_fork(void* stream);

#define ECGMM_RESNET1D_NPARAMS 52
#define ECGMM_RESNET1D_NBUFFERS 27
typedef struct {
  int N, cin, L;        /* signal batch [N, cin, L] fp32 (ecg_signal.unsqueeze(1), PMB:328) */
  int num_classes;      /* classifier[4] out_features (PMB:226 -> signal_dim) */
  int dtype;
  int training;
  float bn_momentum, bn_eps;
  float dropout_p;      /* classifier[3] = Dropout(0.3), PMB:115; applied only when training */
  uint64_t seed, offset;
} ecgmm_resnet1d_desc;

/* replaces ResNet1D_SE.forward (PMB:119-125; signal_model.py:82-88; train_signal_12_af.py:204-210) */
size_t ecgmm_resnet1d_fwd_workspace(const ecgmm_resnet1d_desc* d);
size_t ecgmm_resnet1d_bwd_workspace(const ecgmm_resnet1d_desc* d);
int ecgmm_resnet1d_forward(const ecgmm_resnet1d_desc* d, const float* signal, const void* const* params,
                           void* const* buffers, float* feat_out, void* ws, size_t ws_bytes, void* stream);
/* stages: 0 = classifier+avgpool, 1..3 = layer3..layer1, 4 = stem */
int ecgmm_resnet1d_backward(const ecgmm_resnet1d_desc* d, const float* signal, const float* dfeat,
                            const void* const* params, void* const* grads, void* ws_fwd, void* ws_bwd,
                            size_t ws_bwd_bytes, int stage_begin, int stage_end, void* stream);

/* The multimodal head: everything ECGMultimodalModel.forward does after the three encoders, as ONE call per
 * direction (PMB:326-354 / multimodal.py:440-469: three LayerNorms, three branch Linear heads, AttentionFusion,
 * fusion_classifier = Linear-ReLU-Dropout-Linear, var_loss).  Same kernels as the per-op entry points below.
 * params (19, fp32): image_norm.{weight,bias}, signal_norm.{..}, clinical_norm.{..}, image_classifier.{weight,bias},
 * signal_classifier.{..}, clinical_classifier.{..}, attention_fusion.weights, attention_fusion.norm.{weight,bias},
 * fusion_classifier.0.{weight,bias}, fusion_classifier.3.{weight,bias}.
 * raw[3]: encoder outputs [B, dim[m]];  logits[4]: image / signal / clinical / fusion logits [B, num_classes];
 * var_loss: scalar;  soft_w: softmax of the 3 fusion weights. */
#define ECGMM_HEAD_NPARAMS 19
typedef struct {
  int B;
  int dim[3];        /* image_dim, signal_dim, clinical_dim (PMB:203-206: 256 each; multimodal.py:340-342: 512/128/32) */
  int hidden;        /* fusion_classifier.0 out_features (128, PMB:284) */
  int num_classes;
  int training;      /* Dropout active (model.train()) */
  float ln_eps;
  float dropout_p;   /* fusion_classifier.2 = Dropout(0.3), PMB:287 */
  uint64_t seed, offset;
} ecgmm_head_desc;
size_t ecgmm_head_fwd_workspace(const ecgmm_head_desc* d);
size_t ecgmm_head_bwd_workspace(const ecgmm_head_desc* d);
int ecgmm_head_forward(const ecgmm_head_desc* d, const float* const* raw, const void* const* params,
                       float* const* logits, float* var_loss, float* soft_w, void* ws, size_t ws_bytes, void* stream);
/* replaces the head part of total_loss.backward() (train.py:80).  dlogits[k] / dvar NULL = that output is not in the
 * loss (train.py:78 uses only fusion_logits and var_loss): the branch is skipped, its grads[] entries are left
 * untouched.  grads[i] NULL skips a parameter; draw[m] NULL = encoder m is frozen (train.py:35-40).  Gradients are
 * WRITTEN, not accumulated. */
int ecgmm_head_backward(const ecgmm_head_desc* d, const float* const* raw, const void* const* params,
                        void* const* grads, const float* const* dlogits, const float* dvar, float* const* draw,
                        void* ws_fwd, void* ws_bwd, size_t ws_bwd_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-op entry points (building blocks of the plans; also what the parity tests call one by one)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int N, H, W, Cin, Cout, R, S, stride, pad_h, pad_w;
} ecgmm_conv_desc;

/* layout helpers (the reference's tensors are NCHW / OIHW) */
int ecgmm_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int64_t HW, void* stream);
int ecgmm_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int64_t HW, void* stream);
int ecgmm_cast(int dtype, const float* src, void* dst, int64_t n, void* stream);
int ecgmm_uncast(int dtype, const void* src, float* dst, int64_t n, void* stream);
/* OIHW fp32 -> forward pack [Cout][R*S][Cin] and/or dgrad pack [Cin][R*S][Cout] (either may be NULL) */
int ecgmm_pack_conv_weight(int dtype, const float* w_oihw, void* fwd, void* dgrad, int Cout, int Cin, int RS,
                           void* stream);

/* nn.Conv2d / nn.Conv1d forward (F.conv2d inside torchvision BasicBlock; PMB:71,74,81 Conv1d).
 * stats (nullable): partial BatchNorm sums, ecgmm_conv_stats_rows(N*OH*OW) rows of [2][Cout]. */
int ecgmm_conv_stats_rows(int64_t out_pixels);
/* Stride-1 3x3 (pad 1) and 1x3 (pad 1) bf16 convolutions over whole 256-pixel tiles are served by the halo-resident
 * kernel (csrc/conv_halo.hip) instead of the general implicit-GEMM one: mode 0 = never, 1 = for the shapes it is
 * faster on (default), 2 = wherever it is applicable (A/B timing, tests).  Start-up value: ECGMM_CONV_HALO=0|1|2. */
int ecgmm_conv_halo_enable(int on);
/* Cap on the CUs (one persistent workgroup each) a halo-kernel launch occupies; 0 = all of them (default).  The partial-row
 * counts of its fused BatchNorm reductions follow the cap.  Start-up value: ECGMM_HALO_CUS. */
int ecgmm_conv_halo_cus(int cus);
/* 64 -> 64 channel 3x3 convolutions (ResNet18 layer 1) on 4-wave workgroups, two per CU, instead of one 8-wave
 * workgroup per CU: 0 = off (default: faster stand-alone, slower inside the overlapped step), 1 = on.
 * Start-up value: ECGMM_HALO_W4. */
int ecgmm_conv_halo_w4(int on);
/* Halo conv kernel: waves 4-7 of the 8-wave workgroup run each step's first MFMA block behind the step's barrier instead of
 * in front of it, so the two waves of a SIMD alternate between the matrix pipe and the LDS / fill issue instead of meeting
 * there: 1 = on (default), 0 = lock step.  Pure scheduling (bit-identical results).  Start-up value: ECGMM_HALO_STAGGER. */
int ecgmm_conv_halo_stagger(int on);
/* Halo conv kernel, 64 -> 64 channel 3x3 tiles (ResNet18 layer 1): STREAM form -- the K loop runs on across tile boundaries
 * (weight ring, next tile's halo and step-0 fragments already in flight) and a tile's epilogue runs inside the next tile's
 * first K steps: 1 = on (default), 0 = one tile at a time.  Bit-identical results.  Start-up value: ECGMM_HALO_STREAM. */
int ecgmm_conv_halo_stream(int on);
/* Halo conv kernel, 128-channel tiles: ping-pong K loop (each K step as four read | MFMA phases, waves 4-7 one barrier behind
 * waves 0-3, so one wave of every SIMD feeds the matrix pipe while its partner reads LDS / issues fills): 1 = on (default),
 * 0 = the lock-step loop.  Bit-identical results.  Start-up value: ECGMM_HALO_PP. */
int ecgmm_conv_halo_pingpong(int on);
/* Ring weight-gradient kernel (3x3, two 4-wave groups per workgroup): the groups run half a K step apart (a second barrier
 * in the middle of the step, one barrier of offset), so one group's MFMA-only half pairs with the other's fill issue /
 * fragment-address head: 1 = on, 0 = lock step (default: measured 1 % slower with it).  Bit-identical results.
 * Start-up value: ECGMM_WGRAD_PP. */
int ecgmm_conv_wgrad_pingpong(int on);
/* Weight gradients of the same stride-1 3x3 / 1x3 bf16 convolutions keep their x operand in an LDS ring of pixel rows
 * (wgrad_ring_kernel, csrc/conv_wgrad.hip) instead of one gathered tile per filter tap: 0 = never, 1 = for the shapes
 * it is faster on (default), 2 = wherever applicable (A/B, tests).  Start-up value: ECGMM_WGRAD_RING=0|1|2. */
int ecgmm_conv_wgrad_ring_enable(int on);
int ecgmm_conv_fwd(int dtype, const ecgmm_conv_desc* c, const void* x, const void* w_fwd, const float* bias, void* y,
                   float* stats, int act, void* stream);
/* the same with the BatchNorm partial sums as ONE row per workgroup where the halo-resident kernel serves the shape (*nrows
 * rows of [2][Cout], at most 512; otherwise the per-64-pixel rows above and their count): what the encoder plans call.
 * stats: room for ecgmm_conv_stats_rows(N*OH*OW) + ECGMM_BN_TAIL_ROWS rows. */
int ecgmm_conv_fwd_wgrows(int dtype, const ecgmm_conv_desc* c, const void* x, const void* w_fwd, const float* bias, void* y,
                          float* stats, int* nrows, int act, void* stream);
/* autograd of the above: input gradient (addend, nullable, is added to dx) and weight gradient */
int ecgmm_conv_bwd_data(int dtype, const ecgmm_conv_desc* c, const void* dy, const void* w_dgrad, const void* addend,
                        void* dx, void* stream);
size_t ecgmm_conv_bwd_weight_workspace(int dtype, const ecgmm_conv_desc* c);
int ecgmm_conv_bwd_weight(int dtype, const ecgmm_conv_desc* c, const void* x, const void* dy, float* dw_oihw,
                          int accumulate, void* ws, size_t ws_bytes, void* stream);

/* stem convolutions straight from NCHW / NCL fp32 input: resnet18.conv1 (R = 7) and
 * ResNet1D_SE.initial[0] (R = 1, H = 1; PMB:100) */
size_t ecgmm_stem_packed_elems(int Cin, int R);
int ecgmm_stem_stats_rows(int N, int Cin, int H, int W, int R);
int ecgmm_stem_pack(int dtype, const float* w_oihw, void* packed, int Cin, int R, void* stream);
int ecgmm_stem_fwd(int dtype, const float* x, const void* packed, const float* bias, void* y, float* stats, int N,
                   int Cin, int H, int W, int R, void* stream);
/* bf16 form of ecgmm_stem_fwd whose BatchNorm partial sums stay in registers across a workgroup's tiles: 4 rows per
 * workgroup (ecgmm_stem_wg_stats_rows) instead of 4 per tile -- what the encoder plans call */
int ecgmm_stem_wg_stats_rows(int N, int Cin, int H, int W, int R);
int ecgmm_stem_fwd_wgrows(int dtype, const float* x, const void* packed, const float* bias, void* y, float* stats, int N,
                          int Cin, int H, int W, int R, void* stream);
size_t ecgmm_stem_bwd_weight_workspace(int N, int Cin, int H, int W, int R);
int ecgmm_stem_bwd_weight(int dtype, const float* x, const void* dy, float* dw_oihw, int accumulate, void* ws,
                          size_t ws_bytes, int N, int Cin, int H, int W, int R, void* stream);

/* The 2-D stem BY RECOMPUTE (csrc/conv_stem_fused.hip; bf16, 7x7 stride 2, Cin <= 3): conv1 -> bn1 -> relu -> maxpool of
 * torchvision's resnet18 (the reference's image encoder, multimodal_paper_modal_balance.py:210) without the
 * full-resolution conv output ever reaching HBM.
 *   stem_stats_only   image -> BatchNorm partial sums ([rows][2][64], rows = ecgmm_stem_stats_only_rows; finalize with
 *                     ecgmm_bn_finalize over N*OH*OW values)
 *   stem_pool_fwd     image -> conv -> bn(coef) -> relu -> MaxPool(3,2,1): pooled [N,PH,PW,64] bf16 + arg-max bytes; the
 *                     same values as ecgmm_stem_fwd + ecgmm_bnrelu_maxpool
 *   stem_pool_bwd     gradient w.r.t. pooled -> dgamma, dbeta of bn1 and (dw_oihw non-null) the conv weight gradient: the
 *                     conv is recomputed per tile, the max-pool backward gathered, the BatchNorm backward applied and the
 *                     weight-gradient products taken without writing y or dy.  Same results as ecgmm_pool_bn_bwd +
 *                     ecgmm_stem_bwd_weight up to fp32 summation order. */
int ecgmm_stem_stats_only_rows(int N, int Cin, int H, int W, int R);
int ecgmm_stem_stats_only(int dtype, const float* x, const void* packed, const float* bias, float* stats, int N, int Cin,
                          int H, int W, int R, void* stream);
int ecgmm_stem_pool_fwd(const float* x, const void* packed, const float* coef, void* pooled, uint8_t* idx, int N, int Cin,
                        int H, int W, void* stream);
size_t ecgmm_stem_pool_bwd_workspace(int N, int Cin, int H, int W);
int ecgmm_stem_pool_bwd(const float* x, const void* packed, const float* coef, const float* gamma, const void* dp,
                        const void* pooled, const uint8_t* idx, float* dgamma, float* dbeta, float* dw_oihw, void* ws,
                        size_t ws_bytes, int N, int Cin, int H, int W, void* stream);

/* nn.BatchNorm{1,2}d (train: batch mean / biased var, running update with unbiased var; eval: running
 * stats).  coef = [4][C]: scale, shift, mean, invstd.
 * Partial-sum buffers ([rows][2][C] floats, written by conv_fwd / stem_fwd / col_stats) must be
 * allocated with ECGMM_BN_TAIL_ROWS spare rows after `rows`: bn_finalize folds long buffers into
 * that tail before the final reduction. */
#define ECGMM_BN_TAIL_ROWS 64
int ecgmm_col_stats_rows(int dtype, int64_t M, int C);
int ecgmm_col_stats(int dtype, const void* x, int64_t M, int C, float* partial, void* stream);
int ecgmm_bn_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                      float* coef, void* stream);
int ecgmm_bn_eval_coef(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* coef, void* stream);
/* out = relu?( bn(y) * gate[n][c] + residual ), residual = res or res*rcoef.scale + rcoef.shift
 * (BasicBlock tail; BasicBlock1D tail with the SE gate, PMB:88-93) */
int ecgmm_bn_act(int dtype, const void* y, const float* coef, const void* res, const float* rcoef, const float* gate,
                 int rows_per_sample, int relu, void* out, int64_t M, int C, void* stream);
/* ecgmm_bn_finalize + ecgmm_bn_act as ONE launch: every workgroup of the activation pass folds the (<= 512) partial rows
 * itself -- no dependent finalize launch in between; workgroup 0 writes coef_out [4][C] (for the backward) and updates the
 * running statistics.  More rows / unsupported widths fall back to the two launches.  Same values. */
int ecgmm_bn_act_from_rows(int dtype, const void* y, const float* partial, int rows, double count, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float momentum, float eps, float* coef_out, const void* res, const float* rcoef,
                           const float* gate, int rows_per_sample, int relu, void* out, int64_t M, int C, void* stream);
/* BatchNorm backward with the ReLU mask / SE gate folded in:
 *   dz = [maskref > 0] * dout * gate[n][c] + addc[n][c];  dy, dgamma, dbeta; dz_out = masked dout;
 *   (maskref == y: the mask is recomputed as bn(y) > 0, saving the read of the activated tensor)
 *   dbias (nullable) = column sum of dy (Conv1d bias gradient). scratch: ecgmm_bn_bwd_scratch bytes */
size_t ecgmm_bn_bwd_scratch(int dtype, int64_t M, int C);
int ecgmm_bn_bwd(int dtype, const void* dout, const void* maskref, const float* gate, const float* addc,
                 int rows_per_sample, const void* y, const float* coef, const float* gamma, float* dgamma,
                 float* dbeta, void* dy, void* dz_out, float* dbias, int64_t M, int C, void* scratch, void* stream);

/* The same backward with its reduction pass fused into the convolution that PRODUCES dout (autograd of
 * conv -> BatchNorm -> ReLU chains: torchvision BasicBlock, multimodal_paper_modal_balance.py:210; train.py:80):
 * ecgmm_conv_bwd_data_bnred is ecgmm_conv_bwd_data that also accumulates, in its epilogue, the partial rows
 * (sum g, sum g * (bn_y - mean)) of g = [mask] * dx over the pixels, bn_mask == bn_y (or NULL): mask = (bn(bn_y) > 0)
 * and dx is stored unmasked; otherwise mask = (bn_mask > 0) and dx is stored MASKED.  rows: room for 512 x [2][Cin]
 * floats.  *nrows = rows written, or 0 when this geometry is not served by the fused kernel (dx is then the plain
 * gradient and the caller runs ecgmm_bn_bwd).  ecgmm_bn_bwd_from_rows finishes the backward from such rows
 * (finalize + apply; maskref as in ecgmm_bn_bwd, NULL for an already masked dout). */
int ecgmm_conv_bwd_data_bnred(int dtype, const ecgmm_conv_desc* c, const void* dy, const void* w_dgrad,
                              const void* addend, void* dx, const void* bn_y, const void* bn_mask, const float* bn_coef,
                              float* rows, int* nrows, void* stream);
/* rows a launch of ecgmm_conv_bwd_data_bnred writes for this geometry under the current settings (0: the fused form does
 * not apply and *nrows will be 0) -- for callers whose consumer runs in another call; follows ecgmm_conv_halo_cus */
int ecgmm_conv_bwd_data_bnred_rows(int dtype, const ecgmm_conv_desc* c);
/* Input gradient of a ResNet stage-entry block's two stride-2 branches in one launch (torchvision BasicBlock with
 * downsample; BasicBlock1D, multimodal_paper_modal_balance.py:71-93):  dx = dgrad(conv c, dy) + dgrad(1x1 stride-2 pad-0
 * conv of the same input, dy_down).  The 1x1 branch is one more tap of the stride-2 kernel's parity class (0,0); no
 * temporary is written.  c describes the main convolution (3x3 or 1x3, stride 2, pad <= 1); both gradients are
 * [N][OH][OW][Cout], w_down_dgrad is the 1x1 weight packed for dgrad.  tmp (same size as dx) is only used for geometries
 * the folded form does not serve; NULL is accepted when c has stride 2 and pad <= 1. */
int ecgmm_conv_bwd_data_with_downsample(int dtype, const ecgmm_conv_desc* c, const void* dy, const void* w_dgrad,
                                        const void* dy_down, const void* w_down_dgrad, void* dx, void* tmp,
                                        void* stream);
int ecgmm_bn_bwd_from_rows(int dtype, const void* dout, const void* maskref, const void* y, const float* coef,
                           const float* gamma, float* dgamma, float* dbeta, void* dy, const float* rows, int nrows,
                           int64_t M, int C, void* scratch, void* stream);

/* ResNet18 plan: a BatchNorm-backward reduction is fused into the producing dgrad's epilogue only for layers with at
 * least this many output pixels.  Default (and any negative m): never -- since round 3 the stream form of the layer-1 input
 * gradient plus the separate reduction pass is faster than the fused epilogue; 400000 = the 56x56 stage at batch >= 128
 * (round 2's default), 0 = wherever the halo kernel runs.  Start-up value: ECGMM_BN_FUSE_MIN_M.  Results differ by fp32
 * summation order only. */
int ecgmm_bn_fuse_min_pixels(int64_t m);
/* BatchNorm finalize folded into its consumer pass (every workgroup of bn_act / the backward's apply pass folds the <= 512
 * partial rows itself instead of a separate ~5 us launch in between; csrc/elementwise.hip): 1 = on (default), 0 = separate
 * launches.  Start-up value: ECGMM_BN_FOLD.  Same values up to the fp64 summation grouping of the partial rows. */
int ecgmm_bn_fold(int on);
/* ResNet18 plan: run the stem by recompute (ecgmm_stem_stats_only / stem_pool_fwd / stem_pool_bwd; bf16 only): 1 = on,
 * 0 = the two-pass route that keeps the full-resolution conv output (default: 0.2 ms per step faster at batch 256 although
 * it moves 1.6 GB more -- the recomputing kernels are instruction-bound).  Start-up value: ECGMM_STEM_RECOMPUTE.
 * It changes the plan's workspace layout: switch between steps, never between a forward and its backward. */
int ecgmm_stem_recompute(int on);

/* relu(bn(y)) -> MaxPool(3,2,1) (resnet18.maxpool; ResNet1D_SE.initial[3], PMB:103) and its backward */
int ecgmm_bnrelu_maxpool(int dtype, const void* y, const float* coef, void* out, uint8_t* idx, int N, int H, int W,
                         int C, void* stream);
int ecgmm_maxpool_relu_bwd(int dtype, const void* dp, const void* pooled, const uint8_t* idx, void* dz, int N, int H,
                           int W, int C, void* stream);
/* Backward of [BatchNorm -> ReLU -> MaxPool(3,2,1)] in one call, without the full-resolution pooled-gradient tensor: the
 * BatchNorm reduction runs over the pooled tensors (dp, pooled), the apply pass gathers the max-pool backward on the fly.
 * Same results as ecgmm_maxpool_relu_bwd + ecgmm_bn_bwd up to the rounding of `pooled` (header of the kernels in
 * csrc/elementwise.hip).  scratch: ecgmm_bn_bwd_scratch(dtype, N*H*W, C) bytes.  dbias nullable (sum of dy). */
int ecgmm_pool_bn_bwd(int dtype, const void* dp, const void* pooled, const uint8_t* idx, const void* y, const float* coef,
                      const float* gamma, float* dgamma, float* dbeta, void* dy, float* dbias, int N, int H, int W, int C,
                      void* scratch, void* stream);
/* AdaptiveAvgPool(1) (+ optional per-channel affine of the mean) and its broadcast backward */
int ecgmm_avgpool(int dtype, const void* x, float* out, int N, int R, int C, const float* coef, void* stream);
int ecgmm_bcast_rows(int dtype, const float* v, void* out, int N, int R, int C, float scale, void* stream);
int ecgmm_se_gate_grad(int dtype, const void* dout, const void* maskref, const void* y, const float* coef, float* dg,
                       int N, int R, int C, void* stream);

/* nn.Linear (+bias, +ReLU/Sigmoid) fp32: MFMA (exact f32) when the shape allows, VALU otherwise.
 * (clinical_encoder PMB:256-262, fusion_classifier PMB:283-289, branch heads PMB:269-271, SE fc PMB:53-58) */
int ecgmm_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, int act,
                     void* stream);
size_t ecgmm_linear_bwd_scratch(int B, int In, int Out);
int ecgmm_linear_bwd(const float* dz, const float* x, const float* w, float* dx, float* dw, float* db, int B, int In,
                     int Out, void* scratch, size_t scratch_bytes, void* stream);
int ecgmm_act_bwd(const float* dy, const float* y, float* dz, int64_t n, int act, void* stream);

/* nn.LayerNorm (PMB:223,239,263) and AttentionFusion (PMB:31-46): softmax(weights) * feats -> cat ->
 * LayerNorm, one row kernel.  nseg = 1 (plain LN, fusion_w = NULL) or 3 (fusion). stat = [B][2]. */
int ecgmm_layernorm_fwd(const float* const* seg, const int* dims, int nseg, const float* fusion_w, const float* gamma,
                        const float* beta, float* out, float* stat, float* soft_w, int B, float eps, void* stream);
size_t ecgmm_layernorm_bwd_scratch(int B, int D);
int ecgmm_layernorm_bwd(const float* const* seg, const int* dims, int nseg, const float* fusion_w, const float* gamma,
                        const float* stat, const float* dout, float* const* dseg, int dseg_accumulate, float* dgamma,
                        float* dbeta, float* dfusion_w, int B, void* scratch, void* stream);

/* var_loss (PMB:349-352). scratch = (3*B + 4) floats, kept for the backward */
int ecgmm_varloss_fwd(const float* f0, const float* f1, const float* f2, int B, int D0, int D1, int D2, float* loss,
                      float* scratch, void* stream);
int ecgmm_varloss_bwd(const float* f, int B, int D, const float* gout, const float* scratch, int which, float* df,
                      int accumulate, void* stream);

/* nn.CrossEntropyLoss() (train.py:31) / FocalLoss (signal_model.py:91-106), mean reduction.
 * dcoef = [B] floats kept for the backward; labels are int64 */
int ecgmm_ce_fwd(const float* logits, const int64_t* labels, int B, int C, int focal, float alpha, float gamma,
                 float* loss, float* dcoef, void* stream);
int ecgmm_ce_bwd(const float* logits, const int64_t* labels, int B, int C, const float* dcoef, const float* gout,
                 float* dlogits, void* stream);
/* the step loss of train.py:69-78 in one launch per direction: loss = CrossEntropy(logits, labels) + extra_w * extra[0]
 * (extra = var_loss, extra_w = 0.1); backward: dlogits and dextra[0] = gout * extra_w */
int ecgmm_ce_plus_fwd(const float* logits, const int64_t* labels, int B, int C, const float* extra, float extra_w,
                      float* loss, float* dcoef, void* stream);
int ecgmm_ce_plus_bwd(const float* logits, const int64_t* labels, int B, int C, const float* dcoef, const float* gout,
                      float* dlogits, float* dextra, float extra_w, void* stream);

/* nn.Dropout (PMB:115,260,287): Philox4x32-10, keep-mask bytes saved for the backward */
int ecgmm_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset,
                      void* stream);
int ecgmm_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, void* stream);

/* torch.optim.Adam.step (train.py:43,81; betas/eps defaults; beta1 varies under OneCycleLR,
 * train_signal_12_af.py:250-252) over one contiguous fp32 run; gscale multiplies the gradient
 * (1/world_size after a summed all-reduce) */
int ecgmm_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int64_t step, float gscale, void* stream);
int ecgmm_axpby(float a, const float* x, float b, float* y, int64_t n, void* stream);

/* SURVEY 8(f1) -- the step in front of the hot path, on device: per-signal StandardScaler (optional,
 * per time column) -> moving-average baseline removal (np.convolve(x, ones(w)/w, 'same')) -> IIR
 * low-pass applied forward-backward exactly as scipy.signal.filtfilt(b, a, x) (odd padding 3*(order+1),
 * lfilter_zi initial conditions `zi`).  Replaces preprocess_signal: dataset.py:81-95,
 * train_signal_12_af.py:19-34.  x/out: [S][L] fp32 (S = batch * leads); fp64 arithmetic inside. */
size_t ecgmm_signal_preprocess_workspace(int S, int L, int order);
int ecgmm_signal_preprocess(const float* x, float* out, int S, int L, const float* sc_mean, const float* sc_scale,
                            int window, const double* b, const double* a, const double* zi, int order, void* ws,
                            size_t ws_bytes, void* stream);

/* SURVEY 8(f3) -- the TabNet clinical encoder of multimodal.py:109-148 (pytorch_tabnet.tab_network.TabNetNoEmbeddings:
 * third-party, source and version absent from the reference; restated from its published algorithm).  Row kernels,
 * fp32: GLU gate out = z[:, :D] * sigmoid(z[:, D:]) of a [N, 2D] tensor; sparsemax along rows of [N, D], D <= 64;
 * elementwise helpers of the mask / prior recurrence; mean_n sum_d M log(M + eps). */
enum { ECGMM_EW_MUL = 0, ECGMM_EW_ADD_SCALE = 1, ECGMM_EW_PRIOR = 2, ECGMM_EW_RELU = 3, ECGMM_EW_RELU_BWD = 4,
       ECGMM_EW_SCALE = 5, ECGMM_EW_NEG_MUL = 6, ECGMM_EW_ADD = 7, ECGMM_EW_RSUB = 8 };
int ecgmm_glu_fwd(const float* z, float* out, int64_t N, int D, void* stream);
int ecgmm_glu_bwd(const float* z, const float* dout, float* dz, int64_t N, int D, void* stream);
int ecgmm_sparsemax_fwd(const float* x, float* p, int64_t N, int D, void* stream);
int ecgmm_sparsemax_bwd(const float* p, const float* dp, float* dx, int64_t N, int D, void* stream);
int ecgmm_ew(int op, const float* a, const float* b, float* out, int64_t n, float s, void* stream);
/* x [N, D] -> d = x[:, :nd] (through ReLU when relu != 0), a = x[:, nd:]; bwd merges (gd / ga may be NULL = zero) */
int ecgmm_split_cols(const float* x, float* d, float* a, int64_t N, int D, int nd, int relu, void* stream);
int ecgmm_split_cols_bwd(const float* d, const float* gd, const float* ga, float* gx, int64_t N, int D, int nd, int relu,
                         void* stream);
/* BatchNorm1d over the rows of a small [N, C] fp32 matrix, any C (one block per channel): the 2- / 64- / 128-wide
 * (ghost) batch norms of TabNet.  save = [2][C] mean, invstd.  accumulate != 0 adds into dgamma / dbeta. */
int ecgmm_bn_small_fwd(const float* x, const float* gamma, const float* beta, float* rm, float* rv, long long* nbt,
                       float* y, float* save, int N, int C, int training, float momentum, float eps, void* stream);
int ecgmm_bn_small_bwd(const float* x, const float* dy, const float* gamma, const float* save, float* dx, float* dgamma,
                       float* dbeta, int N, int C, int accumulate, void* stream);
int ecgmm_entropy_fwd(const float* M, float* out, int64_t N, int D, float eps, void* stream);
int ecgmm_entropy_bwd(const float* M, const float* g, float* dM, int64_t N, int D, float eps, void* stream);

/* SURVEY 8(f1), image half: Resize((OH, OW)) -> ToTensor -> Normalize(mean, std) of decoded RGB pictures
 * (dataset.py:61,119-123; train_image_only.py:58-62; dataset_image.py:67-70 when OH==H && OW==W).
 * torchvision hands a PIL picture to Pillow's antialiased BILINEAR resample (8 bpc, two passes, 22-bit
 * fixed-point coefficients, uint8 intermediate); the result is bit-identical to Pillow followed by
 * float32 x/255 and (x-mean)/std.  img: uint8 [B][H][W][3] on the device; out: fp32 [B][3][OH][OW].
 * The coefficient table is built on the host (tables_bytes / tables) and copied to the device by the
 * caller once per (H, W, OH, OW); it is not needed (may be NULL) when no resize takes place. */
size_t ecgmm_image_resize_tables_bytes(int H, int W, int OH, int OW);
int ecgmm_image_resize_tables(int H, int W, int OH, int OW, void* host_tables, size_t bytes);
int ecgmm_image_transform(const void* img, float* out, int B, int H, int W, int OH, int OW, const void* dev_tables,
                          size_t table_bytes, const float* mean3, const float* std3, void* stream);

/* Measurement only (no reference counterpart): HIP-event timing of the conv kernels on their launch
 * stream.  kinds: 0 igemm fwd, 1 igemm dgrad, 2 wgrad, 3 stem fwd, 4 stem wgrad, 5 / 6 igemm fwd / dgrad of the exact-fp32
 * instantiation.  collect()
 * synchronises the recorded events and returns per-kind total ms / algorithmic FLOPs / algorithmic
 * HBM bytes / launches.  enable(1) times every kind, enable(2) only kinds 0 and 1, enable(3) only kinds 5 and 6, enable(0) switches it off. */
int ecgmm_prof_enable(int on);
/* pause(1) suspends the bracketing, pause(0) resumes it; recorded launches are kept (bench.py samples steps). */
int ecgmm_prof_pause(int paused);
int ecgmm_prof_collect(int nkinds, double* ms, double* flops, double* bytes, int64_t* count);
/* Diagnostic step timeline: with ecgmm_tl_enable(1) the ResNet18 plan records a timestamped event on the caller's stream at
 * every phase boundary (ids 100.. forward: 100 start, 101 stem, 102..109 blocks, 110 end; 200.. backward: 200 start, 201 fc,
 * 202..209 blocks 7..0, 210 stem, 211 joined); ecgmm_tl_mark adds the host's own marks; ecgmm_tl_collect returns ids and
 * milliseconds since the first mark and clears the list. */
int ecgmm_tl_enable(int on);
int ecgmm_tl_mark(int id, void* stream);
int ecgmm_tl_collect(int cap, int* ids, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* ECGMM_H */
