// Internal (C++) launch interface between the kernel files, the encoder plans and the C-ABI layer.
#pragma once
#include "common.h"

// conv_igemm.hip
int ecg_conv_stats_rows(long M);
// Optional epilogue extras of a convolution launch (the encoder plans use them; the C ABI passes none).
struct ConvEpi {
  // in -- dgrad only: fuse the REDUCTION pass of the BatchNorm backward that consumes this gradient into the epilogue
  // (done by the halo kernel only: check red_done).  g = [mask] * dst as stored; rows of (sum g, sum g * (y - mean)).
  const void* red_y;      // raw conv output that BatchNorm normalised (same [pixels][channels] layout as dst), or null
  const void* red_mask;   // null / == red_y: mask = (bn(y) > 0), dst stored unmasked; else mask = (red_mask > 0), dst stored MASKED
  const float* red_coef;  // that BatchNorm's forward coefficients [4][C]
  float* red_rows;        // [>= 512 rows][2][C] (one row per workgroup of a channel tile; 64-channel tiles run two workgroups per CU)
  int wg_rows;            // in: 1 = the launch may write ONE partial row per workgroup instead of one per 64 pixels
  // in -- stride-2 dgrad only: fold the input gradient of a 1x1 stride-2 pad-0 convolution of the same input (ResNet
  // downsample branch) into this launch: src2 = that branch's output gradient (same shape as src), wpk2 = its dgrad pack.
  // Done by the parity-class kernel only: check src2_done (0 = run it as its own launch and pass the result as addend).
  const void* src2;
  const void* wpk2;
  // out
  int src2_done;
  int stats_rows;         // rows of `stats` the forward launch wrote
  int red_done;           // 1 = red_rows holds red_rows_n rows; 0 = the caller runs the separate reduction pass
  int red_rows_n;
};
int ecg_conv_igemm(int dtype, int mode, const ConvGeom& g, const void* src, const void* wpk, void* dst,
                   const float* bias, const void* addend, float* stats, int act, hipStream_t stream,
                   ConvEpi* epi = nullptr);
// conv_halo.hip: stride-1 "same" 3x3 / 1x3 convolutions with a halo-resident activation tile (bf16, whole 256-pixel tiles)
bool ecg_conv_halo_ok(int dtype, int mode, const ConvGeom& g);
int ecg_conv_halo_rows(int mode, const ConvGeom& g);
int ecg_conv_halo(int mode, const ConvGeom& g, const void* src, const void* wpk, void* dst, const float* bias,
                  const void* addend, float* stats, int act, ConvEpi* epi, hipStream_t stream);
// conv_wgrad.hip
size_t ecg_conv_wgrad_workspace(int dtype, const ConvGeom& g);
void ecg_conv_wgrad_narrow(bool narrow, int slots = 256);   // slots: four-wave slots of a narrow launch (conv_wgrad.hip, pick_nsplit)
int ecg_conv_wgrad(int dtype, const ConvGeom& g, const void* x, const void* dy, float* grad_oihw, int accumulate,
                   void* workspace, size_t workspace_bytes, hipStream_t stream);
// conv_stem.hip
size_t ecg_stem_packed_elems(int Cin, int R);
int ecg_stem_stats_rows(int N, int Cin, int H, int W, int R);
int ecg_stem_pack(int dtype, const float* w, void* out, int Cin, int R, hipStream_t stream);
int ecg_stem_fwd(int dtype, const float* x, const void* wpk, const float* bias, void* y, float* stats, int N, int Cin,
                 int H, int W, int R, hipStream_t stream);
size_t ecg_stem_wgrad_workspace(int N, int Cin, int H, int W, int R);
int ecg_stem_wgrad(int dtype, const float* x, const void* dy, float* grad, int accumulate, void* workspace,
                   size_t workspace_bytes, int N, int Cin, int H, int W, int R, hipStream_t stream);
int ecg_stem_stats_only_rows(int N, int Cin, int H, int W, int R);
int ecg_stem_wg_stats_rows(int N, int Cin, int H, int W, int R);
int ecg_stem_fwd_wgrows(int dtype, const float* x, const void* wpk, const float* bias, void* y, float* stats, int N, int Cin,
                        int H, int W, int R, hipStream_t stream);   // bf16: statistics rows per workgroup (4 x grid) instead of per tile
int ecg_stem_stats_only(int dtype, const float* x, const void* wpk, const float* bias, float* stats, int N, int Cin, int H,
                        int W, int R, hipStream_t stream);
int ecg_stem_wgrad_reduce(const float* slab, float* grad, int rows, int NG, int accumulate, hipStream_t stream);
// conv_stem_fused.hip: the 2-D stem by recompute (bf16, R = 7, Cin <= 3): conv -> bn -> relu -> max-pool, no full-resolution tensor
bool ecg_stem_fused_ok(int dtype, int Cin, int R);
int ecg_stem_pool_fwd(const float* x, const void* wpk, const float* coef, void* pooled, unsigned char* idx, int N, int Cin,
                      int H, int W, hipStream_t stream);
size_t ecg_stem_pool_bwd_workspace(int N, int Cin, int H, int W);
int ecg_stem_pool_bwd(const float* x, const void* wpk, const float* coef, const float* gamma, const void* dp, const void* pooled,
                      const unsigned char* idx, float* dgamma, float* dbeta, float* dw, void* ws, size_t ws_bytes, int N,
                      int Cin, int H, int W, hipStream_t stream);
// elementwise.hip
int ecg_pool_bn_bwd_reduce(int dtype, const void* dp, const void* pooled, const float* coef, const float* gamma, float* dgamma,
                           float* dbeta, int N, int H, int W, int C, float* scratch, const float** bcoef_out,
                           hipStream_t stream);
int ecg_bn_rows(int dtype, long M, int C);
int ecg_bn_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* beta,
                    float* rm, float* rv, long long* nbt, float momentum, float eps, float* coef, hipStream_t stream);
int ecg_bn_eval_coef(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                     float* coef, hipStream_t stream);
int ecg_col_stats(int dtype, const void* x, long M, int C, float* partial, int* rows_out, hipStream_t stream);
// finalize folded into the consumer (elementwise.hip): the producer's partial rows + what bn_finalize needs
struct EcgBnFold {
  const float* partial; int rows; double count;
  const float* gamma; const float* beta;
  float* rm; float* rv; long long* nbt;   // running statistics (nullable)
  float momentum, eps;
};
bool ecg_bn_fold_ok(int C, int rows);
int ecg_bn_act_fold(int dtype, const void* y, float* coef, const EcgBnFold& f, const void* res, const float* rcoef,
                    const float* gate, int rows_per_sample, int relu, void* out, long M, int C, hipStream_t stream,
                    unsigned char* relu_bits = nullptr);
// relu_bits (nullable; bf16 + relu only): [M][C / 8] bytes, bit j of byte [row][k] = (out[row][8 k + j] > 0) -- the ReLU mask
// for ecg_bn_bwd's mask_bits, 1/16 of the bytes of the activated tensor it would otherwise re-read
int ecg_bn_act(int dtype, const void* y, const float* coef, const void* res, const float* rcoef, const float* gate,
               int rows_per_sample, int relu, void* out, long M, int C, hipStream_t stream, unsigned char* relu_bits = nullptr);
size_t ecg_bn_bwd_scratch(int dtype, long M, int C);
int ecg_bn_bwd(int dtype, const void* dout, const void* maskref, const float* gate, const float* addc,
               int rows_per_sample, const void* y, const float* coef, const float* gamma, float* dgamma, float* dbeta,
               void* dy, void* dz_out, float* dbias, long M, int C, float* scratch, hipStream_t stream,
               const unsigned char* mask_bits = nullptr);
int ecg_bn_bwd_tail(int dtype, const void* dout, const void* maskref, const void* y, const float* coef,
                    const float* gamma, float* dgamma, float* dbeta, void* dy, const float* partial, int rows, long M,
                    int C, float* scratch, hipStream_t stream, const float* gate = nullptr, const float* addc = nullptr,
                    int rows_per_sample = 1, float* dbias = nullptr);
int ecg_se_gate_bn(int dtype, const void* dout, const void* maskref, const void* y, const float* coef, void* dz,
                   float* dg, float* a1, float* a2, float* a3, int N, int R, int C, hipStream_t stream);
int ecg_se_bn_nrows();   // rows ecg_se_bn_rows writes ([rows][2][C])
int ecg_se_bn_rows(const float* a1, const float* a2, const float* a3, const float* gate, const float* addc, int N, int R,
                   int C, float* rows, hipStream_t stream);
int ecg_rows_sum(const float* partial, int rows, int C, float* out, int accumulate, hipStream_t stream);
int ecg_bnrelu_maxpool(int dtype, const void* y, const float* coef, void* out, unsigned char* idx, int N, int H, int W,
                       int C, hipStream_t stream);
int ecg_maxpool_relu_bwd(int dtype, const void* dp, const void* pooled, const unsigned char* idx, void* dz, int N,
                         int H, int W, int C, hipStream_t stream);
int ecg_pool_bn_bwd(int dtype, const void* dp, const void* pooled, const unsigned char* idx, const void* y,
                    const float* coef, const float* gamma, float* dgamma, float* dbeta, void* dy, float* dbias, int N,
                    int H, int W, int C, float* scratch, hipStream_t stream);
bool ecg_stem_fuse_on();
int ecg_avgpool(int dtype, const void* x, float* out, int N, int R, int C, const float* coef, hipStream_t stream);
int ecg_bcast_rows(int dtype, const float* v, void* out, int N, int R, int C, float scale, hipStream_t stream);
int ecg_se_gate_grad(int dtype, const void* dout, const void* maskref, const void* y, const float* coef, float* dg,
                     int N, int R, int C, hipStream_t stream);
constexpr int ECG_PACK_MAX = 24;
struct EcgPackItem {
  const float* w;  // OIHW / OIL fp32 master weight
  void* fwd;       // [Cout][RS][Cin] compute dtype (nullable)
  void* dgrad;     // [Cin][RS][Cout] compute dtype (nullable)
  int Cout, Cin, RS;
};
int ecg_pack_weight_batch(int dtype, const EcgPackItem* items, int n, hipStream_t stream);
// plan_resnet1d.hip: switches the ResNet1D_SE backward's weight-gradient side stream (ecgmm_side_wgrad toggles both plans)
int ecg_resnet1d_side_enable(int on);
int ecg_pack_weight(int dtype, const float* w_oihw, void* fwd, void* dgrad, int Cout, int Cin, int RS,
                    hipStream_t stream);
int ecg_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, long HW, hipStream_t stream);
int ecg_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, long HW, hipStream_t stream);
int ecg_cast(int dtype, const float* src, void* dst, long n, hipStream_t stream);
int ecg_uncast(int dtype, const void* src, float* dst, long n, hipStream_t stream);
// head.hip
int ecg_linear_fwd_valu(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, int act,
                        hipStream_t s);
int ecg_linear_dgrad_valu(const float* dy, const float* w, float* dx, int B, int In, int Out, int accumulate,
                          hipStream_t s);
int ecg_linear_wgrad_valu(const float* dy, const float* x, float* dw, float* db, int B, int In, int Out,
                          int accumulate, hipStream_t s);
int ecg_act_bwd(const float* dy, const float* y, float* dz, long n, int act, hipStream_t s);
int ecg_axpby(float a, const float* x, float b, float* y, long n, hipStream_t s);
int ecg_layernorm_fwd(const float* const* seg, const int* dims, int nseg, const float* fusion_w, const float* gamma,
                      const float* beta, float* out, float* stat, float* soft_w, int B, float eps, hipStream_t s);
size_t ecg_layernorm_bwd_scratch(int B, int D);
int ecg_layernorm_bwd(const float* const* seg, const int* dims, int nseg, const float* fusion_w, const float* gamma,
                      const float* stat, const float* dout, float* const* dseg, int dseg_accumulate, float* dgamma,
                      float* dbeta, float* dfusion_w, int B, float* scratch, hipStream_t s);
// head_fused.hip: the head's row-local work as one kernel per direction + one-wave-per-tile dense kernels
bool ecg_head_fused_ok(const int* dim, int B, int hidden, int num_classes);
int ecg_head_partial_floats(const int* dim, int num_classes);
int ecg_head_bwd_blocks(int B);
int ecg_head_rows_fwd(const float* const* raw, const float* const* ln_g, const float* const* ln_b,
                      const float* const* cls_w, const float* const* cls_b, const float* aw, const float* fg,
                      const float* fb, float* const* feat, float* const* stat, float* const* logits, float* rowvar,
                      float* fused, float* statf, float* soft_w, const int* dim, int B, int NC, float eps,
                      hipStream_t s);
int ecg_head_rows_bwd(const float* const* raw, const float* const* ln_g, const float* const* cls_w, const float* aw,
                      const float* fg, const float* const* feat, const float* const* stat, const float* statf,
                      const float* dfused, const float* const* dlog, const float* dvar, const float* gs,
                      float* const* draw, const int* have, float* partial, const int* dim, int B, int NC,
                      hipStream_t s);
int ecg_head_finalize(const float* partial, int rows, const int* dim, int NC, float* const* g_ln, float* const* g_cls,
                      float* g_aw, float* g_fg, float* g_fb, const float* aw, const int* have, const int* have_cls,
                      int have_fusion, const float* dz, int B, int H, float* fc0_db, hipStream_t s);
bool ecg_se_mlp_fused_ok(int C, int CR);
int ecg_se_mlp_fwd(const float* m, const float* w1, const float* b1, const float* w2, const float* b2, float* h, float* g,
                   int N, int C, int CR, hipStream_t s);
int ecg_se_mlp_bwd(const float* dg, const float* g, const float* h, const float* m, const float* w1, const float* w2,
                   float* ds, float* dh, float* dm, float* dw1, float* db1, float* dw2, float* db2, int N, int C, int CR,
                   float scale, hipStream_t s);
bool ecg_dense16_ok(const void* a, const void* b, const void* c, int B, int In, int Out);
int ecg_dense16_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, int act,
                    hipStream_t s);
int ecg_dense16_dgrad(const float* dy, const float* w, float* dx, int B, int In, int Out, hipStream_t s);
int ecg_dense16_wgrad(const float* dy, const float* x, float* dw, int B, int In, int Out, hipStream_t s);
int ecg_fc_dgrad_drop_relu(const float* dlog, const float* w, const unsigned char* mask, const float* hact, float* dz,
                           int B, int H, int NC, float dropout_p, hipStream_t s);
int ecg_varloss_finish(float* scratch, int B, float* loss, hipStream_t s);
int ecg_varloss_fwd(const float* f0, const float* f1, const float* f2, int B, int D0, int D1, int D2, float* loss,
                    float* scratch, hipStream_t s);
int ecg_varloss_bwd(const float* f, int B, int D, const float* gout, const float* scratch, int m, float* df,
                    int accumulate, hipStream_t s);
int ecg_ce_fwd(const float* logits, const long long* labels, int B, int C, int focal, float alpha, float gamma,
               float* loss, float* dcoef, const float* extra, float extra_w, hipStream_t s);
int ecg_ce_bwd(const float* logits, const long long* labels, int B, int C, const float* dcoef, const float* gout,
               float* dlogits, float* dextra, float extra_w, hipStream_t s);
int ecg_dropout_fwd(const float* x, float* y, unsigned char* mask, long n, float p, unsigned long long seed,
                    unsigned long long offset, hipStream_t s);
int ecg_dropout_bwd(const float* dy, const unsigned char* mask, float* dx, long n, float p, hipStream_t s);
int ecg_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd,
             long step, float gscale, hipStream_t s);

// linear.hip: fp32 Linear that picks the MFMA kernels when the shape allows, the VALU kernels otherwise
size_t ecg_linear_bwd_scratch(int B, int In, int Out);
int ecg_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, int act,
                   float* stats, hipStream_t s);
// dz is the gradient w.r.t. the pre-activation output; dx / dw / db may be null
int ecg_linear_bwd(const float* dz, const float* x, const float* w, float* dx, float* dw, float* db, int B, int In,
                   int Out, void* scratch, size_t scratch_bytes, hipStream_t s);

// prof.hip
enum { ECG_PROF_IGEMM_FWD = 0, ECG_PROF_IGEMM_DGRAD = 1, ECG_PROF_WGRAD = 2, ECG_PROF_STEM_FWD = 3, ECG_PROF_STEM_WGRAD = 4,
       ECG_PROF_IGEMM_F32_FWD = 5, ECG_PROF_IGEMM_F32_DGRAD = 6 };  // exact-fp32 instantiation (other MFMA peak): own kinds
void ecg_prof_begin(int kind, double flops, double bytes, hipStream_t s);
void ecg_prof_end(hipStream_t s);
void ecg_tl_mark(int id, hipStream_t s);   // diagnostic step timeline (prof.hip); ids: 100 + k forward, 200 + k backward

// bump allocator over a caller-owned workspace (base == nullptr: measure only)
struct Arena {
  unsigned char* base;
  size_t off;
  explicit Arena(void* b) : base((unsigned char*)b), off(0) {}
  template <typename T> T* take(size_t count) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
  void* take_bytes(size_t bytes) { return take<unsigned char>(bytes); }
};
