// extern "C" surface of libecgmm_hip.so (see include/ecgmm.h): thin argument adapters only.
#include <stdarg.h>

#include "ops.h"

static thread_local char g_err[512] = "";

void ecg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline ConvGeom geom_of(const ecgmm_conv_desc* c) {
  return make_geom(c->N, c->H, c->W, c->Cin, c->Cout, c->R, c->S, c->stride, c->pad_h, c->pad_w);
}
#define S_(x) ((hipStream_t)(x))

extern "C" {

int ecgmm_version(void) { return ECGMM_VERSION; }
const char* ecgmm_last_error(void) { return g_err; }

int ecgmm_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int64_t HW, void* stream) {
  return ecg_nchw_to_nhwc(dtype, src, dst, N, C, (long)HW, S_(stream));
}
int ecgmm_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int64_t HW, void* stream) {
  return ecg_nhwc_to_nchw(dtype, src, dst, N, C, (long)HW, S_(stream));
}
int ecgmm_cast(int dtype, const float* src, void* dst, int64_t n, void* stream) {
  return ecg_cast(dtype, src, dst, (long)n, S_(stream));
}
int ecgmm_uncast(int dtype, const void* src, float* dst, int64_t n, void* stream) {
  return ecg_uncast(dtype, src, dst, (long)n, S_(stream));
}
int ecgmm_pack_conv_weight(int dtype, const float* w, void* fwd, void* dgrad, int Cout, int Cin, int RS,
                           void* stream) {
  return ecg_pack_weight(dtype, w, fwd, dgrad, Cout, Cin, RS, S_(stream));
}

int ecgmm_conv_stats_rows(int64_t out_pixels) { return ecg_conv_stats_rows((long)out_pixels); }
int ecgmm_conv_fwd(int dtype, const ecgmm_conv_desc* c, const void* x, const void* w_fwd, const float* bias, void* y,
                   float* stats, int act, void* stream) {
  return ecg_conv_igemm(dtype, 0, geom_of(c), x, w_fwd, y, bias, nullptr, stats, act, S_(stream));
}
int ecgmm_conv_bwd_data(int dtype, const ecgmm_conv_desc* c, const void* dy, const void* w_dgrad, const void* addend,
                        void* dx, void* stream) {
  return ecg_conv_igemm(dtype, 1, geom_of(c), dy, w_dgrad, dx, nullptr, addend, nullptr, 0, S_(stream));
}
size_t ecgmm_conv_bwd_weight_workspace(int dtype, const ecgmm_conv_desc* c) {
  return ecg_conv_wgrad_workspace(dtype, geom_of(c));
}
int ecgmm_conv_bwd_weight(int dtype, const ecgmm_conv_desc* c, const void* x, const void* dy, float* dw, int accumulate,
                          void* ws, size_t ws_bytes, void* stream) {
  return ecg_conv_wgrad(dtype, geom_of(c), x, dy, dw, accumulate, ws, ws_bytes, S_(stream));
}

size_t ecgmm_stem_packed_elems(int Cin, int R) { return ecg_stem_packed_elems(Cin, R); }
int ecgmm_stem_stats_rows(int N, int Cin, int H, int W, int R) { return ecg_stem_stats_rows(N, Cin, H, W, R); }
int ecgmm_stem_pack(int dtype, const float* w, void* packed, int Cin, int R, void* stream) {
  return ecg_stem_pack(dtype, w, packed, Cin, R, S_(stream));
}
int ecgmm_stem_fwd(int dtype, const float* x, const void* packed, const float* bias, void* y, float* stats, int N,
                   int Cin, int H, int W, int R, void* stream) {
  return ecg_stem_fwd(dtype, x, packed, bias, y, stats, N, Cin, H, W, R, S_(stream));
}
int ecgmm_stem_wg_stats_rows(int N, int Cin, int H, int W, int R) { return ecg_stem_wg_stats_rows(N, Cin, H, W, R); }
int ecgmm_stem_fwd_wgrows(int dtype, const float* x, const void* packed, const float* bias, void* y, float* stats, int N,
                          int Cin, int H, int W, int R, void* stream) {
  return ecg_stem_fwd_wgrows(dtype, x, packed, bias, y, stats, N, Cin, H, W, R, S_(stream));
}
int ecgmm_stem_stats_only_rows(int N, int Cin, int H, int W, int R) { return ecg_stem_stats_only_rows(N, Cin, H, W, R); }
int ecgmm_stem_stats_only(int dtype, const float* x, const void* packed, const float* bias, float* stats, int N, int Cin,
                          int H, int W, int R, void* stream) {
  return ecg_stem_stats_only(dtype, x, packed, bias, stats, N, Cin, H, W, R, S_(stream));
}
int ecgmm_stem_pool_fwd(const float* x, const void* packed, const float* coef, void* pooled, uint8_t* idx, int N, int Cin,
                        int H, int W, void* stream) {
  return ecg_stem_pool_fwd(x, packed, coef, pooled, idx, N, Cin, H, W, S_(stream));
}
size_t ecgmm_stem_pool_bwd_workspace(int N, int Cin, int H, int W) { return ecg_stem_pool_bwd_workspace(N, Cin, H, W); }
int ecgmm_stem_pool_bwd(const float* x, const void* packed, const float* coef, const float* gamma, const void* dp,
                        const void* pooled, const uint8_t* idx, float* dgamma, float* dbeta, float* dw_oihw, void* ws,
                        size_t ws_bytes, int N, int Cin, int H, int W, void* stream) {
  return ecg_stem_pool_bwd(x, packed, coef, gamma, dp, pooled, idx, dgamma, dbeta, dw_oihw, ws, ws_bytes, N, Cin, H, W,
                           S_(stream));
}
size_t ecgmm_stem_bwd_weight_workspace(int N, int Cin, int H, int W, int R) {
  return ecg_stem_wgrad_workspace(N, Cin, H, W, R);
}
int ecgmm_stem_bwd_weight(int dtype, const float* x, const void* dy, float* dw, int accumulate, void* ws,
                          size_t ws_bytes, int N, int Cin, int H, int W, int R, void* stream) {
  return ecg_stem_wgrad(dtype, x, dy, dw, accumulate, ws, ws_bytes, N, Cin, H, W, R, S_(stream));
}

int ecgmm_col_stats_rows(int dtype, int64_t M, int C) { return ecg_bn_rows(dtype, (long)M, C); }
int ecgmm_col_stats(int dtype, const void* x, int64_t M, int C, float* partial, void* stream) {
  int rows = 0;
  return ecg_col_stats(dtype, x, (long)M, C, partial, &rows, S_(stream));
}
int ecgmm_bn_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* beta,
                      float* rm, float* rv, int64_t* nbt, float momentum, float eps, float* coef, void* stream) {
  return ecg_bn_finalize(partial, rows, C, count, gamma, beta, rm, rv, (long long*)nbt, momentum, eps, coef,
                         S_(stream));
}
int ecgmm_bn_eval_coef(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                       float* coef, void* stream) {
  return ecg_bn_eval_coef(C, gamma, beta, rm, rv, eps, coef, S_(stream));
}
int ecgmm_bn_act(int dtype, const void* y, const float* coef, const void* res, const float* rcoef, const float* gate,
                 int rows_per_sample, int relu, void* out, int64_t M, int C, void* stream) {
  return ecg_bn_act(dtype, y, coef, res, rcoef, gate, rows_per_sample, relu, out, (long)M, C, S_(stream));
}
int ecgmm_bn_act_from_rows(int dtype, const void* y, const float* partial, int rows, double count, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float momentum, float eps, float* coef_out, const void* res, const float* rcoef,
                           const float* gate, int rows_per_sample, int relu, void* out, int64_t M, int C, void* stream) {
  if (!partial || !coef_out) ECG_FAIL(ECGMM_ERR_SHAPE, "bn_act_from_rows: null partial rows / coef");
  if (!ecg_bn_fold_ok(C, rows)) {   // the separate finalize launch, same results
    ECG_TRY(ecg_bn_finalize(partial, rows, C, count, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked,
                            momentum, eps, coef_out, S_(stream)));
    return ecg_bn_act(dtype, y, coef_out, res, rcoef, gate, rows_per_sample, relu, out, (long)M, C, S_(stream));
  }
  EcgBnFold f = {partial, rows, count, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum, eps};
  return ecg_bn_act_fold(dtype, y, coef_out, f, res, rcoef, gate, rows_per_sample, relu, out, (long)M, C, S_(stream));
}
size_t ecgmm_bn_bwd_scratch(int dtype, int64_t M, int C) { return ecg_bn_bwd_scratch(dtype, (long)M, C); }
int ecgmm_bn_bwd(int dtype, const void* dout, const void* maskref, const float* gate, const float* addc,
                 int rows_per_sample, const void* y, const float* coef, const float* gamma, float* dgamma,
                 float* dbeta, void* dy, void* dz_out, float* dbias, int64_t M, int C, void* scratch, void* stream) {
  return ecg_bn_bwd(dtype, dout, maskref, gate, addc, rows_per_sample, y, coef, gamma, dgamma, dbeta, dy, dz_out, dbias,
                    (long)M, C, (float*)scratch, S_(stream));
}

int ecgmm_conv_bwd_data_bnred(int dtype, const ecgmm_conv_desc* c, const void* dy, const void* w_dgrad,
                              const void* addend, void* dx, const void* bn_y, const void* bn_mask, const float* bn_coef,
                              float* rows, int* nrows, void* stream) {
  if (!bn_y || !bn_coef || !rows || !nrows) ECG_FAIL(ECGMM_ERR_SHAPE, "conv_bwd_data_bnred: null BatchNorm operand");
  const ConvGeom g = geom_of(c);
  *nrows = 0;
  if (!ecg_conv_halo_ok(dtype, 1, g))
    return ecg_conv_igemm(dtype, 1, g, dy, w_dgrad, dx, nullptr, addend, nullptr, 0, S_(stream));
  ConvEpi e = {};
  e.wg_rows = 1; e.red_y = bn_y; e.red_mask = bn_mask; e.red_coef = bn_coef; e.red_rows = rows;
  int rc = ecg_conv_igemm(dtype, 1, g, dy, w_dgrad, dx, nullptr, addend, nullptr, 0, S_(stream), &e);
  if (rc == 0 && e.red_done) *nrows = e.red_rows_n;
  return rc;
}
// ecgmm_conv_fwd whose BatchNorm partial sums come as ONE row per workgroup where the halo-resident kernel serves the shape
// (*nrows rows of [2][Cout]; the per-64-pixel layout and its row count otherwise) -- the form the encoder plans use.
int ecgmm_conv_fwd_wgrows(int dtype, const ecgmm_conv_desc* c, const void* x, const void* w_fwd, const float* bias, void* y,
                          float* stats, int* nrows, int act, void* stream) {
  if (!stats || !nrows) ECG_FAIL(ECGMM_ERR_SHAPE, "conv_fwd_wgrows: null statistics buffer / row count");
  const ConvGeom g = geom_of(c);
  ConvEpi e = {};
  e.wg_rows = 1;
  int rc = ecg_conv_igemm(dtype, 0, g, x, w_fwd, y, bias, nullptr, stats, act, S_(stream), &e);
  *nrows = e.stats_rows > 0 ? e.stats_rows : ecg_conv_stats_rows((long)g.N * g.OH * g.OW);
  return rc;
}
// Partial rows a launch of ecgmm_conv_bwd_data_bnred writes for this geometry (0 = the fused form does not apply): the
// number an encoder plan uses when ANOTHER call consumes the rows (csrc/plan_resnet18.hip) -- follows ecgmm_conv_halo_cus.
int ecgmm_conv_bwd_data_bnred_rows(int dtype, const ecgmm_conv_desc* c) {
  const ConvGeom g = geom_of(c);
  return ecg_conv_halo_ok(dtype, 1, g) ? ecg_conv_halo_rows(1, g) : 0;
}
int ecgmm_conv_bwd_data_with_downsample(int dtype, const ecgmm_conv_desc* c, const void* dy, const void* w_dgrad,
                                        const void* dy_down, const void* w_down_dgrad, void* dx, void* tmp,
                                        void* stream) {
  if (!dy_down || !w_down_dgrad) ECG_FAIL(ECGMM_ERR_SHAPE, "conv_bwd_data_with_downsample: null downsample operand");
  const ConvGeom g = geom_of(c);
  if (g.stride != 2) ECG_FAIL(ECGMM_ERR_SHAPE, "conv_bwd_data_with_downsample: stride %d (the main convolution must have stride 2)", g.stride);
  ConvEpi e = {};
  e.src2 = dy_down; e.wpk2 = w_down_dgrad;
  ECG_TRY(ecg_conv_igemm(dtype, 1, g, dy, w_dgrad, dx, nullptr, nullptr, nullptr, 0, S_(stream), &e));
  if (e.src2_done) return 0;
  // geometry not served by the folded form: two launches through the caller's temporary (same shape as dx)
  if (!tmp) ECG_FAIL(ECGMM_ERR_WORKSPACE, "conv_bwd_data_with_downsample: this geometry needs the temporary");
  ConvGeom gd = g;
  gd.R = gd.S = 1; gd.pad_h = gd.pad_w = 0;
  ECG_TRY(ecg_conv_igemm(dtype, 1, gd, dy_down, w_down_dgrad, tmp, nullptr, nullptr, nullptr, 0, S_(stream)));
  return ecg_conv_igemm(dtype, 1, g, dy, w_dgrad, dx, nullptr, tmp, nullptr, 0, S_(stream));
}
int ecgmm_bn_bwd_from_rows(int dtype, const void* dout, const void* maskref, const void* y, const float* coef,
                           const float* gamma, float* dgamma, float* dbeta, void* dy, const float* rows, int nrows,
                           int64_t M, int C, void* scratch, void* stream) {
  return ecg_bn_bwd_tail(dtype, dout, maskref, y, coef, gamma, dgamma, dbeta, dy, rows, nrows, (long)M, C,
                         (float*)scratch, S_(stream));
}

int ecgmm_bnrelu_maxpool(int dtype, const void* y, const float* coef, void* out, uint8_t* idx, int N, int H, int W,
                         int C, void* stream) {
  return ecg_bnrelu_maxpool(dtype, y, coef, out, idx, N, H, W, C, S_(stream));
}
int ecgmm_maxpool_relu_bwd(int dtype, const void* dp, const void* pooled, const uint8_t* idx, void* dz, int N, int H,
                           int W, int C, void* stream) {
  return ecg_maxpool_relu_bwd(dtype, dp, pooled, idx, dz, N, H, W, C, S_(stream));
}
int ecgmm_pool_bn_bwd(int dtype, const void* dp, const void* pooled, const uint8_t* idx, const void* y, const float* coef,
                      const float* gamma, float* dgamma, float* dbeta, void* dy, float* dbias, int N, int H, int W, int C,
                      void* scratch, void* stream) {
  return ecg_pool_bn_bwd(dtype, dp, pooled, idx, y, coef, gamma, dgamma, dbeta, dy, dbias, N, H, W, C, (float*)scratch,
                         S_(stream));
}
int ecgmm_avgpool(int dtype, const void* x, float* out, int N, int R, int C, const float* coef, void* stream) {
  return ecg_avgpool(dtype, x, out, N, R, C, coef, S_(stream));
}
int ecgmm_bcast_rows(int dtype, const float* v, void* out, int N, int R, int C, float scale, void* stream) {
  return ecg_bcast_rows(dtype, v, out, N, R, C, scale, S_(stream));
}
int ecgmm_se_gate_grad(int dtype, const void* dout, const void* maskref, const void* y, const float* coef, float* dg,
                       int N, int R, int C, void* stream) {
  return ecg_se_gate_grad(dtype, dout, maskref, y, coef, dg, N, R, C, S_(stream));
}

int ecgmm_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, int act,
                     void* stream) {
  return ecg_linear_fwd(x, w, bias, y, B, In, Out, act, nullptr, S_(stream));
}
size_t ecgmm_linear_bwd_scratch(int B, int In, int Out) { return ecg_linear_bwd_scratch(B, In, Out); }
int ecgmm_linear_bwd(const float* dz, const float* x, const float* w, float* dx, float* dw, float* db, int B, int In,
                     int Out, void* scratch, size_t scratch_bytes, void* stream) {
  return ecg_linear_bwd(dz, x, w, dx, dw, db, B, In, Out, scratch, scratch_bytes, S_(stream));
}
int ecgmm_act_bwd(const float* dy, const float* y, float* dz, int64_t n, int act, void* stream) {
  return ecg_act_bwd(dy, y, dz, (long)n, act, S_(stream));
}

int ecgmm_layernorm_fwd(const float* const* seg, const int* dims, int nseg, const float* fusion_w, const float* gamma,
                        const float* beta, float* out, float* stat, float* soft_w, int B, float eps, void* stream) {
  return ecg_layernorm_fwd(seg, dims, nseg, fusion_w, gamma, beta, out, stat, soft_w, B, eps, S_(stream));
}
size_t ecgmm_layernorm_bwd_scratch(int B, int D) { return ecg_layernorm_bwd_scratch(B, D); }
int ecgmm_layernorm_bwd(const float* const* seg, const int* dims, int nseg, const float* fusion_w, const float* gamma,
                        const float* stat, const float* dout, float* const* dseg, int dseg_accumulate, float* dgamma,
                        float* dbeta, float* dfusion_w, int B, void* scratch, void* stream) {
  return ecg_layernorm_bwd(seg, dims, nseg, fusion_w, gamma, stat, dout, dseg, dseg_accumulate, dgamma, dbeta,
                           dfusion_w, B, (float*)scratch, S_(stream));
}

int ecgmm_varloss_fwd(const float* f0, const float* f1, const float* f2, int B, int D0, int D1, int D2, float* loss,
                      float* scratch, void* stream) {
  return ecg_varloss_fwd(f0, f1, f2, B, D0, D1, D2, loss, scratch, S_(stream));
}
int ecgmm_varloss_bwd(const float* f, int B, int D, const float* gout, const float* scratch, int which, float* df,
                      int accumulate, void* stream) {
  return ecg_varloss_bwd(f, B, D, gout, scratch, which, df, accumulate, S_(stream));
}

int ecgmm_ce_fwd(const float* logits, const int64_t* labels, int B, int C, int focal, float alpha, float gamma,
                 float* loss, float* dcoef, void* stream) {
  return ecg_ce_fwd(logits, (const long long*)labels, B, C, focal, alpha, gamma, loss, dcoef, nullptr, 0.f, S_(stream));
}
int ecgmm_ce_plus_fwd(const float* logits, const int64_t* labels, int B, int C, const float* extra, float extra_w,
                      float* loss, float* dcoef, void* stream) {
  return ecg_ce_fwd(logits, (const long long*)labels, B, C, 0, 1.f, 0.f, loss, dcoef, extra, extra_w, S_(stream));
}
int ecgmm_ce_plus_bwd(const float* logits, const int64_t* labels, int B, int C, const float* dcoef, const float* gout,
                      float* dlogits, float* dextra, float extra_w, void* stream) {
  return ecg_ce_bwd(logits, (const long long*)labels, B, C, dcoef, gout, dlogits, dextra, extra_w, S_(stream));
}
int ecgmm_ce_bwd(const float* logits, const int64_t* labels, int B, int C, const float* dcoef, const float* gout,
                 float* dlogits, void* stream) {
  return ecg_ce_bwd(logits, (const long long*)labels, B, C, dcoef, gout, dlogits, nullptr, 0.f, S_(stream));
}

int ecgmm_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset,
                      void* stream) {
  return ecg_dropout_fwd(x, y, mask, (long)n, p, seed, offset, S_(stream));
}
int ecgmm_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, void* stream) {
  return ecg_dropout_bwd(dy, mask, dx, (long)n, p, S_(stream));
}

int ecgmm_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int64_t step, float gscale, void* stream) {
  return ecg_adam(p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay, (long)step, gscale, S_(stream));
}
int ecgmm_axpby(float a, const float* x, float b, float* y, int64_t n, void* stream) {
  return ecg_axpby(a, x, b, y, (long)n, S_(stream));
}

}  // extern "C"
