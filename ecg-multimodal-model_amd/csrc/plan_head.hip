// The multimodal HEAD as a native launch plan (forward + backward), fp32: everything ECGMultimodalModel.forward does
// after the three encoders (multimodal_paper_modal_balance.py:326-354; multimodal.py:440-469):
//
//   feat_m   = LayerNorm_m(raw_m)                               m = image, signal, clinical   (PMB:326,330,333)
//   logits_m = Linear(d_m -> classes)(feat_m)                   the three branch heads        (PMB:340-342)
//   fused, w = AttentionFusion(feat_img, feat_sig, feat_clin)   softmax(3) scale, cat, LN     (PMB:31-46,345)
//   fusion   = Linear(H -> classes)(Dropout(ReLU(Linear(D -> H)(fused))))                   (PMB:283-289,346)
//   var_loss = sum of pairwise |mean_b var(feat_m)| differences                              (PMB:349-352)
//
// Through torch.autograd these are ~40 small launches forward and ~45 backward, each behind ~20 us of Python / autograd
// work, at the one point of the step where no other stream has anything queued (the encoders' forward is done, their
// backward not yet enqueued): ~1 ms of a 9 ms step in the kernel trace.  One C call per direction enqueues the same
// kernels back to back.  No new arithmetic: every launch below is an op the per-op parity tests already cover.
//
// Parameter table order (19): image_norm.{weight,bias}, signal_norm.{w,b}, clinical_norm.{w,b},
//   image_classifier.{w,b}, signal_classifier.{w,b}, clinical_classifier.{w,b}, attention_fusion.weights,
//   attention_fusion.norm.{w,b}, fusion_classifier.0.{w,b}, fusion_classifier.3.{w,b}
#include "ops.h"

namespace {

enum { P_LN = 0, P_CLS = 6, P_AW = 12, P_ALN = 13, P_FC0 = 15, P_FC3 = 17 };

struct HeadWs {
  float* feat[3];
  float* stat[3];
  float* fused;
  float* statf;
  float* h;
  float* hd;
  unsigned char* mask;
  float* vscr;
  size_t bytes;
};

int check(const ecgmm_head_desc* d) {
  if (!d) ECG_FAIL(ECGMM_ERR_SHAPE, "head: null desc");
  if (d->B < 1 || d->hidden < 1 || d->num_classes < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "head: bad sizes");
  for (int m = 0; m < 3; ++m)
    if (d->dim[m] < 2) ECG_FAIL(ECGMM_ERR_SHAPE, "head: feature width %d < 2", d->dim[m]);
  if (d->dropout_p < 0.f || d->dropout_p >= 1.f) ECG_FAIL(ECGMM_ERR_SHAPE, "head: dropout p %f", d->dropout_p);
  return 0;
}

void layout(const ecgmm_head_desc& d, void* base, HeadWs& w) {
  Arena a(base);
  const size_t B = d.B, D = (size_t)d.dim[0] + d.dim[1] + d.dim[2];
  for (int m = 0; m < 3; ++m) {
    w.feat[m] = a.take<float>(B * d.dim[m]);
    w.stat[m] = a.take<float>(B * 2);
  }
  w.fused = a.take<float>(B * D);
  w.statf = a.take<float>(B * 2);
  w.h = a.take<float>(B * d.hidden);
  w.hd = a.take<float>(B * d.hidden);
  w.mask = a.take<unsigned char>(B * d.hidden);
  w.vscr = a.take<float>(3 * B + 4);
  w.bytes = align_up(a.off, 256);
}

struct HeadBwdWs {
  float* dfeat[3];
  float* dfused;
  float* dhd;
  float* dz;
  float* ln_scr;
  void* lin_scr;
  size_t lin_bytes;
  float* partial;   // head_rows_bwd_kernel's per-block partial rows (fused path)
  size_t bytes;
};

void layout_bwd(const ecgmm_head_desc& d, void* base, HeadBwdWs& w) {
  Arena a(base);
  const size_t B = d.B;
  const int D = d.dim[0] + d.dim[1] + d.dim[2];
  for (int m = 0; m < 3; ++m) w.dfeat[m] = a.take<float>(B * d.dim[m]);
  w.dfused = a.take<float>(B * D);
  w.dhd = a.take<float>(B * d.hidden);
  w.dz = a.take<float>(B * d.hidden);
  size_t ln = ecg_layernorm_bwd_scratch(d.B, D);
  for (int m = 0; m < 3; ++m) {
    size_t s = ecg_layernorm_bwd_scratch(d.B, d.dim[m]);
    if (s > ln) ln = s;
  }
  w.ln_scr = (float*)a.take_bytes(ln);
  size_t lin = ecg_linear_bwd_scratch(d.B, D, d.hidden);
  size_t l2 = ecg_linear_bwd_scratch(d.B, d.hidden, d.num_classes);
  if (l2 > lin) lin = l2;
  for (int m = 0; m < 3; ++m) {
    size_t s = ecg_linear_bwd_scratch(d.B, d.dim[m], d.num_classes);
    if (s > lin) lin = s;
  }
  w.lin_scr = a.take_bytes(lin);
  w.lin_bytes = lin;
  w.partial = a.take<float>((size_t)ecg_head_bwd_blocks(d.B) * ecg_head_partial_floats(d.dim, d.num_classes > 4 ? 4 : d.num_classes));
  w.bytes = align_up(a.off, 256);
}

inline const float* P(const void* const* params, int i) { return (const float*)params[i]; }
inline float* G(void* const* grads, int i) { return grads ? (float*)grads[i] : nullptr; }

}  // namespace

extern "C" size_t ecgmm_head_fwd_workspace(const ecgmm_head_desc* d) {
  if (check(d)) return 0;
  HeadWs w;
  layout(*d, nullptr, w);
  return w.bytes;
}

extern "C" size_t ecgmm_head_bwd_workspace(const ecgmm_head_desc* d) {
  if (check(d)) return 0;
  HeadBwdWs w;
  layout_bwd(*d, nullptr, w);
  return w.bytes;
}

extern "C" int ecgmm_head_forward(const ecgmm_head_desc* d, const float* const* raw, const void* const* params,
                                  float* const* logits, float* var_loss, float* soft_w, void* ws, size_t ws_bytes,
                                  void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  ECG_TRY(check(d));
  HeadWs w;
  layout(*d, ws, w);
  if (!ws || ws_bytes < w.bytes) ECG_FAIL(ECGMM_ERR_WORKSPACE, "head fwd: workspace %zu < %zu", ws_bytes, w.bytes);
  const int B = d->B, D = d->dim[0] + d->dim[1] + d->dim[2], H = d->hidden, NC = d->num_classes;
  if (ecg_head_fused_ok(d->dim, B, H, NC)) {
    // row-local part in one launch (head_fused.hip), Linear(D -> H) one wave per 16x16 tile
    const float *ln_g[3], *ln_b[3], *cls_w[3], *cls_b[3];
    float* lg[3] = {logits[0], logits[1], logits[2]};
    for (int m = 0; m < 3; ++m) {
      ln_g[m] = P(params, P_LN + 2 * m); ln_b[m] = P(params, P_LN + 2 * m + 1);
      cls_w[m] = P(params, P_CLS + 2 * m); cls_b[m] = P(params, P_CLS + 2 * m + 1);
    }
    ECG_TRY(ecg_head_rows_fwd(raw, ln_g, ln_b, cls_w, cls_b, P(params, P_AW), P(params, P_ALN), P(params, P_ALN + 1),
                              w.feat, w.stat, lg, w.vscr, w.fused, w.statf, soft_w, d->dim, B, NC, d->ln_eps, s));
    if (ecg_dense16_ok(w.fused, P(params, P_FC0), w.h, B, D, H))
      ECG_TRY(ecg_dense16_fwd(w.fused, P(params, P_FC0), P(params, P_FC0 + 1), w.h, B, D, H, ECGMM_ACT_RELU, s));
    else
      ECG_TRY(ecg_linear_fwd(w.fused, P(params, P_FC0), P(params, P_FC0 + 1), w.h, B, D, H, ECGMM_ACT_RELU, nullptr, s));
    const bool drop = d->training && d->dropout_p > 0.f;
    if (drop) ECG_TRY(ecg_dropout_fwd(w.h, w.hd, w.mask, (long)B * H, d->dropout_p, d->seed, d->offset, s));
    ECG_TRY(ecg_linear_fwd(drop ? w.hd : w.h, P(params, P_FC3), P(params, P_FC3 + 1), logits[3], B, H, NC, ECGMM_ACT_NONE,
                           nullptr, s));
    return ecg_varloss_finish(w.vscr, B, var_loss, s);
  }
  for (int m = 0; m < 3; ++m) {
    const float* seg[3] = {raw[m], nullptr, nullptr};
    const int dims[3] = {d->dim[m], 0, 0};
    ECG_TRY(ecg_layernorm_fwd(seg, dims, 1, nullptr, P(params, P_LN + 2 * m), P(params, P_LN + 2 * m + 1), w.feat[m],
                              w.stat[m], nullptr, B, d->ln_eps, s));
    ECG_TRY(ecg_linear_fwd(w.feat[m], P(params, P_CLS + 2 * m), P(params, P_CLS + 2 * m + 1), logits[m], B, d->dim[m],
                           NC, ECGMM_ACT_NONE, nullptr, s));
  }
  {
    const float* seg[3] = {w.feat[0], w.feat[1], w.feat[2]};
    ECG_TRY(ecg_layernorm_fwd(seg, d->dim, 3, P(params, P_AW), P(params, P_ALN), P(params, P_ALN + 1), w.fused, w.statf,
                              soft_w, B, d->ln_eps, s));
  }
  ECG_TRY(ecg_linear_fwd(w.fused, P(params, P_FC0), P(params, P_FC0 + 1), w.h, B, D, H, ECGMM_ACT_RELU, nullptr, s));
  const bool drop = d->training && d->dropout_p > 0.f;
  if (drop)
    ECG_TRY(ecg_dropout_fwd(w.h, w.hd, w.mask, (long)B * H, d->dropout_p, d->seed, d->offset, s));
  ECG_TRY(ecg_linear_fwd(drop ? w.hd : w.h, P(params, P_FC3), P(params, P_FC3 + 1), logits[3], B, H, NC, ECGMM_ACT_NONE,
                         nullptr, s));
  ECG_TRY(ecg_varloss_fwd(w.feat[0], w.feat[1], w.feat[2], B, d->dim[0], d->dim[1], d->dim[2], var_loss, w.vscr, s));
  return 0;
}

// dlogits[k] / dvar: upstream gradients, NULL = that output is not part of the loss (its branch is skipped and its
// parameters get no gradient: grads[] entries of skipped branches are left untouched).  draw[m] NULL = the encoder is
// frozen (train.py:35-40).  Gradients are WRITTEN (not accumulated), like every sink of this library.
extern "C" int ecgmm_head_backward(const ecgmm_head_desc* d, const float* const* raw, const void* const* params,
                                   void* const* grads, const float* const* dlogits, const float* dvar,
                                   float* const* draw, void* ws_fwd, void* ws_bwd, size_t ws_bwd_bytes,
                                   void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  ECG_TRY(check(d));
  HeadWs w;
  layout(*d, ws_fwd, w);
  HeadBwdWs q;
  layout_bwd(*d, ws_bwd, q);
  if (!ws_fwd || !ws_bwd || ws_bwd_bytes < q.bytes)
    ECG_FAIL(ECGMM_ERR_WORKSPACE, "head bwd: workspace %zu < %zu", ws_bwd_bytes, q.bytes);
  const int B = d->B, D = d->dim[0] + d->dim[1] + d->dim[2], H = d->hidden, NC = d->num_classes;
  bool have[3] = {false, false, false};  // dfeat[m] holds something yet?

  if (ecg_head_fused_ok(d->dim, B, H, NC)) {
    const bool fusion = dlogits[3] != nullptr;
    if (fusion) {
      const bool drop = d->training && d->dropout_p > 0.f;
      // fusion_classifier.3: weight / bias gradients; its input gradient goes through Dropout and ReLU in one pass
      ECG_TRY(ecg_linear_bwd(dlogits[3], drop ? w.hd : w.h, P(params, P_FC3), nullptr, G(grads, P_FC3), G(grads, P_FC3 + 1),
                             B, H, NC, q.lin_scr, q.lin_bytes, s));
      ECG_TRY(ecg_fc_dgrad_drop_relu(dlogits[3], P(params, P_FC3), drop ? w.mask : nullptr, w.h, q.dz, B, H, NC,
                                     d->dropout_p, s));
      if (ecg_dense16_ok(q.dz, P(params, P_FC0), q.dfused, B, D, H) && ecg_dense16_ok(w.fused, G(grads, P_FC0), q.dz, B, D, H)) {
        ECG_TRY(ecg_dense16_dgrad(q.dz, P(params, P_FC0), q.dfused, B, D, H, s));
        if (G(grads, P_FC0)) ECG_TRY(ecg_dense16_wgrad(q.dz, w.fused, G(grads, P_FC0), B, D, H, s));
      } else {
        ECG_TRY(ecg_linear_bwd(q.dz, w.fused, P(params, P_FC0), q.dfused, G(grads, P_FC0), nullptr, B, D, H, q.lin_scr,
                               q.lin_bytes, s));
      }
    }
    int hv[3], hc[3];
    const float *ln_g[3], *cls_w[3], *dl[3];
    float *g_ln[6], *g_cls[6];
    for (int m = 0; m < 3; ++m) {
      hc[m] = dlogits[m] != nullptr;
      hv[m] = fusion || dvar != nullptr || hc[m];
      ln_g[m] = P(params, P_LN + 2 * m); cls_w[m] = P(params, P_CLS + 2 * m); dl[m] = dlogits[m];
      g_ln[2 * m] = G(grads, P_LN + 2 * m); g_ln[2 * m + 1] = G(grads, P_LN + 2 * m + 1);
      g_cls[2 * m] = G(grads, P_CLS + 2 * m); g_cls[2 * m + 1] = G(grads, P_CLS + 2 * m + 1);
    }
    ECG_TRY(ecg_head_rows_bwd(raw, ln_g, cls_w, P(params, P_AW), P(params, P_ALN), w.feat, w.stat, w.statf,
                              fusion ? q.dfused : nullptr, dl, dvar, w.vscr + 3 * (size_t)B, draw, hv, q.partial, d->dim, B,
                              NC, s));
    return ecg_head_finalize(q.partial, ecg_head_bwd_blocks(B), d->dim, NC, g_ln, g_cls, G(grads, P_AW), G(grads, P_ALN),
                             G(grads, P_ALN + 1), P(params, P_AW), hv, hc, fusion ? 1 : 0, fusion ? q.dz : nullptr, B, H,
                             fusion ? G(grads, P_FC0 + 1) : nullptr, s);
  }

  if (dlogits[3]) {
    const bool drop = d->training && d->dropout_p > 0.f;
    // fusion_classifier.3, Dropout, ReLU, fusion_classifier.0
    ECG_TRY(ecg_linear_bwd(dlogits[3], drop ? w.hd : w.h, P(params, P_FC3), q.dhd, G(grads, P_FC3), G(grads, P_FC3 + 1),
                           B, H, NC, q.lin_scr, q.lin_bytes, s));
    const float* dh = q.dhd;
    if (drop) {
      ECG_TRY(ecg_dropout_bwd(q.dhd, w.mask, q.dz, (long)B * H, d->dropout_p, s));
      dh = q.dz;
    }
    ECG_TRY(ecg_act_bwd(dh, w.h, q.dz, (long)B * H, ECGMM_ACT_RELU, s));   // (elementwise: in place over dz is fine)
    ECG_TRY(ecg_linear_bwd(q.dz, w.fused, P(params, P_FC0), q.dfused, G(grads, P_FC0), G(grads, P_FC0 + 1), B, D, H,
                           q.lin_scr, q.lin_bytes, s));
    // AttentionFusion: softmax(3) scale + concat + LayerNorm
    const float* seg[3] = {w.feat[0], w.feat[1], w.feat[2]};
    float* dseg[3] = {q.dfeat[0], q.dfeat[1], q.dfeat[2]};
    ECG_TRY(ecg_layernorm_bwd(seg, d->dim, 3, P(params, P_AW), P(params, P_ALN), w.statf, q.dfused, dseg, 0,
                              G(grads, P_ALN), G(grads, P_ALN + 1), G(grads, P_AW), B, q.ln_scr, s));
    have[0] = have[1] = have[2] = true;
  }
  if (dvar) {
    for (int m = 0; m < 3; ++m) {
      ECG_TRY(ecg_varloss_bwd(w.feat[m], B, d->dim[m], dvar, w.vscr, m, q.dfeat[m], have[m] ? 1 : 0, s));
      have[m] = true;
    }
  }
  for (int m = 0; m < 3; ++m) {
    if (dlogits[m]) {
      // branch head: dfeat_m (+)= dlogits_m W_m
      if (have[m]) {
        ECG_TRY(ecg_linear_dgrad_valu(dlogits[m], P(params, P_CLS + 2 * m), q.dfeat[m], B, d->dim[m], NC, 1, s));
        ECG_TRY(ecg_linear_bwd(dlogits[m], w.feat[m], P(params, P_CLS + 2 * m), nullptr, G(grads, P_CLS + 2 * m),
                               G(grads, P_CLS + 2 * m + 1), B, d->dim[m], NC, q.lin_scr, q.lin_bytes, s));
      } else {
        ECG_TRY(ecg_linear_bwd(dlogits[m], w.feat[m], P(params, P_CLS + 2 * m), q.dfeat[m], G(grads, P_CLS + 2 * m),
                               G(grads, P_CLS + 2 * m + 1), B, d->dim[m], NC, q.lin_scr, q.lin_bytes, s));
        have[m] = true;
      }
    }
    if (!have[m]) {   // nothing upstream reaches this branch
      if (draw[m]) (void)hipMemsetAsync(draw[m], 0, (size_t)B * d->dim[m] * sizeof(float), s);
      continue;
    }
    const float* seg[3] = {raw[m], nullptr, nullptr};
    const int dims[3] = {d->dim[m], 0, 0};
    float* dseg[3] = {draw[m], nullptr, nullptr};
    ECG_TRY(ecg_layernorm_bwd(seg, dims, 1, nullptr, P(params, P_LN + 2 * m), w.stat[m], q.dfeat[m], dseg, 0,
                              G(grads, P_LN + 2 * m), G(grads, P_LN + 2 * m + 1), nullptr, B, q.ln_scr, s));
  }
  return 0;
}
