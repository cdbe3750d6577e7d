// fp32 Linear layers of the dense tails (resnet fc, signal head, clinical MLP, fusion classifier,
// SE gates, branch heads).  A Linear is a 1x1 convolution over a [B,1,1,In] channels-last tensor, so
// MFMA-eligible shapes run on the exact-f32 instantiations of the implicit-GEMM kernels (torch's
// [Out][In] weight already IS the packed forward layout); ragged shapes (2-class heads, SE
// bottlenecks) use the VALU kernels of head.hip.
#include "ops.h"

namespace {
inline bool mfma_reduce_ok(int k) { return k % 4 == 0 && k >= 16; }
inline ConvGeom lin_geom(int B, int In, int Out) { return make_geom(B, 1, 1, In, Out, 1, 1, 1, 0, 0); }
// Small dense layers (the tails at batch <= a few hundred: 10-70 MFLOP) run one wave per 16x16 output tile on
// dense16_kernel (head_fused.hip): the implicit-GEMM kernel gives them 2-4 workgroups walking a serial K chain
// (12-57 us a launch); independent waves take 5-14 us.  ECGMM_DENSE16=0: always the implicit-GEMM route.
inline bool dense16_route(const void* a, const void* b, const void* c, int B, int In, int Out) {
  static const bool on = [] { const char* e = getenv("ECGMM_DENSE16"); return !(e && e[0] == '0'); }();
  return on && ecg_dense16_ok(a, b, c, B, In, Out) && (double)B * In * Out <= 134217728.0;
}
}  // namespace

size_t ecg_linear_bwd_scratch(int B, int In, int Out) {
  size_t s = align_up((size_t)In * Out * sizeof(float), 256);  // transposed weight for dgrad
  if (mfma_reduce_ok(In) && mfma_reduce_ok(Out)) s += ecg_conv_wgrad_workspace(ECGMM_F32, lin_geom(B, In, Out));
  return s;
}

int ecg_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, int act,
                   float* stats, hipStream_t s) {
  if (!stats && act != ECGMM_ACT_SIGMOID && dense16_route(x, w, y, B, In, Out))
    return ecg_dense16_fwd(x, w, bias, y, B, In, Out, act, s);
  if (mfma_reduce_ok(In) && Out >= 16 && act != ECGMM_ACT_SIGMOID)
    return ecg_conv_igemm(ECGMM_F32, 0, lin_geom(B, In, Out), x, w, y, bias, nullptr, stats, act, s);
  if (stats) ECG_FAIL(ECGMM_ERR_SHAPE, "linear: fused stats need an MFMA-eligible shape (In=%d Out=%d)", In, Out);
  return ecg_linear_fwd_valu(x, w, bias, y, B, In, Out, act, s);
}

int ecg_linear_bwd(const float* dz, const float* x, const float* w, float* dx, float* dw, float* db, int B, int In,
                   int Out, void* scratch, size_t scratch_bytes, hipStream_t s) {
  const size_t wt_bytes = align_up((size_t)In * Out * sizeof(float), 256);
  if (dx) {
    if (dense16_route(dz, w, dx, B, In, Out)) {
      ECG_TRY(ecg_dense16_dgrad(dz, w, dx, B, In, Out, s));
    } else if (mfma_reduce_ok(Out) && In >= 16) {
      if (!scratch || scratch_bytes < wt_bytes) ECG_FAIL(ECGMM_ERR_WORKSPACE, "linear bwd: scratch too small");
      ECG_TRY(ecg_pack_weight(ECGMM_F32, w, nullptr, scratch, Out, In, 1, s));
      ECG_TRY(ecg_conv_igemm(ECGMM_F32, 1, lin_geom(B, In, Out), dz, scratch, dx, nullptr, nullptr, nullptr, 0, s));
    } else {
      ECG_TRY(ecg_linear_dgrad_valu(dz, w, dx, B, In, Out, 0, s));
    }
  }
  if (dw) {
    if (dense16_route(dz, x, dw, B, In, Out)) {
      ECG_TRY(ecg_dense16_wgrad(dz, x, dw, B, In, Out, s));
      if (db) ECG_TRY(ecg_rows_sum(dz, B, Out, db, 0, s));
    } else if (mfma_reduce_ok(In) && mfma_reduce_ok(Out)) {
      ConvGeom g = lin_geom(B, In, Out);
      size_t need = ecg_conv_wgrad_workspace(ECGMM_F32, g);
      if (!scratch || scratch_bytes < wt_bytes + need) ECG_FAIL(ECGMM_ERR_WORKSPACE, "linear bwd: scratch too small");
      ECG_TRY(ecg_conv_wgrad(ECGMM_F32, g, x, dz, dw, 0, (unsigned char*)scratch + wt_bytes, need, s));
      if (db) ECG_TRY(ecg_rows_sum(dz, B, Out, db, 0, s));
    } else {
      ECG_TRY(ecg_linear_wgrad_valu(dz, x, dw, db, B, In, Out, 0, s));
    }
  } else if (db) {
    ECG_TRY(ecg_rows_sum(dz, B, Out, db, 0, s));
  }
  return 0;
}
