// ResNet1D_SE signal encoder as a native launch plan (forward + backward), gfx950.
// Mirrors multimodal_paper_modal_balance.py:49-125 (== signal_model.py:12-88):
//   Conv1d(cin,64,7,2,3)+BN+ReLU+MaxPool1d(3,2,1) -> 3 x BasicBlock1D(conv3+BN+ReLU, conv3+BN, SE,
//   +identity / conv1 s2 + BN, ReLU) -> global avgpool -> Linear(256,64)+ReLU+Dropout -> Linear(64,n)
// Conv1d layers keep their bias in front of BN, as the reference does.  Activations are [N, L, C]
// channels-last in the compute dtype; the signal enters as [N, cin, L] fp32.
//
// Parameter table (52): initial.0.{weight,bias}, initial.1.{weight,bias}, then per layer
//   conv1.{w,b}, bn1.{w,b}, conv2.{w,b}, bn2.{w,b}, se.fc.0.{w,b}, se.fc.2.{w,b},
//   [downsample.0.{w,b}, downsample.1.{w,b}] (layers 2,3), then classifier.1.{w,b}, classifier.4.{w,b}.
// Buffer table (27): running_mean, running_var, num_batches_tracked per BatchNorm in the same walk.
#include "ops.h"
#include "side_stream.h"

namespace {

struct Blk1 {
  int cin, cout, stride, lin, lout, cr;
  bool down;
  int p0;  // first param index
  int b0;  // first buffer index
};

struct R1D {
  ecgmm_resnet1d_desc d;
  int L1, L2;
  Blk1 blk[3];
  int p_cls;
  size_t max_act;
};

int build(const ecgmm_resnet1d_desc* d, R1D& r) {
  if (!d) ECG_FAIL(ECGMM_ERR_SHAPE, "resnet1d: null desc");
  if (d->dtype != ECGMM_BF16 && d->dtype != ECGMM_F32) ECG_FAIL(ECGMM_ERR_DTYPE, "resnet1d: bad dtype %d", d->dtype);
  if (d->N < 1 || d->L < 64 || d->cin < 1 || d->cin > 24)
    ECG_FAIL(ECGMM_ERR_SHAPE, "resnet1d: bad input N=%d cin=%d L=%d", d->N, d->cin, d->L);
  r.d = *d;
  r.L1 = (d->L + 6 - 7) / 2 + 1;
  r.L2 = (r.L1 + 2 - 3) / 2 + 1;
  int pi = 4, bi = 3, l = r.L2, cin = 64;
  r.max_act = (size_t)d->N * r.L2 * 64;
  for (int i = 0; i < 3; ++i) {
    Blk1& k = r.blk[i];
    k.cin = cin; k.cout = 64 << i; k.stride = i == 0 ? 1 : 2;
    k.lin = l; k.lout = (l + 2 - 3) / k.stride + 1;
    k.cr = k.cout / 16;
    k.down = (k.stride != 1 || k.cin != k.cout);
    k.p0 = pi; k.b0 = bi;
    pi += k.down ? 16 : 12;
    bi += k.down ? 9 : 6;
    size_t a = (size_t)d->N * k.lout * k.cout;
    if (a > r.max_act) r.max_act = a;
    l = k.lout; cin = k.cout;
  }
  r.p_cls = pi;
  if (pi + 4 != ECGMM_RESNET1D_NPARAMS || bi != ECGMM_RESNET1D_NBUFFERS)
    ECG_FAIL(ECGMM_ERR_SHAPE, "resnet1d: internal table mismatch %d %d", pi + 4, bi);
  return 0;
}

struct Fwd1 {
  void* wstem; void* y0; float* coef0; void* p0; unsigned char* idx0;
  struct B {
    void *w1f, *w1d, *w2f, *w2d, *wdf, *wdd;
    void *y1, *a1, *y2, *yd, *out;
    float *coef1, *coef2, *coefd, *m, *h, *g;
  } b[3];
  float *pooled, *h1, *hd;
  unsigned char* dmask;
  float* stats;
  size_t bytes;
};

void layout_fwd(const R1D& r, void* base, Fwd1& w) {
  Arena a(base);
  const size_t es = dtype_size(r.d.dtype);
  const int N = r.d.N;
  w.wstem = a.take_bytes(ecg_stem_packed_elems(r.d.cin, 1) * es);
  w.y0 = a.take_bytes((size_t)N * r.L1 * 64 * es);
  w.coef0 = a.take<float>(4 * 64);
  w.p0 = a.take_bytes((size_t)N * r.L2 * 64 * es);
  w.idx0 = a.take<unsigned char>((size_t)N * r.L2 * 64);
  size_t max_rows_c = (size_t)ecg_stem_stats_rows(N, r.d.cin, 1, r.d.L, 1) * 2 * 64;
  max_rows_c += (size_t)ECG_TAIL_ROWS * 2 * 64;
  for (int i = 0; i < 3; ++i) {
    const Blk1& k = r.blk[i];
    Fwd1::B& b = w.b[i];
    size_t osz = (size_t)N * k.lout * k.cout;
    b.w1f = a.take_bytes((size_t)k.cout * k.cin * 3 * es);
    b.w1d = a.take_bytes((size_t)k.cout * k.cin * 3 * es);
    b.w2f = a.take_bytes((size_t)k.cout * k.cout * 3 * es);
    b.w2d = a.take_bytes((size_t)k.cout * k.cout * 3 * es);
    b.y1 = a.take_bytes(osz * es);
    b.a1 = a.take_bytes(osz * es);
    b.y2 = a.take_bytes(osz * es);
    b.out = a.take_bytes(osz * es);
    b.coef1 = a.take<float>(4 * k.cout);
    b.coef2 = a.take<float>(4 * k.cout);
    b.m = a.take<float>((size_t)N * k.cout);
    b.h = a.take<float>((size_t)N * k.cr);
    b.g = a.take<float>((size_t)N * k.cout);
    if (k.down) {
      b.wdf = a.take_bytes((size_t)k.cout * k.cin * es);
      b.wdd = a.take_bytes((size_t)k.cout * k.cin * es);
      b.yd = a.take_bytes(osz * es);
      b.coefd = a.take<float>(4 * k.cout);
    } else {
      b.wdf = b.wdd = b.yd = nullptr;
      b.coefd = nullptr;
    }
    size_t rows_c = (size_t)(ecg_conv_stats_rows((long)N * k.lout) + ECG_TAIL_ROWS) * 2 * k.cout;
    if (rows_c > max_rows_c) max_rows_c = rows_c;
  }
  w.pooled = a.take<float>((size_t)N * 256);
  w.h1 = a.take<float>((size_t)N * 64);
  w.hd = a.take<float>((size_t)N * 64);
  w.dmask = a.take<unsigned char>((size_t)N * 64);
  w.stats = a.take<float>(max_rows_c);
  w.bytes = align_up(a.off, 256);
}

struct Bwd1 {
  void* X[2]; void *dz, *dy, *dy1, *dyd, *da, *dtmp, *big0, *big1;
  float *dpooled, *dh1, *dfeat_h, *dg, *ds, *dh, *dm, *dbias_scratch;
  float *sa1, *sa2, *sa3, *se_rows;   // per-sample sums of the merged SE / BatchNorm backward pass, and its reduction row
  float* bn_scratch;
  void* wg_ws; size_t wg_bytes;
  void* stem_ws; size_t stem_bytes;
  void* lin_ws; size_t lin_bytes;
  size_t bytes;
};

void layout_bwd(const R1D& r, void* base, Bwd1& w) {
  Arena a(base);
  const size_t es = dtype_size(r.d.dtype);
  const int N = r.d.N;
  for (int i = 0; i < 2; ++i) w.X[i] = a.take_bytes(r.max_act * es);
  w.dz = a.take_bytes(r.max_act * es);
  w.dy = a.take_bytes(r.max_act * es);
  w.dy1 = a.take_bytes(r.max_act * es);  // own buffers for the three wgrad operands: the side stream still reads one
  w.dyd = a.take_bytes(r.max_act * es);  // while the main stream's BatchNorm backward writes the next
  w.da = a.take_bytes(r.max_act * es);
  w.dtmp = a.take_bytes(r.max_act * es);
  size_t big = (size_t)N * r.L1 * 64;
  w.big0 = a.take_bytes(big * es);
  w.big1 = a.take_bytes(big * es);
  w.dpooled = a.take<float>((size_t)N * 256);
  w.dh1 = a.take<float>((size_t)N * 64);
  w.dfeat_h = a.take<float>((size_t)N * 64);
  w.dg = a.take<float>((size_t)N * 256);
  w.ds = a.take<float>((size_t)N * 256);
  w.dh = a.take<float>((size_t)N * 16);
  w.dm = a.take<float>((size_t)N * 256);
  w.dbias_scratch = a.take<float>(256);
  w.sa1 = a.take<float>((size_t)N * 256);
  w.sa2 = a.take<float>((size_t)N * 256);
  w.sa3 = a.take<float>((size_t)N * 256);
  w.se_rows = a.take<float>((size_t)ecg_se_bn_nrows() * 2 * 256);
  size_t bn = ecg_bn_bwd_scratch(r.d.dtype, (long)N * r.L1, 64);
  size_t wg = 0;
  size_t lin = ecg_linear_bwd_scratch(N, 256, 64);
  size_t l2 = ecg_linear_bwd_scratch(N, 64, r.d.num_classes);
  if (l2 > lin) lin = l2;
  for (int i = 0; i < 3; ++i) {
    const Blk1& k = r.blk[i];
    size_t s = ecg_bn_bwd_scratch(r.d.dtype, (long)N * k.lout, k.cout);
    if (s > bn) bn = s;
    size_t g1 = ecg_conv_wgrad_workspace(r.d.dtype, make_geom(N, 1, k.lin, k.cin, k.cout, 1, 3, k.stride, 0, 1));
    size_t g2 = ecg_conv_wgrad_workspace(r.d.dtype, make_geom(N, 1, k.lout, k.cout, k.cout, 1, 3, 1, 0, 1));
    if (g1 > wg) wg = g1;
    if (g2 > wg) wg = g2;
    if (k.down) {
      size_t g3 = ecg_conv_wgrad_workspace(r.d.dtype, make_geom(N, 1, k.lin, k.cin, k.cout, 1, 1, k.stride, 0, 0));
      if (g3 > wg) wg = g3;
    }
    size_t la = ecg_linear_bwd_scratch(N, k.cr, k.cout), lb = ecg_linear_bwd_scratch(N, k.cout, k.cr);
    if (la > lin) lin = la;
    if (lb > lin) lin = lb;
  }
  w.bn_scratch = (float*)a.take_bytes(bn);
  w.wg_ws = a.take_bytes(wg);
  w.wg_bytes = wg;
  w.stem_bytes = ecg_stem_wgrad_workspace(N, r.d.cin, 1, r.d.L, 1);
  w.stem_ws = a.take_bytes(w.stem_bytes);
  w.lin_ws = a.take_bytes(lin);
  w.lin_bytes = lin;
  w.bytes = align_up(a.off, 256);
}

inline const float* P(const void* const* params, int i) { return (const float*)params[i]; }
inline float* G(void* const* grads, int i) { return grads ? (float*)grads[i] : nullptr; }

int bn_coef(const R1D& r, const float* stats, int rows, int C, long count, const void* const* params, int p_bn,
            void* const* buffers, int b_bn, float* coef, hipStream_t s) {
  if (r.d.training)
    return ecg_bn_finalize(stats, rows, C, (double)count, P(params, p_bn), P(params, p_bn + 1), (float*)buffers[b_bn],
                           (float*)buffers[b_bn + 1], (long long*)buffers[b_bn + 2], r.d.bn_momentum, r.d.bn_eps, coef,
                           s);
  return ecg_bn_eval_coef(C, P(params, p_bn), P(params, p_bn + 1), (const float*)buffers[b_bn],
                          (const float*)buffers[b_bn + 1], r.d.bn_eps, coef, s);
}

}  // namespace

// weight-gradient side stream of this plan (side_stream.h); its own instance, see there
static SideStream g_side1;
int ecg_resnet1d_side_enable(int on) {
  ECG_TRY(g_side1.init());
  g_side1.enabled = on != 0;
  return 0;
}
// The ResNet1D_SE plan's own switch.  Beside the image encoder (multimodal model: the signal encoder already runs on a
// side stream of the host's) its weight gradients stay on the encoder's stream: HIP maps streams onto four hardware
// queues, and a fifth busy stream -- with data parallelism the process group's stream is one more -- shares a queue with
// another, whose event waits then serialise both (same-call A/B: 7.15 -> 7.09 ms/step, and the one-rank rehearsal of the
// data-parallel path 7.52 -> 7.17).  Alone (12-lead configuration) the side stream is worth 3 %.
// (a hint ANDed with the global switch ecgmm_side_wgrad / ECGMM_SIDE_WGRAD, so profiling runs that turn every side
// stream off stay serialized)
static bool g_side1_alone = true;
extern "C" int ecgmm_resnet1d_side_wgrad(int on) {
  g_side1_alone = on != 0;
  return 0;
}

extern "C" size_t ecgmm_resnet1d_fwd_workspace(const ecgmm_resnet1d_desc* d) {
  R1D r;
  if (build(d, r)) return 0;
  Fwd1 w;
  layout_fwd(r, nullptr, w);
  return w.bytes;
}
extern "C" size_t ecgmm_resnet1d_bwd_workspace(const ecgmm_resnet1d_desc* d) {
  R1D r;
  if (build(d, r)) return 0;
  Bwd1 w;
  layout_bwd(r, nullptr, w);
  return w.bytes;
}

extern "C" int ecgmm_resnet1d_forward(const ecgmm_resnet1d_desc* d, const float* signal, const void* const* params,
                                      void* const* buffers, float* feat_out, void* ws, size_t ws_bytes,
                                      void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  R1D r;
  ECG_TRY(build(d, r));
  Fwd1 w;
  layout_fwd(r, ws, w);
  if (!ws || ws_bytes < w.bytes) ECG_FAIL(ECGMM_ERR_WORKSPACE, "resnet1d fwd: workspace %zu < %zu", ws_bytes, w.bytes);
  const int dt = r.d.dtype, N = r.d.N, cin = r.d.cin;
  float* st = r.d.training ? w.stats : nullptr;

  {  // every conv weight -> compute-dtype operand layouts, one launch
    EcgPackItem items[ECG_PACK_MAX];
    int n = 0;
    for (int i = 0; i < 3; ++i) {
      const Blk1& k = r.blk[i];
      Fwd1::B& b = w.b[i];
      items[n++] = {P(params, k.p0 + 0), b.w1f, b.w1d, k.cout, k.cin, 3};
      items[n++] = {P(params, k.p0 + 4), b.w2f, b.w2d, k.cout, k.cout, 3};
      if (k.down) items[n++] = {P(params, k.p0 + 12), b.wdf, b.wdd, k.cout, k.cin, 1};
    }
    ECG_TRY(ecg_pack_weight_batch(dt, items, n, s));
  }
  ECG_TRY(ecg_stem_pack(dt, P(params, 0), w.wstem, cin, 1, s));
  // (bf16: statistics rows per workgroup -- sums kept in registers across the workgroup's tiles -- instead of per tile)
  if (dt == ECGMM_BF16) {
    ECG_TRY(ecg_stem_fwd_wgrows(dt, signal, w.wstem, P(params, 1), w.y0, st, N, cin, 1, r.d.L, 1, s));
    ECG_TRY(bn_coef(r, w.stats, ecg_stem_wg_stats_rows(N, cin, 1, r.d.L, 1), 64, (long)N * r.L1, params, 2, buffers, 0,
                    w.coef0, s));
  } else {
    ECG_TRY(ecg_stem_fwd(dt, signal, w.wstem, P(params, 1), w.y0, st, N, cin, 1, r.d.L, 1, s));
    ECG_TRY(bn_coef(r, w.stats, ecg_stem_stats_rows(N, cin, 1, r.d.L, 1), 64, (long)N * r.L1, params, 2, buffers, 0,
                    w.coef0, s));
  }
  ECG_TRY(ecg_bnrelu_maxpool(dt, w.y0, w.coef0, w.p0, w.idx0, N, 1, r.L1, 64, s));

  const void* cur = w.p0;
  for (int i = 0; i < 3; ++i) {
    const Blk1& k = r.blk[i];
    Fwd1::B& b = w.b[i];
    const int p = k.p0, bb = k.b0;
    const long M = (long)N * k.lout;
    const int rows = ecg_conv_stats_rows(M);
    ConvGeom g1 = make_geom(N, 1, k.lin, k.cin, k.cout, 1, 3, k.stride, 0, 1);
    ConvGeom g2 = make_geom(N, 1, k.lout, k.cout, k.cout, 1, 3, 1, 0, 1);
    ECG_TRY(ecg_conv_igemm(dt, 0, g1, cur, b.w1f, b.y1, P(params, p + 1), nullptr, st, 0, s));
    ECG_TRY(bn_coef(r, w.stats, rows, k.cout, M, params, p + 2, buffers, bb, b.coef1, s));
    ECG_TRY(ecg_bn_act(dt, b.y1, b.coef1, nullptr, nullptr, nullptr, 1, 1, b.a1, M, k.cout, s));
    ECG_TRY(ecg_conv_igemm(dt, 0, g2, b.a1, b.w2f, b.y2, P(params, p + 5), nullptr, st, 0, s));
    ECG_TRY(bn_coef(r, w.stats, rows, k.cout, M, params, p + 6, buffers, bb + 3, b.coef2, s));
    // squeeze-excite gate from mean_L(bn2(y2))
    ECG_TRY(ecg_avgpool(dt, b.y2, b.m, N, k.lout, k.cout, b.coef2, s));
    if (ecg_se_mlp_fused_ok(k.cout, k.cr)) {
      ECG_TRY(ecg_se_mlp_fwd(b.m, P(params, p + 8), P(params, p + 9), P(params, p + 10), P(params, p + 11), b.h, b.g, N,
                             k.cout, k.cr, s));
    } else {
      ECG_TRY(ecg_linear_fwd(b.m, P(params, p + 8), P(params, p + 9), b.h, N, k.cout, k.cr, ECGMM_ACT_RELU, nullptr, s));
      ECG_TRY(ecg_linear_fwd(b.h, P(params, p + 10), P(params, p + 11), b.g, N, k.cr, k.cout, ECGMM_ACT_SIGMOID, nullptr,
                             s));
    }
    if (k.down) {
      ConvGeom gd = make_geom(N, 1, k.lin, k.cin, k.cout, 1, 1, k.stride, 0, 0);
      ECG_TRY(ecg_conv_igemm(dt, 0, gd, cur, b.wdf, b.yd, P(params, p + 13), nullptr, st, 0, s));
      ECG_TRY(bn_coef(r, w.stats, rows, k.cout, M, params, p + 14, buffers, bb + 6, b.coefd, s));
      ECG_TRY(ecg_bn_act(dt, b.y2, b.coef2, b.yd, b.coefd, b.g, k.lout, 1, b.out, M, k.cout, s));
    } else {
      ECG_TRY(ecg_bn_act(dt, b.y2, b.coef2, cur, nullptr, b.g, k.lout, 1, b.out, M, k.cout, s));
    }
    cur = b.out;
  }
  const int pc = r.p_cls;
  ECG_TRY(ecg_avgpool(dt, cur, w.pooled, N, r.blk[2].lout, 256, nullptr, s));
  ECG_TRY(ecg_linear_fwd(w.pooled, P(params, pc), P(params, pc + 1), w.h1, N, 256, 64, ECGMM_ACT_RELU, nullptr, s));
  const float* hin = w.h1;
  if (r.d.training && r.d.dropout_p > 0.f) {
    ECG_TRY(ecg_dropout_fwd(w.h1, w.hd, w.dmask, (long)N * 64, r.d.dropout_p, r.d.seed, r.d.offset, s));
    hin = w.hd;
  }
  ECG_TRY(ecg_linear_fwd(hin, P(params, pc + 2), P(params, pc + 3), feat_out, N, 64, r.d.num_classes, 0, nullptr, s));
  return 0;
}

// stages: 0 = classifier + avgpool, 1..3 = blocks 2..0, 4 = stem
extern "C" int ecgmm_resnet1d_backward(const ecgmm_resnet1d_desc* d, const float* signal, const float* dfeat,
                                       const void* const* params, void* const* grads, void* ws_fwd, void* ws_bwd,
                                       size_t ws_bwd_bytes, int stage_begin, int stage_end, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  R1D r;
  ECG_TRY(build(d, r));
  if (!r.d.training) ECG_FAIL(ECGMM_ERR_SHAPE, "resnet1d bwd: forward ran in eval mode (no batch statistics saved)");
  Fwd1 w;
  layout_fwd(r, ws_fwd, w);
  Bwd1 q;
  layout_bwd(r, ws_bwd, q);
  if (!ws_fwd || !ws_bwd || ws_bwd_bytes < q.bytes)
    ECG_FAIL(ECGMM_ERR_WORKSPACE, "resnet1d bwd: workspace %zu < %zu", ws_bwd_bytes, q.bytes);
  const int dt = r.d.dtype, N = r.d.N, cin = r.d.cin;
  ECG_TRY(g_side1.init());
  const bool side = g_side1.enabled && g_side1_alone;
  // weight gradients of this call run beside the dgrad chain: narrow launches (conv_wgrad.hip, pick_nsplit)
  ecg_conv_wgrad_narrow(side);
  struct NarrowOff { ~NarrowOff() { ecg_conv_wgrad_narrow(false); } } narrow_off;
  hipStream_t wst = side ? g_side1.s : s;  // stream of the weight-gradient kernels

  for (int st = stage_begin; st < stage_end; ++st) {
    if (st == 0) {
      g_side1.doneA = g_side1.doneB = g_side1.doneC = nullptr;
      const int pc = r.p_cls;
      const bool drop = r.d.dropout_p > 0.f;
      ECG_TRY(ecg_linear_bwd(dfeat, drop ? w.hd : w.h1, P(params, pc + 2), q.dfeat_h, G(grads, pc + 2),
                             G(grads, pc + 3), N, 64, r.d.num_classes, q.lin_ws, q.lin_bytes, s));
      if (drop) ECG_TRY(ecg_dropout_bwd(q.dfeat_h, w.dmask, q.dfeat_h, (long)N * 64, r.d.dropout_p, s));
      ECG_TRY(ecg_act_bwd(q.dfeat_h, w.h1, q.dh1, (long)N * 64, ECGMM_ACT_RELU, s));
      ECG_TRY(ecg_linear_bwd(q.dh1, w.pooled, P(params, pc), q.dpooled, G(grads, pc), G(grads, pc + 1), N, 256, 64,
                             q.lin_ws, q.lin_bytes, s));
      const int R = r.blk[2].lout;
      ECG_TRY(ecg_bcast_rows(dt, q.dpooled, q.X[0], N, R, 256, 1.f / (float)R, s));
    } else if (st <= 3) {
      const int i = 3 - st;
      const Blk1& k = r.blk[i];
      Fwd1::B& b = w.b[i];
      const int p = k.p0;
      const void* in = i == 0 ? w.p0 : w.b[i - 1].out;
      const void* dcur = q.X[(st - 1) & 1];
      void* din = q.X[st & 1];
      const long M = (long)N * k.lout;
      ConvGeom g1 = make_geom(N, 1, k.lin, k.cin, k.cout, 1, 3, k.stride, 0, 1);
      ConvGeom g2 = make_geom(N, 1, k.lout, k.cout, k.cout, 1, 3, 1, 0, 1);
      // out = relu(bn2(y2) * g + identity); gate gradient first.  Merged form (default): this pass also stores the masked
      // gradient dz and the per-sample sums the BatchNorm-backward reduction needs, so bn2's backward below is finalize +
      // apply only -- one read of (dcur, out, y2) less per block.  ECGMM_SE_MERGE=0: the two-pass form.
      static const bool se_merge = [] { const char* e = getenv("ECGMM_SE_MERGE"); return !(e && e[0] == '0'); }();
      if (se_merge) {
        ECG_TRY(ecg_se_gate_bn(dt, dcur, b.out, b.y2, b.coef2, q.dz, q.dg, q.sa1, q.sa2, q.sa3, N, k.lout, k.cout, s));
      } else {
        ECG_TRY(ecg_se_gate_grad(dt, dcur, b.out, b.y2, b.coef2, q.dg, N, k.lout, k.cout, s));
      }
      if (ecg_se_mlp_fused_ok(k.cout, k.cr)) {   // the SE MLP's backward in two launches (head_fused.hip)
        ECG_TRY(ecg_se_mlp_bwd(q.dg, b.g, b.h, b.m, P(params, p + 8), P(params, p + 10), q.ds, q.dh, q.dm, G(grads, p + 8),
                               G(grads, p + 9), G(grads, p + 10), G(grads, p + 11), N, k.cout, k.cr, 1.f / (float)k.lout, s));
      } else {
        ECG_TRY(ecg_act_bwd(q.dg, b.g, q.ds, (long)N * k.cout, ECGMM_ACT_SIGMOID, s));
        ECG_TRY(ecg_linear_bwd(q.ds, b.h, P(params, p + 10), q.dh, G(grads, p + 10), G(grads, p + 11), N, k.cr, k.cout,
                               q.lin_ws, q.lin_bytes, s));
        ECG_TRY(ecg_act_bwd(q.dh, b.h, q.dh, (long)N * k.cr, ECGMM_ACT_RELU, s));
        ECG_TRY(ecg_linear_bwd(q.dh, b.m, P(params, p + 8), q.dm, G(grads, p + 8), G(grads, p + 9), N, k.cout, k.cr,
                               q.lin_ws, q.lin_bytes, s));
        ECG_TRY(ecg_axpby(1.f / (float)k.lout, q.dm, 0.f, q.dm, (long)N * k.cout, s));
      }
      main_wait(s, g_side1.doneA);  // the previous block's wgrad2 has finished reading q.dy
      if (se_merge) {
        ECG_TRY(ecg_se_bn_rows(q.sa1, q.sa2, q.sa3, b.g, q.dm, N, k.lout, k.cout, q.se_rows, s));
        ECG_TRY(ecg_bn_bwd_tail(dt, q.dz, nullptr, b.y2, b.coef2, P(params, p + 6), G(grads, p + 6), G(grads, p + 7), q.dy,
                                q.se_rows, ecg_se_bn_nrows(), M, k.cout, q.bn_scratch, s, b.g, q.dm, k.lout, G(grads, p + 5)));
      } else {
        ECG_TRY(ecg_bn_bwd(dt, dcur, b.out, b.g, q.dm, k.lout, b.y2, b.coef2, P(params, p + 6), G(grads, p + 6),
                           G(grads, p + 7), q.dy, q.dz, G(grads, p + 5), M, k.cout, q.bn_scratch, s));
      }
      if (G(grads, p + 4)) {
        if (side) g_side1.fork(s);
        ECG_TRY(ecg_conv_wgrad(dt, g2, b.a1, q.dy, G(grads, p + 4), 0, q.wg_ws, q.wg_bytes, wst));
        if (side) g_side1.doneA = g_side1.mark();
      }
      ECG_TRY(ecg_conv_igemm(dt, 1, g2, q.dy, b.w2d, q.da, nullptr, nullptr, nullptr, 0, s));
      main_wait(s, g_side1.doneB);
      ECG_TRY(ecg_bn_bwd(dt, q.da, b.y1 /* mask recomputed from y1 */, nullptr, nullptr, 1, b.y1, b.coef1, P(params, p + 2), G(grads, p + 2),
                         G(grads, p + 3), q.dy1, nullptr, G(grads, p + 1), M, k.cout, q.bn_scratch, s));
      if (G(grads, p + 0)) {
        if (side) g_side1.fork(s);
        ECG_TRY(ecg_conv_wgrad(dt, g1, in, q.dy1, G(grads, p + 0), 0, q.wg_ws, q.wg_bytes, wst));
        if (side) g_side1.doneB = g_side1.mark();
      }
      if (k.down) {
        ConvGeom gd = make_geom(N, 1, k.lin, k.cin, k.cout, 1, 1, k.stride, 0, 0);
        main_wait(s, g_side1.doneC);
        ECG_TRY(ecg_bn_bwd(dt, q.dz, nullptr, nullptr, nullptr, 1, b.yd, b.coefd, P(params, p + 14), G(grads, p + 14),
                           G(grads, p + 15), q.dyd, nullptr, G(grads, p + 13), M, k.cout, q.bn_scratch, s));
        if (G(grads, p + 12)) {
          if (side) g_side1.fork(s);
          ECG_TRY(ecg_conv_wgrad(dt, gd, in, q.dyd, G(grads, p + 12), 0, q.wg_ws, q.wg_bytes, wst));
          if (side) g_side1.doneC = g_side1.mark();
        }
        static const bool fold_on = [] { const char* e = getenv("ECGMM_DOWN_FOLD"); return !(e && e[0] == '0'); }();
        ConvEpi ed = {};   // downsample branch folded into the stride-2 dgrad (see plan_resnet18.hip)
        if (fold_on) { ed.src2 = q.dyd; ed.wpk2 = b.wdd; }
        if (ed.src2) ECG_TRY(ecg_conv_igemm(dt, 1, g1, q.dy1, b.w1d, din, nullptr, nullptr, nullptr, 0, s, &ed));
        if (!ed.src2_done) {
          ECG_TRY(ecg_conv_igemm(dt, 1, gd, q.dyd, b.wdd, q.dtmp, nullptr, nullptr, nullptr, 0, s));
          ECG_TRY(ecg_conv_igemm(dt, 1, g1, q.dy1, b.w1d, din, nullptr, q.dtmp, nullptr, 0, s));
        }
      } else {
        ECG_TRY(ecg_conv_igemm(dt, 1, g1, q.dy1, b.w1d, din, nullptr, q.dz, nullptr, 0, s));
      }
    } else if (st == 4) {
      const void* dp0 = q.X[3 & 1];
      if (ecg_stem_fuse_on()) {
        ECG_TRY(ecg_pool_bn_bwd(dt, dp0, w.p0, w.idx0, w.y0, w.coef0, P(params, 2), G(grads, 2), G(grads, 3), q.big1,
                                G(grads, 1), N, 1, r.L1, 64, q.bn_scratch, s));
      } else {
        ECG_TRY(ecg_maxpool_relu_bwd(dt, dp0, w.p0, w.idx0, q.big0, N, 1, r.L1, 64, s));
        ECG_TRY(ecg_bn_bwd(dt, q.big0, nullptr, nullptr, nullptr, 1, w.y0, w.coef0, P(params, 2), G(grads, 2),
                           G(grads, 3), q.big1, nullptr, G(grads, 1), (long)N * r.L1, 64, q.bn_scratch, s));
      }
      if (G(grads, 0))  // last kernel: stays on the caller's stream (own slab buffer), see plan_resnet18.hip
        ECG_TRY(ecg_stem_wgrad(dt, signal, q.big1, G(grads, 0), 0, q.stem_ws, q.stem_bytes, N, cin, 1, r.d.L, 1, s));
    } else {
      ECG_FAIL(ECGMM_ERR_SHAPE, "resnet1d bwd: stage %d out of range", st);
    }
  }
  if (side) ECG_TRY(g_side1.wait_on(s));  // join: everything the side stream did is ordered before the caller's next work
  return 0;
}
