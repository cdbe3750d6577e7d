// ResNet18 image encoder as a native launch plan (forward + backward), gfx950.
// Mirrors torchvision.models.resnet18() as instantiated by the reference at
// multimodal_paper_modal_balance.py:210,221 and train_image_only.py:92-99:
//   conv7x7/2 -> BN -> ReLU -> maxpool3/2 -> 4 stages x 2 BasicBlocks -> global avgpool -> fc
// Activations live channels-last in the compute dtype inside a caller-owned workspace; the image
// enters as NCHW fp32 exactly as the reference's DataLoader hands it over.
//
// Parameter table order (62 entries, == named_parameters() order of the Python module):
//   conv1.weight, bn1.weight, bn1.bias,
//   layer{1..4}.{0,1}.{conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias,
//                      [downsample.0.weight, downsample.1.weight, downsample.1.bias]},
//   fc.weight, fc.bias
// Buffer table order (60 entries): per BatchNorm in the same walk: running_mean, running_var,
//   num_batches_tracked (int64).
#include <stdlib.h>

#include "ops.h"
#include "side_stream.h"

namespace {
// The 2-D stem by recompute (conv_stem_fused.hip): -1 = read ECGMM_STEM_RECOMPUTE.  DEFAULT OFF: measured in the step
// (same-call A/B, batch 256, round 3) it removes 1.6 GB of fabric traffic per step and is 0.2 ms SLOWER (7.12 -> 7.32 ms):
// its kernels are VALU-issue-bound (~2600 instructions per 128-pixel tile for 96 MFMAs), not byte-bound -- stand-alone
// forward 418 -> 396 us, backward 401 -> 584 us.  Kept as a tested option (memory: -1.2 GB of workspace at batch 256).
int g_stem_recompute = -1;
bool stem_recompute(int dtype) {
  if (g_stem_recompute < 0) { const char* e = getenv("ECGMM_STEM_RECOMPUTE"); g_stem_recompute = (e && e[0] == '1'); }
  return g_stem_recompute != 0 && ecg_stem_fused_ok(dtype, 3, 7);
}
// bn2's backward takes the block's ReLU mask from one bit per element written by the forward activation pass instead of re-reading
// the activated tensor (ECGMM_RELU_BITS=0: re-read it).  Read once per process: forward and backward must agree.
bool relu_bits_on() {
  static const bool on = [] { const char* e = getenv("ECGMM_RELU_BITS"); return !(e && e[0] == '0'); }();
  return on;
}
constexpr long FUSE_NEVER = 1L << 40;
long g_fuse_min_m = -1;  // pixel-count threshold of the fused BatchNorm-backward reductions (-1: read ECGMM_BN_FUSE_MIN_M)

struct BlockCfg {
  int cin, cout, stride, hin, win, hout, wout;
  bool down;
  int p_conv1, p_bn1, p_conv2, p_bn2, p_dconv, p_dbn;  // param indices (weight; bn bias = +1)
  int b_bn1, b_bn2, b_dbn;                              // buffer indices (rm; rv = +1; nbt = +2)
};

struct R18 {
  ecgmm_resnet18_desc d;
  int H1, W1, H2, W2;
  BlockCfg blk[8];
  int p_fc;
  size_t max_act;  // largest block-level activation (elements)
};

int build(const ecgmm_resnet18_desc* d, R18& r) {
  if (!d) ECG_FAIL(ECGMM_ERR_SHAPE, "resnet18: null desc");
  if (d->dtype != ECGMM_BF16 && d->dtype != ECGMM_F32) ECG_FAIL(ECGMM_ERR_DTYPE, "resnet18: bad dtype %d", d->dtype);
  if (d->N < 1 || d->H < 32 || d->W < 32) ECG_FAIL(ECGMM_ERR_SHAPE, "resnet18: bad input %dx%dx%d", d->N, d->H, d->W);
  r.d = *d;
  r.H1 = (d->H + 6 - 7) / 2 + 1;
  r.W1 = (d->W + 6 - 7) / 2 + 1;
  r.H2 = (r.H1 + 2 - 3) / 2 + 1;
  r.W2 = (r.W1 + 2 - 3) / 2 + 1;
  int pi = 3, bi = 3, h = r.H2, w = r.W2, cin = 64;
  r.max_act = 0;
  for (int L = 0; L < 4; ++L) {
    int cout = 64 << L;
    for (int b = 0; b < 2; ++b) {
      BlockCfg& k = r.blk[L * 2 + b];
      k.cin = cin; k.cout = cout; k.stride = (b == 0 && L > 0) ? 2 : 1;
      k.hin = h; k.win = w;
      k.hout = (h + 2 - 3) / k.stride + 1;
      k.wout = (w + 2 - 3) / k.stride + 1;
      k.down = (k.stride != 1 || cin != cout);
      k.p_conv1 = pi; k.p_bn1 = pi + 1; k.p_conv2 = pi + 3; k.p_bn2 = pi + 4; pi += 6;
      k.b_bn1 = bi; k.b_bn2 = bi + 3; bi += 6;
      if (k.down) {
        k.p_dconv = pi; k.p_dbn = pi + 1; pi += 3;
        k.b_dbn = bi; bi += 3;
      } else {
        k.p_dconv = k.p_dbn = k.b_dbn = -1;
      }
      size_t a = (size_t)d->N * k.hin * k.win * k.cin;
      if (a > r.max_act) r.max_act = a;
      a = (size_t)d->N * k.hout * k.wout * k.cout;
      if (a > r.max_act) r.max_act = a;
      h = k.hout; w = k.wout; cin = cout;
    }
  }
  r.p_fc = pi;
  if (pi + 2 != ECGMM_RESNET18_NPARAMS || bi != ECGMM_RESNET18_NBUFFERS)
    ECG_FAIL(ECGMM_ERR_SHAPE, "resnet18: internal table mismatch %d %d", pi + 2, bi);
  return 0;
}

// saved-for-backward workspace
struct FwdWs {
  void* wstem;
  void* y0; float* coef0; void* p0; unsigned char* idx0;
  struct B {
    void *w1f, *w1d, *w2f, *w2d, *wdf, *wdd;
    void *y1, *a1, *y2, *yd, *out;
    unsigned char* bits;   // ReLU mask of `out`, one bit per element (bf16 plans; null otherwise)
    float *coef1, *coef2, *coefd;
  } b[8];
  float* pooled;
  float* stats;  // transient partial-sum rows (largest layer)
  float* stats_d;  // the same for the downsample convolutions (they run on the side stream beside conv1 / conv2)
  size_t bytes;
};

void layout_fwd(const R18& r, void* base, FwdWs& w) {
  Arena a(base);
  const size_t es = dtype_size(r.d.dtype);
  const int N = r.d.N;
  w.wstem = a.take_bytes(ecg_stem_packed_elems(3, 7) * es);
  // (the recomputing stem never materialises the conv output)
  w.y0 = stem_recompute(r.d.dtype) ? nullptr : a.take_bytes((size_t)N * r.H1 * r.W1 * 64 * es);
  w.coef0 = a.take<float>(4 * 64);
  w.p0 = a.take_bytes((size_t)N * r.H2 * r.W2 * 64 * es);
  w.idx0 = a.take<unsigned char>((size_t)N * r.H2 * r.W2 * 64);
  size_t max_rows_c = (size_t)(stem_recompute(r.d.dtype) ? ecg_stem_stats_only_rows(N, 3, r.d.H, r.d.W, 7)
                                                          : ecg_stem_stats_rows(N, 3, r.d.H, r.d.W, 7)) * 2 * 64;
  max_rows_c += (size_t)ECG_TAIL_ROWS * 2 * 64;
  size_t max_rows_d = 64;
  for (int i = 0; i < 8; ++i) {
    const BlockCfg& k = r.blk[i];
    FwdWs::B& b = w.b[i];
    size_t osz = (size_t)N * k.hout * k.wout * k.cout;
    b.w1f = a.take_bytes((size_t)k.cout * k.cin * 9 * es);
    b.w1d = a.take_bytes((size_t)k.cout * k.cin * 9 * es);
    b.w2f = a.take_bytes((size_t)k.cout * k.cout * 9 * es);
    b.w2d = a.take_bytes((size_t)k.cout * k.cout * 9 * es);
    b.y1 = a.take_bytes(osz * es);
    b.a1 = a.take_bytes(osz * es);
    b.y2 = a.take_bytes(osz * es);
    b.out = a.take_bytes(osz * es);
    // ReLU mask of `out` as bits (bf16 training plans): bn2's backward reads these 1/16-size bytes instead of `out`
    b.bits = es == 2 ? (unsigned char*)a.take_bytes(osz / 8) : nullptr;
    b.coef1 = a.take<float>(4 * k.cout);
    b.coef2 = a.take<float>(4 * k.cout);
    if (k.down) {
      b.wdf = a.take_bytes((size_t)k.cout * k.cin * es);
      b.wdd = a.take_bytes((size_t)k.cout * k.cin * es);
      b.yd = a.take_bytes(osz * es);
      b.coefd = a.take<float>(4 * k.cout);
    } else {
      b.wdf = b.wdd = b.yd = nullptr;
      b.coefd = nullptr;
    }
    size_t rows_c = (size_t)(ecg_conv_stats_rows((long)N * k.hout * k.wout) + ECG_TAIL_ROWS) * 2 * k.cout;
    if (rows_c > max_rows_c) max_rows_c = rows_c;
    if (k.down && rows_c > max_rows_d) max_rows_d = rows_c;
  }
  w.pooled = a.take<float>((size_t)N * 512);
  w.stats = a.take<float>(max_rows_c);
  w.stats_d = a.take<float>(max_rows_d);
  w.bytes = align_up(a.off, 256);
}

struct BwdWs {
  void* X[2];   // gradient w.r.t. block outputs (ping-pong)
  void* dz;     // masked residual-branch gradient
  // operands of the side-stream weight-gradient kernels, two of each (blocks alternate): the main stream's next
  // BatchNorm backward then never has to wait for the previous block's wgrad to finish reading, only for the one
  // two blocks back (an exposed cross-stream wait costs 35-140 us here, and the side stream does run behind)
  void* dy[2];   // gradient w.r.t. conv2's raw output
  void* dy1[2];  // gradient w.r.t. conv1's raw output
  void* dyd[2];  // gradient w.r.t. the downsample conv's raw output
  void* da;     // gradient w.r.t. a1
  void* dtmp;   // downsample-path input gradient
  void* big0;   // stem: dz0
  void* big1;   // stem: dy0
  float* dpooled;
  float* bn_scratch;
  float *red1, *red2;  // partial rows of BatchNorm-backward reductions fused into dgrad epilogues (ConvEpi): bn1 / bn2
  void* wg_ws; size_t wg_bytes;
  void* stem_ws; size_t stem_bytes;  // the stem's slab buffer (it runs on the caller's stream, concurrently with side-stream wgrads)
  void* lin_ws; size_t lin_bytes;
  size_t bytes;
};

void layout_bwd(const R18& r, void* base, BwdWs& w) {
  Arena a(base);
  const size_t es = dtype_size(r.d.dtype);
  const int N = r.d.N;
  for (int i = 0; i < 2; ++i) w.X[i] = a.take_bytes(r.max_act * es);
  w.dz = a.take_bytes(r.max_act * es);
  for (int i = 0; i < 2; ++i) {
    w.dy[i] = a.take_bytes(r.max_act * es);
    w.dy1[i] = a.take_bytes(r.max_act * es);
    w.dyd[i] = a.take_bytes(r.max_act * es);
  }
  w.da = a.take_bytes(r.max_act * es);
  w.dtmp = a.take_bytes(r.max_act * es);
  const bool recompute = stem_recompute(r.d.dtype);
  size_t big = recompute ? 0 : (size_t)N * r.H1 * r.W1 * 64;   // (recomputing stem: neither dz0 nor dy0 exist)
  w.big0 = a.take_bytes(big * es);
  w.big1 = a.take_bytes(big * es);
  w.dpooled = a.take<float>((size_t)N * 512);
  size_t bn = ecg_bn_bwd_scratch(r.d.dtype, (long)N * r.H1 * r.W1, 64);
  size_t wg = 0;
  for (int i = 0; i < 8; ++i) {
    const BlockCfg& k = r.blk[i];
    size_t s = ecg_bn_bwd_scratch(r.d.dtype, (long)N * k.hout * k.wout, k.cout);
    if (s > bn) bn = s;
    size_t g1 = ecg_conv_wgrad_workspace(r.d.dtype, make_geom(N, k.hin, k.win, k.cin, k.cout, 3, 3, k.stride, 1, 1));
    size_t g2 = ecg_conv_wgrad_workspace(r.d.dtype, make_geom(N, k.hout, k.wout, k.cout, k.cout, 3, 3, 1, 1, 1));
    if (g1 > wg) wg = g1;
    if (g2 > wg) wg = g2;
    if (k.down) {
      size_t g3 = ecg_conv_wgrad_workspace(r.d.dtype, make_geom(N, k.hin, k.win, k.cin, k.cout, 1, 1, k.stride, 0, 0));
      if (g3 > wg) wg = g3;
    }
  }
  w.bn_scratch = (float*)a.take_bytes(bn);
  w.red1 = a.take<float>((size_t)512 * 2 * 512);
  w.red2 = a.take<float>((size_t)512 * 2 * 512);
  w.wg_ws = a.take_bytes(wg);
  w.wg_bytes = wg;
  w.stem_bytes = recompute ? ecg_stem_pool_bwd_workspace(N, 3, r.d.H, r.d.W) : ecg_stem_wgrad_workspace(N, 3, r.d.H, r.d.W, 7);
  w.stem_ws = a.take_bytes(w.stem_bytes);
  w.lin_bytes = ecg_linear_bwd_scratch(N, 512, r.d.out_dim);
  w.lin_ws = a.take_bytes(w.lin_bytes);
  w.bytes = align_up(a.off, 256);
}

// ---- side stream for the weight-gradient kernels (side_stream.h) ----------------------------------
SideStream g_side;
int side_init() { return g_side.init(); }
inline hipEvent_t side_next_ev() { return g_side.next_ev(); }
inline void side_fork(hipStream_t main) { g_side.fork(main); }
inline hipEvent_t side_mark() { return g_side.mark(); }

inline const float* P(const void* const* params, int i) { return (const float*)params[i]; }
inline float* G(void* const* grads, int i) { return grads ? (float*)grads[i] : nullptr; }

// BN statistics of a fresh conv output -> coefficients (train: batch stats + running update; eval: running stats)
int bn_coef(const R18& r, const float* stats, int rows, int C, long count, const void* const* params, int p_bn,
            void* const* buffers, int b_bn, float* coef, hipStream_t s) {
  if (r.d.training)
    return ecg_bn_finalize(stats, rows, C, (double)count, P(params, p_bn), P(params, p_bn + 1), (float*)buffers[b_bn],
                           (float*)buffers[b_bn + 1], (long long*)buffers[b_bn + 2], r.d.bn_momentum, r.d.bn_eps, coef,
                           s);
  return ecg_bn_eval_coef(C, P(params, p_bn), P(params, p_bn + 1), (const float*)buffers[b_bn],
                          (const float*)buffers[b_bn + 1], r.d.bn_eps, coef, s);
}

// BatchNorm (training statistics from a conv's partial rows, or running statistics in eval mode) + activation pass.
// Where the rows are few enough the finalize is folded into the activation pass itself (ecg_bn_act_fold: one dependent
// ~5 us launch less on the forward's critical path, 16 times per forward).
int bn_then_act(const R18& r, const float* stats, int rows, int C, long count, const void* const* params, int p_bn,
                void* const* buffers, int b_bn, float* coef, const void* y, const void* res, const float* rcoef, void* out,
                hipStream_t s, unsigned char* relu_bits = nullptr) {
  const int dt = r.d.dtype;
  if (r.d.training && ecg_bn_fold_ok(C, rows)) {
    EcgBnFold f = {stats, rows, (double)count, P(params, p_bn), P(params, p_bn + 1), (float*)buffers[b_bn],
                   (float*)buffers[b_bn + 1], (long long*)buffers[b_bn + 2], r.d.bn_momentum, r.d.bn_eps};
    return ecg_bn_act_fold(dt, y, coef, f, res, rcoef, nullptr, 1, 1, out, count, C, s, relu_bits);
  }
  ECG_TRY(bn_coef(r, stats, rows, C, count, params, p_bn, buffers, b_bn, coef, s));
  return ecg_bn_act(dt, y, coef, res, rcoef, nullptr, 1, 1, out, count, C, s, relu_bits);
}

}  // namespace

// Runtime switch for the side-stream weight-gradient overlap (default on; ECGMM_SIDE_WGRAD=0 disables it
// at start-up).  bench.py turns it off for its serialized kernel-timing pass.
extern "C" int ecgmm_side_wgrad(int on) {
  ECG_TRY(side_init());
  g_side.enabled = on != 0;
  return ecg_resnet1d_side_enable(on);
}

// Data-parallel overlap: with defer = 1 a backward call returns WITHOUT joining the side stream to the caller's
// stream (the dgrad chain of the next stage group then keeps overlapping this group's weight-gradient tail); the
// caller orders its consumer (the all-reduce stream) after the weight gradients with ecgmm_side_wait() and must
// run the last stage group with defer = 0, whose join covers everything.
extern "C" int ecgmm_side_defer_join(int defer) {
  ECG_TRY(side_init());
  g_side.defer_join = defer != 0;
  return 0;
}

// Everything the side stream has been given so far happens-before later work on `stream`.
extern "C" int ecgmm_side_wait(void* stream) {
  ECG_TRY(side_init());
  if (!g_side.enabled) return 0;
  return g_side.wait_on((hipStream_t)stream);
}

// The library's weight-gradient side stream (NULL when it is switched off), and "fork": everything enqueued on `stream`
// so far happens-before later work on the side stream.  A data-parallel host issues its early all-reduces from the side
// stream's context (after a fork) instead of from a stream of its own: a fifth busy HIP stream oversubscribes the four
// hardware queues (GPU_MAX_HW_QUEUES), and two streams that share a queue serialise each other's event waits.
extern "C" void* ecgmm_side_stream(void) {
  if (side_init() != 0 || !g_side.enabled) return nullptr;
  return (void*)g_side.s;
}
extern "C" int ecgmm_side_fork(void* stream) {
  ECG_TRY(side_init());
  if (g_side.enabled) g_side.fork((hipStream_t)stream);
  return 0;
}

extern "C" size_t ecgmm_resnet18_fwd_workspace(const ecgmm_resnet18_desc* d) {
  R18 r;
  if (build(d, r)) return 0;
  FwdWs w;
  layout_fwd(r, nullptr, w);
  return w.bytes;
}

extern "C" size_t ecgmm_resnet18_bwd_workspace(const ecgmm_resnet18_desc* d) {
  R18 r;
  if (build(d, r)) return 0;
  BwdWs w;
  layout_bwd(r, nullptr, w);
  return w.bytes;
}

extern "C" int ecgmm_resnet18_forward(const ecgmm_resnet18_desc* d, const float* image, const void* const* params,
                                      void* const* buffers, float* feat_out, void* ws, size_t ws_bytes,
                                      void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  R18 r;
  ECG_TRY(build(d, r));
  FwdWs w;
  layout_fwd(r, ws, w);
  if (!ws || ws_bytes < w.bytes) ECG_FAIL(ECGMM_ERR_WORKSPACE, "resnet18 fwd: workspace %zu < %zu", ws_bytes, w.bytes);
  const int dt = r.d.dtype, N = r.d.N;
  const int stats_rows = r.d.training ? 1 : 0;
  ECG_TRY(side_init());
  static const bool down_side_on = [] { const char* e = getenv("ECGMM_DOWN_SIDE"); return !(e && e[0] == '0'); }();
  const bool side_fwd = g_side.enabled && down_side_on;

  ecg_tl_mark(100, s);
  // ---- every conv weight -> compute-dtype operand layouts, one launch.  (Running it on the side stream underneath
  // the stem convolution was measured: no gain -- a cross-stream event wait costs 35-140 us of GPU-side latency on
  // this platform, more than the 0.1 ms pack hides.)
  {
    EcgPackItem items[ECG_PACK_MAX];
    int n = 0;
    for (int i = 0; i < 8; ++i) {
      const BlockCfg& k = r.blk[i];
      FwdWs::B& b = w.b[i];
      items[n++] = {P(params, k.p_conv1), b.w1f, b.w1d, k.cout, k.cin, 9};
      items[n++] = {P(params, k.p_conv2), b.w2f, b.w2d, k.cout, k.cout, 9};
      if (k.down) items[n++] = {P(params, k.p_dconv), b.wdf, b.wdd, k.cout, k.cin, 1};
    }
    ECG_TRY(ecg_pack_weight_batch(dt, items, n, s));
  }
  // ---- stem
  ECG_TRY(ecg_stem_pack(dt, P(params, 0), w.wstem, 3, 7, s));
  if (stem_recompute(dt)) {
    // pass 1: statistics only; pass 2: conv recomputed -> bn -> relu -> max-pool.  The 64 x H1 x W1 conv output (411 MB at
    // batch 256) is never written: 154 + 154 MB read, 154 MB written instead of 154 + 411 + 411 read/written + 154
    if (stats_rows) ECG_TRY(ecg_stem_stats_only(dt, image, w.wstem, nullptr, w.stats, N, 3, r.d.H, r.d.W, 7, s));
    ECG_TRY(bn_coef(r, w.stats, ecg_stem_stats_only_rows(N, 3, r.d.H, r.d.W, 7), 64, (long)N * r.H1 * r.W1, params, 1,
                    buffers, 0, w.coef0, s));
    ECG_TRY(ecg_stem_pool_fwd(image, w.wstem, w.coef0, w.p0, w.idx0, N, 3, r.d.H, r.d.W, s));
  } else {
    // (bf16: statistics rows per workgroup -- sums kept in registers across the workgroup's tiles -- instead of per tile)
    if (dt == ECGMM_BF16) {
      ECG_TRY(ecg_stem_fwd_wgrows(dt, image, w.wstem, nullptr, w.y0, stats_rows ? w.stats : nullptr, N, 3, r.d.H, r.d.W, 7, s));
      ECG_TRY(bn_coef(r, w.stats, ecg_stem_wg_stats_rows(N, 3, r.d.H, r.d.W, 7), 64, (long)N * r.H1 * r.W1, params, 1,
                      buffers, 0, w.coef0, s));
    } else {
      ECG_TRY(ecg_stem_fwd(dt, image, w.wstem, nullptr, w.y0, stats_rows ? w.stats : nullptr, N, 3, r.d.H, r.d.W, 7, s));
      ECG_TRY(bn_coef(r, w.stats, ecg_stem_stats_rows(N, 3, r.d.H, r.d.W, 7), 64, (long)N * r.H1 * r.W1, params, 1,
                      buffers, 0, w.coef0, s));
    }
    ECG_TRY(ecg_bnrelu_maxpool(dt, w.y0, w.coef0, w.p0, w.idx0, N, r.H1, r.W1, 64, s));
  }

  ecg_tl_mark(101, s);
  const void* cur = w.p0;
  for (int i = 0; i < 8; ++i) {
    const BlockCfg& k = r.blk[i];
    FwdWs::B& b = w.b[i];
    const long M = (long)N * k.hout * k.wout;
    const int rows = ecg_conv_stats_rows(M);
    ConvGeom g1 = make_geom(N, k.hin, k.win, k.cin, k.cout, 3, 3, k.stride, 1, 1);
    ConvGeom g2 = make_geom(N, k.hout, k.wout, k.cout, k.cout, 3, 3, 1, 1, 1);
    // (ConvEpi.wg_rows: the halo kernel writes one partial-sum row per workgroup instead of one per 64 pixels)
    ConvEpi e1 = {}, e2 = {};
    e1.wg_rows = e2.wg_rows = 1;
    // the downsample branch (1x1 stride-2 conv + its BatchNorm statistics) only meets the main branch at the block's
    // last pass: it runs on the weight-gradient side stream (idle in the forward) beside conv1 -> bn1 -> conv2
    hipEvent_t down_done = nullptr;
    const bool down_side = k.down && side_fwd;
    if (down_side) {
      ConvGeom gd = make_geom(N, k.hin, k.win, k.cin, k.cout, 1, 1, k.stride, 0, 0);
      side_fork(s);
      ECG_TRY(ecg_conv_igemm(dt, 0, gd, cur, b.wdf, b.yd, nullptr, nullptr, stats_rows ? w.stats_d : nullptr, 0, g_side.s));
      ECG_TRY(bn_coef(r, w.stats_d, rows, k.cout, M, params, k.p_dbn, buffers, k.b_dbn, b.coefd, g_side.s));
      down_done = side_mark();
    }
    ECG_TRY(ecg_conv_igemm(dt, 0, g1, cur, b.w1f, b.y1, nullptr, nullptr, stats_rows ? w.stats : nullptr, 0, s, &e1));
    ECG_TRY(bn_then_act(r, w.stats, e1.stats_rows, k.cout, M, params, k.p_bn1, buffers, k.b_bn1, b.coef1, b.y1, nullptr,
                        nullptr, b.a1, s));
    ECG_TRY(ecg_conv_igemm(dt, 0, g2, b.a1, b.w2f, b.y2, nullptr, nullptr, stats_rows ? w.stats : nullptr, 0, s, &e2));
    if (k.down) {
      if (down_side) {
        main_wait(s, down_done);
      } else {
        ConvGeom gd = make_geom(N, k.hin, k.win, k.cin, k.cout, 1, 1, k.stride, 0, 0);
        // (own row buffer: bn2's rows in w.stats are still unread -- their finalize is folded into the pass below)
        ECG_TRY(ecg_conv_igemm(dt, 0, gd, cur, b.wdf, b.yd, nullptr, nullptr, stats_rows ? w.stats_d : nullptr, 0, s));
        ECG_TRY(bn_coef(r, w.stats_d, rows, k.cout, M, params, k.p_dbn, buffers, k.b_dbn, b.coefd, s));
      }
      ECG_TRY(bn_then_act(r, w.stats, e2.stats_rows, k.cout, M, params, k.p_bn2, buffers, k.b_bn2, b.coef2, b.y2, b.yd,
                          b.coefd, b.out, s, r.d.training && relu_bits_on() ? b.bits : nullptr));
    } else {
      ECG_TRY(bn_then_act(r, w.stats, e2.stats_rows, k.cout, M, params, k.p_bn2, buffers, k.b_bn2, b.coef2, b.y2, cur,
                          nullptr, b.out, s, r.d.training && relu_bits_on() ? b.bits : nullptr));
    }
    cur = b.out;
    ecg_tl_mark(102 + i, s);
  }
  const BlockCfg& last = r.blk[7];
  ECG_TRY(ecg_avgpool(dt, cur, w.pooled, N, last.hout * last.wout, 512, nullptr, s));
  ECG_TRY(ecg_linear_fwd(w.pooled, P(params, r.p_fc), P(params, r.p_fc + 1), feat_out, N, 512, r.d.out_dim, 0, nullptr,
                         s));
  ecg_tl_mark(110, s);
  return 0;
}

// stages: 0 = fc + avgpool, 1..8 = blocks 7..0, 9 = stem.  Run [stage_begin, stage_end).
extern "C" int ecgmm_resnet18_backward(const ecgmm_resnet18_desc* d, const float* image, const float* dfeat,
                                       const void* const* params, void* const* grads, void* ws_fwd, void* ws_bwd,
                                       size_t ws_bwd_bytes, int stage_begin, int stage_end, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  R18 r;
  ECG_TRY(build(d, r));
  if (!r.d.training) ECG_FAIL(ECGMM_ERR_SHAPE, "resnet18 bwd: forward ran in eval mode (no batch statistics saved)");
  FwdWs w;
  layout_fwd(r, ws_fwd, w);
  BwdWs q;
  layout_bwd(r, ws_bwd, q);
  if (!ws_fwd || !ws_bwd || ws_bwd_bytes < q.bytes)
    ECG_FAIL(ECGMM_ERR_WORKSPACE, "resnet18 bwd: workspace %zu < %zu", ws_bwd_bytes, q.bytes);
  const int dt = r.d.dtype, N = r.d.N;
  ECG_TRY(side_init());
  const bool side = g_side.enabled;
  // weight gradients of this call run beside the dgrad chain: narrow launches (conv_wgrad.hip, pick_nsplit)
  ecg_conv_wgrad_narrow(side, N >= 192 ? 192 : 256);
  struct NarrowOff { ~NarrowOff() { ecg_conv_wgrad_narrow(false); } } narrow_off;
  hipStream_t ws = side ? g_side.s : s;  // stream of the weight-gradient kernels
  // (capping the persistent input-gradient kernels of this call to the CUs the side stream leaves free -- 160 / 192 / 224 of 256 --
  //  measured 6.57-6.60 / 6.62-6.63 / 6.53-6.55 ms against 6.54-6.56 uncapped: the weight gradients hold their CUs only part of the time)

  if (stage_begin == 0) ecg_tl_mark(200, s);
  for (int st = stage_begin; st < stage_end; ++st) {
    if (st > 0) ecg_tl_mark(200 + st, s);   // (mark 200 + st = stage st - 1 enqueued behind it: fc = 201, blocks 7..0 = 202..209)
    if (st == 0) {
      g_side.doneA = g_side.doneB = g_side.doneC = nullptr;
      for (int a = 0; a < 3; ++a) g_side.done2[a][0] = g_side.done2[a][1] = nullptr;
      const BlockCfg& last = r.blk[7];
      ECG_TRY(ecg_linear_bwd(dfeat, w.pooled, P(params, r.p_fc), q.dpooled, G(grads, r.p_fc), G(grads, r.p_fc + 1), N,
                             512, r.d.out_dim, q.lin_ws, q.lin_bytes, s));
      const int R = last.hout * last.wout;
      ECG_TRY(ecg_bcast_rows(dt, q.dpooled, q.X[0], N, R, 512, 1.f / (float)R, s));
    } else if (st <= 8) {
      const int i = 8 - st;
      const BlockCfg& k = r.blk[i];
      FwdWs::B& b = w.b[i];
      const void* in = i == 0 ? w.p0 : w.b[i - 1].out;
      const void* dcur = q.X[(st - 1) & 1];
      void* din = q.X[st & 1];
      const long M = (long)N * k.hout * k.wout;
      ConvGeom g1 = make_geom(N, k.hin, k.win, k.cin, k.cout, 3, 3, k.stride, 1, 1);
      ConvGeom g2 = make_geom(N, k.hout, k.wout, k.cout, k.cout, 3, 3, 1, 1, 1);
      const int pp = st & 1;  // which of the two dy / dy1 / dyd buffers (and their reader events) this block uses
      void *dyb = q.dy[pp], *dy1b = q.dy1[pp], *dydb = q.dyd[pp];
      // The REDUCTION pass of a BatchNorm backward is fused into the epilogue of the dgrad that produces its input,
      // wherever that dgrad runs on the halo kernel (ConvEpi): bn1's into this block's conv2 dgrad; bn2's into the conv1
      // dgrad of the NEXT block (processed one stage earlier), which then also stores the gradient already masked by
      // this block's ReLU -- that buffer IS dz.  Pure function of the geometry, so stage ranges may be split across calls.
      bool fused2 = false;
      int fused2_rows = 0;
      // (ECGMM_BN_FUSE=0: always the separate reduction pass -- A/B switch; same results up to fp32 summation order)
      // Round 2 fused it where it paid then (serialized trace, profiles/r02_*_v1): on the 56x56 tensors the reduction pass
      // it replaces cost more HBM time (52 us) than the epilogue added (20-30 us); from 28x28 down the epilogue's exposed
      // load latency (one persistent workgroup per CU, 3-4 tiles each) costs MORE than the 10-30 us pass.
      // Round 3: DEFAULT OFF everywhere.  The fused epilogue is ~500 VALU per wave and tile (137 us per layer-1 launch), the
      // stream form of the same input gradient without it takes 82 us (101 with the residual addend) + the 50 us pass:
      // same-call A/B of the whole step, fused on layer 1 / never fused = 6.84, 6.88 / 6.75, 6.75 ms.
      // ECGMM_BN_FUSE_MIN_M (ecgmm_bn_fuse_min_pixels) sets the pixel-count threshold: 400000 = round 2's choice (layer 1 at
      // batch 256), 0 = fuse wherever possible.
      static const bool fuse_on = [] { const char* e = getenv("ECGMM_BN_FUSE"); return !(e && e[0] == '0'); }();
      static const bool fold_on = [] { const char* e = getenv("ECGMM_DOWN_FOLD"); return !(e && e[0] == '0'); }();
      if (g_fuse_min_m < 0) { const char* e = getenv("ECGMM_BN_FUSE_MIN_M"); g_fuse_min_m = e ? atol(e) : FUSE_NEVER; }
      const long fuse_min_m = g_fuse_min_m;
      const bool fuse_here = fuse_on && M >= fuse_min_m;
      if (fuse_here && i + 1 < 8 && !r.blk[i + 1].down) {
        const BlockCfg& kn = r.blk[i + 1];
        const ConvGeom gn = make_geom(N, kn.hin, kn.win, kn.cin, kn.cout, 3, 3, 1, 1, 1);
        fused2 = ecg_conv_halo_ok(dt, 1, gn);
        fused2_rows = fused2 ? ecg_conv_halo_rows(1, gn) : 0;
      }
      // out = relu(bn2(y2) + identity)
      main_wait(s, g_side.done2[0][pp]);  // the wgrad2 of two blocks ago has finished reading this dy buffer
      const void* dzp = q.dz;
      if (fused2) {
        dzp = dcur;
        ECG_TRY(ecg_bn_bwd_tail(dt, dcur, nullptr, b.y2, b.coef2, P(params, k.p_bn2), G(grads, k.p_bn2),
                                G(grads, k.p_bn2 + 1), dyb, q.red2, fused2_rows, M, k.cout, q.bn_scratch, s));
      } else {
        ECG_TRY(ecg_bn_bwd(dt, dcur, b.out, nullptr, nullptr, 1, b.y2, b.coef2, P(params, k.p_bn2), G(grads, k.p_bn2),
                           G(grads, k.p_bn2 + 1), dyb, q.dz, nullptr, M, k.cout, q.bn_scratch, s, relu_bits_on() ? b.bits : nullptr));
      }
      if (G(grads, k.p_conv2)) {
        if (side) side_fork(s);
        ECG_TRY(ecg_conv_wgrad(dt, g2, b.a1, dyb, G(grads, k.p_conv2), 0, q.wg_ws, q.wg_bytes, ws));
        if (side) g_side.done2[0][pp] = side_mark();
      }
      // a1 = relu(bn1(y1)); the mask is recomputed from y1
      ConvEpi ea = {};
      if (fuse_here) { ea.wg_rows = 1; ea.red_y = b.y1; ea.red_mask = b.y1; ea.red_coef = b.coef1; ea.red_rows = q.red1; }
      ECG_TRY(ecg_conv_igemm(dt, 1, g2, dyb, b.w2d, q.da, nullptr, nullptr, nullptr, 0, s, &ea));
      main_wait(s, g_side.done2[1][pp]);
      if (ea.red_done) {
        ECG_TRY(ecg_bn_bwd_tail(dt, q.da, b.y1, b.y1, b.coef1, P(params, k.p_bn1), G(grads, k.p_bn1),
                                G(grads, k.p_bn1 + 1), dy1b, q.red1, ea.red_rows_n, M, k.cout, q.bn_scratch, s));
      } else {
        ECG_TRY(ecg_bn_bwd(dt, q.da, b.y1, nullptr, nullptr, 1, b.y1, b.coef1, P(params, k.p_bn1), G(grads, k.p_bn1),
                           G(grads, k.p_bn1 + 1), dy1b, nullptr, nullptr, M, k.cout, q.bn_scratch, s));
      }
      if (G(grads, k.p_conv1)) {
        if (side) side_fork(s);
        ECG_TRY(ecg_conv_wgrad(dt, g1, in, dy1b, G(grads, k.p_conv1), 0, q.wg_ws, q.wg_bytes, ws));
        if (side) g_side.done2[1][pp] = side_mark();
      }
      if (k.down) {
        ConvGeom gd = make_geom(N, k.hin, k.win, k.cin, k.cout, 1, 1, k.stride, 0, 0);
        main_wait(s, g_side.done2[2][pp]);
        ECG_TRY(ecg_bn_bwd(dt, dzp, nullptr, nullptr, nullptr, 1, b.yd, b.coefd, P(params, k.p_dbn),
                           G(grads, k.p_dbn), G(grads, k.p_dbn + 1), dydb, nullptr, nullptr, M, k.cout, q.bn_scratch,
                           s));
        if (G(grads, k.p_dconv)) {
          if (side) side_fork(s);
          ECG_TRY(ecg_conv_wgrad(dt, gd, in, dydb, G(grads, k.p_dconv), 0, q.wg_ws, q.wg_bytes, ws));
          if (side) g_side.done2[2][pp] = side_mark();
        }
        // the downsample branch's input gradient is one more tap of the stride-2 dgrad's parity class (0,0): one launch,
        // no [N][H][W][Cin] temporary written and re-read as addend
        ConvEpi ed = {};
        ed.src2 = dydb; ed.wpk2 = b.wdd;
        if (!fold_on) ed.src2 = nullptr;
        if (ed.src2) {
          ECG_TRY(ecg_conv_igemm(dt, 1, g1, dy1b, b.w1d, din, nullptr, nullptr, nullptr, 0, s, &ed));
        }
        if (!ed.src2_done) {
          ECG_TRY(ecg_conv_igemm(dt, 1, gd, dydb, b.wdd, q.dtmp, nullptr, nullptr, nullptr, 0, s));
          ECG_TRY(ecg_conv_igemm(dt, 1, g1, dy1b, b.w1d, din, nullptr, q.dtmp, nullptr, 0, s));
        }
      } else {
        // this conv1 dgrad produces the previous block's output gradient: fuse that block's bn2 reduction (see above)
        ConvEpi eb = {};
        if (fuse_here && i > 0 && ecg_conv_halo_ok(dt, 1, g1)) {
          const FwdWs::B& pb = w.b[i - 1];
          eb.wg_rows = 1; eb.red_y = pb.y2; eb.red_mask = pb.out; eb.red_coef = pb.coef2; eb.red_rows = q.red2;
        }
        ECG_TRY(ecg_conv_igemm(dt, 1, g1, dy1b, b.w1d, din, nullptr, dzp, nullptr, 0, s, &eb));
      }
    } else if (st == 9) {
      const void* dp0 = q.X[8 & 1];
      if (stem_recompute(dt)) {
        // BatchNorm-backward reduction over the pooled tensors, then ONE kernel: conv recomputed per tile, max-pool backward
        // gathered, BatchNorm backward applied, weight-gradient products taken -- neither y0 nor dy0 exist.  On the
        // caller's stream like the stem wgrad it replaces (see below); own workspace q.stem_ws.
        ECG_TRY(ecg_stem_pool_bwd(image, w.wstem, w.coef0, P(params, 1), dp0, w.p0, w.idx0, G(grads, 1), G(grads, 2),
                                  G(grads, 0), q.stem_ws, q.stem_bytes, N, 3, r.d.H, r.d.W, s));
        continue;
      }
      if (ecg_stem_fuse_on()) {
        ECG_TRY(ecg_pool_bn_bwd(dt, dp0, w.p0, w.idx0, w.y0, w.coef0, P(params, 1), G(grads, 1), G(grads, 2), q.big1,
                                nullptr, N, r.H1, r.W1, 64, q.bn_scratch, s));
      } else {
        ECG_TRY(ecg_maxpool_relu_bwd(dt, dp0, w.p0, w.idx0, q.big0, N, r.H1, r.W1, 64, s));
        ECG_TRY(ecg_bn_bwd(dt, q.big0, nullptr, nullptr, nullptr, 1, w.y0, w.coef0, P(params, 1), G(grads, 1),
                           G(grads, 2), q.big1, nullptr, nullptr, (long)N * r.H1 * r.W1, 64, q.bn_scratch, s));
      }
      if (G(grads, 0)) {
        // the last kernel of the backward stays on the caller's stream: nothing is left there to overlap it with, it
        // overlaps the side stream's remaining wgrads instead, and the join below then waits for an event that has
        // usually fired already (an exposed cross-stream wait costs 35-140 us here).  Own slab buffer: q.stem_ws.
        ECG_TRY(ecg_stem_wgrad(dt, image, q.big1, G(grads, 0), 0, q.stem_ws, q.stem_bytes, N, 3, r.d.H, r.d.W, 7, s));
      }
    } else {
      ECG_FAIL(ECGMM_ERR_SHAPE, "resnet18 bwd: stage %d out of range", st);
    }
  }
  if (stage_end == 10) ecg_tl_mark(210, s);
  if (side && !g_side.defer_join) {  // join: everything the side stream did is ordered before whatever the caller enqueues next
    hipEvent_t e = side_next_ev();
    (void)hipEventRecord(e, g_side.s);
    (void)hipStreamWaitEvent(s, e, 0);
  }
  if (stage_end == 10) ecg_tl_mark(211, s);
  return 0;
}

// The image encoder's stem by recompute (conv_stem_fused.hip): 1 = on (bf16 only), 0 = the two-pass route with the
// full-resolution conv output in memory (default: faster, see above).  Start-up value: ECGMM_STEM_RECOMPUTE.  Changes the workspace layouts: set it
// between steps, never between a forward and its backward.
extern "C" int ecgmm_stem_recompute(int on) {
  g_stem_recompute = on != 0;
  return 0;
}

extern "C" int ecgmm_bn_fuse_min_pixels(int64_t m) {
  g_fuse_min_m = m < 0 ? FUSE_NEVER : (long)m;
  return 0;
}
