// On-device ECG signal pre-processing: the reference runs, per sample on CPU workers,
//   StandardScaler (per time column) -> 200-tap moving-average baseline removal (np.convolve 'same')
//   -> 5th-order Butterworth low-pass applied forward-backward (scipy.signal.filtfilt, odd padding of
//   3*(order+1) samples, steady-state initial conditions)          dataset.py:66-71,81-95;
//   train_signal_12_af.py:19-34 (same functions over [leads, time]).
// One lane = one signal (the IIR recurrence is sequential in time); intermediates live transposed
// ([time][signal]) in a caller-owned fp64 workspace so every step of the recurrence is a coalesced
// access across the wave.  All arithmetic in fp64 (the reference computes in float64), output fp32
// (the reference casts with torch.tensor(..., dtype=torch.float)).
#include "ops.h"

namespace {

struct PreParams {
  const float* x;          // [S][L]
  float* out;              // [S][L]
  const float* sc_mean;    // [L] or null
  const float* sc_scale;   // [L] or null
  double* ws;              // [L + E][S]
  int S, L, window, order;
  double b[8], a[8], zi[8];
};

__global__ __launch_bounds__(64) void signal_preprocess_kernel(PreParams p) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= p.S) return;
  const int L = p.L, S = p.S, M = p.window, N = p.order;
  const int PAD = 3 * (N + 1), E = L + 2 * PAD;
  double* u = p.ws + s;                    // u[t * S]: scaled, baseline-removed signal
  double* f = p.ws + (size_t)L * S + s;    // f[k * S]: forward-filtered padded signal
  const float* xr = p.x + (size_t)s * L;
  auto xs = [&](int t) -> double {
    double v = (double)xr[t];
    if (p.sc_mean) v = (v - (double)p.sc_mean[t]) / (double)p.sc_scale[t];
    return v;
  };
  // ---- moving-average baseline: np.convolve(x, ones(M)/M, 'same') = mean over [i - bk, i + fw]
  const int fw = (M - 1) / 2, bk = M - 1 - fw;
  double run = 0.0;
  for (int t = 0; t <= fw && t < L; ++t) run += xs(t);
  for (int i = 0; i < L; ++i) {
    u[(size_t)i * S] = xs(i) - run / (double)M;
    const int add = i + fw + 1, sub = i - bk;
    if (add < L) run += xs(add);
    if (sub >= 0) run -= xs(sub);
  }
  // ---- forward pass over the odd-extended signal (direct form II transposed, as scipy's lfilter)
  auto ext = [&](int k) -> double {
    if (k < PAD) return 2.0 * u[0] - u[(size_t)(PAD - k) * S];
    if (k < PAD + L) return u[(size_t)(k - PAD) * S];
    return 2.0 * u[(size_t)(L - 1) * S] - u[(size_t)(L - 2 - (k - PAD - L)) * S];
  };
  double z[8];
  const double e0 = ext(0);
  for (int i = 0; i < N; ++i) z[i] = p.zi[i] * e0;
  for (int k = 0; k < E; ++k) {
    const double e = ext(k);
    const double y = p.b[0] * e + z[0];
    for (int i = 0; i < N - 1; ++i) z[i] = p.b[i + 1] * e + z[i + 1] - p.a[i + 1] * y;
    z[N - 1] = p.b[N] * e - p.a[N] * y;
    f[(size_t)k * S] = y;
  }
  // ---- backward pass; keep the un-padded part
  const double r0 = f[(size_t)(E - 1) * S];
  for (int i = 0; i < N; ++i) z[i] = p.zi[i] * r0;
  float* orow = p.out + (size_t)s * L;
  for (int k = 0; k < E; ++k) {
    const double e = f[(size_t)(E - 1 - k) * S];
    const double y = p.b[0] * e + z[0];
    for (int i = 0; i < N - 1; ++i) z[i] = p.b[i + 1] * e + z[i + 1] - p.a[i + 1] * y;
    z[N - 1] = p.b[N] * e - p.a[N] * y;
    const int t = E - 1 - k - PAD;
    if (t >= 0 && t < L) orow[t] = (float)y;
  }
}

}  // namespace

extern "C" size_t ecgmm_signal_preprocess_workspace(int S, int L, int order) {
  return ((size_t)L + (size_t)L + 6 * (order + 1)) * (size_t)S * sizeof(double);
}

extern "C" int ecgmm_signal_preprocess(const float* x, float* out, int S, int L, const float* sc_mean,
                                       const float* sc_scale, int window, const double* b, const double* a,
                                       const double* zi, int order, void* ws, size_t ws_bytes, void* stream) {
  if (order < 1 || order > 7) ECG_FAIL(ECGMM_ERR_SHAPE, "signal_preprocess: filter order %d outside 1..7", order);
  if (S < 1 || L < window || L <= 3 * (order + 1))
    ECG_FAIL(ECGMM_ERR_SHAPE, "signal_preprocess: need L >= window and L > 3*(order+1) (S=%d L=%d window=%d)", S, L,
             window);
  if ((sc_mean == nullptr) != (sc_scale == nullptr)) ECG_FAIL(ECGMM_ERR_SHAPE, "signal_preprocess: scaler needs mean AND scale");
  if (!ws || ws_bytes < ecgmm_signal_preprocess_workspace(S, L, order))
    ECG_FAIL(ECGMM_ERR_WORKSPACE, "signal_preprocess: workspace too small");
  if (a[0] != 1.0) ECG_FAIL(ECGMM_ERR_SHAPE, "signal_preprocess: a[0] must be 1 (normalised transfer function)");
  PreParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.out = out; p.sc_mean = sc_mean; p.sc_scale = sc_scale; p.ws = (double*)ws;
  p.S = S; p.L = L; p.window = window; p.order = order;
  for (int i = 0; i <= order; ++i) { p.b[i] = b[i]; p.a[i] = a[i]; }
  for (int i = 0; i < order; ++i) p.zi[i] = zi[i];
  hipLaunchKernelGGL(signal_preprocess_kernel, dim3(ceil_div(S, 64)), dim3(64), 0, (hipStream_t)stream, p);
  ECG_CHECK_LAUNCH("signal_preprocess");
  return 0;
}
