// On-device ECG signal pre-processing: the reference runs, per sample on CPU workers,
//   StandardScaler (per time column) -> 200-tap moving-average baseline removal (np.convolve 'same')
//   -> 5th-order Butterworth low-pass applied forward-backward (scipy.signal.filtfilt, odd padding of
//   3*(order+1) samples, steady-state initial conditions)          dataset.py:66-71,81-95;
//   train_signal_12_af.py:19-34 (same functions over [leads, time]).
// Two kernels: the LDS-resident chunk-parallel one below (one wave per signal, used whenever the padded
// record fits a CU's 160 KiB LDS -- every ECG length the reference uses), and a one-lane-per-signal
// kernel whose intermediates live transposed ([time][signal]) in a caller-owned fp64 workspace, for
// longer records.  All arithmetic in fp64 (the reference computes in float64), output fp32
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

// ---- LDS-resident variant: one wave per signal, the whole padded signal in LDS (fp64) ----
// The IIR recurrence z' = A z + B x is linear, so the time axis is cut into 64 chunks of C samples:
//   (1) every lane filters its chunk from a ZERO state and keeps only the final state s_t;
//   (2) lane 0 chains the true chunk-entry states  z_{t+1} = A^C z_t + s_t  (A^C built once by lanes
//       0..N-1 running C zero-input steps from the unit states);
//   (3) every lane re-filters its chunk from its true entry state, writing y in place.
// C is forced odd so the 64 lanes' chunk cursors (stride C doubles) fall in distinct LDS banks.
template <int N>
__global__ __launch_bounds__(64) void signal_preprocess_lds_kernel(PreParams p) {
  extern __shared__ double sm[];
  constexpr int T = 64, PAD = 3 * (N + 1);
  const int L = p.L, M = p.window, E = L + 2 * PAD;
  const int lane = threadIdx.x, s = blockIdx.x;
  double* A = sm;            // [L]     scaled input
  double* B = sm + L;        // [E]     odd-extended, baseline-removed signal; filtered in place
  double* st = B + E;        // [T][8]  chunk states
  double* Mx = st + T * 8;   // [8][8]  A^C, row-major
  double b[N + 1], a[N + 1];
#pragma unroll
  for (int i = 0; i <= N; ++i) { b[i] = p.b[i]; a[i] = p.a[i]; }

  const float* xr = p.x + (size_t)s * L;
  for (int t = lane; t < L; t += T) {
    double v = (double)xr[t];
    if (p.sc_mean) v = (v - (double)p.sc_mean[t]) / (double)p.sc_scale[t];
    A[t] = v;
  }
  const int C = ((E + T - 1) / T) | 1;
  if (lane < N) {  // column `lane` of A^C
    double z[N];
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = i == lane ? 1.0 : 0.0;
    for (int k = 0; k < C; ++k) {
      const double y = z[0];
#pragma unroll
      for (int i = 0; i < N - 1; ++i) z[i] = z[i + 1] - a[i + 1] * y;
      z[N - 1] = -a[N] * y;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) Mx[i * 8 + lane] = z[i];
  }
  __syncthreads();
  {  // moving-average baseline, np.convolve(x, ones(M)/M, 'same'): mean over [i - bk, i + fw], zero outside
    const int fw = (M - 1) / 2, bk = M - 1 - fw;
    const int Cm = ((L + T - 1) / T) | 1;
    const int i0 = lane * Cm, i1 = min(L, i0 + Cm);
    if (i0 < L) {
      double run = 0.0;
      for (int j = max(0, i0 - bk); j <= min(L - 1, i0 + fw); ++j) run += A[j];
      for (int i = i0; i < i1; ++i) {
        B[PAD + i] = A[i] - run / (double)M;
        const int add = i + fw + 1, sub = i - bk;
        if (add < L) run += A[add];
        if (sub >= 0) run -= A[sub];
      }
    }
  }
  __syncthreads();
  if (lane < PAD) {  // odd extension about both ends
    B[lane] = 2.0 * B[PAD] - B[PAD + (PAD - lane)];
    B[PAD + L + lane] = 2.0 * B[PAD + L - 1] - B[PAD + L - 2 - lane];
  }
  __syncthreads();

  const int k0 = lane * C, k1 = min(E, k0 + C);
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    auto at = [&](int k) -> double& { return B[pass ? E - 1 - k : k]; };
    double z[N];
    if (k1 - k0 == C) {  // a full chunk hands a state on
#pragma unroll
      for (int i = 0; i < N; ++i) z[i] = 0.0;
      for (int k = k0; k < k1; ++k) {
        const double e = at(k);
        const double y = b[0] * e + z[0];
#pragma unroll
        for (int i = 0; i < N - 1; ++i) z[i] = b[i + 1] * e + z[i + 1] - a[i + 1] * y;
        z[N - 1] = b[N] * e - a[N] * y;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) st[lane * 8 + i] = z[i];
    }
    __syncthreads();
    if (lane == 0) {
      const double e0 = at(0);
      double m[N][N];
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) m[i][j] = Mx[i * 8 + j];
#pragma unroll
      for (int i = 0; i < N; ++i) z[i] = p.zi[i] * e0;
      const int nchunks = (E + C - 1) / C;
      for (int t = 0; t < nchunks; ++t) {
        double sv[N], zn[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { sv[i] = st[t * 8 + i]; st[t * 8 + i] = z[i]; }
#pragma unroll
        for (int i = 0; i < N; ++i) {
          double acc = sv[i];
#pragma unroll
          for (int j = 0; j < N; ++j) acc += m[i][j] * z[j];
          zn[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) z[i] = zn[i];
      }
    }
    __syncthreads();
    if (k0 < E) {
#pragma unroll
      for (int i = 0; i < N; ++i) z[i] = st[lane * 8 + i];
      for (int k = k0; k < k1; ++k) {
        double& r = at(k);
        const double e = r;
        const double y = b[0] * e + z[0];
#pragma unroll
        for (int i = 0; i < N - 1; ++i) z[i] = b[i + 1] * e + z[i + 1] - a[i + 1] * y;
        z[N - 1] = b[N] * e - a[N] * y;
        r = y;
      }
    }
    __syncthreads();
  }
  float* orow = p.out + (size_t)s * L;
  for (int t = lane; t < L; t += T) orow[t] = (float)B[PAD + t];
}

template <int N>
int launch_lds(const PreParams& p, size_t lds, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)signal_preprocess_lds_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  hipLaunchKernelGGL(signal_preprocess_lds_kernel<N>, dim3(p.S), dim3(64), lds, st, p);
  return 0;
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
  // LDS-resident chunk-parallel kernel whenever one padded signal (+ the scaled copy) fits one CU's LDS;
  // longer records take the one-lane-per-signal kernel over the global workspace.
  const size_t lds = ((size_t)L + L + 6 * (order + 1) + 64 * 8 + 64) * sizeof(double);
  if (lds <= 160 * 1024) {
    hipStream_t st = (hipStream_t)stream;
    switch (order) {
      case 1: launch_lds<1>(p, lds, st); break;
      case 2: launch_lds<2>(p, lds, st); break;
      case 3: launch_lds<3>(p, lds, st); break;
      case 4: launch_lds<4>(p, lds, st); break;
      case 5: launch_lds<5>(p, lds, st); break;
      case 6: launch_lds<6>(p, lds, st); break;
      default: launch_lds<7>(p, lds, st); break;
    }
  } else {
    hipLaunchKernelGGL(signal_preprocess_kernel, dim3(ceil_div(S, 64)), dim3(64), 0, (hipStream_t)stream, p);
  }
  ECG_CHECK_LAUNCH("signal_preprocess");
  return 0;
}
