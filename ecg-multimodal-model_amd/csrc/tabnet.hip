// Small fp32 row kernels for the TabNet clinical encoder (reference: multimodal.py:109-148 wraps
// pytorch_tabnet.tab_network.TabNetNoEmbeddings).  Everything here is a few KB per launch:
// GLU gate, sparsemax over <= 64 features (one lane per row, registers only), the mask / prior elementwise
// steps and the mask-entropy term.  Linear layers and (ghost) BatchNorm reuse the library's existing kernels.
#include "ops.h"

namespace {

constexpr int TB = 256;
constexpr int SPMAX_D = 64;

__global__ void glu_fwd_kernel(const float* __restrict__ z, float* __restrict__ out, long N, int D) {
  const long total = N * D;
  for (long i = (long)blockIdx.x * TB + threadIdx.x; i < total; i += (long)gridDim.x * TB) {
    const long n = i / D;
    const int d = (int)(i - n * D);
    const float a = z[n * 2 * D + d], b = z[n * 2 * D + D + d];
    out[i] = a * (1.f / (1.f + expf(-b)));
  }
}
__global__ void glu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dout, float* __restrict__ dz,
                               long N, int D) {
  const long total = N * D;
  for (long i = (long)blockIdx.x * TB + threadIdx.x; i < total; i += (long)gridDim.x * TB) {
    const long n = i / D;
    const int d = (int)(i - n * D);
    const float a = z[n * 2 * D + d], b = z[n * 2 * D + D + d];
    const float s = 1.f / (1.f + expf(-b)), g = dout[i];
    dz[n * 2 * D + d] = g * s;
    dz[n * 2 * D + D + d] = g * a * s * (1.f - s);
  }
}

// sparsemax (Martins & Astudillo 2016) along the row: p = max(x - tau, 0) with tau from the sorted prefix sums
__global__ void sparsemax_fwd_kernel(const float* __restrict__ x, float* __restrict__ p, long N, int D) {
  const long n = (long)blockIdx.x * TB + threadIdx.x;
  if (n >= N) return;
  float v[SPMAX_D];
  float mx = -INFINITY;
  for (int d = 0; d < D; ++d) {
    v[d] = x[n * D + d];
    mx = fmaxf(mx, v[d]);
  }
  for (int d = 0; d < D; ++d) v[d] -= mx;  // (translation invariant; as pytorch_tabnet's implementation does)
  for (int i = 1; i < D; ++i) {            // insertion sort, descending
    const float key = v[i];
    int j = i - 1;
    while (j >= 0 && v[j] < key) {
      v[j + 1] = v[j];
      --j;
    }
    v[j + 1] = key;
  }
  float cum = 0.f, tau = 0.f;
  for (int k = 1; k <= D; ++k) {
    cum += v[k - 1];
    if (1.f + k * v[k - 1] > cum) tau = (cum - 1.f) / k;  // support condition holds for a prefix of the sorted row
  }
  for (int d = 0; d < D; ++d) p[n * D + d] = fmaxf(x[n * D + d] - mx - tau, 0.f);
}
__global__ void sparsemax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, float* __restrict__ dx,
                                     long N, int D) {
  const long n = (long)blockIdx.x * TB + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  int k = 0;
  for (int d = 0; d < D; ++d)
    if (p[n * D + d] > 0.f) {
      s += dp[n * D + d];
      ++k;
    }
  const float vhat = k > 0 ? s / k : 0.f;
  for (int d = 0; d < D; ++d) dx[n * D + d] = p[n * D + d] > 0.f ? dp[n * D + d] - vhat : 0.f;
}

__global__ void ew_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                          long n, float s) {
  for (long i = (long)blockIdx.x * TB + threadIdx.x; i < n; i += (long)gridDim.x * TB) {
    float r;
    switch (op) {
      case ECGMM_EW_MUL: r = a[i] * b[i]; break;
      case ECGMM_EW_ADD_SCALE: r = (a[i] + b[i]) * s; break;
      case ECGMM_EW_PRIOR: r = b[i] * (s - a[i]); break;      // prior' = prior * (gamma - M): a = M, b = prior
      case ECGMM_EW_RELU: r = fmaxf(a[i], 0.f); break;
      case ECGMM_EW_RELU_BWD: r = a[i] > 0.f ? b[i] : 0.f; break;  // a = y, b = dy
      case ECGMM_EW_SCALE: r = a[i] * s; break;
      case ECGMM_EW_NEG_MUL: r = -a[i] * b[i]; break;
      case ECGMM_EW_ADD: r = a[i] + b[i]; break;
      case ECGMM_EW_RSUB: r = s - a[i]; break;
      default: r = 0.f;
    }
    out[i] = r;
  }
}

// out[0] = (1/N) * sum_n sum_d M log(M + eps)   (one block, fixed order -> reproducible)
__global__ __launch_bounds__(1024) void entropy_fwd_kernel(const float* __restrict__ M, float* __restrict__ out, long N,
                                                           int D, float eps) {
  __shared__ double sh[1024];
  double s = 0.0;
  const long total = N * D;
  for (long i = threadIdx.x; i < total; i += 1024) s += (double)(M[i] * logf(M[i] + eps));
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(sh[0] / (double)N);
}
__global__ void entropy_bwd_kernel(const float* __restrict__ M, const float* __restrict__ g, float* __restrict__ dM,
                                   long N, int D, float eps) {
  const long total = N * D;
  const float gs = g[0] / (float)N;
  for (long i = (long)blockIdx.x * TB + threadIdx.x; i < total; i += (long)gridDim.x * TB)
    dM[i] = gs * (logf(M[i] + eps) + M[i] / (M[i] + eps));
}

// x [N, D] -> d = x[:, :nd] (optionally through ReLU), a = x[:, nd:]; and the gradient merge
__global__ void split_kernel(const float* __restrict__ x, float* __restrict__ d, float* __restrict__ a, long N, int D,
                             int nd, int relu) {
  const long total = N * D;
  for (long i = (long)blockIdx.x * TB + threadIdx.x; i < total; i += (long)gridDim.x * TB) {
    const long n = i / D;
    const int j = (int)(i - n * D);
    const float v = x[i];
    if (j < nd) d[n * nd + j] = relu ? fmaxf(v, 0.f) : v;
    else a[n * (D - nd) + (j - nd)] = v;
  }
}
__global__ void split_bwd_kernel(const float* __restrict__ d, const float* __restrict__ gd, const float* __restrict__ ga,
                                 float* __restrict__ gx, long N, int D, int nd, int relu) {
  const long total = N * D;
  for (long i = (long)blockIdx.x * TB + threadIdx.x; i < total; i += (long)gridDim.x * TB) {
    const long n = i / D;
    const int j = (int)(i - n * D);
    float g;
    if (j < nd) g = (gd && !(relu && d[n * nd + j] <= 0.f)) ? gd[n * nd + j] : 0.f;
    else g = ga ? ga[n * (D - nd) + (j - nd)] : 0.f;
    gx[i] = g;
  }
}

// BatchNorm over the rows of a small [N, C] fp32 matrix, any C (TabNet normalises 2-, 64- and 128-wide rows over
// "virtual batches" of <= 128 rows): one block per channel, fp64 sums, one launch forward and one backward.
// save[0][c] = mean, save[1][c] = invstd.  Training updates the running statistics (unbiased variance) like torch.
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* rm, float* rv,
                                                           long long* nbt, float* __restrict__ y, float* __restrict__ save,
                                                           int N, int C, int training, float momentum, float eps) {
  __shared__ double sh[2][256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double mean, var;
  if (training) {
    double s1 = 0.0, s2 = 0.0;
    for (int n = tid; n < N; n += 256) {
      const double v = (double)x[(size_t)n * C + c];
      s1 += v;
      s2 += v * v;
    }
    sh[0][tid] = s1;
    sh[1][tid] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        sh[0][tid] += sh[0][tid + o];
        sh[1][tid] += sh[1][tid + o];
      }
      __syncthreads();
    }
    mean = sh[0][0] / N;
    var = sh[1][0] / N - mean * mean;
    if (var < 0.0) var = 0.0;
    if (tid == 0 && rm) {
      const double unb = N > 1 ? var * N / (N - 1.0) : var;
      rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
      rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
      if (c == 0 && nbt) *nbt += 1;
    }
  } else {
    mean = (double)rm[c];
    var = (double)rv[c];
  }
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f, m = (float)mean;
  if (tid == 0) {
    save[c] = m;
    save[C + c] = invstd;
  }
  for (int n = tid; n < N; n += 256) y[(size_t)n * C + c] = (x[(size_t)n * C + c] - m) * invstd * g + b;
}
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ gamma, const float* __restrict__ save,
                                                           float* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int N, int C, int accumulate) {
  __shared__ double sh[2][256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const float m = save[c], invstd = save[C + c];
  double s1 = 0.0, s2 = 0.0;
  for (int n = tid; n < N; n += 256) {
    const double g = (double)dy[(size_t)n * C + c];
    s1 += g;
    s2 += g * (double)((x[(size_t)n * C + c] - m) * invstd);
  }
  sh[0][tid] = s1;
  sh[1][tid] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      sh[0][tid] += sh[0][tid + o];
      sh[1][tid] += sh[1][tid + o];
    }
    __syncthreads();
  }
  const float sum_dy = (float)sh[0][0], sum_dyx = (float)sh[1][0];
  if (tid == 0) {
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + sum_dyx : sum_dyx;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + sum_dy : sum_dy;
  }
  const float k = (gamma ? gamma[c] : 1.f) * invstd, mdy = sum_dy / N, mdyx = sum_dyx / N;
  for (int n = tid; n < N; n += 256) {
    const float xh = (x[(size_t)n * C + c] - m) * invstd;
    dx[(size_t)n * C + c] = k * (dy[(size_t)n * C + c] - mdy - xh * mdyx);
  }
}

inline int grid_for(long n) {
  long b = (n + TB - 1) / TB;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int ecgmm_glu_fwd(const float* z, float* out, int64_t N, int D, void* stream) {
  if (N < 1 || D < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "glu_fwd: empty input");
  hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_for(N * D)), dim3(TB), 0, (hipStream_t)stream, z, out, (long)N, D);
  ECG_CHECK_LAUNCH("glu_fwd");
  return 0;
}
extern "C" int ecgmm_glu_bwd(const float* z, const float* dout, float* dz, int64_t N, int D, void* stream) {
  if (N < 1 || D < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "glu_bwd: empty input");
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for(N * D)), dim3(TB), 0, (hipStream_t)stream, z, dout, dz, (long)N, D);
  ECG_CHECK_LAUNCH("glu_bwd");
  return 0;
}
extern "C" int ecgmm_sparsemax_fwd(const float* x, float* p, int64_t N, int D, void* stream) {
  if (N < 1 || D < 1 || D > SPMAX_D) ECG_FAIL(ECGMM_ERR_SHAPE, "sparsemax: D=%d outside 1..%d", D, SPMAX_D);
  hipLaunchKernelGGL(sparsemax_fwd_kernel, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, (hipStream_t)stream, x, p,
                     (long)N, D);
  ECG_CHECK_LAUNCH("sparsemax_fwd");
  return 0;
}
extern "C" int ecgmm_sparsemax_bwd(const float* p, const float* dp, float* dx, int64_t N, int D, void* stream) {
  if (N < 1 || D < 1 || D > SPMAX_D) ECG_FAIL(ECGMM_ERR_SHAPE, "sparsemax: D=%d outside 1..%d", D, SPMAX_D);
  hipLaunchKernelGGL(sparsemax_bwd_kernel, dim3((unsigned)((N + TB - 1) / TB)), dim3(TB), 0, (hipStream_t)stream, p, dp,
                     dx, (long)N, D);
  ECG_CHECK_LAUNCH("sparsemax_bwd");
  return 0;
}
extern "C" int ecgmm_ew(int op, const float* a, const float* b, float* out, int64_t n, float s, void* stream) {
  if (n < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "ew: empty input");
  if (op < 0 || op > ECGMM_EW_RSUB) ECG_FAIL(ECGMM_ERR_SHAPE, "ew: unknown op %d", op);
  hipLaunchKernelGGL(ew_kernel, dim3(grid_for(n)), dim3(TB), 0, (hipStream_t)stream, op, a, b, out, (long)n, s);
  ECG_CHECK_LAUNCH("ew");
  return 0;
}
extern "C" int ecgmm_entropy_fwd(const float* M, float* out, int64_t N, int D, float eps, void* stream) {
  if (N < 1 || D < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "entropy: empty input");
  hipLaunchKernelGGL(entropy_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, M, out, (long)N, D, eps);
  ECG_CHECK_LAUNCH("entropy_fwd");
  return 0;
}
extern "C" int ecgmm_entropy_bwd(const float* M, const float* g, float* dM, int64_t N, int D, float eps, void* stream) {
  if (N < 1 || D < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "entropy: empty input");
  hipLaunchKernelGGL(entropy_bwd_kernel, dim3(grid_for(N * D)), dim3(TB), 0, (hipStream_t)stream, M, g, dM, (long)N, D,
                     eps);
  ECG_CHECK_LAUNCH("entropy_bwd");
  return 0;
}
extern "C" int ecgmm_split_cols(const float* x, float* d, float* a, int64_t N, int D, int nd, int relu, void* stream) {
  if (N < 1 || nd < 1 || nd >= D) ECG_FAIL(ECGMM_ERR_SHAPE, "split_cols: need 0 < nd < D (nd=%d D=%d)", nd, D);
  hipLaunchKernelGGL(split_kernel, dim3(grid_for(N * D)), dim3(TB), 0, (hipStream_t)stream, x, d, a, (long)N, D, nd, relu);
  ECG_CHECK_LAUNCH("split_cols");
  return 0;
}
extern "C" int ecgmm_split_cols_bwd(const float* d, const float* gd, const float* ga, float* gx, int64_t N, int D, int nd,
                                    int relu, void* stream) {
  if (N < 1 || nd < 1 || nd >= D) ECG_FAIL(ECGMM_ERR_SHAPE, "split_cols_bwd: need 0 < nd < D (nd=%d D=%d)", nd, D);
  hipLaunchKernelGGL(split_bwd_kernel, dim3(grid_for(N * D)), dim3(TB), 0, (hipStream_t)stream, d, gd, ga, gx, (long)N, D,
                     nd, relu);
  ECG_CHECK_LAUNCH("split_cols_bwd");
  return 0;
}
extern "C" int ecgmm_bn_small_fwd(const float* x, const float* gamma, const float* beta, float* rm, float* rv,
                                  long long* nbt, float* y, float* save, int N, int C, int training, float momentum,
                                  float eps, void* stream) {
  if (N < 1 || C < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "bn_small: empty input");
  if (!training && (!rm || !rv)) ECG_FAIL(ECGMM_ERR_SHAPE, "bn_small: eval mode needs running statistics");
  hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, rm, rv, nbt, y, save,
                     N, C, training, momentum, eps);
  ECG_CHECK_LAUNCH("bn_small_fwd");
  return 0;
}
extern "C" int ecgmm_bn_small_bwd(const float* x, const float* dy, const float* gamma, const float* save, float* dx,
                                  float* dgamma, float* dbeta, int N, int C, int accumulate, void* stream) {
  if (N < 1 || C < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "bn_small: empty input");
  hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, dy, gamma, save, dx, dgamma, dbeta,
                     N, C, accumulate);
  ECG_CHECK_LAUNCH("bn_small_bwd");
  return 0;
}
