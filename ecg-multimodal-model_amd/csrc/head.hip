// Small fp32 kernels of the fusion head and the dense tails (gfx950): generic Linear fwd/bwd for
// ragged shapes (VALU), activation backward, LayerNorm fwd/bwd (wave per row), attention fusion
// (softmax(3) scale + concat + LayerNorm in one row kernel), variance regulariser, cross-entropy /
// focal loss, dropout (Philox4x32-10) and the fused Adam step.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Linear (VALU path): y[b][o] = act(sum_i x[b][i] w[o][i] + bias[o])
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ECGMM_ACT_RELU) return fmaxf(v, 0.f);
  if (act == ECGMM_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}

__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B,
                                                         int In, int Out, int act) {
  // one wave per output element: coalesced reads of x[b][:] and w[o][:], butterfly reduce
  const int lane = threadIdx.x & 63;
  long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  long nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long e = wid; e < (long)B * Out; e += nw) {
    int b = (int)(e / Out), o = (int)(e % Out);
    float s = 0.f;
    for (int i = lane; i < In; i += 64) s += x[(size_t)b * In + i] * w[(size_t)o * In + i];
    s = wave_sum(s);
    if (lane == 0) y[e] = apply_act(s + (bias ? bias[o] : 0.f), act);
  }
}

// dx[b][i] = sum_o dy[b][o] w[o][i]
__global__ void linear_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                    int B, int In, int Out, int accumulate) {
  long total = (long)B * In;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int b = (int)(e / In), i = (int)(e % In);
    float s = 0.f;
    for (int o = 0; o < Out; ++o) s += dy[(size_t)b * Out + o] * w[(size_t)o * In + i];
    dx[e] = accumulate ? dx[e] + s : s;
  }
}

// dw[o][i] = sum_b dy[b][o] x[b][i];  db[o] = sum_b dy[b][o]   (one wave per output element; fixed order)
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dw, float* __restrict__ db, int B, int In,
                                                           int Out, int accumulate) {
  const int lane = threadIdx.x & 63;
  const long total = (long)Out * In;
  long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long e = wid; e < total + Out; e += nw) {
    float s = 0.f;
    if (e < total) {
      const int o = (int)(e / In), i = (int)(e % In);
      for (int b = lane; b < B; b += 64) s += dy[(size_t)b * Out + o] * x[(size_t)b * In + i];
      s = wave_sum(s);
      if (lane == 0) dw[e] = accumulate ? dw[e] + s : s;
    } else if (db) {
      const int o = (int)(e - total);
      for (int b = lane; b < B; b += 64) s += dy[(size_t)b * Out + o];
      s = wave_sum(s);
      if (lane == 0) db[o] = accumulate ? db[o] + s : s;
    }
  }
}

// dz = dy * act'(y) written from the activation OUTPUT y
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz,
                               long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float g = dy[i], v = y[i];
    if (act == ECGMM_ACT_RELU) g = v > 0.f ? g : 0.f;
    else if (act == ECGMM_ACT_SIGMOID) g = g * v * (1.f - v);
    dz[i] = g;
  }
}

__global__ void axpby_kernel(float a, const float* __restrict__ x, float b, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (+ optional attention-fusion scale/concat in front of it).  One wave per row.
// Inputs: up to 3 segments src[m][B][dims[m]] each scaled by sw[m] (sw == null -> one segment, scale 1)
// ------------------------------------------------------------------------------------------------
constexpr int LN_MAXD = 1024;  // 16 values per lane

struct LnParams {
  const float* seg[3];
  int dims[3];
  int nseg;
  const float* fusion_w;  // raw attention weights [3] (softmax applied in-kernel) or null
  const float* gamma;
  const float* beta;
  float* out;             // [B][D]
  float* stat;            // [B][2] mean, rstd
  float* soft_w;          // [3] (written by block 0) or null
  int B, D;
  float eps;
};

__device__ __forceinline__ void softmax3(const float* w, float* o) {
  float m = fmaxf(w[0], fmaxf(w[1], w[2]));
  float e0 = expf(w[0] - m), e1 = expf(w[1] - m), e2 = expf(w[2] - m);
  float inv = 1.f / (e0 + e1 + e2);
  o[0] = e0 * inv; o[1] = e1 * inv; o[2] = e2 * inv;
}

__device__ __forceinline__ float seg_load(const LnParams& p, const float* sw, int row, int col) {
  int m = 0, c = col;
  if (p.nseg > 1) {
    if (c >= p.dims[0]) { c -= p.dims[0]; m = 1; }
    if (m == 1 && c >= p.dims[1]) { c -= p.dims[1]; m = 2; }
  }
  float v = p.seg[m][(size_t)row * p.dims[m] + c];
  return sw ? v * sw[m] : v;
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(LnParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  float swv[3];
  const float* sw = nullptr;
  if (p.fusion_w) {
    softmax3(p.fusion_w, swv);
    sw = swv;
    if (p.soft_w && blockIdx.x == 0 && threadIdx.x < 3) p.soft_w[threadIdx.x] = swv[threadIdx.x];
  }
  if (row >= p.B) return;
  float v[LN_MAXD / 64];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXD / 64; ++k) {
    int c = lane + 64 * k;
    v[k] = c < p.D ? seg_load(p, sw, row, c) : 0.f;
    s += v[k];
  }
  float mean = wave_sum(s) / (float)p.D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXD / 64; ++k) {
    int c = lane + 64 * k;
    float d = c < p.D ? v[k] - mean : 0.f;
    q += d * d;
  }
  float rstd = rsqrtf(wave_sum(q) / (float)p.D + p.eps);
#pragma unroll
  for (int k = 0; k < LN_MAXD / 64; ++k) {
    int c = lane + 64 * k;
    if (c < p.D) p.out[(size_t)row * p.D + c] = (v[k] - mean) * rstd * p.gamma[c] + p.beta[c];
  }
  if (lane == 0) {
    p.stat[2 * row] = mean;
    p.stat[2 * row + 1] = rstd;
  }
}

struct LnBwdParams {
  LnParams f;            // forward description (seg, dims, fusion_w, gamma, stat)
  const float* dout;     // [B][D]
  float* dseg[3];        // grads wrt the (unscaled) segments; nullable entries
  int dseg_accumulate;
  float* partial;        // [grid][2*D + 4]: dgamma, dbeta, dwraw[3] (+pad)
};

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(LnBwdParams q) {
  const LnParams& p = q.f;
  __shared__ float sh[4][2 * LN_MAXD + 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float swv[3];
  const float* sw = nullptr;
  if (p.fusion_w) {
    softmax3(p.fusion_w, swv);
    sw = swv;
  }
  float ag[LN_MAXD / 64], ab[LN_MAXD / 64], aw[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < LN_MAXD / 64; ++k) ag[k] = ab[k] = 0.f;
  for (int row = blockIdx.x * 4 + wv; row < p.B; row += gridDim.x * 4) {
    float mean = p.stat[2 * row], rstd = p.stat[2 * row + 1];
    float xh[LN_MAXD / 64], dg[LN_MAXD / 64];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXD / 64; ++k) {
      int c = lane + 64 * k;
      if (c < p.D) {
        xh[k] = (seg_load(p, sw, row, c) - mean) * rstd;
        float d = q.dout[(size_t)row * p.D + c];
        ag[k] += d * xh[k];
        ab[k] += d;
        dg[k] = d * p.gamma[c];
        s1 += dg[k];
        s2 += dg[k] * xh[k];
      } else {
        xh[k] = dg[k] = 0.f;
      }
    }
    s1 = wave_sum(s1) / (float)p.D;
    s2 = wave_sum(s2) / (float)p.D;
#pragma unroll
    for (int k = 0; k < LN_MAXD / 64; ++k) {
      int c = lane + 64 * k;
      if (c >= p.D) continue;
      float dx = rstd * (dg[k] - s1 - xh[k] * s2);  // grad wrt the (scaled) concat element
      int m = 0, cc = c;
      if (p.nseg > 1) {
        if (cc >= p.dims[0]) { cc -= p.dims[0]; m = 1; }
        if (m == 1 && cc >= p.dims[1]) { cc -= p.dims[1]; m = 2; }
      }
      if (sw) {
        float raw = p.seg[m][(size_t)row * p.dims[m] + cc];
        aw[m] += dx * raw;
        dx *= sw[m];
      }
      if (q.dseg[m]) {
        float* o = q.dseg[m] + (size_t)row * p.dims[m] + cc;
        *o = q.dseg_accumulate ? *o + dx : dx;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < LN_MAXD / 64; ++k) {
    int c = lane + 64 * k;
    if (c < p.D) {
      sh[wv][c] = ag[k];
      sh[wv][p.D + c] = ab[k];
    }
  }
  for (int m = 0; m < 3; ++m) {
    float t = wave_sum(aw[m]);
    if (lane == 0) sh[wv][2 * p.D + m] = t;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * p.D + 3; o += 256)
    q.partial[(size_t)blockIdx.x * (2 * p.D + 4) + o] = (sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o]);
}

// reduce partial rows -> dgamma, dbeta, and softmax-backward of the 3 fusion weights
__global__ void layernorm_bwd_finalize_kernel(const float* __restrict__ partial, int rows, int D, float* dgamma,
                                              float* dbeta, const float* fusion_w, float* dfusion_w) {
  int o = blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = 2 * D + 4;
  if (o < 2 * D) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += partial[(size_t)r * stride + o];
    if (o < D) { if (dgamma) dgamma[o] = s; }
    else if (dbeta) dbeta[o - D] = s;
  }
  if (o == 0 && fusion_w && dfusion_w) {
    float dw[3] = {0.f, 0.f, 0.f}, w[3];
    for (int r = 0; r < rows; ++r)
      for (int m = 0; m < 3; ++m) dw[m] += partial[(size_t)r * stride + 2 * D + m];
    softmax3(fusion_w, w);
    float dot = w[0] * dw[0] + w[1] * dw[1] + w[2] * dw[2];
    for (int m = 0; m < 3; ++m) dfusion_w[m] = w[m] * (dw[m] - dot);
  }
}

// ------------------------------------------------------------------------------------------------
// variance regulariser: v_m = mean_b var_unbiased(f_m[b,:]); loss = |v0-v1| + |v0-v2| + |v1-v2|
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowvar_kernel(const float* __restrict__ f, int B, int D,
                                                     float* __restrict__ rowvar) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += f[(size_t)row * D + c];
  float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int c = lane; c < D; c += 64) {
    float d = f[(size_t)row * D + c] - mean;
    q += d * d;
  }
  q = wave_sum(q);
  if (lane == 0) rowvar[row] = q / (float)(D - 1);
}

// single block: rowvar [3][B] -> loss, sign coefficients gs[3] = dloss/dv_m
__global__ __launch_bounds__(256) void varloss_finish_kernel(const float* __restrict__ rowvar, int B, float* loss,
                                                             float* gs) {
  __shared__ float sh[3][4];
  float v[3];
  for (int m = 0; m < 3; ++m) {
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) s += rowvar[(size_t)m * B + b];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[m][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int m = 0; m < 3; ++m) v[m] = ((sh[m][0] + sh[m][1]) + (sh[m][2] + sh[m][3])) / (float)B;
    auto sgn = [](float a) { return a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f); };
    *loss = fabsf(v[0] - v[1]) + fabsf(v[0] - v[2]) + fabsf(v[1] - v[2]);
    gs[0] = sgn(v[0] - v[1]) + sgn(v[0] - v[2]);
    gs[1] = -sgn(v[0] - v[1]) + sgn(v[1] - v[2]);
    gs[2] = -sgn(v[0] - v[2]) - sgn(v[1] - v[2]);
  }
}

// df[b][j] (+)= gout * gs * 2 (f - mean_b) / ((D-1) B)
__global__ __launch_bounds__(256) void varloss_bwd_kernel(const float* __restrict__ f, int B, int D,
                                                          const float* __restrict__ gout, const float* __restrict__ gs,
                                                          int m, float* __restrict__ df, int accumulate) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += f[(size_t)row * D + c];
  float mean = wave_sum(s) / (float)D;
  float k = gout[0] * gs[m] * 2.f / ((float)(D - 1) * (float)B);
  for (int c = lane; c < D; c += 64) {
    size_t o = (size_t)row * D + c;
    float g = k * (f[o] - mean);
    df[o] = accumulate ? df[o] + g : g;
  }
}

// ------------------------------------------------------------------------------------------------
// cross-entropy / focal loss (mean reduction), single block, fixed summation order
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits,
                                                      const long long* __restrict__ labels, int B, int C, int focal,
                                                      float alpha, float gamma, float* loss, float* dcoef,
                                                      const float* __restrict__ extra, float extra_w) {
  // dcoef[b] = dLoss/dCE_b (before the 1/B), consumed by the backward kernel
  __shared__ float sh[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* z = logits + (size_t)b * C;
    float m = z[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(z[c] - m);
    // a label outside [0, C) never indexes memory: its sample's loss is NaN, so the step's loss is NaN (loud);
    // torch raises IndexError there -- CELossFn does too under ECGMM_CHECK_LABELS=1 (a host sync per call)
    const long long lb = labels[b];
    const float zl = (lb >= 0 && lb < C) ? z[lb] : __builtin_nanf("");
    float ce = (m + logf(se)) - zl;
    float li = ce, dc = 1.f;
    if (focal) {
      float pt = expf(-ce), om = 1.f - pt;
      float w = powf(om, gamma);
      li = alpha * w * ce;
      float dw = (gamma == 0.f) ? 0.f : gamma * powf(om, gamma - 1.f) * pt;
      dc = alpha * (w + dw * ce);
    }
    dcoef[b] = dc;
    acc += li;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  // (+ extra_w * extra: the step loss of train.py:78, CE(fusion_logits) + 0.1 * var_loss, without two more launches)
  if (threadIdx.x == 0) *loss = sh[0] / (float)B + (extra ? extra_w * extra[0] : 0.f);
}

__global__ void ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, int B, int C,
                              const float* __restrict__ dcoef, const float* __restrict__ gout,
                              float* __restrict__ dlogits, float* __restrict__ dextra, float extra_w) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (dextra && b == 0) dextra[0] = gout[0] * extra_w;
  if (b >= B) return;
  const float* z = logits + (size_t)b * C;
  float m = z[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
  float se = 0.f;
  for (int c = 0; c < C; ++c) se += expf(z[c] - m);
  float k = gout[0] * dcoef[b] / (float)B;
  for (int c = 0; c < C; ++c) {
    float pr = expf(z[c] - m) / se;
    dlogits[(size_t)b * C + c] = k * (pr - (c == (int)labels[b] ? 1.f : 0.f));
  }
}

// ------------------------------------------------------------------------------------------------
// dropout: Philox4x32-10 keyed by (seed), counter = (offset + element/4); keep-mask saved as bytes
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0,
                                             unsigned k1) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
  unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
  unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                   unsigned char* __restrict__ mask, long n, float p, unsigned long long seed,
                                   unsigned long long offset) {
  const float scale = 1.f / (1.f - p);
  long groups = (n + 3) / 4;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (long)gridDim.x * blockDim.x) {
    unsigned long long ctr = offset + (unsigned long long)g;
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0, c3 = 0;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c0, c1, c2, c3, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    unsigned rnd[4] = {c0, c1, c2, c3};
    for (int j = 0; j < 4; ++j) {
      long i = g * 4 + j;
      if (i >= n) break;
      float u = (float)(rnd[j] >> 8) * (1.f / 16777216.f);
      unsigned char keep = u >= p;
      mask[i] = keep;
      y[i] = keep ? x[i] * scale : 0.f;
    }
  }
}

__global__ void dropout_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ mask,
                                   float* __restrict__ dx, long n, float p) {
  const float scale = 1.f / (1.f - p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = mask[i] ? dy[i] * scale : 0.f;
}

// ------------------------------------------------------------------------------------------------
// fused Adam over a contiguous fp32 run (torch.optim.Adam semantics, no amsgrad)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                                   float gscale, int vec) {
  long n4 = vec ? (n >> 2) : 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gr = ga[j] * gscale + wd * pa[j];
      ma[j] = b1 * ma[j] + (1.f - b1) * gr;
      va[j] = b2 * va[j] + (1.f - b2) * gr * gr;
      pa[j] -= (lr / bc1) * ma[j] / (sqrtf(va[j]) / bc2s + eps);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  long tail = n4 << 2;
  for (long i = tail + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gr = g[i] * gscale + wd * p[i];
    float mi = b1 * m[i] + (1.f - b1) * gr, vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
  }
}

inline int g1d(long n, int per = 256) {
  long b = (n + per - 1) / per;
  if (b > 2048) b = 2048;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

int ecg_linear_fwd_valu(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, int act,
                        hipStream_t s) {
  long waves = (long)B * Out;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(g1d(waves, 4)), dim3(256), 0, s, x, w, bias, y, B, In, Out, act);
  ECG_CHECK_LAUNCH("linear_fwd");
  return 0;
}
int ecg_linear_dgrad_valu(const float* dy, const float* w, float* dx, int B, int In, int Out, int accumulate,
                          hipStream_t s) {
  hipLaunchKernelGGL(linear_dgrad_kernel, dim3(g1d((long)B * In)), dim3(256), 0, s, dy, w, dx, B, In, Out, accumulate);
  ECG_CHECK_LAUNCH("linear_dgrad");
  return 0;
}
int ecg_linear_wgrad_valu(const float* dy, const float* x, float* dw, float* db, int B, int In, int Out,
                          int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3(g1d((long)Out * In + Out, 4)), dim3(256), 0, s, dy, x, dw, db, B, In, Out,
                     accumulate);
  ECG_CHECK_LAUNCH("linear_wgrad");
  return 0;
}
int ecg_act_bwd(const float* dy, const float* y, float* dz, long n, int act, hipStream_t s) {
  hipLaunchKernelGGL(act_bwd_kernel, dim3(g1d(n)), dim3(256), 0, s, dy, y, dz, n, act);
  ECG_CHECK_LAUNCH("act_bwd");
  return 0;
}
int ecg_axpby(float a, const float* x, float b, float* y, long n, hipStream_t s) {
  hipLaunchKernelGGL(axpby_kernel, dim3(g1d(n)), dim3(256), 0, s, a, x, b, y, n);
  ECG_CHECK_LAUNCH("axpby");
  return 0;
}

static int fill_ln(LnParams& p, const float* const* seg, const int* dims, int nseg, const float* fusion_w,
                   const float* gamma, const float* beta, float* out, float* stat, float* soft_w, int B, float eps) {
  memset(&p, 0, sizeof(p));
  if (nseg < 1 || nseg > 3) ECG_FAIL(ECGMM_ERR_SHAPE, "layernorm: %d segments", nseg);
  if (fusion_w && nseg != 3) ECG_FAIL(ECGMM_ERR_SHAPE, "attention fusion needs 3 segments");
  int D = 0;
  for (int m = 0; m < nseg; ++m) {
    p.seg[m] = seg[m];
    p.dims[m] = dims[m];
    D += dims[m];
  }
  if (D > LN_MAXD) ECG_FAIL(ECGMM_ERR_SHAPE, "layernorm: D=%d > %d", D, LN_MAXD);
  p.nseg = nseg; p.fusion_w = fusion_w; p.gamma = gamma; p.beta = beta; p.out = out; p.stat = stat;
  p.soft_w = soft_w; p.B = B; p.D = D; p.eps = eps;
  return 0;
}

int ecg_layernorm_fwd(const float* const* seg, const int* dims, int nseg, const float* fusion_w, const float* gamma,
                      const float* beta, float* out, float* stat, float* soft_w, int B, float eps, hipStream_t s) {
  LnParams p;
  ECG_TRY(fill_ln(p, seg, dims, nseg, fusion_w, gamma, beta, out, stat, soft_w, B, eps));
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, p);
  ECG_CHECK_LAUNCH("layernorm_fwd");
  return 0;
}

static int ln_bwd_grid(int B) {
  int g = ceil_div(B, 4);
  return g > 256 ? 256 : g;
}
size_t ecg_layernorm_bwd_scratch(int B, int D) { return (size_t)ln_bwd_grid(B) * (2 * D + 4) * sizeof(float); }

int ecg_layernorm_bwd(const float* const* seg, const int* dims, int nseg, const float* fusion_w, const float* gamma,
                      const float* stat, const float* dout, float* const* dseg, int dseg_accumulate, float* dgamma,
                      float* dbeta, float* dfusion_w, int B, float* scratch, hipStream_t s) {
  LnBwdParams q;
  memset(&q, 0, sizeof(q));
  ECG_TRY(fill_ln(q.f, seg, dims, nseg, fusion_w, gamma, nullptr, nullptr, const_cast<float*>(stat), nullptr, B, 0.f));
  q.dout = dout;
  for (int m = 0; m < nseg; ++m) q.dseg[m] = dseg ? dseg[m] : nullptr;
  q.dseg_accumulate = dseg_accumulate;
  q.partial = scratch;
  int grid = ln_bwd_grid(B);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(grid), dim3(256), 0, s, q);
  ECG_CHECK_LAUNCH("layernorm_bwd");
  hipLaunchKernelGGL(layernorm_bwd_finalize_kernel, dim3(ceil_div(2 * q.f.D, 256)), dim3(256), 0, s, scratch, grid,
                     q.f.D, dgamma, dbeta, fusion_w, dfusion_w);
  ECG_CHECK_LAUNCH("layernorm_bwd_finalize");
  return 0;
}

// scratch: rowvar [3][B] + gs[3] (+1 pad)
int ecg_varloss_fwd(const float* f0, const float* f1, const float* f2, int B, int D0, int D1, int D2, float* loss,
                    float* scratch, hipStream_t s) {
  const float* f[3] = {f0, f1, f2};
  int D[3] = {D0, D1, D2};
  for (int m = 0; m < 3; ++m) {
    if (D[m] < 2) ECG_FAIL(ECGMM_ERR_SHAPE, "var_loss: feature dim %d < 2", D[m]);
    hipLaunchKernelGGL(rowvar_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, f[m], B, D[m], scratch + (size_t)m * B);
  }
  hipLaunchKernelGGL(varloss_finish_kernel, dim3(1), dim3(256), 0, s, scratch, B, loss, scratch + 3 * (size_t)B);
  ECG_CHECK_LAUNCH("varloss_fwd");
  return 0;
}
// scratch already holds the three row-variance vectors [3][B] (head_rows_fwd_kernel): loss + sign coefficients only
int ecg_varloss_finish(float* scratch, int B, float* loss, hipStream_t s) {
  hipLaunchKernelGGL(varloss_finish_kernel, dim3(1), dim3(256), 0, s, scratch, B, loss, scratch + 3 * (size_t)B);
  ECG_CHECK_LAUNCH("varloss_finish");
  return 0;
}
int ecg_varloss_bwd(const float* f, int B, int D, const float* gout, const float* scratch, int m, float* df,
                    int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(varloss_bwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, f, B, D, gout,
                     scratch + 3 * (size_t)B, m, df, accumulate);
  ECG_CHECK_LAUNCH("varloss_bwd");
  return 0;
}

int ecg_ce_fwd(const float* logits, const long long* labels, int B, int C, int focal, float alpha, float gamma,
               float* loss, float* dcoef, const float* extra, float extra_w, hipStream_t s) {
  hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), 0, s, logits, labels, B, C, focal, alpha, gamma, loss, dcoef,
                     extra, extra_w);
  ECG_CHECK_LAUNCH("ce_fwd");
  return 0;
}
int ecg_ce_bwd(const float* logits, const long long* labels, int B, int C, const float* dcoef, const float* gout,
               float* dlogits, float* dextra, float extra_w, hipStream_t s) {
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, s, logits, labels, B, C, dcoef, gout,
                     dlogits, dextra, extra_w);
  ECG_CHECK_LAUNCH("ce_bwd");
  return 0;
}

int ecg_dropout_fwd(const float* x, float* y, unsigned char* mask, long n, float p, unsigned long long seed,
                    unsigned long long offset, hipStream_t s) {
  if (!(p >= 0.f && p < 1.f)) ECG_FAIL(ECGMM_ERR_SHAPE, "dropout: p=%f outside [0,1)", p);
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(g1d((n + 3) / 4)), dim3(256), 0, s, x, y, mask, n, p, seed, offset);
  ECG_CHECK_LAUNCH("dropout_fwd");
  return 0;
}
int ecg_dropout_bwd(const float* dy, const unsigned char* mask, float* dx, long n, float p, hipStream_t s) {
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(g1d(n)), dim3(256), 0, s, dy, mask, dx, n, p);
  ECG_CHECK_LAUNCH("dropout_bwd");
  return 0;
}

int ecg_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd,
             long step, float gscale, hipStream_t s) {
  if (step < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "adam: step %ld < 1", step);
  double bc1 = 1.0 - pow((double)b1, (double)step);
  double bc2 = 1.0 - pow((double)b2, (double)step);
  int vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  hipLaunchKernelGGL(adam_kernel, dim3(g1d(vec ? (n >> 2) + 1 : n, 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2,
                     eps, wd, (float)bc1, (float)sqrt(bc2), gscale, vec);
  ECG_CHECK_LAUNCH("adam");
  return 0;
}
