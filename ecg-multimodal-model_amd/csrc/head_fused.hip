// Fused row kernels and small dense kernels of the multimodal head (gfx950, fp32).
//
// plan_head.hip enqueues the head (multimodal_paper_modal_balance.py:326-354) as ~45 launches of 3-20 us each, one
// after the other, at the one point of the step where nothing else can run: 0.43 ms of a 8.3 ms step (tools/head_time.py).
// Everything in the head except the two Linear layers of the fusion classifier is ROW-LOCAL (LayerNorms, branch
// classifiers, attention scaling + concat + LayerNorm, row variances), so it is one wave-per-row kernel per direction:
//
//   head_rows_fwd_kernel : raw_m -> feat_m = LN_m(raw_m), logits_m = feat_m W_m^T + b_m, rowvar_m, fused = LN(cat(w_m feat_m))
//   head_rows_bwd_kernel : d fused, d logits_m, d var -> d raw_m  + per-block partial rows of every parameter gradient
//   head_finalize_kernel : folds the partial rows into the 16 parameter gradients (+ fusion_classifier.0.bias)
//
// and the Linear(D -> H) of the fusion classifier runs on dense16_kernel: one wave per 16x16 output tile, exact-fp32
// MFMA (v_mfma_f32_16x16x4_f32) with operands straight from global memory / L2 -- for B = 256 the implicit-GEMM kernel
// has 2 workgroups walking a 24-stage K chain (57 us); 128 independent waves take ~10 us.
// The arithmetic is that of head.hip's kernels (same formulas, fp32, fixed summation order: bitwise reproducible).
#include "ops.h"

namespace {

__device__ __forceinline__ void softmax3f(const float* w, float* o) {
  float m = fmaxf(w[0], fmaxf(w[1], w[2]));
  float e0 = expf(w[0] - m), e1 = expf(w[1] - m), e2 = expf(w[2] - m);
  float inv = 1.f / (e0 + e1 + e2);
  o[0] = e0 * inv; o[1] = e1 * inv; o[2] = e2 * inv;
}

struct HeadRows {
  const float* raw[3];
  const float* ln_g[3];
  const float* ln_b[3];
  const float* cls_w[3];   // [NC][d_m]
  const float* cls_b[3];
  const float* aw;         // attention weights [3] (raw; softmax in-kernel)
  const float* fg;         // fusion LayerNorm gamma / beta [D]
  const float* fb;
  float* feat[3];          // [B][d_m]
  float* stat[3];          // [B][2] mean, rstd
  float* logits[3];        // [B][NC]
  float* rowvar;           // [3][B]
  float* fused;            // [B][D]
  float* statf;            // [B][2]
  float* soft_w;           // [3] or null
  int dim[3];
  int B, D, NC;
  float eps;
};

// H0, H1, H2 = values per lane of the three modalities (dim_m <= 64 H_m; arrays are sized for the largest, the unused
// slots fold away after unrolling), NCM = max classes.
// Every global load of the row is issued before the first dependent use (the row is a chain of ~20 reductions; with the
// loads left where they are used, each phase paid its own L2/HBM latency: 30 us for 64 workgroups' worth of rows).
template <int H0, int H1, int H2, int NCM>
__global__ __launch_bounds__(256) void head_rows_fwd_kernel(HeadRows p) {
  constexpr int HM = H0 > H1 ? (H0 > H2 ? H0 : H2) : (H1 > H2 ? H1 : H2);
#define HMM(m) ((m) == 0 ? H0 : (m) == 1 ? H1 : H2)
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  float sw[3];
  softmax3f(p.aw, sw);
  if (p.soft_w && blockIdx.x == 0 && threadIdx.x < 3) p.soft_w[threadIdx.x] = sw[threadIdx.x];
  if (row >= p.B) return;
  float v[3][HM], lg[3][HM], lb[3][HM], cw[3][NCM][HM], cb[3][NCM], fgv[3][HM], fbv[3][HM];
  {
    int off = 0;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int d = p.dim[m];
#pragma unroll
      for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
        const int c = lane + 64 * k;
        const bool ok = c < d;
        v[m][k] = ok ? p.raw[m][(size_t)row * d + c] : 0.f;
        lg[m][k] = ok ? p.ln_g[m][c] : 0.f;
        lb[m][k] = ok ? p.ln_b[m][c] : 0.f;
        fgv[m][k] = ok ? p.fg[off + c] : 0.f;
        fbv[m][k] = ok ? p.fb[off + c] : 0.f;
#pragma unroll
        for (int cc = 0; cc < NCM; ++cc) cw[m][cc][k] = (ok && cc < p.NC) ? p.cls_w[m][(size_t)cc * d + c] : 0.f;
      }
#pragma unroll
      for (int cc = 0; cc < NCM; ++cc) cb[m][cc] = cc < p.NC ? p.cls_b[m][cc] : 0.f;
      off += d;
    }
  }
  float f[3][HM];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int d = p.dim[m];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < HM; ++k) if (k < HMM(m)) s += v[m][k];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
      const float dd = lane + 64 * k < d ? v[m][k] - mean : 0.f;
      q += dd * dd;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + p.eps);
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
      const int c = lane + 64 * k;
      f[m][k] = c < d ? (v[m][k] - mean) * rstd * lg[m][k] + lb[m][k] : 0.f;
      if (c < d) p.feat[m][(size_t)row * d + c] = f[m][k];
      s2 += f[m][k];
    }
    if (lane == 0) {
      p.stat[m][2 * row] = mean;
      p.stat[m][2 * row + 1] = rstd;
    }
    // branch classifier (linear_fwd_kernel's order: strided partial sums, butterfly, + bias)
#pragma unroll
    for (int c = 0; c < NCM; ++c) {
      if (c >= p.NC) break;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < HM; ++k) if (k < HMM(m)) t += f[m][k] * cw[m][c][k];
      t = wave_sum(t);
      if (lane == 0) p.logits[m][(size_t)row * p.NC + c] = t + cb[m][c];
    }
    // unbiased variance of the feature row (rowvar_kernel)
    const float fmean = wave_sum(s2) / (float)d;
    float q2 = 0.f;
#pragma unroll
    for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
      const float dd = lane + 64 * k < d ? f[m][k] - fmean : 0.f;
      q2 += dd * dd;
    }
    q2 = wave_sum(q2);
    if (lane == 0) p.rowvar[(size_t)m * p.B + row] = q2 / (float)(d - 1);
  }
  // attention fusion: softmax(3) scale, concat, LayerNorm over D
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
      f[m][k] *= sw[m];   // (0 stays 0 beyond d_m)
      s += f[m][k];
    }
  const float mean = wave_sum(s) / (float)p.D;
  float q = 0.f;
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
      const float dd = lane + 64 * k < p.dim[m] ? f[m][k] - mean : 0.f;
      q += dd * dd;
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)p.D + p.eps);
  int off = 0;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
#pragma unroll
    for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
      const int c = lane + 64 * k;
      if (c < p.dim[m]) p.fused[(size_t)row * p.D + off + c] = (f[m][k] - mean) * rstd * fgv[m][k] + fbv[m][k];
    }
    off += p.dim[m];
  }
  if (lane == 0) {
    p.statf[2 * row] = mean;
    p.statf[2 * row + 1] = rstd;
  }
}

// Partial-row layout (floats), one row per block:
//   [0, D) d fusion gamma | [D, 2D) d fusion beta | [2D, 2D+4) d attention (pre-softmax-backward) sums
//   then per modality m: d gamma_m [d_m] | d beta_m [d_m] | d cls_w_m [NC][d_m] | d cls_b_m [4]
struct HeadRowsBwd {
  HeadRows f;
  const float* dfused;   // [B][D] or null (fusion logits not in the loss)
  const float* dlog[3];  // [B][NC] or null
  const float* dvar;     // scalar or null
  const float* gs;       // var-loss sign coefficients [3] (varloss_finish_kernel)
  float* draw[3];        // nullable (frozen encoder)
  float* partial;
  int P;
  int have[3];           // does anything upstream reach branch m?
};

__host__ __device__ inline int head_part_base(const int* dim, int D, int NC, int m) {
  int o = 2 * D + 4;
  for (int j = 0; j < m; ++j) o += 2 * dim[j] + NC * dim[j] + 4;
  return o;
}

template <int H0, int H1, int H2, int NCM>
__global__ __launch_bounds__(256) void head_rows_bwd_kernel(HeadRowsBwd q) {
  constexpr int HM = H0 > H1 ? (H0 > H2 ? H0 : H2) : (H1 > H2 ? H1 : H2);
  const HeadRows& p = q.f;
  extern __shared__ float shp[];   // [P]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float sw[3];
  softmax3f(p.aw, sw);
  float afg[3][HM], afb[3][HM], ag[3][HM], ab[3][HM], aW[3][NCM][HM], acb[3][NCM], aw3[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 3; ++m) {
#pragma unroll
    for (int k = 0; k < HM; ++k) if (k < HMM(m)) afg[m][k] = afb[m][k] = ag[m][k] = ab[m][k] = 0.f;
#pragma unroll
    for (int c = 0; c < NCM; ++c) {
      acb[m][c] = 0.f;
#pragma unroll
      for (int k = 0; k < HM; ++k) if (k < HMM(m)) aW[m][c][k] = 0.f;
    }
  }
  for (int row = blockIdx.x * 4 + wv; row < p.B; row += gridDim.x * 4) {
    // every load of the row first (see head_rows_fwd_kernel)
    float ft[3][HM], df[3][HM], dfu[3][HM], fgv[3][HM], rw[3][HM], lg[3][HM], cw[3][NCM][HM], gl[3][NCM];
    float mean_m[3], rstd_m[3];
    const float mean_f = q.dfused ? p.statf[2 * row] : 0.f, rstd_f = q.dfused ? p.statf[2 * row + 1] : 0.f;
    const float dvar = q.dvar ? q.dvar[0] : 0.f;
    {
      int off = 0;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int d = p.dim[m];
        mean_m[m] = p.stat[m][2 * row];
        rstd_m[m] = p.stat[m][2 * row + 1];
#pragma unroll
        for (int cc = 0; cc < NCM; ++cc) gl[m][cc] = (q.dlog[m] && cc < p.NC) ? q.dlog[m][(size_t)row * p.NC + cc] : 0.f;
#pragma unroll
        for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
          const int c = lane + 64 * k;
          const bool ok = c < d;
          ft[m][k] = ok ? p.feat[m][(size_t)row * d + c] : 0.f;
          rw[m][k] = ok ? p.raw[m][(size_t)row * d + c] : 0.f;
          lg[m][k] = ok ? p.ln_g[m][c] : 0.f;
          dfu[m][k] = (ok && q.dfused) ? q.dfused[(size_t)row * p.D + off + c] : 0.f;
          fgv[m][k] = (ok && q.dfused) ? p.fg[off + c] : 0.f;
#pragma unroll
          for (int cc = 0; cc < NCM; ++cc)
            cw[m][cc][k] = (ok && q.dlog[m] && cc < p.NC) ? p.cls_w[m][(size_t)cc * d + c] : 0.f;
          df[m][k] = 0.f;
        }
        off += d;
      }
    }
    if (q.dfused) {   // LayerNorm(D) backward over the scaled concat, then the softmax(3) scaling
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int k = 0; k < HM; ++k) if (k < HMM(m))
          if (lane + 64 * k < p.dim[m]) {
            const float xh = (ft[m][k] * sw[m] - mean_f) * rstd_f;
            afg[m][k] += dfu[m][k] * xh;
            afb[m][k] += dfu[m][k];
            const float dg = dfu[m][k] * fgv[m][k];
            s1 += dg;
            s2 += dg * xh;
          }
      s1 = wave_sum(s1) / (float)p.D;
      s2 = wave_sum(s2) / (float)p.D;
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int k = 0; k < HM; ++k) if (k < HMM(m))
          if (lane + 64 * k < p.dim[m]) {
            const float xh = (ft[m][k] * sw[m] - mean_f) * rstd_f;
            const float dx = rstd_f * (dfu[m][k] * fgv[m][k] - s1 - xh * s2);
            aw3[m] += dx * ft[m][k];
            df[m][k] = dx * sw[m];
          }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int d = p.dim[m];
      if (!q.have[m]) {
        if (q.draw[m])
#pragma unroll
          for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
            const int c = lane + 64 * k;
            if (c < d) q.draw[m][(size_t)row * d + c] = 0.f;
          }
        continue;
      }
      if (q.dvar) {   // varloss_bwd_kernel
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < HM; ++k) if (k < HMM(m)) s += ft[m][k];
        const float fmean = wave_sum(s) / (float)d;
        const float kk = dvar * q.gs[m] * 2.f / ((float)(d - 1) * (float)p.B);
#pragma unroll
        for (int k = 0; k < HM; ++k) if (k < HMM(m))
          if (lane + 64 * k < d) df[m][k] += kk * (ft[m][k] - fmean);
      }
      if (q.dlog[m]) {   // branch classifier: d feat += dlogits W, dW += dlogits^T feat, db += dlogits
#pragma unroll
        for (int c = 0; c < NCM; ++c) {
          if (c >= p.NC) break;
          acb[m][c] += gl[m][c];
#pragma unroll
          for (int k = 0; k < HM; ++k) if (k < HMM(m))
            if (lane + 64 * k < d) {
              aW[m][c][k] += gl[m][c] * ft[m][k];
              df[m][k] += gl[m][c] * cw[m][c][k];
            }
        }
      }
      // LayerNorm_m backward
      const float mean = mean_m[m], rstd = rstd_m[m];
      float xh[HM], dg[HM], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
        if (lane + 64 * k < d) {
          xh[k] = (rw[m][k] - mean) * rstd;
          ag[m][k] += df[m][k] * xh[k];
          ab[m][k] += df[m][k];
          dg[k] = df[m][k] * lg[m][k];
          s1 += dg[k];
          s2 += dg[k] * xh[k];
        } else {
          xh[k] = dg[k] = 0.f;
        }
      }
      s1 = wave_sum(s1) / (float)d;
      s2 = wave_sum(s2) / (float)d;
      if (q.draw[m])
#pragma unroll
        for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
          const int c = lane + 64 * k;
          if (c < d) q.draw[m][(size_t)row * d + c] = rstd * (dg[k] - s1 - xh[k] * s2);
        }
    }
  }
  // fold the four waves of the block in a fixed order (wave 0 stores, waves 1..3 add), one partial row per block
  for (int o = threadIdx.x; o < q.P; o += 256) shp[o] = 0.f;   // (padding slots and skipped branches read as 0)
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
      int off = 0;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int d = p.dim[m], base = head_part_base(p.dim, p.D, p.NC, m);
#pragma unroll
        for (int k = 0; k < HM; ++k) if (k < HMM(m)) {
          const int c = lane + 64 * k;
          if (c < d) {
            float* t = shp;
            auto put = [&](int idx, float v) { t[idx] = w == 0 ? v : t[idx] + v; };
            put(off + c, afg[m][k]);
            put(p.D + off + c, afb[m][k]);
            put(base + c, ag[m][k]);
            put(base + d + c, ab[m][k]);
#pragma unroll
            for (int cc = 0; cc < NCM; ++cc)
              if (cc < p.NC) put(base + 2 * d + cc * d + c, aW[m][cc][k]);
          }
        }
        const float t3 = wave_sum(aw3[m]);
        if (lane == 0) shp[2 * p.D + m] = w == 0 ? t3 : shp[2 * p.D + m] + t3;
        if (lane < NCM && lane < p.NC) {
          float v = 0.f;
#pragma unroll
          for (int cc = 0; cc < NCM; ++cc) v = lane == cc ? acb[m][cc] : v;
          const int idx = base + 2 * d + p.NC * d + lane;
          shp[idx] = w == 0 ? v : shp[idx] + v;
        }
        off += d;
      }
    }
    __syncthreads();
  }
  for (int o = threadIdx.x; o < q.P; o += 256) q.partial[(size_t)blockIdx.x * q.P + o] = shp[o];
}

// fold partial rows -> gradient tensors.  A segment copies sum_r partial[r][off + i] to dst[i]; the attention weights go
// through the softmax(3) backward; `col_src` segments are column sums of a [rows][n] matrix (a Linear's bias gradient).
struct HeadSeg {
  float* dst;
  int off, n;
};
struct HeadFinalize {
  HeadSeg seg[16];
  int nseg;
  const float* partial;
  int rows, P;
  const float* aw;        // attention weights (forward) or null
  float* daw;             // their gradient or null
  int aw_off;
  const float* col_src;   // [col_rows][col_n] or null
  float* col_dst;
  int col_rows, col_n;
};

__global__ __launch_bounds__(256) void head_finalize_kernel(HeadFinalize h) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o < h.P) {
    float* dst = nullptr;
    for (int s = 0; s < h.nseg; ++s)
      if (o >= h.seg[s].off && o < h.seg[s].off + h.seg[s].n) dst = h.seg[s].dst ? h.seg[s].dst + (o - h.seg[s].off) : nullptr;
    if (dst) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // eight loads in flight; fixed fold order
      int r = 0;
      for (; r + 8 <= h.rows; r += 8)
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += h.partial[(size_t)(r + u) * h.P + o];
      for (; r < h.rows; ++r) a[0] += h.partial[(size_t)r * h.P + o];
      *dst = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
  } else if (o < h.P + h.col_n && h.col_src) {
    const int j = o - h.P;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = 0;
    for (; r + 8 <= h.col_rows; r += 8)
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += h.col_src[(size_t)(r + u) * h.col_n + j];
    for (; r < h.col_rows; ++r) a[0] += h.col_src[(size_t)r * h.col_n + j];
    h.col_dst[j] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  if (blockIdx.x == 0 && threadIdx.x < 64 && h.daw) {   // one wave: rows across lanes, butterfly, softmax(3) backward
    float dw[3] = {0.f, 0.f, 0.f}, w[3];
    for (int r = threadIdx.x; r < h.rows; r += 64)
#pragma unroll
      for (int m = 0; m < 3; ++m) dw[m] += h.partial[(size_t)r * h.P + h.aw_off + m];
#pragma unroll
    for (int m = 0; m < 3; ++m) dw[m] = wave_sum(dw[m]);
    if (threadIdx.x == 0) {
      softmax3f(h.aw, w);
      const float dot = w[0] * dw[0] + w[1] * dw[1] + w[2] * dw[2];
      for (int m = 0; m < 3; ++m) h.daw[m] = w[m] * (dw[m] - dot);
    }
  }
}

// dz[b][h] = relu'(h_act) * dropout'(mask) * sum_c dlogits[b][c] W[c][h]: the input gradient of Linear(H -> NC) pushed through
// Dropout and ReLU in one elementwise pass (linear_dgrad_kernel + dropout_bwd_kernel + act_bwd_kernel)
__global__ __launch_bounds__(256) void fc_dgrad_drop_relu_kernel(const float* __restrict__ dlog, const float* __restrict__ w,
                                                                  const unsigned char* __restrict__ mask,
                                                                  const float* __restrict__ hact, float* __restrict__ dz,
                                                                  int B, int H, int NC, float keep_scale) {
  const long total = (long)B * H;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int b = (int)(e / H), j = (int)(e % H);
    float s = 0.f;
    for (int c = 0; c < NC; ++c) s += dlog[(size_t)b * NC + c] * w[(size_t)c * H + j];
    if (mask) s = mask[e] ? s * keep_scale : 0.f;
    dz[e] = hact[e] > 0.f ? s : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// dense16_kernel: out[m][n] = act(sum_k A(m,k) B(k,n) + bias[n]), one wave per 16x16 tile, v_mfma_f32_16x16x4_f32.
//   AK: A(m,k) = a[m*lda + k] (k contiguous, 16-B loads), else a[k*lda + m];   BK: B(k,n) = b[n*ldb + k], else b[k*ldb + n]
// Lane (i = lane & 15, g = lane >> 4) feeds row/column i with k = k0 + 4 g + j for the j-th MFMA of a 16-k step: a
// permutation of the K axis applied to both operands alike.  M, N, K multiples of 16.
// ------------------------------------------------------------------------------------------------
template <bool AK, bool BK>
__global__ __launch_bounds__(256) void dense16_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                      const float* __restrict__ bias, float* __restrict__ out, int M,
                                                      int N, int K, int lda, int ldb, int ldo, int act) {
  const int lane = threadIdx.x & 63;
  const int tiles_n = N >> 4;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= (M >> 4) * tiles_n) return;
  const int m0 = (tile / tiles_n) << 4, n0 = (tile % tiles_n) << 4;
  const int i = lane & 15, g = lane >> 4;
  const float* ap = AK ? A + (size_t)(m0 + i) * lda + 4 * g : A + (size_t)(4 * g) * lda + m0 + i;
  const float* bp = BK ? Bm + (size_t)(n0 + i) * ldb + 4 * g : Bm + (size_t)(4 * g) * ldb + n0 + i;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 64) {
    float av[4][4], bv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k0 + 16 * u;
      const bool ok = kk < K;   // wave-uniform
      if (AK) {
        const float4 t = ok ? *reinterpret_cast<const float4*>(ap + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
        av[u][0] = t.x; av[u][1] = t.y; av[u][2] = t.z; av[u][3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) av[u][j] = ok ? ap[(size_t)(kk + j) * lda] : 0.f;
      }
      if (BK) {
        const float4 t = ok ? *reinterpret_cast<const float4*>(bp + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
        bv[u][0] = t.x; bv[u][1] = t.y; bv[u][2] = t.z; bv[u][3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[u][j] = ok ? bp[(size_t)(kk + j) * ldb] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][j], bv[u][j], acc, 0, 0, 0);
  }
  const float bb = bias ? bias[n0 + i] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = acc[r] + bb;
    if (act == ECGMM_ACT_RELU) v = fmaxf(v, 0.f);
    out[(size_t)(m0 + 4 * g + r) * ldo + n0 + i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Squeeze-excite MLP (SEBlock.fc, multimodal_paper_modal_balance.py:35-46): g = sigmoid(W2 relu(W1 m + b1) + b2) per
// sample, C -> C/16 -> C.  As per-op launches it is 2 kernels forward and 7 backward (two act_bwd, two Linear backward
// = dgrad + wgrad each, a scale) of 5-9 us each, in the middle of every residual block's dependency chain; here one
// kernel forward, two backward (per-sample part; weight / bias gradients summed over the batch in a fixed order).
// ------------------------------------------------------------------------------------------------
constexpr int SE_MAXC = 1024, SE_MAXR = 64;
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(const float* __restrict__ m, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ h,
                                                         float* __restrict__ g, int C, int CR) {
  __shared__ float sm[SE_MAXC], shh[SE_MAXR];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < C; c += 256) sm[c] = m[(size_t)n * C + c];
  __syncthreads();
  for (int j = wv; j < CR; j += 4) {   // one wave per hidden unit: strided partial sums, butterfly, + bias (linear_fwd_kernel)
    float t = 0.f;
    for (int c = lane; c < C; c += 64) t += sm[c] * w1[(size_t)j * C + c];
    t = wave_sum(t);
    if (lane == 0) {
      const float v = fmaxf(t + b1[j], 0.f);
      shh[j] = v;
      h[(size_t)n * CR + j] = v;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int j = 0; j < CR; ++j) t += shh[j] * w2[(size_t)c * CR + j];
    g[(size_t)n * C + c] = 1.f / (1.f + expf(-(t + b2[c])));
  }
}

// per sample: ds = dg * g (1 - g);  dh = (ds W2) * [h > 0];  dm = (dh W1) * scale      (scale = 1 / L: the mean over L)
__global__ __launch_bounds__(256) void se_mlp_bwd_samples_kernel(const float* __restrict__ dg, const float* __restrict__ g,
                                                                 const float* __restrict__ h, const float* __restrict__ w1,
                                                                 const float* __restrict__ w2, float* __restrict__ ds,
                                                                 float* __restrict__ dh, float* __restrict__ dm, int C,
                                                                 int CR, float scale) {
  __shared__ float sds[SE_MAXC], sdh[SE_MAXR];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float gv = g[(size_t)n * C + c];
    const float v = dg[(size_t)n * C + c] * gv * (1.f - gv);
    sds[c] = v;
    ds[(size_t)n * C + c] = v;
  }
  __syncthreads();
  for (int j = wv; j < CR; j += 4) {
    float t = 0.f;
    for (int c = lane; c < C; c += 64) t += sds[c] * w2[(size_t)c * CR + j];
    t = wave_sum(t);
    if (lane == 0) {
      const float v = h[(size_t)n * CR + j] > 0.f ? t : 0.f;
      sdh[j] = v;
      dh[(size_t)n * CR + j] = v;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int j = 0; j < CR; ++j) t += sdh[j] * w1[(size_t)j * C + c];
    dm[(size_t)n * C + c] = t * scale;
  }
}

// dW2[c][j] = sum_n ds[n][c] h[n][j], db2[c] = sum_n ds[n][c], dW1[j][c] = sum_n dh[n][j] m[n][c], db1[j] = sum_n dh[n][j]
// 64 outputs per 1024-thread block, the batch cut into 16 slices (n = s mod 16) with four independent load pairs per
// trip, folded in LDS in a fixed order.  Null outputs are skipped.  (4 slices of 128 dependent L2 loads each: 26 us.)
__global__ __launch_bounds__(1024) void se_mlp_bwd_weights_kernel(const float* __restrict__ ds, const float* __restrict__ dh,
                                                                  const float* __restrict__ h, const float* __restrict__ m,
                                                                  float* __restrict__ dw1, float* __restrict__ db1,
                                                                  float* __restrict__ dw2, float* __restrict__ db2, int N,
                                                                  int C, int CR) {
  constexpr int NS = 16;
  __shared__ float sh[NS][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  const int T2 = C * CR, T = 2 * T2 + C + CR;
  float acc = 0.f;
  float* dst = nullptr;
  // every output is sum_n a[n * sa + ia] * (b ? b[n * sb + ib] : 1)
  const float *pa = nullptr, *pb = nullptr;
  size_t sa = 0, sb = 0;
  if (o < T) {
    if (o < T2) {                       // dW2[c][j]
      const int c = o / CR, j = o - c * CR;
      dst = dw2 ? dw2 + o : nullptr;
      pa = ds + c; sa = C; pb = h + j; sb = CR;
    } else if (o < 2 * T2) {            // dW1[j][c]
      const int q = o - T2, j = q / C, c = q - j * C;
      dst = dw1 ? dw1 + q : nullptr;
      pa = dh + j; sa = CR; pb = m + c; sb = C;
    } else if (o < 2 * T2 + C) {        // db2[c]
      const int c = o - 2 * T2;
      dst = db2 ? db2 + c : nullptr;
      pa = ds + c; sa = C;
    } else {                            // db1[j]
      const int j = o - 2 * T2 - C;
      dst = db1 ? db1 + j : nullptr;
      pa = dh + j; sa = CR;
    }
  }
  if (dst) {
    int n = sl;
    for (; n + 3 * NS < N; n += 4 * NS) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = pa[(size_t)(n + NS * u) * sa];
        b[u] = pb ? pb[(size_t)(n + NS * u) * sb] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += a[u] * b[u];
    }
    for (; n < N; n += NS) acc += pa[(size_t)n * sa] * (pb ? pb[(size_t)n * sb] : 1.f);
  }
  sh[sl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sl == 0 && dst) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) t += sh[k][threadIdx.x];
    *dst = t;
  }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

bool ecg_se_mlp_fused_ok(int C, int CR) {
  static const bool on = [] { const char* e = getenv("ECGMM_SE_MLP_FUSED"); return !(e && e[0] == '0'); }();
  return on && C >= 1 && C <= SE_MAXC && CR >= 1 && CR <= SE_MAXR;
}
int ecg_se_mlp_fwd(const float* m, const float* w1, const float* b1, const float* w2, const float* b2, float* h, float* g,
                   int N, int C, int CR, hipStream_t s) {
  hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(N), dim3(256), 0, s, m, w1, b1, w2, b2, h, g, C, CR);
  ECG_CHECK_LAUNCH("se_mlp_fwd");
  return 0;
}
// ds [N][C] and dh [N][CR] are scratch outputs of the per-sample kernel, consumed by the weight kernel
int ecg_se_mlp_bwd(const float* dg, const float* g, const float* h, const float* m, const float* w1, const float* w2,
                   float* ds, float* dh, float* dm, float* dw1, float* db1, float* dw2, float* db2, int N, int C, int CR,
                   float scale, hipStream_t s) {
  hipLaunchKernelGGL(se_mlp_bwd_samples_kernel, dim3(N), dim3(256), 0, s, dg, g, h, w1, w2, ds, dh, dm, C, CR, scale);
  ECG_CHECK_LAUNCH("se_mlp_bwd_samples");
  if (dw1 || db1 || dw2 || db2) {
    const int T = 2 * C * CR + C + CR;
    hipLaunchKernelGGL(se_mlp_bwd_weights_kernel, dim3((T + 63) / 64), dim3(1024), 0, s, (const float*)ds, (const float*)dh, h,
                       m, dw1, db1, dw2, db2, N, C, CR);
    ECG_CHECK_LAUNCH("se_mlp_bwd_weights");
  }
  return 0;
}

// Can the fused head serve this description?  (dims <= 512, classes <= 4, MFMA-tileable sizes)
bool ecg_head_fused_ok(const int* dim, int B, int hidden, int num_classes) {
  static const bool on = [] { const char* e = getenv("ECGMM_HEAD_FUSED"); return !(e && e[0] == '0'); }();
  if (!on) return false;
  const int D = dim[0] + dim[1] + dim[2];
  for (int m = 0; m < 3; ++m)
    if (dim[m] < 2 || dim[m] > 512) return false;
  return num_classes >= 1 && num_classes <= 4 && B % 16 == 0 && D % 16 == 0 && hidden % 16 == 0;
}

int ecg_head_partial_floats(const int* dim, int num_classes) {
  const int D = dim[0] + dim[1] + dim[2];
  return head_part_base(dim, D, num_classes, 3);
}
int ecg_head_bwd_blocks(int B) {
  int g = (B + 3) / 4;
  return g > 64 ? 64 : g;
}

int ecg_head_rows_fwd(const float* const* raw, const float* const* ln_g, const float* const* ln_b,
                      const float* const* cls_w, const float* const* cls_b, const float* aw, const float* fg,
                      const float* fb, float* const* feat, float* const* stat, float* const* logits, float* rowvar,
                      float* fused, float* statf, float* soft_w, const int* dim, int B, int NC, float eps,
                      hipStream_t s) {
  HeadRows p;
  for (int m = 0; m < 3; ++m) {
    p.raw[m] = raw[m]; p.ln_g[m] = ln_g[m]; p.ln_b[m] = ln_b[m]; p.cls_w[m] = cls_w[m]; p.cls_b[m] = cls_b[m];
    p.feat[m] = feat[m]; p.stat[m] = stat[m]; p.logits[m] = logits[m]; p.dim[m] = dim[m];
  }
  p.aw = aw; p.fg = fg; p.fb = fb; p.rowvar = rowvar; p.fused = fused; p.statf = statf; p.soft_w = soft_w;
  p.B = B; p.D = dim[0] + dim[1] + dim[2]; p.NC = NC; p.eps = eps;
  const dim3 grid((B + 3) / 4), block(256);
  const bool pmb = dim[0] <= 256 && dim[1] <= 256 && dim[2] <= 256;    // multimodal_paper_modal_balance.py: 3 x 256
  const bool tab = dim[0] <= 512 && dim[1] <= 128 && dim[2] <= 64;      // multimodal.py: 512 / 128 / 32
  if (pmb && NC <= 2) hipLaunchKernelGGL((head_rows_fwd_kernel<4, 4, 4, 2>), grid, block, 0, s, p);
  else if (tab && NC <= 2) hipLaunchKernelGGL((head_rows_fwd_kernel<8, 2, 1, 2>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((head_rows_fwd_kernel<8, 8, 8, 4>), grid, block, 0, s, p);
  ECG_CHECK_LAUNCH("head_rows_fwd");
  return 0;
}

// partial: ecg_head_bwd_blocks(B) x ecg_head_partial_floats() floats
int ecg_head_rows_bwd(const float* const* raw, const float* const* ln_g, const float* const* cls_w, const float* aw,
                      const float* fg, const float* const* feat, const float* const* stat, const float* statf,
                      const float* dfused, const float* const* dlog, const float* dvar, const float* gs,
                      float* const* draw, const int* have, float* partial, const int* dim, int B, int NC,
                      hipStream_t s) {
  HeadRowsBwd q;
  memset(&q, 0, sizeof(q));
  HeadRows& p = q.f;
  for (int m = 0; m < 3; ++m) {
    p.raw[m] = raw[m]; p.ln_g[m] = ln_g[m]; p.cls_w[m] = cls_w[m]; p.feat[m] = const_cast<float*>(feat[m]);
    p.stat[m] = const_cast<float*>(stat[m]); p.dim[m] = dim[m];
    q.dlog[m] = dlog[m]; q.draw[m] = draw[m]; q.have[m] = have[m];
  }
  p.aw = aw; p.fg = fg; p.statf = const_cast<float*>(statf);
  p.B = B; p.D = dim[0] + dim[1] + dim[2]; p.NC = NC;
  q.dfused = dfused; q.dvar = dvar; q.gs = gs; q.partial = partial;
  q.P = ecg_head_partial_floats(dim, NC);
  const dim3 grid(ecg_head_bwd_blocks(B)), block(256);
  const size_t lds = (size_t)q.P * sizeof(float);   // 18.5 KB at 3 x 256 features, 2 classes; <= 64 KB for every shape ecg_head_fused_ok admits
  const bool pmb = dim[0] <= 256 && dim[1] <= 256 && dim[2] <= 256;
  const bool tab = dim[0] <= 512 && dim[1] <= 128 && dim[2] <= 64;
  if (pmb && NC <= 2) hipLaunchKernelGGL((head_rows_bwd_kernel<4, 4, 4, 2>), grid, block, lds, s, q);
  else if (tab && NC <= 2) hipLaunchKernelGGL((head_rows_bwd_kernel<8, 2, 1, 2>), grid, block, lds, s, q);
  else hipLaunchKernelGGL((head_rows_bwd_kernel<8, 8, 8, 4>), grid, block, lds, s, q);
  ECG_CHECK_LAUNCH("head_rows_bwd");
  return 0;
}

// grads in the head's parameter-table order (plan_head.hip): entries may be null.  fc0_db = column sums of dz [B][H].
int ecg_head_finalize(const float* partial, int rows, const int* dim, int NC, float* const* g_ln, float* const* g_cls,
                      float* g_aw, float* g_fg, float* g_fb, const float* aw, const int* have, const int* have_cls,
                      int have_fusion, const float* dz, int B, int H, float* fc0_db, hipStream_t s) {
  HeadFinalize h;
  memset(&h, 0, sizeof(h));
  const int D = dim[0] + dim[1] + dim[2];
  h.P = ecg_head_partial_floats(dim, NC);
  h.partial = partial; h.rows = rows;
  int n = 0;
  if (have_fusion) {
    h.seg[n++] = {g_fg, 0, D};
    h.seg[n++] = {g_fb, D, D};
    h.aw = aw; h.daw = g_aw; h.aw_off = 2 * D;
  }
  for (int m = 0; m < 3; ++m) {
    const int base = head_part_base(dim, D, NC, m), d = dim[m];
    if (have[m]) {
      h.seg[n++] = {g_ln[2 * m], base, d};
      h.seg[n++] = {g_ln[2 * m + 1], base + d, d};
    }
    if (have_cls[m]) {
      h.seg[n++] = {g_cls[2 * m], base + 2 * d, NC * d};
      h.seg[n++] = {g_cls[2 * m + 1], base + 2 * d + NC * d, NC};
    }
  }
  h.nseg = n;
  if (dz && fc0_db) { h.col_src = dz; h.col_dst = fc0_db; h.col_rows = B; h.col_n = H; }
  const int total = h.P + (h.col_src ? H : 0);
  hipLaunchKernelGGL(head_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, s, h);
  ECG_CHECK_LAUNCH("head_finalize");
  return 0;
}

int ecg_fc_dgrad_drop_relu(const float* dlog, const float* w, const unsigned char* mask, const float* hact, float* dz,
                           int B, int H, int NC, float dropout_p, hipStream_t s) {
  const long total = (long)B * H;
  int grid = (int)((total + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(fc_dgrad_drop_relu_kernel, dim3(grid), dim3(256), 0, s, dlog, w, mask, hact, dz, B, H, NC,
                     1.f / (1.f - dropout_p));
  ECG_CHECK_LAUNCH("fc_dgrad_drop_relu");
  return 0;
}

// y = act(x W^T + b), dx = dy W, dW = dy^T x  for [B][In] x [Out][In], every size a multiple of 16 and 16-B aligned bases
bool ecg_dense16_ok(const void* a, const void* b, const void* c, int B, int In, int Out) {
  return al16(a) && al16(b) && al16(c) && B % 16 == 0 && In % 16 == 0 && Out % 16 == 0;
}
int ecg_dense16_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, int act,
                    hipStream_t s) {
  const int tiles = (B / 16) * (Out / 16);
  hipLaunchKernelGGL((dense16_kernel<true, true>), dim3((tiles + 3) / 4), dim3(256), 0, s, x, w, bias, y, B, Out, In, In,
                     In, Out, act);
  ECG_CHECK_LAUNCH("dense16_fwd");
  return 0;
}
int ecg_dense16_dgrad(const float* dy, const float* w, float* dx, int B, int In, int Out, hipStream_t s) {
  const int tiles = (B / 16) * (In / 16);
  hipLaunchKernelGGL((dense16_kernel<true, false>), dim3((tiles + 3) / 4), dim3(256), 0, s, dy, w, (const float*)nullptr,
                     dx, B, In, Out, Out, In, In, ECGMM_ACT_NONE);
  ECG_CHECK_LAUNCH("dense16_dgrad");
  return 0;
}
int ecg_dense16_wgrad(const float* dy, const float* x, float* dw, int B, int In, int Out, hipStream_t s) {
  const int tiles = (Out / 16) * (In / 16);
  hipLaunchKernelGGL((dense16_kernel<false, false>), dim3((tiles + 3) / 4), dim3(256), 0, s, dy, x,
                     (const float*)nullptr, dw, Out, In, B, Out, In, In, ECGMM_ACT_NONE);
  ECG_CHECK_LAUNCH("dense16_wgrad");
  return 0;
}
