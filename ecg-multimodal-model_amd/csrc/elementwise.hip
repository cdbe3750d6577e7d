// HBM-bound passes over channels-last activations [M pixels][C channels] (gfx950):
// BatchNorm finalize / apply (+ReLU, +residual, +SE gate), BatchNorm backward (reduce + apply),
// 3x3/s2 max-pool (fused with BN+ReLU) and its backward, global average pool, layout packs.
// All tensor traffic is 16 B per lane; each thread keeps a FIXED channel chunk (so per-channel
// coefficients live in registers) and walks rows.
#include "ops.h"

namespace {

constexpr int EW_THREADS = 256;

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
// Tensors a pass streams through ONCE and nobody reads again soon (the gradient and the saved conv output in the BatchNorm
// backward passes, the conv output in the forward activation pass): NON-TEMPORAL loads, so that they do not push the tensors the
// next convolution is about to read out of L2 / the Infinity Cache.  Stand-alone (tools/ew_bench.py) the passes gain 0-4 %; in the
// step, same-call A/B of two builds: 6.66, 6.67 -> 6.56, 6.56 ms.  (-DEW_NO_NT builds the plain-load form.)
__device__ __forceinline__ u32x4 ld16s(const void* p) {
#ifndef EW_NO_NT
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#else
  return *reinterpret_cast<const u32x4*>(p);
#endif
}
// (NOT the pooling passes: their 3x3 / stride-2 windows re-read every row from neighbouring threads -- with non-temporal loads
//  there the step got 0.1 ms SLOWER)
// row -> sample by a magic-number multiply (rows < 2^31): the plain `r / rows_per_sample` on a 64-bit row index is a
// ~100-instruction software division, once or twice per row in the gated passes of the signal encoder
struct RowDiv { unsigned mul, sh; };
__device__ __forceinline__ long row_div(long r, RowDiv d) { return (long)(((unsigned long long)(unsigned)r * d.mul) >> d.sh); }
static RowDiv make_row_div(int d) {
  unsigned dd = d < 1 ? 1u : (unsigned)d;
  int l = 0;
  while ((1u << l) < dd) ++l;
  RowDiv r;
  r.mul = (unsigned)(((1ull << (31 + l)) + dd - 1) / dd);
  r.sh = 31u + (unsigned)l;
  return r;
}
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

// ------------------------------------------------------------------------------------------------
// BatchNorm finalize: partial (sum, sumsq) rows -> mean / invstd / affine coefficients / running stats
// coef layout: [4][C] = scale, shift, mean, invstd
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int rows, int C,
                                                           double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* running_mean,
                                                           float* running_var, long long* nbt, float momentum,
                                                           float eps, float* __restrict__ coef) {
  __shared__ double sh[2][16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;  // 16 row slices
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    int r = slice;
    for (; r + 48 < rows; r += 64) {   // four rows per trip: eight independent loads in flight (same summation order)
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* row = partial + (size_t)(r + 16 * u) * 2 * C;
        a[u] = row[c];
        b[u] = row[C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s1 += (double)a[u];
        s2 += (double)b[u];
      }
    }
    for (; r < rows; r += 16) {
      const float* row = partial + (size_t)r * 2 * C;
      s1 += (double)row[c];
      s2 += (double)row[C + c];
    }
  }
  sh[0][slice][threadIdx.x & 63] = s1;
  sh[1][slice][threadIdx.x & 63] = s2;
  __syncthreads();
  if (slice == 0 && c < C) {
    for (int k = 1; k < 16; ++k) {
      s1 += sh[0][k][threadIdx.x];
      s2 += sh[1][k][threadIdx.x];
    }
    double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    float sc = g * invstd;
    coef[c] = sc;
    coef[C + c] = b - (float)mean * sc;
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = invstd;
    if (running_mean) {
      double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

// stage 1 of a long partial-row reduction: block (cg, b) sums rows [b*chunk, (b+1)*chunk) of NQ*C columns
// into out row b (fp32).  1024 threads = 16 row slices x 64 columns.
__global__ __launch_bounds__(1024) void rows_stage1_kernel(const float* __restrict__ partial, int rows, int width,
                                                           int chunk, float* __restrict__ out) {
  __shared__ double sh[16][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  const int r0 = blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
  double s = 0.0;
  if (col < width) {
    int r = r0 + slice;
    for (; r + 48 < r1; r += 64) {   // four independent loads per trip (same summation order)
      float a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = partial[(size_t)(r + 16 * u) * width + col];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += (double)a[u];
    }
    for (; r < r1; r += 16) s += (double)partial[(size_t)r * width + col];
  }
  sh[slice][threadIdx.x & 63] = s;
  __syncthreads();
  if (slice == 0 && col < width) {
    for (int k = 1; k < 16; ++k) s += sh[k][threadIdx.x];
    out[(size_t)blockIdx.y * width + col] = (float)s;
  }
}

// eval mode: coefficients straight from the running statistics
__global__ void bn_eval_coef_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                    float eps, float* coef) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float invstd = 1.f / sqrtf(rv[c] + eps);
  float sc = (gamma ? gamma[c] : 1.f) * invstd;
  coef[c] = sc;
  coef[C + c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
  coef[2 * C + c] = rm[c];
  coef[3 * C + c] = invstd;
}

// column partial sums of a [M][C] tensor (producer was not a conv epilogue), same row format
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void col_stats_kernel(const T* __restrict__ x, long M, int C,
                                                               float* __restrict__ partial) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float sh[EW_THREADS][2 * VEC + 1];
  const int cpr = C / VEC, rpi = EW_THREADS / cpr;
  const int chunk = threadIdx.x % cpr, r0 = threadIdx.x / cpr;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.f;
  for (long r = (long)blockIdx.x * rpi + r0; r < M; r += (long)gridDim.x * rpi) {
    float f[VEC];
    unpack16<T>(ld16(x + r * C + chunk * VEC), f);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      s1[j] += f[j];
      s2[j] += f[j] * f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sh[threadIdx.x][j] = s1[j];
    sh[threadIdx.x][VEC + j] = s2[j];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += EW_THREADS) {
    int which = o / C, c = o % C;
    int ck = c / VEC, j = c % VEC;
    float s = 0.f;
    for (int k = 0; k < rpi; ++k) s += sh[k * cpr + ck][which * VEC + j];
    partial[(size_t)blockIdx.x * 2 * C + o] = s;
  }
}


// ------------------------------------------------------------------------------------------------
// BatchNorm finalize FOLDED INTO THE CONSUMER (round 3).  The finalize kernels above are ~5 us launches that sit between
// a producer and its consumer on the critical path, ~60 of them per step (+ a dependent-launch boundary each).  Here
// every workgroup of the CONSUMER folds the same <= 512 partial rows itself, in the same fixed order -- every workgroup
// arrives at bit-identical coefficients, so the result does not depend on which workgroup handles which rows -- and
// workgroup 0 publishes them (coef / running statistics / dgamma, dbeta) for the later passes that read them from memory.
// Channels are handled in groups of G = min(C, 128): thread = (channel of the group, row slice), fp64 sums.
// ------------------------------------------------------------------------------------------------
struct BnFin {               // forward: partial rows of (sum y, sum y^2) -> scale, shift, mean, invstd
  const float* partial;      // [rows][2][C]; null = no fold (coefficients come from memory)
  int rows;
  double count;
  const float* gamma;
  const float* beta;
  float* rm;                 // running statistics (nullable)
  float* rv;
  long long* nbt;
  float momentum, eps;
  float* coef_out;           // [4][C], written by workgroup 0
};
struct BnBwdFin {            // backward: partial rows of (sum dz, sum dz * xhat | sum dz * (y - mean)) -> k1, k2, k3
  const float* partial;      // [rows][2][C]; null = no fold
  int rows;
  double count;
  const float* gamma;        // nullable (1)
  float* dgamma;             // nullable, written by workgroup 0
  float* dbeta;
  int centered;              // rows hold sum dz * (y - mean): scale by invstd
};
constexpr int FIN_LDS_DOUBLES = 2 * 1024;   // [2][slices][G] with slices * G == block size (1024 threads)

// sums of the two columns of channel group [g0, g0 + G) over all rows -> (s1, s2) in the first G threads
template <int THREADS>
__device__ __forceinline__ void fin_fold_group(const float* __restrict__ partial, int rows, int C, int g0, int G, double* s_red,
                                               double& s1, double& s2) {
  const int SL = THREADS / G;
  const int cl = threadIdx.x % G, slice = threadIdx.x / G;
  const int c = g0 + cl;
  s1 = 0.0; s2 = 0.0;
  int r = slice;
  for (; r + 3 * SL < rows; r += 4 * SL) {   // four rows per trip: eight independent loads in flight
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* row = partial + (size_t)(r + SL * u) * 2 * C;
      a[u] = row[c];
      b[u] = row[C + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s1 += (double)a[u];
      s2 += (double)b[u];
    }
  }
  for (; r < rows; r += SL) {
    const float* row = partial + (size_t)r * 2 * C;
    s1 += (double)row[c];
    s2 += (double)row[C + c];
  }
  s_red[slice * G + cl] = s1;
  s_red[THREADS + slice * G + cl] = s2;
  __syncthreads();
  if (slice == 0) {
    for (int k = 1; k < SL; ++k) {
      s1 += s_red[k * G + cl];
      s2 += s_red[THREADS + k * G + cl];
    }
  }
}

// forward fold: s_coef = [2][C] (scale, shift) in LDS for this workgroup; workgroup 0 writes coef_out / running statistics
template <int THREADS>
__device__ __forceinline__ void bn_fin_prologue(const BnFin& f, int C, float* s_coef, double* s_red) {
  const int G = C < 128 ? C : 128;
  for (int g0 = 0; g0 < C; g0 += G) {
    double s1, s2;
    fin_fold_group<THREADS>(f.partial, f.rows, C, g0, G, s_red, s1, s2);
    if ((int)threadIdx.x < G) {
      const int c = g0 + threadIdx.x;
      const double mean = s1 / f.count;
      double var = s2 / f.count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
      const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
      const float sc = g * invstd;
      const float sh = b - (float)mean * sc;
      s_coef[c] = sc;
      s_coef[C + c] = sh;
      if (blockIdx.x == 0) {
        f.coef_out[c] = sc;
        f.coef_out[C + c] = sh;
        f.coef_out[2 * C + c] = (float)mean;
        f.coef_out[3 * C + c] = invstd;
        if (f.rm) {
          const double unb = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
          f.rm[c] = (1.f - f.momentum) * f.rm[c] + f.momentum * (float)mean;
          f.rv[c] = (1.f - f.momentum) * f.rv[c] + f.momentum * (float)unb;
        }
      }
    }
    __syncthreads();   // s_red is reused by the next group; s_coef complete after the last one
  }
  if (f.nbt && blockIdx.x == 0 && threadIdx.x == 0) *f.nbt += 1;
}

// backward fold: s_bcoef = [3][C] (gamma * invstd, mean dz, mean dz * xhat) in LDS; workgroup 0 writes dgamma / dbeta
template <int THREADS>
__device__ __forceinline__ void bn_bwd_fin_prologue(const BnBwdFin& f, const float* __restrict__ coef, int C, float* s_bcoef,
                                                    double* s_red) {
  const int G = C < 128 ? C : 128;
  for (int g0 = 0; g0 < C; g0 += G) {
    double s1, s2;
    fin_fold_group<THREADS>(f.partial, f.rows, C, g0, G, s_red, s1, s2);
    if ((int)threadIdx.x < G) {
      const int c = g0 + threadIdx.x;
      const float inv = coef[3 * C + c];
      if (f.centered) s2 *= (double)inv;
      s_bcoef[c] = (f.gamma ? f.gamma[c] : 1.f) * inv;
      s_bcoef[C + c] = (float)(s1 / f.count);
      s_bcoef[2 * C + c] = (float)(s2 / f.count);
      if (blockIdx.x == 0) {
        if (f.dgamma) f.dgamma[c] = (float)s2;
        if (f.dbeta) f.dbeta[c] = (float)s1;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// BN apply: out = relu?( (y*scale+shift) * gate[n][c] + residual ), residual optionally affine
// ------------------------------------------------------------------------------------------------
struct BnActParams {
  const void* y;
  const float* coef;       // [4][C]
  const void* res;         // nullable
  const float* rcoef;      // nullable: residual = res*rscale + rshift
  const float* gate;       // nullable [N][C]
  void* out;
  long M;
  int C, rows_per_sample, relu;
  BnFin fin;               // fin.partial != null: the coefficients are folded from the producer's partial rows here
  RowDiv rdiv;             // rows_per_sample as a magic multiplier
  unsigned char* relu_bits;  // nullable (bf16, relu): bit j of byte [row][chunk] = (out[row][8 chunk + j] > 0) -- the ReLU mask the
                             // BatchNorm backward of a residual block needs, at 1/16 of the activated tensor's bytes
};

#ifndef BN_ACT_THREADS
#define BN_ACT_THREADS 1024
#endif
template <typename T>
__global__ __launch_bounds__(BN_ACT_THREADS) void bn_act_kernel(BnActParams p) {
  constexpr int VEC = Elem<T>::VEC;
  const int cpr = p.C / VEC, rpi = BN_ACT_THREADS / cpr;
  const int chunk = threadIdx.x % cpr, r0 = threadIdx.x / cpr;
  const int c0 = chunk * VEC;
  __shared__ double s_red[FIN_LDS_DOUBLES];
  __shared__ float s_coef[2 * 512];
  const float* cf = p.coef;
  int cfs = p.C;           // stride between the scale and the shift rows
  // (fold form: the first trip's loads are issued BEFORE the fold of the partial rows, whose ~3 us then hide their HBM latency)
  const T* y = (const T*)p.y;
  const T* res = (const T*)p.res;
  const long stride = (long)gridDim.x * rpi;
  const long rfirst = (long)blockIdx.x * rpi + r0;
  u32x4 py[2] = {}, pr[2] = {};
  bool pre = p.fin.partial != nullptr && rfirst < p.M;
  if (pre) {
    const long q2 = rfirst + stride < p.M ? rfirst + stride : rfirst;
    py[0] = ld16s(y + rfirst * p.C + c0);
    py[1] = ld16s(y + q2 * p.C + c0);
    if (res) {
      pr[0] = ld16s(res + rfirst * p.C + c0);
      pr[1] = ld16s(res + q2 * p.C + c0);
    }
  }
  if (p.fin.partial) {
    bn_fin_prologue<BN_ACT_THREADS>(p.fin, p.C, s_coef, s_red);
    cf = s_coef;
  }
  float sc[VEC], sh[VEC], rs[VEC], rb[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = cf[c0 + j];
    sh[j] = cf[cfs + c0 + j];
    rs[j] = p.rcoef ? p.rcoef[c0 + j] : 1.f;
    rb[j] = p.rcoef ? p.rcoef[p.C + c0 + j] : 0.f;
  }
  T* out = (T*)p.out;
  // two rows per iteration, all (up to four) 16-B loads issued before the first use
  for (long r = rfirst; r < p.M; r += 2 * stride) {
    const long r2 = r + stride;
    const bool two = r2 < p.M;
    const long rb2 = two ? r2 : r;
    u32x4 vy[2], vr[2];
    if (pre) {
      vy[0] = py[0]; vy[1] = py[1]; vr[0] = pr[0]; vr[1] = pr[1];
      pre = false;
    } else {
      vy[0] = ld16s(y + r * p.C + c0);
      vy[1] = ld16s(y + rb2 * p.C + c0);
      if (res) {
        vr[0] = ld16s(res + r * p.C + c0);
        vr[1] = ld16s(res + rb2 * p.C + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long rr = u ? rb2 : r;
      float f[VEC], g[VEC];
      unpack16<T>(vy[u], f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = f[j] * sc[j] + sh[j];
      if (p.gate) {
        const float* gp = p.gate + row_div(rr, p.rdiv) * p.C + c0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] *= gp[j];
      }
      if (res) {
        unpack16<T>(vr[u], g);
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] += g[j] * rs[j] + rb[j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] = fmaxf(f[j], 0.f);
      }
      if (VEC == 8 && p.relu_bits && (u == 0 || two)) {
        // (> 0 of the value as STORED: a positive fp32 value below the smallest bf16 rounds to 0 and is masked out by the
        //  backward that compares the stored tensor, too)
        float g[VEC];
        unpack16<T>(pack16<T>(f), g);
        unsigned bits = 0u;
#pragma unroll
        for (int j = 0; j < VEC; ++j) bits |= (g[j] > 0.f ? 1u : 0u) << j;
        p.relu_bits[rr * cpr + chunk] = (unsigned char)bits;
      }
      if (u == 0 || two) st16(out + rr * p.C + c0, pack16<T>(f));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// BN backward.  dz = [maskref > 0] * dout * gate[n][c] + addc[n][c]   (each factor optional)
//   reduce : partial rows of (sum dz, sum dz*xhat), xhat = (y - mean) * invstd
//   finalize: dgamma, dbeta, bcoef = [3][C] (gamma*invstd, mean_dz, mean_dz_xhat)
//   apply  : dy = c1 * (dz - c2 - xhat * c3); optional dz store; optional partial sum of dy (conv bias grad)
// ------------------------------------------------------------------------------------------------
struct BnBwdParams {
  const void* dout;
  const void* maskref;   // nullable
  const float* gate;     // nullable [N][C]
  const float* addc;     // nullable [N][C]
  const void* y;         // raw conv output the BN normalised
  const float* coef;     // forward coef [4][C] (mean at 2C, invstd at 3C)
  const float* bcoef;    // [3][C] (apply only)
  void* dy;              // apply only
  void* dz_out;          // nullable (apply only)
  float* partial;        // reduce: [grid][2][C]; apply: nullable [grid][C] partial sum(dy)
  long M;
  int C, rows_per_sample;
  BnBwdFin fin;          // apply only; fin.partial != null: bcoef is folded from the reduction's partial rows here
  RowDiv rdiv;           // rows_per_sample as a magic multiplier
  const unsigned char* mask_bits;   // nullable (bf16): the ReLU mask as bits (BnActParams::relu_bits) instead of maskref
};

// 1024-thread blocks: at most 256 partial rows per launch, which the finalize / bias-sum kernels fold themselves (a
// separate fold launch per BatchNorm cost ~6 us of dependency latency, 29 times per step), at the same threads in flight.
constexpr int BWD_THREADS = 1024;
template <typename T, bool APPLY>
__global__ __launch_bounds__(BWD_THREADS) void bn_bwd_kernel(BnBwdParams p) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ float shm_dyn[];
  float(*shm)[2 * VEC + 1] = reinterpret_cast<float(*)[2 * VEC + 1]>(shm_dyn);  // [BWD_THREADS][2*VEC+1]
  const int cpr = p.C / VEC, rpi = BWD_THREADS / cpr;
  const int chunk = threadIdx.x % cpr, r0 = threadIdx.x / cpr;
  const int c0 = chunk * VEC;
  __shared__ double s_red[APPLY ? FIN_LDS_DOUBLES : 1];
  __shared__ float s_bcoef[APPLY ? 3 * 512 : 1];
  const float* bc = p.bcoef;
  // (fold form: the first row's loads are issued before the fold, as in bn_act_kernel)
  const long rfirst = (long)blockIdx.x * rpi + r0;
  u32x4 pd = {}, pv = {};
  bool pre = APPLY && p.fin.partial != nullptr && rfirst < p.M;
  if (pre) {
    pd = ld16s((const T*)p.dout + rfirst * p.C + c0);
    pv = ld16s((const T*)p.y + rfirst * p.C + c0);
  }
  if (APPLY && p.fin.partial) {
    bn_bwd_fin_prologue<BWD_THREADS>(p.fin, p.coef, p.C, s_bcoef, s_red);
    bc = s_bcoef;
  }
  float mean[VEC], inv[VEC], k1[VEC], k2[VEC], k3[VEC], a1[VEC], a2[VEC], msc[VEC], msh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    msc[j] = p.coef[c0 + j];
    msh[j] = p.coef[p.C + c0 + j];
    mean[j] = p.coef[2 * p.C + c0 + j];
    inv[j] = p.coef[3 * p.C + c0 + j];
    a1[j] = a2[j] = 0.f;
    if (APPLY) {
      k1[j] = bc[c0 + j];
      k2[j] = bc[p.C + c0 + j];
      k3[j] = bc[2 * p.C + c0 + j];
    }
  }
  const T* dout = (const T*)p.dout;
  const T* mref = (const T*)p.maskref;
  const T* y = (const T*)p.y;
  for (long r = rfirst; r < p.M; r += (long)gridDim.x * rpi) {
    float d[VEC], m[VEC], v[VEC];
    if (pre) {
      unpack16<T>(pd, d);
      unpack16<T>(pv, v);
      pre = false;
    } else {
      unpack16<T>(ld16s(dout + r * p.C + c0), d);
      unpack16<T>(ld16s(y + r * p.C + c0), v);
    }
    if (mref == y) {
      // maskref aliasing y: the ReLU input was bn(y) itself, so the mask is recomputed through the
      // BatchNorm affine instead of reading the activated tensor (one tensor read less)
#pragma unroll
      for (int j = 0; j < VEC; ++j) d[j] = (v[j] * msc[j] + msh[j]) > 0.f ? d[j] : 0.f;
    } else if (VEC == 8 && p.mask_bits) {
      const unsigned bits = p.mask_bits[r * (p.C / VEC) + c0 / VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) d[j] = (bits >> j) & 1u ? d[j] : 0.f;
    } else if (mref) {
      unpack16<T>(ld16s(mref + r * p.C + c0), m);
#pragma unroll
      for (int j = 0; j < VEC; ++j) d[j] = m[j] > 0.f ? d[j] : 0.f;
    }
    // dz_out carries the masked gradient only (the residual-branch gradient), before gate / addc.  With a separate
    // mask tensor the REDUCE pass stores it and the apply pass reads it back as its `dout` (host side below): one
    // tensor read less than re-reading dout and the mask tensor; the values are identical (a masked copy of dout)
    if (p.dz_out) st16((T*)p.dz_out + r * p.C + c0, pack16<T>(d));
    if (p.gate) {
      const float* gp = p.gate + row_div(r, p.rdiv) * p.C + c0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) d[j] *= gp[j];
    }
    if (p.addc) {
      const float* ap = p.addc + row_div(r, p.rdiv) * p.C + c0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) d[j] += ap[j];
    }
    if (!APPLY) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float xh = (v[j] - mean[j]) * inv[j];
        a1[j] += d[j];
        a2[j] += d[j] * xh;
      }
    } else {
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float xh = (v[j] - mean[j]) * inv[j];
        o[j] = k1[j] * (d[j] - k2[j] - xh * k3[j]);
      }
      u32x4 pk = pack16<T>(o);
      st16((T*)p.dy + r * p.C + c0, pk);
      if (p.partial) {
        float back[VEC];
        unpack16<T>(pk, back);  // sum what was actually stored
#pragma unroll
        for (int j = 0; j < VEC; ++j) a1[j] += back[j];
      }
    }
  }
  if (!APPLY || p.partial) {
    constexpr int NQ = APPLY ? 1 : 2;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      shm[threadIdx.x][j] = a1[j];
      if (!APPLY) shm[threadIdx.x][VEC + j] = a2[j];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < NQ * p.C; o += BWD_THREADS) {
      int which = o / p.C, c = o % p.C;
      int ck = c / VEC, j = c % VEC;
      float s = 0.f;
      for (int k = 0; k < rpi; ++k) s += shm[k * cpr + ck][which * VEC + j];
      p.partial[(size_t)blockIdx.x * NQ * p.C + o] = s;
    }
  }
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int rows, int C,
                                                               double count, const float* __restrict__ gamma,
                                                               const float* __restrict__ coef, float* dgamma,
                                                               float* dbeta, float* __restrict__ bcoef, int centered) {
  __shared__ double sh[2][16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    int r = slice;
    for (; r + 48 < rows; r += 64) {   // (as bn_finalize_kernel: four rows per trip)
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* row = partial + (size_t)(r + 16 * u) * 2 * C;
        a[u] = row[c];
        b[u] = row[C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s1 += (double)a[u];
        s2 += (double)b[u];
      }
    }
    for (; r < rows; r += 16) {
      const float* row = partial + (size_t)r * 2 * C;
      s1 += (double)row[c];
      s2 += (double)row[C + c];
    }
  }
  sh[0][slice][threadIdx.x & 63] = s1;
  sh[1][slice][threadIdx.x & 63] = s2;
  __syncthreads();
  if (slice == 0 && c < C) {
    for (int k = 1; k < 16; ++k) {
      s1 += sh[0][k][threadIdx.x];
      s2 += sh[1][k][threadIdx.x];
    }
    // centered: the rows hold sum g * (y - mean) (reduction fused into a conv epilogue), not sum g * xhat
    if (centered) s2 *= (double)coef[3 * C + c];
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    bcoef[c] = (gamma ? gamma[c] : 1.f) * coef[3 * C + c];
    bcoef[C + c] = (float)(s1 / count);
    bcoef[2 * C + c] = (float)(s2 / count);
  }
}

// sum rows of a [rows][C] fp32 partial buffer -> out[C] (conv bias grads, linear bias grads)
__global__ __launch_bounds__(1024) void rows_sum_kernel(const float* __restrict__ partial, int rows, int C, float* out,
                                                        int accumulate) {
  __shared__ double sh[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  double s = 0.0;
  if (c < C) {
    int r = slice;
    for (; r + 48 < rows; r += 64) {   // four independent loads per trip (same summation order)
      float a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = partial[(size_t)(r + 16 * u) * C + c];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += (double)a[u];
    }
    for (; r < rows; r += 16) s += (double)partial[(size_t)r * C + c];
  }
  sh[slice][threadIdx.x & 63] = s;
  __syncthreads();
  if (slice == 0 && c < C) {
    for (int k = 1; k < 16; ++k) s += sh[k][threadIdx.x];
    out[c] = accumulate ? out[c] + (float)s : (float)s;
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 2 / pad 1 max-pool fused with BN apply + ReLU (stem of both encoders; H may be 1)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void bnrelu_maxpool_kernel(const T* __restrict__ y,
                                                                    const float* __restrict__ coef, T* __restrict__ out,
                                                                    unsigned char* __restrict__ idx, int N, int H,
                                                                    int W, int C, int OH, int OW) {
  constexpr int VEC = Elem<T>::VEC;
  const int cpr = C / VEC;
  long total = (long)N * OH * OW * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int chunk = (int)(i % cpr);
    long pix = i / cpr;
    int ow = (int)(pix % OW);
    long t = pix / OW;
    int oh = (int)(t % OH), n = (int)(t / OH);
    int c0 = chunk * VEC;
    float best[VEC];
    int bi[VEC];
    if constexpr (sizeof(T) == 2) {
      // bf16: post-ReLU values are >= 0, so their bf16 bit patterns order like the values.  One sortable key per
      // channel, (bits << 4) | (15 - tap): a single unsigned max implements "largest value, earliest tap on ties".
      unsigned key[VEC];
      float scv[VEC], shv[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        key[j] = 0u;
        scv[j] = coef[c0 + j];
        shv[j] = coef[C + c0 + j];
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        int h = oh * 2 - 1 + kh;
        if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          int w = ow * 2 - 1 + kw;
          if ((unsigned)w >= (unsigned)W) continue;
          float f[VEC];
          unpack16<T>(ld16(y + (((size_t)n * H + h) * W + w) * C + c0), f);
#pragma unroll
          for (int j = 0; j < VEC; ++j) f[j] = fmaxf(f[j] * scv[j] + shv[j], 0.f);
          const u32x4 r = pack16<bf16_t>(f);  // rounded as stored; (-0 cannot win: sign bit masked off)
          const unsigned tcode = 15u - (unsigned)(kh * 3 + kw);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            key[2 * q] = max(key[2 * q], ((r[q] & 0x7fffu) << 4) | tcode);
            key[2 * q + 1] = max(key[2 * q + 1], (((r[q] >> 16) & 0x7fffu) << 4) | tcode);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        best[j] = __uint_as_float((key[j] >> 4) << 16);
        bi[j] = 15 - (int)(key[j] & 15u);
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        best[j] = -INFINITY;
        bi[j] = 0;
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        int h = oh * 2 - 1 + kh;
        if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          int w = ow * 2 - 1 + kw;
          if ((unsigned)w >= (unsigned)W) continue;
          float f[VEC];
          unpack16<T>(ld16(y + (((size_t)n * H + h) * W + w) * C + c0), f);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            float a = fmaxf(f[j] * coef[c0 + j] + coef[C + c0 + j], 0.f);
            if (a > best[j]) {
              best[j] = a;
              bi[j] = kh * 3 + kw;
            }
          }
        }
      }
    }
    st16(out + pix * C + c0, pack16<T>(best));
    // the VEC argmax bytes of this chunk go out as one 8-B (bf16) / 4-B (f32) store
    unsigned pk[VEC / 4];
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q)
      pk[q] = (unsigned)bi[4 * q] | ((unsigned)bi[4 * q + 1] << 8) | ((unsigned)bi[4 * q + 2] << 16) |
              ((unsigned)bi[4 * q + 3] << 24);
    unsigned* ip = reinterpret_cast<unsigned*>(idx + pix * C + c0);
    if constexpr (VEC == 8) *reinterpret_cast<uint2*>(ip) = make_uint2(pk[0], pk[1]);
    else ip[0] = pk[0];
  }
}

// dz[n,h,w,c] = sum over the <=4 windows containing (h,w): dp[win] * [idx[win]==tap] * [p[win] > 0]
// One thread = a 2x2 block of input pixels (h = 2k, 2k+1; w = 2m, 2m+1) x one 16-B channel chunk.  The block
// touches exactly the windows (k, k+1) x (m, m+1), each loaded once (dp, pooled, argmax bytes) with all loads
// issued up front; which tap of which window feeds which pixel is a compile-time table:
//   (2k,2m): w00 tap 4 | (2k,2m+1): w00 tap 5, w01 tap 3 | (2k+1,2m): w00 tap 7, w10 tap 1
//   (2k+1,2m+1): w00 tap 8, w01 tap 6, w10 tap 2, w11 tap 0          (w_ab = window (k+a, m+b))
// (a thread per pixel re-loaded every window 2.25x on average and diverged on the 1/2/4-window cases)
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void maxpool_relu_bwd_kernel(const T* __restrict__ dp,
                                                                      const T* __restrict__ pooled,
                                                                      const unsigned char* __restrict__ idx,
                                                                      T* __restrict__ dz, int N, int H, int W, int C,
                                                                      int OH, int OW) {
  constexpr int VEC = Elem<T>::VEC;
  const int cpr = C / VEC;
  const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
  long total = (long)N * H2 * W2 * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int chunk = (int)(i % cpr);
    long blk = i / cpr;
    int m = (int)(blk % W2);
    long t = blk / W2;
    int k = (int)(t % H2), n = (int)(t / H2);
    int c0 = chunk * VEC;
    float d[4][VEC], pv[4][VEC];
    unsigned pk[4][2];
    bool wok[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int q = a * 2 + b;
        wok[q] = k + a < OH && m + b < OW;
        const size_t o = (((size_t)n * OH + (wok[q] ? k + a : k)) * OW + (wok[q] ? m + b : m)) * C + c0;
        unpack16<T>(ld16(dp + o), d[q]);
        unpack16<T>(ld16(pooled + o), pv[q]);
        if constexpr (VEC == 8) {
          const uint2 t2 = *reinterpret_cast<const uint2*>(idx + o);
          pk[q][0] = t2.x;
          pk[q][1] = t2.y;
        } else {
          pk[q][0] = *reinterpret_cast<const unsigned*>(idx + o);
          pk[q][1] = 0;
        }
      }
    // contribution of window q through tap `tap`, channel j
    auto g = [&](int q, int tap, int j) -> float {
      const int code = (int)((pk[q][j >> 2] >> (8 * (j & 3))) & 0xffu);
      return (wok[q] && code == tap && pv[q][j] > 0.f) ? d[q][j] : 0.f;
    };
    float o00[VEC], o01[VEC], o10[VEC], o11[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      o00[j] = g(0, 4, j);
      o01[j] = g(0, 5, j) + g(1, 3, j);
      o10[j] = g(0, 7, j) + g(2, 1, j);
      o11[j] = ((g(0, 8, j) + g(1, 6, j)) + g(2, 2, j)) + g(3, 0, j);
    }
    const int h0 = 2 * k, w0 = 2 * m;
    T* base = dz + (((size_t)n * H + h0) * W + w0) * C + c0;
    st16(base, pack16<T>(o00));
    if (w0 + 1 < W) st16(base + C, pack16<T>(o01));
    if (h0 + 1 < H) {
      st16(base + (size_t)W * C, pack16<T>(o10));
      if (w0 + 1 < W) st16(base + (size_t)W * C + C, pack16<T>(o11));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Stem backward without the full-resolution intermediate: BatchNorm-backward reduction over the POOLED tensors, and
// max-pool backward + BatchNorm-backward apply in one pass.
//   The max-pool routes each window's gradient to one input position, and there bn(y) equals the pooled value p (> 0;
//   a window with p = 0 passes nothing through the ReLU).  So with g = dp * [p > 0] summed over WINDOWS:
//     sum_pixels dz         = sum g
//     sum_pixels dz (y - m) = sum g * (p - beta) / (gamma * invstd)        (p = gamma * xhat + beta at the arg-max)
//   which reads 2 pooled tensors (1/4 size) instead of dz and y at full resolution, and never needs dz in memory.
//   (bf16: p is stored rounded, so xhat recovered from it carries p's rounding, 2^-9 relative, instead of y's.)
//   A channel with gamma * invstd == 0 has no recoverable xhat: its dgamma row is 0 (torch would give sum dz * xhat;
//   such a channel outputs the constant beta).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BWD_THREADS) void pool_bwd_reduce_kernel(const T* __restrict__ dp,
                                                                      const T* __restrict__ pooled,
                                                                      const float* __restrict__ coef,
                                                                      float* __restrict__ partial, long MP, int C) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ float shm_dyn[];
  float(*shm)[2 * VEC + 1] = reinterpret_cast<float(*)[2 * VEC + 1]>(shm_dyn);
  const int cpr = C / VEC, rpi = BWD_THREADS / cpr;
  const int chunk = threadIdx.x % cpr, r0 = threadIdx.x / cpr;
  const int c0 = chunk * VEC;
  float beta[VEC], rsc[VEC], a1[VEC], a2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const float sc = coef[c0 + j];                               // gamma * invstd
    beta[j] = coef[C + c0 + j] + coef[2 * C + c0 + j] * sc;      // shift + mean * scale
    rsc[j] = sc != 0.f ? 1.f / sc : 0.f;
    a1[j] = a2[j] = 0.f;
  }
  for (long r = (long)blockIdx.x * rpi + r0; r < MP; r += (long)gridDim.x * rpi) {
    float d[VEC], v[VEC];
    unpack16<T>(ld16(dp + r * C + c0), d);
    unpack16<T>(ld16(pooled + r * C + c0), v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float g = v[j] > 0.f ? d[j] : 0.f;
      a1[j] += g;
      a2[j] += g * (v[j] - beta[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    shm[threadIdx.x][j] = a1[j];
    shm[threadIdx.x][VEC + j] = a2[j] * rsc[j];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += BWD_THREADS) {
    int which = o / C, c = o % C;
    int ck = c / VEC, j = c % VEC;
    float sum = 0.f;
    for (int k = 0; k < rpi; ++k) sum += shm[k * cpr + ck][which * VEC + j];
    partial[(size_t)blockIdx.x * 2 * C + o] = sum;
  }
}

// dy = k1 * (dz - k2 - xhat * k3) with dz gathered on the fly from the pooled gradient (same 2x2-pixel-block scheme and
// tap table as maxpool_relu_bwd_kernel above); optional partial rows of sum(dy as stored) for a conv bias gradient.
// Written as dy = k1 * dz + (bn * y + an) with an = k1 * (k3 * invstd * mean - k2), bn = -k1 * k3 * invstd: 3 constants
// per channel.  512-thread blocks: the four windows + four pixels in flight per thread need ~150 VGPRs.
constexpr int POOL_THREADS = 512;
template <typename T>
__global__ __launch_bounds__(POOL_THREADS) void pool_bn_bwd_apply_kernel(const T* __restrict__ dp,
                                                                         const T* __restrict__ pooled,
                                                                         const unsigned char* __restrict__ idx,
                                                                         const T* __restrict__ y,
                                                                         const float* __restrict__ coef,
                                                                         const float* __restrict__ bcoef,
                                                                         T* __restrict__ dy, float* __restrict__ partial,
                                                                         int N, int H, int W, int C, int OH, int OW) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ float shm_dyn[];
  float(*shm)[VEC + 1] = reinterpret_cast<float(*)[VEC + 1]>(shm_dyn);
  const int cpr = C / VEC;
  const int chunk = threadIdx.x % cpr;
  const int c0 = chunk * VEC;
  const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
  const long nblk = (long)N * H2 * W2;
  const int bpi = POOL_THREADS / cpr;
  float k1[VEC], an[VEC], bn[VEC], a1[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const float mean = coef[2 * C + c0 + j], inv = coef[3 * C + c0 + j];
    k1[j] = bcoef[c0 + j];
    bn[j] = -k1[j] * bcoef[2 * C + c0 + j] * inv;
    an[j] = -k1[j] * bcoef[C + c0 + j] - bn[j] * mean;
    a1[j] = 0.f;
  }
  for (long blk = (long)blockIdx.x * bpi + threadIdx.x / cpr; blk < nblk; blk += (long)gridDim.x * bpi) {
    int m = (int)(blk % W2);
    long t = blk / W2;
    int k = (int)(t % H2), n = (int)(t / H2);
    const int h0 = 2 * k, w0 = 2 * m;
    const bool okw = w0 + 1 < W, okh = h0 + 1 < H;
    const size_t base = (((size_t)n * H + h0) * W + w0) * C + c0;
    // all loads up front: 4 windows x (dp, pooled, arg-max bytes) + 4 pixels of y
    u32x4 rd[4], rp[4], ry[4];
    unsigned pk[4][2];
    bool wok[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int q = a * 2 + b;
        wok[q] = k + a < OH && m + b < OW;
        const size_t o = (((size_t)n * OH + (wok[q] ? k + a : k)) * OW + (wok[q] ? m + b : m)) * C + c0;
        rd[q] = ld16(dp + o);
        rp[q] = ld16(pooled + o);
        if constexpr (VEC == 8) {
          const uint2 t2 = *reinterpret_cast<const uint2*>(idx + o);
          pk[q][0] = t2.x;
          pk[q][1] = t2.y;
        } else {
          pk[q][0] = *reinterpret_cast<const unsigned*>(idx + o);
          pk[q][1] = 0;
        }
      }
    ry[0] = ld16(y + base);
    ry[1] = ld16(y + (okw ? base + C : base));
    ry[2] = ld16(y + (okh ? base + (size_t)W * C : base));
    ry[3] = ld16(y + (okh ? base + (size_t)W * C : base) + (okw ? C : 0));
    float d[4][VEC];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float pv[VEC];
      unpack16<T>(rd[q], d[q]);
      unpack16<T>(rp[q], pv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) d[q][j] = (wok[q] && pv[j] > 0.f) ? d[q][j] : 0.f;
    }
    auto g = [&](int q, int tap, int j) -> float {
      const int code = (int)((pk[q][j >> 2] >> (8 * (j & 3))) & 0xffu);
      return code == tap ? d[q][j] : 0.f;
    };
    float o[4][VEC];
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float v[VEC];
      unpack16<T>(ry[px], v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float z;
        if (px == 0) z = g(0, 4, j);
        else if (px == 1) z = g(0, 5, j) + g(1, 3, j);
        else if (px == 2) z = g(0, 7, j) + g(2, 1, j);
        else z = ((g(0, 8, j) + g(1, 6, j)) + g(2, 2, j)) + g(3, 0, j);
        o[px][j] = k1[j] * z + (bn[j] * v[j] + an[j]);
      }
    }
    const u32x4 p00 = pack16<T>(o[0]), p01 = pack16<T>(o[1]), p10 = pack16<T>(o[2]), p11 = pack16<T>(o[3]);
    st16(dy + base, p00);
    if (okw) st16(dy + base + C, p01);
    if (okh) {
      st16(dy + base + (size_t)W * C, p10);
      if (okw) st16(dy + base + (size_t)W * C + C, p11);
    }
    if (partial) {  // sum what was actually stored
      float b0[VEC], b1[VEC], b2[VEC], b3[VEC];
      unpack16<T>(p00, b0); unpack16<T>(p01, b1); unpack16<T>(p10, b2); unpack16<T>(p11, b3);
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        a1[j] += b0[j] + (okw ? b1[j] : 0.f) + (okh ? b2[j] : 0.f) + (okh && okw ? b3[j] : 0.f);
    }
  }
  if (partial) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) shm[threadIdx.x][j] = a1[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += POOL_THREADS) {
      int ck = c / VEC, j = c % VEC;
      float sum = 0.f;
      for (int k = 0; k < bpi; ++k) sum += shm[k * cpr + ck][j];
      partial[(size_t)blockIdx.x * C + c] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// global average pool over the rows of each sample: [N][R][C] T -> [N][C] f32; and its broadcast
// ------------------------------------------------------------------------------------------------
// One block per sample; thread = (16-B channel chunk, row slice); LDS fold over the slices.
// SEGATE = false: out[n][c] = mean_r x[n,r,c] (optionally through the per-channel affine coef)
// SEGATE = true : out[n][c] = sum_r [maskref > 0] * dout * (y * scale + shift)   (x = dout)
template <typename T, bool SEGATE>
__global__ __launch_bounds__(EW_THREADS) void sample_reduce_kernel(const T* __restrict__ x,
                                                                   const T* __restrict__ maskref,
                                                                   const T* __restrict__ y,
                                                                   const float* __restrict__ coef,
                                                                   float* __restrict__ out, int R, int C) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float sh[EW_THREADS][VEC + 1];
  const int cpr = C / VEC, nsl = EW_THREADS / cpr;
  const int chunk = threadIdx.x % cpr, slice = threadIdx.x / cpr;
  const int n = blockIdx.x, c0 = chunk * VEC;
  float acc[VEC], sc[VEC], sf[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    acc[j] = 0.f;
    sc[j] = (SEGATE || coef) ? coef[c0 + j] : 1.f;
    sf[j] = (SEGATE || coef) ? coef[C + c0 + j] : 0.f;
  }
  int r = slice;
  if (!SEGATE) {   // four rows in flight per thread (one load per iteration ran at 3.3 TB/s)
    for (; r + 3 * nsl < R; r += 4 * nsl) {
      u32x4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = ld16(x + ((size_t)n * R + r + u * nsl) * C + c0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[VEC];
        unpack16<T>(q[u], f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += f[j];
      }
    }
  }
  for (; r < R; r += nsl) {
    const size_t o = ((size_t)n * R + r) * C + c0;
    float f[VEC];
    unpack16<T>(ld16(x + o), f);
    if (SEGATE) {
      float m[VEC], v[VEC];
      unpack16<T>(ld16(maskref + o), m);
      unpack16<T>(ld16(y + o), v);
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if (m[j] > 0.f) acc[j] += f[j] * (v[j] * sc[j] + sf[j]);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) sh[threadIdx.x][j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += EW_THREADS) {
    const int ck = c / VEC, j = c % VEC;
    float s_ = 0.f;
    for (int k = 0; k < nsl; ++k) s_ += sh[k * cpr + ck][j];
    if (!SEGATE) {
      s_ /= (float)R;
      if (coef) s_ = s_ * coef[c] + coef[C + c];  // mean of an affine map = affine map of the mean
    }
    out[(size_t)n * C + c] = s_;
  }
}

// SE-block backward, first pass: ONE read of (dout, out, y) per element yields
//   dz[n,r,c]  = [out > 0] * dout                        (the residual-branch gradient, stored)
//   dg[n,c]    = sum_r dz * (y * scale + shift)          (gradient of the SE gate: se_gate_grad)
//   a1, a2, a3 = sum_r dz, sum_r dz * (y - mean), sum_r (y - mean)
// from which the BatchNorm-backward reduction of d = dz * gate[n,c] + addc[n,c] follows WITHOUT another pass over the
// tensors (se_bn_rows_kernel): sum d = sum_n gate a1 + R addc;  sum d (y - mean) = sum_n gate a2 + addc a3.
// (before: se_gate_grad read the three tensors, then bn_bwd's reduce pass read them again and stored dz)
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void se_gate_bn_kernel(const T* __restrict__ dout, const T* __restrict__ maskref,
                                                                const T* __restrict__ y, const float* __restrict__ coef,
                                                                T* __restrict__ dz, float* __restrict__ dg,
                                                                float* __restrict__ a1o, float* __restrict__ a2o,
                                                                float* __restrict__ a3o, int R, int C) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float sh[EW_THREADS][VEC + 1];
  const int cpr = C / VEC, nsl = EW_THREADS / cpr;
  const int chunk = threadIdx.x % cpr, slice = threadIdx.x / cpr;
  const int n = blockIdx.x, c0 = chunk * VEC;
  float sc[VEC], sf[VEC], mu[VEC], ag[VEC], a1[VEC], a2[VEC], a3[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = coef[c0 + j];
    sf[j] = coef[C + c0 + j];
    mu[j] = coef[2 * C + c0 + j];
    ag[j] = a1[j] = a2[j] = a3[j] = 0.f;
  }
  for (int r = slice; r < R; r += nsl) {
    const size_t o = ((size_t)n * R + r) * C + c0;
    float f[VEC], m[VEC], v[VEC];
    unpack16<T>(ld16(dout + o), f);
    unpack16<T>(ld16(maskref + o), m);
    unpack16<T>(ld16(y + o), v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      f[j] = m[j] > 0.f ? f[j] : 0.f;
      const float yc = v[j] - mu[j];
      ag[j] += f[j] * (v[j] * sc[j] + sf[j]);
      a1[j] += f[j];
      a2[j] += f[j] * yc;
      a3[j] += yc;
    }
    st16(dz + o, pack16<T>(f));
  }
  float* const outs[4] = {dg, a1o, a2o, a3o};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) sh[threadIdx.x][j] = q == 0 ? ag[j] : q == 1 ? a1[j] : q == 2 ? a2[j] : a3[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += EW_THREADS) {
      const int ck = c / VEC, j = c % VEC;
      float s_ = 0.f;
      for (int k = 0; k < nsl; ++k) s_ += sh[k * cpr + ck][j];
      outs[q][(size_t)n * C + c] = s_;
    }
    __syncthreads();
  }
}

// SE_ROWS reduction rows [2][C] (sum d, sum d (y - mean)) of d = dz * gate + addc from the per-sample sums above: row j
// covers the samples n = j (mod SE_ROWS); bn_bwd_finalize_kernel folds the rows.  (One row from one block per 64
// channels was a 128-deep chain of dependent L2 loads: 45 us.)
constexpr int SE_ROWS = 16;
__global__ __launch_bounds__(256) void se_bn_rows_kernel(const float* __restrict__ a1, const float* __restrict__ a2,
                                                         const float* __restrict__ a3, const float* __restrict__ gate,
                                                         const float* __restrict__ addc, int N, int R, int C,
                                                         float* __restrict__ rows) {
  __shared__ float sh[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6, j = blockIdx.y;
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int n = j + SE_ROWS * sl; n < N; n += SE_ROWS * 4) {
      const size_t o = (size_t)n * C + c;
      const float g = gate ? gate[o] : 1.f, ad = addc ? addc[o] : 0.f;
      s1 += g * a1[o] + ad * (float)R;
      s2 += g * a2[o] + ad * a3[o];
    }
  sh[0][sl][threadIdx.x & 63] = s1;
  sh[1][sl][threadIdx.x & 63] = s2;
  __syncthreads();
  if (sl == 0 && c < C) {
    float* row = rows + (size_t)j * 2 * C;
    row[c] = (sh[0][0][threadIdx.x] + sh[0][1][threadIdx.x]) + (sh[0][2][threadIdx.x] + sh[0][3][threadIdx.x]);
    row[C + c] = (sh[1][0][threadIdx.x] + sh[1][1][threadIdx.x]) + (sh[1][2][threadIdx.x] + sh[1][3][threadIdx.x]);
  }
}

template <typename T>
__global__ void bcast_rows_kernel(const float* __restrict__ v, T* __restrict__ out, long total, int R, int C,
                                  float scale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long n = i / ((long)R * C);
    Elem<T>::st(out + i, v[n * C + c] * scale);
  }
}

// the same with 16-byte stores: thread = (16-B channel chunk, row slice), one block per CU walking the rows
template <typename T>
__global__ __launch_bounds__(1024) void bcast_rows_vec_kernel(const float* __restrict__ v, T* __restrict__ out, long M,
                                                              RowDiv rdiv, int C, float scale) {
  constexpr int VEC = Elem<T>::VEC;
  const int cpr = C / VEC, rpi = 1024 / cpr;
  const int c0 = (threadIdx.x % cpr) * VEC, r0 = threadIdx.x / cpr;
  for (long r = (long)blockIdx.x * rpi + r0; r < M; r += (long)gridDim.x * rpi) {
    const float* src = v + row_div(r, rdiv) * C + c0;
    float f[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) f[j] = src[j] * scale;
    st16(out + r * C + c0, pack16<T>(f));
  }
}

// ------------------------------------------------------------------------------------------------
// layout / packing
// ------------------------------------------------------------------------------------------------
// OIHW f32 -> fwd pack [Cout][RS][Cin] T and dgrad pack [Cin][RS][Cout] T
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ fwd, T* __restrict__ dgr, int Cout,
                                   int Cin, int RS) {
  long total = (long)Cout * Cin * RS;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int tap = (int)(i % RS);
    long r = i / RS;
    int ci = (int)(r % Cin), co = (int)(r / Cin);
    float v = w[i];
    if (fwd) Elem<T>::st(fwd + ((size_t)co * RS + tap) * Cin + ci, v);
    if (dgr) Elem<T>::st(dgr + ((size_t)ci * RS + tap) * Cout + co, v);
  }
}

// All conv weights of an encoder in ONE launch (a launch costs ~5 us of dependency latency, the copy itself ~1 us):
// the descriptor table travels by value in the kernel arguments; a block finds its tensor by its block offset.
struct PackBatch {
  const float* w[ECG_PACK_MAX];
  void* fwd[ECG_PACK_MAX];
  void* dgr[ECG_PACK_MAX];
  int cout[ECG_PACK_MAX], cin[ECG_PACK_MAX], rs[ECG_PACK_MAX];
  int blk0[ECG_PACK_MAX + 1];  // first block of each tensor
  int n;
};
// One block = a 32 (co) x 32 (ci) tile of one tensor with all its taps, staged through LDS: the OIHW rows are read in
// contiguous 32 * RS-float runs, and both packed layouts are written as 32 consecutive elements per (row, tap) --
// 64-byte bf16 chunks instead of one scattered 2-byte store per element (that version spent 65 us on ResNet18's
// 11.2 M weights, 3.6x its byte floor, at the head of every forward).
constexpr int PACK_T = 32, PACK_RS_MAX = 9;
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(PackBatch b) {
  __shared__ float tile[PACK_T][PACK_T * PACK_RS_MAX + 1];  // [co][ci * RS + tap] (+1: conflict-free column reads)
  int t = 0;
  while (t + 1 < b.n && (int)blockIdx.x >= b.blk0[t + 1]) ++t;
  const int Cout = b.cout[t], Cin = b.cin[t], RS = b.rs[t];
  const float* __restrict__ w = b.w[t];
  T* __restrict__ fwd = (T*)b.fwd[t];
  T* __restrict__ dgr = (T*)b.dgr[t];
  const int ci_tiles = (Cin + PACK_T - 1) / PACK_T;
  const int lb = blockIdx.x - b.blk0[t];
  const int co0 = (lb / ci_tiles) * PACK_T, ci0 = (lb % ci_tiles) * PACK_T;
  const int nci = min(PACK_T, Cin - ci0), nco = min(PACK_T, Cout - co0);
  const int run = nci * RS;  // contiguous floats of one co row inside this tile
  for (int i = threadIdx.x; i < nco * run; i += 256) {
    const int co = i / run, k = i - co * run;
    tile[co][k] = w[((size_t)(co0 + co) * Cin + ci0) * RS + k];
  }
  __syncthreads();
  if (fwd) {  // fwd[co][tap][ci]: 32 consecutive ci per (co, tap)
    for (int i = threadIdx.x; i < nco * RS * PACK_T; i += 256) {
      const int ci = i % PACK_T, ct = i / PACK_T, tap = ct % RS, co = ct / RS;
      if (ci < nci) Elem<T>::st(fwd + ((size_t)(co0 + co) * RS + tap) * Cin + ci0 + ci, tile[co][ci * RS + tap]);
    }
  }
  if (dgr) {  // dgr[ci][tap][co]: 32 consecutive co per (ci, tap)
    for (int i = threadIdx.x; i < nci * RS * PACK_T; i += 256) {
      const int co = i % PACK_T, ct = i / PACK_T, tap = ct % RS, ci = ct / RS;
      if (co < nco) Elem<T>::st(dgr + ((size_t)(ci0 + ci) * RS + tap) * Cout + co0 + co, tile[co][ci * RS + tap]);
    }
  }
}

// NCHW f32 <-> NHWC T through a 32x32 LDS transpose (coalesced both ways)
template <typename T, bool TO_NHWC>
__global__ __launch_bounds__(256) void nchw_nhwc_kernel(const void* __restrict__ src_, void* __restrict__ dst_, int C,
                                                        long HW) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  if (TO_NHWC) {
    const float* src = (const float*)src_ + (size_t)n * C * HW;
    T* dst = (T*)dst_ + (size_t)n * C * HW;
    for (int k = ty; k < 32; k += 8) {
      int c = c0 + k;
      long pp = p0 + tx;
      tile[k][tx] = (c < C && pp < HW) ? src[(size_t)c * HW + pp] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      long pp = p0 + k;
      int c = c0 + tx;
      if (c < C && pp < HW) Elem<T>::st(dst + (size_t)pp * C + c, tile[tx][k]);
    }
  } else {
    const T* src = (const T*)src_ + (size_t)n * C * HW;
    float* dst = (float*)dst_ + (size_t)n * C * HW;
    for (int k = ty; k < 32; k += 8) {
      long pp = p0 + k;
      int c = c0 + tx;
      tile[k][tx] = (c < C && pp < HW) ? Elem<T>::ld(src + (size_t)pp * C + c) : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      int c = c0 + k;
      long pp = p0 + tx;
      if (c < C && pp < HW) dst[(size_t)c * HW + pp] = tile[tx][k];
    }
  }
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    Elem<T>::st(dst + i, src[i]);
}
template <typename T>
__global__ void uncast_kernel(const T* __restrict__ src, float* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = Elem<T>::ld(src + i);
}

inline int ew_grid(long work_items, int per_block) {
  long b = (work_items + per_block - 1) / per_block;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}
inline bool chunk_ok(int C, int dtype) {
  int vec = dtype == ECGMM_BF16 ? 8 : 4;
  if (C % vec) return false;
  int cpr = C / vec;
  return cpr <= EW_THREADS && (EW_THREADS % cpr) == 0;
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32, what)                 \
  if ((dtype) == ECGMM_BF16) { CALL_BF16; }                          \
  else if ((dtype) == ECGMM_F32) { CALL_F32; }                       \
  else ECG_FAIL(ECGMM_ERR_DTYPE, what ": bad dtype %d", (int)(dtype))

int ecg_bn_rows(int dtype, long M, int C) {
  int vec = dtype == ECGMM_BF16 ? 8 : 4;
  if (C < vec) return 1;  // (unsupported width: the kernel launchers reject it with a shape error)
  int rpi = EW_THREADS / (C / vec);
  int g = ew_grid(M, rpi * 8);
  return g > 1024 ? 1024 : g;
}

// Long partial-row buffers are first folded to ECG_TAIL_ROWS rows written into the buffer's tail
// (callers size partial buffers for rows + ECG_TAIL_ROWS rows), so the finalize kernels stay short.
static int fold_rows(const float*& partial, int& rows, int width, hipStream_t stream) {
  if (rows <= 512) return 0;  // the finalize kernels fold short buffers themselves (16 slices x <= 32 iterations)
  float* tail = const_cast<float*>(partial) + (size_t)rows * width;
  int chunk = ceil_div(rows, ECG_TAIL_ROWS);
  int nb = ceil_div(rows, chunk);
  hipLaunchKernelGGL(rows_stage1_kernel, dim3(ceil_div(width, 64), nb), dim3(1024), 0, stream, partial, rows, width,
                     chunk, tail);
  ECG_CHECK_LAUNCH("rows_stage1");
  partial = tail;
  rows = nb;
  return 0;
}

int ecg_bn_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* beta,
                    float* rm, float* rv, long long* nbt, float momentum, float eps, float* coef,
                    hipStream_t stream) {
  ECG_TRY(fold_rows(partial, rows, 2 * C, stream));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, stream, partial, rows, C, count, gamma,
                     beta, rm, rv, nbt, momentum, eps, coef);
  ECG_CHECK_LAUNCH("bn_finalize");
  return 0;
}

int ecg_bn_eval_coef(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                     float* coef, hipStream_t stream) {
  hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, stream, C, gamma, beta, rm, rv, eps,
                     coef);
  ECG_CHECK_LAUNCH("bn_eval_coef");
  return 0;
}

int ecg_col_stats(int dtype, const void* x, long M, int C, float* partial, int* rows_out, hipStream_t stream) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "col_stats: C=%d unsupported", C);
  int grid = ecg_bn_rows(dtype, M, C);
  *rows_out = grid;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(col_stats_kernel<bf16_t>, dim3(grid), dim3(EW_THREADS), 0, stream, (const bf16_t*)x, M,
                                C, partial),
             hipLaunchKernelGGL(col_stats_kernel<float>, dim3(grid), dim3(EW_THREADS), 0, stream, (const float*)x, M, C,
                                partial),
             "col_stats");
  ECG_CHECK_LAUNCH("col_stats");
  return 0;
}

// Finalize folded into the consumer (kernels above; default): ECGMM_BN_FOLD=0 / ecgmm_bn_fold(0) restores the separate launches.
static int g_bn_fold = -1;
extern "C" int ecgmm_bn_fold(int on) {
  g_bn_fold = on != 0;
  return 0;
}
bool ecg_bn_fold_ok(int C, int rows) {
  // Mid-round 3 this measured 6.98 ms with the separate launches against 7.04 ms folded (256 workgroups each re-reading the same
  // 128-256 KB of partial rows cost what the ~5 us launch + its boundary cost the stream).  With the consumers' first loads issued
  // BEFORE the fold -- its ~3 us then cover their HBM latency -- the folded form is the faster one: same-call A/B 6.47, 6.46 ->
  // 6.44, 6.46 ms (multimodal batch 256), 2.96, 2.97 -> 2.95, 2.97 (12-lead batch 512), 3.44, 3.45 -> 3.44, 3.44 (image-only batch
  // 128), and 35 launches fewer per step.  DEFAULT ON; ECGMM_BN_FOLD=0 / ecgmm_bn_fold(0) restores the separate launches.
  if (g_bn_fold < 0) { const char* e = getenv("ECGMM_BN_FOLD"); g_bn_fold = !(e && e[0] == '0'); }
  if (!g_bn_fold || rows < 1 || rows > 512 || C > 512) return false;
  return C >= 128 ? C % 128 == 0 : (C >= 16 && 1024 % C == 0);
}

// bn_act whose coefficients are folded from the producer's partial rows inside the launch (no bn_finalize launch);
// `coef` is an OUTPUT here (workgroup 0 writes it for the backward), as are the running statistics.
int ecg_bn_act_fold(int dtype, const void* y, float* coef, const EcgBnFold& f, const void* res, const float* rcoef,
                    const float* gate, int rows_per_sample, int relu, void* out, long M, int C, hipStream_t stream,
                    unsigned char* relu_bits) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "bn_act: C=%d unsupported", C);
  if (!ecg_bn_fold_ok(C, f.rows)) ECG_FAIL(ECGMM_ERR_SHAPE, "bn_act_fold: C=%d rows=%d not foldable", C, f.rows);
  BnActParams p;
  memset(&p, 0, sizeof(p));
  p.y = y; p.coef = coef; p.res = res; p.rcoef = rcoef; p.gate = gate; p.out = out; p.M = M; p.C = C;
  p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; p.relu = relu;
  p.rdiv = make_row_div(p.rows_per_sample);
  p.fin.partial = f.partial; p.fin.rows = f.rows; p.fin.count = f.count; p.fin.gamma = f.gamma; p.fin.beta = f.beta;
  p.fin.rm = f.rm; p.fin.rv = f.rv; p.fin.nbt = f.nbt; p.fin.momentum = f.momentum; p.fin.eps = f.eps; p.fin.coef_out = coef;
  p.relu_bits = dtype == ECGMM_BF16 && relu ? relu_bits : nullptr;
  int vec = dtype == ECGMM_BF16 ? 8 : 4;
  int grid = ew_grid(M, (BN_ACT_THREADS / (C / vec)) * 4);
  if (grid > 256) grid = 256;
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_kernel<bf16_t>, dim3(grid), dim3(BN_ACT_THREADS), 0, stream, p),
             hipLaunchKernelGGL(bn_act_kernel<float>, dim3(grid), dim3(BN_ACT_THREADS), 0, stream, p), "bn_act");
  ECG_CHECK_LAUNCH("bn_act_fold");
  return 0;
}

int ecg_bn_act(int dtype, const void* y, const float* coef, const void* res, const float* rcoef, const float* gate,
               int rows_per_sample, int relu, void* out, long M, int C, hipStream_t stream, unsigned char* relu_bits) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "bn_act: C=%d unsupported", C);
  BnActParams p;
  memset(&p, 0, sizeof(p));
  p.y = y; p.coef = coef; p.res = res; p.rcoef = rcoef; p.gate = gate; p.out = out; p.M = M; p.C = C;
  p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; p.relu = relu;
  p.rdiv = make_row_div(p.rows_per_sample);
  p.relu_bits = dtype == ECGMM_BF16 && relu ? relu_bits : nullptr;
  int vec = dtype == ECGMM_BF16 ? 8 : 4;
  // (1024-thread blocks, one per CU, each walking rows with a grid stride -- the launch shape of bn_bwd_kernel, which
  // reaches 5.3 TB/s; 256-thread blocks x 4096 reached 4.3: 0.74 ms per step for the 22 launches)
  int grid = ew_grid(M, (BN_ACT_THREADS / (C / vec)) * 4);
  if (grid > 256) grid = 256;   // (sweep 128 / 256 / 384 / 512 blocks: 0.72 / 0.59 / 0.68 / 0.66 ms per step for the 22 launches)
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_kernel<bf16_t>, dim3(grid), dim3(BN_ACT_THREADS), 0, stream, p),
             hipLaunchKernelGGL(bn_act_kernel<float>, dim3(grid), dim3(BN_ACT_THREADS), 0, stream, p), "bn_act");
  ECG_CHECK_LAUNCH("bn_act");
  return 0;
}

static int bn_bwd_rows(int dtype, long M, int C) {
  int vec = dtype == ECGMM_BF16 ? 8 : 4;
  if (C < vec) return 1;
  int rpi = BWD_THREADS / (C / vec);
  int g = ew_grid(M, rpi * 8);
  return g > 256 ? 256 : g;
}
template <typename T, bool APPLY>
static void bn_bwd_launch(const BnBwdParams& p, int grid, hipStream_t stream) {
  constexpr size_t lds = (size_t)BWD_THREADS * (2 * Elem<T>::VEC + 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)bn_bwd_kernel<T, APPLY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((bn_bwd_kernel<T, APPLY>), dim3(grid), dim3(BWD_THREADS), lds, stream, p);
}

// full BN backward: reduce -> finalize -> apply.  scratch: partial rows [rows][2][C] + bcoef [3][C]
size_t ecg_bn_bwd_scratch(int dtype, long M, int C) {
  return ((size_t)(bn_bwd_rows(dtype, M, C) + ECG_TAIL_ROWS) * 2 * C + 3 * C) * sizeof(float);
}

int ecg_bn_bwd(int dtype, const void* dout, const void* maskref, const float* gate, const float* addc,
               int rows_per_sample, const void* y, const float* coef, const float* gamma, float* dgamma, float* dbeta,
               void* dy, void* dz_out, float* dbias, long M, int C, float* scratch, hipStream_t stream,
               const unsigned char* mask_bits) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "bn_bwd: C=%d unsupported", C);
  if (dtype != ECGMM_BF16 || maskref == y) mask_bits = nullptr;
  int grid = bn_bwd_rows(dtype, M, C);
  float* partial = scratch;
  float* bcoef = scratch + (size_t)(grid + ECG_TAIL_ROWS) * 2 * C;
  BnBwdParams p;
  memset(&p, 0, sizeof(p));
  p.dout = dout; p.maskref = maskref; p.gate = gate; p.addc = addc; p.y = y; p.coef = coef; p.M = M; p.C = C;
  p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
  p.rdiv = make_row_div(p.rows_per_sample);
  p.partial = partial;
  p.mask_bits = mask_bits;
  const bool dz_early = dy && dz_out && (mask_bits || (maskref && maskref != y)) && dz_out != dout;
  if (dz_early) p.dz_out = dz_out;
  DISPATCH_T(dtype, (bn_bwd_launch<bf16_t, false>(p, grid, stream)), (bn_bwd_launch<float, false>(p, grid, stream)), "bn_bwd");
  ECG_CHECK_LAUNCH("bn_bwd_reduce");
  // finalize folded into the apply pass (every workgroup folds the <= 256 rows itself) unless the apply pass reuses the
  // row buffer for the bias-gradient sums, or there is no apply pass
  const bool fold = dy && !dbias && ecg_bn_fold_ok(C, grid);
  if (!fold) {
    const float* pr = partial;
    int rows = grid;  // <= 256: folded by the finalize kernel itself (16 slices x 16 rows)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, stream, pr, rows, C, (double)M,
                       gamma, coef, dgamma, dbeta, bcoef, 0);
    ECG_CHECK_LAUNCH("bn_bwd_finalize");
  }
  if (!dy) return 0;
  p.bcoef = bcoef; p.dy = dy; p.dz_out = dz_out;
  if (fold) {
    p.fin.partial = partial; p.fin.rows = grid; p.fin.count = (double)M; p.fin.gamma = gamma; p.fin.dgamma = dgamma;
    p.fin.dbeta = dbeta; p.fin.centered = 0;
  }
  if (dz_early) { p.dout = dz_out; p.maskref = nullptr; p.mask_bits = nullptr; p.dz_out = nullptr; }
  p.partial = dbias ? partial : nullptr;  // reuse (finalize already consumed it; stream-ordered)
  DISPATCH_T(dtype, (bn_bwd_launch<bf16_t, true>(p, grid, stream)), (bn_bwd_launch<float, true>(p, grid, stream)), "bn_bwd");
  ECG_CHECK_LAUNCH("bn_bwd_apply");
  if (dbias) {
    hipLaunchKernelGGL(rows_sum_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, stream, partial, grid, C, dbias, 0);
    ECG_CHECK_LAUNCH("rows_sum");
  }
  return 0;
}

// finalize + apply of a BatchNorm backward whose reduction rows already exist: `partial` = [rows <= 512][2][C] of
// (sum g, sum g * (y - mean)) written by the epilogue of the dgrad that produced `dout` (conv_halo.hip, ConvEpi).
// maskref as in ecg_bn_bwd (null when `dout` was stored already masked).
int ecg_bn_bwd_tail(int dtype, const void* dout, const void* maskref, const void* y, const float* coef,
                    const float* gamma, float* dgamma, float* dbeta, void* dy, const float* partial, int rows, long M,
                    int C, float* scratch, hipStream_t stream, const float* gate, const float* addc,
                    int rows_per_sample, float* dbias) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "bn_bwd: C=%d unsupported", C);
  if (rows < 1 || rows > 512) ECG_FAIL(ECGMM_ERR_SHAPE, "bn_bwd_tail: %d partial rows (1..512)", rows);
  const int grid = bn_bwd_rows(dtype, M, C);
  float* bcoef = scratch + (size_t)(grid + ECG_TAIL_ROWS) * 2 * C;
  const bool fold = ecg_bn_fold_ok(C, rows);   // (`partial` is the producer's buffer, never the apply pass's own rows)
  if (!fold) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, stream, partial, rows, C, (double)M,
                       gamma, coef, dgamma, dbeta, bcoef, 1);
    ECG_CHECK_LAUNCH("bn_bwd_finalize");
  }
  BnBwdParams p;
  memset(&p, 0, sizeof(p));
  p.dout = dout; p.maskref = maskref; p.y = y; p.coef = coef; p.M = M; p.C = C;
  p.gate = gate; p.addc = addc; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
  p.rdiv = make_row_div(p.rows_per_sample);
  p.bcoef = bcoef; p.dy = dy;
  if (fold) {
    p.fin.partial = partial; p.fin.rows = rows; p.fin.count = (double)M; p.fin.gamma = gamma; p.fin.dgamma = dgamma;
    p.fin.dbeta = dbeta; p.fin.centered = 1;
  }
  p.partial = dbias ? scratch : nullptr;   // (the finalize above reads `partial`, a different buffer: scratch's rows are free)
  DISPATCH_T(dtype, (bn_bwd_launch<bf16_t, true>(p, grid, stream)), (bn_bwd_launch<float, true>(p, grid, stream)), "bn_bwd");
  ECG_CHECK_LAUNCH("bn_bwd_apply");
  if (dbias) {
    hipLaunchKernelGGL(rows_sum_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, stream, (const float*)scratch, grid, C, dbias, 0);
    ECG_CHECK_LAUNCH("rows_sum");
  }
  return 0;
}

// Stem backward: [max-pool 3/2/1 <- ReLU <- BatchNorm] in two passes over the pooled tensors + one over y (kernels above).
// scratch as ecg_bn_bwd_scratch(dtype, N*H*W, C).  dbias (nullable): sum of dy (the stem conv's bias gradient).
template <typename T>
static void pool_bwd_launch(const void* dp, const void* pooled, const unsigned char* idx, const void* y, const float* coef,
                            const float* gamma, float* dgamma, float* dbeta, void* dy, float* dbias, int N, int H, int W,
                            int C, int OH, int OW, float* scratch, int rows, hipStream_t stream) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr size_t lds2 = (size_t)BWD_THREADS * (2 * VEC + 1) * sizeof(float), ldsp = (size_t)POOL_THREADS * (VEC + 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)pool_bwd_reduce_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    attr_set = true;
  }
  const long MP = (long)N * OH * OW, M = (long)N * H * W;
  float* partial = scratch;
  float* bcoef = scratch + (size_t)(rows + ECG_TAIL_ROWS) * 2 * C;
  const int rpi = BWD_THREADS / (C / VEC);
  int g1 = ew_grid(MP, rpi * 8);
  g1 = g1 > rows ? rows : g1;
  hipLaunchKernelGGL(pool_bwd_reduce_kernel<T>, dim3(g1), dim3(BWD_THREADS), lds2, stream, (const T*)dp, (const T*)pooled,
                     coef, partial, MP, C);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, stream, (const float*)partial, g1, C,
                     (double)M, gamma, coef, dgamma, dbeta, bcoef, 1);
  const long nblk = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
  int g2 = ew_grid(nblk, POOL_THREADS / (C / VEC));
  if (dbias && g2 > rows) g2 = rows;   // one partial row per block
  hipLaunchKernelGGL(pool_bn_bwd_apply_kernel<T>, dim3(g2), dim3(POOL_THREADS), dbias ? ldsp : 0, stream, (const T*)dp,
                     (const T*)pooled, idx, (const T*)y, coef, (const float*)bcoef, (T*)dy, dbias ? partial : nullptr, N, H, W,
                     C, OH, OW);
  if (dbias) hipLaunchKernelGGL(rows_sum_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, stream, (const float*)partial, g2, C, dbias, 0);
}

// Reduction + finalize of that backward alone (the recomputing stem applies it inside its weight-gradient kernel,
// conv_stem_fused.hip): dgamma, dbeta and *bcoef_out = [3][C] (gamma * invstd, mean dz, mean dz * xhat) inside `scratch`.
int ecg_pool_bn_bwd_reduce(int dtype, const void* dp, const void* pooled, const float* coef, const float* gamma, float* dgamma,
                           float* dbeta, int N, int H, int W, int C, float* scratch, const float** bcoef_out,
                           hipStream_t stream) {
  if (dtype != ECGMM_BF16) ECG_FAIL(ECGMM_ERR_DTYPE, "pool_bn_bwd_reduce: bf16 only");
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "pool_bn_bwd_reduce: C=%d unsupported", C);
  using T = bf16_t;
  constexpr int VEC = Elem<T>::VEC;
  constexpr size_t lds2 = (size_t)BWD_THREADS * (2 * VEC + 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)pool_bwd_reduce_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    attr_set = true;
  }
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const int rows = bn_bwd_rows(dtype, (long)N * H * W, C);
  const long MP = (long)N * OH * OW, M = (long)N * H * W;
  float* partial = scratch;
  float* bcoef = scratch + (size_t)(rows + ECG_TAIL_ROWS) * 2 * C;
  const int rpi = BWD_THREADS / (C / VEC);
  int g1 = ew_grid(MP, rpi * 8);
  g1 = g1 > rows ? rows : g1;
  hipLaunchKernelGGL(pool_bwd_reduce_kernel<T>, dim3(g1), dim3(BWD_THREADS), lds2, stream, (const T*)dp, (const T*)pooled, coef,
                     partial, MP, C);
  ECG_CHECK_LAUNCH("pool_bwd_reduce");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, stream, (const float*)partial, g1, C,
                     (double)M, gamma, coef, dgamma, dbeta, bcoef, 1);
  ECG_CHECK_LAUNCH("bn_bwd_finalize");
  *bcoef_out = bcoef;
  return 0;
}

// ECGMM_STEM_FUSE=0: the plans fall back to max-pool backward + full BatchNorm backward as separate passes (A/B switch)
bool ecg_stem_fuse_on() {
  static const bool on = [] { const char* e = getenv("ECGMM_STEM_FUSE"); return !(e && e[0] == '0'); }();
  return on;
}

int ecg_pool_bn_bwd(int dtype, const void* dp, const void* pooled, const unsigned char* idx, const void* y,
                    const float* coef, const float* gamma, float* dgamma, float* dbeta, void* dy, float* dbias, int N,
                    int H, int W, int C, float* scratch, hipStream_t stream) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "pool_bn_bwd: C=%d unsupported", C);
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const int rows = bn_bwd_rows(dtype, (long)N * H * W, C);
  DISPATCH_T(dtype, (pool_bwd_launch<bf16_t>(dp, pooled, idx, y, coef, gamma, dgamma, dbeta, dy, dbias, N, H, W, C, OH, OW, scratch, rows, stream)),
             (pool_bwd_launch<float>(dp, pooled, idx, y, coef, gamma, dgamma, dbeta, dy, dbias, N, H, W, C, OH, OW, scratch, rows, stream)),
             "pool_bn_bwd");
  ECG_CHECK_LAUNCH("pool_bn_bwd");
  return 0;
}

int ecg_rows_sum(const float* partial, int rows, int C, float* out, int accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(rows_sum_kernel, dim3(ceil_div(C, 64)), dim3(1024), 0, stream, partial, rows, C, out, accumulate);
  ECG_CHECK_LAUNCH("rows_sum");
  return 0;
}

int ecg_bnrelu_maxpool(int dtype, const void* y, const float* coef, void* out, unsigned char* idx, int N, int H, int W,
                       int C, hipStream_t stream) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "maxpool: C=%d unsupported", C);
  int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  int vec = dtype == ECGMM_BF16 ? 8 : 4;
  int grid = ew_grid((long)N * OH * OW * (C / vec), EW_THREADS);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bnrelu_maxpool_kernel<bf16_t>, dim3(grid), dim3(EW_THREADS), 0, stream,
                                (const bf16_t*)y, coef, (bf16_t*)out, idx, N, H, W, C, OH, OW),
             hipLaunchKernelGGL(bnrelu_maxpool_kernel<float>, dim3(grid), dim3(EW_THREADS), 0, stream, (const float*)y,
                                coef, (float*)out, idx, N, H, W, C, OH, OW),
             "maxpool");
  ECG_CHECK_LAUNCH("bnrelu_maxpool");
  return 0;
}

int ecg_maxpool_relu_bwd(int dtype, const void* dp, const void* pooled, const unsigned char* idx, void* dz, int N,
                         int H, int W, int C, hipStream_t stream) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "maxpool bwd: C=%d unsupported", C);
  int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  int vec = dtype == ECGMM_BF16 ? 8 : 4;
  int grid = ew_grid((long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / vec), EW_THREADS);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool_relu_bwd_kernel<bf16_t>, dim3(grid), dim3(EW_THREADS), 0, stream,
                                (const bf16_t*)dp, (const bf16_t*)pooled, idx, (bf16_t*)dz, N, H, W, C, OH, OW),
             hipLaunchKernelGGL(maxpool_relu_bwd_kernel<float>, dim3(grid), dim3(EW_THREADS), 0, stream,
                                (const float*)dp, (const float*)pooled, idx, (float*)dz, N, H, W, C, OH, OW),
             "maxpool bwd");
  ECG_CHECK_LAUNCH("maxpool_relu_bwd");
  return 0;
}

int ecg_avgpool(int dtype, const void* x, float* out, int N, int R, int C, const float* coef, hipStream_t stream) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "avgpool: C=%d unsupported", C);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((sample_reduce_kernel<bf16_t, false>), dim3(N), dim3(EW_THREADS), 0, stream,
                                (const bf16_t*)x, (const bf16_t*)nullptr, (const bf16_t*)nullptr, coef, out, R, C),
             hipLaunchKernelGGL((sample_reduce_kernel<float, false>), dim3(N), dim3(EW_THREADS), 0, stream,
                                (const float*)x, (const float*)nullptr, (const float*)nullptr, coef, out, R, C),
             "avgpool");
  ECG_CHECK_LAUNCH("avgpool");
  return 0;
}

int ecg_bcast_rows(int dtype, const float* v, void* out, int N, int R, int C, float scale, hipStream_t stream) {
  long total = (long)N * R * C;
  const int vecw = dtype == ECGMM_BF16 ? 8 : 4;
  if (C % vecw == 0 && C / vecw <= 1024 && 1024 % (C / vecw) == 0 && ((uintptr_t)out & 15) == 0) {
    const long M = (long)N * R;
    int g = ew_grid(M, (1024 / (C / vecw)) * 4);
    if (g > 256) g = 256;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bcast_rows_vec_kernel<bf16_t>, dim3(g), dim3(1024), 0, stream, v, (bf16_t*)out, M, make_row_div(R), C, scale),
               hipLaunchKernelGGL(bcast_rows_vec_kernel<float>, dim3(g), dim3(1024), 0, stream, v, (float*)out, M, make_row_div(R), C, scale),
               "bcast_rows");
    ECG_CHECK_LAUNCH("bcast_rows_vec");
    return 0;
  }
  int grid = ew_grid(total, 256);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bcast_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, v, (bf16_t*)out, total, R,
                                C, scale),
             hipLaunchKernelGGL(bcast_rows_kernel<float>, dim3(grid), dim3(256), 0, stream, v, (float*)out, total, R, C,
                                scale),
             "bcast_rows");
  ECG_CHECK_LAUNCH("bcast_rows");
  return 0;
}

int ecg_se_gate_grad(int dtype, const void* dout, const void* maskref, const void* y, const float* coef, float* dg,
                     int N, int R, int C, hipStream_t stream) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "se_gate_grad: C=%d unsupported", C);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((sample_reduce_kernel<bf16_t, true>), dim3(N), dim3(EW_THREADS), 0, stream,
                                (const bf16_t*)dout, (const bf16_t*)maskref, (const bf16_t*)y, coef, dg, R, C),
             hipLaunchKernelGGL((sample_reduce_kernel<float, true>), dim3(N), dim3(EW_THREADS), 0, stream,
                                (const float*)dout, (const float*)maskref, (const float*)y, coef, dg, R, C),
             "se_gate_grad");
  ECG_CHECK_LAUNCH("se_gate_grad");
  return 0;
}

// SE-block backward in two tensor passes instead of three (kernels above): se_gate_bn writes dz and the per-sample sums,
// se_bn_rows turns them (with the gate and the pooled-path gradient addc, known only after the SE MLP's backward) into
// the BatchNorm-backward reduction row, ecg_bn_bwd_tail finishes (finalize + apply with gate / addc / dbias).
int ecg_se_gate_bn(int dtype, const void* dout, const void* maskref, const void* y, const float* coef, void* dz,
                   float* dg, float* a1, float* a2, float* a3, int N, int R, int C, hipStream_t stream) {
  if (!chunk_ok(C, dtype)) ECG_FAIL(ECGMM_ERR_SHAPE, "se_gate_bn: C=%d unsupported", C);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(se_gate_bn_kernel<bf16_t>, dim3(N), dim3(EW_THREADS), 0, stream, (const bf16_t*)dout,
                                (const bf16_t*)maskref, (const bf16_t*)y, coef, (bf16_t*)dz, dg, a1, a2, a3, R, C),
             hipLaunchKernelGGL(se_gate_bn_kernel<float>, dim3(N), dim3(EW_THREADS), 0, stream, (const float*)dout,
                                (const float*)maskref, (const float*)y, coef, (float*)dz, dg, a1, a2, a3, R, C),
             "se_gate_bn");
  ECG_CHECK_LAUNCH("se_gate_bn");
  return 0;
}
int ecg_se_bn_rows(const float* a1, const float* a2, const float* a3, const float* gate, const float* addc, int N, int R,
                   int C, float* rows, hipStream_t stream) {
  hipLaunchKernelGGL(se_bn_rows_kernel, dim3(ceil_div(C, 64), SE_ROWS), dim3(256), 0, stream, a1, a2, a3, gate, addc, N, R,
                     C, rows);
  ECG_CHECK_LAUNCH("se_bn_rows");
  return 0;
}
int ecg_se_bn_nrows() { return SE_ROWS; }

int ecg_pack_weight(int dtype, const float* w_oihw, void* fwd, void* dgrad, int Cout, int Cin, int RS,
                    hipStream_t stream) {
  long total = (long)Cout * Cin * RS;
  int grid = ew_grid(total, 256);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, w_oihw, (bf16_t*)fwd,
                                (bf16_t*)dgrad, Cout, Cin, RS),
             hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, stream, w_oihw, (float*)fwd,
                                (float*)dgrad, Cout, Cin, RS),
             "pack_weight");
  ECG_CHECK_LAUNCH("pack_weight");
  return 0;
}

int ecg_pack_weight_batch(int dtype, const EcgPackItem* items, int n, hipStream_t stream) {
  for (int base = 0; base < n; base += ECG_PACK_MAX) {
    PackBatch b;
    memset(&b, 0, sizeof(b));
    b.n = n - base < ECG_PACK_MAX ? n - base : ECG_PACK_MAX;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      const EcgPackItem& it = items[base + i];
      b.w[i] = it.w; b.fwd[i] = it.fwd; b.dgr[i] = it.dgrad; b.cout[i] = it.Cout; b.cin[i] = it.Cin; b.rs[i] = it.RS;
      b.blk0[i] = blocks;
      if (it.RS > 9) ECG_FAIL(ECGMM_ERR_SHAPE, "pack_weight_batch: %d taps unsupported (<= 9)", it.RS);
      blocks += ceil_div(it.Cout, 32) * ceil_div(it.Cin, 32);  // one block per 32 x 32 (co, ci) tile
    }
    b.blk0[b.n] = blocks;
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weight_batch_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, b),
               hipLaunchKernelGGL(pack_weight_batch_kernel<float>, dim3(blocks), dim3(256), 0, stream, b), "pack_weight_batch");
    ECG_CHECK_LAUNCH("pack_weight_batch");
  }
  return 0;
}

int ecg_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, long HW, hipStream_t stream) {
  dim3 grid(ceil_div(HW, 32), ceil_div(C, 32), N);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((nchw_nhwc_kernel<bf16_t, true>), grid, dim3(256), 0, stream, src, dst, C, HW),
             hipLaunchKernelGGL((nchw_nhwc_kernel<float, true>), grid, dim3(256), 0, stream, src, dst, C, HW),
             "nchw_to_nhwc");
  ECG_CHECK_LAUNCH("nchw_to_nhwc");
  return 0;
}

int ecg_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, long HW, hipStream_t stream) {
  dim3 grid(ceil_div(HW, 32), ceil_div(C, 32), N);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((nchw_nhwc_kernel<bf16_t, false>), grid, dim3(256), 0, stream, src, dst, C, HW),
             hipLaunchKernelGGL((nchw_nhwc_kernel<float, false>), grid, dim3(256), 0, stream, src, dst, C, HW),
             "nhwc_to_nchw");
  ECG_CHECK_LAUNCH("nhwc_to_nchw");
  return 0;
}

int ecg_cast(int dtype, const float* src, void* dst, long n, hipStream_t stream) {
  int grid = ew_grid(n, 256);
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, src, (bf16_t*)dst, n),
             hipLaunchKernelGGL(cast_kernel<float>, dim3(grid), dim3(256), 0, stream, src, (float*)dst, n), "cast");
  ECG_CHECK_LAUNCH("cast");
  return 0;
}
int ecg_uncast(int dtype, const void* src, float* dst, long n, hipStream_t stream) {
  int grid = ew_grid(n, 256);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(uncast_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)src, dst, n),
             hipLaunchKernelGGL(uncast_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)src, dst, n),
             "uncast");
  ECG_CHECK_LAUNCH("uncast");
  return 0;
}
