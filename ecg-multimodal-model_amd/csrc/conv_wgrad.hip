// Weight-gradient convolution on MFMA for channels-last activations, gfx950.
//
//   dW[co][r][s][ci] = sum_{n,oh,ow} dy[n,oh,ow,co] * x[n, oh*st-p+r, ow*st-p+s, ci]
//
// GEMM view: M' = Cout, N' = Cin (per tap), K' = output pixels.  Both operands are pixel-major in
// memory ([pixel][channel]), i.e. K is the *slow* axis, so the MFMA fragments are produced by the
// gfx950 transposing LDS read (ds_read_b64_tr_b16) for bf16 and by plain ds_read_b32 for f32.
// One workgroup owns a 64(co) x 64(ci) tile for ALL taps of the filter (the dy tile is staged once
// per K step and reused by every tap; each tap's x tile is a shifted row gather) over a slice of
// the pixel range (split-K).  Partial tiles go to an fp32 slab [split][tap][co][ci] with plain
// stores; a second kernel sums the splits in a fixed order (bitwise reproducible) and scatters into
// the OIHW gradient tensor.
#include "ops.h"

namespace {

struct WgradParams {
  const void* x;
  const void* dy;
  float* slab;
  int H, W, Cin, OH, OW, Cout, S, stride, pad_h, pad_w;
  int M;
  int steps_per_split;
};

template <typename T> struct WgCfg;
template <> struct WgCfg<bf16_t> {
  static constexpr int KP = 32;          // pixels per K step
  static constexpr int ROWSTRIDE = 144;  // 64 ch * 2 B + 16 B pad
  static constexpr int CH = 8;           // 16-B chunks per row
};
template <> struct WgCfg<float> {
  static constexpr int KP = 16;
  static constexpr int ROWSTRIDE = 320;  // 64 ch * 4 B + 64 B pad (k-groups land on disjoint banks)
  static constexpr int CH = 16;
};

template <typename T, int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
  using C = WgCfg<T>;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int TILE_BYTES = C::KP * C::ROWSTRIDE;
  __shared__ __attribute__((aligned(16))) unsigned char smem[(1 + NT) * TILE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave & 1, wci = wave >> 1;
  const int ci_tiles = (p.Cin + 63) / 64;
  const int co0 = (blockIdx.x / ci_tiles) * 64, ci0 = (blockIdx.x % ci_tiles) * 64;
  const int split = blockIdx.y;
  const T* __restrict__ x = (const T*)p.x;
  const T* __restrict__ dy = (const T*)p.dy;

  const int lrow = tid / C::CH, chunk = tid % C::CH;
  const int total_steps = (p.M + C::KP - 1) / C::KP;
  const int s_begin = split * p.steps_per_split;
  const int s_end = min(total_steps, s_begin + p.steps_per_split);

  f32x4 acc[NT][2][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 vdy, vx[NT];
  const int cdy = co0 + chunk * VEC, cx = ci0 + chunk * VEC;
  auto gload = [&](int step) {
    int pix = step * C::KP + lrow;
    bool pok = pix < p.M;
    int pp = pok ? pix : 0;
    int n = pp / (p.OH * p.OW);
    int rem = pp - n * (p.OH * p.OW);
    int oh = rem / p.OW, ow = rem - oh * p.OW;
    vdy = (u32x4){0u, 0u, 0u, 0u};
    if (pok && cdy < p.Cout) vdy = *reinterpret_cast<const u32x4*>(dy + (size_t)pix * p.Cout + cdy);
    const int hb = oh * p.stride - p.pad_h, wb = ow * p.stride - p.pad_w;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int r = t / p.S, s = t - r * p.S;
      int hs = hb + r, ws = wb + s;
      bool ok = pok && cx < p.Cin && (unsigned)hs < (unsigned)p.H && (unsigned)ws < (unsigned)p.W;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x4*>(x + ((size_t)(n * p.H + hs) * p.W + ws) * p.Cin + cx);
      vx[t] = v;
    }
  };
  auto lstore = [&]() {
    unsigned char* base = smem + lrow * C::ROWSTRIDE + chunk * 16;
    *reinterpret_cast<u32x4*>(base) = vdy;
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<u32x4*>(base + (1 + t) * TILE_BYTES) = vx[t];
  };

  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&]() {
    if constexpr (sizeof(T) == 2) {
      // transposing reads: lane 4q+p of a 16-lane group addresses row (kb+q), columns 4p..4p+3;
      // lane i receives column i of those 4 rows.  Group fq takes k rows 8fq..8fq+7.
      const int q = fr >> 2, pq = fr & 3;
      auto frag = [&](const unsigned char* tile, int col0) -> u32x4 {
        const unsigned char* a0 = tile + (8 * fq + q) * C::ROWSTRIDE + (col0 + 4 * pq) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(a0)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(a0 + 4 * C::ROWSTRIDE)));
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return (u32x4){l2.x, l2.y, h2.x, h2.y};
      };
      u32x4 fa[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = frag(smem, wco * 32 + a * 16);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        u32x4 fb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = frag(smem + (1 + t) * TILE_BYTES, wci * 32 + b * 16);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[a]),
                                                                   __builtin_bit_cast(bf16x8_t, fb[b]),
                                                                   acc[t][a][b], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < C::KP / 4; ++ks) {
        const int krow = ks * 4 + fq;
        float fa[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
          fa[a] = *reinterpret_cast<const float*>(smem + krow * C::ROWSTRIDE + (wco * 32 + a * 16 + fr) * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float fb[2];
#pragma unroll
          for (int b = 0; b < 2; ++b)
            fb[b] = *reinterpret_cast<const float*>(smem + (1 + t) * TILE_BYTES + krow * C::ROWSTRIDE +
                                                    (wci * 32 + b * 16 + fr) * 4);
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[t][a][b], 0, 0, 0);
        }
      }
    }
  };

  if (s_begin < s_end) {
    gload(s_begin);
    for (int step = s_begin; step < s_end; ++step) {
      lstore();
      __syncthreads();
      if (step + 1 < s_end) gload(step + 1);
      compute();
      __syncthreads();
    }
  }

  // ---- slab store: D[row = co][col = ci]; lane: ci = fr, co = fq*4 + j
  const int RS = NT;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        int ci = ci0 + wci * 32 + b * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int co = co0 + wco * 32 + a * 16 + fq * 4 + j;
          if (co < p.Cout && ci < p.Cin)
            p.slab[(((size_t)split * RS + t) * p.Cout + co) * p.Cin + ci] = acc[t][a][b][j];
        }
      }
}

// grad[co][ci][tap] (OIHW flattened) = sum_split slab[split][tap][co][ci]   (fixed order)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad, int nsplit, int RS,
                                    int Cout, int Cin, int accumulate) {
  // 256 threads = 64 consecutive outputs x 4 split groups (coalesced 256-B rows, 4-way ILP over splits)
  __shared__ float sh[4][64];
  size_t per = (size_t)RS * Cout * Cin;
  size_t i = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int grp = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (i < per) {
    int k = grp;
    for (; k + 4 < nsplit; k += 8) {
      s0 += slab[(size_t)k * per + i];
      s1 += slab[(size_t)(k + 4) * per + i];
    }
    if (k < nsplit) s0 += slab[(size_t)k * per + i];
  }
  sh[grp][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (grp == 0 && i < per) {
    float s = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    int ci = (int)(i % Cin);
    size_t r = i / Cin;
    int co = (int)(r % Cout);
    int tap = (int)(r / Cout);
    size_t o = ((size_t)co * Cin + ci) * RS + tap;
    grad[o] = accumulate ? grad[o] + s : s;
  }
}

template <typename T>
int launch_wgrad(const ConvGeom& g, WgradParams& p, int nsplit, hipStream_t stream) {
  const int RS = g.R * g.S;
  dim3 grid(ceil_div(g.Cout, 64) * ceil_div(g.Cin, 64), nsplit);
  if (RS == 1) hipLaunchKernelGGL((wgrad_kernel<T, 1>), grid, dim3(256), 0, stream, p);
  else if (RS == 3) hipLaunchKernelGGL((wgrad_kernel<T, 3>), grid, dim3(256), 0, stream, p);
  else if (RS == 9) hipLaunchKernelGGL((wgrad_kernel<T, 9>), grid, dim3(256), 0, stream, p);
  else ECG_FAIL(ECGMM_ERR_SHAPE, "conv wgrad: %dx%d filter unsupported (1, 3 or 9 taps)", g.R, g.S);
  ECG_CHECK_LAUNCH("wgrad_kernel");
  return 0;
}

int pick_nsplit(const ConvGeom& g, int kp) {
  long M = (long)g.N * g.OH * g.OW;
  int tiles = ceil_div(g.Cout, 64) * ceil_div(g.Cin, 64);
  int steps = ceil_div(M, kp);
  int want = ceil_div(1024, tiles);  // ~4 workgroups per CU
  int ns = want < 1 ? 1 : want;
  if (ns > steps) ns = steps;
  if (ns > 512) ns = 512;
  return ns < 1 ? 1 : ns;
}

}  // namespace

size_t ecg_conv_wgrad_workspace(int dtype, const ConvGeom& g) {
  int ns = pick_nsplit(g, dtype == ECGMM_BF16 ? 32 : 16);
  return (size_t)ns * g.R * g.S * g.Cout * g.Cin * sizeof(float);
}

int ecg_conv_wgrad(int dtype, const ConvGeom& g, const void* x, const void* dy, float* grad_oihw, int accumulate,
                   void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const int kp = dtype == ECGMM_BF16 ? 32 : 16;
  const int vec = dtype == ECGMM_BF16 ? 8 : 4;
  if (g.Cin % vec || g.Cout % vec)
    ECG_FAIL(ECGMM_ERR_SHAPE, "conv wgrad: channels (%d,%d) must be multiples of %d", g.Cin, g.Cout, vec);
  long M = (long)g.N * g.OH * g.OW;
  if (M <= 0 || M > 0x7fffffffL) ECG_FAIL(ECGMM_ERR_SHAPE, "conv wgrad: pixel count %ld out of range", M);
  int ns = pick_nsplit(g, kp);
  size_t need = (size_t)ns * g.R * g.S * g.Cout * g.Cin * sizeof(float);
  if (workspace_bytes < need || !workspace)
    ECG_FAIL(ECGMM_ERR_WORKSPACE, "conv wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
  WgradParams p;
  p.x = x; p.dy = dy; p.slab = (float*)workspace;
  p.H = g.H; p.W = g.W; p.Cin = g.Cin; p.OH = g.OH; p.OW = g.OW; p.Cout = g.Cout; p.S = g.S;
  p.stride = g.stride; p.pad_h = g.pad_h; p.pad_w = g.pad_w; p.M = (int)M;
  p.steps_per_split = ceil_div(ceil_div(M, kp), ns);
  if (dtype != ECGMM_BF16 && dtype != ECGMM_F32) ECG_FAIL(ECGMM_ERR_DTYPE, "conv wgrad: bad dtype %d", dtype);
  ecg_prof_begin(ECG_PROF_WGRAD, 2.0 * (double)M * g.Cout * g.R * g.S * g.Cin, stream);
  int rc = dtype == ECGMM_BF16 ? launch_wgrad<bf16_t>(g, p, ns, stream) : launch_wgrad<float>(g, p, ns, stream);
  ecg_prof_end(stream);
  ECG_TRY(rc);
  size_t per = (size_t)g.R * g.S * g.Cout * g.Cin;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(per, 64)), dim3(256), 0, stream, (const float*)workspace,
                     grad_oihw, ns, g.R * g.S, g.Cout, g.Cin, accumulate);
  ECG_CHECK_LAUNCH("wgrad_reduce_kernel");
  return 0;
}
