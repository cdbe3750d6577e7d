// The 2-D stem by RECOMPUTE (round 3), gfx950, bf16:  conv7x7/2 -> BatchNorm -> ReLU -> MaxPool(3,2,1)
// (torchvision resnet18 conv1 / bn1 / relu / maxpool, instantiated by the reference at
// multimodal_paper_modal_balance.py:210) without ever writing the full-resolution conv output.
//
// At batch 256 that tensor is 411 MB (64 x 112 x 112 bf16 per sample) against 154 MB of input and 103 MB of pooled
// output, and the 7x7 conv is 60 GFLOP = 24 us of MFMA: re-running it is cheaper than one pass over its result.
//
//   forward   pass 1  stem_fwd_kernel<bf16, 7, STATS_ONLY>  (conv_stem.hip)     image -> BatchNorm partial sums
//             pass 2  stem_pool_fwd_kernel                                       image -> conv -> bn -> relu -> max-pool
//                     a workgroup computes the (2*4+1) x (2*8+1) conv outputs under a 4 x 8 block of pooled pixels
//                     (one row / column of halo recomputed: 153 instead of 128 positions), rounds them to bf16 exactly as
//                     the stored tensor would have been, applies bn + relu, rounds again, parks the tile in LDS and takes
//                     the 3x3/2 maxima there ("largest value, earliest tap": the same sortable key as
//                     bnrelu_maxpool_kernel, elementwise.hip) -> pooled tensor + arg-max bytes
//   backward  stem_bwd_wgrad_kernel: per 8 x 16 conv tile -- conv recomputed (rounded to bf16), the max-pool backward
//             gathered from the pooled gradient (<= 4 windows per position), BatchNorm-backward apply
//             dy = k1 * dz + (bn * y + an) (the arithmetic of pool_bn_bwd_apply_kernel), dy rounded to bf16 into LDS,
//             and the weight-gradient MFMA loop of stem_wgrad_kernel run from there: neither y nor dy touch HBM.
//             (The BatchNorm-backward REDUCTION runs over the pooled tensors before it: pool_bwd_reduce_kernel.)
//
// Values: bit-identical to the two-pass route (same MFMA order per output, same rounding points) up to the fp32 summation
// order of the statistics / weight-gradient partial sums.
#include "ops.h"

namespace {

constexpr int CO = 64;
constexpr int R7 = 7;
constexpr int KSTEPS_MAX = 6;            // (Cin * 7 + 3) / 4 <= 6: Cin <= 3
constexpr int SROW = 144;                // LDS row stride (bytes) of [pixel][64 ch] bf16 tiles: 16-B aligned, conflict-free b64 writes

struct FusedParams {
  const float* x;        // [N, Cin, H, W] fp32
  const void* wpk;       // [64][KP] bf16 (stem_pack_kernel)
  const float* coef;     // forward BatchNorm coefficients [4][64]: scale, shift, mean, invstd
  const float* bcoef;    // backward: [3][64] = gamma * invstd, mean dz, mean dz * xhat  (bn_bwd_finalize_kernel)
  void* pooled;          // fwd out / bwd in: [N, PH, PW, 64] bf16
  unsigned char* idx;    // fwd out / bwd in: arg-max tap per pooled element
  const void* dp;        // bwd: gradient w.r.t. pooled
  float* slab;           // bwd: [workgroup][64][NG * 8]
  int N, Cin, H, W;      // image
  int OH, OW;            // conv output
  int PH, PW;            // pooled output
  int tiles_h, tiles_w;  // tile positions per image
  int NG;                // Cin * 7
};

__device__ __forceinline__ int xcd_order() {
  const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- input patch of a conv tile of TR x TC outputs whose first output is (r0, c0): rows 2*r0-3 .. , cols 2*c0-3 ..
template <int TR, int TC> struct Patch {
  static constexpr int PHh = (TR - 1) * 2 + R7, PWw = (TC - 1) * 2 + 8, PWS = (PWw + 1) & ~1;
  static constexpr int NPRE = (3 * PHh * PWS + 255) / 256;   // prefetch registers per thread (Cin <= 3)
  static constexpr int ELEMS = NPRE * 256;
};
template <int TR, int TC>
__device__ __forceinline__ void patch_offsets(const FusedParams& p, int r0, int c0, unsigned (&off)[Patch<TR, TC>::NPRE]) {
  using P = Patch<TR, TC>;
  const int total = p.Cin * P::PHh * P::PWS;
  const int ih0 = r0 * 2 - 3, iw0 = c0 * 2 - 3;
#pragma unroll
  for (int k = 0; k < P::NPRE; ++k) {
    const int i = threadIdx.x + 256 * k;
    const int pw = i % P::PWS;
    const int t = i / P::PWS;
    const int ph = t % P::PHh, c = t / P::PHh;
    const int ih = ih0 + ph, iw = iw0 + pw;
    const int ok = (int)(i < total) & (int)(pw < P::PWw) & (int)((unsigned)ih < (unsigned)p.H) & (int)((unsigned)iw < (unsigned)p.W);
    off[k] = ok ? (unsigned)((c * p.H + ih) * p.W + iw) * 4u : 0xFFFFFFFFu;
  }
}
template <int NPRE>
__device__ __forceinline__ void patch_fetch(const FusedParams& p, const unsigned (&off)[NPRE], float (&pre)[NPRE], int n) {
  const size_t img = (size_t)p.Cin * p.H * p.W;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * img), 0, (int)(img * sizeof(float)), 0x00020000);
#pragma unroll
  for (int k = 0; k < NPRE; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
    pre[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off[k], 0, 0));
#endif
  }
}
template <int NPRE> __device__ __forceinline__ void patch_store(bf16_t* patch, const float (&pre)[NPRE]) {
#pragma unroll
  for (int k = 0; k < NPRE; ++k) patch[threadIdx.x + 256 * k] = f2bf(pre[k]);
}

// weights [64][KP] -> LDS rows of KP + 8 elements (16-B reads conflict-free)
__device__ __forceinline__ void stage_weights(bf16_t* sW, const void* wpk, int KP, int WS) {
  const bf16_t* wg = (const bf16_t*)wpk;
  const int vpr = KP / 8;
  for (int i = threadIdx.x; i < CO * vpr; i += 256) {
    const int co = i / vpr, kv = i - co * vpr;
    *reinterpret_cast<u32x4*>(sW + co * WS + kv * 8) = *reinterpret_cast<const u32x4*>(wg + (size_t)co * KP + kv * 8);
  }
}

// patch row offsets of this lane's (c, r) group in every k-step, two 16-bit offsets per register
template <int PHh, int PWS>
__device__ __forceinline__ void group_offsets(int NG, int fq, unsigned (&goff2)[KSTEPS_MAX / 2]) {
#pragma unroll
  for (int ks = 0; ks < KSTEPS_MAX; ++ks) {
    int G = ks * 4 + fq;
    if (G >= NG) G = NG - 1;
    const int c = G / R7, r = G - c * R7;
    const unsigned g = (unsigned)((c * PHh + r) * PWS);
    if (ks & 1) goff2[ks / 2] |= g << 16;
    else goff2[ks / 2] = g;
  }
}

// conv of NB 16-pixel segments x 64 channels from the LDS patch: acc[a][b] (a = 16-channel tile, lane = pixel fr of
// segment b, channels a*16 + fq*4 + j).  Same operand order per output as stem_fwd_kernel.
template <int NB>
__device__ __forceinline__ void conv_tile(f32x4 (&acc)[4][NB], const bf16_t* patch, const bf16_t* wlane, int WS, int ksteps,
                                          const unsigned (&goff2)[KSTEPS_MAX / 2], const int (&boff)[NB]) {
#pragma unroll
  for (int ks = 0; ks < KSTEPS_MAX; ++ks) {
    if (ks < ksteps) {
      u32x4 fb[NB];
      const int goff = (ks & 1) ? (int)(goff2[ks / 2] >> 16) : (int)(goff2[ks / 2] & 0xffffu);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const unsigned* src = reinterpret_cast<const unsigned*>(patch + goff + boff[b]);
        fb[b] = (u32x4){src[0], src[1], src[2], src[3]};
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const u32x4 fa = *reinterpret_cast<const u32x4*>(wlane + a * 16 * WS + ks * 32);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (ks == 0) mfma_bf16_first(acc[a][b], fa, fb[b]);
          else mfma_bf16_inplace(acc[a][b], fa, fb[b]);
        }
      }
    }
  }
  mfma_drain();
}

// ====================================================================================================
// forward pass 2: image -> conv -> (round) -> bn -> relu -> (round) -> 3x3/2 max-pool
// ====================================================================================================
constexpr int PTH = 4, PTW = 8;                       // pooled pixels per tile
constexpr int FR_ = 2 * PTH + 1, FC_ = 2 * PTW + 1;   // conv outputs under them: 9 x 17 = 153
constexpr int FPIX = FR_ * FC_;
constexpr int FNB = 3;                                // 16-pixel segments per wave: 4 x 3 x 16 = 192 >= 153

__global__ __launch_bounds__(256, 3) void stem_pool_fwd_kernel(FusedParams p) {
  using P = Patch<FR_, FC_>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ksteps = (p.NG + 3) / 4, KP = ksteps * 32, WS = KP + 8;
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem);
  const size_t w_bytes = (((size_t)CO * WS * 2) + 15) & ~(size_t)15;
  bf16_t* const patch0 = reinterpret_cast<bf16_t*>(smem + w_bytes);
  unsigned char* const sT = smem + w_bytes + 2 * (size_t)P::ELEMS * 2;   // [FPIX][SROW]: bn + relu'd tile, bf16
  float* const sC = reinterpret_cast<float*>(sT + FPIX * SROW);         // [2][64]: scale, shift (registers are for occupancy)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  stage_weights(sW, p.wpk, KP, WS);

  const int tpi = p.tiles_h * p.tiles_w;
  const int vb = xcd_order();
  const int pos = vb % tpi, slot = vb / tpi, nslots = gridDim.x / tpi;
  const int th_i = pos / p.tiles_w, tw_i = pos - th_i * p.tiles_w;
  const int ph0 = th_i * PTH, pw0 = tw_i * PTW;       // first pooled pixel of the tile
  const int r0 = 2 * ph0 - 1, c0 = 2 * pw0 - 1;       // first conv output (may be -1: never a maximum, see `valid`)

  float pre[P::NPRE];
  unsigned poff[P::NPRE];
  patch_offsets<FR_, FC_>(p, r0, c0, poff);

  // this lane's conv positions: slot s = (wave * FNB + b) * 16 + fr -> (s / 17, s % 17); slots past the tile repeat its last position
  int boff[FNB], trow[FNB];
#pragma unroll
  for (int b = 0; b < FNB; ++b) {
    int s = (wave * FNB + b) * 16 + fr;
    trow[b] = s < FPIX ? s : -1;
    if (s >= FPIX) s = FPIX - 1;
    const int pr = s / FC_, pc = s - pr * FC_;
    boff[b] = pr * 2 * P::PWS + pc * 2;
  }
  unsigned goff2[KSTEPS_MAX / 2];
  group_offsets<P::PHh, P::PWS>(p.NG, fq, goff2);
  const bf16_t* wlane = sW + fr * WS + fq * 8;

  if (tid < 2 * CO) sC[tid] = p.coef[tid];   // scale [64], shift [64]

  // pooling phase: thread = (pooled pixel tid / 8, 16-byte channel chunk tid % 8); position constants:
  const int pp = tid >> 3, chunk = tid & 7;
  const int pi = pp / PTW, pj = pp - pi * PTW;
  const bool pok = ph0 + pi < p.PH && pw0 + pj < p.PW;
  unsigned valid = 0;                                  // bit kh*3+kw: that conv position exists
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int h = r0 + 2 * pi + kh, w = c0 + 2 * pj + kw;
      if ((unsigned)h < (unsigned)p.OH && (unsigned)w < (unsigned)p.OW) valid |= 1u << (kh * 3 + kw);
    }
  const unsigned char* tbase = sT + ((2 * pi) * FC_ + 2 * pj) * SROW + chunk * 16;
  const size_t pimg = (size_t)p.PH * p.PW * CO;
  const size_t pout = ((size_t)(ph0 + pi) * p.PW + pw0 + pj) * CO + chunk * 8;

  int n = slot;
  if (n < p.N) {
    patch_fetch<P::NPRE>(p, poff, pre, n);
    patch_store<P::NPRE>(patch0, pre);
  }
  __syncthreads();
  for (int cur = 0; n < p.N; n += nslots, cur ^= 1) {
    const bf16_t* patch = patch0 + cur * P::ELEMS;
    const bool has_next = n + nslots < p.N;
    if (has_next) patch_fetch<P::NPRE>(p, poff, pre, n + nslots);
    f32x4 acc[4][FNB];
    conv_tile<FNB>(acc, patch, wlane, WS, ksteps, goff2, boff);
    // conv (rounded to bf16 like the stored tensor) -> bn -> relu -> rounded: 4 channels = 8 bytes per (a, b)
#pragma unroll
    for (int b = 0; b < FNB; ++b) {
      if (trow[b] >= 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const f32x4 sc4 = *reinterpret_cast<const f32x4*>(sC + a * 16 + fq * 4);
          const f32x4 sh4 = *reinterpret_cast<const f32x4*>(sC + CO + a * 16 + fq * 4);
          float f[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) f[j] = fmaxf(bf2f(f2bf(acc[a][b][j])) * sc4[j] + sh4[j], 0.f);
          uint2 o;
          o.x = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
          o.y = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
          *reinterpret_cast<uint2*>(sT + trow[b] * SROW + (a * 16 + fq * 4) * 2) = o;
        }
      }
    }
    if (has_next) patch_store<P::NPRE>(patch0 + (cur ^ 1) * P::ELEMS, pre);
    __syncthreads();   // tile complete (and the other patch buffer)
    if (pok) {
      unsigned key[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) key[j] = 0u;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          if ((valid >> (kh * 3 + kw)) & 1u) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(tbase + (kh * FC_ + kw) * SROW);
            const unsigned tcode = 15u - (unsigned)(kh * 3 + kw);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              key[2 * q] = max(key[2 * q], ((r[q] & 0x7fffu) << 4) | tcode);
              key[2 * q + 1] = max(key[2 * q + 1], (((r[q] >> 16) & 0x7fffu) << 4) | tcode);
            }
          }
        }
      u32x4 best;
      unsigned pk[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) best[q] = (key[2 * q] >> 4) | ((key[2 * q + 1] >> 4) << 16);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        pk[q] = (15u - (key[4 * q] & 15u)) | ((15u - (key[4 * q + 1] & 15u)) << 8) | ((15u - (key[4 * q + 2] & 15u)) << 16) |
                ((15u - (key[4 * q + 3] & 15u)) << 24);
      bf16_t* po = (bf16_t*)p.pooled + (size_t)n * pimg + pout;
      *reinterpret_cast<u32x4*>(po) = best;
      *reinterpret_cast<uint2*>(p.idx + (size_t)n * pimg + pout) = make_uint2(pk[0], pk[1]);
    }
    __syncthreads();   // everyone is done reading the tile
  }
}

// ====================================================================================================
// backward: conv recomputed -> max-pool backward gathered -> BatchNorm-backward apply -> weight gradient
// ====================================================================================================
constexpr int BTH = 8, BTW = 16;                       // conv outputs per tile (as stem_wgrad_kernel)
constexpr int WTH = BTH / 2 + 1, WTW = BTW / 2 + 1;    // pooled windows that reach the tile: 5 x 9
constexpr int WPIX = WTH * WTW;                        // 45
constexpr int NWIN = (WPIX * 8 + 255) / 256;           // (window, 16-B chunk) items per thread: 2
constexpr int GROW = 144;                              // LDS row stride of the staged pooled gradient (bf16 x 64 + pad)
constexpr int IROW = 72;                               // ... of the staged arg-max bytes (64 + pad)

__global__ __launch_bounds__(256, 2) void stem_bwd_wgrad_kernel(FusedParams p) {
  using P = Patch<BTH, BTW>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ksteps = (p.NG + 3) / 4, KP = ksteps * 32, WS = KP + 8;
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem);
  const size_t w_bytes = (((size_t)CO * WS * 2) + 15) & ~(size_t)15;
  bf16_t* const patch = reinterpret_cast<bf16_t*>(smem + w_bytes);
  unsigned char* const sDY = smem + w_bytes + (size_t)P::ELEMS * 2;    // [128][SROW] bf16
  unsigned char* const sG = sDY + 128 * SROW;                           // [WPIX][GROW]: dp * [p > 0], bf16
  unsigned char* const sI = sG + WPIX * GROW;                           // [WPIX][IROW]: arg-max taps
  float* const sK = reinterpret_cast<float*>(sI + WPIX * IROW);         // [3][64]: k1, bn, an (registers are for occupancy)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  stage_weights(sW, p.wpk, KP, WS);
  const int NK = p.NG * 8;
  const int ktiles = (NK + 15) / 16;

  const int tpi = p.tiles_h * p.tiles_w;
  const int vb = xcd_order();
  const int pos = vb % tpi, slot = vb / tpi, nslots = gridDim.x / tpi;
  const int th_i = pos / p.tiles_w, tw_i = pos - th_i * p.tiles_w;
  const int oh0 = th_i * BTH, ow0 = tw_i * BTW;         // both even
  const int wh0 = oh0 / 2, ww0 = ow0 / 2;               // first pooled window of the tile

  float pre[P::NPRE];
  unsigned poff[P::NPRE];
  patch_offsets<BTH, BTW>(p, oh0, ow0, poff);

  // ---- forward recompute: this lane's two segments = conv rows 2*wave + b, column fr (stem_fwd_kernel's mapping)
  int boff[2];
  bool okb[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    boff[b] = (wave * 2 + b) * 2 * P::PWS + fr * 2;
    okb[b] = (oh0 + wave * 2 + b < p.OH) & (ow0 + fr < p.OW);
  }
  unsigned goff2[KSTEPS_MAX / 2];
  group_offsets<P::PHh, P::PWS>(p.NG, fq, goff2);
  const bf16_t* wlane = sW + fr * WS + fq * 8;

  // ---- BatchNorm-backward coefficients of the lane's channels: dy = k1 * dz + (bn * y + an)
  if (tid < CO) {
    const int c = tid;
    const float mean = p.coef[2 * CO + c], inv = p.coef[3 * CO + c];
    const float k1 = p.bcoef[c];
    const float bn = -k1 * p.bcoef[2 * CO + c] * inv;
    sK[c] = k1;
    sK[CO + c] = bn;
    sK[2 * CO + c] = -k1 * p.bcoef[CO + c] - bn * mean;
  }
  // ---- windows of the lane's positions.  Row 2*wave + b has parity b: even rows lie in window row k = h/2 only (tap
  // row 1), odd rows in k (tap row 2) and k + 1 (tap row 0); columns likewise with fr.  Window list in the order of
  // maxpool_relu_bwd_kernel's table (w00, w01, w10, w11); a window that does not apply gets tap code 255 (never stored).
  const int wr0 = wave;                                   // tile-relative window row of (2*wave + b) / 2
  const int wc0 = fr >> 1;
  const bool codd = fr & 1;
  // tap = kh * 3 + kw;  b = 0: kh = 1 (second window row unused);  b = 1: kh = 2 in row wr0, kh = 0 in row wr0 + 1
  // even column: kw = 1 (second window column unused);  odd column: kw = 2 in column wc0, kw = 0 in column wc0 + 1
  const int kw_lo = codd ? 2 : 1;
  const unsigned goffc = (unsigned)(wc0 * GROW), ioffc = (unsigned)(wc0 * IROW);

  // ---- weight gradient (stem_wgrad_kernel): wave w owns k-tiles {w, w+4, w+8} x all 4 co-tiles
  f32x4 wacc[4][3];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) wacc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int koff[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int kidx = (wave + 4 * b) * 16 + fr;
    const bool kok = (wave + 4 * b) < ktiles && kidx < NK;
    const int kk = kok ? kidx : 0;
    const int G = kk >> 3, c = G / R7;
    koff[b] = (c * P::PHh + (G - c * R7)) * P::PWS + (kk & 7);
  }

  // ---- pooled-window staging: item i = tid + 256 * k -> (window i / 8, chunk i % 8)
  unsigned woff[NWIN];   // element offset inside one image's pooled tensor, or out of range
  int wlds[NWIN];        // window index (LDS row), or -1
#pragma unroll
  for (int k = 0; k < NWIN; ++k) {
    const int i = tid + 256 * k;
    const int wi = i >> 3, ch = i & 7;
    const int wr = wi / WTW, wc = wi - wr * WTW;
    const bool ok = wi < WPIX && wh0 + wr < p.PH && ww0 + wc < p.PW;
    wlds[k] = wi < WPIX ? wi : -1;
    woff[k] = ok ? (unsigned)(((wh0 + wr) * p.PW + ww0 + wc) * CO + ch * 8) : 0xFFFFFFFFu;
  }
  u32x4 rdp[NWIN], rpp[NWIN];
  uint2 rix[NWIN];
  const size_t pimg = (size_t)p.PH * p.PW * CO;
  auto fetch = [&](int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t r_dp = __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.dp + (size_t)n * pimg), 0, (int)(pimg * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_pp = __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.pooled + (size_t)n * pimg), 0, (int)(pimg * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_ix = __builtin_amdgcn_make_buffer_rsrc((void*)(p.idx + (size_t)n * pimg), 0, (int)pimg, 0x00020000);
#pragma unroll
    for (int k = 0; k < NWIN; ++k) {
      const unsigned o = woff[k];
      rdp[k] = __builtin_amdgcn_raw_buffer_load_b128(r_dp, (int)(o == 0xFFFFFFFFu ? o : o * 2u), 0, 0);
      rpp[k] = __builtin_amdgcn_raw_buffer_load_b128(r_pp, (int)(o == 0xFFFFFFFFu ? o : o * 2u), 0, 0);
      const auto t2 = __builtin_amdgcn_raw_buffer_load_b64(r_ix, (int)o, 0, 0);
      rix[k] = make_uint2(t2[0], t2[1]);
    }
#endif
    patch_fetch<P::NPRE>(p, poff, pre, n);
  };

  if (slot < p.N) fetch(slot);
  for (int n = slot; n < p.N; n += nslots) {
    __syncthreads();   // the previous tile's LDS (patch, dy, windows) fully consumed
    patch_store<P::NPRE>(patch, pre);
#pragma unroll
    for (int k = 0; k < NWIN; ++k) {
      if (wlds[k] >= 0) {
        // g = dp * [p > 0] (exact in bf16: a masked copy); out-of-range windows were zero-filled: p = 0 -> g = 0
        u32x4 g;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned pv = rpp[k][q], dv = rdp[k][q];
          const unsigned lo = (pv & 0x7fffu) != 0u && !(pv & 0x8000u) ? (dv & 0xffffu) : 0u;
          const unsigned hi = ((pv >> 16) & 0x7fffu) != 0u && !(pv & 0x80000000u) ? (dv & 0xffff0000u) : 0u;
          g[q] = lo | hi;
        }
        const int ch = (tid + 256 * k) & 7;
        *reinterpret_cast<u32x4*>(sG + wlds[k] * GROW + ch * 16) = g;
        *reinterpret_cast<uint2*>(sI + wlds[k] * IROW + ch * 8) = rix[k];
      }
    }
    __syncthreads();
    if (n + nslots < p.N) fetch(n + nslots);   // streams in underneath this tile's work

    // (1) conv recomputed
    f32x4 acc[4][2];
    conv_tile<2>(acc, patch, wlane, WS, ksteps, goff2, boff);
    // (2) dz gathered from the windows, dy = k1 * dz + (bn * y + an), rounded, into sDY
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int prow = wave * 2 + b;
      const int px = prow * BTW + fr;
      // window rows / taps for this b
      const int kh_lo = b == 0 ? 1 : 2;
      const unsigned grow0 = (unsigned)((wr0 * WTW) * GROW) + goffc, irow0 = (unsigned)((wr0 * WTW) * IROW) + ioffc;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const unsigned cb = (unsigned)(a * 16 + fq * 4);
        float dz[4] = {0.f, 0.f, 0.f, 0.f};
        // windows in table order: (row lo, col lo), (row lo, col hi), (row hi, col lo), (row hi, col hi)
#pragma unroll
        for (int wr = 0; wr < 2; ++wr) {
          if (wr == 1 && b == 0) continue;                 // even conv row: one window row
          const int kh = wr == 0 ? kh_lo : 0;
#pragma unroll
          for (int wc = 0; wc < 2; ++wc) {
            // (odd column: kw = 2 in the low window, 0 in the high one; even column: kw = 1, high window unused)
            const int kw = wc == 0 ? kw_lo : 0;
            const unsigned tap = (wc == 1 && !codd) ? 255u : (unsigned)(kh * 3 + kw);
            const unsigned go = grow0 + (unsigned)((wr * WTW + wc) * GROW) + cb * 2u;
            const unsigned io = irow0 + (unsigned)((wr * WTW + wc) * IROW) + cb;
            const uint2 gv = *reinterpret_cast<const uint2*>(sG + go);
            const unsigned iv = *reinterpret_cast<const unsigned*>(sI + io);
            const float g0 = __uint_as_float(gv.x << 16), g1 = __uint_as_float(gv.x & 0xffff0000u);
            const float g2 = __uint_as_float(gv.y << 16), g3 = __uint_as_float(gv.y & 0xffff0000u);
            dz[0] += ((iv & 0xffu) == tap) ? g0 : 0.f;
            dz[1] += (((iv >> 8) & 0xffu) == tap) ? g1 : 0.f;
            dz[2] += (((iv >> 16) & 0xffu) == tap) ? g2 : 0.f;
            dz[3] += ((iv >> 24) == tap) ? g3 : 0.f;
          }
        }
        uint2 o = make_uint2(0u, 0u);
        if (okb[b]) {
          const f32x4 k1v = *reinterpret_cast<const f32x4*>(sK + cb);
          const f32x4 bnv = *reinterpret_cast<const f32x4*>(sK + CO + cb);
          const f32x4 anv = *reinterpret_cast<const f32x4*>(sK + 2 * CO + cb);
          float d[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float yv = bf2f(f2bf(acc[a][b][j]));     // the value the stored conv output would have held
            d[j] = k1v[j] * dz[j] + (bnv[j] * yv + anv[j]);
          }
          o.x = (unsigned)f2bf(d[0]) | ((unsigned)f2bf(d[1]) << 16);
          o.y = (unsigned)f2bf(d[2]) | ((unsigned)f2bf(d[3]) << 16);
        }
        *reinterpret_cast<uint2*>(sDY + px * SROW + cb * 2u) = o;
      }
    }
    __syncthreads();
    // (3) weight gradient of this tile: A = dy^T (transposing LDS reads), B = 8 consecutive pixels (stride 2) of one kidx
    {
      const int q = fr >> 2, pq = fr & 3;
      for (int st = 0; st < 4; ++st) {
        u32x4 fa[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const unsigned char* a0 = sDY + (st * 32 + 8 * fq + q) * SROW + (a * 16 + 4 * pq) * 2;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(a0)));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(a0 + 4 * SROW)));
          uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fa[a] = (u32x4){l2.x, l2.y, h2.x, h2.y};
        }
        const int pix0 = st * 32 + 8 * fq;
        const int trow = pix0 / BTW, tcol0 = pix0 % BTW;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const bf16_t* src = patch + koff[b] + trow * 2 * P::PWS + tcol0 * 2;
          u32x4 fb;
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[j] = (unsigned)src[4 * j] | ((unsigned)src[4 * j + 2] << 16);
#pragma unroll
          for (int a = 0; a < 4; ++a) mfma_bf16_inplace(wacc[a][b], fa[a], fb);
        }
      }
    }
  }
  mfma_drain();
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int kidx = (wave + 4 * b) * 16 + fr;
    if ((wave + 4 * b) >= ktiles || kidx >= NK) continue;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) p.slab[((size_t)blockIdx.x * CO + a * 16 + fq * 4 + j) * NK + kidx] = wacc[a][b][j];
  }
}

int fill(FusedParams& p, int N, int Cin, int H, int W) {
  memset(&p, 0, sizeof(p));
  if (Cin < 1 || Cin > 3) ECG_FAIL(ECGMM_ERR_SHAPE, "fused stem: Cin=%d (1..3)", Cin);
  if (H < 8 || W < 8) ECG_FAIL(ECGMM_ERR_SHAPE, "fused stem: %dx%d input too small", H, W);
  p.N = N; p.Cin = Cin; p.H = H; p.W = W;
  p.OH = (H + 6 - 7) / 2 + 1; p.OW = (W + 6 - 7) / 2 + 1;
  p.PH = (p.OH - 1) / 2 + 1; p.PW = (p.OW - 1) / 2 + 1;
  p.NG = Cin * 7;
  if ((double)Cin * H * W * 4.0 >= 2147483648.0 || (double)p.OH * p.OW * CO * 4.0 >= 2147483648.0)
    ECG_FAIL(ECGMM_ERR_SHAPE, "fused stem: one sample of %d x %d x %d exceeds the 2 GiB buffer-addressing range", Cin, H, W);
  return 0;
}

int grid_for(int N, int tpi, int target) {
  int nslots = target / tpi;
  nslots = nslots < 1 ? 1 : (nslots > N ? N : nslots);
  return tpi * nslots;
}

}  // namespace

bool ecg_stem_fused_ok(int dtype, int Cin, int R) { return dtype == ECGMM_BF16 && R == 7 && Cin >= 1 && Cin <= 3; }

// image -> pooled [N, PH, PW, 64] bf16 + arg-max bytes, coef = forward BatchNorm coefficients [4][64]
int ecg_stem_pool_fwd(const float* x, const void* wpk, const float* coef, void* pooled, unsigned char* idx, int N, int Cin,
                      int H, int W, hipStream_t stream) {
  FusedParams p;
  ECG_TRY(fill(p, N, Cin, H, W));
  p.x = x; p.wpk = wpk; p.coef = coef; p.pooled = pooled; p.idx = idx;
  p.tiles_h = ceil_div(p.PH, PTH); p.tiles_w = ceil_div(p.PW, PTW);
  using P = Patch<FR_, FC_>;
  const int ksteps = (p.NG + 3) / 4, WS = ksteps * 32 + 8;
  const size_t lds = align_up((size_t)CO * WS * 2, 16) + 2 * (size_t)P::ELEMS * 2 + (size_t)FPIX * SROW + 2 * CO * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)stem_pool_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    once = true;
  }
  const int grid = grid_for(N, p.tiles_h * p.tiles_w, 2048);
  ecg_prof_begin(ECG_PROF_STEM_FWD, 2.0 * (double)N * p.OH * p.OW * CO * Cin * 49, 4.0 * N * Cin * H * W + 3.0 * N * p.PH * p.PW * CO, stream);
  hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3(grid), dim3(256), lds, stream, p);
  ecg_prof_end(stream);
  ECG_CHECK_LAUNCH("stem_pool_fwd");
  return 0;
}

static int bwd_grid(int N, int tpi) { return grid_for(N, tpi, 1024); }

size_t ecg_stem_bwd_wgrad_workspace(int N, int Cin, int H, int W) {
  FusedParams p;
  if (fill(p, N, Cin, H, W)) return 0;
  const int tpi = ceil_div(p.OH, BTH) * ceil_div(p.OW, BTW);
  return (size_t)bwd_grid(N, tpi) * CO * p.NG * 8 * sizeof(float);
}

// Slab of weight-gradient partial tiles [rows][64][NG * 8]; *rows_out = rows.  The caller folds them
// (stem_wgrad_reduce_kernel through ecg_stem_wgrad_reduce).
int ecg_stem_bwd_wgrad(const float* x, const void* wpk, const float* coef, const float* bcoef, const void* dp,
                       const void* pooled, const unsigned char* idx, float* slab, size_t slab_bytes, int* rows_out, int N,
                       int Cin, int H, int W, hipStream_t stream) {
  FusedParams p;
  ECG_TRY(fill(p, N, Cin, H, W));
  p.x = x; p.wpk = wpk; p.coef = coef; p.bcoef = bcoef; p.dp = dp; p.pooled = const_cast<void*>(pooled);
  p.idx = const_cast<unsigned char*>(idx); p.slab = slab;
  p.tiles_h = ceil_div(p.OH, BTH); p.tiles_w = ceil_div(p.OW, BTW);
  const int grid = bwd_grid(N, p.tiles_h * p.tiles_w);
  const size_t need = (size_t)grid * CO * p.NG * 8 * sizeof(float);
  if (!slab || slab_bytes < need) ECG_FAIL(ECGMM_ERR_WORKSPACE, "fused stem backward: workspace %zu < %zu", slab_bytes, need);
  using P = Patch<BTH, BTW>;
  const int ksteps = (p.NG + 3) / 4, WS = ksteps * 32 + 8;
  const size_t lds = align_up((size_t)CO * WS * 2, 16) + (size_t)P::ELEMS * 2 + 128 * (size_t)SROW + (size_t)WPIX * GROW + (size_t)WPIX * IROW + 3 * CO * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)stem_bwd_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    once = true;
  }
  ecg_prof_begin(ECG_PROF_STEM_WGRAD, 4.0 * (double)N * p.OH * p.OW * CO * Cin * 49, 4.0 * N * Cin * H * W + 5.0 * N * p.PH * p.PW * CO, stream);
  hipLaunchKernelGGL(stem_bwd_wgrad_kernel, dim3(grid), dim3(256), lds, stream, p);
  ecg_prof_end(stream);
  ECG_CHECK_LAUNCH("stem_bwd_wgrad");
  *rows_out = grid;
  return 0;
}

// [BatchNorm -> ReLU -> MaxPool]-backward + conv weight gradient of the stem in one call:
//   reduction over the pooled tensors -> dgamma, dbeta, coefficients;  fused recompute / apply / wgrad -> slab -> dw.
// ws = BatchNorm scratch (ecg_bn_bwd_scratch over the conv output) followed by the slab.  dw nullable (frozen conv).
size_t ecg_stem_pool_bwd_workspace(int N, int Cin, int H, int W) {
  FusedParams p;
  if (fill(p, N, Cin, H, W)) return 0;
  return align_up(ecg_bn_bwd_scratch(ECGMM_BF16, (long)N * p.OH * p.OW, CO), 256) + ecg_stem_bwd_wgrad_workspace(N, Cin, H, W);
}
int ecg_stem_pool_bwd(const float* x, const void* wpk, const float* coef, const float* gamma, const void* dp, const void* pooled,
                      const unsigned char* idx, float* dgamma, float* dbeta, float* dw, void* ws, size_t ws_bytes, int N,
                      int Cin, int H, int W, hipStream_t stream) {
  FusedParams p;
  ECG_TRY(fill(p, N, Cin, H, W));
  const size_t bn_bytes = align_up(ecg_bn_bwd_scratch(ECGMM_BF16, (long)N * p.OH * p.OW, CO), 256);
  const size_t slab_bytes = ecg_stem_bwd_wgrad_workspace(N, Cin, H, W);
  if (!ws || ws_bytes < bn_bytes + slab_bytes) ECG_FAIL(ECGMM_ERR_WORKSPACE, "stem_pool_bwd: workspace %zu < %zu", ws_bytes, bn_bytes + slab_bytes);
  const float* bcoef = nullptr;
  ECG_TRY(ecg_pool_bn_bwd_reduce(ECGMM_BF16, dp, pooled, coef, gamma, dgamma, dbeta, N, p.OH, p.OW, CO, (float*)ws, &bcoef, stream));
  if (!dw) return 0;
  float* slab = reinterpret_cast<float*>((unsigned char*)ws + bn_bytes);
  int rows = 0;
  ECG_TRY(ecg_stem_bwd_wgrad(x, wpk, coef, bcoef, dp, pooled, idx, slab, slab_bytes, &rows, N, Cin, H, W, stream));
  return ecg_stem_wgrad_reduce(slab, dw, rows, p.NG, 0, stream);
}
