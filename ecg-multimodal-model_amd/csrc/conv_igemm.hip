// Implicit-GEMM convolution on MFMA for channels-last (NHWC / NLC) activations, gfx950.
//
//   forward : dst[n,oh,ow,co] = sum_{r,s,ci} src[n, oh*st-p+r, ow*st-p+s, ci] * W[co][r][s][ci]
//   dgrad   : dst[n,h,w,ci]   = sum_{r,s,co} src[n,(h+p-r)/st,(w+p-s)/st,co] * W'[ci][r][s][co]
//
// GEMM view: pixels (n,h,w) are the MFMA *column* index and channels the *row* index, i.e. the packed
// weights are the A operand and the gathered activation rows the B operand.  With that orientation
// each lane of a 16x16 accumulator owns 4 consecutive channels of one pixel, so the epilogue stores
// 8 B (bf16) / 16 B (f32) per lane into the channels-last output and the BatchNorm partial sums are
// a 16-lane butterfly.
//
// Tile: 128 pixels x BN channels x 128 bytes of K per stage (64 bf16 / 32 f32 channels of one tap),
// 256 threads = 4 waves (2 pixel halves x 2 channel halves), double-buffered LDS with register
// staging (issue the next stage's global loads before the MFMAs, write them to LDS after).
// LDS rows are 128 B with a 16-B-chunk XOR swizzle (chunk ^= row & 7) so the ds_read_b128 fragment
// reads of 16 consecutive rows spread over the 64 banks.
// dtype: bf16 -> v_mfma_f32_16x16x32_bf16; f32 -> 4 x v_mfma_f32_16x16x4_f32 per 16-B fragment pair
// (exact fp32 fma chain -- the 1e-3 parity path).
#include "ops.h"

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                  acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), acc, 0, 0, 0);
  }
};

// 16-byte LDS-DMA buffer load (global -> LDS, no VGPR).  The 16-byte form only exists on gfx950; the
// host pass of hipcc checks builtins against its own target, so the body is device-pass only.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_wave_base, unsigned voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, 0, 0,
                                           0);
#endif
}

struct IgemmParams {
  const void* src;
  const void* wpk;
  void* dst;
  const float* bias;
  const void* addend;
  float* stats;  // [2*gridDim.x][2][Cd] partial (sum, sumsq) or null
  int Nimg, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad_h, pad_w;
  int M;  // destination pixels
  int act;
  // stride-2 dgrad only (nullable): the gradient of a 1x1 stride-2 pad-0 convolution of the SAME input (a ResNet
  // downsample branch) -- same [N][Hs][Ws][Cs] shape as src -- and its dgrad-packed weights [Cd][1][Cs]; it is one more
  // "tap" of parity class (0,0) (a 3x3 pad-1 class (0,0) holds the centre tap only: both read source pixel (h/2, w/2))
  const void* src2;
  const void* wpk2;
  // exact division of a (class-local) pixel index by Hc*Wc and by Wc: q = (x * mul) >> sh for x < 2^31, one set per
  // output-pixel parity class (index blockIdx.z; index 0 when the launch has no classes).  A workgroup decodes 4 + TP pixels
  // with two divisions each; with runtime divisors that was ~500 VALU in front of K loops of 2-8 stages (stride-2 dgrad).
  unsigned mul_hw[4], sh_hw[4], mul_w[4], sh_w[4];
};

#ifdef ECG_STAMP
// Diagnostic build only (make stamp): s_memtime brackets around the three segments of a K-loop iteration, summed per
// launch by wave 0 of every workgroup.  Shares, not lengths, are what this build is good for (cdna guide, in-kernel stamps).
__device__ unsigned long long g_stamp[8];
#define ECG_STAMP_AT(t)                                                                   \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#endif

constexpr int BM = 128;       // pixels per workgroup
constexpr int ROWB = 128;     // bytes of K per LDS row per stage
constexpr int NTHREADS = 256;

template <typename T, int BN, int MODE, int ST>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(IgemmParams p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KBE = ROWB / (int)sizeof(T);  // channels per stage
  constexpr int TP = 4;                       // 16-pixel tiles per wave (64 pixels)
  constexpr int TC = BN / 32;                 // 16-channel tiles per wave
  constexpr int NWV = BN / 32;                // weight vectors per thread per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // layout: [2 stages][ (BN + BM) rows ][128 B]
  constexpr int STAGE_BYTES = (BN + BM) * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wp = wave & 1, wc = wave >> 1;  // pixel half, channel half
  // XCD-aware tile order: hardware deals workgroups round-robin over the 8 XCDs, so give each XCD a
  // contiguous run of pixel tiles -- vertically adjacent tiles (which re-read the same input rows for
  // the other filter taps) then share one L2.  Pure speed hint (bijective for any grid size).
  int mt;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = mt * BM, n0 = blockIdx.y * BN;
  const T* __restrict__ src = (const T*)p.src;
  const T* __restrict__ wpk = (const T*)p.wpk;

  // ---- gather bookkeeping.  All global reads are LDS-DMA buffer loads (buffer_load_dwordx4 ... lds):
  // a 32-bit byte offset from a per-workgroup base (the sample of the tile's first pixel), hardware
  // zero-fill for anything out of range (padding, ragged tiles, partial channel stages), no VGPR
  // staging and no ds_write.  One wave-instruction fills 1 KiB = 8 LDS rows x 128 B; the XOR swizzle
  // is applied on the SOURCE side (lane -> logical chunk), the LDS image stays lane-linear.
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int lrow8 = lane >> 3;                    // row within an 8-row DMA piece
  const int lchunk = (lane & 7) ^ (lrow8 & 7);    // logical 16-B chunk this lane fetches
  // Stride-2 dgrad runs one grid slice (blockIdx.z) per output-pixel parity class (h&1, w&1): inside a
  // class every pixel sees the same subset of filter taps (r = r0, r0+2, ...), so no MFMA work is spent
  // on taps that fall between the strided samples.  Pixels are enumerated over the class sub-lattice.
  constexpr bool PAR = (MODE == 1 && ST == 2);
  constexpr int TSTEP = PAR ? 2 : 1;
  int ph = 0, pw = 0, Hc = p.Hd, Wc = p.Wd, r0 = 0, s0 = 0, Rc = p.R, Sc = p.S, Mc = p.M;
  // (grid slices are dispatched in blockIdx.z order: the class with the MOST taps -- (1,1): four of a 3x3 filter -- goes first and
  //  the one-tap class last, so the launch's tail is made of its shortest workgroups: batch 256, layers 2 / 3 / 4 input gradient
  //  87 / 63 / 66 -> 83 / 59 / 54 us.  One-row tensors (the 1-D encoder: classes 2 and 3 are empty) keep the plain order, which
  //  measured 10 % faster there.)
  const int cz = PAR ? (p.Hd > 1 ? 3 - (int)blockIdx.z : (int)blockIdx.z) : 0;
  if (PAR) {
    ph = cz >> 1;
    pw = cz & 1;
    Hc = (p.Hd - ph + 1) >> 1;
    Wc = (p.Wd - pw + 1) >> 1;
    r0 = (ph + p.pad_h) & 1;
    s0 = (pw + p.pad_w) & 1;
    Rc = p.R > r0 ? (p.R - r0 + 1) >> 1 : 0;
    Sc = p.S > s0 ? (p.S - s0 + 1) >> 1 : 0;
    Mc = p.Nimg * Hc * Wc;
    if (m0 >= Mc) return;  // block-uniform
  }
  const int HWd = Hc * Wc;
  const int cls = PAR ? cz : 0;
  const unsigned mhw = p.mul_hw[cls], shw = p.sh_hw[cls], mw = p.mul_w[cls], sw_ = p.sh_w[cls];
  auto div_hw = [&](int x) -> int { return (int)(((unsigned long long)(unsigned)x * mhw) >> shw); };
  auto div_w = [&](int x) -> int { return (int)(((unsigned long long)(unsigned)x * mw) >> sw_); };
  const int n_first = div_hw(m0);
  const size_t img_elems = (size_t)p.Hs * p.Ws * p.Cs;
  const size_t left = ((size_t)p.Nimg - n_first) * img_elems * sizeof(T);
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(src + (size_t)n_first * img_elems), 0, left > 0xFFFFFFF0ull ? (int)0xFFFFFFF0u : (int)left, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      (void*)wpk, 0, (int)((size_t)p.Cd * p.R * p.S * p.Cs * sizeof(T)), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFFFu;
  // Per gathered row: source-space coordinates (hq, wq) and byte offset (offb) of filter tap (0,0); tap
  // (g_r, g_s) then sits at hq + SGN*g_r, wq + SGN*g_s, i.e. a wave-uniform byte delta away.
  constexpr int SGN = MODE == 0 ? 1 : -1;
  const unsigned pix_bytes = (unsigned)p.Cs * (unsigned)sizeof(T);
  int hq[4], wq[4];
  unsigned offb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int pix = m0 + wave * 32 + i * 8 + lrow8;
    bool rok = pix < Mc;
    int pp = rok ? pix : m0;
    int n = div_hw(pp);
    int rem = pp - n * HWd;
    int hd = div_w(rem), wd = rem - hd * Wc;
    if (PAR) {
      hd = hd * 2 + ph;
      wd = wd * 2 + pw;
    }
    if (MODE == 0) {
      hq[i] = hd * p.stride - p.pad_h;
      wq[i] = wd * p.stride - p.pad_w;
    } else if (ST == 2) {  // (hd + pad - r0, wd + pad - s0 are even by construction of the parity class)
      hq[i] = (hd + p.pad_h - r0) >> 1;
      wq[i] = (wd + p.pad_w - s0) >> 1;
    } else {
      hq[i] = hd + p.pad_h;
      wq[i] = wd + p.pad_w;
    }
    offb[i] = (unsigned)(((n - n_first) * p.Hs + hq[i]) * p.Ws + wq[i]) * pix_bytes;
    if (!rok) hq[i] = 1 << 28;  // every tap of a padding row falls out of range
  }
  const int RS = p.R * p.S;
  const int cpt = (p.Cs + KBE - 1) / KBE;  // stages per tap
  const int nk_main = Rc * Sc * cpt;
  const bool has2 = PAR && p.src2 != nullptr && cz == 0;   // block-uniform
  const int nk = nk_main + (has2 ? cpt : 0);
  unsigned wrow[NWV];
#pragma unroll
  for (int i = 0; i < NWV; ++i) {
    int j = n0 + wave * (BN / 4) + i * 8 + lrow8;
    wrow[i] = j < p.Cd ? (unsigned)j * (unsigned)RS * pix_bytes : OOB;
  }
  const __amdgpu_buffer_rsrc_t rs_src2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const T*)(has2 ? p.src2 : p.src) + (size_t)n_first * img_elems), 0,
      left > 0xFFFFFFF0ull ? (int)0xFFFFFFF0u : (int)left, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(has2 ? p.wpk2 : p.wpk), 0, (int)((size_t)p.Cd * p.Cs * sizeof(T)), 0x00020000);
  int g_n = 0;  // stages issued so far (wave-uniform)

  int g_r = 0, g_s = 0, g_cc = 0;  // wave-uniform tap / channel-stage counters (stages are issued in order)
  auto dma = [&](int stage) {
    unsigned char* sW = smem + stage * STAGE_BYTES;
    unsigned char* sX = sW + BN * ROWB;
    if (PAR && g_n >= nk_main) {   // the folded 1x1 branch: source pixel (hq, wq) itself, its own weights [Cd][Cs]
      const int ch = (g_n - nk_main) * KBE + lchunk * VEC;
      const unsigned chb = ch < p.Cs ? (unsigned)ch * (unsigned)sizeof(T) : OOB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = (unsigned)hq[i] < (unsigned)p.Hs && (unsigned)wq[i] < (unsigned)p.Ws;
        dma16(rs_src2, sX + (wv * 4 + i) * 1024, (ok && chb != OOB) ? offb[i] + chb : OOB);
      }
#pragma unroll
      for (int i = 0; i < NWV; ++i)
        dma16(rs_w2, sW + (wv * NWV + i) * 1024, (wrow[i] != OOB && chb != OOB) ? wrow[i] / (unsigned)RS + chb : OOB);
      ++g_n;
      return;
    }
    ++g_n;
    const int r = r0 + g_r * TSTEP, s = s0 + g_s * TSTEP;
    const int ch = g_cc * KBE + lchunk * VEC;
    const unsigned chb = ch < p.Cs ? (unsigned)ch * (unsigned)sizeof(T) : OOB;
    const unsigned tapd = (unsigned)(SGN * (g_r * p.Ws + g_s)) * pix_bytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = (unsigned)(hq[i] + SGN * g_r) < (unsigned)p.Hs && (unsigned)(wq[i] + SGN * g_s) < (unsigned)p.Ws;
      dma16(rs_src, sX + (wv * 4 + i) * 1024, (ok && chb != OOB) ? offb[i] + tapd + chb : OOB);
    }
    const unsigned tapb = (unsigned)(r * p.S + s) * pix_bytes + chb;
#pragma unroll
    for (int i = 0; i < NWV; ++i)
      dma16(rs_w, sW + (wv * NWV + i) * 1024, (wrow[i] != OOB && chb != OOB) ? wrow[i] + tapb : OOB);
    if (++g_cc == cpt) {
      g_cc = 0;
      if (++g_s == Sc) {
        g_s = 0;
        ++g_r;
      }
    }
  };

  f32x4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int stage) {
    const unsigned char* sW = smem + stage * STAGE_BYTES;
    const unsigned char* sX = sW + BN * ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[TC], fb[TP];
      const int lch = ks * 4 + fq;
#pragma unroll
      for (int a = 0; a < TC; ++a) {
        int row = wc * (BN / 2) + a * 16 + fr;
        fa[a] = *reinterpret_cast<const u32x4*>(sW + row * ROWB + ((lch ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        int row = wp * 64 + b * 16 + fr;
        fb[b] = *reinterpret_cast<const u32x4*>(sX + row * ROWB + ((lch ^ (row & 7)) << 4));
      }
      // (raised issue priority over the MFMA block: +5 % on the long-K layers 3-4, a loss on the 16-MFMA stages of BN = 64)
      if (BN == 128) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) Mma<T>::run(acc[a][b], fa[a], fb[b]);
      if (BN == 128) __builtin_amdgcn_s_setprio(0);
    }
  };

  if (nk > 0) {
    dma(0);
    __syncthreads();  // (drains vmcnt: the LDS-DMA of stage 0 has landed for every wave)
  }
#ifdef ECG_STAMP
  unsigned long long sA = 0, sB = 0, sC = 0, t0, t1, t2, t3, tk0, tk1;
  ECG_STAMP_AT(tk0);
#endif
  for (int it = 0; it < nk; ++it) {
#ifdef ECG_STAMP
    ECG_STAMP_AT(t0);
#endif
    if (it + 1 < nk) dma((it + 1) & 1);  // that buffer was last read in iteration it-1, fenced by its barrier
#ifdef ECG_STAMP
    ECG_STAMP_AT(t1);
#endif
    compute(it & 1);
#ifdef ECG_STAMP
    ECG_STAMP_AT(t2);
#endif
    __syncthreads();
#ifdef ECG_STAMP
    ECG_STAMP_AT(t3);
    sA += t1 - t0; sB += t2 - t1; sC += t3 - t2;
#endif
  }
#ifdef ECG_STAMP
  ECG_STAMP_AT(tk1);
#endif

  // ---- epilogue: lane owns pixel (fr) x 4 consecutive channels (fq*4 + j) of each 16x16 tile
  T* __restrict__ dst = (T*)p.dst;
  const T* __restrict__ addend = (const T*)p.addend;
  const bool vec_ok = (p.Cd & 3) == 0;
  // bf16 with whole 32-channel groups in range: the 8-byte packs of two neighbouring channel tiles are exchanged
  // between lane rows (v_permlane16_swap) so that every lane stores 16 B and an instruction writes 64 contiguous
  // bytes per pixel.  (8 B per lane = 32-byte segments: WRITE_SIZE showed ~1.45x the output bytes.)
  const bool wide = sizeof(T) == 2 && (p.Cd & 31) == 0;
#ifdef ECG_STAMP
  unsigned long long te1;
#endif
  // Fast path for the common case -- whole tile in range, bf16, not a parity class (bias / addend / ReLU by uniform branches):
  // straight-line code with no per-lane branches (the generic epilogue below is ~3000 instructions with ~250
  // exec-mask branches; in-kernel stamps showed it holding 12-30 % of a workgroup's lifetime).
  bool fast = false;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (sizeof(T) == 2) {
    fast = wide && m0 + BM <= Mc && n0 + BN <= p.Cd;
    if (fast) {
      // element offset of this lane's pixel of fragment b (a parity class enumerates its own sub-lattice)
      size_t prow[TP];
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        int pix = m0 + wp * 64 + b * 16 + fr;
        if (PAR) {
          const int n = div_hw(pix), rem = pix - n * HWd;
          const int h2 = div_w(rem), w2 = rem - h2 * Wc;
          pix = (n * p.Hd + h2 * 2 + ph) * p.Wd + w2 * 2 + pw;
        }
        prow[b] = (size_t)pix * p.Cd + n0 + wc * (BN / 2);
      }
      float s1[TC][4], s2[TC][4];
      f32x4 bias4[TC];
#pragma unroll
      for (int a = 0; a < TC; ++a) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[a][j] = s2[a][j] = 0.f;
        bias4[a] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + wc * (BN / 2) + a * 16 + fq * 4)
                          : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#ifdef ECG_STAMP
      unsigned long long tf0;
      ECG_STAMP_AT(tf0);
#endif
#pragma unroll
      for (int a = 0; a < TC; a += 2)
#pragma unroll
        for (int b = 0; b < TP; ++b) {
          f32x4 v0 = acc[a][b], v1 = acc[a + 1][b];
          if (p.bias) {
            v0 += bias4[a];
            v1 += bias4[a + 1];
          }
          if (addend) {  // the lane's own 2 x 4 channels of this pixel: two 8-byte loads
            const T* ap = addend + prow[b] + fq * 4;
            const uint2 p0 = *reinterpret_cast<const uint2*>(ap + a * 16), p1 = *reinterpret_cast<const uint2*>(ap + (a + 1) * 16);
            v0 += (f32x4){__uint_as_float(p0.x << 16), __uint_as_float(p0.x & 0xFFFF0000u), __uint_as_float(p0.y << 16),
                          __uint_as_float(p0.y & 0xFFFF0000u)};
            v1 += (f32x4){__uint_as_float(p1.x << 16), __uint_as_float(p1.x & 0xFFFF0000u), __uint_as_float(p1.y << 16),
                          __uint_as_float(p1.y & 0xFFFF0000u)};
          }
          if (p.act == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v0[j] = fmaxf(v0[j], 0.f);
              v1[j] = fmaxf(v1[j], 0.f);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s1[a][j] += v0[j];
            s2[a][j] += v0[j] * v0[j];
            s1[a + 1][j] += v1[j];
            s2[a + 1][j] += v1[j] * v1[j];
          }
          const unsigned x0 = (unsigned)f2bf(v0[0]) | ((unsigned)f2bf(v0[1]) << 16);
          const unsigned y0 = (unsigned)f2bf(v0[2]) | ((unsigned)f2bf(v0[3]) << 16);
          const unsigned x1 = (unsigned)f2bf(v1[0]) | ((unsigned)f2bf(v1[1]) << 16);
          const unsigned y1 = (unsigned)f2bf(v1[2]) | ((unsigned)f2bf(v1[3]) << 16);
          auto lo = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
          auto hi = __builtin_amdgcn_permlane16_swap(y0, y1, false, false);
          const int ch = (fq & 1) ? (a + 1) * 16 + (fq - 1) * 4 : a * 16 + fq * 4;
          *reinterpret_cast<u32x4*>(dst + prow[b] + ch) = (u32x4){lo[0], hi[0], lo[1], hi[1]};
        }
#ifdef ECG_STAMP
      unsigned long long tf1;
      ECG_STAMP_AT(tf1);
      if (threadIdx.x == 0) {
        atomicAdd(&g_stamp[7], tf1 - tf0);   // pair loop: values, packing, swaps, stores
      }
#endif
      if (p.stats) {
        float* srow = p.stats + (size_t)(mt * 2 + wp) * 2 * p.Cd + n0 + wc * (BN / 2) + fq * 4;
#pragma unroll
        for (int a = 0; a < TC; ++a) {
          f32x4 r1, r2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            r1[j] = row16_sum(s1[a][j]);
            r2[j] = row16_sum(s2[a][j]);
          }
          if (fr == 0) {
            *reinterpret_cast<f32x4*>(srow + a * 16) = r1;
            *reinterpret_cast<f32x4*>(srow + p.Cd + a * 16) = r2;
          }
        }
      }
#ifdef ECG_STAMP
      ECG_STAMP_AT(te1);
#endif
    }
  }
#endif
  if (!fast) {
  uint2 opk[TC][TP];
  int wpix[TP];
#pragma unroll
  for (int a = 0; a < TC; ++a) {
    const int ch0 = n0 + wc * (BN / 2) + a * 16 + fq * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ch0 + j < p.Cd) bv[j] = p.bias[ch0 + j];
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      int pix = m0 + wp * 64 + b * 16 + fr;
      const bool pok = pix < Mc;
      if (PAR) {  // class-local index -> destination pixel
        int pp = pok ? pix : m0;
        int n = div_hw(pp);
        int rem = pp - n * HWd;
        int h2 = div_w(rem), w2 = rem - h2 * Wc;
        pix = (n * p.Hd + h2 * 2 + ph) * p.Wd + w2 * 2 + pw;
      }
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j] + bv[j];
      if (addend && pok) {
        const T* ap = addend + (size_t)pix * p.Cd + ch0;
        if (vec_ok && ch0 + 3 < p.Cd) {  // 4 consecutive channels: one 8-B (bf16) / 16-B (f32) load
          if (sizeof(T) == 2) {
            uint2 pk = *reinterpret_cast<const uint2*>(ap);
            v[0] += __uint_as_float(pk.x << 16);
            v[1] += __uint_as_float(pk.x & 0xFFFF0000u);
            v[2] += __uint_as_float(pk.y << 16);
            v[3] += __uint_as_float(pk.y & 0xFFFF0000u);
          } else {
            float4 f4 = *reinterpret_cast<const float4*>(ap);
            v[0] += f4.x; v[1] += f4.y; v[2] += f4.z; v[3] += f4.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (ch0 + j < p.Cd) v[j] += Elem<T>::ld(ap + j);
        }
      }
      if (p.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (wide) {
        opk[a][b].x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        opk[a][b].y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        wpix[b] = pok ? pix : -1;
      }
      if (pok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s1[j] += v[j];
          s2[j] += v[j] * v[j];
        }
        T* o = dst + (size_t)pix * p.Cd + ch0;
        if (wide) {
        } else if (vec_ok && ch0 + 3 < p.Cd) {
          if (sizeof(T) == 2) {
            uint2 pk;
            pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
            pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(o) = pk;
          } else {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (ch0 + j < p.Cd) Elem<T>::st(o + j, v[j]);
        }
      }
    }
    if (p.stats) {
      // butterfly over the 16 pixel lanes, then lane fr==0 of each quad writes 4 channels
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] = row16_sum(s1[j]);
        s2[j] = row16_sum(s2[j]);
      }
      if (fr == 0) {
        float* row = p.stats + (size_t)(mt * 2 + wp) * 2 * p.Cd;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ch0 + j < p.Cd) {
            row[ch0 + j] = s1[j];
            row[p.Cd + ch0 + j] = s2[j];
          }
      }
    }
  }
#ifdef ECG_STAMP
  ECG_STAMP_AT(te1);
#endif
  if (wide) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int a = 0; a < TC; a += 2)
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        // rows of 16 lanes = fq: after the swaps an even-fq lane holds [its own tile-a pack | its right neighbour's],
        // an odd-fq lane [its left neighbour's tile-(a+1) pack | its own]: 8 consecutive channels either way
        auto lo = __builtin_amdgcn_permlane16_swap(opk[a][b].x, opk[a + 1][b].x, false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(opk[a][b].y, opk[a + 1][b].y, false, false);
        const int ch = n0 + wc * (BN / 2) + ((fq & 1) ? (a + 1) * 16 + (fq - 1) * 4 : a * 16 + fq * 4);
        if (wpix[b] >= 0 && n0 + wc * (BN / 2) + a * 16 < p.Cd)  // (Cd % 32 == 0: a pair is wholly in or out of range)
          *reinterpret_cast<u32x4*>(dst + (size_t)wpix[b] * p.Cd + ch) = (u32x4){lo[0], hi[0], lo[1], hi[1]};
      }
#endif
  }
  }  // generic epilogue
#ifdef ECG_STAMP
  {
    unsigned long long te;
    ECG_STAMP_AT(te);
    if (threadIdx.x == 0) {
      atomicAdd(&g_stamp[0], sA);
      atomicAdd(&g_stamp[1], sB);
      atomicAdd(&g_stamp[2], sC);
      atomicAdd(&g_stamp[3], te1 - tk1);    // epilogue part 1: values, BatchNorm sums, packing
      atomicAdd(&g_stamp[4], tk1 - tk0);    // K loop
      atomicAdd(&g_stamp[5], te - tk1);     // epilogue
      atomicAdd(&g_stamp[6], 1ull);         // workgroups
    }
  }
#endif
}

#ifdef ECG_STAMP
extern "C" int ecgmm_stamp_read(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamp), sizeof(g_stamp)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

template <typename T, int BN, int MODE, int ST>
int launch_one(const IgemmParams& p, hipStream_t stream) {
  dim3 grid(ceil_div(p.M, BM), ceil_div(p.Cd, BN));
  if (MODE == 1 && ST == 2) {  // 4 parity classes; x sized for the largest one
    long mc = (long)p.Nimg * ((p.Hd + 1) / 2) * ((p.Wd + 1) / 2);
    grid = dim3(ceil_div(mc, BM), ceil_div(p.Cd, BN), 4);
  }
  size_t lds = 2 * (size_t)(BN + BM) * ROWB;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)igemm_kernel<T, BN, MODE, ST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((igemm_kernel<T, BN, MODE, ST>), grid, dim3(NTHREADS), lds, stream, p);
  ECG_CHECK_LAUNCH("igemm_kernel");
  return 0;
}

template <typename T>
int launch_T(const IgemmParams& p, int mode, hipStream_t stream) {
  const bool wide = p.Cd > 64;
  if (mode == 0) return wide ? launch_one<T, 128, 0, 1>(p, stream) : launch_one<T, 64, 0, 1>(p, stream);
  if (p.stride == 1) return wide ? launch_one<T, 128, 1, 1>(p, stream) : launch_one<T, 64, 1, 1>(p, stream);
  if (p.stride == 2) return wide ? launch_one<T, 128, 1, 2>(p, stream) : launch_one<T, 64, 1, 2>(p, stream);
  ECG_FAIL(ECGMM_ERR_SHAPE, "conv dgrad: stride %d unsupported (1 or 2)", p.stride);
}

}  // namespace

// rows of the BatchNorm partial-sum buffer a forward launch writes (2 per 128-pixel tile)
int ecg_conv_stats_rows(long M) { return 2 * ceil_div(M, BM); }

static inline double conv_flops(const ConvGeom& g) {
  return 2.0 * (double)g.N * g.OH * g.OW * g.Cout * g.R * g.S * g.Cin;
}

// mode 0: forward (src = x, dst = y); mode 1: dgrad (src = dy, dst = dx; g still describes the FORWARD conv)
int ecg_conv_igemm(int dtype, int mode, const ConvGeom& g, const void* src, const void* wpk, void* dst,
                   const float* bias, const void* addend, float* stats, int act, hipStream_t stream, ConvEpi* epi) {
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = src; p.wpk = wpk; p.dst = dst; p.bias = bias; p.addend = addend; p.stats = stats; p.act = act;
  p.Nimg = g.N; p.R = g.R; p.S = g.S; p.stride = g.stride; p.pad_h = g.pad_h; p.pad_w = g.pad_w;
  if (mode == 0) {
    p.Hs = g.H; p.Ws = g.W; p.Cs = g.Cin; p.Hd = g.OH; p.Wd = g.OW; p.Cd = g.Cout;
  } else {
    p.Hs = g.OH; p.Ws = g.OW; p.Cs = g.Cout; p.Hd = g.H; p.Wd = g.W; p.Cd = g.Cin;
  }
  long M = (long)g.N * p.Hd * p.Wd;
  if (M <= 0 || M > 0x7fffffffL) ECG_FAIL(ECGMM_ERR_SHAPE, "conv: pixel count %ld out of range", M);
  {
    auto magic = [](unsigned d, unsigned& m, unsigned& sh) {   // exact for dividends < 2^31
      if (d < 1) d = 1;
      int l = 0;
      while ((1u << l) < d) ++l;
      m = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
      sh = 31 + l;
    };
    const bool par = mode == 1 && g.stride == 2;
    for (int c = 0; c < 4; ++c) {
      const int ph_ = par ? c >> 1 : 0, pw_ = par ? c & 1 : 0;
      const int Hc_ = par ? (p.Hd - ph_ + 1) >> 1 : p.Hd, Wc_ = par ? (p.Wd - pw_ + 1) >> 1 : p.Wd;
      magic((unsigned)(Hc_ * Wc_), p.mul_hw[c], p.sh_hw[c]);
      magic((unsigned)Wc_, p.mul_w[c], p.sh_w[c]);
    }
  }
  p.M = (int)M;
  const int vec = dtype == ECGMM_BF16 ? 8 : 4;
  if (p.Cs % vec != 0) ECG_FAIL(ECGMM_ERR_SHAPE, "conv: reduction channels %d not a multiple of %d", p.Cs, vec);
  if (dtype != ECGMM_BF16 && dtype != ECGMM_F32) ECG_FAIL(ECGMM_ERR_DTYPE, "conv: bad dtype %d", dtype);
  {  // algorithmic bytes: source and destination tensors once each (+ addend), + the packed weights
    const double esz = (double)dtype_size(dtype);
    double bytes = esz * ((double)g.N * p.Hs * p.Ws * p.Cs + (double)M * p.Cd * (addend ? 2.0 : 1.0) +
                          (double)g.R * g.S * g.Cin * g.Cout);
    if (epi && epi->red_y) bytes += esz * (double)M * p.Cd * (epi->red_mask ? 2.0 : 1.0);  // fused BN-backward reduction operands
    if (epi && epi->src2 && mode == 1 && g.stride == 2) bytes += esz * ((double)g.N * p.Hs * p.Ws * p.Cs + (double)g.Cin * g.Cout);
    ecg_prof_begin(dtype == ECGMM_F32 ? (mode == 0 ? ECG_PROF_IGEMM_F32_FWD : ECG_PROF_IGEMM_F32_DGRAD)
                                      : (mode == 0 ? ECG_PROF_IGEMM_FWD : ECG_PROF_IGEMM_DGRAD),
                   conv_flops(g), bytes, stream);
  }
  if (epi) epi->src2_done = 0;
  if (epi && epi->src2 && mode == 1 && g.stride == 2 && g.pad_h <= 1 && g.pad_w <= 1 && !addend) {
    p.src2 = epi->src2; p.wpk2 = epi->wpk2;
    epi->src2_done = 1;
  }
  int rc;
  if (ecg_conv_halo_ok(dtype, mode, g)) {  // stride-1 3x3 / 1x3 on whole 256-pixel tiles: halo-resident kernel (conv_halo.hip)
    rc = ecg_conv_halo(mode, g, src, wpk, dst, bias, addend, stats, act, epi, stream);
  } else {
    if (epi) {
      epi->stats_rows = ecg_conv_stats_rows(M);
      epi->red_done = 0;
      epi->red_rows_n = 0;
    }
    rc = dtype == ECGMM_BF16 ? launch_T<bf16_t>(p, mode, stream) : launch_T<float>(p, mode, stream);
  }
  ecg_prof_end(stream);
  return rc;
}
