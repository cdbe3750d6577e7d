// Stem convolutions (few input channels, 7-wide filter, stride 2) on MFMA, gfx950:
//   2-D: Conv2d(3,64,7,2,3) over the NCHW fp32 image  (torchvision resnet18.conv1)
//   1-D: Conv1d(cin,64,7,2,3) over the NCL fp32 signal (ResNet1D_SE.initial[0]; R = 1)
// The input is consumed in its native NCHW layout: a workgroup stages the (TH*2+R-2) x (TW*2+6)
// input patch of every channel into LDS with W-contiguous (coalesced) global reads, and the MFMA
// operand fragments are gathered straight from that patch -- the filter taps are regrouped as
// K = (c, r) groups x 8 (7 s-taps + one zero-weight pad), so a lane's 8-deep k-slice is 8 consecutive
// floats of one patch row.  Output is channels-last [N, OH, OW, 64] in the compute dtype.
// Forward: A = packed weights (rows = out channels), B = patch (cols = pixels).
// Wgrad:   A = dy^T via transposing LDS reads, B = patch values of 8 consecutive pixels (stride 2).
#include "ops.h"

__device__ __host__ inline size_t align_up_dev(size_t a) { return (a + 15) & ~(size_t)15; }

#ifndef STEM_ABL
#define STEM_ABL 0  // diagnostic builds (tools/stem_ablate.sh): 1 no output stores, 2 no statistics, 3 no MFMA loop, 4 no patch fetch
#endif

namespace {

constexpr int STEM_CO = 64;
constexpr int MAXG = 24;  // (c, r) groups supported (2-D: 21, 1-D 12-lead: 12)

struct StemParams {
  const float* x;    // [N, Cin, H, W] fp32
  const void* wpk;   // fwd: [64][KP] T, k = G*8 + s (zero for s == 7 and padded groups)
  void* y;           // fwd out / wgrad dy: [N, OH, OW, 64] T
  const float* bias;
  float* stats;      // fwd: [4*grid][2][64]
  float* slab;       // wgrad: [split][64][NG*8]
  int N, Cin, H, W, OH, OW, R, pad_h;
  int TH, TW;        // output tile (TH*TW == 128)
  int tiles_h, tiles_w;
  int NG;            // Cin * R
};

template <typename T> struct StemCfg;
template <> struct StemCfg<bf16_t> { static constexpr int WPAD = 8; };   // row stride (KP + 8) bf16: 16-B reads conflict-free
template <> struct StemCfg<float> { static constexpr int WPAD = 4; };

// tile / patch geometry per filter height (R = 7: 2-D stem, 8x16 output tile; R = 1: 1-D stem, 1x128)
template <int R> struct StemDims {
  static constexpr int TH = R == 1 ? 1 : 8, TW = R == 1 ? 128 : 16;
  static constexpr int PH = (TH - 1) * 2 + R, PW = (TW - 1) * 2 + 8, PWS = (PW + 1) & ~1;
  // patch elements a thread prefetches (registers): 2-D 3 channels x 21 x 38 = 2394 <= 10 * 256; 1-D up to 15 leads.
  // (16 for both kept the 2-D kernels at 184 VGPRs = 2 waves per SIMD; 10 lets a third workgroup onto the CU.)
  static constexpr int NPRE = R == 1 ? 16 : 10;
};

// A workgroup keeps ONE tile position (th, tw) and walks images, so everything that depends only on the position is
// computed once: which image-relative byte a thread fetches for patch slot k (padding and out-of-patch slots get an
// out-of-range offset: hardware zero fill), the pixel validity and output offsets of the epilogue.  Per image only
// the buffer base (scalar) moves.  (Walking arbitrary tiles cost ~900 VALU per wave-tile for 48 MFMAs -- two runtime
// tile-index divisions, per-element bounds tests and 64-bit addresses; SQ counters)
template <int R>
__device__ __forceinline__ void stem_patch_offsets(const StemParams& p, int oh0, int ow0,
                                                   unsigned (&off)[StemDims<R>::NPRE]) {
  constexpr int PH = StemDims<R>::PH, PW = StemDims<R>::PW, PWS = StemDims<R>::PWS, NPRE = StemDims<R>::NPRE;
  const int total = p.Cin * PH * PWS;
  const int ih0 = oh0 * 2 - p.pad_h, iw0 = ow0 * 2 - 3;
#pragma unroll
  for (int k = 0; k < NPRE; ++k) {
    const int i = threadIdx.x + 256 * k;
    const int pw = i % PWS;
    const int t = i / PWS;
    const int ph = t % PH, c = t / PH;
    const int ih = ih0 + ph, iw = iw0 + pw;
    const int ok = (int)(i < total) & (int)(pw < PW) & (int)((unsigned)ih < (unsigned)p.H) &
                   (int)((unsigned)iw < (unsigned)p.W);
    off[k] = ok ? (unsigned)((c * p.H + ih) * p.W + iw) * 4u : 0xFFFFFFFFu;
  }
}
template <int R>
__device__ __forceinline__ void stem_fetch_image_patch(const StemParams& p, const unsigned (&off)[StemDims<R>::NPRE],
                                                       float (&pre)[StemDims<R>::NPRE], int n) {
  const size_t img = (size_t)p.Cin * p.H * p.W;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * img), 0,
                                                                      (int)(img * sizeof(float)), 0x00020000);
#pragma unroll
  for (int k = 0; k < StemDims<R>::NPRE; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
    pre[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off[k], 0, 0));
#endif
  }
}
// patch buffers hold NPRE * 256 elements, so every slot is stored (the slots past the patch carry zeros).  The bf16
// kernel keeps the patch in bf16: converted once per element here instead of once per use in the MFMA loop (an element
// feeds ~12 operand fragments), half the LDS bytes per fragment read, and a workgroup's LDS drops to 35 KB.
template <typename T, int R>
__device__ __forceinline__ void stem_store_slots(T* patch, const float (&pre)[StemDims<R>::NPRE]) {
#pragma unroll
  for (int k = 0; k < StemDims<R>::NPRE; ++k) {
    if constexpr (sizeof(T) == 2) patch[threadIdx.x + 256 * k] = f2bf(pre[k]);
    else patch[threadIdx.x + 256 * k] = pre[k];
  }
}

// XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous run of (slot,
// position) ids -- neighbouring tile positions, whose input patches overlap by the filter halo, then share one L2.
// Pure speed hint (bijective for any grid size).
__device__ __forceinline__ int stem_xcd_order() {
  const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// STATS_ONLY (round 3, the recomputing stem: conv_stem_fused.hip): the BatchNorm partial sums are all this pass is for --
// no output store, and the sums stay in registers across the workgroup's images (a workgroup keeps one tile position):
// one set of four statistics rows per WORKGROUP, written at the end, instead of one per tile (128 DPP adds per
// wave-tile and 13x the rows for the finalize to fold).
// SMODE 2 (round 3): the ordinary forward (output stored) with the same per-workgroup statistics rows: the per-tile form
// spent ~30 % of a wave-tile's ~1 000 instructions on 128 DPP adds + row stores (ablation: 49 of 249 us), and its 100 k
// rows needed a fold launch in front of the finalize.  SMODE 0 keeps the per-tile rows of the stand-alone C entry point.
template <typename T, int R, int SMODE = 0>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemParams p) {
  constexpr bool STATS_ONLY = SMODE == 1, STATS_WG = SMODE != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TH = StemDims<R>::TH, TW = StemDims<R>::TW, PH = StemDims<R>::PH, PWS = StemDims<R>::PWS;
  constexpr int NPRE = StemDims<R>::NPRE, PB = NPRE * 256;  // elements per patch buffer
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KSMAX = (MAXG + 3) / 4;  // 6 (even: offsets are kept in pairs)
  const int ksteps = (p.NG + 3) / 4;
  const int KP = ksteps * 32;
  const int WS = KP + StemCfg<T>::WPAD;  // weight row stride (elements; rows stay 16-B aligned)
  T* sW = reinterpret_cast<T*>(smem);
  T* const patch0 = reinterpret_cast<T*>(smem + align_up_dev((size_t)STEM_CO * WS * sizeof(T)));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;

  // stage the packed weights once per workgroup (16-B vectors)
  {
    const T* wg = (const T*)p.wpk;
    const int vpr = KP / VEC;  // vectors per row
    for (int i = tid; i < STEM_CO * vpr; i += 256) {
      int co = i / vpr, kv = i - co * vpr;
      *reinterpret_cast<u32x4*>(sW + co * WS + kv * VEC) = *reinterpret_cast<const u32x4*>(wg + (size_t)co * KP + kv * VEC);
    }
  }
  // grid = tile positions per image x image slots
  const int tpi = p.tiles_h * p.tiles_w;
  const int vb = stem_xcd_order();
  const int pos = vb % tpi, slot = vb / tpi, nslots = gridDim.x / tpi;
  const int th_i = pos / p.tiles_w, tw_i = pos - th_i * p.tiles_w;
  const int oh0 = th_i * TH, ow0 = tw_i * TW;

  float pre[NPRE];
  unsigned poff[NPRE];
  stem_patch_offsets<R>(p, oh0, ow0, poff);

  // this lane's two 16-pixel segments: patch offset of the pixel, validity and output offset (position constants)
  constexpr int segs = TW / 16;
  int prow[2], pcol[2], boff[2], oofs[2];
  bool okb[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int t = wave * 2 + b;  // 16-pixel segment index within the workgroup (8 segments)
    prow[b] = t / segs;
    pcol[b] = (t % segs) * 16 + fr;
    boff[b] = prow[b] * 2 * PWS + pcol[b] * 2;
    okb[b] = (oh0 + prow[b] < p.OH) & (ow0 + pcol[b] < p.OW);
    oofs[b] = ((oh0 + prow[b]) * p.OW + ow0 + pcol[b]) * STEM_CO;
  }
  // patch row of this lane's (c, r) group in every k-step (padded groups carry zero weights; keep the read in bounds)
  // (two 16-bit offsets per register: the kernel sits at the 96-VGPR edge of a fourth wave per SIMD)
  unsigned goff2[KSMAX / 2];
#pragma unroll
  for (int ks = 0; ks < KSMAX; ++ks) {
    int G = ks * 4 + fq;
    if (G >= p.NG) G = p.NG - 1;
    const int c = G / R, r = G - c * R;
    const unsigned g = (unsigned)((c * PH + r) * PWS);
    if (ks & 1) goff2[ks / 2] |= g << 16;
    else goff2[ks / 2] = g;
  }
  const T* wlane = sW + fr * WS + fq * 8;  // bf16 A fragments: + a * 16 * WS + ks * 32

  // Two patch buffers: the next image's patch is fetched into registers at the top of a tile and written to the
  // OTHER buffer between this tile's MFMA loop and its epilogue, so the wait for those loads does not also wait for
  // the previous epilogue's output stores (stores and loads share vmcnt), and a tile needs one barrier, not two.
  float ws1[STATS_WG ? 4 : 1][4], ws2[STATS_WG ? 4 : 1][4];   // per-lane sums over all of the workgroup's tiles
  if (STATS_WG) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) ws1[a][j] = ws2[a][j] = 0.f;
  }
  int n = slot;
  if (n < p.N) {
    stem_fetch_image_patch<R>(p, poff, pre, n);
    stem_store_slots<T, R>(patch0, pre);
  }
  __syncthreads();  // weights and the first patch staged
  for (int cur = 0; n < p.N; n += nslots, cur ^= 1) {
  const T* patch = patch0 + cur * PB;
  const bool has_next = n + nslots < p.N;
  if (has_next && (STEM_ABL != 4 || p.N < 0)) stem_fetch_image_patch<R>(p, poff, pre, n + nslots);  // streams in underneath this tile's MFMAs

  f32x4 acc[4][2];
  if constexpr (sizeof(T) != 2) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < KSMAX; ++ks) {
      if (ks < ksteps && (STEM_ABL != 3 || p.N < 0)) {
        u32x4 fb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          // 8 consecutive bf16 of one patch row; the start is only 4-byte aligned (pixel column * 2 elements)
          const int goff = (ks & 1) ? (int)(goff2[ks / 2] >> 16) : (int)(goff2[ks / 2] & 0xffffu);
          const unsigned* src = reinterpret_cast<const unsigned*>(patch + goff + boff[b]);
          fb[b] = (u32x4){src[0], src[1], src[2], src[3]};
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          u32x4 fa = *reinterpret_cast<const u32x4*>(wlane + a * 16 * WS + ks * 32);
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            if (ks == 0) mfma_bf16_first(acc[a][b], fa, fb[b]);  // ksteps >= 1 always: C = 0 inline
            else mfma_bf16_inplace(acc[a][b], fa, fb[b]);
          }
        }
      }
    }
    mfma_drain();
  } else {
    // f32: the k index of a 16x16x4 step is the lane quad; walk the 4 groups x 8 taps of every kstep
    for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int G2 = ks * 4 + g;
        if (G2 >= p.NG) G2 = p.NG - 1;
        const int c2 = G2 / R, r2 = G2 - c2 * R;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int s = hf * 4 + fq;
          float fbv[2];
#pragma unroll
          for (int b = 0; b < 2; ++b) fbv[b] = patch[(c2 * PH + r2) * PWS + boff[b] + s];
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            float fav = sW[(a * 16 + fr) * WS + (ks * 4 + g) * 8 + s];
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fav, fbv[b], acc[a][b], 0, 0, 0);
          }
        }
      }
    }
  }

  if (has_next) stem_store_slots<T, R>(patch0 + (cur ^ 1) * PB, pre);
  if constexpr (STATS_ONLY) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (okb[b]) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = p.bias ? acc[a][b][j] + p.bias[a * 16 + fq * 4 + j] : acc[a][b][j];
            ws1[a][j] += v;
            ws2[a][j] += v * v;
          }
        }
    __syncthreads();
    continue;
  }
  // epilogue: lane = pixel fr of segment b, channels a*16 + fq*4 + j
  T* yimg = (T*)p.y + (size_t)n * p.OH * p.OW * STEM_CO;
  const int tile = n * tpi + pos;
  // bf16 output and statistics rows go through buffer stores: scalar base + 32-bit lane offset (no 64-bit address
  // per store), and a pixel outside the image gets an out-of-range offset instead of an exec-masked branch
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)yimg, 0, (int)((size_t)p.OH * p.OW * STEM_CO * sizeof(T)), 0x00020000);
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.stats ? p.stats + (size_t)tile * 4 * 2 * STEM_CO : nullptr), 0, p.stats ? 4 * 2 * STEM_CO * 4 : 0, 0x00020000);
  // bf16: the 8-byte packs of neighbouring channel tiles are exchanged between lane rows (v_permlane16_swap) so every
  // lane stores 16 B and an instruction writes 64 contiguous bytes per pixel (see conv_igemm.hip's epilogue)
#pragma unroll
  for (int ap = 0; ap < 4; ap += 2) {  // channel tiles in pairs: a pair's packs are swapped and stored before the next pair
  uint2 opk[2][2];
#pragma unroll
  for (int a = ap; a < ap + 2; ++a) {
    const int ch0 = a * 16 + fq * 4;
    float bvv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bvv[j] = p.bias[ch0 + j];
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = p.bias ? acc[a][b][j] + bvv[j] : acc[a][b][j];
      if (okb[b]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s1[j] += v[j];
          s2[j] += v[j] * v[j];
        }
        if (sizeof(T) != 2) *reinterpret_cast<float4*>(yimg + oofs[b] + ch0) = make_float4(v[0], v[1], v[2], v[3]);
      }
      if (sizeof(T) == 2) {
        opk[a - ap][b].x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        opk[a - ap][b].y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
      }
    }
    if (STATS_WG) {
      if (p.stats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ws1[a][j] += s1[j];
          ws2[a][j] += s2[j];
        }
      }
    } else if (p.stats && (STEM_ABL != 2 || p.N < 0)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] = row16_sum(s1[j]);
        s2[j] = row16_sum(s2[j]);
      }
#if defined(__HIP_DEVICE_COMPILE__)
      {
        // lanes fr == 0 hold the row sums: [wave][2][64] floats of this tile's four statistics rows
        const unsigned so = fr == 0 ? (unsigned)((wave * 2 * STEM_CO + ch0) * 4) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(s1[0]), __float_as_uint(s1[1]), __float_as_uint(s1[2]), __float_as_uint(s1[3])},
                                               srs, (int)so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(s2[0]), __float_as_uint(s2[1]), __float_as_uint(s2[2]), __float_as_uint(s2[3])},
                                               srs, (int)so, STEM_CO * 4, 0);
      }
#endif
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (sizeof(T) == 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      auto lo = __builtin_amdgcn_permlane16_swap(opk[0][b].x, opk[1][b].x, false, false);
      auto hi = __builtin_amdgcn_permlane16_swap(opk[0][b].y, opk[1][b].y, false, false);
      const int ch = (fq & 1) ? (ap + 1) * 16 + (fq - 1) * 4 : ap * 16 + fq * 4;
      const unsigned yo = okb[b] ? (unsigned)(oofs[b] + ch) * 2u : 0xFFFFFFFFu;
      if (STEM_ABL != 1 || p.N < 0)
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){lo[0], hi[0], lo[1], hi[1]}, yrs, (int)yo, 0, 0);
    }
  }
#endif
  }  // channel-tile pairs
  __syncthreads();  // the other buffer is complete; everyone is done reading this one
  }  // image loop
  if (STATS_WG && p.stats) {
    // rows [blockIdx.x * 4 + wave][2][64]: 16-lane DPP sums once per workgroup
    float* srow = p.stats + ((size_t)blockIdx.x * 4 + wave) * 2 * STEM_CO;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      f32x4 r1, r2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r1[j] = row16_sum(ws1[a][j]);
        r2[j] = row16_sum(ws2[a][j]);
      }
      if (fr == 0) {
        *reinterpret_cast<f32x4*>(srow + a * 16 + fq * 4) = r1;
        *reinterpret_cast<f32x4*>(srow + STEM_CO + a * 16 + fq * 4) = r2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad: slab[split][co][kidx] = sum over this split's tiles of dy[pix][co] * patch(pix, kidx)
// wave w owns k-tiles {w, w+4, w+8} (16 kidx each) x all 4 co-tiles.
// ---------------------------------------------------------------------------------------------------
template <typename T, int R>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int DYS = sizeof(T) == 2 ? 144 : 320;  // dy row stride in bytes (64 ch + pad)
  constexpr int TH = StemDims<R>::TH, TW = StemDims<R>::TW, PH = StemDims<R>::PH, PWS = StemDims<R>::PWS;
  unsigned char* sDY = smem;                                  // [128 pix][DYS]
  float* patch = reinterpret_cast<float*>(smem + 128 * DYS);  // [Cin][PH][PWS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int NK = p.NG * 8;
  const int ktiles = (NK + 15) / 16;
  // grid = tile positions per image x image slots: a workgroup keeps one tile position and walks its slot's images
  // (fetch offsets and validity are position constants, see stem_fwd_kernel); its slab row is blockIdx.x
  const int tpi = p.tiles_h * p.tiles_w;
  const int vb = stem_xcd_order();
  const int pos = vb % tpi, slot = vb / tpi, nslots = gridDim.x / tpi;
  const int th_i = pos / p.tiles_w, tw_i = pos - th_i * p.tiles_w;
  const int oh0 = th_i * TH, ow0 = tw_i * TW;

  f32x4 acc[4][3];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this lane's B column = kidx of each owned k-tile -> (c, r, s)
  // (columns past NK read patch element 0: their products are never stored)
  int koff[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int kidx = (wave + 4 * b) * 16 + fr;
    const bool kok = (wave + 4 * b) < ktiles && kidx < NK;
    const int kk = kok ? kidx : 0;
    const int G = kk >> 3, c = G / R;
    koff[b] = (c * PH + (G - c * R)) * PWS + (kk & 7);
  }

  const T* dy = (const T*)p.y;
  constexpr int CH = 64 * (int)sizeof(T) / 16;  // 16-B chunks per dy row
  constexpr int NDY = 128 * CH / 256;           // dy vectors per thread per tile (4 bf16 / 8 f32)
  float pre[StemDims<R>::NPRE];
  u32x4 pdy[NDY];
  unsigned poff[StemDims<R>::NPRE], dyoff[NDY];
  stem_patch_offsets<R>(p, oh0, ow0, poff);
#pragma unroll
  for (int k = 0; k < NDY; ++k) {
    const int i = tid + 256 * k;
    const int pix = i / CH, chunk = i % CH;
    const int oh = oh0 + pix / TW, ow = ow0 + pix % TW;
    dyoff[k] = ((oh < p.OH) & (ow < p.OW)) ? (unsigned)(((oh * p.OW + ow) * STEM_CO) * (int)sizeof(T) + chunk * 16) : 0xFFFFFFFFu;
  }
  auto fetch = [&](int n) {
    const size_t dimg = (size_t)p.OH * p.OW * STEM_CO;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (size_t)n * dimg), 0,
                                                                          (int)(dimg * sizeof(T)), 0x00020000);
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
      pdy[k] = __builtin_amdgcn_raw_buffer_load_b128(drs, (int)dyoff[k], 0, 0);
#endif
    }
    stem_fetch_image_patch<R>(p, poff, pre, n);
  };
  if (slot < p.N) fetch(slot);
  for (int n = slot; n < p.N; n += nslots) {
    __syncthreads();  // previous tile's LDS fully consumed
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
      const int i = tid + 256 * k;
      *reinterpret_cast<u32x4*>(sDY + (i / CH) * DYS + (i % CH) * 16) = pdy[k];
    }
    stem_store_slots<float, R>(patch, pre);
    __syncthreads();
    if (n + nslots < p.N) fetch(n + nslots);  // streams in underneath this tile's MFMAs

    if constexpr (sizeof(T) == 2) {
      const int q = fr >> 2, pq = fr & 3;
      for (int st = 0; st < 4; ++st) {  // 32 pixels per step = two 16-pixel segments
        // A: dy^T, k rows = pixels st*32 + 8*fq .. +7
        u32x4 fa[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const unsigned char* a0 = sDY + (st * 32 + 8 * fq + q) * DYS + (a * 16 + 4 * pq) * 2;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(a0)));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(a0 + 4 * DYS)));
          uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fa[a] = (u32x4){l2.x, l2.y, h2.x, h2.y};
        }
        // B: 8 consecutive pixels (st*32 + 8*fq + j) of one kidx
        const int pix0 = st * 32 + 8 * fq;
        const int trow = pix0 / TW, tcol0 = pix0 % TW;  // 8 | 16 | TW: the 8 pixels share a row
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          float bv[8];
          const float* src = patch + koff[b] + trow * 2 * PWS + tcol0 * 2;
#pragma unroll
          for (int j = 0; j < 8; ++j) bv[j] = src[2 * j];
          u32x4 fb = pack16<bf16_t>(bv);
#pragma unroll
          for (int a = 0; a < 4; ++a)
            mfma_bf16_inplace(acc[a][b], fa[a], fb);
        }
      }
    } else {
      for (int st = 0; st < 32; ++st) {  // 4 pixels per step
        const int pix = st * 4 + fq;
        const int trow = pix / TW, tcol = pix % TW;
        float fa[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const float*>(sDY + pix * DYS + (a * 16 + fr) * 4);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          float fb = patch[koff[b] + trow * 2 * PWS + tcol * 2];
#pragma unroll
          for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb, acc[a][b], 0, 0, 0);
        }
      }
    }
  }

  if (sizeof(T) == 2) mfma_drain();
  // D[row = co = a*16 + fq*4 + j][col = kidx]
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    int kidx = (wave + 4 * b) * 16 + fr;
    if ((wave + 4 * b) >= ktiles || kidx >= NK) continue;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        p.slab[((size_t)blockIdx.x * STEM_CO + a * 16 + fq * 4 + j) * NK + kidx] = acc[a][b][j];
  }
}

// grad[co][c][r][s] = sum_split slab[split][co][(c*R + r)*8 + s]   (s < 7)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ slab,
                                                                float* __restrict__ grad, int nsplit, int NG,
                                                                int accumulate) {
  // 256 threads = 16 outputs x 16 split slices
  __shared__ float sh[16][17];
  const int oi = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + oi;
  const int total = STEM_CO * NG * 7;
  const int NK = NG * 8;
  float acc = 0.f;
  if (i < total) {
    int s = i % 7;
    int t = i / 7;
    int G = t % NG, co = t / NG;
    const float* src = slab + (size_t)co * NK + G * 8 + s;
    const size_t stride = (size_t)STEM_CO * NK;
    int k = sl;
    for (; k + 48 < nsplit; k += 64) {   // four independent loads per trip (same summation order)
      const float a0 = src[(size_t)k * stride], a1 = src[(size_t)(k + 16) * stride];
      const float a2 = src[(size_t)(k + 32) * stride], a3 = src[(size_t)(k + 48) * stride];
      acc += a0;
      acc += a1;
      acc += a2;
      acc += a3;
    }
    for (; k < nsplit; k += 16) acc += src[(size_t)k * stride];
  }
  sh[sl][oi] = acc;
  __syncthreads();
  if (sl == 0 && i < total) {
    float v = 0.f;
    for (int k = 0; k < 16; ++k) v += sh[k][oi];
    grad[i] = accumulate ? grad[i] + v : v;
  }
}

// OIHW fp32 [64][Cin][R][7] -> [64][KP] T with k = (c*R + r)*8 + s
template <typename T>
__global__ void stem_pack_kernel(const float* __restrict__ w, T* __restrict__ out, int NG, int KP) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= STEM_CO * KP) return;
  int co = i / KP, k = i - co * KP;
  int G = k >> 3, s = k & 7;
  float v = (G < NG && s < 7) ? w[((size_t)co * NG + G) * 7 + s] : 0.f;
  Elem<T>::st(out + i, v);
}

struct StemShape {
  int OH, OW, TH, TW, tiles_h, tiles_w, NG, KP;
};
int stem_shape(int Cin, int H, int W, int R, StemShape& s) {
  if (R != 1 && R != 7) ECG_FAIL(ECGMM_ERR_SHAPE, "stem: R=%d (1 or 7)", R);
  s.NG = Cin * R;
  if (s.NG > MAXG) ECG_FAIL(ECGMM_ERR_SHAPE, "stem: Cin*R=%d > %d", s.NG, MAXG);
  int pad_h = R / 2;
  s.OH = (H + 2 * pad_h - R) / 2 + 1;
  s.OW = (W + 6 - 7) / 2 + 1;
  if (R == 1) { s.TH = 1; s.TW = 128; }
  else { s.TH = 8; s.TW = 16; }
  s.tiles_h = ceil_div(s.OH, s.TH);
  s.tiles_w = ceil_div(s.OW, s.TW);
  s.KP = ((s.NG + 3) / 4) * 32;
  {
    int PH = (s.TH - 1) * 2 + R, PW = (s.TW - 1) * 2 + 8, PWS = (PW + 1) & ~1;
    if ((double)Cin * H * W * 4.0 >= 2147483648.0 || (double)s.OH * s.OW * STEM_CO * 4.0 >= 2147483648.0)
      ECG_FAIL(ECGMM_ERR_SHAPE, "stem: one sample of %d x %d x %d exceeds the 2 GiB buffer-addressing range", Cin, H, W);
    if (Cin * PH * PWS > (R == 1 ? 16 : 10) * 256) ECG_FAIL(ECGMM_ERR_SHAPE, "stem: input patch of %d channels exceeds the prefetch registers", Cin);
  }
  return 0;
}
void fill_params(StemParams& p, const StemShape& s, int N, int Cin, int H, int W, int R) {
  memset(&p, 0, sizeof(p));
  p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.OH = s.OH; p.OW = s.OW; p.R = R; p.pad_h = R / 2;
  p.TH = s.TH; p.TW = s.TW; p.tiles_h = s.tiles_h; p.tiles_w = s.tiles_w; p.NG = s.NG;
}
size_t stem_patch_buffer_bytes(int R, size_t esz) { return (size_t)(R == 1 ? 16 : 10) * 256 * esz; }  // NPRE * 256 slots
size_t patch_bytes(const StemShape& s, int Cin, int R) {
  int PH = (s.TH - 1) * 2 + R, PW = (s.TW - 1) * 2 + 8, PWS = (PW + 1) & ~1;
  return (size_t)Cin * PH * PWS * sizeof(float);
}

}  // namespace

size_t ecg_stem_packed_elems(int Cin, int R) { return (size_t)STEM_CO * (((Cin * R + 3) / 4) * 32); }
int ecg_stem_stats_rows(int N, int Cin, int H, int W, int R) {
  StemShape s;
  if (stem_shape(Cin, H, W, R, s)) return -1;
  return 4 * N * s.tiles_h * s.tiles_w;
}

int ecg_stem_pack(int dtype, const float* w, void* out, int Cin, int R, hipStream_t stream) {
  int NG = Cin * R, KP = ((NG + 3) / 4) * 32;
  int n = STEM_CO * KP;
  if (dtype == ECGMM_BF16)
    hipLaunchKernelGGL(stem_pack_kernel<bf16_t>, dim3(ceil_div(n, 256)), dim3(256), 0, stream, w, (bf16_t*)out, NG, KP);
  else
    hipLaunchKernelGGL(stem_pack_kernel<float>, dim3(ceil_div(n, 256)), dim3(256), 0, stream, w, (float*)out, NG, KP);
  ECG_CHECK_LAUNCH("stem_pack");
  return 0;
}

// rows of the per-workgroup statistics form (ecg_stem_fwd_wgrows): 4 per workgroup
int ecg_stem_wg_stats_rows(int N, int Cin, int H, int W, int R) { return ecg_stem_stats_only_rows(N, Cin, H, W, R); }
static int stem_fwd_impl(int dtype, const float* x, const void* wpk, const float* bias, void* y, float* stats, int N, int Cin,
                         int H, int W, int R, bool wg_rows, hipStream_t stream);
int ecg_stem_fwd_wgrows(int dtype, const float* x, const void* wpk, const float* bias, void* y, float* stats, int N, int Cin,
                        int H, int W, int R, hipStream_t stream) {
  return stem_fwd_impl(dtype, x, wpk, bias, y, stats, N, Cin, H, W, R, true, stream);
}
int ecg_stem_fwd(int dtype, const float* x, const void* wpk, const float* bias, void* y, float* stats, int N, int Cin,
                 int H, int W, int R, hipStream_t stream) {
  return stem_fwd_impl(dtype, x, wpk, bias, y, stats, N, Cin, H, W, R, false, stream);
}
static int stem_fwd_impl(int dtype, const float* x, const void* wpk, const float* bias, void* y, float* stats, int N, int Cin,
                         int H, int W, int R, bool wg_rows, hipStream_t stream) {
  StemShape s;
  ECG_TRY(stem_shape(Cin, H, W, R, s));
  StemParams p;
  fill_params(p, s, N, Cin, H, W, R);
  p.x = x; p.wpk = wpk; p.y = y; p.bias = bias; p.stats = stats;
  const size_t esz = dtype_size(dtype);
  const int WS = s.KP + (dtype == ECGMM_BF16 ? 8 : 4);
  size_t lds = align_up((size_t)STEM_CO * WS * esz, 16) + 2 * stem_patch_buffer_bytes(R, esz);  // two patch buffers
  // a workgroup = one tile position x one image slot; it stages the weights once, then walks its slot's images
  const int tpi = s.tiles_h * s.tiles_w;
  int nslots = 2048 / tpi;
  nslots = nslots < 1 ? 1 : (nslots > N ? N : nslots);
  dim3 grid(tpi * nslots);
  ecg_prof_begin(ECG_PROF_STEM_FWD, 2.0 * (double)N * s.OH * s.OW * STEM_CO * Cin * R * 7,
                 4.0 * N * Cin * H * W + (double)dtype_size(dtype) * N * s.OH * s.OW * STEM_CO, stream);
  if (dtype == ECGMM_BF16 && wg_rows) {
    if (R == 7) hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, 7, 2>), grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, 1, 2>), grid, dim3(256), lds, stream, p);
  } else if (wg_rows) {
    ECG_FAIL(ECGMM_ERR_DTYPE, "stem: per-workgroup statistics rows are a bf16 form");
  } else if (dtype == ECGMM_BF16) {
    if (R == 7) hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, 7>), grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, 1>), grid, dim3(256), lds, stream, p);
  } else if (dtype == ECGMM_F32) {
    static bool once = false;
    if (!once) {
      (void)hipFuncSetAttribute((const void*)stem_fwd_kernel<float, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
      (void)hipFuncSetAttribute((const void*)stem_fwd_kernel<float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
      once = true;
    }
    if (R == 7) hipLaunchKernelGGL((stem_fwd_kernel<float, 7>), grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((stem_fwd_kernel<float, 1>), grid, dim3(256), lds, stream, p);
  } else {
    ECG_FAIL(ECGMM_ERR_DTYPE, "stem: bad dtype %d", dtype);
  }
  ecg_prof_end(stream);
  ECG_CHECK_LAUNCH("stem_fwd");
  return 0;
}

// Statistics-only pass of the recomputing stem (conv_stem_fused.hip): rows = 4 per workgroup.
static int stem_stats_grid(int N, int tpi) {
  int nslots = 2048 / tpi;
  nslots = nslots < 1 ? 1 : (nslots > N ? N : nslots);
  return tpi * nslots;
}
int ecg_stem_stats_only_rows(int N, int Cin, int H, int W, int R) {
  StemShape s;
  if (stem_shape(Cin, H, W, R, s)) return -1;
  return 4 * stem_stats_grid(N, s.tiles_h * s.tiles_w);
}
int ecg_stem_stats_only(int dtype, const float* x, const void* wpk, const float* bias, float* stats, int N, int Cin,
                        int H, int W, int R, hipStream_t stream) {
  if (dtype != ECGMM_BF16) ECG_FAIL(ECGMM_ERR_DTYPE, "stem statistics-only pass: bf16 only");
  StemShape s;
  ECG_TRY(stem_shape(Cin, H, W, R, s));
  StemParams p;
  fill_params(p, s, N, Cin, H, W, R);
  p.x = x; p.wpk = wpk; p.bias = bias; p.stats = stats;
  const int WS = s.KP + 8;
  size_t lds = align_up((size_t)STEM_CO * WS * 2, 16) + 2 * stem_patch_buffer_bytes(R, 2);
  dim3 grid(stem_stats_grid(N, s.tiles_h * s.tiles_w));
  ecg_prof_begin(ECG_PROF_STEM_FWD, 2.0 * (double)N * s.OH * s.OW * STEM_CO * Cin * R * 7, 4.0 * N * Cin * H * W, stream);
  if (R == 7) hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, 7, 1>), grid, dim3(256), lds, stream, p);
  else hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, 1, 1>), grid, dim3(256), lds, stream, p);
  ecg_prof_end(stream);
  ECG_CHECK_LAUNCH("stem_stats_only");
  return 0;
}

// fold of a [rows][64][NG * 8] slab of weight-gradient partial tiles into the OIHW gradient (also used by conv_stem_fused.hip)
int ecg_stem_wgrad_reduce(const float* slab, float* grad, int rows, int NG, int accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(ceil_div(STEM_CO * NG * 7, 16)), dim3(256), 0, stream, slab, grad, rows, NG,
                     accumulate);
  ECG_CHECK_LAUNCH("stem_wgrad_reduce");
  return 0;
}

// wgrad grid = slab rows: tile positions per image x image slots (about 1024 workgroups)
static int stem_wgrad_grid(int N, int tpi) {
  int nslots = 1024 / tpi;
  nslots = nslots < 1 ? 1 : (nslots > N ? N : nslots);
  return tpi * nslots;
}

size_t ecg_stem_wgrad_workspace(int N, int Cin, int H, int W, int R) {
  StemShape s;
  if (stem_shape(Cin, H, W, R, s)) return 0;
  return (size_t)stem_wgrad_grid(N, s.tiles_h * s.tiles_w) * STEM_CO * s.NG * 8 * sizeof(float);
}

int ecg_stem_wgrad(int dtype, const float* x, const void* dy, float* grad, int accumulate, void* workspace,
                   size_t workspace_bytes, int N, int Cin, int H, int W, int R, hipStream_t stream) {
  StemShape s;
  ECG_TRY(stem_shape(Cin, H, W, R, s));
  const int grid = stem_wgrad_grid(N, s.tiles_h * s.tiles_w);
  size_t need = (size_t)grid * STEM_CO * s.NG * 8 * sizeof(float);
  if (!workspace || workspace_bytes < need)
    ECG_FAIL(ECGMM_ERR_WORKSPACE, "stem wgrad: workspace %zu < %zu", workspace_bytes, need);
  StemParams p;
  fill_params(p, s, N, Cin, H, W, R);
  p.x = x; p.y = const_cast<void*>(dy); p.slab = (float*)workspace;
  size_t lds = 128 * (size_t)(dtype == ECGMM_BF16 ? 144 : 320) + stem_patch_buffer_bytes(R, sizeof(float));
  if (dtype != ECGMM_BF16 && dtype != ECGMM_F32) ECG_FAIL(ECGMM_ERR_DTYPE, "stem wgrad: bad dtype %d", dtype);
  ecg_prof_begin(ECG_PROF_STEM_WGRAD, 2.0 * (double)N * s.OH * s.OW * STEM_CO * Cin * R * 7,
                 4.0 * N * Cin * H * W + (double)dtype_size(dtype) * N * s.OH * s.OW * STEM_CO, stream);
  if (dtype == ECGMM_BF16) {
    if (R == 7) hipLaunchKernelGGL((stem_wgrad_kernel<bf16_t, 7>), dim3(grid), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((stem_wgrad_kernel<bf16_t, 1>), dim3(grid), dim3(256), lds, stream, p);
  } else {
    if (R == 7) hipLaunchKernelGGL((stem_wgrad_kernel<float, 7>), dim3(grid), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((stem_wgrad_kernel<float, 1>), dim3(grid), dim3(256), lds, stream, p);
  }
  ecg_prof_end(stream);
  ECG_CHECK_LAUNCH("stem_wgrad");
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(ceil_div(STEM_CO * s.NG * 7, 16)), dim3(256), 0, stream,
                     (const float*)workspace, grad, grid, s.NG, accumulate);
  ECG_CHECK_LAUNCH("stem_wgrad_reduce");
  return 0;
}
