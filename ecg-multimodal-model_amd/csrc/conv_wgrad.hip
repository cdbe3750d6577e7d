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
  int N, H, W, Cin, OH, OW, Cout, S, stride, pad_h, pad_w;
  int M;
  int steps_per_split;
  unsigned mul_hw, sh_hw, mul_w, sh_w;  // exact division by OH*OW and OW for x < 2^31 (see magic_div)
};

// 16-byte LDS-DMA buffer load (device pass only: the host pass checks builtins against its own target)
__device__ __forceinline__ void wg_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_wave_base, unsigned voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, 0, 0,
                                           0);
#endif
}

#ifdef ECG_STAMP
// diagnostic build only (make stamp; tools/stamp_igemm.py --wgrad): see conv_igemm.hip
__device__ unsigned long long g_wstamp[8];
#define ECG_WSTAMP_AT(t)                                                                 \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
extern "C" int ecgmm_wstamp_read(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wstamp), sizeof(g_wstamp)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wstamp), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

// Every tile is KP pixel rows x 64 channels = 4 KiB, filled by ONE 1-KiB LDS-DMA piece per wave; the
// XOR swizzle of the 16-B chunk index (applied on the source side) makes the fragment reads spread
// over the banks: bf16 transposing reads touch rows {q, 8+q} per 32-lane half, f32 reads rows {k, k+1}.
template <typename T> struct WgCfg;
template <> struct WgCfg<bf16_t> {
  static constexpr int KP = 32;    // pixels per K step
  static constexpr int RB = 128;   // row bytes (64 ch)
  static constexpr int CPR = 8;    // 16-B chunks per row
  static constexpr int RPP = 8;    // rows per DMA piece
  static __device__ __forceinline__ int swz(int row) { return ((row & 3) ^ ((row >> 3) & 1)) << 1; }
};
template <> struct WgCfg<float> {
  static constexpr int KP = 16;
  static constexpr int RB = 256;
  static constexpr int CPR = 16;
  static constexpr int RPP = 4;
  static __device__ __forceinline__ int swz(int row) { return (row & 1) << 2; }
};

// GROUPS = 2: a 512-thread workgroup is two 4-wave groups, each running the K loop below over its own half of the
// workgroup's pixel slice with its own LDS stages; at the end group 1 hands its accumulators to group 0 through LDS
// and ONE partial tile goes to the slab.  Same waves per CU as two 256-thread workgroups, half the split count: half the
// fp32 slab bytes written here and re-read by the reduce kernel (75 MB -> 37 MB per 3x3 launch at batch 256).
template <typename T, int NT, int GROUPS>
__global__ __launch_bounds__(256 * GROUPS) void wgrad_kernel(WgradParams p) {
  using C = WgCfg<T>;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int TILE_BYTES = C::KP * C::RB;  // 4096
  constexpr int STAGE_BYTES = (1 + NT) * TILE_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][1 + NT tiles]

  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int grp = GROUPS == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 8);
  unsigned char* const smem_g = smem + grp * 2 * STAGE_BYTES;
  // Wave layout: every wave owns ALL 64 output channels (4 dy fragments, loaded once per K step and reused by every
  // tap) x its own 16 input channels (one x fragment per tap): 4 + NT fragment reads per 4 NT MFMAs.  The 2 x 2 layout
  // (32 x 32 per wave) read 2 + 2 NT fragments for the same MFMAs -- 0.56 vs 0.36 KB of LDS per MFMA at 9 taps, and the
  // LDS read bytes of two resident workgroups were 110 % of their MFMA time.
  constexpr int TA = 4, TB = 1;
  const int ci_tiles = (p.Cin + 63) / 64;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest).  All (co, ci)
  // tiles of one split read the same pixels of x and dy, so a split's tiles are steered onto ONE XCD (its operand
  // range then passes through one L2 instead of eight); splits are spread over the XCDs.  Bijective when the split
  // count is a multiple of 8 (layers 2-4: 128 / 32 / 8 splits), identity otherwise.
  int tile_id = blockIdx.x, split = blockIdx.y;
  if ((gridDim.y & 7) == 0 && gridDim.x > 1) {
    const int NT_ = gridDim.x, L = blockIdx.y * NT_ + blockIdx.x;
    const int k = L >> 3;
    split = (L & 7) + 8 * (k / NT_);
    tile_id = k % NT_;
  }
  const int co0 = (tile_id / ci_tiles) * 64, ci0 = (tile_id % ci_tiles) * 64;
  const T* __restrict__ x = (const T*)p.x;
  const T* __restrict__ dy = (const T*)p.dy;

  const int total_steps = (p.M + C::KP - 1) / C::KP;
  const int wg_begin = split * p.steps_per_split;
  const int wg_end = min(total_steps, wg_begin + p.steps_per_split);
  const int wg_len = max(wg_end - wg_begin, 0);
  const int iters = GROUPS == 1 ? wg_len : (wg_len + 1) / 2;       // group 0's step count (>= group 1's)
  const int s_begin = wg_begin + grp * iters;
  const int s_end = GROUPS == 1 ? wg_end : min(wg_end, s_begin + iters);
  const int my_len = max(s_end - s_begin, 0);

  f32x4 acc[NT][TA][TB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) acc[t][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- DMA bookkeeping: this lane fills (row = wave*RPP + lane/CPR, physical chunk lane%CPR) of every tile
  constexpr unsigned OOB = 0xFFFFFFFFu;
  const int drow = wave * C::RPP + lane / C::CPR;
  const int lchunk = (lane % C::CPR) ^ C::swz(drow);
  const int HW = p.OH * p.OW;
  const long pix0 = (long)s_begin * C::KP;                       // first pixel of this split
  const int n0 = (int)(((unsigned long long)(unsigned)pix0 * p.mul_hw) >> p.sh_hw);
  const size_t img = (size_t)p.H * p.W * p.Cin;
  const size_t x_left = ((size_t)p.N - n0) * img * sizeof(T);
  const size_t dy_left = ((size_t)p.M - pix0) * p.Cout * sizeof(T);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(x + (size_t)n0 * img), 0, x_left > 0xFFFFFFF0ull ? (int)0xFFFFFFF0u : (int)x_left, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(dy + (size_t)pix0 * p.Cout), 0, dy_left > 0xFFFFFFF0ull ? (int)0xFFFFFFF0u : (int)dy_left, 0x00020000);
  const int cdy = co0 + lchunk * VEC, cx = ci0 + lchunk * VEC;
  const unsigned dy_lane = cdy < p.Cout ? (unsigned)cdy * (unsigned)sizeof(T) : OOB;
  const unsigned x_lane = cx < p.Cin ? (unsigned)cx * (unsigned)sizeof(T) : OOB;
  const unsigned dy_row_bytes = (unsigned)p.Cout * (unsigned)sizeof(T), x_pix_bytes = (unsigned)p.Cin * (unsigned)sizeof(T);

  // filter width is 3 or 1 (taps = R*S in {1, 3, 9}); per-tap source deltas in bytes, wave-uniform
  const bool s3 = p.S == 3;
  unsigned tap_delta[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tap_delta[t] = (unsigned)((s3 ? t / 3 : t) * p.W + (s3 ? t % 3 : 0)) * x_pix_bytes;
  // Pixel state of this lane's DMA row.  Steps are issued strictly in order, each KP pixels after the last, so
  // (oh, ow), the byte offset of tap (0,0)'s source pixel and the dy row offset are ADVANCED by wave-uniform
  // constants with at most one carry per dimension, instead of being decoded from the pixel index every step (that
  // decode was two 64-bit magic divisions and five more quarter-rate multiplies per K step: ~40 % of the step's
  // VALU cycles, conv_wgrad DMA issue = 40 % of a K step in the s_memtime stamps).
  const unsigned sW_b = (unsigned)p.stride * x_pix_bytes, sH_b = (unsigned)(p.stride * p.W) * x_pix_bytes;
  const unsigned img_b = (unsigned)(p.H * p.W) * x_pix_bytes;
  const int d_n = C::KP / HW, d_r = C::KP - d_n * HW, d_oh = d_r / p.OW, d_ow = d_r - d_oh * p.OW;  // KP pixels ahead
  const unsigned adv = (unsigned)d_n * img_b + (unsigned)d_oh * sH_b + (unsigned)d_ow * sW_b;
  const unsigned carry_w = sH_b - (unsigned)p.OW * sW_b;   // ow wrapped: one output row down
  const unsigned carry_h = img_b - (unsigned)p.OH * sH_b;  // oh wrapped: next image
  int st_oh, st_ow;
  unsigned st_off, st_dy;
  {
    const int pix = s_begin * C::KP + drow;
    const unsigned pp = pix < p.M ? (unsigned)pix : 0u;
    const int n = (int)(((unsigned long long)pp * p.mul_hw) >> p.sh_hw);
    const unsigned rem = pp - (unsigned)n * (unsigned)HW;
    st_oh = (int)(((unsigned long long)rem * p.mul_w) >> p.sh_w);
    st_ow = (int)rem - st_oh * p.OW;
    st_off = (unsigned)(((n - n0) * p.H + st_oh * p.stride - p.pad_h) * p.W + st_ow * p.stride - p.pad_w) * x_pix_bytes + x_lane;
    st_dy = (unsigned)(pix - (int)pix0) * dy_row_bytes + dy_lane;
  }
  const int lane_x = (int)(x_lane != OOB), lane_dy = (int)(dy_lane != OOB);
  auto dma = [&](int stage, int step) {
    unsigned char* base = smem_g + stage * STAGE_BYTES + wv * 1024;
    const int pok = (int)(step * C::KP + drow < p.M);
    wg_dma16(rs_dy, base, (pok & lane_dy) ? st_dy : OOB);
    const int hb = __mul24(st_oh, p.stride) - p.pad_h, wb = __mul24(st_ow, p.stride) - p.pad_w;
    // tap (r, s) is a wave-uniform delta away from tap (0,0).
    // (bitwise, not short-circuit: with && the compiler turned every tap's DMA into two exec-masked branches)
    const int lane_ok = pok & lane_x;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int r = s3 ? t / 3 : t, s = s3 ? t % 3 : 0;  // (t is a constant after unrolling; s3 is wave-uniform)
      const int ok = lane_ok & (int)((unsigned)(hb + r) < (unsigned)p.H) & (int)((unsigned)(wb + s) < (unsigned)p.W);
      wg_dma16(rs_x, base + (1 + t) * TILE_BYTES, ok ? st_off + tap_delta[t] : OOB);
    }
    // advance to the next step's pixel
    st_dy += (unsigned)C::KP * dy_row_bytes;
    st_ow += d_ow;
    const bool cw = st_ow >= p.OW;
    st_ow -= cw ? p.OW : 0;
    st_oh += d_oh + (cw ? 1 : 0);
    const bool ch = st_oh >= p.OH;
    st_oh -= ch ? p.OH : 0;
    st_off += adv + (cw ? carry_w : 0u) + (ch ? carry_h : 0u);
  };

  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int stage) {
    const unsigned char* st = smem_g + stage * STAGE_BYTES;
    if constexpr (sizeof(T) == 2) {
      // transposing reads: lane 4q+p of a 16-lane group addresses row (kb+q), columns 4p..4p+3;
      // lane i receives column i of those 4 rows.  Group fq takes k rows 8fq..8fq+7.
      const int q = fr >> 2, pq = fr & 3;
      const int row = 8 * fq + q;
      const int sw = C::swz(row);  // == swz(row + 4)
      auto frag = [&](const unsigned char* tile, int col0) -> u32x4 {
        const int byte = (col0 + 4 * pq) * 2;
        const unsigned char* a0 = tile + row * C::RB + ((((byte >> 4) ^ sw)) << 4) + (byte & 15);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(a0)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(a0 + 4 * C::RB)));
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return (u32x4){l2.x, l2.y, h2.x, h2.y};
      };
      u32x4 fa[TA];
#pragma unroll
      for (int a = 0; a < TA; ++a) fa[a] = frag(st, a * 16);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        u32x4 fb[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) fb[b] = frag(st + (1 + t) * TILE_BYTES, wave * 16 + b * 16);
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
          for (int b = 0; b < TB; ++b)
            acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[a]),
                                                                   __builtin_bit_cast(bf16x8_t, fb[b]),
                                                                   acc[t][a][b], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < C::KP / 4; ++ks) {
        const int krow = ks * 4 + fq;
        const int sw = C::swz(krow);
        auto rd = [&](const unsigned char* tile, int col) -> float {
          return *reinterpret_cast<const float*>(tile + krow * C::RB + ((((col >> 2) ^ sw)) << 4) + (col & 3) * 4);
        };
        float fa[TA];
#pragma unroll
        for (int a = 0; a < TA; ++a) fa[a] = rd(st, a * 16 + fr);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float fb[TB];
#pragma unroll
          for (int b = 0; b < TB; ++b) fb[b] = rd(st + (1 + t) * TILE_BYTES, wave * 16 + b * 16 + fr);
#pragma unroll
          for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b)
              acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[t][a][b], 0, 0, 0);
        }
      }
    }
  };

#ifdef ECG_STAMP
  unsigned long long sA = 0, sB = 0, sC = 0, t0, t1, t2, t3, tk0, tk1;
  ECG_WSTAMP_AT(tk0);
#endif
  if (iters > 0) {   // (every wave of the workgroup takes the same `iters` trips: the barriers below are workgroup-wide)
    if (my_len > 0) dma(0, s_begin);
    __syncthreads();  // drains vmcnt: stage 0 has landed for every wave
    for (int k = 0; k < iters; ++k) {
#ifdef ECG_STAMP
      ECG_WSTAMP_AT(t0);
#endif
      if (k + 1 < my_len) dma((k + 1) & 1, s_begin + k + 1);  // buffer last read in iteration k-1, fenced by its barrier
#ifdef ECG_STAMP
      ECG_WSTAMP_AT(t1);
#endif
      if (GROUPS == 1 || k < my_len) compute(k & 1);
#ifdef ECG_STAMP
      ECG_WSTAMP_AT(t2);
#endif
      __syncthreads();
#ifdef ECG_STAMP
      ECG_WSTAMP_AT(t3);
      sA += t1 - t0; sB += t2 - t1; sC += t3 - t2;
#endif
    }
  }
#ifdef ECG_STAMP
  ECG_WSTAMP_AT(tk1);
  if (threadIdx.x == 0) {
    atomicAdd(&g_wstamp[0], sA);
    atomicAdd(&g_wstamp[1], sB);
    atomicAdd(&g_wstamp[2], sC);
    atomicAdd(&g_wstamp[4], tk1 - tk0);
    atomicAdd(&g_wstamp[6], 1ull);
    atomicAdd(&g_wstamp[7], (unsigned long long)(s_end - s_begin));
  }
#endif

  if constexpr (GROUPS == 2) {
    // group 1 -> LDS -> group 0 (register r of thread t at float [r * 256 + t]: every access a contiguous 1-KiB row).
    // The K loop's last barrier has retired every fragment read and every DMA, so the stages can be overwritten.
    float* red = reinterpret_cast<float*>(smem) + (tid & 255);
    if (grp == 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
          for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(((t * TA + a) * TB + b) * 4 + j) * 256] = acc[t][a][b][j];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[t][a][b][j] += red[(((t * TA + a) * TB + b) * 4 + j) * 256];
  }

  // ---- slab store: D[row = co][col = ci]; lane: ci = fr, co = fq*4 + j
  const int RS = NT;
  if (co0 + 64 <= p.Cout && ci0 + 64 <= p.Cin) {  // whole tile in range: no per-element branches
    float* base = p.slab + (((size_t)split * RS) * p.Cout + co0 + fq * 4) * p.Cin + ci0 + wave * 16 + fr;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            base[((size_t)t * p.Cout + a * 16 + j) * p.Cin + b * 16] = acc[t][a][b][j];
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        int ci = ci0 + wave * 16 + b * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int co = co0 + a * 16 + fq * 4 + j;
          if (co < p.Cout && ci < p.Cin)
            p.slab[(((size_t)split * RS + t) * p.Cout + co) * p.Cin + ci] = acc[t][a][b][j];
        }
      }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad_ring_kernel: the same 64(co) x 64(ci) x all-taps tile and split-K, for stride-1 "same" convolutions (3x3 pad 1,
// 1x3 pad 1) in bf16 -- with the x operand kept in a RING of pixel rows instead of one gathered tile per tap.
//
// In wgrad_kernel every K step (32 pixels) moves 1 + NT tiles = 40 KB through the CU's vector-memory path for 36 MFMAs per
// wave (10 LDS-DMA pieces per wave and step: ~40 % of a step in the s_memtime stamps), and the NT x tiles are NT shifted
// copies of (almost) the same pixels.  With stride 1 the source pixel of tap (r, s) for output pixel p is the flat pixel
// p + (r - ph) * W + (s - pw) of the SAME channels-last tensor, so one window of consecutive pixel rows serves every
// tap: a 256-row ring (32 KB) indexed by (flat pixel & 255) holds [p0 - HL, p0 + 32 + HL) and takes in 32 new rows per
// step -- 2 pieces per wave and step instead of 10.  A tap that leaves the image reads a zero row instead (per-lane
// address select; the ring row it would hit holds a neighbouring pixel).  Ring rows advance by 32 per step, which
// leaves the bank swizzle (bits 0-3 of the row) unchanged: the 2 x NT fragment addresses of a lane are computed once
// and only rotated (+4096 B mod 32 KB) per step.
// ------------------------------------------------------------------------------------------------------------------
struct WgradRingParams {
  const void* x;
  const void* dy;
  float* slab;
  int H, W, Cin, Cout, pad_h;
  int M;
  int steps_per_split;
  int HL, HLa;           // halo pixels (pad_h * W + 1) and the same rounded up to a multiple of 8
  unsigned mul_hw, sh_hw, mul_w, sh_w;
  int pingpong;          // two-group workgroups, 9 taps: the groups run half a K step apart (see the K loop)
};

constexpr int RING_ROWS = 256;
constexpr int RING_BYTES = RING_ROWS * 128;
constexpr int RING_D = 3;                      // fills run RING_D steps ahead of the step being computed
constexpr int RING_NDY = RING_D + 1;           // dy stages (4 KB each)
constexpr int RING_DYB = RING_NDY * 4096;
constexpr int RING_LDS = RING_BYTES + RING_DYB + 128;   // per group: ring, dy stages, one all-zero 128-byte row
// Workgroup layout (round 3): [ring of group 0 | ring of group 1 | dy stages 0 | dy stages 1 | zero rows] -- every ring starts on a
// multiple of its own 32 KB, so a fragment address is ((A + rot) & 0x7FFF) | ring base: add + and-or on ABSOLUTE LDS addresses
// (the reads took `group base + offset` before: one more add per read, 26 reads per step).

// LDS-DMA as inline asm (M0 = LDS byte address of the wave's 1-KiB piece, saved and restored): hipcc does not see these
// loads, so it neither drains them with an s_waitcnt vmcnt(0) in front of the next ds_read_b64_tr_b16 (which it does for
// the builtin: an intrinsic without memory operands "may alias" the LDS-DMA, conv_wgrad.hip notes) nor counts them --
// the kernel retires them with its own counted s_waitcnt vmcnt(N) + s_barrier, RING_D steps after issue.
__device__ __forceinline__ void ring_dma16(const u32x4& rsrc, unsigned lds_addr, unsigned voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(lds_addr), "s"(rsrc)
      : "memory");
#endif
}
template <int N> __device__ __forceinline__ void ring_wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ u32x4 ring_rsrc(const void* ptr, size_t bytes) {
  const unsigned long long a = (unsigned long long)ptr;
  return (u32x4){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu)),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bytes), 0x00020000u};
}

// GROUPS = 2: two 4-wave groups per 512-thread workgroup, each with its own ring / dy stages over its half of the
// workgroup's steps, one slab tile per workgroup (LDS hand-over at the end) -- as wgrad_kernel<T, NT, 2>.
template <int NT, int GROUPS>
__global__ __launch_bounds__(256 * GROUPS) void wgrad_ring_kernel(WgradRingParams p) {
  using C = WgCfg<bf16_t>;
  constexpr int R = NT / 3;  // filter rows (S == 3)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_wg[];

  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int grp = GROUPS == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 8);
  const unsigned lds_wg = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_wg;   // LDS address of smem_wg[0]
  const unsigned ring_at = lds_wg + (unsigned)(grp * RING_BYTES);
  const unsigned dy_at = lds_wg + (unsigned)(GROUPS * RING_BYTES + grp * RING_DYB);
  const unsigned zero_at = lds_wg + (unsigned)(GROUPS * (RING_BYTES + RING_DYB) + grp * 128);
  // (absolute ring addresses are composed with OR: the dynamic LDS segment of this kernel starts at 0 -- it has no static LDS --
  //  and if a toolchain ever placed it elsewhere the ring would still have to sit on a 32 KB boundary)
  if ((ring_at & (unsigned)(RING_BYTES - 1)) != 0u) __builtin_trap();
  auto lds_ptr = [](unsigned a) { return (__attribute__((address_space(3))) unsigned char*)(size_t)a; };
  constexpr int TA = 4;   // wave layout as in wgrad_kernel: all 64 output channels x the wave's own 16 input channels
  // PING-PONG between the two groups (round 3; the two waves of a SIMD are wave w of group 0 and wave w of group 1).  In lock
  // step both groups run a K step's head (fill issue, the dy fragments, 18 tap addresses: no MFMA) together and then
  // contend for the matrix pipe with 36 MFMAs each.  A second workgroup barrier in the middle of the step (behind the MFMAs
  // of tap PP_SPLIT - 1) and ONE barrier of offset between the groups (an extra s_barrier in front of group 1's loop, one
  // behind group 0's) make every interval pair one group's head + first MFMAs with the other group's MFMA-only second half.
  // The groups share no LDS (own ring, own dy stages): every hazard is inside a group, whose waves stay in the same phase.
  constexpr bool PP_OK = GROUPS == 2 && NT == 9;
  constexpr int PP_SPLIT = 4;
  const bool pp = PP_OK && p.pingpong != 0;
  const int ci_tiles = p.Cin / 64;
  int tile_id = blockIdx.x, split = blockIdx.y;
  if ((gridDim.y & 7) == 0 && gridDim.x > 1) {   // a split's tiles on one XCD (see wgrad_kernel)
    const int NT_ = gridDim.x, L = blockIdx.y * NT_ + blockIdx.x;
    const int k = L >> 3;
    split = (L & 7) + 8 * (k / NT_);
    tile_id = k % NT_;
  }
  const int co0 = (tile_id / ci_tiles) * 64, ci0 = (tile_id % ci_tiles) * 64;
  const bf16_t* __restrict__ x = (const bf16_t*)p.x;
  const bf16_t* __restrict__ dy = (const bf16_t*)p.dy;
  const int total_steps = (p.M + 31) / 32;
  const int wg_begin = split * p.steps_per_split;
  const int wg_len = max(min(total_steps, wg_begin + p.steps_per_split) - wg_begin, 0);
  const int iters = GROUPS == 1 ? wg_len : (wg_len + 1) / 2;   // group 0's step count (>= group 1's)
  const int s_begin = wg_begin + grp * iters;
  const int s_end = min(wg_begin + wg_len, s_begin + iters);

  f32x4 acc[NT][TA];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < TA; ++a) acc[t][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if ((tid & 255) < 8) *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(lds_ptr(zero_at + (tid & 255) * 16)) = (u32x4){0u, 0u, 0u, 0u};

  // ---- DMA bookkeeping (whole tensors are addressed from their start: the host keeps them under 2 GiB)
  constexpr unsigned OOB = 0xFFFFFFFFu;
  const u32x4 rs_x = ring_rsrc(x, (size_t)p.M * p.Cin * 2), rs_dy = ring_rsrc(dy, (size_t)p.M * p.Cout * 2);
  const int lrow8 = lane >> 3;
  const unsigned x_pix_bytes = (unsigned)p.Cin * 2u, dy_row_bytes = (unsigned)p.Cout * 2u;
  const int P0 = s_begin * 32;
  // dy: lane fills (row = wave*8 + lane/8, physical chunk lane%8) of the step's 32 x 64 tile
  const int drow = wave * 8 + lrow8;
  const unsigned dy_lane = (unsigned)(co0 + (((lane & 7) ^ C::swz(drow)) * 8)) * 2u;
  // x ring: a piece is 8 consecutive flat pixels; piece rows are 8-aligned in the ring, so the swizzle is per lane
  auto x_piece = [&](int q0) {   // q0: first flat pixel of the piece (multiple of 8, may be negative or >= M)
    const int q = q0 + lrow8;
    const int rrow = q & (RING_ROWS - 1);
    const unsigned src = (unsigned)q * x_pix_bytes + (unsigned)(ci0 + (((lane & 7) ^ C::swz(rrow)) * 8)) * 2u;
    ring_dma16(rs_x, ring_at + (unsigned)((q0 & (RING_ROWS - 1)) * 128), (q >= 0 && q < p.M) ? src : OOB);
  };
  auto dy_piece = [&](int stage, int step) {
    const int pix = step * 32 + drow;
    ring_dma16(rs_dy, dy_at + (unsigned)(stage * 4096 + wv * 1024), pix < p.M ? (unsigned)pix * dy_row_bytes + dy_lane : OOB);
  };
  // fill group of `step` (2 DMAs per wave): its dy tile, and the 32 ring rows it needs beyond the previous step's window
  auto dma_step = [&](int stage, int step) {
    dy_piece(stage, step);
    x_piece(step * 32 + p.HLa + wv * 8);
  };

  // ---- fragment addresses.  Lane (fq, q = fr >> 2, pq = fr & 3) supplies the row address of k rows 8 fq + q (lo) and
  // + 4 (hi) of a step; per tap the ring-relative byte address of the lo / hi source pixel at the split's first step.
  const int fr = lane & 15, fq = lane >> 4;
  const int q4 = fr >> 2, pq = fr & 3;
  const int kl = 8 * fq + q4;
  unsigned A_lo[NT], A_hi[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r = t / 3, sx = t - r * 3;
    const int d = (r - p.pad_h) * p.W + (sx - 1);
    const int byte0 = wave * 32 + 8 * pq;
    const int rl = (P0 + kl + d) & (RING_ROWS - 1), rh = (P0 + kl + 4 + d) & (RING_ROWS - 1);
    A_lo[t] = (unsigned)(rl * 128 + ((((byte0 >> 4) ^ C::swz(rl))) << 4) + (byte0 & 15));
    A_hi[t] = (unsigned)(rh * 128 + ((((byte0 >> 4) ^ C::swz(rh))) << 4) + (byte0 & 15));
  }
  // (oh, ow) of the lane's lo pixel, advanced by 32 pixels per step with one carry per dimension
  const int HW = p.H * p.W;
  int st_oh, st_ow;
  {
    const unsigned pp = (unsigned)(P0 + kl) < (unsigned)p.M ? (unsigned)(P0 + kl) : 0u;
    const int n = (int)(((unsigned long long)pp * p.mul_hw) >> p.sh_hw);
    const unsigned rem = pp - (unsigned)n * (unsigned)HW;
    st_oh = (int)(((unsigned long long)rem * p.mul_w) >> p.sh_w);
    st_ow = (int)rem - st_oh * p.W;
  }
  const int d_r = 32 % HW, d_oh = d_r / p.W, d_ow = d_r - d_oh * p.W;

  // tap addresses of a step: per tap the ring byte the lane's lo / hi source pixel sits at (or the zero row when the tap
  // leaves the image) -- pure VALU, no LDS / fill dependency.  SOFTWARE-PIPELINED one step ahead (round 3): the addresses of
  // step k + 1 are computed inside step k's tap loop, two taps' worth behind each tap's MFMAs, where the VALU work runs
  // under the matrix pipe; in front of the step's first MFMA it was ~100 VALU per wave that both groups executed at the
  // same time with the pipe idle (ablation, layer 2: the kernel without its MFMAs took 57 of 85 us -- the MFMA time sat
  // entirely on top of the rest).
  // Vector-instruction ISSUE is what bounds this loop (a 16x16x32 MFMA holds the SIMD's vector issue for 8 of its 16 cycles: two
  // VALU per MFMA are free, every further one costs 4 cycles, for both waves of the SIMD; the loop had 125 VALU per 36 MFMAs and
  // ran at the 1.8 k cycles per step that predicts).  So the validity of a step's taps is 8 lane masks (row above / below and
  // column left / right in range, for the lo and the hi pixel) -- 8 compares per step, combined per tap on the SCALAR unit --
  // instead of four add + compare range checks per tap, and a tap's address is add, and, select.
  unsigned al[NT], ah[NT];
  // (the masks are ballots -- 64-bit scalars -- and the select takes its mask from the scalar pair: as `bool`s hipcc turned them
  //  into 0/1 bytes in vector registers and the loop grew to 139 VALU)
  typedef unsigned long long lanemask_t;
  struct TapOk { lanemask_t lo_r[3], lo_c[3], hi_r[3], hi_c[3]; };
  auto tap_ok = [&](int oh_l, int ow_l) -> TapOk {
    int oh_h = oh_l, ow_h = ow_l + 4;
    if (ow_h >= p.W) { ow_h -= p.W; ++oh_h; }
    if (oh_h >= p.H) oh_h -= p.H;
    TapOk k;
    const lanemask_t all = ~0ull;
    // filter row r reads image row oh + r - pad_h (pad_h = 1 for three filter rows, 0 for one), column sx reads ow + sx - 1
    k.lo_r[0] = R == 1 ? all : __builtin_amdgcn_ballot_w64(oh_l >= 1); k.lo_r[1] = all; k.lo_r[2] = __builtin_amdgcn_ballot_w64(oh_l <= p.H - 2);
    k.hi_r[0] = R == 1 ? all : __builtin_amdgcn_ballot_w64(oh_h >= 1); k.hi_r[1] = all; k.hi_r[2] = __builtin_amdgcn_ballot_w64(oh_h <= p.H - 2);
    k.lo_c[0] = __builtin_amdgcn_ballot_w64(ow_l >= 1); k.lo_c[1] = all; k.lo_c[2] = __builtin_amdgcn_ballot_w64(ow_l <= p.W - 2);
    k.hi_c[0] = __builtin_amdgcn_ballot_w64(ow_h >= 1); k.hi_c[1] = all; k.hi_c[2] = __builtin_amdgcn_ballot_w64(ow_h <= p.W - 2);
    return k;
  };
  auto sel = [](lanemask_t m, unsigned if_set, unsigned if_clear) -> unsigned {
    unsigned r_;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r_) : "v"(if_clear), "v"(if_set), "s"(m));
#else
    r_ = if_set;
#endif
    return r_;
  };
  const unsigned zrow = zero_at + (unsigned)(8 * (pq & 1));
  // (x & 0x7FFF) | ring base as ONE v_and_or_b32: mask from a scalar register, base from a vector register -- with the mask as a
  // literal (gfx9 VOP3 takes none) hipcc emits v_and + v_or
  unsigned ring_at_v;
  const unsigned ring_mask = (unsigned)(RING_BYTES - 1);
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mov_b32 %0, %1" : "=v"(ring_at_v) : "s"(ring_at));
#else
  ring_at_v = ring_at;
#endif
  auto wrap = [&](unsigned x_) -> unsigned {
    unsigned r_;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r_) : "v"(x_), "s"(ring_mask), "v"(ring_at_v));
#else
    r_ = (x_ & ring_mask) | ring_at_v;
#endif
    return r_;
  };
  auto tap_addr = [&](int t, const TapOk& k, unsigned rot_, unsigned& a_lo, unsigned& a_hi) {
    const int r = t / 3, sx = t - r * 3;
    a_lo = sel(k.lo_r[r] & k.lo_c[sx], wrap(A_lo[t] + rot_), zrow);
    a_hi = sel(k.hi_r[r] & k.hi_c[sx], wrap(A_hi[t] + rot_), zrow);
  };
  {
    const TapOk k0 = tap_ok(st_oh, st_ow);
#pragma unroll
    for (int t = 0; t < NT; ++t) tap_addr(t, k0, 0u, al[t], ah[t]);   // the first step's
  }

  unsigned dyfrag[TA];
  {
    const int row = 8 * fq + q4;
    const int sw = C::swz(row);
#pragma unroll
    for (int a = 0; a < TA; ++a) {
      const int byte = (a * 16 + 4 * pq) * 2;
      dyfrag[a] = (unsigned)(row * 128 + ((((byte >> 4) ^ sw)) << 4) + (byte & 15));
    }
  }
  auto compute = [&](int stage, unsigned rot) {
    // dy fragments (A operand): the step's own tile, rows 8 fq + q (+ 4), as in wgrad_kernel (per-lane offsets fixed for the kernel)
    const unsigned dyt = dy_at + (unsigned)(stage * 4096);
    u32x4 fa[TA];
#pragma unroll
    for (int a = 0; a < TA; ++a) {
      const unsigned a0 = dyt + dyfrag[a];
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_ptr(a0));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_ptr(a0 + 512u));
      uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      fa[a] = (u32x4){l2.x, l2.y, h2.x, h2.y};
    }
    // next step: 32 pixels on
    int n_ow = st_ow + d_ow;
    const bool cw = n_ow >= p.W;
    n_ow -= cw ? p.W : 0;
    int n_oh = st_oh + d_oh + (cw ? 1 : 0);
    if (n_oh >= p.H) n_oh -= p.H;
    const unsigned n_rot = (rot + 4096u) & (RING_BYTES - 1);
    const TapOk nk = tap_ok(n_oh, n_ow);
    (void)rot;
    // the fragment reads run ONE TAP AHEAD of the MFMAs that consume them: with the reads of tap t issued right in front of
    // its MFMAs a wave waited out the LDS latency nine times per step (the kernel ran at ~30 % MFMA-busy whatever its fill scheme)
    auto rd_tap = [&](int t, u32x4& fb) {
#if defined(WG_ABL) && WG_ABL == 2   // diagnostic: no x-tap fragment reads (operands = address bits)
      fb = (u32x4){al[t], ah[t], 0x3c003c00u, 0x3c003c00u};
      return;
#endif
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_ptr(al[t]));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_ptr(ah[t]));
      uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      fb = (u32x4){l2.x, l2.y, h2.x, h2.y};
    };
    u32x4 fbA, fbB;
    rd_tap(0, fbA);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      u32x4& cur = (t & 1) ? fbB : fbA;
      u32x4& nxt = (t & 1) ? fbA : fbB;
      if (t + 1 < NT) rd_tap(t + 1, nxt);
#if defined(WG_ABL) && WG_ABL == 1   // diagnostic: no MFMAs (fragments kept alive)
      asm volatile("" ::"v"(cur), "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]));
#else
#pragma unroll
      for (int a = 0; a < TA; ++a)
        acc[t][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[a]),
                                                            __builtin_bit_cast(bf16x8_t, cur), acc[t][a], 0, 0, 0);
#endif
      // (tap t's addresses were consumed by rd_tap(t) one iteration ago: replace them by the next step's, under these MFMAs)
      if (t >= 1) tap_addr(t - 1, nk, n_rot, al[t - 1], ah[t - 1]);
      if (PP_OK && t == PP_SPLIT - 1 && pp) {   // the ping-pong's mid-step barrier (the K loop below)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    tap_addr(NT - 1, nk, n_rot, al[NT - 1], ah[NT - 1]);
    st_oh = n_oh;
    st_ow = n_ow;
  };

  const int nsteps = max(s_end - s_begin, 0);
  if (iters > 0) {   // (every wave of the workgroup takes the same `iters` trips: the barriers are workgroup-wide)
    // prologue: the window of the first step, [P0 - HLa, P0 + 32 + HLa), its dy tile, and the fill groups of the next
    // RING_D - 1 steps; everything waited for once
    if (nsteps > 0) {
      for (int pc = wv; pc * 8 < 32 + 2 * p.HLa; pc += 4) x_piece(P0 - p.HLa + pc * 8);
      dy_piece(0, s_begin);
      for (int j = 1; j < RING_D && j < nsteps; ++j) dma_step(j, s_begin + j);
    }
    ring_wait_vmcnt<0>();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the zero row's ds_write
    __builtin_amdgcn_s_barrier();
    unsigned rot = 0u;
    if (pp && grp == 1) __builtin_amdgcn_s_barrier();   // group 1: one barrier (= half a step) behind, for the whole loop
    for (int k = 0; k < iters; ++k) {
      if (GROUPS == 1 || k < nsteps) {
        // fills RING_D steps ahead: dy stage (k + RING_D) & 3 was read in step k - 1 (every wave is past its barrier); the
        // ring slots they overwrite hold pixels 256 rows back, behind this step's window (host: HL + HLa + 32 RING_D + 32 <= 256)
#if !(defined(WG_ABL) && WG_ABL == 3)   // diagnostic 3: no fills inside the K loop
        if (k + RING_D < nsteps) dma_step((k + RING_D) % RING_NDY, s_begin + k + RING_D);
#endif
        compute(k % RING_NDY, rot);
        rot = (rot + 4096u) & (RING_BYTES - 1);
        // step k + 1's fills have landed: all but the groups of the steps after it (2 DMAs each) -- this wave's, then
        // (barrier) everybody's; the LDS reads of this step are back, so its dy stage and ring rows may be overwritten
        const int younger = (k + RING_D < nsteps ? k + RING_D : nsteps - 1) - (k + 1);
        if (younger >= 2) ring_wait_vmcnt<4>();
        else if (younger == 1) ring_wait_vmcnt<2>();
        else ring_wait_vmcnt<0>();
      } else if (pp) {
        __builtin_amdgcn_s_barrier();       // (a group that has run out of steps keeps the barrier count: the mid-step one)
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
    }
    if (pp && grp == 0) __builtin_amdgcn_s_barrier();   // level again
  }

  if constexpr (GROUPS == 2) {   // group 1 -> LDS -> group 0 (see wgrad_kernel)
    float* red = reinterpret_cast<float*>(smem_wg) + (tid & 255);
    if (grp == 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
          for (int j = 0; j < 4; ++j) red[((t * TA + a) * 4 + j) * 256] = acc[t][a][j];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][a][j] += red[((t * TA + a) * 4 + j) * 256];
  }

  // ---- slab store (identical layout to wgrad_kernel: [split][tap][co][ci])
  float* base = p.slab + (((size_t)split * NT) * p.Cout + co0 + fq * 4) * p.Cin + ci0 + wave * 16 + fr;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) base[((size_t)t * p.Cout + a * 16 + j) * p.Cin] = acc[t][a][j];
}

// exact unsigned division by d for dividends < 2^31: q = (x * m) >> sh
void magic_div(unsigned d, unsigned& m, unsigned& sh) {
  int l = 0;
  while ((1u << l) < d) ++l;
  m = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
  sh = 31 + l;
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
    for (; k + 12 < nsplit; k += 16) {   // four independent loads per trip (same two interleaved sums as below)
      const float a0 = slab[(size_t)k * per + i], a1 = slab[(size_t)(k + 4) * per + i];
      const float a2 = slab[(size_t)(k + 8) * per + i], a3 = slab[(size_t)(k + 12) * per + i];
      s0 += a0;
      s1 += a1;
      s0 += a2;
      s1 += a3;
    }
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

template <typename T, int NT, int GROUPS>
int launch_wgrad_nt(WgradParams& p, dim3 grid, hipStream_t stream) {
  size_t lds = (size_t)GROUPS * 2 * (1 + NT) * 4096;
  if (GROUPS == 2 && lds < (size_t)NT * 16 * 1024) lds = (size_t)NT * 16 * 1024;   // the hand-over buffer: NT*16 floats x 256 threads
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<T, NT, GROUPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_kernel<T, NT, GROUPS>), grid, dim3(256 * GROUPS), lds, stream, p);
  ECG_CHECK_LAUNCH("wgrad_kernel");
  return 0;
}

// two 4-wave groups per workgroup (see wgrad_kernel): bf16 filters with 3 or 9 taps, where the slabs are the larger part of
// the launch's traffic.  ECGMM_WGRAD_GROUPS=1 keeps 256-thread workgroups (A/B switch).
int g_wgrad_groups = -1;
int wgrad_groups(int dtype, const ConvGeom& g) {
  if (g_wgrad_groups < 0) {
    const char* e = getenv("ECGMM_WGRAD_GROUPS");
    g_wgrad_groups = e && e[0] == '1' ? 1 : 2;
  }
  return dtype == ECGMM_BF16 && g.R * g.S >= 3 ? g_wgrad_groups : 1;
}

template <typename T>
int launch_wgrad(const ConvGeom& g, WgradParams& p, int nsplit, int groups, hipStream_t stream) {
  const int RS = g.R * g.S;
  dim3 grid(ceil_div(g.Cout, 64) * ceil_div(g.Cin, 64), nsplit);
  if (g.S != 1 && g.S != 3) ECG_FAIL(ECGMM_ERR_SHAPE, "conv wgrad: filter width %d unsupported (1 or 3)", g.S);
  if constexpr (sizeof(T) == 2) {
    if (groups == 2 && RS == 3) return launch_wgrad_nt<T, 3, 2>(p, grid, stream);
    if (groups == 2 && RS == 9) return launch_wgrad_nt<T, 9, 2>(p, grid, stream);
  }
  if (RS == 1) return launch_wgrad_nt<T, 1, 1>(p, grid, stream);
  if (RS == 3) return launch_wgrad_nt<T, 3, 1>(p, grid, stream);
  if (RS == 9) return launch_wgrad_nt<T, 9, 1>(p, grid, stream);
  ECG_FAIL(ECGMM_ERR_SHAPE, "conv wgrad: %dx%d filter unsupported (1, 3 or 9 taps)", g.R, g.S);
}

int g_wgrad_ring = -1;   // ECGMM_WGRAD_RING: 0 = wgrad_kernel everywhere, 1 = ring kernel where it is faster (default), 2 = wherever applicable

bool wgrad_ring_ok(int dtype, const ConvGeom& g) {
  if (g_wgrad_ring < 0) {
    const char* e = getenv("ECGMM_WGRAD_RING");
    g_wgrad_ring = e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1;
  }
  if (!g_wgrad_ring || dtype != ECGMM_BF16 || g.stride != 1 || g.S != 3 || g.pad_w != 1) return false;
  if (!((g.R == 3 && g.pad_h == 1) || (g.R == 1 && g.pad_h == 0))) return false;
  if (g.OH != g.H || g.OW != g.W || g.W < 4 || g.Cin % 64 || g.Cout % 64) return false;
  const int HL = g.pad_h * g.W + 1, HLa = (HL + 7) / 8 * 8;
  if (HL + HLa + 32 * RING_D + 32 > RING_ROWS) return false;
  const double M = (double)g.N * g.H * g.W;
  if (!(M * g.Cin * 2.0 < 2.0e9 && M * g.Cout * 2.0 < 2.0e9)) return false;
  if (g_wgrad_ring == 2) return true;
  // Same-call A/B at B = 256 with the 4 x 1 wave layout (profiles/r02_wgrad_ring_ab.txt, second table): the ring form is
  // 5-8 % faster on every 3x3 layer and on the 128/256-channel 1-D k = 3 layers, 12 % slower on the 64-channel 1-D layer
  // (its halo is a small share of a 1250-pixel row; wgrad_kernel's two-group form wins there).
  return g.R == 3 || g.Cin >= 128;
}

template <int NT, int GROUPS>
int launch_wgrad_ring(const WgradRingParams& p, dim3 grid, hipStream_t stream) {
  int lds = GROUPS * RING_LDS;
  if (GROUPS == 2 && lds < NT * 16 * 1024) lds = NT * 16 * 1024;   // the hand-over buffer
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_ring_kernel<NT, GROUPS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_ring_kernel<NT, GROUPS>), grid, dim3(256 * GROUPS), lds, stream, p);
  ECG_CHECK_LAUNCH("wgrad_ring_kernel");
  return 0;
}

bool g_wgrad_narrow = false;
int g_wgrad_narrow_slots = 256;
int pick_nsplit(const ConvGeom& g, int kp, int groups) {
  long M = (long)g.N * g.OH * g.OW;
  int tiles = ceil_div(g.Cout, 64) * ceil_div(g.Cin, 64);
  int steps = ceil_div(M, kp);
  // One resident round.  Alone on the GPU a launch wants every CU (512 four-wave slots); on the weight-gradient SIDE
  // stream it runs beside the critical path's dgrad -> BatchNorm chain, and -- an 8-wave, 144 KB-LDS workgroup shares
  // its CU with nothing -- every CU it holds is one the chain waits for.  Narrow (half the CUs, twice as long) the step is
  // 0.2 ms faster: sweep 512 / 384 / 256 / 192 / 128 slots = 7.39 / 7.24 / 7.18 / 7.21 / 7.82 ms per step (round 2), and the
  // slabs shrink with the split count.  Round 3, after the conv and weight-gradient kernels got faster (same-call sweep of the
  // multimodal step at batch 256, tools/ab_env.sh): 384 / 320 / 256 / 224 / 192 / 160 slots = 6.95 / 6.93 / 6.88 / 6.84 / 6.81 /
  // 6.85 ms -- there the dgrad -> BatchNorm chain is the critical path of the backward and the side stream has slack, so the
  // ResNet18 plan asks for 192 (96 of the 256 CUs) from batch 192 up; the image-only step at batch 128 (3.445 / 3.48 ms) and
  // the 12-lead signal encoder at batch 512 (2.95 / 3.13 ms) are faster at 256 and keep it (the caller's choice:
  // ecg_conv_wgrad_narrow's second argument).  ECGMM_WGRAD_WGS overrides.
  static const int slots_env = [] { const char* e = getenv("ECGMM_WGRAD_WGS"); return e ? atoi(e) : 0; }();
  const int slots = slots_env > 0 ? slots_env : (g_wgrad_narrow ? g_wgrad_narrow_slots : 512);
  int want = ceil_div(slots / groups, tiles);
  int ns = want < 1 ? 1 : want;
  if (ns > steps) ns = steps;
  if (ns > 512) ns = 512;
  return ns < 1 ? 1 : ns;
}

}  // namespace

// Ping-pong between the two wave groups of wgrad_ring_kernel<9, 2> (see its K loop): 1 = on, 0 = lock step (DEFAULT: measured
// stand-alone at batch 256, layers 1-4: 86.2 / 85.3 / 81.4 / 90.6 us in lock step, 86.6 / 86.7 / 82.4 / 91.8 us with it -- this
// kernel's K step is not limited by the two groups meeting at the matrix pipe).  Start-up value: ECGMM_WGRAD_PP.  Bit-identical results.
static int g_wgrad_pp = -1;
extern "C" int ecgmm_conv_wgrad_pingpong(int on) {
  g_wgrad_pp = on != 0;
  return 0;
}
// Runtime switch (same-process A/B, tools/conv_bench.py --ring): 0 = every weight gradient on wgrad_kernel.
extern "C" int ecgmm_conv_wgrad_ring_enable(int on) {
  g_wgrad_ring = on < 0 ? 0 : on > 2 ? 2 : on;
  return 0;
}

// narrow = the caller runs its weight gradients on a side stream beside other work (see pick_nsplit)
void ecg_conv_wgrad_narrow(bool narrow, int slots) { g_wgrad_narrow = narrow; g_wgrad_narrow_slots = slots > 0 ? slots : 256; }

size_t ecg_conv_wgrad_workspace(int dtype, const ConvGeom& g) {
  const bool keep = g_wgrad_narrow;
  g_wgrad_narrow = false;   // (upper bound over both settings)
  struct Restore { bool v; ~Restore() { g_wgrad_narrow = v; } } restore{keep};
  int ns = pick_nsplit(g, dtype == ECGMM_BF16 ? 32 : 16, 1);   // (upper bound over the kernel variants)
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
  const bool ring = wgrad_ring_ok(dtype, g);
  const int groups = wgrad_groups(dtype, g);
  int ns = pick_nsplit(g, kp, groups);
  size_t need = (size_t)ns * g.R * g.S * g.Cout * g.Cin * sizeof(float);
  if (workspace_bytes < need || !workspace)
    ECG_FAIL(ECGMM_ERR_WORKSPACE, "conv wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
  WgradParams p;
  p.x = x; p.dy = dy; p.slab = (float*)workspace;
  p.N = g.N; p.H = g.H; p.W = g.W; p.Cin = g.Cin; p.OH = g.OH; p.OW = g.OW; p.Cout = g.Cout; p.S = g.S;
  p.stride = g.stride; p.pad_h = g.pad_h; p.pad_w = g.pad_w; p.M = (int)M;
  p.steps_per_split = ceil_div(ceil_div(M, kp), ns);
  magic_div((unsigned)(g.OH * g.OW), p.mul_hw, p.sh_hw);
  magic_div((unsigned)g.OW, p.mul_w, p.sh_w);
  if (dtype != ECGMM_BF16 && dtype != ECGMM_F32) ECG_FAIL(ECGMM_ERR_DTYPE, "conv wgrad: bad dtype %d", dtype);
  ecg_prof_begin(ECG_PROF_WGRAD, 2.0 * (double)M * g.Cout * g.R * g.S * g.Cin,
                 (double)dtype_size(dtype) * ((double)g.N * g.H * g.W * g.Cin + (double)M * g.Cout) +
                     4.0 * g.R * g.S * g.Cin * g.Cout,
                 stream);
  int rc;
  if (ring) {
    WgradRingParams q;
    q.x = x; q.dy = dy; q.slab = (float*)workspace;
    q.H = g.H; q.W = g.W; q.Cin = g.Cin; q.Cout = g.Cout; q.pad_h = g.pad_h; q.M = (int)M;
    q.steps_per_split = p.steps_per_split;
    q.HL = g.pad_h * g.W + 1;
    q.HLa = (q.HL + 7) / 8 * 8;
    q.mul_hw = p.mul_hw; q.sh_hw = p.sh_hw; q.mul_w = p.mul_w; q.sh_w = p.sh_w;
    if (g_wgrad_pp < 0) { const char* e = getenv("ECGMM_WGRAD_PP"); g_wgrad_pp = (e && e[0] == '1'); }
    q.pingpong = g_wgrad_pp;
    dim3 grid((g.Cout / 64) * (g.Cin / 64), ns);
    if (groups == 2) rc = g.R == 3 ? launch_wgrad_ring<9, 2>(q, grid, stream) : launch_wgrad_ring<3, 2>(q, grid, stream);
    else rc = g.R == 3 ? launch_wgrad_ring<9, 1>(q, grid, stream) : launch_wgrad_ring<3, 1>(q, grid, stream);
  } else {
    rc = dtype == ECGMM_BF16 ? launch_wgrad<bf16_t>(g, p, ns, groups, stream) : launch_wgrad<float>(g, p, ns, 1, stream);
  }
  ecg_prof_end(stream);
  ECG_TRY(rc);
  size_t per = (size_t)g.R * g.S * g.Cout * g.Cin;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(per, 64)), dim3(256), 0, stream, (const float*)workspace,
                     grad_oihw, ns, g.R * g.S, g.Cout, g.Cin, accumulate);
  ECG_CHECK_LAUNCH("wgrad_reduce_kernel");
  return 0;
}
