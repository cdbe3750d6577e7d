// Stride-1 "same" convolutions (3x3 pad 1, 1x3 pad 1) on MFMA with a HALO-resident activation tile, gfx950, bf16.
//
// Why a second implicit-GEMM kernel (conv_igemm.hip stays the general one): on the 128 x 128 tile of conv_igemm every
// K stage moves (128 + 128) rows x 128 B through the CU's vector-memory path for 32 MFMAs per wave -- at 64 B/clk that
// path is as busy as the matrix pipe (measured: 8 LDS-DMA pieces per wave take ~40 % of an iteration), and the nine
// taps of a 3x3 filter fetch (almost) the same pixels nine times.  Here a workgroup owns 256 consecutive pixels of the
// flat N*H*W index (channels-last => a CONTIGUOUS byte range) and keeps, per 64-channel slice, that range plus W+1
// pixels on each side in LDS: tap (dr, ds) of pixel q is row q + dr*W + ds of the same image, so all nine taps read one
// resident copy.  Per K step (one tap x 64 channels) only the weights move: BN rows instead of BN + 256.
//
//   tile        256 pixels x BN channels (BN = 128, or 64 for the 64-channel layers); 512 threads = 8 waves = 4 pixel
//               quarters x 2 channel halves, 2 waves per SIMD, ONE workgroup per CU (<= 150 KB of LDS)
//   LDS         2 halo buffers (the next 64-channel slice streams in while this one is used) + a 4-slot weight ring;
//               every fill is LDS-DMA (buffer_load ... lds), two steps ahead, retired by a COUNTED s_waitcnt vmcnt(N)
//               and one raw s_barrier per step (nothing drains to vmcnt(0) inside the loop)
//   fragments   two register sets: the reads of k-step 1 are issued before the MFMAs of k-step 0, and the reads of the
//               NEXT step's k-step 0 (its slot landed one barrier earlier) before the MFMAs of k-step 1
//   borders     a tap that leaves the image (or crosses into the neighbouring image / row) must read zeros although its
//               LDS row holds a real neighbouring pixel: every lane keeps, per (pixel tile, tap), the LDS address it
//               reads from, and invalid taps point at an all-zero LDS row.  No per-step validity arithmetic.
//
// Numerics: same products as conv_igemm, summed channel-slice-major instead of tap-major (fp32 accumulation order).
// Reference semantics: torchvision BasicBlock 3x3 convs (multimodal_paper_modal_balance.py:210) and BasicBlock1D's
// k=3 Conv1d (:71-81), forward and input gradient.
#include <type_traits>

#include "ops.h"

namespace {

constexpr int HBM_ = 256;        // pixels per workgroup
constexpr int HTHREADS = 512;
constexpr unsigned H_OOB = 0x80000000u;  // stays out of range after a channel-slice offset is added (tensors < 2 GiB)

struct HaloParams {
  const void* src;
  const void* wpk;
  void* dst;
  const float* bias;
  const void* addend;
  float* stats;
  int M, H, W, Cs, Cd, ph, pw, act;
  int ncs;    // Cs / 64
  int HL;     // ph * W + pw: halo pixels on each side
  int hrows;  // 256 + 2 * HL
  int ntm;    // pixel tiles (M / 256)
  int ntiles; // ntm * (Cd / BN)
};

__device__ __forceinline__ void hdma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_wave_base, unsigned voffset,
                                       unsigned soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset,
                                           soffset, 0, 0);
#endif
}

// s_waitcnt vmcnt(N): all but the wave's N youngest vector-memory operations (here: LDS-DMA fills) have completed.
// Inline asm on purpose: hipcc does not know this wait, and it must not replace it by its own vmcnt(0).
template <int N> __device__ __forceinline__ void wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// s_waitcnt lgkmcnt(0) as the BUILTIN: hipcc's own LDS-read bookkeeping must see it, or it waits again (too strictly)
// in front of the MFMAs that follow the next fragment reads
__device__ __forceinline__ void wait_lds() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
}

// two floats -> packed bf16 pair with ONE v_cvt_pk_bf16_f32 (round to nearest even, NaN kept: same values as f2bf)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}

template <int BN> struct HaloCfg {
  static constexpr int HCAP = BN == 128 ? 320 : 384;  // halo rows a buffer holds (256 + 2 * (W + 1) must fit)
  static constexpr int HBUF = (HCAP + 8) * 128;       // + one 8-row piece whose first row is the all-zero row
  static constexpr int WSTAGE = BN * 128;
  static constexpr int NWS = 4;
  static constexpr int WBASE = 2 * HBUF;
  static constexpr int LDS = WBASE + NWS * WSTAGE;
  static constexpr int WPS = BN / 64;                 // weight DMA pieces per wave per step
  static constexpr int NPW_MAX = HCAP / 64;           // halo DMA pieces per wave per channel slice
};

template <int BN, int RS, int MODE>
__global__ __launch_bounds__(HTHREADS) void conv_halo_kernel(HaloParams p) {
  using C = HaloCfg<BN>;
  constexpr int TP = 4;         // 16-pixel tiles per wave (64 pixels)
  constexpr int TC = BN / 32;   // 16-channel tiles per wave
  constexpr int HPS = RS == 9 ? 1 : C::NPW_MAX;  // halo pieces a wave issues per step (3-tap filters: all in the slice's first step)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wv & 3, wc = wv >> 2;
  const int fr = lane & 15, fq = lane >> 4;
  const int G = gridDim.x;
  // Workgroups are PERSISTENT: workgroup b takes tiles perm(b), perm(b) + G, ... of the (channel tile, pixel tile) list.
  // perm gives the workgroups that share an XCD (equal b % 8: MI355X_MICROARCH.md, dispatch) a contiguous run of pixel
  // tiles, whose halos overlap, so they share that XCD's L2 -- speed only, bijective for any G.
  int T;
  {
    const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    T = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const unsigned pixb = (unsigned)p.Cs * 2u;  // bytes per pixel row of the source
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((size_t)p.M * pixb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wpk, 0, (int)((size_t)p.Cd * RS * p.Cs * 2), 0x00020000);

  // ---- DMA bookkeeping: lane -> (row of an 8-row piece, logical 16-B chunk); the XOR swizzle sits on the SOURCE side
  const int lrow8 = lane >> 3;
  const unsigned lchunkb = (unsigned)(((lane & 7) ^ lrow8) << 4);
  unsigned hoff[C::NPW_MAX];  // byte offset (channel slice 0) of this lane's 16 B of halo piece kk, or out of range
  auto set_hoff = [&](int m0) {
#pragma unroll
    for (int kk = 0; kk < C::NPW_MAX; ++kk) {
      const int row = (kk * 8 + wv) * 8 + lrow8;
      const int pix = m0 - p.HL + row;
      const bool ok = row < p.hrows && pix >= 0 && pix < p.M;
      hoff[kk] = ok ? (unsigned)pix * pixb + lchunkb : H_OOB;
    }
  };
  auto set_wrow = [&](unsigned (&wr)[C::WPS], int n0) {
#pragma unroll
    for (int i = 0; i < C::WPS; ++i) wr[i] = (unsigned)(n0 + (i * 8 + wv) * 8 + lrow8) * (unsigned)RS * pixb + lchunkb;
  };
  // halo pieces [k0, k1) of this wave, channel slice `cs`, into buffer `hbuf` (a piece wholly past the halo's last row
  // is all out of range: zero fill into rows nobody reads -- the piece count stays a compile-time constant)
  auto dma_halo = [&](unsigned hbuf, int cs, int k0, int k1) {
#pragma unroll
    for (int kk = 0; kk < C::NPW_MAX; ++kk)
      if (kk >= k0 && kk < k1) hdma16(rs_src, smem + hbuf + (kk * 8 + wv) * 1024, hoff[kk], (unsigned)cs * 128u);
  };
  auto dma_w = [&](const unsigned (&wr)[C::WPS], int slot, int cs, int t) {
    const unsigned so = ((unsigned)t * (unsigned)p.Cs + (unsigned)cs * 64u) * 2u;
#pragma unroll
    for (int i = 0; i < C::WPS; ++i) hdma16(rs_w, smem + C::WBASE + slot * C::WSTAGE + (i * 8 + wv) * 1024, wr[i], so);
  };

  // ---- fragment addresses
  // weights (A operand): row = channel, conflict-free b128 reads via chunk ^ (row & 7)
  const unsigned aoff0 = (unsigned)C::WBASE + (unsigned)(wc * (BN / 2) + fr) * 128u + (unsigned)((fq ^ (fr & 7)) << 4);
  // activations (B operand): per (pixel tile b, tap t) the halo-relative LDS byte this lane reads for k-step 0
  // (k-step 1 = the same address ^ 64); a tap outside the image reads the zero row
  // (two 16-bit addresses per register: 4 x RS of them live through the whole K loop)
  constexpr int RSP = (RS + 1) / 2;
  unsigned baddr[TP][RSP];
  auto bad = [&](int b, int t) -> unsigned { return (t & 1) ? baddr[b][t >> 1] >> 16 : baddr[b][t >> 1] & 0xFFFFu; };
  const int HW = p.H * p.W;
  auto set_baddr = [&](int m0) {
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const int pl = wp * 64 + b * 16 + fr;
      const int q = m0 + pl;
      const int n = q / HW, rem = q - n * HW;
      const int h = rem / p.W, w = rem - h * p.W;
      bool okh[3], okw[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        okh[j] = (unsigned)(h + (MODE == 0 ? j - p.ph : p.ph - j)) < (unsigned)p.H;
        okw[j] = (unsigned)(w + (MODE == 0 ? j - p.pw : p.pw - j)) < (unsigned)p.W;
      }
#pragma unroll
      for (int t2 = 0; t2 < RSP; ++t2) baddr[b][t2] = 0u;
#pragma unroll
      for (int t = 0; t < RS; ++t) {
        const int r = t / 3, s = t - r * 3;
        const int dr = MODE == 0 ? r - p.ph : p.ph - r;
        const int ds = MODE == 0 ? s - p.pw : p.pw - s;
        const int R = p.HL + pl + dr * p.W + ds;
        const unsigned a = (okh[r] && okw[s]) ? (unsigned)R * 128u + (unsigned)(((fq ^ R) & 7) << 4)
                                              : (unsigned)C::HCAP * 128u + (unsigned)(fq << 4);
        baddr[b][t >> 1] |= a << (16 * (t & 1));
      }
    }
  };

  f32x4 acc[TC][TP];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  auto ld_a = [&](u32x4 (&f)[TC], int slot, unsigned ksx) {
#if defined(HALO_ABL) && HALO_ABL == 2   // diagnostic: no LDS fragment reads (operands = address bits)
#pragma unroll
    for (int a = 0; a < TC; ++a) f[a] = (u32x4){aoff0, ksx, (unsigned)slot, 0x3c003c00u};
    return;
#endif
    const unsigned char* base = smem + ((aoff0 ^ ksx) + (unsigned)(slot * C::WSTAGE));
#pragma unroll
    for (int a = 0; a < TC; ++a) f[a] = *reinterpret_cast<const u32x4*>(base + a * 2048);
  };
  auto ld_b = [&](unsigned off) -> u32x4 {
#if defined(HALO_ABL) && HALO_ABL == 2
    return (u32x4){off, 0x3c003c00u, off, 0x3c003c00u};
#endif
    return *reinterpret_cast<const u32x4*>(smem + off);
  };
  auto mma = [&](const u32x4 (&fa)[TC], const u32x4 (&fb)[TP]) {
#if defined(HALO_ABL) && HALO_ABL == 1   // timing-only diagnostic build (make halo_abl): no MFMAs, fragments kept alive
#pragma unroll
    for (int a = 0; a < TC; ++a) asm volatile("" ::"v"(fa[a]));
#pragma unroll
    for (int b = 0; b < TP; ++b) asm volatile("" ::"v"(fb[b]));
    return;
#endif
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[a]),
                                                            __builtin_bit_cast(bf16x8_t, fb[b]), acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- tile loop.  Per tile: the fills of its first halo slice and first two weight slots are ALREADY in flight
  // (issued before the loop for the first tile, and right after the previous tile's last K step otherwise: they land
  // while that tile's epilogue runs), the address table is built, then one wait + barrier, then the K loop.
  int nt = T / p.ntm, mt = T - nt * p.ntm;
  int m0 = mt * HBM_, n0 = nt * BN;
  unsigned wrow[C::WPS];
  int g0 = 0;  // K steps done so far (all tiles): step g uses weight slot g & 3
  int qs = 0;  // channel slices done so far: slice q uses halo buffer q & 1
  auto tile_fills = [&](int m0_, int n0_) {
    set_hoff(m0_);
    set_wrow(wrow, n0_);
    dma_halo((unsigned)(qs & 1) * C::HBUF, 0, 0, C::NPW_MAX);
    dma_w(wrow, g0 & 3, 0, 0);
    dma_w(wrow, (g0 + 1) & 3, 0, 1);
    dma_w(wrow, (g0 + 2) & 3, 0, 2);
  };
  if (tid < 16) *reinterpret_cast<u32x4*>(smem + (tid >> 3) * C::HBUF + C::HCAP * 128 + (tid & 7) * 16) = (u32x4){0u, 0u, 0u, 0u};
  tile_fills(m0, n0);

  u32x4 fa0[TC], fb0[TP], fa1[TC], fb1[TP];

  // One channel slice = RS steps, fully unrolled and branch-free, so that every wait count is a compile-time constant
  // and hipcc's LDS-read bookkeeping stays exact.  LAST = the tile's last slice: no fills past the tile's end.
  auto slice = [&](int cs, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const unsigned hb = (unsigned)(qs & 1) * C::HBUF, hbn = (unsigned)((qs + 1) & 1) * C::HBUF;
    // keep the fragment addresses opaque per slice: otherwise their loop-invariant variants (^ 64, + buffer base) are
    // hoisted out of the loop and the kernel spills
#pragma unroll
    for (int b = 0; b < TP; ++b)
#pragma unroll
      for (int t2 = 0; t2 < RSP; ++t2) asm volatile("" : "+v"(baddr[b][t2]));
#pragma unroll
    for (int t = 0; t < RS; ++t) {
      const int s = g0 + t;
      // (a) k-step 1 of this step into the second register set
      ld_a(fa1, s & 3, 64u);
#pragma unroll
      for (int b = 0; b < TP; ++b) fb1[b] = ld_b((bad(b, t) ^ 64u) + hb);
      // (b) fills THREE steps ahead: weights of step s + 3 go into the slot step s - 1 read (every wave is past that step's
      // barrier), the next slice's halo into the other halo buffer.  Two steps' fills stay in flight across each barrier.
      const auto n_halo = [](int tt, bool last) { return (!last && tt < RS - 1 && tt * HPS < C::NPW_MAX)
                                                      ? ((tt + 1) * HPS < C::NPW_MAX ? HPS : C::NPW_MAX - tt * HPS) : 0; };
      const auto n_fill = [&](int tt, bool last) { return ((last && tt + 3 >= RS) ? 0 : C::WPS) + n_halo(tt, last); };
      const bool wrap = t + 3 >= RS;
#if !(defined(HALO_ABL) && HALO_ABL == 3)   // diagnostic 3: no fills inside the K loop
      if (!(LAST && wrap)) dma_w(wrow, (s + 3) & 3, cs + (wrap ? 1 : 0), (t + 3) % RS);
      if (n_halo(t, LAST) > 0) dma_halo(hbn, cs + 1, t * HPS, (t + 1) * HPS);
#endif
      // (c) k-step 0
      mma(fa0, fb0);
      // (d) the fills of step s + 1 (issued two steps ago) have landed -- this wave's, then (barrier) everybody's; all that
      // may still be in flight are the previous step's and this step's fills.  (The step before a slice's first one is a
      // non-final slice's last step, or the tile prologue, whose fills were all waited for: a larger count is then moot.)
      // The LDS reads of (a) are back too, so the slots they read may be refilled by the next step.
      const int outstanding = (t == 0 ? n_fill(RS - 1, false) : n_fill(t - 1, LAST)) + n_fill(t, LAST);
      static_assert(2 * (C::WPS + C::NPW_MAX) <= 16, "wait_vmcnt table");
      if (outstanding == 0) wait_vmcnt<0>();
      else if (outstanding == 1) wait_vmcnt<1>();
      else if (outstanding == 2) wait_vmcnt<2>();
      else if (outstanding == 3) wait_vmcnt<3>();
      else if (outstanding == 4) wait_vmcnt<4>();
      else if (outstanding == 5) wait_vmcnt<5>();
      else if (outstanding == 6) wait_vmcnt<6>();
      else if (outstanding == 7) wait_vmcnt<7>();
      else if (outstanding == 8) wait_vmcnt<8>();
      else if (outstanding == 9) wait_vmcnt<9>();
      else if (outstanding == 10) wait_vmcnt<10>();
      else if (outstanding == 11) wait_vmcnt<11>();
      else if (outstanding == 12) wait_vmcnt<12>();
      else wait_vmcnt<0>();
      wait_lds();
#if !(defined(HALO_ABL) && HALO_ABL == 5)   // diagnostic 5: no per-step barrier
      __builtin_amdgcn_s_barrier();
#endif
      // (e) k-step 0 of the next step (none after the tile's last step)
      if (!(LAST && t == RS - 1)) {
        const int t1 = (t + 1) % RS;
        ld_a(fa0, (s + 1) & 3, 0u);
        const unsigned hb1 = t + 1 < RS ? hb : hbn;
#pragma unroll
        for (int b = 0; b < TP; ++b) fb0[b] = ld_b(bad(b, t1) + hb1);
      }
      // (f) k-step 1
      mma(fa1, fb1);
    }
    g0 += RS;
    ++qs;
  };

  bf16_t* __restrict__ dst = (bf16_t*)p.dst;
  const bf16_t* __restrict__ addend = (const bf16_t*)p.addend;
  bool first_tile = true;
  for (;;) {
    set_baddr(m0);
    zero_acc();
    // This tile's first fills have landed.  They were issued BEFORE the previous tile's output stores, and vmcnt retires
    // in issue order: waiting for "all but the N youngest", N = the store instructions of that epilogue, leaves the
    // stores in flight (a vmcnt(0) here exposed their whole write latency once per tile).  N must not exceed what was
    // really issued -- the statistics rows are a wave-uniform runtime choice; operations hipcc adds on its own (scratch)
    // only make the wait stricter.
    constexpr int N_DATA = TC * TP / 2, N_STAT = 2 * TC;
    if (first_tile) wait_vmcnt<0>();
    else if (p.stats) wait_vmcnt<N_DATA + N_STAT>();
    else wait_vmcnt<N_DATA>();
    first_tile = false;
    wait_lds();
    __builtin_amdgcn_s_barrier();
    ld_a(fa0, g0 & 3, 0u);
    {
      const unsigned hb = (unsigned)(qs & 1) * C::HBUF;
#pragma unroll
      for (int b = 0; b < TP; ++b) fb0[b] = ld_b(bad(b, 0) + hb);
    }
    for (int cs = 0; cs + 1 < p.ncs; ++cs) slice(cs, std::integral_constant<bool, false>{});
    slice(p.ncs - 1, std::integral_constant<bool, true>{});
    // every wave is past the last step's barrier, i.e. has read its last fragments: the halo buffer and the two weight
    // slots the next tile starts with are free -- start its fills now, under this tile's epilogue
    const int Tn = T + G;
    const bool more = Tn < p.ntiles;
    const int mt_cur = mt, m0_cur = m0, n0_cur = n0;
    if (more) {
      nt = Tn / p.ntm;
      mt = Tn - nt * p.ntm;
      m0 = mt * HBM_;
      n0 = nt * BN;
      T = Tn;
      tile_fills(m0, n0);
    }

#if !(defined(HALO_ABL) && HALO_ABL == 4)   // diagnostic 4: no epilogue (accumulators kept alive)
    // ---- epilogue of this tile (whole tiles only: the host dispatches this kernel for M % 256 == 0, Cd % BN == 0).
    // lane owns pixel fr x 4 consecutive channels fq*4.. of each 16 x 16 tile; the 8-byte packs of two neighbouring
    // channel tiles are exchanged between lane rows (v_permlane16_swap) so every lane stores 16 contiguous bytes
    {
      size_t prow[TP];
#pragma unroll
      for (int b = 0; b < TP; ++b) prow[b] = (size_t)(m0_cur + wp * 64 + b * 16 + fr) * p.Cd + n0_cur + wc * (BN / 2);
      float s1[TC][4], s2[TC][4];
      f32x4 bias4[TC];
#pragma unroll
      for (int a = 0; a < TC; ++a) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[a][j] = s2[a][j] = 0.f;
        bias4[a] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0_cur + wc * (BN / 2) + a * 16 + fq * 4)
                          : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int a = 0; a < TC; a += 2)
#pragma unroll
        for (int b = 0; b < TP; ++b) {
          f32x4 v0 = acc[a][b], v1 = acc[a + 1][b];
          if (p.bias) {
            v0 += bias4[a];
            v1 += bias4[a + 1];
          }
          if (addend) {
            const bf16_t* ap = addend + prow[b] + fq * 4;
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
          const unsigned x0 = pack_bf16x2(v0[0], v0[1]), y0 = pack_bf16x2(v0[2], v0[3]);
          const unsigned x1 = pack_bf16x2(v1[0], v1[1]), y1 = pack_bf16x2(v1[2], v1[3]);
          auto lo = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
          auto hi = __builtin_amdgcn_permlane16_swap(y0, y1, false, false);
          const int ch = (fq & 1) ? (a + 1) * 16 + (fq - 1) * 4 : a * 16 + fq * 4;
          *reinterpret_cast<u32x4*>(dst + prow[b] + ch) = (u32x4){lo[0], hi[0], lo[1], hi[1]};
        }
#endif
      if (p.stats) {  // BatchNorm partial sums: one row per 64 pixels (the rows conv_igemm writes: 2 per 128 pixels)
        float* srow = p.stats + (size_t)(mt_cur * 4 + wp) * 2 * p.Cd + n0_cur + wc * (BN / 2) + fq * 4;
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
    }
#else
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) asm volatile("" ::"v"(acc[a][b]));
#endif
    if (!more) break;
  }
}

template <int BN, int RS, int MODE>
int launch_halo(const HaloParams& p, hipStream_t stream) {
  using C = HaloCfg<BN>;
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)conv_halo_kernel<BN, RS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS) != hipSuccess)
      ECG_FAIL(ECGMM_ERR_LAUNCH, "conv_halo: cannot reserve %d bytes of LDS", C::LDS);
    attr_set[dev] = true;
  }
  static int ncu[16] = {0};
  if (dev >= 0 && dev < 16 && ncu[dev] == 0) {
    hipDeviceProp_t prop;
    ncu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  HaloParams q = p;
  q.ntm = p.M / HBM_;
  q.ntiles = q.ntm * (p.Cd / BN);
  const int cus = dev >= 0 && dev < 16 ? ncu[dev] : 256;
  dim3 grid(q.ntiles < cus ? q.ntiles : cus);   // persistent: one workgroup per CU walks the tile list
  hipLaunchKernelGGL((conv_halo_kernel<BN, RS, MODE>), grid, dim3(HTHREADS), C::LDS, stream, q);
  ECG_CHECK_LAUNCH("conv_halo_kernel");
  return 0;
}

int g_halo_enabled = -1;  // read once from ECGMM_CONV_HALO: 0 = off, 1 = where it is the faster kernel (default), 2 = wherever applicable

}  // namespace

// Runtime switch (A/B against conv_igemm from one process: tools/conv_bench.py): 0 = never take the halo kernel.
extern "C" int ecgmm_conv_halo_enable(int on) {
  g_halo_enabled = on < 0 ? 0 : on > 2 ? 2 : on;
  return 0;
}

// Is the halo kernel applicable to this (stride-1, "same") convolution?  mode 0 = forward, 1 = input gradient.
bool ecg_conv_halo_ok(int dtype, int mode, const ConvGeom& g) {
  if (g_halo_enabled < 0) {
    const char* e = getenv("ECGMM_CONV_HALO");
    g_halo_enabled = e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1;
  }
  if (!g_halo_enabled || dtype != ECGMM_BF16 || g.stride != 1 || g.S != 3) return false;
  if (!((g.R == 3 && g.pad_h == 1) || (g.R == 1 && g.pad_h == 0)) || g.pad_w != 1) return false;
  if (g.OH != g.H || g.OW != g.W) return false;
  const int Cs = mode == 0 ? g.Cin : g.Cout, Cd = mode == 0 ? g.Cout : g.Cin;
  const long M = (long)g.N * g.H * g.W;
  if (M % HBM_ != 0 || Cs % 64 != 0 || Cd % 64 != 0 || (Cd > 64 && Cd % 128 != 0)) return false;
  const int HL = g.pad_h * g.W + g.pad_w, hrows = 256 + 2 * HL;
  if (hrows > (Cd > 64 ? HaloCfg<128>::HCAP : HaloCfg<64>::HCAP)) return false;
  if ((double)M * Cs * 2.0 > 2.0e9 || (double)M * Cd * 2.0 > 8.0e9) return false;
  if (g_halo_enabled == 2) return true;
  // Where it pays (B = 256 layer shapes, same-call A/B against conv_igemm, profiles/r02_conv_bench_layers.txt): every 3x3
  // layer (+5...+18 %) and the 1-D encoder's last stage.  The short 1-D reductions (64-128 input channels x 3 taps = 3-6
  // K steps per tile) do not amortise the per-tile prologue / epilogue of the one resident workgroup: conv_igemm's two
  // independent workgroups per CU are faster or equal there.
  return g.R == 3 || Cs >= 256;
}

int ecg_conv_halo(int mode, const ConvGeom& g, const void* src, const void* wpk, void* dst, const float* bias,
                  const void* addend, float* stats, int act, hipStream_t stream) {
  HaloParams p;
  memset(&p, 0, sizeof(p));
  p.src = src; p.wpk = wpk; p.dst = dst; p.bias = bias; p.addend = addend; p.stats = stats; p.act = act;
  p.M = g.N * g.H * g.W; p.H = g.H; p.W = g.W; p.ph = g.pad_h; p.pw = g.pad_w;
  p.Cs = mode == 0 ? g.Cin : g.Cout;
  p.Cd = mode == 0 ? g.Cout : g.Cin;
  p.ncs = p.Cs / 64;
  p.HL = g.pad_h * g.W + g.pad_w;
  p.hrows = 256 + 2 * p.HL;
  const bool wide = p.Cd > 64;
  if (g.R == 3) {
    if (mode == 0) return wide ? launch_halo<128, 9, 0>(p, stream) : launch_halo<64, 9, 0>(p, stream);
    return wide ? launch_halo<128, 9, 1>(p, stream) : launch_halo<64, 9, 1>(p, stream);
  }
  if (mode == 0) return wide ? launch_halo<128, 3, 0>(p, stream) : launch_halo<64, 3, 0>(p, stream);
  return wide ? launch_halo<128, 3, 1>(p, stream) : launch_halo<64, 3, 1>(p, stream);
}
