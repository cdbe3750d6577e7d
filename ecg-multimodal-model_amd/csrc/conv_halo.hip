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
constexpr unsigned H_OOB = 0x80000000u;  // stays out of range after a channel-slice offset is added (tensors < 2 GiB)

#ifdef ECG_STAMP
// Diagnostic build only (make stamp): s_memtime stamps at the phase boundaries of a tile, summed per launch by lane 0 of wave 0
// of every workgroup: [0] address table, [1] wait for the tile's first fills + barrier, [2] K loop, [3] epilogue, [4] tiles,
// [5] whole workgroup.  Shares, not lengths (tools/stamp_halo.py).
__device__ unsigned long long g_hstamp[8];
#define HSTAMP_AT(t)                                                          \
  do {                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                        \
  } while (0)
#endif

struct HaloParams {
  const void* src;
  const void* wpk;
  void* dst;
  const float* bias;
  const void* addend;
  float* stats;          // forward: BatchNorm partial sums (sum y, sum y^2) or null
  // dgrad: reduction pass of the BatchNorm backward that consumes this gradient, fused into the epilogue (or null):
  //   g = [mask] * dst (as stored, bf16);  rows of (sum g, sum g * (y - mean))
  //   red_mask == red_y: the ReLU input was bn(y) itself, mask = (y * scale + shift > 0), dst is stored unmasked;
  //   otherwise mask = (red_mask > 0) and dst is stored MASKED (the residual-branch gradient the apply pass reads)
  const void* red_y;
  const void* red_mask;
  const float* red_coef;  // forward coefficients [4][Cd]: scale, shift, mean, invstd
  float* red_rows;
  int wg_rows;            // 1: ONE partial row per workgroup (accumulated over its tiles in LDS, written at the end);
                          // 0: one row per 64 pixels, the layout conv_igemm writes (forward statistics only)
  int M, H, W, Cs, Cd, ph, pw, act;
  int ncs;    // Cs / 64
  int HL;     // ph * W + pw: halo pixels on each side
  int hrows;  // 256 + 2 * HL
  int ntm;    // pixel tiles (M / 256)
  int ntn;    // channel tiles (Cd / BN); gridDim.x is a multiple of it
  int stagger;  // 1: waves 4-7 run their first MFMA block of a step AFTER the step's barrier (see slice())
  unsigned mul_hw, sh_hw, mul_w, sh_w;   // exact division of a pixel index by H*W and by W: q = (x * mul) >> sh (x < 2^31)
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

// NW = waves per workgroup.  8 (4 pixel quarters x 2 channel halves, 2 waves per SIMD, ONE workgroup per CU, two halo
// buffers, 4-slot weight ring) is the kernel described above.  NW = 4 (round 3; BN = 64 only): 4 pixel quarters x all 64
// channels -- the same 64 x 64 wave tile as the BN = 128 form -- in a workgroup small enough for TWO per CU (one halo buffer,
// 3-slot weight ring: 76 KB): a 64-channel tile has half the MFMAs per tile of a 128-channel one for the same prologue /
// epilogue, and 8 waves in lock step cannot overlap the two -- two independent workgroups per CU do (one in its K loop while
// the other stores / builds its address table / waits for its halo).
template <int BN, int NW = 8> struct HaloCfg {
  static_assert(NW == 8 || (NW == 4 && BN == 64), "4-wave workgroups: 64-channel tiles only");
  static constexpr int HCAP = BN == 128 ? 320 : 384;  // halo rows a buffer holds (256 + 2 * (W + 1) must fit)
  static constexpr int HBUF = (HCAP + 8) * 128;       // + one 8-row piece whose first row is the all-zero row
  static constexpr int NHB = NW == 8 ? 2 : 1;          // halo buffers
  static constexpr int WSTAGE = BN * 128;
  static constexpr int NWS = NW == 8 ? 4 : 3;          // weight ring slots; fills run NWS - 1 steps ahead
  static constexpr int WBASE = NHB * HBUF;
  static constexpr int SBASE = WBASE + NWS * WSTAGE;   // fp32 [4 pixel quarters][2][BN] partial-row accumulators
  static constexpr int CBASE = SBASE + 4 * 2 * BN * 4;   // fp32 [3][BN]: scale, shift, mean of the fused BatchNorm backward
  static constexpr int TRASH = CBASE + 3 * BN * 4;       // 256 B nobody reads: destination of the L2-prefetch DMAs
  // stream form (64-channel tiles): per-pixel validity masks of the address pairs, [2 tiles][6 pairs (5 used)][256 pixels] dwords
  static constexpr int MBASE = TRASH + 256;
  static constexpr int MBUF = 6 * 256 * 4;
  static constexpr int LDS = MBASE + (BN == 64 && NW == 8 ? 2 * MBUF : 0);
  static constexpr int WPS = BN / (8 * NW);           // weight DMA pieces per wave per step
  static constexpr int NPW_MAX = HCAP / (8 * NW);     // halo DMA pieces per wave per channel slice
  static constexpr int WCN = NW / 4;                   // channel groups of waves
  static constexpr int CPW = BN / WCN;                 // channels per wave
};

// NCS1: instantiation for ONE channel slice per tile (Cs = 64).  The second halo buffer is then idle during a tile's K
// loop, so the NEXT tile's halo is fetched there during this tile's K loop (through the same in-loop piece schedule
// that otherwise fetches the next channel slice) instead of from the epilogue, where only ~0.6 us of work covered its
// HBM latency: 1.5-2 us of exposed wait per tile, 12 tiles per workgroup on the 56x56 layers.
// PP (round 3, the guide's 8-phase GEMM discipline applied to this loop): PING-PONG between the two waves of a SIMD.  A K step
// becomes four phases -- read the k-step-0 fragments (+ issue the step's fills) | barrier | its 16 MFMAs | barrier | read the
// k-step-1 fragments | barrier | its 16 MFMAs | barrier -- and waves 4-7 run ONE BARRIER behind waves 0-3 (an extra
// s_barrier in front of their K loop, one behind the others'): in every interval one wave of each SIMD issues MFMAs while its
// partner issues LDS reads / fills, instead of both meeting at the matrix pipe and then both at the LDS.  One fragment
// register set instead of two.  Same products in the same order: results bit-identical to the lock-step form.
// ST (round 3): STREAM form of the one-slice (64 -> 64 channel) tiles.  tools/stamp_halo.py (profiles/r03_stamp_halo.txt): of a
// layer-1 tile's ~11.4 k cycles the K loop is 58 %; the rest is the address table, the wait for the tile's first fills + barrier,
// and the epilogue -- phases in which the matrix pipe of all four SIMDs idles, 12 times per workgroup.  With one channel slice
// per tile the weights are the same 9 taps for every tile and the next tile's halo already streams into the other buffer, so the
// K loop never has to stop: the weight ring runs on across the tile boundary, the step-0 fragments of tile i+1 are read in step 8
// of tile i, the first MFMA of a tile starts from a zero C operand (no accumulator clearing), the finished accumulators are
// copied aside and tile i's epilogue (bias / ReLU / statistics / bf16 pack / 16-byte stores) runs in four pieces inside steps
// 1-4 of tile i+1, and the address table of tile i+1 is rebuilt in place, register pair by register pair, as soon as tile i has
// used a pair for the last time (validity masks built once per pixel by the whole workgroup, through a small LDS table).  The loop
// body is branch-free apart from the stagger of waves 4-7.  Same products, same order of accumulation, same statistics grouping:
// bit-identical results.  Batch 256, layer 1 (tools/conv_bench.py --stream 0|1, profiles/r03_halo_stream.txt): forward with
// statistics 81 -> 72.5 us, input gradient 91 -> 80 us.
template <int BN, int RS, int MODE, bool NCS1 = false, int NW = 8, bool PP = false, bool ST = false, bool AD = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 2) void conv_halo_kernel(HaloParams p) {
  using C = HaloCfg<BN, NW>;
  static_assert(!ST || (BN == 64 && RS == 9 && NCS1 && NW == 8 && !PP && MODE != 1), "stream form: 64 -> 64 channel 3x3 tiles, no fused reduction");
  static_assert(!AD || (ST && MODE == 2), "AD = the stream form of the input gradient with a residual addend");
  static_assert(!PP || (NW == 8 && !NCS1), "ping-pong: 8-wave workgroups");
  static_assert(!(NCS1 && NW == 4), "one halo buffer: the next tile's halo cannot stream in during the K loop");
  static_assert(NW == 8 || RS % 3 == 0, "3-slot ring: the slot of a step is its tap index mod 3");
  constexpr int HTHREADS = NW * 64;
  constexpr int D = C::NWS - 1;   // weight fills run D steps ahead
  constexpr int TP = 4;         // 16-pixel tiles per wave (64 pixels)
  constexpr int TC = C::CPW / 16;   // 16-channel tiles per wave
#ifndef HALO_HPS9
#define HALO_HPS9 1   // (all of a slice's halo pieces in its first step, as the 3-tap filters do: 1-2 % slower on layers 2-4)
#endif
  constexpr int HPS = RS == 9 ? HALO_HPS9 : C::NPW_MAX;  // halo pieces a wave issues per step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wv & 3, wc = wv >> 2;   // (NW == 4: wc == 0)
  // (64-channel tiles only -- measured stand-alone, batch 256: layer 1 forward 97.8 -> 91.2 us, input gradient 94.3 -> 88.9 us;
  // the 128-channel tiles of layers 2-4 got 3-5 % SLOWER with it (their two MFMA blocks per step already fill the interval,
  // and the branch cost them registers: 12 -> 52 B of scratch), so they stay in lock step)
  const bool late = !PP && NW == 8 && BN == 64 && p.stagger != 0 && wv >= 4;
  // weight ring slot of K step g (tap t): 4 slots -> g & 3; 3 slots -> t % 3 (every slice has RS = 3 or 9 steps)
  auto slot_of = [](int g, int t) -> int { return C::NWS == 4 ? (g & 3) : (t % 3); };
  auto hbuf_of = [](int q) -> unsigned { return C::NHB == 2 ? (unsigned)(q & 1) * C::HBUF : 0u; };
  const int fr = lane & 15, fq = lane >> 4;
  const int G = gridDim.x;
  // Workgroups are PERSISTENT, and a workgroup stays on ONE channel tile: workgroup b owns channel tile b % ntn and the
  // pixel tiles b / ntn, b / ntn + G / ntn, ... (G is a multiple of ntn).  Equal b % 8 = same XCD (MI355X_MICROARCH.md,
  // dispatch) and ntn divides 8, so an XCD's L2 holds one channel tile's weights -- speed only.  One channel tile per
  // workgroup is what lets the per-channel partial sums accumulate over all of its tiles.
  const int nt = (int)blockIdx.x % p.ntn, kq = (int)blockIdx.x / p.ntn, Gk = G / p.ntn;
  const int n0 = nt * BN;
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
      const int row = (kk * NW + wv) * 8 + lrow8;
      const int pix = m0 - p.HL + row;
      const bool ok = row < p.hrows && pix >= 0 && pix < p.M;
      hoff[kk] = ok ? (unsigned)pix * pixb + lchunkb : H_OOB;
    }
  };
  auto set_wrow = [&](unsigned (&wr)[C::WPS], int n0) {
#pragma unroll
    for (int i = 0; i < C::WPS; ++i) wr[i] = (unsigned)(n0 + (i * NW + wv) * 8 + lrow8) * (unsigned)RS * pixb + lchunkb;
  };
  // halo pieces [k0, k1) of this wave, channel slice `cs`, into buffer `hbuf` (a piece wholly past the halo's last row
  // is all out of range: zero fill into rows nobody reads -- the piece count stays a compile-time constant)
  auto dma_halo = [&](unsigned hbuf, int cs, int k0, int k1) {
#pragma unroll
    for (int kk = 0; kk < C::NPW_MAX; ++kk)
      if (kk >= k0 && kk < k1) hdma16(rs_src, smem + hbuf + (kk * NW + wv) * 1024, hoff[kk], (unsigned)cs * 128u);
  };
  auto dma_w = [&](const unsigned (&wr)[C::WPS], int slot, int cs, int t) {
    const unsigned so = ((unsigned)t * (unsigned)p.Cs + (unsigned)cs * 64u) * 2u;
#pragma unroll
    for (int i = 0; i < C::WPS; ++i) hdma16(rs_w, smem + C::WBASE + slot * C::WSTAGE + (i * NW + wv) * 1024, wr[i], so);
  };

  // ---- fragment addresses
  // weights (A operand): row = channel, conflict-free b128 reads via chunk ^ (row & 7)
  const unsigned aoff0 = (unsigned)C::WBASE + (unsigned)(wc * C::CPW + fr) * 128u + (unsigned)((fq ^ (fr & 7)) << 4);
  // activations (B operand): per (pixel tile b, tap t) the halo-relative LDS byte this lane reads for k-step 0
  // (k-step 1 = the same address ^ 64); a tap outside the image reads the zero row
  // (two 16-bit addresses per register: 4 x RS of them live through the whole K loop)
  constexpr int RSP = (RS + 1) / 2;
  unsigned baddr[TP][RSP];
  auto bad = [&](int b, int t) -> unsigned { return (t & 1) ? baddr[b][t >> 1] >> 16 : baddr[b][t >> 1] & 0xFFFFu; };
  const int HW = p.H * p.W;
  // The address a tap reads when it is VALID does not depend on the tile (halo-relative); only which taps leave the image does.
  // AVALID (64-channel tiles, which have the registers): the valid addresses are built once per kernel, a tile only builds the
  // validity masks and selects (2 VALU per tap instead of ~8): the table was ~350 VALU per tile, 12 tiles per launch on layer 1.
  constexpr bool AVALID = BN == 64;
  unsigned avalid[AVALID ? TP : 1][AVALID ? RSP : 1];
  const unsigned zaddr = (unsigned)C::HCAP * 128u + (unsigned)(fq << 4);
  auto tap_addr = [&](int pl, int t) -> unsigned {
    const int r = t / 3, s = t - r * 3;
    const int dr = MODE == 0 ? r - p.ph : p.ph - r;
    const int ds = MODE == 0 ? s - p.pw : p.pw - s;
    const int R = p.HL + pl + dr * p.W + ds;
    return (unsigned)R * 128u + (unsigned)(((fq ^ R) & 7) << 4);
  };
  if (AVALID) {
#pragma unroll
    for (int b = 0; b < TP; ++b) {
#pragma unroll
      for (int t2 = 0; t2 < RSP; ++t2) avalid[b][t2] = 0u;
#pragma unroll
      for (int t = 0; t < RS; ++t) avalid[b][t >> 1] |= tap_addr(wp * 64 + b * 16 + fr, t) << (16 * (t & 1));
    }
  }
  auto set_baddr = [&](int m0) {
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const int pl = wp * 64 + b * 16 + fr;
      const int q = m0 + pl;
      // (magic-number division: the two runtime-divisor divisions per pixel tile were ~60 VALU each, four times per tile)
      const int n = (int)(((unsigned long long)(unsigned)q * p.mul_hw) >> p.sh_hw), rem = q - n * HW;
      const int h = (int)(((unsigned long long)(unsigned)rem * p.mul_w) >> p.sh_w), w = rem - h * p.W;
      bool okh[3], okw[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        okh[j] = (unsigned)(h + (MODE == 0 ? j - p.ph : p.ph - j)) < (unsigned)p.H;
        okw[j] = (unsigned)(w + (MODE == 0 ? j - p.pw : p.pw - j)) < (unsigned)p.W;
      }
      if (AVALID) {
        const unsigned zz = zaddr | (zaddr << 16);
#pragma unroll
        for (int t2 = 0; t2 < RSP; ++t2) {
          const int t0 = 2 * t2, t1 = 2 * t2 + 1;
          const bool v0 = okh[t0 / 3] && okw[t0 % 3];
          const bool v1 = t1 < RS && okh[(t1 < RS ? t1 : 0) / 3] && okw[(t1 < RS ? t1 : 0) % 3];
          const unsigned mask = (v0 ? 0xFFFFu : 0u) | (v1 ? 0xFFFF0000u : 0u);
          baddr[b][t2] = (avalid[b][t2] & mask) | (zz & ~mask);   // (an odd tap count leaves the last high half unused)
        }
      } else {
#pragma unroll
        for (int t2 = 0; t2 < RSP; ++t2) baddr[b][t2] = 0u;
#pragma unroll
        for (int t = 0; t < RS; ++t) {
          const int r = t / 3, s = t - r * 3;
          const unsigned a = (okh[r] && okw[s]) ? tap_addr(pl, t) : zaddr;
          baddr[b][t >> 1] |= a << (16 * (t & 1));
        }
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
  // (fresh: the tile's first MFMA block starts from a zero C operand -- the stream form never clears its accumulators)
  auto mma_fresh = [&](const u32x4 (&fa)[TC], const u32x4 (&fb)[TP]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[a]),
                                                            __builtin_bit_cast(bf16x8_t, fb[b]), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
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
  int mt = kq;
  int m0 = mt * HBM_;
  unsigned wrow[C::WPS];
  set_wrow(wrow, n0);
  int g0 = 0;  // K steps done so far (all tiles): step g uses weight slot g & 3
  int qs = 0;  // channel slices done so far: slice q uses halo buffer q & 1
  auto tile_fills = [&](int m0_, bool with_halo) {
    if (with_halo) {
      set_hoff(m0_);
      dma_halo(hbuf_of(qs), 0, 0, C::NPW_MAX);
    }
#pragma unroll
    for (int t = 0; t < D; ++t) dma_w(wrow, slot_of(g0 + t, t), 0, t);
  };
  if (tid < 8 * C::NHB) *reinterpret_cast<u32x4*>(smem + (tid >> 3) * C::HBUF + C::HCAP * 128 + (tid & 7) * 16) = (u32x4){0u, 0u, 0u, 0u};
  float* const sacc = reinterpret_cast<float*>(smem + C::SBASE);
  for (int i = tid; i < 4 * 2 * BN; i += HTHREADS) sacc[i] = 0.f;
  float* const scoef = reinterpret_cast<float*>(smem + C::CBASE);
  if (MODE == 1 && p.red_y)
    for (int i = tid; i < 3 * BN; i += HTHREADS) scoef[i] = p.red_coef[(i / BN) * p.Cd + n0 + (i % BN)];
  tile_fills(m0, true);

  u32x4 fa0[TC], fb0[TP], fa1[TC], fb1[TP];

  // The epilogue's operand tensors (residual addend, fused BatchNorm-backward y and mask) are read once per tile, right
  // when the tile ends: their HBM latency would be exposed once per tile.  At the start of the tile's last K slice every
  // wave touches the 64 lines it will need (its 64 pixels x its channel range = one 128-byte line each) with ONE 4-byte
  // LDS-DMA per tensor into a trash area: no registers, and the lines wait in L2.  Always three DMAs (an absent tensor
  // is replaced by one line of the weights), so the counted waits of the slice stay compile-time constants.
  const bf16_t* const pf_base[3] = {(const bf16_t*)p.addend, MODE == 1 ? (const bf16_t*)p.red_y : nullptr,
                                    MODE == 1 && p.red_mask != p.red_y ? (const bf16_t*)p.red_mask : nullptr};
  auto prefetch_epilogue_operands = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    const size_t e = (size_t)(m0 + wp * 64 + lane) * p.Cd + n0 + wc * C::CPW;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const bf16_t* g = pf_base[k] ? pf_base[k] + e : (const bf16_t*)p.wpk;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(smem + C::TRASH), 4, 0, 0);
    }
#endif
  };

  // One channel slice = RS steps, fully unrolled and branch-free, so that every wait count is a compile-time constant
  // and hipcc's LDS-read bookkeeping stays exact.  LAST = the tile's last slice: no fills past the tile's end.
  // LATE (round 3, the guide's "two waves that run the same program with one barrier per block: try a stagger"): the two
  // waves of a SIMD (w and w + 4) otherwise reach their MFMA blocks, their fragment reads, the fill issue and the barrier
  // together -- the matrix pipe serves both at once and then neither.  Waves 4-7 therefore run the step's first MFMA block
  // AFTER the step's barrier instead of before it: right behind a barrier they issue MFMAs while waves 0-3 issue fragment
  // reads, and in front of the next one they issue reads / fills while waves 0-3 finish their MFMA block.  Same
  // instructions, same barriers and wait counts, same results bit for bit.
  // (A wave-uniform branch around the one MFMA block, not two copies of the slice: duplicating the slice pushed the 128-channel
  // instantiations into scratch.  Both paths leave the same LDS reads pending at the joins, so hipcc's wait bookkeeping stays exact.)
  auto slice = [&](int cs, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const unsigned hb = hbuf_of(qs), hbn = hbuf_of(qs + 1);
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
      ld_a(fa1, slot_of(s, t), 64u);
#pragma unroll
      for (int b = 0; b < TP; ++b) fb1[b] = ld_b((bad(b, t) ^ 64u) + hb);
      // (b) fills THREE steps ahead: weights of step s + 3 go into the slot step s - 1 read (every wave is past that step's
      // barrier), the next slice's halo into the other halo buffer.  Two steps' fills stay in flight across each barrier.
      const auto n_halo = [](int tt, bool last) { return (!(last && !NCS1) && tt < RS - 1 && tt * HPS < C::NPW_MAX)
                                                      ? ((tt + 1) * HPS < C::NPW_MAX ? HPS : C::NPW_MAX - tt * HPS) : 0; };
      const auto n_fill = [&](int tt, bool last) {
        return ((last && tt + D >= RS) ? 0 : C::WPS) + n_halo(tt, last) + ((last && tt == 0) ? 3 : 0);
      };
      const bool wrap = t + D >= RS;
      if (LAST && t == 0) prefetch_epilogue_operands();   // 3 DMAs, counted in n_fill
#if !(defined(HALO_ABL) && HALO_ABL == 3)   // diagnostic 3: no fills inside the K loop
      if (!(LAST && wrap)) dma_w(wrow, slot_of(s + D, (t + D) % RS), cs + (wrap ? 1 : 0), (t + D) % RS);
      if (n_halo(t, LAST) > 0) dma_halo(hbn, NCS1 ? 0 : cs + 1, t * HPS, (t + 1) * HPS);   // (NCS1: the next TILE's halo, hoff set at the tile's start)
#endif
      // (c) k-step 0
      if (!late) mma(fa0, fb0);
      // (d) the fills of step s + 1 (issued two steps ago) have landed -- this wave's, then (barrier) everybody's; all that
      // may still be in flight are the previous step's and this step's fills.  (The step before a slice's first one is a
      // non-final slice's last step, or the tile prologue, whose fills were all waited for: a larger count is then moot.)
      // The LDS reads of (a) are back too, so the slots they read may be refilled by the next step.
      // (fills D steps ahead: the ones issued in the last D - 1 steps may still be in flight)
      const int outstanding = (D == 3 ? (t == 0 ? n_fill(RS - 1, false) : n_fill(t - 1, LAST)) : 0) + n_fill(t, LAST);
      // (a count past the table falls back to vmcnt(0): stricter, never wrong)
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
      else if (outstanding == 13) wait_vmcnt<13>();
      else if (outstanding == 14) wait_vmcnt<14>();
      else if (outstanding == 15) wait_vmcnt<15>();
      else wait_vmcnt<0>();
      wait_lds();
#if !(defined(HALO_ABL) && HALO_ABL == 5)   // diagnostic 5: no per-step barrier
      __builtin_amdgcn_s_barrier();
#endif
      if (late) mma(fa0, fb0);
      // (e) k-step 0 of the next step (none after the tile's last step)
      if (!(LAST && t == RS - 1)) {
        const int t1 = (t + 1) % RS;
        ld_a(fa0, slot_of(s + 1, t1), 0u);
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

  // ---- the ping-pong form of a slice (PP): see the kernel's header.  Hazards, with waves 4-7 one barrier behind:
  //   fills of step s+1 (issued at step s-2)  are waited for in front of B3 of step s by every wave; the first read of that slot
  //                                           is waves 0-3's phase 1 of step s+1, behind B4 (= waves 4-7's B3): everybody's landed;
  //   the slot step s+3's fill overwrites      (step s-1's) was last read in phase 3 of step s-1, and those reads are complete
  //                                           (lgkmcnt(0)) in front of that phase's barrier B3 -- waves 4-7's B3 of step s-1 is the
  //                                           barrier waves 0-3 pass last (their B4) before phase 1 of step s issues the fill.
  auto slice_pp = [&](int cs, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const unsigned hb = hbuf_of(qs), hbn = hbuf_of(qs + 1);
#pragma unroll
    for (int b = 0; b < TP; ++b)
#pragma unroll
      for (int t2 = 0; t2 < RSP; ++t2) asm volatile("" : "+v"(baddr[b][t2]));
#pragma unroll
    for (int t = 0; t < RS; ++t) {
      const int s = g0 + t;
      const auto n_halo = [](int tt, bool last) { return (!last && tt < RS - 1 && tt * HPS < C::NPW_MAX)
                                                      ? ((tt + 1) * HPS < C::NPW_MAX ? HPS : C::NPW_MAX - tt * HPS) : 0; };
      const auto n_fill = [&](int tt, bool last) {
        return ((last && tt + D >= RS) ? 0 : C::WPS) + n_halo(tt, last) + ((last && tt == 0) ? 3 : 0);
      };
      const bool wrap = t + D >= RS;
      // phase 1: k-step-0 fragments, then this step's fills
      ld_a(fa0, slot_of(s, t), 0u);
#pragma unroll
      for (int b = 0; b < TP; ++b) fb0[b] = ld_b(bad(b, t) + hb);
      if (LAST && t == 0) prefetch_epilogue_operands();   // 3 DMAs, counted in n_fill
      if (!(LAST && wrap)) dma_w(wrow, slot_of(s + D, (t + D) % RS), cs + (wrap ? 1 : 0), (t + D) % RS);
      if (n_halo(t, LAST) > 0) dma_halo(hbn, cs + 1, t * HPS, (t + 1) * HPS);
      __builtin_amdgcn_s_barrier();                       // B1
      wait_lds();
      // phase 2
      mma(fa0, fb0);
      __builtin_amdgcn_s_barrier();                       // B2
      // phase 3: k-step-1 fragments; the fills of step s + 1 have landed (all but the last two steps' fills)
      ld_a(fa0, slot_of(s, t), 64u);
#pragma unroll
      for (int b = 0; b < TP; ++b) fb0[b] = ld_b((bad(b, t) ^ 64u) + hb);
      const int outstanding = (D == 3 ? (t == 0 ? n_fill(RS - 1, false) : n_fill(t - 1, LAST)) : 0) + n_fill(t, LAST);
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
      __builtin_amdgcn_s_barrier();                       // B3
      // phase 4
      mma(fa0, fb0);
      __builtin_amdgcn_s_barrier();                       // B4
    }
    g0 += RS;
    ++qs;
  };

  bf16_t* __restrict__ dst = (bf16_t*)p.dst;
  const bf16_t* __restrict__ addend = (const bf16_t*)p.addend;
  // RACC (64-channel tiles, one row per workgroup): the per-channel sums stay in each lane's registers across ALL of the
  // workgroup's tiles and go through the 16-lane DPP reduction once, at the end -- per tile that reduction was 8 (BN = 64:
  // 16 values x 4 DPP adds x 2 + LDS read-modify-write) of a 64-channel tile's ~100 us (profiles/r02_halo_ablation.txt).
  // The 128-channel instantiations have no registers to spare (they sit at 256 with scratch): per-tile LDS accumulation stays.
  // (Forward only: with 16 more live registers the fused-reduction input gradient (MODE 1) spills 72 B per lane.)
#ifdef HALO_NO_RACC   // A/B build (make halo_noracc): per-tile LDS accumulation everywhere
  constexpr bool RACC = false;
#else
  constexpr bool RACC = BN == 64 && MODE == 0 && NW == 8;
#endif
  constexpr int TA_ = TC / 2;
  float rs1[RACC && MODE == 0 ? TC : 1][4], rs2[RACC && MODE == 0 ? TC : 1][4];
  float rq1[RACC && MODE == 1 ? TA_ : 1][8], rq2[RACC && MODE == 1 ? TA_ : 1][8];
#pragma unroll
  for (int a = 0; a < (RACC && MODE == 0 ? TC : 1); ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) rs1[a][j] = rs2[a][j] = 0.f;
#pragma unroll
  for (int a = 0; a < (RACC && MODE == 1 ? TA_ : 1); ++a)
#pragma unroll
    for (int j = 0; j < 8; ++j) rq1[a][j] = rq2[a][j] = 0.f;
  const bool racc = RACC && p.wg_rows != 0;
  bool first_tile = true;
#ifdef ECG_STAMP
  unsigned long long hs_a, hs_b, hs_c, hs_d, hs_e, hs_w0, hs_acc[4] = {0, 0, 0, 0}, hs_tiles = 0, hs_bar = 0;
  HSTAMP_AT(hs_w0);
#endif
  if constexpr (ST) {
    // ---- the stream form (see the kernel's header) ------------------------------------------------------------------
    // The address table of a tile = the tile-independent valid addresses (avalid) selected against the zero row by per-pixel
    // validity masks.  Building those masks in every lane for its four pixels (two divisions, six range checks, five mask
    // pairs each) was ~250 VALU per wave and tile -- and VALU time is exposed here (both waves of a SIMD run the same code at
    // the same time; tools/conv_bench.py with make halo_abl: 100 VALU per tile cost ~8 % of the kernel).  So the masks of a
    // tile are built ONCE per pixel by the workgroup's 512 threads (thread -> pixel tid & 255; waves 0-3 pairs 0-2, waves 4-7
    // pairs 3-4), written to a small LDS table during the previous tile's step 0, and every lane fetches the dword of each of
    // its (pixel, pair)s -- 4 ds_read_b32 + 4 v_bfi per pair -- when the pair's registers come free.
    auto produce_masks = [&](int m0_, unsigned buf) {
      const int pixel = tid & 255;
      const int q = m0_ + pixel;
      const int n = (int)(((unsigned long long)(unsigned)q * p.mul_hw) >> p.sh_hw), rem = q - n * HW;
      const int h = (int)(((unsigned long long)(unsigned)rem * p.mul_w) >> p.sh_w), w = rem - h * p.W;
      // 3x3, pad 1: tap (r, s) reads row h + r - 1 (forward) or h + 1 - r (input gradient), likewise columns
      const unsigned first_r = MODE == 0 ? 0x007u : 0x1C0u, last_r = MODE == 0 ? 0x1C0u : 0x007u;
      const unsigned first_s = MODE == 0 ? 0x049u : 0x124u, last_s = MODE == 0 ? 0x124u : 0x049u;
      unsigned v = 0x1FFu;
      v &= h == 0 ? ~first_r : ~0u;
      v &= h == p.H - 1 ? ~last_r : ~0u;
      v &= w == 0 ? ~first_s : ~0u;
      v &= w == p.W - 1 ? ~last_s : ~0u;
      const int k0 = wc * 3;   // pairs k0 .. k0 + 2 (pair 5 does not exist: its slot is never read)
      unsigned* tb = reinterpret_cast<unsigned*>(smem + C::MBASE + buf * C::MBUF) + k0 * 256 + pixel;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)v, 2 * (k0 + j), 1) & 0xFFFFu;
        const unsigned hi = (unsigned)__builtin_amdgcn_sbfe((int)v, 2 * (k0 + j) + 1, 1) & 0xFFFF0000u;   // (tap 9: bit clear)
        tb[j * 256] = lo | hi;
      }
    };
    unsigned mk[TP];   // masks of the pair being rebuilt: read in front of the step's fragment reads, applied behind its barrier
    const unsigned mlane = (unsigned)C::MBASE + (unsigned)(wp * 64 + fr) * 4u;
    auto fetch_masks = [&](int t2, unsigned buf) {
#pragma unroll
      for (int b = 0; b < TP; ++b) mk[b] = *reinterpret_cast<const unsigned*>(smem + mlane + buf * C::MBUF + t2 * 1024 + b * 64);
    };
    auto rebuild = [&](int t2) {
      const unsigned zz = zaddr | (zaddr << 16);
#pragma unroll
      for (int b = 0; b < TP; ++b) baddr[b][t2] = (avalid[b][t2] & mk[b]) | (zz & ~mk[b]);
    };
    // vector-memory operations a wave issues in step t, in this order: the weight piece(s) of step s + 3, the next tile's halo
    // piece t, one 16-byte store of the previous tile's output (steps 1-4; the first tile stores zeros to its own rows, see
    // below, so the counts are constants).  vmcnt retires in issue order: the wait of step t needs the WEIGHTS issued in
    // step t - 2 (step t + 1 reads them), so everything younger may stay in flight -- that step's halo piece and store, and
    // all of steps t - 1 and t.  A store has three and a half steps (~1.2 us) for its write acknowledgement; a halo piece
    // (HBM) has landed by the wait of step 8 at the latest (pieces are issued in steps 0-5), in front of whose barrier nobody
    // reads the next tile's halo.
    // AD (input gradient + residual addend): a quarter's 16-byte addend load is issued THREE steps ahead of the quarter
    // (loads in steps 1-4 into three rotating register sets, quarters in steps 4-7: hipcc waits for them itself, with a count
    // that is exact in this straight-line body), and their lines were pulled into L2 by one 4-byte LDS-DMA per wave in step 5 of
    // the tile that computed them (a wave's 64 pixels x its 64 bytes: one lane per line) -- a load waited for one step after its
    // issue, or served from HBM, would stall the whole in-order vmcnt queue.
    constexpr auto st_ = [](int t) { t = (t + RS) % RS; return AD ? (t >= 4 && t <= 7 ? 1 : 0) : (t >= 1 && t <= 4 ? 1 : 0); };
    constexpr auto ld_ = [](int t) { t = (t + RS) % RS; return AD && t >= 1 && t <= 4 ? 1 : 0; };
    constexpr auto pf_ = [](int t) { t = (t + RS) % RS; return AD && t == 5 ? 1 : 0; };
    constexpr auto ex_ = [=](int t) { return st_(t) + ld_(t) + pf_(t); };   // vector-memory operations of a step's VALU part
    constexpr auto hl_ = [](int t) { t = (t + RS) % RS; return t < C::NPW_MAX ? 1 : 0; };
    static_assert(C::NPW_MAX <= RS - 1 && HPS == 1, "one halo piece per step, none in the tile's last step");
    const int cw = n0 + wc * C::CPW;
    const int cl = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;
    f32x4 accp[TC][TP];
    static_assert(TC == 2, "one channel-tile pair per wave");
    // one quarter (pixel tile b) of the previous tile's epilogue: the tile-at-once epilogue's arithmetic, value for value
    // (no bias, no activation: the host dispatches the stream form only then -- every convolution in front of a BatchNorm --
    //  so the pieces are straight-line code the scheduler can place between the MFMAs of the step's last block)
    // (ONE 16-byte load per lane in the STORE layout -- the 8 consecutive channels of the pixel this lane stores -- instead of an
    //  8-byte load per channel tile in the accumulator layout: whole 64-byte runs per pixel; the row swap that builds the store
    //  layout is its own inverse, so two v_permlane16_swap take the addend back to the accumulators' lanes)
    u32x4 adv[3];
    auto load_addend = [&](int b, int m0_) {
      adv[b % 3] = *reinterpret_cast<const u32x4*>(addend + (size_t)(m0_ + wp * 64 + b * 16 + fr) * p.Cd + cw + cl);
    };
    auto prefetch_addend = [&](int m0_) {
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(addend + (size_t)(m0_ + wp * 64 + lane) * p.Cd + cw),
                                       (__attribute__((address_space(3))) void*)(smem + C::TRASH), 4, 0, 0);
#endif
    };
    auto epi_unit = [&](int b, int m0_) {
#if defined(__HIP_DEVICE_COMPILE__)
      f32x4 v0 = accp[0][b], v1 = accp[1][b];
      if constexpr (AD) {
        const u32x4 ao = adv[b % 3];
        auto ax = __builtin_amdgcn_permlane16_swap(ao[0], ao[2], false, false);   // channels 0-1 of tile 0 | of tile 1
        auto ay = __builtin_amdgcn_permlane16_swap(ao[1], ao[3], false, false);   // channels 2-3
        v0 += (f32x4){__uint_as_float(ax[0] << 16), __uint_as_float(ax[0] & 0xFFFF0000u), __uint_as_float(ay[0] << 16), __uint_as_float(ay[0] & 0xFFFF0000u)};
        v1 += (f32x4){__uint_as_float(ax[1] << 16), __uint_as_float(ax[1] & 0xFFFF0000u), __uint_as_float(ay[1] << 16), __uint_as_float(ay[1] & 0xFFFF0000u)};
      }
      if constexpr (MODE == 0) {
        // statistics: quarter b < 2 sums channel tile b over ALL four pixel tiles (the finished accumulators are all still
        // there) -- the tile's own sums first, in pixel-tile order, then into the running sums: the tile-at-once epilogue's
        // grouping, bit for bit, with no partial sums held in registers across steps
        if (b < TC) {
          f32x4 t1, t2;
#pragma unroll
          for (int bb = 0; bb < TP; ++bb) {
            const f32x4 v = accp[b][bb];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              t1[j] = bb == 0 ? v[j] : t1[j] + v[j];
              t2[j] = __builtin_fmaf(v[j], v[j], bb == 0 ? 0.f : t2[j]);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            rs1[b][j] += t1[j];
            rs2[b][j] += t2[j];
          }
        }
      }
      const unsigned x0 = pack_bf16x2(v0[0], v0[1]), y0 = pack_bf16x2(v0[2], v0[3]);
      const unsigned x1 = pack_bf16x2(v1[0], v1[1]), y1 = pack_bf16x2(v1[2], v1[3]);
      auto lo = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
      auto hi = __builtin_amdgcn_permlane16_swap(y0, y1, false, false);
#if defined(HALO_ABL) && HALO_ABL == 11   // diagnostic 11 (timing only): epilogue arithmetic kept, its store replaced by a 4-byte DMA
      asm volatile("" ::"v"(lo[0]), "v"(hi[0]), "v"(lo[1]), "v"(hi[1]));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p.wpk,
                                       (__attribute__((address_space(3))) void*)(smem + C::TRASH), 4, 0, 0);
#else
      *reinterpret_cast<u32x4*>(dst + (size_t)(m0_ + wp * 64 + b * 16 + fr) * p.Cd + cw + cl) = (u32x4){lo[0], hi[0], lo[1], hi[1]};
#endif
#endif
    };
    auto dummy_vm = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p.wpk,
                                       (__attribute__((address_space(3))) void*)(smem + C::TRASH), 4, 0, 0);
#endif
    };
    set_baddr(m0);
    produce_masks(m0, 0u);   // (pair 4 is rebuilt from the table at every tile's step 0, the first tile's too)
    set_hoff(mt + Gk < p.ntm ? (mt + Gk) * HBM_ : p.M + 4 * HBM_);
    wait_vmcnt<0>();
    wait_lds();
    __builtin_amdgcn_s_barrier();
    ld_a(fa0, slot_of(g0, 0), 0u);
#pragma unroll
    for (int b = 0; b < TP; ++b) fb0[b] = ld_b(bad(b, 0) + hbuf_of(qs));
    // The first tile has no previous tile: its four pieces run on zero accumulators (statistics + 0, exactly) and store zeros to
    // the first tile's OWN rows, which the same lanes overwrite with the real values a tile later (a thread's stores to one
    // address stay in order) -- no branch in the loop, constant vector-memory counts.
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) accp[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int m0_prev = m0;
    for (;;) {
      const bool more = mt + Gk < p.ntm;
      const int m0n = (mt + Gk) * HBM_;   // (past the last tile: what is built from it is never used for a product that is kept)
      const unsigned hb = hbuf_of(qs), hbn = hbuf_of(qs + 1);
      const unsigned mcur = (unsigned)(qs & 1);   // this tile's half of the mask table
#pragma unroll
      for (int b = 0; b < TP; ++b)
#pragma unroll
        for (int t2 = 0; t2 < RSP; ++t2) asm volatile("" : "+v"(baddr[b][t2]));
#pragma unroll
      for (int t = 0; t < RS; ++t) {
        const int s = g0 + t;
        // (a) k-step 1 of this step; in front of it the masks of the address pair this step rebuilds (the step's LDS wait covers
        // them): tap 8's pair for THIS tile in step 0 (it was still in use when the others were rebuilt), pairs 0-3 of the next tile
        if (t == 0) fetch_masks(4, mcur);
        if (t == 2 || t == 4 || t == 6 || t == 8) fetch_masks(t / 2 - 1, mcur ^ 1u);
        ld_a(fa1, slot_of(s, t), 64u);
#pragma unroll
        for (int b = 0; b < TP; ++b) fb1[b] = ld_b((bad(b, t) ^ 64u) + hb);
        // (b) this step's vector-memory operations, in the order the wait counts assume
#if !(defined(HALO_ABL) && HALO_ABL == 13)
        dma_w(wrow, slot_of(s + D, (t + D) % RS), 0, (t + D) % RS);
        if (hl_(t)) dma_halo(hbn, 0, t, t + 1);
#endif
        // Everything else a step does besides feeding the matrix pipe -- a quarter of the previous tile's epilogue (steps 1-4),
        // the next tile's validity bits (steps 0-1) and address pairs (even steps) -- is VALU work, and the two waves of a SIMD
        // should not do it at the same time (the matrix pipe would idle for both).  It sits at the END of the step: behind the
        // barrier waves 0-3 have one MFMA block left and waves 4-7 (one block late, see `late`) two, so waves 0-3 reach it
        // while their partners issue their second block, and waves 4-7 while waves 0-3 are in the next step's first block.
        // In vector-memory order the store sits between this step's halo piece and the next step's weights; the step's own
        // wait comes before it (one operation fewer in flight there).
        auto extra = [&]() {
          if (st_(t)) {
#if defined(HALO_ABL) && HALO_ABL == 10   // diagnostic 10 (timing only): the stream form without its in-loop epilogue pieces (same vm-op counts)
            dummy_vm();
#else
            epi_unit(AD ? t - 4 : t - 1, m0_prev);
#endif
          }
          if (ld_(t)) load_addend(t - 1, m0_prev);   // (behind the quarter that used this register set)
          if (pf_(t)) prefetch_addend(m0);            // this tile's own rows: read from step 1 of the next tile on
          // address table of the next tile, in place: pair t2 is free once step 2 t2 + 1 has read its taps; tap 8 (pair 4) is
          // first read in step 7, so it is rebuilt at the tile's own start.  Step 0 also builds the NEXT tile's masks (read from
          // step 2 on: a barrier and every writer's LDS wait lie between; that half of the table was last read a tile ago).
          if (t == 0) {
            rebuild(4);
            produce_masks(m0n, mcur ^ 1u);
          }
          if (t == 2) rebuild(0);
          if (t == 4) rebuild(1);
          if (t == 6) rebuild(2);
          if (t == 8) rebuild(3);
        };
        // (c) k-step 0
        if (!late) { if (t == 0) mma_fresh(fa0, fb0); else mma(fa0, fb0); }
        // (d)
        // (this step's own store / loads come behind its wait: they are not in the count)
        const int outstanding = hl_(t - 2) + ex_(t - 2) + (C::WPS + hl_(t - 1) + ex_(t - 1)) + (C::WPS + hl_(t));
        auto wait_n = [&](int n) {
          if (n == 1) wait_vmcnt<1>();
          else if (n == 2) wait_vmcnt<2>();
          else if (n == 3) wait_vmcnt<3>();
          else if (n == 4) wait_vmcnt<4>();
          else if (n == 5) wait_vmcnt<5>();
          else if (n == 6) wait_vmcnt<6>();
          else if (n == 7) wait_vmcnt<7>();
          else if (n == 8) wait_vmcnt<8>();
          else if (n == 9) wait_vmcnt<9>();
          else if (n == 10) wait_vmcnt<10>();
          else if (n == 11) wait_vmcnt<11>();
          else if (n == 12) wait_vmcnt<12>();
          else wait_vmcnt<0>();
        };
#if defined(HALO_ABL) && (HALO_ABL == 12 || HALO_ABL == 13)   // diagnostic 12 (timing only, wrong results): no per-step wait / barrier; 13: also no in-loop fills
        wait_lds();
#else
        wait_n(outstanding);
        wait_lds();
        __builtin_amdgcn_s_barrier();
#endif
        if (late) { if (t == 0) mma_fresh(fa0, fb0); else mma(fa0, fb0); }
        // (e) k-step 0 of the next step -- after step 8: of the NEXT TILE's step 0 (its halo buffer, its address table)
        {
          const int t1 = (t + 1) % RS;
          ld_a(fa0, slot_of(s + 1, t1), 0u);
          const unsigned hb1 = t + 1 < RS ? hb : hbn;
#pragma unroll
          for (int b = 0; b < TP; ++b) fb0[b] = ld_b(bad(b, t1) + hb1);
        }
        // (f) k-step 1, (g) the step's VALU work.  Both are straight-line code (no branch since the first tile runs its pieces on
        // zero accumulators), so hipcc schedules the VALU instructions -- and the next step's fragment reads and fills -- between
        // the MFMAs by itself: layer 1 forward 77.5 -> 72.5 us.  (Explicit sched_group_barrier groups of one MFMA + six VALU on top
        // of that measured the same forward and 3 % slower input gradient: not kept.)  A reordering of this region's vector-memory
        // operations cannot break the counted waits: both fills write LDS and stay in source order, the store may move among them.
        mma(fa1, fb1);
        extra();
      }
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) accp[a][b] = acc[a][b];
      m0_prev = m0;
      g0 += RS;
      ++qs;
      if (!more) break;
      mt += Gk;
      m0 = mt * HBM_;
      set_hoff(mt + Gk < p.ntm ? (mt + Gk) * HBM_ : p.M + 4 * HBM_);
    }
    // the last tile's epilogue; the ring's fills past the last tile must have landed before the workgroup's LDS is given back
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      if constexpr (AD) load_addend(b, m0_prev);
      epi_unit(b, m0_prev);
    }
    wait_vmcnt<0>();
  } else
  for (;;) {
#ifdef ECG_STAMP
    HSTAMP_AT(hs_a);
#endif
    set_baddr(m0);
    zero_acc();
#ifdef ECG_STAMP
    HSTAMP_AT(hs_b);
#endif
    if (NCS1) set_hoff(mt + Gk < p.ntm ? (mt + Gk) * HBM_ : p.M + 4 * HBM_);   // next tile's halo offsets (none left: all out of range)
    // This tile's first fills have landed.  They were issued BEFORE the previous tile's output stores, and vmcnt retires
    // in issue order: waiting for "all but the N youngest", N = the store instructions of that epilogue, leaves the
    // stores in flight (a vmcnt(0) here exposed their whole write latency once per tile).  N must not exceed what was
    // really issued -- the statistics rows are a wave-uniform runtime choice; operations hipcc adds on its own (scratch)
    // only make the wait stricter.
    constexpr int N_DATA = TC * TP / 2, N_STAT = 2 * TC;
#if defined(HALO_ABL) && HALO_ABL == 8     // diagnostic 8 (timing only, wrong results): the tile does not wait for its first weight fills
    if (first_tile) wait_vmcnt<0>();
    else wait_vmcnt<N_DATA + D * C::WPS>();
#elif defined(HALO_ABL) && HALO_ABL == 9   // diagnostic 9 (timing only): no wait at all at the tile's start
    if (first_tile) wait_vmcnt<0>();
#else
    if (first_tile) wait_vmcnt<0>();
    else if (MODE == 0 && p.stats && !p.wg_rows) wait_vmcnt<N_DATA + N_STAT>();
    else wait_vmcnt<N_DATA>();
#endif
    first_tile = false;
#ifdef ECG_STAMP
    unsigned long long hs_b2;
    HSTAMP_AT(hs_b2);
#endif
    wait_lds();
    __builtin_amdgcn_s_barrier();
#ifdef ECG_STAMP
    HSTAMP_AT(hs_c);
#endif
    using F_ = std::integral_constant<bool, false>;
    using T_ = std::integral_constant<bool, true>;
    if constexpr (PP) {
      if (wv >= 4) __builtin_amdgcn_s_barrier();          // waves 4-7: one barrier behind, for the whole K loop
      for (int cs = 0; cs + 1 < p.ncs; ++cs) slice_pp(cs, F_{});
      slice_pp(p.ncs - 1, T_{});
      if (wv < 4) __builtin_amdgcn_s_barrier();           // ... and level again: every LDS read of the tile is complete
    } else {
      ld_a(fa0, slot_of(g0, 0), 0u);
      {
        const unsigned hb = hbuf_of(qs);
#pragma unroll
        for (int b = 0; b < TP; ++b) fb0[b] = ld_b(bad(b, 0) + hb);
      }
      for (int cs = 0; cs + 1 < p.ncs; ++cs) slice(cs, F_{});
      slice(p.ncs - 1, T_{});
    }
    // every wave is past the last step's barrier, i.e. has read its last fragments: the halo buffer and the weight
    // slots the next tile starts with are free; its fills are issued from inside this tile's epilogue (below)
    // the K loop ends HERE for every accumulator (hipcc otherwise sinks the last slice's MFMAs of the second pixel half
    // below the epilogue's first phase and keeps their fragments in scratch), and nothing of the epilogue -- its operand
    // loads! -- is scheduled up into the K loop
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) asm volatile("" : "+v"(acc[a][b]));
    __builtin_amdgcn_sched_barrier(0);
#ifdef ECG_STAMP
    HSTAMP_AT(hs_d);
#endif
    const bool more = mt + Gk < p.ntm;
    const int mt_cur = mt, m0_cur = m0, n0_cur = n0;
    if (more) {
      mt += Gk;
      m0 = mt * HBM_;
    }

#if !(defined(HALO_ABL) && HALO_ABL == 4)   // diagnostic 4: no epilogue (accumulators kept alive)
    // ---- epilogue of this tile (whole tiles only: the host dispatches this kernel for M % 256 == 0, Cd % BN == 0).
    // A lane owns pixel fr x 4 consecutive channels (fq*4..) of each 16 x 16 tile; the 8-byte bf16 packs of two
    // neighbouring channel tiles are exchanged between lane rows (v_permlane16_swap), so that a lane ends up with 8
    // consecutive channels of one pixel: 16-byte stores -- and 16-byte loads of the fused BatchNorm-backward operands.
    //
    // vmcnt retires vector-memory operations IN ORDER, loads and stores alike: a load issued behind a store (or behind
    // the next tile's fills) cannot be waited for without waiting for that store's write acknowledgement (or for the
    // fills to land).  So the epilogue runs in phases over the two pixel halves h of the wave's 64 pixels:
    //     loads(0) -> compute(0) -> loads(1) -> next tile's fills -> stores(0) -> compute(1) -> stores(1)
    // every wait for loads(h) leaves only younger operations in flight, and the wait that opens the next tile ("fills
    // landed") leaves exactly the N_DATA stores.  The operand lines were pulled into L2 during the last K slice (below).
    {
      constexpr int TA = TC / 2;   // channel-tile pairs
      const bf16_t* __restrict__ red_y = MODE == 1 ? (const bf16_t*)p.red_y : nullptr;
      const bf16_t* __restrict__ red_m = MODE == 1 ? (const bf16_t*)p.red_mask : nullptr;
      const bool red = red_y != nullptr, red_sep = red && red_m != red_y;
      const int cw = n0_cur + wc * C::CPW;                                       // first channel of this wave
      const int cl = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;                      // + 32 * pair: first of the lane's 8 stored channels
      size_t prow[TP];
#pragma unroll
      for (int b = 0; b < TP; ++b) prow[b] = (size_t)(m0_cur + wp * 64 + b * 16 + fr) * p.Cd + cw;
      float s1[TC][4], s2[TC][4];          // forward statistics: fp32 values, lane's 4 + 4 channels of a tile pair
      float q1[TA][8], q2[TA][8];          // backward reduction: stored values, lane's 8 stored channels of a pair
      f32x4 bias4[TC];
#pragma unroll
      for (int a = 0; a < TC; ++a) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[a][j] = s2[a][j] = 0.f;
        bias4[a] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + cw + a * 16 + fq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ap = 0; ap < TA; ++ap)
#pragma unroll
        for (int j = 0; j < 8; ++j) q1[ap][j] = q2[ap][j] = 0.f;
      auto unpk = [](unsigned lo, unsigned hi) -> f32x4 {
        return (f32x4){__uint_as_float(lo << 16), __uint_as_float(lo & 0xFFFF0000u), __uint_as_float(hi << 16),
                       __uint_as_float(hi & 0xFFFF0000u)};
      };
      uint2 adv[2][TC][2];       // [half][channel tile][pixel tile of the half]: residual addend, pre-swap layout
      u32x4 yq[2][TA][2], mq[2][TA][2], oq[2][TA][2];
      auto loads = [&](auto h_tag) {
        constexpr int h = decltype(h_tag)::value;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int b = h * 2 + bb;
          if (addend) {
#pragma unroll
            for (int a = 0; a < TC; ++a) adv[h][a][bb] = *reinterpret_cast<const uint2*>(addend + prow[b] + a * 16 + fq * 4);
          }
          if (MODE == 1 && red) {
#pragma unroll
            for (int ap = 0; ap < TA; ++ap) {
              yq[h][ap][bb] = *reinterpret_cast<const u32x4*>(red_y + prow[b] + ap * 32 + cl);
              if (red_sep) mq[h][ap][bb] = *reinterpret_cast<const u32x4*>(red_m + prow[b] + ap * 32 + cl);
            }
          }
        }
      };
      auto compute = [&](auto h_tag) {
        constexpr int h = decltype(h_tag)::value;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int ap = 0; ap < TA; ++ap) {
          const int a = 2 * ap;
          f32x4 c_sc[2], c_sh[2], c_mu[2];
          if (MODE == 1 && red) {   // coefficients of the lane's 8 stored channels (staged in LDS at kernel start)
            const float* cf = scoef + wc * C::CPW + ap * 32 + cl;
            c_sc[0] = *reinterpret_cast<const f32x4*>(cf); c_sc[1] = *reinterpret_cast<const f32x4*>(cf + 4);
            c_sh[0] = *reinterpret_cast<const f32x4*>(cf + BN); c_sh[1] = *reinterpret_cast<const f32x4*>(cf + BN + 4);
            c_mu[0] = *reinterpret_cast<const f32x4*>(cf + 2 * BN); c_mu[1] = *reinterpret_cast<const f32x4*>(cf + 2 * BN + 4);
          }
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const int b = h * 2 + bb;
            f32x4 v0 = acc[a][b], v1 = acc[a + 1][b];
            if (p.bias) {
              v0 += bias4[a];
              v1 += bias4[a + 1];
            }
            if (addend) {
              v0 += unpk(adv[h][a][bb].x, adv[h][a][bb].y);
              v1 += unpk(adv[h][a + 1][bb].x, adv[h][a + 1][bb].y);
            }
            if (p.act == 1) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                v0[j] = fmaxf(v0[j], 0.f);
                v1[j] = fmaxf(v1[j], 0.f);
              }
            }
            if (MODE == 0) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                s1[a][j] += v0[j];
                s2[a][j] = __builtin_fmaf(v0[j], v0[j], s2[a][j]);   // (explicit: the stream form's pieces must round the same way)
                s1[a + 1][j] += v1[j];
                s2[a + 1][j] = __builtin_fmaf(v1[j], v1[j], s2[a + 1][j]);
              }
            }
            const unsigned x0 = pack_bf16x2(v0[0], v0[1]), y0 = pack_bf16x2(v0[2], v0[3]);
            const unsigned x1 = pack_bf16x2(v1[0], v1[1]), y1 = pack_bf16x2(v1[2], v1[3]);
            auto lo = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
            auto hi = __builtin_amdgcn_permlane16_swap(y0, y1, false, false);
            u32x4 o = (u32x4){lo[0], hi[0], lo[1], hi[1]};    // 8 consecutive channels (cl + 32 * ap ..) of pixel b
            if (MODE == 1 && red) {
              // the BatchNorm backward that consumes this gradient sums what is STORED (bf16), masked by its ReLU
              const f32x4 d[2] = {unpk(o[0], o[1]), unpk(o[2], o[3])};
              const u32x4 yr = yq[h][ap][bb];
              const f32x4 yv[2] = {unpk(yr[0], yr[1]), unpk(yr[2], yr[3])};
              f32x4 g[2];
              if (red_sep) {
                const u32x4 mr = mq[h][ap][bb];
                const f32x4 mv[2] = {unpk(mr[0], mr[1]), unpk(mr[2], mr[3])};
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                  for (int j = 0; j < 4; ++j) g[u][j] = mv[u][j] > 0.f ? d[u][j] : 0.f;
                o = (u32x4){pack_bf16x2(g[0][0], g[0][1]), pack_bf16x2(g[0][2], g[0][3]), pack_bf16x2(g[1][0], g[1][1]),
                            pack_bf16x2(g[1][2], g[1][3])};   // exact: the kept values are bf16 already
              } else {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                  for (int j = 0; j < 4; ++j) g[u][j] = (yv[u][j] * c_sc[u][j] + c_sh[u][j]) > 0.f ? d[u][j] : 0.f;
              }
#pragma unroll
              for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  q1[ap][u * 4 + j] += g[u][j];
                  q2[ap][u * 4 + j] += g[u][j] * (yv[u][j] - c_mu[u][j]);
                }
            }
            oq[h][ap][bb] = o;
          }
        }
#endif
      };
      auto stores = [&](auto h_tag) {
        constexpr int h = decltype(h_tag)::value;
#if defined(HALO_ABL) && HALO_ABL == 6   // diagnostic 6: no output stores (values kept alive)
#pragma unroll
        for (int ap = 0; ap < TA; ++ap)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) asm volatile("" ::"v"(oq[h][ap][bb]));
        return;
#endif
#pragma unroll
        for (int ap = 0; ap < TA; ++ap)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) *reinterpret_cast<u32x4*>(dst + prow[h * 2 + bb] + ap * 32 + cl) = oq[h][ap][bb];
      };
      using H0 = std::integral_constant<int, 0>;
      using H1 = std::integral_constant<int, 1>;
      if (addend == nullptr && !red) {
        // nothing to load: fills first (they have the whole epilogue to land), stores as soon as a half is computed
        if (more) tile_fills(m0, !NCS1);
        __builtin_amdgcn_sched_barrier(0);
        compute(H0{});
        stores(H0{});
        compute(H1{});
        stores(H1{});
      } else {
        loads(H0{});
        compute(H0{});
        loads(H1{});
        __builtin_amdgcn_sched_barrier(0);
        if (more) tile_fills(m0, !NCS1);
        __builtin_amdgcn_sched_barrier(0);
        stores(H0{});
        compute(H1{});
        stores(H1{});
      }
#if defined(HALO_ABL) && HALO_ABL == 7   // diagnostic 7: no statistics reduction
      if (false) {
#else
      if ((MODE == 0 && p.stats) || (MODE == 1 && red)) {
#endif
        // per-channel partial sums of this wave's 64 pixels: 16-lane DPP reduction, then either a row of the per-64-pixel
        // layout (what conv_igemm writes: forward only), or -- wg_rows -- added to this wave's LDS accumulators (only this
        // wave's fr == 0 lanes touch its [pixel quarter][channel] slots: plain read-modify-write, fixed order, reproducible)
        float* lrow = sacc + wp * 2 * BN + wc * C::CPW;
        if (RACC && racc) {
          if constexpr (RACC && MODE == 0) {
#pragma unroll
            for (int a = 0; a < TC; ++a)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                rs1[a][j] += s1[a][j];
                rs2[a][j] += s2[a][j];
              }
          }
          if constexpr (RACC && MODE == 1) {
#pragma unroll
            for (int ap = 0; ap < TA; ++ap)
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                rq1[ap][j] += q1[ap][j];
                rq2[ap][j] += q2[ap][j];
              }
          }
        } else if (MODE == 0) {
          float* srow = p.stats + (size_t)(mt_cur * 4 + wp) * 2 * p.Cd + cw + fq * 4;
#pragma unroll
          for (int a = 0; a < TC; ++a) {
            f32x4 r1, r2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              r1[j] = row16_sum(s1[a][j]);
              r2[j] = row16_sum(s2[a][j]);
            }
            if (fr == 0) {
              if (p.wg_rows) {
                *reinterpret_cast<f32x4*>(lrow + a * 16 + fq * 4) += r1;
                *reinterpret_cast<f32x4*>(lrow + BN + a * 16 + fq * 4) += r2;
              } else {
                *reinterpret_cast<f32x4*>(srow + a * 16) = r1;
                *reinterpret_cast<f32x4*>(srow + p.Cd + a * 16) = r2;
              }
            }
          }
        } else {
#pragma unroll
          for (int ap = 0; ap < TA; ++ap) {
            f32x4 r1[2], r2[2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              r1[j >> 2][j & 3] = row16_sum(q1[ap][j]);
              r2[j >> 2][j & 3] = row16_sum(q2[ap][j]);
            }
            if (fr == 0) {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                *reinterpret_cast<f32x4*>(lrow + ap * 32 + cl + u * 4) += r1[u];
                *reinterpret_cast<f32x4*>(lrow + BN + ap * 32 + cl + u * 4) += r2[u];
              }
            }
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
#ifdef ECG_STAMP
    HSTAMP_AT(hs_e);
    hs_acc[0] += hs_b - hs_a; hs_acc[1] += hs_c - hs_b; hs_bar += hs_c - hs_b2; hs_acc[2] += hs_d - hs_c; hs_acc[3] += hs_e - hs_d; ++hs_tiles;
#endif
    if (!more) break;
  }
#ifdef ECG_STAMP
  if (tid == 0) {
    unsigned long long hs_w1;
    HSTAMP_AT(hs_w1);
    for (int i = 0; i < 4; ++i) atomicAdd(&g_hstamp[i], hs_acc[i]);
    atomicAdd(&g_hstamp[4], hs_tiles);
    atomicAdd(&g_hstamp[5], hs_w1 - hs_w0);
    atomicAdd(&g_hstamp[6], 1ull);
    atomicAdd(&g_hstamp[7], hs_bar);
  }
#endif
  if (RACC && racc && ((MODE == 0 && p.stats) || (MODE == 1 && p.red_y))) {
    // the register sums of all tiles: one 16-lane DPP reduction, into this wave's (zero-filled) LDS slots
    float* lrow = sacc + wp * 2 * BN + wc * C::CPW;
    const int cl = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;
    if constexpr (RACC && MODE == 0) {
#pragma unroll
      for (int a = 0; a < TC; ++a) {
        f32x4 r1, r2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          r1[j] = row16_sum(rs1[a][j]);
          r2[j] = row16_sum(rs2[a][j]);
        }
        if (fr == 0) {
          *reinterpret_cast<f32x4*>(lrow + a * 16 + fq * 4) = r1;
          *reinterpret_cast<f32x4*>(lrow + BN + a * 16 + fq * 4) = r2;
        }
      }
    }
    if constexpr (RACC && MODE == 1) {
#pragma unroll
      for (int ap = 0; ap < TA_; ++ap) {
        f32x4 r1[2], r2[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          r1[j >> 2][j & 3] = row16_sum(rq1[ap][j]);
          r2[j >> 2][j & 3] = row16_sum(rq2[ap][j]);
        }
        if (fr == 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<f32x4*>(lrow + ap * 32 + cl + u * 4) = r1[u];
            *reinterpret_cast<f32x4*>(lrow + BN + ap * 32 + cl + u * 4) = r2[u];
          }
        }
      }
    }
  }
  if (p.wg_rows && (p.stats || p.red_y)) {
    // one partial row per workgroup: the four pixel quarters' accumulators, summed in fixed order
    __syncthreads();
    float* rows = p.red_y ? p.red_rows : p.stats;
    for (int i = tid; i < 2 * BN; i += HTHREADS) {
      const int which = i / BN, c = i - which * BN;
      const float v = ((sacc[0 * 2 * BN + i] + sacc[1 * 2 * BN + i]) + sacc[2 * 2 * BN + i]) + sacc[3 * 2 * BN + i];
      rows[((size_t)kq * 2 + which) * p.Cd + n0 + c] = v;
    }
  }
}

// Workgroups per channel tile of a launch (= partial rows a wg_rows launch writes per channel tile): one persistent
// workgroup per CU -- of at most ECGMM_HALO_CUS CUs -- shared out over the ntn channel tiles, never more than the pixel
// tiles.  The ONE place this is computed: the launch and ecg_conv_halo_rows() (whose caller sizes the read of a row
// buffer another launch fills) must agree for every setting of the cap.
// exact unsigned division by d for dividends < 2^31: q = (x * m) >> sh
void halo_magic_div(unsigned d, unsigned& m, unsigned& sh) {
  int l = 0;
  while ((1u << l) < d) ++l;
  m = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
  sh = 31 + l;
}
int g_halo_cu_cap = -1;  // -1: read ECGMM_HALO_CUS at first use; <= 0 after that: no cap
int halo_gk(int ntn, int ntm, int wg_per_cu = 1) {
  static int ncu[16] = {0};
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16) {
    if (ncu[dev] == 0) {
      hipDeviceProp_t prop;
      ncu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    cus = ncu[dev];
  }
  if (g_halo_cu_cap < 0) { const char* e = getenv("ECGMM_HALO_CUS"); g_halo_cu_cap = e ? atoi(e) : 0; if (g_halo_cu_cap < 0) g_halo_cu_cap = 0; }
  if (g_halo_cu_cap > 0 && cus > g_halo_cu_cap) cus = g_halo_cu_cap;
  int Gk = cus * wg_per_cu / (ntn < 1 ? 1 : ntn);
  if (Gk > ntm) Gk = ntm;
  if (Gk > 512) Gk = 512;   // (partial-row buffers hold 512 rows: ECGMM_BN_RED_ROWS)
  return Gk < 1 ? 1 : Gk;
}

template <int BN, int RS, int MODE, bool NCS1 = false, int NW = 8, bool PP = false, bool ST = false, bool AD = false>
int launch_halo(const HaloParams& p, int* rows_out, hipStream_t stream) {
  using C = HaloCfg<BN, NW>;
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)conv_halo_kernel<BN, RS, MODE, NCS1, NW, PP, ST, AD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS) != hipSuccess)
      ECG_FAIL(ECGMM_ERR_LAUNCH, "conv_halo: cannot reserve %d bytes of LDS", C::LDS);
    attr_set[dev] = true;
  }
  HaloParams q = p;
  q.ntm = p.M / HBM_;
  q.ntn = p.Cd / BN;
  // persistent: one workgroup per CU; Gk workgroups per channel tile (each walks pixel tiles k, k + Gk, ...)
  // (4-wave workgroups: two per CU)
  const int Gk = halo_gk(q.ntn, q.ntm, NW == 4 ? 2 : 1);
  *rows_out = Gk;
  hipLaunchKernelGGL((conv_halo_kernel<BN, RS, MODE, NCS1, NW, PP, ST, AD>), dim3(Gk * q.ntn), dim3(NW * 64), C::LDS, stream, q);
  ECG_CHECK_LAUNCH("conv_halo_kernel");
  return 0;
}

// 64 -> 64 channel 3x3 layers on 4-wave workgroups, two per CU (-1: read ECGMM_HALO_W4).  DEFAULT OFF.  Measured (round 3,
// batch 256): stand-alone forward 94.5 -> 90.5 us, input gradient 94.7 -> 82.5 us (same-call A/B, tools/conv_bench.py --w4),
// but the whole step 6.93 -> 7.28 ms: 512 half-size workgroups that CAN share a CU with other streams' waves lose more to the
// concurrent weight-gradient / signal-encoder kernels than the overlap of their own phases gains.  The instantiation with the
// fused BatchNorm-backward reduction (MODE 1) is never dispatched on 4 waves: it needs 432 B of scratch per lane, and with
// compiler-inserted scratch traffic inside the counted-vmcnt K loop its partial rows were NOT reproducible run to run once
// other streams shared the GPU (tools/det_check_mm.py; every scratch-free instantiation is bit-reproducible).
#ifdef ECG_STAMP
}  // namespace
extern "C" int ecgmm_hstamp_read(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_hstamp), sizeof(g_hstamp)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_hstamp), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
namespace {
#endif
int g_halo_w4 = -1;
int g_halo_pp = -1;       // ping-pong K loop on the 128-channel tiles (-1: read ECGMM_HALO_PP, default on)
int g_halo_stream = -1;   // stream form of the 64 -> 64 channel 3x3 tiles (-1: read ECGMM_HALO_STREAM, default on)
int g_halo_stagger = -1;  // waves 4-7 staggered by one MFMA block (-1: read ECGMM_HALO_STAGGER, default on)
int g_halo_enabled = -1;  // read once from ECGMM_CONV_HALO: 0 = off, 1 = where it is the faster kernel (default), 2 = wherever applicable

}  // namespace

// Runtime switch (A/B against conv_igemm from one process: tools/conv_bench.py): 0 = never take the halo kernel.
extern "C" int ecgmm_conv_halo_enable(int on) {
  g_halo_enabled = on < 0 ? 0 : on > 2 ? 2 : on;
  return 0;
}

extern "C" int ecgmm_conv_halo_pingpong(int on) {
  g_halo_pp = on != 0;
  return 0;
}
extern "C" int ecgmm_conv_halo_stream(int on) {
  g_halo_stream = on != 0;
  return 0;
}
extern "C" int ecgmm_conv_halo_stagger(int on) {
  g_halo_stagger = on != 0;
  return 0;
}
extern "C" int ecgmm_conv_halo_w4(int on) {
  g_halo_w4 = on != 0;
  return 0;
}

// Cap on the CUs (= persistent workgroups) a halo launch occupies: 0 = all (default).  Start-up value: ECGMM_HALO_CUS.
extern "C" int ecgmm_conv_halo_cus(int cus) {
  g_halo_cu_cap = cus < 0 ? 0 : cus;
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
  // Where it pays (B = 256 / 512 layer shapes, same-call A/B against conv_igemm, profiles/r02_conv_bench_layers.txt and
  // tools/halo_1d.sh): every 3x3 layer (+5...+18 %) and the 1-D encoder's 128- and 256-channel stages (+7...+20 %).  The
  // 64-channel 1-D layer (3 K steps per tile) does not amortise the per-tile prologue / epilogue of the one resident
  // workgroup: conv_igemm's two independent workgroups per CU are 8 % faster there.
  return g.R == 3 || Cs >= 128;
}

// partial rows a halo launch with ConvEpi.wg_rows writes: one per workgroup of a channel tile
static bool halo_w4(const ConvGeom& g, int Cs, int Cd) {
  if (g_halo_w4 < 0) { const char* e = getenv("ECGMM_HALO_W4"); g_halo_w4 = (e && e[0] == '1'); }
  return g_halo_w4 && g.R == 3 && Cs == 64 && Cd == 64;
}
// (rows of the fused BatchNorm-backward reduction: that instantiation always runs one 8-wave workgroup per CU)
int ecg_conv_halo_rows(int mode, const ConvGeom& g) {
  const int Cd = mode == 0 ? g.Cout : g.Cin;
  const int ntn = Cd > 64 ? Cd / 128 : 1, ntm = g.N * g.H * g.W / HBM_;
  return halo_gk(ntn, ntm, 1);
}

int ecg_conv_halo(int mode, const ConvGeom& g, const void* src, const void* wpk, void* dst, const float* bias,
                  const void* addend, float* stats, int act, ConvEpi* epi, hipStream_t stream) {
  HaloParams p;
  memset(&p, 0, sizeof(p));
  p.src = src; p.wpk = wpk; p.dst = dst; p.bias = bias; p.addend = addend; p.stats = stats; p.act = act;
  p.M = g.N * g.H * g.W; p.H = g.H; p.W = g.W; p.ph = g.pad_h; p.pw = g.pad_w;
  p.Cs = mode == 0 ? g.Cin : g.Cout;
  p.Cd = mode == 0 ? g.Cout : g.Cin;
  p.ncs = p.Cs / 64;
  p.HL = g.pad_h * g.W + g.pad_w;
  p.hrows = 256 + 2 * p.HL;
  halo_magic_div((unsigned)(g.H * g.W), p.mul_hw, p.sh_hw);
  halo_magic_div((unsigned)g.W, p.mul_w, p.sh_w);
  if (epi) {
    p.wg_rows = epi->wg_rows;
    if (epi->red_y) {
      if (stats || !epi->red_coef || !epi->red_rows || !epi->wg_rows)
        ECG_FAIL(ECGMM_ERR_SHAPE, "conv_halo: fused BatchNorm-backward reduction needs coefficients, rows, wg_rows and no forward statistics");
      p.red_y = epi->red_y; p.red_mask = epi->red_mask ? epi->red_mask : epi->red_y; p.red_coef = epi->red_coef;
      p.red_rows = epi->red_rows;
    }
  }
  if (g_halo_stagger < 0) { const char* e = getenv("ECGMM_HALO_STAGGER"); g_halo_stagger = !(e && e[0] == '0'); }
  if (g_halo_pp < 0) { const char* e = getenv("ECGMM_HALO_PP"); g_halo_pp = !(e && e[0] == '0'); }
  p.stagger = g_halo_stagger;
  int wg = 0;
  const bool wide = p.Cd > 64;
  int rc;
  // MODE 2 = input gradient WITHOUT the fused BatchNorm-backward reduction compiled in: the reduction's operand registers
  // push the BN = 128 instantiation over its 256-VGPR budget (396 B of scratch per lane) whether or not a launch uses it --
  // 13 us of a 88 us layer-2 launch (tools/halo_abl.sh, ablation 7)
  static const bool ncs1_on = [] { const char* e = getenv("ECGMM_HALO_NCS1"); return !(e && e[0] == '0'); }();
  if (halo_w4(g, p.Cs, p.Cd) && !p.red_y) {   // the 64 -> 64 channel 3x3 layers: 4-wave workgroups, two per CU (option)
    if (mode == 0) rc = launch_halo<64, 9, 0, false, 4>(p, &wg, stream);
    else rc = launch_halo<64, 9, 2, false, 4>(p, &wg, stream);
  } else if (g.R == 3 && !wide && p.ncs == 1 && ncs1_on) {   // the same on one 8-wave workgroup per CU: next tile's halo during the K loop
    // (stream form: forward with per-workgroup statistics rows or none, input gradient without addend / fused reduction)
    if (g_halo_stream < 0) { const char* e = getenv("ECGMM_HALO_STREAM"); g_halo_stream = !(e && e[0] == '0'); }
    if (mode == 0 && g_halo_stream && (!p.stats || p.wg_rows) && !p.addend && !p.bias && p.act != 1) rc = launch_halo<64, 9, 0, true, 8, false, true>(p, &wg, stream);
    else if (mode != 0 && g_halo_stream && !p.red_y && !p.addend && !p.bias && p.act != 1) rc = launch_halo<64, 9, 2, true, 8, false, true>(p, &wg, stream);
    else if (mode != 0 && g_halo_stream && !p.red_y && p.addend && !p.bias && p.act != 1) rc = launch_halo<64, 9, 2, true, 8, false, true, true>(p, &wg, stream);
    else if (mode == 0) rc = launch_halo<64, 9, 0, true>(p, &wg, stream);
    else if (p.red_y) rc = launch_halo<64, 9, 1, true>(p, &wg, stream);
    else rc = launch_halo<64, 9, 2, true>(p, &wg, stream);
  } else if (wide && g.R == 3 && g_halo_pp) {
    // 128-channel 3x3 tiles: ping-pong K loop.  Same-call A/B against the lock-step loop (tools/conv_bench.py --pp, batch 256):
    // layer 2 fwd 79-82 -> 74 us, dgrad 75-82 -> 70-75; layer 3 64.5-65.7 -> 62.5 / 62.4-63.2 -> 60.0; layer 4 58.3 -> 55-56 /
    // 56.3 -> 52.7-53.7 (-4...-10 %).  The 1x3 layers of the signal encoder (3 steps per slice) measured 2 % slower: lock step.
    if (mode == 0) rc = launch_halo<128, 9, 0, false, 8, true>(p, &wg, stream);
    else if (p.red_y) rc = launch_halo<128, 9, 1, false, 8, true>(p, &wg, stream);
    else rc = launch_halo<128, 9, 2, false, 8, true>(p, &wg, stream);
  } else if (g.R == 3) {
    if (mode == 0) rc = wide ? launch_halo<128, 9, 0>(p, &wg, stream) : launch_halo<64, 9, 0>(p, &wg, stream);
    else if (p.red_y) rc = wide ? launch_halo<128, 9, 1>(p, &wg, stream) : launch_halo<64, 9, 1>(p, &wg, stream);
    else rc = wide ? launch_halo<128, 9, 2>(p, &wg, stream) : launch_halo<64, 9, 2>(p, &wg, stream);
  } else {
    if (mode == 0) rc = wide ? launch_halo<128, 3, 0>(p, &wg, stream) : launch_halo<64, 3, 0>(p, &wg, stream);
    else if (p.red_y) rc = wide ? launch_halo<128, 3, 1>(p, &wg, stream) : launch_halo<64, 3, 1>(p, &wg, stream);
    else rc = wide ? launch_halo<128, 3, 2>(p, &wg, stream) : launch_halo<64, 3, 2>(p, &wg, stream);
  }
  if (epi) {
    epi->stats_rows = p.wg_rows ? wg : 2 * ceil_div(p.M, 128);
    epi->red_done = p.red_y != nullptr;
    epi->red_rows_n = p.red_y ? wg : 0;
  }
  return rc;
}
