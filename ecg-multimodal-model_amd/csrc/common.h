// Shared device/host helpers for libecgmm_hip.so (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ecgmm.h"

typedef unsigned short bf16_t;  // raw bf16 bits; all arithmetic goes through f32
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define ECG_WAVE 64
#define ECG_TAIL_ROWS 64  // spare rows every BatchNorm partial-sum buffer carries (see fold_rows)

// ---- error plumbing (thread-local last error, C return codes) -------------------------------
void ecg_set_error(const char* fmt, ...);
#define ECG_FAIL(code, ...)       \
  do {                            \
    ecg_set_error(__VA_ARGS__);   \
    return (code);                \
  } while (0)
#define ECG_CHECK_LAUNCH(name)                                                     \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) ECG_FAIL(ECGMM_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)
#define ECG_TRY(expr)             \
  do {                            \
    int rc_ = (expr);             \
    if (rc_ != 0) return rc_;     \
  } while (0)

// ---- bf16 <-> f32 ------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16 B
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte vector <-> float[VEC]
template <typename T> __device__ __forceinline__ void unpack16(const u32x4& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
  }
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* f);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
  return v;
}
template <> __device__ __forceinline__ u32x4 pack16<bf16_t>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (unsigned)f2bf(f[2 * i]) | ((unsigned)f2bf(f[2 * i + 1]) << 16);
  return v;
}

// ---- wave / block reductions -----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15) with VALU-only DPP moves (no ds_bpermute /
// LDS traffic): quad xor-1, quad xor-2, half-row mirror, row mirror.  Every lane ends with the total.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

// bf16 MFMA with the accumulator tied in place.  With the builtin, loops whose accumulator arrays live across several
// basic blocks (the stem kernels' k-step loops) come out of register allocation with the tiles shifted between
// iterations and a v_accvgpr_read / _write pair per accumulator register per iteration; an asm operand tied "+a"
// cannot move.  hipcc does not pad hazards of asm statements: the leading s_nop 1 covers a VALU-written A/B operand,
// no accumulator may be used by two consecutive MFMAs, and mfma_drain() must follow the last one before any other
// instruction reads an accumulator.
__device__ __forceinline__ void mfma_bf16_inplace(f32x4& acc, const u32x4& a, const u32x4& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#endif
}
// first MFMA of an accumulation: C = 0 as an inline constant (no accumulator zero-fill, no zero registers)
__device__ __forceinline__ void mfma_bf16_first(f32x4& acc, const u32x4& a, const u32x4& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
#endif
}
__device__ __forceinline__ void mfma_drain() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");  // >= 18 wait states: last MFMA's D -> any reader
#endif
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
static inline size_t dtype_size(int dt) { return dt == ECGMM_BF16 ? 2 : 4; }

// ---- conv geometry shared by fwd / dgrad / wgrad ------------------------------------------------------
struct ConvGeom {
  int N, H, W, Cin;     // input  (NHWC)
  int OH, OW, Cout;     // output (NHWC)
  int R, S, stride, pad_h, pad_w;
};
static inline ConvGeom make_geom(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad_h,
                                 int pad_w) {
  ConvGeom g;
  g.N = N; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.R = R; g.S = S; g.stride = stride;
  g.pad_h = pad_h; g.pad_w = pad_w;
  g.OH = (H + 2 * pad_h - R) / stride + 1;
  g.OW = (W + 2 * pad_w - S) / stride + 1;
  return g;
}
