// On-device image transform (image half): the reference's per-sample CPU pipeline
//   transforms.Resize((224, 224)) -> ToTensor() -> Normalize([0.5]*3, [0.5]*3)      dataset.py:119-123
// applied to the decoded RGB picture (dataset.py:61; JPEG decoding itself stays on the host).
// torchvision's Resize of a PIL image is Pillow's antialiased BILINEAR resample: byte/integer work --
// two passes (horizontal, then vertical), triangle filter stretched by the down-scale factor,
// coefficients quantised to 22 fractional bits, uint8 intermediate.  Results are bit-identical to Pillow.
//
// HBM-bound: a 2500x250 RGB picture is 1.875 MB in, 0.6 MB (fp32 3x224x224) out.  One workgroup owns a band
// of output rows of one picture: it streams the source rows the band needs through LDS with 16-byte loads,
// resamples each horizontally into an LDS-resident uint8 strip, then runs the vertical pass + ToTensor +
// Normalize out of LDS and writes fp32 NCHW rows (coalesced along x).
#include <math.h>

#include <vector>

#include "ops.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct Axis {
  int ksize;
  std::vector<int> bounds;  // [out][2]: first source index, tap count
  std::vector<int> kk;      // [out][ksize]
};

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the whole axis.
Axis make_axis(int in_size, int out_size) {
  Axis ax;
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  ax.ksize = (int)ceil(support) * 2 + 1;
  ax.bounds.assign((size_t)out_size * 2, 0);
  ax.kk.assign((size_t)out_size * ax.ksize, 0);
  std::vector<double> w(ax.ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double t = (x + xmin - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      w[x] = t < 1.0 ? 1.0 - t : 0.0;
      ww += w[x];
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) w[x] /= ww;
      ax.kk[(size_t)xx * ax.ksize + x] =
          w[x] < 0 ? (int)(-0.5 + w[x] * (1 << PRECISION_BITS)) : (int)(0.5 + w[x] * (1 << PRECISION_BITS));
    }
    ax.bounds[2 * xx] = xmin;
    ax.bounds[2 * xx + 1] = xmax;
  }
  return ax;
}

struct Layout {  // int32 offsets inside the device table
  int ksh, ksv, hb, hk, vb, vk, total;
};
Layout table_layout(int H, int W, int OH, int OW) {
  Layout l;
  const double sh = (double)W / OW, sv = (double)H / OH;
  l.ksh = (int)ceil(sh < 1.0 ? 1.0 : sh) * 2 + 1;
  l.ksv = (int)ceil(sv < 1.0 ? 1.0 : sv) * 2 + 1;
  l.hb = 0;
  l.hk = l.hb + 2 * OW;
  l.vb = l.hk + OW * l.ksh;
  l.vk = l.vb + 2 * OH;
  l.total = l.vk + OH * l.ksv;
  return l;
}

// pixel (< 2^8) x coefficient (< 2^23): the 24-bit multiplier is full rate, v_mul_lo_u32 is not
__device__ __forceinline__ int mul24(unsigned char px, int k) { return __mul24((int)px, k); }

__device__ __forceinline__ unsigned clip8(int v) {
  v >>= PRECISION_BITS;
  return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

struct XfParams {
  const unsigned char* in;  // [B][H][W][3]
  float* out;               // [B][3][OH][OW]
  const int* tab;
  Layout l;
  int B, H, W, OH, OW;
  int OB;         // output rows per workgroup
  int raw_bytes;  // LDS bytes reserved for one source row's 16-byte-aligned span
  int strip_rows;
  float mean[3], std[3];
};

__global__ __launch_bounds__(256) void image_transform_kernel(XfParams p) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int OW = p.OW, OH = p.OH, W = p.W, H = p.H, ksh = p.l.ksh, ksv = p.l.ksv;
  const int* hb = p.tab + p.l.hb;
  const int* vb = p.tab + p.l.vb;
  const int* vkg = p.tab + p.l.vk;
  unsigned char* raw = lds;
  unsigned char* strip = lds + p.raw_bytes;                                  // [strip_rows][OW*3]
  int* hk = (int*)(strip + (((size_t)p.strip_rows * OW * 3 + 15) & ~(size_t)15));  // [OW][ksh]
  for (int i = tid; i < OW * ksh; i += 256) hk[i] = p.tab[p.l.hk + i];

  const int oy0 = blockIdx.x * p.OB, oy1 = min(OH, oy0 + p.OB);
  const int y_lo = vb[2 * oy0], y_hi = vb[2 * (oy1 - 1)] + vb[2 * (oy1 - 1) + 1];
  const size_t total = (size_t)p.B * H * W * 3;
  const int row_bytes = W * 3;

  for (int r = y_lo; r < y_hi; ++r) {
    // ---- source row -> LDS: the 16-byte-aligned span that covers it
    const size_t rb = ((size_t)(b * (size_t)H + r) * W) * 3;
    const uintptr_t addr = (uintptr_t)p.in + rb;
    const uintptr_t a0 = addr & ~(uintptr_t)15;
    const int off = (int)(addr - a0);
    const int nch = (off + row_bytes + 15) >> 4;
    __syncthreads();  // previous row's readers are done with `raw`
    for (int j = tid; j < nch; j += 256) {
      const uintptr_t q = a0 + (uintptr_t)j * 16;
      u32x4 v;
      if (q >= (uintptr_t)p.in && q + 16 <= (uintptr_t)p.in + total) {
        v = *(const u32x4*)q;
      } else {  // first / last chunk of the whole buffer: stay inside it
        unsigned char t[16];
        for (int i = 0; i < 16; ++i) {
          const uintptr_t a = q + i;
          t[i] = (a >= (uintptr_t)p.in && a < (uintptr_t)p.in + total) ? *(const unsigned char*)a : 0;
        }
        for (int i = 0; i < 4; ++i)
          v[i] = (unsigned)t[4 * i] | ((unsigned)t[4 * i + 1] << 8) | ((unsigned)t[4 * i + 2] << 16) |
                 ((unsigned)t[4 * i + 3] << 24);
      }
      *(u32x4*)(raw + j * 16) = v;
    }
    __syncthreads();
    // ---- horizontal pass: one lane per output column, all three channels
    unsigned char* srow = strip + (size_t)(r - y_lo) * OW * 3;
    for (int ox = tid; ox < OW; ox += 256) {
      const int xmin = hb[2 * ox], n = hb[2 * ox + 1];
      const unsigned char* px = raw + off + xmin * 3;
      const int* k = hk + ox * ksh;
      int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
      for (int x = 0; x < n; ++x) {
        const int kv = k[x];
        s0 += mul24(px[3 * x], kv);
        s1 += mul24(px[3 * x + 1], kv);
        s2 += mul24(px[3 * x + 2], kv);
      }
      srow[ox * 3] = (unsigned char)clip8(s0);
      srow[ox * 3 + 1] = (unsigned char)clip8(s1);
      srow[ox * 3 + 2] = (unsigned char)clip8(s2);
    }
  }
  __syncthreads();
  // ---- vertical pass + ToTensor + Normalize, fp32 NCHW
  const int nout = (oy1 - oy0) * 3 * OW;
  for (int idx = tid; idx < nout; idx += 256) {
    const int ox = idx % OW, c = (idx / OW) % 3, oy = oy0 + idx / (3 * OW);
    const int ymin = vb[2 * oy], n = vb[2 * oy + 1];
    const unsigned char* col = strip + (size_t)(ymin - y_lo) * OW * 3 + ox * 3 + c;
    const int* k = vkg + oy * ksv;
    int s = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < n; ++y) s += mul24(col[(size_t)y * OW * 3], k[y]);
    const float t = __fdiv_rn((float)clip8(s), 255.0f);
    p.out[(((size_t)b * 3 + c) * OH + oy) * OW + ox] = __fdiv_rn(t - p.mean[c], p.std[c]);
  }
}

// Fast variant for OW <= 256 (one lane owns one output column for the whole band): its <= KMAX horizontal
// coefficients stay in registers (zero-padded, so the tap loop is fully unrolled with immediate LDS offsets),
// and the next source row's 16-byte chunks are already in flight while the current row is resampled.
template <int KMAX>
__global__ __launch_bounds__(256) void image_transform_fast_kernel(XfParams p) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int NPF = 4;  // chunks of one source row a lane may carry (host guarantees nch <= 1024)
  const int tid = threadIdx.x, b = blockIdx.y;
  const int OW = p.OW, OH = p.OH, W = p.W, H = p.H, ksv = p.l.ksv;
  const int* vb = p.tab + p.l.vb;
  const int* vkg = p.tab + p.l.vk;
  unsigned char* raw = lds;
  unsigned char* strip = lds + p.raw_bytes;
  float* lut = (float*)(strip + (((size_t)p.strip_rows * OW * 3 + 15) & ~(size_t)15));  // [3][256] normalised values
  for (int i = tid; i < 768; i += 256) lut[i] = __fdiv_rn(__fdiv_rn((float)(i & 255), 255.0f) - p.mean[i >> 8], p.std[i >> 8]);
  int kreg[KMAX];
  int xmin3 = 0;
  const bool col = tid < OW;
  if (col) {
    xmin3 = p.tab[p.l.hb + 2 * tid] * 3;
    const int n = p.tab[p.l.hb + 2 * tid + 1];
#pragma unroll
    for (int x = 0; x < KMAX; ++x) kreg[x] = x < n ? p.tab[p.l.hk + tid * p.l.ksh + x] : 0;
  }
  const int oy0 = blockIdx.x * p.OB, oy1 = min(OH, oy0 + p.OB);
  const int y_lo = vb[2 * oy0], y_hi = vb[2 * (oy1 - 1)] + vb[2 * (oy1 - 1) + 1];
  const uintptr_t lo = (uintptr_t)p.in, hi = lo + (size_t)p.B * H * W * 3;
  const int row_bytes = W * 3;

  u32x4 pre[NPF];
  int off = 0;
  auto fetch = [&](int r) {
    const uintptr_t addr = lo + ((size_t)(b * (size_t)H + r) * W) * 3;
    const uintptr_t a0 = addr & ~(uintptr_t)15;
    off = (int)(addr - a0);
    const int nch = (off + row_bytes + 15) >> 4;
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int c = tid + 256 * j;
      if (c < nch) {
        const uintptr_t q = a0 + (uintptr_t)c * 16;
        if (q >= lo && q + 16 <= hi) {
          pre[j] = *(const u32x4*)q;
        } else {
          u32x4 v = {0, 0, 0, 0};
          for (int i = 0; i < 16; ++i)
            if (q + i >= lo && q + i < hi) v[i >> 2] |= (unsigned)(*(const unsigned char*)(q + i)) << (8 * (i & 3));
          pre[j] = v;
        }
      }
    }
    return nch;
  };
  int nch = fetch(y_lo);
  for (int r = y_lo; r < y_hi; ++r) {
    const int cur_off = off;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int c = tid + 256 * j;
      if (c < nch) *(u32x4*)(raw + c * 16) = pre[j];
    }
    __syncthreads();
    if (r + 1 < y_hi) nch = fetch(r + 1);
    if (col) {
      const unsigned char* px = raw + cur_off + xmin3;
      int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
      for (int x = 0; x < KMAX; ++x) {
        s0 += mul24(px[3 * x], kreg[x]);
        s1 += mul24(px[3 * x + 1], kreg[x]);
        s2 += mul24(px[3 * x + 2], kreg[x]);
      }
      unsigned char* d = strip + (size_t)(r - y_lo) * OW * 3 + tid * 3;
      d[0] = (unsigned char)clip8(s0);
      d[1] = (unsigned char)clip8(s1);
      d[2] = (unsigned char)clip8(s2);
    }
  }
  __syncthreads();
  // vertical pass: row index and taps are wave-uniform (scalar loads); ToTensor/Normalize through the 3x256 table
  if (col) {
    for (int oy = oy0; oy < oy1; ++oy) {
      const int ymin = vb[2 * oy], n = vb[2 * oy + 1];
      const unsigned char* cp = strip + (size_t)(ymin - y_lo) * OW * 3 + tid * 3;
      const int* k = vkg + oy * ksv;
      int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
      for (int y = 0; y < n; ++y) {
        const int kv = k[y];
        s0 += mul24(cp[0], kv);
        s1 += mul24(cp[1], kv);
        s2 += mul24(cp[2], kv);
        cp += OW * 3;
      }
      float* o = p.out + (((size_t)b * 3) * OH + oy) * OW + tid;
      o[0] = lut[clip8(s0)];
      o[(size_t)OH * OW] = lut[256 + clip8(s1)];
      o[(size_t)2 * OH * OW] = lut[512 + clip8(s2)];
    }
  }
}

template <int KMAX>
void launch_fast(const XfParams& p, size_t lds, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)image_transform_fast_kernel<KMAX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  hipLaunchKernelGGL(image_transform_fast_kernel<KMAX>, dim3(ceil_div(p.OH, p.OB), p.B), dim3(256), lds, st, p);
}

// no resize (dataset_image.py:67-70): uint8 HWC -> normalised fp32 CHW, 4 pixels (12 bytes) per lane
__global__ __launch_bounds__(256) void to_tensor_normalize_kernel(const unsigned char* in, float* out, size_t npix_per,
                                                                 int B, float m0, float m1, float m2, float s0,
                                                                 float s1, float s2) {
  const size_t total = (size_t)B * npix_per;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / npix_per, px = i - b * npix_per;
    const unsigned char* q = in + i * 3;
    float* o = out + b * 3 * npix_per + px;
    o[0] = __fdiv_rn(__fdiv_rn((float)q[0], 255.0f) - m0, s0);
    o[npix_per] = __fdiv_rn(__fdiv_rn((float)q[1], 255.0f) - m1, s1);
    o[2 * npix_per] = __fdiv_rn(__fdiv_rn((float)q[2], 255.0f) - m2, s2);
  }
}

// 4 pixels per lane: three aligned dword loads, one 16-byte store per colour plane
__global__ __launch_bounds__(256) void to_tensor_normalize_x4_kernel(const unsigned* in, float* out, size_t nquad_per,
                                                                    int B, float m0, float m1, float m2, float s0,
                                                                    float s1, float s2) {
  const size_t total = (size_t)B * nquad_per;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / nquad_per, q = i - b * nquad_per;
    const unsigned w0 = in[3 * i], w1 = in[3 * i + 1], w2 = in[3 * i + 2];
    const unsigned long long lo = w0 | ((unsigned long long)w1 << 32);
    unsigned char px[12];
#pragma unroll
    for (int k = 0; k < 8; ++k) px[k] = (unsigned char)(lo >> (8 * k));
#pragma unroll
    for (int k = 0; k < 4; ++k) px[8 + k] = (unsigned char)(w2 >> (8 * k));
    const float m[3] = {m0, m1, m2}, s[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = __fdiv_rn(__fdiv_rn((float)px[3 * k + c], 255.0f) - m[c], s[c]);
      *(f32x4*)(out + (b * 3 + c) * nquad_per * 4 + q * 4) = v;
    }
  }
}

}  // namespace

extern "C" size_t ecgmm_image_resize_tables_bytes(int H, int W, int OH, int OW) {
  if (H < 1 || W < 1 || OH < 1 || OW < 1) return 0;
  return (size_t)table_layout(H, W, OH, OW).total * sizeof(int);
}

extern "C" int ecgmm_image_resize_tables(int H, int W, int OH, int OW, void* host_tables, size_t bytes) {
  if (H < 1 || W < 1 || OH < 1 || OW < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "image_resize_tables: empty picture");
  const Layout l = table_layout(H, W, OH, OW);
  if (!host_tables || bytes < (size_t)l.total * sizeof(int)) ECG_FAIL(ECGMM_ERR_WORKSPACE, "image_resize_tables: buffer too small");
  const Axis h = make_axis(W, OW), v = make_axis(H, OH);
  if (h.ksize != l.ksh || v.ksize != l.ksv) ECG_FAIL(ECGMM_ERR_SHAPE, "image_resize_tables: internal tap-count mismatch");
  int* t = (int*)host_tables;
  memcpy(t + l.hb, h.bounds.data(), h.bounds.size() * sizeof(int));
  memcpy(t + l.hk, h.kk.data(), h.kk.size() * sizeof(int));
  memcpy(t + l.vb, v.bounds.data(), v.bounds.size() * sizeof(int));
  memcpy(t + l.vk, v.kk.data(), v.kk.size() * sizeof(int));
  return 0;
}

extern "C" int ecgmm_image_transform(const void* img, float* out, int B, int H, int W, int OH, int OW,
                                     const void* dev_tables, size_t table_bytes, const float* mean3,
                                     const float* std3, void* stream) {
  if (B < 1 || H < 1 || W < 1 || OH < 1 || OW < 1) ECG_FAIL(ECGMM_ERR_SHAPE, "image_transform: empty batch or picture");
  if (!mean3 || !std3) ECG_FAIL(ECGMM_ERR_SHAPE, "image_transform: mean/std required");
  for (int c = 0; c < 3; ++c)
    if (std3[c] == 0.0f) ECG_FAIL(ECGMM_ERR_SHAPE, "image_transform: std[%d] is zero", c);
  hipStream_t st = (hipStream_t)stream;
  if (H == OH && W == OW) {  // Pillow returns a copy when the size is unchanged
    const size_t npix = (size_t)H * W;
    if (npix % 4 == 0 && ((uintptr_t)img & 3) == 0 && ((uintptr_t)out & 15) == 0) {
      const size_t nq = npix / 4 * B;
      const int grid4 = (int)((nq + 255) / 256 < 16384 ? (nq + 255) / 256 : 16384);
      hipLaunchKernelGGL(to_tensor_normalize_x4_kernel, dim3(grid4), dim3(256), 0, st, (const unsigned*)img, out,
                         npix / 4, B, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
      ECG_CHECK_LAUNCH("image_transform(to_tensor x4)");
      return 0;
    }
    const int grid = (int)((npix * B + 255) / 256 < 65536 ? (npix * B + 255) / 256 : 65536);
    hipLaunchKernelGGL(to_tensor_normalize_kernel, dim3(grid), dim3(256), 0, st, (const unsigned char*)img, out, npix,
                       B, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    ECG_CHECK_LAUNCH("image_transform(to_tensor)");
    return 0;
  }
  const Layout l = table_layout(H, W, OH, OW);
  if (!dev_tables || table_bytes < (size_t)l.total * sizeof(int))
    ECG_FAIL(ECGMM_ERR_WORKSPACE, "image_transform: coefficient table too small (build it with ecgmm_image_resize_tables)");
  if ((size_t)B * H * W * 3 >= ((size_t)1 << 40)) ECG_FAIL(ECGMM_ERR_SHAPE, "image_transform: batch too large");
  // band height: the largest of 16/8/4/2/1 output rows (32 measured slower: fewer workgroups in flight) whose source strip fits LDS next to the row buffer
  const Axis v = make_axis(H, OH), h = make_axis(W, OW);
  int nmax = 0;
  for (int ox = 0; ox < OW; ++ox) nmax = h.bounds[2 * ox + 1] > nmax ? h.bounds[2 * ox + 1] : nmax;
  const int kmax = nmax <= 4 ? 4 : nmax <= 8 ? 8 : nmax <= 16 ? 16 : nmax <= 24 ? 24 : nmax <= 32 ? 32 : nmax <= 48 ? 48 : 0;
  const bool fast = OW <= 256 && kmax != 0 && (size_t)W * 3 + 31 <= 1024 * 16;
  XfParams p;
  memset(&p, 0, sizeof(p));
  p.raw_bytes = (int)align_up((size_t)W * 3 + 32 + (fast ? 3 * kmax : 0), 16);
  const size_t hk_bytes = fast ? 768 * sizeof(float) : (size_t)OW * l.ksh * sizeof(int);
  size_t lds = 0;
  for (int ob = 16; ob >= 1; ob >>= 1) {
    int rows = 0;
    for (int oy0 = 0; oy0 < OH; oy0 += ob) {
      const int oy1 = (oy0 + ob < OH ? oy0 + ob : OH) - 1;
      const int n = v.bounds[2 * oy1] + v.bounds[2 * oy1 + 1] - v.bounds[2 * oy0];
      rows = n > rows ? n : rows;
    }
    lds = (size_t)p.raw_bytes + align_up((size_t)rows * OW * 3, 16) + hk_bytes;
    if (lds <= 64 * 1024 || (ob == 1 && lds <= 160 * 1024)) {
      p.OB = ob;
      p.strip_rows = rows;
      break;
    }
  }
  if (p.OB == 0) ECG_FAIL(ECGMM_ERR_SHAPE, "image_transform: %dx%d -> %dx%d needs %zu B of LDS per row band", H, W, OH, OW, lds);
  p.in = (const unsigned char*)img; p.out = out; p.tab = (const int*)dev_tables; p.l = l;
  p.B = B; p.H = H; p.W = W; p.OH = OH; p.OW = OW;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.std[c] = std3[c]; }
  if (B > 65535) ECG_FAIL(ECGMM_ERR_SHAPE, "image_transform: at most 65535 pictures per call");
  if (fast) {
    switch (kmax) {
      case 4: launch_fast<4>(p, lds, st); break;
      case 8: launch_fast<8>(p, lds, st); break;
      case 16: launch_fast<16>(p, lds, st); break;
      case 24: launch_fast<24>(p, lds, st); break;
      case 32: launch_fast<32>(p, lds, st); break;
      default: launch_fast<48>(p, lds, st); break;
    }
  } else {
    (void)hipFuncSetAttribute((const void*)image_transform_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(image_transform_kernel, dim3(ceil_div(OH, p.OB), B), dim3(256), lds, st, p);
  }
  ECG_CHECK_LAUNCH("image_transform");
  return 0;
}
