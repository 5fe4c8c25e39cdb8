// Kernels of the reference's `MobileNet` factory name (S/nets.py:277-311: torchvision deeplabv3_mobilenet_v3_large) that the other families
// do not have: depthwise k x k convolutions with stride and dilation (forward with the BatchNorm batch statistics in the epilogue, data
// gradient, weight-gradient partials), the squeeze-excite scale passes and hardsigmoid.  hardswish lives in the BatchNorm apply / backward
// passes (elementwise.hip: act / mode 3).  All HBM-bound: NHWC, 16-byte channel chunks per thread, channel-lane x pixel-lane workgroups.
#include <stdlib.h>

#include "common.h"

namespace cvcs {

#define MB_DT(dt) ((dt) == CVCS_F32 || (dt) == CVCS_BF16)

static int mb_view(const char* fn, const void* p, int64_t ld, int C, int es) {
  CVCS_CHECK_ARG(p != nullptr, "%s: null tensor", fn);
  CVCS_CHECK_ARG(ld >= C && (ld * es) % 16 == 0 && ((uintptr_t)p % 16) == 0, "%s: view (ld=%lld, C=%d) must be 16-byte aligned", fn, (long long)ld, C);
  return CVCS_OK;
}
static inline unsigned mb_grid(int64_t items, int cap = 256 * 32) {
  int64_t g = (items + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}
// channel-chunk lanes of a workgroup: the largest divisor of CC that is <= cap (the widths of MobileNetV3 - 96, 160, 224, 480, 672, 960 -
// are no powers of two); PL = 256 / lanes pixel lanes, the threads beyond lanes * PL idle
static inline int chunk_lanes(int CC, int cap) {
  int best = 1;
  for (int d = 1; d <= cap && d <= CC; ++d)
    if (CC % d == 0) best = d;
  return best;
}

// ------------------------------------------------------------------------------------------------ depthwise forward / data gradient
constexpr int kDwPixPerLane = 4, kDwMaxWgs = 8192, kDwWgradWgs = 4096;
struct DwgArgs {
  const char* in; char* out; const float* w;        // w f32 [K*K][C] (tap-major)
  int64_t in_ld, out_ld;
  int B, Hi, Wi, Ho, Wo, C, K, s, d, pad, ccw;     // Hi x Wi: the map read; Ho x Wo: the map written
  float* stat_sum; float* stat_m2; float* stat_cnt;
  int xcd;       // XCD-aware pixel order (CVCS_XCD_ORDER=0 switches it off: a tuning knob)
};

// FWD:  out[b, oy, ox, c] = sum_t in[b, oy*s - pad + ky*d, ox*s - pad + kx*d, c] * w[t][c]                 (in = x, out = y)
// DGRAD: out[b, iy, ix, c] = sum_t in[b, (iy + pad - ky*d) / s, (ix + pad - kx*d) / s, c] * w[t][c]       (in = dy, out = dx; exact divisions only)
// K is a template parameter: the K*K taps are fully unrolled - every 16-byte load of a window row is issued before the first multiply (the
// rolled loop with one load in flight per thread ran at 0.7 TB/s); the filter sits in LDS (K = 5: 200 values per thread would not fit registers)
template <typename T, int K, bool DGRAD>
__global__ __launch_bounds__(256) void dwconv_kernel(DwgArgs p) {
  constexpr int ES = sizeof(T), V = 16 / ES, KK = K * K, kDwRowUnroll = K == 3 ? 3 : 1;
  __shared__ float wl[KK * 32 * V];
  __shared__ float red[2][256 * V];
  __shared__ int cntl[256];
  const int ccw = p.ccw, PL = 256 / ccw;
  const int tid = threadIdx.x, cl = tid % ccw, pl = tid / ccw;
  const bool active = pl < PL;
  const int cc = blockIdx.y * ccw + cl;
  // XCD-aware pixel order: the workgroups of one XCD (ids 8 apart) take a CONTIGUOUS span of pixels per grid stride - whole bands of image rows -
  // so the K window rows a pixel needs are hits of that XCD's own L2 (dealt round-robin, every row was fetched from HBM by several XCDs:
  // PMC 212 MB per launch for 115 MB of operands)
  const unsigned bid = p.xcd ? xcd_order(blockIdx.x, gridDim.x) : blockIdx.x;
  for (int i = tid; i < KK * ccw * V; i += 256) {
    const int t = i / (ccw * V), c = i - t * (ccw * V);
    wl[i] = p.w[(int64_t)t * p.C + blockIdx.y * ccw * V + c];
  }
  __syncthreads();
  const char* __restrict__ in = p.in + cc * 16;
  char* __restrict__ out = p.out + cc * 16;
  const int64_t items = (int64_t)p.B * p.Ho * p.Wo;
  const int64_t in_row = (int64_t)p.Wi * p.in_ld * ES, in_pix = p.in_ld * ES;
  float s1[V], s2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  int cnt = 0;
  if (active)
    for (unsigned it = bid * PL + pl; it < (unsigned)items; it += gridDim.x * PL) {      // (32-bit index arithmetic: items < 2^31, checked by the host;
      const unsigned t0 = it / (unsigned)p.Wo;                                                  //  64-bit divisions cost more than the window's loads on thin layers)
      const int ox = (int)(it - t0 * (unsigned)p.Wo);
      const int64_t b = t0 / (unsigned)p.Ho;
      const int oy = (int)(t0 - (unsigned)b * (unsigned)p.Ho);
      int ix[K];
      bool vx[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if constexpr (!DGRAD) {
          ix[k] = ox * p.s - p.pad + k * p.d;
          vx[k] = (unsigned)ix[k] < (unsigned)p.Wi;
        } else {
          const int tx = ox + p.pad - k * p.d;
          ix[k] = tx / p.s;
          vx[k] = tx >= 0 && ix[k] * p.s == tx && ix[k] < p.Wi;
        }
      }
      const char* img = in + b * p.Hi * in_row;
      float acc[V];
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = 0.f;
      // K = 3: all nine loads in flight; K = 5: one window row (five loads) per trip of a rolled loop - unrolled, the 25 loads of a pixel
      // (100 VGPRs) leave one wave per SIMD
#pragma unroll kDwRowUnroll
      for (int ky = 0; ky < K; ++ky) {
        int iy;
        bool vy;
        if constexpr (!DGRAD) {
          iy = oy * p.s - p.pad + ky * p.d;
          vy = (unsigned)iy < (unsigned)p.Hi;
        } else {
          const int ty = oy + p.pad - ky * p.d;
          iy = ty / p.s;
          vy = ty >= 0 && iy * p.s == ty && iy < p.Hi;
        }
        uint4 raw[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          raw[kx] = make_uint4(0u, 0u, 0u, 0u);
          if (vy && vx[kx]) raw[kx] = *reinterpret_cast<const uint4*>(img + iy * in_row + ix[kx] * in_pix);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          float f[V];
          Elem<T>::unpack(raw[kx], f);
          const float* wt = wl + ((ky * K + kx) * ccw + cl) * V;
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += f[k] * wt[k];
        }
      }
      const uint4 pk = Elem<T>::pack(acc);
      *reinterpret_cast<uint4*>(out + ((int64_t)it * p.out_ld) * ES) = pk;
      if constexpr (!DGRAD) {
        float r[V];
        Elem<T>::unpack(pk, r);      // the statistics of the values as stored
#pragma unroll
        for (int k = 0; k < V; ++k) { s1[k] += r[k]; s2[k] += r[k] * r[k]; }
        ++cnt;
      }
    }
  if constexpr (!DGRAD) {
    if (p.stat_sum == nullptr) return;
#pragma unroll
    for (int k = 0; k < V; ++k) { red[0][tid * V + k] = s1[k]; red[1][tid * V + k] = s2[k]; }
    cntl[tid] = active ? cnt : 0;
    __syncthreads();
    if (pl == 0) {
      int n = 0;
      for (int q = 0; q < PL; ++q) n += cntl[q * ccw + cl];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        double S = 0.0, Q = 0.0;
        for (int q = 0; q < PL; ++q) { S += (double)red[0][(q * ccw + cl) * V + k]; Q += (double)red[1][(q * ccw + cl) * V + k]; }
        double m2 = n > 0 ? Q - S * S / (double)n : 0.0;
        if (m2 < 0.0) m2 = 0.0;
        const int64_t o = (int64_t)bid * p.C + cc * V + k;
        p.stat_sum[o] = (float)S;
        p.stat_m2[o] = (float)m2;
      }
      if (blockIdx.y == 0 && cl == 0) p.stat_cnt[bid] = (float)n;
    }
  }
}

// ------------------------------------------------------------------------------------------------ depthwise weight gradient
// part[blockIdx.x][t][c] = sum over this workgroup's output pixels of dy[b, oy, ox, c] * x[b, oy*s - pad + ky*d, ox*s - pad + kx*d, c].
// One filter ROW per workgroup (ky = blockIdx.z): K x V accumulators per thread instead of K*K x 4 (K = 5: 130 VGPRs, three waves per SIMD,
// 8-byte loads - 0.3-0.4 ms per launch), 16-byte loads, K + 1 loads in flight per pixel; the pixel lanes are combined through LDS one tap at a
// time, in lane order
struct DwgWgArgs {
  const char* x; const char* dy; float* part;
  int64_t x_ld, dy_ld;
  int B, Hi, Wi, Ho, Wo, C, s, d, pad, ccw, rows;
};
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DwgWgArgs p) {
  constexpr int ES = sizeof(T), V = 16 / ES, KK = K * K;
  __shared__ float red[256 * V];
  const int ccw = p.ccw, PL = 256 / ccw;
  const int tid = threadIdx.x, cl = tid % ccw, pl = tid / ccw;
  const bool active = pl < PL;
  const int cc = blockIdx.y * ccw + cl;
  // the K workgroups of one pixel set (one per filter row) read the same x / dy: they get workgroup ids 8 apart - the SAME XCD (ids are dealt
  // round-robin over the 8 XCDs, each with its own L2), back to back - so the K passes over the operands are L2 hits (with ky on blockIdx.z
  // they ran a whole grid apart: 441 MB of HBM reads per launch for ~90 MB of operands)
  const int grp = blockIdx.x / (8 * K), rem = blockIdx.x % (8 * K);
  const int ky = rem / 8, row = grp * 8 + rem % 8;
  if (row >= p.rows) return;
  const char* __restrict__ xb = p.x + cc * 16;
  const char* __restrict__ gb = p.dy + cc * 16;
  const int64_t x_row = (int64_t)p.Wi * p.x_ld * ES, x_pix = p.x_ld * ES;
  float acc[K][V];
#pragma unroll
  for (int t = 0; t < K; ++t)
#pragma unroll
    for (int k = 0; k < V; ++k) acc[t][k] = 0.f;
  const int64_t items = (int64_t)p.B * p.Ho * p.Wo;
  if (active)
    for (unsigned it = row * PL + pl; it < (unsigned)items; it += p.rows * PL) {
      const unsigned t0 = it / (unsigned)p.Wo;
      const int ox = (int)(it - t0 * (unsigned)p.Wo);
      const int64_t b = t0 / (unsigned)p.Ho;
      const int oy = (int)(t0 - (unsigned)b * (unsigned)p.Ho);
      const int iy = oy * p.s - p.pad + ky * p.d;
      if ((unsigned)iy >= (unsigned)p.Hi) continue;
      const char* rowp = xb + (b * p.Hi + iy) * x_row;
      const uint4 graw = *reinterpret_cast<const uint4*>(gb + ((int64_t)it * p.dy_ld) * ES);
      uint4 raw[K];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int ix = ox * p.s - p.pad + kx * p.d;
        raw[kx] = make_uint4(0u, 0u, 0u, 0u);
        if ((unsigned)ix < (unsigned)p.Wi) raw[kx] = *reinterpret_cast<const uint4*>(rowp + ix * x_pix);
      }
      float g[V];
      Elem<T>::unpack(graw, g);
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        float f[V];
        Elem<T>::unpack(raw[kx], f);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[kx][k] += g[k] * f[k];
      }
    }
#pragma unroll
  for (int t = 0; t < K; ++t) {
#pragma unroll
    for (int k = 0; k < V; ++k) red[tid * V + k] = active ? acc[t][k] : 0.f;
    __syncthreads();
    if (pl == 0) {
#pragma unroll
      for (int k = 0; k < V; ++k) {
        float a = 0.f;
        for (int q = 0; q < PL; ++q) a += red[(q * ccw + cl) * V + k];
        p.part[((int64_t)row * KK + ky * K + t) * p.C + cc * V + k] = a;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ squeeze-excite
// out[b, p, c] = x[b, p, c] * s[b, c] (+ add[b, c] * add_scale): the forward scale, and its backward w.r.t. x with the gradient of the
// global average riding along (x = g, add = the gradient of the pooled vector, add_scale = 1 / HW)
template <typename T>
__global__ __launch_bounds__(256) void se_scale_kernel(const char* x, int64_t x_ld, const char* s, int64_t s_ld, const char* add, int64_t add_ld,
                                                       float add_scale, int B, int HW, int C, char* out, int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int64_t total = (int64_t)B * HW * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    const int64_t b = pix / HW;
    float f[V], sv[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + pix * x_ld * ES + cc * 16), f);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(s + b * s_ld * ES + cc * 16), sv);
#pragma unroll
    for (int k = 0; k < V; ++k) f[k] *= sv[k];
    if (add) {
      float av[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(add + b * add_ld * ES + cc * 16), av);
#pragma unroll
      for (int k = 0; k < V; ++k) f[k] += av[k] * add_scale;
    }
    *reinterpret_cast<uint4*>(out + pix * out_ld * ES + cc * 16) = Elem<T>::pack(f);
  }
}

// out[b, c] = sum over the pixels of image b of a[b, p, c] * x[b, p, c] (the gradient of the squeeze-excite scale)
template <typename T>
__global__ __launch_bounds__(256) void image_dot_kernel(const char* a, int64_t a_ld, const char* x, int64_t x_ld, int HW, int C, char* out,
                                                        int64_t out_ld, int ccw) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  __shared__ float red[256 * V];
  const int PL = 256 / ccw;
  const int cl = threadIdx.x % ccw, pl = threadIdx.x / ccw;
  const int cc = blockIdx.y * ccw + cl;
  const int b = blockIdx.x;
  float s[V];
#pragma unroll
  for (int k = 0; k < V; ++k) s[k] = 0.f;
  if (pl < PL) {
    int p = pl;
    for (; p + 3 * PL < HW; p += 4 * PL) {      // four pixels per trip, eight loads in flight
      uint4 ra[4], rx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ra[u] = *reinterpret_cast<const uint4*>(a + ((int64_t)b * HW + p + u * PL) * a_ld * ES + cc * 16);
        rx[u] = *reinterpret_cast<const uint4*>(x + ((int64_t)b * HW + p + u * PL) * x_ld * ES + cc * 16);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[V], g[V];
        Elem<T>::unpack(ra[u], g);
        Elem<T>::unpack(rx[u], f);
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += f[k] * g[k];
      }
    }
    for (; p < HW; p += PL) {
      float f[V], g[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(a + ((int64_t)b * HW + p) * a_ld * ES + cc * 16), g);
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + ((int64_t)b * HW + p) * x_ld * ES + cc * 16), f);
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] += f[k] * g[k];
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) red[threadIdx.x * V + k] = s[k];
  __syncthreads();
  if (pl == 0) {
    float r[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float acc = 0.f;
      for (int q = 0; q < PL; ++q) acc += red[(q * ccw + cl) * V + k];
      r[k] = acc;
    }
    *reinterpret_cast<uint4*>(out + (int64_t)b * out_ld * ES + cc * 16) = Elem<T>::pack(r);
  }
}

// hardsigmoid(t) = clamp(t / 6 + 1/2, 0, 1); backward: g * (|t| < 3 ? 1/6 : 0) (g == NULL: the forward)
template <typename T>
__global__ __launch_bounds__(256) void hardsigmoid_kernel(const char* t, int64_t t_ld, const char* g, int64_t g_ld, int64_t M, int C, char* out,
                                                          int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < M * CC; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t row = id / CC;
    float f[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(t + row * t_ld * ES + cc * 16), f);
    if (g) {
      float gv[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(g + row * g_ld * ES + cc * 16), gv);
#pragma unroll
      for (int k = 0; k < V; ++k) f[k] = (f[k] > -3.f && f[k] < 3.f) ? gv[k] * (1.f / 6.f) : 0.f;
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) f[k] = fminf(fmaxf(f[k] * (1.f / 6.f) + 0.5f, 0.f), 1.f);
    }
    *reinterpret_cast<uint4*>(out + row * out_ld * ES + cc * 16) = Elem<T>::pack(f);
  }
}

// out = scale[c] * y + shift[c] + res: the linear bottleneck's BatchNorm apply with the residual add (no activation: MobileNetV2 / V3 project
// layers); res NULL: the BatchNorm apply alone
template <typename T>
__global__ __launch_bounds__(256) void bn_add_kernel(const char* y, int64_t y_ld, const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const char* res, int64_t res_ld, int64_t M, int C, char* out, int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < M * CC; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    float f[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + pix * y_ld * ES + cc * 16), f);
#pragma unroll
    for (int k = 0; k < V; ++k) f[k] = f[k] * scale[cc * V + k] + shift[cc * V + k];
    if (res) {
      float r[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(res + pix * res_ld * ES + cc * 16), r);
#pragma unroll
      for (int k = 0; k < V; ++k) f[k] += r[k];
    }
    *reinterpret_cast<uint4*>(out + pix * out_ld * ES + cc * 16) = Elem<T>::pack(f);
  }
}

// rows (= gridDim.x) of a depthwise launch: ~pix_per_lane pixels per thread, at most max_wgs workgroups over the whole (rows x channel slabs) grid
static int dw_rows(int64_t M, int ccw, int slabs, int pix_per_lane, int max_wgs) {
  const int PL = 256 / ccw;
  int64_t r = (M + (int64_t)PL * pix_per_lane - 1) / ((int64_t)PL * pix_per_lane);
  int64_t cap = max_wgs / slabs;
  if (cap < 64) cap = 64;
  return (int)(r < 1 ? 1 : (r > cap ? cap : r));
}

}  // namespace cvcs

using namespace cvcs;

extern "C" int cvcs_dwconv_rows(int64_t M, int C, int dtype) {
  if (M <= 0 || C <= 0 || !MB_DT(dtype)) return 0;
  const int V = dtype == CVCS_F32 ? 4 : 8;
  if (C % V) return 0;
  const int ccw = chunk_lanes(C / V, 32);
  return dw_rows(M, ccw, C / V / ccw, kDwPixPerLane, kDwMaxWgs);
}

static int dwconv_launch(const char* fn, bool dgrad, const void* in, int64_t in_ld, int B, int Hi, int Wi, int C, const float* w, int K, int stride, int pad,
                         int dil, void* out, int64_t out_ld, int Ho, int Wo, float* stat_sum, float* stat_m2, float* stat_cnt, int dtype, void* stream) {
  CVCS_CHECK_ARG(MB_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % V == 0 && w, "%s: bad shape", fn);
  CVCS_CHECK_ARG((K == 3 || K == 5) && stride >= 1 && dil >= 1 && pad >= 0, "%s: bad geometry (kernel size 3 or 5)", fn);
  // the geometry of the FORWARD convolution: (Hi, Wi) -> (Ho, Wo) forward, (Ho, Wo) <- (Hi, Wi) for the data gradient
  const int Hx = dgrad ? Ho : Hi, Wx = dgrad ? Wo : Wi, Hy = dgrad ? Hi : Ho, Wy = dgrad ? Wi : Wo;
  CVCS_CHECK_ARG(Hy == (Hx + 2 * pad - dil * (K - 1) - 1) / stride + 1 && Wy == (Wx + 2 * pad - dil * (K - 1) - 1) / stride + 1, "%s: output size mismatch", fn);
  CVCS_CHECK_ARG(!stat_sum == !stat_m2 && !stat_sum == !stat_cnt && !(dgrad && stat_sum), "%s: statistics pointers", fn);
  CVCS_CHECK_ARG((int64_t)B * Ho * Wo < (1ll << 31) - (1 << 22), "%s: more than 2^31 output pixels", fn);
  int rc;
  if ((rc = mb_view(fn, in, in_ld, C, es)) || (rc = mb_view(fn, out, out_ld, C, es))) return rc;
  DwgArgs a;
  a.in = (const char*)in; a.out = (char*)out; a.w = w; a.in_ld = in_ld; a.out_ld = out_ld;
  a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.C = C; a.K = K; a.s = stride; a.d = dil; a.pad = pad;
  a.ccw = chunk_lanes(C / V, 32);
  a.stat_sum = stat_sum; a.stat_m2 = stat_m2; a.stat_cnt = stat_cnt;
  static const int xcd_on = getenv("CVCS_XCD_ORDER") ? atoi(getenv("CVCS_XCD_ORDER")) : 1;   // tuning knob
  a.xcd = xcd_on;
  const dim3 grid((unsigned)dw_rows((int64_t)B * Ho * Wo, a.ccw, C / V / a.ccw, kDwPixPerLane, kDwMaxWgs), (unsigned)(C / V / a.ccw));
  hipStream_t st = (hipStream_t)stream;
#define DW_L(TT, KK_)                                                                                  \
  do {                                                                                                 \
    if (dgrad) hipLaunchKernelGGL((dwconv_kernel<TT, KK_, true>), grid, dim3(256), 0, st, a);          \
    else hipLaunchKernelGGL((dwconv_kernel<TT, KK_, false>), grid, dim3(256), 0, st, a);               \
  } while (0)
  if (dtype == CVCS_F32) { if (K == 3) DW_L(float, 3); else DW_L(float, 5); }
  else { if (K == 3) DW_L(bf16_t, 3); else DW_L(bf16_t, 5); }
#undef DW_L
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_dwconv(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, int K, int stride, int pad, int dil, void* y,
                           int64_t y_ld, int Ho, int Wo, float* stat_sum, float* stat_m2, float* stat_cnt, int dtype, void* stream) {
  return dwconv_launch("cvcs_dwconv", false, x, x_ld, B, H, W, C, w, K, stride, pad, dil, y, y_ld, Ho, Wo, stat_sum, stat_m2, stat_cnt, dtype, stream);
}

extern "C" int cvcs_dwconv_dgrad(const void* dy, int64_t dy_ld, int B, int Ho, int Wo, int C, const float* w, int K, int stride, int pad, int dil, void* dx,
                                 int64_t dx_ld, int H, int W, int dtype, void* stream) {
  return dwconv_launch("cvcs_dwconv_dgrad", true, dy, dy_ld, B, Ho, Wo, C, w, K, stride, pad, dil, dx, dx_ld, H, W, nullptr, nullptr, nullptr, dtype, stream);
}

static int dw_wgrad_rows(int64_t M, int C, int K, int V, int* ccw_out) {
  const int ccw = chunk_lanes(C / V, 32);
  if (ccw_out) *ccw_out = ccw;
  return dw_rows(M, ccw, (C / V / ccw) * K, 2 * kDwPixPerLane, kDwWgradWgs);
}

extern "C" int cvcs_dwconv_wgrad_rows(int64_t M, int C, int K, int dtype) {
  if (M <= 0 || C <= 0 || !MB_DT(dtype) || (K != 3 && K != 5)) return 0;
  const int V = dtype == CVCS_F32 ? 4 : 8;
  if (C % V) return 0;
  return dw_wgrad_rows(M, C, K, V, nullptr);
}

extern "C" int cvcs_dwconv_wgrad(const void* x, int64_t x_ld, const void* dy, int64_t dy_ld, int B, int H, int W, int C, int K, int stride, int pad, int dil,
                                 int Ho, int Wo, float* part, int dtype, void* stream) {
  const char* fn = "cvcs_dwconv_wgrad";
  CVCS_CHECK_ARG(MB_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % V == 0 && part, "%s: bad shape", fn);
  CVCS_CHECK_ARG((K == 3 || K == 5) && stride >= 1 && dil >= 1 && pad >= 0, "%s: kernel size 3 or 5", fn);
  CVCS_CHECK_ARG(Ho == (H + 2 * pad - dil * (K - 1) - 1) / stride + 1 && Wo == (W + 2 * pad - dil * (K - 1) - 1) / stride + 1, "%s: output size mismatch", fn);
  CVCS_CHECK_ARG((int64_t)B * Ho * Wo < (1ll << 31) - (1 << 22), "%s: more than 2^31 output pixels", fn);
  int rc;
  if ((rc = mb_view(fn, x, x_ld, C, es)) || (rc = mb_view(fn, dy, dy_ld, C, es))) return rc;
  DwgWgArgs a;
  a.x = (const char*)x; a.dy = (const char*)dy; a.part = part; a.x_ld = x_ld; a.dy_ld = dy_ld;
  a.B = B; a.Hi = H; a.Wi = W; a.Ho = Ho; a.Wo = Wo; a.C = C; a.s = stride; a.d = dil; a.pad = pad;
  a.rows = dw_wgrad_rows((int64_t)B * Ho * Wo, C, K, V, &a.ccw);
  const dim3 grid((unsigned)((a.rows + 7) / 8 * 8 * K), (unsigned)(C / V / a.ccw));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) {
    if (K == 3) hipLaunchKernelGGL((dwconv_wgrad_kernel<float, 3>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwconv_wgrad_kernel<float, 5>), grid, dim3(256), 0, st, a);
  } else {
    if (K == 3) hipLaunchKernelGGL((dwconv_wgrad_kernel<bf16_t, 3>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwconv_wgrad_kernel<bf16_t, 5>), grid, dim3(256), 0, st, a);
  }
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_se_scale(const void* x, int64_t x_ld, const void* s, int64_t s_ld, const void* add, int64_t add_ld, float add_scale, int B, int HW, int C,
                             void* out, int64_t out_ld, int dtype, void* stream) {
  const char* fn = "cvcs_se_scale";
  CVCS_CHECK_ARG(MB_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % (16 / es) == 0, "%s: bad shape", fn);
  int rc;
  if ((rc = mb_view(fn, x, x_ld, C, es)) || (rc = mb_view(fn, s, s_ld, C, es)) || (rc = mb_view(fn, out, out_ld, C, es))) return rc;
  if (add && (rc = mb_view(fn, add, add_ld, C, es))) return rc;
  const dim3 grid(mb_grid((int64_t)B * HW * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((se_scale_kernel<float>), grid, dim3(256), 0, st, (const char*)x, x_ld, (const char*)s, s_ld, (const char*)add, add_ld, add_scale, B, HW, C,
                       (char*)out, out_ld);
  else
    hipLaunchKernelGGL((se_scale_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)x, x_ld, (const char*)s, s_ld, (const char*)add, add_ld, add_scale, B, HW, C,
                       (char*)out, out_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_image_dot(const void* a, int64_t a_ld, const void* x, int64_t x_ld, int B, int HW, int C, void* out, int64_t out_ld, int dtype,
                              void* stream) {
  const char* fn = "cvcs_image_dot";
  CVCS_CHECK_ARG(MB_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % V == 0, "%s: bad shape", fn);
  int rc;
  if ((rc = mb_view(fn, a, a_ld, C, es)) || (rc = mb_view(fn, x, x_ld, C, es)) || (rc = mb_view(fn, out, out_ld, C, es))) return rc;
  const int ccw = chunk_lanes(C / V, 32);
  const dim3 grid((unsigned)B, (unsigned)(C / V / ccw));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((image_dot_kernel<float>), grid, dim3(256), 0, st, (const char*)a, a_ld, (const char*)x, x_ld, HW, C, (char*)out, out_ld, ccw);
  else
    hipLaunchKernelGGL((image_dot_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)a, a_ld, (const char*)x, x_ld, HW, C, (char*)out, out_ld, ccw);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_hardsigmoid(const void* t, int64_t t_ld, const void* g, int64_t g_ld, int64_t M, int C, void* out, int64_t out_ld, int dtype,
                                void* stream) {
  const char* fn = "cvcs_hardsigmoid";
  CVCS_CHECK_ARG(MB_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(M > 0 && C > 0 && C % (16 / es) == 0, "%s: bad shape", fn);
  int rc;
  if ((rc = mb_view(fn, t, t_ld, C, es)) || (rc = mb_view(fn, out, out_ld, C, es))) return rc;
  if (g && (rc = mb_view(fn, g, g_ld, C, es))) return rc;
  const dim3 grid(mb_grid(M * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((hardsigmoid_kernel<float>), grid, dim3(256), 0, st, (const char*)t, t_ld, (const char*)g, g_ld, M, C, (char*)out, out_ld);
  else
    hipLaunchKernelGGL((hardsigmoid_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)t, t_ld, (const char*)g, g_ld, M, C, (char*)out, out_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_bn_add(const void* y, int64_t y_ld, const float* scale, const float* shift, const void* res, int64_t res_ld, int64_t M, int C, void* out,
                           int64_t out_ld, int dtype, void* stream) {
  const char* fn = "cvcs_bn_add";
  CVCS_CHECK_ARG(MB_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(M > 0 && C > 0 && C % (16 / es) == 0 && scale && shift, "%s: bad shape", fn);
  int rc;
  if ((rc = mb_view(fn, y, y_ld, C, es)) || (rc = mb_view(fn, out, out_ld, C, es))) return rc;
  if (res && (rc = mb_view(fn, res, res_ld, C, es))) return rc;
  const dim3 grid(mb_grid(M * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((bn_add_kernel<float>), grid, dim3(256), 0, st, (const char*)y, y_ld, scale, shift, (const char*)res, res_ld, M, C, (char*)out, out_ld);
  else
    hipLaunchKernelGGL((bn_add_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)y, y_ld, scale, shift, (const char*)res, res_ld, M, C, (char*)out, out_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}
