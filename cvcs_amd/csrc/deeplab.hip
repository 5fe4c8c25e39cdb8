// HBM-bound kernels of the DeepLabV3+ path (BASELINE.json configs[2]; the reference wraps torchvision's DeepLabV3 at
// S/nets.py:234-311 behind the factory S/utils.py:174-195) on gfx950:
//   * bilinear resize by an integer factor, align_corners=False - F.interpolate as torchvision's DeepLab heads call it
//     (the wrapper returns d['out'], S/nets.py:248-250, which torchvision resizes to the input size) - for NHWC
//     activations and for the NCHW f32 logits, forward and backward;
//   * ASPP image pooling: per-image spatial mean and its broadcast (AdaptiveAvgPool2d(1) ... F.interpolate of a 1x1 map);
//   * the 1x1 classifier on any channel count (the U-Net head kernels of head_loss.hip are built for 64 channels):
//     NHWC activation -> NCHW f32 logits, fused argmax, backward (dx + partial dW/db rows).
// Every backward is a GATHER (each input element sums the outputs it fed, fixed order): no atomics, bitwise reproducible.
#include "common.h"

namespace cvcs {

static inline unsigned dl_grid(int64_t total, int cap = 256 * 32) {
  int64_t g = cdiv(total, 256);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// source taps of output coordinate o at integer scale s (align_corners=False): src = (o + 0.5)/s - 0.5, clamped at 0
__device__ __forceinline__ void bil_taps(int o, int s, int n, int& i0, int& i1, float& w0, float& w1) {
  float src = ((float)o + 0.5f) / (float)s - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + 1 < n ? i0 + 1 : n - 1;
  w1 = src - (float)i0;
  w0 = 1.f - w1;
}

// ------------------------------------------------------------------------------------------------ NHWC resize
template <typename T>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const char* in, int64_t in_ld, int B, int H, int W, int C, int s,
                                                        char* out, int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V, Ho = H * s, Wo = W * s;
  const int64_t total = (int64_t)B * Ho * Wo * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t op = id / CC;
    const int ox = (int)(op % Wo);
    const int64_t t = op / Wo;
    const int oy = (int)(t % Ho);
    const int64_t b = t / Ho;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bil_taps(oy, s, H, y0, y1, wy0, wy1);
    bil_taps(ox, s, W, x0, x1, wx0, wx1);
    float a[V], c[V], d[V], e[V], r[V];
    const char* base = in + (b * H * W) * in_ld * ES + cc * 16;
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)y0 * W + x0) * in_ld * ES), a);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)y0 * W + x1) * in_ld * ES), c);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)y1 * W + x0) * in_ld * ES), d);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)y1 * W + x1) * in_ld * ES), e);
#pragma unroll
    for (int k = 0; k < V; ++k) r[k] = wy0 * (wx0 * a[k] + wx1 * c[k]) + wy1 * (wx0 * d[k] + wx1 * e[k]);
    *reinterpret_cast<uint4*>(out + op * out_ld * ES + cc * 16) = Elem<T>::pack(r);
  }
}

// weight with which output coordinate o (scale s, n inputs) reads input i
__device__ __forceinline__ float bil_weight(int o, int s, int n, int i) {
  int i0, i1;
  float w0, w1;
  bil_taps(o, s, n, i0, i1, w0, w1);
  return (i0 == i ? w0 : 0.f) + (i1 == i ? w1 : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const char* g, int64_t g_ld, int B, int H, int W, int C, int s, char* gin,
                                                        int64_t gin_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V, Ho = H * s, Wo = W * s;
  const int64_t total = (int64_t)B * H * W * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t ip = id / CC;
    const int ix = (int)(ip % W);
    const int64_t t = ip / W;
    const int iy = (int)(t % H);
    const int64_t b = t / H;
    // outputs whose source lies within one input pixel of (iy, ix): o in ((i - 1 + 0.5) s - 0.5, (i + 1 + 0.5) s - 0.5)
    int oy_lo = (iy - 1) * s + s / 2 - 1, oy_hi = (iy + 1) * s + (s + 1) / 2;
    int ox_lo = (ix - 1) * s + s / 2 - 1, ox_hi = (ix + 1) * s + (s + 1) / 2;
    oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi; ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const float wy = bil_weight(oy, s, H, iy);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const float w = wy * bil_weight(ox, s, W, ix);
        if (w == 0.f) continue;
        float f[V];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(g + (((b * Ho + oy) * Wo + ox) * g_ld) * ES + cc * 16), f);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += w * f[k];
      }
    }
    *reinterpret_cast<uint4*>(gin + ip * gin_ld * ES + cc * 16) = Elem<T>::pack(acc);
  }
}

// ------------------------------------------------------------------------------------------------ NCHW f32 resize (logits)
// Both kernels were VALU-instruction-bound (a float division and two integer divisions per output element; ~100 taps with a division each per
// input element in the backward): four outputs per thread with one 16-byte store, and tap weights from a multiplication by 1/s.
__device__ __forceinline__ void bil_taps_mul(int o, float inv_s, int n, int& i0, int& i1, float& w0, float& w1) {
  float src = ((float)o + 0.5f) * inv_s - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + 1 < n ? i0 + 1 : n - 1;
  w1 = src - (float)i0;
  w0 = 1.f - w1;
}
__device__ __forceinline__ float bil_weight_mul(int o, float inv_s, int n, int i) {
  int i0, i1;
  float w0, w1;
  bil_taps_mul(o, inv_s, n, i0, i1, w0, w1);
  return (i0 == i ? w0 : 0.f) + (i1 == i ? w1 : 0.f);
}

// s % 4 == 0: a thread writes 4 consecutive outputs of a row
__global__ __launch_bounds__(256) void resize_nchw_fwd_kernel(const float* __restrict__ in, int64_t planes, int H, int W, int s,
                                                             float* __restrict__ out) {
  const int Ho = H * s, Wo = W * s, Wq = Wo / 4;
  const float inv_s = 1.f / (float)s;
  const int64_t total = planes * Ho * Wq;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int ox0 = (int)(id % Wq) * 4;
    const int64_t t = id / Wq;
    const int oy = (int)(t % Ho);
    const int64_t pl = t / Ho;
    int y0, y1;
    float wy0, wy1;
    bil_taps_mul(oy, inv_s, H, y0, y1, wy0, wy1);
    const float* r0 = in + pl * H * W + (int64_t)y0 * W;
    const float* r1 = in + pl * H * W + (int64_t)y1 * W;
    float4 o;
    float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x0, x1;
      float wx0, wx1;
      bil_taps_mul(ox0 + j, inv_s, W, x0, x1, wx0, wx1);
      ov[j] = wy0 * (wx0 * r0[x0] + wx1 * r0[x1]) + wy1 * (wx0 * r1[x0] + wx1 * r1[x1]);
    }
    *reinterpret_cast<float4*>(out + (pl * Ho + oy) * Wo + ox0) = o;
  }
}

// any s: one output per thread
__global__ __launch_bounds__(256) void resize_nchw_fwd1_kernel(const float* __restrict__ in, int64_t planes, int H, int W, int s,
                                                              float* __restrict__ out) {
  const int Ho = H * s, Wo = W * s;
  const int64_t total = planes * Ho * Wo;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int ox = (int)(id % Wo);
    const int64_t t = id / Wo;
    const int oy = (int)(t % Ho);
    const int64_t pl = t / Ho;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bil_taps(oy, s, H, y0, y1, wy0, wy1);
    bil_taps(ox, s, W, x0, x1, wx0, wx1);
    const float* p = in + pl * H * W;
    out[id] = wy0 * (wx0 * p[y0 * W + x0] + wx1 * p[y0 * W + x1]) + wy1 * (wx0 * p[y1 * W + x0] + wx1 * p[y1 * W + x1]);
  }
}

template <bool MUL>
__global__ __launch_bounds__(256) void resize_nchw_bwd_kernel(const float* __restrict__ g, int64_t planes, int H, int W, int s,
                                                             float* __restrict__ gin) {
  const int Ho = H * s, Wo = W * s;
  const float inv_s = 1.f / (float)s;
  const int64_t total = planes * H * W;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int ix = (int)(id % W);
    const int64_t t = id / W;
    const int iy = (int)(t % H);
    const int64_t pl = t / H;
    int oy_lo = (iy - 1) * s + s / 2 - 1, oy_hi = (iy + 1) * s + (s + 1) / 2;
    int ox_lo = (ix - 1) * s + s / 2 - 1, ox_hi = (ix + 1) * s + (s + 1) / 2;
    oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi; ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
    const float* p = g + pl * Ho * Wo;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const float wy = MUL ? bil_weight_mul(oy, inv_s, H, iy) : bil_weight(oy, s, H, iy);
      if (wy == 0.f) continue;
      const float* row = p + (int64_t)oy * Wo;
      float racc = 0.f;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) racc += (MUL ? bil_weight_mul(ox, inv_s, W, ix) : bil_weight(ox, s, W, ix)) * row[ox];
      acc += wy * racc;
    }
    gin[id] = acc;
  }
}

// The same gradient, one workgroup per (plane, low-resolution row): at scale 16 the per-element kernel above makes every thread walk a
// 34 x 34 window with a stride of 64 bytes between neighbouring lanes (2.7 ms for the [32,16,512,512] logit gradient of MobileNet / DeepLabV3).
// Here every high-resolution row of the window is staged ONCE into LDS with coalesced loads; a thread owns (ix, j) pairs - column j of ix's
// window - and accumulates wy * wx * g over the rows; the JW partial sums of an ix are added in a fixed order at the end.
constexpr int kRszPairs = 8;
template <bool MUL>
__global__ __launch_bounds__(256) void resize_nchw_bwd_rows_kernel(const float* __restrict__ g, int H, int W, int s, float* __restrict__ gin) {
  extern __shared__ __align__(16) float rsz_sm[];
  const int Ho = H * s, Wo = W * s, JW = 2 * s + 2, NPAIR = W * JW;
  float* row = rsz_sm;
  float* part = rsz_sm + Wo;
  const float inv_s = 1.f / (float)s;
  const int iy = blockIdx.x % H;
  const int64_t pl = blockIdx.x / H;
  const int tid = threadIdx.x;
  int ox[kRszPairs];
  float wx[kRszPairs], acc[kRszPairs];
#pragma unroll
  for (int k = 0; k < kRszPairs; ++k) {
    const int pr = tid + k * 256;
    acc[k] = 0.f; wx[k] = 0.f; ox[k] = 0;
    if (pr < NPAIR) {
      const int ix = pr / JW, j = pr - ix * JW;
      const int o = (ix - 1) * s + s / 2 - 1 + j;
      if (o >= 0 && o < Wo) {
        ox[k] = o;
        wx[k] = MUL ? bil_weight_mul(o, inv_s, W, ix) : bil_weight(o, s, W, ix);
      }
    }
  }
  int oy_lo = (iy - 1) * s + s / 2 - 1, oy_hi = (iy + 1) * s + (s + 1) / 2;
  oy_lo = oy_lo < 0 ? 0 : oy_lo;
  oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi;
  const float* p = g + pl * Ho * Wo;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float wy = MUL ? bil_weight_mul(oy, inv_s, H, iy) : bil_weight(oy, s, H, iy);
    if (wy == 0.f) continue;      // (uniform over the workgroup)
    for (int i = tid * 4; i < Wo; i += 1024) *reinterpret_cast<float4*>(row + i) = *reinterpret_cast<const float4*>(p + (int64_t)oy * Wo + i);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRszPairs; ++k) acc[k] += wy * (wx[k] * row[ox[k]]);
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < kRszPairs; ++k)
    if (tid + k * 256 < NPAIR) part[tid + k * 256] = acc[k];
  __syncthreads();
  if (tid < W) {
    float a = 0.f;
    for (int j = 0; j < JW; ++j) a += part[tid * JW + j];
    gin[(pl * H + iy) * W + tid] = a;
  }
}

// ------------------------------------------------------------------------------------------------ image pooling
// out[b][c] = scale * sum over the HW pixels of image b; one workgroup per (image, group of ccw 16-byte chunks): ccw channel lanes x
// 256 / ccw pixel lanes, fixed order.  (ccw = 32 where it divides: a 2048-channel ASPP input is then 8 x B workgroups of 8 pixel lanes
// instead of B workgroups walking all pixels one by one - 1.0 -> 0.15 ms on DeepLabV3-ResNet101.)
template <typename T>
__global__ __launch_bounds__(256) void image_sum_kernel(const char* x, int64_t ld, int HW, int C, float scale, char* out,
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
  // four pixels per trip, their loads issued together (B x slabs workgroups only: one load in flight per thread left a 64 x 64 map at 60 us)
  int p = pl;
  for (; p + 3 * PL < HW; p += 4 * PL) {
    uint4 raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(x + ((int64_t)b * HW + p + u * PL) * ld * ES + cc * 16);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[V];
      Elem<T>::unpack(raw[u], f);
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] += f[k];
    }
  }
  for (; p < HW; p += PL) {
    float f[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + ((int64_t)b * HW + p) * ld * ES + cc * 16), f);
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] += f[k];
  }
#pragma unroll
  for (int k = 0; k < V; ++k) red[threadIdx.x * V + k] = s[k];
  __syncthreads();
  if (pl == 0) {
    float r[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float a = 0.f;
      for (int q = 0; q < PL; ++q) a += red[(q * ccw + cl) * V + k];
      r[k] = a * scale;
    }
    *reinterpret_cast<uint4*>(out + (int64_t)b * out_ld * ES + cc * 16) = Elem<T>::pack(r);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void image_broadcast_kernel(const char* v, int64_t v_ld, int B, int HW, int C, float scale,
                                                             char* out, int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int64_t total = (int64_t)B * HW * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    const int64_t b = pix / HW;
    float f[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(v + b * v_ld * ES + cc * 16), f);
#pragma unroll
    for (int k = 0; k < V; ++k) f[k] *= scale;
    *reinterpret_cast<uint4*>(out + pix * out_ld * ES + cc * 16) = Elem<T>::pack(f);
  }
}

// ------------------------------------------------------------------------------------------------ 1x1 classifier, any C
constexpr int kLinMaxNC = 32;
// forward / argmax: one pixel per thread, the NC x C weights in LDS, the pixel row streamed 16 bytes at a time
template <typename T, bool ARGMAX>
__global__ __launch_bounds__(256) void linear_head_kernel(const char* x, int64_t x_ld, int64_t P, int64_t HW, int C,
                                                         const float* __restrict__ w, const float* __restrict__ bias, int NC,
                                                         float* logits, uint8_t* labels) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  extern __shared__ float sw[];   // [NC][C] + [NC]
  for (int i = threadIdx.x; i < NC * C; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < NC; i += 256) sw[NC * C + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  for (int64_t pidx = (int64_t)blockIdx.x * 256 + threadIdx.x; pidx < P; pidx += (int64_t)gridDim.x * 256) {
    float acc[kLinMaxNC];
#pragma unroll
    for (int c = 0; c < kLinMaxNC; ++c) acc[c] = c < NC ? sw[NC * C + c] : 0.f;
    const char* row = x + pidx * x_ld * ES;
    for (int k0 = 0; k0 < C; k0 += V) {
      float f[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(row + k0 * ES), f);
#pragma unroll
      for (int c = 0; c < kLinMaxNC; ++c)
        if (c < NC) {
#pragma unroll
          for (int k = 0; k < V; ++k) acc[c] += f[k] * sw[c * C + k0 + k];
        }
    }
    if constexpr (ARGMAX) {
      float best = acc[0];
      int arg = 0;
#pragma unroll
      for (int c = 1; c < kLinMaxNC; ++c)
        if (c < NC && acc[c] > best) { best = acc[c]; arg = c; }   // strict '>': the FIRST maximum
      labels[pidx] = (uint8_t)arg;
    } else {
      const int64_t b = pidx / HW, hw = pidx - b * HW;
#pragma unroll
      for (int c = 0; c < kLinMaxNC; ++c)
        if (c < NC) logits[(b * NC + c) * HW + hw] = acc[c];
    }
  }
}

// backward: dx[p][k] = sum_c dl[p][c] w[c][k]; partial dW[c][k] = sum_p dl[p][c] x[p][k], db[c] = sum_p dl[p][c] per
// workgroup row (reduced by cvcs_colsum_finalize).  Workgroup: strips of 64 pixels staged in LDS (x as f32, dl);
// thread t owns channel k = t (+256 j): its dW column of NC accumulators stays in registers over the whole strip loop.
template <typename T>
__global__ __launch_bounds__(256) void linear_head_bwd_kernel(const char* x, int64_t x_ld, const float* __restrict__ dl, int64_t P,
                                                             int64_t HW, int C, const float* __restrict__ w, int NC, char* dx,
                                                             int64_t dx_ld, float* part, int TP /* pixels per strip */) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  extern __shared__ float sm[];                // sx [TP][C] | sd [TP][NCP] | sw [NC][C]
  const int NCP = NC + 1;
  float* sx = sm;
  float* sd = sx + TP * C;
  float* sw = sd + TP * NCP;
  const int tid = threadIdx.x;
  for (int i = tid; i < NC * C; i += 256) sw[i] = w[i];
  const int KPT = (C + 255) / 256;             // channels per thread (C <= 1024)
  float acc[4][kLinMaxNC];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < kLinMaxNC; ++c) acc[j][c] = 0.f;
  float accb = 0.f;                            // db: thread c < NC sums its class
  for (int64_t p0 = (int64_t)blockIdx.x * TP; p0 < P; p0 += (int64_t)gridDim.x * TP) {
    __syncthreads();
    for (int id = tid; id < TP * (C / V); id += 256) {
      const int r = id / (C / V), cv = id - r * (C / V);
      float f[V];
      if (p0 + r < P) Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + (p0 + r) * x_ld * ES + cv * 16), f);
      else
#pragma unroll
        for (int q = 0; q < V; ++q) f[q] = 0.f;
#pragma unroll
      for (int q = 0; q < V; ++q) sx[r * C + cv * V + q] = f[q];
    }
    for (int id = tid; id < TP * NC; id += 256) {
      const int c = id / TP, r = id - c * TP;
      float v = 0.f;
      if (p0 + r < P) {
        const int64_t pidx = p0 + r, b = pidx / HW, hw = pidx - b * HW;
        v = dl[(b * NC + c) * HW + hw];
      }
      sd[r * NCP + c] = v;
    }
    __syncthreads();
    if (tid < NC)
      for (int r = 0; r < TP; ++r) accb += sd[r * NCP + tid];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = tid + 256 * j;
      if (j < KPT && k < C) {
        for (int r = 0; r < TP; ++r) {
          const float xv = sx[r * C + k];
#pragma unroll
          for (int c = 0; c < kLinMaxNC; ++c)
            if (c < NC) acc[j][c] += sd[r * NCP + c] * xv;
        }
      }
    }
    // dx of the strip: thread -> (pixel r = id / (C/V), chunk): V channels
    for (int id = tid; id < TP * (C / V); id += 256) {
      const int r = id / (C / V), cv = id - r * (C / V);
      if (p0 + r >= P) continue;
      float o[V];
#pragma unroll
      for (int q = 0; q < V; ++q) o[q] = 0.f;
      for (int c = 0; c < NC; ++c) {
        const float d = sd[r * NCP + c];
#pragma unroll
        for (int q = 0; q < V; ++q) o[q] += d * sw[c * C + cv * V + q];
      }
      *reinterpret_cast<uint4*>(dx + (p0 + r) * dx_ld * ES + cv * 16) = Elem<T>::pack(o);
    }
  }
  float* row = part + (int64_t)blockIdx.x * (NC * C + NC);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = tid + 256 * j;
    if (j < KPT && k < C) {
#pragma unroll
      for (int c = 0; c < kLinMaxNC; ++c)
        if (c < NC) row[c * C + k] = acc[j][c];
    }
  }
  if (tid < NC) row[NC * C + tid] = accb;
}

// The same backward on the matrix cores (bf16 activations, HW % 32 == 0, C % 64 == 0): strips of 32 pixels of ONE image.
//   dx^T[k][p] = sum_c w[c][k] dl[p][c]    A = w^T rows from an LDS image [C][32 classes] (bf16, zero rows for c >= NC), B = dl of the
//                                          strip (the gradient of the logits enters every other data gradient of the bf16 path in bf16 too);
//                                          the accumulators hold 4 consecutive channels of one pixel: 8-byte stores
//   dW[c][k] = sum_p dl[p][c] x[p][k]      A = dl^T[c][p] (f32 strip image in LDS, split into bf16 high + low parts: two MFMAs keep the
//                                          f32 gradient of the logits), B = x^T by ds_read_b64_tr_b16 from the row-major strip
//                                          [32 px][C] (+32 B per row: the eight rows a 32-lane half reads fall on distinct banks);
//                                          accumulators stay in registers over the workgroup's strips -> part[row][c][k]
//   db[c]     = sum_p dl[p][c]             wave 0, from the same fragments (f32)
// 4 waves: dx tiles split (pixel tile, channel half); dW channel tiles split in quarters.
typedef __attribute__((ext_vector_type(4))) short lin_s16x4;
typedef __attribute__((ext_vector_type(8))) short lin_s16x8;
template <int MT>   // class tiles of 16: NC <= 16 -> 1, NC <= 32 -> 2
__global__ __launch_bounds__(256, 2) void linear_head_bwd_mfma_kernel(const char* x, int64_t x_ld, const float* __restrict__ dl, int64_t P,
                                                                     int64_t HW, int C, const float* __restrict__ w, int NC, char* dx,
                                                                     int64_t dx_ld, float* part) {
  extern __shared__ __attribute__((aligned(16))) char lsm[];
  const int RS = C * 2 + 32;                   // x strip row stride (bytes)
  char* sx = lsm;                              // [32][RS]
  char* swT = sx + 32 * RS;                    // [C][64 B]: 32 classes of channel k, 16-byte chunk q stored at q ^ ((k >> 2) & 3)
  float* sdl = reinterpret_cast<float*>(swT + C * 64);   // [32 classes][32 px] f32
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, g = l >> 4, c16 = l & 15;
  for (int id = tid; id < C * 4; id += 256) {
    const int k = id >> 2, q = id & 3;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 8 * q + e < NC ? w[(int64_t)(8 * q + e) * C + k] : 0.f;
    *reinterpret_cast<uint4*>(swT + k * 64 + ((q ^ ((k >> 2) & 3)) << 4)) = Elem<bf16_t>::pack(f);
  }
  for (int id = tid; id < 32 * 32; id += 256) sdl[id] = 0.f;
  const int NTW = C / 64;                      // dW channel tiles per wave (C / 16 tiles over 4 waves)
  f32x4 acc[MT][8];                            // up to C = 512 per pass; wider heads take a second pass over the tiles
  float accb[MT] = {};
  const int npass = (NTW + 7) / 8;
  float* prow = part + (int64_t)blockIdx.x * ((int64_t)NC * C + NC);
  const int64_t nstrips = P / 32;
  for (int pass = 0; pass < npass; ++pass) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 8; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int64_t s = blockIdx.x; s < nstrips; s += gridDim.x) {
      const int64_t p0 = s * 32, b = p0 / HW, hw0 = p0 - b * HW;
      __syncthreads();
      for (int id = tid; id < 32 * (C / 8); id += 256) {
        const int r = id / (C / 8), cv = id - r * (C / 8);
        *reinterpret_cast<uint4*>(sx + r * RS + cv * 16) = *reinterpret_cast<const uint4*>(x + (p0 + r) * x_ld * 2 + cv * 16);
      }
      for (int id = tid; id < NC * 32; id += 256) {
        const int cc = id >> 5, pp = id & 31;
        sdl[id] = dl[(b * NC + cc) * HW + hw0 + pp];
      }
      __syncthreads();
      if (pass == 0) {
        // ---- dx: wave -> pixel tile (wv & 1), channel tiles of half (wv >> 1)
        const int pt = wv & 1;
        float d8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) d8[q] = sdl[(8 * g + q) * 32 + 16 * pt + c16];
        const bf16x8 bfrag = __builtin_bit_cast(bf16x8, Elem<bf16_t>::pack(d8));
        const int kt0 = (wv >> 1) * (C / 32), kt1 = kt0 + C / 32;
        char* drow = dx + (p0 + 16 * pt + c16) * dx_ld * 2;
        for (int kt = kt0; kt < kt1; ++kt) {
          const int k = 16 * kt + c16;
          const uint4 a = *reinterpret_cast<const uint4*>(swT + k * 64 + ((g ^ ((k >> 2) & 3)) << 4));
          const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), bfrag, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          *reinterpret_cast<uint2*>(drow + (16 * kt + 4 * g) * 2) = make_uint2(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]));
        }
      }
      // ---- dW: A = dl^T[class 16m + c16][pixels 4g..4g+3, 16+4g..] as high + low bf16 parts
      bf16x8 ahi[MT], alo[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(&sdl[(16 * m + c16) * 32 + 4 * g]);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(&sdl[(16 * m + c16) * 32 + 16 + 4 * g]);
        float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]}, lo[8];
        const uint4 hi = Elem<bf16_t>::pack(f);
        float fh[8];
        Elem<bf16_t>::unpack(hi, fh);
#pragma unroll
        for (int e = 0; e < 8; ++e) lo[e] = f[e] - fh[e];
        ahi[m] = __builtin_bit_cast(bf16x8, hi);
        alo[m] = __builtin_bit_cast(bf16x8, Elem<bf16_t>::pack(lo));
        if (pass == 0 && wv == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) accb[m] += f[e];
        }
      }
      const int q4 = (l >> 2) & 3, pp4 = l & 3;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const int nt = wv * NTW + pass * 8 + n;
        if (pass * 8 + n < NTW) {
          const char* a0 = sx + (4 * g + q4) * RS + (16 * nt + 4 * pp4) * 2;
          const lin_s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lin_s16x4*)a0);
          const lin_s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lin_s16x4*)(a0 + 16 * RS));
          const bf16x8 bx = __builtin_bit_cast(bf16x8, (lin_s16x8)__builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[m], bx, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[m], bx, acc[m][n], 0, 0, 0);
          }
        }
      }
    }
    // partial row: acc tile (m, n): lane holds classes 16m + 4g + r of channel 16nt + c16
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int nt = wv * NTW + pass * 8 + n;
      if (pass * 8 + n < NTW) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int cc = 16 * m + 4 * g + r;
            if (cc < NC) prow[(int64_t)cc * C + 16 * nt + c16] = acc[m][n][r];
          }
      }
    }
  }
  if (wv == 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float v = accb[m];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (g == 0 && 16 * m + c16 < NC) prow[(int64_t)NC * C + 16 * m + c16] = v;
    }
  }
}

static int dl_check_view(const char* fn, const void* ptr, int64_t ld, int C, int es) {
  CVCS_CHECK_ARG(ptr != nullptr, "%s: null tensor", fn);
  CVCS_CHECK_ARG(((uintptr_t)ptr % 16) == 0 && ld >= C && (ld * es) % 16 == 0, "%s: view must be 16-byte aligned with ld >= C", fn);
  return CVCS_OK;
}

}  // namespace cvcs

using namespace cvcs;
#define DL_DT_OK(dt) ((dt) == CVCS_F32 || (dt) == CVCS_BF16)

extern "C" int cvcs_resize_bilinear_fwd(const void* in, int64_t in_ld, int B, int H, int W, int C, int scale, void* out,
                                        int64_t out_ld, int dtype, void* stream) {
  const char* fn = "cvcs_resize_bilinear_fwd";
  CVCS_CHECK_ARG(DL_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0 && scale >= 1 && scale <= 32, "%s: bad shape", fn);
  int rc;
  if ((rc = dl_check_view(fn, in, in_ld, C, es)) || (rc = dl_check_view(fn, out, out_ld, C, es))) return rc;
  const dim3 grid(dl_grid((int64_t)B * H * W * scale * scale * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((resize_fwd_kernel<float>), grid, dim3(256), 0, st, (const char*)in, in_ld, B, H, W, C, scale, (char*)out, out_ld);
  else hipLaunchKernelGGL((resize_fwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)in, in_ld, B, H, W, C, scale, (char*)out, out_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_resize_bilinear_bwd(const void* gout, int64_t gout_ld, int B, int H, int W, int C, int scale, void* gin,
                                        int64_t gin_ld, int dtype, void* stream) {
  const char* fn = "cvcs_resize_bilinear_bwd";
  CVCS_CHECK_ARG(DL_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0 && scale >= 1 && scale <= 32, "%s: bad shape", fn);
  int rc;
  if ((rc = dl_check_view(fn, gout, gout_ld, C, es)) || (rc = dl_check_view(fn, gin, gin_ld, C, es))) return rc;
  const dim3 grid(dl_grid((int64_t)B * H * W * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((resize_bwd_kernel<float>), grid, dim3(256), 0, st, (const char*)gout, gout_ld, B, H, W, C, scale, (char*)gin, gin_ld);
  else hipLaunchKernelGGL((resize_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)gout, gout_ld, B, H, W, C, scale, (char*)gin, gin_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_resize_bilinear_nchw_fwd(const float* in, int64_t planes, int H, int W, int scale, float* out, void* stream) {
  const char* fn = "cvcs_resize_bilinear_nchw_fwd";
  CVCS_CHECK_ARG(in && out && planes > 0 && H > 0 && W > 0 && scale >= 1 && scale <= 32, "%s: bad arguments", fn);
  // (1/s is exact for the powers of two the networks use - the multiplication then gives the division's taps bit for bit; other scales keep the division)
  if (scale % 4 == 0 && (scale & (scale - 1)) == 0 && ((uintptr_t)out % 16) == 0)
    hipLaunchKernelGGL(resize_nchw_fwd_kernel, dim3(dl_grid(planes * H * W * scale * scale / 4)), dim3(256), 0, (hipStream_t)stream, in, planes, H, W, scale, out);
  else
    hipLaunchKernelGGL(resize_nchw_fwd1_kernel, dim3(dl_grid(planes * H * W * scale * scale)), dim3(256), 0, (hipStream_t)stream, in, planes, H, W, scale, out);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_resize_bilinear_nchw_bwd(const float* gout, int64_t planes, int H, int W, int scale, float* gin, void* stream) {
  const char* fn = "cvcs_resize_bilinear_nchw_bwd";
  CVCS_CHECK_ARG(gout && gin && planes > 0 && H > 0 && W > 0 && scale >= 1 && scale <= 32, "%s: bad arguments", fn);
  const int64_t Wo = (int64_t)W * scale, npair = (int64_t)W * (2 * scale + 2);
  const bool pow2 = (scale & (scale - 1)) == 0;
  if (scale >= 4 && W <= 256 && Wo % 4 == 0 && npair <= 256 * kRszPairs && planes * H <= 0x7fffffff && ((uintptr_t)gout % 16) == 0) {
    const dim3 grid((unsigned)(planes * H));
    const size_t lds = (size_t)(Wo + npair) * sizeof(float);
    if (pow2) hipLaunchKernelGGL(resize_nchw_bwd_rows_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, gout, H, W, scale, gin);
    else hipLaunchKernelGGL(resize_nchw_bwd_rows_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, gout, H, W, scale, gin);
  } else if (pow2) {
    hipLaunchKernelGGL(resize_nchw_bwd_kernel<true>, dim3(dl_grid(planes * H * W)), dim3(256), 0, (hipStream_t)stream, gout, planes, H, W, scale, gin);
  } else {
    hipLaunchKernelGGL(resize_nchw_bwd_kernel<false>, dim3(dl_grid(planes * H * W)), dim3(256), 0, (hipStream_t)stream, gout, planes, H, W, scale, gin);
  }
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_image_sum(const void* x, int64_t x_ld, int B, int HW, int C, float scale, void* out, int64_t out_ld, int dtype,
                              void* stream) {
  const char* fn = "cvcs_image_sum";
  CVCS_CHECK_ARG(DL_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % (16 / es) == 0, "%s: bad shape", fn);
  const int CC = C / (16 / es);
  int ccw = CC % 32 == 0 ? 32 : (CC < 256 && 256 % CC == 0 ? CC : 1);
  if (ccw == 1)      // widths like 96 / 160 / 960 (MobileNetV3): the largest divisor up to 32; the 256 - ccw * PL threads beyond the pixel lanes idle
    for (int d = 1; d <= 32 && d <= CC; ++d)
      if (CC % d == 0) ccw = d;
  int rc;
  if ((rc = dl_check_view(fn, x, x_ld, C, es)) || (rc = dl_check_view(fn, out, out_ld, C, es))) return rc;
  const dim3 grid((unsigned)B, (unsigned)(CC / ccw));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((image_sum_kernel<float>), grid, dim3(256), 0, st, (const char*)x, x_ld, HW, C, scale, (char*)out, out_ld, ccw);
  else hipLaunchKernelGGL((image_sum_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)x, x_ld, HW, C, scale, (char*)out, out_ld, ccw);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_image_broadcast(const void* v, int64_t v_ld, int B, int HW, int C, float scale, void* out, int64_t out_ld,
                                    int dtype, void* stream) {
  const char* fn = "cvcs_image_broadcast";
  CVCS_CHECK_ARG(DL_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % (16 / es) == 0, "%s: bad shape", fn);
  int rc;
  if ((rc = dl_check_view(fn, v, v_ld, C, es)) || (rc = dl_check_view(fn, out, out_ld, C, es))) return rc;
  const dim3 grid(dl_grid((int64_t)B * HW * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((image_broadcast_kernel<float>), grid, dim3(256), 0, st, (const char*)v, v_ld, B, HW, C, scale, (char*)out, out_ld);
  else hipLaunchKernelGGL((image_broadcast_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)v, v_ld, B, HW, C, scale, (char*)out, out_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}


// ------------------------------------------------------------------------------------------------ dropout (counter-based mask)
// nn.Dropout(p) of torchvision's ASPP (after the 1x1 projection; reached through S/nets.py:242-246 in train mode): out = keep ? x / (1 - p) : 0
// with keep decided by a counter-based hash of (seed, step, logical element index) - no generator state, so the mask is replayable on the
// host (tests) and identical in forward and backward (the backward is the same call on the gradient).  state = {seed, step} on the device:
// a recorded launch plan replays with fixed arguments, the step advances through cvcs_counter_add.
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long step, unsigned long long idx, unsigned thr24) {
  unsigned long long z = (seed ^ (step * 0xD6E8FEB86659FD93ull)) + idx * 0x9E3779B97F4A7C15ull;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 40) < thr24;
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const char* x, int64_t x_ld, int64_t M, int C, char* out, int64_t out_ld,
                                                      const unsigned long long* __restrict__ state, unsigned thr24, float inv_keep) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const unsigned long long seed = state[0], step = state[1];
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < M * CC; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    float f[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + (pix * x_ld) * ES + cc * 16), f);
#pragma unroll
    for (int k = 0; k < V; ++k)
      f[k] = dropout_keep(seed, step, (unsigned long long)pix * C + cc * V + k, thr24) ? f[k] * inv_keep : 0.f;
    *reinterpret_cast<uint4*>(out + (pix * out_ld) * ES + cc * 16) = Elem<T>::pack(f);
  }
}

__global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) { *c += inc; }

static int lin_args_ok(const char* fn, int C, int NC, int dtype) {
  CVCS_CHECK_ARG(DL_DT_OK(dtype), "%s: bad dtype", fn);
  CVCS_CHECK_ARG(C > 0 && C % 64 == 0 && C <= 1024, "%s: C=%d must be a multiple of 64, at most 1024", fn, C);
  CVCS_CHECK_ARG(NC >= 1 && NC <= kLinMaxNC, "%s: NC=%d out of [1,%d]", fn, NC, kLinMaxNC);
  return CVCS_OK;
}

extern "C" int cvcs_linear_head_fwd(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, const float* bias,
                                    int NC, float* logits, uint8_t* labels, int dtype, void* stream) {
  const char* fn = "cvcs_linear_head_fwd";
  int rc;
  if ((rc = lin_args_ok(fn, C, NC, dtype))) return rc;
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(x && w && (logits != nullptr) != (labels != nullptr) && B > 0 && H > 0 && W > 0, "%s: exactly one of logits / labels", fn);
  if ((rc = dl_check_view(fn, x, x_ld, C, es))) return rc;
  const int64_t P = (int64_t)B * H * W, HW = (int64_t)H * W;
  const size_t lds = (size_t)(NC * C + NC) * 4;
  const dim3 grid(dl_grid(P, 2048));
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_LIN(TT, AM)                                                                                                    \
  do {                                                                                                                        \
    static bool attr_done = false;                                                                                            \
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_head_kernel<TT, AM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_done = true; } \
    hipLaunchKernelGGL((linear_head_kernel<TT, AM>), grid, dim3(256), lds, st, (const char*)x, x_ld, P, HW, C, w, bias, NC, logits, labels); \
  } while (0)
  if (dtype == CVCS_F32) { if (labels) LAUNCH_LIN(float, true); else LAUNCH_LIN(float, false); }
  else { if (labels) LAUNCH_LIN(bf16_t, true); else LAUNCH_LIN(bf16_t, false); }
#undef LAUNCH_LIN
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

static int lin_strip(int C) { return C <= 256 ? 64 : (C <= 512 ? 32 : 16); }   // pixels per LDS-staged strip

extern "C" int cvcs_linear_head_bwd_rows(int64_t P) {
  int64_t r = cdiv(P, 64 * 8);   // ~8 strips (of at most 64 pixels) per workgroup
  return (int)(r < 1 ? 1 : (r > 512 ? 512 : r));
}

extern "C" int cvcs_linear_head_bwd(const void* x, int64_t x_ld, const float* dlogits, int B, int H, int W, int C, const float* w,
                                    int NC, void* dx, int64_t dx_ld, float* part_dw, int dtype, void* stream) {
  const char* fn = "cvcs_linear_head_bwd";
  int rc;
  if ((rc = lin_args_ok(fn, C, NC, dtype))) return rc;
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(x && dlogits && w && dx && part_dw && B > 0 && H > 0 && W > 0, "%s: null argument", fn);
  if ((rc = dl_check_view(fn, x, x_ld, C, es)) || (rc = dl_check_view(fn, dx, dx_ld, C, es))) return rc;
  const int64_t P = (int64_t)B * H * W, HW = (int64_t)H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_BF16 && HW % 32 == 0 && C % 64 == 0) {   // matrix-core path (see linear_head_bwd_mfma_kernel)
    const size_t lds_m = (size_t)32 * (C * 2 + 32) + (size_t)C * 64 + 32 * 32 * 4;
    const dim3 grid_m((unsigned)cvcs_linear_head_bwd_rows(P));
    if (NC <= 16) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_head_bwd_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((linear_head_bwd_mfma_kernel<1>), grid_m, dim3(256), lds_m, st, (const char*)x, x_ld, dlogits, P, HW, C, w, NC, (char*)dx, dx_ld, part_dw);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_head_bwd_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((linear_head_bwd_mfma_kernel<2>), grid_m, dim3(256), lds_m, st, (const char*)x, x_ld, dlogits, P, HW, C, w, NC, (char*)dx, dx_ld, part_dw);
    }
    CVCS_CHECK_LAUNCH(fn);
    return CVCS_OK;
  }
  const int TP = lin_strip(C);
  const size_t lds = (size_t)(TP * C + TP * (NC + 1) + NC * C) * 4;
  CVCS_CHECK_ARG(lds <= 160 * 1024, "%s: C=%d, NC=%d need %zu bytes of LDS", fn, C, NC, lds);
  const dim3 grid((unsigned)cvcs_linear_head_bwd_rows(P));
#define LAUNCH_LINB(TT)                                                                                                       \
  do {                                                                                                                        \
    static bool attr_done = false;                                                                                            \
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_head_bwd_kernel<TT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_done = true; } \
    hipLaunchKernelGGL((linear_head_bwd_kernel<TT>), grid, dim3(256), lds, st, (const char*)x, x_ld, dlogits, P, HW, C, w, NC, (char*)dx, dx_ld, part_dw, TP); \
  } while (0)
  if (dtype == CVCS_F32) LAUNCH_LINB(float); else LAUNCH_LINB(bf16_t);
#undef LAUNCH_LINB
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_dropout(const void* x, int64_t x_ld, int64_t M, int C, void* out, int64_t out_ld, const uint64_t* state, float p,
                            int dtype, void* stream) {
  const char* fn = "cvcs_dropout";
  CVCS_CHECK_ARG(DL_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(M > 0 && C > 0 && C % (16 / es) == 0 && state && p >= 0.f && p < 1.f, "%s: bad argument", fn);
  int rc;
  if ((rc = dl_check_view(fn, x, x_ld, C, es)) || (rc = dl_check_view(fn, out, out_ld, C, es))) return rc;
  const float keep = 1.f - p;
  const unsigned thr24 = (unsigned)(keep * 16777216.0f);
  const dim3 grid(dl_grid(M * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((dropout_kernel<float>), grid, dim3(256), 0, st, (const char*)x, x_ld, M, C, (char*)out, out_ld, (const unsigned long long*)state, thr24, 1.f / keep);
  else
    hipLaunchKernelGGL((dropout_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)x, x_ld, M, C, (char*)out, out_ld, (const unsigned long long*)state, thr24, 1.f / keep);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_counter_add(uint64_t* counter, uint64_t inc, void* stream) {
  CVCS_CHECK_ARG(counter != nullptr, "cvcs_counter_add: null counter");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)counter, (unsigned long long)inc);
  CVCS_CHECK_LAUNCH("cvcs_counter_add");
  return CVCS_OK;
}
