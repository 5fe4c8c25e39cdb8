// Fused GroupNorm + SiLU on gfx950 (north star: "fused BN/GN+SiLU"; the `decoder_norm: gn_silu` option of the ResNet-UNets -
// the reference itself only uses BatchNorm + ReLU, S/blocks.py:8-49).  NHWC activations, statistics per (image, group).
//
//   forward : one streaming pass for the per-(image, channel) sums (sum y, sum y^2), a tiny finalize that merges the channels of
//             a group in f64 and emits per-(image, channel) scale / shift, ONE pass that normalises, applies the affine and
//             the SiLU (z * sigmoid(z)) - normalisation and activation never touch HBM separately;
//   backward: dz = g * silu'(z) is recomputed from (y, g) in both passes; pass 1 the per-(image, channel) sums (sum dz,
//             sum dz*y), finalize -> dgamma, dbeta and three per-(image, channel) coefficients, pass 2
//             dy = ca*dz + cb + cc*y.  No atomics; fixed-order f64 merges.
#include "common.h"

namespace cvcs {

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad(float z) {
  const float s = 1.f / (1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}

// per-(image, channel) partial sums over a slice of the image's pixels: part[b][row][0|1][c]
//   MODE 0 (forward statistics): u = y, v = y^2        MODE 1 (backward): u = dz, v = dz * y, dz = g * act'(scale*y + shift)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void gn_sums_kernel(const char* y, int64_t y_ld, const char* g, int64_t g_ld, int HW, int C,
                                                     const float* __restrict__ scale, const float* __restrict__ shift, int silu,
                                                     float* part, int rows) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  __shared__ float red[2][256 * V];
  const int CC = C / V;
  const int ccw = CC < 256 ? CC : 256;
  const int PL = 256 / ccw;
  const int cl = threadIdx.x % ccw, pl = threadIdx.x / ccw;
  const int cc = blockIdx.z * ccw + cl;
  const int b = blockIdx.y, row = blockIdx.x;
  float sc[V], sh[V], s0[V], s1[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    s0[k] = s1[k] = 0.f;
    sc[k] = MODE ? scale[(int64_t)b * C + cc * V + k] : 0.f;
    sh[k] = MODE ? shift[(int64_t)b * C + cc * V + k] : 0.f;
  }
  for (int p = row * PL + pl; p < HW; p += rows * PL) {
    const int64_t pix = (int64_t)b * HW + p;
    float yv[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + pix * y_ld * ES + cc * 16), yv);
    if constexpr (MODE == 0) {
#pragma unroll
      for (int k = 0; k < V; ++k) { s0[k] += yv[k]; s1[k] += yv[k] * yv[k]; }
    } else {
      float gv[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(g + pix * g_ld * ES + cc * 16), gv);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float dz = silu ? gv[k] * silu_grad(sc[k] * yv[k] + sh[k]) : gv[k];
        s0[k] += dz; s1[k] += dz * yv[k];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) { red[0][threadIdx.x * V + k] = s0[k]; red[1][threadIdx.x * V + k] = s1[k]; }
  __syncthreads();
  if (pl == 0) {
    float* o = part + (((int64_t)b * rows + row) * 2) * C + cc * V;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float a = 0.f, c2 = 0.f;
      for (int q = 0; q < PL; ++q) { a += red[0][(q * ccw + cl) * V + k]; c2 += red[1][(q * ccw + cl) * V + k]; }
      o[k] = a; o[C + k] = c2;
    }
  }
}

// one thread per (image, group): forward finalize
__global__ void gn_finalize_kernel(const float* __restrict__ part, int rows, int B, int HW, int C, int G,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* scale,
                                   float* shift, float* mean_out, float* invstd_out) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * G) return;
  const int b = id / G, gi = id % G, cg = C / G;
  double s = 0.0, q = 0.0;
  for (int r = 0; r < rows; ++r) {
    const float* p = part + (((int64_t)b * rows + r) * 2) * C + gi * cg;
    for (int c = 0; c < cg; ++c) { s += (double)p[c]; q += (double)p[C + c]; }
  }
  const double n = (double)HW * cg;
  const double mean = s / n;
  double var = q / n - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  mean_out[id] = (float)mean; invstd_out[id] = (float)invstd;
  for (int c = 0; c < cg; ++c) {
    const int ch = gi * cg + c;
    const double sc = (double)gamma[ch] * invstd;
    scale[(int64_t)b * C + ch] = (float)sc;
    shift[(int64_t)b * C + ch] = (float)((double)beta[ch] - mean * sc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_act_kernel(const char* y, int64_t y_ld, int B, int HW, int C, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int silu, char* out, int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int64_t total = (int64_t)B * HW * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    const int64_t b = pix / HW;
    float f[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + pix * y_ld * ES + cc * 16), f);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float z = f[k] * scale[b * C + cc * V + k] + shift[b * C + cc * V + k];
      f[k] = silu ? silu_f(z) : z;
    }
    *reinterpret_cast<uint4*>(out + pix * out_ld * ES + cc * 16) = Elem<T>::pack(f);
  }
}

// backward finalize, one thread per (image, group): per-(image, channel) coefficients of pass 2 and the per-image shares of
// dgamma / dbeta (tmp[b][0|1][c]; summed over images by gn_param_grad_kernel in a fixed order)
__global__ void gn_bwd_finalize_kernel(const float* __restrict__ part, int rows, int B, int HW, int C, int G,
                                       const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                                       float* tmp, float* ca, float* cb, float* cc) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * G) return;
  const int b = id / G, gi = id % G, cg = C / G;
  const double mu = mean[id], is = invstd[id];
  double s1 = 0.0, s2 = 0.0;   // sum_c gamma * sum dz,  sum_c gamma * sum dz * xhat
  for (int c = 0; c < cg; ++c) {
    const int ch = gi * cg + c;
    double sdz = 0.0, sdzy = 0.0;
    for (int r = 0; r < rows; ++r) {
      const float* p = part + (((int64_t)b * rows + r) * 2) * C + ch;
      sdz += (double)p[0]; sdzy += (double)p[C];
    }
    const double sdzx = is * (sdzy - mu * sdz);          // sum dz * xhat
    tmp[((int64_t)b * 2 + 0) * C + ch] = (float)sdzx;    // dgamma share
    tmp[((int64_t)b * 2 + 1) * C + ch] = (float)sdz;     // dbeta share
    s1 += (double)gamma[ch] * sdz;
    s2 += (double)gamma[ch] * sdzx;
  }
  const double n = (double)HW * cg;
  // dy = is*gamma*dz - is*s1/n - is*xhat*s2/n,  xhat = (y - mu)*is
  const double kb = -is * s1 / n + is * is * mu * s2 / n, kc = -is * is * s2 / n;
  for (int c = 0; c < cg; ++c) {
    const int ch = gi * cg + c;
    ca[(int64_t)b * C + ch] = (float)(is * (double)gamma[ch]);
    cb[(int64_t)b * C + ch] = (float)kb;
    cc[(int64_t)b * C + ch] = (float)kc;
  }
}

__global__ void gn_param_grad_kernel(const float* __restrict__ tmp, int B, int C, float* dgamma, float* dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, d = 0.0;
  for (int b = 0; b < B; ++b) { a += (double)tmp[((int64_t)b * 2 + 0) * C + c]; d += (double)tmp[((int64_t)b * 2 + 1) * C + c]; }
  dgamma[c] = (float)a; dbeta[c] = (float)d;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const char* y, int64_t y_ld, const char* g, int64_t g_ld, int B, int HW, int C,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, int silu,
                                                          const float* __restrict__ ca, const float* __restrict__ cb,
                                                          const float* __restrict__ cc, char* dy, int64_t dy_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int64_t total = (int64_t)B * HW * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cch = (int)(id % CC);
    const int64_t pix = id / CC;
    const int64_t b = pix / HW;
    float yv[V], gv[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + pix * y_ld * ES + cch * 16), yv);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(g + pix * g_ld * ES + cch * 16), gv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int64_t i = b * C + cch * V + k;
      const float dz = silu ? gv[k] * silu_grad(scale[i] * yv[k] + shift[i]) : gv[k];
      gv[k] = ca[i] * dz + cb[i] + cc[i] * yv[k];
    }
    *reinterpret_cast<uint4*>(dy + pix * dy_ld * ES + cch * 16) = Elem<T>::pack(gv);
  }
}

static int gn_check(const char* fn, int B, int HW, int C, int G, int dtype) {
  CVCS_CHECK_ARG(dtype == CVCS_F32 || dtype == CVCS_BF16, "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % V == 0, "%s: bad shape", fn);
  CVCS_CHECK_ARG(G <= 0 || (C % G == 0), "%s: C=%d is no multiple of the %d groups", fn, C, G);
  const int CC = C / V, ccw = CC < 256 ? CC : 256;
  CVCS_CHECK_ARG(256 % ccw == 0 && CC % ccw == 0, "%s: C/%d must divide 256 or be a multiple of 256", fn, V);
  return CVCS_OK;
}
static int gn_view(const char* fn, const void* p, int64_t ld, int C, int es) {
  CVCS_CHECK_ARG(p && ((uintptr_t)p % 16) == 0 && ld >= C && (ld * es) % 16 == 0, "%s: view must be 16-byte aligned with ld >= C", fn);
  return CVCS_OK;
}

}  // namespace cvcs

using namespace cvcs;

extern "C" int cvcs_gn_rows(int HW) {   // partial rows per image of the two sum passes
  int r = (HW + 1023) / 1024;
  return r < 1 ? 1 : (r > 64 ? 64 : r);
}

static int gn_sums(const char* fn, int mode, const void* y, int64_t y_ld, const void* g, int64_t g_ld, int B, int HW, int C,
                   const float* scale, const float* shift, int silu, float* part, int dtype, void* stream) {
  int rc;
  if ((rc = gn_check(fn, B, HW, C, 0, dtype))) return rc;
  const int es = dtype == CVCS_F32 ? 4 : 2;
  if ((rc = gn_view(fn, y, y_ld, C, es))) return rc;
  if (mode && (rc = gn_view(fn, g, g_ld, C, es))) return rc;
  CVCS_CHECK_ARG(part && (!mode || (scale && shift)), "%s: null argument", fn);
  const int rows = cvcs_gn_rows(HW);
  const int CC = C / (16 / es), ccw = CC < 256 ? CC : 256;
  const dim3 grid((unsigned)rows, (unsigned)B, (unsigned)(CC / ccw));
  hipStream_t st = (hipStream_t)stream;
#define GN_SUMS(TT, MD) hipLaunchKernelGGL((gn_sums_kernel<TT, MD>), grid, dim3(256), 0, st, (const char*)y, y_ld, (const char*)g, g_ld, HW, C, scale, shift, silu, part, rows)
  if (dtype == CVCS_F32) { if (mode) GN_SUMS(float, 1); else GN_SUMS(float, 0); }
  else { if (mode) GN_SUMS(bf16_t, 1); else GN_SUMS(bf16_t, 0); }
#undef GN_SUMS
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_gn_stats(const void* y, int64_t y_ld, int B, int HW, int C, float* part, int dtype, void* stream) {
  return gn_sums("cvcs_gn_stats", 0, y, y_ld, nullptr, 0, B, HW, C, nullptr, nullptr, 0, part, dtype, stream);
}

extern "C" int cvcs_gn_finalize(const float* part, int B, int HW, int C, int G, const float* gamma, const float* beta, float eps,
                                float* scale, float* shift, float* mean, float* invstd, void* stream) {
  const char* fn = "cvcs_gn_finalize";
  CVCS_CHECK_ARG(part && gamma && beta && scale && shift && mean && invstd && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "%s: bad arguments", fn);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)cdiv(B * G, 64)), dim3(64), 0, (hipStream_t)stream, part, cvcs_gn_rows(HW), B, HW, C, G,
                     gamma, beta, eps, scale, shift, mean, invstd);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_gn_act_fwd(const void* y, int64_t y_ld, int B, int HW, int C, const float* scale, const float* shift, int silu,
                               void* out, int64_t out_ld, int dtype, void* stream) {
  const char* fn = "cvcs_gn_act_fwd";
  int rc;
  if ((rc = gn_check(fn, B, HW, C, 0, dtype))) return rc;
  const int es = dtype == CVCS_F32 ? 4 : 2;
  if ((rc = gn_view(fn, y, y_ld, C, es)) || (rc = gn_view(fn, out, out_ld, C, es))) return rc;
  CVCS_CHECK_ARG(scale && shift, "%s: null argument", fn);
  int64_t gx = cdiv((int64_t)B * HW * (C / (16 / es)), 256);
  const dim3 grid((unsigned)(gx > 8192 ? 8192 : gx));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((gn_act_kernel<float>), grid, dim3(256), 0, st, (const char*)y, y_ld, B, HW, C, scale, shift, silu, (char*)out, out_ld);
  else hipLaunchKernelGGL((gn_act_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)y, y_ld, B, HW, C, scale, shift, silu, (char*)out, out_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_gn_act_bwd_reduce(const void* y, int64_t y_ld, const void* g, int64_t g_ld, int B, int HW, int C,
                                      const float* scale, const float* shift, int silu, float* part, int dtype, void* stream) {
  return gn_sums("cvcs_gn_act_bwd_reduce", 1, y, y_ld, g, g_ld, B, HW, C, scale, shift, silu, part, dtype, stream);
}

extern "C" int cvcs_gn_bwd_finalize(const float* part, int B, int HW, int C, int G, const float* gamma, const float* mean,
                                    const float* invstd, float* tmp, float* dgamma, float* dbeta, float* ca, float* cb, float* cc,
                                    void* stream) {
  const char* fn = "cvcs_gn_bwd_finalize";
  CVCS_CHECK_ARG(part && gamma && mean && invstd && tmp && dgamma && dbeta && ca && cb && cc && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0,
                 "%s: bad arguments", fn);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((unsigned)cdiv(B * G, 64)), dim3(64), 0, st, part, cvcs_gn_rows(HW), B, HW, C, G, gamma, mean,
                     invstd, tmp, ca, cb, cc);
  hipLaunchKernelGGL(gn_param_grad_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, st, tmp, B, C, dgamma, dbeta);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_gn_act_bwd_apply(const void* y, int64_t y_ld, const void* g, int64_t g_ld, int B, int HW, int C, const float* scale,
                                     const float* shift, int silu, const float* ca, const float* cb, const float* cc, void* dy,
                                     int64_t dy_ld, int dtype, void* stream) {
  const char* fn = "cvcs_gn_act_bwd_apply";
  int rc;
  if ((rc = gn_check(fn, B, HW, C, 0, dtype))) return rc;
  const int es = dtype == CVCS_F32 ? 4 : 2;
  if ((rc = gn_view(fn, y, y_ld, C, es)) || (rc = gn_view(fn, g, g_ld, C, es)) || (rc = gn_view(fn, dy, dy_ld, C, es))) return rc;
  CVCS_CHECK_ARG(scale && shift && ca && cb && cc, "%s: null argument", fn);
  int64_t gx = cdiv((int64_t)B * HW * (C / (16 / es)), 256);
  const dim3 grid((unsigned)(gx > 8192 ? 8192 : gx));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((gn_bwd_apply_kernel<float>), grid, dim3(256), 0, st, (const char*)y, y_ld, (const char*)g, g_ld, B, HW, C, scale, shift, silu, ca, cb, cc, (char*)dy, dy_ld);
  else hipLaunchKernelGGL((gn_bwd_apply_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)y, y_ld, (const char*)g, g_ld, B, HW, C, scale, shift, silu, ca, cb, cc, (char*)dy, dy_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}
