// BatchNorm behind a 1x1 convolution without a stored convolution output (gfx950).
//
// For y = W a (a: [M pixels][m] bf16, W: [C][m] bf16, f32 accumulation) every quantity a train-mode BatchNorm of y needs is a function of
// the m x m second-moment matrix G = sum_p a_p a_p^T and the column sums s = sum_p a_p (cvcs_gram, one pass over `a`):
//   forward   mean_c = w_c . s / M,   E[y_c^2] = w_c^T G w_c / M            -> scale / shift, applied in the conv's own epilogue
//   backward  with dz the gradient w.r.t. the BatchNorm output, R = dz^T a (the plain weight-gradient GEMM) and sum_dz (its bias sums):
//             sum_p dz_c y_c = w_c . R_c        -> dbeta = sum_dz, dgamma = invstd (w_c . R_c - mean_c sum_dz_c)
//             dy = alpha dz + beta0 + kappa y    (alpha = gamma invstd, kappa = -alpha invstd dgamma / M, beta0 = -alpha dbeta / M - kappa mean)
//             dW_c = alpha_c R_c + beta0_c s + kappa_c (W G)_c
//             da = W^T dy = (W^T diag(alpha)) dz + (W^T diag(kappa) W) a + W^T beta0   -> one GEMM over the sources [dz | a]
// In a ResNet bottleneck C = 4 m: the 4x wider tensors y and dy are never written or read (they were 3 + 4 tensor passes of the
// forward / backward of every block).  The finalize kernels here are O(C m^2) on the vector ALUs - f64 where sums can cancel.
//
// Reference ops replaced: nn.BatchNorm2d behind the last 1x1 convolution of torchvision's Bottleneck (the encoders of S/nets.py:234-311),
// forward (train mode) and backward (S/train.py:125).
#include "common.h"

namespace cvcs {

constexpr int kGfCB = 8;      // channels per workgroup of the forward finalize

// One workgroup = kGfCB output channels.  v[cb][j] = sum_k G[k][j] w[c][k] (a thread owns column j: coalesced reads of G's rows, G symmetric),
// then q_c = sum_j v[cb][j] w[c][j] and mean_c = sum_k w[c][k] s[k] / M.  All in f64: var = q / M - mean^2 cancels.
// The chain is latency-bound (one dependent L2 read per k): for m < 256 the k range is split over the 256 / m thread groups of the workgroup and
// merged through LDS in group order; every thread keeps 8 independent loads in flight.
__global__ __launch_bounds__(256) void bn_gram_finalize_kernel(const double* __restrict__ G, const double* __restrict__ s, const bf16_t* __restrict__ W,
                                                              int C, int m, int64_t M, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* rmean, float* rvar, float momentum, float eps,
                                                              float* scale, float* shift, float* save_mean, float* save_invstd, float* q_out) {
  extern __shared__ double sm[];
  double* wl = sm;                      // [kGfCB][m]
  double* red = sm + kGfCB * m;         // [2][kGfCB][256], also the [groups][kGfCB][m] merge buffer of the k split
  const int c0 = blockIdx.x * kGfCB, tid = threadIdx.x;
  for (int i = tid; i < kGfCB * m; i += 256) {
    const int cb = i / m, k = i - cb * m;
    wl[i] = c0 + cb < C ? (double)bf16_to_f32(W[(int64_t)(c0 + cb) * m + k]) : 0.0;
  }
  __syncthreads();
  double pq[kGfCB], pm[kGfCB];
#pragma unroll
  for (int cb = 0; cb < kGfCB; ++cb) { pq[cb] = 0.0; pm[cb] = 0.0; }
  const int groups = m >= 256 ? 1 : 256 / m;          // thread groups sharing a column
  const int jt = m >= 256 ? tid : tid % m, kg = m >= 256 ? 0 : tid / m;
  const int kper = m / groups;                          // (m and groups are powers of two times 32 / 64: m % groups == 0 for m = 32 ... 512)
  for (int j = jt; j < m; j += 256) {
    double v[kGfCB];
#pragma unroll
    for (int cb = 0; cb < kGfCB; ++cb) v[cb] = 0.0;
    const int kb = kg * kper, ke = kb + kper;
    int k = kb;
    for (; k + 8 <= ke; k += 8) {
      double g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) g[u] = G[(int64_t)(k + u) * m + j];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int cb = 0; cb < kGfCB; ++cb) v[cb] += g[u] * wl[cb * m + k + u];
    }
    for (; k < ke; ++k) {
      const double g = G[(int64_t)k * m + j];
#pragma unroll
      for (int cb = 0; cb < kGfCB; ++cb) v[cb] += g * wl[cb * m + k];
    }
    if (groups > 1) {
      // merge the k ranges in group order (one j per thread when m < 256)
#pragma unroll
      for (int cb = 0; cb < kGfCB; ++cb) red[(kg * kGfCB + cb) * m + j] = v[cb];
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int cb = 0; cb < kGfCB; ++cb) {
          double t = red[cb * m + j];
          for (int q = 1; q < groups; ++q) t += red[(q * kGfCB + cb) * m + j];
          v[cb] = t;
        }
      }
      __syncthreads();
    }
    if (kg == 0) {
      const double sj = s[j];
#pragma unroll
      for (int cb = 0; cb < kGfCB; ++cb) {
        if (c0 + cb < C) q_out[(int64_t)(c0 + cb) * m + j] = (float)v[cb];
        pq[cb] += v[cb] * wl[cb * m + j];
        pm[cb] += sj * wl[cb * m + j];
      }
    }
  }
#pragma unroll
  for (int cb = 0; cb < kGfCB; ++cb) { red[cb * 256 + tid] = pq[cb]; red[(kGfCB + cb) * 256 + tid] = pm[cb]; }
  __syncthreads();
  // 16 lanes per channel sum the 256 per-thread partials (threads that own no column hold zeros), merged in lane order
  {
    const int cb = tid >> 4, l = tid & 15;
    double q = 0.0, mu = 0.0;
    if (cb < kGfCB)
      for (int t = l; t < 256; t += 16) { q += red[cb * 256 + t]; mu += red[(kGfCB + cb) * 256 + t]; }
    __syncthreads();
    if (cb < kGfCB) { red[cb * 16 + l] = q; red[(kGfCB + cb) * 16 + l] = mu; }
    __syncthreads();
  }
  if (tid < kGfCB && c0 + tid < C) {
    const int c = c0 + tid;
    double q = 0.0, mu = 0.0;
    for (int t = 0; t < 16; ++t) { q += red[tid * 16 + t]; mu += red[(kGfCB + tid) * 16 + t]; }
    const double Md = (double)M;
    mu /= Md;
    double var = q / Md - mu * mu;
    var = var > 0.0 ? var : 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[c] * is;
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - mu * sc);
    save_mean[c] = (float)mu;
    save_invstd[c] = (float)is;
    const double unb = M > 1 ? var * Md / (Md - 1.0) : var;
    rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mu);
    rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
  }
}

// Backward coefficients + weight gradient + the dz columns of the packed data-gradient weight.  One workgroup = 16 channels x 16 column lanes.
__global__ __launch_bounds__(256) void bn_gram_bwd_kernel(const float* __restrict__ R, const float* __restrict__ sum_dz, const bf16_t* __restrict__ W,
                                                         const float* __restrict__ Q, const double* __restrict__ s, int C, int m, int64_t M,
                                                         const float* __restrict__ scale, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, float* dgamma, float* dbeta, float* dw,
                                                         bf16_t* wd2, float* coef) {
  __shared__ double red[16][17];
  __shared__ float al[16], be[16], ka[16];
  __shared__ bf16_t tile[16][16 + 2];
  const int cl = threadIdx.x >> 4, jl = threadIdx.x & 15;
  const int c = blockIdx.x * 16 + cl;
  const bool cok = c < C;
  double u = 0.0;
  if (cok)
    for (int j = jl; j < m; j += 16) u += (double)bf16_to_f32(W[(int64_t)c * m + j]) * (double)R[(int64_t)c * m + j];
  red[cl][jl] = u;
  __syncthreads();
  if (jl == 0 && cok) {
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t += red[cl][q];
    const double Md = (double)M, sdz = (double)sum_dz[c], mu = (double)mean[c], is = (double)invstd[c], alpha = (double)scale[c];
    const double dgam = is * (t - mu * sdz);
    dbeta[c] = (float)sdz;
    dgamma[c] = (float)dgam;
    const double kappa = -alpha * is * dgam / Md;
    const double beta0 = -alpha * sdz / Md - kappa * mu;
    al[cl] = (float)alpha; be[cl] = (float)beta0; ka[cl] = (float)kappa;
    coef[c] = (float)kappa;
    coef[C + c] = (float)beta0;
  }
  __syncthreads();
  const int ktot = C + m;
  const int c0 = blockIdx.x * 16;
  for (int j0 = 0; j0 < m; j0 += 16) {
    const int j = j0 + jl;
    if (cok) {
      const float w = bf16_to_f32(W[(int64_t)c * m + j]);
      dw[(int64_t)c * m + j] = al[cl] * R[(int64_t)c * m + j] + be[cl] * (float)s[j] + ka[cl] * Q[(int64_t)c * m + j];
      tile[cl][jl] = f32_to_bf16(al[cl] * w);
    } else {
      tile[cl][jl] = 0;
    }
    __syncthreads();
    // transposed store: row j0 + cl of wd2, columns c0 + jl
    if (c0 + jl < C) wd2[(int64_t)(j0 + cl) * ktot + c0 + jl] = tile[jl][cl];
    __syncthreads();
  }
}

// Mmat[j][k] = sum_c kappa_c w_cj w_ck -> wd2[j][C + k]; bias[j] = sum_c beta0_c w_cj.  Latency matters more than FLOPs here (the launch sits
// on the backward's critical chain): one workgroup = a 32 x 32 output tile x ONE chunk of kMmCK channels, partial tiles [chunk][m][m] (+ bias rows
// [chunk][m]) to a workspace; bn_gram_mmat_reduce_kernel sums the chunks in order and writes the bf16 columns.
constexpr int kMmCK = 64;
__global__ __launch_bounds__(256) void bn_gram_mmat_kernel(const bf16_t* __restrict__ W, const float* __restrict__ coef, int C, int m, float* part) {
  __shared__ float wj[kMmCK][33], wk[kMmCK][33], b0[kMmCK];
  const int j0 = blockIdx.y * 32, k0 = blockIdx.x * 32, cc = blockIdx.z * kMmCK;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  for (int i = threadIdx.x; i < kMmCK * 32; i += 256) {
    const int r = i >> 5, q = i & 31;
    const int c = cc + r;
    const bool ok = c < C;
    const float kv = ok ? coef[c] : 0.f;
    wj[r][q] = ok ? bf16_to_f32(W[(int64_t)c * m + j0 + q]) : 0.f;
    wk[r][q] = ok ? bf16_to_f32(W[(int64_t)c * m + k0 + q]) * kv : 0.f;
    if (q == 0) b0[r] = ok ? coef[C + c] : 0.f;
  }
  __syncthreads();
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll 8
  for (int r = 0; r < kMmCK; ++r) {
    const float a0 = wj[r][ty], a1 = wj[r][ty + 16], c0 = wk[r][tx], c1 = wk[r][tx + 16];
    acc[0][0] += a0 * c0; acc[0][1] += a0 * c1; acc[1][0] += a1 * c0; acc[1][1] += a1 * c1;
  }
  float* pt = part + (int64_t)blockIdx.z * ((int64_t)m * m + m);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) pt[(int64_t)(j0 + ty + 16 * a) * m + k0 + tx + 16 * b] = acc[a][b];
  if (blockIdx.x == 0 && threadIdx.x < 32) {
    float bacc = 0.f;
#pragma unroll 8
    for (int r = 0; r < kMmCK; ++r) bacc += b0[r] * wj[r][threadIdx.x];
    pt[(int64_t)m * m + j0 + threadIdx.x] = bacc;
  }
}
__global__ __launch_bounds__(256) void bn_gram_mmat_reduce_kernel(const float* __restrict__ part, int nchunk, int C, int m, bf16_t* wd2, float* bias) {
  const int64_t total = (int64_t)m * m + m;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  float a = 0.f;
  int z = 0;
  for (; z + 4 <= nchunk; z += 4) {
    const float v0 = part[(int64_t)z * total + id], v1 = part[(int64_t)(z + 1) * total + id];
    const float v2 = part[(int64_t)(z + 2) * total + id], v3 = part[(int64_t)(z + 3) * total + id];
    a += v0; a += v1; a += v2; a += v3;
  }
  for (; z < nchunk; ++z) a += part[(int64_t)z * total + id];
  if (id < (int64_t)m * m) {
    const int j = (int)(id / m), k = (int)(id - (int64_t)j * m);
    wd2[(int64_t)j * (C + m) + C + k] = f32_to_bf16(a);
  } else {
    bias[id - (int64_t)m * m] = a;
  }
}

// out[c][0 .. m1) = bf16(scale1[c] * w1[c][:]), out[c][m1 .. m1 + m2) = bf16(scale2[c] * w2[c][:]), bias[c] = shift1[c] + shift2[c]: the operand of
// ONE two-source GEMM relu([a | h] out^T + bias) = relu(BN3(W3 a) + BNd(Wd h)), the tail of a bottleneck block with a same-resolution projection
// shortcut whose two BatchNorms both come from Gram matrices (neither conv output is ever stored).  A thread converts 8 consecutive columns.
__global__ __launch_bounds__(256) void bn_gram_fold_kernel(const bf16_t* __restrict__ w1, const float* __restrict__ scale1, const float* __restrict__ shift1, int m1,
                                                          const bf16_t* __restrict__ w2, const float* __restrict__ scale2, const float* __restrict__ shift2, int m2,
                                                          int C, bf16_t* __restrict__ out, float* __restrict__ bias) {
  const int per = (m1 + m2) / 8;
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id < C) bias[id] = shift1[id] + shift2[id];
  if (id >= C * per) return;
  const int c = id / per, k0 = (id - c * per) * 8;
  const bool first = k0 < m1;
  const bf16_t* src = first ? w1 + (int64_t)c * m1 + k0 : w2 + (int64_t)c * m2 + (k0 - m1);
  const float sc = first ? scale1[c] : scale2[c];
  float f[8];
  Elem<bf16_t>::unpack(*reinterpret_cast<const uint4*>(src), f);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] *= sc;
  *reinterpret_cast<uint4*>(out + (int64_t)c * (m1 + m2) + k0) = Elem<bf16_t>::pack(f);
}

}  // namespace cvcs

using namespace cvcs;

extern "C" int cvcs_bn_gram_finalize(const double* gram, const double* colsum, const void* w_bf16, int C, int m, int64_t M,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                     float* scale, float* shift, float* save_mean, float* save_invstd, float* q_out, void* stream) {
  CVCS_CHECK_ARG(gram && colsum && w_bf16 && gamma && beta && running_mean && running_var && scale && shift && save_mean && save_invstd && q_out,
                 "cvcs_bn_gram_finalize: null argument");
  CVCS_CHECK_ARG(C > 0 && m >= 32 && m % 32 == 0 && m <= 512 && M > 0, "cvcs_bn_gram_finalize: C=%d m=%d (m: a multiple of 32 up to 512)", C, m);
  const size_t lds = (size_t)(kGfCB * m + 2 * kGfCB * 256) * sizeof(double);      // (the k-split merge buffer [256 / m][kGfCB][m] has the same size)
  hipLaunchKernelGGL(bn_gram_finalize_kernel, dim3((unsigned)cdiv(C, kGfCB)), dim3(256), lds, (hipStream_t)stream, gram, colsum, (const bf16_t*)w_bf16, C, m, M,
                     gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd, q_out);
  CVCS_CHECK_LAUNCH("cvcs_bn_gram_finalize");
  return CVCS_OK;
}

extern "C" int cvcs_bn_gram_bwd(const float* r, const float* sum_dz, const void* w_bf16, const float* q, const double* colsum, int C, int m, int64_t M,
                                const float* scale, const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* dw,
                                void* wd2_bf16, float* coef, void* stream) {
  CVCS_CHECK_ARG(r && sum_dz && w_bf16 && q && colsum && scale && save_mean && save_invstd && dgamma && dbeta && dw && wd2_bf16 && coef,
                 "cvcs_bn_gram_bwd: null argument");
  CVCS_CHECK_ARG(C > 0 && C % 16 == 0 && m >= 32 && m % 32 == 0 && m <= 512 && M > 0, "cvcs_bn_gram_bwd: C=%d m=%d", C, m);
  hipLaunchKernelGGL(bn_gram_bwd_kernel, dim3((unsigned)(C / 16)), dim3(256), 0, (hipStream_t)stream, r, sum_dz, (const bf16_t*)w_bf16, q, colsum, C, m, M,
                     scale, save_mean, save_invstd, dgamma, dbeta, dw, (bf16_t*)wd2_bf16, coef);
  CVCS_CHECK_LAUNCH("cvcs_bn_gram_bwd");
  return CVCS_OK;
}

extern "C" int64_t cvcs_bn_gram_mmat_workspace_floats(int C, int m) {
  if (C <= 0 || m < 32 || m % 32 != 0 || m > 512) return CVCS_EINVAL;
  return cdiv(C, kMmCK) * ((int64_t)m * m + m);
}

extern "C" int cvcs_bn_gram_mmat(const void* w_bf16, const float* coef, int C, int m, void* wd2_bf16, float* bias, float* workspace, void* stream) {
  CVCS_CHECK_ARG(w_bf16 && coef && wd2_bf16 && bias && workspace, "cvcs_bn_gram_mmat: null argument");
  CVCS_CHECK_ARG(C > 0 && m >= 32 && m % 32 == 0 && m <= 512, "cvcs_bn_gram_mmat: C=%d m=%d", C, m);
  const int nchunk = (int)cdiv(C, kMmCK);
  hipLaunchKernelGGL(bn_gram_mmat_kernel, dim3((unsigned)(m / 32), (unsigned)(m / 32), (unsigned)nchunk), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w_bf16, coef, C, m, workspace);
  CVCS_CHECK_LAUNCH("cvcs_bn_gram_mmat");
  hipLaunchKernelGGL(bn_gram_mmat_reduce_kernel, dim3((unsigned)cdiv((int64_t)m * m + m, 256)), dim3(256), 0, (hipStream_t)stream, workspace, nchunk, C, m,
                     (bf16_t*)wd2_bf16, bias);
  CVCS_CHECK_LAUNCH("cvcs_bn_gram_mmat(reduce)");
  return CVCS_OK;
}

extern "C" int cvcs_bn_gram_fold(const void* w1_bf16, const float* scale1, const float* shift1, int m1, const void* w2_bf16, const float* scale2,
                                 const float* shift2, int m2, int C, void* w_out_bf16, float* bias, void* stream) {
  CVCS_CHECK_ARG(w1_bf16 && scale1 && shift1 && w2_bf16 && scale2 && shift2 && w_out_bf16 && bias, "cvcs_bn_gram_fold: null argument");
  CVCS_CHECK_ARG(C > 0 && m1 > 0 && m2 > 0 && m1 % 8 == 0 && m2 % 8 == 0, "cvcs_bn_gram_fold: C=%d m1=%d m2=%d (rows of whole 16-byte chunks)", C, m1, m2);
  CVCS_CHECK_ARG(((uintptr_t)w1_bf16 % 16) == 0 && ((uintptr_t)w2_bf16 % 16) == 0 && ((uintptr_t)w_out_bf16 % 16) == 0, "cvcs_bn_gram_fold: alignment");
  const int64_t n = (int64_t)C * ((m1 + m2) / 8);
  hipLaunchKernelGGL(bn_gram_fold_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w1_bf16, scale1, shift1, m1,
                     (const bf16_t*)w2_bf16, scale2, shift2, m2, C, (bf16_t*)w_out_bf16, bias);
  CVCS_CHECK_LAUNCH("cvcs_bn_gram_fold");
  return CVCS_OK;
}
