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

// One workgroup = kGfCB output channels.  v[cb][j] = sum_k G[k][j] w[c][k] (thread = column j: coalesced reads of G's rows, G symmetric),
// then q_c = sum_j v[cb][j] w[c][j] and mean_c = sum_k w[c][k] s[k] / M.  All in f64: var = q / M - mean^2 cancels.
__global__ __launch_bounds__(256) void bn_gram_finalize_kernel(const double* __restrict__ G, const double* __restrict__ s, const bf16_t* __restrict__ W,
                                                              int C, int m, int64_t M, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* rmean, float* rvar, float momentum, float eps,
                                                              float* scale, float* shift, float* save_mean, float* save_invstd, float* q_out) {
  extern __shared__ double sm[];
  double* wl = sm;                      // [kGfCB][m]
  double* red = sm + kGfCB * m;         // [2][kGfCB][256]
  const int c0 = blockIdx.x * kGfCB, tid = threadIdx.x;
  for (int i = tid; i < kGfCB * m; i += 256) {
    const int cb = i / m, k = i - cb * m;
    wl[i] = c0 + cb < C ? (double)bf16_to_f32(W[(int64_t)(c0 + cb) * m + k]) : 0.0;
  }
  __syncthreads();
  double pq[kGfCB], pm[kGfCB];
#pragma unroll
  for (int cb = 0; cb < kGfCB; ++cb) { pq[cb] = 0.0; pm[cb] = 0.0; }
  for (int j = tid; j < m; j += 256) {
    double v[kGfCB];
#pragma unroll
    for (int cb = 0; cb < kGfCB; ++cb) v[cb] = 0.0;
    for (int k = 0; k < m; ++k) {
      const double g = G[(int64_t)k * m + j];
#pragma unroll
      for (int cb = 0; cb < kGfCB; ++cb) v[cb] += g * wl[cb * m + k];
    }
    const double sj = s[j];
#pragma unroll
    for (int cb = 0; cb < kGfCB; ++cb) {
      if (c0 + cb < C) q_out[(int64_t)(c0 + cb) * m + j] = (float)v[cb];
      pq[cb] += v[cb] * wl[cb * m + j];
      pm[cb] += sj * wl[cb * m + j];
    }
  }
#pragma unroll
  for (int cb = 0; cb < kGfCB; ++cb) { red[cb * 256 + tid] = pq[cb]; red[(kGfCB + cb) * 256 + tid] = pm[cb]; }
  __syncthreads();
  if (tid < kGfCB && c0 + tid < C) {
    const int c = c0 + tid;
    double q = 0.0, mu = 0.0;
    const int n = m < 256 ? m : 256;     // (threads beyond m contributed zeros)
    for (int t = 0; t < n; ++t) { q += red[tid * 256 + t]; mu += red[(kGfCB + tid) * 256 + t]; }
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

// Mmat[j][k] = sum_c kappa_c w_cj w_ck -> wd2[j][C + k]; bias[j] = sum_c beta0_c w_cj.  32 x 32 outputs per workgroup, 2 x 2 per thread.
__global__ __launch_bounds__(256) void bn_gram_mmat_kernel(const bf16_t* __restrict__ W, const float* __restrict__ coef, int C, int m, bf16_t* wd2,
                                                          float* bias) {
  constexpr int CK = 32;
  __shared__ float wj[CK][33], wk[CK][33], kap[CK], b0[CK];
  const int j0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  float bacc = 0.f;
  const bool do_bias = blockIdx.x == 0;
  for (int cc = 0; cc < C; cc += CK) {
    for (int i = threadIdx.x; i < CK * 32; i += 256) {
      const int r = i >> 5, q = i & 31;
      const int c = cc + r;
      const bool ok = c < C;
      const float kv = ok ? coef[c] : 0.f;
      wj[r][q] = ok ? bf16_to_f32(W[(int64_t)c * m + j0 + q]) : 0.f;
      wk[r][q] = ok ? bf16_to_f32(W[(int64_t)c * m + k0 + q]) * kv : 0.f;
      if (q == 0) { kap[r] = kv; b0[r] = ok ? coef[C + c] : 0.f; }
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < CK; ++r) {
      const float a0 = wj[r][ty], a1 = wj[r][ty + 16], c0 = wk[r][tx], c1 = wk[r][tx + 16];
      acc[0][0] += a0 * c0; acc[0][1] += a0 * c1; acc[1][0] += a1 * c0; acc[1][1] += a1 * c1;
    }
    if (do_bias && threadIdx.x < 32) {
#pragma unroll 8
      for (int r = 0; r < CK; ++r) bacc += b0[r] * wj[r][threadIdx.x];
    }
    __syncthreads();
  }
  const int ktot = C + m;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) wd2[(int64_t)(j0 + ty + 16 * a) * ktot + C + k0 + tx + 16 * b] = f32_to_bf16(acc[a][b]);
  if (do_bias && threadIdx.x < 32) bias[j0 + threadIdx.x] = bacc;
}

}  // namespace cvcs

using namespace cvcs;

extern "C" int cvcs_bn_gram_finalize(const double* gram, const double* colsum, const void* w_bf16, int C, int m, int64_t M,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                     float* scale, float* shift, float* save_mean, float* save_invstd, float* q_out, void* stream) {
  CVCS_CHECK_ARG(gram && colsum && w_bf16 && gamma && beta && running_mean && running_var && scale && shift && save_mean && save_invstd && q_out,
                 "cvcs_bn_gram_finalize: null argument");
  CVCS_CHECK_ARG(C > 0 && m >= 32 && m % 32 == 0 && m <= 512 && M > 0, "cvcs_bn_gram_finalize: C=%d m=%d (m: a multiple of 32 up to 512)", C, m);
  const size_t lds = (size_t)(kGfCB * m + 2 * kGfCB * 256) * sizeof(double);
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

extern "C" int cvcs_bn_gram_mmat(const void* w_bf16, const float* coef, int C, int m, void* wd2_bf16, float* bias, void* stream) {
  CVCS_CHECK_ARG(w_bf16 && coef && wd2_bf16 && bias, "cvcs_bn_gram_mmat: null argument");
  CVCS_CHECK_ARG(C > 0 && m >= 32 && m % 32 == 0 && m <= 512, "cvcs_bn_gram_mmat: C=%d m=%d", C, m);
  hipLaunchKernelGGL(bn_gram_mmat_kernel, dim3((unsigned)(m / 32), (unsigned)(m / 32)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w_bf16, coef, C, m,
                     (bf16_t*)wd2_bf16, bias);
  CVCS_CHECK_LAUNCH("cvcs_bn_gram_mmat");
  return CVCS_OK;
}
