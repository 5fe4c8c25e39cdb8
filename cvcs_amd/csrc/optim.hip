// Fused multi-tensor optimiser steps over one flat f32 parameter buffer (gfx950, HBM-bound: 16-byte lanes).
// The flat buffer is the same one the gradient all-reduce uses, so one launch updates all 31 M parameters.
//
// Reference ops replaced: torch.optim.SGD(lr, momentum=0.9, weight_decay=1e-5).step() (S/utils.py:214) and
// torch.optim.Adam(lr=0.005).step() (S/utils.py:217), called at S/train.py:126.
#include "common.h"

namespace cvcs {

// torch.optim.SGD: g += wd*p; buf = g (first step) | mom*buf + g; p -= lr*buf
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                 int64_t n, float lr, float mom, float wd, float gs, int first) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 bv = first ? make_float4(0, 0, 0, 0) : reinterpret_cast<float4*>(buf)[i];
    float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w}, be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = ge[k] * gs + wd * pe[k];
      be[k] = first ? gg : mom * be[k] + gg;
      pe[k] = pe[k] - lr * be[k];
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    reinterpret_cast<float4*>(buf)[i] = make_float4(be[0], be[1], be[2], be[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float gg = g[i] * gs + wd * p[i];
    const float b = first ? gg : mom * buf[i] + gg;
    buf[i] = b;
    p[i] -= lr * b;
  }
}

// torch.optim.Adam (amsgrad off): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int64_t n, float step_size, float inv_sqrt_bc2,
                                                  float b1, float b2, float eps, float wd, float gs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gg = g[i] * gs;
    if (wd != 0.f) gg += wd * p[i];
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
    p[i] -= step_size * (mm / denom);
  }
}

}  // namespace cvcs

namespace cvcs {
// x *= *scalar unless *scalar == 1 (decided on the device: no host round trip, capturable) - the incoming gradient of the fused loss
__global__ __launch_bounds__(256) void scale_unless_one_kernel(float* x, int64_t n, const float* __restrict__ scalar) {
  const float v = *scalar;
  if (v == 1.0f) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= v;
}
__global__ __launch_bounds__(256) void scale_unless_one_bf16_kernel(bf16_t* x, int64_t n, const float* __restrict__ scalar) {
  const float v = *scalar;
  if (v == 1.0f) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = f32_to_bf16(bf16_to_f32(x[i]) * v);
}
}  // namespace cvcs

using namespace cvcs;

extern "C" int cvcs_scale_unless_one_bf16(void* x, int64_t n, const float* scalar_device, void* stream) {
  CVCS_CHECK_ARG(x && scalar_device && n > 0, "cvcs_scale_unless_one_bf16: bad argument");
  int64_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(scale_unless_one_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, n, scalar_device);
  CVCS_CHECK_LAUNCH("cvcs_scale_unless_one_bf16");
  return CVCS_OK;
}

extern "C" int cvcs_scale_unless_one(float* x, int64_t n, const float* scalar_device, void* stream) {
  CVCS_CHECK_ARG(x && scalar_device && n > 0, "cvcs_scale_unless_one: bad argument");
  int64_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(scale_unless_one_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, scalar_device);
  CVCS_CHECK_LAUNCH("cvcs_scale_unless_one");
  return CVCS_OK;
}

extern "C" int cvcs_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                             float weight_decay, float grad_scale, int first_step, void* stream) {
  CVCS_CHECK_ARG(p && g && momentum_buf && n > 0, "cvcs_sgd_step: null argument");
  CVCS_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)momentum_buf % 16) == 0,
                 "cvcs_sgd_step: buffers must be 16-byte aligned");
  int64_t blocks = cdiv(n / 4 + 1, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, momentum_buf, n, lr, momentum,
                     weight_decay, grad_scale, first_step);
  CVCS_CHECK_LAUNCH("cvcs_sgd_step");
  return CVCS_OK;
}

extern "C" int cvcs_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, float grad_scale, int step, void* stream) {
  CVCS_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "cvcs_adam_step: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  int64_t blocks = cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)(lr / bc1),
                     (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay, grad_scale);
  CVCS_CHECK_LAUNCH("cvcs_adam_step");
  return CVCS_OK;
}
