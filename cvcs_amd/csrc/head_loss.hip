// Classifier head (1x1 conv 64 -> NC), per-pixel softmax cross-entropy, argmax + confusion matrix (gfx950).
//
// All of these are HBM-bound per-pixel kernels (2*64*NC FLOP against (64+NC) elements per pixel): one lane owns
// one pixel, reads its 64 channels as whole 16-byte chunks, keeps the NC logits in registers, and the NCHW f32
// logits are written/read class-plane by class-plane so that a wave touches 256 contiguous bytes per class.
// Reductions over pixels (loss, sum of weights, dW of the head, confusion counts) are workgroup partials +
// a fixed-order finalize (loss, dW) or LDS-privatised integer histograms flushed with 64-bit atomics (counts).
//
// Reference ops replaced: nn.Conv2d(64, NC, 1) (S/nets.py:172), nn.CrossEntropyLoss(weight, ignore_index)
// (S/utils.py:230,238; S/train.py:122), torch.max/argmax (S/utils.py:90,158), MulticlassConfusionMatrix.update
// (S/utils.py:93-94) and the backward of the first two (S/train.py:125).
#include <stdlib.h>
#include "common.h"

namespace cvcs {

constexpr int kHeadC = 64;
constexpr int kMaxNC = 32;

template <typename T>
__device__ __forceinline__ void load_pixel64(const char* base, float* x) {
  constexpr int V = 16 / sizeof(T);
#pragma unroll
  for (int c = 0; c < kHeadC / V; ++c) Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + c * 16), x + c * V);
}

template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const char* x, int64_t x_ld, int64_t P, int64_t HW,
                                                      const float* __restrict__ w, const float* __restrict__ bias, int NC,
                                                      float* logits) {
  __shared__ float sw[kMaxNC * kHeadC + kMaxNC];
  for (int i = threadIdx.x; i < NC * kHeadC; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < NC; i += 256) sw[kMaxNC * kHeadC + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  for (int64_t pidx = (int64_t)blockIdx.x * 256 + threadIdx.x; pidx < P; pidx += (int64_t)gridDim.x * 256) {
    float xv[kHeadC];
    load_pixel64<T>(x + pidx * x_ld * sizeof(T), xv);
    const int64_t b = pidx / HW, hw = pidx - b * HW;
    for (int c = 0; c < NC; ++c) {
      float s = sw[kMaxNC * kHeadC + c];
#pragma unroll
      for (int k = 0; k < kHeadC; ++k) s += xv[k] * sw[c * kHeadC + k];
      logits[(b * NC + c) * HW + hw] = s;
    }
  }
}

// The last BatchNorm of the network is followed only by the 1x1 head: z = scale*r + shift feeds logits = W z + b, so
// logits = (W diag(scale)) r + (b + W shift) - the BN-apply pass over the largest activation is replaced by folding the
// affine into the NC x 64 head weights (forward), and by un-folding the weight gradient (backward):
//   dW[c][k] = sum_p dl[p][c] z[p][k] = scale[k] * (sum_p dl[p][c] r[p][k]) + shift[k] * db[c].
__global__ void head_fold_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ scale,
                                 const float* __restrict__ shift, int NC, float* w_out, float* bias_out) {
  const int c = blockIdx.x, k = threadIdx.x;   // one workgroup of 64 lanes per class
  const float wv = w[c * kHeadC + k];
  w_out[c * kHeadC + k] = wv * scale[k];
  const float t = wave_sum(wv * shift[k]);
  if (k == 0) bias_out[c] = (bias ? bias[c] : 0.f) + t;
}

__global__ void head_unfold_grad_kernel(float* dw, const float* __restrict__ db, const float* __restrict__ scale,
                                        const float* __restrict__ shift, int NC) {
  const int c = blockIdx.x, k = threadIdx.x;
  dw[c * kHeadC + k] = scale[k] * dw[c * kHeadC + k] + shift[k] * db[c];
}

// head + argmax in one pass (prediction: S/utils.py:88-90, S/inference.py:45-47 without the logits round trip): the NC
// logits of a pixel are formed exactly as in head_fwd_kernel (same order of operations -> the same labels as
// head_fwd + argmax), only the index of the first maximum leaves the kernel.
template <typename T>
__global__ __launch_bounds__(256) void head_argmax_kernel(const char* x, int64_t x_ld, int64_t P, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int NC, uint8_t* labels) {
  __shared__ float sw[kMaxNC * kHeadC + kMaxNC];
  for (int i = threadIdx.x; i < NC * kHeadC; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < NC; i += 256) sw[kMaxNC * kHeadC + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  for (int64_t pidx = (int64_t)blockIdx.x * 256 + threadIdx.x; pidx < P; pidx += (int64_t)gridDim.x * 256) {
    float xv[kHeadC];
    load_pixel64<T>(x + pidx * x_ld * sizeof(T), xv);
    float best = -INFINITY;
    int arg = 0;
    for (int c = 0; c < NC; ++c) {
      float s = sw[kMaxNC * kHeadC + c];
#pragma unroll
      for (int k = 0; k < kHeadC; ++k) s += xv[k] * sw[c * kHeadC + k];
      if (s > best || c == 0) { best = s; arg = c; }   // strict '>' keeps the FIRST maximum
    }
    labels[pidx] = (uint8_t)arg;
  }
}

// Head backward, one pass over (x, dl) per 64-pixel tile:
//   partial dW[c][k] = sum_p dl[p][c] * x[p][k], db[c] = sum_p dl[p][c]  (one partial row per workgroup:
//                      part[row][c*64 + k] for c < NC, then part[row][NC*64 + c]; fixed-order finalize elsewhere)
//   dx[p][k]         = sum_c dl[p][c] * w[c][k]                         (written as whole NHWC pixel rows)
// Both are contractions on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32).  The tile is staged in LDS (dl as
// [class][pixel], x as f32 [pixel][channel]; pitches 130 / 80 floats keep the ds_read_b32 fragment reads conflict-free).
// dW: A = dl^T [class][pixel], B = x [pixel][channel], wave w owns channels 16w..16w+15.
// dx: computed TRANSPOSED, A = w^T [channel][class] (held in registers for the whole kernel), B = dl [class][pixel], so a
// lane ends up with four consecutive channels of one pixel - one packed 8/16-byte write into the staging tile (the x
// tile's LDS, free by then), from which the 64 pixel rows leave as contiguous 16-byte-per-lane stores.
constexpr int kDwTile = 64;   // pixels staged per iteration
constexpr int kSdPitch = 130, kSxPitch = 80;
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const char* x, int64_t x_ld, const float* __restrict__ dl, int64_t P,
                                                      int64_t HW, const float* __restrict__ w, int NC, char* dx, int64_t dx_ld,
                                                      float* part) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  constexpr int OROW = kHeadC * ES + 16;          // staging row: one pixel's 64 channels + 16 bytes against bank conflicts
  static_assert(kDwTile * OROW <= kDwTile * kSxPitch * 4, "the dx staging tile reuses the x tile");
  __shared__ __attribute__((aligned(16))) float sx[kDwTile * kSxPitch];
  __shared__ float sd[kMaxNC * kSdPitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int ncb = (NC + 15) / 16;                  // class blocks of dW
  const int nks = (NC + 3) / 4;                    // class k-steps of dx
  for (int i = tid; i < kMaxNC * kSdPitch; i += 256) sd[i] = 0.f;   // class rows >= NC stay zero
  float wa[kMaxNC / 4];                            // A fragments of dx: w[class 4s+fg][channel 16*wave+fr]
#pragma unroll
  for (int s = 0; s < kMaxNC / 4; ++s) {
    const int c = 4 * s + fg;
    wa[s] = c < NC ? w[c * kHeadC + wave * 16 + fr] : 0.f;
  }
  f32x4 acc[kMaxNC / 16];
#pragma unroll
  for (int cb = 0; cb < kMaxNC / 16; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float accb[kMaxNC / 4];                          // db partials: this wave's lanes cover class wave + 4k
#pragma unroll
  for (int k = 0; k < kMaxNC / 4; ++k) accb[k] = 0.f;
  char* stage = reinterpret_cast<char*>(sx);
  for (int64_t p0 = (int64_t)blockIdx.x * kDwTile; p0 < P; p0 += (int64_t)gridDim.x * kDwTile) {
    __syncthreads();
    for (int id = tid; id < kDwTile * (kHeadC / V); id += 256) {
      const int r = id / (kHeadC / V), c = id - r * (kHeadC / V);
      float f[V];
      if (p0 + r < P) Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + (p0 + r) * x_ld * ES + c * 16), f);
      else
#pragma unroll
        for (int q = 0; q < V; ++q) f[q] = 0.f;
#pragma unroll
      for (int q = 0; q < V; ++q) sx[r * kSxPitch + c * V + q] = f[q];
    }
#pragma unroll
    for (int k = 0; k < kMaxNC / 4; ++k) {
      const int c = wave + 4 * k;                  // id = tid + 256k -> class id / 64 = wave + 4k, pixel lane
      if (c < NC) {
        float v = 0.f;
        if (p0 + lane < P) {
          const int64_t pidx = p0 + lane, b = pidx / HW, hw = pidx - b * HW;
          v = dl[(b * NC + c) * HW + hw];
        }
        sd[c * kSdPitch + lane] = v;
        accb[k] += v;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < kDwTile / 4; ++s) {
      const float bv = sx[(4 * s + fg) * kSxPitch + wave * 16 + fr];
#pragma unroll
      for (int cb = 0; cb < kMaxNC / 16; ++cb)
        if (cb < ncb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(sd[(cb * 16 + fr) * kSdPitch + 4 * s + fg], bv, acc[cb], 0, 0, 0);
    }
    f32x4 dxa[kDwTile / 16];
#pragma unroll
    for (int nb = 0; nb < kDwTile / 16; ++nb) {
      dxa[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < kMaxNC / 4; ++s)
        if (s < nks) dxa[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], sd[(4 * s + fg) * kSdPitch + nb * 16 + fr], dxa[nb], 0, 0, 0);
    }
    __syncthreads();                               // every wave is done reading the x tile
#pragma unroll
    for (int nb = 0; nb < kDwTile / 16; ++nb) {    // D layout: row (channel) = fg*4 + r, col (pixel) = fr
      char* dst = stage + (nb * 16 + fr) * OROW + (wave * 16 + fg * 4) * ES;
      if constexpr (ES == 2) {
        uint2 u;
        u.x = pack2_bf16(dxa[nb][0], dxa[nb][1]);
        u.y = pack2_bf16(dxa[nb][2], dxa[nb][3]);
        *reinterpret_cast<uint2*>(dst) = u;
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(dxa[nb][0], dxa[nb][1], dxa[nb][2], dxa[nb][3]);
      }
    }
    __syncthreads();
    constexpr int CPR = kHeadC * ES / 16;          // 16-byte chunks per pixel row
    for (int id = tid; id < kDwTile * CPR; id += 256) {
      const int r = id / CPR, c = id - r * CPR;
      if (p0 + r < P)
        *reinterpret_cast<uint4*>(dx + (p0 + r) * dx_ld * ES + c * 16) = *reinterpret_cast<const uint4*>(stage + r * OROW + c * 16);
    }
  }
  float* row = part + (int64_t)blockIdx.x * (NC * kHeadC + NC);
#pragma unroll
  for (int cb = 0; cb < kMaxNC / 16; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = cb * 16 + fg * 4 + r;   // D layout: row (class) = fg*4 + r, col (channel) = fr
      if (cb < ncb && c < NC) row[c * kHeadC + wave * 16 + fr] = acc[cb][r];
    }
#pragma unroll
  for (int k = 0; k < kMaxNC / 4; ++k) {
    const float sum = wave_sum(accb[k]);
    if (lane == 0 && wave + 4 * k < NC) row[NC * kHeadC + wave + 4 * k] = sum;
  }
}

// bf16 variant of the pass above (the throughput path).  The x tile stays bf16 in LDS as it is in memory (a 16-byte copy per
// lane instead of eight 4-byte LDS writes), dW runs on the bf16 MFMA: B = x [32 pixels][16 channels] by transposed LDS
// reads (ds_read_b64_tr_b16), A = dl^T [16 classes][32 pixels] from the f32 dl tile in LDS (two 16-byte reads per lane and
// K-step, in exactly the pixel order the transposed reads deliver), split into a bf16 head and a bf16 remainder (dl = hi + lo
// to 2^-17: two MFMAs, products exact to that) - x is exactly bf16 already.  dx runs on the bf16 MFMA too, with both of ITS
// operands split (below).
constexpr int kXPitch = kHeadC * 2 + 16;   // bytes: a pixel row of the bf16 x / dx tile
constexpr int kSdPitch4 = 68;              // floats: a class row of the dl tile (16-byte aligned rows, conflict-free b128 reads)
constexpr int kSdpPitch = kMaxNC + 4;      // floats: a pixel row of the pixel-major copy of the dl tile (16-byte aligned rows)
__global__ __launch_bounds__(256) void head_bwd_bf16_kernel(const char* x, int64_t x_ld, const float* __restrict__ dl, int64_t P,
                                                           int64_t HW, const float* __restrict__ w, int NC, char* dx,
                                                           int64_t dx_ld, float* part) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  __shared__ __attribute__((aligned(16))) char sx[kDwTile * kXPitch];
  __shared__ __attribute__((aligned(16))) float sd[kMaxNC * kSdPitch4];
  __shared__ __attribute__((aligned(16))) float sdp[kDwTile * kSdpPitch];   // the same dl tile, pixel-major (dx B operand)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int ncb = (NC + 15) / 16;
  for (int i = tid; i < kMaxNC * kSdPitch4; i += 256) sd[i] = 0.f;   // class rows >= NC stay zero
  for (int i = tid; i < kDwTile * kSdpPitch; i += 256) sdp[i] = 0.f;
  // dx = w^T dl on the bf16 MFMA with both operands split into a bf16 head and a bf16 remainder (exact to 2^-17, the
  // result is stored as bf16): one K = 32 instruction contracts 16 classes TWICE - k-groups 0,1 hold the heads of dl,
  // k-groups 2,3 the remainders of the same classes, against the same w values - so w_head x (dl_head + dl_rem) is one
  // MFMA and w_rem x (...) a second (the exact-f32 MFMA took 16 instructions of 32 cycles per 64-pixel tile, and a
  // third of this kernel's time).  A fragment: lane (fr = channel, fg) holds classes cb*16 + (fg & 1)*8 + 0..7.
  bf16x8 wah[kMaxNC / 16], wal[kMaxNC / 16];
#pragma unroll
  for (int cb = 0; cb < kMaxNC / 16; ++cb) {
    float f[8], hf[8], rem[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cb * 16 + (fg & 1) * 8 + i;
      f[i] = c < NC ? w[c * kHeadC + wave * 16 + fr] : 0.f;
    }
    const uint4 head = Elem<bf16_t>::pack(f);
    Elem<bf16_t>::unpack(head, hf);
#pragma unroll
    for (int i = 0; i < 8; ++i) rem[i] = f[i] - hf[i];
    wah[cb] = __builtin_bit_cast(bf16x8, head);
    wal[cb] = __builtin_bit_cast(bf16x8, Elem<bf16_t>::pack(rem));
  }
  f32x4 acc[kMaxNC / 16];
#pragma unroll
  for (int cb = 0; cb < kMaxNC / 16; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float accb[kMaxNC / 4];                          // db partials: this wave's lanes cover class wave + 4k
#pragma unroll
  for (int k = 0; k < kMaxNC / 4; ++k) accb[k] = 0.f;
  const int q = fr >> 2, pp = fr & 3;
  const char* xfrag = sx + (wave * 16) * 2 + pp * 8;      // transposed-read base: channel block of this wave
  for (int64_t p0 = (int64_t)blockIdx.x * kDwTile; p0 < P; p0 += (int64_t)gridDim.x * kDwTile) {
    // a 64-pixel tile never straddles two images (HW is a multiple of 64) nor the end of the tensor
    const int64_t b = p0 / HW, hw0 = p0 - b * HW;
    const float* dlb = dl + b * NC * HW + hw0;
    __syncthreads();
    for (int id = tid; id < kDwTile * 8; id += 256) {
      const int r = id >> 3, c = id & 7;
      *reinterpret_cast<uint4*>(sx + r * kXPitch + c * 16) = *reinterpret_cast<const uint4*>(x + (p0 + r) * x_ld * 2 + c * 16);
    }
#pragma unroll
    for (int k = 0; k < kMaxNC / 4; ++k) {
      const int c = wave + 4 * k;                  // item tid + 256k: class wave + 4k, pixel lane
      if (c < NC) {
        const float v = dlb[(int64_t)c * HW + lane];
        sd[c * kSdPitch4 + lane] = v;
        sdp[lane * kSdpPitch + c] = v;
        accb[k] += v;
      }
    }
    __syncthreads();
    // dW: A = dl^T [16 classes][32 pixels] from the LDS tile, element i of k-group fg <-> pixel 4fg+i | 16+4fg+i-4 (the
    // order the transposed x reads deliver), split into bf16 head + remainder; B = x [32 pixels][16 channels]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = ks * 32 + 4 * fg + q, r1 = r0 + 16;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xfrag + r0 * kXPitch));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xfrag + r1 * kXPitch));
      const bf16x8 xf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int cb = 0; cb < kMaxNC / 16; ++cb)
        if (cb < ncb) {
          const float4 a0 = *reinterpret_cast<const float4*>(sd + (cb * 16 + fr) * kSdPitch4 + ks * 32 + 4 * fg);
          const float4 a1 = *reinterpret_cast<const float4*>(sd + (cb * 16 + fr) * kSdPitch4 + ks * 32 + 16 + 4 * fg);
          const float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const uint4 head = Elem<bf16_t>::pack(f);
          float hf[8], rem[8];
          Elem<bf16_t>::unpack(head, hf);
#pragma unroll
          for (int k = 0; k < 8; ++k) rem[k] = f[k] - hf[k];
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, head), xf, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Elem<bf16_t>::pack(rem)), xf, acc[cb], 0, 0, 0);
        }
    }
    // dx = w^T dl (see the A fragments above): B fragment of lane (fr = pixel, fg) = eight classes of its pixel from the
    // pixel-major tile, heads in k-groups 0,1 and remainders in k-groups 2,3
    f32x4 dxa[kDwTile / 16];
#pragma unroll
    for (int nb = 0; nb < kDwTile / 16; ++nb) {
      dxa[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cb = 0; cb < kMaxNC / 16; ++cb)
        if (cb < ncb) {
          const float* src = sdp + (nb * 16 + fr) * kSdpPitch + cb * 16 + (fg & 1) * 8;
          const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
          const float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const uint4 head = Elem<bf16_t>::pack(f);
          float hf[8], rem[8];
          Elem<bf16_t>::unpack(head, hf);
#pragma unroll
          for (int k = 0; k < 8; ++k) rem[k] = f[k] - hf[k];
          const uint4 lo = Elem<bf16_t>::pack(rem);
          const bool hi_lane = fg < 2;
          const uint4 sel = make_uint4(hi_lane ? head.x : lo.x, hi_lane ? head.y : lo.y, hi_lane ? head.z : lo.z, hi_lane ? head.w : lo.w);
          const bf16x8 bfrag = __builtin_bit_cast(bf16x8, sel);
          dxa[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wah[cb], bfrag, dxa[nb], 0, 0, 0);
          dxa[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wal[cb], bfrag, dxa[nb], 0, 0, 0);
        }
    }
    __syncthreads();                               // every wave is done reading the x tile: reuse it as the dx staging tile
#pragma unroll
    for (int nb = 0; nb < kDwTile / 16; ++nb) {    // D layout: row (channel) = fg*4 + r, col (pixel) = fr
      uint2 u;
      u.x = pack2_bf16(dxa[nb][0], dxa[nb][1]);
      u.y = pack2_bf16(dxa[nb][2], dxa[nb][3]);
      *reinterpret_cast<uint2*>(sx + (nb * 16 + fr) * kXPitch + (wave * 16 + fg * 4) * 2) = u;
    }
    __syncthreads();
    for (int id = tid; id < kDwTile * 8; id += 256) {
      const int r = id >> 3, c = id & 7;
      *reinterpret_cast<uint4*>(dx + (p0 + r) * dx_ld * 2 + c * 16) = *reinterpret_cast<const uint4*>(sx + r * kXPitch + c * 16);
    }
  }
  float* row = part + (int64_t)blockIdx.x * (NC * kHeadC + NC);
#pragma unroll
  for (int cb = 0; cb < kMaxNC / 16; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = cb * 16 + fg * 4 + r;   // D layout: row (class) = fg*4 + r, col (channel) = fr
      if (cb < ncb && c < NC) row[c * kHeadC + wave * 16 + fr] = acc[cb][r];
    }
#pragma unroll
  for (int k = 0; k < kMaxNC / 4; ++k) {
    const float sum = wave_sum(accb[k]);
    if (lane == 0 && wave + 4 * k < NC) row[NC * kHeadC + wave + 4 * k] = sum;
  }
}

// ------------------------------------------------------------------------------------------------ cross entropy
// workspace layout (floats): [0] = sum of target weights, [1] = loss numerator, [2..2+R) partial weights,
// [2+R..2+2R) partial numerators, R = ce_rows(P).
__host__ __device__ inline int ce_rows(int64_t P) {
  int64_t r = (P + 1023) / 1024;
  return (int)(r < 1 ? 1 : (r > 4096 ? 4096 : r));
}

__device__ __forceinline__ int load_target(const void* t, int is_u8, int64_t i) {
  return is_u8 ? (int)reinterpret_cast<const uint8_t*>(t)[i] : (int)reinterpret_cast<const int64_t*>(t)[i];
}

__device__ __forceinline__ float block_sum_256(float v, float* sbuf) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sbuf[wave] = v;
  __syncthreads();
  return sbuf[0] + sbuf[1] + sbuf[2] + sbuf[3];
}

__global__ __launch_bounds__(256) void ce_weight_kernel(const void* target, int is_u8, int64_t P, int NC,
                                                       const float* __restrict__ cw, int ignore, float* ws) {
  __shared__ float sbuf[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
    const int t = load_target(target, is_u8, i);
    if (t != ignore && (unsigned)t < (unsigned)NC) s += cw ? cw[t] : 1.f;
  }
  s = block_sum_256(s, sbuf);
  if (threadIdx.x == 0) ws[2 + blockIdx.x] = s;
}

__global__ void ce_reduce_kernel(float* ws, int R, int which, float* loss_out) {
  // which 0: ws[0] = sum of partial weights.  which 1: ws[1] = numerator; loss_out = ws[1]/ws[0]
  // one wave; fixed lane-strided order + xor-tree in f64 (deterministic)
  double a = 0.0;
  const float* part = ws + 2 + (which ? R : 0);
  for (int r = threadIdx.x; r < R; r += 64) a += (double)part[r];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (threadIdx.x != 0) return;
  ws[which] = (float)a;
  if (which) *loss_out = (float)(a / (double)ws[0]);
}

template <int NCMAX>
__global__ __launch_bounds__(256) void ce_main_kernel(const float* __restrict__ logits, const void* target, int is_u8,
                                                     int64_t P, int64_t HW, int NC, const float* __restrict__ cw, int ignore,
                                                     float grad_scale, float* dlogits, float* ws, int R) {
  __shared__ float sbuf[4];
  const float den = ws[0];
  // every pixel ignored: the loss is 0/0 = NaN as in torch, but its gradient is ZERO (F.cross_entropy with an all-ignored
  // target returns grad == 0; the reference's step S/train.py:121-126 then leaves the weights finite)
  const float gmul = den > 0.f ? grad_scale / den : 0.f;
  float num = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HW, hw = i - b * HW;
    const float* zp = logits + b * NC * HW + hw;
    const int t = load_target(target, is_u8, i);
    const bool valid = (t != ignore) && ((unsigned)t < (unsigned)NC);
    const float wt = valid ? (cw ? cw[t] : 1.f) : 0.f;
    float z[NCMAX];
    float m = -INFINITY, zt = 0.f;
#pragma unroll
    for (int c = 0; c < NCMAX; ++c)
      if (c < NC) { z[c] = zp[c * HW]; m = fmaxf(m, z[c]); zt = (c == t) ? z[c] : zt; }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < NCMAX; ++c)
      if (c < NC) { z[c] = expf(z[c] - m); se += z[c]; }   // z[c] now holds exp(z_c - max): reused by the gradient
    const float lse = m + logf(se);
    const float inv_se = 1.f / se;
    if (valid) num += wt * (lse - zt);
    if (dlogits) {
      float* dp = dlogits + b * NC * HW + hw;
      const float f = wt * gmul;
#pragma unroll
      for (int c = 0; c < NCMAX; ++c)
        if (c < NC) dp[c * HW] = valid ? f * (z[c] * inv_se - ((c == t) ? 1.f : 0.f)) : 0.f;   // (0 * inf/NaN logits stay 0)
    }
  }
  num = block_sum_256(num, sbuf);
  if (threadIdx.x == 0) ws[2 + R + blockIdx.x] = num;
}

// ------------------------------------------------------------------------------------------------ argmax + confusion
template <int NCMAX>
__global__ __launch_bounds__(256) void argmax_conf_kernel(const float* __restrict__ logits, int64_t P, int64_t HW, int NC,
                                                         uint8_t* labels, const void* target, int is_u8, int ignore, int K,
                                                         unsigned long long* conf) {
  __shared__ unsigned int hist[kMaxNC * kMaxNC];
  const bool do_conf = conf != nullptr;
  if (do_conf) {
    for (int i = threadIdx.x; i < K * K; i += 256) hist[i] = 0;
    __syncthreads();
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HW, hw = i - b * HW;
    const float* zp = logits + b * NC * HW + hw;
    float best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < NCMAX; ++c)
      if (c < NC) {
        const float v = zp[c * HW];
        if (v > best || (c == 0)) { best = v; arg = c; }   // strict '>' keeps the FIRST maximum
      }
    if (labels) labels[i] = (uint8_t)arg;
    if (do_conf) {
      const int t = load_target(target, is_u8, i);
      if (t != ignore && (unsigned)t < (unsigned)K && arg < K) atomicAdd(&hist[t * K + arg], 1u);
    }
  }
  if (do_conf) {
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += 256)
      if (hist[i]) atomicAdd(&conf[i], (unsigned long long)hist[i]);
  }
}

// confusion matrix from label maps (networks that return labels, not logits: the Ensemble, S/utils.py:89-91)
__global__ __launch_bounds__(256) void label_conf_kernel(const uint8_t* __restrict__ pred, const void* target, int is_u8,
                                                        int64_t P, int ignore, int K, unsigned long long* conf) {
  __shared__ unsigned int hist[kMaxNC * kMaxNC];
  for (int i = threadIdx.x; i < K * K; i += 256) hist[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
    const int t = load_target(target, is_u8, i);
    const int a = pred[i];
    if (t != ignore && (unsigned)t < (unsigned)K && a < K) atomicAdd(&hist[t * K + a], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K; i += 256)
    if (hist[i]) atomicAdd(&conf[i], (unsigned long long)hist[i]);
}

// per-pixel majority vote over M label maps [M][P]; ties -> the smallest label (torch.mode, S/utils.py:504-506)
constexpr int kMaxVoters = 16;
__global__ __launch_bounds__(256) void vote_kernel(const uint8_t* __restrict__ labels, int M, int64_t P, uint8_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
    int v[kMaxVoters];
#pragma unroll
    for (int m = 0; m < kMaxVoters; ++m) v[m] = m < M ? labels[(int64_t)m * P + i] : -1;
    int best = 256, best_n = 0;
#pragma unroll
    for (int a = 0; a < kMaxVoters; ++a) {
      if (a >= M) continue;
      int n = 0;
#pragma unroll
      for (int b = 0; b < kMaxVoters; ++b) n += (b < M && v[b] == v[a]) ? 1 : 0;
      if (n > best_n || (n == best_n && v[a] < best)) { best_n = n; best = v[a]; }
    }
    out[i] = (uint8_t)best;
  }
}

// ------------------------------------------------------------------------------------------------ inference boundary
// crop n tiles (S x S, origin = tile origin - (S-p)/2, zero outside the image) out of one u8 CHW image
__global__ __launch_bounds__(256) void crop_tiles_kernel(const uint8_t* __restrict__ src, int C, int H, int W, uint8_t* dst,
                                                        int n, int first_tile, int tiles_per_row, int p, int S) {
  const int off = -(S - p) / 2;
  const int64_t total = (int64_t)n * C * S * S;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int x = (int)(id % S);
    int64_t t = id / S;
    const int y = (int)(t % S); t /= S;
    const int c = (int)(t % C);
    const int tile = first_tile + (int)(t / C);
    const int ty = tile / tiles_per_row, tx = tile - ty * tiles_per_row;
    const int sy = ty * p + off + y, sx = tx * p + off + x;
    dst[id] = ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) ? src[((int64_t)c * H + sy) * W + sx] : (uint8_t)0;
  }
}

// tile producer: gather n tiles (S x S) at arbitrary origins (int32 [n][2] = (top, left), e.g. randomly shifted grid
// positions) out of one resident u8 CHW image / HW index mask; zero outside the image
__global__ __launch_bounds__(256) void gather_tiles_kernel(const uint8_t* __restrict__ src, int C, int H, int W,
                                                          const int* __restrict__ origins, int n, int S, uint8_t* dst) {
  const int64_t total = (int64_t)n * C * S * S;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int x = (int)(id % S);
    int64_t t = id / S;
    const int y = (int)(t % S); t /= S;
    const int c = (int)(t % C);
    const int ti = (int)(t / C);
    const int sy = origins[2 * ti] + y, sx = origins[2 * ti + 1] + x;
    dst[id] = ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) ? src[((int64_t)c * H + sy) * W + sx] : (uint8_t)0;
  }
}

// per-class pixel counts of a u8 label image (Loader._get_class_count, S/dataset.py:346-358): LDS-privatised histogram
__global__ __launch_bounds__(256) void label_histogram_kernel(const uint8_t* __restrict__ lab, int64_t n, int K,
                                                             unsigned long long* counts) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) atomicAdd(&h[lab[i]], 1u);
  __syncthreads();
  if ((int)threadIdx.x < K && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// argmax over classes of the centre p x p window of every tile, palette lookup, write into the stitched maps
template <int NCMAX>
__global__ __launch_bounds__(256) void argmax_stitch_kernel(const float* __restrict__ logits, int n, int NC, int S, int p,
                                                           int first_tile, int tiles_per_row, const uint8_t* __restrict__ palette,
                                                           uint8_t* rgb, uint8_t* labels, int Hout, int Wout) {
  const int m = (S - p) / 2;
  const int64_t total = (int64_t)n * p * p;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int x = (int)(id % p);
    int64_t t = id / p;
    const int y = (int)(t % p);
    const int ti = (int)(t / p);
    const float* zp = logits + (((int64_t)ti * NC) * S + (y + m)) * S + (x + m);
    float best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < NCMAX; ++c)
      if (c < NC) {
        const float v = zp[(int64_t)c * S * S];
        if (v > best || c == 0) { best = v; arg = c; }
      }
    const int tile = first_tile + ti;
    const int ty = tile / tiles_per_row, tx = tile - ty * tiles_per_row;
    const int oy = ty * p + y, ox = tx * p + x;
    if (oy < Hout && ox < Wout) {
      const int64_t o = (int64_t)oy * Wout + ox;
      if (labels) labels[o] = (uint8_t)arg;
      if (rgb) { rgb[o * 3] = palette[arg * 3]; rgb[o * 3 + 1] = palette[arg * 3 + 1]; rgb[o * 3 + 2] = palette[arg * 3 + 2]; }
    }
  }
}

// the same stitching from per-tile LABEL maps [n][S][S] (fused head+argmax output, or an Ensemble's vote)
__global__ __launch_bounds__(256) void label_stitch_kernel(const uint8_t* __restrict__ tiles, int n, int S, int p, int first_tile,
                                                          int tiles_per_row, const uint8_t* __restrict__ palette, uint8_t* rgb,
                                                          uint8_t* labels, int Hout, int Wout) {
  const int m = (S - p) / 2;
  const int64_t total = (int64_t)n * p * p;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int x = (int)(id % p);
    int64_t t = id / p;
    const int y = (int)(t % p);
    const int ti = (int)(t / p);
    const int arg = tiles[((int64_t)ti * S + (y + m)) * S + (x + m)];
    const int tile = first_tile + ti;
    const int ty = tile / tiles_per_row, tx = tile - ty * tiles_per_row;
    const int oy = ty * p + y, ox = tx * p + x;
    if (oy < Hout && ox < Wout) {
      const int64_t o = (int64_t)oy * Wout + ox;
      if (labels) labels[o] = (uint8_t)arg;
      if (rgb) { rgb[o * 3] = palette[arg * 3]; rgb[o * 3 + 1] = palette[arg * 3 + 1]; rgb[o * 3 + 2] = palette[arg * 3 + 2]; }
    }
  }
}

static inline int pix_grid(int64_t P, int cap = 256 * 16) {
  int64_t g = cdiv(P, 256);
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}


// ------------------------------------------------------------------------------------------------ thin heads (C = 16 | 32 input channels)
// The light U-Net decoders end in 16 (or 32) channels at full resolution (decoder widths 256/128/64/32/16): the head is then even more
// HBM-bound than the 64-channel one (2*C*NC FLOP against C + NC elements per pixel) and runs on the VALU.  Forward / argmax: one lane
// owns one pixel, weights in LDS.  Backward: 256-pixel tiles staged in LDS (x as f32 [pixel][C], dl as [class][pixel]); dx per pixel by
// its lane; dW: each thread owns (class, channel) pairs and sums over the tile's pixels; one partial row per workgroup in the layout of
// the 64-channel head (part[row][c*C + k], then part[row][NC*C + c]) for the same fixed-order finalize.
template <typename T, int C>
__device__ __forceinline__ void load_pixel_c(const char* base, float* x) {
  constexpr int V = 16 / sizeof(T);
#pragma unroll
  for (int c = 0; c < C / V; ++c) Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + c * 16), x + c * V);
}

template <typename T, int C>
__global__ __launch_bounds__(256) void thin_head_fwd_kernel(const char* x, int64_t x_ld, int64_t P, int64_t HW, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int NC, float* logits, uint8_t* labels) {
  __shared__ float sw[kMaxNC * C + kMaxNC];
  for (int i = threadIdx.x; i < NC * C; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < NC; i += 256) sw[kMaxNC * C + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  for (int64_t pidx = (int64_t)blockIdx.x * 256 + threadIdx.x; pidx < P; pidx += (int64_t)gridDim.x * 256) {
    float xv[C];
    load_pixel_c<T, C>(x + pidx * x_ld * sizeof(T), xv);
    const int64_t b = pidx / HW, hw = pidx - b * HW;
    float best = -INFINITY;
    int arg = 0;
    for (int c = 0; c < NC; ++c) {
      float s = sw[kMaxNC * C + c];
#pragma unroll
      for (int k = 0; k < C; ++k) s += xv[k] * sw[c * C + k];
      if (logits) logits[(b * NC + c) * HW + hw] = s;
      if (s > best || c == 0) { best = s; arg = c; }   // strict '>' keeps the FIRST maximum
    }
    if (labels) labels[pidx] = (uint8_t)arg;
  }
}

constexpr int kThinTile = 256;
template <typename T, int C>
__global__ __launch_bounds__(256) void thin_head_bwd_kernel(const char* x, int64_t x_ld, const float* __restrict__ dl, int64_t P, int64_t HW,
                                                           const float* __restrict__ w, int NC, char* dx, int64_t dx_ld, float* part) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  constexpr int XP = C + 4;                       // pitch of the x tile (floats): 16-byte aligned rows, the +4 keeps pixel lanes on different banks
  constexpr int KG = C / 4;                       // 4-channel groups
  __shared__ __attribute__((aligned(16))) float sx[kThinTile * XP];
  __shared__ float sd[kMaxNC * kThinTile];
  __shared__ float sw[kMaxNC * C];
  const int tid = threadIdx.x;
  for (int i = tid; i < NC * C; i += 256) sw[i] = w[i];
  // dW: a thread owns one (class, 4-channel group) block and every Q-th pixel of a tile: ONE dl read (shared by the block's 4 FMAs) and
  // ONE 16-byte x read per pixel; the Q pixel phases of a block are summed at the end.  nblk = NC * C / 4 <= 256.
  const int nblk = NC * KG;
  const int Q = 256 / nblk;                       // pixel phases (>= 1)
  const int blk = tid % nblk, q = tid / nblk;     // threads beyond Q * nblk idle in the dW part
  const int bc = blk / KG, bk = (blk - bc * KG) * 4;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, ab = 0.f;
  const int64_t ntiles = (P + kThinTile - 1) / kThinTile;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t pidx = t * kThinTile + tid;
    const bool ok = pidx < P;
    __syncthreads();                              // the previous tile's readers are done
    float xv[C];
    if (ok) load_pixel_c<T, C>(x + pidx * x_ld * ES, xv);
#pragma unroll
    for (int k = 0; k < C; k += 4)
      *reinterpret_cast<float4*>(&sx[tid * XP + k]) = ok ? make_float4(xv[k], xv[k + 1], xv[k + 2], xv[k + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t b = ok ? pidx / HW : 0, hw = ok ? pidx - b * HW : 0;
    // the NC class planes of this pixel: eight loads in flight at a time (a loop of load -> LDS store pays the HBM latency per class)
    float dlv[kMaxNC];
#pragma unroll
    for (int c0 = 0; c0 < kMaxNC; c0 += 8) {
      if (c0 < NC) {
#pragma unroll
        for (int c = c0; c < c0 + 8; ++c) dlv[c] = (ok && c < NC) ? dl[(b * NC + c) * HW + hw] : 0.f;
#pragma unroll
        for (int c = c0; c < c0 + 8; ++c)
          if (c < NC) sd[c * kThinTile + tid] = dlv[c];
      }
    }
    __syncthreads();
    if (ok) {                                     // dx[p][k] = sum_c dl[p][c] w[c][k]  (w by wave-uniform scalar loads, dl from registers)
      float o[C];
#pragma unroll
      for (int k = 0; k < C; ++k) o[k] = 0.f;
#pragma unroll
      for (int c = 0; c < kMaxNC; ++c) {
        if (c < NC) {
#pragma unroll
          for (int k = 0; k < C; ++k) o[k] += dlv[c] * w[c * C + k];
        }
      }
#pragma unroll
      for (int k = 0; k < C / V; ++k) *reinterpret_cast<uint4*>(dx + pidx * dx_ld * ES + k * 16) = Elem<T>::pack(o + k * V);
    }
    if (q < Q) {
      for (int pp = q; pp < kThinTile; pp += Q) {
        const float d = sd[bc * kThinTile + pp];
        const float4 xx = *reinterpret_cast<const float4*>(&sx[pp * XP + bk]);
        a0 += d * xx.x; a1 += d * xx.y; a2 += d * xx.z; a3 += d * xx.w;
        ab += d;
      }
    }
  }
  // sum the Q pixel phases of every block through LDS (the x tile is free), fixed order
  __syncthreads();
  float* red = sx;                                // [Q][nblk][5]
  if (q < Q) {
    float* r = red + ((int64_t)q * nblk + blk) * 5;
    r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3; r[4] = ab;
  }
  __syncthreads();
  float* row = part + (int64_t)blockIdx.x * (NC * C + NC);
  if (tid < nblk) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, sb = 0.f;
    for (int j = 0; j < Q; ++j) {
      const float* r = red + ((int64_t)j * nblk + tid) * 5;
      s0 += r[0]; s1 += r[1]; s2 += r[2]; s3 += r[3]; sb += r[4];
    }
    row[bc * C + bk] = s0; row[bc * C + bk + 1] = s1; row[bc * C + bk + 2] = s2; row[bc * C + bk + 3] = s3;
    if (bk == 0) row[NC * C + bc] = sb;
  }
}


// ===================================================================================================================
// 1x1 head + softmax cross-entropy + their backward in ONE pass over the last decoder activation (bf16, C = 16 / 32 / 64 channels,
// NC <= 32 classes): S/train.py:121-123 (`net(x)` ... `criterion(...)` ... `loss.backward()`) without the [B, NC, H, W] f32 logits and their
// gradient ever touching HBM - per step of the 32 x 512^2 headline that is 2 x 537 MB written and 2 x 537 MB read back.
//
// Everything is a chain of v_mfma_f32_16x16x16_bf16 tiles over 16 pixels whose register layouts feed each other without leaving the lane:
//   S^T [class][pixel] = W x^T            A = W rows (hi + lo bf16 parts of the f32 weights: exact to 2^-17), B = x^T: lane (pixel l%16,
//                                         channels 4g..4g+3) IS an 8-byte global load; D: lane (pixel l%16) holds classes 4g..4g+3
//   softmax over the classes of a pixel = the lane's 4 values x the 4 lane groups (two xor-shuffles); loss, dl = (p - onehot) w_t / den
//   dx^T [channel][pixel] = W^T dl^T      B = dl^T: k = class 4g..4g+3 of pixel l%16 = exactly the D layout above (hi + lo parts);
//                                         D: lane (pixel l%16, channels 4g..4g+3) = an 8-byte global store
//   dW [class][channel] += dl^T x         K = pixels: both operands transposed through a 512-byte LDS tile per wave (ds_read_b64_tr_b16)
// f32 accumulation everywhere; partial dW | db rows per workgroup and the loss numerator per workgroup are reduced in fixed order.
typedef short bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mma16(const uint2& a, const uint2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4_t, a), __builtin_bit_cast(bf16x4_t, b), c, 0, 0, 0);
}
// four f32 -> the bf16 heads and the bf16 remainders (v = hi + lo to 2^-17 relative)
__device__ __forceinline__ void split4(const float* v, uint2& hi, uint2& lo) {
  float h[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bf16_t b = f32_to_bf16(v[k]);
    h[k] = v[k];
    l[k] = v[k] - bf16_to_f32(b);
  }
  hi.x = pack2_bf16(h[0], h[1]); hi.y = pack2_bf16(h[2], h[3]);
  lo.x = pack2_bf16(l[0], l[1]); lo.y = pack2_bf16(l[2], l[3]);
}
__device__ __forceinline__ uint2 tr_read16(unsigned addr) {
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}

template <int C, int NT>
__global__ __launch_bounds__(256) void head_ce_kernel(const char* __restrict__ x, int64_t x_ld, int64_t P, const float* __restrict__ w,
                                                     const float* __restrict__ wdx, const float* __restrict__ bias, int NC, const void* target, int is_u8,
                                                     const float* __restrict__ cw, int ignore, float grad_scale, char* dx, int64_t dx_ld,
                                                     float* part_dw, float* ws, int R) {
  constexpr int CT = C / 16;
  constexpr int ROW = NT * 16 * C + NT * 16;          // floats of a wave's partial row in LDS: dW[NT*16][C] | db[NT*16]
  __shared__ __attribute__((aligned(16))) char tiles[4][(CT + 2 * NT) * 512];
  __shared__ float comb[4][ROW];
  __shared__ float sbuf[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const unsigned lds_w = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)tiles[wave];
  const unsigned wr_off = fr * 32 + g * 8;                          // this lane's 8 bytes of a [16 pixels][32 B] tile
  const unsigned tr_off = (4 * g + (fr >> 2)) * 32 + (fr & 3) * 8;  // transposed read: column fr, rows 4g..4g+3

  // ---- weight fragments (constant for the kernel)
  uint2 Whi[NT][CT], Wlo[NT][CT], WThi[CT][NT], WTlo[CT][NT];
  f32x4 b0[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float v[4], t[4];
      const int n = nt * 16 + fr;                  // A of S^T: row = class fr, k = channels 4g..4g+3
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = n < NC ? w[(int64_t)n * C + ct * 16 + 4 * g + k] : 0.f;
      split4(v, Whi[nt][ct], Wlo[nt][ct]);
      const int c = ct * 16 + fr;                  // A of dx^T: row = channel fr, k = classes 4g..4g+3
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int nn = nt * 16 + 4 * g + k; t[k] = nn < NC ? wdx[(int64_t)nn * C + c] : 0.f; }
      split4(t, WThi[ct][nt], WTlo[ct][nt]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int n = nt * 16 + 4 * g + r; b0[nt][r] = n < NC ? bias[n] : 0.f; }
  }
  const float den = ws[0];
  const float gmul = den > 0.f ? grad_scale / den : 0.f;      // (an all-ignored batch: NaN loss, ZERO gradient - as torch)

  f32x4 dW[NT][CT];
  float db[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dW[nt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) db[nt][r] = 0.f;
  }
  float num = 0.f;

  const int64_t ntiles = (P + 15) / 16;
  const int64_t tstep = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  // software pipeline: the loads of tile i + 1 are in flight while tile i is computed
  uint2 bx[CT], bxn[CT];
  int tg = 0, tgn = 0;
  auto fetch = [&](int64_t tl, uint2* bxx, int& tt) {
    int64_t p = tl * 16 + fr; p = p < P ? p : P - 1;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bxx[ct] = *reinterpret_cast<const uint2*>(x + (p * x_ld + ct * 16 + 4 * g) * 2);
    tt = load_target(target, is_u8, p);
  };
  if (tile < ntiles) fetch(tile, bx, tg);
  for (; tile < ntiles; tile += tstep) {
    if (tile + tstep < ntiles) fetch(tile + tstep, bxn, tgn);
    const int64_t p = tile * 16 + fr;
    const bool pok = p < P;
    // ---- logits (transposed): lane = pixel fr, classes nt*16 + 4g + r
    f32x4 S[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      S[nt] = b0[nt];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        S[nt] = mma16(Whi[nt][ct], bx[ct], S[nt]);
        S[nt] = mma16(Wlo[nt][ct], bx[ct], S[nt]);
      }
    }
    const int t = tg;
    const bool valid = pok && (t != ignore) && ((unsigned)t < (unsigned)NC);
    const float wt = valid ? (cw ? cw[t] : 1.f) : 0.f;
    float m = -INFINITY, zt = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nt * 16 + 4 * g + r;
        if (n < NC) { m = fmaxf(m, S[nt][r]); zt = (n == t) ? S[nt][r] : zt; }
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
    zt += __shfl_xor(zt, 16, 64); zt += __shfl_xor(zt, 32, 64);
    float se = 0.f;
    f32x4 e[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nt * 16 + 4 * g + r;
        e[nt][r] = n < NC ? expf(S[nt][r] - m) : 0.f;
        se += e[nt][r];
      }
    se += __shfl_xor(se, 16, 64); se += __shfl_xor(se, 32, 64);
    const float lse = m + logf(se), inv_se = 1.f / se;
    if (valid && g == 0) num += wt * (lse - zt);
    // ---- dl = (softmax - onehot) * w_t / den, split into bf16 head + remainder
    const float f = wt * gmul;
    uint2 dlh[NT], dll[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float d[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nt * 16 + 4 * g + r;
        d[r] = (valid && n < NC) ? f * (e[nt][r] * inv_se - ((n == t) ? 1.f : 0.f)) : 0.f;
        db[nt][r] += d[r];
      }
      split4(d, dlh[nt], dll[nt]);
    }
    // ---- dx (transposed): lane = pixel fr, channels ct*16 + 4g + r
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        a = mma16(WThi[ct][nt], dlh[nt], a);
        a = mma16(WThi[ct][nt], dll[nt], a);
        a = mma16(WTlo[ct][nt], dlh[nt], a);
      }
      if (pok) {
        uint2 o;
        o.x = pack2_bf16(a[0], a[1]); o.y = pack2_bf16(a[2], a[3]);
        *reinterpret_cast<uint2*>(dx + (p * dx_ld + ct * 16 + 4 * g) * 2) = o;
      }
    }
    // ---- dW += dl^T x: both operands through this wave's LDS tiles, read back transposed (K = the 16 pixels)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the previous tile's transposed reads are done)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const uint2 xv = pok ? bx[ct] : make_uint2(0u, 0u);
      asm volatile("ds_write_b64 %0, %1" : : "v"(lds_w + ct * 512 + wr_off), "v"(xv) : "memory");
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      asm volatile("ds_write_b64 %0, %1" : : "v"(lds_w + (CT + 2 * nt) * 512 + wr_off), "v"(dlh[nt]) : "memory");
      asm volatile("ds_write_b64 %0, %1" : : "v"(lds_w + (CT + 2 * nt + 1) * 512 + wr_off), "v"(dll[nt]) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint2 xt[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) xt[ct] = tr_read16(lds_w + ct * 512 + tr_off);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const uint2 ah = tr_read16(lds_w + (CT + 2 * nt) * 512 + tr_off), al = tr_read16(lds_w + (CT + 2 * nt + 1) * 512 + tr_off);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        dW[nt][ct] = mma16(ah, xt[ct], dW[nt][ct]);
        dW[nt][ct] = mma16(al, xt[ct], dW[nt][ct]);
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bx[ct] = bxn[ct];
    tg = tgn;
  }

  // ---- per-workgroup partial row [NC*C | NC]: the four waves merged in wave order.  dW D layout: lane (channel ct*16 + fr, classes nt*16+4g+r)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) comb[wave][(nt * 16 + 4 * g + r) * C + ct * 16 + fr] = dW[nt][ct][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = db[nt][r];                        // over the 16 pixels (lanes) of the group
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
      if (fr == 0) comb[wave][NT * 16 * C + nt * 16 + 4 * g + r] = s;
    }
  }
  num = block_sum_256(num, sbuf);                 // (contains the barrier that publishes comb)
  if (tid == 0) ws[2 + R + blockIdx.x] = num;
  float* row = part_dw + (int64_t)blockIdx.x * (NC * C + NC);
  for (int i = tid; i < NC * C + NC; i += 256) {
    const int src = i < NC * C ? i : NT * 16 * C + (i - NC * C);
    row[i] = (comb[0][src] + comb[1][src]) + (comb[2][src] + comb[3][src]);
  }
}

}  // namespace cvcs

using namespace cvcs;

static bool thin_c(int C) { return C == 16 || C == 32; }

extern "C" int cvcs_head_fwd(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, const float* bias, int NC,
                             float* logits, int dtype, void* stream) {
  CVCS_CHECK_ARG(dtype == CVCS_F32 || dtype == CVCS_BF16, "cvcs_head_fwd: bad dtype");
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(x && w && logits && B > 0 && H > 0 && W > 0, "cvcs_head_fwd: null argument");
  CVCS_CHECK_ARG(C == kHeadC || thin_c(C), "cvcs_head_fwd: head is built for %d (S/nets.py:172), 32 or 16 input channels, got %d", kHeadC, C);
  CVCS_CHECK_ARG(NC >= 1 && NC <= kMaxNC, "cvcs_head_fwd: NC=%d out of [1,%d]", NC, kMaxNC);
  CVCS_CHECK_ARG(((uintptr_t)x % 16) == 0 && (x_ld * es) % 16 == 0 && x_ld >= C, "cvcs_head_fwd: x view alignment");
  const int64_t P = (int64_t)B * H * W, HW = (int64_t)H * W;
  dim3 grid(pix_grid(P));
  if (thin_c(C)) {
#define THIN_FWD(TT, CC) hipLaunchKernelGGL((thin_head_fwd_kernel<TT, CC>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, P, HW, w, bias, NC, logits, (uint8_t*)nullptr)
    if (dtype == CVCS_F32) { if (C == 16) THIN_FWD(float, 16); else THIN_FWD(float, 32); }
    else { if (C == 16) THIN_FWD(bf16_t, 16); else THIN_FWD(bf16_t, 32); }
#undef THIN_FWD
    CVCS_CHECK_LAUNCH("cvcs_head_fwd(thin)");
    return CVCS_OK;
  }
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((head_fwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, P, HW, w, bias, NC, logits);
  else
    hipLaunchKernelGGL((head_fwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, P, HW, w, bias, NC, logits);
  CVCS_CHECK_LAUNCH("cvcs_head_fwd");
  return CVCS_OK;
}

extern "C" int cvcs_head_fold(const float* w, const float* bias, const float* scale, const float* shift, int NC, float* w_out,
                              float* bias_out, void* stream) {
  CVCS_CHECK_ARG(w && scale && shift && w_out && bias_out && NC >= 1 && NC <= kMaxNC, "cvcs_head_fold: bad argument");
  hipLaunchKernelGGL(head_fold_kernel, dim3(NC), dim3(kHeadC), 0, (hipStream_t)stream, w, bias, scale, shift, NC, w_out, bias_out);
  CVCS_CHECK_LAUNCH("cvcs_head_fold");
  return CVCS_OK;
}

extern "C" int cvcs_head_unfold_grad(float* dw, const float* db, const float* scale, const float* shift, int NC, void* stream) {
  CVCS_CHECK_ARG(dw && db && scale && shift && NC >= 1 && NC <= kMaxNC, "cvcs_head_unfold_grad: bad argument");
  hipLaunchKernelGGL(head_unfold_grad_kernel, dim3(NC), dim3(kHeadC), 0, (hipStream_t)stream, dw, db, scale, shift, NC);
  CVCS_CHECK_LAUNCH("cvcs_head_unfold_grad");
  return CVCS_OK;
}

extern "C" int cvcs_head_argmax(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, const float* bias, int NC,
                                uint8_t* labels, int dtype, void* stream) {
  CVCS_CHECK_ARG(dtype == CVCS_F32 || dtype == CVCS_BF16, "cvcs_head_argmax: bad dtype");
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(x && w && labels && B > 0 && H > 0 && W > 0, "cvcs_head_argmax: null argument");
  CVCS_CHECK_ARG(C == kHeadC || thin_c(C), "cvcs_head_argmax: head is built for %d, 32 or 16 input channels, got %d", kHeadC, C);
  CVCS_CHECK_ARG(NC >= 1 && NC <= kMaxNC, "cvcs_head_argmax: NC=%d out of [1,%d]", NC, kMaxNC);
  CVCS_CHECK_ARG(((uintptr_t)x % 16) == 0 && (x_ld * es) % 16 == 0 && x_ld >= C, "cvcs_head_argmax: view alignment");
  const int64_t P = (int64_t)B * H * W;
  if (thin_c(C)) {
    const int64_t HW = (int64_t)H * W;
#define THIN_ARG(TT, CC) hipLaunchKernelGGL((thin_head_fwd_kernel<TT, CC>), dim3(pix_grid(P)), dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, P, HW, w, bias, NC, (float*)nullptr, labels)
    if (dtype == CVCS_F32) { if (C == 16) THIN_ARG(float, 16); else THIN_ARG(float, 32); }
    else { if (C == 16) THIN_ARG(bf16_t, 16); else THIN_ARG(bf16_t, 32); }
#undef THIN_ARG
    CVCS_CHECK_LAUNCH("cvcs_head_argmax(thin)");
    return CVCS_OK;
  }
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((head_argmax_kernel<float>), dim3(pix_grid(P)), dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, P, w, bias, NC, labels);
  else
    hipLaunchKernelGGL((head_argmax_kernel<bf16_t>), dim3(pix_grid(P)), dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, P, w, bias, NC, labels);
  CVCS_CHECK_LAUNCH("cvcs_head_argmax");
  return CVCS_OK;
}

extern "C" int cvcs_label_stitch(const uint8_t* tiles, int n, int S, int p, int first_tile, int tiles_per_row, const uint8_t* palette,
                                 uint8_t* rgb, uint8_t* labels, int Hout, int Wout, void* stream) {
  CVCS_CHECK_ARG(tiles && n > 0 && S >= p && p > 0 && (S - p) % 2 == 0 && tiles_per_row > 0 && first_tile >= 0 && Hout > 0 &&
                     Wout > 0, "cvcs_label_stitch: bad argument");
  CVCS_CHECK_ARG((rgb == nullptr || palette != nullptr) && (rgb || labels), "cvcs_label_stitch: rgb needs a palette; nothing to write");
  hipLaunchKernelGGL(label_stitch_kernel, dim3(pix_grid((int64_t)n * p * p, 4096)), dim3(256), 0, (hipStream_t)stream, tiles, n, S, p,
                     first_tile, tiles_per_row, palette, rgb, labels, Hout, Wout);
  CVCS_CHECK_LAUNCH("cvcs_label_stitch");
  return CVCS_OK;
}

extern "C" int cvcs_head_bwd_rows(int64_t P) {
  static const int cap = getenv("CVCS_HEAD_ROWS") ? atoi(getenv("CVCS_HEAD_ROWS")) : 1024;   // tuning knob: resident workgroups
  int64_t r = cdiv(P, kDwTile);
  return (int)(r < 1 ? 1 : (r > cap ? cap : r));
}

extern "C" int cvcs_head_bwd(const void* x, int64_t x_ld, const float* dlogits, int B, int H, int W, int C, const float* w, int NC,
                             void* dx, int64_t dx_ld, float* part_dw, int dtype, void* stream) {
  CVCS_CHECK_ARG(dtype == CVCS_F32 || dtype == CVCS_BF16, "cvcs_head_bwd: bad dtype");
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(x && dlogits && w && dx && part_dw && B > 0 && H > 0 && W > 0, "cvcs_head_bwd: null argument");
  CVCS_CHECK_ARG(C == kHeadC || thin_c(C), "cvcs_head_bwd: head is built for %d, 32 or 16 input channels, got %d", kHeadC, C);
  CVCS_CHECK_ARG(NC >= 1 && NC <= kMaxNC, "cvcs_head_bwd: NC=%d out of [1,%d]", NC, kMaxNC);
  CVCS_CHECK_ARG(((uintptr_t)x % 16) == 0 && (x_ld * es) % 16 == 0 && x_ld >= C && ((uintptr_t)dx % 16) == 0 &&
                     (dx_ld * es) % 16 == 0 && dx_ld >= C, "cvcs_head_bwd: view alignment");
  const int64_t P = (int64_t)B * H * W, HW = (int64_t)H * W;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cvcs_head_bwd_rows(P));
  if (thin_c(C)) {
#define THIN_BWD(TT, CC) hipLaunchKernelGGL((thin_head_bwd_kernel<TT, CC>), grid, dim3(256), 0, st, (const char*)x, x_ld, dlogits, P, HW, w, NC, (char*)dx, dx_ld, part_dw)
    if (dtype == CVCS_F32) { if (C == 16) THIN_BWD(float, 16); else THIN_BWD(float, 32); }
    else { if (C == 16) THIN_BWD(bf16_t, 16); else THIN_BWD(bf16_t, 32); }
#undef THIN_BWD
    CVCS_CHECK_LAUNCH("cvcs_head_bwd(thin)");
    return CVCS_OK;
  }
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((head_bwd_kernel<float>), grid, dim3(256), 0, st, (const char*)x, x_ld, dlogits, P, HW, w, NC, (char*)dx, dx_ld, part_dw);
  else if (HW % kDwTile == 0)   // tiles of 64 pixels inside one image (every tile side that is a multiple of 16)
    hipLaunchKernelGGL(head_bwd_bf16_kernel, grid, dim3(256), 0, st, (const char*)x, x_ld, dlogits, P, HW, w, NC, (char*)dx, dx_ld, part_dw);
  else
    hipLaunchKernelGGL((head_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)x, x_ld, dlogits, P, HW, w, NC, (char*)dx, dx_ld, part_dw);
  CVCS_CHECK_LAUNCH("cvcs_head_bwd");
  return CVCS_OK;
}

extern "C" int cvcs_ce_workspace_floats(int64_t P) { return 2 + 2 * ce_rows(P); }

extern "C" int cvcs_ce_weight_sum(const void* target, int target_is_u8, int B, int NC, int64_t HW, const float* class_weight,
                                  int ignore_index, float* workspace, void* stream) {
  CVCS_CHECK_ARG(target && workspace && B > 0 && HW > 0, "cvcs_ce_weight_sum: null argument");
  CVCS_CHECK_ARG(NC >= 1 && NC <= kMaxNC, "cvcs_ce_weight_sum: NC=%d out of [1,%d]", NC, kMaxNC);
  const int64_t P = (int64_t)B * HW;
  const int R = ce_rows(P);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_weight_kernel, dim3(R), dim3(256), 0, st, target, target_is_u8, P, NC, class_weight, ignore_index, workspace);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(64), 0, st, workspace, R, 0, (float*)nullptr);
  CVCS_CHECK_LAUNCH("cvcs_ce_weight_sum");
  return CVCS_OK;
}

extern "C" int cvcs_ce_fwd_bwd(const float* logits, const void* target, int target_is_u8, int B, int NC, int64_t HW,
                               const float* class_weight, int ignore_index, float grad_scale, float* loss_out, float* dlogits,
                               float* workspace, int external_denominator, void* stream) {
  CVCS_CHECK_ARG(logits && target && loss_out && workspace && B > 0 && HW > 0, "cvcs_ce_fwd_bwd: null argument");
  CVCS_CHECK_ARG(NC >= 1 && NC <= kMaxNC, "cvcs_ce_fwd_bwd: NC=%d out of [1,%d]", NC, kMaxNC);
  const int64_t P = (int64_t)B * HW;
  const int R = ce_rows(P);
  hipStream_t st = (hipStream_t)stream;
  if (!external_denominator) {
    hipLaunchKernelGGL(ce_weight_kernel, dim3(R), dim3(256), 0, st, target, target_is_u8, P, NC, class_weight, ignore_index, workspace);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(64), 0, st, workspace, R, 0, loss_out);
  }
  if (NC <= 8)
    hipLaunchKernelGGL((ce_main_kernel<8>), dim3(R), dim3(256), 0, st, logits, target, target_is_u8, P, HW, NC, class_weight, ignore_index, grad_scale, dlogits, workspace, R);
  else if (NC <= 16)
    hipLaunchKernelGGL((ce_main_kernel<16>), dim3(R), dim3(256), 0, st, logits, target, target_is_u8, P, HW, NC, class_weight, ignore_index, grad_scale, dlogits, workspace, R);
  else
    hipLaunchKernelGGL((ce_main_kernel<32>), dim3(R), dim3(256), 0, st, logits, target, target_is_u8, P, HW, NC, class_weight, ignore_index, grad_scale, dlogits, workspace, R);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(64), 0, st, workspace, R, 1, loss_out);
  CVCS_CHECK_LAUNCH("cvcs_ce_fwd_bwd");
  return CVCS_OK;
}

extern "C" int cvcs_argmax_confusion(const float* logits, int B, int NC, int64_t HW, uint8_t* labels, const void* target,
                                     int target_is_u8, int ignore_index, int K, int64_t* conf, void* stream) {
  CVCS_CHECK_ARG(logits && B > 0 && HW > 0, "cvcs_argmax_confusion: null argument");
  CVCS_CHECK_ARG(NC >= 1 && NC <= kMaxNC, "cvcs_argmax_confusion: NC=%d out of [1,%d]", NC, kMaxNC);
  CVCS_CHECK_ARG(labels || conf, "cvcs_argmax_confusion: nothing to produce");
  if (conf) CVCS_CHECK_ARG(target && K >= 1 && K <= kMaxNC, "cvcs_argmax_confusion: confusion needs targets and K in [1,%d]", kMaxNC);
  const int64_t P = (int64_t)B * HW;
  dim3 grid(pix_grid(P, 1024));
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* c = reinterpret_cast<unsigned long long*>(conf);
  if (NC <= 8)
    hipLaunchKernelGGL((argmax_conf_kernel<8>), grid, dim3(256), 0, st, logits, P, HW, NC, labels, target, target_is_u8, ignore_index, K, c);
  else if (NC <= 16)
    hipLaunchKernelGGL((argmax_conf_kernel<16>), grid, dim3(256), 0, st, logits, P, HW, NC, labels, target, target_is_u8, ignore_index, K, c);
  else
    hipLaunchKernelGGL((argmax_conf_kernel<32>), grid, dim3(256), 0, st, logits, P, HW, NC, labels, target, target_is_u8, ignore_index, K, c);
  CVCS_CHECK_LAUNCH("cvcs_argmax_confusion");
  return CVCS_OK;
}

extern "C" int cvcs_label_confusion(const uint8_t* pred, const void* target, int target_is_u8, int64_t P, int ignore_index, int K,
                                    int64_t* conf, void* stream) {
  CVCS_CHECK_ARG(pred && target && conf && P > 0, "cvcs_label_confusion: null argument");
  CVCS_CHECK_ARG(K >= 1 && K <= kMaxNC, "cvcs_label_confusion: K=%d out of [1,%d]", K, kMaxNC);
  hipLaunchKernelGGL(label_conf_kernel, dim3(pix_grid(P, 1024)), dim3(256), 0, (hipStream_t)stream, pred, target, target_is_u8, P,
                     ignore_index, K, reinterpret_cast<unsigned long long*>(conf));
  CVCS_CHECK_LAUNCH("cvcs_label_confusion");
  return CVCS_OK;
}

extern "C" int cvcs_vote_labels(const uint8_t* labels, int M, int64_t P, uint8_t* out, void* stream) {
  CVCS_CHECK_ARG(labels && out && P > 0, "cvcs_vote_labels: null argument");
  CVCS_CHECK_ARG(M >= 1 && M <= kMaxVoters, "cvcs_vote_labels: M=%d voters out of [1,%d]", M, kMaxVoters);
  hipLaunchKernelGGL(vote_kernel, dim3(pix_grid(P, 2048)), dim3(256), 0, (hipStream_t)stream, labels, M, P, out);
  CVCS_CHECK_LAUNCH("cvcs_vote_labels");
  return CVCS_OK;
}

extern "C" int cvcs_crop_tiles(const uint8_t* src, int C, int H, int W, uint8_t* dst, int n, int first_tile, int tiles_per_row,
                               int p, int S, void* stream) {
  CVCS_CHECK_ARG(src && dst && C > 0 && H > 0 && W > 0 && n > 0 && tiles_per_row > 0 && p > 0 && S >= p && (S - p) % 2 == 0 &&
                     first_tile >= 0, "cvcs_crop_tiles: bad argument");
  hipLaunchKernelGGL(crop_tiles_kernel, dim3(pix_grid((int64_t)n * C * S * S, 4096)), dim3(256), 0, (hipStream_t)stream, src, C, H,
                     W, dst, n, first_tile, tiles_per_row, p, S);
  CVCS_CHECK_LAUNCH("cvcs_crop_tiles");
  return CVCS_OK;
}

extern "C" int cvcs_argmax_stitch(const float* logits, int n, int NC, int S, int p, int first_tile, int tiles_per_row,
                                  const uint8_t* palette, uint8_t* rgb, uint8_t* labels, int Hout, int Wout, void* stream) {
  CVCS_CHECK_ARG(logits && n > 0 && S >= p && p > 0 && (S - p) % 2 == 0 && tiles_per_row > 0 && first_tile >= 0 && Hout > 0 &&
                     Wout > 0, "cvcs_argmax_stitch: bad argument");
  CVCS_CHECK_ARG(NC >= 1 && NC <= kMaxNC, "cvcs_argmax_stitch: NC=%d out of [1,%d]", NC, kMaxNC);
  CVCS_CHECK_ARG((rgb == nullptr || palette != nullptr) && (rgb || labels), "cvcs_argmax_stitch: rgb needs a palette; nothing to write");
  dim3 grid(pix_grid((int64_t)n * p * p, 4096));
  hipStream_t st = (hipStream_t)stream;
  if (NC <= 8)
    hipLaunchKernelGGL((argmax_stitch_kernel<8>), grid, dim3(256), 0, st, logits, n, NC, S, p, first_tile, tiles_per_row, palette, rgb, labels, Hout, Wout);
  else if (NC <= 16)
    hipLaunchKernelGGL((argmax_stitch_kernel<16>), grid, dim3(256), 0, st, logits, n, NC, S, p, first_tile, tiles_per_row, palette, rgb, labels, Hout, Wout);
  else
    hipLaunchKernelGGL((argmax_stitch_kernel<32>), grid, dim3(256), 0, st, logits, n, NC, S, p, first_tile, tiles_per_row, palette, rgb, labels, Hout, Wout);
  CVCS_CHECK_LAUNCH("cvcs_argmax_stitch");
  return CVCS_OK;
}

extern "C" int cvcs_gather_tiles(const uint8_t* src, int C, int H, int W, const int32_t* origins, int n, int S, uint8_t* dst,
                                 void* stream) {
  CVCS_CHECK_ARG(src && dst && origins && C > 0 && H > 0 && W > 0 && n > 0 && S > 0, "cvcs_gather_tiles: bad argument");
  hipLaunchKernelGGL(gather_tiles_kernel, dim3(pix_grid((int64_t)n * C * S * S, 4096)), dim3(256), 0, (hipStream_t)stream, src, C,
                     H, W, origins, n, S, dst);
  CVCS_CHECK_LAUNCH("cvcs_gather_tiles");
  return CVCS_OK;
}

extern "C" int cvcs_label_histogram(const uint8_t* labels, int64_t n, int K, int64_t* counts, void* stream) {
  CVCS_CHECK_ARG(labels && counts && n > 0 && K >= 1 && K <= 256, "cvcs_label_histogram: bad argument");
  hipLaunchKernelGGL(label_histogram_kernel, dim3(pix_grid(n, 1024)), dim3(256), 0, (hipStream_t)stream, labels, n, K,
                     reinterpret_cast<unsigned long long*>(counts));
  CVCS_CHECK_LAUNCH("cvcs_label_histogram");
  return CVCS_OK;
}

// the partial dW | db rows cvcs_head_ce writes (one per workgroup), also the row count of its loss-numerator partials
extern "C" int cvcs_head_ce_rows(int64_t P) { return ce_rows(P); }

extern "C" int cvcs_head_ce(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, const float* w_dx, const float* bias, int NC,
                            const void* target, int target_is_u8, const float* class_weight, int ignore_index, float grad_scale,
                            void* dx, int64_t dx_ld, float* part_dw, float* workspace, float* loss_out, int dtype, void* stream) {
  CVCS_CHECK_ARG(x && w && bias && target && dx && part_dw && workspace && loss_out, "cvcs_head_ce: null argument");
  CVCS_CHECK_ARG(dtype == CVCS_BF16, "cvcs_head_ce: built for bf16 activations (the f32 parity path keeps its logits)");
  CVCS_CHECK_ARG(C == 16 || C == 32 || C == 64, "cvcs_head_ce: C=%d (16 | 32 | 64 input channels)", C);
  CVCS_CHECK_ARG(NC >= 1 && NC <= kMaxNC, "cvcs_head_ce: NC=%d out of [1,%d]", NC, kMaxNC);
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && x_ld >= C && dx_ld >= C && (x_ld * 2) % 8 == 0 && (dx_ld * 2) % 8 == 0 &&
                 ((uintptr_t)x % 8) == 0 && ((uintptr_t)dx % 8) == 0, "cvcs_head_ce: views");
  const int64_t P = (int64_t)B * H * W;
  const int R = ce_rows(P);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_weight_kernel, dim3(R), dim3(256), 0, st, target, target_is_u8, P, NC, class_weight, ignore_index, workspace);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(64), 0, st, workspace, R, 0, loss_out);
#define LAUNCH_HC(C_, NT_)                                                                                                               \
  hipLaunchKernelGGL((head_ce_kernel<C_, NT_>), dim3(R), dim3(256), 0, st, (const char*)x, x_ld, P, w, w_dx ? w_dx : w, bias, NC, target, target_is_u8,    \
                     class_weight, ignore_index, grad_scale, (char*)dx, dx_ld, part_dw, workspace, R)
  if (NC <= 16) { if (C == 16) LAUNCH_HC(16, 1); else if (C == 32) LAUNCH_HC(32, 1); else LAUNCH_HC(64, 1); }
  else          { if (C == 16) LAUNCH_HC(16, 2); else if (C == 32) LAUNCH_HC(32, 2); else LAUNCH_HC(64, 2); }
#undef LAUNCH_HC
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(64), 0, st, workspace, R, 1, loss_out);
  CVCS_CHECK_LAUNCH("cvcs_head_ce");
  return CVCS_OK;
}
