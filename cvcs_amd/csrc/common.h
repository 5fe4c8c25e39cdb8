// Shared helpers for libcvcs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/cvcs_hip.h"

namespace cvcs {

void set_error(const char* fmt, ...);

#define CVCS_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      ::cvcs::set_error(__VA_ARGS__);        \
      return CVCS_EINVAL;                    \
    }                                        \
  } while (0)

#define CVCS_CHECK_LAUNCH(name)                                                  \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      ::cvcs::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
      return CVCS_EHIP;                                                          \
    }                                                                            \
  } while (0)

typedef uint16_t bf16_t;  // storage type of a bfloat16
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the hardware convert (keeps NaN a NaN)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}

// two floats -> one dword of two bf16 (low half = a) with ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, h);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPer16 = 4;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  // unpack a 16-byte chunk into kPer16 floats / pack back
  __device__ static __forceinline__ void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct Elem<bf16_t> {
  static constexpr int kPer16 = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  __device__ static __forceinline__ void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    uint4 u;
    u.x = pack2_bf16(f[0], f[1]);
    u.y = pack2_bf16(f[2], f[3]);
    u.z = pack2_bf16(f[4], f[5]);
    u.w = pack2_bf16(f[6], f[7]);
    return u;
  }
};

// XCD-aware workgroup order (cdna_hip_programming.md T1): the dispatcher deals consecutive workgroups round-robin over the 8 XCDs, each
// with its own L2; the blocks with the same (id % 8) share an XCD.  Logical ids are handed out so that each XCD's blocks get a contiguous
// range: tiles that share an operand panel (consecutive logical ids) then run on ONE XCD, back to back, and the panel is an L2 hit
// instead of a separate fetch per XCD.  Bijective for any workgroup count.  A speed choice only: correctness never depends on placement.
__device__ __forceinline__ unsigned xcd_order(unsigned bid, unsigned nwg) {
  if (nwg < 16) return bid;
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- nn.Upsample(x2, bilinear) computed on the fly while a thin 3x3 kernel stages its input (conv3x3_thin_kernel<.., UP>, wgrad_thin_kernel<.., UP>):
// the 6 x 66-pixel halo of a 4 x 64 tile of the UP-SAMPLED map is interpolated LDS -> LDS from the 4 x 34-pixel patch of the low-resolution map
// under it (staged by LDS-DMA with clamped coordinates), with cvcs_upsample2x_fwd's own taps and expression and the same bf16 rounding - the
// matrix cores see the operand the stored up-sampled tensor would have given them.
__device__ __forceinline__ void up2_taps(int o, int n, int& i0, int& i1, float& w0, float& w1) {      // = up_taps of elementwise.hip
  const int k = o >> 1;
  if (o & 1) { i0 = k; i1 = k + 1 < n ? k + 1 : k; w0 = 0.75f; w1 = 0.25f; }
  else       { i0 = k > 0 ? k - 1 : 0; i1 = k; w0 = k > 0 ? 0.25f : 0.f; w1 = k > 0 ? 0.75f : 1.f; }
}
constexpr int kUpPatchRows = 4, kUpPatchCols = 34;      // low-resolution pixels under a 6 x 66 halo (tile origin a multiple of 4 x 64)
// halo pixel hp = hy * 66 + hx of the tile at (ty0, tx0), 16-byte chunk ch: interpolated from the patch in LDS (pixel-major rows of CI * 2 bytes)
template <int CI>
__device__ __forceinline__ uint4 up2_halo_chunk(const char* patch, int hp, int ch, int ty0, int tx0, int H, int W) {
  const int hy = hp / 66, hx = hp - hy * 66;
  const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
  if ((unsigned)gy >= (unsigned)H || (unsigned)gx >= (unsigned)W) return make_uint4(0u, 0u, 0u, 0u);      // the conv's zero padding
  int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
  up2_taps(gy, H >> 1, y0, y1, wy0, wy1);
  up2_taps(gx, W >> 1, x0, x1, wx0, wx1);
  const int ly0 = (ty0 >> 1) - 1, lx0 = (tx0 >> 1) - 1;        // patch origin (patch pixel (pr, pc) holds the low-resolution pixel clamped from (ly0 + pr, lx0 + pc))
  const int r0 = (y0 - ly0) * kUpPatchCols, r1 = (y1 - ly0) * kUpPatchCols, c0 = x0 - lx0, c1 = x1 - lx0;
  float a[8], bb[8], c[8], d[8], o[8];
  Elem<bf16_t>::unpack(*reinterpret_cast<const uint4*>(patch + (r0 + c0) * (CI * 2) + ch * 16), a);
  Elem<bf16_t>::unpack(*reinterpret_cast<const uint4*>(patch + (r0 + c1) * (CI * 2) + ch * 16), bb);
  Elem<bf16_t>::unpack(*reinterpret_cast<const uint4*>(patch + (r1 + c0) * (CI * 2) + ch * 16), c);
  Elem<bf16_t>::unpack(*reinterpret_cast<const uint4*>(patch + (r1 + c1) * (CI * 2) + ch * 16), d);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = wy0 * (wx0 * a[k] + wx1 * bb[k]) + wy1 * (wx0 * c[k] + wx1 * d[k]);
  return Elem<bf16_t>::pack(o);
}

// Workgroups of `fn` (block size, dynamic LDS) the whole chip holds at once: the grid of a persistent (grid-stride) kernel.  Falls back to
// `fallback` when there is no device to ask (the launch-plan dry runs of the CPU tests).
inline int resident_workgroups(const void* fn, int threads, size_t lds, int fallback = 1024) {
  int dev = 0, cus = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0 ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, threads, lds) != hipSuccess || per <= 0) {
    (void)hipGetLastError();
    return fallback;
  }
  return per * cus;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One LDS-DMA wave-instruction (global_load_lds_dwordx4: lane l's 16 bytes land at lds_dst + 16*l) issued from inline
// asm: the compiler then does not know a DMA is in flight and schedules its LDS fragment reads freely with its own
// lgkmcnt waits (seeing the DMA, it would drain vmcnt before every LDS read).  Completion is tracked by hand: a counted
// `s_waitcnt vmcnt(k)` + barrier before the data are read.  M0 (the LDS base) is saved and restored around the load.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst /* wave-uniform LDS byte address */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// ---- optional fp8 side output of a producer pass (the tensors the fp8 convolutions of conv_fp8.hip read): besides its bf16 output the
// pass writes the e4m3 / e5m2 image of THE SAME (bf16-rounded) values and feeds the tensor's amax - no separate quantisation sweep.
// slot = { amax, scale, 1 / scale, fmax } + the partial amax lines (cvcs_hip.h, kF8SlotFloats).  q == NULL: none.
struct Q8Out { char* q; int64_t ld; float* slot; int fmt; int take_amax; };

// eight bf16 values as stored (one 16-byte chunk) -> eight fp8 bytes; am <- max(am, |values|)
__device__ __forceinline__ uint2 q8_pack8(const uint4& packed, float inv, float fmax, int fmt, float& am) {
  float f[8];
  f[0] = __uint_as_float(packed.x << 16); f[1] = __uint_as_float(packed.x & 0xffff0000u);
  f[2] = __uint_as_float(packed.y << 16); f[3] = __uint_as_float(packed.y & 0xffff0000u);
  f[4] = __uint_as_float(packed.z << 16); f[5] = __uint_as_float(packed.z & 0xffff0000u);
  f[6] = __uint_as_float(packed.w << 16); f[7] = __uint_as_float(packed.w & 0xffff0000u);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    am = fmaxf(am, fabsf(f[k]));
    f[k] = fminf(fmaxf(f[k] * inv, -fmax), fmax);
  }
  unsigned lo = 0, hi = 0;
  if (fmt == CVCS_E4M3) {
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false); hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  } else {
    lo = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], lo, false); lo = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_bf8_f32(f[4], f[5], hi, false); hi = __builtin_amdgcn_cvt_pk_bf8_f32(f[6], f[7], hi, true);
  }
  return make_uint2(lo, hi);
}

// a thread's running amax -> the slot: wave maximum, then ONE atomic per wave into one of kF8Parts partial words, each on a 64-byte line of
// its own behind the slot's state (thousands of waves finishing together on ONE word queue at the memory side, ~12 ns each: measured
// +75 us per producer launch); cvcs_fp8_update_scales takes the maximum over the words.  atomicMax on the bits of a non-negative float is
// exact and order-independent, so the step stays bitwise reproducible.  Every lane of the wave must call it.
constexpr int kF8Parts = 64;            // partial amax words per slot
constexpr int kF8SlotFloats = 16 * (1 + kF8Parts);   // a slot: one 64-byte line of state + kF8Parts lines of one partial word each
__device__ __forceinline__ void q8_commit_amax(float am, float* slot) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
  if ((threadIdx.x & 63) == 0 && am == am && am > 0.f) {
    const unsigned w = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) + blockIdx.y * 7u) & (kF8Parts - 1);
    atomicMax(reinterpret_cast<unsigned*>(slot + 16 * (1 + w)), __float_as_uint(am));
  }
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace cvcs
