// fp8 3x3 / stride 1 / pad 1 convolution for gfx950 on the block-scaled matrix instruction
// v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 rate; the plain 16x16x32 fp8 form runs AT the bf16 rate) - the
// "mixed bf16/fp8 convs" of BASELINE.json configs[4] (ResNet50-UNet 1024x1024).  The reference has no such path (it is f32
// throughout, S/train.py:121); the op replaced is the same nn.Conv2d k3 p1 (S/blocks.py:3-4) and the input-gradient half of
// its convolution_backward, for the layers whose channel counts are multiples of 128.
//
// Numerics: x8 = e4m3(x / s_x) (forward) or e5m2(dy / s_dy) (data gradient), w8 = e4m3(w / s_w) with PER-TENSOR scales
// (delayed scaling: s = 2 * amax(previous step) / fmax, cvcs_fp8_update_scales); products are exact in the MFMA, accumulation
// f32, the E8M0 block scales of the instruction are all 2^0 and the epilogue multiplies by s_x * s_w.  Output bf16.
//
// Geometry (one workgroup = one 16x16 output tile of ONE image x 128 output channels, 4 waves as 2 (eight image rows each) x 2
// (64-channel halves): one wave per SIMD with the whole 512-register file - two waves per SIMD leave 256 registers for 64
// accumulators and 8-register operand fragments, which spilled - and one workgroup per CU):
//   * a K-slice is 128 input channels = ONE MFMA K-step; LDS rows are 128 bytes (one pixel / one output channel x 128 fp8).
//   * the 18x18-pixel halo of a slice is staged ONCE (41 LDS-DMA pieces of 8 rows) and serves all nine taps; weight tiles
//     (128 x 128 B per tap) stream through a 3-slot ring, two taps ahead.  Counted vmcnt + raw barrier, never drained.
//   * lane (fr, fg) of an operand fragment takes the 16-byte chunks fg and fg + 4 of its row (the K order is the same
//     permutation on both operands, so the contraction is unchanged); chunk c of the row with swizzle key r is stored at
//     c ^ (((r >> 1) & 3) << 1).  A ds_read_b128 lane group (8 lanes of fg = 2m, 8 of fg = 2m + 1, 16 consecutive rows
//     together) then covers all 16 slots of the 256-byte bank row: bit 0 of the chunk separates the two fg, the swizzle
//     spreads the four rows of equal parity - conflict-free for ANY first halo column.
//   * the step barrier sits in the MIDDLE of a step: fragments of tap t+1 are read right behind it and are in flight under
//     the second half of tap t's MFMAs, so no wave starts a step with an empty matrix pipe.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace cvcs {

struct Conv8Args {
  const char* in;
  const char* wt;
  char* out;
  const float* scale_in; const float* scale_w;
  const float* pre_scale; const float* pre_shift;
  float* stat_sum; float* stat_m2; float* stat_cnt;
  int64_t in_ld, out_ld;
  int B, H, W, Cin, Cout;
  int relu;
  int tiles_x, tiles_y;
};

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ uint4 g_zero16_f8;   // zero word: LDS-DMA source of the padding ring

template <int I, int N, typename F>
__device__ __forceinline__ void static_for8(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for8<I + 1, N>(f);
  }
}

template <int K>
__device__ __forceinline__ void wait_vm_barrier8() {
  static_assert(K >= 0 && K < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" : : "n"(K) : "memory");
}

constexpr int kF8BN = 128, kF8NW = 4, kF8HS = 18, kF8HROWS = 324, kF8HPIECES = 41;
constexpr int kF8ABytes = kF8HPIECES * 1024;       // 41984: one halo buffer (328 rows of 128 B)
constexpr int kF8Slot = kF8BN * 128;               // 16384: one tap's weight tile
constexpr int kF8Lds = 2 * kF8ABytes + 3 * kF8Slot;   // 133120

template <int BFMT>   // format of the pixel operand: 0 = e4m3 (forward), 1 = e5m2 (data gradient); weights are e4m3
__global__ __launch_bounds__(256, 1) void conv3x3_fp8_kernel(Conv8Args p) {
  constexpr int BN = kF8BN, NW = kF8NW, NT = NW * 64, WN = 2, MREP = 8, NREP = 4, WNC = 64;
  constexpr int HS = kF8HS, HROWS = kF8HROWS, HP = kF8HPIECES, A_BYTES = kF8ABytes, SLOT = kF8Slot;
  constexpr int NPA = (HP + NW - 1) / NW;          // halo pieces per wave and slice (11)
  constexpr int BPW = BN / 8 / NW;                 // weight pieces per wave and tap (4)
  constexpr int RB = HS * 128;                     // bytes between image rows of the halo tile
  constexpr int OROW = BN * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fg = lane >> 4;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int b = blockIdx.x / tiles_per_img;
  const int trem = blockIdx.x - b * tiles_per_img;
  const int ty0 = (trem / p.tiles_x) * 16, tx0 = (trem % p.tiles_x) * 16;
  const int n0 = blockIdx.y * BN;
  const int nslice = p.Cin >> 7;
  const int64_t wt_tap = (int64_t)p.Cout * p.Cin;      // bytes between taps
  const int64_t img_base = (int64_t)b * p.H * p.W;
  const int rr = lane >> 3, pc = lane & 7;             // row / physical chunk of this lane inside an 8-row DMA piece
  const float dq = p.scale_in[0] * p.scale_w[0];       // dequantisation factor of the accumulators

  // ---- DMA sources.  Weight pieces of this wave: g = wave + NW j, rows 8 g .. 8 g + 7 of the tap's tile
  const char* bsrc[BPW];
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int row = (wave + NW * j) * 8 + rr;
    const int c = pc ^ (((row >> 1) & 3) << 1);
    bsrc[j] = p.wt + (int64_t)(n0 + row) * p.Cin + c * 16;
  }
  auto dma_w = [&](int cs, int tap, int slot) {
    const int64_t soff = (int64_t)tap * wt_tap + (int64_t)cs * 128;   // wave-uniform
#pragma unroll
    for (int j = 0; j < BPW; ++j) dma16(bsrc[j] + soff, lds0 + 2 * A_BYTES + slot * SLOT + (wave + NW * j) * 1024);
  };
  int aoff[NPA];          // byte offset of this lane's 16 bytes inside the image (slice 0), -1: padding / beyond the halo
  const char* img_ptr = p.in + img_base * p.in_ld;
#pragma unroll
  for (int q = 0; q < NPA; ++q) {
    const int g = q * NW + wave;
    const int row = g * 8 + rr;
    const int hy = row / HS, hx = row - hy * HS;
    const int c = pc ^ (((hx >> 1) & 3) << 1);       // the halo swizzle is keyed on the COLUMN (rows of one column differ by immediates)
    const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
    const bool ok = g < HP && row < HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    aoff[q] = ok ? (int)(((int64_t)iy * p.W + ix) * p.in_ld + c * 16) : -1;
  }
  auto dma_halo = [&](int q, int cs, int buf) {
    const int g = q * NW + wave;
    int o = aoff[q];
    asm volatile("" : "+v"(o));      // keep the 32-bit offset: hoisted out of the loop, six 64-bit source pointers spilled
    const char* src = o >= 0 ? img_ptr + o + (int64_t)cs * 128 : reinterpret_cast<const char*>(&g_zero16_f8);
    dma16(src, lds0 + buf * A_BYTES + g * 1024);
  };
  // ---- prologue: weights of tap 0, the whole halo of slice 0, weights of tap 1
  dma_w(0, 0, 0);
#pragma unroll
  for (int q = 0; q < NPA; ++q)
    if (q * NW + wave < HP) dma_halo(q, 0, 0);
  dma_w(0, 1, 1);
  wait_vm_barrier8<0>();

  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int i = 0; i < MREP; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses (bytes from smem)
  unsigned wbase[2], abase[3][2];
  {
    const int r0 = wn * WNC + fr;
    const int key = ((fr >> 1) & 3) << 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) wbase[h] = 2 * A_BYTES + r0 * 128 + (((fg + 4 * h) ^ key) << 4);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int hx = fr + kw;
      const int k2 = ((hx >> 1) & 3) << 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) abase[kw][h] = ((wm * MREP) * HS + hx) * 128 + (((fg + 4 * h) ^ k2) << 4);
    }
  }
  // Operand fragments of a tap: 4 weight blocks (16 channels each) x 4 image rows, 8 registers each.  Register plan (a full second
  // set would not fit beside the 64 accumulators at two waves per SIMD): the pixel fragments are double-buffered (A0 / A1 by step
  // parity), the weight fragments are not - a tap's MFMAs run in two halves, channel blocks {0, 1} then {2, 3}: W23 of the CURRENT
  // tap is read at the start of the step under the first half, and behind the barrier W01 and the pixel fragments of the NEXT tap
  // are read under the second half.
  i32x8 A0[MREP], A1[MREP], W01[2], W23[2];
  auto load8 = [&](unsigned lo_off, unsigned hi_off) -> i32x8 {
    const i32x4 lo = *reinterpret_cast<const i32x4*>(smem + lo_off);
    const i32x4 hi = *reinterpret_cast<const i32x4*>(smem + hi_off);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto load_w = [&](auto tap_, auto j0_, i32x8 (&w)[2]) {
    constexpr int tap = decltype(tap_)::value, j0 = decltype(j0_)::value;
    constexpr int WOFF = (tap % 3) * SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j) w[j] = load8(wbase[0] + WOFF + (j0 + j) * 2048, wbase[1] + WOFF + (j0 + j) * 2048);
  };
  auto load_a = [&](auto par_, auto tap_, i32x8 (&a)[MREP]) {
    constexpr int PAR = decltype(par_)::value, tap = decltype(tap_)::value;
    constexpr int kh = tap / 3, kw = tap - kh * 3;
    constexpr int AOFF = PAR * A_BYTES + kh * RB;
#pragma unroll
    for (int i = 0; i < MREP; ++i) a[i] = load8(abase[kw][0] + AOFF + i * RB, abase[kw][1] + AOFF + i * RB);
  };
  constexpr int ONE = 0x7f7f7f7f;    // E8M0 block scales 2^0
  auto mma_half = [&](const i32x8 (&w)[2], const i32x8 (&a)[MREP], int j0) {
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)   // A = weights (rows = channels), B = pixels: D[channel][pixel]
        acc[i][j0 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w[j], a[i], acc[i][j0 + j], 0, BFMT, 0, ONE, 0, ONE);
    // pin the MFMAs to this point of the step: the intrinsic touches no memory, so the optimiser would otherwise sink whole
    // accumulation chains past the barriers (sched_barrier only binds the machine scheduler) and spill every fragment on the way
    asm volatile("" : "+a"(acc[0][j0]), "+a"(acc[0][j0 + 1]), "+a"(acc[1][j0]), "+a"(acc[1][j0 + 1]), "+a"(acc[2][j0]), "+a"(acc[2][j0 + 1]),
                      "+a"(acc[3][j0]), "+a"(acc[3][j0 + 1]), "+a"(acc[4][j0]), "+a"(acc[4][j0 + 1]), "+a"(acc[5][j0]), "+a"(acc[5][j0 + 1]),
                      "+a"(acc[6][j0]), "+a"(acc[6][j0 + 1]), "+a"(acc[7][j0]), "+a"(acc[7][j0 + 1]));
  };
  using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>;

  // ---- main loop.  Global step g = 9 cs + tap, weight ring slot = tap % 3.  Per step:
  //   A  issue the weight DMA of step g+2 (slot last read by step g-1, whose reads every wave finished before ITS barrier)
  //      and, in taps 0..5, this wave's next piece of the NEXT slice's halo (the other buffer: free since the last barrier);
  //      read W23 of this tap
  //   B  first half of this tap's MFMAs (W01 and the pixel fragments were read in step g-1 behind its barrier)
  //   C  counted vmcnt + barrier: the weights of step g+1 (issued in step g-1) have landed for every wave; in tap 8 also
  //      the next slice's halo (its pieces were issued in taps 0..5, nothing younger than them is exempt there)
  //   D  read W01 and the pixel fragments of step g+1          E  second half of the MFMAs
  load_w(I0{}, I0{}, W01);
  load_a(I0{}, I0{}, A0);
  // LAST = the final slice: no next halo, weights only while they belong to this slice - every other slice issues the same DMA in
  // every instance of a step, so the counted waits are compile-time constants and the body is straight-line code
  auto slice_body = [&](auto par_, auto last_, int cs) {
    constexpr int PAR = decltype(par_)::value;
    constexpr bool LAST = decltype(last_)::value;
    static_for8<0, 9>([&](auto tap_) {
      constexpr int tap = decltype(tap_)::value;
      constexpr bool ODD = ((PAR * 9 + tap) & 1) != 0;
      constexpr int tn = (tap + 1) % 9;
      constexpr int PARN = tap == 8 ? (PAR ^ 1) : PAR;
      constexpr int t2 = (tap + 2) % 9;
      constexpr bool WEIGHTS = !LAST || tap + 2 < 9;
      // halo pieces of the next slice: q = 2 tap, 2 tap + 1 in taps 0..4, q = 10 (wave 0 only: piece 40) in tap 5
      constexpr int NH = LAST ? 0 : (tap < 5 ? 2 : (tap == 5 ? 1 : 0));
      // (in tap 5 the other waves wait as if they had issued piece 10 too: stricter, never wrong)
      constexpr int k = (tap < 5 ? NH : 0) + (WEIGHTS ? BPW : 0);
      if constexpr (NH == 2) { dma_halo(2 * tap, cs + 1, PAR ^ 1); dma_halo(2 * tap + 1, cs + 1, PAR ^ 1); }
      if constexpr (NH == 1) { if (wave == 0) dma_halo(10, cs + 1, PAR ^ 1); }
      if constexpr (WEIGHTS) dma_w(cs + (tap + 2) / 9, t2, t2 % 3);
      load_w(tap_, I2{}, W23);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ODD) mma_half(W01, A1, 0); else mma_half(W01, A0, 0);
      __builtin_amdgcn_sched_barrier(0);
      wait_vm_barrier8<k>();
      if constexpr (!(LAST && tap == 8)) {
        load_w(std::integral_constant<int, tn>{}, I0{}, W01);
        if constexpr (ODD) load_a(std::integral_constant<int, PARN>{}, std::integral_constant<int, tn>{}, A0);
        else load_a(std::integral_constant<int, PARN>{}, std::integral_constant<int, tn>{}, A1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ODD) mma_half(W23, A1, 2); else mma_half(W23, A0, 2);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
  for (int cs = 0; cs + 2 < nslice; cs += 2) {
    slice_body(P0{}, std::false_type{}, cs);
    slice_body(P1{}, std::false_type{}, cs + 1);
  }
  if (nslice & 1) {
    slice_body(P0{}, std::true_type{}, nslice - 1);
  } else {
    slice_body(P0{}, std::false_type{}, nslice - 2);
    slice_body(P1{}, std::true_type{}, nslice - 1);
  }
  __syncthreads();   // every wave is done with the staging area before the output tile overwrites it

  // ---- epilogue.  acc[i][j][r]: image row wm*4 + i, pixel fr, channel wn*64 + j*16 + fg*4 + r
  int nx = p.W - tx0; nx = nx > 16 ? 16 : nx;
  int nyt = p.H - ty0; nyt = nyt > 16 ? 16 : nyt;
  const bool full = nyt == 16 && nx == 16;
  const bool zero_oob = p.stat_sum != nullptr && !full;
#pragma unroll
  for (int j = 0; j < NREP; ++j) {
    const int nb = n0 + wn * WNC + j * 16 + fg * 4;
    float s1[4] = {dq, dq, dq, dq}, t1[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.pre_scale) {
      const float4 a = *reinterpret_cast<const float4*>(p.pre_scale + nb), c = *reinterpret_cast<const float4*>(p.pre_shift + nb);
      s1[0] = dq * a.x; s1[1] = dq * a.y; s1[2] = dq * a.z; s1[3] = dq * a.w; t1[0] = c.x; t1[1] = c.y; t1[2] = c.z; t1[3] = c.w;
    }
    const float bound = p.relu ? 0.f : -INFINITY;
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
      const int lrow = (wm * MREP + i) * 16 + fr;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[i][j][r] * s1[r] + t1[r], bound);
      uint2 u;
      u.x = pack2_bf16(v[0], v[1]);
      u.y = pack2_bf16(v[2], v[3]);
      if (zero_oob && !(wm * MREP + i < nyt && fr < nx)) u = make_uint2(0u, 0u);   // outside the image: not in the sums
      *reinterpret_cast<uint2*>(smem + lrow * OROW + (wn * WNC + j * 16 + fg * 4) * 2) = u;
    }
  }
  __syncthreads();
  if (p.stat_sum) {
    // BatchNorm partial statistics of the staged tile (as STORED: bf16) on the matrix cores, exactly as the bf16 halo kernel
    // takes them (conv_igemm.hip): wave w owns channels 16w .. 16w+15, column sums = ones^T X, sums of squares = diag(X^T X)
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const int q = fr >> 2, pp = fr & 3;
    const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
    const int nvalid = nyt * nx;
    const int64_t row = blockIdx.x;
#pragma unroll
    for (int cb = 0; cb < BN / 16 / NW; ++cb) {
      const int blk = wave + NW * cb;                 // 16-channel block of this pass
      f32x4 ssum = (f32x4){0.f, 0.f, 0.f, 0.f}, ssq = (f32x4){0.f, 0.f, 0.f, 0.f};
      const char* base = smem + (blk * 16) * 2 + pp * 8;
#pragma unroll
      for (int pg = 0; pg < 8; ++pg) {
        const int r0 = pg * 32 + 4 * fg + q, r1 = r0 + 16;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + r0 * OROW));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + r1 * OROW));
        const bf16x8 xf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        ssum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xf, ssum, 0, 0, 0);
        ssq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, xf, ssq, 0, 0, 0);
      }
      if (fg == q) {                                   // lane fr of group fr>>2 holds the diagonal entry of its channel
        const float sx = ssum[0];
        const float s2 = pp == 0 ? ssq[0] : (pp == 1 ? ssq[1] : (pp == 2 ? ssq[2] : ssq[3]));
        float m2 = s2 - sx * sx / (float)nvalid;
        m2 = m2 < 0.f ? 0.f : m2;
        const int n = n0 + blk * 16 + fr;
        p.stat_sum[row * p.Cout + n] = sx;
        p.stat_m2[row * p.Cout + n] = m2;
      }
    }
    if (blockIdx.y == 0 && tid == 0) p.stat_cnt[row] = (float)nvalid;
  }
  {
    // whole pixel rows out: a lane keeps its (pixel column, 16-byte chunk) and walks down the staged rows two image rows at a time
    constexpr int CPR = BN * 2 / 16, RPT = NT / CPR, ITERS = 256 / RPT, RPI = RPT / 16;
    const int lrow0 = tid / CPR, c = tid - lrow0 * CPR;
    const int x = tx0 + (lrow0 & 15), y0 = ty0 + (lrow0 >> 4);
    char* dst = p.out + ((img_base + (int64_t)y0 * p.W + x) * p.out_ld + n0) * 2 + c * 16;
    const int64_t dstep = (int64_t)RPI * p.W * p.out_ld * 2;
    uint4 v[ITERS];
#pragma unroll
    for (int k = 0; k < ITERS; ++k) v[k] = *reinterpret_cast<const uint4*>(smem + (lrow0 + k * RPT) * OROW + c * 16);
    if (full) {
#pragma unroll
      for (int k = 0; k < ITERS; ++k) *reinterpret_cast<uint4*>(dst + k * dstep) = v[k];
    } else {
#pragma unroll
      for (int k = 0; k < ITERS; ++k)
        if (y0 + k * RPI < p.H && x < p.W) *reinterpret_cast<uint4*>(dst + k * dstep) = v[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------ quantisation
// state of one per-tensor scale ("slot", kF8SlotFloats floats): { amax, scale, 1 / scale, fmax, amax the scale was made from } on the first
// 64-byte line, then kF8Parts lines holding one partial amax word each (q8_commit_amax)
template <int FMT>
__device__ __forceinline__ unsigned pack4_f8(float a, float b, float c, float d, float fmax) {
  a = fminf(fmaxf(a, -fmax), fmax); b = fminf(fmaxf(b, -fmax), fmax);
  c = fminf(fmaxf(c, -fmax), fmax); d = fminf(fmaxf(d, -fmax), fmax);
  unsigned w = 0;
  if constexpr (FMT == 0) {
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  } else {
    w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
  }
  return w;
}

// x (bf16 view) -> q (fp8 view) = fmt(x * inv_scale), and amax <- max(amax, max |x|) (order-independent: atomicMax on the bits of a
// non-negative float).  q == NULL: the amax only (calibration of a slot that has no history yet).
template <int FMT>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const char* __restrict__ x, int64_t x_ld, int64_t M, int C, char* __restrict__ q,
                                                           int64_t q_ld, float* __restrict__ slot, int take_amax) {
  const int CC = C / 16;
  const int64_t total = M * CC;
  const float inv = slot[2], fmax = slot[3];
  float am = 0.f;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    const uint4 u0 = *reinterpret_cast<const uint4*>(x + (pix * x_ld + cc * 16) * 2);
    const uint4 u1 = *reinterpret_cast<const uint4*>(x + (pix * x_ld + cc * 16) * 2 + 16);
    float f[16];
    Elem<bf16_t>::unpack(u0, f);
    Elem<bf16_t>::unpack(u1, f + 8);
#pragma unroll
    for (int k = 0; k < 16; ++k) am = fmaxf(am, fabsf(f[k]));
    if (q) {
      uint4 o;
      o.x = pack4_f8<FMT>(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv, fmax);
      o.y = pack4_f8<FMT>(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv, fmax);
      o.z = pack4_f8<FMT>(f[8] * inv, f[9] * inv, f[10] * inv, f[11] * inv, fmax);
      o.w = pack4_f8<FMT>(f[12] * inv, f[13] * inv, f[14] * inv, f[15] * inv, fmax);
      *reinterpret_cast<uint4*>(q + pix * q_ld + cc * 16) = o;
    }
  }
  if (!take_amax) return;      // (kernel argument: uniform)
  q8_commit_amax(am, slot);
}

// delayed scaling: scale <- margin * amax / fmax of the step that just ended (kept when the tensor was not produced: amax == 0),
// amax <- 0.  slots [n][4]; fmax is the slot's own (448 for e4m3, 57344 for e5m2).
__global__ void fp8_update_scales_kernel(float* slots, int n, float margin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float* s = slots + (int64_t)kF8SlotFloats * i;
  float am = s[0];
  for (int j = 0; j < kF8Parts; ++j) {
    am = fmaxf(am, s[16 * (1 + j)]);
    s[16 * (1 + j)] = 0.f;
  }
  s[4] = am;                   // (kept for inspection: the amax the current scale was made from)
  if (am > 0.f) {
    const float sc = margin * am / s[3];
    s[1] = sc;
    s[2] = 1.f / sc;
  }
  s[0] = 0.f;
}

}  // namespace cvcs

using namespace cvcs;

extern "C" int cvcs_conv3x3_fp8(const cvcs_conv8_desc* d, void* stream) {
  CVCS_CHECK_ARG(d != nullptr, "cvcs_conv3x3_fp8: null descriptor");
  CVCS_CHECK_ARG(d->in && d->wt && d->out && d->scale_in && d->scale_w, "cvcs_conv3x3_fp8: null tensor");
  CVCS_CHECK_ARG(d->B > 0 && d->H >= 8 && d->W >= 8, "cvcs_conv3x3_fp8: bad shape (maps of at least 8 pixels)");
  CVCS_CHECK_ARG(d->Cin > 0 && d->Cin % 128 == 0, "cvcs_conv3x3_fp8: Cin=%d must be a multiple of 128 (one MFMA K-step)", d->Cin);
  CVCS_CHECK_ARG(d->Cout > 0 && d->Cout % 128 == 0, "cvcs_conv3x3_fp8: Cout=%d must be a multiple of 128", d->Cout);
  CVCS_CHECK_ARG(d->in_fmt == CVCS_E4M3 || d->in_fmt == CVCS_E5M2, "cvcs_conv3x3_fp8: in_fmt");
  CVCS_CHECK_ARG(d->in_ld >= d->Cin && d->in_ld % 16 == 0, "cvcs_conv3x3_fp8: in_ld");
  CVCS_CHECK_ARG(d->out_ld >= d->Cout && (d->out_ld * 2) % 16 == 0, "cvcs_conv3x3_fp8: out_ld");
  CVCS_CHECK_ARG(((uintptr_t)d->in % 16) == 0 && ((uintptr_t)d->out % 16) == 0 && ((uintptr_t)d->wt % 16) == 0,
                 "cvcs_conv3x3_fp8: tensors must be 16-byte aligned");
  CVCS_CHECK_ARG((int64_t)d->H * d->W * d->in_ld < (1ll << 31), "cvcs_conv3x3_fp8: an image must stay below 2 GiB");
  CVCS_CHECK_ARG((d->stat_sum == nullptr) == (d->stat_m2 == nullptr) && (d->stat_sum == nullptr) == (d->stat_cnt == nullptr),
                 "cvcs_conv3x3_fp8: stat_sum/stat_m2/stat_cnt go together");
  CVCS_CHECK_ARG((d->pre_scale == nullptr) == (d->pre_shift == nullptr), "cvcs_conv3x3_fp8: a folded BatchNorm needs both scale and shift");
  Conv8Args a;
  a.in = (const char*)d->in; a.wt = (const char*)d->wt; a.out = (char*)d->out;
  a.scale_in = d->scale_in; a.scale_w = d->scale_w; a.pre_scale = d->pre_scale; a.pre_shift = d->pre_shift;
  a.stat_sum = d->stat_sum; a.stat_m2 = d->stat_m2; a.stat_cnt = d->stat_cnt;
  a.in_ld = d->in_ld; a.out_ld = d->out_ld;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.relu = d->relu;
  a.tiles_x = (int)cdiv(d->W, 16); a.tiles_y = (int)cdiv(d->H, 16);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_fp8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kF8Lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_fp8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kF8Lds);
    attr_done = true;
  }
  dim3 grid((unsigned)(a.B * a.tiles_x * a.tiles_y), (unsigned)(a.Cout / kF8BN));
  if (d->in_fmt == CVCS_E4M3) hipLaunchKernelGGL((conv3x3_fp8_kernel<0>), grid, dim3(kF8NW * 64), kF8Lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((conv3x3_fp8_kernel<1>), grid, dim3(kF8NW * 64), kF8Lds, (hipStream_t)stream, a);
  CVCS_CHECK_LAUNCH("cvcs_conv3x3_fp8");
  return CVCS_OK;
}

extern "C" int cvcs_quantize_fp8(const void* x, int64_t x_ld, int64_t M, int C, void* q, int64_t q_ld, int fmt, float* slot, int take_amax,
                                 void* stream) {
  CVCS_CHECK_ARG(x && slot && M > 0 && C > 0 && C % 16 == 0, "cvcs_quantize_fp8: bad argument (C must be a multiple of 16)");
  CVCS_CHECK_ARG(fmt == CVCS_E4M3 || fmt == CVCS_E5M2, "cvcs_quantize_fp8: fmt");
  CVCS_CHECK_ARG(q || take_amax, "cvcs_quantize_fp8: nothing to do (no output view and no amax)");
  CVCS_CHECK_ARG(x_ld >= C && (x_ld * 2) % 16 == 0 && ((uintptr_t)x % 16) == 0, "cvcs_quantize_fp8: x view");
  CVCS_CHECK_ARG(!q || (q_ld >= C && q_ld % 16 == 0 && ((uintptr_t)q % 16) == 0), "cvcs_quantize_fp8: q view");
  const int64_t total = M * (C / 16);
  int64_t blocks = cdiv(total, 256);
  blocks = blocks > 256 * 8 ? 256 * 8 : blocks;
  if (fmt == CVCS_E4M3)
    hipLaunchKernelGGL((quantize_fp8_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, M, C, (char*)q, q_ld, slot, take_amax);
  else
    hipLaunchKernelGGL((quantize_fp8_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, M, C, (char*)q, q_ld, slot, take_amax);
  CVCS_CHECK_LAUNCH("cvcs_quantize_fp8");
  return CVCS_OK;
}

extern "C" int cvcs_fp8_update_scales(float* slots, int n, float margin, void* stream) {
  CVCS_CHECK_ARG(slots && n > 0 && margin >= 1.f, "cvcs_fp8_update_scales: bad argument");
  hipLaunchKernelGGL(fp8_update_scales_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, slots, n, margin);
  CVCS_CHECK_LAUNCH("cvcs_fp8_update_scales");
  return CVCS_OK;
}
