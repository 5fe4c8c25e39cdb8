// Weight gradient of a convolution on gfx950: dW[tap][co][ci] = sum_p dy[p][co] * x[pix(p,tap)][ci].
//
// GEMM view per tap: M = Cout, N = Cin, K = B*Ho*Wo output pixels.  NHWC tensors are channel-contiguous, i.e.
// K runs along LDS *rows*: the MFMA fragments are therefore read TRANSPOSED from LDS -
//   bf16: ds_read_b64_tr_b16 (4 pixel rows x 16 channels -> 4 k-values of one channel per lane),
//   f32 : four ds_read_b32 (one channel of one pixel each) feeding v_mfma_f32_16x16x4_f32.
// A K-tile is a TH x TW = 32-pixel rectangle of one image; its dy tile [32][64 co] and its x halo tile
// [((TH-1)s+KH) x ((TW-1)s+KW)][64 ci] are staged once and reused by all KH*KW taps (a tap is a row offset into
// the halo tile, and every lane supplies its own row address to the transposed read, so shifts are free).
// k -> pixel permutation: lane group g takes tile pixels {4g..4g+3} and {16+4g..16+4g+3}; with the 32-byte row
// padding the two 4x16 blocks of a 32-lane half fall on 8 distinct 32-byte bank slots (conflict-free, stride 1).
//
// Workgroup: 256 threads = 4 waves 2(co) x 2(ci), tile 64 co x 64 ci x all taps; wave tile 32 x 32 per tap
// (NT x 4 accumulators).  Split-K over contiguous K-tile ranges; partials [slice][tap][co][ci] f32, reduced in a
// fixed order (bitwise reproducible) straight into the reference's OIHW parameter layout.
//
// Reference op replaced: weight half of convolution_backward for nn.Conv2d / nn.ConvTranspose2d (S/train.py:125).
#include <type_traits>

#include "common.h"

namespace cvcs {

struct WgradArgs {
  const char* x;
  const char* dy;
  float* ws;
  int64_t x_ld, dy_ld;
  int B, H, W, Cin, Ho, Wo, Cout;
  int KW, stride, pad;
  int stride_w, pad_w;        // generic kernel: W axis (== stride / pad unless cvcs_wgrad_desc.aniso)
  int64_t x_row_pitch, x_img_pitch;   // generic kernel: pitches of `x` in elements
  // dilated 3x3 weight gradients run tap by tap as SHIFTED 1x1 problems (x pixel = dy pixel + (sh_y, sh_x), zero outside the
  // image): partial slabs are laid out [slice][9 taps][co][ci] (slice_stride != 0 overrides the kernel's own stride)
  int sh_y, sh_x;
  int64_t slice_stride;
  int TH, TW, HR, HC;         // K-tile and halo-tile extents
  int tiles_x, tiles_y;       // K-tiles per image
  int ktiles, per_slice;      // total K-tiles, K-tiles per slice
  int ntile_n;                // Cin tiles
  int tiles_mn, xcd_order;    // GEMM kernel: (co, ci) tiles per K-slice; XCD-aware workgroup order (CVCS_XCD_ORDER=0: off)
  float* bias_ws;             // GEMM kernel: partial bias gradients [slice][Cout] (NULL: none)
  int x_s2;                   // GEMM kernel: x is read at stride 2 (1x1 / stride-2 projection shortcuts: output pixel (b, oy, ox) <- x pixel (b, 2 oy, 2 ox))
};

__device__ uint4 g_wzero16;  // zero word: LDS-DMA source of padding pixels / absent channels

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// four transposed LDS reads (4 pixel rows x 16 channels each -> 4 k-values of one channel per lane) + their wait
__device__ __forceinline__ void tr_read4(u32x2& o0, u32x2& o1, u32x2& o2, u32x2& o3, unsigned a0, unsigned a1, unsigned a2,
                                         unsigned a3) {
  asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\tds_read_b64_tr_b16 %2, %6\n\t"
               "ds_read_b64_tr_b16 %3, %7\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
               : "memory");
}

__device__ __forceinline__ void wait_vm_barrier_n(int k) {
  switch (k) {
    case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;   // unknown count: drain
  }
}

// Staging is LDS-DMA (global_load_lds_dwordx4, 1 KiB lane-linear pieces) into a ring of NS stages, NS-1 K-tiles ahead;
// each iteration ends with a counted vmcnt (the pieces this wave issued in this iteration may stay in flight) and a
// raw barrier.  bf16 LDS rows are 128 B, 32-byte chunk cc of row r is stored at chunk cc ^ ((r>>1)&3) (swizzle applied
// on the DMA source address): the two 4x16 blocks a 32-lane half of ds_read_b64_tr_b16 reads - eight consecutive
// tile rows - then fall on eight distinct 32-byte bank slots for any starting row.
template <typename T, int NT, int NS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs p) {
  constexpr int ES = sizeof(T);
  constexpr int ROWB = 64 * ES;            // LDS row: 64 channels of one pixel
  constexpr int RPP = 1024 / ROWB;         // rows per DMA piece (8 | 4)
  constexpr int CPR = ROWB / 16;           // 16-byte chunks per row (8 | 16)
  constexpr int DY_BYTES = 32 * ROWB;
  constexpr int X_BYTES = 160 * ROWB;      // kXRows halo rows
  constexpr int STAGE = DY_BYTES + X_BYTES;
  constexpr int DYP = 32 / RPP;            // dy pieces per K-tile (4 | 8)
  constexpr int D = NS - 1;                // prefetch distance in K-tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x;
  const int tn = tile % p.ntile_n, tm = tile / p.ntile_n;
  const int co0 = tm * 64, ci0 = tn * 64;
  const int slice = blockIdx.y;
  const int kt_begin = slice * p.per_slice;
  int kt_end = kt_begin + p.per_slice;
  if (kt_end > p.ktiles) kt_end = p.ktiles;

  f32x4 acc[NT][2][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int hrows = p.HR * p.HC;
  const int xpieces = (hrows + RPP - 1) / RPP;
  const int cin_rem = p.Cin - ci0;
  const int cout_rem = p.Cout - co0;            // < 64 in the last row tile of a Cout that is no multiple of 64 (Swin's 96 / 288)
  const int rr = lane / CPR, pc = lane % CPR;   // row / physical 16-byte chunk of this lane inside a DMA piece
  const int my_pieces = DYP / 4 + (xpieces - wave + 3) / 4;   // DMA instructions this wave issues per K-tile
  const bool m0ok = wm * 32 < cout_rem, m1ok = wm * 32 + 16 < cout_rem, n0ok = wn * 32 < cin_rem, n1ok = wn * 32 + 16 < cin_rem;

  auto swz_chunk = [&](int row, int chunk) -> int {   // logical <-> physical 16-byte chunk (involution)
    if constexpr (ES == 2) return chunk ^ (((row >> 1) & 3) << 1);
    else return chunk;
  };
  auto issue_tile = [&](int kt, int st) {
    const int b = kt / (p.tiles_x * p.tiles_y);
    const int tr = kt - b * (p.tiles_x * p.tiles_y);
    const int ty0 = (tr / p.tiles_x) * p.TH, tx0 = (tr % p.tiles_x) * p.TW;
    char* sdy = smem + st * STAGE;
    char* sx = sdy + DY_BYTES;
    for (int pi = wave; pi < DYP; pi += 4) {
      const int k = pi * RPP + rr;
      const int c = swz_chunk(k, pc);
      const int oy = ty0 + k / p.TW, ox = tx0 + k % p.TW;
      const char* src = reinterpret_cast<const char*>(&g_wzero16);
      if (oy < p.Ho && ox < p.Wo && c * (16 / ES) < cout_rem)
        src = p.dy + ((((int64_t)b * p.Ho + oy) * p.Wo + ox) * p.dy_ld + co0) * ES + c * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sdy + pi * 1024), 16, 0, 0);
    }
    const int iy_base = ty0 * p.stride - p.pad, ix_base = tx0 * p.stride_w - p.pad_w;
    for (int pi = wave; pi < xpieces; pi += 4) {
      const int r = pi * RPP + rr;
      const int c = swz_chunk(r, pc);
      const int iy = iy_base + r / p.HC, ix = ix_base + r % p.HC;
      const char* src = reinterpret_cast<const char*>(&g_wzero16);
      if (r < hrows && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c * (16 / ES) < cin_rem)
        src = p.x + ((int64_t)b * p.x_img_pitch + (int64_t)iy * p.x_row_pitch + (int64_t)ix * p.x_ld + ci0) * ES + c * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sx + pi * 1024), 16, 0, 0);
    }
  };

  // ---- prologue: the first D K-tiles
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (kt_begin + d < kt_end) issue_tile(kt_begin + d, d);
  wait_vm_barrier_n(0);

  int st = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    int issued = 0;
    if (kt + D < kt_end) {
      int st2 = st + D; if (st2 >= NS) st2 -= NS;
      issue_tile(kt + D, st2);
      issued = my_pieces;
    }
    if constexpr (ES == 2) {
      const unsigned dyb = lds0 + st * STAGE, xb = dyb + DY_BYTES;
      // lane l: group g = l>>4, i = l&15 -> supplies row q = i>>2 and 8 bytes at column 4*(i&3) of its group's block
      const int q = fr >> 2, pp = fr & 3;
      const int k0 = 4 * fg + q, k1 = 16 + 4 * fg + q;      // tile pixels of the two transposed reads
      auto addr = [&](unsigned base, int row, int cc) -> unsigned {
        return base + row * ROWB + ((cc ^ ((row >> 1) & 3)) << 5) + pp * 8;
      };
      u32x2 a0l, a0h, a1l, a1h;
      tr_read4(a0l, a0h, a1l, a1h, addr(dyb, k0, wm * 2), addr(dyb, k1, wm * 2), addr(dyb, k0, wm * 2 + 1),
               addr(dyb, k1, wm * 2 + 1));
      const bf16x8 af0 = __builtin_bit_cast(bf16x8, make_uint4(a0l.x, a0l.y, a0h.x, a0h.y));
      const bf16x8 af1 = __builtin_bit_cast(bf16x8, make_uint4(a1l.x, a1l.y, a1h.x, a1h.y));
      const int hb0 = ((k0 / p.TW) * p.stride) * p.HC + (k0 % p.TW) * p.stride_w;
      const int hb1 = ((k1 / p.TW) * p.stride) * p.HC + (k1 % p.TW) * p.stride_w;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int kh = t / p.KW, kw = t - kh * p.KW;
        const int off = kh * p.HC + kw;
        u32x2 b0l, b0h, b1l, b1h;
        tr_read4(b0l, b0h, b1l, b1h, addr(xb, hb0 + off, wn * 2), addr(xb, hb1 + off, wn * 2), addr(xb, hb0 + off, wn * 2 + 1),
                 addr(xb, hb1 + off, wn * 2 + 1));
        const bf16x8 bf0 = __builtin_bit_cast(bf16x8, make_uint4(b0l.x, b0l.y, b0h.x, b0h.y));
        const bf16x8 bf1 = __builtin_bit_cast(bf16x8, make_uint4(b1l.x, b1l.y, b1h.x, b1h.y));
        // thin layers (Cout / Cin of 16 | 32 inside the 64 x 64 tile): the 16 x 16 blocks that lie wholly in the zero padding are skipped
        // (wave-uniform tests; a 16 -> 16 layer leaves ONE of the sixteen blocks of a tile)
        if (m0ok && n0ok) acc[t][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bf0, acc[t][0][0], 0, 0, 0);
        if (m0ok && n1ok) acc[t][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bf1, acc[t][0][1], 0, 0, 0);
        if (m1ok && n0ok) acc[t][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bf0, acc[t][1][0], 0, 0, 0);
        if (m1ok && n1ok) acc[t][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bf1, acc[t][1][1], 0, 0, 0);
      }
    } else {
      // f32 (parity path): plain LDS reads; MFMA pass s contracts tile pixels k = 4s + g (g = lane>>4)
      const char* sdy = smem + st * STAGE;
      const char* sx = sdy + DY_BYTES;
#pragma unroll 1
      for (int s = 0; s < 8; ++s) {
        const int k = 4 * s + fg;
        float af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          af[i] = *reinterpret_cast<const float*>(sdy + k * ROWB + (wm * 32 + i * 16 + fr) * 4);
        const int hb = ((k / p.TW) * p.stride) * p.HC + (k % p.TW) * p.stride_w;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int kh = t / p.KW, kw = t - kh * p.KW;
          const int row = hb + kh * p.HC + kw;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float bv = *reinterpret_cast<const float*>(sx + row * ROWB + (wn * 32 + j * 16 + fr) * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bv, acc[t][i][j], 0, 0, 0);
          }
        }
      }
    }
    wait_vm_barrier_n(D >= 2 ? issued : 0);
    if (++st == NS) st = 0;
  }

  // ---- partials: ws[slice][tap][co][ci]; D layout: row (co) = fg*4 + r, col (ci) = fr
  const int64_t slice_stride = p.slice_stride ? p.slice_stride : (int64_t)NT * p.Cout * p.Cin;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ci = ci0 + wn * 32 + j * 16 + fr;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wm * 32 + i * 16 + fg * 4 + r;
          if (co < p.Cout) p.ws[slice * slice_stride + ((int64_t)t * p.Cout + co) * p.Cin + ci] = acc[t][i][j][r];
        }
      }
}

typedef __attribute__((ext_vector_type(4))) short s16x4v;
typedef __attribute__((ext_vector_type(8))) short s16x8v;
__device__ __forceinline__ bf16x8 tr_pair(const char* smem_base, unsigned off0, unsigned off1) {
  const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(smem_base + off0));
  const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(smem_base + off1));
  const s16x8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}


// ===================================================================================================================
// 1x1 / stride 1 weight gradients (bf16): one transposed GEMM  dW[co][ci] = sum_p dy[p][co] * x[p][ci]  over all pixels.
// The ResNet bottlenecks make two thirds of their encoder's multiplies in 1x1 convolutions; on the generic kernel above a
// K-tile feeds 16 MFMAs from 12 KB of LDS fill (206 TFLOP/s measured on ResNet50-UNet).  Here a workgroup owns BM x BN
// channels (128 x 128, 128 x 64 or 64 x 128), 4 waves 2 x 2, wave tile BM/2 x BN/2: a 32-pixel K-tile feeds
// (BM/32)*(BN/32) MFMAs per wave from (BM+BN)*64 bytes - 4x the MFMAs per byte.  Both operands are pixel-major, so every
// fragment is a transposed LDS read (ds_read_b64_tr_b16) exactly as in the kernel above; the tile is kept as 64-channel
// column blocks of [32 pixels][128 B] so that its bank swizzle is that kernel's.  Same LDS-DMA ring (3 stages, two K-tiles
// ahead, counted vmcnt), same split-K partial slabs + fixed-order reduce.
// WMV waves along the co axis x 2 along ci (4 or 8 waves).  The kernel is bound by the L2 -> LDS path, not by the matrix cores: a BM x BN
// tile moves (BM + BN) * 64 bytes per 32-pixel K-tile for BM * BN * 64 FLOP, i.e. BM*BN/(BM+BN) FLOP per byte - 64 for 128 x 128, 85 for
// 256 x 128 (8 waves, still two workgroups per CU), 128 for 256 x 256 (8 waves, 96 KB of LDS, one workgroup per CU).
template <int BM, int BN, int WMV = 2, int NSTAGE = 3>
__global__ __launch_bounds__(WMV * 128, (BM * BN > 256 * 128 ? 1 : 2)) void wgrad_gemm_kernel(WgradArgs p) {
  constexpr int NWV = WMV * 2;                    // waves
  constexpr int MB = BM / 64, NB = BN / 64;       // 64-channel blocks of the dy / x tile
  constexpr int BLK = 32 * 128;                   // bytes of one block: 32 pixels x 64 channels
  constexpr int STAGE = (MB + NB) * BLK;
  constexpr int NS = NSTAGE, D = NSTAGE - 1;      // ring stages, K-tiles in flight ahead of the one being multiplied
  constexpr int PIECES = (MB + NB) * 4;           // 1 KiB DMA pieces per K-tile (8 pixel rows of one block each)
  constexpr int PPW = PIECES / NWV;               // per wave
  static_assert(PIECES % NWV == 0 && PPW <= 4, "DMA pieces must divide over the waves");
  constexpr int MR = BM / WMV / 16, NR = BN / 32; // 16-channel fragments per wave (wave tile BM/WMV x BN/2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // 1-D grid in XCD-aware order, the (co, ci) tiles of one K-slice consecutive: they all read that slice's pixels of dy and x, so the
  // slice crosses the fabric once per XCD-resident group instead of once per tile
  const unsigned lid = p.xcd_order ? xcd_order(blockIdx.x, gridDim.x) : blockIdx.x;
  const int tile = (int)(lid % p.tiles_mn), kslice = (int)(lid / p.tiles_mn);
  const int tn = tile % p.ntile_n, tm = tile / p.ntile_n;
  const int co0 = tm * BM, ci0 = tn * BN;
  const int64_t M = (int64_t)p.B * p.Ho * p.Wo;
  const int kt_begin = kslice * p.per_slice;
  int kt_end = kt_begin + p.per_slice;
  if (kt_end > p.ktiles) kt_end = p.ktiles;

  // per-lane DMA sources of this wave's pieces: piece -> (block, 8-row group); lane -> (row in group, physical chunk)
  const int rr = lane >> 3, pc = lane & 7;
  const char* src0[PPW];
  int64_t rowb[PPW];      // bytes per pixel row of the source
  int prow[PPW];          // pixel row inside the K-tile
  bool colok[PPW];        // channel counts that are no multiple of the tile (96 * 2^k): chunks beyond the last channel read zeros
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pi = wave + NWV * i;
    const int blk = pi >> 2, row = (pi & 3) * 8 + rr;
    const int c = pc ^ (((row >> 1) & 3) << 1);   // logical 16-byte chunk stored at physical chunk pc
    prow[i] = row;
    if (blk < MB) { src0[i] = p.dy + ((int64_t)co0 + blk * 64) * 2 + c * 16; rowb[i] = p.dy_ld * 2; colok[i] = co0 + blk * 64 + c * 8 < p.Cout; }
    else          { src0[i] = p.x + ((int64_t)ci0 + (blk - MB) * 64) * 2 + c * 16; rowb[i] = p.x_ld * 2; colok[i] = ci0 + (blk - MB) * 64 + c * 8 < p.Cin; }
  }
  const bool shifted = (p.sh_y | p.sh_x) != 0;
  const int64_t sh_lin = (int64_t)p.sh_y * p.W + p.sh_x;
  auto issue_tile = [&](int kt, int st) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int64_t pix = (int64_t)kt * 32 + prow[i];
      bool ok = pix < M && colok[i];
      if (shifted && (wave + NWV * i) >= 4 * MB) {      // an x piece of a shifted tap: the source pixel must stay inside its image
        const int ox = (int)(pix % p.W), oy = (int)((pix / p.W) % p.H);
        ok = ok && (unsigned)(oy + p.sh_y) < (unsigned)p.H && (unsigned)(ox + p.sh_x) < (unsigned)p.W;
        pix += sh_lin;
      }
      if (p.x_s2 && (wave + NWV * i) >= 4 * MB && ok) {     // an x piece of a stride-2 1x1 layer: (b, oy, ox) of the output pixel -> (b, 2 oy, 2 ox) of x
        const int ox = (int)(pix % p.Wo);
        const int64_t t = pix / p.Wo;
        const int oy = (int)(t % p.Ho);
        pix = ((t / p.Ho) * p.H + 2 * oy) * p.W + 2 * ox;
      }
      const char* src = ok ? src0[i] + pix * rowb[i] : reinterpret_cast<const char*>(&g_wzero16);
      dma16(src, lds0 + st * STAGE + (wave + NWV * i) * 1024);
    }
  };

  f32x4 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient (cvcs_wgrad_desc.dbias): the ci-tile-0 workgroups of every K-slice also sum their dy fragments over the pixels
  // (lane = channel fr of fragment i, eight pixels per fragment; the four lane groups hold different pixels)
  const bool do_bias = p.bias_ws != nullptr && tn == 0 && wn == 0;
  float bsum[MR];
#pragma unroll
  for (int i = 0; i < MR; ++i) bsum[i] = 0.f;

  const int fr = lane & 15, fg = lane >> 4;
  const int q = fr >> 2, pp = fr & 3;
  const int k0 = 4 * fg + q, k1 = 16 + 4 * fg + q;   // tile pixels of the two transposed reads of a fragment
  // byte offsets inside a block of the (k0 | k1, 16-channel chunk cc) reads
  unsigned off0[4], off1[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    off0[cc] = k0 * 128 + ((cc ^ ((k0 >> 1) & 3)) << 5) + pp * 8;
    off1[cc] = k1 * 128 + ((cc ^ ((k1 >> 1) & 3)) << 5) + pp * 8;
  }

#pragma unroll
  for (int d = 0; d < D; ++d)
    if (kt_begin + d < kt_end) issue_tile(kt_begin + d, d);
  wait_vm_barrier_n(0);
  int st = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    if (kt + D < kt_end) {
      int st2 = st + D; if (st2 >= NS) st2 -= NS;
      issue_tile(kt + D, st2);
    }
    // K-tiles that may stay in flight behind the next one (kt + 1 must have landed)
    int ahead = (kt + D < kt_end ? kt + D : kt_end - 1) - (kt + 1);
    ahead = ahead < 0 ? 0 : ahead;
    const char* sb = smem + st * STAGE;
    bf16x8 af[MR], bfr[NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int ch = wm * (BM / WMV) + i * 16;
      const char* blk = sb + (ch >> 6) * BLK;
      af[i] = tr_pair(blk, off0[(ch >> 4) & 3], off1[(ch >> 4) & 3]);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int ch = wn * (BN / 2) + j * 16;
      const char* blk = sb + (MB + (ch >> 6)) * BLK;
      bfr[j] = tr_pair(blk, off0[(ch >> 4) & 3], off1[(ch >> 4) & 3]);
    }
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        float f[8];
        Elem<bf16_t>::unpack(__builtin_bit_cast(uint4, af[i]), f);
        bsum[i] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
      }
    }
    wait_vm_barrier_n(ahead * PPW);
    if (++st == NS) st = 0;
  }
  // partials ws[slice][co][ci]; D layout: row (co) = fg*4 + r, col (ci) = fr
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int co = co0 + wm * (BM / WMV) + i * 16 + fr;
      if (fg == 0 && co < p.Cout) p.bias_ws[(int64_t)kslice * p.Cout + co] = v;
    }
  }
  float* ws = p.ws + (int64_t)kslice * (p.slice_stride ? p.slice_stride : (int64_t)p.Cout * p.Cin);
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int ci = ci0 + wn * (BN / 2) + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * (BM / WMV) + i * 16 + fg * 4 + r;
        if (co < p.Cout && ci < p.Cin) ws[(int64_t)co * p.Cin + ci] = acc[i][j][r];
      }
    }
}

// ===================================================================================================================
// Fast path: bf16, 3x3 / stride 1 / pad 1, K-tiles of 1 x 32 pixels (Wo > 16).  Same algorithm and LDS-DMA ring as the
// generic kernel, with everything the inner loop needs made loop-invariant:
//   * the halo tile is stored with a row pitch of 40 pixels (34 used): a filter row is then 40*128 bytes = a multiple
//     of the swizzle period, so the swizzled address of a transposed read depends on (k + kw) only -> 12 per-lane
//     registers cover all nine taps, the tap row and the ring stage are instruction immediates;
//   * per-lane DMA source offsets are computed once (no divisions per K-tile), the K-tile origin is scalar;
//   * the LDS-DMA is issued from inline asm, the transposed reads are builtins: the compiler interleaves reads and MFMAs
//     with its own fine-grained lgkmcnt waits (it would drain vmcnt before every LDS read if it saw the DMA).
template <int I, int N, typename F>
__device__ __forceinline__ void wg_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wg_static_for<I + 1, N>(f);
  }
}

// K-tile shape TH x TW (TH*TW = 32).  The halo tile is what crosses L2->LDS per K-tile: 3x34 pixels for 1x32, 6x10 for
// 4x8 - the square-ish tile re-reads the input 1.9x instead of 3.2x (the deep levels of this kernel are MALL-bound and
// the shallow ones HBM-bound on exactly that traffic).
template <int TH, int TW>
__global__ __launch_bounds__(256, 2) void wgrad_fast_kernel(WgradArgs p) {
  static_assert(TH * TW == 32, "a K-tile is 32 output pixels");
  constexpr int ROWB = 128;                        // LDS row bytes (64 bf16)
  constexpr int HP = (TW + 2 + 7) / 8 * 8;         // halo pitch in pixels, a multiple of the swizzle period
  constexpr int HR = TH + 2;
  constexpr int DY_BYTES = 32 * ROWB, X_ROWS = HR * HP, X_BYTES = X_ROWS * ROWB;
  constexpr int STAGE = DY_BYTES + X_BYTES;
  constexpr int XP = X_ROWS / 8;                   // halo DMA pieces
  constexpr int XJ = (XP + 3) / 4;                 // ... per wave
  constexpr int NS = 3;                            // ring stages: two K-tiles in flight (a third buys nothing: measured)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;
  const int tn = tile % p.ntile_n, tm = tile / p.ntile_n;
  const int co0 = tm * 64, ci0 = tn * 64;
  const int slice = blockIdx.y;
  const int kt_begin = slice * p.per_slice;
  int kt_end = kt_begin + p.per_slice;
  if (kt_end > p.ktiles) kt_end = p.ktiles;

  // wave w owns ALL 64 output channels x input channels [16w, 16w+16) for all nine taps: the dy fragments (4 blocks) are
  // read once per K-tile and reused by every tap, each x fragment feeds four MFMAs - 26 transposed reads per 36 MFMAs
  // (a 32x32 wave tile needs 40): this kernel is LDS-bandwidth-bound (fragment reads + DMA writes), not latency-bound.
  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int cin_rem = p.Cin - ci0;
  const int rr = lane >> 3, pc = lane & 7;   // row / physical 16-byte chunk of this lane inside a DMA piece
  auto swzc = [](int row, int chunk) { return chunk ^ (((row >> 1) & 3) << 1); };

  // ---- per-lane DMA sources.  dy: piece `wave`, tile pixel k = wave*8 + rr.  halo: pieces wave + 4j (j < 4).
  const int kdy = wave * 8 + rr;
  const int kdy_y = kdy / TW, kdy_x = kdy % TW;
  const int off_dy = ((kdy_y * p.Wo + kdy_x) * (int)p.dy_ld + co0) * 2 + swzc(kdy, pc) * 16;
  int off_x[XJ], hyx[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int r = (wave + 4 * j) * 8 + rr;
    const int hy = r / HP, hx = r - hy * HP;
    const int c = swzc(r, pc);
    const bool ok = (wave + 4 * j) < XP && hx < TW + 2 && c * 8 < cin_rem;
    off_x[j] = ((hy * p.W + hx) * (int)p.x_ld + ci0) * 2 + c * 16;
    hyx[j] = ok ? ((hy << 8) | hx) : -1;
  }
  const int my_pieces = 1 + (XP - wave + 3) / 4;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  // K-tiles are issued strictly in sequence: the coordinates of the next one are kept incrementally (no divisions per issue)
  int ib, ity, itx;
  {
    ib = kt_begin / tiles_per_img;
    const int tr = kt_begin - ib * tiles_per_img;
    ity = tr / p.tiles_x;
    itx = tr - ity * p.tiles_x;
  }
  // byte offsets of the current tile ROW (tx0 = 0) inside dy and x, recomputed only when the row changes
  int64_t dy_row, x_row;
  auto set_row = [&]() {
    dy_row = ((((int64_t)ib * p.Ho + ity * TH) * p.Wo) * p.dy_ld) * 2;
    x_row = ((((int64_t)ib * p.H + ity * TH - 1) * p.W - 1) * p.x_ld) * 2;
  };
  set_row();
  const char* zsrc = reinterpret_cast<const char*>(&g_wzero16);
  asm volatile("" : "+s"(zsrc));   // keep the zero word's address in SGPRs (otherwise re-materialised pc-relative per use)
  auto issue_tile = [&](int st) {
    const int oy = ity * TH, tx0 = itx * TW;
    const unsigned sdy = lds0 + st * STAGE;
    const unsigned sx = sdy + DY_BYTES;
    const char* dyrow = p.dy + dy_row + (int64_t)(tx0 * (int)p.dy_ld * 2);
    const char* src = (oy + kdy_y < p.Ho && tx0 + kdy_x < p.Wo) ? dyrow + off_dy : zsrc;
    dma16(src, sdy + wave * 1024);
    const int iy0 = oy - 1, ix0 = tx0 - 1;
    const char* xorg = p.x + x_row + (int64_t)(tx0 * (int)p.x_ld * 2);   // may lie before the tensor: only dereferenced for valid pixels
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      if (wave + 4 * j < XP) {
        const int hy = hyx[j] >> 8, hx = hyx[j] & 255;
        const bool ok = hyx[j] >= 0 && (unsigned)(iy0 + hy) < (unsigned)p.H && (unsigned)(ix0 + hx) < (unsigned)p.W;
        const char* s2 = ok ? xorg + off_x[j] : zsrc;
        dma16(s2, sx + (wave + 4 * j) * 1024);
      }
    }
    if (++itx == p.tiles_x) {
      itx = 0;
      if (++ity == p.tiles_y) { ity = 0; ++ib; }
      set_row();
    }
  };

  // ---- per-lane transposed-read offsets (relative to the stage base)
  const int q = fr >> 2, pp = fr & 3;
  const int k0 = 4 * fg + q, k1 = 16 + 4 * fg + q;
  auto taddr = [&](int row, int cc) -> unsigned { return row * ROWB + ((cc ^ ((row >> 1) & 3)) << 5) + pp * 8; };
  unsigned ad[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { ad[i][0] = taddr(k0, i); ad[i][1] = taddr(k1, i); }
  const int h0 = (k0 / TW) * HP + k0 % TW, h1 = (k1 / TW) * HP + k1 % TW;   // halo rows of the two tile pixels (tap 0,0)
  unsigned ax[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) { ax[kw][0] = taddr(h0 + kw, wave); ax[kw][1] = taddr(h1 + kw, wave); }

  // ---- prologue
  if (kt_begin < kt_end) issue_tile(0);
  if (kt_begin + 1 < kt_end) issue_tile(1);
  wait_vm_barrier_n(0);

  auto tile_body = [&](auto st_, int kt) {
    constexpr int ST = decltype(st_)::value;
    int issued = 0;
    if (kt + 2 < kt_end) { issue_tile((ST + 2) % NS); issued = my_pieces; }
    const char* sb = smem + ST * STAGE;
    if (wave * 16 < cin_rem) {   // a wave whose 16 input channels do not exist (the first layer stores 8) only moves data
    bf16x8 af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = tr_pair(sb, ad[i][0], ad[i][1]);
    wg_static_for<0, 9>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      constexpr int kh = t / 3, kw = t - kh * 3;
      const char* xb = sb + DY_BYTES + kh * HP * ROWB;
      const bf16x8 bf = tr_pair(xb, ax[kw][0], ax[kw][1]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[t][i], 0, 0, 0);
    });
    }
    wait_vm_barrier_n(issued);   // tile kt+1 has landed; this iteration's pieces (tile kt+2) may stay in flight
  };
  for (int kt = kt_begin; kt < kt_end; kt += 3) {
    tile_body(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < kt_end) tile_body(std::integral_constant<int, 1>{}, kt + 1);
    if (kt + 2 < kt_end) tile_body(std::integral_constant<int, 2>{}, kt + 2);
  }

  const int64_t slice_stride = (int64_t)9 * p.Cout * p.Cin;
  const int ci = ci0 + wave * 16 + fr;
  if (ci < p.Cin) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + i * 16 + fg * 4 + r;
          p.ws[slice * slice_stride + ((int64_t)t * p.Cout + co) * p.Cin + ci] = acc[t][i][r];
        }
  }
}

// ===================================================================================================================
// 128-channel variant of the fast path.  Experiments on the 64 x 64 kernel above (MFMAs removed / DMA removed / reads
// removed) showed it is bound by the global->LDS fill: 16 KB of DMA per K-tile and workgroup, 7.5 TB/s chip-wide = the
// measured LDS-DMA ceiling, with the MFMA pipes idle a third of the time.  Here a workgroup is 8 waves and owns
// 128 output channels x 64 input channels x 9 taps: the halo tile of x (the larger, 1.9x over-fetched operand) is staged
// once for twice the MFMA work, 10 KB of fill per 64x64x9 unit instead of 16.  Wave (wm, wn) = 64 channels of dy x 16
// channels of x; the dy tile is kept as two 32 x 64-channel blocks so that addressing, swizzle and the conflict-free
// transposed reads are those of the kernel above.
// S2 = the 2x2 / stride 2 / pad 0 geometry (ConvTranspose2d(k2,s2) weight gradient): four taps, the x tile is the 8 x 16
// fine pixels under the 4 x 8 K-tile, stored DE-INTERLEAVED - LDS block (fy, kw) holds the eight pixels (fy, 2kx + kw) -
// so that the eight tile pixels of a transposed read are eight consecutive LDS rows for every tap, exactly as in the
// 3x3 case (stride-2 rows would all start on the same 32 banks), and a tap is an immediate offset of kh*2048 + kw*1024.
template <int TH, int TW, int NS, bool S2 = false>
__global__ __launch_bounds__(512, 2) void wgrad_fast128_kernel(WgradArgs p) {
  static_assert(TH * TW == 32, "a K-tile is 32 output pixels");
  constexpr int ROWB = 128;
  constexpr int HP = (TW + 2 + 7) / 8 * 8;
  constexpr int HR = TH + 2;
  constexpr int NTAPS = S2 ? 4 : 9;
  static_assert(!S2 || (TH == 4 && TW == 8), "the stride-2 layout is written for 4 x 8 K-tiles");
  constexpr int DYH_BYTES = 32 * ROWB, DY_BYTES = 2 * DYH_BYTES, X_ROWS = S2 ? 128 : HR * HP, X_BYTES = X_ROWS * ROWB;
  constexpr int STAGE = DY_BYTES + X_BYTES;
  constexpr int XP = X_ROWS / 8;                   // halo DMA pieces
  constexpr int XJ = (XP + 7) / 8;                 // ... per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tile = blockIdx.x;
  const int tn = tile % p.ntile_n, tm = tile / p.ntile_n;
  const int co0 = tm * 128, ci0 = tn * 64;
  const int slice = blockIdx.y;
  const int kt_begin = slice * p.per_slice;
  int kt_end = kt_begin + p.per_slice;
  if (kt_end > p.ktiles) kt_end = p.ktiles;

  f32x4 acc[NTAPS][4];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int cin_rem = p.Cin - ci0;
  const int rr = lane >> 3, pc = lane & 7;
  auto swzc = [](int row, int chunk) { return chunk ^ (((row >> 1) & 3) << 1); };

  // ---- per-lane DMA sources.  dy: piece `wave` = channel half wm, tile pixels (wave&3)*8 + rr.  halo: pieces wave + 8j.
  const int kdy = wn * 8 + rr;
  const int kdy_y = kdy / TW, kdy_x = kdy % TW;
  const int off_dy = ((kdy_y * p.Wo + kdy_x) * (int)p.dy_ld + co0 + wm * 64) * 2 + swzc(kdy, pc) * 16;
  int off_x[XJ], hyx[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int r = (wave + 8 * j) * 8 + rr;
    // tile-relative source pixel of LDS row r.  3x3: halo row r / HP, column r % HP; stride 2: block (fy, kw) = piece index
    const int hy = S2 ? (wave + 8 * j) >> 1 : r / HP, hx = S2 ? 2 * rr + ((wave + 8 * j) & 1) : r - (r / HP) * HP;
    const int c = swzc(r, pc);
    const bool ok = (wave + 8 * j) < XP && (S2 || hx < TW + 2) && c * 8 < cin_rem;
    off_x[j] = ((hy * p.W + hx) * (int)p.x_ld + ci0) * 2 + c * 16;
    hyx[j] = ok ? ((hy << 8) | hx) : -1;
  }
  const int my_pieces = 1 + (XP - wave + 7) / 8;
  const int tiles_per_img = p.tiles_x * p.tiles_y;

  // K-tiles are issued strictly in sequence (each wave issues every tile once): the coordinates of the next one are kept
  // incrementally - two integer divisions per issue were a third of this kernel's scalar instruction stream, which is as
  // long as its MFMA stream (SQ_ACTIVE_INST_ANY 47 % of the wave time)
  int ib, ity, itx;
  {
    ib = kt_begin / tiles_per_img;
    const int tr = kt_begin - ib * tiles_per_img;
    ity = tr / p.tiles_x;
    itx = tr - ity * p.tiles_x;
  }
  // byte offsets of the current tile ROW (tx0 = 0) inside dy and x, recomputed only when the row changes
  int64_t dy_row, x_row;
  auto set_row = [&]() {
    dy_row = ((((int64_t)ib * p.Ho + ity * TH) * p.Wo) * p.dy_ld) * 2;
    x_row = S2 ? ((((int64_t)ib * p.H + ity * TH * 2) * p.W) * p.x_ld) * 2 : ((((int64_t)ib * p.H + ity * TH - 1) * p.W - 1) * p.x_ld) * 2;
  };
  set_row();
  const char* zsrc = reinterpret_cast<const char*>(&g_wzero16);
  asm volatile("" : "+s"(zsrc));   // keep the zero word's address in SGPRs (otherwise re-materialised pc-relative per use)
  auto issue_tile = [&](int st) {
    const int oy = ity * TH, tx0 = itx * TW;
    const unsigned sdy = lds0 + st * STAGE;
    const unsigned sx = sdy + DY_BYTES;
    const char* dyrow = p.dy + dy_row + (int64_t)(tx0 * (int)p.dy_ld * 2);
    const char* src = (oy + kdy_y < p.Ho && tx0 + kdy_x < p.Wo) ? dyrow + off_dy : zsrc;
    dma16(src, sdy + wave * 1024);
    const int iy0 = S2 ? 2 * oy : oy - 1, ix0 = S2 ? 2 * tx0 : tx0 - 1;
    const char* xorg = p.x + x_row + (int64_t)((S2 ? 2 * tx0 : tx0) * (int)p.x_ld * 2);   // may lie before the tensor: only dereferenced for valid pixels
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      if (wave + 8 * j < XP) {
        const int hy = hyx[j] >> 8, hx = hyx[j] & 255;
        const bool ok = hyx[j] >= 0 && (unsigned)(iy0 + hy) < (unsigned)p.H && (unsigned)(ix0 + hx) < (unsigned)p.W;
        const char* s2 = ok ? xorg + off_x[j] : zsrc;
        dma16(s2, sx + (wave + 8 * j) * 1024);
      }
    }
    if (++itx == p.tiles_x) {
      itx = 0;
      if (++ity == p.tiles_y) { ity = 0; ++ib; }
      set_row();
    }
  };

  const int q = fr >> 2, pp = fr & 3;
  const int k0 = 4 * fg + q, k1 = 16 + 4 * fg + q;
  auto taddr = [&](int row, int cc) -> unsigned { return row * ROWB + ((cc ^ ((row >> 1) & 3)) << 5) + pp * 8; };
  unsigned ad[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { ad[i][0] = wm * DYH_BYTES + taddr(k0, i); ad[i][1] = wm * DYH_BYTES + taddr(k1, i); }
  // LDS rows of the two tile pixels at tap (0,0): 3x3 = halo row; stride 2 = ky*32 + kx (blocks (2ky+kh, kw) of 8 rows)
  const int h0 = S2 ? (k0 / TW) * 32 + k0 % TW : (k0 / TW) * HP + k0 % TW, h1 = S2 ? (k1 / TW) * 32 + k1 % TW : (k1 / TW) * HP + k1 % TW;
  constexpr int NAX = S2 ? 1 : 3;
  unsigned ax[NAX][2];
#pragma unroll
  for (int kw = 0; kw < NAX; ++kw) { ax[kw][0] = taddr(h0 + kw, wn); ax[kw][1] = taddr(h1 + kw, wn); }

  // ring of NS stages, NS-1 K-tiles ahead: the fetch of a K-tile (HBM / Infinity-Cache latency) has NS-2 whole
  // iterations to land before it is waited for
  constexpr int D = NS - 1;
  wg_static_for<0, D>([&](auto d_) {
    constexpr int d = decltype(d_)::value;
    if (kt_begin + d < kt_end) issue_tile(d);
  });
  wait_vm_barrier_n(0);

  auto tile_body = [&](auto st_, int kt) {
    constexpr int ST = decltype(st_)::value;
    // vmcnt: the pieces of tiles kt+2 .. kt+D may stay in flight, tile kt+1 must have landed
    int issued = 0;
    // Waves 0-3 issue their DMA at the start of the iteration, their SIMD partners (waves 4-7) at its end.  A wave whose
    // global_load_lds blocks at issue (the vector-memory pipeline backs up under real traffic: with every DMA pointed at
    // one zero word this kernel runs 1.6x faster) cannot issue MFMAs either, and partners running the same program
    // would block together.  Measured (bench.py, wgrad family): all at start 850, all at end 846, partners at taps
    // 2 / 4 / 6 / after tap 8: 878 / 905 / 942 / 950 TFLOP/s.
    const bool late = wave >= 4;
    const bool more = kt + D < kt_end;
    if (!late && more) issue_tile((ST + D) % NS);
    { int ahead = kt_end - (kt + 2); ahead = ahead < 0 ? 0 : (ahead > D - 1 ? D - 1 : ahead); issued = ahead * my_pieces; }
    const char* sb = smem + ST * STAGE;
    bf16x8 af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = tr_pair(sb, ad[i][0], ad[i][1]);
    wg_static_for<0, NTAPS>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      constexpr int kh = S2 ? t / 2 : t / 3, kw = S2 ? t % 2 : t - kh * 3;
      const char* xb = sb + DY_BYTES + (S2 ? kh * 2048 + kw * 1024 : kh * HP * ROWB);
      const bf16x8 bf = tr_pair(xb, ax[S2 ? 0 : kw][0], ax[S2 ? 0 : kw][1]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[t][i], 0, 0, 0);
      if constexpr (t == NTAPS - 1) {
        if (late && more) issue_tile((ST + D) % NS);
      }
    });
    wait_vm_barrier_n(issued);
  };
  for (int kt = kt_begin; kt < kt_end; kt += NS) {
    wg_static_for<0, NS>([&](auto s_) {
      constexpr int s = decltype(s_)::value;
      if (kt + s < kt_end) tile_body(std::integral_constant<int, s>{}, kt + s);
    });
  }

  const int64_t slice_stride = (int64_t)NTAPS * p.Cout * p.Cin;
  const int ci = ci0 + wn * 16 + fr;
  if (ci < p.Cin) {
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wm * 64 + i * 16 + fg * 4 + r;
          p.ws[slice * slice_stride + ((int64_t)t * p.Cout + co) * p.Cin + ci] = acc[t][i][r];
        }
  }
}

// dw[co][ci][tap] (OIHW, ci < Cin_real) = sum over slices, fixed order.  One (co, ci) pair and all TAPS taps per lane: the
// slab reads are coalesced along ci and the TAPS results leave as one contiguous run of the OIHW tensor.  KS lanes share a
// pair (each sums every KS-th slice; merged in fixed order through LDS) so that small layers with many slices still fill
// the chip.
template <int TAPS, int KS>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslice,
                                                          int Cout, int Cin, int Cin_real) {
  constexpr int PAIRS = 256 / KS;
  __shared__ float sh[KS > 1 ? KS : 1][PAIRS][TAPS + 1];
  const int64_t total = (int64_t)Cout * Cin_real;
  const int64_t tap_stride = (int64_t)Cout * Cin;
  const int64_t slice_stride = (int64_t)TAPS * tap_stride;
  const int pl = threadIdx.x % PAIRS, kg = threadIdx.x / PAIRS;
  for (int64_t base = (int64_t)blockIdx.x * PAIRS; base < total; base += (int64_t)gridDim.x * PAIRS) {
    const int64_t id = base + pl;
    const bool ok = id < total;
    float s[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) s[t] = 0.f;
    if (ok) {
      const int ci = (int)(id % Cin_real);
      const int co = (int)(id / Cin_real);
      const float* src = ws + (int64_t)co * Cin + ci;
      // four slices' loads in flight per thread (a single slice per trip left the reduce latency-bound: 0.5 TB/s over the
      // 512 partial slabs of a 64 x 64 layer); summed in slice order all the same
      constexpr int UK = TAPS > 1 ? 4 : 8;
      int k = kg;
      for (; k + (UK - 1) * KS < nslice; k += UK * KS) {
        float v[UK][TAPS];
#pragma unroll
        for (int u = 0; u < UK; ++u) {
#pragma unroll
          for (int t = 0; t < TAPS; ++t) v[u][t] = src[(int64_t)(k + u * KS) * slice_stride + t * tap_stride];
        }
#pragma unroll
        for (int u = 0; u < UK; ++u) {
#pragma unroll
          for (int t = 0; t < TAPS; ++t) s[t] += v[u][t];
        }
      }
      for (; k < nslice; k += KS) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) s[t] += src[(int64_t)k * slice_stride + t * tap_stride];
      }
    }
    if constexpr (KS > 1) {
      __syncthreads();
#pragma unroll
      for (int t = 0; t < TAPS; ++t) sh[kg][pl][t] = s[t];
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          float a = sh[0][pl][t];
          for (int g = 1; g < KS; ++g) a += sh[g][pl][t];
          s[t] = a;
        }
      }
    }
    if (ok && kg == 0) {
      float* dst = dw + id * TAPS;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) dst[t] = s[t];
    }
  }
}

constexpr int kXRows = 160;   // halo-tile capacity of the generic kernel in LDS rows (3x3 / stride 2 needs 9 x 17 = 153)
struct WgradPlan {
  int TH, TW, HR, HC, tiles_x, tiles_y, ktiles, nslice, per_slice, tiles_mn;
};

// fast: 0 = generic kernel, 1 = 64 x 64 fast path, 2 = 128 x 64 fast path (8 waves, one workgroup per CU)
static WgradPlan make_plan(int B, int Ho, int Wo, int Cout, int Cin, int KH, int KW, int stride, int fast = 0, int stride_w = 0) {
  WgradPlan pl;
  if (stride_w <= 0) stride_w = stride;
  int tw = 1;
  while (tw < Wo && tw < 32) tw <<= 1;
  pl.TW = tw;
  pl.TH = 32 / tw;
  if (fast) { pl.TW = 8; pl.TH = 4; }   // the bf16 3x3 fast path works on 4 x 8-pixel K-tiles
  // the halo tile of a K-tile has to fit kXRows LDS rows: strided / tall filters take the 32-pixel rectangle with the
  // smallest halo (3x3 / stride 2: 8 x 4 pixels -> 17 x 9 rows)
  if (!fast && ((pl.TH - 1) * stride + KH) * ((pl.TW - 1) * stride_w + KW) > kXRows) {
    int best = 1 << 30;
    for (int t = 1; t <= 32; t <<= 1) {
      const int h = ((32 / t - 1) * stride + KH) * ((t - 1) * stride_w + KW);
      if (h < best) { best = h; pl.TW = t; pl.TH = 32 / t; }
    }
  }
  pl.HR = (pl.TH - 1) * stride + KH;
  pl.HC = (pl.TW - 1) * stride_w + KW;
  pl.tiles_x = (int)cdiv(Wo, pl.TW);
  pl.tiles_y = (int)cdiv(Ho, pl.TH);
  pl.ktiles = B * pl.tiles_x * pl.tiles_y;
  pl.tiles_mn = (fast == 2 ? Cout / 128 : (int)cdiv(Cout, 64)) * (int)cdiv(Cin, 64);
  // ~2 resident workgroups per CU (one for the 8-wave kernel); fewer slices = less partial-slab traffic
  int want = (int)cdiv(fast == 2 ? 256 : 512, pl.tiles_mn);
  if (want > pl.ktiles) want = pl.ktiles;
  if (want < 1) want = 1;
  pl.per_slice = (int)cdiv(pl.ktiles, want);
  pl.nslice = (int)cdiv(pl.ktiles, pl.per_slice);
  return pl;
}

template <typename T, int NT>
static int launch(const WgradArgs& a, const WgradPlan& pl, hipStream_t st) {
  constexpr int ES = sizeof(T);
  constexpr int NS = ES == 2 ? 3 : 2;   // bf16: two K-tiles in flight; f32 (parity path): one
  const size_t lds = (size_t)NS * (32 + kXRows) * 64 * ES;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, NT, NS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid((unsigned)pl.tiles_mn, (unsigned)pl.nslice);
  hipLaunchKernelGGL((wgrad_kernel<T, NT, NS>), grid, dim3(256), lds, st, a);
  CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad");
  return CVCS_OK;
}

}  // namespace cvcs

using namespace cvcs;

// bf16, 3x3 / stride 1 / pad 1, maps of at least 4 x 8 pixels, 32-bit per-image offsets
static bool fast_shape(int dtype, int KH, int KW, int stride, int pad, int H, int W, int Ho, int Wo, int64_t x_ld, int64_t dy_ld) {
  return dtype == CVCS_BF16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && Ho >= 4 && Wo >= 8 &&
         (int64_t)H * W * x_ld * 2 < (1ll << 31) && (int64_t)Ho * Wo * dy_ld * 2 < (1ll << 31);
}
// bf16, 2x2 / stride 2 / pad 0 (the ConvTranspose2d weight gradient), 128-channel tiles
static bool s2_shape(const cvcs_wgrad_desc* d) {
  static const int on = getenv("CVCS_WGRAD_S2") ? atoi(getenv("CVCS_WGRAD_S2")) : 1;   // tuning knob
  return on && d->dtype == CVCS_BF16 && d->KH == 2 && d->KW == 2 && d->stride == 2 && d->pad == 0 && d->Ho >= 4 && d->Wo >= 8 &&
         d->Cout % 128 == 0 && d->H == 2 * d->Ho && d->W == 2 * d->Wo &&
         (int64_t)d->H * d->W * d->x_ld * 2 < (1ll << 31) && (int64_t)d->Ho * d->Wo * d->dy_ld * 2 < (1ll << 31);
}

// ===================================================================================================================
// Thin weight gradient (bf16, 3x3 / stride 1 / pad 1, Cout and Cin in {16, 32}): the light U-Net decoder stages at 256^2 / 512^2 pixels.
// These layers are HBM-bound (a 16 -> 16 layer at 512^2 x 32 tiles reads 0.54 GB for 39 GFLOP); on the generic kernel above they ran
// 4-5 x over their HBM floor (64 x 64 channel tiles that are 15/16 padding, 32-pixel K-tiles with a barrier each).  Here:
//   * a workgroup takes 4 x 64-pixel tiles of one image (grid-stride over all tiles); the dy tile and the 6 x 66-pixel x halo are
//     staged by LDS-DMA as pixel-major rows of C * 2 bytes (the tensors' own layout: whole 1 KiB pieces are contiguous runs);
//   * wave w owns tile row w: K = 32 consecutive pixels of the row per MFMA, operands by transposed LDS reads (ds_read_b64_tr_b16 turns
//     4 pixels x 16 channels into the k-major fragment; a tap is a pixel offset into the halo rows) - one dy fragment pair serves all
//     nine taps;
//   * every wave keeps its own [9][Cout][Cin] accumulators over all its tiles and writes ONE partial slab (slice = 4 * workgroup + wave);
//     the fixed-order reduce kernel above sums the slabs - no atomics, bitwise reproducible.
// Single-buffered: several workgroups per CU overlap each other's DMA waits (21.5 KB of LDS for 16 / 16 channels, 42 KB for 32 / 32).
struct ThinWgArgs {
  const char* x; const char* dy; float* ws;
  int64_t x_ld, dy_ld;       // elements
  int B, H, W, Cin, Cout;
  int tiles_x, tiles_y, ntiles;
};
constexpr int kThinTH = 4, kThinTW = 64, kThinHC = kThinTW + 2, kThinHR = kThinTH + 2;

__device__ uint4 g_tzero16;

// UP instance (cvcs_wgrad_desc.x_up2): x is the map at HALF the resolution and the layer's input its bilinear x2 up-sampling, interpolated LDS -> LDS
// from the 4 x 34-pixel patch under the tile's halo (common.h: up2_halo_chunk; the forward's conv3x3_thin_kernel<.., UP> stages the same way)
template <int CO, int CI, bool UP = false>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(ThinWgArgs p) {
  constexpr int DYB = kThinTH * kThinTW * CO * 2;                      // bytes of the dy tile
  constexpr int XPIX = kThinHR * kThinHC;                              // 396 halo pixels
  constexpr int XB = (XPIX * CI * 2 + 1023) / 1024 * 1024;
  constexpr int NDY = DYB / 1024, NX = XB / 1024;
  constexpr int MB = CO / 16, NB = CI / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int tpi = p.tiles_x * p.tiles_y;
  f32x4 acc[9][MB][NB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane position inside a DMA piece: byte offset 16 * lane of 1 KiB = (pixel, 16-byte chunk) of a pixel-major image
  constexpr int DPP = 1024 / (CO * 2), DCH = CO * 2 / 16;             // dy pixels per piece, chunks per pixel
  constexpr int XPP = 1024 / (CI * 2), XCH = CI * 2 / 16;
  const int dpx = lane / DCH, dch = lane % DCH;
  const int xpx = lane / XCH, xch = lane % XCH;
  // fragment addresses: lane (i = fr, g = fg) of a transposed read supplies row q = i >> 2 (a pixel), 8 bytes at channel 4 * (i & 3)
  const int q = fr >> 2, pp = fr & 3;

  for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    const int b = t / tpi, tr = t - b * tpi;
    const int ty0 = (tr / p.tiles_x) * kThinTH, tx0 = (tr % p.tiles_x) * kThinTW;
    // ---- stage the tile (every wave its share of the pieces)
    for (int pi = wave; pi < NDY; pi += 4) {
      const int k = pi * DPP + dpx;                                    // tile pixel: row k / 64, column k % 64
      const int oy = ty0 + (k >> 6), ox = tx0 + (k & 63);
      const char* src = (oy < p.H && ox < p.W) ? p.dy + ((((int64_t)b * p.H + oy) * p.W + ox) * p.dy_ld) * 2 + dch * 16
                                               : reinterpret_cast<const char*>(&g_tzero16);
      dma16(src, lds0 + pi * 1024);
    }
    if constexpr (UP) {
      constexpr int PPIX = kUpPatchRows * kUpPatchCols, NP = (PPIX * CI * 2 + 1023) / 1024;
      const int ly0 = (ty0 >> 1) - 1, lx0 = (tx0 >> 1) - 1, Hl = p.H >> 1, Wl = p.W >> 1;
      for (int pi = wave; pi < NP; pi += 4) {                          // the low-resolution patch behind the halo region (clamped coordinates)
        const int r = pi * XPP + xpx;
        const int pr = r / kUpPatchCols, pc = r - pr * kUpPatchCols;
        int ly = ly0 + pr, lx = lx0 + pc;
        ly = ly < 0 ? 0 : (ly >= Hl ? Hl - 1 : ly);
        lx = lx < 0 ? 0 : (lx >= Wl ? Wl - 1 : lx);
        const char* src = r < PPIX ? p.x + ((((int64_t)b * Hl + ly) * Wl + lx) * p.x_ld) * 2 + xch * 16 : reinterpret_cast<const char*>(&g_tzero16);
        dma16(src, lds0 + DYB + XB + pi * 1024);
      }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      for (int id = tid; id < XPIX * XCH; id += 256) {
        const int hp = id / XCH, ch = id - hp * XCH;
        *reinterpret_cast<uint4*>(smem + DYB + hp * (CI * 2) + ch * 16) = up2_halo_chunk<CI>(smem + DYB + XB, hp, ch, ty0, tx0, p.H, p.W);
      }
      __syncthreads();
    } else {
    for (int pi = wave; pi < NX; pi += 4) {
      const int r = pi * XPP + xpx;                                    // halo pixel: row r / 66, column r % 66
      const int hy = r / kThinHC, hx = r - hy * kThinHC;
      const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
      const char* src = (r < XPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                            ? p.x + ((((int64_t)b * p.H + iy) * p.W + ix) * p.x_ld) * 2 + xch * 16
                            : reinterpret_cast<const char*>(&g_tzero16);
      dma16(src, lds0 + DYB + pi * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // ---- wave `wave` = tile row `wave`: two K-steps of 32 pixels
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int px0 = wave * kThinTW + ks * 32 + 8 * fg + q;           // dy tile pixel of this lane's first transposed read
      bf16x8 af[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i)
        af[i] = tr_pair(smem, px0 * (CO * 2) + i * 32 + pp * 8, (px0 + 4) * (CO * 2) + i * 32 + pp * 8);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - kh * 3;
        const int hp = (wave + kh) * kThinHC + ks * 32 + 8 * fg + q + kw;   // halo pixel under tile pixel px0 for this tap
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const bf16x8 bf = tr_pair(smem, DYB + hp * (CI * 2) + j * 32 + pp * 8, DYB + (hp + 4) * (CI * 2) + j * 32 + pp * 8);
#pragma unroll
          for (int i = 0; i < MB; ++i) acc[tap][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[tap][i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();            // every wave is done with the tile before the next one is staged over it
  }
  // ---- this wave's partial slab: ws[slice][tap][co][ci]; D layout: row (co) = fg*4 + r, col (ci) = fr
  float* ws = p.ws + (int64_t)(blockIdx.x * 4 + wave) * 9 * CO * CI;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ws[((int64_t)tap * CO + i * 16 + fg * 4 + r) * CI + j * 16 + fr] = acc[tap][i][j][r];
}

// The same layer under a WIDE input (Cin a multiple of 128 under Cout = 16 | 32: the 128 -> 32 stage of the light decoder, 256^2 pixels).  As
// 32-channel groups of the kernel above it took one launch per group - the thin dy re-read every time, and a 64-byte slice of every 256-byte
// pixel fetched as whole lines: 4 x 0.54 GB for 0.67 GB of operands.  Here the four waves of a workgroup own one 32-channel group EACH and all
// walk the same 4 x 32-pixel tile (K = one tile row per MFMA): the dy tile and the 6 x 34-pixel halo of all 128 channels are staged once.
// A halo pixel is 256 bytes = the whole bank period, so its sixteen 16-byte chunks are stored XOR-swizzled by a key of the pixel index
// (chunk c of halo pixel p sits at chunk c ^ key(p)): the sixteen pixels of a transposed read then spread over all banks (two-way instead of
// sixteen-way).  Slabs ws[group][workgroup][tap][co][32], summed per group by the reduce kernel below (blockIdx.y = group).
constexpr int kWideTH = 4, kWideTW = 32, kWideHC = kWideTW + 2, kWideHR = kWideTH + 2;
__device__ __forceinline__ unsigned wide_key(unsigned p) { return 2u * ((p & 3u) | (((p >> 3) & 1u) << 2)); }

template <int CO>
__global__ __launch_bounds__(256) void wgrad_thin_wide_kernel(ThinWgArgs p) {
  constexpr int DYB = kWideTH * kWideTW * CO * 2;                      // 4 | 8 KiB of dy
  constexpr int XPIX = kWideHR * kWideHC;                              // 204 halo pixels x 256 bytes = 51 pieces of 1 KiB exactly
  constexpr int NDY = DYB / 1024, NX = XPIX * 256 / 1024;
  constexpr int MB = CO / 16, NB = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int tpi = p.tiles_x * p.tiles_y;
  const int ci0 = blockIdx.y * 128;                                    // this launch row's 128 input channels
  f32x4 acc[9][MB][NB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int DPP = 1024 / (CO * 2), DCH = CO * 2 / 16;             // dy pixels per piece, chunks per pixel
  const int dpx = lane / DCH, dch = lane % DCH;
  const int xs = lane >> 4, xc = lane & 15;                            // x: pixel slot of the piece, chunk POSITION in LDS
  const int q = fr >> 2, pp = fr & 3;

  for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    const int b = t / tpi, tr = t - b * tpi;
    const int ty0 = (tr / p.tiles_x) * kWideTH, tx0 = (tr % p.tiles_x) * kWideTW;
    for (int pi = wave; pi < NDY; pi += 4) {
      const int k = pi * DPP + dpx;                                    // tile pixel: row k / 32, column k % 32
      const int oy = ty0 + (k >> 5), ox = tx0 + (k & 31);
      const char* src = (oy < p.H && ox < p.W) ? p.dy + ((((int64_t)b * p.H + oy) * p.W + ox) * p.dy_ld) * 2 + dch * 16
                                               : reinterpret_cast<const char*>(&g_tzero16);
      dma16(src, lds0 + pi * 1024);
    }
    for (int pi = wave; pi < NX; pi += 4) {
      const int r = pi * 4 + xs;                                       // halo pixel: row r / 34, column r % 34
      const int hy = r / kWideHC, hx = r - hy * kWideHC;
      const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
      const unsigned c = (unsigned)xc ^ wide_key((unsigned)r);        // the chunk that belongs at this position
      const char* src = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                            ? p.x + ((((int64_t)b * p.H + iy) * p.W + ix) * p.x_ld + ci0) * 2 + c * 16
                            : reinterpret_cast<const char*>(&g_tzero16);
      dma16(src, lds0 + DYB + pi * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int r = 0; r < kWideTH; ++r) {
      const int px0 = r * kWideTW + 8 * fg + q;                        // dy tile pixel of this lane's first transposed read
      bf16x8 af[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i)
        af[i] = tr_pair(smem, px0 * (CO * 2) + i * 32 + pp * 8, (px0 + 4) * (CO * 2) + i * 32 + pp * 8);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - kh * 3;
        const unsigned hp = (unsigned)((r + kh) * kWideHC + 8 * fg + q + kw);   // halo pixel under tile pixel px0 for this tap
        const unsigned k0 = wide_key(hp), k1 = wide_key(hp + 4);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const unsigned c = (unsigned)(wave * 4 + j * 2 + (pp >> 1));           // 16-byte chunk of this lane's four channels
          const bf16x8 bf = tr_pair(smem, DYB + hp * 256 + ((c ^ k0) << 4) + (pp & 1) * 8, DYB + (hp + 4) * 256 + ((c ^ k1) << 4) + (pp & 1) * 8);
#pragma unroll
          for (int i = 0; i < MB; ++i) acc[tap][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[tap][i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();            // every wave is done with the tile before the next one is staged over it
  }
  // ---- this wave's partial slab of its channel group: ws[group][workgroup][tap][co][32]
  float* ws = p.ws + ((int64_t)(blockIdx.y * 4 + wave) * gridDim.x + blockIdx.x) * 9 * CO * 32;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ws[((int64_t)tap * CO + i * 16 + fg * 4 + r) * 32 + j * 16 + fr] = acc[tap][i][j][r];
}

// sum of the thin kernel's S partial slabs ws[s][tap][co][ci] -> dw[co][ci][tap] (OIHW), fixed order: 1024 threads = 16 slice lanes x 64
// consecutive slab elements (a wave reads 256 contiguous bytes of one slab); slice lane l sums slabs l, l + 16, ... eight loads in
// flight, then the 16 lanes are added in order.  (The generic reduce kernel above walks the slices with 16 lanes per OUTPUT PAIR: with
// thousands of slabs of a few thousand elements it took 0.3 ms - 3 x the thin kernel itself.)
__global__ __launch_bounds__(1024) void wgrad_thin_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int S, int CO, int CI, int CI_total = 0,
                                                                 int ci0 = 0) {
  __shared__ float sh[16][64];
  const int n = 9 * CO * CI;
  ws += (int64_t)blockIdx.y * S * n;      // (the wide-input kernel: one launch row per 32-channel group, its S slabs back to back)
  ci0 += blockIdx.y * CI;
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), l = threadIdx.x >> 6;
  float a = 0.f;
  if (e < n) {
    int s = l;
    for (; s + 7 * 16 < S; s += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ws[(int64_t)(s + u * 16) * n + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; s < S; s += 16) a += ws[(int64_t)s * n + e];
  }
  sh[l][threadIdx.x & 63] = a;
  __syncthreads();
  if (l == 0 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][threadIdx.x];
    const int ci = e % CI, r = e / CI, co = r % CO, tap = r / CO;
    dw[((int64_t)co * (CI_total ? CI_total : CI) + ci0 + ci) * 9 + tap] = t;
  }
}

static bool thin_wgrad_shape(const cvcs_wgrad_desc* d) {
  static const int on = getenv("CVCS_WGRAD_THIN") ? atoi(getenv("CVCS_WGRAD_THIN")) : 1;   // tuning knob
  const bool special = d->aniso != 0 || d->x_row_pitch != 0 || d->x_img_pitch != 0;
  return on && d->dtype == CVCS_BF16 && !special && d->dil <= 1 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
         (d->Cout == 16 || d->Cout == 32) && (d->Cin == 16 || d->Cin == 32 || (d->Cin % 32 == 0 && d->Cin <= 256)) && d->Cin_real == d->Cin &&
         d->H == d->Ho && d->W == d->Wo && d->H >= 4 && d->W >= 32 &&
         (!d->x_up2 || (d->Cout == 16 && d->Cin == 32 && d->H % 2 == 0 && d->W % 2 == 0));
}
static int thin_wgrad_groups(const cvcs_wgrad_desc* d) {
  const int64_t ntiles = (int64_t)d->B * cdiv(d->H, kThinTH) * cdiv(d->W, kThinTW);
  return (int)(ntiles < 768 ? ntiles : 768);       // three workgroups per CU
}

static bool thin_wide_shape(const cvcs_wgrad_desc* d) {
  static const int on = getenv("CVCS_WGRAD_THIN_WIDE") ? atoi(getenv("CVCS_WGRAD_THIN_WIDE")) : 1;   // tuning knob
  return on && thin_wgrad_shape(d) && d->Cin % 128 == 0;
}
static int thin_wide_groups(const cvcs_wgrad_desc* d) {
  const int64_t ntiles = (int64_t)d->B * cdiv(d->H, kWideTH) * cdiv(d->W, kWideTW);
  return (int)(ntiles < 512 ? ntiles : 512);       // two workgroups per CU (59 KiB of LDS each)
}

static int fast_path(const cvcs_wgrad_desc* d) {
  if (d->Cout % 64 != 0) return 0;   // a last row tile that is not full: the generic kernel masks it
  static const int force64 = getenv("CVCS_WGRAD_64") ? atoi(getenv("CVCS_WGRAD_64")) : 0;   // tuning knob
  if (s2_shape(d)) return 2;
  if (!fast_shape(d->dtype, d->KH, d->KW, d->stride, d->pad, d->H, d->W, d->Ho, d->Wo, d->x_ld, d->dy_ld)) return 0;
  return (d->Cout % 128 == 0 && !force64) ? 2 : 1;
}

// db[co] = sum over the K-slices of the partial bias gradients: 8 slice lanes x 32 channels per workgroup, every lane sums slices
// sl, sl + 8, ... four loads at a time, the eight lanes are merged in lane order (one thread per channel walking up to 512 slices took 37 us)
__global__ __launch_bounds__(256) void wgrad_bias_reduce_kernel(const float* __restrict__ part, int nslice, int Cout, float* __restrict__ db) {
  __shared__ float sh[8][33];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int co = blockIdx.x * 32 + cl;
  float a = 0.f;
  if (co < Cout) {
    int s = sl;
    for (; s + 24 < nslice; s += 32) {
      const float v0 = part[(int64_t)s * Cout + co], v1 = part[(int64_t)(s + 8) * Cout + co];
      const float v2 = part[(int64_t)(s + 16) * Cout + co], v3 = part[(int64_t)(s + 24) * Cout + co];
      a += v0; a += v1; a += v2; a += v3;
    }
    for (; s < nslice; s += 8) a += part[(int64_t)s * Cout + co];
  }
  sh[sl][cl] = a;
  __syncthreads();
  if (sl == 0 && co < Cout) {
    float t = sh[0][cl];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += sh[q][cl];
    db[co] = t;
  }
}

static int wgrad_xcd_order() {
  static const int on = getenv("CVCS_XCD_ORDER") ? atoi(getenv("CVCS_XCD_ORDER")) : 1;   // tuning knob
  return on;
}

// 1x1 / stride 1 GEMM path: tile shape and split-K plan
struct GemmPlan { int BM, BN, ktiles, per_slice, nslice, tiles_mn, ntile_n; };
static bool gemm_shape(int KH, int KW, int stride, int Cout, int Cin) {
  static const int on = getenv("CVCS_WGRAD_GEMM") ? atoi(getenv("CVCS_WGRAD_GEMM")) : 1;   // tuning knob
  // channel counts: multiples of 8 (a 16-byte chunk); the last tile of a count that is no multiple of the tile is masked
  // (chunks beyond the last channel read zeros, their outputs are not stored) - the 96 * 2^k widths of Swin
  return on && KH == 1 && KW == 1 && stride == 1 && Cout % 8 == 0 && Cin % 8 == 0 && (Cout > 64 || Cin > 64 || (Cout == 64 && Cin == 64));
}
// 1x1 / stride 2 / pad 0 on even maps (the projection shortcuts of the down-sampling ResNet blocks): the same GEMM over the OUTPUT pixels, x read
// at the even pixels in place - no gathered copy of the input (round 3 gathered the even pixels first: cvcs_im2col with a 1x1 window)
static bool gemm_s2_shape(const cvcs_wgrad_desc* d) {
  return d->dtype == CVCS_BF16 && d->KH == 1 && d->KW == 1 && d->stride == 2 && d->pad == 0 && d->dil <= 1 && !d->aniso && !d->x_row_pitch && !d->x_img_pitch &&
         d->H == 2 * d->Ho && d->W == 2 * d->Wo && gemm_shape(1, 1, 1, d->Cout, d->Cin) && !d->dbias;
}
static GemmPlan gemm_plan(int B, int Ho, int Wo, int Cout, int Cin) {
  GemmPlan g;
  g.BM = Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : 128);
  g.BN = Cin % 128 == 0 ? 128 : (Cin % 64 == 0 ? 64 : 128);
  if (g.BM == 64 && g.BN == 64 && !(Cout == 64 && Cin == 64)) {           // (the 64 x 64 instance serves the 64 -> 64 layer only)
    if (Cout >= Cin) g.BM = 128; else g.BN = 128;
  }
  // 8-wave 256 x 128 tiles for LARGE problems only (>= 16 tiles of 128 x 128 and >= 64 K pixels): measured +2 % on the DeepLabV3-ResNet101
  // step (1024 x 256 and 2048 x 512 gradients over 131 072 pixels), -1 % on ResNet50-UNet (the same matrices over 32 768 pixels: fewer,
  // longer workgroups); 256 x 256 (one workgroup per CU) measured no better anywhere.  CVCS_WGRAD_GEMM_BIG: 0 = never, 2 = also 256 x 256
  static const int big = getenv("CVCS_WGRAD_GEMM_BIG") ? atoi(getenv("CVCS_WGRAD_GEMM_BIG")) : 1;
  const int64_t pixels = (int64_t)B * Ho * Wo;
  if (big >= 1 && g.BM == 128 && g.BN == 128 && Cout % 256 == 0 && (Cout / 128) * (Cin / 128) >= 16 && pixels >= 65536) g.BM = 256;
  if (big >= 2 && g.BM == 256 && Cin % 256 == 0) g.BN = 256;
  g.ktiles = (int)cdiv((int64_t)B * Ho * Wo, 32);
  g.ntile_n = (int)cdiv(Cin, g.BN);
  g.tiles_mn = (int)cdiv(Cout, g.BM) * g.ntile_n;
  int want = (int)cdiv(512, g.tiles_mn);     // two resident workgroups per CU
  if (want > g.ktiles) want = g.ktiles;
  if (want < 1) want = 1;
  g.per_slice = (int)cdiv(g.ktiles, want);
  g.nslice = (int)cdiv(g.ktiles, g.per_slice);
  return g;
}

// worst case over both kernels (the slice count of the generic and of the fast plan can differ)
extern "C" int cvcs_wgrad_slices(int B, int Ho, int Wo, int Cout, int Cin, int KH, int KW, int stride) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || Cout < 16 || Cin <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return CVCS_EINVAL;
  int n = make_plan(B, Ho, Wo, Cout, Cin, KH, KW, stride, 0).nslice;
  if (gemm_shape(KH, KW, stride, Cout, Cin)) {
    const int g = gemm_plan(B, Ho, Wo, Cout, Cin).nslice;
    n = g > n ? g : n;
  }
  if (KH == 3 && KW == 3 && stride == 1 && Ho >= 4 && Wo >= 8) {
    const int b = make_plan(B, Ho, Wo, Cout, Cin, KH, KW, stride, 1).nslice;
    n = b > n ? b : n;
    if (Cout % 128 == 0) {
      const int c = make_plan(B, Ho, Wo, Cout, Cin, KH, KW, stride, 2).nslice;
      n = c > n ? c : n;
    }
  }
  if (KH == 2 && KW == 2 && stride == 2 && Ho >= 4 && Wo >= 8 && Cout % 128 == 0) {
    const int c = make_plan(B, Ho, Wo, Cout, Cin, KH, KW, stride, 2).nslice;
    n = c > n ? c : n;
  }
  return n;
}

// Dilated 3x3 / stride 1 / pad = dil (ASPP rates 6 / 12 / 18 and the dilated last ResNet stage of DeepLabV3+): at those rates
// a K-tile's nine taps touch nine disjoint windows, so there is no halo to share - each tap is a 1x1 weight gradient between
// dy and the SHIFTED input, run on the 1x1 GEMM kernel (bf16) or the generic kernel with a signed padding (f32 parity path),
// nine launches into one [slice][tap][co][ci] partial slab, one fixed-order reduce.
static int wgrad_dilated(const cvcs_wgrad_desc* d, hipStream_t st) {
  const int es = d->dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == d->dil && !d->aniso && !d->x_row_pitch && !d->x_img_pitch,
                 "cvcs_conv2d_wgrad: dilation is built for 3x3 / stride 1 / pad = dil");
  CVCS_CHECK_ARG(d->H == d->Ho && d->W == d->Wo && d->Cin_real == d->Cin, "cvcs_conv2d_wgrad(dilated): same-size maps, no channel padding");
  CVCS_CHECK_ARG(d->x_ld >= d->Cin && d->x_ld * es % 16 == 0 && d->dy_ld >= d->Cout && d->dy_ld * es % 16 == 0 &&
                 ((uintptr_t)d->x % 16) == 0 && ((uintptr_t)d->dy % 16) == 0, "cvcs_conv2d_wgrad(dilated): views");
  WgradArgs a;
  a.tiles_mn = 1; a.xcd_order = wgrad_xcd_order(); a.bias_ws = nullptr; a.x_s2 = 0;
  a.x = (const char*)d->x; a.dy = (const char*)d->dy;
  a.x_ld = d->x_ld; a.dy_ld = d->dy_ld;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KW = 1; a.stride = 1; a.stride_w = 1;
  a.x_row_pitch = (int64_t)d->W * d->x_ld; a.x_img_pitch = (int64_t)d->H * a.x_row_pitch;
  a.slice_stride = (int64_t)9 * d->Cout * d->Cin;
  const bool gemm = d->dtype == CVCS_BF16 && gemm_shape(1, 1, 1, d->Cout, d->Cin);
  int nslice;
  if (gemm) {
    const GemmPlan g = gemm_plan(d->B, d->Ho, d->Wo, d->Cout, d->Cin);
    a.ktiles = g.ktiles; a.per_slice = g.per_slice; a.ntile_n = g.ntile_n;
    nslice = g.nslice;
    const dim3 grid((unsigned)(g.tiles_mn * g.nslice));
    a.tiles_mn = g.tiles_mn;
    for (int t = 0; t < 9; ++t) {
      a.sh_y = (t / 3 - 1) * d->dil; a.sh_x = (t % 3 - 1) * d->dil;
      a.ws = d->workspace + (int64_t)t * d->Cout * d->Cin;
      a.pad = 0; a.pad_w = 0;
#define LAUNCH_GEMM_D(BM_, BN_, WMV_)                                                                                           \
  do {                                                                                                                          \
    const int lds = 3 * ((BM_) / 64 + (BN_) / 64) * 4096;                                                                       \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_gemm_kernel<BM_, BN_, WMV_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL((wgrad_gemm_kernel<BM_, BN_, WMV_>), grid, dim3((WMV_) * 128), lds, st, a);                              \
  } while (0)
      if (g.BM == 256 && g.BN == 256) LAUNCH_GEMM_D(256, 256, 4);
      else if (g.BM == 256) LAUNCH_GEMM_D(256, 128, 4);
      else if (g.BM == 128 && g.BN == 128) LAUNCH_GEMM_D(128, 128, 2);
      else if (g.BM == 128) LAUNCH_GEMM_D(128, 64, 2);
      else if (g.BN == 64) LAUNCH_GEMM_D(64, 64, 2);
      else LAUNCH_GEMM_D(64, 128, 2);
#undef LAUNCH_GEMM_D
    }
  } else {
    WgradPlan pl = make_plan(d->B, d->Ho, d->Wo, d->Cout, d->Cin, 1, 1, 1, 0);
    a.TH = pl.TH; a.TW = pl.TW; a.HR = pl.HR; a.HC = pl.HC;
    a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.ktiles = pl.ktiles; a.per_slice = pl.per_slice;
    a.ntile_n = (int)cdiv(d->Cin, 64);
    a.sh_y = a.sh_x = 0;
    nslice = pl.nslice;
    for (int t = 0; t < 9; ++t) {
      a.pad = -(t / 3 - 1) * d->dil; a.pad_w = -(t % 3 - 1) * d->dil;   // signed padding = the tap's shift (loads are bounds-checked)
      a.ws = d->workspace + (int64_t)t * d->Cout * d->Cin;
      int rc = d->dtype == CVCS_F32 ? launch<float, 1>(a, pl, st) : launch<bf16_t, 1>(a, pl, st);
      if (rc != CVCS_OK) return rc;
    }
  }
  CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(dilated)");
  const int64_t total = (int64_t)d->Cout * d->Cin;
  int ks = 1;
  while (ks < 16 && total * ks < 65536 && ks * 2 <= nslice) ks *= 4;
  const unsigned blocks = (unsigned)(cdiv(total, 256 / ks) > 4096 ? 4096 : cdiv(total, 256 / ks));
  if (ks >= 16) hipLaunchKernelGGL((wgrad_reduce_kernel<9, 16>), dim3(blocks), dim3(256), 0, st, d->workspace, d->dw, nslice, d->Cout, d->Cin, d->Cin_real);
  else if (ks >= 4) hipLaunchKernelGGL((wgrad_reduce_kernel<9, 4>), dim3(blocks), dim3(256), 0, st, d->workspace, d->dw, nslice, d->Cout, d->Cin, d->Cin_real);
  else hipLaunchKernelGGL((wgrad_reduce_kernel<9, 1>), dim3(blocks), dim3(256), 0, st, d->workspace, d->dw, nslice, d->Cout, d->Cin, d->Cin_real);
  CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(dilated reduce)");
  return CVCS_OK;
}

// exact workspace size of one descriptor (covers the anisotropic / pitched case, which cvcs_wgrad_slices cannot express)
extern "C" int cvcs_wgrad_takes_bias(const cvcs_wgrad_desc* d) {
  if (!d) return 0;
  const bool special = d->aniso != 0 || d->x_row_pitch != 0 || d->x_img_pitch != 0;
  return d->dtype == CVCS_BF16 && d->dil <= 1 && !special && d->pad == 0 && d->H == d->Ho && d->W == d->Wo &&
         gemm_shape(d->KH, d->KW, d->stride, d->Cout, d->Cin);
}

// floats of the dW partial slabs (the bias partials of cvcs_wgrad_desc.dbias follow them)
static int64_t wgrad_slab_floats(const cvcs_wgrad_desc* d);

extern "C" int64_t cvcs_wgrad_workspace_floats(const cvcs_wgrad_desc* d) {
  if (!d || d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout < 16 || d->Cin <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0) return CVCS_EINVAL;
  int64_t n = wgrad_slab_floats(d);
  if (d->dbias && cvcs_wgrad_takes_bias(d)) n += (int64_t)gemm_plan(d->B, d->Ho, d->Wo, d->Cout, d->Cin).nslice * d->Cout;
  return n;
}

static int64_t wgrad_slab_floats(const cvcs_wgrad_desc* d) {
  const bool special = d->aniso != 0 || d->x_row_pitch != 0 || d->x_img_pitch != 0;
  int n;
  if (gemm_s2_shape(d)) {      // (the generic plan of the same descriptor may use more slices: take the larger)
    const int64_t g = (int64_t)gemm_plan(d->B, d->Ho, d->Wo, d->Cout, d->Cin).nslice * d->Cout * d->Cin;
    const int64_t o = (int64_t)cvcs_wgrad_slices(d->B, d->Ho, d->Wo, d->Cout, d->Cin, 1, 1, 2) * d->Cout * d->Cin;
    return g > o ? g : o;
  }
  if (thin_wgrad_shape(d)) {
    if (thin_wide_shape(d)) return (int64_t)thin_wide_groups(d) * (d->Cin / 32) * 9 * d->Cout * 32;      // every group's slabs at once
    return (int64_t)thin_wgrad_groups(d) * 4 * 9 * d->Cout * (d->Cin > 32 ? 32 : d->Cin);     // (one 32-channel group at a time)
  }
  if (d->dil > 1) {   // nine shifted 1x1 problems into one [slice][9][co][ci] slab
    const int a = make_plan(d->B, d->Ho, d->Wo, d->Cout, d->Cin, 1, 1, 1, 0).nslice;
    const int g = gemm_shape(1, 1, 1, d->Cout, d->Cin) ? gemm_plan(d->B, d->Ho, d->Wo, d->Cout, d->Cin).nslice : 0;
    return (int64_t)(a > g ? a : g) * 9 * d->Cout * d->Cin;
  }
  if (special) n = make_plan(d->B, d->Ho, d->Wo, d->Cout, d->Cin, d->KH, d->KW, d->stride, 0, d->aniso ? d->stride_w : d->stride).nslice;
  else n = cvcs_wgrad_slices(d->B, d->Ho, d->Wo, d->Cout, d->Cin, d->KH, d->KW, d->stride);
  return (int64_t)n * d->KH * d->KW * d->Cout * d->Cin;
}

extern "C" int cvcs_conv2d_wgrad(const cvcs_wgrad_desc* d, void* stream) {
  CVCS_CHECK_ARG(d != nullptr, "cvcs_conv2d_wgrad: null descriptor");
  CVCS_CHECK_ARG(d->dtype == CVCS_F32 || d->dtype == CVCS_BF16, "cvcs_conv2d_wgrad: bad dtype");
  const int es = d->dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(d->x && d->dy && d->dw && d->workspace, "cvcs_conv2d_wgrad: null tensor");
  CVCS_CHECK_ARG(d->Cout > 0 && d->Cout % 16 == 0, "cvcs_conv2d_wgrad: Cout=%d must be a multiple of 16", d->Cout);
  CVCS_CHECK_ARG(d->Cin > 0 && d->Cin % (16 / es) == 0, "cvcs_conv2d_wgrad: Cin=%d must be a multiple of %d", d->Cin, 16 / es);
  CVCS_CHECK_ARG(d->Cin_real > 0 && d->Cin_real <= d->Cin, "cvcs_conv2d_wgrad: Cin_real");
  const int taps = d->KH * d->KW;
  CVCS_CHECK_ARG((d->KH == 3 && d->KW == 3) || (d->KH == 2 && d->KW == 2) || (d->KH == 1 && d->KW == 1) || (d->KH == 7 && d->KW == 1),
                 "cvcs_conv2d_wgrad: filter %dx%d not built (3x3, 2x2, 1x1, 7x1)", d->KH, d->KW);
  if (d->dil > 1) {
    CVCS_CHECK_ARG(d->dbias == nullptr, "cvcs_conv2d_wgrad: no dbias with dilation");
    return wgrad_dilated(d, (hipStream_t)stream);
  }
  const bool aniso = d->aniso != 0, pitched = d->x_row_pitch != 0 || d->x_img_pitch != 0;
  const int stride_w = aniso ? d->stride_w : d->stride, pad_w = aniso ? d->pad_w : d->pad;
  CVCS_CHECK_ARG(d->stride >= 1 && d->stride <= 2 && d->pad >= 0 && stride_w >= 1 && stride_w <= 2 && pad_w >= 0, "cvcs_conv2d_wgrad: stride/pad");
  const int eh = (d->H + 2 * d->pad - (d->KH - 1) - 1) / d->stride + 1;
  const int ew = (d->W + 2 * pad_w - (d->KW - 1) - 1) / stride_w + 1;
  CVCS_CHECK_ARG(eh == d->Ho && ew == d->Wo, "cvcs_conv2d_wgrad: Ho,Wo=%d,%d but geometry gives %d,%d", d->Ho, d->Wo, eh, ew);
  CVCS_CHECK_ARG((pitched || d->x_ld >= d->Cin) && d->x_ld * es % 16 == 0 && d->dy_ld >= d->Cout && d->dy_ld * es % 16 == 0,
                 "cvcs_conv2d_wgrad: leading dimensions");
  CVCS_CHECK_ARG(((uintptr_t)d->x % 16) == 0 && ((uintptr_t)d->dy % 16) == 0, "cvcs_conv2d_wgrad: alignment");
  const int64_t row_pitch = d->x_row_pitch ? d->x_row_pitch : (int64_t)d->W * d->x_ld;
  const int64_t img_pitch = d->x_img_pitch ? d->x_img_pitch : (int64_t)d->H * row_pitch;
  CVCS_CHECK_ARG(row_pitch * es % 16 == 0 && img_pitch * es % 16 == 0 && row_pitch > 0 && img_pitch >= row_pitch, "cvcs_conv2d_wgrad: pitches of x");
  if (d->x_up2) CVCS_CHECK_ARG(thin_wgrad_shape(d), "cvcs_conv2d_wgrad: x_up2 (the weight gradient over the bilinear x2 up-sampling of `x`) is built for the thin bf16 3x3 32 -> 16 instance, even H, W");
  if (thin_wgrad_shape(d)) {
    ThinWgArgs ta;
    ta.x = (const char*)d->x; ta.dy = (const char*)d->dy; ta.ws = d->workspace; ta.x_ld = d->x_ld; ta.dy_ld = d->dy_ld;
    ta.B = d->B; ta.H = d->H; ta.W = d->W; ta.Cin = d->Cin; ta.Cout = d->Cout;
    ta.tiles_x = (int)cdiv(d->W, kThinTW); ta.tiles_y = (int)cdiv(d->H, kThinTH); ta.ntiles = d->B * ta.tiles_x * ta.tiles_y;
    hipStream_t tst = (hipStream_t)stream;
    if (thin_wide_shape(d)) {
      ta.tiles_x = (int)cdiv(d->W, kWideTW); ta.tiles_y = (int)cdiv(d->H, kWideTH); ta.ntiles = d->B * ta.tiles_x * ta.tiles_y;
      const int wg = thin_wide_groups(d);
      const size_t lds = (size_t)kWideTH * kWideTW * d->Cout * 2 + (size_t)kWideHR * kWideHC * 256;
      const dim3 grid((unsigned)wg, (unsigned)(d->Cin / 128));
      if (d->Cout == 16) hipLaunchKernelGGL((wgrad_thin_wide_kernel<16>), grid, dim3(256), lds, tst, ta);
      else hipLaunchKernelGGL((wgrad_thin_wide_kernel<32>), grid, dim3(256), lds, tst, ta);
      CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(thin, wide input)");
      hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3((unsigned)cdiv(9 * d->Cout * 32, 64), (unsigned)(d->Cin / 32)), dim3(1024), 0, tst, d->workspace, d->dw, wg,
                         d->Cout, 32, d->Cin, 0);
      CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(thin reduce)");
      return CVCS_OK;
    }
    const int groups = thin_wgrad_groups(d);
    if (d->x_up2) {
      const int lds = kThinTH * kThinTW * 16 * 2 + (kThinHR * kThinHC * 32 * 2 + 1023) / 1024 * 1024 + (kUpPatchRows * kUpPatchCols * 32 * 2 + 1023) / 1024 * 1024;
      hipLaunchKernelGGL((wgrad_thin_kernel<16, 32, true>), dim3((unsigned)groups), dim3(256), lds, tst, ta);
      CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(thin, up-sampled input)");
      hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3((unsigned)cdiv(9 * 16 * 32, 64)), dim3(1024), 0, tst, d->workspace, d->dw, groups * 4, 16, 32, 32, 0);
      CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(thin reduce)");
      return CVCS_OK;
    }
#define LAUNCH_THIN(CO_, CI_)                                                                                              \
  do {                                                                                                                     \
    const int lds = kThinTH * kThinTW * (CO_) * 2 + (kThinHR * kThinHC * (CI_) * 2 + 1023) / 1024 * 1024;                  \
    hipLaunchKernelGGL((wgrad_thin_kernel<CO_, CI_>), dim3((unsigned)groups), dim3(256), lds, tst, ta);                    \
  } while (0)
    // a wide input (Cin = 64 ... 256 under a thin Cout: the 128 -> 32 decoder layer) runs as 32-channel groups of the same kernel: the
    // generic kernel took 552 us there, four group launches take a third less and re-read only the thin dy
    const int cgrp = d->Cin > 32 ? 32 : d->Cin;
    for (int ci0 = 0; ci0 < d->Cin; ci0 += cgrp) {
      ta.x = (const char*)d->x + (int64_t)ci0 * 2;
      ta.Cin = cgrp;
      if (d->Cout == 16 && cgrp == 16) LAUNCH_THIN(16, 16);
      else if (d->Cout == 16) LAUNCH_THIN(16, 32);
      else if (cgrp == 16) LAUNCH_THIN(32, 16);
      else LAUNCH_THIN(32, 32);
      CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(thin)");
      hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3((unsigned)cdiv(9 * d->Cout * cgrp, 64)), dim3(1024), 0, tst, d->workspace, d->dw, groups * 4,
                         d->Cout, cgrp, d->Cin, ci0);
      CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(thin reduce)");
    }
#undef LAUNCH_THIN
    return CVCS_OK;
  }
  const int fastp = (aniso || pitched) ? 0 : fast_path(d);
  WgradPlan pl = make_plan(d->B, d->Ho, d->Wo, d->Cout, d->Cin, d->KH, d->KW, d->stride, fastp, stride_w);
  CVCS_CHECK_ARG(pl.HR * pl.HC <= (fastp ? 128 : kXRows), "cvcs_conv2d_wgrad: halo tile too large");
  WgradArgs a;
  a.tiles_mn = 1; a.xcd_order = wgrad_xcd_order(); a.bias_ws = nullptr;
  a.x = (const char*)d->x; a.dy = (const char*)d->dy; a.ws = d->workspace;
  a.x_ld = d->x_ld; a.dy_ld = d->dy_ld;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.stride_w = stride_w; a.pad_w = pad_w; a.x_row_pitch = row_pitch; a.x_img_pitch = img_pitch;
  a.sh_y = a.sh_x = 0; a.slice_stride = 0;
  a.TH = pl.TH; a.TW = pl.TW; a.HR = pl.HR; a.HC = pl.HC;
  a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.ktiles = pl.ktiles; a.per_slice = pl.per_slice;
  a.ntile_n = (int)cdiv(d->Cin, 64);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  int nslice_used = pl.nslice;
  static const int s2_on = getenv("CVCS_WGRAD_GEMM_S2") ? atoi(getenv("CVCS_WGRAD_GEMM_S2")) : 1;   // tuning knob
  const bool s2 = s2_on && gemm_s2_shape(d);
  a.x_s2 = s2 ? 1 : 0;
  if (s2 || (d->dtype == CVCS_BF16 && !aniso && !pitched && d->pad == 0 && gemm_shape(d->KH, d->KW, d->stride, d->Cout, d->Cin) &&
             d->H == d->Ho && d->W == d->Wo)) {
    const GemmPlan g = gemm_plan(d->B, d->Ho, d->Wo, d->Cout, d->Cin);
    a.ktiles = g.ktiles; a.per_slice = g.per_slice; a.ntile_n = g.ntile_n;
    nslice_used = g.nslice;
    a.tiles_mn = g.tiles_mn;
    if (d->dbias) a.bias_ws = d->workspace + wgrad_slab_floats(d);
    const dim3 grid((unsigned)(g.tiles_mn * g.nslice));
    // ring depth of the 4-wave tiles (K-tiles in flight = depth - 1): these launches are bound by the latency of the L2 -> LDS path - two
    // 16 KiB tiles in flight per workgroup, ~1 us per 32-pixel K-tile - and LDS has room for a deeper ring at two workgroups per CU
    static const int gemm_ns = getenv("CVCS_WGRAD_GEMM_NS") ? atoi(getenv("CVCS_WGRAD_GEMM_NS")) : 3;   // tuning knob (3 | 4 | 5)
#define LAUNCH_GEMM_NS(BM_, BN_, WMV_, NS_)                                                                                     \
  do {                                                                                                                          \
    const int lds = (NS_) * ((BM_) / 64 + (BN_) / 64) * 4096;                                                                   \
    static bool attr_done = false;                                                                                              \
    if (!attr_done) {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_gemm_kernel<BM_, BN_, WMV_, NS_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
      attr_done = true;                                                                                                         \
    }                                                                                                                           \
    hipLaunchKernelGGL((wgrad_gemm_kernel<BM_, BN_, WMV_, NS_>), grid, dim3((WMV_) * 128), lds, st, a);                         \
  } while (0)
#define LAUNCH_GEMM(BM_, BN_, WMV_) LAUNCH_GEMM_NS(BM_, BN_, WMV_, 3)
#define LAUNCH_GEMM4(BM_, BN_)                                                  \
  do {                                                                          \
    if (gemm_ns >= 5 && (BM_) + (BN_) <= 192) LAUNCH_GEMM_NS(BM_, BN_, 2, 5);   \
    else if (gemm_ns >= 4) LAUNCH_GEMM_NS(BM_, BN_, 2, 4);                      \
    else LAUNCH_GEMM_NS(BM_, BN_, 2, 3);                                        \
  } while (0)
    if (g.BM == 256 && g.BN == 256) LAUNCH_GEMM(256, 256, 4);
    else if (g.BM == 256) LAUNCH_GEMM(256, 128, 4);
    else if (g.BM == 128 && g.BN == 128) LAUNCH_GEMM4(128, 128);
    else if (g.BM == 128) LAUNCH_GEMM4(128, 64);
    else if (g.BN == 64) LAUNCH_GEMM4(64, 64);
    else LAUNCH_GEMM4(64, 128);
#undef LAUNCH_GEMM4
#undef LAUNCH_GEMM
#undef LAUNCH_GEMM_NS
    if (d->dbias)
      hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((unsigned)cdiv(d->Cout, 32)), dim3(256), 0, st, a.bias_ws, g.nslice, d->Cout, d->dbias);
    CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(gemm)");
    rc = CVCS_OK;
  } else if (d->dbias) {
    set_error("cvcs_conv2d_wgrad: dbias is built for the descriptors cvcs_wgrad_takes_bias() accepts (bf16 1x1 / stride 1 GEMM path)");
    return CVCS_EUNSUPPORTED;
  } else if (d->dtype == CVCS_F32)
    rc = taps == 9 ? launch<float, 9>(a, pl, st) : taps == 7 ? launch<float, 7>(a, pl, st) : taps == 4 ? launch<float, 4>(a, pl, st) : launch<float, 1>(a, pl, st);
  else if (fastp && s2_shape(d)) {
    const int lds = 3 * (64 + 128) * 128;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fast128_kernel<4, 8, 3, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_done = true;
    }
    hipLaunchKernelGGL((wgrad_fast128_kernel<4, 8, 3, true>), dim3((unsigned)pl.tiles_mn, (unsigned)pl.nslice), dim3(512), lds, st, a);
    CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(s2)");
    rc = CVCS_OK;
  } else if (fastp == 2) {
    constexpr int kStages = 3;   // deeper rings (4-6 stages) measured no faster: the fetch is not latency-bound
    const int lds = kStages * (64 + 6 * 16) * 128;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fast128_kernel<4, 8, kStages>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_done = true;
    }
    hipLaunchKernelGGL((wgrad_fast128_kernel<4, 8, kStages>), dim3((unsigned)pl.tiles_mn, (unsigned)pl.nslice), dim3(512), lds, st, a);
    CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(fast128)");
    rc = CVCS_OK;
  } else if (fastp) {
    // 4 x 8-pixel K-tiles (plan already made for that shape by make_plan)
    static bool attr_done = false;
    const int lds = 3 * (32 + 6 * 16) * 128;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fast_kernel<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_done = true;
    }
    hipLaunchKernelGGL((wgrad_fast_kernel<4, 8>), dim3((unsigned)pl.tiles_mn, (unsigned)pl.nslice), dim3(256), lds, st, a);
    CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(fast)");
    rc = CVCS_OK;
  } else
    rc = taps == 9 ? launch<bf16_t, 9>(a, pl, st) : taps == 7 ? launch<bf16_t, 7>(a, pl, st) : taps == 4 ? launch<bf16_t, 4>(a, pl, st) : launch<bf16_t, 1>(a, pl, st);
  if (rc != CVCS_OK) return rc;
  const int64_t total = (int64_t)d->Cout * d->Cin_real;
  // lanes per (co, ci) pair: enough threads for ~64K in flight, never more than the slices there are
  int ks = 1;
  while (ks < 16 && total * ks < 65536 && ks * 2 <= nslice_used) ks *= 4;
#define LAUNCH_RED(TAPS, KS)                                                                                        \
  hipLaunchKernelGGL((wgrad_reduce_kernel<TAPS, KS>), dim3((unsigned)(cdiv(total, 256 / KS) > 4096 ? 4096 : cdiv(total, 256 / KS))), \
                     dim3(256), 0, st, d->workspace, d->dw, nslice_used, d->Cout, d->Cin, d->Cin_real)
#define LAUNCH_RED_T(TAPS)                                   \
  do {                                                       \
    if (ks >= 16) LAUNCH_RED(TAPS, 16);                      \
    else if (ks >= 4) LAUNCH_RED(TAPS, 4);                   \
    else LAUNCH_RED(TAPS, 1);                                \
  } while (0)
  if (taps == 9) LAUNCH_RED_T(9);
  else if (taps == 7) LAUNCH_RED_T(7);
  else if (taps == 4) LAUNCH_RED_T(4);
  else LAUNCH_RED_T(1);
#undef LAUNCH_RED_T
#undef LAUNCH_RED
  CVCS_CHECK_LAUNCH("cvcs_conv2d_wgrad(reduce)");
  return CVCS_OK;
}


// ===================================================================================================================
// Second-moment ("Gram") matrix of an activation view: G[i][j] = sum_p a[p][i] a[p][j], s[i] = sum_p a[p][i] over the M pixels - the
// sufficient statistics of EVERY BatchNorm behind a 1x1 convolution of a (cvcs_bn_gram_finalize): mean_c = w_c . s / M,
// E[y_c^2] = w_c^T G w_c / M.  It is the 1x1 weight-gradient GEMM with dy = x = a (same kernel, same LDS-DMA ring), K-split into slabs
// that are summed in f64 in slice order (bitwise reproducible) - one pass over a tensor that is 4x narrower than the conv output whose
// statistics it yields.
namespace cvcs {
template <int KS>
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias_ws, int nslice, int C,
                                                         double* __restrict__ gram, double* __restrict__ colsum) {
  constexpr int EL = 256 / KS;
  __shared__ double sh[KS][EL];
  const int64_t total = (int64_t)C * C + C;       // the matrix, then the column sums
  const int el = threadIdx.x % EL, kg = threadIdx.x / EL;
  const int64_t id = (int64_t)blockIdx.x * EL + el;
  double a = 0.0;
  if (id < total) {
    const bool mat = id < (int64_t)C * C;
    const float* src = mat ? ws + id : bias_ws + (id - (int64_t)C * C);
    const int64_t stride = mat ? (int64_t)C * C : C;
    int k = kg;
    for (; k + 3 * KS < nslice; k += 4 * KS) {
      const float v0 = src[(int64_t)k * stride], v1 = src[(int64_t)(k + KS) * stride];
      const float v2 = src[(int64_t)(k + 2 * KS) * stride], v3 = src[(int64_t)(k + 3 * KS) * stride];
      a += (double)v0; a += (double)v1; a += (double)v2; a += (double)v3;
    }
    for (; k < nslice; k += KS) a += (double)src[(int64_t)k * stride];
  }
  sh[kg][el] = a;
  __syncthreads();
  if (kg == 0 && id < total) {
    double t = sh[0][el];
#pragma unroll
    for (int q = 1; q < KS; ++q) t += sh[q][el];
    if (id < (int64_t)C * C) gram[id] = t; else colsum[id - (int64_t)C * C] = t;
  }
}
struct GramPlan { int BM, tiles, ktiles, per_slice, nslice; };
static GramPlan gram_plan(int64_t M, int C, int bm = 0) {
  GramPlan g;
  g.BM = bm ? bm : (C % 128 == 0 ? 128 : 64);
  g.tiles = (C / g.BM) * (C / g.BM);
  g.ktiles = (int)cdiv(M, 32);
  int want = (int)cdiv(512, g.tiles);
  // partial slabs are written and read once each: keep them below ~8 MB (at 128 channels 512 slices would be 32 MB for a 34 MB input)
  const int64_t cap = (8ll << 20) / ((int64_t)C * C * 4);
  if (want > cap) want = (int)(cap < 1 ? 1 : cap);
  if (want > g.ktiles) want = g.ktiles;
  if (want < 1) want = 1;
  g.per_slice = (int)cdiv(g.ktiles, want);
  g.nslice = (int)cdiv(g.ktiles, g.per_slice);
  // the slab cap leaves 128 .. 512-channel matrices with 128 workgroups of 128 x 128 tiles - half the chip idle, 30-35 us for 4 GFLOP:
  // 64 x 64 tiles give four times the workgroups under the same cap
  static const int tile64 = getenv("CVCS_GRAM_TILE64") ? atoi(getenv("CVCS_GRAM_TILE64")) : 1;   // tuning knob
  if (!bm && tile64 && g.BM == 128 && g.tiles * g.nslice < 256) return gram_plan(M, C, 64);
  return g;
}
}  // namespace cvcs

extern "C" int64_t cvcs_gram_workspace_floats(int64_t M, int C) {
  if (M <= 0 || C < 64 || C % 64 != 0 || M >= (1ll << 31)) return CVCS_EINVAL;
  const GramPlan g = gram_plan(M, C);
  return (int64_t)g.nslice * ((int64_t)C * C + C);
}

extern "C" int cvcs_gram(const void* x, int64_t x_ld, int64_t M, int C, double* gram, double* colsum, float* workspace, void* stream) {
  CVCS_CHECK_ARG(x && gram && colsum && workspace, "cvcs_gram: null argument");
  CVCS_CHECK_ARG(M > 0 && M < (1ll << 31) && C >= 64 && C % 64 == 0 && C <= 2048, "cvcs_gram: M=%lld C=%d (channels: a multiple of 64)", (long long)M, C);
  CVCS_CHECK_ARG(x_ld >= C && (x_ld * 2) % 16 == 0 && ((uintptr_t)x % 16) == 0, "cvcs_gram: view");
  const GramPlan g = gram_plan(M, C);
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const char*)x; a.dy = (const char*)x; a.x_ld = x_ld; a.dy_ld = x_ld;
  a.B = 1; a.H = 1; a.W = (int)M; a.Ho = 1; a.Wo = (int)M; a.Cin = C; a.Cout = C;
  a.KW = 1; a.stride = 1; a.stride_w = 1;
  a.ws = workspace; a.bias_ws = workspace + (int64_t)g.nslice * C * C;
  a.ktiles = g.ktiles; a.per_slice = g.per_slice; a.ntile_n = C / g.BM; a.tiles_mn = g.tiles; a.xcd_order = wgrad_xcd_order();
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(g.tiles * g.nslice));
  if (g.BM == 128) {
    const int lds = 3 * 4 * 4096;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_gemm_kernel<128, 128, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((wgrad_gemm_kernel<128, 128, 2>), grid, dim3(256), lds, st, a);
  } else {
    const int lds = 3 * 2 * 4096;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_gemm_kernel<64, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((wgrad_gemm_kernel<64, 64, 2>), grid, dim3(256), lds, st, a);
  }
  CVCS_CHECK_LAUNCH("cvcs_gram");
  const int64_t total = (int64_t)C * C + C;
  // lanes per element: enough threads in flight for the small matrices, never more than the slices there are
  int ks = 1;
  while (ks < 16 && total * ks < 131072 && ks * 4 <= g.nslice) ks *= 4;
  if (ks >= 16) hipLaunchKernelGGL((gram_reduce_kernel<16>), dim3((unsigned)cdiv(total, 16)), dim3(256), 0, st, workspace, a.bias_ws, g.nslice, C, gram, colsum);
  else if (ks >= 4) hipLaunchKernelGGL((gram_reduce_kernel<4>), dim3((unsigned)cdiv(total, 64)), dim3(256), 0, st, workspace, a.bias_ws, g.nslice, C, gram, colsum);
  else hipLaunchKernelGGL((gram_reduce_kernel<1>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, workspace, a.bias_ws, g.nslice, C, gram, colsum);
  CVCS_CHECK_LAUNCH("cvcs_gram(reduce)");
  return CVCS_OK;
}
