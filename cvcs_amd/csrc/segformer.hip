// Kernels of the reference's `SegformerMod` (/root/reference/source/scripts/nets.py:313-349: transformers' Mix-Transformer segmentation
// model + the reference's ConvTranspose tail) that the ResNet / Swin kernels do not cover.  gfx950 only.
//
//   * patch matrices: every convolution of this model whose filter is neither 1x1 nor 3x3 / stride 1 (the 7x7 / stride 4 and 3x3 / stride 2
//     overlapping patch embeddings, the k = s = 8 | 4 | 2 sequence-reduction convolutions, the two ConvTranspose2d of the tail taken phase
//     by phase) is ONE gather of its patches into rows (`cvcs_im2col`, zero-padded to a K-group) followed by the GEMM kernels of the 1x1
//     convolutions; `cvcs_col2im` is the transposed gather (data gradient).  These layers work on 3 ... 320 channels at 1/4 ... 1/32 of
//     the tile (and on NC <= 32 channels in the tail): HBM-bound byte shuffling, no case for a dedicated MFMA pipeline each.
//   * depthwise 3x3 of the Mix-FFN (forward, data gradient = the same kernel on the flipped filter, weight + bias gradient partial rows);
//   * spatial-reduction attention: N queries x Nk = N / sr^2 keys per image and head, head dimension 32 (forward with the log-sum-exp
//     saved; backward recomputes the probabilities: a query-major kernel for dQ, a key-major kernel for dK / dV partials, fixed-order sum);
//   * DropPath as a per-sample scale (counter-based, replayable on the host) fused into the residual add;
//   * table-driven weight gather: f32 master tensors in the reference's layouts -> the GEMM operands of the patch-matrix layers.
#include <stdlib.h>

#include "common.h"

namespace cvcs {

#define SF_DT(dt) ((dt) == CVCS_F32 || (dt) == CVCS_BF16)

static int sf_view(const char* fn, const void* p, int64_t ld, int C, int es) {
  CVCS_CHECK_ARG(p != nullptr, "%s: null tensor", fn);
  CVCS_CHECK_ARG(ld >= C && (ld * es) % 16 == 0 && ((uintptr_t)p % 16) == 0, "%s: view (ld=%lld, C=%d) must be 16-byte aligned", fn, (long long)ld, C);
  return CVCS_OK;
}
static inline unsigned sf_grid(int64_t items, int cap = 256 * 32) {
  int64_t g = cdiv(items, 256);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------ patch gather / scatter
// col[b, oy, ox, (ky*KW + kx)*C + c] = in[b, oy*s + oy0 + ky*dir, ox*s + ox0 + kx*dir, c]   (0 outside the map, 0 in the K padding)
// a convolution: oy0 = ox0 = -pad, dir = +1; phase (a, b) of a stride-2 ConvTranspose2d(k, p): s = 1, dir = -1, oy0 = (a + p) >> 1
struct ColArgs {
  const char* in; char* col;
  int64_t in_ld, col_ld;
  int B, H, W, C, Ho, Wo, KH, KW, s, oy0, ox0, dir;
};

template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(ColArgs p) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int cpt = p.C / V;                                   // chunks per tap
  const int cpr = (int)(p.col_ld / V);                       // chunks per row (K padding included)
  const int64_t total = (int64_t)p.B * p.Ho * p.Wo * cpr;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int ch = (int)(id % cpr);
    const int64_t pix = id / cpr;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    const int tap = ch / cpt;
    if (tap < p.KH * p.KW) {
      const int cc = ch - tap * cpt;
      const int ox = (int)(pix % p.Wo);
      const int64_t t = pix / p.Wo;
      const int oy = (int)(t % p.Ho);
      const int64_t b = t / p.Ho;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      const int iy = oy * p.s + p.oy0 + ky * p.dir, ix = ox * p.s + p.ox0 + kx * p.dir;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v = *reinterpret_cast<const uint4*>(p.in + (((b * p.H + iy) * p.W + ix) * p.in_ld) * ES + cc * 16);
    }
    *reinterpret_cast<uint4*>(p.col + (pix * p.col_ld) * ES + ch * 16) = v;
  }
}

// the first patch embedding reads the tile itself: planar u8 / f32 [B,3,H,W], normalised as nets.py:337-340 does ((x - mean) / std on the
// raw 0..255 scale, f32 division), K index (ky*KW + kx)*3 + c
struct StemColArgs {
  const void* src; char* col;
  int64_t col_ld;
  int B, H, W, Ho, Wo, KH, KW, s, pad, is_u8;
  float mean[3], std[3];
};
template <typename T>
__global__ __launch_bounds__(256) void im2col_stem_kernel(StemColArgs p) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int cpr = (int)(p.col_ld / V);
  const int K = p.KH * p.KW * 3;
  const int64_t total = (int64_t)p.B * p.Ho * p.Wo * cpr;
  const int64_t plane = (int64_t)p.H * p.W;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int ch = (int)(id % cpr);
    const int64_t pix = id / cpr;
    const int ox = (int)(pix % p.Wo);
    const int64_t t = pix / p.Wo;
    const int oy = (int)(t % p.Ho);
    const int64_t b = t / p.Ho;
    float f[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int kk = ch * V + k;
      float v = 0.f;
      if (kk < K) {
        const int c = kk % 3, tap = kk / 3;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int iy = oy * p.s - p.pad + ky, ix = ox * p.s - p.pad + kx;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
          const int64_t o = (b * 3 + c) * plane + (int64_t)iy * p.W + ix;
          const float x = p.is_u8 ? (float)reinterpret_cast<const uint8_t*>(p.src)[o] : reinterpret_cast<const float*>(p.src)[o];
          v = (x - p.mean[c]) / p.std[c];
        }
      }
      f[k] = v;
    }
    *reinterpret_cast<uint4*>(p.col + (pix * p.col_ld) * ES + ch * 16) = Elem<T>::pack(f);
  }
}

// dx[b, iy, ix, c] = sum over phases and taps of col_ph[b, oy, ox, tap*C + c] with oy*s + oy0[ph] + ky*dir == iy (f32 sum, fixed order)
struct Col2imArgs {
  const char* col; char* dx;
  int64_t col_ld, dx_ld, phase_bytes;
  int B, H, W, C, Ho, Wo, KH, KW, s, dir, nph;
  int oy0[4], ox0[4];
};
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(Col2imArgs p) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int cpt = p.C / V;
  const int64_t total = (int64_t)p.B * p.H * p.W * cpt;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % cpt);
    const int64_t pix = id / cpt;
    const int ix = (int)(pix % p.W);
    const int64_t t = pix / p.W;
    const int iy = (int)(t % p.H);
    const int64_t b = t / p.H;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int ph = 0; ph < p.nph; ++ph) {
      const char* col = p.col + ph * p.phase_bytes;
      for (int ky = 0; ky < p.KH; ++ky) {
        const int ny = iy - p.oy0[ph] - ky * p.dir;
        if (ny < 0 || ny % p.s) continue;
        const int oy = ny / p.s;
        if (oy >= p.Ho) continue;
        for (int kx = 0; kx < p.KW; ++kx) {
          const int nx = ix - p.ox0[ph] - kx * p.dir;
          if (nx < 0 || nx % p.s) continue;
          const int ox = nx / p.s;
          if (ox >= p.Wo) continue;
          float f[V];
          Elem<T>::unpack(*reinterpret_cast<const uint4*>(col + (((b * p.Ho + oy) * p.Wo + ox) * p.col_ld + (int64_t)(ky * p.KW + kx) * p.C) * ES + cc * 16), f);
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += f[k];
        }
      }
    }
    *reinterpret_cast<uint4*>(p.dx + (pix * p.dx_ld) * ES + cc * 16) = Elem<T>::pack(acc);
  }
}

// the four phase outputs [4][B,H,W,C] of a stride-2 ConvTranspose2d -> [B,2H,2W,C] (+ReLU); the transposed move (ReLU mask from the output)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void phase_shuffle_kernel(const char* ph, int64_t ph_ld, int64_t phase_bytes, int B, int H, int W, int C, char* full,
                                                            int64_t full_ld, const char* mask, int64_t mask_ld, int relu) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int cpt = C / V;
  const int64_t total = (int64_t)B * 4 * H * W * cpt;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % cpt);
    const int64_t pix = id / cpt;                         // pixel of the full-resolution map
    const int ox = (int)(pix % (2 * W));
    const int64_t t = pix / (2 * W);
    const int oy = (int)(t % (2 * H));
    const int64_t b = t / (2 * H);
    const int phase = (oy & 1) * 2 + (ox & 1);
    const int64_t lo = (b * H + (oy >> 1)) * W + (ox >> 1);
    char* pp = const_cast<char*>(ph) + phase * phase_bytes + (lo * ph_ld) * ES + cc * 16;
    char* fp = full + (pix * full_ld) * ES + cc * 16;
    float f[V];
    if constexpr (!BWD) {
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(pp), f);
      if (relu) {
#pragma unroll
        for (int k = 0; k < V; ++k) f[k] = fmaxf(f[k], 0.f);
      }
      *reinterpret_cast<uint4*>(fp) = Elem<T>::pack(f);
    } else {
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(fp), f);
      if (mask) {
        float m[V];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(mask + (pix * mask_ld) * ES + cc * 16), m);
#pragma unroll
        for (int k = 0; k < V; ++k) f[k] = m[k] > 0.f ? f[k] : 0.f;
      }
      *reinterpret_cast<uint4*>(pp) = Elem<T>::pack(f);
    }
  }
}

// ------------------------------------------------------------------------------------------------ depthwise 3x3 (Mix-FFN)
// nn.Conv2d(C, C, 3, 1, 1, groups=C): w [C][9] f32 as stored ([C,1,3,3]); flip = 1 runs the data gradient (taps reversed, no bias).
// A workgroup = (<= 32 channel chunks) x (256 / chunks consecutive columns) walks a band of kDwRows rows of one image top to bottom with a
// sliding 3x3 window in registers: three new loads per output pixel, the neighbouring columns' loads are L1 hits, the band's halo rows the
// only re-read.  Work items (image, band, column strip) are handed out in XCD-aware order so that adjacent strips share an L2.
constexpr int kDwChunks = 32, kDwRows = 16, kDwRing = 6;
struct DwArgs {
  const char* x; const char* dy; char* out;
  const float* w; const float* bias; float* part;
  int64_t x_ld, dy_ld, out_ld;
  int B, H, W, C, flip;
};

template <int NB>
__device__ __forceinline__ void dw_load_row(const char* base, int64_t ld_bytes, int yy, int x, int H, int W, bool col_ok, uint4 (&r)[3]) {
  // the three horizontally adjacent chunks of row yy around column x (zeros outside the map); NB = bytes per chunk (16 | 8)
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int xx = x + d - 1;
    r[d] = make_uint4(0u, 0u, 0u, 0u);
    if (col_ok && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
      const char* src = base + ((int64_t)yy * W + xx) * ld_bytes;
      if constexpr (NB == 16) r[d] = *reinterpret_cast<const uint4*>(src);
      else { const uint2 t = *reinterpret_cast<const uint2*>(src); r[d].x = t.x; r[d].y = t.y; }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(DwArgs p) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = p.C / V;
  const int ccw = CC < kDwChunks ? CC : kDwChunks;
  const int PL = 256 / ccw;
  const int cl = threadIdx.x % ccw, pl = threadIdx.x / ccw;
  const int cc = blockIdx.y * ccw + cl;
  if (pl >= PL || cc >= CC) return;
  float wv[9][V], bv[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = cc * V + k;
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t][k] = p.w[c * 9 + (p.flip ? 8 - t : t)];
    bv[k] = p.bias ? p.bias[c] : 0.f;
  }
  const int xs = (p.W + PL - 1) / PL, ys = (p.H + kDwRows - 1) / kDwRows;
  const int item = (int)xcd_order(blockIdx.x, gridDim.x);
  const int xb = item % xs, yb = (item / xs) % ys, b = item / (xs * ys);
  const int x = xb * PL + pl;
  const bool col_ok = x < p.W;
  const int y0 = yb * kDwRows, y1 = y0 + kDwRows < p.H ? y0 + kDwRows : p.H;
  const char* base = p.x + ((int64_t)b * p.H * p.W * p.x_ld) * ES + cc * 16;
  char* obase = p.out + ((int64_t)b * p.H * p.W * p.out_ld) * ES + cc * 16;
  // a ring of kDwRing window rows, the loop fully unrolled: the loads of rows y + 2 ... y + kDwRing - 2 are in flight while row y is computed
  // (at ~200 VGPRs two waves share a SIMD: with one row ahead the kernel held 24 KB per CU in flight - 2.1 TB/s; three rows ahead triple it)
  uint4 R[kDwRing][3];
#pragma unroll
  for (int r = 0; r < kDwRing - 1; ++r)
    dw_load_row<16>(base, p.x_ld * ES, y0 - 1 + r, x, p.H, p.W, col_ok && y0 - 1 + r <= y1, R[r]);
#pragma unroll
  for (int i = 0; i < kDwRows; ++i) {
    const int y = y0 + i;
    if (y >= y1) break;
    dw_load_row<16>(base, p.x_ld * ES, y + kDwRing - 2, x, p.H, p.W, col_ok && y + kDwRing - 2 <= y1, R[(i + kDwRing - 1) % kDwRing]);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = bv[k];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float f0[V], f1[V], f2[V];
      Elem<T>::unpack(R[i % kDwRing][d], f0);
      Elem<T>::unpack(R[(i + 1) % kDwRing][d], f1);
      Elem<T>::unpack(R[(i + 2) % kDwRing][d], f2);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += f0[k] * wv[d][k] + f1[k] * wv[3 + d][k] + f2[k] * wv[6 + d][k];
    }
    if (col_ok) *reinterpret_cast<uint4*>(obase + ((int64_t)y * p.W + x) * p.out_ld * ES) = Elem<T>::pack(acc);
  }
}

// partial sums of dW[c][tap] = sum dy[p] * x[p + off(tap)] and db[c] = sum dy[p]: part[blockIdx.x][c*9 + tap | 9C + c].  Four channels per
// thread (40 accumulators), the same sliding window; workgroup w takes the contiguous range of work items [w * n / rows, (w + 1) * n / rows)
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(DwArgs p) {
  constexpr int ES = sizeof(T), V = 4, NB = V * ES;
  __shared__ float red[256][V + 1];
  const int CC = p.C / V;
  const int ccw = CC < 64 ? CC : 64;
  const int PL = 256 / ccw;
  const int cl = threadIdx.x % ccw, pl = threadIdx.x / ccw;
  const int cc = blockIdx.y * ccw + cl;
  const bool live = pl < PL && cc < CC;
  float acc[10][V];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int k = 0; k < V; ++k) acc[t][k] = 0.f;
  const int xs = (p.W + PL - 1) / PL, ys = (p.H + kDwRows - 1) / kDwRows;
  const int nitems = p.B * xs * ys;
  const int i0 = (int)((int64_t)blockIdx.x * nitems / gridDim.x), i1 = (int)((int64_t)(blockIdx.x + 1) * nitems / gridDim.x);
  if (live) {
    for (int item = i0; item < i1; ++item) {
      const int xb = item % xs, yb = (item / xs) % ys, b = item / (xs * ys);
      const int x = xb * PL + pl;
      const bool col_ok = x < p.W;
      const int y0 = yb * kDwRows, y1 = y0 + kDwRows < p.H ? y0 + kDwRows : p.H;
      const char* base = p.x + ((int64_t)b * p.H * p.W * p.x_ld) * ES + cc * NB;
      const char* gbase = p.dy + ((int64_t)b * p.H * p.W * p.dy_ld) * ES + cc * NB;
      uint4 R[kDwRing][3], G[kDwRing - 2];
      auto load_g = [&](int yy, uint4& gr) {
        gr = make_uint4(0u, 0u, 0u, 0u);
        if (col_ok && yy < y1) {
          const char* src = gbase + ((int64_t)yy * p.W + x) * p.dy_ld * ES;
          if constexpr (NB == 16) gr = *reinterpret_cast<const uint4*>(src);
          else { const uint2 t = *reinterpret_cast<const uint2*>(src); gr.x = t.x; gr.y = t.y; }
        }
      };
      auto unpack4 = [](const uint4& raw, float* f) {
        if constexpr (ES == 4) { Elem<float>::unpack(raw, f); }
        else { f[0] = __uint_as_float(raw.x << 16); f[1] = __uint_as_float(raw.x & 0xffff0000u); f[2] = __uint_as_float(raw.y << 16); f[3] = __uint_as_float(raw.y & 0xffff0000u); }
      };
#pragma unroll
      for (int r = 0; r < kDwRing - 1; ++r)
        dw_load_row<NB>(base, p.x_ld * ES, y0 - 1 + r, x, p.H, p.W, col_ok && y0 - 1 + r <= y1, R[r]);
#pragma unroll
      for (int r = 0; r < kDwRing - 3; ++r) load_g(y0 + r, G[r]);
#pragma unroll
      for (int i = 0; i < kDwRows; ++i) {
        const int y = y0 + i;
        if (y >= y1) break;
        dw_load_row<NB>(base, p.x_ld * ES, y + kDwRing - 2, x, p.H, p.W, col_ok && y + kDwRing - 2 <= y1, R[(i + kDwRing - 1) % kDwRing]);
        load_g(y + kDwRing - 3, G[(i + kDwRing - 3) % (kDwRing - 2)]);
        float g[V];
        unpack4(G[i % (kDwRing - 2)], g);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[9][k] += g[k];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) {
            float f[V];
            unpack4(R[(i + rr) % kDwRing][d], f);
#pragma unroll
            for (int k = 0; k < V; ++k) acc[rr * 3 + d][k] += g[k] * f[k];
          }
        }
      }
    }
  }
  // combine the PL column lanes, one tap at a time, in lane order
  float* row = p.part + (int64_t)blockIdx.x * 10 * p.C;
  for (int t = 0; t < 10; ++t) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k) red[threadIdx.x][k] = acc[t][k];
    __syncthreads();
    if (live && pl == 0) {
#pragma unroll
      for (int k = 0; k < V; ++k) {
        float s_ = 0.f;
        for (int q = 0; q < PL; ++q) s_ += red[q * ccw + cl][k];
        const int c = cc * V + k;
        if (t < 9) row[c * 9 + t] = s_; else row[9 * p.C + c] = s_;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ DropPath
// SegformerDropPath: per sample, out = x / keep * floor(u + keep).  u = 24 random bits of a counter-based hash of (seed, step, call, sample):
// scale[call][b] = (u24 < keep * 2^24) ? 1 / keep : 0   (rate 0 -> 1).  Replayed on the host by cvcs_amd.ops.drop_path_scales_host.
__global__ void drop_path_scales_kernel(const unsigned long long* __restrict__ state, const float* __restrict__ rates, int n, int B, float* out) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n * B) return;
  const int call = id / B;
  const float keep = 1.f - rates[call];
  float s = 1.f;
  if (rates[call] > 0.f) {
    unsigned long long z = ((state[0] ^ 0x5DEECE66Dull) ^ (state[1] * 0xD6E8FEB86659FD93ull)) + (unsigned long long)id * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    const unsigned thr24 = (unsigned)(keep * 16777216.0f);
    s = (unsigned)(z >> 40) < thr24 ? 1.f / keep : 0.f;
  }
  out[id] = s;
}

// out[b, n, :] = (res ? res[b, n, :] : 0) + scale[b] * x[b, n, :]
template <typename T>
__global__ __launch_bounds__(256) void scale_rows_add_kernel(const char* x, int64_t x_ld, const float* __restrict__ scale, const char* res, int64_t res_ld,
                                                             int64_t HW, int64_t M, int C, char* out, int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < M * CC; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    const float s = scale ? scale[pix / HW] : 1.f;
    float f[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + (pix * x_ld) * ES + cc * 16), f);
    if (res) {
      float r[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(res + (pix * res_ld) * ES + cc * 16), r);
#pragma unroll
      for (int k = 0; k < V; ++k) f[k] = r[k] + s * f[k];
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) f[k] = s * f[k];
    }
    *reinterpret_cast<uint4*>(out + (pix * out_ld) * ES + cc * 16) = Elem<T>::pack(f);
  }
}

// ------------------------------------------------------------------------------------------------ spatial-reduction attention
// q [B, N, heads*32], kv [B, Nk, 2*heads*32] (keys in channels [0, C), values in [C, 2C)), out [B, N, heads*32], lse [B*heads, N] f32.
// softmax(q k^T / sqrt(32)) v per image and head.  One thread = one query (its 32 q values and 32 accumulators in registers); the keys and
// values of the (image, head) are staged in LDS as f32, 256 at a time, and read as broadcasts; the softmax runs online over blocks of
// 8 keys.  (The matrix-core version is the next step: at Nk = 256 this VALU form costs ~70 instructions per query-key pair.)
constexpr int kSrKC = 256;
struct SrArgs {
  const char* q; const char* kv; const char* o; const char* go; char* out; char* gq;
  float* lse; float* delta; float* part;
  int64_t q_ld, kv_ld, o_ld, go_ld, out_ld, gq_ld;
  int B, N, Nk, heads, QC;
  float scale;
};

template <typename T, int kSrD>
__device__ __forceinline__ void sr_load_row(const char* base, float* f) {   // the D consecutive channels of one head of one token -> f32
  constexpr int ES = sizeof(T), V = 16 / ES;
#pragma unroll
  for (int i = 0; i < kSrD / V; ++i) Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + i * 16), f + i * V);
}
template <typename T, int kSrD>
__device__ __forceinline__ void sr_store_row(char* base, const float* f) {
  constexpr int ES = sizeof(T), V = 16 / ES;
#pragma unroll
  for (int i = 0; i < kSrD / V; ++i) *reinterpret_cast<uint4*>(base + i * 16) = Elem<T>::pack(f + i * V);
}
// stage keys [j0, j0 + n) of (b, h): sk[j][32], sv[j][32] as f32
template <typename T, int kSrD>
__device__ __forceinline__ void sr_stage_kv(const SrArgs& p, int b, int h, int j0, int n, float* sk, float* sv) {
  constexpr int ES = sizeof(T);
  const int C = p.heads * kSrD;
  for (int id = threadIdx.x; id < n * 2; id += blockDim.x) {
    const int j = id >> 1, which = id & 1;
    float f[kSrD];
    sr_load_row<T, kSrD>(p.kv + (((int64_t)b * p.Nk + j0 + j) * p.kv_ld + which * C + h * kSrD) * ES, f);
    float* dst = (which ? sv : sk) + j * kSrD;
#pragma unroll
    for (int d = 0; d < kSrD; d += 4) *reinterpret_cast<f32x4*>(dst + d) = (f32x4){f[d], f[d + 1], f[d + 2], f[d + 3]};
  }
}

template <typename T, int kSrD>
__global__ __launch_bounds__(256) void sr_attn_fwd_kernel(SrArgs p) {
  constexpr int ES = sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sr_smem[];
  float* sk = sr_smem;
  float* sv = sr_smem + kSrKC * kSrD;
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int n = blockIdx.x * 256 + threadIdx.x;
  const bool live = n < p.N;
  float q[kSrD], acc[kSrD];
  if (live) sr_load_row<T, kSrD>(p.q + (((int64_t)b * p.N + n) * p.q_ld + h * kSrD) * ES, q);
#pragma unroll
  for (int d = 0; d < kSrD; ++d) { q[d] = live ? q[d] : 0.f; acc[d] = 0.f; }
  float m = -INFINITY, l = 0.f;
  for (int j0 = 0; j0 < p.Nk; j0 += kSrKC) {
    const int nk = p.Nk - j0 < kSrKC ? p.Nk - j0 : kSrKC;
    __syncthreads();
    sr_stage_kv<T, kSrD>(p, b, h, j0, nk, sk, sv);
    __syncthreads();
    for (int jb = 0; jb < nk; jb += 8) {
      float s[8];
      float bm = -INFINITY;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float a = 0.f;
        if (jb + u < nk) {
#pragma unroll
          for (int d = 0; d < kSrD; d += 4) {
            const f32x4 kk = *reinterpret_cast<const f32x4*>(sk + (jb + u) * kSrD + d);
            a += q[d] * kk[0] + q[d + 1] * kk[1] + q[d + 2] * kk[2] + q[d + 3] * kk[3];
          }
          a *= p.scale;
        } else {
          a = -INFINITY;
        }
        s[u] = a;
        bm = fmaxf(bm, a);
      }
      const float mn = fmaxf(m, bm);
      const float corr = __expf(m - mn);           // (m = -inf on the first block: exp(-inf) = 0)
      l *= corr;
#pragma unroll
      for (int d = 0; d < kSrD; ++d) acc[d] *= corr;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (jb + u < nk) {
          const float pr = __expf(s[u] - mn);
          l += pr;
#pragma unroll
          for (int d = 0; d < kSrD; d += 4) {
            const f32x4 vv = *reinterpret_cast<const f32x4*>(sv + (jb + u) * kSrD + d);
            acc[d] += pr * vv[0]; acc[d + 1] += pr * vv[1]; acc[d + 2] += pr * vv[2]; acc[d + 3] += pr * vv[3];
          }
        }
      }
      m = mn;
    }
  }
  if (live) {
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < kSrD; ++d) acc[d] *= inv;
    sr_store_row<T, kSrD>(p.out + (((int64_t)b * p.N + n) * p.out_ld + h * kSrD) * ES, acc);
    if (p.lse) p.lse[(int64_t)bh * p.N + n] = m + __logf(l);
  }
}

// dQ (query-major, the forward's structure): p = exp(s - lse), dp = go . v, ds = p (dp - delta) scale, dq += ds k;  delta = go . o saved
template <typename T, int kSrD>
__global__ __launch_bounds__(256) void sr_attn_bwd_dq_kernel(SrArgs p) {
  constexpr int ES = sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sr_smem[];
  float* sk = sr_smem;
  float* sv = sr_smem + kSrKC * kSrD;
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int n = blockIdx.x * 256 + threadIdx.x;
  const bool live = n < p.N;
  float q[kSrD], go[kSrD], dq[kSrD];
  float delta = 0.f, lse = 0.f;
  if (live) {
    sr_load_row<T, kSrD>(p.q + (((int64_t)b * p.N + n) * p.q_ld + h * kSrD) * ES, q);
    sr_load_row<T, kSrD>(p.go + (((int64_t)b * p.N + n) * p.go_ld + h * kSrD) * ES, go);
    float o[kSrD];
    sr_load_row<T, kSrD>(p.o + (((int64_t)b * p.N + n) * p.o_ld + h * kSrD) * ES, o);
#pragma unroll
    for (int d = 0; d < kSrD; ++d) delta += go[d] * o[d];
    lse = p.lse[(int64_t)bh * p.N + n];
    p.delta[(int64_t)bh * p.N + n] = delta;
  }
#pragma unroll
  for (int d = 0; d < kSrD; ++d) { if (!live) { q[d] = 0.f; go[d] = 0.f; } dq[d] = 0.f; }
  for (int j0 = 0; j0 < p.Nk; j0 += kSrKC) {
    const int nk = p.Nk - j0 < kSrKC ? p.Nk - j0 : kSrKC;
    __syncthreads();
    sr_stage_kv<T, kSrD>(p, b, h, j0, nk, sk, sv);
    __syncthreads();
    for (int j = 0; j < nk; ++j) {
      float s = 0.f, dp = 0.f;
      f32x4 kk[kSrD / 4];
#pragma unroll
      for (int d = 0; d < kSrD; d += 4) {
        kk[d / 4] = *reinterpret_cast<const f32x4*>(sk + j * kSrD + d);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(sv + j * kSrD + d);
        s += q[d] * kk[d / 4][0] + q[d + 1] * kk[d / 4][1] + q[d + 2] * kk[d / 4][2] + q[d + 3] * kk[d / 4][3];
        dp += go[d] * vv[0] + go[d + 1] * vv[1] + go[d + 2] * vv[2] + go[d + 3] * vv[3];
      }
      const float pr = __expf(s * p.scale - lse);
      const float ds = pr * (dp - delta) * p.scale;
#pragma unroll
      for (int d = 0; d < kSrD; d += 4) {
        dq[d] += ds * kk[d / 4][0]; dq[d + 1] += ds * kk[d / 4][1]; dq[d + 2] += ds * kk[d / 4][2]; dq[d + 3] += ds * kk[d / 4][3];
      }
    }
  }
  if (live) sr_store_row<T, kSrD>(p.gq + (((int64_t)b * p.N + n) * p.gq_ld + h * kSrD) * ES, dq);
}

// dK / dV (key-major): one thread = one key (k, v, dk, dv in registers), the workgroup walks the queries [chunk*QC, (chunk+1)*QC) staged
// 64 at a time; part[(bh * chunks + chunk)][key][dk 32 | dv 32]
template <typename T, int kSrD>
__global__ __launch_bounds__(256) void sr_attn_bwd_dkv_kernel(SrArgs p) {
  constexpr int ES = sizeof(T);
  __shared__ __attribute__((aligned(16))) float sq[64 * kSrD];
  __shared__ __attribute__((aligned(16))) float sg[64 * kSrD];
  __shared__ float sl[64], sd[64];
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int chunk = blockIdx.x, chunks = gridDim.x;
  const int j = blockIdx.z * 256 + threadIdx.x;
  const bool live = j < p.Nk;
  const int C = p.heads * kSrD;
  float k[kSrD], v[kSrD], dk[kSrD], dv[kSrD];
  if (live) {
    sr_load_row<T, kSrD>(p.kv + (((int64_t)b * p.Nk + j) * p.kv_ld + h * kSrD) * ES, k);
    sr_load_row<T, kSrD>(p.kv + (((int64_t)b * p.Nk + j) * p.kv_ld + C + h * kSrD) * ES, v);
  }
#pragma unroll
  for (int d = 0; d < kSrD; ++d) { if (!live) { k[d] = 0.f; v[d] = 0.f; } dk[d] = 0.f; dv[d] = 0.f; }
  const int n0 = chunk * p.QC;
  const int n1 = n0 + p.QC < p.N ? n0 + p.QC : p.N;
  for (int nb = n0; nb < n1; nb += 64) {
    const int nq = n1 - nb < 64 ? n1 - nb : 64;
    __syncthreads();
    for (int id = threadIdx.x; id < nq * 2; id += 256) {
      const int r = id >> 1, which = id & 1;
      float f[kSrD];
      const int64_t tok = (int64_t)b * p.N + nb + r;
      if (which) sr_load_row<T, kSrD>(p.go + (tok * p.go_ld + h * kSrD) * ES, f);
      else sr_load_row<T, kSrD>(p.q + (tok * p.q_ld + h * kSrD) * ES, f);
      float* dst = (which ? sg : sq) + r * kSrD;
#pragma unroll
      for (int d = 0; d < kSrD; d += 4) *reinterpret_cast<f32x4*>(dst + d) = (f32x4){f[d], f[d + 1], f[d + 2], f[d + 3]};
    }
    if (threadIdx.x < nq) {
      sl[threadIdx.x] = p.lse[(int64_t)bh * p.N + nb + threadIdx.x];
      sd[threadIdx.x] = p.delta[(int64_t)bh * p.N + nb + threadIdx.x];
    }
    __syncthreads();
    if (live) {
      for (int r = 0; r < nq; ++r) {
        float s = 0.f, dp = 0.f;
        f32x4 qq[kSrD / 4], gg[kSrD / 4];
#pragma unroll
        for (int d = 0; d < kSrD; d += 4) {
          qq[d / 4] = *reinterpret_cast<const f32x4*>(sq + r * kSrD + d);
          gg[d / 4] = *reinterpret_cast<const f32x4*>(sg + r * kSrD + d);
          s += k[d] * qq[d / 4][0] + k[d + 1] * qq[d / 4][1] + k[d + 2] * qq[d / 4][2] + k[d + 3] * qq[d / 4][3];
          dp += v[d] * gg[d / 4][0] + v[d + 1] * gg[d / 4][1] + v[d + 2] * gg[d / 4][2] + v[d + 3] * gg[d / 4][3];
        }
        const float pr = __expf(s * p.scale - sl[r]);
        const float ds = pr * (dp - sd[r]) * p.scale;
#pragma unroll
        for (int d = 0; d < kSrD; d += 4) {
          dv[d] += pr * gg[d / 4][0]; dv[d + 1] += pr * gg[d / 4][1]; dv[d + 2] += pr * gg[d / 4][2]; dv[d + 3] += pr * gg[d / 4][3];
          dk[d] += ds * qq[d / 4][0]; dk[d + 1] += ds * qq[d / 4][1]; dk[d + 2] += ds * qq[d / 4][2]; dk[d + 3] += ds * qq[d / 4][3];
        }
      }
    }
  }
  if (live) {
    float* dst = p.part + (((int64_t)bh * chunks + chunk) * p.Nk + j) * (2 * kSrD);
#pragma unroll
    for (int d = 0; d < kSrD; d += 4) {
      *reinterpret_cast<f32x4*>(dst + d) = (f32x4){dk[d], dk[d + 1], dk[d + 2], dk[d + 3]};
      *reinterpret_cast<f32x4*>(dst + kSrD + d) = (f32x4){dv[d], dv[d + 1], dv[d + 2], dv[d + 3]};
    }
  }
}

// gkv[b, j, which*C + h*D + d] = sum over chunks (in chunk order) of part[(bh, chunk)][j][which*D + d]
template <typename T, int kSrD>
__global__ __launch_bounds__(256) void sr_attn_dkv_reduce_kernel(const float* __restrict__ part, int chunks, int B, int Nk, int heads, char* gkv, int64_t gkv_ld) {
  constexpr int E = 2 * kSrD;
  const int64_t total = (int64_t)B * heads * Nk * E;
  const int C = heads * kSrD;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int e = (int)(id % E);
    const int64_t t = id / E;
    const int j = (int)(t % Nk);
    const int bh = (int)(t / Nk);
    const int b = bh / heads, h = bh - b * heads;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += part[(((int64_t)bh * chunks + c) * Nk + j) * E + e];
    Elem<T>::st(reinterpret_cast<T*>(gkv) + ((int64_t)b * Nk + j) * gkv_ld + (e / kSrD) * C + h * kSrD + (e % kSrD), s);
  }
}

// ---- the same attention on the matrix cores (bf16, Nk a multiple of 32, head dimension D = 32 | 64 = one | two K-steps of
// v_mfma_f32_16x16x32_bf16).  Scores are computed TRANSPOSED, S^T = K Q^T (A = 16 key rows from LDS, B = the wave's 16 queries, read once):
// the accumulator layout (lane = query l % 16, registers = keys (l / 16) * 4 + r) is then exactly a B operand of the second MFMA,
// O^T = V^T P^T, if its 32 K slots are fed as [block 2j keys g*4..g*4+3 | block 2j+1 keys g*4..g*4+3] - the probabilities go from accumulators
// to operands without leaving the registers; V^T ([D][keys], row pitch + 8 bytes: conflict-free 8-byte reads) supplies the matching slots.
// Softmax statistics: per lane over its own keys, combined across the four lane groups by two shuffles; 256 keys at a time with the usual
// online rescale.
constexpr int kSrVtPitch = kSrKC * 2 + 8;     // bytes per V^T (or K^T) row in LDS

// rows [j0, j0 + n) of the key (which = 0) / value (which = 1) projection of (b, h): row-major [n][D] bf16 at `rows` and / or transposed
// [D][n] (pitch kSrVtPitch) at `tr`
template <int D>
__device__ __forceinline__ void sr_stage_bf16(const SrArgs& p, int b, int h, int which, int j0, int n, char* rows, char* tr) {
  const int C = p.heads * D;
  constexpr int CH = D / 8;
  for (int id = threadIdx.x; id < n * CH; id += blockDim.x) {
    const int j = id / CH, c = id % CH;
    const uint4 v = *reinterpret_cast<const uint4*>(p.kv + (((int64_t)b * p.Nk + j0 + j) * p.kv_ld + which * C + h * D + c * 8) * 2);
    if (rows) *reinterpret_cast<uint4*>(rows + j * (D * 2) + c * 16) = v;
    if (tr) {
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 8; ++k)
        *reinterpret_cast<uint16_t*>(tr + (c * 8 + k) * kSrVtPitch + j * 2) = (uint16_t)(k & 1 ? w[k >> 1] >> 16 : w[k >> 1] & 0xffffu);
    }
  }
}

__device__ __forceinline__ bf16x8 sr_pack8(const f32x4& a, const f32x4& b) {
  uint4 u;
  u.x = pack2_bf16(a[0], a[1]); u.y = pack2_bf16(a[2], a[3]); u.z = pack2_bf16(b[0], b[1]); u.w = pack2_bf16(b[2], b[3]);
  return __builtin_bit_cast(bf16x8, u);
}
// A operand of the transposed product: row d = db*16 + fr of a [D][keys] LDS image, K slots = keys {2j*16 + g*4 .., (2j+1)*16 + g*4 ..}
__device__ __forceinline__ bf16x8 sr_tr_operand(const char* tr, int pitch, int db, int fr, int g, int jp) {
  const char* row = tr + (db * 16 + fr) * pitch;
  const uint2 lo = *reinterpret_cast<const uint2*>(row + ((2 * jp) * 16 + g * 4) * 2);
  const uint2 hi = *reinterpret_cast<const uint2*>(row + ((2 * jp + 1) * 16 + g * 4) * 2);
  return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}
// the 16-byte operand chunk (channels ks*32 + g*8 ..) of token `tok` of a [tokens][ld] bf16 tensor, head offset included in `base`
__device__ __forceinline__ bf16x8 sr_tok_operand(const char* base, int64_t tok, int64_t ld, int ks, int g, bool ok) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (ok) v = *reinterpret_cast<const uint4*>(base + (tok * ld + ks * 32 + g * 8) * 2);
  return __builtin_bit_cast(bf16x8, v);
}

template <int D>
__global__ __launch_bounds__(256) void sr_attn_fwd_mfma_kernel(SrArgs p) {
  constexpr int KS = D / 32, DB = D / 16;
  extern __shared__ __attribute__((aligned(16))) char sr_lds[];
  char* skr = sr_lds;                              // K rows [kSrKC][D bf16]
  char* svt = sr_lds + kSrKC * D * 2;              // V^T [D][kSrVtPitch]
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int nbase = blockIdx.x * 256 + wave * 64;
  const char* qbase = p.q + (int64_t)h * D * 2;
  bf16x8 bq[4][KS];
  f32x4 o[4][DB];
  float m[4], l[4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const int n = nbase + qb * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bq[qb][ks] = sr_tok_operand(qbase, (int64_t)b * p.N + n, p.q_ld, ks, g, n < p.N);
#pragma unroll
    for (int db = 0; db < DB; ++db) o[qb][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m[qb] = -INFINITY; l[qb] = 0.f;
  }
  for (int j0 = 0; j0 < p.Nk; j0 += kSrKC) {
    const int nk = p.Nk - j0 < kSrKC ? p.Nk - j0 : kSrKC;       // a multiple of 32
    __syncthreads();
    sr_stage_bf16<D>(p, b, h, 0, j0, nk, skr, nullptr);
    sr_stage_bf16<D>(p, b, h, 1, j0, nk, nullptr, svt);
    __syncthreads();
    const int nblk = nk >> 4;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      f32x4 s[kSrKC / 16];
      float cm = -INFINITY;
#pragma unroll
      for (int jb = 0; jb < kSrKC / 16; ++jb) {
        if (jb < nblk) {
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 ka = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(skr + (jb * 16 + fr) * (D * 2) + ks * 64 + g * 16));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, bq[qb][ks], acc, 0, 0, 0);
          }
          s[jb] = acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[jb][r] *= p.scale; cm = fmaxf(cm, s[jb][r]); }
        }
      }
      cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
      cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
      const float mn = fmaxf(m[qb], cm);
      const float corr = __expf(m[qb] - mn);
      m[qb] = mn;
      l[qb] *= corr;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[qb][db][r] *= corr;
#pragma unroll
      for (int jb = 0; jb < kSrKC / 16; ++jb) {
        if (jb < nblk) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[jb][r] = __expf(s[jb][r] - mn); l[qb] += s[jb][r]; }
        }
      }
#pragma unroll
      for (int jp = 0; jp < kSrKC / 32; ++jp) {
        if (2 * jp < nblk) {
          const bf16x8 pb = sr_pack8(s[2 * jp], s[2 * jp + 1]);
#pragma unroll
          for (int db = 0; db < DB; ++db)
            o[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sr_tr_operand(svt, kSrVtPitch, db, fr, g, jp), pb, o[qb][db], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    float lt = l[qb];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const int n = nbase + qb * 16 + fr;
    if (n < p.N) {
      const float inv = 1.f / lt;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        uint2 u;
        u.x = pack2_bf16(o[qb][db][0] * inv, o[qb][db][1] * inv);
        u.y = pack2_bf16(o[qb][db][2] * inv, o[qb][db][3] * inv);
        *reinterpret_cast<uint2*>(p.out + (((int64_t)b * p.N + n) * p.out_ld + h * D + db * 16 + g * 4) * 2) = u;
      }
      if (p.lse && g == 0) p.lse[(int64_t)bh * p.N + n] = m[qb] + __logf(lt);
    }
  }
}

// dQ on the matrix cores: the forward's structure.  S^T and dP^T = V dO^T per 16-key block, dS^T = P^T (dP^T - delta) scale in the accumulator
// layout, dQ^T += K^T dS^T with K^T read from a transposed LDS image.  delta = rowsum(dO o O) is formed here (8 channels per lane and K-step,
// two shuffles) and saved for the key-major kernel.
template <int D>
__global__ __launch_bounds__(256) void sr_attn_bwd_dq_mfma_kernel(SrArgs p) {
  constexpr int KS = D / 32, DB = D / 16;
  extern __shared__ __attribute__((aligned(16))) char sr_lds[];
  char* skr = sr_lds;                              // K rows
  char* svr = sr_lds + kSrKC * D * 2;              // V rows
  char* skt = sr_lds + 2 * kSrKC * D * 2;          // K^T
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int nbase = blockIdx.x * 256 + wave * 64;
  const int64_t hoff = (int64_t)h * D * 2;
  bf16x8 bq[4][KS], bgo[4][KS];
  f32x4 dq[4][DB];
  float lse[4], delta[4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const int n = nbase + qb * 16 + fr;
    const int64_t tok = (int64_t)b * p.N + n;
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bq[qb][ks] = sr_tok_operand(p.q + hoff, tok, p.q_ld, ks, g, n < p.N);
      bgo[qb][ks] = sr_tok_operand(p.go + hoff, tok, p.go_ld, ks, g, n < p.N);
      const bf16x8 bo = sr_tok_operand(p.o + hoff, tok, p.o_ld, ks, g, n < p.N);
      float fg_[8], fo[8];
      Elem<bf16_t>::unpack(__builtin_bit_cast(uint4, bgo[qb][ks]), fg_);
      Elem<bf16_t>::unpack(__builtin_bit_cast(uint4, bo), fo);
#pragma unroll
      for (int k = 0; k < 8; ++k) d += fg_[k] * fo[k];
    }
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    delta[qb] = d;
    lse[qb] = n < p.N ? p.lse[(int64_t)bh * p.N + n] : INFINITY;
    if (n < p.N && g == 0) p.delta[(int64_t)bh * p.N + n] = d;
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[qb][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int j0 = 0; j0 < p.Nk; j0 += kSrKC) {
    const int nk = p.Nk - j0 < kSrKC ? p.Nk - j0 : kSrKC;
    __syncthreads();
    sr_stage_bf16<D>(p, b, h, 0, j0, nk, skr, skt);
    sr_stage_bf16<D>(p, b, h, 1, j0, nk, svr, nullptr);
    __syncthreads();
    const int npair = nk >> 5;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      for (int jp = 0; jp < npair; ++jp) {
        f32x4 ds[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int jb = 2 * jp + e;
          f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = st;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 ka = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(skr + (jb * 16 + fr) * (D * 2) + ks * 64 + g * 16));
            const bf16x8 va = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(svr + (jb * 16 + fr) * (D * 2) + ks * 64 + g * 16));
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, bq[qb][ks], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, bgo[qb][ks], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[e][r] = __expf(st[r] * p.scale - lse[qb]) * (dp[r] - delta[qb]) * p.scale;
        }
        const bf16x8 pb = sr_pack8(ds[0], ds[1]);
#pragma unroll
        for (int db = 0; db < DB; ++db)
          dq[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sr_tr_operand(skt, kSrVtPitch, db, fr, g, jp), pb, dq[qb][db], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const int n = nbase + qb * 16 + fr;
    if (n < p.N) {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        uint2 u;
        u.x = pack2_bf16(dq[qb][db][0], dq[qb][db][1]);
        u.y = pack2_bf16(dq[qb][db][2], dq[qb][db][3]);
        *reinterpret_cast<uint2*>(p.gq + (((int64_t)b * p.N + n) * p.gq_ld + h * D + db * 16 + g * 4) * 2) = u;
      }
    }
  }
}

// dK / dV on the matrix cores (key-major): wave w owns the key blocks {i*4 + w} of this workgroup's 256 keys (their K / V rows stay in
// registers as B operands), the workgroup walks its query chunk 32 queries at a time.  S = Q K^T and dP = dO V^T come out with lane = key,
// registers = queries - exactly the A operand (rows = keys, K slots = the 32 queries) of dV += P^T dO and dK += dS^T Q, whose B operands are
// the step's dO^T / Q^T tiles from a transposed LDS image.  Partials as the VALU kernel's: part[(bh, chunk)][key][dk D | dv D].
constexpr int kSrQtPitch = 32 * 2 + 8;
template <int D>
__global__ __launch_bounds__(256) void sr_attn_bwd_dkv_mfma_kernel(SrArgs p) {
  constexpr int KS = D / 32, DB = D / 16, CH = D / 8;
  __shared__ __attribute__((aligned(16))) char sqt[D * kSrQtPitch];      // Q^T  [D][32 queries]
  __shared__ __attribute__((aligned(16))) char sgt[D * kSrQtPitch];      // dO^T
  __shared__ __attribute__((aligned(16))) float sl[32];
  __shared__ __attribute__((aligned(16))) float sd[32];
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int chunk = blockIdx.x, chunks = gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int C = p.heads * D;
  const int64_t hoff = (int64_t)h * D * 2;
  bf16x8 kop[4][KS], vop[4][KS];
  f32x4 dk[4][DB], dv[4][DB];
  bool live[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = blockIdx.z * 256 + (i * 4 + wave) * 16 + fr;
    live[i] = blockIdx.z * 256 + (i * 4 + wave) * 16 < p.Nk;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kop[i][ks] = sr_tok_operand(p.kv + hoff, (int64_t)b * p.Nk + key, p.kv_ld, ks, g, live[i]);
      vop[i][ks] = sr_tok_operand(p.kv + hoff + (int64_t)C * 2, (int64_t)b * p.Nk + key, p.kv_ld, ks, g, live[i]);
    }
#pragma unroll
    for (int db = 0; db < DB; ++db) dk[i][db] = dv[i][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int n0 = chunk * p.QC;
  const int n1 = n0 + p.QC < p.N ? n0 + p.QC : p.N;
  for (int nb = n0; nb < n1; nb += 32) {
    // this step's Q / dO rows as A operands (lane = query fr of block a, channels ks*32 + g*8 ..)
    bf16x8 qa[2][KS], ga[2][KS];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int n = nb + a * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        qa[a][ks] = sr_tok_operand(p.q + hoff, (int64_t)b * p.N + n, p.q_ld, ks, g, n < n1);
        ga[a][ks] = sr_tok_operand(p.go + hoff, (int64_t)b * p.N + n, p.go_ld, ks, g, n < n1);
      }
    }
    __syncthreads();
    // the transposed tiles: work item = (tensor, query, 8-channel chunk)
    for (int id = threadIdx.x; id < 2 * 32 * CH; id += 256) {
      const int t = id / (32 * CH), r = (id / CH) & 31, c = id % CH;
      const int n = nb + r;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (n < n1) {
        const int64_t tok = (int64_t)b * p.N + n;
        v = t ? *reinterpret_cast<const uint4*>(p.go + hoff + (tok * p.go_ld + c * 8) * 2)
              : *reinterpret_cast<const uint4*>(p.q + hoff + (tok * p.q_ld + c * 8) * 2);
      }
      char* tr = t ? sgt : sqt;
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 8; ++k)
        *reinterpret_cast<uint16_t*>(tr + (c * 8 + k) * kSrQtPitch + r * 2) = (uint16_t)(k & 1 ? w[k >> 1] >> 16 : w[k >> 1] & 0xffffu);
    }
    if (threadIdx.x < 32) {
      const int nn = nb + threadIdx.x;
      sl[threadIdx.x] = nn < n1 ? p.lse[(int64_t)bh * p.N + nn] : INFINITY;
      sd[threadIdx.x] = nn < n1 ? p.delta[(int64_t)bh * p.N + nn] : 0.f;
    }
    __syncthreads();
    bf16x8 qT[DB], gT[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      qT[db] = sr_tr_operand(sqt, kSrQtPitch, db, fr, g, 0);
      gT[db] = sr_tr_operand(sgt, kSrQtPitch, db, fr, g, 0);
    }
    f32x4 ls[2], dl[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      ls[a] = *reinterpret_cast<const f32x4*>(sl + a * 16 + g * 4);
      dl[a] = *reinterpret_cast<const f32x4*>(sd + a * 16 + g * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!live[i]) continue;
      f32x4 pr[2], ds[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[a][ks], kop[i][ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[a][ks], vop[i][ks], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[a][r] = __expf(s[r] * p.scale - ls[a][r]);
          ds[a][r] = pr[a][r] * (dp[r] - dl[a][r]) * p.scale;
        }
      }
      const bf16x8 pa = sr_pack8(pr[0], pr[1]), da = sr_pack8(ds[0], ds[1]);
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        dv[i][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, gT[db], dv[i][db], 0, 0, 0);
        dk[i][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, qT[db], dk[i][db], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!live[i]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = blockIdx.z * 256 + (i * 4 + wave) * 16 + g * 4 + r;
      float* dst = p.part + (((int64_t)bh * chunks + chunk) * p.Nk + key) * (2 * D);
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        dst[db * 16 + fr] = dk[i][db][r];
        dst[D + db * 16 + fr] = dv[i][db][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ ConvTranspose2d(k, stride 2, p) as ONE convolution
// out[2i + a, 2j + b, co] = sum_{jy, jx, ci} in[i + da - jy, j + db - jx, ci] w[ci][co][ay + 2 jy][ax + 2 jx],  da = (a + p) >> 1, ay = (a + p) & 1.
// Over the four phases the input offsets span a T x T window, T = k/2 + 1 (k = 4: 3x3, k = 8: 5x5, padding T/2): the layer is an ordinary
// stride-1 convolution from C to 4C channels (channel = phase * CP + co; taps a phase does not touch are zero) followed by the pixel
// shuffle - no zero-stuffed input, no per-phase patch matrices.  This kernel writes that convolution's operands from the master tensor:
//   layout 0 (conv kernels):  wf [T*T][4CP][CPk],  wd [T*T flipped][CP][4CP]           (forward / data-gradient images of cvcs_conv2d)
//   layout 1 (patch GEMM):    wf [4CP][Kp], K = (ty*T + tx)*CP + ci,  wd [Kp][4CP]      (1x1 images over the cvcs_im2col matrix)
// and bias4[phase*CP + co] = bias[co].  deconv_unpack_grad_kernel is the transposed move for the gradients.
struct DeconvArgs {
  const float* w; const float* bias; void* wf; void* wd; float* bias4;
  const float* dw; const float* db4; float* gw; float* gb;
  int NC, k, p, CP, CPk, Kp, layout;
};
__device__ __forceinline__ bool deconv_tap(int k, int p, int T, int a, int t, int& kk) {   // union tap t of phase bit a -> filter index kk
  const int da = (a + p) >> 1, ay = (a + p) & 1;
  const int j = da + T / 2 - t;
  kk = ay + 2 * j;
  return j >= 0 && j < k / 2;
}
template <typename T_>
__global__ __launch_bounds__(256) void deconv_pack_kernel(DeconvArgs p) {
  const int T = p.k / 2 + 1, C4 = 4 * p.CP;
  const int64_t total = (int64_t)T * T * C4 * p.CP;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int ci = (int)(id % p.CP);
    int64_t r = id / p.CP;
    const int oc = (int)(r % C4);
    const int t = (int)(r / C4);
    const int ph = oc / p.CP, co = oc - ph * p.CP, ty = t / T, tx = t - ty * T;
    int ky, kx;
    const bool oky = deconv_tap(p.k, p.p, T, ph >> 1, ty, ky), okx = deconv_tap(p.k, p.p, T, ph & 1, tx, kx);
    const bool ok = oky && okx;
    const float v = (ok && ci < p.NC && co < p.NC) ? p.w[(((int64_t)ci * p.NC + co) * p.k + ky) * p.k + kx] : 0.f;
    if (p.layout == 0) {
      Elem<T_>::st(reinterpret_cast<T_*>(p.wf) + ((int64_t)t * C4 + oc) * p.CPk + ci, v);
      Elem<T_>::st(reinterpret_cast<T_*>(p.wd) + ((int64_t)(T * T - 1 - t) * p.CP + ci) * C4 + oc, v);
    } else {
      Elem<T_>::st(reinterpret_cast<T_*>(p.wf) + (int64_t)oc * p.Kp + t * p.CP + ci, v);
      Elem<T_>::st(reinterpret_cast<T_*>(p.wd) + ((int64_t)t * p.CP + ci) * C4 + oc, v);
    }
    if (id < C4) p.bias4[id] = (id % p.CP) < p.NC ? p.bias[id % p.CP] : 0.f;
  }
}
// gw[ci][co][ky][kx] = dW element of the (phase, tap) that uses filter tap (ky, kx) - exactly one; gb[co] = sum over the phases of db4
__global__ __launch_bounds__(256) void deconv_unpack_grad_kernel(DeconvArgs p) {
  const int T = p.k / 2 + 1;
  const int64_t total = (int64_t)p.NC * p.NC * p.k * p.k;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total + p.NC; id += (int64_t)gridDim.x * 256) {
    if (id >= total) {
      const int co = (int)(id - total);
      p.gb[co] = ((p.db4[co] + p.db4[p.CP + co]) + p.db4[2 * p.CP + co]) + p.db4[3 * p.CP + co];
      continue;
    }
    const int kx = (int)(id % p.k);
    int64_t r = id / p.k;
    const int ky = (int)(r % p.k);
    r /= p.k;
    const int co = (int)(r % p.NC), ci = (int)(r / p.NC);
    // the phase whose parity matches: ay = (a + p) & 1 = ky & 1
    const int a = ((ky & 1) - p.p) & 1, b = ((kx & 1) - p.p) & 1;
    const int ty = ((a + p.p) >> 1) + T / 2 - (ky >> 1), tx = ((b + p.p) >> 1) + T / 2 - (kx >> 1);
    const int oc = (a * 2 + b) * p.CP + co, t = ty * T + tx;
    p.gw[id] = p.layout == 0 ? p.dw[(((int64_t)oc * p.CP + ci) * T + ty) * T + tx] : p.dw[(int64_t)oc * p.Kp + t * p.CP + ci];
  }
}

// ------------------------------------------------------------------------------------------------ NHWC <-> NCHW f32 logits
// the NC-channel tail ends in an NHWC map with padded channels; the nn.Module contract (S/nets.py:346-349) returns NCHW f32 logits.
// One thread = one pixel: plane-wise accesses are coalesced across the wave, the pixel's own row is one or two 16-byte chunks per lane.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void planes_kernel(char* x, int64_t x_ld, int64_t M, int64_t HW, int C, int NC, float* planes) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < M; pix += (int64_t)gridDim.x * 256) {
    const int64_t b = pix / HW, hw = pix - b * HW;
    float* pl = planes + b * NC * HW + hw;
    for (int c0 = 0; c0 < C; c0 += V) {
      float f[V];
      if constexpr (!BWD) {
        if (c0 >= NC) break;
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + (pix * x_ld + c0) * ES), f);
#pragma unroll
        for (int k = 0; k < V; ++k)
          if (c0 + k < NC) pl[(int64_t)(c0 + k) * HW] = f[k];
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) f[k] = c0 + k < NC ? pl[(int64_t)(c0 + k) * HW] : 0.f;
        *reinterpret_cast<uint4*>(x + (pix * x_ld + c0) * ES) = Elem<T>::pack(f);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gather
// dst[r][c] (row pitch Cp) = r < Rv && c < Cv ? src[base + idx3(r; rd1, rd2, rs*) + idx3(c; cd1, cd2, cs*)] : 0,
// idx3(x; d1, d2, s0, s1, s2) = (x / (d1*d2)) * s0 + ((x / d2) % d1) * s1 + (x % d2) * s2
template <typename T>
__global__ __launch_bounds__(256) void gather_weights_kernel(const cvcs_gather_item* __restrict__ items) {
  const cvcs_gather_item it = items[blockIdx.y];
  const int64_t total = (int64_t)it.R * it.Cp;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int c = (int)(id % it.Cp), r = (int)(id / it.Cp);
    float v = 0.f;
    if (r < it.Rv && c < it.Cv && (it.rv2 == 0 || r % it.rd2 < it.rv2) && (it.cv2 == 0 || c % it.cd2 < it.cv2)) {
      const int64_t ri = (int64_t)(r / (it.rd1 * it.rd2)) * it.rs0 + (int64_t)((r / it.rd2) % it.rd1) * it.rs1 + (int64_t)(r % it.rd2) * it.rs2;
      const int64_t ci = (int64_t)(c / (it.cd1 * it.cd2)) * it.cs0 + (int64_t)((c / it.cd2) % it.cd1) * it.cs1 + (int64_t)(c % it.cd2) * it.cs2;
      v = it.src[it.base + ri + ci];
    }
    if (it.f32_out) reinterpret_cast<float*>(it.dst)[id] = v;
    else Elem<T>::st(reinterpret_cast<T*>(it.dst) + id, v);
  }
}

// the transposed gather: dst (f32, the master tensor's gradient) [base + idx(r) + idx(c)] = src[r][c] for r < Rv, c < Cv (every master element is
// hit exactly once by construction of the table: plain stores)
__global__ __launch_bounds__(256) void scatter_weight_grads_kernel(const cvcs_gather_item* __restrict__ items) {
  const cvcs_gather_item it = items[blockIdx.y];
  const int64_t total = (int64_t)it.Rv * it.Cv;
  float* master = const_cast<float*>(it.src);
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int c = (int)(id % it.Cv), r = (int)(id / it.Cv);
    if ((it.rv2 != 0 && r % it.rd2 >= it.rv2) || (it.cv2 != 0 && c % it.cd2 >= it.cv2)) continue;
    const int64_t ri = (int64_t)(r / (it.rd1 * it.rd2)) * it.rs0 + (int64_t)((r / it.rd2) % it.rd1) * it.rs1 + (int64_t)(r % it.rd2) * it.rs2;
    const int64_t ci = (int64_t)(c / (it.cd1 * it.cd2)) * it.cs0 + (int64_t)((c / it.cd2) % it.cd1) * it.cs1 + (int64_t)(c % it.cd2) * it.cs2;
    master[it.base + ri + ci] = reinterpret_cast<const float*>(it.dst)[(int64_t)r * it.Cp + c];
  }
}

}  // namespace cvcs

using namespace cvcs;

extern "C" int cvcs_im2col(const void* in, int64_t in_ld, int B, int H, int W, int C, int KH, int KW, int stride, int oy0, int ox0, int dir, int Ho,
                           int Wo, void* col, int64_t col_ld, int dtype, void* stream) {
  const char* fn = "cvcs_im2col";
  CVCS_CHECK_ARG(SF_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % V == 0, "%s: bad shape (C=%d must be a multiple of %d)", fn, C, V);
  CVCS_CHECK_ARG(KH >= 1 && KW >= 1 && KH * KW <= 64 && stride >= 1 && (dir == 1 || dir == -1), "%s: bad filter geometry", fn);
  CVCS_CHECK_ARG(col_ld >= (int64_t)KH * KW * C, "%s: col_ld=%lld < KH*KW*C", fn, (long long)col_ld);
  int rc;
  if ((rc = sf_view(fn, in, in_ld, C, es)) || (rc = sf_view(fn, col, col_ld, KH * KW * C, es))) return rc;
  ColArgs a{(const char*)in, (char*)col, in_ld, col_ld, B, H, W, C, Ho, Wo, KH, KW, stride, oy0, ox0, dir};
  const dim3 grid(sf_grid((int64_t)B * Ho * Wo * (col_ld / V)));
  if (dtype == CVCS_F32) hipLaunchKernelGGL((im2col_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((im2col_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, a);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_im2col_stem(const void* src, int src_is_u8, int B, int H, int W, int KH, int KW, int stride, int pad, const float* mean3,
                                const float* std3, int Ho, int Wo, void* col, int64_t col_ld, int dtype, void* stream) {
  const char* fn = "cvcs_im2col_stem";
  CVCS_CHECK_ARG(SF_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(src && mean3 && std3 && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0, "%s: bad argument", fn);
  CVCS_CHECK_ARG(col_ld >= (int64_t)KH * KW * 3 && col_ld % V == 0, "%s: col_ld=%lld", fn, (long long)col_ld);
  CVCS_CHECK_ARG(Ho == (H + 2 * pad - KH) / stride + 1 && Wo == (W + 2 * pad - KW) / stride + 1, "%s: Ho, Wo do not follow from the geometry", fn);
  int rc;
  if ((rc = sf_view(fn, col, col_ld, KH * KW * 3, es))) return rc;
  StemColArgs a;
  a.src = src; a.col = (char*)col; a.col_ld = col_ld; a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.KH = KH; a.KW = KW; a.s = stride; a.pad = pad;
  a.is_u8 = src_is_u8;
  for (int i = 0; i < 3; ++i) { a.mean[i] = mean3[i]; a.std[i] = std3[i]; }
  const dim3 grid(sf_grid((int64_t)B * Ho * Wo * (col_ld / V)));
  if (dtype == CVCS_F32) hipLaunchKernelGGL((im2col_stem_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((im2col_stem_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, a);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_col2im(const void* col, int64_t col_ld, int64_t phase_elems, int nph, const int* oy0, const int* ox0, int dir, int stride, int KH,
                           int KW, int B, int Ho, int Wo, int C, int H, int W, void* dx, int64_t dx_ld, int dtype, void* stream) {
  const char* fn = "cvcs_col2im";
  CVCS_CHECK_ARG(SF_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % V == 0, "%s: bad shape", fn);
  CVCS_CHECK_ARG(nph >= 1 && nph <= 4 && oy0 && ox0 && KH >= 1 && KW >= 1 && KH * KW <= 64 && stride >= 1 && (dir == 1 || dir == -1), "%s: bad geometry", fn);
  CVCS_CHECK_ARG(col_ld >= (int64_t)KH * KW * C && (nph == 1 || phase_elems >= (int64_t)B * Ho * Wo * col_ld), "%s: col_ld / phase stride", fn);
  int rc;
  if ((rc = sf_view(fn, col, col_ld, KH * KW * C, es)) || (rc = sf_view(fn, dx, dx_ld, C, es))) return rc;
  CVCS_CHECK_ARG((phase_elems * es) % 16 == 0, "%s: phase stride alignment", fn);
  Col2imArgs a;
  a.col = (const char*)col; a.dx = (char*)dx; a.col_ld = col_ld; a.dx_ld = dx_ld; a.phase_bytes = phase_elems * es;
  a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = Ho; a.Wo = Wo; a.KH = KH; a.KW = KW; a.s = stride; a.dir = dir; a.nph = nph;
  for (int i = 0; i < 4; ++i) { a.oy0[i] = i < nph ? oy0[i] : 0; a.ox0[i] = i < nph ? ox0[i] : 0; }
  const dim3 grid(sf_grid((int64_t)B * H * W * (C / V)));
  if (dtype == CVCS_F32) hipLaunchKernelGGL((col2im_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((col2im_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, a);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

static int phase_shuffle_impl(const char* fn, bool bwd, const void* ph, int64_t ph_ld, int64_t phase_elems, int B, int H, int W, int C, void* full,
                              int64_t full_ld, const void* mask, int64_t mask_ld, int relu, int dtype, void* stream) {
  CVCS_CHECK_ARG(SF_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  // phase slabs (phase_elems >= B*H*W*ph_ld) or phases interleaved in the channels of one map (ph_ld >= 4 * phase_elems, phase_elems >= C)
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % V == 0 && (phase_elems * es) % 16 == 0 &&
                 (phase_elems >= (int64_t)B * H * W * ph_ld || (phase_elems >= C && ph_ld >= 4 * phase_elems)), "%s: bad shape", fn);
  int rc;
  if ((rc = sf_view(fn, ph, ph_ld, C, es)) || (rc = sf_view(fn, full, full_ld, C, es))) return rc;
  if (mask && (rc = sf_view(fn, mask, mask_ld, C, es))) return rc;
  const dim3 grid(sf_grid((int64_t)B * 4 * H * W * (C / V)));
  hipStream_t st = (hipStream_t)stream;
#define PS_L(TT, BW) hipLaunchKernelGGL((phase_shuffle_kernel<TT, BW>), grid, dim3(256), 0, st, (const char*)ph, ph_ld, phase_elems * es, B, H, W, C, (char*)full, full_ld, (const char*)mask, mask_ld, relu)
  if (dtype == CVCS_F32) { if (bwd) PS_L(float, true); else PS_L(float, false); }
  else { if (bwd) PS_L(bf16_t, true); else PS_L(bf16_t, false); }
#undef PS_L
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_phase_shuffle(const void* ph, int64_t ph_ld, int64_t phase_elems, int B, int H, int W, int C, void* out, int64_t out_ld, int relu,
                                  int dtype, void* stream) {
  return phase_shuffle_impl("cvcs_phase_shuffle", false, ph, ph_ld, phase_elems, B, H, W, C, out, out_ld, nullptr, 0, relu, dtype, stream);
}

extern "C" int cvcs_phase_unshuffle(const void* g, int64_t g_ld, const void* relu_out, int64_t relu_out_ld, int B, int H, int W, int C, void* gph,
                                    int64_t gph_ld, int64_t phase_elems, int dtype, void* stream) {
  return phase_shuffle_impl("cvcs_phase_unshuffle", true, gph, gph_ld, phase_elems, B, H, W, C, const_cast<void*>(g), g_ld, relu_out, relu_out_ld, 0, dtype, stream);
}

static int dw_common(const char* fn, const void* x, int64_t x_ld, int B, int H, int W, int C, int dtype) {
  CVCS_CHECK_ARG(SF_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % V == 0, "%s: bad shape (C=%d must be a multiple of %d)", fn, C, V);
  return sf_view(fn, x, x_ld, C, es);
}

extern "C" int cvcs_dwconv3x3(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, const float* bias, int flip, void* out,
                              int64_t out_ld, int dtype, void* stream) {
  const char* fn = "cvcs_dwconv3x3";
  int rc;
  if ((rc = dw_common(fn, x, x_ld, B, H, W, C, dtype))) return rc;
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(w != nullptr, "%s: null weight", fn);
  if ((rc = sf_view(fn, out, out_ld, C, es))) return rc;
  DwArgs a{(const char*)x, nullptr, (char*)out, w, bias, nullptr, x_ld, 0, out_ld, B, H, W, C, flip};
  const int CC = C / V, ccw = CC < kDwChunks ? CC : kDwChunks, PL = 256 / ccw;
  const int64_t items = (int64_t)B * cdiv(H, kDwRows) * cdiv(W, PL);
  CVCS_CHECK_ARG(items < (1ll << 31), "%s: too many work items", fn);
  const dim3 grid((unsigned)items, (unsigned)cdiv(CC, ccw));
  if (dtype == CVCS_F32) hipLaunchKernelGGL((dwconv3x3_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((dwconv3x3_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, a);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_dwconv3x3_wgrad_rows(int64_t M) {
  int64_t r = cdiv(M, 256);
  return (int)(r < 1 ? 1 : (r > 512 ? 512 : r));
}

extern "C" int cvcs_dwconv3x3_wgrad(const void* x, int64_t x_ld, const void* dy, int64_t dy_ld, int B, int H, int W, int C, float* part, int dtype,
                                    void* stream) {
  const char* fn = "cvcs_dwconv3x3_wgrad";
  int rc;
  if ((rc = dw_common(fn, x, x_ld, B, H, W, C, dtype))) return rc;
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  if ((rc = sf_view(fn, dy, dy_ld, C, es))) return rc;
  CVCS_CHECK_ARG(part != nullptr, "%s: null partial buffer", fn);
  DwArgs a{(const char*)x, (const char*)dy, nullptr, nullptr, nullptr, part, x_ld, dy_ld, 0, B, H, W, C, 0};
  (void)V;
  const int CC = C / 4, ccw = CC < 64 ? CC : 64;        // four channels per thread (the wgrad kernel's own chunking)
  CVCS_CHECK_ARG(C % 4 == 0, "%s: C %% 4 != 0", fn);
  const dim3 grid((unsigned)cvcs_dwconv3x3_wgrad_rows((int64_t)B * H * W), (unsigned)cdiv(CC, ccw));
  if (dtype == CVCS_F32) hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, a);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_drop_path_scales(const uint64_t* state, const float* rates, int n, int B, float* out, void* stream) {
  CVCS_CHECK_ARG(state && rates && out && n > 0 && B > 0, "cvcs_drop_path_scales: bad argument");
  hipLaunchKernelGGL(drop_path_scales_kernel, dim3((unsigned)cdiv((int64_t)n * B, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)state, rates, n, B, out);
  CVCS_CHECK_LAUNCH("cvcs_drop_path_scales");
  return CVCS_OK;
}

extern "C" int cvcs_scale_rows_add(const void* x, int64_t x_ld, const float* scale, const void* res, int64_t res_ld, int B, int64_t HW, int C, void* out,
                                   int64_t out_ld, int dtype, void* stream) {
  const char* fn = "cvcs_scale_rows_add";
  CVCS_CHECK_ARG(SF_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % V == 0, "%s: bad shape", fn);
  int rc;
  if ((rc = sf_view(fn, x, x_ld, C, es)) || (rc = sf_view(fn, out, out_ld, C, es))) return rc;
  if (res && (rc = sf_view(fn, res, res_ld, C, es))) return rc;
  const int64_t M = (int64_t)B * HW;
  const dim3 grid(sf_grid(M * (C / V)));
  if (dtype == CVCS_F32) hipLaunchKernelGGL((scale_rows_add_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, scale, (const char*)res, res_ld, HW, M, C, (char*)out, out_ld);
  else hipLaunchKernelGGL((scale_rows_add_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, scale, (const char*)res, res_ld, HW, M, C, (char*)out, out_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

static int sr_check(const char* fn, int B, int N, int Nk, int heads, int head_dim, int dtype) {
  CVCS_CHECK_ARG(SF_DT(dtype), "%s: bad dtype", fn);
  CVCS_CHECK_ARG(B > 0 && N > 0 && Nk > 0 && heads > 0 && heads <= 64 && (int64_t)B * heads <= 65535, "%s: bad shape", fn);
  CVCS_CHECK_ARG(head_dim == 32 || head_dim == 64, "%s: head_dim=%d (built: 32 = MiT-b0, 64 = MiT-b1 ... b5)", fn, head_dim);
  return CVCS_OK;
}
static int sr_chunk(int N) {     // queries per key-major workgroup
  int qc = N / 16;
  return qc < 256 ? 256 : qc;
}
static int sr_use_mfma() {
  static const int on = getenv("CVCS_SR_ATTN_MFMA") ? atoi(getenv("CVCS_SR_ATTN_MFMA")) : 1;      // tuning / A-B knob
  return on;
}
template <typename K>
static void sr_lds_attr(K kernel, size_t lds) {
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <int D>
static int sr_fwd_launch(const SrArgs& a, const char* fn, int dtype, hipStream_t st) {
  const dim3 grid((unsigned)cdiv(a.N, 256), (unsigned)(a.B * a.heads));
  if (dtype == CVCS_BF16 && sr_use_mfma() && a.Nk % 32 == 0) {
    const size_t lds = (size_t)kSrKC * D * 2 + (size_t)D * kSrVtPitch;
    sr_lds_attr(&sr_attn_fwd_mfma_kernel<D>, lds);
    hipLaunchKernelGGL((sr_attn_fwd_mfma_kernel<D>), grid, dim3(256), lds, st, a);
  } else {
    const size_t lds = 2 * (size_t)kSrKC * D * 4;
    if (dtype == CVCS_F32) { sr_lds_attr(&sr_attn_fwd_kernel<float, D>, lds); hipLaunchKernelGGL((sr_attn_fwd_kernel<float, D>), grid, dim3(256), lds, st, a); }
    else { sr_lds_attr(&sr_attn_fwd_kernel<bf16_t, D>, lds); hipLaunchKernelGGL((sr_attn_fwd_kernel<bf16_t, D>), grid, dim3(256), lds, st, a); }
  }
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_sr_attention_fwd(const void* q, int64_t q_ld, const void* kv, int64_t kv_ld, int B, int N, int Nk, int heads, int head_dim,
                                     void* out, int64_t out_ld, float* lse, int dtype, void* stream) {
  const char* fn = "cvcs_sr_attention_fwd";
  int rc;
  if ((rc = sr_check(fn, B, N, Nk, heads, head_dim, dtype))) return rc;
  const int es = dtype == CVCS_F32 ? 4 : 2, C = heads * head_dim;
  if ((rc = sf_view(fn, q, q_ld, C, es)) || (rc = sf_view(fn, kv, kv_ld, 2 * C, es)) || (rc = sf_view(fn, out, out_ld, C, es))) return rc;
  SrArgs a{};
  a.q = (const char*)q; a.kv = (const char*)kv; a.out = (char*)out; a.lse = lse; a.q_ld = q_ld; a.kv_ld = kv_ld; a.out_ld = out_ld;
  a.B = B; a.N = N; a.Nk = Nk; a.heads = heads; a.scale = 1.0f / sqrtf((float)head_dim);
  return head_dim == 32 ? sr_fwd_launch<32>(a, fn, dtype, (hipStream_t)stream) : sr_fwd_launch<64>(a, fn, dtype, (hipStream_t)stream);
}

// floats: delta [B*heads*N] then the key-major partials [B*heads][chunks][Nk][2*head_dim]
extern "C" int64_t cvcs_sr_attention_bwd_workspace(int B, int N, int Nk, int heads, int head_dim) {
  if (B <= 0 || N <= 0 || Nk <= 0 || heads <= 0 || (head_dim != 32 && head_dim != 64)) return CVCS_EINVAL;
  const int64_t chunks = cdiv(N, sr_chunk(N));
  const int64_t delta = ((int64_t)B * heads * N + 3) / 4 * 4;
  return delta + (int64_t)B * heads * chunks * Nk * 2 * head_dim;
}

template <int D>
static int sr_bwd_launch(const SrArgs& a, const char* fn, int dtype, void* gkv, int64_t gkv_ld, hipStream_t st) {
  const int chunks = (int)cdiv(a.N, a.QC);
  const dim3 gq_grid((unsigned)cdiv(a.N, 256), (unsigned)(a.B * a.heads));
  const dim3 gk_grid((unsigned)chunks, (unsigned)(a.B * a.heads), (unsigned)cdiv(a.Nk, 256));
  const dim3 gr_grid(sf_grid((int64_t)a.B * a.heads * a.Nk * 2 * D));
  if (dtype == CVCS_BF16 && sr_use_mfma() && a.Nk % 32 == 0) {
    const size_t lds = 2 * (size_t)kSrKC * D * 2 + (size_t)D * kSrVtPitch;
    sr_lds_attr(&sr_attn_bwd_dq_mfma_kernel<D>, lds);
    hipLaunchKernelGGL((sr_attn_bwd_dq_mfma_kernel<D>), gq_grid, dim3(256), lds, st, a);
    hipLaunchKernelGGL((sr_attn_bwd_dkv_mfma_kernel<D>), gk_grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL((sr_attn_dkv_reduce_kernel<bf16_t, D>), gr_grid, dim3(256), 0, st, a.part, chunks, a.B, a.Nk, a.heads, (char*)gkv, gkv_ld);
  } else if (dtype == CVCS_F32) {
    const size_t lds = 2 * (size_t)kSrKC * D * 4;
    sr_lds_attr(&sr_attn_bwd_dq_kernel<float, D>, lds);
    hipLaunchKernelGGL((sr_attn_bwd_dq_kernel<float, D>), gq_grid, dim3(256), lds, st, a);
    hipLaunchKernelGGL((sr_attn_bwd_dkv_kernel<float, D>), gk_grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL((sr_attn_dkv_reduce_kernel<float, D>), gr_grid, dim3(256), 0, st, a.part, chunks, a.B, a.Nk, a.heads, (char*)gkv, gkv_ld);
  } else {
    const size_t lds = 2 * (size_t)kSrKC * D * 4;
    sr_lds_attr(&sr_attn_bwd_dq_kernel<bf16_t, D>, lds);
    hipLaunchKernelGGL((sr_attn_bwd_dq_kernel<bf16_t, D>), gq_grid, dim3(256), lds, st, a);
    hipLaunchKernelGGL((sr_attn_bwd_dkv_kernel<bf16_t, D>), gk_grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL((sr_attn_dkv_reduce_kernel<bf16_t, D>), gr_grid, dim3(256), 0, st, a.part, chunks, a.B, a.Nk, a.heads, (char*)gkv, gkv_ld);
  }
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_sr_attention_bwd(const void* q, int64_t q_ld, const void* kv, int64_t kv_ld, const void* o, int64_t o_ld, const void* go, int64_t go_ld,
                                     const float* lse, int B, int N, int Nk, int heads, int head_dim, void* gq, int64_t gq_ld, void* gkv,
                                     int64_t gkv_ld, float* workspace, int dtype, void* stream) {
  const char* fn = "cvcs_sr_attention_bwd";
  int rc;
  if ((rc = sr_check(fn, B, N, Nk, heads, head_dim, dtype))) return rc;
  const int es = dtype == CVCS_F32 ? 4 : 2, C = heads * head_dim;
  CVCS_CHECK_ARG(lse && workspace, "%s: null argument", fn);
  if ((rc = sf_view(fn, q, q_ld, C, es)) || (rc = sf_view(fn, kv, kv_ld, 2 * C, es)) || (rc = sf_view(fn, o, o_ld, C, es)) ||
      (rc = sf_view(fn, go, go_ld, C, es)) || (rc = sf_view(fn, gq, gq_ld, C, es)) || (rc = sf_view(fn, gkv, gkv_ld, 2 * C, es))) return rc;
  SrArgs a{};
  a.q = (const char*)q; a.kv = (const char*)kv; a.o = (const char*)o; a.go = (const char*)go; a.gq = (char*)gq;
  a.lse = const_cast<float*>(lse);
  a.delta = workspace;
  a.part = workspace + ((int64_t)B * heads * N + 3) / 4 * 4;
  a.q_ld = q_ld; a.kv_ld = kv_ld; a.o_ld = o_ld; a.go_ld = go_ld; a.gq_ld = gq_ld;
  a.B = B; a.N = N; a.Nk = Nk; a.heads = heads; a.QC = sr_chunk(N); a.scale = 1.0f / sqrtf((float)head_dim);
  return head_dim == 32 ? sr_bwd_launch<32>(a, fn, dtype, gkv, gkv_ld, (hipStream_t)stream) : sr_bwd_launch<64>(a, fn, dtype, gkv, gkv_ld, (hipStream_t)stream);
}

extern "C" int cvcs_gather_weights(const cvcs_gather_item* items_device, int n_items, int dtype, void* stream) {
  CVCS_CHECK_ARG(SF_DT(dtype), "cvcs_gather_weights: bad dtype");
  CVCS_CHECK_ARG(items_device && n_items > 0 && n_items <= 65535, "cvcs_gather_weights: bad table");
  const dim3 grid(64, (unsigned)n_items);
  if (dtype == CVCS_F32) hipLaunchKernelGGL((gather_weights_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, items_device);
  else hipLaunchKernelGGL((gather_weights_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, items_device);
  CVCS_CHECK_LAUNCH("cvcs_gather_weights");
  return CVCS_OK;
}

extern "C" int cvcs_scatter_weight_grads(const cvcs_gather_item* items_device, int n_items, void* stream) {
  CVCS_CHECK_ARG(items_device && n_items > 0 && n_items <= 65535, "cvcs_scatter_weight_grads: bad table");
  hipLaunchKernelGGL(scatter_weight_grads_kernel, dim3(64, (unsigned)n_items), dim3(256), 0, (hipStream_t)stream, items_device);
  CVCS_CHECK_LAUNCH("cvcs_scatter_weight_grads");
  return CVCS_OK;
}

static int planes_impl(const char* fn, bool bwd, void* x, int64_t x_ld, int B, int64_t HW, int C, int NC, float* planes, int dtype, void* stream) {
  CVCS_CHECK_ARG(SF_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % V == 0 && NC >= 1 && NC <= C && planes, "%s: bad argument", fn);
  int rc;
  if ((rc = sf_view(fn, x, x_ld, C, es))) return rc;
  const int64_t M = (int64_t)B * HW;
  const dim3 grid(sf_grid(M));
  hipStream_t st = (hipStream_t)stream;
#define PL_L(TT, BW) hipLaunchKernelGGL((planes_kernel<TT, BW>), grid, dim3(256), 0, st, (char*)x, x_ld, M, HW, C, NC, planes)
  if (dtype == CVCS_F32) { if (bwd) PL_L(float, true); else PL_L(float, false); }
  else { if (bwd) PL_L(bf16_t, true); else PL_L(bf16_t, false); }
#undef PL_L
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_planes_from_nhwc(const void* x, int64_t x_ld, int B, int64_t HW, int C, int NC, float* planes, int dtype, void* stream) {
  return planes_impl("cvcs_planes_from_nhwc", false, const_cast<void*>(x), x_ld, B, HW, C, NC, planes, dtype, stream);
}

extern "C" int cvcs_nhwc_from_planes(const float* planes, int B, int64_t HW, int NC, void* x, int64_t x_ld, int C, int dtype, void* stream) {
  return planes_impl("cvcs_nhwc_from_planes", true, x, x_ld, B, HW, C, NC, const_cast<float*>(planes), dtype, stream);
}

static int deconv_args_ok(const char* fn, int NC, int k, int p, int CP, int layout) {
  CVCS_CHECK_ARG(NC >= 1 && CP >= NC && CP % 16 == 0 && (k == 4 || k == 8) && p >= 0 && p < k / 2 && (layout == 0 || layout == 1), "%s: bad argument", fn);
  return CVCS_OK;
}

extern "C" int cvcs_deconv_pack(const float* w, const float* bias, int NC, int k, int p, int CP, int layout, int CPk, int Kp, void* wf, void* wd,
                                float* bias4, int dtype, void* stream) {
  const char* fn = "cvcs_deconv_pack";
  CVCS_CHECK_ARG(SF_DT(dtype), "%s: bad dtype", fn);
  int rc;
  if ((rc = deconv_args_ok(fn, NC, k, p, CP, layout))) return rc;
  const int T = k / 2 + 1;
  CVCS_CHECK_ARG(w && bias && wf && wd && bias4 && CPk >= CP && (layout == 0 || Kp >= T * T * CP), "%s: null tensor / pitches", fn);
  DeconvArgs a{};
  a.w = w; a.bias = bias; a.wf = wf; a.wd = wd; a.bias4 = bias4; a.NC = NC; a.k = k; a.p = p; a.CP = CP; a.CPk = CPk; a.Kp = Kp; a.layout = layout;
  const dim3 grid(sf_grid((int64_t)T * T * 4 * CP * CP, 1024));
  if (dtype == CVCS_F32) hipLaunchKernelGGL((deconv_pack_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((deconv_pack_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, a);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_deconv_unpack_grad(const float* dw, const float* db4, int NC, int k, int p, int CP, int layout, int Kp, float* gw, float* gb,
                                       void* stream) {
  const char* fn = "cvcs_deconv_unpack_grad";
  int rc;
  if ((rc = deconv_args_ok(fn, NC, k, p, CP, layout))) return rc;
  CVCS_CHECK_ARG(dw && db4 && gw && gb, "%s: null tensor", fn);
  DeconvArgs a{};
  a.dw = dw; a.db4 = db4; a.gw = gw; a.gb = gb; a.NC = NC; a.k = k; a.p = p; a.CP = CP; a.Kp = Kp; a.layout = layout;
  hipLaunchKernelGGL(deconv_unpack_grad_kernel, dim3(sf_grid((int64_t)NC * NC * k * k + NC, 1024)), dim3(256), 0, (hipStream_t)stream, a);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

