// HBM-bound kernels of the residual (ResNet-encoder) networks on gfx950: the block tail (BatchNorm apply + shortcut add +
// ReLU) and its backward, the stem's 3x3 / stride 2 max-pooling, zero-dilation for strided data gradients, and the
// boundary packing of the 7x7 / stride 2 stem (virtual-pixel layout, see cvcs_conv_desc.aniso in include/cvcs_hip.h).
//
// The reference has no residual network (its factory S/utils.py:174-195 is the seam these models plug into; BASELINE.json
// configs 1, 2, 5 name them).  Same conventions as elementwise.hip: one 16-byte chunk (8 bf16 / 4 f32 channels of one
// pixel) per lane and access, lanes of a wave on consecutive chunks, no atomics (gather formulations), grid-stride loops.
#include "common.h"

namespace cvcs {

static inline unsigned res_grid(int64_t total) {
  int64_t g = cdiv(total, 256);
  return (unsigned)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}

// ------------------------------------------------------------------------------------------------ block tail forward
// a thread keeps ONE 16-byte channel chunk for its whole life (the four per-channel vectors are loaded once, not per pixel)
// and walks pixels with the grid stride; a workgroup covers min(C/V, 256) chunks x 256/that pixels per step.
template <typename T>
__global__ __launch_bounds__(256) void bn_add_act_kernel(const char* y1, int64_t y1_ld, const float* __restrict__ s1,
                                                        const float* __restrict__ b1, const char* y2, int64_t y2_ld,
                                                        const float* __restrict__ s2, const float* __restrict__ b2, int64_t M,
                                                        int C, char* out, int64_t out_ld, Q8Out q8) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int ccw = CC < 256 ? CC : 256;
  const int PL = 256 / ccw;
  const int cl = threadIdx.x % ccw, pl = threadIdx.x / ccw;
  const int cc = blockIdx.y * ccw + cl;
  if (cc >= CC) return;
  float am = 0.f;
  const float q_inv = q8.q ? q8.slot[2] : 0.f, q_fmax = q8.q ? q8.slot[3] : 0.f;
  float a1[V], c1[V], a2[V], c2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = cc * V + k;
    a1[k] = s1[c]; c1[k] = b1[c];
    a2[k] = s2 ? s2[c] : 1.f; c2[k] = s2 ? b2[c] : 0.f;
  }
  for (int64_t pix = (int64_t)blockIdx.x * PL + pl; pix < M; pix += (int64_t)gridDim.x * PL) {
    float a[V], b[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y1 + pix * y1_ld * ES + cc * 16), a);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y2 + pix * y2_ld * ES + cc * 16), b);
#pragma unroll
    for (int k = 0; k < V; ++k) a[k] = fmaxf((a[k] * a1[k] + c1[k]) + (b[k] * a2[k] + c2[k]), 0.f);
    const uint4 pk = Elem<T>::pack(a);
    *reinterpret_cast<uint4*>(out + pix * out_ld * ES + cc * 16) = pk;
    if constexpr (ES == 2) {
      if (q8.q) *reinterpret_cast<uint2*>(q8.q + pix * q8.ld + cc * 8) = q8_pack8(pk, q_inv, q_fmax, q8.fmt, am);
    }
  }
  if constexpr (ES == 2) {
    if (q8.q && q8.take_amax) q8_commit_amax(am, q8.slot);
  }
}

// ------------------------------------------------------------------------------------------------ block tail backward
struct SumArgs {
  const char* out; const char* g[3]; char* dz;
  int64_t out_ld, g_ld[3], dz_ld;
  int half[3];
  int B, H, W, C;
};

template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_sum_kernel(SumArgs p) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = p.C / V;
  const int64_t total = (int64_t)p.B * p.H * p.W * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    const int x = (int)(pix % p.W);
    const int64_t t = pix / p.W;
    const int y = (int)(t % p.H);
    const int64_t b = t / p.H;
    float s[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (!p.g[j]) continue;
      int64_t gp = pix;
      if (p.half[j]) {
        if ((x | y) & 1) continue;   // a stride-2 1x1 shortcut only saw the even pixels
        gp = (b * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1);
      }
      float f[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(p.g[j] + gp * p.g_ld[j] * ES + cc * 16), f);
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] += f[k];
    }
    if (p.out) {
      float o[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(p.out + pix * p.out_ld * ES + cc * 16), o);
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] = o[k] > 0.f ? s[k] : 0.f;
    }
    *reinterpret_cast<uint4*>(p.dz + pix * p.dz_ld * ES + cc * 16) = Elem<T>::pack(s);
  }
}

// The same pass with the FIRST pass of the BatchNorm backward(s) that consume dz riding on it: the block's last BatchNorm (no ReLU between
// it and the sum: mode 2) and, in a block with a projection shortcut, the downsample BatchNorm see the same gradient dz, so
//   part_dz[row][c] = sum dz,  part_dzx[j][row][c] = sum dz * (y_j - mean_j) * invstd_j        (row = workgroup, j = 0 | 1)
// come from the values this kernel has in registers; the separate reduce launches (read dz, read y) become one extra read of y here.
// Work distribution of bn_bwd_kernel: <= 32 channel chunks x pixel lanes per workgroup, blockIdx.y over wider layers.
struct TailBwdArgs {
  SumArgs s;
  const char* y[2]; int64_t y_ld[2];
  const float* mean[2]; const float* invstd[2];
  float* part_dz; float* part_dzx[2];
  // POOLG: one more gradient term comes through a MaxPool2d(3, 2, 1) of this map (the ResNet stem): pooled gradients pg[0] (+ pg[1]) at
  // (H + 1) / 2 x (W + 1) / 2 with the arg-max tap index of every pooled element - the max-pool backward without a gradient map of its own
  const char* pg[2]; int64_t pg_ld[2]; const uint8_t* pidx;
};
template <typename T, bool POOLG>
__global__ __launch_bounds__(256) void relu_bwd_sum_bn_kernel(TailBwdArgs q) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  __shared__ float red[3][256 * V];
  const SumArgs& p = q.s;
  const int CC = p.C / V;
  const int ccw = CC < 32 ? CC : 32;
  const int PL = 256 / ccw;
  const int tid = threadIdx.x, cl = tid % ccw, pl = tid / ccw;
  const int cc = blockIdx.y * ccw + cl;
  float mu[2][V], is[2][V], s0[V], s1[2][V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    s0[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      mu[j][k] = q.y[j] ? q.mean[j][cc * V + k] : 0.f;
      is[j][k] = q.y[j] ? q.invstd[j][cc * V + k] : 0.f;
      s1[j][k] = 0.f;
    }
  }
  const bool any_half = p.half[0] | p.half[1] | p.half[2];
  const int64_t M = (int64_t)p.B * p.H * p.W;
  for (int64_t pix = (int64_t)blockIdx.x * PL + pl; pix < M; pix += (int64_t)gridDim.x * PL) {
    int x = 0, y = 0;
    int64_t b = 0;
    if (any_half || POOLG) {
      x = (int)(pix % p.W);
      const int64_t t = pix / p.W;
      y = (int)(t % p.H);
      b = t / p.H;
    }
    // every load of this pixel first
    uint4 gr[3], yr[2], orw = make_uint4(0u, 0u, 0u, 0u);
    bool gok[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      gok[j] = p.g[j] != nullptr;
      int64_t gp = pix;
      if (gok[j] && p.half[j]) {
        gok[j] = !((x | y) & 1);
        gp = (b * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1);
      }
      gr[j] = gok[j] ? *reinterpret_cast<const uint4*>(p.g[j] + gp * p.g_ld[j] * ES + cc * 16) : make_uint4(0u, 0u, 0u, 0u);
    }
    if (p.out) orw = *reinterpret_cast<const uint4*>(p.out + pix * p.out_ld * ES + cc * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      yr[j] = q.y[j] ? *reinterpret_cast<const uint4*>(q.y[j] + pix * q.y_ld[j] * ES + cc * 16) : make_uint4(0u, 0u, 0u, 0u);
    float s[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (!gok[j]) continue;
      float f[V];
      Elem<T>::unpack(gr[j], f);
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] += f[k];
    }
    if constexpr (POOLG) {
      // windows (oy, ox) that contain (y, x): 2*oy - 1 <= y <= 2*oy + 1, i.e. oy = y >> 1 and, for odd y, (y + 1) >> 1 as well; the pooled
      // gradient goes to the element the forward recorded.  All four candidate windows are fetched unconditionally (clamped addresses, a
      // duplicate or out-of-range window counts for nothing): twelve independent loads in flight instead of a chain of guarded ones.
      const int Ho = (p.H + 1) >> 1, Wo = (p.W + 1) >> 1;
      const int oyv[2] = {y >> 1, (y + 1) >> 1}, oxv[2] = {x >> 1, (x + 1) >> 1};
      const bool oyok[2] = {true, (y & 1) && oyv[1] < Ho}, oxok[2] = {true, (x & 1) && oxv[1] < Wo};
      uint2 iwr[4];
      uint4 g0r[4], g1r[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int oy = oyok[w >> 1] ? oyv[w >> 1] : oyv[0], ox = oxok[w & 1] ? oxv[w & 1] : oxv[0];
        const int64_t op = (b * Ho + oy) * Wo + ox;
        const uint8_t* ip = q.pidx + op * p.C + cc * V;
        if constexpr (V == 8) iwr[w] = *reinterpret_cast<const uint2*>(ip);
        else iwr[w] = make_uint2(*reinterpret_cast<const unsigned*>(ip), 0u);
        g0r[w] = *reinterpret_cast<const uint4*>(q.pg[0] + op * q.pg_ld[0] * ES + cc * 16);
        g1r[w] = q.pg[1] ? *reinterpret_cast<const uint4*>(q.pg[1] + op * q.pg_ld[1] * ES + cc * 16) : make_uint4(0u, 0u, 0u, 0u);
      }
      float sp[V];       // (summed apart and rounded to the storage type first: the value cvcs_maxpool3x3s2_bwd would have stored)
#pragma unroll
      for (int k = 0; k < V; ++k) sp[k] = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const bool ok = oyok[w >> 1] && oxok[w & 1];
        const int want = ok ? (y - (2 * oyv[w >> 1] - 1)) * 3 + (x - (2 * oxv[w & 1] - 1)) : 255;      // (tap indices are 0 .. 8)
        float f[V], f1[V];
        Elem<T>::unpack(g0r[w], f);
        Elem<T>::unpack(g1r[w], f1);
        const unsigned iw[2] = {iwr[w].x, iwr[w].y};
#pragma unroll
        for (int k = 0; k < V; ++k)
          if ((int)((iw[k >> 2] >> ((k & 3) * 8)) & 0xff) == want) sp[k] += f[k] + f1[k];
      }
      float spr[V];
      Elem<T>::unpack(Elem<T>::pack(sp), spr);
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] += spr[k];
    }
    if (p.out) {
      float o[V];
      Elem<T>::unpack(orw, o);
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] = o[k] > 0.f ? s[k] : 0.f;
    }
    const uint4 pk = Elem<T>::pack(s);
    *reinterpret_cast<uint4*>(p.dz + pix * p.dz_ld * ES + cc * 16) = pk;
    // the sums are those of the STORED gradient (what the apply pass will read back)
    float d[V];
    Elem<T>::unpack(pk, d);
#pragma unroll
    for (int k = 0; k < V; ++k) s0[k] += d[k];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!q.y[j]) continue;
      float yv[V];
      Elem<T>::unpack(yr[j], yv);
#pragma unroll
      for (int k = 0; k < V; ++k) s1[j][k] += d[k] * ((yv[k] - mu[j][k]) * is[j][k]);
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) { red[0][tid * V + k] = s0[k]; red[1][tid * V + k] = s1[0][k]; red[2][tid * V + k] = s1[1][k]; }
  __syncthreads();
  {
    // (every thread sums a slice of the pixel lanes of ONE of the workgroup's n_out channels, the slices are merged in order: bn_bwd_kernel)
    const int n_out = ccw * V, P = 256 / n_out > 0 ? 256 / n_out : 1;
    const int e = tid % n_out, part = tid / n_out;
    float a = 0.f, b0 = 0.f, b1 = 0.f;
    if (part < P)
      for (int w = part; w < PL; w += P) { a += red[0][w * n_out + e]; b0 += red[1][w * n_out + e]; b1 += red[2][w * n_out + e]; }
    __syncthreads();
    if (part < P) { red[0][part * n_out + e] = a; red[1][part * n_out + e] = b0; red[2][part * n_out + e] = b1; }
    __syncthreads();
    if (tid < n_out) {
      a = 0.f; b0 = 0.f; b1 = 0.f;
      for (int j = 0; j < P; ++j) { a += red[0][j * n_out + e]; b0 += red[1][j * n_out + e]; b1 += red[2][j * n_out + e]; }
      const int64_t o = (int64_t)blockIdx.x * p.C + (int64_t)blockIdx.y * n_out + e;
      q.part_dz[o] = a;
      if (q.y[0]) q.part_dzx[0][o] = b0;
      if (q.y[1]) q.part_dzx[1][o] = b1;
    }
  }
}

// ------------------------------------------------------------------------------------------------ MaxPool2d(3, 2, 1)
template <typename T>
__global__ __launch_bounds__(256) void maxpool3_fwd_kernel(const char* x, int64_t x_ld, int B, int H, int W, int C, char* out,
                                                          int64_t out_ld, uint8_t* idx) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t total = (int64_t)B * Ho * Wo * CC;
  const unsigned xbid = xcd_order(blockIdx.x, gridDim.x);      // (XCD-aware order: overlapping windows of neighbouring rows in ONE L2)
  for (int64_t id = (int64_t)xbid * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t op = id / CC;
    const int ox = (int)(op % Wo);
    const int64_t t = op / Wo;
    const int oy = (int)(t % Ho);
    const int64_t b = t / Ho;
    float best[V];
    int arg[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { best[k] = -INFINITY; arg[k] = -1; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * oy - 1 + kh;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = 2 * ox - 1 + kw;
        if ((unsigned)ix >= (unsigned)W) continue;
        float f[V];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + ((b * H + iy) * W + ix) * x_ld * ES + cc * 16), f);
#pragma unroll
        for (int k = 0; k < V; ++k)
          if (f[k] > best[k] || arg[k] < 0) { best[k] = f[k]; arg[k] = kh * 3 + kw; }   // strict '>': FIRST maximum in scan order
      }
    }
    *reinterpret_cast<uint4*>(out + op * out_ld * ES + cc * 16) = Elem<T>::pack(best);
    uint8_t* ip = idx + op * C + cc * V;
    if constexpr (V == 8) {
      uint2 w;
      w.x = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
      w.y = (unsigned)arg[4] | ((unsigned)arg[5] << 8) | ((unsigned)arg[6] << 16) | ((unsigned)arg[7] << 24);
      *reinterpret_cast<uint2*>(ip) = w;
    } else {
      *reinterpret_cast<unsigned*>(ip) = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool3_bwd_kernel(const char* g0, int64_t g0_ld, const char* g1, int64_t g1_ld,
                                                          const uint8_t* __restrict__ idx, int B, int H, int W, int C, char* dx,
                                                          int64_t dx_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t total = (int64_t)B * H * W * CC;
  const unsigned xbid = xcd_order(blockIdx.x, gridDim.x);      // (XCD-aware order: overlapping windows of neighbouring rows in ONE L2)
  for (int64_t id = (int64_t)xbid * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    const int x = (int)(pix % W);
    const int64_t t = pix / W;
    const int y = (int)(t % H);
    const int64_t b = t / H;
    float s[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.f;
    // windows (oy, ox) that contain (y, x): 2*oy - 1 <= y <= 2*oy + 1
    const int oy0 = y >> 1, oy1 = (y + 1) >> 1, ox0 = x >> 1, ox1 = (x + 1) >> 1;
    for (int oy = oy0; oy <= oy1; ++oy) {
      if (oy >= Ho) continue;
      const int kh = y - (2 * oy - 1);
      for (int ox = ox0; ox <= ox1; ++ox) {
        if (ox >= Wo) continue;
        const int want = kh * 3 + (x - (2 * ox - 1));
        const int64_t op = (b * Ho + oy) * Wo + ox;
        const uint8_t* ip = idx + op * C + cc * V;
        unsigned iw[2];
        if constexpr (V == 8) { const uint2 w = *reinterpret_cast<const uint2*>(ip); iw[0] = w.x; iw[1] = w.y; }
        else { iw[0] = *reinterpret_cast<const unsigned*>(ip); iw[1] = 0; }
        float f[V];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(g0 + op * g0_ld * ES + cc * 16), f);
        if (g1) {
          float f1[V];
          Elem<T>::unpack(*reinterpret_cast<const uint4*>(g1 + op * g1_ld * ES + cc * 16), f1);
#pragma unroll
          for (int k = 0; k < V; ++k) f[k] += f1[k];
        }
#pragma unroll
        for (int k = 0; k < V; ++k)
          if ((int)((iw[k >> 2] >> ((k & 3) * 8)) & 0xff) == want) s[k] += f[k];
      }
    }
    *reinterpret_cast<uint4*>(dx + pix * dx_ld * ES + cc * 16) = Elem<T>::pack(s);
  }
}

// ------------------------------------------------------------------------------------------------ zero dilation x2
template <typename T>
__global__ __launch_bounds__(256) void dilate2x_kernel(const char* in, int64_t in_ld, int B, int H, int W, int C, char* out,
                                                      int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int64_t total = (int64_t)B * (2 * H) * (2 * W) * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    const int x = (int)(pix % (2 * W));
    const int64_t t = pix / (2 * W);
    const int y = (int)(t % (2 * H));
    const int64_t b = t / (2 * H);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!((x | y) & 1)) v = *reinterpret_cast<const uint4*>(in + ((b * H + (y >> 1)) * W + (x >> 1)) * in_ld * ES + cc * 16);
    *reinterpret_cast<uint4*>(out + pix * out_ld * ES + cc * 16) = v;
  }
}

// ------------------------------------------------------------------------------------------------ sub-grid layouts (dilated stages)
// Layout L_d of a [B, H, W, C] map (d | H, d | W): image n = b*d*d + sy*d + sx holds the pixels (y*d + sy, x*d + sx), y < H/d, x < W/d.
// A 3x3 convolution with dilation d and padding d on the map IS a plain 3x3 / pad 1 convolution on the B*d*d sub-images of L_d (a tap at
// +-d lands on the neighbouring pixel of the same sub-image, and beyond its border exactly where the dilated tap meets the zero padding):
// the dilated stages of the DeepLab encoders then run on the halo kernels (forward, data and weight gradient) like every other 3x3.
// This gather moves a map from L_a to L_b (a, b >= 1); its backward is the same gather from L_b to L_a.
template <typename T>
__global__ __launch_bounds__(256) void regrid_kernel(const char* in, int64_t in_ld, int B, int H, int W, int C, int a, int b_, char* out,
                                                    int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int Hb = H / b_, Wb = W / b_, Ha = H / a, Wa = W / a;
  const int64_t total = (int64_t)B * H * W * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;                      // linear pixel of the destination (layout L_b)
    const int xb = (int)(pix % Wb);
    int64_t t = pix / Wb;
    const int yb = (int)(t % Hb);
    t /= Hb;
    const int sb = (int)(t % (b_ * b_));
    const int64_t img = t / (b_ * b_);
    const int Y = yb * b_ + sb / b_, X = xb * b_ + sb % b_;
    const int64_t src = ((img * (a * a) + (Y % a) * a + (X % a)) * Ha + Y / a) * Wa + X / a;
    *reinterpret_cast<uint4*>(out + pix * out_ld * ES + cc * 16) = *reinterpret_cast<const uint4*>(in + src * in_ld * ES + cc * 16);
  }
}

// ------------------------------------------------------------------------------------------------ stem boundary
template <typename T, typename S>
__global__ __launch_bounds__(256) void pack_input_stem_kernel(const S* __restrict__ src, int B, int H, int W, T* dst) {
  const int WP = W + 8;
  const int64_t total = (int64_t)B * H * WP;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int xp = (int)(id % WP);
    const int64_t t = id / WP;
    const int y = (int)(t % H);
    const int64_t b = t / H;
    const int x = xp - 3;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)x < (unsigned)W) {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (float)src[((b * 3 + c) * H + y) * W + x];
    }
    T* d = dst + id * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) Elem<T>::st(d + c, v[c]);
  }
}

template <typename T>
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, int Cout, T* wf) {
  // wf[kh][co][kw*4 + c], kw in 0..7 (kw = 7 and c = 3 are zero)
  const int total = 7 * Cout * 32;
  for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
    const int v = id % 32;
    const int co = (id / 32) % Cout;
    const int kh = id / (32 * Cout);
    const int kw = v >> 2, c = v & 3;
    const float val = (kw < 7 && c < 3) ? w[((co * 3 + c) * 7 + kh) * 7 + kw] : 0.f;
    Elem<T>::st(wf + id, val);
  }
}

__global__ void unpack_stem_wgrad_kernel(const float* __restrict__ tmp, int Cout, float* __restrict__ dw) {
  // tmp[co][v = kw*4 + c][kh]  ->  dw[co][c][kh][kw]
  const int total = Cout * 3 * 49;
  for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
    const int kw = id % 7;
    const int kh = (id / 7) % 7;
    const int c = (id / 49) % 3;
    const int co = id / 147;
    dw[id] = tmp[(co * 32 + kw * 4 + c) * 7 + kh];
  }
}

static int res_check_view(const char* fn, const void* ptr, int64_t ld, int C, int es) {
  CVCS_CHECK_ARG(ptr != nullptr, "%s: null tensor", fn);
  CVCS_CHECK_ARG(((uintptr_t)ptr % 16) == 0 && ld >= C && (ld * es) % 16 == 0, "%s: view must be 16-byte aligned with ld >= C", fn);
  return CVCS_OK;
}

}  // namespace cvcs

using namespace cvcs;

#define RES_DT_OK(dt) ((dt) == CVCS_F32 || (dt) == CVCS_BF16)

static int bn_add_act_impl(const char* fn, const void* y1, int64_t y1_ld, const float* s1, const float* b1, const void* y2, int64_t y2_ld,
                           const float* s2, const float* b2, int64_t M, int C, void* out, int64_t out_ld, int dtype, void* stream, Q8Out q8) {
  CVCS_CHECK_ARG(RES_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(M > 0 && C > 0 && C % (16 / es) == 0, "%s: bad shape", fn);
  CVCS_CHECK_ARG(s1 && b1 && ((s2 == nullptr) == (b2 == nullptr)), "%s: scale / shift vectors", fn);
  int rc;
  if ((rc = res_check_view(fn, y1, y1_ld, C, es)) || (rc = res_check_view(fn, y2, y2_ld, C, es)) ||
      (rc = res_check_view(fn, out, out_ld, C, es)))
    return rc;
  if (q8.q) {
    CVCS_CHECK_ARG(dtype == CVCS_BF16 && q8.slot && (q8.fmt == CVCS_E4M3 || q8.fmt == CVCS_E5M2), "%s: fp8 side output (bf16 storage, slot, format)", fn);
    CVCS_CHECK_ARG(q8.ld >= C && q8.ld % 8 == 0 && ((uintptr_t)q8.q % 8) == 0, "%s: fp8 view", fn);
  }
  const int CC = C / (16 / es), ccw = CC < 256 ? CC : 256;
  CVCS_CHECK_ARG(256 % ccw == 0 && CC % ccw == 0, "%s: C/%d must divide 256 or be a multiple of 256", fn, 16 / es);
  int64_t gx = cdiv(M, (256 / ccw) * 4);   // ~4 pixels per thread
  const dim3 grid((unsigned)(gx < 1 ? 1 : (gx > 8192 ? 8192 : gx)), (unsigned)(CC / ccw));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((bn_add_act_kernel<float>), grid, dim3(256), 0, st, (const char*)y1, y1_ld, s1, b1, (const char*)y2, y2_ld, s2, b2, M, C, (char*)out, out_ld, q8);
  else
    hipLaunchKernelGGL((bn_add_act_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)y1, y1_ld, s1, b1, (const char*)y2, y2_ld, s2, b2, M, C, (char*)out, out_ld, q8);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_bn_add_act(const void* y1, int64_t y1_ld, const float* s1, const float* b1, const void* y2, int64_t y2_ld,
                               const float* s2, const float* b2, int64_t M, int C, void* out, int64_t out_ld, int dtype,
                               void* stream) {
  return bn_add_act_impl("cvcs_bn_add_act", y1, y1_ld, s1, b1, y2, y2_ld, s2, b2, M, C, out, out_ld, dtype, stream, Q8Out{nullptr, 0, nullptr, 0, 0});
}

extern "C" int cvcs_bn_add_act_q8(const void* y1, int64_t y1_ld, const float* s1, const float* b1, const void* y2, int64_t y2_ld,
                                  const float* s2, const float* b2, int64_t M, int C, void* out, int64_t out_ld, void* q8, int64_t q8_ld,
                                  int fmt, float* slot, int take_amax, int dtype, void* stream) {
  CVCS_CHECK_ARG(q8 != nullptr, "cvcs_bn_add_act_q8: null fp8 output");
  return bn_add_act_impl("cvcs_bn_add_act_q8", y1, y1_ld, s1, b1, y2, y2_ld, s2, b2, M, C, out, out_ld, dtype, stream,
                         Q8Out{(char*)q8, q8_ld, slot, fmt, take_amax});
}

extern "C" int cvcs_relu_bwd_sum(const void* out, int64_t out_ld, const void* g0, int64_t g0_ld, int g0_half, const void* g1,
                                 int64_t g1_ld, int g1_half, const void* g2, int64_t g2_ld, int g2_half, int B, int H, int W,
                                 int C, void* dz, int64_t dz_ld, int dtype, void* stream) {
  const char* fn = "cvcs_relu_bwd_sum";
  CVCS_CHECK_ARG(RES_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0, "%s: bad shape", fn);
  CVCS_CHECK_ARG(g0 != nullptr, "%s: g0 is required", fn);
  CVCS_CHECK_ARG(!(g0_half || (g1 && g1_half) || (g2 && g2_half)) || (H % 2 == 0 && W % 2 == 0), "%s: a half-resolution gradient needs even H, W", fn);
  int rc;
  if (out && (rc = res_check_view(fn, out, out_ld, C, es))) return rc;
  if ((rc = res_check_view(fn, g0, g0_ld, C, es)) || (rc = res_check_view(fn, dz, dz_ld, C, es))) return rc;
  if (g1 && (rc = res_check_view(fn, g1, g1_ld, C, es))) return rc;
  if (g2 && (rc = res_check_view(fn, g2, g2_ld, C, es))) return rc;
  SumArgs a;
  a.out = (const char*)out; a.out_ld = out_ld; a.dz = (char*)dz; a.dz_ld = dz_ld;
  a.g[0] = (const char*)g0; a.g[1] = (const char*)g1; a.g[2] = (const char*)g2;
  a.g_ld[0] = g0_ld; a.g_ld[1] = g1_ld; a.g_ld[2] = g2_ld;
  a.half[0] = g0_half; a.half[1] = g1_half; a.half[2] = g2_half;
  a.B = B; a.H = H; a.W = W; a.C = C;
  const dim3 grid(res_grid((int64_t)B * H * W * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((relu_bwd_sum_kernel<float>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((relu_bwd_sum_kernel<bf16_t>), grid, dim3(256), 0, st, a);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_maxpool3x3s2_fwd(const void* x, int64_t x_ld, int B, int H, int W, int C, void* out, int64_t out_ld,
                                     uint8_t* idx, int dtype, void* stream) {
  const char* fn = "cvcs_maxpool3x3s2_fwd";
  CVCS_CHECK_ARG(RES_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0 && idx && ((uintptr_t)idx % 8) == 0, "%s: bad shape / idx", fn);
  int rc;
  if ((rc = res_check_view(fn, x, x_ld, C, es)) || (rc = res_check_view(fn, out, out_ld, C, es))) return rc;
  const dim3 grid(res_grid((int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((maxpool3_fwd_kernel<float>), grid, dim3(256), 0, st, (const char*)x, x_ld, B, H, W, C, (char*)out, out_ld, idx);
  else hipLaunchKernelGGL((maxpool3_fwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)x, x_ld, B, H, W, C, (char*)out, out_ld, idx);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_maxpool3x3s2_bwd(const void* g0, int64_t g0_ld, const void* g1, int64_t g1_ld, const uint8_t* idx, int B,
                                     int H, int W, int C, void* dx, int64_t dx_ld, int dtype, void* stream) {
  const char* fn = "cvcs_maxpool3x3s2_bwd";
  CVCS_CHECK_ARG(RES_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0 && idx && ((uintptr_t)idx % 8) == 0, "%s: bad shape / idx", fn);
  int rc;
  if ((rc = res_check_view(fn, g0, g0_ld, C, es)) || (rc = res_check_view(fn, dx, dx_ld, C, es))) return rc;
  if (g1 && (rc = res_check_view(fn, g1, g1_ld, C, es))) return rc;
  const dim3 grid(res_grid((int64_t)B * H * W * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((maxpool3_bwd_kernel<float>), grid, dim3(256), 0, st, (const char*)g0, g0_ld, (const char*)g1, g1_ld, idx, B, H, W, C, (char*)dx, dx_ld);
  else hipLaunchKernelGGL((maxpool3_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)g0, g0_ld, (const char*)g1, g1_ld, idx, B, H, W, C, (char*)dx, dx_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_dilate2x(const void* in, int64_t in_ld, int B, int H, int W, int C, void* out, int64_t out_ld, int dtype,
                             void* stream) {
  const char* fn = "cvcs_dilate2x";
  CVCS_CHECK_ARG(RES_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0, "%s: bad shape", fn);
  int rc;
  if ((rc = res_check_view(fn, in, in_ld, C, es)) || (rc = res_check_view(fn, out, out_ld, C, es))) return rc;
  const dim3 grid(res_grid((int64_t)B * 4 * H * W * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((dilate2x_kernel<float>), grid, dim3(256), 0, st, (const char*)in, in_ld, B, H, W, C, (char*)out, out_ld);
  else hipLaunchKernelGGL((dilate2x_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)in, in_ld, B, H, W, C, (char*)out, out_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_regrid(const void* in, int64_t in_ld, int B, int H, int W, int C, int a, int b, void* out, int64_t out_ld, int dtype,
                           void* stream) {
  const char* fn = "cvcs_regrid";
  CVCS_CHECK_ARG(RES_DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0 && a >= 1 && b >= 1 && H % a == 0 && W % a == 0 && H % b == 0 && W % b == 0,
                 "%s: bad shape (both grid factors must divide H and W)", fn);
  int rc;
  if ((rc = res_check_view(fn, in, in_ld, C, es)) || (rc = res_check_view(fn, out, out_ld, C, es))) return rc;
  const dim3 grid(res_grid((int64_t)B * H * W * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((regrid_kernel<float>), grid, dim3(256), 0, st, (const char*)in, in_ld, B, H, W, C, a, b, (char*)out, out_ld);
  else hipLaunchKernelGGL((regrid_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)in, in_ld, B, H, W, C, a, b, (char*)out, out_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_pack_input_stem(const void* src, int src_is_u8, int B, int H, int W, void* dst, int dtype, void* stream) {
  const char* fn = "cvcs_pack_input_stem";
  CVCS_CHECK_ARG(RES_DT_OK(dtype), "%s: bad dtype", fn);
  CVCS_CHECK_ARG(src && dst && B > 0 && H > 0 && W > 0 && W % 2 == 0 && ((uintptr_t)dst % 16) == 0, "%s: bad arguments", fn);
  const dim3 grid(res_grid((int64_t)B * H * (W + 8)));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) {
    if (src_is_u8) hipLaunchKernelGGL((pack_input_stem_kernel<float, uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)src, B, H, W, (float*)dst);
    else hipLaunchKernelGGL((pack_input_stem_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, B, H, W, (float*)dst);
  } else {
    if (src_is_u8) hipLaunchKernelGGL((pack_input_stem_kernel<bf16_t, uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)src, B, H, W, (bf16_t*)dst);
    else hipLaunchKernelGGL((pack_input_stem_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const float*)src, B, H, W, (bf16_t*)dst);
  }
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_pack_stem_weight(const float* w, int Cout, void* w_fwd, int dtype, void* stream) {
  const char* fn = "cvcs_pack_stem_weight";
  CVCS_CHECK_ARG(RES_DT_OK(dtype) && w && w_fwd && Cout > 0, "%s: bad arguments", fn);
  const dim3 grid((unsigned)cdiv(7 * Cout * 32, 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((pack_stem_weight_kernel<float>), grid, dim3(256), 0, st, w, Cout, (float*)w_fwd);
  else hipLaunchKernelGGL((pack_stem_weight_kernel<bf16_t>), grid, dim3(256), 0, st, w, Cout, (bf16_t*)w_fwd);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_unpack_stem_wgrad(const float* tmp, int Cout, float* dw, void* stream) {
  const char* fn = "cvcs_unpack_stem_wgrad";
  CVCS_CHECK_ARG(tmp && dw && Cout > 0, "%s: bad arguments", fn);
  hipLaunchKernelGGL(unpack_stem_wgrad_kernel, dim3((unsigned)cdiv(Cout * 147, 256)), dim3(256), 0, (hipStream_t)stream, tmp, Cout, dw);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_bn_bwd_rows(int64_t M);

extern "C" int cvcs_relu_bwd_sum_bn(const cvcs_tail_bwd_desc* d, void* stream) {
  const char* fn = "cvcs_relu_bwd_sum_bn";
  CVCS_CHECK_ARG(d != nullptr, "%s: null descriptor", fn);
  CVCS_CHECK_ARG(RES_DT_OK(d->dtype), "%s: bad dtype", fn);
  const int es = d->dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->C % V == 0, "%s: bad shape", fn);
  const int CC = d->C / V;
  CVCS_CHECK_ARG((CC & (CC - 1)) == 0 || CC % 32 == 0, "%s: C/%d must be a power of two or a multiple of 32", fn, V);
  CVCS_CHECK_ARG((d->g[0] != nullptr || d->pool_idx != nullptr) && d->dz && d->y[0] && d->part_dz && d->part_dzx[0],
                 "%s: a gradient (g[0] or the pooled one), dz, y[0] and the partial buffers are required", fn);
  CVCS_CHECK_ARG(!d->pool_idx || d->pool_g[0], "%s: pool_idx needs pool_g[0]", fn);
  CVCS_CHECK_ARG(!(d->g_half[0] || (d->g[1] && d->g_half[1]) || (d->g[2] && d->g_half[2])) || (d->H % 2 == 0 && d->W % 2 == 0),
                 "%s: a half-resolution gradient needs even H, W", fn);
  int rc;
  if (d->out && (rc = res_check_view(fn, d->out, d->out_ld, d->C, es))) return rc;
  if ((rc = res_check_view(fn, d->dz, d->dz_ld, d->C, es))) return rc;
  TailBwdArgs a{};
  for (int j = 0; j < 3; ++j) {
    if (d->g[j] && (rc = res_check_view(fn, d->g[j], d->g_ld[j], d->C, es))) return rc;
    a.s.g[j] = (const char*)d->g[j]; a.s.g_ld[j] = d->g_ld[j]; a.s.half[j] = d->g[j] ? d->g_half[j] : 0;
  }
  for (int j = 0; j < 2; ++j) {
    if (d->y[j]) {
      if ((rc = res_check_view(fn, d->y[j], d->y_ld[j], d->C, es))) return rc;
      CVCS_CHECK_ARG(d->mean[j] && d->invstd[j] && d->part_dzx[j], "%s: BatchNorm %d needs mean, invstd and its partial buffer", fn, j);
    }
    a.y[j] = (const char*)d->y[j]; a.y_ld[j] = d->y_ld[j]; a.mean[j] = d->mean[j]; a.invstd[j] = d->invstd[j]; a.part_dzx[j] = d->part_dzx[j];
  }
  a.s.out = (const char*)d->out; a.s.out_ld = d->out_ld; a.s.dz = (char*)d->dz; a.s.dz_ld = d->dz_ld;
  a.s.B = d->B; a.s.H = d->H; a.s.W = d->W; a.s.C = d->C;
  a.part_dz = d->part_dz;
  const int ccw = CC < 32 ? CC : 32;
  const dim3 grid((unsigned)cvcs_bn_bwd_rows((int64_t)d->B * d->H * d->W), (unsigned)(CC / ccw));
  hipStream_t st = (hipStream_t)stream;
  if (d->pool_idx) {
    for (int j = 0; j < 2; ++j) {
      if (d->pool_g[j] && (rc = res_check_view(fn, d->pool_g[j], d->pool_g_ld[j], d->C, es))) return rc;
      a.pg[j] = (const char*)d->pool_g[j]; a.pg_ld[j] = d->pool_g_ld[j];
    }
    a.pidx = d->pool_idx;
    if (d->dtype == CVCS_F32) hipLaunchKernelGGL((relu_bwd_sum_bn_kernel<float, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((relu_bwd_sum_bn_kernel<bf16_t, true>), grid, dim3(256), 0, st, a);
  } else {
    if (d->dtype == CVCS_F32) hipLaunchKernelGGL((relu_bwd_sum_bn_kernel<float, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((relu_bwd_sum_bn_kernel<bf16_t, false>), grid, dim3(256), 0, st, a);
  }
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

