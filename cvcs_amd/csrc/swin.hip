// Kernels of the Swin-T + UPerNet path (BASELINE.json configs[3]; north star: "patch-embed + windowed attention for the
// ViT/Swin encoder variant"; the reference documents TSwin / BSwin at configs/train/README.txt:44-45 without a definition) on
// gfx950.  Tokens are NHWC pixels: a token row = the C channels of one pixel, so every linear layer is a 1x1 convolution on the
// GEMM kernels of conv_igemm.hip / wgrad.hip; this file holds what is NOT a GEMM:
//   LayerNorm fwd / bwd (one wave per token, values held in registers, f32 statistics saved for the backward),
//   GELU (erf) fwd / bwd, 4x4 patch packing of the u8 tile, 2x2 patch-merging gather / its transpose,
//   window gather with zero padding to a multiple of 7 and cyclic shift / its inverse fused with the residual add,
//   window-7 multi-head attention fwd / bwd: QK^T * scale + relative-position bias (+ the -100 region mask of shifted windows),
//     softmax, PV - one wave per (window, head), scores in registers, K / V in LDS; the backward recomputes the probabilities,
//     and the gradient of the bias table comes out as per-chunk partial sums reduced in a fixed order (no atomics),
//   adaptive average pooling (UPerNet's pyramid pooling) fwd / bwd, bilinear resize between arbitrary sizes fwd / bwd (gather).
// A first, correct-by-construction version: the attention runs on the VALU (49 x 49 x 32 per window-head is 0.5 % of the
// model's multiplies); its MFMA version (49 padded to 64) is listed in DESIGN.md section 7.
#include "common.h"

namespace cvcs {

static inline unsigned sw_grid(int64_t total, int cap = 256 * 32) {
  int64_t g = cdiv(total, 256);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// one wave per token; lane l holds the 16-byte chunks l, l+64, ... (C <= 1536: 3 chunks per lane in bf16, 6 in f32)
template <typename T> struct LnChunks { static constexpr int n = sizeof(T) == 4 ? 6 : 3; };
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const char* x, int64_t x_ld, int64_t M, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, char* out, int64_t out_ld,
                                                           float* mean_out, float* invstd_out) {
  constexpr int ES = sizeof(T), V = 16 / ES, kLnMaxChunks = LnChunks<T>::n;
  const int CC = C / V;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t tok = (int64_t)blockIdx.x * 4 + wave; tok < M; tok += (int64_t)gridDim.x * 4) {
    float v[kLnMaxChunks][V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxChunks; ++i) {
      const int cc = lane + 64 * i;
      if (cc < CC) {
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + tok * x_ld * ES + cc * 16), v[i]);
#pragma unroll
        for (int k = 0; k < V; ++k) s += v[i][k];
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxChunks; ++i)
      if (lane + 64 * i < CC) {
#pragma unroll
        for (int k = 0; k < V; ++k) { const float d = v[i][k] - mean; q += d * d; }
      }
    const float invstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < kLnMaxChunks; ++i) {
      const int cc = lane + 64 * i;
      if (cc < CC) {
#pragma unroll
        for (int k = 0; k < V; ++k) v[i][k] = (v[i][k] - mean) * invstd * gamma[cc * V + k] + beta[cc * V + k];
        *reinterpret_cast<uint4*>(out + tok * out_ld * ES + cc * 16) = Elem<T>::pack(v[i]);
      }
    }
    if (lane == 0) { mean_out[tok] = mean; invstd_out[tok] = invstd; }
  }
}

// backward: dx = invstd * (gamma*g - mean_c(gamma*g) - xhat * mean_c(gamma*g*xhat)); partial rows part[block][0|1][C] of
// dgamma = sum_tokens g*xhat and dbeta = sum_tokens g (reduced by cvcs_colsum_finalize over 2C columns)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const char* x, int64_t x_ld, const char* g, int64_t g_ld, int64_t M, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, char* dx, int64_t dx_ld, float* part) {
  constexpr int ES = sizeof(T), V = 16 / ES, kLnMaxChunks = LnChunks<T>::n;
  extern __shared__ float sred[];      // [4 waves][2][C]
  const int CC = C / V;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[kLnMaxChunks][V], db[kLnMaxChunks][V];
#pragma unroll
  for (int i = 0; i < kLnMaxChunks; ++i)
#pragma unroll
    for (int k = 0; k < V; ++k) { dg[i][k] = 0.f; db[i][k] = 0.f; }
  for (int64_t tok = (int64_t)blockIdx.x * 4 + wave; tok < M; tok += (int64_t)gridDim.x * 4) {
    const float mu = mean[tok], is = invstd[tok];
    float xh[kLnMaxChunks][V], gg[kLnMaxChunks][V];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxChunks; ++i) {
      const int cc = lane + 64 * i;
      if (cc < CC) {
        float gv[V];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + tok * x_ld * ES + cc * 16), xh[i]);
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(g + tok * g_ld * ES + cc * 16), gv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
          xh[i][k] = (xh[i][k] - mu) * is;
          dg[i][k] += gv[k] * xh[i][k];
          db[i][k] += gv[k];
          gg[i][k] = gv[k] * gamma[cc * V + k];
          s1 += gg[i][k];
          s2 += gg[i][k] * xh[i][k];
        }
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < kLnMaxChunks; ++i) {
      const int cc = lane + 64 * i;
      if (cc < CC) {
        float o[V];
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = is * (gg[i][k] - s1 - xh[i][k] * s2);
        *reinterpret_cast<uint4*>(dx + tok * dx_ld * ES + cc * 16) = Elem<T>::pack(o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kLnMaxChunks; ++i) {
    const int cc = lane + 64 * i;
    if (cc < CC)
#pragma unroll
      for (int k = 0; k < V; ++k) {
        sred[(wave * 2 + 0) * C + cc * V + k] = dg[i][k];
        sred[(wave * 2 + 1) * C + cc * V + k] = db[i][k];
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    part[(int64_t)blockIdx.x * 2 * C + i] = sred[(0 * 2 + which) * C + c] + sred[(1 * 2 + which) * C + c] + sred[(2 * 2 + which) * C + c] +
                                            sred[(3 * 2 + which) * C + c];
  }
}

// ------------------------------------------------------------------------------------------------ GELU (erf)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const char* x, int64_t x_ld, const char* g, int64_t g_ld, int64_t M, int C, char* out,
                                                  int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int64_t total = M * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t tok = id / CC;
    float f[V], gv[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + tok * x_ld * ES + cc * 16), f);
    if constexpr (BWD) Elem<T>::unpack(*reinterpret_cast<const uint4*>(g + tok * g_ld * ES + cc * 16), gv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float cdf = 0.5f * (1.f + erff(f[k] * 0.70710678118654752f));
      if constexpr (BWD) f[k] = gv[k] * (cdf + f[k] * 0.3989422804014327f * __expf(-0.5f * f[k] * f[k]));
      else f[k] = f[k] * cdf;
    }
    *reinterpret_cast<uint4*>(out + tok * out_ld * ES + cc * 16) = Elem<T>::pack(f);
  }
}

// ------------------------------------------------------------------------------------------------ patch packing / merging
// u8 | f32 NCHW [B,3,S,S] -> [B, S/4, S/4, 64]: channel c*16 + kh*4 + kw = pixel (4y+kh, 4x+kw) of colour c (the flattening of the
// [96,3,4,4] projection weight), channels 48..63 zero
template <typename T, typename S>
__global__ __launch_bounds__(256) void pack_patches_kernel(const S* __restrict__ src, int B, int H, int W, T* dst) {
  const int Hp = H / 4, Wp = W / 4;
  const int64_t total = (int64_t)B * Hp * Wp * 64;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int ch = (int)(id & 63);
    const int64_t tok = id >> 6;
    const int px = (int)(tok % Wp);
    const int64_t t = tok / Wp;
    const int py = (int)(t % Hp);
    const int64_t b = t / Hp;
    float v = 0.f;
    if (ch < 48) {
      const int c = ch >> 4, kh = (ch >> 2) & 3, kw = ch & 3;
      v = (float)src[((b * 3 + c) * H + 4 * py + kh) * W + 4 * px + kw];
    }
    Elem<T>::st(dst + id, v);
  }
}

// patch merging gather (SwinPatchMerging): out[b, y, x, blk*C + c] = in[b, 2y + row, 2x + col, c], blk = col*2 + row;
// REVERSE = its transpose: gin[b, y, x, c] = gout[b, y/2, x/2, ((x&1)*2 + (y&1))*C + c]
template <typename T, bool REVERSE>
__global__ __launch_bounds__(256) void patch_merge_kernel(const char* in, int64_t in_ld, int B, int H, int W, int C, char* out, int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;               // chunks per source pixel
  const int64_t total = (int64_t)B * H * W * CC;   // over the FINE pixels (H x W = the unmerged map)
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t pix = id / CC;
    const int x = (int)(pix % W);
    const int64_t t = pix / W;
    const int y = (int)(t % H);
    const int64_t b = t / H;
    const int blk = (x & 1) * 2 + (y & 1);
    const int64_t mp = (b * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1);
    if constexpr (REVERSE)
      *reinterpret_cast<uint4*>(out + pix * out_ld * ES + cc * 16) = *reinterpret_cast<const uint4*>(in + (mp * in_ld + blk * C) * ES + cc * 16);
    else
      *reinterpret_cast<uint4*>(out + (mp * out_ld + blk * C) * ES + cc * 16) = *reinterpret_cast<const uint4*>(in + pix * in_ld * ES + cc * 16);
  }
}

// ------------------------------------------------------------------------------------------------ window gather / reverse
// window token (b, wy, wx, i, j) <-> map pixel: the map is zero-padded to Hp x Wp (multiples of 7) and rolled by -shift, so the
// token at rolled position (wy*7+i, wx*7+j) is the padded pixel ((.. + shift) mod Hp, (.. + shift) mod Wp), a zero if outside H x W.
// GATHER : win[tok] = map[src(tok)] | 0          (forward partition; backward of the reverse)
// REVERSE: out[pix] = (res ? res[pix] : 0) + win[tok(pix)]   (forward un-partition + residual add; backward of the gather)
template <typename T, bool REVERSE>
__global__ __launch_bounds__(256) void window_kernel(const char* a, int64_t a_ld, const char* res, int64_t res_ld, int B, int H, int W, int C,
                                                    int shift, char* out, int64_t out_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES, WS = 7;
  const int CC = C / V;
  const int Hp = (H + WS - 1) / WS * WS, Wp = (W + WS - 1) / WS * WS;
  const int nwx = Wp / WS, nwy = Hp / WS;
  const int64_t total = (REVERSE ? (int64_t)B * H * W : (int64_t)B * Hp * Wp) * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t e = id / CC;
    if constexpr (!REVERSE) {
      // e = window token index: ((b*nwy + wy)*nwx + wx)*49 + i*7 + j
      const int t49 = (int)(e % 49);
      const int64_t w = e / 49;
      const int wx = (int)(w % nwx);
      const int64_t t = w / nwx;
      const int wy = (int)(t % nwy);
      const int64_t b = t / nwy;
      int sy = wy * WS + t49 / 7 + shift, sx = wx * WS + t49 % 7 + shift;
      sy = sy >= Hp ? sy - Hp : sy; sx = sx >= Wp ? sx - Wp : sx;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (sy < H && sx < W) v = *reinterpret_cast<const uint4*>(a + (((b * H + sy) * W + sx) * a_ld) * ES + cc * 16);
      *reinterpret_cast<uint4*>(out + e * out_ld * ES + cc * 16) = v;
    } else {
      const int x = (int)(e % W);
      const int64_t t = e / W;
      const int y = (int)(t % H);
      const int64_t b = t / H;
      int py = y - shift, px = x - shift;
      py = py < 0 ? py + Hp : py; px = px < 0 ? px + Wp : px;
      const int64_t tok = ((b * nwy + py / WS) * nwx + px / WS) * 49 + (py % WS) * 7 + px % WS;
      float f[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(a + tok * a_ld * ES + cc * 16), f);
      if (res) {
        float r[V];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(res + e * res_ld * ES + cc * 16), r);
#pragma unroll
        for (int k = 0; k < V; ++k) f[k] += r[k];
      }
      *reinterpret_cast<uint4*>(out + e * out_ld * ES + cc * 16) = Elem<T>::pack(f);
    }
  }
}

// ------------------------------------------------------------------------------------------------ window attention
// qkv [T][3C] (q | k | v, head h = channels h*32 .. h*32+31 of each), T = nwin*49 window tokens; one wave per (window, head):
// lane t < 49 owns query row t.  bias_table [169][heads] f32.  Shifted windows: region(token) from its rolled position.
struct AttnArgs {
  const char* qkv; int64_t qkv_ld;
  const char* dout; int64_t dout_ld;   // backward: gradient of the attention output [T][C]
  char* out; int64_t out_ld;           // forward: output [T][C]; backward: dqkv [T][3C]
  const float* table; float* dtable_part;
  int nwin, heads, C, nwy, nwx, Hp, Wp, shift;
  float scale;
  int wins_per_chunk;
};

__device__ __forceinline__ int attn_region(int y, int x, int Hp, int Wp, int shift) {
  const int hr = (y >= Hp - 7) + (y >= Hp - shift), wr = (x >= Wp - 7) + (x >= Wp - shift);
  return hr * 3 + wr;
}

template <typename T>
__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnArgs p) {
  constexpr int ES = sizeof(T), D = 32;
  __shared__ float sk[49][D + 1], sv[49][D + 1], sb[169];
  const int h = blockIdx.y, t = threadIdx.x;
  for (int i = t; i < 169; i += 64) sb[i] = p.table[i * p.heads + h];
  for (int w = blockIdx.x; w < p.nwin; w += gridDim.x) {
    const int64_t tok0 = (int64_t)w * 49;
    __syncthreads();
    float q[D];
    if (t < 49) {
      const char* row = p.qkv + (tok0 + t) * p.qkv_ld * ES;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        q[d] = Elem<T>::ld(reinterpret_cast<const T*>(row) + h * D + d) * p.scale;
        sk[t][d] = Elem<T>::ld(reinterpret_cast<const T*>(row) + p.C + h * D + d);
        sv[t][d] = Elem<T>::ld(reinterpret_cast<const T*>(row) + 2 * p.C + h * D + d);
      }
    }
    __syncthreads();
    if (t < 49) {
      const int wi = w % (p.nwy * p.nwx), wy = wi / p.nwx, wx = wi % p.nwx;
      const int ti = t / 7, tj = t % 7;
      const int myreg = p.shift ? attn_region(wy * 7 + ti, wx * 7 + tj, p.Hp, p.Wp, p.shift) : 0;
      float s[49];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 49; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) a += q[d] * sk[j][d];
        a += sb[(ti - j / 7 + 6) * 13 + (tj - j % 7 + 6)];
        if (p.shift && attn_region(wy * 7 + j / 7, wx * 7 + j % 7, p.Hp, p.Wp, p.shift) != myreg) a += -100.f;
        s[j] = a;
        mx = fmaxf(mx, a);
      }
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < 49; ++j) { s[j] = __expf(s[j] - mx); se += s[j]; }
      const float inv = 1.f / se;
      float o[D];
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] = 0.f;
#pragma unroll
      for (int j = 0; j < 49; ++j) {
        const float pj = s[j] * inv;
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] += pj * sv[j][d];
      }
      T* orow = reinterpret_cast<T*>(p.out + (tok0 + t) * p.out_ld * ES) + h * D;
#pragma unroll
      for (int d = 0; d < D; ++d) Elem<T>::st(orow + d, o[d]);
    }
  }
}

// backward: grid (chunks, heads); a workgroup walks `wins_per_chunk` windows of one head and keeps row t of the summed dS in
// registers -> dtable_part[chunk][head][49*49]
template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_kernel(AttnArgs p) {
  constexpr int ES = sizeof(T), D = 32;
  __shared__ float sq[49][D + 1], sk[49][D + 1], sv[49][D + 1], sdo[49][D + 1], sP[49][50], sdS[49][50], sb[169];
  const int h = blockIdx.y, t = threadIdx.x;
  for (int i = t; i < 169; i += 64) sb[i] = p.table[i * p.heads + h];
  float dsum[49];
#pragma unroll
  for (int j = 0; j < 49; ++j) dsum[j] = 0.f;
  const int w0 = blockIdx.x * p.wins_per_chunk;
  for (int w = w0; w < w0 + p.wins_per_chunk && w < p.nwin; ++w) {
    const int64_t tok0 = (int64_t)w * 49;
    __syncthreads();
    if (t < 49) {
      const char* row = p.qkv + (tok0 + t) * p.qkv_ld * ES;
      const char* drow = p.dout + (tok0 + t) * p.dout_ld * ES;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        sq[t][d] = Elem<T>::ld(reinterpret_cast<const T*>(row) + h * D + d);
        sk[t][d] = Elem<T>::ld(reinterpret_cast<const T*>(row) + p.C + h * D + d);
        sv[t][d] = Elem<T>::ld(reinterpret_cast<const T*>(row) + 2 * p.C + h * D + d);
        sdo[t][d] = Elem<T>::ld(reinterpret_cast<const T*>(drow) + h * D + d);
      }
    }
    __syncthreads();
    if (t < 49) {
      const int wi = w % (p.nwy * p.nwx), wy = wi / p.nwx, wx = wi % p.nwx;
      const int ti = t / 7, tj = t % 7;
      const int myreg = p.shift ? attn_region(wy * 7 + ti, wx * 7 + tj, p.Hp, p.Wp, p.shift) : 0;
      float s[49];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 49; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) a += sq[t][d] * sk[j][d];
        a = a * p.scale + sb[(ti - j / 7 + 6) * 13 + (tj - j % 7 + 6)];
        if (p.shift && attn_region(wy * 7 + j / 7, wx * 7 + j % 7, p.Hp, p.Wp, p.shift) != myreg) a += -100.f;
        s[j] = a;
        mx = fmaxf(mx, a);
      }
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < 49; ++j) { s[j] = __expf(s[j] - mx); se += s[j]; }
      const float inv = 1.f / se;
      float delta = 0.f;
      float dp[49];
#pragma unroll
      for (int j = 0; j < 49; ++j) {
        s[j] *= inv;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) a += sdo[t][d] * sv[j][d];
        dp[j] = a;
        delta += s[j] * a;
      }
      float dq[D];
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] = 0.f;
#pragma unroll
      for (int j = 0; j < 49; ++j) {
        const float ds = s[j] * (dp[j] - delta);
        dsum[j] += ds;
        sP[t][j] = s[j];
        sdS[t][j] = ds;
#pragma unroll
        for (int d = 0; d < D; ++d) dq[d] += ds * sk[j][d];
      }
      T* orow = reinterpret_cast<T*>(p.out + (tok0 + t) * p.out_ld * ES);
#pragma unroll
      for (int d = 0; d < D; ++d) Elem<T>::st(orow + h * D + d, dq[d] * p.scale);
    }
    __syncthreads();
    if (t < 49) {   // column t: dk_t = scale * sum_i dS[i][t] q_i ; dv_t = sum_i P[i][t] dO_i
      float dk[D], dv[D];
#pragma unroll
      for (int d = 0; d < D; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
      for (int i = 0; i < 49; ++i) {
        const float ds = sdS[i][t], pi = sP[i][t];
#pragma unroll
        for (int d = 0; d < D; ++d) { dk[d] += ds * sq[i][d]; dv[d] += pi * sdo[i][d]; }
      }
      T* orow = reinterpret_cast<T*>(p.out + (tok0 + t) * p.out_ld * ES);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        Elem<T>::st(orow + p.C + h * D + d, dk[d] * p.scale);
        Elem<T>::st(orow + 2 * p.C + h * D + d, dv[d]);
      }
    }
  }
  if (t < 49) {
    float* o = p.dtable_part + (((int64_t)blockIdx.x * p.heads + h) * 49 + t) * 49;
#pragma unroll
    for (int j = 0; j < 49; ++j) o[j] = dsum[j];
  }
}

// dtable[r][h] = sum over chunks and the (t, j) pairs whose relative index is r, in a fixed order
__global__ void attn_dtable_kernel(const float* __restrict__ part, int chunks, int heads, float* dtable) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= 169 * heads) return;
  const int r = id / heads, h = id % heads;
  const int di = r / 13 - 6, dj = r % 13 - 6;     // ti - ji, tj - jj
  double a = 0.0;
  for (int c = 0; c < chunks; ++c) {
    const float* pp = part + ((int64_t)c * heads + h) * 49 * 49;
    for (int ti = 0; ti < 7; ++ti) {
      const int ji = ti - di;
      if (ji < 0 || ji > 6) continue;
      for (int tj = 0; tj < 7; ++tj) {
        const int jj = tj - dj;
        if (jj < 0 || jj > 6) continue;
        a += (double)pp[(ti * 7 + tj) * 49 + ji * 7 + jj];
      }
    }
  }
  dtable[id] = (float)a;
}

// ------------------------------------------------------------------------------------------------ adaptive average pooling
__device__ __forceinline__ void ap_window(int i, int n_in, int n_out, int& lo, int& hi) {
  lo = (i * n_in) / n_out;
  hi = ((i + 1) * n_in + n_out - 1) / n_out;
}
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void adaptive_pool_kernel(const char* in, int64_t in_ld, int B, int H, int W, int C, int S, char* out,
                                                           int64_t out_ld) {
  // forward: in [B,H,W,C] -> out [B,S,S,C];  backward: in = gout [B,S,S,C] -> out = gin [B,H,W,C] (every input pixel gathers the
  // windows that contain it)
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int64_t total = (BWD ? (int64_t)B * H * W : (int64_t)B * S * S) * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t e = id / CC;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    if constexpr (!BWD) {
      const int j = (int)(e % S);
      const int64_t t = e / S;
      const int i = (int)(t % S);
      const int64_t b = t / S;
      int y0, y1, x0, x1;
      ap_window(i, H, S, y0, y1);
      ap_window(j, W, S, x0, x1);
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
          float f[V];
          Elem<T>::unpack(*reinterpret_cast<const uint4*>(in + (((b * H + y) * W + x) * in_ld) * ES + cc * 16), f);
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += f[k];
        }
      const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] *= inv;
    } else {
      const int x = (int)(e % W);
      const int64_t t = e / W;
      const int y = (int)(t % H);
      const int64_t b = t / H;
      for (int i = 0; i < S; ++i) {
        int y0, y1;
        ap_window(i, H, S, y0, y1);
        if (y < y0 || y >= y1) continue;
        for (int j = 0; j < S; ++j) {
          int x0, x1;
          ap_window(j, W, S, x0, x1);
          if (x < x0 || x >= x1) continue;
          float f[V];
          Elem<T>::unpack(*reinterpret_cast<const uint4*>(in + (((b * S + i) * S + j) * in_ld) * ES + cc * 16), f);
          const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += f[k] * inv;
        }
      }
    }
    *reinterpret_cast<uint4*>(out + e * out_ld * ES + cc * 16) = Elem<T>::pack(acc);
  }
}

// ------------------------------------------------------------------------------------------------ bilinear resize, any sizes
// F.interpolate(size=(Ho,Wo), mode='bilinear', align_corners=False): src = (o + 0.5) * in/out - 0.5, clamped at 0
__device__ __forceinline__ void rs_taps(int o, float scale, int n, int& i0, int& i1, float& w0, float& w1) {
  float src = ((float)o + 0.5f) * scale - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 > n - 1 ? n - 1 : i0;
  i1 = i0 + 1 < n ? i0 + 1 : n - 1;
  w1 = src - (float)i0;
  w0 = 1.f - w1;
}
__device__ __forceinline__ float rs_weight(int o, float scale, int n, int i) {
  int i0, i1;
  float w0, w1;
  rs_taps(o, scale, n, i0, i1, w0, w1);
  return (i0 == i ? w0 : 0.f) + (i1 == i ? w1 : 0.f);
}
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void resize_any_kernel(const char* a, int64_t a_ld, int B, int Hi, int Wi, int Ho, int Wo, int C, char* out,
                                                        int64_t out_ld, int accumulate) {
  // forward: a = small/any input [B,Hi,Wi,C] -> out [B,Ho,Wo,C] (accumulate: out += - the FPN top-down add);
  // backward: a = gradient of the output [B,Ho,Wo,C] -> out = gradient of the input [B,Hi,Wi,C] (gather over the outputs that read it)
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  const int64_t total = (BWD ? (int64_t)B * Hi * Wi : (int64_t)B * Ho * Wo) * CC;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    const int64_t e = id / CC;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    if constexpr (!BWD) {
      const int ox = (int)(e % Wo);
      const int64_t t = e / Wo;
      const int oy = (int)(t % Ho);
      const int64_t b = t / Ho;
      int y0, y1, x0, x1;
      float wy0, wy1, wx0, wx1;
      rs_taps(oy, sy, Hi, y0, y1, wy0, wy1);
      rs_taps(ox, sx, Wi, x0, x1, wx0, wx1);
      float p00[V], p01[V], p10[V], p11[V];
      const char* base = a + (b * Hi * Wi) * a_ld * ES + cc * 16;
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)y0 * Wi + x0) * a_ld * ES), p00);
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)y0 * Wi + x1) * a_ld * ES), p01);
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)y1 * Wi + x0) * a_ld * ES), p10);
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)y1 * Wi + x1) * a_ld * ES), p11);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = wy0 * (wx0 * p00[k] + wx1 * p01[k]) + wy1 * (wx0 * p10[k] + wx1 * p11[k]);
      if (accumulate) {
        float o[V];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(out + e * out_ld * ES + cc * 16), o);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += o[k];
      }
    } else {
      const int ix = (int)(e % Wi);
      const int64_t t = e / Wi;
      const int iy = (int)(t % Hi);
      const int64_t b = t / Hi;
      // outputs whose source lies within one input pixel of (iy, ix)
      int oy_lo = (int)floorf(((float)iy - 1.f + 0.5f) / sy - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.f + 0.5f) / sy - 0.5f) + 1;
      int ox_lo = (int)floorf(((float)ix - 1.f + 0.5f) / sx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.f + 0.5f) / sx - 0.5f) + 1;
      oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
      oy_hi = oy_hi > Ho - 1 ? Ho - 1 : oy_hi; ox_hi = ox_hi > Wo - 1 ? Wo - 1 : ox_hi;
      if (iy == 0) oy_lo = 0;             // clamped sources (src < 0) all read input 0
      if (ix == 0) ox_lo = 0;
      for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const float wy = rs_weight(oy, sy, Hi, iy);
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
          const float w = wy * rs_weight(ox, sx, Wi, ix);
          if (w == 0.f) continue;
          float f[V];
          Elem<T>::unpack(*reinterpret_cast<const uint4*>(a + (((b * Ho + oy) * Wo + ox) * a_ld) * ES + cc * 16), f);
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += w * f[k];
        }
      }
    }
    *reinterpret_cast<uint4*>(out + e * out_ld * ES + cc * 16) = Elem<T>::pack(acc);
  }
}

static int sw_view(const char* fn, const void* ptr, int64_t ld, int C, int es) {
  CVCS_CHECK_ARG(ptr != nullptr && ((uintptr_t)ptr % 16) == 0 && ld >= C && (ld * es) % 16 == 0, "%s: view must be 16-byte aligned with ld >= C", fn);
  return CVCS_OK;
}
#define SW_DT(dt) ((dt) == CVCS_F32 || (dt) == CVCS_BF16)

}  // namespace cvcs

using namespace cvcs;

extern "C" int cvcs_layernorm_rows(int64_t M) {
  int64_t r = cdiv(M, 64);
  return (int)(r < 1 ? 1 : (r > 1024 ? 1024 : r));
}

extern "C" int cvcs_layernorm_fwd(const void* x, int64_t x_ld, int64_t M, int C, const float* gamma, const float* beta, float eps, void* out,
                                  int64_t out_ld, float* mean, float* invstd, int dtype, void* stream) {
  const char* fn = "cvcs_layernorm_fwd";
  CVCS_CHECK_ARG(SW_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(M > 0 && C > 0 && C % V == 0 && C <= 1536, "%s: C=%d must be a multiple of %d, at most 1536", fn, C, V);
  CVCS_CHECK_ARG(gamma && beta && mean && invstd, "%s: null argument", fn);
  int rc;
  if ((rc = sw_view(fn, x, x_ld, C, es)) || (rc = sw_view(fn, out, out_ld, C, es))) return rc;
  const dim3 grid((unsigned)(cdiv(M, 4) > 16384 ? 16384 : cdiv(M, 4)));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((layernorm_fwd_kernel<float>), grid, dim3(256), 0, st, (const char*)x, x_ld, M, C, gamma, beta, eps, (char*)out, out_ld, mean, invstd);
  else hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const char*)x, x_ld, M, C, gamma, beta, eps, (char*)out, out_ld, mean, invstd);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_layernorm_bwd(const void* x, int64_t x_ld, const void* g, int64_t g_ld, int64_t M, int C, const float* gamma,
                                  const float* mean, const float* invstd, void* dx, int64_t dx_ld, float* part, int dtype, void* stream) {
  const char* fn = "cvcs_layernorm_bwd";
  CVCS_CHECK_ARG(SW_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(M > 0 && C > 0 && C % V == 0 && C <= 1536, "%s: bad C", fn);
  CVCS_CHECK_ARG(gamma && mean && invstd && part, "%s: null argument", fn);
  int rc;
  if ((rc = sw_view(fn, x, x_ld, C, es)) || (rc = sw_view(fn, g, g_ld, C, es)) || (rc = sw_view(fn, dx, dx_ld, C, es))) return rc;
  const dim3 grid((unsigned)cvcs_layernorm_rows(M));
  const size_t lds = (size_t)4 * 2 * C * 4;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((layernorm_bwd_kernel<float>), grid, dim3(256), lds, st, (const char*)x, x_ld, (const char*)g, g_ld, M, C, gamma, mean, invstd, (char*)dx, dx_ld, part);
  else hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t>), grid, dim3(256), lds, st, (const char*)x, x_ld, (const char*)g, g_ld, M, C, gamma, mean, invstd, (char*)dx, dx_ld, part);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_gelu(const void* x, int64_t x_ld, const void* g, int64_t g_ld, int64_t M, int C, void* out, int64_t out_ld, int dtype,
                         void* stream) {
  const char* fn = "cvcs_gelu";
  CVCS_CHECK_ARG(SW_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(M > 0 && C > 0 && C % (16 / es) == 0, "%s: bad shape", fn);
  int rc;
  if ((rc = sw_view(fn, x, x_ld, C, es)) || (rc = sw_view(fn, out, out_ld, C, es))) return rc;
  if (g && (rc = sw_view(fn, g, g_ld, C, es))) return rc;
  const dim3 grid(sw_grid(M * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
#define GELU_L(TT, BW) hipLaunchKernelGGL((gelu_kernel<TT, BW>), grid, dim3(256), 0, st, (const char*)x, x_ld, (const char*)g, g_ld, M, C, (char*)out, out_ld)
  if (dtype == CVCS_F32) { if (g) GELU_L(float, true); else GELU_L(float, false); }
  else { if (g) GELU_L(bf16_t, true); else GELU_L(bf16_t, false); }
#undef GELU_L
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_pack_patches(const void* src, int src_is_u8, int B, int H, int W, void* dst, int dtype, void* stream) {
  const char* fn = "cvcs_pack_patches";
  CVCS_CHECK_ARG(SW_DT(dtype) && src && dst && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0 && ((uintptr_t)dst % 16) == 0, "%s: bad arguments", fn);
  const dim3 grid(sw_grid((int64_t)B * (H / 4) * (W / 4) * 64));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) {
    if (src_is_u8) hipLaunchKernelGGL((pack_patches_kernel<float, uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)src, B, H, W, (float*)dst);
    else hipLaunchKernelGGL((pack_patches_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, B, H, W, (float*)dst);
  } else {
    if (src_is_u8) hipLaunchKernelGGL((pack_patches_kernel<bf16_t, uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)src, B, H, W, (bf16_t*)dst);
    else hipLaunchKernelGGL((pack_patches_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const float*)src, B, H, W, (bf16_t*)dst);
  }
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_patch_merge(const void* in, int64_t in_ld, int B, int H, int W, int C, void* out, int64_t out_ld, int reverse, int dtype,
                                void* stream) {
  const char* fn = "cvcs_patch_merge";
  CVCS_CHECK_ARG(SW_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % (16 / es) == 0, "%s: bad shape (H, W = the unmerged map, even)", fn);
  int rc;
  if ((rc = sw_view(fn, in, in_ld, reverse ? 4 * C : C, es)) || (rc = sw_view(fn, out, out_ld, reverse ? C : 4 * C, es))) return rc;
  const dim3 grid(sw_grid((int64_t)B * H * W * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
#define PM_L(TT, RV) hipLaunchKernelGGL((patch_merge_kernel<TT, RV>), grid, dim3(256), 0, st, (const char*)in, in_ld, B, H, W, C, (char*)out, out_ld)
  if (dtype == CVCS_F32) { if (reverse) PM_L(float, true); else PM_L(float, false); }
  else { if (reverse) PM_L(bf16_t, true); else PM_L(bf16_t, false); }
#undef PM_L
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_window_gather(const void* map, int64_t map_ld, int B, int H, int W, int C, int shift, void* win, int64_t win_ld, int dtype,
                                  void* stream) {
  const char* fn = "cvcs_window_gather";
  CVCS_CHECK_ARG(SW_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0 && shift >= 0 && shift < 7, "%s: bad shape", fn);
  int rc;
  if ((rc = sw_view(fn, map, map_ld, C, es)) || (rc = sw_view(fn, win, win_ld, C, es))) return rc;
  const int Hp = (H + 6) / 7 * 7, Wp = (W + 6) / 7 * 7;
  const dim3 grid(sw_grid((int64_t)B * Hp * Wp * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((window_kernel<float, false>), grid, dim3(256), 0, st, (const char*)map, map_ld, (const char*)nullptr, (int64_t)0, B, H, W, C, shift, (char*)win, win_ld);
  else hipLaunchKernelGGL((window_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const char*)map, map_ld, (const char*)nullptr, (int64_t)0, B, H, W, C, shift, (char*)win, win_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_window_reverse(const void* win, int64_t win_ld, const void* res, int64_t res_ld, int B, int H, int W, int C, int shift, void* out,
                                   int64_t out_ld, int dtype, void* stream) {
  const char* fn = "cvcs_window_reverse";
  CVCS_CHECK_ARG(SW_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0 && shift >= 0 && shift < 7, "%s: bad shape", fn);
  int rc;
  if ((rc = sw_view(fn, win, win_ld, C, es)) || (rc = sw_view(fn, out, out_ld, C, es))) return rc;
  if (res && (rc = sw_view(fn, res, res_ld, C, es))) return rc;
  const dim3 grid(sw_grid((int64_t)B * H * W * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((window_kernel<float, true>), grid, dim3(256), 0, st, (const char*)win, win_ld, (const char*)res, res_ld, B, H, W, C, shift, (char*)out, out_ld);
  else hipLaunchKernelGGL((window_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const char*)win, win_ld, (const char*)res, res_ld, B, H, W, C, shift, (char*)out, out_ld);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

static int attn_setup(const char* fn, AttnArgs& a, const void* qkv, int64_t qkv_ld, int B, int H, int W, int C, int heads, int shift, const float* table,
                      int dtype) {
  CVCS_CHECK_ARG(SW_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && heads > 0 && C == heads * 32 && shift >= 0 && shift < 7 && table, "%s: bad shape (head dim 32, window 7)", fn);
  int rc;
  if ((rc = sw_view(fn, qkv, qkv_ld, 3 * C, es))) return rc;
  a.qkv = (const char*)qkv; a.qkv_ld = qkv_ld; a.table = table;
  a.Hp = (H + 6) / 7 * 7; a.Wp = (W + 6) / 7 * 7; a.nwy = a.Hp / 7; a.nwx = a.Wp / 7;
  a.nwin = B * a.nwy * a.nwx; a.heads = heads; a.C = C; a.shift = shift;
  a.scale = 0.17677669529663687f;   // 32^-0.5
  a.dout = nullptr; a.dout_ld = 0; a.dtable_part = nullptr; a.wins_per_chunk = 1;
  return CVCS_OK;
}

extern "C" int cvcs_window_attention_fwd(const void* qkv, int64_t qkv_ld, int B, int H, int W, int C, int heads, int shift, const float* table,
                                         void* out, int64_t out_ld, int dtype, void* stream) {
  const char* fn = "cvcs_window_attention_fwd";
  AttnArgs a;
  int rc;
  if ((rc = attn_setup(fn, a, qkv, qkv_ld, B, H, W, C, heads, shift, table, dtype))) return rc;
  if ((rc = sw_view(fn, out, out_ld, C, dtype == CVCS_F32 ? 4 : 2))) return rc;
  a.out = (char*)out; a.out_ld = out_ld;
  const dim3 grid((unsigned)(a.nwin > 16384 ? 16384 : a.nwin), (unsigned)heads);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) hipLaunchKernelGGL((attn_fwd_kernel<float>), grid, dim3(64), 0, st, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), grid, dim3(64), 0, st, a);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

static int attn_chunking(int nwin, int& per) {   // windows per backward workgroup; chunks = ceil(nwin / per)
  per = nwin / 2048 + 1;
  if (per < 4) per = nwin >= 1024 ? 4 : 1;
  return (int)cdiv(nwin, per);
}

extern "C" int64_t cvcs_window_attention_bwd_workspace_floats(int B, int H, int W, int heads) {
  if (B <= 0 || H <= 0 || W <= 0 || heads <= 0) return CVCS_EINVAL;
  const int nwin = B * ((H + 6) / 7) * ((W + 6) / 7);
  int per;
  return (int64_t)attn_chunking(nwin, per) * heads * 49 * 49;
}

extern "C" int cvcs_window_attention_bwd(const void* qkv, int64_t qkv_ld, const void* dout, int64_t dout_ld, int B, int H, int W, int C, int heads,
                                         int shift, const float* table, void* dqkv, int64_t dqkv_ld, float* dtable, float* workspace, int dtype,
                                         void* stream) {
  const char* fn = "cvcs_window_attention_bwd";
  AttnArgs a;
  int rc;
  if ((rc = attn_setup(fn, a, qkv, qkv_ld, B, H, W, C, heads, shift, table, dtype))) return rc;
  const int es = dtype == CVCS_F32 ? 4 : 2;
  if ((rc = sw_view(fn, dout, dout_ld, C, es)) || (rc = sw_view(fn, dqkv, dqkv_ld, 3 * C, es))) return rc;
  CVCS_CHECK_ARG(dtable && workspace, "%s: null argument", fn);
  a.dout = (const char*)dout; a.dout_ld = dout_ld; a.out = (char*)dqkv; a.out_ld = dqkv_ld; a.dtable_part = workspace;
  const int chunks = attn_chunking(a.nwin, a.wins_per_chunk);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)chunks, (unsigned)heads);
  if (dtype == CVCS_F32) hipLaunchKernelGGL((attn_bwd_kernel<float>), grid, dim3(64), 0, st, a);
  else hipLaunchKernelGGL((attn_bwd_kernel<bf16_t>), grid, dim3(64), 0, st, a);
  hipLaunchKernelGGL(attn_dtable_kernel, dim3((unsigned)cdiv(169 * heads, 64)), dim3(64), 0, st, workspace, chunks, heads, dtable);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_adaptive_avg_pool(const void* in, int64_t in_ld, int B, int H, int W, int C, int S, void* out, int64_t out_ld, int backward,
                                      int dtype, void* stream) {
  const char* fn = "cvcs_adaptive_avg_pool";
  CVCS_CHECK_ARG(SW_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0 && S >= 1, "%s: bad shape", fn);
  int rc;
  if ((rc = sw_view(fn, in, in_ld, C, es)) || (rc = sw_view(fn, out, out_ld, C, es))) return rc;
  const dim3 grid(sw_grid((backward ? (int64_t)B * H * W : (int64_t)B * S * S) * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
#define AP_L(TT, BW) hipLaunchKernelGGL((adaptive_pool_kernel<TT, BW>), grid, dim3(256), 0, st, (const char*)in, in_ld, B, H, W, C, S, (char*)out, out_ld)
  if (dtype == CVCS_F32) { if (backward) AP_L(float, true); else AP_L(float, false); }
  else { if (backward) AP_L(bf16_t, true); else AP_L(bf16_t, false); }
#undef AP_L
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_resize_bilinear_any(const void* a, int64_t a_ld, int B, int Hi, int Wi, int Ho, int Wo, int C, void* out, int64_t out_ld,
                                        int backward, int accumulate, int dtype, void* stream) {
  const char* fn = "cvcs_resize_bilinear_any";
  CVCS_CHECK_ARG(SW_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % (16 / es) == 0 && !(backward && accumulate), "%s: bad shape", fn);
  int rc;
  if ((rc = sw_view(fn, a, a_ld, C, es)) || (rc = sw_view(fn, out, out_ld, C, es))) return rc;
  const dim3 grid(sw_grid((backward ? (int64_t)B * Hi * Wi : (int64_t)B * Ho * Wo) * (C / (16 / es))));
  hipStream_t st = (hipStream_t)stream;
#define RS_L(TT, BW) hipLaunchKernelGGL((resize_any_kernel<TT, BW>), grid, dim3(256), 0, st, (const char*)a, a_ld, B, Hi, Wi, Ho, Wo, C, (char*)out, out_ld, accumulate)
  if (dtype == CVCS_F32) { if (backward) RS_L(float, true); else RS_L(float, false); }
  else { if (backward) RS_L(bf16_t, true); else RS_L(bf16_t, false); }
#undef RS_L
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}
