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
// The f32 attention (parity path) runs on the VALU, one wave per (window, head); the bf16 attention on the matrix cores with the 49
// tokens padded to 64 (attn_fwd_mfma_kernel / attn_bwd_mfma_kernel below).
#include "common.h"

namespace cvcs {

static inline unsigned sw_grid(int64_t total, int cap = 256 * 32) {
  int64_t g = cdiv(total, 256);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// G lanes per token (G = 64: one wave per token, lane l holds the 16-byte chunks l, l+64, ... - C <= 2048: 4 chunks per lane in bf16, 8 in
// f32 (Swin-B merges 4 x 512 channels); G = 16 / 32: the narrow stages - C = 96 is 12 chunks in bf16 - put 4 / 2 tokens on a wave, one chunk per lane)
template <typename T> struct LnChunks { static constexpr int n = sizeof(T) == 4 ? 8 : 4; };
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T, int G>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const char* x, int64_t x_ld, int64_t M, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, char* out, int64_t out_ld,
                                                           float* mean_out, float* invstd_out) {
  constexpr int ES = sizeof(T), V = 16 / ES, NCH = G == 64 ? LnChunks<T>::n : 1, TPW = 64 / G;
  const int CC = C / V;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / G, sl = lane % G;
  for (int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * TPW; t0 < M; t0 += (int64_t)gridDim.x * 4 * TPW) {
    const int64_t tok = t0 + sub;
    const bool live = tok < M;
    float v[NCH][V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = sl + G * i;
      if (live && cc < CC) {
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + tok * x_ld * ES + cc * 16), v[i]);
#pragma unroll
        for (int k = 0; k < V; ++k) s += v[i][k];
      }
    }
    const float mean = group_sum<G>(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (live && sl + G * i < CC) {
#pragma unroll
        for (int k = 0; k < V; ++k) { const float d = v[i][k] - mean; q += d * d; }
      }
    const float invstd = 1.f / sqrtf(group_sum<G>(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = sl + G * i;
      if (live && cc < CC) {
#pragma unroll
        for (int k = 0; k < V; ++k) v[i][k] = (v[i][k] - mean) * invstd * gamma[cc * V + k] + beta[cc * V + k];
        *reinterpret_cast<uint4*>(out + tok * out_ld * ES + cc * 16) = Elem<T>::pack(v[i]);
      }
    }
    if (live && sl == 0) { mean_out[tok] = mean; invstd_out[tok] = invstd; }
  }
}

// backward: dx = invstd * (gamma*g - mean_c(gamma*g) - xhat * mean_c(gamma*g*xhat)); partial rows part[block][0|1][C] of
// dgamma = sum_tokens g*xhat and dbeta = sum_tokens g (reduced by cvcs_colsum_finalize over 2C columns)
template <typename T, int G>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const char* x, int64_t x_ld, const char* g, int64_t g_ld, int64_t M, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, char* dx, int64_t dx_ld, float* part) {
  constexpr int ES = sizeof(T), V = 16 / ES, NCH = G == 64 ? LnChunks<T>::n : 1, TPW = 64 / G;
  extern __shared__ float sred[];      // [4 waves * TPW token slots][2][C]
  const int CC = C / V;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / G, sl = lane % G;
  float dg[NCH][V], db[NCH][V];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int k = 0; k < V; ++k) { dg[i][k] = 0.f; db[i][k] = 0.f; }
  for (int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * TPW; t0 < M; t0 += (int64_t)gridDim.x * 4 * TPW) {
    const int64_t tok = t0 + sub;
    const bool live = tok < M;
    const float mu = live ? mean[tok] : 0.f, is = live ? invstd[tok] : 0.f;
    float xh[NCH][V], gg[NCH][V];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = sl + G * i;
      if (live && cc < CC) {
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
    s1 = group_sum<G>(s1) / (float)C;
    s2 = group_sum<G>(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = sl + G * i;
      if (live && cc < CC) {
        float o[V];
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = is * (gg[i][k] - s1 - xh[i][k] * s2);
        *reinterpret_cast<uint4*>(dx + tok * dx_ld * ES + cc * 16) = Elem<T>::pack(o);
      }
    }
  }
  const int slot = wave * TPW + sub;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int cc = sl + G * i;
    if (cc < CC)
#pragma unroll
      for (int k = 0; k < V; ++k) {
        sred[(slot * 2 + 0) * C + cc * V + k] = dg[i][k];
        sred[(slot * 2 + 1) * C + cc * V + k] = db[i][k];
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    float a = 0.f;
#pragma unroll
    for (int sidx = 0; sidx < 4 * TPW; ++sidx) a += sred[(sidx * 2 + which) * C + c];
    part[(int64_t)blockIdx.x * 2 * C + i] = a;
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

// ------------------------------------------------------------------------------------------------ window attention on the matrix cores (bf16)
// One wave per (window, head); the 49 tokens are padded to 64 (keys j >= 49 get a bias of -1e30, queries i >= 49 are discarded).
// Everything is computed TRANSPOSED so that no accumulator ever has to change lanes (cdna_hip_programming.md, "an accumulator tile
// as the next MFMA's operand"):
//   S^T[j][i] = sum_d K[j][d] Q[i][d]      A = K rows, B = Q rows: both fragments are 16-byte loads of a token row (no LDS);
//                                          the accumulators START at bias[i][j] / scale (an LDS image in accumulator layout)
//   softmax over j for a fixed i           i is the lane's column: 16 values in registers, two cross-lane steps
//   O^T[d][i] = sum_j V[j][d] P[i][j]      B = P^T straight from the S^T accumulators (lane group g holds j = 4g..4g+3 and
//                                          16+4g..16+4g+3 of each 32-key step); A = V^T by ds_read_b64_tr_b16 from a row-major V
//                                          tile, whose 4-row blocks are exactly that k permutation
//   the O^T accumulators hold 4 consecutive d of one token: 8-byte stores.
// LDS V tile: [64 rows][64 B], the two 32-byte halves of a row swapped on rows with bit 2 set (lane groups g, g+1 of a 32-lane
// half then read different banks); rows 49..63 are zeroed once.
constexpr int kAttnWaves = 8;      // forward: waves per workgroup
constexpr int kAttnBwdWaves = 4;   // backward

__device__ __forceinline__ unsigned attn_tile_off(int row, int half) { return (unsigned)(row * 64 + ((half ^ ((row >> 2) & 1)) << 5)); }

typedef __attribute__((ext_vector_type(4))) short at_s16x4;
typedef __attribute__((ext_vector_type(8))) short at_s16x8;
// A-operand fragment of X^T (X row-major [64][32] bf16 in LDS): lane (row d = 16*dt + l%16, group g), k = X rows 32*ks + {4g..4g+3, 16+4g..}
__device__ __forceinline__ bf16x8 attn_tr_frag(const char* tile, int dt, int ks, int l) {
  const int g = l >> 4, q = (l >> 2) & 3, pp = l & 3;
  const int r0 = 32 * ks + 4 * g + q;
  const at_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(tile + attn_tile_off(r0, dt) + 8 * pp));
  const at_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(tile + attn_tile_off(r0 + 16, dt) + 8 * pp));
  return __builtin_bit_cast(bf16x8, (at_s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ f32x4 attn_mfma(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 attn_mfma(const bf16x8& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// two accumulator tiles (k rows 4g..4g+3 and 16+4g..) -> one 8-element bf16 operand fragment
__device__ __forceinline__ uint4 attn_pack(const f32x4& a, const f32x4& b) {
  return make_uint4(pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]), pack2_bf16(b[0], b[1]), pack2_bf16(b[2], b[3]));
}
__device__ __forceinline__ float attn_red_max(float v) {   // over the four lane groups (lanes l, l^16, l^32, l^48)
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float attn_red_sum(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}
// bias image in accumulator layout: img[(a*4 + b)*64 + lane][r]; transposed = false: the tile (a, b) = (key tile, query tile) of S^T;
// true: (query tile, key tile) of S.  Keys >= 49: -1e30; queries >= 49: 0.
__device__ __forceinline__ void attn_bias_image(float* img, const float* __restrict__ table, int heads, int h, float inv_scale, bool transposed,
                                                int tid, int nthreads) {
  for (int e = tid; e < 16 * 64 * 4; e += nthreads) {
    const int r = e & 3, ln = (e >> 2) & 63, t = e >> 8, a = t >> 2, b = t & 3;
    const int row = 16 * a + 4 * (ln >> 4) + r, col = 16 * b + (ln & 15);
    const int i = transposed ? row : col, j = transposed ? col : row;
    float v;
    if (j >= 49) v = -1e30f;
    else if (i >= 49) v = 0.f;
    else v = table[((i / 7 - j / 7 + 6) * 13 + (i % 7 - j % 7 + 6)) * heads + h] * inv_scale;
    img[e] = v;
  }
}
// region ids (shifted windows) of the tokens this lane meets: lane-indexed tokens 16x + c (x = 0..3) and row-indexed tokens
// 16x + 4g + r (x = 0..3, r = 0..3); tokens >= 49 get region 15 (never equal matters not: their scores are masked or discarded)
struct AttnRegions { int lane_tok[4]; int row_tok[4][4]; };
__device__ __forceinline__ int attn_tok_region(const AttnArgs& p, int wy, int wx, int t) {
  return t >= 49 ? 15 : attn_region(wy * 7 + t / 7, wx * 7 + t % 7, p.Hp, p.Wp, p.shift);
}

__global__ __launch_bounds__(kAttnWaves * 64) void attn_fwd_mfma_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) float sbias[16 * 64 * 4];
  __shared__ __attribute__((aligned(16))) char sv_all[kAttnWaves][64 * 64];
  const int h = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, l = tid & 63, g = l >> 4, c = l & 15;
  attn_bias_image(sbias, p.table, p.heads, h, 1.f / p.scale, false, tid, kAttnWaves * 64);
  char* sv = sv_all[wv];
  if (l < 60) *reinterpret_cast<uint4*>(sv + 49 * 64 + l * 16) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const int64_t ldb = p.qkv_ld * 2;
  for (int w = blockIdx.x * kAttnWaves + wv; w < p.nwin; w += gridDim.x * kAttnWaves) {
    const char* base = p.qkv + (int64_t)w * 49 * ldb + h * 64;
    uint4 qf[4], kf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = min(16 * t + c, 48);
      qf[t] = *reinterpret_cast<const uint4*>(base + row * ldb + g * 16);
      kf[t] = *reinterpret_cast<const uint4*>(base + p.C * 2 + row * ldb + g * 16);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = l + 64 * k;
      if (e < 196) {
        const int row = e >> 2, ch = e & 3;
        const uint4 v = *reinterpret_cast<const uint4*>(base + 2 * p.C * 2 + row * ldb + ch * 16);
        *reinterpret_cast<uint4*>(sv + attn_tile_off(row, ch >> 1) + ((ch & 1) << 4)) = v;
      }
    }
    f32x4 acc[4][4];     // [key tile mt][query tile nt]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = attn_mfma(kf[mt], qf[nt], *reinterpret_cast<const f32x4*>(&sbias[((mt * 4 + nt) * 64 + l) * 4]));
    const int wi = w % (p.nwy * p.nwx), wy = wi / p.nwx, wx = wi % p.nwx;
    const bool masked = p.shift != 0 && (wy == p.nwy - 1 || wx == p.nwx - 1);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] *= p.scale;
    if (masked) {
      int rl[4], rr[4][4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        rl[x] = attn_tok_region(p, wy, wx, 16 * x + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) rr[x][r] = attn_tok_region(p, wy, wx, 16 * x + 4 * g + r);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][nt][r] += rr[mt][r] != rl[nt] ? -100.f : 0.f;
    }
    uint4 pf[4][2];      // P^T fragments [query tile][key step]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float mx = -INFINITY;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[mt][nt][r]);
      mx = attn_red_max(mx);
      float se = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[mt][nt][r] = __expf(acc[mt][nt][r] - mx); se += acc[mt][nt][r]; }
      const float inv = 1.f / attn_red_sum(se);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] *= inv;
      pf[nt][0] = attn_pack(acc[0][nt], acc[1][nt]);
      pf[nt][1] = attn_pack(acc[2][nt], acc[3][nt]);
    }
    // (the V tile was written by this wave's own ds_write above: LDS operations of one wave complete in order)
    bf16x8 vf[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) vf[dt][ks] = attn_tr_frag(sv, dt, ks, l);
    char* obase = p.out + (int64_t)w * 49 * p.out_ld * 2 + h * 64;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int i = 16 * nt + c;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        o = attn_mfma(vf[dt][0], pf[nt][0], o);
        o = attn_mfma(vf[dt][1], pf[nt][1], o);
        if (i < 49) *reinterpret_cast<uint2*>(obase + (int64_t)i * p.out_ld * 2 + (16 * dt + 4 * g) * 2) = make_uint2(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]));
      }
    }
  }
}

// Backward.  Per (window, head), with P recomputed:
//   T pass (per 16-query column block): S^T, dP^T[j][i] = sum_d V[j][d] dO[i][d]  ->  softmax statistics and delta_i = sum_j P dP
//     (kept in LDS for the second pass), dS^T = P^T (dP^T - delta)  ->  dQ^T[d][i] = sum_j K[j][d] dS^T[j][i]  (A = K^T by transposed read)
//   N pass (per 16-key column block): S[i][j], dP[i][j] in the other orientation (the same row fragments with A and B swapped), P and dS
//     from the stored statistics  ->  dV^T[d][j] = sum_i dO[i][d] P[i][j],  dK^T[d][j] = sum_i Q[i][d] dS[i][j]  (A = dO^T, Q^T by
//     transposed reads); dS is also accumulated over the wave's windows for the gradient of the bias table.
// LDS per wave: Q, K, dO tiles ([64][64 B], as the forward's V tile) + 3 x 64 statistics; per workgroup the two bias images.
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_mfma_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  float* sbiasT = reinterpret_cast<float*>(at_smem);
  float* sbiasN = sbiasT + 16 * 64 * 4;
  const int h = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, l = tid & 63, g = l >> 4, c = l & 15;
  char* wbase = at_smem + 2 * 16384 + wv * (3 * 4096 + 768);
  char* sq = wbase;
  char* sk = wbase + 4096;
  char* sdo = wbase + 8192;
  float* stat = reinterpret_cast<float*>(wbase + 12288);     // max[64] | 1/sum[64] | delta[64]
  attn_bias_image(sbiasT, p.table, p.heads, h, 1.f / p.scale, false, tid, NW * 64);
  attn_bias_image(sbiasN, p.table, p.heads, h, 1.f / p.scale, true, tid, NW * 64);
  if (l < 60) {
    *reinterpret_cast<uint4*>(sq + 49 * 64 + l * 16) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(sk + 49 * 64 + l * 16) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(sdo + 49 * 64 + l * 16) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  f32x4 dsum[4][4];    // [query tile it][key tile jt], summed over this wave's windows
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) dsum[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t ldb = p.qkv_ld * 2, ldd = p.dout_ld * 2, ldo = p.out_ld * 2;
  for (int w = blockIdx.x * NW + wv; w < p.nwin; w += gridDim.x * NW) {
    const char* base = p.qkv + (int64_t)w * 49 * ldb + h * 64;
    const char* dbase = p.dout + (int64_t)w * 49 * ldd + h * 64;
    char* obase = p.out + (int64_t)w * 49 * ldo + h * 64;
    uint4 qf[4], kf[4], vf[4], dof[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = min(16 * t + c, 48);
      qf[t] = *reinterpret_cast<const uint4*>(base + row * ldb + g * 16);
      kf[t] = *reinterpret_cast<const uint4*>(base + p.C * 2 + row * ldb + g * 16);
      vf[t] = *reinterpret_cast<const uint4*>(base + 2 * p.C * 2 + row * ldb + g * 16);
      dof[t] = *reinterpret_cast<const uint4*>(dbase + row * ldd + g * 16);
      if (16 * t + c < 49) {   // the same fragments fill the tiles the transposed reads take (chunk g of row 16t + c)
        const unsigned off = attn_tile_off(16 * t + c, g >> 1) + ((g & 1) << 4);
        *reinterpret_cast<uint4*>(sq + off) = qf[t];
        *reinterpret_cast<uint4*>(sk + off) = kf[t];
        *reinterpret_cast<uint4*>(sdo + off) = dof[t];
      }
    }
    const int wi = w % (p.nwy * p.nwx), wy = wi / p.nwx, wx = wi % p.nwx;
    const bool masked = p.shift != 0 && (wy == p.nwy - 1 || wx == p.nwx - 1);
    unsigned rl = 0, rr[4] = {0, 0, 0, 0};     // 4-bit region ids: rl nibble x = token 16x + c; rr[x] nibble r = token 16x + 4g + r
    if (masked) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        rl |= (unsigned)attn_tok_region(p, wy, wx, 16 * x + c) << (4 * x);
#pragma unroll
        for (int r = 0; r < 4; ++r) rr[x] |= (unsigned)attn_tok_region(p, wy, wx, 16 * x + 4 * g + r) << (4 * r);
      }
    }
    // ---- T pass: queries on the lanes
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 s[4], dp[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        s[mt] = attn_mfma(kf[mt], qf[nt], *reinterpret_cast<const f32x4*>(&sbiasT[((mt * 4 + nt) * 64 + l) * 4])) * p.scale;
        dp[mt] = attn_mfma(vf[mt], dof[nt], (f32x4){0.f, 0.f, 0.f, 0.f});
      }
      if (masked) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[mt][r] += ((rr[mt] >> (4 * r)) & 15u) != ((rl >> (4 * nt)) & 15u) ? -100.f : 0.f;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[mt][r]);
      mx = attn_red_max(mx);
      float se = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[mt][r] = __expf(s[mt][r] - mx); se += s[mt][r]; }
      const float inv = 1.f / attn_red_sum(se);
      float delta = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[mt][r] *= inv; delta += s[mt][r] * dp[mt][r]; }
      delta = attn_red_sum(delta);
      if (g == 0) { stat[16 * nt + c] = mx; stat[64 + 16 * nt + c] = inv; stat[128 + 16 * nt + c] = delta; }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[mt][r] *= dp[mt][r] - delta;      // dS^T
      const uint4 ds0 = attn_pack(s[0], s[1]), ds1 = attn_pack(s[2], s[3]);
      const int i = 16 * nt + c;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        o = attn_mfma(attn_tr_frag(sk, dt, 0, l), ds0, o);
        o = attn_mfma(attn_tr_frag(sk, dt, 1, l), ds1, o);
        o *= p.scale;
        if (i < 49) *reinterpret_cast<uint2*>(obase + (int64_t)i * ldo + (16 * dt + 4 * g) * 2) = make_uint2(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]));
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the column blocks apart: interleaved, their live tiles exceed the 256 registers of 2 waves / SIMD
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's statistics are in LDS (same-wave ordering; compiler fence)
    // ---- N pass: keys on the lanes
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      f32x4 s[4], dp[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        s[it] = attn_mfma(qf[it], kf[jt], *reinterpret_cast<const f32x4*>(&sbiasN[((it * 4 + jt) * 64 + l) * 4])) * p.scale;
        dp[it] = attn_mfma(dof[it], vf[jt], (f32x4){0.f, 0.f, 0.f, 0.f});
      }
      if (masked) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[it][r] += ((rr[it] >> (4 * r)) & 15u) != ((rl >> (4 * jt)) & 15u) ? -100.f : 0.f;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const f32x4 rmx = *reinterpret_cast<const f32x4*>(&stat[16 * it + 4 * g]);
        const f32x4 rinv = *reinterpret_cast<const f32x4*>(&stat[64 + 16 * it + 4 * g]);
        const f32x4 rdel = *reinterpret_cast<const f32x4*>(&stat[128 + 16 * it + 4 * g]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool valid = 16 * it + 4 * g + r < 49;
          const float pv = valid ? __expf(s[it][r] - rmx[r]) * rinv[r] : 0.f;
          const float dsv = pv * (dp[it][r] - rdel[r]);
          s[it][r] = pv;
          dp[it][r] = valid ? dsv : 0.f;
          dsum[it][jt][r] += dp[it][r];
        }
      }
      const uint4 p0 = attn_pack(s[0], s[1]), p1 = attn_pack(s[2], s[3]);
      const uint4 d0 = attn_pack(dp[0], dp[1]), d1 = attn_pack(dp[2], dp[3]);
      const int j = 16 * jt + c;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4 ov = {0.f, 0.f, 0.f, 0.f}, ok = {0.f, 0.f, 0.f, 0.f};
        ov = attn_mfma(attn_tr_frag(sdo, dt, 0, l), p0, ov);
        ov = attn_mfma(attn_tr_frag(sdo, dt, 1, l), p1, ov);
        ok = attn_mfma(attn_tr_frag(sq, dt, 0, l), d0, ok);
        ok = attn_mfma(attn_tr_frag(sq, dt, 1, l), d1, ok);
        ok *= p.scale;
        if (j < 49) {
          char* row = obase + (int64_t)j * ldo + (16 * dt + 4 * g) * 2;
          *reinterpret_cast<uint2*>(row + p.C * 2) = make_uint2(pack2_bf16(ok[0], ok[1]), pack2_bf16(ok[2], ok[3]));
          *reinterpret_cast<uint2*>(row + 2 * p.C * 2) = make_uint2(pack2_bf16(ov[0], ov[1]), pack2_bf16(ov[2], ov[3]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tiles and statistics are rewritten by the next window
  }
  // per-wave partial of the bias-table gradient: part[chunk = workgroup * waves + wave][head][i * 49 + j]
  float* o = p.dtable_part + ((int64_t)(blockIdx.x * NW + wv) * p.heads + h) * 49 * 49;
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * it + 4 * g + r, j = 16 * jt + c;
        if (i < 49 && j < 49) o[i * 49 + j] = dsum[it][jt][r];
      }
}

// sum[h][t*49 + j] = sum over chunks of part[chunk][h][t*49 + j], chunk 0 first (coalesced over the entries; double accumulation)
__global__ __launch_bounds__(256) void attn_dtable_sum_kernel(const float* __restrict__ part, int chunks, int heads, float* __restrict__ sum) {
  const int id = blockIdx.x * 256 + threadIdx.x, n = heads * 49 * 49;
  if (id >= n) return;
  double a = 0.0;
#pragma unroll 8
  for (int c = 0; c < chunks; ++c) a += (double)part[(int64_t)c * n + id];
  sum[id] = (float)a;
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
  CVCS_CHECK_ARG(M > 0 && C > 0 && C % V == 0 && C <= 2048, "%s: C=%d must be a multiple of %d, at most 2048", fn, C, V);
  CVCS_CHECK_ARG(gamma && beta && mean && invstd, "%s: null argument", fn);
  int rc;
  if ((rc = sw_view(fn, x, x_ld, C, es)) || (rc = sw_view(fn, out, out_ld, C, es))) return rc;
  const dim3 grid((unsigned)(cdiv(M, 16) > 16384 ? 16384 : cdiv(M, 16)));
  hipStream_t st = (hipStream_t)stream;
  const int CC = C / (16 / es), G = CC <= 16 ? 16 : (CC <= 32 ? 32 : 64);     // lanes per token
#define LN_F(TT, GG) hipLaunchKernelGGL((layernorm_fwd_kernel<TT, GG>), grid, dim3(256), 0, st, (const char*)x, x_ld, M, C, gamma, beta, eps, (char*)out, out_ld, mean, invstd)
  if (dtype == CVCS_F32) { if (G == 16) LN_F(float, 16); else if (G == 32) LN_F(float, 32); else LN_F(float, 64); }
  else { if (G == 16) LN_F(bf16_t, 16); else if (G == 32) LN_F(bf16_t, 32); else LN_F(bf16_t, 64); }
#undef LN_F
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_layernorm_bwd(const void* x, int64_t x_ld, const void* g, int64_t g_ld, int64_t M, int C, const float* gamma,
                                  const float* mean, const float* invstd, void* dx, int64_t dx_ld, float* part, int dtype, void* stream) {
  const char* fn = "cvcs_layernorm_bwd";
  CVCS_CHECK_ARG(SW_DT(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2, V = 16 / es;
  CVCS_CHECK_ARG(M > 0 && C > 0 && C % V == 0 && C <= 2048, "%s: bad C", fn);
  CVCS_CHECK_ARG(gamma && mean && invstd && part, "%s: null argument", fn);
  int rc;
  if ((rc = sw_view(fn, x, x_ld, C, es)) || (rc = sw_view(fn, g, g_ld, C, es)) || (rc = sw_view(fn, dx, dx_ld, C, es))) return rc;
  const dim3 grid((unsigned)cvcs_layernorm_rows(M));
  const int CC = C / V, G = CC <= 16 ? 16 : (CC <= 32 ? 32 : 64);
  const size_t lds = (size_t)4 * (64 / G) * 2 * C * 4;
  hipStream_t st = (hipStream_t)stream;
#define LN_B(TT, GG) hipLaunchKernelGGL((layernorm_bwd_kernel<TT, GG>), grid, dim3(256), lds, st, (const char*)x, x_ld, (const char*)g, g_ld, M, C, gamma, mean, invstd, (char*)dx, dx_ld, part)
  if (dtype == CVCS_F32) { if (G == 16) LN_B(float, 16); else if (G == 32) LN_B(float, 32); else LN_B(float, 64); }
  else { if (G == 16) LN_B(bf16_t, 16); else if (G == 32) LN_B(bf16_t, 32); else LN_B(bf16_t, 64); }
#undef LN_B
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

// workgroups along x of the MFMA kernels: every wave of a workgroup walks windows of one head; about `per_cu` workgroups per CU in all
static int attn_mfma_groups(int nwin, int heads, int per_cu, int waves) {
  int g = (256 * per_cu) / heads;
  if (g < 1) g = 1;
  const int need = (int)cdiv(nwin, waves);
  return need < g ? need : g;
}

extern "C" int cvcs_window_attention_fwd(const void* qkv, int64_t qkv_ld, int B, int H, int W, int C, int heads, int shift, const float* table,
                                         void* out, int64_t out_ld, int dtype, void* stream) {
  const char* fn = "cvcs_window_attention_fwd";
  AttnArgs a;
  int rc;
  if ((rc = attn_setup(fn, a, qkv, qkv_ld, B, H, W, C, heads, shift, table, dtype))) return rc;
  if ((rc = sw_view(fn, out, out_ld, C, dtype == CVCS_F32 ? 4 : 2))) return rc;
  a.out = (char*)out; a.out_ld = out_ld;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) {
    const dim3 grid((unsigned)(a.nwin > 16384 ? 16384 : a.nwin), (unsigned)heads);
    hipLaunchKernelGGL((attn_fwd_kernel<float>), grid, dim3(64), 0, st, a);
  } else {
    const dim3 grid((unsigned)attn_mfma_groups(a.nwin, heads, 4, kAttnWaves), (unsigned)heads);
    hipLaunchKernelGGL(attn_fwd_mfma_kernel, grid, dim3(kAttnWaves * 64), 0, st, a);
  }
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
  const int valu = attn_chunking(nwin, per), mfma = attn_mfma_groups(nwin, heads, 1, kAttnBwdWaves) * kAttnBwdWaves;   // (one size for both dtypes)
  return (int64_t)((valu > mfma ? valu : mfma) + 1) * heads * 49 * 49;   // (+1: the slab of the summed partials)
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
  hipStream_t st = (hipStream_t)stream;
  int chunks;
  if (dtype == CVCS_F32) {
    chunks = attn_chunking(a.nwin, a.wins_per_chunk);
    hipLaunchKernelGGL((attn_bwd_kernel<float>), dim3((unsigned)chunks, (unsigned)heads), dim3(64), 0, st, a);
  } else {
    // one workgroup of 4 waves per CU (84 KB of LDS, one wave per SIMD with up to 512 registers): with 8 waves the 256-register
    // limit of two waves per SIMD spills 79 registers of the bias-gradient accumulators and the launch measured 8 % slower
    const int groups = attn_mfma_groups(a.nwin, heads, 1, kAttnBwdWaves);
    chunks = groups * kAttnBwdWaves;
    constexpr int lds = 2 * 16384 + kAttnBwdWaves * (3 * 4096 + 768);
    (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<kAttnBwdWaves>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(attn_bwd_mfma_kernel<kAttnBwdWaves>, dim3((unsigned)groups, (unsigned)heads), dim3(kAttnBwdWaves * 64), lds, st, a);
  }
  // partials -> one [heads][49*49] slab behind them -> the 169 x heads table entries
  float* slab = workspace + (int64_t)chunks * heads * 49 * 49;
  hipLaunchKernelGGL(attn_dtable_sum_kernel, dim3((unsigned)cdiv(heads * 49 * 49, 256)), dim3(256), 0, st, workspace, chunks, heads, slab);
  hipLaunchKernelGGL(attn_dtable_kernel, dim3((unsigned)cdiv(169 * heads, 64)), dim3(64), 0, st, slab, 1, heads, dtable);
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
