// HBM-bound kernels of the per-tile path on gfx950: BatchNorm finalize / apply (+ReLU, +MaxPool 2x2) and its
// two-pass backward, bilinear x2, and the NCHW<->packed layout changes at the boundary.
//
// Every activation access is one 16-byte chunk per lane (8 bf16 / 4 f32 channels of one pixel), lanes of a
// wave walking consecutive chunks: whole 128-byte lines per pixel row.  Per-channel reductions keep a channel
// chunk fixed per thread, accumulate in registers over a grid-stride of pixels, combine the workgroup's pixel
// lanes through LDS and write one partial row per workgroup; a tiny f64 finalize sums the rows in a fixed
// order (bitwise reproducible, no float atomics).
//
// Reference ops replaced: nn.BatchNorm2d (S/blocks.py:14,42,45), nn.ReLU (S/blocks.py:17,41,44), nn.MaxPool2d(2,2)
// (S/nets.py:130,135,140,145), nn.Upsample(x2, bilinear) (S/blocks.py:29), `.type(torch.float32)` (S/train.py:121)
// and their autograd backward (S/train.py:125).
#include "common.h"

namespace cvcs {

// ------------------------------------------------------------------------------------------------ BN finalize
// Chan et al. combination of per-block (count, sum, centred M2) -> mean / biased var; f64 throughout.
// Workgroup = 16 row-lanes x 16 channels: each thread merges a strided subset of the partial rows, the 16 lanes of a
// channel are merged through LDS.  With more than kDirectRows partial rows a first stage (one workgroup per
// (row block, 16 channels)) shrinks them to <= kDirectRows rows of larger pixel count.
struct Moments { double n, mean, m2; };
__device__ __forceinline__ void merge(Moments& a, double nb, double mean_b, double m2_b) {
  if (nb <= 0.0) return;
  const double n = a.n + nb;
  const double d = mean_b - a.mean;
  a.mean += d * (nb / n);
  a.m2 += m2_b + d * d * (a.n * nb / n);
  a.n = n;
}

constexpr int kDirectRows = 2048;

// merge partial rows r0, r0+rstep, .. < r1 (row r covers scnt[r] pixels) for channel c.
// Rows of EQUAL pixel count (every full tile; the common case) need no per-row division: with S = sum of the row sums, Q = sum of their
// squares and M = sum of the row M2s, the merged moments are mean = S / (k n), M2 = M + Q / n - S^2 / (k n) - three f64 FMAs per row instead of
// two f64 divisions (the serial Chan merge took 15 us per BatchNorm: 63 of them per ResNet50-UNet step).  In f64 the cancellation costs
// ~1e-16 (mean / sigma)^2 of relative accuracy.  A lane that meets rows of different counts (ragged tiles) merges them pairwise as before.
__device__ __forceinline__ Moments merge_rows(const float* __restrict__ ssum, const float* __restrict__ sm2,
                                              const float* __restrict__ scnt, int r0, int r1, int rstep, int C, int c) {
  double S = 0.0, Q = 0.0, Mm = 0.0, n0 = -1.0;
  int k = 0;
  bool uniform = true;
  auto take = [&](float cn, float sv, float mv) {
    const double n = (double)cn;
    if (n <= 0.0) return;
    if (n0 < 0.0) n0 = n;
    uniform = uniform && n == n0;
    const double s_ = (double)sv;
    S += s_;
    Q += s_ * s_;
    Mm += (double)mv;
    ++k;
  };
  int r = r0;
  // sixteen, then four rows' loads in flight: these kernels are chains of dependent memory round trips (~1.5 us each), not arithmetic
  for (; r + 15 * rstep < r1; r += 16 * rstep) {
    float cn[16], sv[16], mv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      cn[u] = scnt[r + u * rstep];
      sv[u] = ssum[(int64_t)(r + u * rstep) * C + c];
      mv[u] = sm2[(int64_t)(r + u * rstep) * C + c];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) take(cn[u], sv[u], mv[u]);
  }
  for (; r + 3 * rstep < r1; r += 4 * rstep) {
    float cn[4], sv[4], mv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      cn[u] = scnt[r + u * rstep];
      sv[u] = ssum[(int64_t)(r + u * rstep) * C + c];
      mv[u] = sm2[(int64_t)(r + u * rstep) * C + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) take(cn[u], sv[u], mv[u]);
  }
  for (; r < r1; r += rstep) take(scnt[r], ssum[(int64_t)r * C + c], sm2[(int64_t)r * C + c]);
  if (k == 0) return Moments{0.0, 0.0, 0.0};
  if (uniform) {
    const double N = n0 * (double)k;
    double m2 = Mm + (Q / n0 - S * S / N);
    if (m2 < 0.0) m2 = 0.0;
    return Moments{N, S / N, m2};
  }
  Moments a{0.0, 0.0, 0.0};
  for (int r = r0; r < r1; r += rstep) {
    const double n = (double)scnt[r];
    if (n <= 0.0) continue;
    merge(a, n, (double)ssum[(int64_t)r * C + c] / n, (double)sm2[(int64_t)r * C + c]);
  }
  return a;
}

// finalize workgroups are 1024 threads = kRL row-lanes x 16 channels (these kernels are latency-bound serial f64 loops
// over the partial rows: 64 row-lanes cut the loop 4x against 16); lanes are merged in two levels, in a fixed order
constexpr int kRL = 64;
__device__ __forceinline__ Moments merge_lanes(Moments a, int rl, int cl) {
  // The 64 lane results of a channel without a chain of pairwise merges (two f64 divisions each: 16 dependent merges were a third of the
  // kernel): N = sum n, mean = sum(n mean) / N - ONE division - then M2 = sum(m2 + n (mean_q - mean)^2), the standard parallel form (no
  // cancellation).  Eight lanes sum eight entries each, in a fixed order.
  __shared__ double sh[3][kRL][17];
  __shared__ double part[3][8][17];
  sh[0][rl][cl] = a.n; sh[1][rl][cl] = a.mean; sh[2][rl][cl] = a.m2;
  __syncthreads();
  if (rl < 8) {
    double n = 0.0, s1 = 0.0;
    for (int q = rl; q < kRL; q += 8) { n += sh[0][q][cl]; s1 += sh[0][q][cl] * sh[1][q][cl]; }
    part[0][rl][cl] = n; part[1][rl][cl] = s1;
  }
  __syncthreads();
  double N = 0.0, S = 0.0;
  if (rl < 8) {
    for (int q = 0; q < 8; ++q) { N += part[0][q][cl]; S += part[1][q][cl]; }
    const double mean = N > 0.0 ? S / N : 0.0;
    double m2 = 0.0;
    for (int q = rl; q < kRL; q += 8) {
      const double d = sh[1][q][cl] - mean;
      m2 += sh[2][q][cl] + sh[0][q][cl] * d * d;
    }
    part[2][rl][cl] = m2;
  }
  __syncthreads();
  Moments t{0.0, 0.0, 0.0};
  if (rl == 0) {
    double m2 = 0.0;
    for (int q = 0; q < 8; ++q) m2 += part[2][q][cl];
    t.n = N; t.mean = N > 0.0 ? S / N : 0.0; t.m2 = m2;
  }
  return t;
}

// stage 1: grid (nblk, C/64), 1024 threads = 16 row-lanes x 64 channels (a wave reads 256 contiguous bytes of a partial
// row); block b merges rows [b*R, (b+1)*R) -> out_sum/out_m2 [nblk][C] (sum = mean*n), out_cnt [nblk]
__global__ __launch_bounds__(1024) void bn_stats_reduce_kernel(const float* __restrict__ ssum, const float* __restrict__ sm2,
                                                             const float* __restrict__ scnt, int rows, int R, int C,
                                                             float* out_sum, float* out_m2, float* out_cnt) {
  __shared__ double sh[3][16][65];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int cx = blockIdx.y * 64 + cl;
  const int c = cx < C ? cx : C - 1;             // thin layers (C = 16 | 32 | 48): the lanes beyond C repeat the last channel, write nothing
  const int r0 = blockIdx.x * R;
  int r1 = r0 + R;
  if (r1 > rows) r1 = rows;
  Moments a = merge_rows(ssum, sm2, scnt, r0 + rl, r1, 16, C, c);
  sh[0][rl][cl] = a.n; sh[1][rl][cl] = a.mean; sh[2][rl][cl] = a.m2;
  __syncthreads();
  if (rl == 0 && cx < C) {
    Moments t{0.0, 0.0, 0.0};
    for (int q = 0; q < 16; ++q) merge(t, sh[0][q][cl], sh[1][q][cl], sh[2][q][cl]);
    out_sum[(int64_t)blockIdx.x * C + c] = (float)(t.mean * t.n);
    out_m2[(int64_t)blockIdx.x * C + c] = (float)t.m2;
    if (blockIdx.y == 0 && cl == 0) out_cnt[blockIdx.x] = (float)t.n;   // exact: pixel counts < 2^24
  }
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ ssum, const float* __restrict__ sm2,
                                                         const float* __restrict__ scnt, int rows, int64_t M, int C,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* running_mean, float* running_var, float momentum, float eps,
                                                         int train, float* scale, float* shift, float* save_mean,
                                                         float* save_invstd, double* moments) {
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  if (!train) {
    if (rl == 0) {
      const float is = 1.0f / sqrtf(running_var[c] + eps);
      const float sc = gamma[c] * is;
      scale[c] = sc;
      shift[c] = beta[c] - running_mean[c] * sc;
    }
    return;
  }
  Moments a = merge_rows(ssum, sm2, scnt, rl, rows, kRL, C, c);
  a = merge_lanes(a, rl, cl);
  if (rl != 0) return;
  if (moments) {   // this rank's (n, sum x, sum x^2) per channel, to be summed over ranks (cvcs_bn_moments)
    moments[c] = a.n;
    moments[C + c] = a.n * a.mean;
    moments[2 * C + c] = a.m2 + a.n * a.mean * a.mean;
    return;
  }
  const double mean = a.mean, m2 = a.m2;
  const double var = m2 / (double)M;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  const double unbiased = M > 1 ? m2 / (double)(M - 1) : var;
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}

// scale/shift/saved statistics/running statistics from all-rank moments (n, sum x, sum x^2) [3][C] in f64
__global__ __launch_bounds__(256) void bn_finalize_moments_kernel(const double* __restrict__ mom, int C,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* running_mean, float* running_var, float momentum, float eps,
                                                                float* scale, float* shift, float* save_mean, float* save_invstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double n = mom[c], mean = mom[C + c] / n;
  double m2 = mom[2 * C + c] - n * mean * mean;
  if (m2 < 0.0) m2 = 0.0;
  const double var = m2 / n;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}

__global__ __launch_bounds__(256) void bn_bwd_coeffs_kernel(const float* __restrict__ sums, double M, int C, float* ca, float* cb) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  ca[c] = (float)((double)sums[c] / M);
  cb[c] = (float)((double)sums[C + c] / M);
}

// ------------------------------------------------------------------------------------------------ BN apply fwd
// nn.Hardswish (the activation of MobileNetV3's "HS" blocks, reached through S/nets.py:277-311): x * relu6(x + 3) / 6 and its derivative
// (0 below -3, x / 3 + 1/2 up to 3, 1 above - torch's hardswish_backward)
__device__ __forceinline__ float hardswish(float z) { return z * fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f); }
__device__ __forceinline__ float hardswish_grad(float z) { return z < -3.f ? 0.f : (z <= 3.f ? z * (1.f / 3.f) + 0.5f : 1.f); }

struct BnActArgs {
  const char* y; char* out; char* pool;
  int64_t y_ld, out_ld, pool_ld;
  const float* scale; const float* shift;
  int B, H, W, C, relu;
  Q8Out q8;
};

// RELU (0 none, 1 ReLU, 2 hardswish) is a template parameter (as an argument: a scalar branch per element).  Work items are indexed in 32 bits
// (the host refuses maps of 2^31 chunks); when the grid stride is a multiple of the row's chunk count - every power-of-two width - a thread
// keeps its channel chunk, and its 2 x V scale / shift values, for the whole loop.
template <typename T, bool POOL, int RELU>
__global__ __launch_bounds__(256) void bn_act_kernel(BnActArgs p) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const unsigned CC = (unsigned)(p.C / V);
  const int64_t items = POOL ? (int64_t)p.B * (p.H / 2) * (p.W / 2) : (int64_t)p.B * p.H * p.W;
  const unsigned total = (unsigned)(items * CC);
  float am = 0.f;
  const float q_inv = p.q8.q ? p.q8.slot[2] : 0.f, q_fmax = p.q8.q ? p.q8.slot[3] : 0.f;
  const unsigned S = gridDim.x * 256u;
  const bool fixed = S % CC == 0;                 // wave-uniform
  const unsigned id0 = blockIdx.x * 256u + threadIdx.x;
  unsigned cc = id0 % CC, it32 = id0 / CC;
  const unsigned it_step = S / CC;
  float sc[V], sh[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { sc[k] = p.scale[cc * V + k]; sh[k] = p.shift[cc * V + k]; }
  for (unsigned id = id0; id < total; id += S) {
    if (!fixed) {
      cc = id % CC; it32 = id / CC;
#pragma unroll
      for (int k = 0; k < V; ++k) { sc[k] = p.scale[cc * V + k]; sh[k] = p.shift[cc * V + k]; }
    }
    const int64_t it = it32;
    if (fixed) it32 += it_step;
    if constexpr (!POOL) {
      float f[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(p.y + (it * p.y_ld) * ES + cc * 16), f);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        f[k] = f[k] * sc[k] + sh[k];
        if constexpr (RELU == 1) f[k] = fmaxf(f[k], 0.f);
        else if constexpr (RELU == 2) f[k] = hardswish(f[k]);
      }
      const uint4 pk = Elem<T>::pack(f);
      *reinterpret_cast<uint4*>(p.out + (it * p.out_ld) * ES + cc * 16) = pk;
      if constexpr (ES == 2) {
        if (p.q8.q) *reinterpret_cast<uint2*>(p.q8.q + it * p.q8.ld + cc * 8) = q8_pack8(pk, q_inv, q_fmax, p.q8.fmt, am);
      }
    } else {
      const int W2 = p.W / 2, H2 = p.H / 2;
      const int px = (int)(it % W2);
      const int64_t t = it / W2;
      const int py = (int)(t % H2);
      const int64_t b = t / H2;
      float mx[V];
#pragma unroll
      for (int k = 0; k < V; ++k) mx[k] = -INFINITY;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int64_t pix = (b * p.H + 2 * py + (w >> 1)) * p.W + 2 * px + (w & 1);
        float f[V];
        Elem<T>::unpack(*reinterpret_cast<const uint4*>(p.y + (pix * p.y_ld) * ES + cc * 16), f);
#pragma unroll
        for (int k = 0; k < V; ++k) {
          f[k] = f[k] * sc[k] + sh[k];
          if constexpr (RELU != 0) f[k] = fmaxf(f[k], 0.f);
          mx[k] = fmaxf(mx[k], f[k]);
        }
        if (p.out) *reinterpret_cast<uint4*>(p.out + (pix * p.out_ld) * ES + cc * 16) = Elem<T>::pack(f);
      }
      *reinterpret_cast<uint4*>(p.pool + (it * p.pool_ld) * ES + cc * 16) = Elem<T>::pack(mx);
    }
  }
  if constexpr (ES == 2 && !POOL) {
    if (p.q8.q && p.q8.take_amax) q8_commit_amax(am, p.q8.slot);
  }
}

// ------------------------------------------------------------------------------------------------ BN backward
struct BnBwdArgs {
  const char* y; const char* g1; const char* g2; char* dy;
  int64_t y_ld, g1_ld, g2_ld, dy_ld;
  const float* scale; const float* shift; const float* mean; const float* invstd;
  const float* ca; const float* cb;
  float* part0; float* part1;   // reduce: (dz, dz*xhat); apply: (db, unused)
  int B, H, W, C, mode, ccw;    // ccw: channel-chunk lanes per workgroup (bn_bwd_chunk_lanes)
  Q8Out q8;                     // apply: fp8 image of dy (the data gradient of an fp8 convolution reads it)
};

// One work item = one pixel (POOL=false) or one 2x2 window (POOL=true) x one 16-byte channel chunk.
// APPLY=false: accumulate sum(dz), sum(dz*xhat).  APPLY=true: write dy, accumulate sum(dy).
constexpr int kBnBwdChunks = 32;
// MODE (the activation around the BatchNorm: 0 ReLU after, 1 ReLU before, 2 none, 3 hardswish after) is a template parameter: as a kernel
// argument it cost two or three scalar branches PER ELEMENT inside the unrolled loop (the passes ran 30-40 % under a plain 2R1W stream)
template <typename T, bool POOL, bool APPLY, int MODE>
__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdArgs p) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  constexpr int NPIX = POOL ? 4 : 1;
  __shared__ float red[2][256 * V];
  const int ccw = p.ccw;                    // chunk lanes per workgroup (a 512-byte slab of a pixel at most: wide layers spread over
                                            // blockIdx.y, so that small maps still fill the chip)
  const int PL = 256 / ccw;                 // pixel lanes per workgroup; widths like 96 / 160 / 960 leave 256 - ccw * PL threads idle
  const int tid = threadIdx.x;
  const int cl = tid % ccw, pl = tid / ccw;
  const int cc = blockIdx.y * ccw + cl;
  const int64_t items = pl < PL ? (POOL ? (int64_t)p.B * (p.H / 2) * (p.W / 2) : (int64_t)p.B * p.H * p.W) : 0;
  float sc[V], sh[V], mu[V], is[V], ca[V], cb[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = cc * V + k;
    sc[k] = p.scale[c]; sh[k] = p.shift[c]; mu[k] = p.mean[c]; is[k] = p.invstd[c];
    ca[k] = APPLY ? p.ca[c] : 0.f; cb[k] = APPLY ? p.cb[c] : 0.f;
  }
  float s0[V], s1[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { s0[k] = 0.f; s1[k] = 0.f; }
  float am = 0.f;
  const float q_inv = (APPLY && p.q8.q) ? p.q8.slot[2] : 0.f, q_fmax = (APPLY && p.q8.q) ? p.q8.slot[3] : 0.f;

  // U independent work items per trip: all their 16-byte loads are issued before the first use (a single item in flight per
  // thread left the small layers latency-bound: 0.8-3 TB/s)
  // The U items of a trip are ADJACENT pixel groups (a workgroup streams U * PL consecutive pixels, consecutive workgroups consecutive
  // blocks): with the items a whole grid stride apart (gridDim.x * PL pixels = a power-of-two number of MiB) a thread's loads in flight all
  // fell on the same memory channels, and the passes ran at 3.6-4.9 TB/s where a plain two-stream add reaches 6.
  constexpr int U = POOL ? 1 : 4;
  int64_t step = (int64_t)gridDim.x * PL;
  int64_t it = (int64_t)blockIdx.x * PL + pl;
  if constexpr (!POOL) {
    const int64_t grp = (int64_t)U * PL;
    const int64_t nfull = items / grp;                  // whole groups (0 for the idle threads)
    for (int64_t gi = blockIdx.x; gi < nfull; gi += gridDim.x) {
      const int64_t it0 = gi * grp + pl;
      uint4 yr[U], gr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t px = it0 + u * PL;
        yr[u] = *reinterpret_cast<const uint4*>(p.y + (px * p.y_ld) * ES + cc * 16);
        gr[u] = *reinterpret_cast<const uint4*>(p.g1 + (px * p.g1_ld) * ES + cc * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t px = it0 + u * PL;
        float yv[V], gv[V], out[V];
        Elem<T>::unpack(yr[u], yv);
        Elem<T>::unpack(gr[u], gv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const float xh = (yv[k] - mu[k]) * is[k];
          float dz = gv[k];
          if constexpr (MODE == 0) dz = (yv[k] * sc[k] + sh[k] > 0.f) ? dz : 0.f;
          else if constexpr (MODE == 3) dz *= hardswish_grad(yv[k] * sc[k] + sh[k]);
          if constexpr (!APPLY) {
            s0[k] += dz;
            s1[k] += dz * xh;
          } else {
            float d = sc[k] * (dz - ca[k] - xh * cb[k]);
            if constexpr (MODE == 1) d = (yv[k] > 0.f) ? d : 0.f;
            out[k] = d;
            s0[k] += d;
          }
        }
        if constexpr (APPLY) {
          const uint4 pk = Elem<T>::pack(out);
          *reinterpret_cast<uint4*>(p.dy + (px * p.dy_ld) * ES + cc * 16) = pk;
          if constexpr (ES == 2) {
            if (p.q8.q) *reinterpret_cast<uint2*>(p.q8.q + px * p.q8.ld + cc * 8) = q8_pack8(pk, q_inv, q_fmax, p.q8.fmt, am);
          }
        }
      }
    }
  }
  if constexpr (!POOL) {      // the last, partial group: one workgroup, one pixel at a time
    const int64_t grp = (int64_t)U * PL, nfull = items / grp;
    it = (int64_t)blockIdx.x == nfull % gridDim.x ? nfull * grp + pl : items;
    step = PL;
  }
  for (; it < items; it += step) {
    int64_t pix[NPIX];
    if constexpr (POOL) {
      const int W2 = p.W / 2, H2 = p.H / 2;
      const int px = (int)(it % W2);
      const int64_t t = it / W2;
      const int py = (int)(t % H2);
      const int64_t b = t / H2;
#pragma unroll
      for (int w = 0; w < 4; ++w) pix[w] = (b * p.H + 2 * py + (w >> 1)) * p.W + 2 * px + (w & 1);
    } else {
      pix[0] = it;
    }
    float yv[NPIX][V], gv[NPIX][V];
#pragma unroll
    for (int w = 0; w < NPIX; ++w) {
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(p.y + (pix[w] * p.y_ld) * ES + cc * 16), yv[w]);
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(p.g1 + (pix[w] * p.g1_ld) * ES + cc * 16), gv[w]);
    }
    if constexpr (POOL) {
      // MaxPool2d backward: the pooled gradient goes to the FIRST maximum of the window in scan order
      float g2[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(p.g2 + (it * p.g2_ld) * ES + cc * 16), g2);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        float best = -INFINITY;
        int arg = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float a = yv[w][k] * sc[k] + sh[k];
          a = fmaxf(a, 0.f);
          if (a > best) { best = a; arg = w; }
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) gv[w][k] += (w == arg) ? g2[k] : 0.f;
      }
    }
#pragma unroll
    for (int w = 0; w < NPIX; ++w) {
      float out[V];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float xh = (yv[w][k] - mu[k]) * is[k];
        float dz = gv[w][k];
        if constexpr (MODE == 0) dz = (yv[w][k] * sc[k] + sh[k] > 0.f) ? dz : 0.f;  // encoder: ReLU after BN
        else if constexpr (MODE == 3) dz *= hardswish_grad(yv[w][k] * sc[k] + sh[k]);  // hardswish after BN
        if constexpr (!APPLY) {
          s0[k] += dz;
          s1[k] += dz * xh;
        } else {
          float d = sc[k] * (dz - ca[k] - xh * cb[k]);
          if constexpr (MODE == 1) d = (yv[w][k] > 0.f) ? d : 0.f;                  // decoder: ReLU before BN
          out[k] = d;
          s0[k] += d;
        }
      }
      if constexpr (APPLY) {
        const uint4 pk = Elem<T>::pack(out);
        *reinterpret_cast<uint4*>(p.dy + (pix[w] * p.dy_ld) * ES + cc * 16) = pk;
        if constexpr (ES == 2) {
          if (p.q8.q) *reinterpret_cast<uint2*>(p.q8.q + pix[w] * p.q8.ld + cc * 8) = q8_pack8(pk, q_inv, q_fmax, p.q8.fmt, am);
        }
      }
    }
  }
  if constexpr (APPLY && ES == 2) {
    if (p.q8.q && p.q8.take_amax) q8_commit_amax(am, p.q8.slot);
  }
  // ---- combine the PL pixel lanes: red[.][q][e], e = one of the workgroup's n_out = ccw * V channels.  Every thread sums a slice of the
  // lanes of ONE channel, the slices are merged in order (ccw threads walking all PL lanes of V channels each - 2048 dependent LDS reads
  // for a 16-channel map - were a serial tail of 10-50 us behind every launch)
#pragma unroll
  for (int k = 0; k < V; ++k) { red[0][tid * V + k] = s0[k]; red[1][tid * V + k] = s1[k]; }
  __syncthreads();
  {
    const int n_out = ccw * V, P = 256 / n_out > 0 ? 256 / n_out : 1;
    const int e = tid % n_out, part = tid / n_out;
    float a = 0.f, b2 = 0.f;
    if (part < P)
      for (int q = part; q < PL; q += P) { a += red[0][q * n_out + e]; if constexpr (!APPLY) b2 += red[1][q * n_out + e]; }
    __syncthreads();
    if (part < P) { red[0][part * n_out + e] = a; red[1][part * n_out + e] = b2; }
    __syncthreads();
    if (tid < n_out) {
      a = 0.f; b2 = 0.f;
      for (int j = 0; j < P; ++j) { a += red[0][j * n_out + e]; if constexpr (!APPLY) b2 += red[1][j * n_out + e]; }
      const int64_t o = (int64_t)blockIdx.x * p.C + (int64_t)blockIdx.y * n_out + e;
      p.part0[o] = a;
      if constexpr (!APPLY) p.part1[o] = b2;
    }
  }
}

// partial per-channel sums of a view: part[blockIdx.x][c] = sum over this workgroup's pixels (bias gradients)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const char* x, int64_t ld, int64_t M, int C, float* part) {
  // any C that is a multiple of the 16-byte chunk (the 3 * 2^k widths of Swin are no powers of two): a workgroup covers
  // ccw = min(C/V, 256) chunks x PL = floor(256 / ccw) pixel lanes; the threads beyond ccw * PL idle
  constexpr int ES = sizeof(T), V = 16 / ES;
  __shared__ float red[256 * V];
  const int CC = C / V;
  const int ccw = CC < 256 ? CC : 256;
  const int PL = 256 / ccw;
  const int tid = threadIdx.x;
  const int cl = tid % ccw, pl = tid / ccw;
  const int cc = blockIdx.y * ccw + cl;
  const bool active = pl < PL && cc < CC;
  float s[V];
#pragma unroll
  for (int k = 0; k < V; ++k) s[k] = 0.f;
  if (active)
    for (int64_t it = (int64_t)blockIdx.x * PL + pl; it < M; it += (int64_t)gridDim.x * PL) {
      float f[V];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + (it * ld) * ES + cc * 16), f);
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] += f[k];
    }
#pragma unroll
  for (int k = 0; k < V; ++k) red[tid * V + k] = s[k];
  __syncthreads();
  if (pl == 0 && cc < CC) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float a = 0.f;
      for (int q = 0; q < PL; ++q) a += red[(q * ccw + cl) * V + k];
      part[(int64_t)blockIdx.x * C + cc * V + k] = a;
    }
  }
}

// column sums of up to 1024 partial rows: workgroup = kRL row-lanes x 16 channels, f64 accumulate, fixed order
__device__ __forceinline__ double colsum16(const float* __restrict__ part, int rows, int C, int c, int rl, int cl, int slot) {
  __shared__ double sh[2][kRL][17];
  double a = 0.0;
  int r = rl;
  for (; r + 15 * kRL < rows; r += 16 * kRL) {        // sixteen, then four loads in flight per lane
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = part[(int64_t)(r + u * kRL) * C + c];
#pragma unroll
    for (int u = 0; u < 16; ++u) a += (double)v[u];
  }
  for (; r + 3 * kRL < rows; r += 4 * kRL) {
    const float v0 = part[(int64_t)r * C + c], v1 = part[(int64_t)(r + kRL) * C + c];
    const float v2 = part[(int64_t)(r + 2 * kRL) * C + c], v3 = part[(int64_t)(r + 3 * kRL) * C + c];
    a += (double)v0; a += (double)v1; a += (double)v2; a += (double)v3;
  }
  for (; r < rows; r += kRL) a += (double)part[(int64_t)r * C + c];
  sh[slot][rl][cl] = a;
  __syncthreads();
  if (rl < 8) {                                       // two levels, fixed order: lane rl sums entries rl, rl + 8, ...
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < kRL / 8; ++q) t += sh[slot][rl + 8 * q][cl];
    sh[slot][rl][cl] = t;
  }
  __syncthreads();
  double t = 0.0;
  if (rl == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sh[slot][q][cl];
  }
  return t;
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ pdz, const float* __restrict__ pdzx,
                                                             int rows, int64_t M, int C, const float* __restrict__ gamma,
                                                             const float* __restrict__ invstd, float* dgamma, float* dbeta,
                                                             float* ca, float* cb) {
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const double a = colsum16(pdz, rows, C, c, rl, cl, 0);
  const double b = colsum16(pdzx, rows, C, c, rl, cl, 1);
  if (rl != 0) return;
  dbeta[c] = (float)a;
  dgamma[c] = (float)b;
  ca[c] = (float)(a / (double)M);
  cb[c] = (float)(b / (double)M);
}

__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ part, int rows, int C, float* out) {
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const bool ok = c < C;
  const double a = colsum16(part, ok ? rows : 0, C, ok ? c : 0, rl, cl, 0);
  if (rl == 0 && ok) out[c] = (float)a;
}

// ------------------------------------------------------------------------------------------------ bilinear x2
// align_corners=False, scale 2: out row 2k   = .25*in[k-1] + .75*in[k]   (k=0: in[0])
//                               out row 2k+1 = .75*in[k]   + .25*in[k+1] (k=H-1: in[H-1]); columns alike.
__device__ __forceinline__ void up_taps(int o, int n, int& i0, int& i1, float& w0, float& w1) {
  const int k = o >> 1;
  if (o & 1) { i0 = k; i1 = k + 1 < n ? k + 1 : k; w0 = 0.75f; w1 = 0.25f; }
  else       { i0 = k > 0 ? k - 1 : 0; i1 = k; w0 = k > 0 ? 0.25f : 0.f; w1 = k > 0 ? 0.75f : 1.f; }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const char* in, int64_t in_ld, int B, int H, int W, int C,
                                                            char* out, int64_t out_ld, Q8Out q8) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int64_t total = (int64_t)B * 2 * H * 2 * W * CC;
  float am = 0.f;
  const float q_inv = q8.q ? q8.slot[2] : 0.f, q_fmax = q8.q ? q8.slot[3] : 0.f;
  const unsigned xbid = xcd_order(blockIdx.x, gridDim.x);
  // XCD-aware order: the workgroups of one XCD take a contiguous span of rows per grid stride, so the rows two neighbouring outputs share are
  // hits of that XCD's L2 (dealt round-robin, vertical neighbours sat in different L2s: PMC 2.3x the algorithmic reads for upsample2x_bwd)
  for (int64_t id = (int64_t)xbid * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    int64_t t = id / CC;
    const int ox = (int)(t % (2 * W)); t /= 2 * W;
    const int oy = (int)(t % (2 * H));
    const int64_t b = t / (2 * H);
    int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
    up_taps(oy, H, y0, y1, wy0, wy1);
    up_taps(ox, W, x0, x1, wx0, wx1);
    float a[V], bb[V], c[V], d[V], o[V];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(in + (((b * H + y0) * W + x0) * in_ld) * ES + cc * 16), a);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(in + (((b * H + y0) * W + x1) * in_ld) * ES + cc * 16), bb);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(in + (((b * H + y1) * W + x0) * in_ld) * ES + cc * 16), c);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(in + (((b * H + y1) * W + x1) * in_ld) * ES + cc * 16), d);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = wy0 * (wx0 * a[k] + wx1 * bb[k]) + wy1 * (wx0 * c[k] + wx1 * d[k]);
    const int64_t opix = ((b * 2 * H) + oy) * (2 * W) + ox;
    const uint4 pk = Elem<T>::pack(o);
    *reinterpret_cast<uint4*>(out + (opix * out_ld) * ES + cc * 16) = pk;
    if constexpr (ES == 2) {
      if (q8.q) *reinterpret_cast<uint2*>(q8.q + opix * q8.ld + cc * 8) = q8_pack8(pk, q_inv, q_fmax, q8.fmt, am);
    }
  }
  if constexpr (ES == 2) {
    if (q8.q && q8.take_amax) q8_commit_amax(am, q8.slot);
  }
}

// gather form of the transpose: input row y collects output rows 2y-1 (.25), 2y (.75 | 1), 2y+1 (.75 | 1), 2y+2 (.25)
__device__ __forceinline__ void up_taps_t(int i, int n, int o[4], float w[4]) {
  o[0] = 2 * i - 1; w[0] = i > 0 ? 0.25f : 0.f;
  o[1] = 2 * i;     w[1] = i > 0 ? 0.75f : 1.f;
  o[2] = 2 * i + 1; w[2] = i < n - 1 ? 0.75f : 1.f;
  o[3] = 2 * i + 2; w[3] = i < n - 1 ? 0.25f : 0.f;
  if (i == 0) o[0] = 0;
  if (i == n - 1) o[3] = 2 * n - 1;
}

// A thread owns one (column, 16-byte channel chunk) and walks a STRIP of kUpStrip low-resolution rows: row y needs the high-resolution rows
// 2y-1 .. 2y+2, of which 2y+1 and 2y+2 are the first two rows of y + 1 - they stay in registers, already reduced along x (8 loads per output
// instead of 16, and every high-resolution row of a strip is fetched ONCE: the per-pixel gather re-read each row for the low-resolution row
// above and below it, 2.3 x the algorithmic bytes in the PMC counters).
constexpr int kUpStrip = 8;
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const char* g, int64_t g_ld, int B, int H, int W, int C,
                                                            char* gin, int64_t gin_ld) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int CC = C / V;
  const int strips = (H + kUpStrip - 1) / kUpStrip;
  const int64_t total = (int64_t)B * strips * W * CC;
  const unsigned xbid = xcd_order(blockIdx.x, gridDim.x);
  for (int64_t id = (int64_t)xbid * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int cc = (int)(id % CC);
    int64_t t = id / CC;
    const int x = (int)(t % W); t /= W;
    const int sy = (int)(t % strips);
    const int64_t b = t / strips;
    int ox[4]; float wx[4];
    up_taps_t(x, W, ox, wx);
    const char* gb = g + (b * 2 * H) * (int64_t)(2 * W) * g_ld * ES + cc * 16;
    // one high-resolution row reduced along x with this column's four weights
    auto hrow = [&](int oy, float* h) {
      uint4 raw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) raw[j] = *reinterpret_cast<const uint4*>(gb + ((int64_t)oy * (2 * W) + ox[j]) * g_ld * ES);
#pragma unroll
      for (int k = 0; k < V; ++k) h[k] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float f[V];
        Elem<T>::unpack(raw[j], f);
#pragma unroll
        for (int k = 0; k < V; ++k) h[k] += wx[j] * f[k];      // (a zero weight belongs to a clamped, valid address)
      }
    };
    const int y0 = sy * kUpStrip;
    const int y1 = y0 + kUpStrip < H ? y0 + kUpStrip : H;
    float hA[V], hB[V], hC[V], hD[V];
    {
      int oy[4]; float wy[4];
      up_taps_t(y0, H, oy, wy);
      hrow(oy[0], hA);
      hrow(oy[1], hB);
    }
    for (int y = y0; y < y1; ++y) {
      int oy[4]; float wy[4];
      up_taps_t(y, H, oy, wy);
      hrow(oy[2], hC);
      hrow(oy[3], hD);
      float acc[V];
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = wy[0] * hA[k] + wy[1] * hB[k] + wy[2] * hC[k] + wy[3] * hD[k];
      *reinterpret_cast<uint4*>(gin + (((b * H + y) * W + x) * gin_ld) * ES + cc * 16) = Elem<T>::pack(acc);
#pragma unroll
      for (int k = 0; k < V; ++k) { hA[k] = hC[k]; hB[k] = hD[k]; }      // rows 2y+1, 2y+2 = rows 2(y+1)-1, 2(y+1)
    }
  }
}

// ------------------------------------------------------------------------------------------------ boundary packs
template <typename T, typename S>
__global__ __launch_bounds__(256) void pack_input_kernel(const S* __restrict__ src, int B, int C, int H, int W, char* dst,
                                                        int Cpad) {
  constexpr int ES = sizeof(T), V = 16 / ES;
  const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
  for (int64_t pidx = (int64_t)blockIdx.x * 256 + threadIdx.x; pidx < total; pidx += (int64_t)gridDim.x * 256) {
    const int64_t b = pidx / HW, hw = pidx - b * HW;
    for (int c0 = 0; c0 < Cpad; c0 += V) {
      float f[V];
#pragma unroll
      for (int k = 0; k < V; ++k) f[k] = (c0 + k < C) ? (float)src[(b * C + c0 + k) * HW + hw] : 0.f;
      *reinterpret_cast<uint4*>(dst + (pidx * Cpad + c0) * ES) = Elem<T>::pack(f);
    }
  }
}

template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int KW, int Cin_pad,
                                        T* wf, T* wd) {
  const int taps = KH * KW;
  const int64_t nf = (int64_t)taps * Cout * Cin_pad;
  const int64_t nd = wd ? (int64_t)taps * Cin * Cout : 0;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < nf + nd; id += (int64_t)gridDim.x * blockDim.x) {
    if (id < nf) {
      const int ci = (int)(id % Cin_pad);
      const int64_t r = id / Cin_pad;
      const int co = (int)(r % Cout);
      const int t = (int)(r / Cout);
      const float v = ci < Cin ? w[((int64_t)co * Cin + ci) * taps + t] : 0.f;
      Elem<T>::st(wf + id, v);
    } else {
      // data-gradient operand: wd[t'][ci][co] = w[co][ci][KH-1-kh'][KW-1-kw'] (180-degree flip, roles swapped)
      const int64_t j = id - nf;
      const int co = (int)(j % Cout);
      const int64_t r = j / Cout;
      const int ci = (int)(r % Cin);
      const int tp = (int)(r / Cin);
      const int t = taps - 1 - tp;
      Elem<T>::st(wd + j, w[((int64_t)co * Cin + ci) * taps + t]);
    }
  }
}

// every conv of the network in ONE launch (the per-step re-pack after the optimiser): blockIdx.y = table entry.
// A workgroup moves a block of 32 output x 32 input channels x all taps through LDS: the OIHW source is read as 32
// contiguous runs (coalesced; a thread-per-output-element gather re-fetched every line ~30x), the forward operand leaves
// as 64-byte rows wf[t][co][ci0..ci0+31], the flipped data-gradient operand as 64-byte pieces wd[t'][ci][co0..co0+31]
// (16 x 64 blocks left wd as scattered 32-byte pieces - half a memory burst each: the launch wrote at a quarter of the rate it read).
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const cvcs_pack_item* __restrict__ items) {
  constexpr int CO = 32, CI = 32, MAXT = 9;
  __shared__ float sw[CO][CI * MAXT + CI * MAXT / 32 + 1];      // element j of a row sits at j + j / 32: the 16-byte gathers below step 8 floats per lane
  auto at = [](int j) { return j + (j >> 5); };
  const cvcs_pack_item it = items[blockIdx.y];
  const float* __restrict__ w = it.w;
  T* wf = reinterpret_cast<T*>(it.w_fwd);
  T* wd = reinterpret_cast<T*>(it.w_dgrad);
  const int Cout = it.Cout, Cin = it.Cin, Cin_pad = it.Cin_pad, taps = it.KH * it.KW;
  const int Cout_pad = it.Cout_pad > 0 ? it.Cout_pad : Cout;      // row length of the data-gradient image (columns Cout.. stay as allocated: zero)
  // input channels per block: as many 32-channel groups as the 288-float LDS rows hold (1x1 convs: 288 channels)
  const int CIb = CI * (MAXT / taps);
  const int nci = (Cin_pad + CIb - 1) / CIb, nco = (Cout + CO - 1) / CO;      // (a 16-channel layer: the rows beyond Cout load zeros, store nothing)
  for (int blk = blockIdx.x; blk < nco * nci; blk += gridDim.x) {
    const int co0 = (blk / nci) * CO, ci0 = (blk % nci) * CIb;
    const int cin_here = Cin - ci0 < CIb ? (Cin - ci0 < 0 ? 0 : Cin - ci0) : CIb;    // real input channels in this block
    const int run = cin_here * taps;                                                   // contiguous floats per output channel
    const int cpad_here = Cin_pad - ci0 < CIb ? Cin_pad - ci0 : CIb;                   // a multiple of V (Cin_pad is a K-group)
    __syncthreads();
    for (int id = threadIdx.x; id < CO * cpad_here * taps; id += 256) {
      const int c = id / (cpad_here * taps), r = id - c * (cpad_here * taps);
      sw[c][at(r)] = (r < run && co0 + c < Cout) ? w[((int64_t)(co0 + c) * Cin + ci0) * taps + r] : 0.f;   // r = ci_local * taps + t
    }
    __syncthreads();
    constexpr int V = 16 / sizeof(T);                                                  // elements per 16-byte store
    for (int id = threadIdx.x; id < taps * CO * (cpad_here / V); id += 256) {          // wf[t][co][ci]: 16 bytes of ci per lane
      const int cv = id % (cpad_here / V), r = id / (cpad_here / V), c = r % CO, t = r / CO;
      float f[V];
#pragma unroll
      for (int k = 0; k < V; ++k) f[k] = sw[c][at((cv * V + k) * taps + t)];
      if (co0 + c < Cout) *reinterpret_cast<uint4*>(wf + ((int64_t)t * Cout + co0 + c) * Cin_pad + ci0 + cv * V) = Elem<T>::pack(f);
    }
    if (wd) {
      for (int id = threadIdx.x; id < taps * cin_here * (CO / V); id += 256) {         // wd[t'][ci][co]: 16 bytes of co per lane
        const int cv = id % (CO / V), r = id / (CO / V), ci = r % cin_here, tp = r / cin_here;
        float f[V];
#pragma unroll
        for (int k = 0; k < V; ++k) f[k] = sw[cv * V + k][at(ci * taps + (taps - 1 - tp))];
        if (co0 + cv * V < Cout) *reinterpret_cast<uint4*>(wd + ((int64_t)tp * Cin + ci0 + ci) * Cout_pad + co0 + cv * V) = Elem<T>::pack(f);
      }
    }
  }
}

// ConvTranspose2d(k2,s2) weight [Cin][Cout][2][2]:
//   forward  operand wf[n = t*Cout + co][ci] (a 1x1 conv to 4*Cout columns, pixel-shuffled by the epilogue)
//   backward operand wd[t][ci][co]           (a 2x2 stride-2 conv over the output gradient)
template <typename T>
__global__ void pack_convT_weight_kernel(const float* __restrict__ w, int Cin, int Cout, T* wf, float* bias4,
                                         const float* __restrict__ bias, T* wd) {
  const int64_t n = (int64_t)4 * Cout * Cin;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < 2 * n + 4 * Cout; id += (int64_t)gridDim.x * blockDim.x) {
    if (id < n) {
      const int ci = (int)(id % Cin);
      const int64_t r = id / Cin;
      const int co = (int)(r % Cout);
      const int t = (int)(r / Cout);
      Elem<T>::st(wf + id, w[((int64_t)ci * Cout + co) * 4 + t]);
    } else if (id < 2 * n) {
      const int64_t j = id - n;
      const int co = (int)(j % Cout);
      const int64_t r = j / Cout;
      const int ci = (int)(r % Cin);
      const int t = (int)(r / Cin);
      if (wd) Elem<T>::st(wd + j, w[((int64_t)ci * Cout + co) * 4 + t]);
    } else {
      const int j = (int)(id - 2 * n);
      bias4[j] = bias ? bias[j % Cout] : 0.f;
    }
  }
}

static inline int grid_for(int64_t total, int per_block = 256, int cap = 256 * 16) {
  int64_t g = cdiv(total, per_block);
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace cvcs

using namespace cvcs;

#define DT_OK(dt) ((dt) == CVCS_F32 || (dt) == CVCS_BF16)

extern "C" int cvcs_bn_finalize_workspace_floats(int rows, int C) {
  if (rows <= kDirectRows) return 0;
  const int R = (int)cdiv(rows, 1024);
  return (int)cdiv(rows, R) * (2 * C + 1);
}

// merge > kDirectRows partial rows down to <= 1024 block rows in the workspace (stage 1 of the two-stage merge)
static int bn_stage1(const char* fn, const float*& stat_sum, const float*& stat_m2, const float*& stat_cnt, int& rows, int C,
                     float* workspace, hipStream_t st) {
  if (rows <= kDirectRows) return CVCS_OK;
  CVCS_CHECK_ARG(workspace != nullptr, "%s: %d partial rows need the workspace", fn, rows);
  CVCS_CHECK_ARG(C % 16 == 0, "%s: the two-stage merge needs C %% 16 == 0", fn);
  const int R = (int)cdiv(rows, 1024);
  const int nblk = (int)cdiv(rows, R);
  float* o_sum = workspace;
  float* o_m2 = workspace + (int64_t)nblk * C;
  float* o_cnt = workspace + (int64_t)2 * nblk * C;
  hipLaunchKernelGGL(bn_stats_reduce_kernel, dim3((unsigned)nblk, (unsigned)cdiv(C, 64)), dim3(1024), 0, st, stat_sum, stat_m2,
                     stat_cnt, rows, R, C, o_sum, o_m2, o_cnt);
  stat_sum = o_sum; stat_m2 = o_m2; stat_cnt = o_cnt; rows = nblk;
  return CVCS_OK;
}

extern "C" int cvcs_bn_finalize(const float* stat_sum, const float* stat_m2, const float* stat_cnt, int rows, int64_t M, int C,
                                const float* gamma, const float* beta, float* running_mean, float* running_var,
                                float momentum, float eps, int train, float* scale, float* shift, float* save_mean,
                                float* save_invstd, float* workspace, void* stream) {
  CVCS_CHECK_ARG(C > 0 && C % 16 == 0 && gamma && beta && running_mean && running_var && scale && shift,
                 "cvcs_bn_finalize: null argument or C %% 16 != 0");
  hipStream_t st = (hipStream_t)stream;
  if (train) {
    CVCS_CHECK_ARG(stat_sum && stat_m2 && stat_cnt && save_mean && save_invstd && M > 0 && rows > 0,
                   "cvcs_bn_finalize: train needs statistics");
    int rc;
    if ((rc = bn_stage1("cvcs_bn_finalize", stat_sum, stat_m2, stat_cnt, rows, C, workspace, st))) return rc;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)(C / 16)), dim3(16 * kRL), 0, st, stat_sum, stat_m2, stat_cnt, rows, M, C,
                     gamma, beta, running_mean, running_var, momentum, eps, train, scale, shift, save_mean, save_invstd,
                     (double*)nullptr);
  CVCS_CHECK_LAUNCH("cvcs_bn_finalize");
  return CVCS_OK;
}

extern "C" int cvcs_bn_moments(const float* stat_sum, const float* stat_m2, const float* stat_cnt, int rows, int C,
                               double* moments, float* workspace, void* stream) {
  CVCS_CHECK_ARG(C > 0 && C % 16 == 0 && stat_sum && stat_m2 && stat_cnt && moments && rows > 0,
                 "cvcs_bn_moments: null argument or C %% 16 != 0");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if ((rc = bn_stage1("cvcs_bn_moments", stat_sum, stat_m2, stat_cnt, rows, C, workspace, st))) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)(C / 16)), dim3(16 * kRL), 0, st, stat_sum, stat_m2, stat_cnt, rows,
                     (int64_t)0, C, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, 0.f, 1,
                     (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, moments);
  CVCS_CHECK_LAUNCH("cvcs_bn_moments");
  return CVCS_OK;
}

extern "C" int cvcs_bn_finalize_moments(const double* moments, int C, const float* gamma, const float* beta, float* running_mean,
                                        float* running_var, float momentum, float eps, float* scale, float* shift,
                                        float* save_mean, float* save_invstd, void* stream) {
  CVCS_CHECK_ARG(C > 0 && moments && gamma && beta && running_mean && running_var && scale && shift && save_mean && save_invstd,
                 "cvcs_bn_finalize_moments: null argument");
  hipLaunchKernelGGL(bn_finalize_moments_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, moments, C, gamma,
                     beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd);
  CVCS_CHECK_LAUNCH("cvcs_bn_finalize_moments");
  return CVCS_OK;
}

extern "C" int cvcs_bn_bwd_coeffs(const float* sums, int64_t M, int C, float* coef_a, float* coef_b, void* stream) {
  CVCS_CHECK_ARG(sums && coef_a && coef_b && C > 0 && M > 0, "cvcs_bn_bwd_coeffs: bad argument");
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, (double)M, C,
                     coef_a, coef_b);
  CVCS_CHECK_LAUNCH("cvcs_bn_bwd_coeffs");
  return CVCS_OK;
}

static int check_view(const char* fn, const void* p, int64_t ld, int C, int es) {
  CVCS_CHECK_ARG(p != nullptr, "%s: null tensor", fn);
  CVCS_CHECK_ARG(((uintptr_t)p % 16) == 0 && (ld * es) % 16 == 0 && ld >= C, "%s: view must be 16-byte aligned, ld >= C", fn);
  return CVCS_OK;
}

static int check_q8(const char* fn, const Q8Out& q, int C, int dtype) {
  if (!q.q) return CVCS_OK;
  CVCS_CHECK_ARG(dtype == CVCS_BF16, "%s: the fp8 side output goes with bf16 storage", fn);
  CVCS_CHECK_ARG(q.slot && (q.fmt == CVCS_E4M3 || q.fmt == CVCS_E5M2), "%s: fp8 side output needs a scale slot and a format", fn);
  CVCS_CHECK_ARG(q.ld >= C && q.ld % 8 == 0 && ((uintptr_t)q.q % 8) == 0, "%s: fp8 view", fn);
  return CVCS_OK;
}

static int bn_act_impl(const void* y, int64_t y_ld, int B, int H, int W, int C, const float* scale, const float* shift,
                       int relu, void* out, int64_t out_ld, void* pool, int64_t pool_ld, int dtype, void* stream, Q8Out q8);

extern "C" int cvcs_bn_act(const void* y, int64_t y_ld, int B, int H, int W, int C, const float* scale, const float* shift,
                           int relu, void* out, int64_t out_ld, void* pool, int64_t pool_ld, int dtype, void* stream) {
  return bn_act_impl(y, y_ld, B, H, W, C, scale, shift, relu, out, out_ld, pool, pool_ld, dtype, stream, Q8Out{nullptr, 0, nullptr, 0, 0});
}

extern "C" int cvcs_bn_act_q8(const void* y, int64_t y_ld, int B, int H, int W, int C, const float* scale, const float* shift,
                              int relu, void* out, int64_t out_ld, void* q8, int64_t q8_ld, int fmt, float* slot, int take_amax, int dtype,
                              void* stream) {
  CVCS_CHECK_ARG(q8 != nullptr && out != nullptr, "cvcs_bn_act_q8: null output");
  return bn_act_impl(y, y_ld, B, H, W, C, scale, shift, relu, out, out_ld, nullptr, 0, dtype, stream, Q8Out{(char*)q8, q8_ld, slot, fmt, take_amax});
}

static int bn_act_impl(const void* y, int64_t y_ld, int B, int H, int W, int C, const float* scale, const float* shift,
                       int relu, void* out, int64_t out_ld, void* pool, int64_t pool_ld, int dtype, void* stream, Q8Out q8) {
  CVCS_CHECK_ARG(DT_OK(dtype), "cvcs_bn_act: bad dtype");
  { int rq = check_q8("cvcs_bn_act", q8, C, dtype); if (rq) return rq; }
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0, "cvcs_bn_act: bad shape");
  int rc;
  if ((rc = check_view("cvcs_bn_act", y, y_ld, C, es))) return rc;
  CVCS_CHECK_ARG(out || pool, "cvcs_bn_act: nothing to write");
  if (out && (rc = check_view("cvcs_bn_act", out, out_ld, C, es))) return rc;
  if (pool) {
    if ((rc = check_view("cvcs_bn_act", pool, pool_ld, C, es))) return rc;
    CVCS_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "cvcs_bn_act: pooling needs even H, W");
  }
  CVCS_CHECK_ARG(scale && shift, "cvcs_bn_act: null scale/shift");
  BnActArgs a{(const char*)y, (char*)out, (char*)pool, y_ld, out_ld, pool_ld, scale, shift, B, H, W, C, relu, q8};
  const int64_t total = (pool ? (int64_t)B * (H / 2) * (W / 2) : (int64_t)B * H * W) * (C / (16 / es));
  dim3 grid(grid_for(total, 256, 256 * 32));
  hipStream_t st = (hipStream_t)stream;
  CVCS_CHECK_ARG(total < (1ll << 31), "cvcs_bn_act: map too large (2^31 16-byte chunks)");
  CVCS_CHECK_ARG(relu >= 0 && relu <= 2 && !(pool && relu == 2), "cvcs_bn_act: relu=%d", relu);
#define LAUNCH_BNA(TT, PP)                                                                                   \
  do {                                                                                                       \
    if (relu == 0) hipLaunchKernelGGL((bn_act_kernel<TT, PP, 0>), grid, dim3(256), 0, st, a);                \
    else if (relu == 1) hipLaunchKernelGGL((bn_act_kernel<TT, PP, 1>), grid, dim3(256), 0, st, a);           \
    else hipLaunchKernelGGL((bn_act_kernel<TT, PP, 2>), grid, dim3(256), 0, st, a);                          \
  } while (0)
  if (dtype == CVCS_F32) { if (pool) LAUNCH_BNA(float, true); else LAUNCH_BNA(float, false); }
  else { if (pool) LAUNCH_BNA(bf16_t, true); else LAUNCH_BNA(bf16_t, false); }
#undef LAUNCH_BNA
  CVCS_CHECK_LAUNCH("cvcs_bn_act");
  return CVCS_OK;
}

extern "C" int cvcs_bn_bwd_rows(int64_t M) {
  static const int cap = getenv("CVCS_BN_BWD_ROWS") ? atoi(getenv("CVCS_BN_BWD_ROWS")) : 1024;   // tuning knob: workgroups (= partial rows)
  int64_t r = cdiv(M, 64);       // >= 64 pixels per workgroup row; maps of 16^2 x 32 tiles still get 128 rows x C/256 slabs
  return (int)(r < 1 ? 1 : (r > cap ? cap : r));
}

// Channel-chunk lanes per workgroup of the BatchNorm-backward passes: the largest divisor of the row's 16-byte chunk count up to kBnBwdChunks, so
// that gridDim.y * lanes covers a pixel row EXACTLY; the 256 / lanes pixel lanes then leave 256 - lanes * (256 / lanes) threads idle, which
// the kernel keeps out of the pixel loop (`items = pl < PL ? ... : 0`).  History (round 3, the MobileNet bring-up at 4 x 64^2 in f32): an
// interim rule lanes = min(chunks, 32) admitted widths like 96 (24 chunks of 4 floats: 10 pixel lanes, 16 threads left over) while the kernel
// had no such guard - the left-over threads took pixel lane 10, i.e. pixels of the NEXT workgroup and, in the last workgroup of the 4 x 4-pixel
// maps, up to 3 grid strides past the last pixel of y and g: the "memory access fault" of that session (last launch cvcs_bn_bwd_reduce).
extern "C" int cvcs_bn_bwd_chunk_lanes(int C, int dtype) {
  if (!DT_OK(dtype) || C <= 0) return CVCS_EINVAL;
  const int V = dtype == CVCS_F32 ? 4 : 8;
  if (C % V != 0) return CVCS_EINVAL;
  const int CC = C / V;
  int ccw = 1;      // (powers of two and multiples of 32 use every thread)
  for (int d = 1; d <= kBnBwdChunks && d <= CC; ++d)
    if (CC % d == 0) ccw = d;
  return ccw;
}

static int bn_bwd_common(const char* fn, bool apply, const void* y, int64_t y_ld, const void* g1, int64_t g1_ld,
                         const void* g2, int64_t g2_ld, int B, int H, int W, int C, const float* scale, const float* shift,
                         const float* mean, const float* invstd, const float* ca, const float* cb, int mode, void* dy,
                         int64_t dy_ld, float* part0, float* part1, int dtype, void* stream, Q8Out q8 = Q8Out{nullptr, 0, nullptr, 0, 0}) {
  CVCS_CHECK_ARG(DT_OK(dtype), "%s: bad dtype", fn);
  { int rq = check_q8(fn, q8, C, dtype); if (rq) return rq; }
  CVCS_CHECK_ARG(!q8.q || (apply && !g2), "%s: the fp8 side output belongs to the apply pass without pooling", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  const int V = 16 / es;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % V == 0, "%s: bad shape", fn);
  const int CC = C / V;
  CVCS_CHECK_ARG(mode >= 0 && mode <= 3, "%s: bad mode", fn);
  CVCS_CHECK_ARG(!(g2 && mode == 1), "%s: pooled gradient only in encoder mode", fn);
  int rc;
  if ((rc = check_view(fn, y, y_ld, C, es))) return rc;
  if ((rc = check_view(fn, g1, g1_ld, C, es))) return rc;
  if (g2) {
    if ((rc = check_view(fn, g2, g2_ld, C, es))) return rc;
    CVCS_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "%s: pooling needs even H, W", fn);
  }
  if (apply && (rc = check_view(fn, dy, dy_ld, C, es))) return rc;
  CVCS_CHECK_ARG(scale && shift && mean && invstd && part0 && (apply ? (ca && cb) : part1 != nullptr), "%s: null argument", fn);
  BnBwdArgs a{(const char*)y, (const char*)g1, (const char*)g2, (char*)dy, y_ld, g1_ld, g2_ld, dy_ld, scale, shift, mean,
              invstd, ca, cb, part0, part1, B, H, W, C, mode, 0, q8};
  const int ccw = cvcs_bn_bwd_chunk_lanes(C, dtype);
  a.ccw = ccw;
  dim3 grid((unsigned)cvcs_bn_bwd_rows((int64_t)B * H * W), (unsigned)(CC / ccw));
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BNB3(TT, PP, AA)                                                                                       \
  do {                                                                                                                \
    switch (mode) {                                                                                                   \
      case 0: hipLaunchKernelGGL((bn_bwd_kernel<TT, PP, AA, 0>), grid, dim3(256), 0, st, a); break;                   \
      case 1: hipLaunchKernelGGL((bn_bwd_kernel<TT, PP, AA, 1>), grid, dim3(256), 0, st, a); break;                   \
      case 2: hipLaunchKernelGGL((bn_bwd_kernel<TT, PP, AA, 2>), grid, dim3(256), 0, st, a); break;                   \
      default: hipLaunchKernelGGL((bn_bwd_kernel<TT, PP, AA, 3>), grid, dim3(256), 0, st, a); break;                  \
    }                                                                                                                 \
  } while (0)
#define LAUNCH_BNB(TT)                                                                                                \
  do {                                                                                                                \
    if (g2) { if (apply) LAUNCH_BNB3(TT, true, true); else LAUNCH_BNB3(TT, true, false); }                            \
    else    { if (apply) LAUNCH_BNB3(TT, false, true); else LAUNCH_BNB3(TT, false, false); }                          \
  } while (0)
  if (dtype == CVCS_F32) LAUNCH_BNB(float); else LAUNCH_BNB(bf16_t);
#undef LAUNCH_BNB3
#undef LAUNCH_BNB
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_bn_bwd_reduce(const void* y, int64_t y_ld, const void* g1, int64_t g1_ld, const void* g2, int64_t g2_ld,
                                  int B, int H, int W, int C, const float* scale, const float* shift, const float* save_mean,
                                  const float* save_invstd, int mode, float* part_dz, float* part_dzx, int dtype, void* stream) {
  return bn_bwd_common("cvcs_bn_bwd_reduce", false, y, y_ld, g1, g1_ld, g2, g2_ld, B, H, W, C, scale, shift, save_mean,
                       save_invstd, nullptr, nullptr, mode, nullptr, 0, part_dz, part_dzx, dtype, stream);
}

extern "C" int cvcs_bn_bwd_finalize(const float* part_dz, const float* part_dzx, int rows, int64_t M, int C, const float* gamma,
                                    const float* save_invstd, float* dgamma, float* dbeta, float* coef_a, float* coef_b,
                                    void* stream) {
  CVCS_CHECK_ARG(part_dz && part_dzx && dgamma && dbeta && coef_a && coef_b && C > 0 && M > 0, "cvcs_bn_bwd_finalize: null argument");
  CVCS_CHECK_ARG(rows >= 1 && rows <= 4096, "cvcs_bn_bwd_finalize: rows=%d out of [1,4096]", rows);
  CVCS_CHECK_ARG(C % 16 == 0, "cvcs_bn_bwd_finalize: C %% 16 != 0");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)(C / 16)), dim3(16 * kRL), 0, (hipStream_t)stream, part_dz, part_dzx,
                     rows, M, C, gamma, save_invstd, dgamma, dbeta, coef_a, coef_b);
  CVCS_CHECK_LAUNCH("cvcs_bn_bwd_finalize");
  return CVCS_OK;
}

extern "C" int cvcs_bn_bwd_apply(const void* y, int64_t y_ld, const void* g1, int64_t g1_ld, const void* g2, int64_t g2_ld,
                                 int B, int H, int W, int C, const float* scale, const float* shift, const float* save_mean,
                                 const float* save_invstd, const float* coef_a, const float* coef_b, int mode, void* dy,
                                 int64_t dy_ld, float* part_db, int dtype, void* stream) {
  return bn_bwd_common("cvcs_bn_bwd_apply", true, y, y_ld, g1, g1_ld, g2, g2_ld, B, H, W, C, scale, shift, save_mean,
                       save_invstd, coef_a, coef_b, mode, dy, dy_ld, part_db, nullptr, dtype, stream);
}

extern "C" int cvcs_bn_bwd_apply_q8(const void* y, int64_t y_ld, const void* g1, int64_t g1_ld, int B, int H, int W, int C, const float* scale,
                                    const float* shift, const float* save_mean, const float* save_invstd, const float* coef_a,
                                    const float* coef_b, int mode, void* dy, int64_t dy_ld, float* part_db, void* q8, int64_t q8_ld, int fmt,
                                    float* slot, int take_amax, int dtype, void* stream) {
  CVCS_CHECK_ARG(q8 != nullptr, "cvcs_bn_bwd_apply_q8: null fp8 output");
  return bn_bwd_common("cvcs_bn_bwd_apply_q8", true, y, y_ld, g1, g1_ld, nullptr, 0, B, H, W, C, scale, shift, save_mean, save_invstd, coef_a,
                       coef_b, mode, dy, dy_ld, part_db, nullptr, dtype, stream, Q8Out{(char*)q8, q8_ld, slot, fmt, take_amax});
}

extern "C" int cvcs_colsum_finalize(const float* part, int rows, int C, float* out, void* stream) {
  CVCS_CHECK_ARG(part && out && rows > 0 && C > 0, "cvcs_colsum_finalize: bad argument");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)cdiv(C, 16)), dim3(16 * kRL), 0, (hipStream_t)stream, part, rows, C, out);
  CVCS_CHECK_LAUNCH("cvcs_colsum_finalize");
  return CVCS_OK;
}

extern "C" int cvcs_colsum_partial(const void* x, int64_t x_ld, int64_t M, int C, float* part, int dtype, void* stream) {
  CVCS_CHECK_ARG(DT_OK(dtype), "cvcs_colsum_partial: bad dtype");
  const int es = dtype == CVCS_F32 ? 4 : 2;
  const int V = 16 / es;
  CVCS_CHECK_ARG(M > 0 && C > 0 && C % V == 0 && part, "cvcs_colsum_partial: bad shape");
  const int CC = C / V;
  int rc;
  if ((rc = check_view("cvcs_colsum_partial", x, x_ld, C, es))) return rc;
  const int ccw = CC < 256 ? CC : 256;
  dim3 grid((unsigned)cvcs_bn_bwd_rows(M), (unsigned)cdiv(CC, ccw));
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, M, C, part);
  else
    hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)x, x_ld, M, C, part);
  CVCS_CHECK_LAUNCH("cvcs_colsum_partial");
  return CVCS_OK;
}

static int upsample2x_fwd_impl(const char* fn, const void* in, int64_t in_ld, int B, int H, int W, int C, void* out, int64_t out_ld,
                               int dtype, void* stream, Q8Out q8) {
  CVCS_CHECK_ARG(DT_OK(dtype), "%s: bad dtype", fn);
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0, "%s: bad shape", fn);
  int rc;
  if ((rc = check_view(fn, in, in_ld, C, es))) return rc;
  if ((rc = check_view(fn, out, out_ld, C, es))) return rc;
  if ((rc = check_q8(fn, q8, C, dtype))) return rc;
  const int64_t total = (int64_t)B * 4 * H * W * (C / (16 / es));
  dim3 grid(grid_for(total, 256, 256 * 32));
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((upsample2x_fwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)in, in_ld, B, H, W, C, (char*)out, out_ld, q8);
  else
    hipLaunchKernelGGL((upsample2x_fwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)in, in_ld, B, H, W, C, (char*)out, out_ld, q8);
  CVCS_CHECK_LAUNCH(fn);
  return CVCS_OK;
}

extern "C" int cvcs_upsample2x_fwd(const void* in, int64_t in_ld, int B, int H, int W, int C, void* out, int64_t out_ld,
                                   int dtype, void* stream) {
  return upsample2x_fwd_impl("cvcs_upsample2x_fwd", in, in_ld, B, H, W, C, out, out_ld, dtype, stream, Q8Out{nullptr, 0, nullptr, 0, 0});
}

extern "C" int cvcs_upsample2x_fwd_q8(const void* in, int64_t in_ld, int B, int H, int W, int C, void* out, int64_t out_ld, void* q8,
                                      int64_t q8_ld, int fmt, float* slot, int take_amax, int dtype, void* stream) {
  CVCS_CHECK_ARG(q8 != nullptr, "cvcs_upsample2x_fwd_q8: null fp8 output");
  return upsample2x_fwd_impl("cvcs_upsample2x_fwd_q8", in, in_ld, B, H, W, C, out, out_ld, dtype, stream, Q8Out{(char*)q8, q8_ld, slot, fmt, take_amax});
}

extern "C" int cvcs_upsample2x_bwd(const void* gout, int64_t gout_ld, int B, int H, int W, int C, void* gin, int64_t gin_ld,
                                   int dtype, void* stream) {
  CVCS_CHECK_ARG(DT_OK(dtype), "cvcs_upsample2x_bwd: bad dtype");
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % (16 / es) == 0, "cvcs_upsample2x_bwd: bad shape");
  int rc;
  if ((rc = check_view("cvcs_upsample2x_bwd", gout, gout_ld, C, es))) return rc;
  if ((rc = check_view("cvcs_upsample2x_bwd", gin, gin_ld, C, es))) return rc;
  const int64_t total = (int64_t)B * ((H + kUpStrip - 1) / kUpStrip) * W * (C / (16 / es));      // one thread per (strip of rows, column, channel chunk)
  dim3 grid(grid_for(total, 256, 256 * 32));
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((upsample2x_bwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)gout, gout_ld, B, H, W, C, (char*)gin, gin_ld);
  else
    hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)gout, gout_ld, B, H, W, C, (char*)gin, gin_ld);
  CVCS_CHECK_LAUNCH("cvcs_upsample2x_bwd");
  return CVCS_OK;
}

extern "C" int cvcs_pack_input(const void* src, int src_is_u8, int B, int C, int H, int W, void* dst, int Cpad, int dtype,
                               void* stream) {
  CVCS_CHECK_ARG(DT_OK(dtype), "cvcs_pack_input: bad dtype");
  const int es = dtype == CVCS_F32 ? 4 : 2;
  CVCS_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && Cpad % (16 / es) == 0, "cvcs_pack_input: bad shape");
  CVCS_CHECK_ARG(((uintptr_t)dst % 16) == 0, "cvcs_pack_input: dst alignment");
  dim3 grid(grid_for((int64_t)B * H * W, 256, 256 * 32));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVCS_F32) {
    if (src_is_u8) hipLaunchKernelGGL((pack_input_kernel<float, uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)src, B, C, H, W, (char*)dst, Cpad);
    else hipLaunchKernelGGL((pack_input_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, B, C, H, W, (char*)dst, Cpad);
  } else {
    if (src_is_u8) hipLaunchKernelGGL((pack_input_kernel<bf16_t, uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)src, B, C, H, W, (char*)dst, Cpad);
    else hipLaunchKernelGGL((pack_input_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const float*)src, B, C, H, W, (char*)dst, Cpad);
  }
  CVCS_CHECK_LAUNCH("cvcs_pack_input");
  return CVCS_OK;
}

extern "C" int cvcs_pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, int Cin_pad, void* w_fwd, void* w_dgrad,
                                     int dtype, void* stream) {
  CVCS_CHECK_ARG(DT_OK(dtype), "cvcs_pack_conv_weight: bad dtype");
  CVCS_CHECK_ARG(w && w_fwd && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && Cin_pad >= Cin, "cvcs_pack_conv_weight: bad argument");
  const int64_t total = (int64_t)KH * KW * Cout * (Cin_pad + (w_dgrad ? Cin : 0));
  dim3 grid(grid_for(total));
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((pack_conv_weight_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, KH, KW, Cin_pad, (float*)w_fwd, (float*)w_dgrad);
  else
    hipLaunchKernelGGL((pack_conv_weight_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, KH, KW, Cin_pad, (bf16_t*)w_fwd, (bf16_t*)w_dgrad);
  CVCS_CHECK_LAUNCH("cvcs_pack_conv_weight");
  return CVCS_OK;
}

extern "C" int cvcs_pack_conv_weights(const cvcs_pack_item* items_device, int n_items, int dtype, void* stream) {
  CVCS_CHECK_ARG(DT_OK(dtype), "cvcs_pack_conv_weights: bad dtype");
  CVCS_CHECK_ARG(items_device && n_items > 0 && n_items <= 65535, "cvcs_pack_conv_weights: bad table");
  dim3 grid(128, (unsigned)n_items);   // entries must have Cout %% 16 == 0 and at most 9 taps (checked where the table is built)
  if (dtype == CVCS_F32) hipLaunchKernelGGL((pack_conv_weights_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, items_device);
  else hipLaunchKernelGGL((pack_conv_weights_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, items_device);
  CVCS_CHECK_LAUNCH("cvcs_pack_conv_weights");
  return CVCS_OK;
}

extern "C" int cvcs_pack_convT_weight(const float* w, int Cin, int Cout, void* w_fwd, float* bias4, const float* bias,
                                      void* w_dgrad, int dtype, void* stream) {
  CVCS_CHECK_ARG(DT_OK(dtype), "cvcs_pack_convT_weight: bad dtype");
  CVCS_CHECK_ARG(w && w_fwd && bias4 && Cin > 0 && Cout > 0, "cvcs_pack_convT_weight: bad argument");
  const int64_t total = (int64_t)8 * Cout * Cin + 4 * Cout;
  dim3 grid(grid_for(total));
  if (dtype == CVCS_F32)
    hipLaunchKernelGGL((pack_convT_weight_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, (float*)w_fwd, bias4, bias, (float*)w_dgrad);
  else
    hipLaunchKernelGGL((pack_convT_weight_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, (bf16_t*)w_fwd, bias4, bias, (bf16_t*)w_dgrad);
  CVCS_CHECK_LAUNCH("cvcs_pack_convT_weight");
  return CVCS_OK;
}
