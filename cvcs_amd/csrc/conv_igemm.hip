// Implicit-GEMM convolution for gfx950: out[M pixels][Cout] = gather(in)[M][KH*KW*Cin] x wt[Cout][KH*KW*Cin]^T.
//
// GEMM view: M = B*Ho*Wo output pixels (rows), N = Cout, K = taps x Cin.  Both operands are K-contiguous in memory
// (NHWC activations: channels of one pixel; weights packed [tap][Cout][Cin]), so every MFMA fragment is one
// 16-byte LDS read.  A K-step is one filter tap x one 64-byte channel slice (32 bf16 / 16 f32 channels): the A
// tile of a step is a *gather* of 128 shifted pixels (zero for the padding ring), never an im2col buffer.
//
// Workgroup: 256 threads = 4 waves as 2(M) x 2(N); tile 128 pixels x BN channels (BN = 64 | 128); wave tile
// 64 x BN/2 as 4 x (BN/32) MFMA 16x16 accumulators.  Two LDS stages; the global loads of step k+1 are issued
// before the MFMAs of step k and written to LDS after them (issue-early / write-late), one barrier per step.
// LDS rows are 64 B; 16-byte chunk c of row r is stored at chunk c ^ (((r>>2)&1)<<1), which makes both the
// ds_write_b128 (8 lanes = 2 rows) and the ds_read_b128 fragment reads (16 rows x one chunk) conflict-free.
//
// Epilogue (fused): + bias, optional ReLU, per-64-row-block per-channel (sum, centred M2) for BatchNorm, cast,
// transpose through LDS and write whole pixel rows (BN*sizeof(T) contiguous bytes), optionally pixel-shuffled
// (ConvTranspose2d k2 s2 == 1x1 conv to 4*Cr columns + shuffle).
//
// Reference ops replaced: S/blocks.py:3-4 (conv3x3 via :13,:40,:43), S/nets.py:150,156,162,168 (ConvTranspose2d)
// and the data-gradient of both (convolution_backward, S/train.py:125) through pre-flipped weights.
#include "common.h"

namespace cvcs {

struct ConvArgs {
  const char* in;
  const char* wt;
  const float* bias;
  char* out;
  float* stat_sum;
  float* stat_m2;
  int64_t in_ld, out_ld;  // in elements
  int B, H, W, Cin, Ho, Wo, Cout;
  int KH, KW, stride, pad, dil;
  int relu, pixel_shuffle;
  int M;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // exact-f32 MFMA (v_mfma_f32_16x16x4_f32): lane (r = l&15, g = l>>4) feeds k = g; with 16-byte fragments lane g
  // holds channels 4g..4g+3, so pass j contracts channels {4g+j}: the four passes cover the 16-channel slice.
  __device__ static __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

constexpr int kBM = 128;

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ (((row >> 2) & 1) << 1); }

template <typename T, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
  constexpr int ES = sizeof(T);
  constexpr int KG = 64 / ES;      // channels per K-step
  constexpr int WN = BN / 2;       // wave tile N
  constexpr int NREP = WN / 16;
  constexpr int MREP = 4;
  constexpr int A_BYTES = kBM * 64;
  constexpr int B_BYTES = BN * 64;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int BROWS = BN / 64;   // B rows per loader thread
  constexpr int OROW = BN * ES + 16;  // epilogue LDS row pitch (bytes)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * kBM;
  const int n0 = blockIdx.y * BN;
  const int lr = tid >> 2, lc = tid & 3;

  // ---- loader rows: A rows lr, lr+64 ; B rows lr + 64*j
  int iy0[2], ix0[2];
  int64_t pbase[2];
  bool mvalid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + lr + 64 * i;
    mvalid[i] = m < p.M;
    int mm = mvalid[i] ? m : 0;
    int ox = mm % p.Wo;
    int t = mm / p.Wo;
    int oy = t % p.Ho;
    int b = t / p.Ho;
    iy0[i] = oy * p.stride - p.pad;
    ix0[i] = ox * p.stride - p.pad;
    pbase[i] = (int64_t)b * p.H * p.W;
  }
  const int taps = p.KH * p.KW;
  const int nslice = p.Cin / KG;
  const int nsteps = taps * nslice;
  const int64_t wt_tap_stride = (int64_t)p.Cout * p.Cin;  // elements

  uint4 ra[2], rb[BROWS];
  auto load_regs = [&](int tap, int cs) {
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int iy = iy0[i] + kh * p.dil, ix = ix0[i] + kw * p.dil;
      bool ok = mvalid[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      ra[i] = make_uint4(0, 0, 0, 0);
      if (ok) {
        const char* src = p.in + ((pbase[i] + (int64_t)iy * p.W + ix) * p.in_ld + (int64_t)cs * KG) * ES + lc * 16;
        ra[i] = *reinterpret_cast<const uint4*>(src);
      }
    }
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      int n = n0 + lr + 64 * j;
      const char* src = p.wt + ((int64_t)tap * wt_tap_stride + (int64_t)n * p.Cin + (int64_t)cs * KG) * ES + lc * 16;
      rb[j] = *reinterpret_cast<const uint4*>(src);
    }
  };
  auto write_lds = [&](int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int r = lr + 64 * i;
      *reinterpret_cast<uint4*>(sa + r * 64 + swz(r, lc) * 16) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      int r = lr + 64 * j;
      *reinterpret_cast<uint4*>(sb + r * 64 + swz(r, lc) * 16) = rb[j];
    }
  };

  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int i = 0; i < MREP; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int buf) {
    const char* sa = smem + buf * STAGE;
    const char* sb = sa + A_BYTES;
    uint4 af[MREP], bfr[NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
      int r = wm * 64 + i * 16 + fr;
      af[i] = *reinterpret_cast<const uint4*>(sa + r * 64 + swz(r, fg) * 16);
    }
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      int r = wn * WN + j * 16 + fr;
      bfr[j] = *reinterpret_cast<const uint4*>(sb + r * 64 + swz(r, fg) * 16);
    }
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
      for (int j = 0; j < NREP; ++j) acc[i][j] = Mma<T>::run(af[i], bfr[j], acc[i][j]);
  };

  // ---- main loop: slice-major, tap-minor (neighbouring taps re-touch the same lines in L1/L2)
  int tap = 0, cs = 0;
  load_regs(0, 0);
  write_lds(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    int ntap = tap + 1, ncs = cs;
    if (ntap == taps) { ntap = 0; ncs = cs + 1; }
    const bool more = (s + 1 < nsteps);
    if (more) load_regs(ntap, ncs);
    compute(s & 1);
    if (more) write_lds((s + 1) & 1);
    __syncthreads();
    tap = ntap; cs = ncs;
  }

  // ---- epilogue: bias / ReLU
#pragma unroll
  for (int j = 0; j < NREP; ++j) {
    const int n = n0 + wn * WN + j * 16 + fr;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv;
        if (p.relu) v = fmaxf(v, 0.f);
        acc[i][j][r] = v;
      }
  }
  // ---- BatchNorm partial statistics of this wave's 64 rows (row = wm*64 + i*16 + fg*4 + r)
  if (p.stat_sum) {
    const int rbase = m0 + wm * 64;
    int nvalid = p.M - rbase;
    nvalid = nvalid < 0 ? 0 : (nvalid > 64 ? 64 : nvalid);
    const float inv = nvalid > 0 ? 1.f / (float)nvalid : 0.f;
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (i * 16 + fg * 4 + r < nvalid) s += acc[i][j][r];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mean = s * inv;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (i * 16 + fg * 4 + r < nvalid) {
            float d = acc[i][j][r] - mean;
            q += d * d;
          }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (fg == 0) {
        const int n = n0 + wn * WN + j * 16 + fr;
        const int64_t row = (int64_t)blockIdx.x * 2 + wm;
        p.stat_sum[row * p.Cout + n] = s;
        p.stat_m2[row * p.Cout + n] = q;
      }
    }
  }
  // ---- transpose through LDS, store whole pixel rows.  (The trailing barrier of the main loop already
  //      separates the last fragment reads from these writes.)
#pragma unroll
  for (int i = 0; i < MREP; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = wm * 64 + i * 16 + fg * 4 + r;
        int col = wn * WN + j * 16 + fr;
        Elem<T>::st(reinterpret_cast<T*>(smem + row * OROW) + col, acc[i][j][r]);
      }
  __syncthreads();
  constexpr int CPR = BN * ES / 16;  // 16-byte chunks per row
  int Cr = 0, q = 0, co0 = n0;
  if (p.pixel_shuffle) {
    Cr = p.Cout >> 2;
    q = n0 / Cr;
    co0 = n0 - q * Cr;
  }
  for (int id = tid; id < kBM * CPR; id += 256) {
    int row = id / CPR, c = id - row * CPR;
    int m = m0 + row;
    if (m >= p.M) continue;
    uint4 v = *reinterpret_cast<const uint4*>(smem + row * OROW + c * 16);
    int64_t pix = m;
    if (p.pixel_shuffle) {
      int ox = m % p.Wo;
      int t = m / p.Wo;
      int oy = t % p.Ho;
      int b = t / p.Ho;
      pix = ((int64_t)b * (2 * p.Ho) + 2 * oy + (q >> 1)) * (2 * p.Wo) + 2 * ox + (q & 1);
    }
    char* dst = p.out + (pix * p.out_ld + co0) * ES + c * 16;
    *reinterpret_cast<uint4*>(dst) = v;
  }
}

template <typename T, int BN>
static int launch(const ConvArgs& a, hipStream_t st) {
  constexpr int ES = sizeof(T);
  size_t stage = 2 * (size_t)(kBM * 64 + BN * 64);
  size_t epi = (size_t)kBM * (BN * ES + 16);
  size_t lds = stage > epi ? stage : epi;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, BN>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid((unsigned)cdiv(a.M, kBM), (unsigned)(a.Cout / BN));
  hipLaunchKernelGGL((conv_igemm_kernel<T, BN>), grid, dim3(256), lds, st, a);
  CVCS_CHECK_LAUNCH("cvcs_conv2d");
  return CVCS_OK;
}

}  // namespace cvcs

using namespace cvcs;

extern "C" int cvcs_conv_stat_rows(int64_t M) { return (int)(cdiv(M, kBM) * 2); }

extern "C" int cvcs_conv2d(const cvcs_conv_desc* d, void* stream) {
  CVCS_CHECK_ARG(d != nullptr, "cvcs_conv2d: null descriptor");
  CVCS_CHECK_ARG(d->dtype == CVCS_F32 || d->dtype == CVCS_BF16, "cvcs_conv2d: bad dtype %d", d->dtype);
  const int es = d->dtype == CVCS_F32 ? 4 : 2;
  const int kg = 64 / es;
  CVCS_CHECK_ARG(d->in && d->wt && d->out, "cvcs_conv2d: null tensor");
  CVCS_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0, "cvcs_conv2d: bad shape");
  CVCS_CHECK_ARG(d->Cin > 0 && d->Cin % kg == 0, "cvcs_conv2d: Cin=%d must be a multiple of %d", d->Cin, kg);
  CVCS_CHECK_ARG(d->Cout > 0 && d->Cout % 64 == 0, "cvcs_conv2d: Cout=%d must be a multiple of 64", d->Cout);
  CVCS_CHECK_ARG(d->KH >= 1 && d->KW >= 1 && d->stride >= 1 && d->dil >= 1 && d->pad >= 0, "cvcs_conv2d: bad filter geometry");
  // output extent must match the filter geometry (every gathered pixel is range-checked in the kernel, but the
  // host refuses inconsistent shapes so that no tile silently reads the wrong window)
  const int eh = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  const int ew = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  CVCS_CHECK_ARG(eh == d->Ho && ew == d->Wo, "cvcs_conv2d: Ho,Wo=%d,%d but geometry gives %d,%d", d->Ho, d->Wo, eh, ew);
  CVCS_CHECK_ARG(d->in_ld >= d->Cin && d->in_ld * es % 16 == 0, "cvcs_conv2d: in_ld");
  const int cout_store = d->pixel_shuffle ? d->Cout / 4 : d->Cout;
  CVCS_CHECK_ARG(d->out_ld >= cout_store && d->out_ld * es % 16 == 0, "cvcs_conv2d: out_ld");
  CVCS_CHECK_ARG(((uintptr_t)d->in % 16) == 0 && ((uintptr_t)d->out % 16) == 0 && ((uintptr_t)d->wt % 16) == 0,
                 "cvcs_conv2d: tensors must be 16-byte aligned");
  const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
  CVCS_CHECK_ARG(M < (1ll << 31) - kBM, "cvcs_conv2d: too many output pixels");
  CVCS_CHECK_ARG((d->stat_sum == nullptr) == (d->stat_m2 == nullptr), "cvcs_conv2d: stat_sum/stat_m2 go together");
  int bn = (d->Cout % 128 == 0) ? 128 : 64;
  if (d->pixel_shuffle) {
    CVCS_CHECK_ARG(d->Cout % 4 == 0 && (d->Cout / 4) % 64 == 0, "cvcs_conv2d: pixel_shuffle needs Cout/4 %% 64 == 0");
    if ((d->Cout / 4) % 128 != 0) bn = 64;  // a column tile must not straddle two (dy,dx) groups
  }
  ConvArgs a;
  a.in = (const char*)d->in; a.wt = (const char*)d->wt; a.bias = d->bias; a.out = (char*)d->out;
  a.stat_sum = d->stat_sum; a.stat_m2 = d->stat_m2;
  a.in_ld = d->in_ld; a.out_ld = d->out_ld;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  a.relu = d->relu; a.pixel_shuffle = d->pixel_shuffle; a.M = (int)M;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == CVCS_F32) return bn == 128 ? launch<float, 128>(a, st) : launch<float, 64>(a, st);
  return bn == 128 ? launch<bf16_t, 128>(a, st) : launch<bf16_t, 64>(a, st);
}
