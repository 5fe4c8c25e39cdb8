/* cvcs_hip.h - C-ABI of libcvcs_hip.so: the MI355X (gfx950) replacement for the torch/ATen ops on the
 * per-tile segmentation hot path of theElandor/CVCS.
 *
 * The reference has no FFI of its own: its seam is the Python factories `load_network`, `load_loss`,
 * `load_optimizer` (source/scripts/utils.py:174-242) and the nn.Module contract (source/scripts/nets.py:12-33).
 * Each entry point below replaces the ATen op(s) one of those call sites issues; the citation after "replaces:"
 * is the reference line that issues it (S/ = source/scripts/).
 *
 * Conventions
 *   - every function returns 0 on success, a negative CVCS_E* code otherwise; cvcs_last_error() gives the
 *     message of the calling thread's last failure.
 *   - all pointers are DEVICE pointers unless the name ends in _host; buffers are caller-owned.
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and allocates nothing.
 *   - activations inside the library are NHWC ("pixel-major") so that the contraction axis of every MFMA
 *     operand is contiguous; the NCHW u8/f32 tensors of the reference appear only at the boundary calls
 *     (cvcs_pack_input, cvcs_head_fwd, cvcs_ce_*, cvcs_argmax_*).
 *   - an activation "view" is (pointer, ld): element (pixel p, channel c) lives at ptr[p*ld + c]; this is how
 *     torch.concat (S/nets.py:185,189,193,197) disappears: producers write into channel ranges of one buffer.
 *   - dtype: CVCS_F32 (parity path, exact-f32 MFMA) or CVCS_BF16 (throughput path, bf16 MFMA, f32 accumulate).
 */
#ifndef CVCS_HIP_H
#define CVCS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVCS_ABI_VERSION 14

enum { CVCS_F32 = 0, CVCS_BF16 = 1 };
enum { CVCS_E4M3 = 0, CVCS_E5M2 = 1 };   /* OCP fp8 formats of the fp8 convolution path (gfx950: e4m3fn / e5m2, not the MI300 fnuz forms) */
enum { CVCS_OK = 0, CVCS_EINVAL = -1, CVCS_EHIP = -2, CVCS_EUNSUPPORTED = -3 };

const char* cvcs_last_error(void);
int cvcs_abi_version(void);
/* sizeof(cvcs_conv_desc) / sizeof(cvcs_wgrad_desc) as this library was compiled: a binding in another language checks its
 * own struct declaration against them when it loads the library (cvcs_amd/_lib.py does)                            */
int cvcs_sizeof_conv_desc(void);
int cvcs_sizeof_wgrad_desc(void);
/* number of split-K slices cvcs_conv2d_wgrad will use; workspace = slices*KH*KW*Cout*Cin floats */
int cvcs_wgrad_slices(int B, int Ho, int Wo, int Cout, int Cin, int KH, int KW, int stride);

/* ---- implicit-GEMM convolution (forward, and data-gradient with pre-flipped weights) ------------------
 * replaces: nn.Conv2d k3 p1 (S/blocks.py:3-4 via :13,:40,:43), nn.ConvTranspose2d k2 s2 (S/nets.py:150,156,162,168;
 * pixel_shuffle=1), and the input-gradient half of convolution_backward for both (S/train.py:125).
 *   out[p, n] = act( bias[n] + sum_{kh,kw,c} in[pix(p,kh,kw), c] * wt[kh*KW+kw][n][c] )
 * M = B*Ho*Wo output pixels; wt is [KH*KW][Cout][Cin] in `dtype`; Cin % (64/sizeof(dtype)) == 0, Cout % 32 == 0 (a Cout that
 * is no multiple of 64 - the 96 / 288 widths of Swin-T's first stage - runs on the generic kernel with a masked last tile).
 * stat_sum/stat_m2 (optional, f32 [rows][Cout]) and stat_cnt (f32 [rows]), rows = cvcs_conv_stat_rows(desc): per
 * wave-sized block of output pixels and channel, the pixel count, the sum and the second moment centred on the
 * block's own mean of the values written - the BatchNorm batch statistics, fused into the epilogue.
 * 3x3 / stride 1 / pad 1 convolutions run a specialised kernel that stages each 18x18 input halo once for all
 * nine taps; every other geometry runs the generic gather kernel.                                          */
typedef struct {
  const void* in;   int64_t in_ld;  int32_t B, H, W, Cin;
  const void* wt;   const float* bias;
  void* out;        int64_t out_ld; int32_t Ho, Wo, Cout;
  int32_t KH, KW, stride, pad, dil;
  int32_t relu;           /* 1: ReLU in the epilogue (decoder order conv->ReLU->BN, S/blocks.py:40-45) */
  int32_t pixel_shuffle;  /* 1: Cout = 4*Cr, column (dy*2+dx)*Cr+co goes to pixel (2y+dy, 2x+dx), channel co */
  float* stat_sum;  float* stat_m2;  float* stat_cnt;
  int32_t dtype;
  /* eval-mode BatchNorm folded into the epilogue (f32 [Cout] each, NULL = none): v = acc + bias;
   * v = v*pre_scale + pre_shift; ReLU if `relu`; v = v*post_scale + post_shift.  "pre" is the encoder order
   * conv->BN->ReLU (S/blocks.py:13-17), "post" the decoder order conv->ReLU->BN (S/blocks.py:40-45); scale/shift come
   * from cvcs_bn_finalize(train=0).                                                                              */
  const float* pre_scale;   const float* pre_shift;
  const float* post_scale;  const float* post_shift;
  /* 0, or the number of channels the `in` rows really hold (a multiple of 16 bytes, <= Cin): channels Cin_valid..Cin-1
   * read as zero without being fetched.  The first layer's 3-channel tile is stored 16 bytes per pixel while the
   * contraction runs over one zero-padded K-group (`wt` is [KH*KW][Cout][Cin]).                                  */
  int32_t Cin_valid;
  /* optional fused nn.MaxPool2d(2,2) (S/nets.py:130,135,140,145) of the values written to `out` (after bias / folded
   * BatchNorm / ReLU): pool_out [B, Ho/2, Wo/2, Cout] NHWC in `dtype`, leading dimension pool_ld elements; NULL = none.
   * Eval mode only needs it (in train mode the pooling is part of the BatchNorm-apply pass); 3x3 / stride 1 / pad 1
   * convolutions on maps of at least 8 pixels with even Ho, Wo.                                                   */
  void* pool_out;   int64_t pool_ld;
  /* optional fused FIRST pass of the BatchNorm backward that consumes this launch's output (cvcs_bn_bwd_reduce without its
   * own sweep over the tensors; bf16 3x3 / stride 1 / pad 1 launches, i.e. the data gradient of the conv ABOVE a
   * conv->BN->ReLU (mode 0) or conv->ReLU->BN (mode 1) block, S/blocks.py:13-17 / :40-45).  `out` is the gradient g w.r.t.
   * that block's output; bwd_y [B, Ho, Wo, Cout] is the block's saved conv output (mode 1: after the ReLU), leading
   * dimension bwd_y_ld; the four per-channel vectors are those of cvcs_bn_bwd_reduce.  Written: one partial row per
   * 16x16 output tile (cvcs_conv_stat_rows rows x Cout floats each) of sum(dz) and sum(dz * xhat) taken over the values
   * as STORED (bf16), which cvcs_colsum_partial + cvcs_bn_bwd_finalize reduce.  NULL bwd_y = none.                 */
  const void* bwd_y;   int64_t bwd_y_ld;
  const float* bwd_scale;  const float* bwd_shift;  const float* bwd_mean;  const float* bwd_invstd;
  int32_t bwd_mode;
  float* bwd_part_dz;  float* bwd_part_dzx;
  /* ---- ABI 4: anisotropic geometry and explicit input pitches (generic gather kernel only; all 0 = the isotropic,
   * densely packed case above).  aniso = 1: the W axis uses stride_w / pad_w instead of stride / pad (H keeps stride / pad).
   * in_row_pitch / in_img_pitch (elements, 0 = W*in_ld / H*row pitch): pixel (b, y, x) of `in` lives at
   * b*in_img_pitch + y*in_row_pitch + x*in_ld, and in_ld may then be SMALLER than Cin - consecutive "pixels" overlap.
   * This is how the 7x7 / stride 2 stem of the ResNet encoders runs without an im2col buffer: the 3-channel tile is stored
   * 4 channels per pixel with a 3-pixel left margin, a "virtual pixel" is the 64-byte window of 8 real pixels starting at
   * real column 2*xo - 3 (in_ld = 8 elements = two real pixels), and the stem is a KH = 7, KW = 1 convolution with
   * stride (2, 1) over 32 virtual channels = (kw, c) pairs - seven K-steps instead of 49.                          */
  int32_t aniso, stride_w, pad_w;
  int64_t in_row_pitch, in_img_pitch;
  /* ---- ABI 13 (bf16 1x1 / stride 1 launches; all NULL / 0 = none) --------------------------------------------------------
   * res: a residual tile ADDED between the pre-affine and the ReLU - out = relu?(pre_scale*acc + pre_shift + r), r = res[p, n]
   * or res_scale[n]*res[p, n] + res_shift[n] (the projection shortcut's own BatchNorm).  With pre_scale / pre_shift from
   * cvcs_bn_gram_finalize this is the whole tail of a bottleneck block - conv3 -> BatchNorm -> + shortcut -> ReLU - in the conv's
   * epilogue: the conv output itself is never stored (torchvision Bottleneck.forward as used by S/nets.py:234-311).
   * in2: a SECOND contraction source of the same pixels - out[p, n] = sum_c in[p, c] wt[n][c] + sum_c in2[p, c] wt[n][Cin + c],
   * `wt` rows Cin + Cin2 long (the data gradient of such a block, see cvcs_bn_gram_bwd).                                    */
  const void* res;   int64_t res_ld;
  const float* res_scale;  const float* res_shift;
  const void* in2;   int64_t in2_ld;  int32_t Cin2;
  /* mask: out = (mask[p, n] > 0) ? value : 0 after everything else - the ReLU backward of the activation this gradient belongs to.  With
   * res = the gradient arriving over the identity shortcut, the data-gradient launch of a block's first 1x1 conv writes the PREVIOUS block's
   * dz = (g_conv1 + g_shortcut) * (out > 0) directly (cvcs_relu_bwd_sum without its pass over three tensors).                  */
  const void* mask;  int64_t mask_ld;
  /* res2: a second tile added beside res (no affine); res2_half = 1: it lives at HALF the resolution and contributes at the even pixels only -
   * the data gradient of a 1x1 / stride-2 projection shortcut, so that the block below a down-sampling block gets its dz from ONE launch too */
  const void* res2;  int64_t res2_ld;  int32_t res2_half;
  /* (ABI 14) the ReLU mask as BITS, one byte per 16-byte chunk of a pixel row ([pixels][Cout / 8], bit e = channel 8 * chunk + e):
   * mask_bits_out - a forward launch that applies a ReLU also writes (value > 0) of what it stores; mask_bits - a data-gradient launch reads the
   * mask from there instead of `mask` (the block output itself: 16 x the bytes).  bf16 1x1 launches of the taps kernel (Cout % 128 == 0).   */
  void* mask_bits_out;  const void* mask_bits;
  /* (ABI 14, opt-in) in_up2 = 1: `in` is the map at HALF the resolution ([B][H/2][W/2], H and W of this descriptor being those of the conv's input)
   * and the convolution runs on its bilinear x2 up-sampling (nn.Upsample(2, 'bilinear'), S/blocks.py:29), interpolated while the input tile is
   * staged - bit-identical to cvcs_upsample2x_fwd followed by this conv, without the up-sampled tensor.  Thin bf16 3x3 32 -> 16 instance.      */
  int32_t in_up2;
} cvcs_conv_desc;
int cvcs_conv2d(const cvcs_conv_desc* d, void* stream);
/* number of partial-statistics rows cvcs_conv2d writes for this descriptor */
int cvcs_conv_stat_rows(const cvcs_conv_desc* d);

/* ---- weight gradient ---------------------------------------------------------------------------------
 * replaces: the weight half of convolution_backward (S/train.py:125).
 *   dW[n][c][kh][kw] = sum_p dy[p, n] * x[pix(p,kh,kw), c]     (p over B*Ho*Wo; pix = p*stride - pad + (kh,kw))
 * `dw` is f32 in the reference's parameter layout OIHW [Cout][Cin_real][KH][KW]; Cin_real <= Cin lets the
 * zero-padded first layer drop its padding.  A ConvTranspose2d(k2,s2) weight [Cin_T][Cout_T][2][2] is the same
 * call with dy := the layer's input, x := the gradient of its output, KH=KW=2, stride=2, pad=0.
 * Filters built: 3x3, 2x2, 1x1 (and 7x1, see `aniso`).  workspace: cvcs_wgrad_slices(...) * KH*KW*Cout*Cin floats.               */
typedef struct {
  const void* x;    int64_t x_ld;   int32_t B, H, W, Cin;
  const void* dy;   int64_t dy_ld;  int32_t Ho, Wo, Cout;
  int32_t KH, KW, stride, pad;
  float* dw;        int32_t Cin_real;
  float* workspace;
  int32_t dtype;
  /* ABI 4: as in cvcs_conv_desc (generic kernel only): separate W stride / padding, explicit pitches of `x`; filters
   * 7x1 are built for this case (the stem's weight gradient over virtual channels).                              */
  int32_t aniso, stride_w, pad_w;
  int64_t x_row_pitch, x_img_pitch;
  /* ABI 5: dilation (0 / 1 = none).  dil > 1 is built for 3x3 / stride 1 / pad = dil (DeepLabV3+: ASPP rates, dilated last
   * stage): nine shifted 1x1 weight gradients into one partial slab (workspace: cvcs_wgrad_workspace_floats).       */
  int32_t dil;
  /* ABI 9: optional bias gradient db[n] = sum_p dy[p, n] (f32 [Cout]) out of the SAME launch - the weight-gradient GEMM already streams dy
   * once, a separate column-sum pass would read it again (nn.Linear layers of the Swin encoder).  Built for the descriptors
   * cvcs_wgrad_takes_bias() accepts (bf16, 1x1 / stride 1 on the GEMM kernel); NULL = none.  The workspace grows by slices * Cout floats
   * (cvcs_wgrad_workspace_floats accounts for it when dbias is set).                                                                       */
  float* dbias;
  /* (ABI 14, opt-in) x_up2 = 1: `x` is the map at HALF the resolution and the layer's input its bilinear x2 up-sampling (H, W of this descriptor),
   * interpolated while the tile is staged - the weight half of cvcs_conv_desc.in_up2.  Thin bf16 3x3 instance Cin = 32, Cout = 16.                */
  int32_t x_up2;
} cvcs_wgrad_desc;
int cvcs_conv2d_wgrad(const cvcs_wgrad_desc* d, void* stream);
/* 1 if cvcs_conv2d_wgrad can also produce `dbias` for this descriptor, else 0 */
int cvcs_wgrad_takes_bias(const cvcs_wgrad_desc* d);
/* exact workspace size (floats) of one descriptor; covers the anisotropic / pitched case cvcs_wgrad_slices cannot express */
int64_t cvcs_wgrad_workspace_floats(const cvcs_wgrad_desc* d);

/* ---- BatchNorm2d ---------------------------------------------------------------------------------------
 * replaces: nn.BatchNorm2d (S/blocks.py:14,42,45) forward in train / eval mode and its backward.
 * cvcs_bn_finalize: combine the conv's partial statistics (rows x C) into scale/shift (f32 [C] each), save
 * mean/invstd for backward and update running stats (momentum 0.1, unbiased var).  train=0: scale/shift from the
 * running stats, partials ignored.                                                                        */
int cvcs_bn_finalize_workspace_floats(int rows, int C);   /* 0 when the partial rows are merged directly */
int cvcs_bn_finalize(const float* stat_sum, const float* stat_m2, const float* stat_cnt, int rows, int64_t M, int C,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float momentum, float eps, int train,
                     float* scale, float* shift, float* save_mean, float* save_invstd, float* workspace, void* stream);

/* Cross-rank ("SyncBN") variant, used when tiles of ONE batch are sharded over ranks and the result must equal the
 * reference's single-process batch statistics (S/blocks.py:14 over the whole batch): cvcs_bn_moments writes this rank's
 * per-channel (n, sum x, sum x^2) as f64 [3][C]; the host sums them over ranks (one small all-reduce);
 * cvcs_bn_finalize_moments then does what cvcs_bn_finalize(train=1) does, from the summed moments.             */
int cvcs_bn_moments(const float* stat_sum, const float* stat_m2, const float* stat_cnt, int rows, int C,
                    double* moments, float* workspace, void* stream);
int cvcs_bn_finalize_moments(const double* moments, int C, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps,
                             float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);

/* (out may be NULL when pool is given: pooling-only pass over an already activated tensor, scale = 1, shift = 0)
 * y -> out = relu?(scale*y + shift), optionally also pool[p/2] = max 2x2 (nn.MaxPool2d(2,2), S/nets.py:130,135,140,145).
 * replaces: BN apply + nn.ReLU (S/blocks.py:17) (+ MaxPool2d).  relu=1: encoder order; relu=0: decoder BN apply. */
int cvcs_bn_act(const void* y, int64_t y_ld, int B, int H, int W, int C,
                const float* scale, const float* shift, int relu,
                void* out, int64_t out_ld, void* pool, int64_t pool_ld, int dtype, void* stream);

/* BatchNorm(+ReLU)(+MaxPool) backward, two passes over (y, g):
 *   mode 0 (encoder, a = relu(bn(y))): dz = (g1 + unpool(g2)) * (a > 0);  mode 1 (decoder, z = bn(r), r = relu(conv)):
 *   dz = g1, and the ReLU mask (r > 0) is applied to the result;  mode 2 (no ReLU on either side: the last BatchNorm of a
 *   residual block, whose ReLU comes after the add - cvcs_relu_bwd_sum): dz = g1.
 * pass 1 -> partial sums of dz and dz*xhat ([rows][C] each, rows = cvcs_bn_bwd_rows(B*H*W); or taken by the conv
 *           launch that produced g1 - cvcs_conv_desc.bwd_y - and reduced to at most 1024 rows by cvcs_colsum_partial);
 * cvcs_bn_bwd_finalize -> dgamma, dbeta and the two per-channel coefficients;
 * pass 2 -> dy (gradient w.r.t. the conv output) and partial column sums of dy (conv bias gradient).        */
int cvcs_bn_bwd_rows(int64_t M);
/* channel-chunk lanes per workgroup of the two passes for a C-channel tensor (a divisor of C's 16-byte chunk count, <= 32): any C that is a
 * multiple of the chunk is accepted (96, 160, 224, 480, 672, 960 of MobileNetV3 included); exported so that the rule is testable without a GPU */
int cvcs_bn_bwd_chunk_lanes(int C, int dtype);
int cvcs_bn_bwd_reduce(const void* y, int64_t y_ld, const void* g1, int64_t g1_ld, const void* g2, int64_t g2_ld,
                       int B, int H, int W, int C, const float* scale, const float* shift,
                       const float* save_mean, const float* save_invstd, int mode,
                       float* part_dz, float* part_dzx, int dtype, void* stream);
int cvcs_bn_bwd_finalize(const float* part_dz, const float* part_dzx, int rows, int64_t M, int C,
                         const float* gamma, const float* save_invstd,
                         float* dgamma, float* dbeta, float* coef_a, float* coef_b, void* stream);
/* SyncBN backward: coef_a/coef_b from all-rank sums [2][C] (sum dz, sum dz*xhat; f32) over M = all ranks' pixels */
int cvcs_bn_bwd_coeffs(const float* sums, int64_t M, int C, float* coef_a, float* coef_b, void* stream);
int cvcs_bn_bwd_apply(const void* y, int64_t y_ld, const void* g1, int64_t g1_ld, const void* g2, int64_t g2_ld,
                      int B, int H, int W, int C, const float* scale, const float* shift,
                      const float* save_mean, const float* save_invstd, const float* coef_a, const float* coef_b,
                      int mode, void* dy, int64_t dy_ld, float* part_db, int dtype, void* stream);
/* column-sum of `rows` partial rows: out[c] = sum_r part[r][c] (f64 accumulate) - conv bias gradients */
int cvcs_colsum_finalize(const float* part, int rows, int C, float* out, void* stream);
/* partial per-channel sums of an activation view over its M pixels: part[cvcs_bn_bwd_rows(M)][C]; with
 * cvcs_colsum_finalize this is the bias gradient of a conv that has no BatchNorm behind it
 * (nn.ConvTranspose2d S/nets.py:150, UnetUpscaleLayer's conv S/blocks.py:30).                               */
int cvcs_colsum_partial(const void* x, int64_t x_ld, int64_t M, int C, float* part, int dtype, void* stream);

/* ---- bilinear x2 upsampling (Urnet only) -----------------------------------------------------------------
 * replaces: nn.Upsample(scale_factor=2, mode='bilinear') (align_corners=False) (S/blocks.py:29) fwd / bwd. */
int cvcs_upsample2x_fwd(const void* in, int64_t in_ld, int B, int H, int W, int C, void* out, int64_t out_ld,
                        int dtype, void* stream);
int cvcs_upsample2x_bwd(const void* gout, int64_t gout_ld, int B, int H, int W, int C, void* gin, int64_t gin_ld,
                        int dtype, void* stream);

/* ---- boundary: NCHW <-> internal ----------------------------------------------------------------------------
 * cvcs_pack_input replaces `image.type(torch.float32)` (S/train.py:121): u8 (src_is_u8=1) or f32 NCHW [B,3,H,W]
 * -> NHWC [B,H,W,Cpad] in `dtype`, channels >= 3 zero.                                                    */
int cvcs_pack_input(const void* src, int src_is_u8, int B, int C, int H, int W, void* dst, int Cpad, int dtype,
                    void* stream);
/* the same for a whole table of convs in one launch (the per-step re-pack of every layer after the optimiser
 * step): `items` is an array of n_items entries IN DEVICE MEMORY (pointers do not change between steps).     */
typedef struct {
  const float* w;   void* w_fwd;   void* w_dgrad;   /* w_dgrad may be NULL */
  int32_t Cout, Cin, KH, KW, Cin_pad;
  int32_t Cout_pad;   /* row length of w_dgrad ([flipped tap][Cin][Cout_pad]); 0 = Cout.  Cout_pad > Cout: the layer's data gradient contracts
                         over a K-group wider than its output channels (thin layers, Cout = 16): the caller zero-fills the columns once */
} cvcs_pack_item;
int cvcs_pack_conv_weights(const cvcs_pack_item* items_device, int n_items, int dtype, void* stream);

/* parameter layout changes: OIHW f32 -> [KH*KW][Cout][Cin_pad] (fwd) and -> flipped/transposed [KH*KW][Cin][Cout]
 * (data-gradient operand); ConvTranspose2d IOHW -> [1][(dy*2+dx)*Cout+co][Cin] (fwd) and [4 taps][Cin][Cout] (bwd) */
int cvcs_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int KH, int KW, int Cin_pad,
                          void* w_fwd, void* w_dgrad, int dtype, void* stream);
int cvcs_pack_convT_weight(const float* w_iohw, int Cin, int Cout, void* w_fwd, float* bias4, const float* bias,
                           void* w_dgrad, int dtype, void* stream);

/* ---- 1x1 classifier head + loss --------------------------------------------------------------------------
 * cvcs_head_fwd replaces nn.Conv2d(64, NC, 1) (S/nets.py:172): NHWC `dtype` [P, C] -> NCHW f32 logits [B, NC, H, W]. */
int cvcs_head_fwd(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, const float* bias, int NC,
                  float* logits, int dtype, void* stream);
/* BatchNorm in front of the head folded into it (train mode; replaces the BN-apply pass of S/blocks.py:45 for the last
 * decoder layer): w_out[c][k] = w[c][k]*scale[k], bias_out[c] = bias[c] + sum_k w[c][k]*shift[k]; cvcs_head_unfold_grad
 * turns the gradient w.r.t. the folded input back into dW of the real head: dw = scale*dw + shift (x) db.        */
int cvcs_head_fold(const float* w, const float* bias, const float* scale, const float* shift, int NC, float* w_out,
                   float* bias_out, void* stream);
int cvcs_head_unfold_grad(float* dw, const float* db, const float* scale, const float* shift, int NC, void* stream);
/* head + argmax fused (prediction, S/utils.py:88-90, S/inference.py:45-47): u8 labels [B,H,W], ties -> lowest class;
 * identical to cvcs_head_fwd followed by cvcs_argmax_confusion(labels) without the logits round trip.        */
int cvcs_head_argmax(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, const float* bias, int NC,
                     uint8_t* labels, int dtype, void* stream);
/* backward of the head: dx (NHWC dtype), partial dW/db rows -> finalize with cvcs_colsum_finalize            */
int cvcs_head_bwd_rows(int64_t P);
int cvcs_head_bwd(const void* x, int64_t x_ld, const float* dlogits, int B, int H, int W, int C, const float* w, int NC,
                  void* dx, int64_t dx_ld, float* part_dw /* [rows][NC*C + NC] */, int dtype, void* stream);

/* nn.CrossEntropyLoss(weight, ignore_index) (S/utils.py:230,238; called S/train.py:122, S/utils.py:120):
 *   loss = sum_{t!=ignore} -w[t] log_softmax(z)[t] / sum_{t!=ignore} w[t];   dlogits likewise (may be NULL).
 * logits f32 NCHW [B,NC,H,W]; target u8 (target_is_u8=1) or int64 [B,H,W]; class_weight f32 [NC] or NULL.
 * workspace: cvcs_ce_workspace_floats(P) floats; loss_out: 1 float (device).
 * Sharded batches: the mean is over the non-ignored pixels of the WHOLE batch, so with tiles of one batch on several
 * ranks call cvcs_ce_weight_sum (writes this rank's sum of w[t] to workspace[0]), sum workspace[0] over ranks, then
 * cvcs_ce_fwd_bwd(..., external_denominator=1): loss_out = this rank's numerator / global denominator (the ranks'
 * values add up to the reference's loss), dlogits = grad_scale * d(global loss)/d(logits).                    */
int cvcs_ce_workspace_floats(int64_t P);
int cvcs_ce_weight_sum(const void* target, int target_is_u8, int B, int NC, int64_t HW, const float* class_weight,
                       int ignore_index, float* workspace, void* stream);
int cvcs_ce_fwd_bwd(const float* logits, const void* target, int target_is_u8, int B, int NC, int64_t HW,
                    const float* class_weight, int ignore_index, float grad_scale,
                    float* loss_out, float* dlogits, float* workspace, int external_denominator, void* stream);

/* argmax over classes (ties -> lowest index; torch.max/argmax S/utils.py:90,158) -> u8 labels [B,H,W], and
 * confusion-matrix accumulation conf[t*K + p] += 1 for target != ignore (torchmetrics MulticlassConfusionMatrix,
 * S/utils.py:76-78,93-94; ignore_index < 0: none).  conf: int64 [K*K], accumulated.                          */
int cvcs_argmax_confusion(const float* logits, int B, int NC, int64_t HW, uint8_t* labels,
                          const void* target, int target_is_u8, int ignore_index, int K, int64_t* conf, void* stream);
/* the same accumulation from predicted LABELS (u8 [P]) for networks with returns_logits = False (the Ensemble,
 * S/utils.py:89-94).                                                                                        */
int cvcs_label_confusion(const uint8_t* pred, const void* target, int target_is_u8, int64_t P, int ignore_index, int K,
                         int64_t* conf, void* stream);
/* per-pixel majority vote over M label maps labels[M][P] -> out[P]; ties -> smallest label.
 * replaces: torch.mode(torch.stack(preds), dim=0) in Ensemble.forward (S/utils.py:504-507).  M <= 16.        */
int cvcs_vote_labels(const uint8_t* labels, int M, int64_t P, uint8_t* out, void* stream);

/* ---- whole-image inference boundary -----------------------------------------------------------------------------
 * replaces: GID15.__getitem__ crop / _get_padded_patch (S/dataset.py:18-23,70-96), CenterCrop + argmax + iconvert +
 * per-tile PNG + stitching (S/utils.py:145-171, S/inference.py:40-57, S/converters.py:23-36).
 * cvcs_crop_tiles: tiles first_tile .. first_tile+n-1 (row-major, tiles_per_row per image row, stride p) of one u8
 * CHW image -> u8 [n,C,S,S]; S >= p is the border-corrected size, the S-window is CENTRED on the p-tile (the
 * reference crops it at offset S-p, S/dataset.py:19-22, and re-centres by (S-p)/2: its quirk 6), zero outside.
 * cvcs_argmax_stitch: f32 logits [n,NC,S,S] -> argmax of the centre p x p window -> palette colour (u8 [K][3]) and/or
 * label written at the tile's place in the stitched maps (u8 [Hout][Wout][3] / [Hout][Wout]).                    */
int cvcs_crop_tiles(const uint8_t* src, int C, int H, int W, uint8_t* dst, int n, int first_tile, int tiles_per_row,
                    int p, int S, void* stream);
int cvcs_argmax_stitch(const float* logits, int n, int NC, int S, int p, int first_tile, int tiles_per_row,
                       const uint8_t* palette, uint8_t* rgb, uint8_t* labels, int Hout, int Wout, void* stream);
/* the same stitching from per-tile label maps u8 [n][S][S] (cvcs_head_argmax output, or an Ensemble's vote) */
int cvcs_label_stitch(const uint8_t* tiles, int n, int S, int p, int first_tile, int tiles_per_row,
                      const uint8_t* palette, uint8_t* rgb, uint8_t* labels, int Hout, int Wout, void* stream);

/* ---- tile producer --------------------------------------------------------------------------------------------
 * replaces: IterableChunk's eager Python pre-crop of every patch (S/dataset.py:136-172, _random_shift :25-26) and the
 * per-class pixel count behind the wCEL weights (S/dataset.py:346-358).  Decoded images stay resident as u8 on the
 * device; a batch is one gather launch over n origins (int32 [n][2] = top, left; out-of-image pixels read 0).      */
int cvcs_gather_tiles(const uint8_t* src, int C, int H, int W, const int32_t* origins, int n, int S, uint8_t* dst,
                      void* stream);
int cvcs_label_histogram(const uint8_t* labels, int64_t n, int K, int64_t* counts /* int64 [K], accumulated */, void* stream);

/* ---- residual networks (ResNet-encoder U-Nets of BASELINE.json configs 1, 2, 5; no counterpart in the reference, whose
 * factory S/utils.py:174-195 is the seam these models plug into) ------------------------------------------------------
 * cvcs_bn_add_act: the tail of a residual block, out = relu(s1*y1 + b1 + (s2 ? s2*y2 + b2 : y2)): BatchNorm apply of the
 * main branch + the identity (s2 = NULL: y2 is the block input as stored) or the BatchNorm of the 1x1 projection
 * shortcut + the add + the ReLU in one pass.  All views NHWC `dtype`, C % (16/sizeof) == 0.                          */
int cvcs_bn_add_act(const void* y1, int64_t y1_ld, const float* s1, const float* b1,
                    const void* y2, int64_t y2_ld, const float* s2, const float* b2,
                    int64_t M, int C, void* out, int64_t out_ld, int dtype, void* stream);
/* backward of that tail: dz = (g0 + g1 + g2) * (out > 0) - the gradient w.r.t. the block output arrives from up to three
 * consumers (next block's first conv, next block's shortcut, a decoder skip); g1 / g2 may be NULL.  A gradient with
 * gN_half = 1 lives at HALF resolution [B, H/2, W/2, C] and contributes to pixel (2y, 2x) only: the data gradient of a
 * 1x1 / stride 2 projection shortcut, scattered on the fly.  `out` NULL: no ReLU mask (plain sum).                  */
int cvcs_relu_bwd_sum(const void* out, int64_t out_ld,
                      const void* g0, int64_t g0_ld, int g0_half,
                      const void* g1, int64_t g1_ld, int g1_half,
                      const void* g2, int64_t g2_ld, int g2_half,
                      int B, int H, int W, int C, void* dz, int64_t dz_ld, int dtype, void* stream);
/* nn.MaxPool2d(3, stride 2, padding 1) of the ResNet stem and its backward.  idx (u8 [B,Ho,Wo,C]) = window position
 * kh*3+kw of the FIRST maximum in scan order (what ATen's max_pool2d_with_indices keeps); backward gathers: every input
 * pixel sums the gradients of the (at most four) windows whose idx points at it - no atomics, bitwise reproducible.
 * g1 (optional) is a second gradient of the pooled tensor added on the fly (two consumers of the pooled map).       */
int cvcs_maxpool3x3s2_fwd(const void* x, int64_t x_ld, int B, int H, int W, int C, void* out, int64_t out_ld,
                          uint8_t* idx, int dtype, void* stream);
int cvcs_maxpool3x3s2_bwd(const void* g0, int64_t g0_ld, const void* g1, int64_t g1_ld, const uint8_t* idx,
                          int B, int H, int W, int C, void* dx, int64_t dx_ld, int dtype, void* stream);
/* zero-dilation x2: out[b, 2y, 2x] = in[b, y, x], zero elsewhere (out is [B, 2H, 2W, C]).  The data gradient of a
 * 3x3 / stride 2 / pad 1 convolution is the 3x3 / stride 1 / pad 1 convolution of the dilated output gradient with the
 * flipped weights - it then runs on the halo kernel like every other data gradient.                                 */
int cvcs_dilate2x(const void* in, int64_t in_ld, int B, int H, int W, int C, void* out, int64_t out_ld, int dtype,
                  void* stream);
/* sub-grid layout change of a [B, H, W, C] map (the dilated stages of torchvision's DeepLab encoders, S/nets.py:234-257: a 3x3
 * convolution with dilation d and padding d equals a plain 3x3 / pad 1 convolution on the d*d sub-sampled images).  Layout L_d:
 * image n = b*d*d + sy*d + sx of [B*d*d, H/d, W/d, C] holds the pixels (y*d + sy, x*d + sx).  in is in L_a, out in L_b (a, b >= 1
 * divide H and W, which are the sizes of the ORIGINAL map); the gradient of the operation is the same call with a and b swapped.     */
int cvcs_regrid(const void* in, int64_t in_ld, int B, int H, int W, int C, int a, int b, void* out, int64_t out_ld, int dtype,
                void* stream);
/* stem boundary: u8 | f32 NCHW [B,3,H,W] -> [B, H, W + 8, 4] `dtype` (real column x at index x + 3, margins and the
 * 4th channel zero) - the layout the virtual-pixel stem convolution reads (cvcs_conv_desc.aniso).                    */
int cvcs_pack_input_stem(const void* src, int src_is_u8, int B, int H, int W, void* dst, int dtype, void* stream);
/* stem weight OIHW f32 [Cout][3][7][7] -> [7 (kh)][Cout][32] `dtype`, virtual channel kw*4 + c (others zero); and the
 * way back for its gradient: tmp f32 [Cout][32][7] (what cvcs_conv2d_wgrad writes for the 7x1 virtual filter) -> OIHW */
int cvcs_pack_stem_weight(const float* w, int Cout, void* w_fwd, int dtype, void* stream);
int cvcs_unpack_stem_wgrad(const float* tmp, int Cout, float* dw, void* stream);

/* ---- DeepLabV3+ (BASELINE.json configs[2]; the reference wraps torchvision's DeepLabV3 at S/nets.py:234-311 and returns
 * d['out'] :248-250, which torchvision resizes bilinearly to the input size) ------------------------------------------
 * Bilinear resize by an integer factor, align_corners=False (F.interpolate): NHWC activations [B,H,W,C] -> [B,H*s,W*s,C]
 * and the NCHW f32 logits (`planes` = B*NC maps of H x W); the backward calls take the gradient of the LARGE map and
 * write the gradient of the small one (H, W are the small map's), as gathers (no atomics).                          */
int cvcs_resize_bilinear_fwd(const void* in, int64_t in_ld, int B, int H, int W, int C, int scale, void* out, int64_t out_ld,
                             int dtype, void* stream);
int cvcs_resize_bilinear_bwd(const void* gout, int64_t gout_ld, int B, int H, int W, int C, int scale, void* gin,
                             int64_t gin_ld, int dtype, void* stream);
int cvcs_resize_bilinear_nchw_fwd(const float* in, int64_t planes, int H, int W, int scale, float* out, void* stream);
int cvcs_resize_bilinear_nchw_bwd(const float* gout, int64_t planes, int H, int W, int scale, float* gin, void* stream);
/* ASPP image pooling: out[b][c] = scale * sum over the HW pixels of image b (nn.AdaptiveAvgPool2d(1): scale = 1/HW; the
 * backward of the broadcast below: scale = 1), and the broadcast of a [B][C] vector over HW pixels times `scale`
 * (F.interpolate of a 1x1 map: scale = 1; the backward of the mean: scale = 1/HW).                                   */
int cvcs_image_sum(const void* x, int64_t x_ld, int B, int HW, int C, float scale, void* out, int64_t out_ld, int dtype,
                   void* stream);
int cvcs_image_broadcast(const void* v, int64_t v_ld, int B, int HW, int C, float scale, void* out, int64_t out_ld, int dtype,
                         void* stream);
/* 1x1 classifier on any channel count C (multiple of 64, <= 1024; the replaced `classifier[4]` of S/nets.py:243-244):
 * NHWC `dtype` -> NCHW f32 logits [B,NC,H,W] (labels NULL) or u8 argmax labels [B,H,W] (logits NULL; ties -> lowest
 * class); backward: dx NHWC and partial rows [cvcs_linear_head_bwd_rows(P)][NC*C + NC] of dW | db for
 * cvcs_colsum_finalize.                                                                                             */
int cvcs_linear_head_fwd(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, const float* bias, int NC,
                         float* logits, uint8_t* labels, int dtype, void* stream);
int cvcs_linear_head_bwd_rows(int64_t P);
int cvcs_linear_head_bwd(const void* x, int64_t x_ld, const float* dlogits, int B, int H, int W, int C, const float* w, int NC,
                         void* dx, int64_t dx_ld, float* part_dw, int dtype, void* stream);

/* ---- fused GroupNorm + SiLU (north star "fused BN/GN+SiLU"; the `decoder_norm: gn_silu` option of the ResNet-UNets; the
 * reference itself only has BatchNorm + ReLU, S/blocks.py:8-49) ----------------------------------------------------------
 * nn.GroupNorm(G, C) statistics are per (image, group).  Forward: cvcs_gn_stats (per-(image, channel) partial sums of y and
 * y^2: part[B][cvcs_gn_rows(HW)][2][C]) -> cvcs_gn_finalize (merges the channels of a group in f64; per-(image, channel)
 * scale / shift [B][C], mean / invstd [B][G]) -> cvcs_gn_act_fwd: out = act(scale*y + shift), act = SiLU (silu = 1) or identity
 * - normalisation, affine and activation in ONE pass.  Backward: cvcs_gn_act_bwd_reduce (partial sums of dz and dz*y,
 * dz = g * act'(z) recomputed) -> cvcs_gn_bwd_finalize (dgamma, dbeta; tmp [B][2][C]; coefficients ca, cb, cc [B][C]) ->
 * cvcs_gn_act_bwd_apply: dy = ca*dz + cb + cc*y.                                                                        */
int cvcs_gn_rows(int HW);
int cvcs_gn_stats(const void* y, int64_t y_ld, int B, int HW, int C, float* part, int dtype, void* stream);
int cvcs_gn_finalize(const float* part, int B, int HW, int C, int G, const float* gamma, const float* beta, float eps,
                     float* scale, float* shift, float* mean, float* invstd, void* stream);
int cvcs_gn_act_fwd(const void* y, int64_t y_ld, int B, int HW, int C, const float* scale, const float* shift, int silu,
                    void* out, int64_t out_ld, int dtype, void* stream);
int cvcs_gn_act_bwd_reduce(const void* y, int64_t y_ld, const void* g, int64_t g_ld, int B, int HW, int C, const float* scale,
                           const float* shift, int silu, float* part, int dtype, void* stream);
int cvcs_gn_bwd_finalize(const float* part, int B, int HW, int C, int G, const float* gamma, const float* mean,
                         const float* invstd, float* tmp, float* dgamma, float* dbeta, float* ca, float* cb, float* cc,
                         void* stream);
int cvcs_gn_act_bwd_apply(const void* y, int64_t y_ld, const void* g, int64_t g_ld, int B, int HW, int C, const float* scale,
                          const float* shift, int silu, const float* ca, const float* cb, const float* cc, void* dy,
                          int64_t dy_ld, int dtype, void* stream);

/* ---- Swin-T + UPerNet (BASELINE.json configs[3]; north star "patch-embed + windowed attention for the ViT/Swin encoder
 * variant"; the reference documents TSwin / BSwin at configs/train/README.txt:44-45 without a definition) ------------------
 * Tokens are NHWC pixels, every nn.Linear is a 1x1 cvcs_conv2d (+ cvcs_conv2d_wgrad, bias gradient by cvcs_colsum_*).
 * LayerNorm over the C channels of each of M tokens (eps as given); mean / invstd f32 [M] are saved for the backward, which
 * writes dx and partial rows part[cvcs_layernorm_rows(M)][2][C] of (dgamma | dbeta) for cvcs_colsum_finalize(2C columns).  */
int cvcs_layernorm_rows(int64_t M);
int cvcs_layernorm_fwd(const void* x, int64_t x_ld, int64_t M, int C, const float* gamma, const float* beta, float eps, void* out,
                       int64_t out_ld, float* mean, float* invstd, int dtype, void* stream);
int cvcs_layernorm_bwd(const void* x, int64_t x_ld, const void* g, int64_t g_ld, int64_t M, int C, const float* gamma, const float* mean,
                       const float* invstd, void* dx, int64_t dx_ld, float* part, int dtype, void* stream);
/* GELU (erf form): g == NULL: out = gelu(x); else out = g * gelu'(x)                                                        */
int cvcs_gelu(const void* x, int64_t x_ld, const void* g, int64_t g_ld, int64_t M, int C, void* out, int64_t out_ld, int dtype, void* stream);
/* patch embedding input: u8 | f32 NCHW [B,3,H,W] -> [B,H/4,W/4,64] `dtype`, channel c*16 + kh*4 + kw (the flattening of the
 * [96,3,4,4] projection weight; channels 48..63 zero): the 4x4 / stride 4 convolution becomes a 1x1 convolution               */
int cvcs_pack_patches(const void* src, int src_is_u8, int B, int H, int W, void* dst, int dtype, void* stream);
/* SwinPatchMerging gather: [B,H,W,C] -> [B,H/2,W/2,4C], block col*2+row = pixel (2y+row, 2x+col); reverse = its transpose  */
int cvcs_patch_merge(const void* in, int64_t in_ld, int B, int H, int W, int C, void* out, int64_t out_ld, int reverse, int dtype, void* stream);
/* window partition: the [B,H,W,C] token map zero-padded to multiples of 7 and rolled by -shift -> windows [B*nW*49][C]; and
 * the inverse fused with the residual add: out = (res ? res : 0) + un-partitioned windows (padding dropped).  Each is the
 * other's backward.                                                                                                        */
int cvcs_window_gather(const void* map, int64_t map_ld, int B, int H, int W, int C, int shift, void* win, int64_t win_ld, int dtype, void* stream);
int cvcs_window_reverse(const void* win, int64_t win_ld, const void* res, int64_t res_ld, int B, int H, int W, int C, int shift, void* out,
                        int64_t out_ld, int dtype, void* stream);
/* window-7 multi-head attention (head dim 32) over the window tokens: qkv [T][3C] = (q | k | v), table f32 [169][heads]
 * (relative-position bias), shift > 0 adds Swin's -100 region mask; out [T][C].  Backward: dqkv [T][3C] and dtable
 * [169][heads] (workspace: cvcs_window_attention_bwd_workspace_floats; fixed-order reduction).                             */
int cvcs_window_attention_fwd(const void* qkv, int64_t qkv_ld, int B, int H, int W, int C, int heads, int shift, const float* table,
                              void* out, int64_t out_ld, int dtype, void* stream);
int64_t cvcs_window_attention_bwd_workspace_floats(int B, int H, int W, int heads);
int cvcs_window_attention_bwd(const void* qkv, int64_t qkv_ld, const void* dout, int64_t dout_ld, int B, int H, int W, int C, int heads,
                              int shift, const float* table, void* dqkv, int64_t dqkv_ld, float* dtable, float* workspace, int dtype,
                              void* stream);
/* UPerNet: nn.AdaptiveAvgPool2d(S) forward ([B,H,W,C] -> [B,S,S,C]) / backward (in = gradient [B,S,S,C] -> [B,H,W,C]);
 * F.interpolate(size=(Ho,Wo), 'bilinear', align_corners=False) between arbitrary sizes: forward [B,Hi,Wi,C] -> [B,Ho,Wo,C]
 * (accumulate = 1: out += - the FPN top-down add) / backward (a = gradient [B,Ho,Wo,C] -> gradient [B,Hi,Wi,C], a gather). */
int cvcs_adaptive_avg_pool(const void* in, int64_t in_ld, int B, int H, int W, int C, int S, void* out, int64_t out_ld, int backward,
                           int dtype, void* stream);
int cvcs_resize_bilinear_any(const void* a, int64_t a_ld, int B, int Hi, int Wi, int Ho, int Wo, int C, void* out, int64_t out_ld,
                             int backward, int accumulate, int dtype, void* stream);

/* ---- fused optimisers over a flat f32 parameter buffer -------------------------------------------------------
 * replaces: torch.optim.SGD(momentum, weight_decay).step() / torch.optim.Adam.step() (S/utils.py:214,217; S/train.py:126).
 * grad_scale multiplies the gradient first (1/world_size after the sum all-reduce).                           */
int cvcs_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                  float weight_decay, float grad_scale, int first_step, void* stream);
int cvcs_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, float grad_scale, int step, void* stream);

/* ---- ABI 10: fp8 convolutions (BASELINE.json configs[4]: "mixed bf16/fp8 convs") ---------------------------------------
 * replaces: the same nn.Conv2d k3 p1 (S/blocks.py:3-4) and the input-gradient half of its convolution_backward (S/train.py:125) as
 * cvcs_conv2d, for 3x3 / stride 1 / pad 1 layers whose channel counts are multiples of 128 - on the block-scaled matrix instruction
 * (v_mfma_scale_f32_16x16x128_f8f6f4, twice the bf16 rate).  The reference computes in f32 only; this is the reduced-precision
 * throughput path the north star names, checked against the oracle's fp8-emulation mode.
 *   out[p, n] (bf16) = act( s_in * s_w * sum_{kh,kw,c} in8[pix(p,kh,kw), c] * wt8[kh*3+kw][n][c] )
 * in8: fp8 NHWC view (e4m3 for activations, e5m2 for gradients; in_ld in BYTES = elements), wt8: e4m3 [9][Cout][Cin];
 * scale_in / scale_w: DEVICE scalars (f32) holding the dequantisation scales s_in, s_w (element [1] of a scale slot, below).
 * stat_*: BatchNorm partial statistics of the bf16 values written, one row per 16x16 tile (B * ceil(H/16) * ceil(W/16) rows) as
 * the bf16 kernel of cvcs_conv2d writes them.  pre_scale / pre_shift / relu: the eval-mode BatchNorm fold of cvcs_conv_desc.  */
typedef struct {
  const void* in;   int64_t in_ld;  int32_t B, H, W, Cin;   int32_t in_fmt;   /* CVCS_E4M3 | CVCS_E5M2 */
  const void* wt;
  void* out;        int64_t out_ld; int32_t Cout;
  const float* scale_in;  const float* scale_w;
  int32_t relu;
  const float* pre_scale;  const float* pre_shift;
  float* stat_sum;  float* stat_m2;  float* stat_cnt;
} cvcs_conv8_desc;
int cvcs_sizeof_conv8_desc(void);
int cvcs_conv3x3_fp8(const cvcs_conv8_desc* d, void* stream);
#define CVCS_FP8_SLOT_FLOATS 1040
/* A scale slot is CVCS_FP8_SLOT_FLOATS floats: [0..4] = { amax (unused by the kernels: 0), scale, 1 / scale, fmax of its format (448 | 57344),
 * the amax the current scale was made from }, then 64 partial amax words, one per 64-byte line at [16 * (1 + j)] - producers
 * atomically raise ONE of them per wave, cvcs_fp8_update_scales takes the maximum over all and clears them.
 * cvcs_quantize_fp8: q (fp8 view, ld in bytes) = fmt(clamp(x * slot[2], +-slot[3])) of a bf16 view x [M pixels][C], round to nearest
 * even, and the slot's amax words are raised to max|x| (atomic max on the float's bits: order-independent, so a step stays reproducible).
 * take_amax = 0: slot[0] is left alone (evaluation passes use the trained scales without feeding the next step's).
 * q == NULL: the amax only (calibration of a slot without history).  C % 16 == 0.
 * cvcs_fp8_update_scales: delayed scaling at the end of a step - for each of n consecutive slots with amax > 0: scale = margin * amax / fmax,
 * then the amax words are cleared.  (margin >= 1 leaves headroom for the next step's growth; values beyond saturate.)                          */
int cvcs_quantize_fp8(const void* x, int64_t x_ld, int64_t M, int C, void* q, int64_t q_ld, int fmt, float* slot, int take_amax, void* stream);
int cvcs_fp8_update_scales(float* slots, int n, float margin, void* stream);
/* Producer passes WITH an fp8 side output: exactly cvcs_bn_act (no pooling) / cvcs_bn_bwd_apply (no pooled gradient) / cvcs_upsample2x_fwd /
 * cvcs_bn_add_act, which besides their bf16 output write the fp8 image q8 (view, ld in bytes) of the SAME bf16-rounded values, quantised with
 * `slot` as cvcs_quantize_fp8 does, and feed the slot's amax (take_amax) - the tensor an fp8 convolution reads never gets a sweep of its
 * own.  Several producers may share one slot (the two channel ranges of a decoder's concat buffer are one tensor with one scale).  bf16 only. */
int cvcs_bn_act_q8(const void* y, int64_t y_ld, int B, int H, int W, int C, const float* scale, const float* shift, int relu, void* out,
                   int64_t out_ld, void* q8, int64_t q8_ld, int fmt, float* slot, int take_amax, int dtype, void* stream);
int cvcs_bn_bwd_apply_q8(const void* y, int64_t y_ld, const void* g1, int64_t g1_ld, int B, int H, int W, int C, const float* scale,
                         const float* shift, const float* save_mean, const float* save_invstd, const float* coef_a, const float* coef_b,
                         int mode, void* dy, int64_t dy_ld, float* part_db, void* q8, int64_t q8_ld, int fmt, float* slot, int take_amax,
                         int dtype, void* stream);
int cvcs_upsample2x_fwd_q8(const void* in, int64_t in_ld, int B, int H, int W, int C, void* out, int64_t out_ld, void* q8, int64_t q8_ld,
                           int fmt, float* slot, int take_amax, int dtype, void* stream);
int cvcs_bn_add_act_q8(const void* y1, int64_t y1_ld, const float* s1, const float* b1, const void* y2, int64_t y2_ld, const float* s2,
                       const float* b2, int64_t M, int C, void* out, int64_t out_ld, void* q8, int64_t q8_ld, int fmt, float* slot,
                       int take_amax, int dtype, void* stream);

/* ---- ABI 10: dropout with a counter-based mask --------------------------------------------------------------------------------
 * replaces: the nn.Dropout(0.5) of torchvision's DeepLabHead / ASPP (after the 1x1 projection) that the reference's `Resnet101` trains with
 * (S/nets.py:242-246 builds torchvision.models.segmentation.deeplabv3_resnet101; net.train() at S/train.py:113 switches it on).
 * out[i] = keep(i) ? x[i] / (1 - p) : 0, keep(i) = (splitmix64(seed ^ step * 0xD6E8FEB86659FD93 + i * 0x9E3779B97F4A7C15) >> 40) < (1 - p) * 2^24,
 * i = pixel * C + channel; state = {seed, step} (two uint64 on the device).  The same call on the gradient is the backward; the mask is
 * a pure function of (seed, step, i), so a test replays it on the host.  cvcs_counter_add advances `step` (a recorded plan replays with
 * fixed arguments).  In place (out == x) is allowed.                                                                            */
int cvcs_dropout(const void* x, int64_t x_ld, int64_t M, int C, void* out, int64_t out_ld, const uint64_t* state, float p, int dtype,
                 void* stream);
int cvcs_counter_add(uint64_t* counter, uint64_t inc, void* stream);

/* ---- ABI 11: the reference's `SegformerMod` (/root/reference/source/scripts/nets.py:313-349; factory name /root/reference/source/scripts/utils.py:191-192):
 * transformers' Mix-Transformer segmentation model + the reference's ConvTranspose tail.  Tokens are NHWC pixels, every nn.Linear is a 1x1
 * convolution (cvcs_conv2d / cvcs_conv2d_wgrad), LayerNorm / GELU / bilinear resizes / BatchNorm are the Swin and ResNet entry points.
 *
 * Patch matrices.  cvcs_im2col replaces the input side of nn.Conv2d(k, stride, padding) for the overlapping patch embeddings (7/4/3, 3/2/1),
 * the sequence-reduction convolutions (k = s = 8 | 4 | 2) and - phase by phase - the two nn.ConvTranspose2d of the tail (nets.py:331-334):
 *   col[b, oy, ox, (ky*KW + kx)*C + c] = in[b, oy*stride + oy0 + ky*dir, ox*stride + ox0 + kx*dir, c]    (0 outside the map and for K >= KH*KW*C)
 * convolution: oy0 = ox0 = -padding, dir = +1.  Phase (a, b) of ConvTranspose2d(k, stride 2, padding p), i.e. the output pixels (2i + a, 2j + b):
 * stride = 1, dir = -1, KH = KW = k / 2, oy0 = (a + p) >> 1, ox0 = (b + p) >> 1, and tap (jy, jx) multiplies w[ci][co][((a+p)&1) + 2*jy][((b+p)&1) + 2*jx].
 * The layer is then a 1x1 convolution of `col`.  cvcs_im2col_stem is the same gather on the tile itself (planar u8 / f32 [B,3,H,W], K index
 * (ky*KW + kx)*3 + c) with the reference's preprocessor folded in: (x - mean[c]) / std[c] on the RAW 0..255 values (nets.py:337-340, 347),
 * padding pixels 0 AFTER the normalisation.  cvcs_col2im is the transposed gather (data gradient), summed over nph phase slabs (phase_elems
 * elements apart) in f32 in a fixed order.  cvcs_phase_shuffle interleaves the four phase outputs [4][B,H,W,C] into [B,2H,2W,C] (+ ReLU);
 * cvcs_phase_unshuffle is the transposed move with the ReLU mask taken from the forward output (relu_out, NULL = none).                     */
int cvcs_im2col(const void* in, int64_t in_ld, int B, int H, int W, int C, int KH, int KW, int stride, int oy0, int ox0, int dir, int Ho, int Wo,
                void* col, int64_t col_ld, int dtype, void* stream);
int cvcs_im2col_stem(const void* src, int src_is_u8, int B, int H, int W, int KH, int KW, int stride, int pad, const float* mean3, const float* std3,
                     int Ho, int Wo, void* col, int64_t col_ld, int dtype, void* stream);
int cvcs_col2im(const void* col, int64_t col_ld, int64_t phase_elems, int nph, const int* oy0, const int* ox0, int dir, int stride, int KH, int KW,
                int B, int Ho, int Wo, int C, int H, int W, void* dx, int64_t dx_ld, int dtype, void* stream);
int cvcs_phase_shuffle(const void* ph, int64_t ph_ld, int64_t phase_elems, int B, int H, int W, int C, void* out, int64_t out_ld, int relu, int dtype,
                       void* stream);
int cvcs_phase_unshuffle(const void* g, int64_t g_ld, const void* relu_out, int64_t relu_out_ld, int B, int H, int W, int C, void* gph, int64_t gph_ld,
                         int64_t phase_elems, int dtype, void* stream);
/* Depthwise 3x3 of the Mix-FFN (transformers SegformerDepthWiseConv: nn.Conv2d(C, C, 3, 1, 1, groups=C)): w f32 [C][9] as the module stores it,
 * bias f32 [C] or NULL.  flip = 1 applies the filter reversed = the data gradient.  cvcs_dwconv3x3_wgrad: partial rows
 * part[cvcs_dwconv3x3_wgrad_rows(B*H*W)][C*9 (c-major) | C] of dW and db; cvcs_colsum_finalize sums them into a weight | bias gradient pair. */
int cvcs_dwconv3x3(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, const float* bias, int flip, void* out, int64_t out_ld,
                   int dtype, void* stream);
int cvcs_dwconv3x3_wgrad_rows(int64_t M);
int cvcs_dwconv3x3_wgrad(const void* x, int64_t x_ld, const void* dy, int64_t dy_ld, int B, int H, int W, int C, float* part, int dtype, void* stream);
/* DropPath (transformers SegformerDropPath, train mode): scales[call][b] = (u24 < keep * 2^24) ? 1 / keep : 0 with keep = 1 - rates[call] and
 * u24 = splitmix64((seed ^ 0x5DEECE66D) ^ step * 0xD6E8FEB86659FD93 + (call * B + b) * 0x9E3779B97F4A7C15) >> 40; rate 0 -> 1.  state = {seed, step} as
 * for cvcs_dropout; rates f32 [n] on the device.  cvcs_scale_rows_add: out[b, n, :] = res[b, n, :] + scale[b] * x[b, n, :] (res NULL: the scaled
 * rows alone - the branch gradient; scale NULL = 1).                                                                                          */
int cvcs_drop_path_scales(const uint64_t* state, const float* rates, int n, int B, float* out, void* stream);
int cvcs_scale_rows_add(const void* x, int64_t x_ld, const float* scale, const void* res, int64_t res_ld, int B, int64_t HW, int C, void* out,
                        int64_t out_ld, int dtype, void* stream);
/* Spatial-reduction attention (transformers SegformerAttention + eager_attention_forward): q [B, N, heads*D], kv [B, Nk, 2*heads*D] (key
 * projection in channels [0, C), value projection in [C, 2C): the two nn.Linear run as one GEMM), head_dim D = 32 (MiT-b0) | 64 (MiT-b1 ... b5),
 * out = softmax(q k^T / sqrt(D)) v per image and head; lse f32 [B*heads][N] (log-sum-exp of the scaled scores) is what the backward recomputes the
 * probabilities from.
 * cvcs_sr_attention_bwd: gq, gkv from go (gradient of out); workspace of cvcs_sr_attention_bwd_workspace floats.                              */
int cvcs_sr_attention_fwd(const void* q, int64_t q_ld, const void* kv, int64_t kv_ld, int B, int N, int Nk, int heads, int head_dim, void* out,
                          int64_t out_ld, float* lse, int dtype, void* stream);
int64_t cvcs_sr_attention_bwd_workspace(int B, int N, int Nk, int heads, int head_dim);
int cvcs_sr_attention_bwd(const void* q, int64_t q_ld, const void* kv, int64_t kv_ld, const void* o, int64_t o_ld, const void* go, int64_t go_ld,
                          const float* lse, int B, int N, int Nk, int heads, int head_dim, void* gq, int64_t gq_ld, void* gkv, int64_t gkv_ld,
                          float* workspace, int dtype, void* stream);
/* ---- MobileNetV3 pieces (factory name `MobileNet`: S/nets.py:277-311, S/utils.py:183-184 = torchvision deeplabv3_mobilenet_v3_large) ------------
 * Depthwise K x K convolution (nn.Conv2d(C, C, K, stride, pad, dil, groups=C), K = 3 | 5; no bias - a BatchNorm follows), NHWC:
 *   w f32 [K*K][C] tap-major (the engine gathers it from the module's [C][1][K][K]).
 * cvcs_dwconv: forward; stat_* (NULL = none): the BatchNorm partial statistics of the values written, cvcs_dwconv_rows(B*Ho*Wo, C, dtype)
 *   rows in the format cvcs_bn_finalize reads (sum, second moment about the row's own mean, count).
 * cvcs_dwconv_dgrad: dx [B,H,W,C] from dy [B,Ho,Wo,C].  cvcs_dwconv_wgrad (K = 3 | 5): partial rows
 *   part[cvcs_dwconv_wgrad_rows(B*Ho*Wo, C, K, dtype)][K*K][C]; cvcs_colsum_finalize(part, rows, K*K*C) sums them into the tap-major gradient.
 * hardswish is an activation code of the BatchNorm passes: cvcs_bn_act(relu = 2), cvcs_bn_bwd_reduce / _apply(mode = 3).            */
int cvcs_dwconv_rows(int64_t M, int C, int dtype);
int cvcs_dwconv(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, int K, int stride, int pad, int dil, void* y, int64_t y_ld,
                int Ho, int Wo, float* stat_sum, float* stat_m2, float* stat_cnt, int dtype, void* stream);
int cvcs_dwconv_dgrad(const void* dy, int64_t dy_ld, int B, int Ho, int Wo, int C, const float* w, int K, int stride, int pad, int dil, void* dx,
                      int64_t dx_ld, int H, int W, int dtype, void* stream);
int cvcs_dwconv_wgrad_rows(int64_t M, int C, int K, int dtype);
int cvcs_dwconv_wgrad(const void* x, int64_t x_ld, const void* dy, int64_t dy_ld, int B, int H, int W, int C, int K, int stride, int pad, int dil,
                      int Ho, int Wo, float* part, int dtype, void* stream);
/* Squeeze-excite (torchvision SqueezeExcitation: x * hardsigmoid(fc2(relu(fc1(avgpool(x)))))); the pooled mean is cvcs_image_sum, the two fully
 * connected layers are 1x1 convolutions on [B,1,1,C] maps.
 * cvcs_se_scale: out[b,p,c] = x[b,p,c] * s[b,c] (+ add[b,c] * add_scale; add NULL = none) - the forward, and the backward w.r.t. x with the
 *   gradient of the mean riding along (x = g, add = gradient of the pooled vector, add_scale = 1/HW);
 * cvcs_image_dot: out[b,c] = sum_p a[b,p,c] * x[b,p,c] (the gradient of s);
 * cvcs_hardsigmoid: out = clamp(t/6 + 1/2, 0, 1) (g NULL) or the backward g * (|t| < 3 ? 1/6 : 0).                                     */
int cvcs_se_scale(const void* x, int64_t x_ld, const void* s, int64_t s_ld, const void* add, int64_t add_ld, float add_scale, int B, int HW, int C,
                  void* out, int64_t out_ld, int dtype, void* stream);
int cvcs_image_dot(const void* a, int64_t a_ld, const void* x, int64_t x_ld, int B, int HW, int C, void* out, int64_t out_ld, int dtype, void* stream);
/* cvcs_bn_add: out = scale[c] * y + shift[c] (+ res): the BatchNorm apply of a linear bottleneck's projection with the residual add, no activation */
int cvcs_bn_add(const void* y, int64_t y_ld, const float* scale, const float* shift, const void* res, int64_t res_ld, int64_t M, int C, void* out,
                int64_t out_ld, int dtype, void* stream);
int cvcs_hardsigmoid(const void* t, int64_t t_ld, const void* g, int64_t g_ld, int64_t M, int C, void* out, int64_t out_ld, int dtype, void* stream);
/* Table-driven weight gather: the f32 master tensors keep the reference's layouts (Conv2d OIHW, ConvTranspose2d IOHW); the GEMM operands of the
 * patch-matrix layers are re-gathered from them every step in ONE launch:
 *   dst[r][c] (R rows of pitch Cp; `dtype`, or f32 when f32_out) = valid(r, c) ? src[base + idx3(r; rd1, rd2, rs0..2) + idx3(c; cd1, cd2, cs0..2)] : 0
 *   valid(r, c) = r < Rv && c < Cv && (rv2 == 0 || r % rd2 < rv2) && (cv2 == 0 || c % cd2 < cv2)   (rv2 / cv2: channel padding inside a tap block)
 *   idx3(x; d1, d2, s0, s1, s2) = (x / (d1*d2)) * s0 + ((x / d2) % d1) * s1 + (x % d2) * s2
 * cvcs_scatter_weight_grads is the transposed move for gradients: src[base + idx3(r) + idx3(c)] = dst[r][c] (dst f32) for r < Rv, c < Cv.       */
typedef struct {
  const float* src;
  void* dst;
  int64_t base, rs0, rs1, rs2, cs0, cs1, cs2;
  int32_t R, Cp, Rv, Cv, rd1, rd2, cd1, cd2, f32_out, rv2, cv2, pad_;
} cvcs_gather_item;
int cvcs_gather_weights(const cvcs_gather_item* items_device, int n_items, int dtype, void* stream);
int cvcs_scatter_weight_grads(const cvcs_gather_item* items_device, int n_items, void* stream);
/* NHWC map [B*HW][C] (`dtype`, channels >= NC are padding) <-> the NCHW f32 logits / logit gradients [B][NC][HW] of the nn.Module contract
 * (S/nets.py:346-349 returns `self.seq(out)`, NCHW).  cvcs_nhwc_from_planes zero-fills the padding channels.                                */
int cvcs_planes_from_nhwc(const void* x, int64_t x_ld, int B, int64_t HW, int C, int NC, float* planes, int dtype, void* stream);
int cvcs_nhwc_from_planes(const float* planes, int B, int64_t HW, int NC, void* x, int64_t x_ld, int C, int dtype, void* stream);
/* nn.ConvTranspose2d(NC, NC, k, stride 2, padding p) (k = 4 | 8; the reference's tail, S/nets.py:331-334) as ONE stride-1 convolution from CP to
 * 4*CP channels (channel = phase*CP + co, phase = (oy & 1)*2 + (ox & 1)) over a T x T window, T = k/2 + 1, padding T/2, followed by
 * cvcs_phase_shuffle with the phases interleaved in the channels (ph_ld = 4*CP, phase_elems = CP).  cvcs_deconv_pack writes the operands from
 * the master weight w [NC][NC][k][k] (IOHW) and bias: layout 0 = cvcs_conv2d images wf [T*T][4CP][CPk], wd [T*T][CP][4CP] (flipped taps);
 * layout 1 = 1x1 images over the cvcs_im2col patch matrix, wf [4CP][Kp] with K = (ty*T + tx)*CP + ci, wd [Kp][4CP]; bias4 [4CP].
 * cvcs_deconv_unpack_grad: the master gradients gw [NC][NC][k][k], gb [NC] from that convolution's dW (layout 0: OIHW [4CP][CP][T][T] f32,
 * layout 1: [4CP][Kp] f32) and db4 [4CP].                                                                                                   */
int cvcs_deconv_pack(const float* w, const float* bias, int NC, int k, int p, int CP, int layout, int CPk, int Kp, void* wf, void* wd, float* bias4,
                     int dtype, void* stream);
int cvcs_deconv_unpack_grad(const float* dw, const float* db4, int NC, int k, int p, int CP, int layout, int Kp, float* gw, float* gb, void* stream);
/* cvcs_relu_bwd_sum with the first pass of the BatchNorm backward(s) that consume dz (S/blocks.py has no residual blocks; this is the tail of
 * torchvision's BasicBlock / Bottleneck as BASELINE's ResNet encoders use it): the block's last BatchNorm y[0] and, with a projection shortcut, the
 * downsample BatchNorm y[1] (NULL: none) both receive dz, so part_dz[row][c] = sum dz and part_dzx[j][row][c] = sum dz * (y_j - mean_j) * invstd_j
 * (rows = cvcs_bn_bwd_rows(B*H*W), the layout cvcs_bn_bwd_finalize reads) come out of the same launch.                                        */
typedef struct {
  const void* out; int64_t out_ld;
  const void* g[3]; int64_t g_ld[3]; int32_t g_half[3]; int32_t dtype;
  void* dz; int64_t dz_ld;
  int32_t B, H, W, C;
  const void* y[2]; int64_t y_ld[2];
  const float* mean[2]; const float* invstd[2];
  float* part_dz; float* part_dzx[2];
  /* (ABI 14) one more term of the sum arrives THROUGH a MaxPool2d(3, 2, 1) of this map - the ResNet stem (torchvision resnet.py: conv1 -> bn1 -> relu
   * -> maxpool, the relu output also being the U-Net's first skip): pool_g[0] (+ pool_g[1]) are gradients of the pooled map [(H+1)/2][(W+1)/2],
   * pool_idx the arg-max tap of every pooled element as cvcs_maxpool3x3s2_fwd wrote it.  NULL pool_idx: none.  Replaces cvcs_maxpool3x3s2_bwd +
   * cvcs_relu_bwd_sum + cvcs_bn_bwd_reduce of the stem by one launch (g[0] may then be NULL: no other gradient).                                  */
  const void* pool_g[2]; int64_t pool_g_ld[2]; const uint8_t* pool_idx;
} cvcs_tail_bwd_desc;
int cvcs_relu_bwd_sum_bn(const cvcs_tail_bwd_desc* d, void* stream);
/* x[0..n) *= *scalar_device unless it is exactly 1 (decided on the device: no host synchronisation).  Used for the incoming gradient of the fused
 * cross-entropy (`loss.backward()` hands a 1; `(loss / k).backward()` a 1 / k) - replaces a float(tensor) comparison on the host.              */
int cvcs_scale_unless_one(float* x, int64_t n, const float* scalar_device, void* stream);
/* the same on a contiguous bf16 buffer (the head's data gradient written by cvcs_head_ce before the incoming gradient of the loss is known) */
int cvcs_scale_unless_one_bf16(void* x, int64_t n, const float* scalar_device, void* stream);
/* ---- BatchNorm behind a 1x1 convolution WITHOUT materialising the convolution output (ABI 13) -----------------------------------
 * replaces: the last conv1x1 -> nn.BatchNorm2d -> (+ shortcut) -> ReLU of a ResNet bottleneck (torchvision Bottleneck, the encoders of
 * S/nets.py:234-311) in train mode, forward and backward, for y = W a over the M pixels of a map (a: [M][m] bf16, W: [C][m] bf16).
 * The batch statistics of y are functions of the m x m second-moment ("Gram") matrix of a: mean_c = w_c . s / M,
 * E[y_c^2] = w_c^T G w_c / M with G = sum_p a_p a_p^T, s = sum_p a_p - so y (4x wider than a in a bottleneck) is never written or read:
 *   cvcs_gram              G (f64 [m][m]) and s (f64 [m]) of a view: one streaming pass on the matrix cores (partial slabs summed in f64,
 *                          fixed order); workspace = cvcs_gram_workspace_floats(M, m) floats
 *   cvcs_bn_gram_finalize  what cvcs_bn_finalize(train=1) does, from (G, s): scale / shift / saved mean / invstd / running statistics;
 *                          also Q = W G (f32 [C][m]) for the backward
 *   (forward: cvcs_conv2d with pre_scale / pre_shift (+ res) applies the BatchNorm (+ shortcut, ReLU) in the conv's epilogue)
 *   cvcs_bn_gram_bwd       given R = dz^T a (f32 [C][m], the plain weight-gradient GEMM of dz against a) and sum_dz (f32 [C], its dbias):
 *                          sum_p dz_c y_c = w_c . R_c, so dgamma / dbeta need no pass over y either; with dy = alpha dz + beta0 + kappa y
 *                          (the BatchNorm backward, per channel) the weight gradient is dW_c = alpha_c R_c + beta0_c s + kappa_c Q_c, and the
 *                          data gradient W^T dy = (W^T diag(alpha)) dz + (W^T diag(kappa) W) a + W^T beta0: ONE GEMM over the two sources
 *                          [dz | a] (cvcs_conv_desc.in2) with the packed weight wd2 [m][C + m] bf16 and bias [m] written here (first C columns)
 *                          and by cvcs_bn_gram_mmat (last m columns + bias), coefficients kappa / beta0 in coef [2][C].                  */
int64_t cvcs_gram_workspace_floats(int64_t M, int C);
int cvcs_gram(const void* x, int64_t x_ld, int64_t M, int C, double* gram, double* colsum, float* workspace, void* stream);
int cvcs_bn_gram_finalize(const double* gram, const double* colsum, const void* w_bf16, int C, int m, int64_t M,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                          float* scale, float* shift, float* save_mean, float* save_invstd, float* q_out, void* stream);
int cvcs_bn_gram_bwd(const float* r, const float* sum_dz, const void* w_bf16, const float* q, const double* colsum, int C, int m, int64_t M,
                     const float* scale, const float* save_mean, const float* save_invstd,
                     float* dgamma, float* dbeta, float* dw, void* wd2_bf16, float* coef, void* stream);
int64_t cvcs_bn_gram_mmat_workspace_floats(int C, int m);
int cvcs_bn_gram_mmat(const void* w_bf16, const float* coef, int C, int m, void* wd2_bf16, float* bias, float* workspace, void* stream);
/* (ABI 14) a block with a SAME-RESOLUTION projection shortcut (torchvision Bottleneck.downsample at stride 1: layer1.0, the dilated stages of the
 * DeepLab encoders; torchvision/models/resnet.py as S/nets.py:8 imports it): both 1x1 convs of the tail read narrow inputs (a: m1 channels, the block
 * input h: m2 channels), both BatchNorms come from Gram matrices, and relu(BN3(W3 a) + BNd(Wd h)) is ONE two-source GEMM (cvcs_conv_desc.in2) whose
 * operand [C][m1 + m2] = [scale3 * W3 | scaled * Wd] (bf16) and bias = shift3 + shiftd this call writes: the projection's output is never stored.   */
int cvcs_bn_gram_fold(const void* w1_bf16, const float* scale1, const float* shift1, int m1, const void* w2_bf16, const float* scale2,
                      const float* shift2, int m2, int C, void* w_out_bf16, float* bias, void* stream);
/* ---- 1x1 head + cross-entropy + their backward in ONE pass (ABI 13) ----------------------------------------------------------------
 * replaces: `logits = net(x)` (the head nn.Conv2d(C, NC, 1), S/nets.py:172), `criterion(logits, target)` (nn.CrossEntropyLoss, S/utils.py:230,238)
 * and the head's half of `loss.backward()` (S/train.py:121-125) for bf16 activations with C = 16 | 32 | 64 channels and NC <= 32 classes:
 * the [B, NC, H, W] f32 logits and their gradient are never written.  x: the last decoder activation [B*H*W][C] (view), target as cvcs_ce_fwd_bwd,
 * w f32 [NC][C], bias f32 [NC].  Writes loss_out (mean over the non-ignored pixels, class weights as S/utils.py:230), dx (the gradient
 * w.r.t. x, bf16 view) and cvcs_head_ce_rows(P) partial rows [NC*C | NC] of dW | db (summed by cvcs_colsum_finalize, as cvcs_head_bwd's).
 * workspace: cvcs_ce_workspace_floats(P) floats.  grad_scale multiplies the gradients (1 for a plain loss.backward()).
 * w_dx (NULL = w): the weights of the data gradient dx = w_dx^T dl when they differ from those of the logits - the reference U-Nets fold their
 * last BatchNorm into the head (cvcs_head_fold: logits from the folded weights on the pre-BatchNorm activation, dx w.r.t. the BatchNorm output
 * from the real ones, dW w.r.t. the folded ones, un-folded by cvcs_head_unfold_grad).                                                      */
int cvcs_head_ce_rows(int64_t P);
int cvcs_head_ce(const void* x, int64_t x_ld, int B, int H, int W, int C, const float* w, const float* w_dx, const float* bias, int NC,
                 const void* target, int target_is_u8, const float* class_weight, int ignore_index, float grad_scale,
                 void* dx, int64_t dx_ld, float* part_dw, float* workspace, float* loss_out, int dtype, void* stream);
/* A recorded launch plan driven from C (host time per step < 0.5 ms).  A plan is the list of launch entry points of
 * THIS header that one pass of a network issues for one input shape, with their arguments (every pointer a persistent buffer).  cvcs_call holds one
 * launch: the function, its integer-class arguments in declaration order WITHOUT the trailing stream (pointers, int, int64_t), and its float
 * arguments in declaration order.  cvcs_replay issues calls[0..n) on `stream`; on the first non-zero return it stops, stores the index and
 * returns that code.  (x86-64 System V only: one 28 + 8 slot prototype calls every entry point; no struct is passed by value anywhere.)        */
#define CVCS_CALL_MAX_INT 28
#define CVCS_CALL_MAX_FLT 8
typedef struct {
  void* fn;
  int32_t nint, nflt;
  int64_t i[CVCS_CALL_MAX_INT];
  float f[CVCS_CALL_MAX_FLT];
} cvcs_call;
int cvcs_sizeof_call(void);
int cvcs_replay(const cvcs_call* calls, int n, void* stream, int* failed_index);

#ifdef __cplusplus
}
#endif
#endif
