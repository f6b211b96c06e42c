/*
 * cineflow.h -- C ABI of libcineflow_hip.so (MI355X / gfx950 only).
 *
 * The reference (nicolas1805961/Cardiac-Segmentation-Optical-flow) has no FFI layer: its operator
 * boundary is the `forward` of a handful of PyTorch modules.  Each entry point
 * below replaces the device work of one such `forward` (or a fused group of them) and cites it as
 * file:line relative to the reference root.  INTEGRATION.md shows the ctypes stub a reference maintainer
 * would add to route those modules here.
 *
 * Conventions
 *   - all tensors are caller-owned device pointers, contiguous, fp32 unless stated, NCHW ("channel-first");
 *   - no allocation, no ownership transfer, no host synchronisation inside the library; workspaces are
 *     passed in; every launch goes on the caller's `stream` (a hipStream_t passed as void*), so every
 *     entry point is safe to capture in a hipGraph and to call concurrently on different streams;
 *   - return value 0 = success, <0 = error (CF_ERR_*); cf_last_error() gives the text for the calling
 *     thread.  Shapes are validated on the host before any launch.
 */
#ifndef CINEFLOW_H
#define CINEFLOW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_OK 0
#define CF_ERR_ARG (-1)     /* bad shape / null pointer / unsupported size */
#define CF_ERR_LAUNCH (-2)  /* hipGetLastError() after a launch            */

/* activation codes used by the fused epilogues */
#define CF_ACT_NONE 0
#define CF_ACT_GELU 1    /* exact erf GELU (torch.nn.GELU default)       */
#define CF_ACT_RELU 2
#define CF_ACT_LRELU 3   /* LeakyReLU(0.01), generic_UNet.py:38          */
#define CF_ACT_TANH 4
#define CF_ACT_SIGMOID 5

/* residual placement for cf_group_norm */
#define CF_RES_NONE 0
#define CF_RES_BEFORE_ACT 1  /* SingleConv, lib/utils.py:1258-1261  */
#define CF_RES_AFTER_ACT 2   /* DoubleConv, lib/utils.py:1207-1212  */

const char* cf_last_error(void);
int cf_version(void);

/* ---------------------------------------------------------------- VoxelMorph warp family
 * SpatialTransformer.forward, nnunet/network_architecture/integration.py:61-79 (2-D branch):
 * out[b,c,i,j] = bilinear(src[b,c], i + flow[b,0,i,j], j + flow[b,1,i,j]), align_corners=True, zero padding,
 * with the reference's normalise/un-normalise fp32 rounding path reproduced. */
int cf_warp_bilinear_2d(const float* flow, const float* src, float* out, int B, int C, int H, int W, void* stream);

/* VecInt.forward, integration.py:95-99: v = v/2^nsteps; nsteps x { v = v + warp(v, v) }.
 * `tmp` is a caller workspace of B*2*H*W floats; result in `out` (may not alias `vec`). */
int cf_vecint_2d(const float* vec, float* out, float* tmp, int B, int H, int W, int nsteps, void* stream);

/* warp_linear, nnunet/network_architecture/SegFlowGaussian.py:3571-3580, fused:
 * one_hot(labels, K) -> warp with flow[t] -> argmax over K (first maximum wins).
 * flow [T,B,2,H,W]; labels uint8 [B,H,W] (ED label map, values < K <= 8); out uint8 [T,B,H,W]. */
int cf_warp_labels_2d(const float* flow, const uint8_t* labels, uint8_t* out, int T, int B, int K, int H, int W,
                      void* stream);

/* The 6-channel memory-encoder input of SegFlowGaussian.py:1427-1433 / :1898-1904, fused:
 * reg = warp(cum, xt); out = cat[x0, xt, cum(2), x0 - reg, reg].  x0, xt [B,1,H,W]; cum [B,2,H,W]; out [B,6,H,W]. */
int cf_memory_input(const float* x0, const float* xt, const float* cum, float* out, int B, int H, int W, void* stream);

/* jacobian_determinant, nnunet/compute_jacobian.py:16-59 (2-D branch), np.gradient semantics, float64 math.
 * disp [B,2,H,W] channel-first (channel 0 along axis H); det float64 [B,H,W].  H,W >= 2. */
int cf_jacobian_det_2d(const float* disp, double* det, int B, int H, int W, void* stream);

/* 3-D branches of the same two functions: SpatialTransformer.forward for a volume (integration.py:75-77; trilinear,
 * align_corners=True, zeros outside; flow [B,3,D,H,W] with channel i displacing along axis i of (D,H,W), src/out
 * [B,C,D,H,W], out must not alias src) and jacobian_determinant's 3-D case (compute_jacobian.py:42-52;
 * disp [B,3,D,H,W] -> det float64 [B,D,H,W]).  All sizes >= 2. */
int cf_warp_trilinear_3d(const float* flow, const float* src, float* out, int B, int C, int D, int H, int W, void* stream);
int cf_jacobian_det_3d(const float* disp, double* det, int B, int D, int H, int W, void* stream);

/* ---------------------------------------------------------------- correlation
 * CorrVolume(radius, stride)(cur, prev) -- source absent from the reference; call sites
 * SegFlowGaussian.py:256-261, :1376-1377; spec in DESIGN.md ("parity unpinned"):
 * out[b,(dy+r)(2r+1)+(dx+r),y,x] = mean_c cur[b,c,y,x] * prev[b,c,y+dy*stride,x+dx*stride], zero outside. */
int cf_corr_volume(const float* cur, const float* prev, float* out, int B, int C, int H, int W, int radius, int stride,
                   void* stream);
/* A/B knob without a reference counterpart: CorrVolume calls with radius 4, dilation 1 / 2 / 4, C % 16 == 0, W % 64 == 0 and H % (8 x dilation) == 0
 * run on the f16-MFMA kernel (csrc/corr_mfma.hip: 2-D banded products of hi/lo-split operands, fp32 accumulation, within 1e-5 of the fp32 kernel);
 * 0 keeps them on the fp32 vector kernel (also CF_CORR_MFMA=0).  Returns the previous setting. */
int cf_corr_mfma_enable(int on);

/* CorrBlock (published RAFT; call site SegFlowGaussian.py:929): all-pairs volume
 * corr[b, n1, n2] = sum_c f1[b,c,n1] f2[b,c,n2] / sqrt(C) into pyr level 0, then `levels`-1 2x2 average pools.
 * pyr: one buffer, level l at element offset sum_{j<l} B*N*(H>>j)*(W>>j), N = H*W.  H*W % 64 == 0. */
int cf_corr_pyramid(const float* f1, const float* f2, float* pyr, int B, int C, int H, int W, int levels, void* stream);

/* CorrBlock.__call__(coords), call site SegFlowGaussian.py:935: radius-r bilinear lookup on every level.
 * coords [B,2,H,W] (x,y); out [B, levels*(2r+1)^2, H, W]. */
int cf_corr_lookup(const float* pyr, const float* coords, float* out, int B, int H, int W, int levels, int radius,
                   void* stream);

/* SegFlowGaussian.upsample_flow, SegFlowGaussian.py:846-857: convex 8x upsampling.
 * flow [B,C,h,w], mask [B,576,h,w] -> out [B,C,8h,8w] (C = 2 for flow, 4 for upsample_seg :860-871). */
int cf_convex_upsample(const float* flow, const float* mask, float* out, int B, int C, int h, int w, void* stream);

/* ---------------------------------------------------------------- convolution (implicit GEMM on fp32 MFMA)
 * nn.Conv2d forward as used by DoubleConv (lib/utils.py:1182-1215), ConvDropoutNormNonlin
 * (generic_UNet.py:64-68), ConvGRUCell (convGRU.py:57-66), the nn.Linear / MultiheadAttention projections of
 * TransformerFlowLayer (lib/vit_transformer.py:1253-1267, as 1x1 convs on channel-first tokens) and the RAFT
 * update block.  The input is the channel concatenation cat[x1 (C1 ch), x2 (C2 ch)] (x2 may be NULL/0): the
 * reference's torch.cat((skip, x), 1) never materialises.
 *   wt    : weights pre-transposed to [K = (C1+C2)*KH*KW][Cout]  (K index = ci*KH*KW + kh*KW + kw);
 *           w_bstride != 0 selects per-sample weights (element stride) -- used for the all-pairs GEMM
 *   out   : written into channels [out_coff, out_coff+Cout) of a [B,out_ctotal,Ho,Wo] tensor
 *   value : act(alpha * conv + bias[co]) + res[b,co,oy,ox]      (bias, res nullable; res is [B,Cout,Ho,Wo]) */
int cf_conv2d(const float* x1, int C1, const float* x2, int C2, const float* wt, long w_bstride, const float* bias,
              const float* res, float* out, int out_ctotal, int out_coff, int B, int H, int W, int Cout, int KH,
              int KW, int stride, int pad_h, int pad_w, int act, float alpha, void* stream);

/* Same operator on the f16 MFMA with a 3-term hi/lo operand split (conv_f16s.hip): ~2^-22 relative operand error
 * (fp32-class accuracy, measured in DESIGN.md) at up to 5.3x the fp32-MFMA rate.  Supported: 3x3 pad 1 or 1x1 pad 0 at
 * stride 1 or 2, and the separable 1x5 pad (0,2) / 5x1 pad (2,0) convolutions of RAFT's SepConvGRU at stride 1; anything
 * else returns CF_ERR_ARG (the caller decides: cf_conv2d is the exact fp32 operator).  `wpk` = weights packed by the host
 * in MFMA fragment order as fp16 hi/lo planes and pre-scaled by 2^s (cineflow/ops.py pack_conv_weight_f16s);
 * pass alpha * 2^-s as `alpha`.  With a second input (C2 > 0) whose split C1 is not a multiple of the chunk size (16
 * channels for 3x3, 32 otherwise) the weights must be packed split-aware (pack_conv_weight_f16s(w, c1=C1): x1's channels
 * padded to whole chunks) -- the kernel switches input pointers at a chunk boundary.
 * Sizes: the kernel addresses its inputs with 32-bit offsets; ONE SAMPLE of x1 / x2 must stay below 2 GiB, a batch of any
 * size is cut into sub-batches inside the library (same kernel, same numbers).
 * Activation range: |x| < 65504 (fp16's range for the hi half).  Larger magnitudes, Inf and NaN are not clamped: they come
 * out as NaN, as loudly as in an fp32 convolution fed with NaN, and stay inside their own sample (the zero-weight channel tail of a chunk
 * is never fetched, so nothing of sample b + 1 reaches sample b).  cf_count_out_of_range is the debug check.  Below 2^-14 |x| the lo half goes subnormal (absolute error
 * 2^-25), which is under the fp32 rounding of the sum for the normalised activations this path is built for.
 * gn_ws (nullable, 2*B*gn_groups doubles): on return it holds the GroupNorm / InstanceNorm statistics (sum, sum of squares
 * per (sample, group)) of the OUTPUT, accumulated in the conv epilogue (or by a statistics pass when a workgroup spans
 * several samples); feed it to cf_group_norm_apply.  Requires a dense output (out_coff 0, out_ctotal == Cout).
 * A NEGATIVE gn_groups (= -groups) declares that gn_ws already holds zeros (slices of a pool the caller cleared with one
 * memset): the per-launch memset is skipped. */
int cf_conv2d_f16s(const float* x1, int C1, const float* x2, int C2, const void* wpk, const float* bias, const float* res,
                   float* out, int out_ctotal, int out_coff, int B, int H, int W, int Cout, int KH, int KW, int stride,
                   int pad_h, int pad_w, int act, float alpha, double* gn_ws, int gn_groups, void* stream);
/* Product mode of the f16-MFMA convolutions launched FROM THE CALLING THREAD afterwards (thread-local; returns the previous mode):
 * 3 (default) = hi/lo operand split, three MFMAs per k-step, f32-class results; 1 = hi x hi only, i.e. operands rounded to fp16 with fp32
 * accumulation -- what the reference's fp16 autocast gives the segmentation path under mixed_precision=True
 * (nnunet/network_architecture/neural_network.py:140-146; the flow path forces it off, SegFlowGaussian.py:2905-2909).  In mode 1 every
 * cf_conv2d_f16s* / cf_conv_transpose2d_k2s2_f16s call takes the one-tile kernel's hi x hi instantiation (a third of the MFMAs) and
 * cf_conv2d_wino_ok answers 0. */
int cf_conv_terms(int terms);
int cf_conv_transpose2d_k2s2_f16s(const float* x, const void* wpk, const float* bias, float* out, int out_ctotal,
                                  int out_coff, int B, int Cin, int H, int W, int Cout, float alpha, double* gn_ws, int gn_groups,
                                  void* stream);

/* nn.ConvTranspose2d(k=2, s=2) of PatchExpand2DGroup (lib/utils.py:1982-1994) and Generic_UNet.tu
 * (generic_UNet.py:343-345).  w is the torch layout [Cin][Cout][2][2] (no transposition needed); bias nullable.
 * out is written into channels [out_coff, out_coff+Cout) of a [B,out_ctotal,2H,2W] tensor. */
int cf_conv_transpose2d_k2s2(const float* x, const float* w, const float* bias, float* out, int out_ctotal,
                             int out_coff, int B, int Cin, int H, int W, int Cout, void* stream);

/* ---------------------------------------------------------------- normalisation
 * nn.GroupNorm(groups, C) / nn.InstanceNorm2d(C, affine) (groups == C) + activation + residual, two launches
 * (fp64 statistics, then apply).  ws: caller workspace of 2*B*groups doubles.  gamma/beta nullable (1 / 0).
 * out = act(gn(x) [+ res]) [+ res]  per res_mode.  In-place (out == x) allowed. */
int cf_group_norm(const float* x, const float* gamma, const float* beta, const float* res, float* out, int B, int C,
                  int HW, int groups, float eps, int act, int res_mode, double* ws, void* stream);

/* The apply pass of cf_group_norm alone; `ws` holds the statistics (from cf_conv2d_f16s' fused epilogue). */
int cf_group_norm_apply(const float* x, const float* gamma, const float* beta, const float* res, float* out, int B, int C,
                        int HW, int groups, float eps, int act, int res_mode, const double* ws, void* stream);
/* The same apply pass when the residual is a raw convolution output that still awaits ITS OWN GroupNorm (the 1x1 conv + GroupNorm
 * `downsample` branch of DoubleConv, nnunet/lib/utils.py:1192-1195, :1208-1213): res is normalised on the fly with res_ws (its
 * {sum, sum of squares} pairs, same groups and eps) and res_gamma / res_beta, saving that branch's own apply pass. */
int cf_group_norm_apply_res_norm(const float* x, const float* gamma, const float* beta, const float* res, float* out, int B, int C,
                                 int HW, int groups, float eps, int act, int res_mode, const double* ws, const double* res_ws,
                                 const float* res_gamma, const float* res_beta, void* stream);
/* Deferred normalisation: instead of running the apply pass on a convolution output, hand its per-(sample, channel) coefficients to
 * the consuming convolution.  cf_group_norm_coef: ws ({sum, sum of squares} pairs) -> coef float [B][3][C] = {mean, rstd * gamma,
 * beta}.  cf_conv2d_f16s_prenorm: 3x3 / stride 1 / pad 1 convolution of lrelu((x - mean) * scale + shift, in_slope) (in_slope < 0:
 * GELU, for the second convolution of a DoubleConv, nnunet/lib/utils.py:1182-1215) -- the second
 * convolution of a Generic_UNet stage, nnunet/network_architecture/generic_UNet.py:79-144, consuming the first one's raw output --
 * dense output, optional fused statistics as in cf_conv2d_f16s.  Built for the vector-staging layer shapes only;
 * cf_conv2d_f16s_prenorm_ok(B, C, H, W, Cout) returns 1 when a shape qualifies (no launch), and the call fails otherwise. */
int cf_group_norm_coef(const double* ws, const float* gamma, const float* beta, int B, int C, int HW, int groups, float eps, float* coef,
                       void* stream);
/* Deferred normalisation into a 1x1 head: out[b][k][p] = bias[k] + sum_c w[k][c] * lrelu((x[b][c][p] - mean) * scale + shift, slope) with
 * coef from cf_group_norm_coef -- Generic_UNet's seg_outputs[-1] (a bias-free 1x1 convolution, generic_UNet.py:405-408) behind the last
 * decoder convolution's InstanceNorm + LeakyReLU, in one pass over that convolution's raw output.  K in {2, 4, 8}, HW % 4 == 0, fp32 FMAs
 * in channel order; w is the [K][C] weight, bias may be NULL. */
int cf_norm_head_1x1(const float* x, const float* coef, float slope, const float* w, const float* bias, float* out, int B, int C, int HW,
                     int K, void* stream);
/* Tuning / A-B knob without a reference counterpart: the short-K layers on large maps (3x3, stride 1, 32 or 64 output channels, >= 1024 tiles of
 * 16 / 8 rows x 32 pixels) run on a persistent software-pipelined kernel (csrc/conv_stream.hip) that sums the taps of a chunk in (kx, ky)
 * instead of (ky, kx) order: outputs equal the one-tile-per-workgroup kernel's to fp32 summation-order noise (<= 4e-6 of the output scale).
 * level 0 routes everything back to conv_f16s (also CF_CONV_STREAM=0), 1 (default) takes the shapes measured faster there (all but the
 * 64-channel layers with deferred input normalisation), 2 every shape the kernel can run.  Returns the previous level. */
int cf_conv_stream_enable(int on);
int cf_conv2d_f16s_prenorm_ok(int B, int C, int H, int W, int Cout);
int cf_conv2d_f16s_prenorm(const float* x, int C, const float* in_norm, float in_slope, const void* wpk, const float* bias, float* out,
                           int B, int H, int W, int Cout, float alpha, double* gn_ws, int gn_groups, void* stream);

/* The same operator (nn.Conv2d 3x3 / stride 1 / padding 1 of DoubleConv, nnunet/lib/utils.py:1182-1215; ConvDropoutNormNonlin,
 * generic_UNet.py:26-69; ConvGRUCell's gate convolutions, convGRU.py:57-66) as a ROW Winograd F(2,3) on the f16 MFMA with the 3-term hi/lo
 * split (csrc/conv_wino.hip): the input transform (d0 - d2, d1 + d2, d2 - d1, d1 - d3) runs in fp32 before the split, the kernel transform
 * (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2) on the host in fp64 (cineflow/ops.py pack_conv_weight_wino -- `wpk` is NOT the
 * direct kernels' packing), 12 instead of 18 MFMA k-steps per pair of output columns; results equal cf_conv2d_f16s' to fp32 summation
 * noise (3-5e-7 of max|y| against fp64 at 128 / 256 channels).  Built for Cout in whole 128-channel blocks (or a last block >= 96: the
 * U-Net's 480), W % 16 == 0, H a multiple of the tile rows (4 / 8 at W >= 32, 8 / 16 at W = 16), 16-byte aligned inputs, one sample of
 * each tensor < 2 GiB; arguments as cf_conv2d_f16s (cat[x1, x2] packed split-aware when C1 % 16 != 0; alpha carries 2^-s; fused GroupNorm
 * statistics with gn_ws), the _prenorm form as cf_conv2d_f16s_prenorm.  cf_conv2d_wino_ok(B, C1, C2, H, W, Cout, prenorm) returns 1 when
 * a shape qualifies (no launch); the calls fail with CF_ERR_ARG otherwise (the caller stays on cf_conv2d_f16s).  CF_CONV_WINO=0 in the
 * environment makes cf_conv2d_wino_ok answer 0 for every shape (A/B knob). */
int cf_conv2d_wino_ok(int B, int C1, int C2, int H, int W, int Cout, int prenorm);
/* route level (tests, A/B runs): 0 = off, 1 = automatic (default: the persistent wave-specialised kernel where a layer has at least two items per
 * CU, else one tile per workgroup), 2 / 4 = force the one-tile kernel with 2 / 4 unit tiles per wave, 8 = force the persistent kernel -- each
 * where the geometry allows (initial value from CF_CONV_WINO).  Returns the previous level. */
int cf_conv_wino_enable(int level);
int cf_conv2d_wino(const float* x1, int C1, const float* x2, int C2, const void* wpk, const float* bias, const float* res, float* out,
                   int out_ctotal, int out_coff, int B, int H, int W, int Cout, int act, float alpha, double* gn_ws, int gn_groups,
                   void* stream);
int cf_conv2d_wino_prenorm(const float* x, int C, const float* in_norm, float in_slope, const void* wpk, const float* bias, float* out,
                           int B, int H, int W, int Cout, float alpha, double* gn_ws, int gn_groups, void* stream);

/* nn.LayerNorm(C) over the channel axis of channel-first tokens x [B,C,N] (lib/vit_transformer.py:1257,1261,1265);
 * the residual add is done by the preceding conv epilogue. */
int cf_layer_norm_cf(const float* x, const float* gamma, const float* beta, float* out, int B, int C, int N, float eps,
                     void* stream);

/* ---------------------------------------------------------------- attention
 * nn.MultiheadAttention core (need_weights path: q scaled by 1/sqrt(d) first), channel-first:
 * q [B,heads*d,Nq], k,v [B,heads*d,Nk] with element batch strides q_bs,k_bs,v_bs (so they may be channel slices
 * of a fused projection buffer); out [B,heads*d,Nq] contiguous.  d in {8,16,32,64}; Nq,Nk multiples of 32. */
int cf_attention_cf(const float* q, long q_bs, const float* k, long k_bs, const float* v, long v_bs, float* out, int B,
                    int heads, int d, int Nq, int Nk, void* stream);
/* cf_attention_cf for sequences whose length is not a multiple of 32: the caller zero-pads q / k / v to Nq, Nk (multiples of 32); only
 * the first nk_valid keys take part in the softmax (nk_valid in the last 32-key block); padded query rows of `out` are garbage. */
int cf_attention_cf_masked(const float* q, long q_bs, const float* k, long k_bs, const float* v, long v_bs, float* out, int B,
                           int heads, int d, int Nq, int Nk, int nk_valid, void* stream);

/* ---------------------------------------------------------------- ConvGRU gating (convGRU.py:60-68)
 * gates [B,2C,HW] = sigmoid(conv_gates(...)) with channels [0,C) = reset, [C,2C) = update. */
int cf_gru_reset_mul(const float* gates, const float* h, float* rh, int B, int C, int HW, void* stream);
int cf_gru_blend(const float* gates, const float* h, const float* cand, float* out, int B, int C, int HW, void* stream);

/* ---------------------------------------------------------------- small tensor plumbing kernels */
#define CF_OP_ADD 0
#define CF_OP_SUB 1
#define CF_OP_MUL 2
/* out[i] = a[i] op b[i % b_period]  (b_period = n for no broadcast) */
int cf_binary(int op, const float* a, const float* b, float* out, long n, long b_period, void* stream);
/* Debug aid for the f16-split convolutions (no reference counterpart: the reference's autocast has no range check either): *counter +=
 * number of elements of x[0..n) that are NaN / Inf or have |x| >= limit.  cineflow.ops runs it on every cf_conv2d_f16s* input when
 * CF_F16S_RANGE_CHECK=1 (limit 65504, the largest fp16 value) and reports through ops.f16s_range_violations(); inputs outside the range come
 * out of those kernels as NaN (see cf_conv2d_f16s).  The exact fp32 route for such data is cf_conv2d (cineflow.ops.set_conv_mode("f32")). */
int cf_count_out_of_range(const float* x, long n, float limit, unsigned long long* counter, void* stream);
/* dst[b, dst_coff + c, :] = act(src[b, src_coff + c, :])  for c < C : cat / split / activation of channel slices */
int cf_copy_channels(const float* src, int src_ctotal, int src_coff, float* dst, int dst_ctotal, int dst_coff, int B,
                     int C, int HW, int act, void* stream);
/* coords_grid (published RAFT; call site SegFlowGaussian.py:838-839): out [B,2,H,W], channel 0 = x, 1 = y */
int cf_coords_grid(float* out, int B, int H, int W, void* stream);
/* dst [N,h,w] = src [N,H,W][y0:y0+h, x0:x0+w]   (Processor.crop_data, processor.py:134-138) */
int cf_crop2d(const float* src, float* dst, int N, int H, int W, int y0, int x0, int h, int w, void* stream);
/* dst [N,H,W] = zero-pad of src [N,h,w] placed at (y0,x0)  (Processor.uncrop_no_registration, processor.py:178-186) */
int cf_pad2d(const float* src, float* dst, int N, int h, int w, int y0, int x0, int H, int W, void* stream);

/* ---------------------------------------------------------------- sliding-window / TTA (neural_network.py)
 * _internal_maybe_mirror_and_pred_2D :596-615, one term: acc += weight * flip(softmax_K(flip-input logits)).
 * logits [B,K,H,W] are the network output on the (possibly flipped) input; flips undo the mirroring. */
int cf_tta_accumulate(const float* logits, float* acc, int B, int K, int H, int W, int flip_h, int flip_w, float weight,
                      void* stream);
int cf_flip2d(const float* src, float* dst, int N, int H, int W, int flip_h, int flip_w, void* stream);
/* :711-731: agg[:, lx:lx+ph, ly:ly+pw] += pred * g ; cnt[...] += g   (g nullable = ones). pred [K,ph,pw]. */
int cf_tile_accumulate(const float* pred, const float* gauss, float* agg, float* cnt, int K, int X, int Y, int lx,
                       int ly, int ph, int pw, void* stream);
/* :741-744: probs = agg/cnt ; seg = argmax_K probs (first maximum).  seg uint8 [X,Y]. */
int cf_tile_finalize(const float* agg, const float* cnt, float* probs, uint8_t* seg, int K, int X, int Y, void* stream);
/* 3-D twins: _internal_maybe_mirror_and_pred_3D (neural_network.py:506-571; logits/acc [B,K,D,H,W], the three flags undo
 * the mirroring of axes 2,3,4) and the tile aggregation of _internal_predict_3D_3Dconv_tiled (:381-398; pred [K,px,py,pz],
 * gauss [px,py,pz] nullable, agg/cnt [K,X,Y,Z]).  cf_tile_finalize serves 3-D volumes as (K, X, Y*Z). */
int cf_tta_accumulate_3d(const float* logits, float* acc, int B, int K, int D, int H, int W, int flip_d, int flip_h, int flip_w,
                         float weight, void* stream);
int cf_flip3d(const float* src, float* dst, int N, int D, int H, int W, int flip_d, int flip_h, int flip_w, void* stream);
int cf_tile_accumulate_3d(const float* pred, const float* gauss, float* agg, float* cnt, int K, int X, int Y, int Z, int lx, int ly,
                          int lz, int px, int py, int pz, void* stream);
/* argmax over K of [B,K,HW] -> uint8 [B,HW] */
int cf_argmax_channels(const float* x, uint8_t* out, int B, int K, int HW, void* stream);

/* ---------------------------------------------------------------- export post-processing
 * remove_all_but_the_largest_connected_component, nnunet/postprocessing/connected_components.py:51-107, on the device.
 * scipy.ndimage.label's default structure (face neighbours).  labels int32 [n]: 0 = outside the region, else 1 + the
 * smallest voxel index of the component once cf_cc_sweep has converged (*changed stays 0 after a sweep; the caller
 * zeroes it before each sweep and reads it back).  `classes`: HOST array of the 1..8 label values forming the region. */
int cf_cc_init(const uint8_t* image, int* labels, long n, const uint8_t* classes, int nclasses, void* stream);
int cf_cc_sweep(int* labels, int D, int H, int W, int* changed, void* stream);
/* counts int32 [n], zero-initialised by the caller: counts[l-1] += 1 for every voxel with label l */
int cf_cc_count(const int* labels, int* counts, long n, void* stream);
/* image[i] = 0 where the voxel's component size differs from max_count and (min_valid < 0 or size*volume_per_voxel < min_valid) */
int cf_cc_remove(uint8_t* image, const int* labels, const int* counts, long n, int max_count, double volume_per_voxel,
                 double min_valid, void* stream);

/* resample_data_or_seg, nnunet/preprocessing/preprocessing.py:111-200 (export path, orders 0 and 1): src [N,X,Y,Z] ->
 * dst [N,X2,Y2,Z2], sampling at src = (n/n2)*(dst+0.5)-0.5 with edge clamping (skimage resize mode='edge',
 * map_coordinates mode='nearest'); per axis linear (1) or nearest (0). */
int cf_resize3d(const float* src, float* dst, int N, int X, int Y, int Z, int X2, int Y2, int Z2, int linear_x, int linear_y,
                int linear_z, void* stream);

/* Stem convolutions (nn.Conv2d with 1, 2 or 6 input channels, 3x3 pad 1 or 1x1, stride 1, e.g. Generic_UNet's first conv,
 * nnunet/network_architecture/generic_UNet.py:70-86, and the encoders' first DoubleConv, nnunet/lib/utils.py:1175-1210) as a direct
 * fp32 kernel: weight fp32 [Cout][Cin][K][K] (the checkpoint layout), out dense [B][Cout][H][W].  gn_ws (may be NULL): device fp64
 * [B][gn_groups][2] = {sum, sum of squares} of the output per (sample, group), zeroed here; gn_groups <= 64 divides Cout. */
int cf_conv2d_small_cin(const float* x, const float* weight, const float* bias, float* out, int B, int Cin, int H, int W, int Cout, int K,
                        double* gn_ws, int gn_groups, void* stream);
/* Head convolutions (nn.Conv2d, 3x3 pad 1 stride 1, 1..4 OUTPUT channels: the flow head `final_conv` of Decoder2D,
 * nnunet/lib/decoder_alt.py:890-892, and FlowHead.conv2 of the published RAFT update block) as a direct, exact fp32 kernel:
 * out = conv(x) + bias (+ res): weight fp32 [Cout][Cin][3][3] (checkpoint layout), x [B][Cin][H][W], res / out dense [B][Cout][H][W];
 * bias and res may be NULL. */
int cf_conv2d_small_cout(const float* x, const float* weight, const float* bias, const float* res, float* out, int B, int Cin, int H, int W,
                         int Cout, void* stream);

/* ---------------------------------------------------------------- test-time preprocessing (the step before the path)
 * create_nonzero_mask, nnunet/preprocessing/cropping.py:25-32: mask[v] = any_c data[c][v] != 0 (uint8 [V]). */
int cf_nonzero_mask(const float* data, int C, long V, uint8_t* mask, void* stream);
/* scipy.ndimage.binary_fill_holes of cropping.py:31 -- `labels` are the converged cf_cc_init/cf_cc_sweep labels of the BACKGROUND
 * (class value 0) of `mask`; background components without a voxel on the array border become 1.  touch: int32 [D*H*W] scratch.
 * ndim 3: all six faces are border; ndim 2 (D == 1): the four edges. */
int cf_fill_holes(uint8_t* mask, const int* labels, int* touch, int D, int H, int W, int ndim, void* stream);
/* get_bbox_from_mask, cropping.py:47-55: bbox (device int32 [6]) = {min z, max z, min y, max y, min x, max x} over mask != 0
 * (inclusive maxima; {INT_MAX, -1, ...} for an empty mask).  Synchronises the stream once for its initialisation copy. */
int cf_mask_bbox(const uint8_t* mask, int D, int H, int W, int* bbox, void* stream);
/* One axis of skimage.transform.resize(order=3, mode='edge', anti_aliasing=False) (= scipy.ndimage.zoom(order=3, mode='nearest',
 * grid_mode=True)) as used by resample_data_or_seg, preprocessing.py:111-200: src fp64 [outer][n][inner] -> dst [outer][m][inner]. */
int cf_spline3_resample_axis(const double* src, double* dst, long outer, int n, long inner, int m, void* stream);
/* resize's clip=True: range of every slab (c, s) of x fp64 [C][A][S][B] -> minmax fp64 [C*S][2]; then y (same layout, resampled
 * A and B) clipped to its slab's range (minmax may be NULL: no clipping) and rounded to fp32. */
int cf_slab_minmax_chunks(int C, int A, int S, long B);   /* k: `partial` of cf_slab_minmax is fp64 [C*S][k][2] scratch */
int cf_slab_minmax(const double* x, int C, int A, int S, long B, double* minmax, double* partial, void* stream);
int cf_slab_clip_to_f32(const double* y, float* out, int C, int A, int S, long B, const double* minmax, void* stream);
/* Intensity normalisation, preprocessing.py:274-320.  cf_masked_moments: out3 (device fp64) = {sum, sum of squares, count} over the
 * voxels with seg >= 0 (seg may be NULL: all) and, when use_range, lo < x < hi.  cf_normalize, in place:
 * x = ((clip ? clamp(x, lo, hi) : x) - sub) / div, then 0 where zero_outside and seg < 0. */
int cf_masked_moments(const float* x, const float* seg, long n, int use_range, float lo, float hi, double* out3, void* stream);
int cf_normalize(float* x, const float* seg, long n, int clip, float lo, float hi, float sub, float div, int zero_outside, void* stream);
/* preprocessing.py:251 `data[np.isnan(data)] = 0`, in place */
int cf_nan_to_zero(float* x, long n, void* stream);
/* batchgenerators resize_segmentation (orders > 0): dst[i] = value where the resized indicator src[i] >= thr */
int cf_assign_where_ge(float* dst, const float* src, long n, float thr, float value, void* stream);
/* cropping.py:128-135: seg [C][V] gets `label` where seg == 0 and mask [V] == 0 (a zero-filled seg yields the created one) */
int cf_seg_outside_mask(float* seg, const uint8_t* mask, int C, long V, float label, void* stream);

/* ---------------------------------------------------------------- downstream metrics (consumers of the output layout)
 * ConfusionMatrix.compute, nnunet/evaluation/metrics.py:65-82: counts3 (device u64 [3]) = {TP, FP, FN} of test != 0 vs
 * reference != 0 over n voxels (TN = n - TP - FP - FN). */
int cf_confusion_counts(const uint8_t* test, const uint8_t* reference, long n, unsigned long long* counts3, void* stream);
/* all classes of nnunet/compute_metrics.py:96-106 in one pass: hist (device u64 [K*K + 1]), hist[t*K + r] = #voxels with test label t
 * and reference label r; hist[K*K] counts voxels with a label >= K.  K <= 16. */
int cf_label_confusion(const uint8_t* test, const uint8_t* reference, long n, int K, unsigned long long* hist, void* stream);
/* medpy.metric.binary.__surface_distances (behind metrics.py:323-392): border voxels of mask [D,H,W] (mask ^ binary_erosion(mask,
 * generate_binary_structure(ndim, 1))) compacted as int32 (z, y, x) triples into coords (capacity 3*D*H*W); *count (device int32)
 * receives their number.  Then dist[i] = min_j ||spacing * (a[i] - b[j])|| in fp64 (= distance_transform_edt(~border_b, sampling)
 * read at border_a), and {max, sum} of a non-negative fp64 array -> out2. */
int cf_surface_border(const uint8_t* mask, int D, int H, int W, int ndim, int* coords, int* count, void* stream);
int cf_surface_min_dist(const int* a, int na, const int* b, int nb, double sz, double sy, double sx, double* dist, void* stream);
int cf_max_sum_nonneg(const double* x, long n, double* out2, void* stream);
/* nnunet/compute_jacobian.py:160-186: per label k < K (<= 16) of `labels`, stats (device fp64 [3*K]) = {sum x, count, #(x < 0)} */
int cf_region_stats(const double* x, const uint8_t* labels, long n, int K, double* stats, void* stream);
/* kornia.filters.spatial_gradient3d(mode='diff', order=1) of compute_jacobian.py:146: x [N][D][H][W] -> out [N][3][D][H][W]
 * (component 0 along W, 1 along H, 2 along D), replicate-padded central differences times 0.5; and the sums of |x| per slab
 * (c, s) of x [C][A][S][B] -> sums fp64 [C*S] behind the per-frame means of :147-159. */
int cf_spatial_gradient3d(const float* x, float* out, long N, int D, int H, int W, void* stream);
int cf_slab_abs_sum(const float* x, int C, int A, int S, long B, double* sums, void* stream);
/* skimage.metrics.structural_similarity as nnunet/compute_SSIM.py:91 calls it (2-D, uniform win x win window, sample covariance):
 * S (fp64 [H][W]) = SSIM of every pixel from the five local means with scipy's 'reflect' boundary; C1 = (K1 R)^2, C2 = (K2 R)^2,
 * cov_norm = NP / (NP - 1).  The mean over the interior (cf_region_stats on the cropped map) is the score. */
int cf_ssim_map(const double* im1, const double* im2, int H, int W, int win, double C1, double C2, double cov_norm, double* S, void* stream);

/* SwinCrossAttention (nnunet/lib/swin_cross_attention.py:13-112) after the projections: qk [B][2C][H][W] = the q and k maps of the
 * "rescaler" input, v [B][C][H][W] the value map of the "rescaled" input, bias_table [(2w-1)^2][heads] the learned relative position
 * bias; out [B][C][H][W] = softmax(q k^T / sqrt(C/heads) + bias + shift mask) v per window of w x w tokens (w <= 8 dividing H and W),
 * windows taken on the map cyclically shifted by `shift` and written back un-shifted. */
int cf_window_attention(const float* qk, const float* v, const float* bias_table, float* out, int B, int C, int H, int W, int heads,
                        int window, int shift, void* stream);

/* Processor.get_mean_centroid's masks_to_boxes (training/network_training/processor.py:140-160): per frame n of x [N][H][W]
 * (uint8, or float32 when is_float) boxes[n] = {x1, y1, x2, y2} of the non-zero pixels, or {-1,-1,-1,-1} for an all-zero frame. */
int cf_frame_boxes(const void* x, int is_float, int* boxes, int N, int H, int W, void* stream);
/* SpatialTransformerContour.forward(new_locs, original) (network_architecture/integration.py:5-34) as get_strain.py calls it:
 * out[b,c,p] = bilinear sample (align_corners, zeros outside) of field [B][C][H][W] at the contour point (pts[b,0,p], pts[b,1,p]) --
 * channel 0 is normalised by W - 1 and used as grid x, channel 1 by H - 1 as grid y (the reference's shape[~i]). */
int cf_sample_points_2d(const float* field, const float* pts, float* out, int B, int C, int H, int W, int P, void* stream);

/* ---------------------------------------------------------------- measurement hooks (bench.py only; no reference analogue)
 * cf_profile_enable(n): pre-create n event pairs and time every conv / CorrVolume launch with a (start, stop) pair that
 * brackets exactly that kernel on its own stream (hipExtLaunchKernelGGL); 0 disables.  Kernel ids:
 * 0,1,2 = conv_igemm MT 1,2,4 (work = flops); 3,4,5 = corr_volume_p7 stride 1,2,4 (work = algorithmic bytes);
 * 6 = conv_f16s (work = flops); 7 = RAFT all-pairs volume + pyramid pooling, 8 = RAFT correlation lookup, 9 = convex
 * upsampling, 10 = GroupNorm apply passes, 11 = 2-D bilinear warp, 12 = 2-D label warp, 13 = 2-D Jacobian determinant (work =
 * algorithmic bytes), 14 = direct 3x3 convolution to <= 4 channels (bytes), 15 = conv_stream, 16 = conv_wino (work = direct-form flops).
 * cf_profile_read sums kernel durations [ms], work and launches since the last cf_profile_reset (it synchronises: call it
 * outside the timed region). */
int cf_profile_enable(int max_launches);
int cf_profile_reset(void);
int cf_profile_read(int kernel_id, double* total_ms, double* total_work, long* launches);

#ifdef __cplusplus
}
#endif
#endif /* CINEFLOW_H */
