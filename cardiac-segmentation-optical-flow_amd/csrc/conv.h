// Internal interface of the implicit-GEMM convolution (conv.hip), shared with corr.hip (all-pairs GEMM).
#pragma once
#include "common.h"

namespace cf {

struct ConvParams {
    const float* x1;      // [B,C1,H,W]
    const float* x2;      // [B,C2,H,W] or nullptr
    const float* wt;      // [K][Cout] (+ b * w_bstride)
    const float* bias;    // [Cout] or nullptr
    const float* res;     // [B,Cout,Ho,Wo] or nullptr
    float* out;           // [B,out_ctotal,Ho*up,Wo*up]
    long w_bstride;
    int C1, C2, B, H, W, Cout, KH, KW, stride, pad_h, pad_w;
    int Ho, Wo;
    int out_ctotal, out_coff;
    int act;
    float alpha;
    int scatter2x2;       // 1: ConvTranspose k2s2 epilogue (GEMM row m = co*4 + dy*2 + dx)
    double* gn_ws;        // optional: accumulate GroupNorm statistics of the OUTPUT (sum, sum of squares per (sample, group))
    int gn_groups;
    int gn_prezeroed = 0;   // gn_ws is already zero (caller-managed pool)
    // optional: x1 is a raw convolution output whose normalisation + LeakyReLU has been deferred to this consumer (conv_f16s.hip,
    // vector staging only): per (sample, channel) {mean, scale, shift}, float [B][3][C1]; value = lrelu((x - mean) * scale + shift)
    const float* in_norm = nullptr;
    float in_slope = 1.0f;   // LeakyReLU slope; in_slope < 0 selects GELU (erf form) instead
    int terms = 3;          // f16-split kernels: 3 = hi/lo split, three MFMAs per k-step (f32-class, the default); 1 = hi x hi only ("mixed precision":
                            // operands rounded to fp16, fp32 accumulation -- the segmentation path under mixed_precision=True, cf_conv_terms)
    int probe = 0;          // 1: run the dispatch and its checks only, launch nothing (capability query)
    int profile_kid = -1;   // >= 0: time this launch under that profile id with `profile_work` instead of the conv's own id / flops
    double profile_work = 0.0;
};

// validates nothing; callers validate.  Returns CF_OK / CF_ERR_LAUNCH.
int launch_conv(const ConvParams& p, hipStream_t s);

// f16 hi/lo-split kernel (conv_f16s.hip)
int conv_terms();           // this thread's cf_conv_terms setting (1 | 3)
bool conv_f16s_supported(const ConvParams& p);
int launch_conv_f16s(const ConvParams& p, const _Float16* wpk, hipStream_t s);

// persistent software-pipelined variant of the f16-split kernel for short-K layers on large maps (conv_stream.hip); the caller zeroes p.gn_ws
bool conv_stream_applicable(const ConvParams& p);
int launch_conv_stream(const ConvParams& p, const _Float16* wpk, hipStream_t s);

// row Winograd F(2,3) form of the f16-split kernel for 3x3 / stride 1 layers with >= 128 output channels (conv_wino.hip); `wpk` is packed
// by cineflow.ops.pack_conv_weight_wino (NOT the direct kernels' packing)
bool conv_wino_applicable(const ConvParams& p);
int launch_conv_wino(const ConvParams& p, const _Float16* wpk, hipStream_t s);

// CorrVolume (radius 4, dilation 1 / 2 / 4) on the f16 MFMA in 2-D banded form (corr_mfma.hip); corr.hip keeps every other shape
bool corr_mfma_applicable(int C, int H, int W, int stride, const float* cur, const float* prev);
int launch_corr_volume_mfma(const float* cur, const float* prev, float* out, int B, int C, int H, int W, int stride, hipStream_t s);

// RAFT all-pairs volume + pyramid in one kernel (allpairs.hip); returns 1 when the shape is not one it is built for
int allpairs_pyramid_fused(const float* f1, const float* f2, float* pyr, int B, int C, int H, int W, int levels, hipStream_t stream);

// GroupNorm statistics pass (norm.hip): ws[2*(b*groups+g)] = sum, +1 = sum of squares, fp64
int launch_gn_stats(const float* x, double* ws, int B, int C, int HW, int groups, hipStream_t s);

}  // namespace cf
