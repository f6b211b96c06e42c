// GroupNorm / InstanceNorm (+activation, +residual) and channel-first LayerNorm.  HBM-bound.
//
// Statistics are accumulated in fp64 (sum, sum of squares): E[x^2]-E[x]^2 is then exact to ~1e-16 relative, so the
// result matches PyTorch's two-pass/Welford CPU kernels to fp32 rounding.  In NCHW a (sample, group) slab is one
// contiguous run of (C/groups)*HW floats, so the statistics pass is a plain segmented reduction.
#include "conv.h"

namespace cf {

// small slabs: one wave per slab, no atomics
__global__ void __launch_bounds__(256) gn_stats_wave_kernel(const float* __restrict__ x, double* __restrict__ ws, long nslabs,
                                                           int L) {
    const int lane = threadIdx.x & 63;
    long slab = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slab >= nslabs) return;
    const float* xs = x + slab * L;
    double s = 0.0, ss = 0.0;
    for (int i = lane; i < L; i += 64) {
        double v = (double)xs[i];
        s += v;
        ss += v * v;
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    if (lane == 0) {
        ws[2 * slab] = s;
        ws[2 * slab + 1] = ss;
    }
}

// large slabs: `segs` blocks per slab, float4 loads, one fp64 atomic pair per block (ws zeroed by the caller)
__global__ void __launch_bounds__(256) gn_stats_block_kernel(const float* __restrict__ x, double* __restrict__ ws, int L,
                                                            int segs) {
    const long slab = blockIdx.x / segs;
    const int seg = blockIdx.x % segs;
    const float* xs = x + slab * L;
    const int L4 = L >> 2;  // L % 4 == 0 guaranteed by the host
    const float4* x4 = reinterpret_cast<const float4*>(xs);
    double s = 0.0, ss = 0.0;
    for (int i = seg * 256 + threadIdx.x; i < L4; i += segs * 256) {
        float4 v = x4[i];
        float a = v.x + v.y, b = v.z + v.w;  // short fp32 pre-sums, then fp64
        float qa = v.x * v.x + v.y * v.y, qb = v.z * v.z + v.w * v.w;
        s += (double)a + (double)b;
        ss += (double)qa + (double)qb;
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[2 * w] = s; red[2 * w + 1] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = red[0] + red[2] + red[4] + red[6];
        double tss = red[1] + red[3] + red[5] + red[7];
        atomicAdd(&ws[2 * slab], ts);
        atomicAdd(&ws[2 * slab + 1], tss);
    }
}

// streaming (non-temporal) accesses of the large-plane apply pass: every element is touched exactly once by this kernel
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
#define NT_LOAD4(p) ([&]() { const nt_f32x4 _v = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(p)); return make_float4(_v.x, _v.y, _v.z, _v.w); }())
#define NT_STORE4(p, a, b, c, d) __builtin_nontemporal_store(nt_f32x4{a, b, c, d}, reinterpret_cast<nt_f32x4*>(p))

// optional normalisation of the residual operand (all NULL: the residual is added as it is)
struct ResNorm {
    const double* ws;
    const float* gamma;
    const float* beta;
};

// Apply pass.  A block works inside ONE (sample, channel) plane, so mean / rstd / gamma / beta are block-uniform scalars
// (computed once per block from the fp64 sums) and the element loop is a pure float4 stream: y = act((x-mean)*rstd*g+b [+res]) [+res].
template <bool VEC>
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ res,
                                                       float* __restrict__ out, const double* __restrict__ ws, int C, int HW,
                                                       int groups, float eps, int act, int res_mode, int segs, ResNorm rn) {
    const long plane = blockIdx.x / segs;  // b*C + c
    const int seg = blockIdx.x % segs;
    const int c = (int)(plane % C);
    const int cpg = C / groups;
    const long slab = plane / cpg;  // b*groups + g
    const double invL = 1.0 / ((double)cpg * HW);
    const double mean_d = ws[2 * slab] * invL;
    double var = ws[2 * slab + 1] * invL - mean_d * mean_d;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mean = (float)mean_d;
    const float g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    // residual that is itself a raw convolution output awaiting ITS GroupNorm (the 1x1 downsample branch of DoubleConv): r is
    // normalised here on the fly, r' = (r - rm) * ra + rb, instead of in a pass of its own (rm = 0, ra = 1, rb = 0: r' = r exactly)
    float rm = 0.f, ra = 1.f, rb = 0.f;
    if (rn.ws) {
        const double rmean = rn.ws[2 * slab] * invL;
        double rvar = rn.ws[2 * slab + 1] * invL - rmean * rmean;
        if (rvar < 0.0) rvar = 0.0;
        const float rrstd = (float)(1.0 / sqrt(rvar + (double)eps)), rg = rn.gamma ? rn.gamma[c] : 1.f;
        rm = (float)rmean;
        ra = rrstd * rg;
        rb = rn.beta ? rn.beta[c] : 0.f;
    }
    const long base = plane * HW;
    if (VEC) {
        const int n4 = HW >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* r4 = res ? reinterpret_cast<const float4*>(res + base) : nullptr;
        float4* o4 = reinterpret_cast<float4*>(out + base);
        for (int i = seg * 256 + threadIdx.x; i < n4; i += segs * 256) {
            float4 v = NT_LOAD4(x4 + i);
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (res_mode != CF_RES_NONE) r = NT_LOAD4(r4 + i);
            float t[4] = {v.x, v.y, v.z, v.w};
            const float rr[4] = {(r.x - rm) * ra + rb, (r.y - rm) * ra + rb, (r.z - rm) * ra + rb, (r.w - rm) * ra + rb};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float y = (t[k] - mean) * rstd * g + bb;
                if (res_mode == CF_RES_BEFORE_ACT) y += rr[k];
                y = act_apply(y, act);
                if (res_mode == CF_RES_AFTER_ACT) y += rr[k];
                t[k] = y;
            }
            NT_STORE4(o4 + i, t[0], t[1], t[2], t[3]);
        }
    } else {
        for (int i = seg * 256 + threadIdx.x; i < HW; i += segs * 256) {
            float y = (x[base + i] - mean) * rstd * g + bb;
            const float r = res_mode != CF_RES_NONE ? (res[base + i] - rm) * ra + rb : 0.f;
            if (res_mode == CF_RES_BEFORE_ACT) y += r;
            y = act_apply(y, act);
            if (res_mode == CF_RES_AFTER_ACT) y += r;
            out[base + i] = y;
        }
    }
}

// small planes (HW < 2048): a block takes PPB whole planes (~4096 elements); one thread per plane turns the fp64 sums into
// (mean, rstd, gamma, beta) in LDS once, then every thread streams its elements with the same formula as the per-plane kernel.
// (The first version recomputed the fp64 mean / sqrt / divide and two 64-bit divisions PER ELEMENT: 0.4-1.7 TB/s.)
constexpr int GN_SMALL_MAXP = 512;
template <bool VEC>
__global__ void __launch_bounds__(256) gn_apply_small_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ res,
                                                             float* __restrict__ out, const double* __restrict__ ws, int C, int HW,
                                                             int groups, float eps, int act, int res_mode, long planes, int ppb, ResNorm rn) {
    __shared__ float coef[GN_SMALL_MAXP][4];
    __shared__ float rcoef[GN_SMALL_MAXP][4];
    const long p0 = (long)blockIdx.x * ppb;
    const int np = (int)(planes - p0 < ppb ? planes - p0 : ppb);
    const int cpg = C / groups;
    const double invL = 1.0 / ((double)cpg * HW);
    for (int t = threadIdx.x; t < np; t += 256) {
        const long pl = p0 + t;
        const int c = (int)(pl % C);
        const long slab = pl / cpg;
        const double mean = ws[2 * slab] * invL;
        double var = ws[2 * slab + 1] * invL - mean * mean;
        if (var < 0.0) var = 0.0;
        coef[t][0] = (float)mean;
        coef[t][1] = (float)(1.0 / sqrt(var + (double)eps));
        coef[t][2] = gamma ? gamma[c] : 1.f;
        coef[t][3] = beta ? beta[c] : 0.f;
        float rm = 0.f, ra = 1.f, rb = 0.f;
        if (rn.ws) {
            const double rmean = rn.ws[2 * slab] * invL;
            double rvar = rn.ws[2 * slab + 1] * invL - rmean * rmean;
            if (rvar < 0.0) rvar = 0.0;
            rm = (float)rmean;
            ra = (float)(1.0 / sqrt(rvar + (double)eps)) * (rn.gamma ? rn.gamma[c] : 1.f);
            rb = rn.beta ? rn.beta[c] : 0.f;
        }
        rcoef[t][0] = rm;
        rcoef[t][1] = ra;
        rcoef[t][2] = rb;
    }
    __syncthreads();
    const long base = p0 * HW;
    const float inv_hw = 1.0f / (float)HW;
    if (VEC) {
        const int n4 = (np * HW) >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* r4 = res ? reinterpret_cast<const float4*>(res + base) : nullptr;
        float4* o4 = reinterpret_cast<float4*>(out + base);
        for (int i = threadIdx.x; i < n4; i += 256) {
            int pl = (int)(((float)(4 * i) + 0.5f) * inv_hw);      // plane of the float4 (HW % 4 == 0: it never straddles two)
            if ((pl + 1) * HW <= 4 * i) ++pl;                       // exact for any HW: correct the float estimate
            if (pl * HW > 4 * i) --pl;
            const float mean = coef[pl][0], rstd = coef[pl][1], g = coef[pl][2], bb = coef[pl][3];
            const float rm = rcoef[pl][0], ra = rcoef[pl][1], rb = rcoef[pl][2];
            const float4 v = x4[i];
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (res_mode != CF_RES_NONE) r = r4[i];
            float t[4] = {v.x, v.y, v.z, v.w};
            const float rr[4] = {(r.x - rm) * ra + rb, (r.y - rm) * ra + rb, (r.z - rm) * ra + rb, (r.w - rm) * ra + rb};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float y = (t[k] - mean) * rstd * g + bb;
                if (res_mode == CF_RES_BEFORE_ACT) y += rr[k];
                y = act_apply(y, act);
                if (res_mode == CF_RES_AFTER_ACT) y += rr[k];
                t[k] = y;
            }
            o4[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
    } else {
        const int n = np * HW;
        for (int i = threadIdx.x; i < n; i += 256) {
            int pl = (int)(((float)i + 0.5f) * inv_hw);
            if ((pl + 1) * HW <= i) ++pl;
            if (pl * HW > i) --pl;
            float y = (x[base + i] - coef[pl][0]) * coef[pl][1] * coef[pl][2] + coef[pl][3];
            const float r = res_mode != CF_RES_NONE ? (res[base + i] - rcoef[pl][0]) * rcoef[pl][1] + rcoef[pl][2] : 0.f;
            if (res_mode == CF_RES_BEFORE_ACT) y += r;
            y = act_apply(y, act);
            if (res_mode == CF_RES_AFTER_ACT) y += r;
            out[base + i] = y;
        }
    }
}

// LayerNorm over C of channel-first tokens [B, C, N].  Block = 64 consecutive tokens (lanes: coalesced along N) x 8 waves,
// wave g owning channels g, g+8, ...: fp64 partial sums per thread, one LDS combine, then the affine pass.  Up to 32
// channels per thread stay in registers between the two passes (C <= 256); larger C re-reads (L2 hits).
constexpr int LN_G = 8, LN_REG = 32;
__global__ void __launch_bounds__(64 * LN_G) layer_norm_cf_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ out, int B,
                                                                 int C, int N, float eps) {
    __shared__ double red[LN_G][64][2];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long t = (long)blockIdx.x * 64 + lane;      // token index over (b, n); a block never straddles... it may: handled per thread
    const bool valid = t < (long)B * N;
    const int b = valid ? (int)(t / N) : 0;
    const int n = valid ? (int)(t - (long)b * N) : 0;
    const float* xb = x + (long)b * C * N + n;
    float* ob = out + (long)b * C * N + n;
    const bool inreg = C <= LN_G * LN_REG;
    float v[LN_REG];
    double s = 0.0, ss = 0.0;
    if (valid) {
        if (inreg) {
#pragma unroll
            for (int k = 0; k < LN_REG; ++k) {
                const int c = g + k * LN_G;
                v[k] = c < C ? xb[(long)c * N] : 0.f;
                s += (double)v[k];
                ss += (double)v[k] * (double)v[k];
            }
        } else {
            for (int c = g; c < C; c += LN_G) {
                const double u = (double)xb[(long)c * N];
                s += u;
                ss += u * u;
            }
        }
    }
    red[g][lane][0] = s;
    red[g][lane][1] = ss;
    __syncthreads();
    s = ss = 0.0;
#pragma unroll
    for (int k = 0; k < LN_G; ++k) {
        s += red[k][lane][0];
        ss += red[k][lane][1];
    }
    if (!valid) return;
    const double mean = s / C;
    double var = ss / C - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mf = (float)mean;
    if (inreg) {
#pragma unroll
        for (int k = 0; k < LN_REG; ++k) {
            const int c = g + k * LN_G;
            if (c < C) ob[(long)c * N] = (v[k] - mf) * rstd * gamma[c] + beta[c];
        }
    } else {
        for (int c = g; c < C; c += LN_G) ob[(long)c * N] = (xb[(long)c * N] - mf) * rstd * gamma[c] + beta[c];
    }
}

}  // namespace cf

namespace cf {
int launch_gn_stats(const float* x, double* ws, int B, int C, int HW, int groups, hipStream_t s) {
    const long nslabs = (long)B * groups;
    const int L = (C / groups) * HW;
    if (L <= 4096 || (L & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) {
        hipLaunchKernelGGL(gn_stats_wave_kernel, dim3(cdiv(nslabs, 4)), dim3(256), 0, s, x, ws, nslabs, L);
    } else {
        if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * nslabs, s) != hipSuccess) { set_error("gn stats: memset failed"); return CF_ERR_LAUNCH; }
        int segs = (int)((L / 4 + 256 * 8 - 1) / (256 * 8));  // ~8 float4 per thread
        if (segs < 1) segs = 1;
        if (segs > 64) segs = 64;
        hipLaunchKernelGGL(gn_stats_block_kernel, dim3((unsigned)(nslabs * segs)), dim3(256), 0, s, x, ws, L, segs);
    }
    if (hipGetLastError() != hipSuccess) { set_error("gn stats: launch failed"); return CF_ERR_LAUNCH; }
    return CF_OK;
}
}  // namespace cf

using namespace cf;

static int gn_apply_launch(const float* x, const float* gamma, const float* beta, const float* res, float* out, int B, int C, int HW,
                           int groups, float eps, int act, int res_mode, const double* ws, hipStream_t s, ResNorm rn = ResNorm{nullptr, nullptr, nullptr});

extern "C" int cf_group_norm(const float* x, const float* gamma, const float* beta, const float* res, float* out, int B, int C,
                             int HW, int groups, float eps, int act, int res_mode, double* ws, void* stream) {
    CF_REQUIRE(x && out && ws, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "bad shape B=%d C=%d HW=%d groups=%d", B, C, HW, groups);
    CF_REQUIRE(res_mode == CF_RES_NONE || res, "residual mode %d without residual tensor", res_mode);
    CF_REQUIRE(act >= CF_ACT_NONE && act <= CF_ACT_SIGMOID, "bad activation %d", act);
    hipStream_t s = as_stream(stream);
    CF_REQUIRE((long)(C / groups) * HW < (1L << 31), "slab too large");
    int rc = launch_gn_stats(x, ws, B, C, HW, groups, s);
    if (rc != CF_OK) return rc;
    return gn_apply_launch(x, gamma, beta, res, out, B, C, HW, groups, eps, act, res_mode, ws, s);
}

// Apply pass only: `ws` already holds the (sum, sum of squares) pairs -- produced by cf_conv2d_f16s' fused epilogue.
extern "C" int cf_group_norm_apply(const float* x, const float* gamma, const float* beta, const float* res, float* out, int B, int C,
                                   int HW, int groups, float eps, int act, int res_mode, const double* ws, void* stream) {
    CF_REQUIRE(x && out && ws, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "bad shape B=%d C=%d HW=%d groups=%d", B, C, HW, groups);
    CF_REQUIRE(res_mode == CF_RES_NONE || res, "residual mode %d without residual tensor", res_mode);
    CF_REQUIRE(act >= CF_ACT_NONE && act <= CF_ACT_SIGMOID, "bad activation %d", act);
    return gn_apply_launch(x, gamma, beta, res, out, B, C, HW, groups, eps, act, res_mode, ws, as_stream(stream));
}

// {mean, scale, shift} per (sample, channel) from the fp64 sums: what a consumer needs to apply the normalisation itself
// (cf_conv2d_f16s_prenorm): y = (x - mean) * scale + shift, scale = rstd * gamma, shift = beta.  coef float [B][3][C].
__global__ void __launch_bounds__(256) gn_coef_kernel(const double* __restrict__ ws, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int B, int C, int HW, int groups, float eps, float* __restrict__ coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C, cpg = C / groups;
    const long slab = (long)b * groups + c / cpg;
    const double invL = 1.0 / ((double)cpg * HW);
    const double mean = ws[2 * slab] * invL;
    double var = ws[2 * slab + 1] * invL - mean * mean;
    if (var < 0.0) var = 0.0;
    float* o = coef + (long)b * 3 * C + c;
    o[0] = (float)mean;
    o[C] = (float)(1.0 / sqrt(var + (double)eps)) * (gamma ? gamma[c] : 1.f);
    o[2 * C] = beta ? beta[c] : 0.f;
}

extern "C" int cf_group_norm_coef(const double* ws, const float* gamma, const float* beta, int B, int C, int HW, int groups, float eps, float* coef,
                                  void* stream) {
    CF_REQUIRE(ws && coef, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0 && (long)B * C < (1L << 31), "bad shape B=%d C=%d HW=%d groups=%d", B, C, HW, groups);
    hipLaunchKernelGGL(gn_coef_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, as_stream(stream), ws, gamma, beta, B, C, HW, groups, eps, coef);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

// Deferred normalisation into a 1x1 head (Generic_UNet's seg_outputs[-1] behind the last decoder stack, generic_UNet.py:405-408):
//   out[b, k, p] = bias[k] + sum_c w[k, c] * lrelu((x[b, c, p] - mean[b, c]) * scale[b, c] + shift[b, c], slope)
// x is the last convolution's raw output.  One read of x (4 C bytes per pixel) and one write of the K planes instead of the apply pass
// (read + write 4 C) followed by a 1x1 convolution (read 4 C): the op is HBM-bound, the K * C FMAs per pixel run in fp32 in channel order.
// Four consecutive pixels per thread (16-byte loads / stores); a block stays inside one sample so the coefficients are block-uniform and
// sit in LDS together with the weights.
template <int K>
__global__ void __launch_bounds__(256) norm_head_1x1_kernel(const float* __restrict__ x, const float* __restrict__ coef, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int C, int HW, float slope,
                                                           int blocks_per_sample) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ float sm[];            // [3 C] coefficients of this sample, then [K C] weights
    const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x - b * blocks_per_sample;
    for (int i = threadIdx.x; i < 3 * C; i += 256) sm[i] = coef[(long)b * 3 * C + i];
    for (int i = threadIdx.x; i < K * C; i += 256) sm[3 * C + i] = w[i];
    __syncthreads();
    const int p = (blk * 256 + threadIdx.x) * 4;
    if (p >= HW) return;
    const float* xb = x + (long)b * C * HW + p;
    f32x4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { const float bv = bias ? bias[k] : 0.f; acc[k] = f32x4{bv, bv, bv, bv}; }
    for (int c = 0; c < C; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long)c * HW);
        const float m = sm[c], sc = sm[C + c], sh = sm[2 * C + c];
        f32x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = (v[j] - m) * sc + sh;
            t[j] = u > 0.f ? u : u * slope;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float wk = sm[3 * C + k * C + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][j] = fmaf(wk, t[j], acc[k][j]);
        }
    }
    float* ob = out + (long)b * K * HW + p;
#pragma unroll
    for (int k = 0; k < K; ++k) *reinterpret_cast<f32x4*>(ob + (long)k * HW) = acc[k];
}

extern "C" int cf_norm_head_1x1(const float* x, const float* coef, float slope, const float* w, const float* bias, float* out, int B, int C,
                                int HW, int K, void* stream) {
    CF_REQUIRE(x && coef && w && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && C <= 1024 && HW > 0 && (HW & 3) == 0 && (K == 2 || K == 4 || K == 8), "bad shape B=%d C=%d HW=%d K=%d (K in {2,4,8}, HW %% 4 == 0)",
               B, C, HW, K);
    CF_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "x and out must be 16-byte aligned");
    CF_REQUIRE(slope >= 0.f, "LeakyReLU slope must be >= 0");
    const int bps = (HW / 4 + 255) / 256;
    const long nblk = (long)B * bps;
    CF_REQUIRE(nblk < (1L << 31), "grid too large");
    const size_t lds = (size_t)(3 + K) * C * sizeof(float);
    hipStream_t s = as_stream(stream);
    if (K == 2) hipLaunchKernelGGL(norm_head_1x1_kernel<2>, dim3((unsigned)nblk), dim3(256), lds, s, x, coef, w, bias, out, C, HW, slope, bps);
    else if (K == 4) hipLaunchKernelGGL(norm_head_1x1_kernel<4>, dim3((unsigned)nblk), dim3(256), lds, s, x, coef, w, bias, out, C, HW, slope, bps);
    else hipLaunchKernelGGL(norm_head_1x1_kernel<8>, dim3((unsigned)nblk), dim3(256), lds, s, x, coef, w, bias, out, C, HW, slope, bps);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

// cf_group_norm_apply with a residual that still awaits its own GroupNorm (same group count and eps): res_ws holds the residual's
// (sum, sum of squares) pairs, res_gamma / res_beta its affine parameters.
extern "C" int cf_group_norm_apply_res_norm(const float* x, const float* gamma, const float* beta, const float* res, float* out, int B, int C,
                                            int HW, int groups, float eps, int act, int res_mode, const double* ws, const double* res_ws,
                                            const float* res_gamma, const float* res_beta, void* stream) {
    CF_REQUIRE(x && out && ws && res && res_ws, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "bad shape B=%d C=%d HW=%d groups=%d", B, C, HW, groups);
    CF_REQUIRE(res_mode == CF_RES_BEFORE_ACT || res_mode == CF_RES_AFTER_ACT, "a normalised residual needs a residual mode, got %d", res_mode);
    CF_REQUIRE(act >= CF_ACT_NONE && act <= CF_ACT_SIGMOID, "bad activation %d", act);
    return gn_apply_launch(x, gamma, beta, res, out, B, C, HW, groups, eps, act, res_mode, ws, as_stream(stream), ResNorm{res_ws, res_gamma, res_beta});
}

static int gn_apply_launch(const float* x, const float* gamma, const float* beta, const float* res, float* out, int B, int C, int HW,
                           int groups, float eps, int act, int res_mode, const double* ws, hipStream_t s, ResNorm rn) {
    if (HW < 2048) {
        const long planes_s = (long)B * C;
        int ppb = 4096 / HW;
        if (ppb < 1) ppb = 1;
        if (ppb > GN_SMALL_MAXP) ppb = GN_SMALL_MAXP;
        const long nblk = (planes_s + ppb - 1) / ppb;
        CF_REQUIRE(nblk < (1L << 31), "grid too large");
        const bool vec_s = (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res)) & 15) == 0;
        if (vec_s)
            hipLaunchKernelGGL((gn_apply_small_kernel<true>), dim3((unsigned)nblk), dim3(256), 0, s, x, gamma, beta, res, out, ws, C, HW, groups, eps,
                               act, res_mode, planes_s, ppb, rn);
        else
            hipLaunchKernelGGL((gn_apply_small_kernel<false>), dim3((unsigned)nblk), dim3(256), 0, s, x, gamma, beta, res, out, ws, C, HW, groups, eps,
                               act, res_mode, planes_s, ppb, rn);
        CF_CHECK_LAUNCH();
        return CF_OK;
    }
    const long planes = (long)B * C;
    const bool vec = (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res)) & 15) == 0;
    const int per_block = vec ? 256 * 4 * 4 : 256 * 4;  // ~4 vector (or scalar) elements per thread
    int asegs = (HW + per_block - 1) / per_block;
    if (asegs < 1) asegs = 1;
    CF_REQUIRE(planes * asegs < (1L << 31), "grid too large");
    if (vec)
        hipLaunchKernelGGL((gn_apply_kernel<true>), dim3((unsigned)(planes * asegs)), dim3(256), 0, s, x, gamma, beta, res, out, ws, C, HW, groups,
                           eps, act, res_mode, asegs, rn);
    else
        hipLaunchKernelGGL((gn_apply_kernel<false>), dim3((unsigned)(planes * asegs)), dim3(256), 0, s, x, gamma, beta, res, out, ws, C, HW, groups,
                           eps, act, res_mode, asegs, rn);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_layer_norm_cf(const float* x, const float* gamma, const float* beta, float* out, int B, int C, int N, float eps,
                                void* stream) {
    CF_REQUIRE(x && gamma && beta && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && N > 0, "bad shape");
    const long total = (long)B * N;
    CF_REQUIRE((total + 63) / 64 < (1L << 31), "too many tokens");
    hipLaunchKernelGGL(layer_norm_cf_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64 * LN_G), 0, as_stream(stream), x, gamma, beta,
                       out, B, C, N, eps);
    CF_CHECK_LAUNCH();
    return CF_OK;
}
