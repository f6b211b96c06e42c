// VoxelMorph spatial-transformer family: bilinear warp, scaling-and-squaring, fused label propagation,
// fused memory-encoder input, Jacobian determinant.  All HBM-bound: one thread per output pixel, flow read
// once (coalesced along W), the four taps served by L1/L2 (neighbouring lanes sample neighbouring pixels).
#include "common.h"

namespace cf {

// out[b,c,i,j] for all c: flow read once per pixel, taps reused across channels.
__global__ void __launch_bounds__(256) warp_bilinear_2d_kernel(const float* __restrict__ flow,
                                                               const float* __restrict__ src,
                                                               float* __restrict__ out, int B, int C, int H, int W,
                                                               float add_scale /*0: out=warp ; else out = src_add + warp*/,
                                                               const float* __restrict__ addend) {
    const long HW = (long)H * W;
    const long total = (long)B * HW;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int b = (int)(idx / HW);
        int p = (int)(idx - (long)b * HW);
        int i = p / W, j = p - i * W;
        const float* fb = flow + (long)b * 2 * HW;
        float y = st_coord((float)i, fb[p], (float)(H - 1));
        float x = st_coord((float)j, fb[HW + p], (float)(W - 1));
        Taps t = make_taps(y, x, H, W);
        const float* sb = src + (long)b * C * HW;
        float* ob = out + (long)b * C * HW;
        for (int c = 0; c < C; ++c) {
            float v = sample_taps(sb + (long)c * HW, t, W);
            if (addend) v = __fadd_rn(addend[(long)b * C * HW + (long)c * HW + p], v);
            ob[(long)c * HW + p] = v;
        }
    }
}

__global__ void __launch_bounds__(256) scale_kernel(const float* __restrict__ in, float* __restrict__ out, long n, float s) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i] * s;
}

// one_hot -> warp -> argmax without materialising the K planes: the warped one-hot value of class k is the sum of
// the tap weights whose label equals k.  Accumulation order per class follows sample_taps (nw, ne, sw, se).
__global__ void __launch_bounds__(256) warp_labels_2d_kernel(const float* __restrict__ flow,
                                                             const uint8_t* __restrict__ labels,
                                                             uint8_t* __restrict__ out, int T, int B, int K, int H, int W) {
    const long HW = (long)H * W;
    const long total = (long)T * B * HW;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long tb = idx / HW;
        int p = (int)(idx - tb * HW);
        int b = (int)(tb % B);
        int i = p / W, j = p - i * W;
        const float* fb = flow + tb * 2 * HW;
        float y = st_coord((float)i, fb[p], (float)(H - 1));
        float x = st_coord((float)j, fb[HW + p], (float)(W - 1));
        Taps t = make_taps(y, x, H, W);
        const uint8_t* lb = labels + (long)b * HW + (long)t.y0 * W + t.x0;
        int l00 = t.v00 ? lb[0] : -1, l01 = t.v01 ? lb[1] : -1, l10 = t.v10 ? lb[W] : -1, l11 = t.v11 ? lb[W + 1] : -1;
        float best = -1.f;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            float v = __fadd_rn(__fadd_rn(__fadd_rn(l00 == k ? t.w00 : 0.f, l01 == k ? t.w01 : 0.f), l10 == k ? t.w10 : 0.f),
                                l11 == k ? t.w11 : 0.f);
            if (v > best) { best = v; arg = k; }
        }
        out[idx] = (uint8_t)arg;
    }
}

__global__ void __launch_bounds__(256) memory_input_kernel(const float* __restrict__ x0, const float* __restrict__ xt,
                                                           const float* __restrict__ cum, float* __restrict__ out, int B,
                                                           int H, int W) {
    const long HW = (long)H * W;
    const long total = (long)B * HW;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int b = (int)(idx / HW);
        int p = (int)(idx - (long)b * HW);
        int i = p / W, j = p - i * W;
        float f0 = cum[(long)b * 2 * HW + p], f1 = cum[(long)b * 2 * HW + HW + p];
        float y = st_coord((float)i, f0, (float)(H - 1));
        float x = st_coord((float)j, f1, (float)(W - 1));
        Taps t = make_taps(y, x, H, W);
        float reg = sample_taps(xt + (long)b * HW, t, W);
        float a = x0[idx];
        float* ob = out + (long)b * 6 * HW + p;
        ob[0] = a;
        ob[HW] = xt[idx];
        ob[2 * HW] = f0;
        ob[3 * HW] = f1;
        ob[4 * HW] = __fsub_rn(a, reg);
        ob[5 * HW] = reg;
    }
}

// np.gradient: central differences in the interior, one-sided at the borders; float64 like numpy.
__global__ void __launch_bounds__(256) jacobian_det_2d_kernel(const float* __restrict__ disp, double* __restrict__ det,
                                                              int B, int H, int W) {
    const long HW = (long)H * W;
    const long total = (long)B * HW;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int b = (int)(idx / HW);
        int p = (int)(idx - (long)b * HW);
        int i = p / W, j = p - i * W;
        const float* d0 = disp + (long)b * 2 * HW;  // phi0 = d0 + i
        const float* d1 = d0 + HW;                  // phi1 = d1 + j
        int iu = i > 0 ? i - 1 : i, id = i < H - 1 ? i + 1 : i;
        int jl = j > 0 ? j - 1 : j, jr = j < W - 1 ? j + 1 : j;
        double hi = (double)(id - iu), hj = (double)(jr - jl);
        // d/d(axis0)
        double p0_a0 = (((double)d0[id * W + j] + id) - ((double)d0[iu * W + j] + iu)) / hi;
        double p1_a0 = (((double)d1[id * W + j] + j) - ((double)d1[iu * W + j] + j)) / hi;
        // d/d(axis1)
        double p0_a1 = (((double)d0[i * W + jr] + i) - ((double)d0[i * W + jl] + i)) / hj;
        double p1_a1 = (((double)d1[i * W + jr] + jr) - ((double)d1[i * W + jl] + jl)) / hj;
        det[idx] = p0_a0 * p1_a1 - p0_a1 * p1_a0;
    }
}

}  // namespace cf

using namespace cf;

extern "C" int cf_warp_bilinear_2d(const float* flow, const float* src, float* out, int B, int C, int H, int W, void* stream) {
    CF_REQUIRE(flow && src && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, "bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    CF_REQUIRE(out != src, "out must not alias src");
    long n = (long)B * H * W;
    hipLaunchKernelGGL(warp_bilinear_2d_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, as_stream(stream), flow, src, out, B, C,
                       H, W, 0.f, (const float*)nullptr);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_vecint_2d(const float* vec, float* out, float* tmp, int B, int H, int W, int nsteps, void* stream) {
    CF_REQUIRE(vec && out && tmp, "null pointer");
    CF_REQUIRE(B > 0 && H > 1 && W > 1 && nsteps >= 0 && nsteps < 31, "bad shape");
    CF_REQUIRE(out != vec && tmp != vec && tmp != out, "buffers must not alias");
    long n = (long)B * 2 * H * W;
    hipStream_t s = as_stream(stream);
    // ping-pong so that the last step lands in `out`
    float* cur = (nsteps % 2 == 0) ? out : tmp;
    float* nxt = (nsteps % 2 == 0) ? tmp : out;
    hipLaunchKernelGGL(scale_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, s, vec, cur, n, 1.0f / (float)(1 << nsteps));
    CF_CHECK_LAUNCH();
    long np = (long)B * H * W;
    for (int it = 0; it < nsteps; ++it) {
        hipLaunchKernelGGL(warp_bilinear_2d_kernel, dim3(flat_grid(np, 256)), dim3(256), 0, s, cur, cur, nxt, B, 2, H, W, 1.f,
                           (const float*)cur);
        CF_CHECK_LAUNCH();
        float* t = cur; cur = nxt; nxt = t;
    }
    return CF_OK;
}

extern "C" int cf_warp_labels_2d(const float* flow, const uint8_t* labels, uint8_t* out, int T, int B, int K, int H, int W,
                                 void* stream) {
    CF_REQUIRE(flow && labels && out, "null pointer");
    CF_REQUIRE(T > 0 && B > 0 && K > 0 && K <= 8 && H > 1 && W > 1, "bad shape");
    long n = (long)T * B * H * W;
    hipLaunchKernelGGL(warp_labels_2d_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, as_stream(stream), flow, labels, out, T, B,
                       K, H, W);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_memory_input(const float* x0, const float* xt, const float* cum, float* out, int B, int H, int W, void* stream) {
    CF_REQUIRE(x0 && xt && cum && out, "null pointer");
    CF_REQUIRE(B > 0 && H > 1 && W > 1, "bad shape");
    long n = (long)B * H * W;
    hipLaunchKernelGGL(memory_input_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, as_stream(stream), x0, xt, cum, out, B, H, W);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_jacobian_det_2d(const float* disp, double* det, int B, int H, int W, void* stream) {
    CF_REQUIRE(disp && det, "null pointer");
    CF_REQUIRE(B > 0 && H >= 2 && W >= 2, "bad shape");
    long n = (long)B * H * W;
    hipLaunchKernelGGL(jacobian_det_2d_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, as_stream(stream), disp, det, B, H, W);
    CF_CHECK_LAUNCH();
    return CF_OK;
}
