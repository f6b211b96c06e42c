// Small HBM-bound kernels around the convolutions: GRU gating, channel-slice copies (cat/split), TTA and
// sliding-window accumulation, crop / pad.  One pass each, coalesced along the innermost (x) axis.
#include "common.h"

namespace cf {

static thread_local std::string g_last_error;
void set_error(const std::string& s) { g_last_error = s; }

#define GRID_STRIDE(i, n) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

__global__ void __launch_bounds__(256) gru_reset_mul_kernel(const float* __restrict__ gates, const float* __restrict__ h,
                                                           float* __restrict__ rh, int C, long CHW, long total) {
    GRID_STRIDE(i, total) {
        long b = i / CHW, r = i - b * CHW;
        rh[i] = gates[b * 2 * CHW + r] * h[i];
    }
}

__global__ void __launch_bounds__(256) gru_blend_kernel(const float* __restrict__ gates, const float* __restrict__ h,
                                                       const float* __restrict__ cand, float* __restrict__ out, long CHW,
                                                       long total) {
    GRID_STRIDE(i, total) {
        long b = i / CHW, r = i - b * CHW;
        float u = gates[b * 2 * CHW + CHW + r];
        out[i] = (1.0f - u) * h[i] + u * cand[i];
    }
}

__global__ void __launch_bounds__(256) binary_kernel(int op, const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ out, long n, long period) {
    GRID_STRIDE(i, n) {
        float x = a[i], y = b[period == n ? i : i % period];
        out[i] = op == CF_OP_ADD ? x + y : (op == CF_OP_SUB ? x - y : x * y);
    }
}

// debug aid of the f16-split convolutions (CF_F16S_RANGE_CHECK=1): elements that the hi/lo split cannot represent -- non-finite, or
// |x| >= limit (65504 = fp16 max) -- are counted into one device counter; one ballot + one atomic per wave that saw any
__global__ void __launch_bounds__(256) count_out_of_range_kernel(const float* __restrict__ x, long n, float limit, unsigned long long* __restrict__ counter) {
    GRID_STRIDE(i, n) {
        const float v = x[i];
        const bool bad = !(fabsf(v) < limit);            // NaN compares false: counted
        const unsigned long long m = __ballot(bad);
        if (m && (threadIdx.x & 63) == (unsigned)__ffsll((long long)m) - 1u) atomicAdd(counter, (unsigned long long)__popcll(m));
    }
}

__global__ void __launch_bounds__(256) copy_channels_kernel(const float* __restrict__ src, int sct, int sco, float* __restrict__ dst,
                                                           int dct, int dco, int C, int HW, int act, long total) {
    GRID_STRIDE(i, total) {
        int p = (int)(i % HW);
        long r = i / HW;
        int c = (int)(r % C);
        long b = r / C;
        dst[(b * dct + dco + c) * HW + p] = act_apply(src[(b * sct + sco + c) * HW + p], act);
    }
}

__global__ void __launch_bounds__(256) coords_grid_kernel(float* __restrict__ out, int H, int W, long total) {
    const int HW = H * W;
    GRID_STRIDE(i, total) {
        int p = (int)(i % HW);
        int ch = (int)((i / HW) & 1);
        out[i] = ch == 0 ? (float)(p % W) : (float)(p / W);
    }
}

__global__ void __launch_bounds__(256) crop2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int y0,
                                                    int x0, int h, int w, long total) {
    GRID_STRIDE(i, total) {
        int x = (int)(i % w);
        long r = i / w;
        int y = (int)(r % h);
        long n = r / h;
        dst[i] = src[(n * H + y0 + y) * W + x0 + x];
    }
}

__global__ void __launch_bounds__(256) pad2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int y0,
                                                   int x0, int H, int W, long total) {
    GRID_STRIDE(i, total) {
        int x = (int)(i % W);
        long r = i / W;
        int y = (int)(r % H);
        long n = r / H;
        int ys = y - y0, xs = x - x0;
        dst[i] = ((unsigned)ys < (unsigned)h && (unsigned)xs < (unsigned)w) ? src[(n * h + ys) * w + xs] : 0.f;
    }
}

// acc[b,k,y,x] += weight * softmax_k(logits[b,:,yf,xf])  with (yf,xf) the mirrored position
__global__ void __launch_bounds__(256) tta_accumulate_kernel(const float* __restrict__ logits, float* __restrict__ acc, int K, int H,
                                                            int W, int fh, int fw, float weight, long total) {
    const long HW = (long)H * W;
    GRID_STRIDE(i, total) {  // i over (b, y, x)
        int p = (int)(i % HW);
        long b = i / HW;
        int y = p / W, x = p - y * W;
        int ys = fh ? H - 1 - y : y, xs = fw ? W - 1 - x : x;
        const float* lp = logits + b * K * HW + (long)ys * W + xs;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, lp[k * HW]);
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += expf(lp[k * HW] - mx);
        float* ap = acc + b * K * HW + p;
        for (int k = 0; k < K; ++k) ap[k * HW] += weight * (expf(lp[k * HW] - mx) / sum);
    }
}

__global__ void __launch_bounds__(256) flip2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int fh,
                                                    int fw, long total) {
    const long HW = (long)H * W;
    GRID_STRIDE(i, total) {
        int p = (int)(i % HW);
        long n = i / HW;
        int y = p / W, x = p - y * W;
        dst[i] = src[n * HW + (long)(fh ? H - 1 - y : y) * W + (fw ? W - 1 - x : x)];
    }
}

__global__ void __launch_bounds__(256) tile_accumulate_kernel(const float* __restrict__ pred, const float* __restrict__ gauss,
                                                             float* __restrict__ agg, float* __restrict__ cnt, int X, int Y, int lx,
                                                             int ly, int ph, int pw, long total) {
    GRID_STRIDE(i, total) {  // i over (k, py, px)
        int px = (int)(i % pw);
        long r = i / pw;
        int py = (int)(r % ph);
        long k = r / ph;
        float g = gauss ? gauss[py * pw + px] : 1.f;
        long o = (k * X + lx + py) * Y + ly + px;
        agg[o] += pred[i];  // pred already carries the Gaussian weight (mult in _internal_maybe_mirror_and_pred_2D)
        cnt[o] += g;
    }
}


// ---- 3-D twins for _internal_maybe_mirror_and_pred_3D / _internal_predict_3D_3Dconv_tiled (neural_network.py:506-571, :292-430)
__global__ void __launch_bounds__(256) tta_accumulate_3d_kernel(const float* __restrict__ logits, float* __restrict__ acc, int K, int D,
                                                               int H, int W, int fd, int fh, int fw, float weight, long total) {
    const long HW = (long)H * W, V = (long)D * HW;
    GRID_STRIDE(i, total) {  // i over (b, z, y, x)
        long p = i % V, b = i / V;
        int z = (int)(p / HW), q = (int)(p - (long)z * HW);
        int y = q / W, x = q - y * W;
        int zs = fd ? D - 1 - z : z, ys = fh ? H - 1 - y : y, xs = fw ? W - 1 - x : x;
        const float* lp = logits + b * K * V + (long)zs * HW + (long)ys * W + xs;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, lp[k * V]);
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += expf(lp[k * V] - mx);
        float* ap = acc + b * K * V + p;
        for (int k = 0; k < K; ++k) ap[k * V] += weight * (expf(lp[k * V] - mx) / sum);
    }
}

__global__ void __launch_bounds__(256) flip3d_kernel(const float* __restrict__ src, float* __restrict__ dst, int D, int H, int W, int fd,
                                                    int fh, int fw, long total) {
    const long HW = (long)H * W, V = (long)D * HW;
    GRID_STRIDE(i, total) {
        long p = i % V, n = i / V;
        int z = (int)(p / HW), q = (int)(p - (long)z * HW);
        int y = q / W, x = q - y * W;
        dst[i] = src[n * V + (long)(fd ? D - 1 - z : z) * HW + (long)(fh ? H - 1 - y : y) * W + (fw ? W - 1 - x : x)];
    }
}

__global__ void __launch_bounds__(256) tile_accumulate_3d_kernel(const float* __restrict__ pred, const float* __restrict__ gauss,
                                                                float* __restrict__ agg, float* __restrict__ cnt, int X, int Y, int Z,
                                                                int lx, int ly, int lz, int px, int py, int pz, long total) {
    const long pv = (long)px * py * pz;
    GRID_STRIDE(i, total) {  // i over (k, tx, ty, tz)
        long t = i % pv, k = i / pv;
        int tz = (int)(t % pz);
        long r = t / pz;
        int ty = (int)(r % py), tx = (int)(r / py);
        float g = gauss ? gauss[t] : 1.f;
        long o = ((k * X + lx + tx) * Y + ly + ty) * Z + lz + tz;
        agg[o] += pred[i];  // pred already carries the Gaussian weight (mult in _internal_maybe_mirror_and_pred_3D)
        cnt[o] += g;
    }
}

__global__ void __launch_bounds__(256) tile_finalize_kernel(const float* __restrict__ agg, const float* __restrict__ cnt,
                                                           float* __restrict__ probs, uint8_t* __restrict__ seg, int K, long XY) {
    GRID_STRIDE(i, XY) {
        float best = -INFINITY;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            float v = agg[k * XY + i] / cnt[k * XY + i];
            probs[k * XY + i] = v;
            if (v > best) { best = v; arg = k; }
        }
        seg[i] = (uint8_t)arg;
    }
}

__global__ void __launch_bounds__(256) argmax_channels_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int K, long HW,
                                                             long total) {
    GRID_STRIDE(i, total) {
        long b = i / HW, p = i - b * HW;
        float best = -INFINITY;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            float v = x[(b * K + k) * HW + p];
            if (v > best) { best = v; arg = k; }
        }
        out[i] = (uint8_t)arg;
    }
}

}  // namespace cf

using namespace cf;

#define LAUNCH_FLAT(kernel, n, ...)                                                                              \
    hipLaunchKernelGGL(kernel, dim3(flat_grid((n), 256)), dim3(256), 0, as_stream(stream), __VA_ARGS__);         \
    CF_CHECK_LAUNCH();                                                                                           \
    return CF_OK

extern "C" const char* cf_last_error(void) { return g_last_error.c_str(); }
extern "C" int cf_version(void) { return 100; }

extern "C" int cf_gru_reset_mul(const float* gates, const float* h, float* rh, int B, int C, int HW, void* stream) {
    CF_REQUIRE(gates && h && rh, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && HW > 0, "bad shape");
    long chw = (long)C * HW, total = (long)B * chw;
    LAUNCH_FLAT(gru_reset_mul_kernel, total, gates, h, rh, C, chw, total);
}

extern "C" int cf_gru_blend(const float* gates, const float* h, const float* cand, float* out, int B, int C, int HW, void* stream) {
    CF_REQUIRE(gates && h && cand && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && HW > 0, "bad shape");
    long chw = (long)C * HW, total = (long)B * chw;
    LAUNCH_FLAT(gru_blend_kernel, total, gates, h, cand, out, chw, total);
}

extern "C" int cf_binary(int op, const float* a, const float* b, float* out, long n, long b_period, void* stream) {
    CF_REQUIRE(a && b && out, "null pointer");
    CF_REQUIRE(n > 0 && b_period > 0 && b_period <= n && op >= CF_OP_ADD && op <= CF_OP_MUL, "bad arguments");
    LAUNCH_FLAT(binary_kernel, n, op, a, b, out, n, b_period);
}

extern "C" int cf_count_out_of_range(const float* x, long n, float limit, unsigned long long* counter, void* stream) {
    CF_REQUIRE(x && counter && n > 0 && limit > 0.f, "bad arguments");
    LAUNCH_FLAT(count_out_of_range_kernel, n, x, n, limit, counter);
}

extern "C" int cf_copy_channels(const float* src, int src_ctotal, int src_coff, float* dst, int dst_ctotal, int dst_coff, int B,
                                int C, int HW, int act, void* stream) {
    CF_REQUIRE(src && dst, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && HW > 0 && src_coff >= 0 && dst_coff >= 0 && src_coff + C <= src_ctotal && dst_coff + C <= dst_ctotal,
               "channel slice out of range");
    CF_REQUIRE(act >= CF_ACT_NONE && act <= CF_ACT_SIGMOID, "bad activation");
    long total = (long)B * C * HW;
    LAUNCH_FLAT(copy_channels_kernel, total, src, src_ctotal, src_coff, dst, dst_ctotal, dst_coff, C, HW, act, total);
}

extern "C" int cf_coords_grid(float* out, int B, int H, int W, void* stream) {
    CF_REQUIRE(out && B > 0 && H > 0 && W > 0, "bad arguments");
    long total = (long)B * 2 * H * W;
    LAUNCH_FLAT(coords_grid_kernel, total, out, H, W, total);
}

extern "C" int cf_crop2d(const float* src, float* dst, int N, int H, int W, int y0, int x0, int h, int w, void* stream) {
    CF_REQUIRE(src && dst, "null pointer");
    CF_REQUIRE(N > 0 && h > 0 && w > 0 && y0 >= 0 && x0 >= 0 && y0 + h <= H && x0 + w <= W, "crop window out of range");
    long total = (long)N * h * w;
    LAUNCH_FLAT(crop2d_kernel, total, src, dst, H, W, y0, x0, h, w, total);
}

extern "C" int cf_pad2d(const float* src, float* dst, int N, int h, int w, int y0, int x0, int H, int W, void* stream) {
    CF_REQUIRE(src && dst, "null pointer");
    CF_REQUIRE(N > 0 && h > 0 && w > 0 && y0 >= 0 && x0 >= 0 && y0 + h <= H && x0 + w <= W, "pad window out of range");
    long total = (long)N * H * W;
    LAUNCH_FLAT(pad2d_kernel, total, src, dst, h, w, y0, x0, H, W, total);
}

extern "C" int cf_tta_accumulate(const float* logits, float* acc, int B, int K, int H, int W, int flip_h, int flip_w, float weight,
                                 void* stream) {
    CF_REQUIRE(logits && acc, "null pointer");
    CF_REQUIRE(B > 0 && K > 0 && K <= 64 && H > 0 && W > 0, "bad shape");
    long total = (long)B * H * W;
    LAUNCH_FLAT(tta_accumulate_kernel, total, logits, acc, K, H, W, flip_h, flip_w, weight, total);
}

extern "C" int cf_flip2d(const float* src, float* dst, int N, int H, int W, int flip_h, int flip_w, void* stream) {
    CF_REQUIRE(src && dst && src != dst, "null or aliased pointer");
    CF_REQUIRE(N > 0 && H > 0 && W > 0, "bad shape");
    long total = (long)N * H * W;
    LAUNCH_FLAT(flip2d_kernel, total, src, dst, H, W, flip_h, flip_w, total);
}

extern "C" int cf_tile_accumulate(const float* pred, const float* gauss, float* agg, float* cnt, int K, int X, int Y, int lx, int ly,
                                  int ph, int pw, void* stream) {
    CF_REQUIRE(pred && agg && cnt, "null pointer");
    CF_REQUIRE(K > 0 && lx >= 0 && ly >= 0 && lx + ph <= X && ly + pw <= Y && ph > 0 && pw > 0, "tile out of range");
    long total = (long)K * ph * pw;
    LAUNCH_FLAT(tile_accumulate_kernel, total, pred, gauss, agg, cnt, X, Y, lx, ly, ph, pw, total);
}

extern "C" int cf_tile_finalize(const float* agg, const float* cnt, float* probs, uint8_t* seg, int K, int X, int Y, void* stream) {
    CF_REQUIRE(agg && cnt && probs && seg, "null pointer");
    CF_REQUIRE(K > 0 && K <= 255 && X > 0 && Y > 0, "bad shape");
    long xy = (long)X * Y;
    LAUNCH_FLAT(tile_finalize_kernel, xy, agg, cnt, probs, seg, K, xy);
}

extern "C" int cf_argmax_channels(const float* x, uint8_t* out, int B, int K, int HW, void* stream) {
    CF_REQUIRE(x && out, "null pointer");
    CF_REQUIRE(B > 0 && K > 0 && K <= 255 && HW > 0, "bad shape");
    long total = (long)B * HW;
    LAUNCH_FLAT(argmax_channels_kernel, total, x, out, K, (long)HW, total);
}

extern "C" int cf_tta_accumulate_3d(const float* logits, float* acc, int B, int K, int D, int H, int W, int flip_d, int flip_h, int flip_w,
                                    float weight, void* stream) {
    CF_REQUIRE(logits && acc, "null pointer");
    CF_REQUIRE(B > 0 && K > 0 && D > 0 && H > 0 && W > 0, "bad shape");
    long total = (long)B * D * H * W;
    LAUNCH_FLAT(tta_accumulate_3d_kernel, total, logits, acc, K, D, H, W, flip_d, flip_h, flip_w, weight, total);
}

extern "C" int cf_flip3d(const float* src, float* dst, int N, int D, int H, int W, int flip_d, int flip_h, int flip_w, void* stream) {
    CF_REQUIRE(src && dst && src != dst, "null or aliased pointer");
    CF_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "bad shape");
    long total = (long)N * D * H * W;
    LAUNCH_FLAT(flip3d_kernel, total, src, dst, D, H, W, flip_d, flip_h, flip_w, total);
}

extern "C" int cf_tile_accumulate_3d(const float* pred, const float* gauss, float* agg, float* cnt, int K, int X, int Y, int Z, int lx,
                                     int ly, int lz, int px, int py, int pz, void* stream) {
    CF_REQUIRE(pred && agg && cnt, "null pointer");
    CF_REQUIRE(K > 0 && lx >= 0 && ly >= 0 && lz >= 0 && lx + px <= X && ly + py <= Y && lz + pz <= Z && px > 0 && py > 0 && pz > 0,
               "tile out of range");
    long total = (long)K * px * py * pz;
    LAUNCH_FLAT(tile_accumulate_3d_kernel, total, pred, gauss, agg, cnt, X, Y, Z, lx, ly, lz, px, py, pz, total);
}


// ---------------------------------------------------------------------------------------------------------------------
// Per-frame bounding boxes of the non-zero pixels (torchvision.ops.masks_to_boxes as Processor.get_mean_centroid uses it,
// processor.py:140-160) and bilinear sampling of a field at a list of points (SpatialTransformerContour, integration.py:5-34).
namespace cf {

template <typename T>
__global__ void __launch_bounds__(256) frame_boxes_kernel(const T* __restrict__ x, int* __restrict__ boxes, int H, int W) {
    __shared__ int red[4];   // x1, y1 (min), x2, y2 (max)
    if (threadIdx.x < 4) red[threadIdx.x] = threadIdx.x < 2 ? INT_MAX : -1;
    __syncthreads();
    const T* f = x + (long)blockIdx.x * H * W;
    int x1 = INT_MAX, y1 = INT_MAX, x2 = -1, y2 = -1;
    for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
        if (f[i] != (T)0) {
            const int yy = i / W, xx = i - yy * W;
            x1 = min(x1, xx); y1 = min(y1, yy); x2 = max(x2, xx); y2 = max(y2, yy);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        x1 = min(x1, __shfl_xor(x1, o, 64)); y1 = min(y1, __shfl_xor(y1, o, 64));
        x2 = max(x2, __shfl_xor(x2, o, 64)); y2 = max(y2, __shfl_xor(y2, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&red[0], x1); atomicMin(&red[1], y1); atomicMax(&red[2], x2); atomicMax(&red[3], y2); }
    __syncthreads();
    if (threadIdx.x < 4) boxes[blockIdx.x * 4 + threadIdx.x] = red[2] < 0 ? -1 : red[threadIdx.x];
}

// out[b, c, p] = bilinear sample of field[b, c] at the point whose grid_sample coordinates are (gx, gy) = the normalised
// (pts[b,0,p], pts[b,1,p]): 2 * (v / (size - 1) - 0.5) with size = W for channel 0 and H for channel 1 (the reference's shape[~i]),
// align_corners = True, zeros outside -- every intermediate rounded as PyTorch rounds it.
__global__ void __launch_bounds__(256) sample_points_kernel(const float* __restrict__ field, const float* __restrict__ pts, float* __restrict__ out,
                                                           int B, int C, int H, int W, int P) {
    const long total = (long)B * C * P;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int p = (int)(idx % P);
        const long r = idx / P;
        const int c = (int)(r % C), b = (int)(r / C);
        const float px = pts[((long)b * 2 + 0) * P + p], py = pts[((long)b * 2 + 1) * P + p];
        const float gx = __fmul_rn(2.0f, __fsub_rn(__fdiv_rn(px, (float)(W - 1)), 0.5f));
        const float gy = __fmul_rn(2.0f, __fsub_rn(__fdiv_rn(py, (float)(H - 1)), 0.5f));
        const float xs = __fmul_rn(__fdiv_rn(__fadd_rn(gx, 1.0f), 2.0f), (float)(W - 1));
        const float ys = __fmul_rn(__fdiv_rn(__fadd_rn(gy, 1.0f), 2.0f), (float)(H - 1));
        Taps t = make_taps(ys, xs, H, W);
        out[idx] = sample_taps(field + ((long)b * C + c) * H * W, t, W);
    }
}

}  // namespace cf

extern "C" int cf_frame_boxes(const void* x, int is_float, int* boxes, int N, int H, int W, void* stream) {
    CF_REQUIRE(x && boxes, "null pointer");
    CF_REQUIRE(N > 0 && H > 0 && W > 0 && (long)H * W < (1L << 31), "bad shape N=%d H=%d W=%d", N, H, W);
    if (is_float) hipLaunchKernelGGL(cf::frame_boxes_kernel<float>, dim3((unsigned)N), dim3(256), 0, cf::as_stream(stream), (const float*)x, boxes, H, W);
    else hipLaunchKernelGGL(cf::frame_boxes_kernel<uint8_t>, dim3((unsigned)N), dim3(256), 0, cf::as_stream(stream), (const uint8_t*)x, boxes, H, W);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_sample_points_2d(const float* field, const float* pts, float* out, int B, int C, int H, int W, int P, void* stream) {
    CF_REQUIRE(field && pts && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1 && P > 0, "bad shape");
    const long total = (long)B * C * P;
    hipLaunchKernelGGL(cf::sample_points_kernel, dim3(cf::flat_grid(total, 256)), dim3(256), 0, cf::as_stream(stream), field, pts, out, B, C, H, W, P);
    CF_CHECK_LAUNCH();
    return CF_OK;
}
