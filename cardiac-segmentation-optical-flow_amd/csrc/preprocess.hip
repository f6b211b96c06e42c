// Test-time preprocessing, the step right before the hot path, on the device:
//   * crop-to-nonzero (nnunet/preprocessing/cropping.py:25-61, :104-137): non-zero mask over the modalities, scipy's
//     binary_fill_holes (background reachable from the array border through face neighbours stays background -- the background
//     components come from the connected-component sweeps of postprocess.hip), bounding box;
//   * cubic-spline resampling (preprocessing.py:111-200 through skimage.transform.resize(order=3, mode='edge') =
//     scipy.ndimage.zoom(order=3, mode='nearest', grid_mode=True)): one separable pass per axis -- edge-pad 12 samples, the recursive
//     B-spline prefilter of scipy's ni_splines.c (pole sqrt(3)-2, mirror initialisation) unrolled into its impulse response, 4-tap evaluation at
//     src = (dst + 0.5) n/m - 0.5.  All in fp64 like the reference, which resamples `data.astype(float)`;
//   * per-slab clipping to the input range (resize's clip=True) and intensity normalisation (preprocessing.py:274-320).
// Volumes are a few MB: launch-latency scale work; every kernel is one thread per voxel (no serial per-line recursion) and reductions
// combine per block before they touch an atomic.
#include <limits.h>

#include "common.h"

namespace cf {

#define GRID_STRIDE(i, n) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

__global__ void __launch_bounds__(256) nonzero_mask_kernel(const float* __restrict__ data, int C, long V, uint8_t* __restrict__ mask) {
    GRID_STRIDE(i, V) {
        bool nz = false;
        for (int c = 0; c < C; ++c) nz |= data[(long)c * V + i] != 0.f;   // NaN != 0 is true, as in numpy
        mask[i] = nz ? 1 : 0;
    }
}

// background components (labels of postprocess.hip's sweeps over mask == 0) that own a voxel on the array border
__global__ void __launch_bounds__(256) border_touch_kernel(const int* __restrict__ labels, int D, int H, int W, int z_faces,
                                                           int* __restrict__ touch) {
    const long HW = (long)H * W, n = (long)D * HW;
    GRID_STRIDE(i, n) {
        const int l = labels[i];
        if (!l) continue;
        const int z = (int)(i / HW), q = (int)(i - (long)z * HW), y = q / W, x = q - y * W;
        if (x == 0 || x == W - 1 || y == 0 || y == H - 1 || (z_faces && (z == 0 || z == D - 1))) touch[l - 1] = 1;
    }
}

__global__ void __launch_bounds__(256) fill_holes_kernel(uint8_t* __restrict__ mask, const int* __restrict__ labels,
                                                         const int* __restrict__ touch, long n) {
    GRID_STRIDE(i, n) {
        const int l = labels[i];
        if (l && !touch[l - 1]) mask[i] = 1;
    }
}

// bbox = {min z, max z, min y, max y, min x, max x} over mask != 0 (host initialises mins to INT_MAX and maxes to -1)
__global__ void __launch_bounds__(256) mask_bbox_kernel(const uint8_t* __restrict__ mask, int D, int H, int W, int* __restrict__ bbox) {
    const long HW = (long)H * W, n = (long)D * HW;
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {-1, -1, -1};
    GRID_STRIDE(i, n) {
        if (!mask[i]) continue;
        const int z = (int)(i / HW), q = (int)(i - (long)z * HW), y = q / W, x = q - y * W;
        lo[0] = min(lo[0], z); hi[0] = max(hi[0], z);
        lo[1] = min(lo[1], y); hi[1] = max(hi[1], y);
        lo[2] = min(lo[2], x); hi[2] = max(hi[2], x);
    }
    // wave reduce, then one LDS combine per block: a few hundred same-address atomics in total instead of one set per wave
    __shared__ int s_lo[3][4], s_hi[3][4];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = max(hi[a], __shfl_xor(hi[a], o, 64));
        }
        if ((threadIdx.x & 63) == 0) { s_lo[a][wave] = lo[a]; s_hi[a][wave] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const int l = min(min(s_lo[a][0], s_lo[a][1]), min(s_lo[a][2], s_lo[a][3]));
        const int h = max(max(s_hi[a][0], s_hi[a][1]), max(s_hi[a][2], s_hi[a][3]));
        if (h >= 0) {
            atomicMin(&bbox[2 * a], l);
            atomicMax(&bbox[2 * a + 1], h);
        }
    }
}

// One cubic-spline resampling pass along the middle axis of src[outer][n][inner] -> dst[outer][m][inner], one thread per OUTPUT
// sample.  scipy's recursive prefilter (causal + anticausal first-order sections, pole z = sqrt(3) - 2, mirror initialisation, on
// the line edge-padded by 12 samples) is the convolution of the mirror-extended padded line with h[d] = -6z/(1 - z^2) z^|d|;
// |z|^24 = 2e-14, so 24 taps either side reproduce the fp64 recursion to 1e-14 relative (checked against scipy.ndimage.zoom in
// tests/test_preprocess.py) with no serial dependence: the 4 B-spline taps and the prefilter fold into one 52-tap filter per
// output.  The serial one-thread-per-line recursion this replaces took 1.2 ms for a 10 x 236 x 200 volume (latency of ~500
// dependent global accesses); this form is bandwidth-shaped.
constexpr int SPLINE_PAD = 12;      // scipy.ndimage._prepad_for_spline_filter for mode 'nearest'
constexpr int SPLINE_J = 24;        // prefilter taps either side
__global__ void __launch_bounds__(256) spline3_axis_kernel(const double* __restrict__ src, double* __restrict__ dst, long outer, int n,
                                                           long inner, int m) {
    __shared__ double zp[SPLINE_J + 1];
    const double z = -0.26794919243112270647;          // sqrt(3) - 2
    if (threadIdx.x == 0) {
        double v = -6.0 * z / (1.0 - z * z);
        for (int k = 0; k <= SPLINE_J; ++k) { zp[k] = v; v *= z; }
    }
    __syncthreads();
    const long total = outer * m * inner;
    const int np = n + 2 * SPLINE_PAD;
    const double scale = (double)n / (double)m;
    GRID_STRIDE(i, total) {
        const long in = i % inner, r = i / inner;
        const int j = (int)(r % m);
        const long o = r / m;
        const double* s = src + o * n * inner + in;
        const double t = scale * ((double)j + 0.5) - 0.5 + (double)SPLINE_PAD;
        const double fl = floor(t), f = t - fl, g = 1.0 - f;
        const int i0 = (int)fl - 1;
        const double w[4] = {g * g * g / 6.0, (4.0 - 6.0 * f * f + 3.0 * f * f * f) / 6.0, (4.0 - 6.0 * g * g + 3.0 * g * g * g) / 6.0, f * f * f / 6.0};
        double acc = 0.0;
        for (int u = i0 - SPLINE_J; u <= i0 + 3 + SPLINE_J; ++u) {
            double coef = 0.0;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int d = abs(u - i0 - l);
                if (d <= SPLINE_J) coef += w[l] * zp[d];
            }
            int q = u < 0 ? -u : u;                     // mirror extension of the padded line ...
            if (q > np - 1) q = 2 * (np - 1) - q;
            q = min(max(q - SPLINE_PAD, 0), n - 1);     // ... whose own samples are the edge-clamped input
            acc += coef * s[q * inner];
        }
        dst[i] = acc;
    }
}

// min / max of every slab (c, s) of x[C][A][S][B]: block (slab, chunk) reduces its share into part[(slab * nchunk + chunk) * 2 + {0, 1}],
// slab_minmax_combine_kernel folds the chunks -> mm[slab * 2 + {0, 1}]
__global__ void __launch_bounds__(256) slab_minmax_kernel(const double* __restrict__ x, int A, int S, long B, double* __restrict__ part) {
    const int slab = blockIdx.x, c = slab / S, s = slab - c * S;
    const long cnt = (long)A * B;
    double lo = INFINITY, hi = -INFINITY;
    for (long e = blockIdx.y * (long)blockDim.x + threadIdx.x; e < cnt; e += (long)gridDim.y * blockDim.x) {
        const long a = e / B, b = e - a * B;
        const double v = x[(((long)c * A + a) * S + s) * B + b];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o, 64));
        hi = fmax(hi, __shfl_xor(hi, o, 64));
    }
    __shared__ double slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* q = part + 2 * ((long)slab * gridDim.y + blockIdx.y);
        q[0] = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3]));
        q[1] = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
    }
}
__global__ void __launch_bounds__(64) slab_minmax_combine_kernel(const double* __restrict__ part, int nchunk, double* __restrict__ mm) {
    const int slab = blockIdx.x;
    double lo = INFINITY, hi = -INFINITY;
    for (int k = threadIdx.x; k < nchunk; k += 64) {
        lo = fmin(lo, part[2 * ((long)slab * nchunk + k)]);
        hi = fmax(hi, part[2 * ((long)slab * nchunk + k) + 1]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o, 64));
        hi = fmax(hi, __shfl_xor(hi, o, 64));
    }
    if (threadIdx.x == 0) { mm[2 * slab] = lo; mm[2 * slab + 1] = hi; }
}

// y[C][A][S][B] (fp64) clipped to its slab's range and rounded to fp32 (the reference's `.astype(dtype_data)`)
__global__ void __launch_bounds__(256) slab_clip_kernel(const double* __restrict__ y, float* __restrict__ out, int A, int S, long B, long total,
                                                        const double* __restrict__ mm) {
    GRID_STRIDE(i, total) {
        const long r = i / B;
        const int s = (int)(r % S);
        const long c = r / S / A;
        double v = y[i];
        if (mm) {
            const double* q = mm + 2 * (c * S + s);
            v = fmin(fmax(v, q[0]), q[1]);
        }
        out[i] = (float)v;
    }
}

// sum, sum of squares and count of the voxels selected by (seg >= 0) and / or (lo < x < hi) -> out[3] (fp64 atomics)
__global__ void __launch_bounds__(256) masked_moments_kernel(const float* __restrict__ x, const float* __restrict__ seg, long n, int use_range,
                                                             float lo, float hi, double* __restrict__ out) {
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    GRID_STRIDE(i, n) {
        const float v = x[i];
        bool in = true;
        if (seg) in = seg[i] >= 0.f;
        if (use_range) in = in && v > lo && v < hi;
        if (in) { s1 += (double)v; s2 += (double)v * (double)v; cnt += 1.0; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    // one LDS combine per block, then one atomic triple per block (same-address fp64 atomics from every wave serialise)
    __shared__ double red[3][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; red[2][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (v != 0.0) atomicAdd(&out[threadIdx.x], v);
    }
}

// x = ((clip ? clamp(x, lo, hi) : x) - sub) / div in fp32 like the reference's float32 arrays; voxels with seg < 0 become 0 when
// zero_outside; with only_inside the voxels outside the mask are left for zero_outside to clear (preprocessing.py:311-315)
__global__ void __launch_bounds__(256) normalize_kernel(float* __restrict__ x, const float* __restrict__ seg, long n, int clip, float lo, float hi,
                                                        float sub, float div, int zero_outside) {
    GRID_STRIDE(i, n) {
        float v = x[i];
        if (clip) v = fminf(fmaxf(v, lo), hi);
        v = __fdiv_rn(__fsub_rn(v, sub), div);
        if (zero_outside && seg[i] < 0.f) v = 0.f;
        x[i] = v;
    }
}

__global__ void __launch_bounds__(256) nan_to_zero_kernel(float* __restrict__ x, long n) {
    GRID_STRIDE(i, n) {
        const float v = x[i];
        if (v != v) x[i] = 0.f;
    }
}

// dst[i] = value where src[i] >= thr (resize_segmentation: a label wins where its resized indicator reaches 0.5)
__global__ void __launch_bounds__(256) assign_where_ge_kernel(float* __restrict__ dst, const float* __restrict__ src, long n, float thr, float value) {
    GRID_STRIDE(i, n) {
        if (src[i] >= thr) dst[i] = value;
    }
}

// cropping.py:128-135: seg[(seg == 0) & (mask == 0)] = label
__global__ void __launch_bounds__(256) seg_outside_mask_kernel(float* __restrict__ seg, const uint8_t* __restrict__ mask, int C, long V, float label) {
    GRID_STRIDE(i, (long)C * V) {
        if (seg[i] == 0.f && !mask[i % V]) seg[i] = label;
    }
}

}  // namespace cf

using namespace cf;

#define LAUNCH_FLAT(kernel, n, ...)                                                                              \
    do {                                                                                                         \
        hipLaunchKernelGGL(kernel, dim3(flat_grid((n), 256)), dim3(256), 0, as_stream(stream), __VA_ARGS__);      \
        CF_CHECK_LAUNCH();                                                                                       \
        return CF_OK;                                                                                            \
    } while (0)

extern "C" int cf_nonzero_mask(const float* data, int C, long V, uint8_t* mask, void* stream) {
    CF_REQUIRE(data && mask, "null pointer");
    CF_REQUIRE(C > 0 && V > 0, "bad shape C=%d V=%ld", C, V);
    LAUNCH_FLAT(nonzero_mask_kernel, V, data, C, V, mask);
}

extern "C" int cf_fill_holes(uint8_t* mask, const int* labels, int* touch, int D, int H, int W, int ndim, void* stream) {
    CF_REQUIRE(mask && labels && touch, "null pointer");
    CF_REQUIRE(D > 0 && H > 0 && W > 0 && (long)D * H * W < (1L << 31) - 1, "bad shape");
    CF_REQUIRE(ndim == 3 || (ndim == 2 && D == 1), "ndim must be 3, or 2 with D == 1");
    const long n = (long)D * H * W;
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(touch, 0, n * sizeof(int), s) != hipSuccess) { set_error("cf_fill_holes: memset failed"); return CF_ERR_LAUNCH; }
    hipLaunchKernelGGL(border_touch_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, s, labels, D, H, W, ndim == 3 ? 1 : 0, touch);
    hipLaunchKernelGGL(fill_holes_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, s, mask, labels, touch, n);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

// {min, max} pairs start at {INT_MAX, -1}; written by a kernel on the caller's stream (a host-to-device copy of a stack array needed a
// stream synchronisation here, against the library's own "no host sync" rule)
__global__ void bbox_init_kernel(int* bbox) {
    if (threadIdx.x < 6) bbox[threadIdx.x] = (threadIdx.x & 1) ? -1 : INT_MAX;
}

extern "C" int cf_mask_bbox(const uint8_t* mask, int D, int H, int W, int* bbox, void* stream) {
    CF_REQUIRE(mask && bbox, "null pointer");
    CF_REQUIRE(D > 0 && H > 0 && W > 0 && (long)D * H * W < (1L << 31) - 1, "bad shape");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, s, bbox);
    const long n = (long)D * H * W;
    hipLaunchKernelGGL(mask_bbox_kernel, dim3(flat_grid(n, 256, 16)), dim3(256), 0, s, mask, D, H, W, bbox);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_spline3_resample_axis(const double* src, double* dst, long outer, int n, long inner, int m, void* stream) {
    CF_REQUIRE(src && dst && src != dst, "null or aliased pointer");
    CF_REQUIRE(outer > 0 && n > 0 && inner > 0 && m > 0, "bad shape outer=%ld n=%d inner=%ld m=%d", outer, n, inner, m);
    const long total = outer * m * inner;
    LAUNCH_FLAT(spline3_axis_kernel, total, src, dst, outer, n, inner, m);
}

extern "C" int cf_slab_minmax_chunks(int C, int A, int S, long B) {
    // chunks per slab: enough blocks for the chip, at least ~2K elements per block
    (void)C;
    const long cnt = (long)A * B;
    long k = (cnt + 2047) / 2048;
    if (k < 1) k = 1;
    if (k > 64) k = 64;
    return (int)k;
}

extern "C" int cf_slab_minmax(const double* x, int C, int A, int S, long B, double* minmax, double* partial, void* stream) {
    CF_REQUIRE(x && minmax && partial, "null pointer");
    CF_REQUIRE(C > 0 && A > 0 && S > 0 && B > 0 && (long)C * S < (1L << 31), "bad shape");
    const int nchunk = cf_slab_minmax_chunks(C, A, S, B);
    hipLaunchKernelGGL(slab_minmax_kernel, dim3((unsigned)(C * S), (unsigned)nchunk), dim3(256), 0, as_stream(stream), x, A, S, B, partial);
    hipLaunchKernelGGL(slab_minmax_combine_kernel, dim3((unsigned)(C * S)), dim3(64), 0, as_stream(stream), partial, nchunk, minmax);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_slab_clip_to_f32(const double* y, float* out, int C, int A, int S, long B, const double* minmax, void* stream) {
    CF_REQUIRE(y && out, "null pointer");
    CF_REQUIRE(C > 0 && A > 0 && S > 0 && B > 0, "bad shape");
    const long total = (long)C * A * S * B;
    LAUNCH_FLAT(slab_clip_kernel, total, y, out, A, S, B, total, minmax);
}

extern "C" int cf_masked_moments(const float* x, const float* seg, long n, int use_range, float lo, float hi, double* out3, void* stream) {
    CF_REQUIRE(x && out3 && n > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(out3, 0, 3 * sizeof(double), s) != hipSuccess) { set_error("cf_masked_moments: memset failed"); return CF_ERR_LAUNCH; }
    hipLaunchKernelGGL(masked_moments_kernel, dim3(flat_grid(n, 256, 16)), dim3(256), 0, s, x, seg, n, use_range, lo, hi, out3);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_normalize(float* x, const float* seg, long n, int clip, float lo, float hi, float sub, float div, int zero_outside, void* stream) {
    CF_REQUIRE(x && n > 0, "bad arguments");
    CF_REQUIRE(!zero_outside || seg, "zero_outside needs the segmentation");
    CF_REQUIRE(div != 0.f, "zero divisor");
    LAUNCH_FLAT(normalize_kernel, n, x, seg, n, clip, lo, hi, sub, div, zero_outside);
}

extern "C" int cf_nan_to_zero(float* x, long n, void* stream) {
    CF_REQUIRE(x && n > 0, "bad arguments");
    LAUNCH_FLAT(nan_to_zero_kernel, n, x, n);
}

extern "C" int cf_assign_where_ge(float* dst, const float* src, long n, float thr, float value, void* stream) {
    CF_REQUIRE(dst && src && n > 0, "bad arguments");
    LAUNCH_FLAT(assign_where_ge_kernel, n, dst, src, n, thr, value);
}

extern "C" int cf_seg_outside_mask(float* seg, const uint8_t* mask, int C, long V, float label, void* stream) {
    CF_REQUIRE(seg && mask && C > 0 && V > 0, "bad arguments");
    LAUNCH_FLAT(seg_outside_mask_kernel, (long)C * V, seg, mask, C, V, label);
}
