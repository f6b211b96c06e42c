// Downstream metrics of the output layout, on the device:
//   * nnunet/evaluation/metrics.py:27-105 ConfusionMatrix (TP / FP / TN / FN of `test != 0` vs `reference != 0`) and a K x K label
//     confusion histogram (one pass instead of one ConfusionMatrix per class, nnunet/compute_metrics.py:96-106);
//   * medpy.metric.binary's surface distances behind hausdorff_distance / avg_surface_distance[_symmetric]
//     (metrics.py:323-392): border = mask ^ binary_erosion(mask, face structure), distance of every border voxel of A to the
//     nearest border voxel of B -- scipy's exact EDT evaluated only where it is read, as a brute-force minimum over B's border
//     (a few thousand points per object; both sets staged through LDS), spacing-weighted in fp64, summed z, y, x like scipy;
//   * nnunet/compute_jacobian.py:146-186: kornia.filters.spatial_gradient3d(mode='diff', order=1) (replicate-padded central
//     differences times 0.5) and the per-structure statistics of the Jacobian determinant (sum, count, negatives per label).
#include "common.h"

namespace cf {

#define GRID_STRIDE(i, n) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

typedef unsigned long long u64;

__device__ __forceinline__ u64 wave_sum(u64 v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// out[0..2] += tp, fp, fn (tn = n - tp - fp - fn on the host)
__global__ void __launch_bounds__(256) confusion_kernel(const uint8_t* __restrict__ test, const uint8_t* __restrict__ ref, long n,
                                                        u64* __restrict__ out) {
    u64 tp = 0, fp = 0, fn = 0;
    GRID_STRIDE(i, n) {
        const bool t = test[i] != 0, r = ref[i] != 0;
        tp += t && r;
        fp += t && !r;
        fn += !t && r;
    }
    tp = wave_sum(tp); fp = wave_sum(fp); fn = wave_sum(fn);
    __shared__ u64 red[3][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = tp; red[1][threadIdx.x >> 6] = fp; red[2][threadIdx.x >> 6] = fn; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const u64 v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (v) atomicAdd(&out[threadIdx.x], v);
    }
}

// hist[t * K + r] += 1 for every voxel (labels >= K are counted nowhere and reported through hist[K * K])
__global__ void __launch_bounds__(256) label_confusion_kernel(const uint8_t* __restrict__ test, const uint8_t* __restrict__ ref, long n, int K,
                                                              u64* __restrict__ hist) {
    __shared__ unsigned int h[16 * 16 + 1];
    for (int k = threadIdx.x; k <= K * K; k += blockDim.x) h[k] = 0;
    __syncthreads();
    GRID_STRIDE(i, n) {
        const int t = test[i], r = ref[i];
        atomicAdd(&h[(t < K && r < K) ? t * K + r : K * K], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= K * K; k += blockDim.x)
        if (h[k]) atomicAdd(&hist[k], (u64)h[k]);
}

// border voxels of mask (set, with a face neighbour that is unset or outside the array: binary_erosion's border_value = 0),
// compacted as (z, y, x) triples; *count must be zero on entry.  ndim 2 (D == 1): no z neighbours.
__global__ void __launch_bounds__(256) surface_border_kernel(const uint8_t* __restrict__ mask, int D, int H, int W, int ndim, int* __restrict__ coords,
                                                             int* __restrict__ count) {
    const long HW = (long)H * W, n = (long)D * HW;
    const long nround = (n + 63) / 64 * 64;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nround; i += (long)gridDim.x * blockDim.x) {
        bool border = false;
        int z = 0, y = 0, x = 0;
        if (i < n && mask[i]) {
            z = (int)(i / HW);
            const int q = (int)(i - (long)z * HW);
            y = q / W;
            x = q - y * W;
            border = x == 0 || x == W - 1 || y == 0 || y == H - 1 || !mask[i - 1] || !mask[i + 1] || !mask[i - W] || !mask[i + W];
            if (ndim == 3) border = border || z == 0 || z == D - 1 || !mask[i - HW] || !mask[i + HW];
        }
        const u64 ballot = __ballot(border);
        if (ballot) {
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == 0) base = atomicAdd(count, __popcll(ballot));
            base = __shfl(base, 0, 64);
            if (border) {
                const int slot = base + __popcll(ballot & ((1ull << lane) - 1ull));
                coords[3 * slot] = z;
                coords[3 * slot + 1] = y;
                coords[3 * slot + 2] = x;
            }
        }
    }
}

// out[i] = min_j || spacing * (A[i] - B[j]) ||  (fp64; squares summed z, y, x like scipy's distance_transform_edt)
__global__ void __launch_bounds__(256) surface_min_dist_kernel(const int* __restrict__ A, int nA, const int* __restrict__ B, int nB, double sz,
                                                               double sy, double sx, double* __restrict__ out) {
    __shared__ int tile[256 * 3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int az = 0, ay = 0, ax = 0;
    if (i < nA) { az = A[3 * i]; ay = A[3 * i + 1]; ax = A[3 * i + 2]; }
    double best = INFINITY;
    for (int j0 = 0; j0 < nB; j0 += 256) {
        const int cnt = min(256, nB - j0);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt * 3; k += blockDim.x) tile[k] = B[3 * j0 + k];
        __syncthreads();
        if (i < nA) {
            for (int j = 0; j < cnt; ++j) {
                const double dz = sz * (double)(az - tile[3 * j]), dy = sy * (double)(ay - tile[3 * j + 1]), dx = sx * (double)(ax - tile[3 * j + 2]);
                best = fmin(best, __dadd_rn(__dadd_rn(__dmul_rn(dz, dz), __dmul_rn(dy, dy)), __dmul_rn(dx, dx)));
            }
        }
    }
    if (i < nA) out[i] = sqrt(best);
}

// out[0] = max(out[0], max x), out[1] += sum x for non-negative fp64 x (the bit pattern of a non-negative double is monotone)
__global__ void __launch_bounds__(256) max_sum_kernel(const double* __restrict__ x, long n, double* __restrict__ out) {
    double mx = 0.0, sm = 0.0;
    GRID_STRIDE(i, n) { mx = fmax(mx, x[i]); sm += x[i]; }
    for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o, 64)); sm += __shfl_xor(sm, o, 64); }
    __shared__ double rm[4], rs[4];
    if ((threadIdx.x & 63) == 0) { rm[threadIdx.x >> 6] = mx; rs[threadIdx.x >> 6] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(rm[0], rm[1]), fmax(rm[2], rm[3]));
        sm = rs[0] + rs[1] + rs[2] + rs[3];
        atomicMax(reinterpret_cast<u64*>(&out[0]), (u64)__double_as_longlong(mx));
        atomicAdd(&out[1], sm);
    }
}

// per label k < K: stats[3k] += sum x, stats[3k+1] += count, stats[3k+2] += #(x < 0)   (compute_jacobian.py:160-170)
__global__ void __launch_bounds__(256) region_stats_kernel(const double* __restrict__ x, const uint8_t* __restrict__ lab, long n, int K,
                                                           double* __restrict__ stats) {
    __shared__ double s[16 * 3];
    for (int k = threadIdx.x; k < K * 3; k += blockDim.x) s[k] = 0.0;
    __syncthreads();
    GRID_STRIDE(i, n) {
        const int k = lab[i];
        if (k < K) {
            const double v = x[i];
            atomicAdd(&s[3 * k], v);
            atomicAdd(&s[3 * k + 1], 1.0);
            if (v < 0.0) atomicAdd(&s[3 * k + 2], 1.0);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K * 3; k += blockDim.x)
        if (s[k] != 0.0) atomicAdd(&stats[k], s[k]);
}

// kornia.filters.spatial_gradient3d(x, mode='diff', order=1): x [N][D][H][W] -> out [N][3][D][H][W], component 0 along W, 1 along H,
// 2 along D: 0.5 * (x[+1] - x[-1]) with replicate padding
__global__ void __launch_bounds__(256) spatial_gradient3d_kernel(const float* __restrict__ x, float* __restrict__ out, long N, int D, int H, int W) {
    const long HW = (long)H * W, V = (long)D * HW, total = N * V;
    GRID_STRIDE(i, total) {
        const long nidx = i / V, p = i - nidx * V;
        const int z = (int)(p / HW), q = (int)(p - (long)z * HW), y = q / W, xx = q - y * W;
        const float* s = x + nidx * V;
        auto at = [&](int a, int b, int c) { return s[(long)min(max(a, 0), D - 1) * HW + (long)min(max(b, 0), H - 1) * W + min(max(c, 0), W - 1)]; };
        float* o = out + nidx * 3 * V + p;
        o[0] = 0.5f * (at(z, y, xx + 1) - at(z, y, xx - 1));
        o[V] = 0.5f * (at(z, y + 1, xx) - at(z, y - 1, xx));
        o[2 * V] = 0.5f * (at(z + 1, y, xx) - at(z - 1, y, xx));
    }
}

// sums[c * S + s] = sum |x[c][a][s][b]| over a, b (fp64): the per-frame means of |gradient| in compute_jacobian.py:147-159
__global__ void __launch_bounds__(256) slab_abs_sum_kernel(const float* __restrict__ x, int A, int S, long B, double* __restrict__ sums) {
    const int slab = blockIdx.x, c = slab / S, s = slab - c * S;
    const long cnt = (long)A * B;
    double acc = 0.0;
    for (long e = blockIdx.y * (long)blockDim.x + threadIdx.x; e < cnt; e += (long)gridDim.y * blockDim.x) {
        const long a = e / B, b = e - a * B;
        acc += fabs((double)x[(((long)c * A + a) * S + s) * B + b]);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double r[4];
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[slab], r[0] + r[1] + r[2] + r[3]);
}

// skimage.metrics.structural_similarity (uniform window, use_sample_covariance) of nnunet/compute_SSIM*.py: the five local means
// (x, y, xx, yy, xy over a win x win window, scipy 'reflect' boundary) and the SSIM value of every pixel, in fp64.
__global__ void __launch_bounds__(256) ssim_map_kernel(const double* __restrict__ a, const double* __restrict__ b, int H, int W, int win, double C1,
                                                       double C2, double cov_norm, double* __restrict__ S) {
    const long n = (long)H * W;
    const int r = win / 2;
    const double inv = 1.0 / ((double)win * win);
    GRID_STRIDE(i, n) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        double sx = 0.0, sy = 0.0, sxx = 0.0, syy = 0.0, sxy = 0.0;
        for (int dy = -r; dy <= r; ++dy) {
            int yy = y + dy;
            yy = yy < 0 ? -yy - 1 : (yy >= H ? 2 * H - 1 - yy : yy);       // reflect: (d c b a | a b c d | d c b a)
            yy = min(max(yy, 0), H - 1);
            for (int dx = -r; dx <= r; ++dx) {
                int xx = x + dx;
                xx = xx < 0 ? -xx - 1 : (xx >= W ? 2 * W - 1 - xx : xx);
                xx = min(max(xx, 0), W - 1);
                const double u = a[(long)yy * W + xx], v = b[(long)yy * W + xx];
                sx += u; sy += v; sxx += u * u; syy += v * v; sxy += u * v;
            }
        }
        const double ux = sx * inv, uy = sy * inv;
        const double vx = cov_norm * (sxx * inv - ux * ux), vy = cov_norm * (syy * inv - uy * uy), vxy = cov_norm * (sxy * inv - ux * uy);
        S[i] = ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
}

}  // namespace cf

using namespace cf;

#define CHECK_HIP(expr, what)                                              \
    do {                                                                   \
        if ((expr) != hipSuccess) { set_error(what); return CF_ERR_LAUNCH; } \
    } while (0)

extern "C" int cf_confusion_counts(const uint8_t* test, const uint8_t* reference, long n, unsigned long long* counts3, void* stream) {
    CF_REQUIRE(test && reference && counts3 && n > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    CHECK_HIP(hipMemsetAsync(counts3, 0, 3 * sizeof(u64), s), "cf_confusion_counts: memset failed");
    hipLaunchKernelGGL(confusion_kernel, dim3(flat_grid(n, 256, 16)), dim3(256), 0, s, test, reference, n, counts3);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_label_confusion(const uint8_t* test, const uint8_t* reference, long n, int K, unsigned long long* hist, void* stream) {
    CF_REQUIRE(test && reference && hist && n > 0, "bad arguments");
    CF_REQUIRE(K >= 1 && K <= 16, "K must be 1..16, got %d", K);
    hipStream_t s = as_stream(stream);
    CHECK_HIP(hipMemsetAsync(hist, 0, (K * K + 1) * sizeof(u64), s), "cf_label_confusion: memset failed");
    hipLaunchKernelGGL(label_confusion_kernel, dim3(flat_grid(n, 256, 16)), dim3(256), 0, s, test, reference, n, K, hist);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_surface_border(const uint8_t* mask, int D, int H, int W, int ndim, int* coords, int* count, void* stream) {
    CF_REQUIRE(mask && coords && count, "null pointer");
    CF_REQUIRE(D > 0 && H > 0 && W > 0 && (long)D * H * W < (1L << 29), "bad shape");
    CF_REQUIRE(ndim == 3 || (ndim == 2 && D == 1), "ndim must be 3, or 2 with D == 1");
    hipStream_t s = as_stream(stream);
    CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int), s), "cf_surface_border: memset failed");
    const long n = (long)D * H * W;
    hipLaunchKernelGGL(surface_border_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, s, mask, D, H, W, ndim, coords, count);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_surface_min_dist(const int* a, int na, const int* b, int nb, double sz, double sy, double sx, double* dist, void* stream) {
    CF_REQUIRE(a && b && dist, "null pointer");
    CF_REQUIRE(na > 0 && nb > 0, "empty point set (na=%d nb=%d)", na, nb);
    hipLaunchKernelGGL(surface_min_dist_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, as_stream(stream), a, na, b, nb, sz, sy, sx, dist);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_max_sum_nonneg(const double* x, long n, double* out2, void* stream) {
    CF_REQUIRE(x && out2 && n > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    CHECK_HIP(hipMemsetAsync(out2, 0, 2 * sizeof(double), s), "cf_max_sum_nonneg: memset failed");
    hipLaunchKernelGGL(max_sum_kernel, dim3(flat_grid(n, 256, 16)), dim3(256), 0, s, x, n, out2);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_region_stats(const double* x, const uint8_t* labels, long n, int K, double* stats, void* stream) {
    CF_REQUIRE(x && labels && stats && n > 0, "bad arguments");
    CF_REQUIRE(K >= 1 && K <= 16, "K must be 1..16, got %d", K);
    hipStream_t s = as_stream(stream);
    CHECK_HIP(hipMemsetAsync(stats, 0, 3 * K * sizeof(double), s), "cf_region_stats: memset failed");
    hipLaunchKernelGGL(region_stats_kernel, dim3(flat_grid(n, 256, 16)), dim3(256), 0, s, x, labels, n, K, stats);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_spatial_gradient3d(const float* x, float* out, long N, int D, int H, int W, void* stream) {
    CF_REQUIRE(x && out && x != out, "null or aliased pointer");
    CF_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "bad shape");
    const long total = N * D * H * W;
    hipLaunchKernelGGL(spatial_gradient3d_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, as_stream(stream), x, out, N, D, H, W);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_slab_abs_sum(const float* x, int C, int A, int S, long B, double* sums, void* stream) {
    CF_REQUIRE(x && sums, "null pointer");
    CF_REQUIRE(C > 0 && A > 0 && S > 0 && B > 0 && (long)C * S < (1L << 31), "bad shape");
    hipStream_t s = as_stream(stream);
    CHECK_HIP(hipMemsetAsync(sums, 0, (size_t)C * S * sizeof(double), s), "cf_slab_abs_sum: memset failed");
    long k = ((long)A * B + 4095) / 4096;
    if (k < 1) k = 1;
    if (k > 64) k = 64;
    hipLaunchKernelGGL(slab_abs_sum_kernel, dim3((unsigned)(C * S), (unsigned)k), dim3(256), 0, s, x, A, S, B, sums);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_ssim_map(const double* im1, const double* im2, int H, int W, int win, double C1, double C2, double cov_norm, double* S, void* stream) {
    CF_REQUIRE(im1 && im2 && S, "null pointer");
    CF_REQUIRE(H > 0 && W > 0 && win >= 3 && (win & 1) && win <= H && win <= W, "bad shape H=%d W=%d win=%d (odd window no larger than the image)", H, W, win);
    const long n = (long)H * W;
    hipLaunchKernelGGL(ssim_map_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, as_stream(stream), im1, im2, H, W, win, C1, C2, cov_norm, S);
    CF_CHECK_LAUNCH();
    return CF_OK;
}
