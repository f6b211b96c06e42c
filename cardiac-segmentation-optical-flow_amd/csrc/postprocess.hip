// Largest-connected-component filter of the export step (nnunet/postprocessing/connected_components.py:31-107) on the device:
// scipy.ndimage.label's default structure (face neighbours: 4 in 2-D, 6 in 3-D) by min-label propagation with pointer jumping,
// component sizes by one atomic histogram, removal in one pass.  Labels are "index of a voxel of the component + 1".
#include "common.h"

namespace cf {

#define GRID_STRIDE(i, n) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// labels[i] = i+1 where image[i] is one of the (up to 8) class values of the region, else 0
__global__ void __launch_bounds__(256) cc_init_kernel(const uint8_t* __restrict__ image, int* __restrict__ labels, long n,
                                                      unsigned long long classes, int nclasses) {
    GRID_STRIDE(i, n) {
        const unsigned v = image[i];
        bool in = false;
        for (int k = 0; k < nclasses; ++k) in |= (v == ((classes >> (8 * k)) & 0xffu));
        labels[i] = in ? (int)(i + 1) : 0;
    }
}

// one sweep: take the smallest label among the face neighbours, then jump to that label's own label (path halving).
// Labels only ever decrease, so racing reads of neighbours are harmless; *changed is raised when anything moved.
__global__ void __launch_bounds__(256) cc_sweep_kernel(int* __restrict__ labels, int D, int H, int W, int* __restrict__ changed) {
    const long HW = (long)H * W, n = (long)D * HW;
    GRID_STRIDE(i, n) {
        int l = labels[i];
        if (l == 0) continue;
        const int z = (int)(i / HW), q = (int)(i - (long)z * HW), y = q / W, x = q - y * W;
        int m = l;
        int t;
        if (x > 0 && (t = labels[i - 1]) != 0) m = min(m, t);
        if (x < W - 1 && (t = labels[i + 1]) != 0) m = min(m, t);
        if (y > 0 && (t = labels[i - W]) != 0) m = min(m, t);
        if (y < H - 1 && (t = labels[i + W]) != 0) m = min(m, t);
        if (z > 0 && (t = labels[i - HW]) != 0) m = min(m, t);
        if (z < D - 1 && (t = labels[i + HW]) != 0) m = min(m, t);
        t = labels[m - 1];          // voxel m-1 belongs to the same component: its label is a smaller-or-equal representative
        if (t != 0 && t < m) m = t;
        if (m < l) {
            labels[i] = m;
            *changed = 1;
        }
    }
}

__global__ void __launch_bounds__(256) cc_count_kernel(const int* __restrict__ labels, int* __restrict__ counts, long n) {
    GRID_STRIDE(i, n) {
        const int l = labels[i];
        if (l) atomicAdd(&counts[l - 1], 1);
    }
}

// connected_components.py:90-101: every object whose size differs from the largest one is removed when it is smaller than
// min_valid (min_valid < 0: always)
__global__ void __launch_bounds__(256) cc_remove_kernel(uint8_t* __restrict__ image, const int* __restrict__ labels,
                                                        const int* __restrict__ counts, long n, int max_count, double volume_per_voxel,
                                                        double min_valid) {
    GRID_STRIDE(i, n) {
        const int l = labels[i];
        if (!l) continue;
        const int c = counts[l - 1];
        if (c != max_count && (min_valid < 0.0 || (double)c * volume_per_voxel < min_valid)) image[i] = 0;
    }
}

// Export resampling (resample_data_or_seg, nnunet/preprocessing/preprocessing.py:111-200): skimage.transform.resize(mode='edge',
// anti_aliasing=False) / map_coordinates(mode='nearest') sample at src = scale * (dst + 0.5) - 0.5 with edge clamping; per axis
// either linear (order 1) or nearest (order 0, floor(src + 0.5)).  Coordinates in double like the reference.
struct ResizeAxis {
    int i0, i1;
    float w1;
};
__device__ __forceinline__ ResizeAxis resize_axis(int o, int n_src, int n_dst, int linear) {
    const double s = ((double)n_src / (double)n_dst) * ((double)o + 0.5) - 0.5;
    ResizeAxis a;
    if (linear) {
        const double f = floor(s);
        int i = (int)f;
        a.w1 = (float)(s - f);
        a.i0 = min(max(i, 0), n_src - 1);
        a.i1 = min(max(i + 1, 0), n_src - 1);
    } else {
        a.i0 = a.i1 = min(max((int)floor(s + 0.5), 0), n_src - 1);
        a.w1 = 0.f;
    }
    return a;
}

__global__ void __launch_bounds__(256) resize3d_kernel(const float* __restrict__ src, float* __restrict__ dst, int X, int Y, int Z, int X2,
                                                       int Y2, int Z2, int lx, int ly, int lz, long total) {
    const long V2 = (long)X2 * Y2 * Z2, V = (long)X * Y * Z;
    GRID_STRIDE(i, total) {
        const long n = i / V2, p = i - n * V2;
        const int z = (int)(p % Z2);
        const long r = p / Z2;
        const int y = (int)(r % Y2), x = (int)(r / Y2);
        const ResizeAxis ax = resize_axis(x, X, X2, lx), ay = resize_axis(y, Y, Y2, ly), az = resize_axis(z, Z, Z2, lz);
        const float* s = src + n * V;
        auto at = [&](int a, int b, int c) { return s[((long)a * Y + b) * Z + c]; };
        auto lerp = [](float u, float v, float w) { return u + (v - u) * w; };
        const float c00 = lerp(at(ax.i0, ay.i0, az.i0), at(ax.i0, ay.i0, az.i1), az.w1);
        const float c01 = lerp(at(ax.i0, ay.i1, az.i0), at(ax.i0, ay.i1, az.i1), az.w1);
        const float c10 = lerp(at(ax.i1, ay.i0, az.i0), at(ax.i1, ay.i0, az.i1), az.w1);
        const float c11 = lerp(at(ax.i1, ay.i1, az.i0), at(ax.i1, ay.i1, az.i1), az.w1);
        dst[i] = lerp(lerp(c00, c01, ay.w1), lerp(c10, c11, ay.w1), ax.w1);
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

extern "C" int cf_cc_init(const uint8_t* image, int* labels, long n, const uint8_t* classes, int nclasses, void* stream) {
    CF_REQUIRE(image && labels && classes, "null pointer");
    CF_REQUIRE(n > 0 && n < (1L << 31) - 1 && nclasses >= 1 && nclasses <= 8, "bad size");
    unsigned long long packed = 0;
    for (int k = 0; k < nclasses; ++k) packed |= (unsigned long long)classes[k] << (8 * k);   // host array of class values
    LAUNCH_FLAT(cc_init_kernel, n, image, labels, n, packed, nclasses);
}

extern "C" int cf_cc_sweep(int* labels, int D, int H, int W, int* changed, void* stream) {
    CF_REQUIRE(labels && changed, "null pointer");
    CF_REQUIRE(D > 0 && H > 0 && W > 0 && (long)D * H * W < (1L << 31) - 1, "bad shape");
    const long n = (long)D * H * W;
    LAUNCH_FLAT(cc_sweep_kernel, n, labels, D, H, W, changed);
}

extern "C" int cf_cc_count(const int* labels, int* counts, long n, void* stream) {
    CF_REQUIRE(labels && counts && n > 0, "bad arguments");
    LAUNCH_FLAT(cc_count_kernel, n, labels, counts, n);
}

extern "C" int cf_cc_remove(uint8_t* image, const int* labels, const int* counts, long n, int max_count, double volume_per_voxel,
                            double min_valid, void* stream) {
    CF_REQUIRE(image && labels && counts && n > 0, "bad arguments");
    LAUNCH_FLAT(cc_remove_kernel, n, image, labels, counts, n, max_count, volume_per_voxel, min_valid);
}

extern "C" int cf_resize3d(const float* src, float* dst, int N, int X, int Y, int Z, int X2, int Y2, int Z2, int linear_x, int linear_y,
                           int linear_z, void* stream) {
    CF_REQUIRE(src && dst && src != dst, "null or aliased pointer");
    CF_REQUIRE(N > 0 && X > 0 && Y > 0 && Z > 0 && X2 > 0 && Y2 > 0 && Z2 > 0, "bad shape");
    const long total = (long)N * X2 * Y2 * Z2;
    LAUNCH_FLAT(resize3d_kernel, total, src, dst, X, Y, Z, X2, Y2, Z2, linear_x, linear_y, linear_z, total);
}
