// VoxelMorph spatial-transformer family: bilinear warp, scaling-and-squaring, fused label propagation,
// fused memory-encoder input, Jacobian determinant.  All HBM-bound: one thread per output pixel, flow read
// once (coalesced along W), the four taps served by L1/L2 (neighbouring lanes sample neighbouring pixels).
#include "common.h"
#include "profile.h"

// every product and sum below rounds on its own, as in ATen / numpy (hipcc would otherwise contract a * b + c into one FMA)
#pragma clang fp contract(off)

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


// ---------------------------------------------------------------------------------------------------------- W % 4 == 0 forms
// Four consecutive pixels of one row per thread: the flow is read as two 16-byte loads, the outputs leave as one 16-byte (float) /
// 4-byte (label) / 2 x 16-byte (double) store, and the sixteen tap gathers of a thread are independent loads in flight together.
// The taps go through a buffer resource over the whole source tensor: a tap outside the image gets an out-of-range offset and the
// hardware returns 0 for it -- no branch and no 64-bit address per tap (the one-pixel kernels spend more instructions on those than on
// the interpolation).  Index arithmetic is 32-bit; the entry points send tensors that do not fit (>= 2 GiB) to the one-pixel kernels.
// The arithmetic per pixel is the scalar kernels' (same helpers, same order): results are bit-identical.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr unsigned TAP_OOB = 0x80000000u;

template <bool ADD>
__global__ void __launch_bounds__(256) warp_bilinear_2d_v4_kernel(const float* __restrict__ flow, const float* __restrict__ src,
                                                                  float* __restrict__ out, int B, int C, int H, int W,
                                                                  const float* __restrict__ addend) {
    const unsigned HW = (unsigned)(H * W);
    const unsigned nq = (unsigned)B * HW / 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)((unsigned)B * C * HW * 4u), 0x00020000);
    for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
        const unsigned e = q * 4;
        const unsigned b = e / HW;
        const unsigned p = e - b * HW;
        const unsigned i = p / (unsigned)W, j = p - i * (unsigned)W;
        const float* fb = flow + (size_t)b * 2 * HW + p;
        const f32x4 fy = *reinterpret_cast<const f32x4*>(fb);
        const f32x4 fx = *reinterpret_cast<const f32x4*>(fb + HW);
        unsigned off[4][4];
        float w[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Taps t = make_taps(st_coord((float)i, fy[k], (float)(H - 1)), st_coord((float)(j + k), fx[k], (float)(W - 1)), H, W);
            const unsigned base = ((unsigned)((int)(b * C * HW) + t.y0 * W + t.x0)) * 4u;
            off[k][0] = t.v00 ? base : TAP_OOB;
            off[k][1] = t.v01 ? base + 4u : TAP_OOB;
            off[k][2] = t.v10 ? base + 4u * W : TAP_OOB;
            off[k][3] = t.v11 ? base + 4u * W + 4u : TAP_OOB;
            w[k][0] = t.w00; w[k][1] = t.w01; w[k][2] = t.w10; w[k][3] = t.w11;
        }
        const size_t ob = (size_t)b * C * HW + p;
        for (int c = 0; c < C; ++c) {
            const unsigned soff = (unsigned)c * HW * 4u;
            float g[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < 4; ++n) g[k][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off[k][n], soff, 0));
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k)   // sample_taps' association: nw*w + ne*w + sw*w + se*w
                v[k] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g[k][0], w[k][0]), __fmul_rn(g[k][1], w[k][1])), __fmul_rn(g[k][2], w[k][2])),
                                 __fmul_rn(g[k][3], w[k][3]));
            if (ADD) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(addend + ob + (size_t)c * HW);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = __fadd_rn(a[k], v[k]);
            }
            *reinterpret_cast<f32x4*>(out + ob + (size_t)c * HW) = v;
        }
    }
}

template <int KT>   // KT > 0: the class count as a compile-time constant (the class loop unrolls); 0: run-time K
__global__ void __launch_bounds__(256) warp_labels_2d_v4_kernel(const float* __restrict__ flow, const uint8_t* __restrict__ labels,
                                                                uint8_t* __restrict__ out, int T, int B, int Krt, int H, int W) {
    const int K = KT ? KT : Krt;
    const unsigned HW = (unsigned)(H * W);
    const unsigned nq = (unsigned)T * B * HW / 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(labels), 0, (int)((unsigned)B * HW), 0x00020000);
    for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
        const unsigned e = q * 4;
        const unsigned tb = e / HW;
        const unsigned p = e - tb * HW;
        const unsigned b = tb % (unsigned)B;
        const unsigned i = p / (unsigned)W, j = p - i * (unsigned)W;
        const float* fb = flow + (size_t)tb * 2 * HW + p;
        const f32x4 fy = *reinterpret_cast<const f32x4*>(fb);
        const f32x4 fx = *reinterpret_cast<const f32x4*>(fb + HW);
        int lab[4][4];
        float w[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Taps t = make_taps(st_coord((float)i, fy[k], (float)(H - 1)), st_coord((float)(j + k), fx[k], (float)(W - 1)), H, W);
            const unsigned base = (unsigned)((int)(b * HW) + t.y0 * W + t.x0);
            // a tap outside the image reads 0 through the out-of-range offset; its weight is zeroed instead of its label being set to
            // "no class" (the class sums are the same: + 0.f either way)
            lab[k][0] = __builtin_amdgcn_raw_buffer_load_b8(rs, t.v00 ? base : TAP_OOB, 0, 0);
            lab[k][1] = __builtin_amdgcn_raw_buffer_load_b8(rs, t.v01 ? base + 1u : TAP_OOB, 0, 0);
            lab[k][2] = __builtin_amdgcn_raw_buffer_load_b8(rs, t.v10 ? base + W : TAP_OOB, 0, 0);
            lab[k][3] = __builtin_amdgcn_raw_buffer_load_b8(rs, t.v11 ? base + W + 1u : TAP_OOB, 0, 0);
            w[k][0] = t.v00 ? t.w00 : 0.f; w[k][1] = t.v01 ? t.w01 : 0.f; w[k][2] = t.v10 ? t.w10 : 0.f; w[k][3] = t.v11 ? t.w11 : 0.f;
        }
        unsigned packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float best = -1.f;
            int arg = 0;
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const float v = __fadd_rn(__fadd_rn(__fadd_rn(lab[k][0] == c ? w[k][0] : 0.f, lab[k][1] == c ? w[k][1] : 0.f),
                                                    lab[k][2] == c ? w[k][2] : 0.f), lab[k][3] == c ? w[k][3] : 0.f);
                if (v > best) { best = v; arg = c; }
            }
            packed |= (unsigned)arg << (8 * k);
        }
        *reinterpret_cast<unsigned*>(out + e) = packed;
    }
}

// np.gradient spacings are 1 (one-sided, at the border) or 2 (central): the division becomes a multiplication by 1.0 or 0.5, which is
// exact and therefore the same double; an IEEE float64 division costs more than the rest of the pixel.
__global__ void __launch_bounds__(256) jacobian_det_2d_v4_kernel(const float* __restrict__ disp, double* __restrict__ det, int B, int H,
                                                                 int W) {
    const unsigned HW = (unsigned)(H * W);
    const unsigned nq = (unsigned)B * HW / 4;
    // XCD-banded block order (grid is a multiple of 8): neighbouring blocks, which read each other's border rows, run on ONE XCD and meet
    // in its L2 -- round-robin they fetched every border row once per XCD (PMC FETCH_SIZE 1.59x the algorithmic read before this)
    const unsigned bid = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    for (unsigned q = bid * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
        const unsigned e = q * 4;
        const unsigned b = e / HW;
        const unsigned p = e - b * HW;
        const int i = (int)(p / (unsigned)W), j = (int)(p - (unsigned)i * (unsigned)W);
        const float* d0 = disp + (size_t)b * 2 * HW;
        const float* d1 = d0 + HW;
        const int iu = i > 0 ? i - 1 : i, id = i < H - 1 ? i + 1 : i;
        const double rhi = id - iu == 2 ? 0.5 : 1.0;
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(d0 + iu * W + j), n0 = *reinterpret_cast<const f32x4*>(d0 + id * W + j);
        const f32x4 u1 = *reinterpret_cast<const f32x4*>(d1 + iu * W + j), n1 = *reinterpret_cast<const f32x4*>(d1 + id * W + j);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(d0 + i * W + j), c1 = *reinterpret_cast<const f32x4*>(d1 + i * W + j);
        // row i, columns j-1 .. j+4; at the row ends np.gradient differences the border element with its inner neighbour
        const bool first = j == 0, last = j + 4 == W;
        const float r0[6] = {first ? c0[0] : d0[i * W + j - 1], c0[0], c0[1], c0[2], c0[3], last ? c0[3] : d0[i * W + j + 4]};
        const float r1[6] = {first ? c1[0] : d1[i * W + j - 1], c1[0], c1[1], c1[2], c1[3], last ? c1[3] : d1[i * W + j + 4]};
        double o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jj = j + k;
            const bool one_l = k == 0 && first, one_r = k == 3 && last;   // one-sided difference
            const int jl = one_l ? jj : jj - 1, jr = one_r ? jj : jj + 1;
            const double rhj = (one_l || one_r) ? 1.0 : 0.5;
            const double p0_a0 = (((double)n0[k] + id) - ((double)u0[k] + iu)) * rhi;
            const double p1_a0 = (((double)n1[k] + jj) - ((double)u1[k] + jj)) * rhi;
            const double p0_a1 = (((double)r0[k + 2] + i) - ((double)r0[k] + i)) * rhj;
            const double p1_a1 = (((double)r1[k + 2] + jr) - ((double)r1[k] + jl)) * rhj;
            o[k] = p0_a0 * p1_a1 - p0_a1 * p1_a0;
        }
        *reinterpret_cast<f64x2*>(det + e) = f64x2{o[0], o[1]};
        *reinterpret_cast<f64x2*>(det + e + 2) = f64x2{o[2], o[3]};
    }
}

// 3-D branch of SpatialTransformer (integration.py:75-77): trilinear, align_corners=True, zeros outside.  Weights and the
// accumulation order follow ATen's scalar grid_sampler_3d ((x1 - ix) * (y1 - iy) * (z1 - iz), corners tnw, tne, tsw, tse,
// bnw, bne, bsw, bse).  flow channel i displaces along axis i of (D, H, W).
__global__ void __launch_bounds__(256) warp_trilinear_3d_kernel(const float* __restrict__ flow, const float* __restrict__ src,
                                                                float* __restrict__ out, int B, int C, int D, int H, int W) {
    const long HW = (long)H * W, V = (long)D * HW;
    const long total = (long)B * V;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / V);
        const long p = idx - (long)b * V;
        const int z = (int)(p / HW);
        const int q = (int)(p - (long)z * HW);
        const int y = q / W, x = q - y * W;
        const float* fb = flow + (long)b * 3 * V + p;
        const float iz = st_coord((float)z, fb[0], (float)(D - 1));
        const float iy = st_coord((float)y, fb[V], (float)(H - 1));
        const float ix = st_coord((float)x, fb[2 * V], (float)(W - 1));
        float zf = floorf(iz), yf = floorf(iy), xf = floorf(ix);
        // weights before the clamp (the clamp only guards the float->int conversion of far-away / NaN coordinates)
        const float wz1 = __fsub_rn(__fadd_rn(zf, 1.0f), iz), wz0 = __fsub_rn(iz, zf);
        const float wy1 = __fsub_rn(__fadd_rn(yf, 1.0f), iy), wy0 = __fsub_rn(iy, yf);
        const float wx1 = __fsub_rn(__fadd_rn(xf, 1.0f), ix), wx0 = __fsub_rn(ix, xf);
        zf = fminf(fmaxf(zf, -2.0f), (float)D + 1.0f);
        yf = fminf(fmaxf(yf, -2.0f), (float)H + 1.0f);
        xf = fminf(fmaxf(xf, -2.0f), (float)W + 1.0f);
        const int z0 = (int)zf, y0 = (int)yf, x0 = (int)xf;
        const bool zv0 = z0 >= 0 && z0 < D, zv1 = z0 + 1 >= 0 && z0 + 1 < D;
        const bool yv0 = y0 >= 0 && y0 < H, yv1 = y0 + 1 >= 0 && y0 + 1 < H;
        const bool xv0 = x0 >= 0 && x0 < W, xv1 = x0 + 1 >= 0 && x0 + 1 < W;
        float w[8];
        bool v[8];
        long off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {   // k = dz*4 + dy*2 + dx : tnw, tne, tsw, tse, bnw, bne, bsw, bse
            const int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
            w[k] = __fmul_rn(__fmul_rn(dx ? wx0 : wx1, dy ? wy0 : wy1), dz ? wz0 : wz1);
            v[k] = (dz ? zv1 : zv0) && (dy ? yv1 : yv0) && (dx ? xv1 : xv0);
            off[k] = (long)(z0 + dz) * HW + (long)(y0 + dy) * W + (x0 + dx);
        }
        for (int c = 0; c < C; ++c) {
            const float* sp = src + ((long)b * C + c) * V;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (v[k]) acc = __fadd_rn(acc, __fmul_rn(sp[off[k]], w[k]));
            out[((long)b * C + c) * V + p] = acc;
        }
    }
}

// 3-D variant of compute_jacobian.py:42-52 (np.gradient in float64; displacement channel i along axis i of (D, H, W))
__global__ void __launch_bounds__(256) jacobian_det_3d_kernel(const float* __restrict__ disp, double* __restrict__ det, int B, int D,
                                                              int H, int W) {
    const long HW = (long)H * W, V = (long)D * HW;
    const long total = (long)B * V;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / V);
        const long p = idx - (long)b * V;
        const int i = (int)(p / HW);
        const int q = (int)(p - (long)i * HW);
        const int j = q / W, k = q - j * W;
        const float* d = disp + (long)b * 3 * V;
        const int lo[3] = {i > 0 ? i - 1 : i, j > 0 ? j - 1 : j, k > 0 ? k - 1 : k};
        const int hi[3] = {i < D - 1 ? i + 1 : i, j < H - 1 ? j + 1 : j, k < W - 1 ? k + 1 : k};
        const int pos[3] = {i, j, k};
        double g[3][3];   // g[axis][component] = d(phi_component)/d(axis)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int ph[3] = {i, j, k}, pl[3] = {i, j, k};
            ph[a] = hi[a];
            pl[a] = lo[a];
            const long oh = (long)ph[0] * HW + (long)ph[1] * W + ph[2], ol = (long)pl[0] * HW + (long)pl[1] * W + pl[2];
            const double h = (double)(hi[a] - lo[a]);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                g[a][c] = (((double)d[c * V + oh] + (c == a ? ph[a] : pos[c])) - ((double)d[c * V + ol] + (c == a ? pl[a] : pos[c]))) / h;
        }
        const double d0 = g[0][0] * (g[1][1] * g[2][2] - g[1][2] * g[2][1]);
        const double d1 = g[0][1] * (g[1][0] * g[2][2] - g[1][2] * g[2][0]);
        const double d2 = g[0][2] * (g[1][0] * g[2][1] - g[1][1] * g[2][0]);
        det[idx] = d0 - d1 + d2;
    }
}

}  // namespace cf

using namespace cf;

// CF_WARP_SCALAR=1: one-pixel-per-thread kernels also for W % 4 == 0 (A/B knob)
static bool warp_v4(int W, double max_tensor_bytes) {
    static const bool scalar = [] { const char* e = getenv("CF_WARP_SCALAR"); return e && e[0] == '1'; }();
    return W % 4 == 0 && !scalar && max_tensor_bytes < 2147483648.0;   // 32-bit indices and buffer offsets inside the four-pixel kernels
}

extern "C" int cf_warp_bilinear_2d(const float* flow, const float* src, float* out, int B, int C, int H, int W, void* stream) {
    CF_REQUIRE(flow && src && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, "bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    CF_REQUIRE(out != src, "out must not alias src");
    long n = (long)B * H * W;
    const double bytes = 4.0 * n * (2.0 + 2.0 * C);   // flow + src read once, out written once
    if (warp_v4(W, 4.0 * n * (C > 2 ? C : 2)))
        launch_profiled(PK_WARP, bytes, warp_bilinear_2d_v4_kernel<false>, dim3(flat_grid(n / 4, 256)), dim3(256), as_stream(stream), flow,
                        src, out, B, C, H, W, (const float*)nullptr);
    else
        launch_profiled(PK_WARP, bytes, warp_bilinear_2d_kernel, dim3(flat_grid(n, 256)), dim3(256), as_stream(stream), flow, src, out, B, C,
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
        if (warp_v4(W, 8.0 * np))
            hipLaunchKernelGGL(warp_bilinear_2d_v4_kernel<true>, dim3(flat_grid(np / 4, 256)), dim3(256), 0, s, cur, cur, nxt, B, 2, H, W,
                               (const float*)cur);
        else
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
    const double bytes = 10.0 * n;   // flow read (8 B), one label gathered (1 B) and one written (1 B) per pixel
    if (warp_v4(W, 8.0 * n))
        launch_profiled(PK_WARP_LABELS, bytes, K == 4 ? warp_labels_2d_v4_kernel<4> : warp_labels_2d_v4_kernel<0>, dim3(flat_grid(n / 4, 256)),
                        dim3(256), as_stream(stream), flow, labels, out, T, B, K, H, W);
    else
        launch_profiled(PK_WARP_LABELS, bytes, warp_labels_2d_kernel, dim3(flat_grid(n, 256)), dim3(256), as_stream(stream), flow, labels,
                        out, T, B, K, H, W);
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
    const double bytes = 16.0 * n;   // displacement read (8 B), determinant written (8 B, float64 like numpy)
    if (warp_v4(W, 8.0 * n))
        launch_profiled(PK_JACOBIAN, bytes, jacobian_det_2d_v4_kernel, dim3(flat_grid(n / 4, 256)), dim3(256), as_stream(stream), disp, det, B,
                        H, W);
    else
        launch_profiled(PK_JACOBIAN, bytes, jacobian_det_2d_kernel, dim3(flat_grid(n, 256)), dim3(256), as_stream(stream), disp, det, B, H, W);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_warp_trilinear_3d(const float* flow, const float* src, float* out, int B, int C, int D, int H, int W, void* stream) {
    CF_REQUIRE(flow && src && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && D > 1 && H > 1 && W > 1, "bad shape B=%d C=%d D=%d H=%d W=%d", B, C, D, H, W);
    CF_REQUIRE(out != src, "out must not alias src");
    long n = (long)B * D * H * W;
    hipLaunchKernelGGL(warp_trilinear_3d_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, as_stream(stream), flow, src, out, B, C, D, H, W);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_jacobian_det_3d(const float* disp, double* det, int B, int D, int H, int W, void* stream) {
    CF_REQUIRE(disp && det, "null pointer");
    CF_REQUIRE(B > 0 && D >= 2 && H >= 2 && W >= 2, "bad shape");
    long n = (long)B * D * H * W;
    hipLaunchKernelGGL(jacobian_det_3d_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, as_stream(stream), disp, det, B, D, H, W);
    CF_CHECK_LAUNCH();
    return CF_OK;
}
