// Correlation kernels: the local cost volume (CorrVolume), the RAFT all-pairs pyramid + lookup, convex upsampling.
#include "conv.h"
#include "profile.h"

namespace cf {

// =====================================================================================================================
// Local cost volume, radius 4 (9x9 = 81 displacements), dilation s in {1,2,4}:
//   out[b,(dy+4)*9+(dx+4),y,x] = (1/C) sum_c cur[b,c,y,x] * prev[b,c,y+dy*s,x+dx*s]      (zero outside)
//
// Algorithmic traffic: read cur+prev once (2*C*H*W*4 B), write 81*H*W*4 B; 2*81*C flops per pixel -> ~12-17 flop/B,
// i.e. right at the fp32-VALU / HBM ridge of gfx950, so the inner loop must run near the VALU rate:
//   * a dilated problem splits into independent residue classes: pixel (y,x) only meets prev pixels with the same
//     (y mod s, x mod s).  A workgroup takes 4 sub-rows of ONE y-class x 64 real columns (all x-classes), stages the
//     tile de-interleaved by x-class in LDS ([c][row][x-class][i]) so every thread's operands are contiguous;
//   * thread = (dy, sub-row, 8 same-class pixels): 72 accumulators, per channel 2+4 ds_read_b128 feed 72 FMAs
//     (12 FMA per LDS read instruction, 4.5 per loaded float) -> VALU-bound, not LDS-bound;
//   * 9 dy x 32 (row,octet) = 288 threads (5 waves, last half-wave idle in the FMA loop, all 320 threads stage).
// Global reads are full contiguous row segments (coalesced); writes of one wave cover whole 256-B row segments.
// =====================================================================================================================
constexpr int CV_CC = 4;        // channels per LDS chunk
constexpr int CV_THREADS = 512; // waves 0-4: compute (288 lanes used), waves 5-7: staging
constexpr int CV_STAGERS = 192;
constexpr int CV_MAXT = 2;      // float4 staging tasks per staging thread (12*(64+8S)/4 + 64 groups <= 2*192 for S in {1,2,4})

template <int S>
__global__ void __launch_bounds__(CV_THREADS, 4) corr_volume_r4_kernel(const float* __restrict__ cur, const float* __restrict__ prev,
                                                                   float* __restrict__ out, int B, int C, int H, int W,
                                                                   int tiles_x, int tiles_y) {
    constexpr int NI = 64 / S;        // same-class pixels per tile row
    constexpr int PW = NI + 8;        // prev sub-row length (halo 4 each side)
    constexpr int PROW = S * PW;      // floats per (c, prev row) == real columns staged per row
    constexpr int CROW = 64;          // floats per (c, cur row)
    constexpr int PREV_C = 12 * PROW; // per channel
    constexpr int CUR_C = 4 * CROW;
    constexpr int SLAB = PREV_C + CUR_C;        // floats per channel: [prev | cur]
    constexpr int BUF = CV_CC * SLAB;           // floats per buffer
    constexpr int PGRP = PROW / 4;              // float4 groups per prev row
    constexpr int NGRP = 12 * PGRP + 4 * 16;    // staging groups per channel
    static_assert(NGRP <= CV_MAXT * CV_STAGERS, "staging tasks");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    // ---- tile decode; blocks sharing an XCD (bid % 8) take a contiguous band of the tile list
    const int nb = gridDim.x;
    int bid = blockIdx.x;
    {
        int xcd = bid & 7, qn = nb >> 3, rn = nb & 7;
        bid = ((xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
    }
    const int tx = bid % tiles_x;
    int t = bid / tiles_x;
    const int ry = t % S;  // y residue class
    t /= S;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int xtile = tx * 64;
    const int ytile = ty * 4 * S;
    const int HW = H * W;
    const unsigned HW4 = (unsigned)HW * 4u;

    // ---- staging tasks: one float4 (4 consecutive real x) of one tile row, for each of the CV_CC channels of a chunk.
    // Raw buffer loads: rows / columns outside the image park the offset at 2 GiB, channels >= C run past num_records
    // (the descriptor covers exactly this sample's C planes) -> the range check returns 0, no branches, no masks.
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x;
    const bool stager = tid >= 320;   // wave-uniform role split: staging registers never live in the FMA waves
    const int st = tid - 320;
    unsigned t_off[CV_MAXT];   // byte offset inside the sample (channel 0)
    int t_lds[CV_MAXT];        // float index inside a channel slab of the first of the 4 elements, -1: no task
    bool t_prev[CV_MAXT];
#pragma unroll
    for (int k = 0; k < CV_MAXT; ++k) {
        const int grp = stager ? st + k * CV_STAGERS : NGRP;
        t_off[k] = OOB;
        t_lds[k] = -1;
        t_prev[k] = true;
        if (grp < 12 * PGRP) {
            const int pr = grp / PGRP, col = (grp - pr * PGRP) * 4;   // real column offset inside the staged row
            const int y = ytile + S * (pr - 4) + ry, x = xtile - 4 * S + col;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) t_off[k] = (unsigned)(y * W + x) * 4u;
            // de-interleave by x class: element e (0..3) of the float4 has class (col+e)%S and index (col+e)/S
            t_lds[k] = pr * PROW + (col % S) * PW + col / S;
        } else if (grp < NGRP) {
            const int g2 = grp - 12 * PGRP;
            const int rr = g2 >> 4, col = (g2 & 15) * 4;
            const int y = ytile + S * rr + ry, x = xtile + col;
            if (y < H && x < W) t_off[k] = (unsigned)(y * W + x) * 4u;
            t_lds[k] = PREV_C + rr * CROW + (col % S) * NI + col / S;
            t_prev[k] = false;
        }
    }
    const long sample = (long)b * C * HW;
    const __amdgpu_buffer_rsrc_t rs_prev = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(prev + sample), 0, (int)((long)C * HW4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cur + sample), 0, (int)((long)C * HW4), 0x00020000);

    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int nchunk = (C + CV_CC - 1) / CV_CC;
    if (stager) {
        // ---- staging waves.  Two register sets (A, B) keep the loads of TWO chunks in flight, so the wait in front of
        // each LDS write is for loads issued a whole chunk period earlier (vmcnt retires in order: waiting for the older
        // set leaves the younger one in flight).  The loop is unrolled by two so that the sets are statically indexed.
        // Same number of barriers as the FMA branch (1 + nchunk).
        f32x4 sa[CV_MAXT][CV_CC], sb[CV_MAXT][CV_CC];
        auto issue_loads = [&](int ch, f32x4 (&stg)[CV_MAXT][CV_CC]) {
            const int c0 = ch * CV_CC;
#pragma unroll
            for (int k = 0; k < CV_MAXT; ++k)
#pragma unroll
                for (int c = 0; c < CV_CC; ++c) {
                    const unsigned off = t_off[k] + (unsigned)(c0 + c) * HW4;
                    stg[k][c] = t_prev[k] ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_prev, off, 0, 0))
                                          : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_cur, off, 0, 0));
                }
        };
        auto write_stage = [&](int ch, const f32x4 (&stg)[CV_MAXT][CV_CC]) {
            float* base = lds + (ch & 1) * BUF;
#pragma unroll
            for (int k = 0; k < CV_MAXT; ++k) {
                if (t_lds[k] < 0) continue;
                const int cls_stride = t_prev[k] ? PW : NI;
#pragma unroll
                for (int c = 0; c < CV_CC; ++c) {
                    float* d = base + c * SLAB + t_lds[k];
                    if (S == 1) {
                        *reinterpret_cast<f32x4*>(d) = stg[k][c];
                    } else if (S == 2) {  // classes 0,1,0,1 : elements (0,2) and (1,3) are neighbours inside their class rows
                        d[0] = stg[k][c][0]; d[1] = stg[k][c][2];
                        d[cls_stride] = stg[k][c][1]; d[cls_stride + 1] = stg[k][c][3];
                    } else {              // S == 4: one element per class
                        d[0] = stg[k][c][0]; d[cls_stride] = stg[k][c][1]; d[2 * cls_stride] = stg[k][c][2]; d[3 * cls_stride] = stg[k][c][3];
                    }
                }
            }
        };
        if (S == 4) {
            // the scalar de-interleaving LDS writes of the stride-4 variant need the registers of the second set:
            // one chunk in flight only (no spill at 128 VGPRs)
            issue_loads(0, sa);
            write_stage(0, sa);
            __syncthreads();
            for (int ch = 0; ch < nchunk; ++ch) {
                if (ch + 1 < nchunk) {
                    issue_loads(ch + 1, sa);
                    write_stage(ch + 1, sa);
                }
                __syncthreads();
            }
            return;
        }
        issue_loads(0, sa);
        if (nchunk > 1) issue_loads(1, sb);
        write_stage(0, sa);
        __syncthreads();
        int ch = 0;
        while (true) {
            // chunk ch is in LDS; set B holds chunk ch+1 (in flight); set A is free
            if (ch + 2 < nchunk) issue_loads(ch + 2, sa);
            if (ch + 1 < nchunk) write_stage(ch + 1, sb);
            __syncthreads();
            if (++ch >= nchunk) break;
            // chunk ch is in LDS; set A holds chunk ch+1; set B is free
            if (ch + 2 < nchunk) issue_loads(ch + 2, sb);
            if (ch + 1 < nchunk) write_stage(ch + 1, sa);
            __syncthreads();
            if (++ch >= nchunk) break;
        }
        return;
    }

    // ---- FMA waves (the roles live in separate loops so that staging registers and accumulators never coexist)
    const bool compute = tid < 288;
    const int dy = tid >> 5;          // 0..8  (displacement dy-4)
    const int qd = tid & 31;
    const int r = qd >> 3;            // sub-row 0..3
    const int u = qd & 7;
    const int rx = u % S;             // x residue class
    const int o = u / S;              // octet within the class row
    const int prev_off = ((r + dy) * S + rx) * PW + 8 * o;       // + c*SLAB ; window [8o, 8o+16)
    const int cur_off = PREV_C + r * CROW + rx * NI + 8 * o;     // + c*SLAB

    float acc[8][9];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int d = 0; d < 9; ++d) acc[j][d] = 0.f;

    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        if (compute) {
            const float* xb = lds + (ch & 1) * BUF;
#pragma unroll
            for (int c = 0; c < CV_CC; ++c) {
                const float4* pp = reinterpret_cast<const float4*>(xb + c * SLAB + prev_off);
                const float4* cp = reinterpret_cast<const float4*>(xb + c * SLAB + cur_off);
                float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
                float4 c0v = cp[0], c1v = cp[1];
                const float pv[16] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w, p3.x, p3.y, p3.z, p3.w};
                const float cv[8] = {c0v.x, c0v.y, c0v.z, c0v.w, c1v.x, c1v.y, c1v.z, c1v.w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int d = 0; d < 9; ++d) acc[j][d] = fmaf(cv[j], pv[j + d], acc[j][d]);
            }
        }
        __syncthreads();
    }
    if (!compute) return;
    const int y = ytile + S * r + ry;
    if (y >= H) return;
    const float invC = 1.0f / (float)C;
    float* ob = out + (long)b * 81 * HW + (long)y * W;
#pragma unroll
    for (int d = 0; d < 9; ++d) {
        float* oc = ob + (long)(dy * 9 + d) * HW;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int x = xtile + S * (8 * o + j) + rx;
            if (x < W) oc[x] = acc[j][d] * invC;
        }
    }
}

// generic fallback (any radius / stride): one thread per output element
__global__ void __launch_bounds__(256) corr_volume_generic_kernel(const float* __restrict__ cur, const float* __restrict__ prev,
                                                                 float* __restrict__ out, int B, int C, int H, int W, int radius,
                                                                 int stride) {
    const int D = 2 * radius + 1;
    const long HW = (long)H * W;
    const long total = (long)B * D * D * HW;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int p = (int)(idx % HW);
        long rest = idx / HW;
        int ch = (int)(rest % (D * D));
        int b = (int)(rest / (D * D));
        int y = p / W, x = p - y * W;
        int yy = y + (ch / D - radius) * stride, xx = x + (ch % D - radius) * stride;
        float s = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            const float* cp = cur + (long)b * C * HW + p;
            const float* pp = prev + (long)b * C * HW + (long)yy * W + xx;
            for (int c = 0; c < C; ++c) s = fmaf(cp[(long)c * HW], pp[(long)c * HW], s);
        }
        out[idx] = s / (float)C;
    }
}

// =====================================================================================================================
// RAFT all-pairs pyramid: level 0 by the MFMA GEMM (conv.hip, per-sample "weights" = fmap1), then 2x2 average pools.
// =====================================================================================================================
__global__ void __launch_bounds__(256) avgpool2x2_kernel(const float* __restrict__ in, float* __restrict__ out, long planes, int H,
                                                        int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = planes * Ho * Wo;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int xo = (int)(idx % Wo);
        long rest = idx / Wo;
        int yo = (int)(rest % Ho);
        long pl = rest / Ho;
        const float* ip = in + pl * H * W + (long)(2 * yo) * W + 2 * xo;
        // ATen avg_pool2d sums the window in row-major order then divides
        out[idx] = (((ip[0] + ip[1]) + ip[W]) + ip[W + 1]) / 4.0f;
    }
}

// lookup: out[b, l*D*D + i*D + j, y, x] = bilinear(level_l[b, n=(y,x)], cx/2^l + (i-r), cy/2^l + (j-r)), zeros outside,
// with RAFT's bilinear_sampler normalisation (2*x/(W-1) - 1) and grid_sample's align_corners un-normalisation.
__global__ void __launch_bounds__(256) corr_lookup_kernel(const float* __restrict__ pyr, const float* __restrict__ coords,
                                                         float* __restrict__ out, int B, int H, int W, int levels, int radius) {
    const int D = 2 * radius + 1;
    const int N = H * W;
    const int CH = levels * D * D;
    const long total = (long)B * CH * N;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int n = (int)(idx % N);
        long rest = idx / N;
        int ch = (int)(rest % CH);
        int b = (int)(rest / CH);
        int l = ch / (D * D);
        int ij = ch - l * D * D;
        int i = ij / D, j = ij - i * D;
        long off = 0;
        for (int q = 0; q < l; ++q) off += (long)B * N * (H >> q) * (W >> q);
        const int Hl = H >> l, Wl = W >> l;
        const float* plane = pyr + off + ((long)b * N + n) * Hl * Wl;
        const float inv = 1.0f / (float)(1 << l);
        float cx = __fadd_rn(__fmul_rn(coords[(long)b * 2 * N + n], inv), (float)(i - radius));       // x + dy[i]  (RAFT ordering)
        float cy = __fadd_rn(__fmul_rn(coords[(long)b * 2 * N + N + n], inv), (float)(j - radius));   // y + dx[j]
        float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, cx), (float)(Wl - 1)), 1.0f);
        float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, cy), (float)(Hl - 1)), 1.0f);
        float x = __fmul_rn(__fdiv_rn(__fadd_rn(gx, 1.0f), 2.0f), (float)(Wl - 1));
        float y = __fmul_rn(__fdiv_rn(__fadd_rn(gy, 1.0f), 2.0f), (float)(Hl - 1));
        Taps t = make_taps(y, x, Hl, Wl);
        out[idx] = sample_taps(plane, t, Wl);
    }
}

// convex upsampling: thread = (b, i, j, y, x) with x fastest -> the 9 mask reads are coalesced
__global__ void __launch_bounds__(256) convex_upsample_kernel(const float* __restrict__ flow, const float* __restrict__ mask,
                                                             float* __restrict__ out, int B, int C, int h, int w) {
    const int hw = h * w;
    const long total = (long)B * 64 * hw;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int p = (int)(idx % hw);
        long rest = idx / hw;
        int ij = (int)(rest % 64);
        int b = (int)(rest / 64);
        int y = p / w, x = p - y * w;
        const float* mb = mask + ((long)b * 576 + ij) * hw + p;
        float m[9];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) { m[k] = mb[(long)k * 64 * hw]; mx = fmaxf(mx, m[k]); }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); sum += m[k]; }
        const int i = ij >> 3, j = ij & 7;
        for (int c = 0; c < C; ++c) {
            const float* fb = flow + ((long)b * C + c) * hw;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
                float f = ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) ? 8.0f * fb[yy * w + xx] : 0.f;
                acc += (m[k] / sum) * f;
            }
            out[(((long)b * C + c) * (8 * h) + 8 * y + i) * (8 * w) + 8 * x + j] = acc;
        }
    }
}

}  // namespace cf

using namespace cf;

extern "C" int cf_corr_volume(const float* cur, const float* prev, float* out, int B, int C, int H, int W, int radius, int stride,
                              void* stream) {
    CF_REQUIRE(cur && prev && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && radius >= 0 && radius <= 8 && stride >= 1, "bad shape");
    hipStream_t s = as_stream(stream);
    if (radius == 4 && (stride == 1 || stride == 2 || stride == 4) && (W & 3) == 0 && (long)C * H * W * 4 < (1L << 31) &&
        ((reinterpret_cast<uintptr_t>(cur) | reinterpret_cast<uintptr_t>(prev)) & 15) == 0) {
        int tiles_x = (W + 63) / 64, tiles_y = (H + 4 * stride - 1) / (4 * stride);
        long nblk = (long)B * tiles_y * stride * tiles_x;
        CF_REQUIRE(nblk < (1L << 31), "grid too large");
        dim3 grid((unsigned)nblk), block(CV_THREADS);
        // algorithmic bytes: read cur + prev once, write the 81-channel volume once
        const double bytes = 4.0 * (double)B * H * W * (2.0 * C + 81.0);
        if (stride == 1) launch_profiled(PK_CORRVOL_S1, bytes, corr_volume_r4_kernel<1>, grid, block, s, cur, prev, out, B, C, H, W, tiles_x, tiles_y);
        else if (stride == 2) launch_profiled(PK_CORRVOL_S2, bytes, corr_volume_r4_kernel<2>, grid, block, s, cur, prev, out, B, C, H, W, tiles_x, tiles_y);
        else launch_profiled(PK_CORRVOL_S4, bytes, corr_volume_r4_kernel<4>, grid, block, s, cur, prev, out, B, C, H, W, tiles_x, tiles_y);
    } else {
        long total = (long)B * (2 * radius + 1) * (2 * radius + 1) * H * W;
        hipLaunchKernelGGL(corr_volume_generic_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, s, cur, prev, out, B, C, H, W, radius,
                           stride);
    }
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_corr_pyramid(const float* f1, const float* f2, float* pyr, int B, int C, int H, int W, int levels, void* stream) {
    CF_REQUIRE(f1 && f2 && pyr, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && levels >= 1 && levels <= 6, "bad shape");
    CF_REQUIRE((H * W) % 64 == 0, "H*W must be a multiple of 64 (got %d)", H * W);
    CF_REQUIRE((H >> (levels - 1)) >= 2 && (W >> (levels - 1)) >= 2, "coarsest level must be at least 2x2");
    hipStream_t s = as_stream(stream);
    const int N = H * W;
    // level 0: out[b, n1, n2] = sum_c f1[b,c,n1] * f2[b,c,n2] / sqrt(C): a 1x1 "conv" of f2 with per-sample weights Wt[k=c][m=n1] = f1[b]
    ConvParams p;
    p.x1 = f2; p.x2 = nullptr; p.wt = f1; p.bias = nullptr; p.res = nullptr; p.out = pyr; p.w_bstride = (long)C * N;
    p.C1 = C; p.C2 = 0; p.B = B; p.H = H; p.W = W; p.Cout = N; p.KH = 1; p.KW = 1; p.stride = 1; p.pad_h = 0; p.pad_w = 0;
    p.Ho = H; p.Wo = W; p.out_ctotal = N; p.out_coff = 0; p.act = CF_ACT_NONE; p.alpha = (float)(1.0 / sqrt((double)C));
    p.scatter2x2 = 0; p.gn_ws = nullptr; p.gn_groups = 0;
    int rc = launch_conv(p, s);
    if (rc != CF_OK) return rc;
    long off = 0;
    for (int l = 1; l < levels; ++l) {
        int Hl = H >> (l - 1), Wl = W >> (l - 1);
        long planes = (long)B * N;
        const float* in = pyr + off;
        off += planes * Hl * Wl;
        long total = planes * (Hl >> 1) * (Wl >> 1);
        hipLaunchKernelGGL(avgpool2x2_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, s, in, pyr + off, planes, Hl, Wl);
        CF_CHECK_LAUNCH();
    }
    return CF_OK;
}

extern "C" int cf_corr_lookup(const float* pyr, const float* coords, float* out, int B, int H, int W, int levels, int radius,
                              void* stream) {
    CF_REQUIRE(pyr && coords && out, "null pointer");
    CF_REQUIRE(B > 0 && H > 0 && W > 0 && levels >= 1 && levels <= 6 && radius >= 0 && radius <= 8, "bad shape");
    CF_REQUIRE((H >> (levels - 1)) >= 2 && (W >> (levels - 1)) >= 2, "coarsest level must be at least 2x2");
    long total = (long)B * levels * (2 * radius + 1) * (2 * radius + 1) * H * W;
    hipLaunchKernelGGL(corr_lookup_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, as_stream(stream), pyr, coords, out, B, H, W,
                       levels, radius);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_convex_upsample(const float* flow, const float* mask, float* out, int B, int C, int h, int w, void* stream) {
    CF_REQUIRE(flow && mask && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0, "bad shape");
    long total = (long)B * 64 * h * w;
    hipLaunchKernelGGL(convex_upsample_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, as_stream(stream), flow, mask, out, B, C, h, w);
    CF_CHECK_LAUNCH();
    return CF_OK;
}
