// RAFT all-pairs correlation volume + its 4-level average pyramid in ONE kernel (CorrBlock.__init__ of the published RAFT; call site
// nnunet/network_architecture/SegFlowGaussian.py:929).
//
//   level 0 : corr[b, n1, n2] = (1 / sqrt(C)) * sum_c f1[b, c, n1] * f2[b, c, n2]          n1, n2 over the H x W feature map
//   level l : avg_pool2d(level l-1, 2, 2) over the (y2, x2) axes of n2
//
// Algorithmic traffic: read f1 and f2 once (2 C N 4 B), write N^2 4 B (1 + 1/4 + 1/16 + 1/64) -- 7.67 MB per
// 256-channel 32x32 pair, 82 % of it the write: the op is HBM-write bound, IF the 0.54 GFLOP of the product hide under the stores.
// The first version ran level 0 as an fp32-MFMA implicit GEMM with per-sample "weights" (48 TFLOP/s) and pooled in three more
// launches that re-read what had just been written: 708 us at B = 64 = 0.69 TB/s.  Here:
//   * the product runs on the f16 MFMA with the 3-term hi/lo split of conv_f16s.hip (fp32-class: 2^-22 relative operand error);
//   * f1 / f2 tiles are transposed and split on the fly into 80-byte pixel records in LDS (16 channels per chunk, double buffered);
//   * a wave owns 32 n1 rows x 8 n2 map rows (8 accumulator tiles): lanes run along x2, so level-0 stores are 128-byte row
//     segments, and every pooled level comes out of the SAME registers -- x2 neighbours by __shfl_xor(1 / 2 / 4), y2 neighbours are
//     accumulator tiles of the same wave -- in ATen's summation order on the rounded level-0 values, bit-compatible with pooling the
//     stored level;
//   * workgroup = 8 waves = 128 n1 x 16 map rows (512 n2); grid = B x (N / 128) x (H / 16), n1 blocks fastest so that the
//     workgroups sharing an f2 tile run together.
// Built for W == 32, H % 16 == 0, C % 16 == 0 (RAFT at 1/8 resolution of 256 x 256: 32 x 32 x 256); other shapes keep the generic path.
#include <hip/hip_fp16.h>

#include "common.h"
#include "profile.h"

namespace cf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int AP_CK = 16;                  // channels per chunk
constexpr int AP_REC = AP_CK * 4 + 16;     // 80-byte record: [hi 16 | lo 16 | pad]
constexpr int AP_M = 128, AP_ROWS = 16;    // n1 per workgroup, n2 map rows per workgroup (x 32 columns)
constexpr int AP_NPOS = AP_M + AP_ROWS * 32;   // positions staged per chunk: 128 of f1 + 512 of f2
constexpr int AP_TASKS = AP_NPOS * 2;      // (position, 8-channel group)
constexpr int AP_MAXT = (AP_TASKS + 511) / 512;

__global__ void __launch_bounds__(512, 2) allpairs_pyramid_kernel(const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ pyr,
                                                                  int B, int C, int H, float scale) {
    constexpr int W = 32;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * AP_NPOS * AP_REC];
    const int N = H * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int mblocks = N / AP_M, rgroups = H / AP_ROWS;
    // XCD-banded: consecutive logical ids (the workgroups that share an f2 tile, then the two row groups of a sample) run on ONE XCD and
    // meet in its L2 -- dispatched round-robin they re-fetched every tile once per XCD (PMC: reads 2.25x algorithmic before this)
    int bid = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int mb = bid % mblocks;
    bid /= mblocks;
    const int rg = bid % rgroups;
    const int b = bid / rgroups;
    const int n1_0 = mb * AP_M, y2_0 = rg * AP_ROWS;
    const int wm = wave & 3, wg = wave >> 2;           // n1 sub-block (32 rows), row group (8 map rows)

    // ---- staging tasks: position p < 128 -> f1[n1_0 + p], else f2[(y2_0 * 32) + p - 128]; 8 channels per task
    const float* src[AP_MAXT];
    int dst[AP_MAXT];
#pragma unroll
    for (int t = 0; t < AP_MAXT; ++t) {
        const int task = tid + t * 512;
        dst[t] = -1;
        src[t] = f1;
        if (task < AP_TASKS) {
            const int grp = task / AP_NPOS, p = task - grp * AP_NPOS;
            const float* base = p < AP_M ? f1 + (long)b * C * N + n1_0 + p : f2 + (long)b * C * N + y2_0 * W + (p - AP_M);
            src[t] = base + (long)grp * 8 * N;
            dst[t] = p * AP_REC + grp * 16;
        }
    }
    float stg[AP_MAXT][8];
    auto issue = [&](int chunk) {
#pragma unroll
        for (int t = 0; t < AP_MAXT; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) stg[t][j] = dst[t] >= 0 ? src[t][((long)chunk * AP_CK + j) * N] : 0.f;
    };
    auto write = [&](int chunk) {
        unsigned char* base = lds + (chunk & 1) * (AP_NPOS * AP_REC);
#pragma unroll
        for (int t = 0; t < AP_MAXT; ++t) {
            if (dst[t] < 0) continue;
            f16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const _Float16 h = (_Float16)stg[t][j];
                hi[j] = h;
                lo[j] = (_Float16)(stg[t][j] - (float)h);
            }
            *reinterpret_cast<f16x8*>(base + dst[t]) = hi;
            *reinterpret_cast<f16x8*>(base + dst[t] + AP_CK * 2) = lo;
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunk = C / AP_CK;
    issue(0);
    write(0);
    __syncthreads();
    const int a_off = (wm * 32 + l31) * AP_REC + half * 16;
    const int b_off = (AP_M + (wg * 8) * 32 + l31) * AP_REC + half * 16;
    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        if (more) issue(c + 1);
        const unsigned char* xb = lds + (c & 1) * (AP_NPOS * AP_REC);
        const f16x8 ah = *reinterpret_cast<const f16x8*>(xb + a_off);
        const f16x8 al = *reinterpret_cast<const f16x8*>(xb + a_off + AP_CK * 2);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const unsigned char* rp = xb + b_off + t * 32 * AP_REC;
            const f16x8 bh = *reinterpret_cast<const f16x8*>(rp);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(rp + AP_CK * 2);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
        }
        if (more) write(c + 1);
        __syncthreads();
    }

    // ---- epilogue: level 0 and the three pooled levels from the same registers
    const long NN = (long)N;
    float* l0 = pyr;
    float* l1 = l0 + (long)B * NN * N;
    float* l2 = l1 + (long)B * NN * (N >> 2);
    float* l3 = l2 + (long)B * NN * (N >> 4);
    const int y2w = y2_0 + wg * 8;                       // first map row of this wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n1 = n1_0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const long row = (long)b * NN + n1;
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            v[t] = acc[t][r] * scale;
            l0[(row * H + y2w + t) * W + l31] = v[t];
        }
        // level 1: ((ip[0] + ip[1]) + ip[W]) + ip[W+1]) / 4 on the rounded level-0 values (ATen avg_pool2d order)
        float p1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float top = v[2 * t] + __shfl_xor(v[2 * t], 1, 64);                 // (even lanes: own + right neighbour)
            const float bot_r = __shfl_xor(v[2 * t + 1], 1, 64);
            p1[t] = ((top + v[2 * t + 1]) + bot_r) / 4.0f;
            if ((l31 & 1) == 0) l1[(row * (H >> 1) + (y2w >> 1) + t) * (W >> 1) + (l31 >> 1)] = p1[t];
        }
        float p2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float top = p1[2 * t] + __shfl_xor(p1[2 * t], 2, 64);
            const float bot_r = __shfl_xor(p1[2 * t + 1], 2, 64);
            p2[t] = ((top + p1[2 * t + 1]) + bot_r) / 4.0f;
            if ((l31 & 3) == 0) l2[(row * (H >> 2) + (y2w >> 2) + t) * (W >> 2) + (l31 >> 2)] = p2[t];
        }
        {
            const float top = p2[0] + __shfl_xor(p2[0], 4, 64);
            const float bot_r = __shfl_xor(p2[1], 4, 64);
            const float p3 = ((top + p2[1]) + bot_r) / 4.0f;
            if ((l31 & 7) == 0) l3[(row * (H >> 3) + (y2w >> 3)) * (W >> 3) + (l31 >> 3)] = p3;
        }
    }
}

// returns CF_OK when the fused kernel took the job, 1 when the shape is outside what it is built for (caller uses the generic path)
int allpairs_pyramid_fused(const float* f1, const float* f2, float* pyr, int B, int C, int H, int W, int levels, hipStream_t stream) {
    if (!(W == 32 && H % AP_ROWS == 0 && C % AP_CK == 0 && levels == 4 && H >= 16)) return 1;
    const int N = H * W;
    const long nblk = (long)B * (N / AP_M) * (H / AP_ROWS);
    if (nblk >= (1L << 31) || (nblk & 7)) return 1;
    // algorithmic bytes: f1 + f2 read once, the four levels written once
    const double bytes = 4.0 * B * (2.0 * C * N + (double)N * N * (1.0 + 0.25 + 0.0625 + 0.015625));
    launch_profiled(PK_ALLPAIRS, bytes, allpairs_pyramid_kernel, dim3((unsigned)nblk), dim3(512), stream, f1, f2, pyr, B, C, H,
                    (float)(1.0 / sqrt((double)C)));
    CF_CHECK_LAUNCH();
    return CF_OK;
}

}  // namespace cf
