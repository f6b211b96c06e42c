// Direct fp32 convolution for the layers whose input has a handful of channels (the networks' stems: 1 -> 32 / 64, 6 -> 64; 3x3 pad 1
// or 1x1, stride 1).  On the MFMA kernels these layers pad K from 9 * Cin to 9 * 16 and run at ~6 TF -- 0.5-0.7 ms for what is a
// 1 GB write.  Here one thread owns one output column of PY rows: its (PY + 2) x 3 x Cin input neighbourhood sits in registers,
// the weights of the current output channel are wave-uniform scalar loads, every store is a 256-byte row segment per wave, and the
// GroupNorm / InstanceNorm statistics of the output are reduced per (sample, group) -- registers over the thread's rows, shuffles
// over the wave, LDS over the block, ONE fp64 atomic pair per group and block (same contract as conv_f16s.hip's fused statistics).
// Roofline: HBM; algorithmic bytes = 4 * B * H * W * (Cin + Cout).
#include "common.h"
#include "conv.h"
#include "profile.h"

namespace cf {

template <int CIN, int KS, int PY>
__global__ void __launch_bounds__(256) conv_small_cin_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ out, int H, int W, int Cout, int groups,
                                                             double* __restrict__ gn_ws) {
    constexpr int R = KS / 2;                 // halo
    constexpr int NR = PY + 2 * R, NC = KS;
    __shared__ float red[64 * 2];             // per group (<= 64): sum, sum of squares of this block
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64 + tx, y0 = (blockIdx.y * 4 + ty) * PY, b = blockIdx.z;
    const long HW = (long)H * W;
    const float* xb = x + (long)b * CIN * HW;
    float* ob = out + (long)b * Cout * HW;
    if (groups > 0) {
        for (int k = threadIdx.x; k < 2 * groups; k += 256) red[k] = 0.f;
        __syncthreads();
    }
    float in[CIN][NR][NC];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int yy = y0 + r - R, xx = x0 + c - R;
                in[ci][r][c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xb[ci * HW + (long)yy * W + xx] : 0.f;
            }
    bool ok[PY];
#pragma unroll
    for (int r = 0; r < PY; ++r) ok[r] = x0 < W && y0 + r < H;
    const int cpg = groups > 0 ? Cout / groups : Cout;
    const int ng = groups > 0 ? groups : 1;
    for (int g = 0; g < ng; ++g) {
        float gs = 0.f, gq = 0.f;
        for (int c = 0; c < cpg; ++c) {
            const int co = g * cpg + c;
            const float* wc = w + (long)co * CIN * KS * KS;     // wave-uniform: scalar loads
            const float bv = bias ? bias[co] : 0.f;
            float wr[CIN * KS * KS];
#pragma unroll
            for (int k = 0; k < CIN * KS * KS; ++k) wr[k] = wc[k];
#pragma unroll
            for (int r = 0; r < PY; ++r) {
                float acc = bv;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) acc = fmaf(wr[(ci * KS + ky) * KS + kx], in[ci][r + ky][kx], acc);
                if (ok[r]) {
                    ob[(long)co * HW + (long)(y0 + r) * W + x0] = acc;
                    gs += acc;
                    gq += acc * acc;
                }
            }
        }
        if (groups > 0) {
            for (int o = 32; o > 0; o >>= 1) {
                gs += __shfl_xor(gs, o, 64);
                gq += __shfl_xor(gq, o, 64);
            }
            if (tx == 0) {
                atomicAdd(&red[2 * g], gs);
                atomicAdd(&red[2 * g + 1], gq);
            }
        }
    }
    if (groups > 0) {
        __syncthreads();
        if ((int)threadIdx.x < 2 * groups) atomicAdd(&gn_ws[2L * b * groups + threadIdx.x], (double)red[threadIdx.x]);
    }
}

template <int CIN, int KS, int PY>
static int launch_small_cin(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cout, int groups, double* ws,
                            hipStream_t s) {
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((H + 4 * PY - 1) / (4 * PY)), (unsigned)B);
    hipLaunchKernelGGL((conv_small_cin_kernel<CIN, KS, PY>), grid, dim3(256), 0, s, x, w, bias, out, H, W, Cout, groups, ws);
    return CF_OK;
}

// The opposite corner: 3x3 / pad 1 / stride 1 convolutions with a handful of OUTPUT channels -- the 64 -> 2 flow head of Decoder2D
// (decoder_alt.py:890-892) and RAFT's FlowHead.conv2.  On the MFMA kernel two output channels fill 1/16 of a 32-channel m-tile: 16 TF, 1.2 ms
// at 128 x 256 x 256 for what is a 2.1 GB read.  Here a thread owns one output column of PY rows and all COUT channels (COUT * PY
// accumulators); per input channel it reads its (PY + 2) x 3 neighbourhood (the row segments of a wave are contiguous: whole 256-byte
// lines, the 3x overlap between neighbouring lanes is L1 traffic), the COUT * 9 weights of the channel are wave-uniform scalar loads.
// Exact fp32.  Roofline: HBM; algorithmic bytes = 4 * B * H * W * (Cin + COUT).
template <int COUT, int PY>
__global__ void __launch_bounds__(256, 4) conv3x3_small_cout_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                    const float* __restrict__ res, float* __restrict__ out, int Cin, int H, int W) {
    constexpr int NR = PY + 2;
    constexpr unsigned OOB = 0x80000000u;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64 + tx, y0 = (blockIdx.y * 4 + ty) * PY, b = blockIdx.z;
    const long HW = (long)H * W;
    // one buffer resource over this sample's input: the 30 byte offsets of the thread's neighbourhood in channel 0 are computed once (a tap
    // outside the image is parked out of range: the range check returns the zero padding), the channel term is one add per load
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long)b * Cin * HW), 0, (int)((long)Cin * HW * 4), 0x00020000);
    unsigned off[NR][3];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int yy = y0 + r - 1, xx = x0 + c - 1;
            off[r][c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (unsigned)(yy * W + xx) * 4u : OOB;
        }
    float acc[COUT][PY];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int r = 0; r < PY; ++r) acc[co][r] = bias ? bias[co] : 0.f;
    const unsigned HW4 = (unsigned)HW * 4u;
#pragma unroll 1
    for (int ci = 0; ci < Cin; ++ci) {
        const unsigned cofs = (unsigned)ci * HW4;
        float in[NR][3];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) in[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off[r][c] + cofs, 0, 0));
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float* wc = w + ((long)co * Cin + ci) * 9;      // wave-uniform: scalar loads
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float wv = wc[ky * 3 + kx];
#pragma unroll
                    for (int r = 0; r < PY; ++r) acc[co][r] = fmaf(wv, in[r + ky][kx], acc[co][r]);
                }
        }
    }
    if (x0 >= W) return;
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int r = 0; r < PY; ++r) {
            if (y0 + r >= H) continue;
            const long o = ((long)b * COUT + co) * HW + (long)(y0 + r) * W + x0;
            out[o] = res ? acc[co][r] + res[o] : acc[co][r];
        }
}

}  // namespace cf

using namespace cf;

extern "C" int cf_conv2d_small_cin(const float* x, const float* weight, const float* bias, float* out, int B, int Cin, int H, int W, int Cout, int K,
                                   double* gn_ws, int gn_groups, void* stream) {
    CF_REQUIRE(x && weight && out && x != out, "null or aliased pointer");
    CF_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && Cout > 0, "bad shape B=%d H=%d W=%d Cout=%d", B, H, W, Cout);
    CF_REQUIRE((K == 3 || K == 1) && (Cin == 1 || Cin == 2 || Cin == 6), "built for Cin 1, 2, 6 and 3x3 (pad 1) / 1x1 kernels, got Cin=%d K=%d", Cin, K);
    CF_REQUIRE(!gn_ws || (gn_groups > 0 && gn_groups <= 64 && Cout % gn_groups == 0), "bad GroupNorm statistics request (groups <= 64 dividing Cout)");
    hipStream_t s = as_stream(stream);
    const int groups = gn_ws ? gn_groups : 0;
    if (gn_ws && hipMemsetAsync(gn_ws, 0, 2L * B * groups * sizeof(double), s) != hipSuccess) { set_error("cf_conv2d_small_cin: memset failed"); return CF_ERR_LAUNCH; }
    if (K == 3) {
        if (Cin == 1) launch_small_cin<1, 3, 8>(x, weight, bias, out, B, H, W, Cout, groups, gn_ws, s);
        else if (Cin == 2) launch_small_cin<2, 3, 4>(x, weight, bias, out, B, H, W, Cout, groups, gn_ws, s);
        else launch_small_cin<6, 3, 2>(x, weight, bias, out, B, H, W, Cout, groups, gn_ws, s);
    } else {
        if (Cin == 1) launch_small_cin<1, 1, 8>(x, weight, bias, out, B, H, W, Cout, groups, gn_ws, s);
        else if (Cin == 2) launch_small_cin<2, 1, 8>(x, weight, bias, out, B, H, W, Cout, groups, gn_ws, s);
        else launch_small_cin<6, 1, 8>(x, weight, bias, out, B, H, W, Cout, groups, gn_ws, s);
    }
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_conv2d_small_cout(const float* x, const float* weight, const float* bias, const float* res, float* out, int B, int Cin, int H, int W,
                                    int Cout, void* stream) {
    CF_REQUIRE(x && weight && out && x != out, "null or aliased pointer");
    CF_REQUIRE(B > 0 && B < 65536 && Cin > 0 && H > 0 && W > 0, "bad shape B=%d Cin=%d H=%d W=%d", B, Cin, H, W);
    CF_REQUIRE(Cout >= 1 && Cout <= 4, "built for 1..4 output channels, got %d", Cout);
    CF_REQUIRE((double)Cin * H * W * 4.0 < 2147483648.0, "one input sample must stay below 2 GiB (32-bit buffer offsets)");
    hipStream_t s = as_stream(stream);
    constexpr int PY = 8;
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((H + 4 * PY - 1) / (4 * PY)), (unsigned)B);
    hipEvent_t e0, e1;
    const double bytes = 4.0 * B * H * W * (Cin + Cout);
    const bool prof = profile_on() && profile_events(PK_CONV_SMALL_COUT, bytes, &e0, &e1);
#define CF_LAUNCH_SC(N)                                                                                                                          \
    if (prof) hipExtLaunchKernelGGL((conv3x3_small_cout_kernel<N, PY>), grid, dim3(256), 0, s, e0, e1, 0, x, weight, bias, res, out, Cin, H, W);   \
    else hipLaunchKernelGGL((conv3x3_small_cout_kernel<N, PY>), grid, dim3(256), 0, s, x, weight, bias, res, out, Cin, H, W)
    switch (Cout) {
        case 1: CF_LAUNCH_SC(1); break;
        case 2: CF_LAUNCH_SC(2); break;
        case 3: CF_LAUNCH_SC(3); break;
        default: CF_LAUNCH_SC(4); break;
    }
#undef CF_LAUNCH_SC
    CF_CHECK_LAUNCH();
    return CF_OK;
}
