// Implicit-GEMM convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products, fp32 accumulate
// (bit-equivalent to a k-ordered fmaf chain), so the flow recurrence keeps the reference's fp32 numerics.
//
//   GEMM view   D[m = co][n = (b,oy,ox)] = sum_k  Wt[k][co] * X[b, ci(k), oy*s+kh(k)-ph, ox*s+kw(k)-pw]
//   k = ci*KH*KW + kh*KW + kw  (the natural weight order), two k per MFMA: lanes 0-31 take k, lanes 32-63 k+1.
//
// Wave tile: (32*MT) output channels x (32*NT) consecutive flat output pixels; a workgroup is four waves on four
// consecutive pixel tiles of the same channel block, so the A (weight) fragments hit L1 and the B (input) fragments
// are 128-byte coalesced rows shifted by (kh,kw) -- the 3x3 halo is served by L1/L2, never re-read from HBM.
// Operands are prefetched PF k-steps ahead in registers (no LDS, no barriers: every wave is independent and may exit
// early).  Epilogue fuses alpha, bias, activation, residual, channel-offset (torch.cat elimination) and the
// ConvTranspose2d(k=2,s=2) pixel scatter.
#include <stdlib.h>

#include "conv.h"
#include "profile.h"

namespace cf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// blocks that share `bid % 8` share an XCD (and its L2): give each XCD a contiguous band of pixel tiles so that the
// rows shared by vertically adjacent tiles are fetched into one L2 only.  Bijective for any nb.
__device__ __forceinline__ int xcd_band_remap(int bid, int nb) {
    int xcd = bid & 7, q = nb >> 3, r = nb & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <int MT, int NT>
__global__ void __launch_bounds__(256) conv_igemm_f32_kernel(const ConvParams p) {
    constexpr int PF = 4;  // k-steps of register prefetch
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int HoWo = p.Ho * p.Wo;
    const long Ntot = (long)p.B * HoWo;
    const int tile = xcd_band_remap(blockIdx.x, gridDim.x);
    const long n0 = ((long)tile * 4 + wave) * (32 * NT);
    if (n0 >= Ntot) return;  // wave-uniform
    const int m0 = blockIdx.y * (32 * MT);
    const int Cin = p.C1 + p.C2;
    const int K = Cin * p.KH * p.KW;
    const int HW = p.H * p.W;

    // ---- per-lane pixel decode (B operand columns)
    bool pv[NT];
    int iy0[NT], ix0[NT];
    const float* base1[NT];
    const float* base2[NT];
    long obase[NT];  // output offset without the channel term
    long rbase[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        long n = n0 + nt * 32 + l31;
        pv[nt] = n < Ntot;
        long nn = pv[nt] ? n : 0;
        int b = (int)(nn / HoWo);
        int r = (int)(nn - (long)b * HoWo);
        int oy = r / p.Wo, ox = r - oy * p.Wo;
        iy0[nt] = oy * p.stride - p.pad_h;
        ix0[nt] = ox * p.stride - p.pad_w;
        long sp = (long)iy0[nt] * p.W + ix0[nt];
        base1[nt] = p.x1 + (long)b * p.C1 * HW + sp;
        base2[nt] = p.x2 ? p.x2 + (long)b * p.C2 * HW + sp : nullptr;
        if (p.scatter2x2)
            obase[nt] = ((long)b * p.out_ctotal + p.out_coff) * (4L * HoWo) + (long)(2 * oy) * (2 * p.Wo) + 2 * ox;
        else
            obase[nt] = ((long)b * p.out_ctotal + p.out_coff) * (long)HoWo + r;
        rbase[nt] = (long)b * p.Cout * HoWo + r;
    }
    const float* wt = p.wt;
    if (p.w_bstride) wt += (n0 / HoWo) * p.w_bstride;  // per-sample weights: tile never straddles samples (host-checked)

    bool mv[MT];
    int mco[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        mco[mt] = m0 + mt * 32 + l31;
        mv[mt] = mco[mt] < p.Cout;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // ---- k cursor of this half-wave for the NEXT load
    int lk = half;  // k index
    int ci = 0, kh = 0, kw = half;
    // normalise (kw may exceed KW-1 when KW == 1, etc.)
    while (kw >= p.KW) { kw -= p.KW; ++kh; }
    while (kh >= p.KH) { kh -= p.KH; ++ci; }

    float a[PF][MT], bq[PF][NT];

    auto load_step = [&](int slot) {
        const bool kv = lk < K;
        const long wrow = (long)lk * p.Cout;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[slot][mt] = (kv && mv[mt]) ? wt[wrow + mco[mt]] : 0.f;
        const int koff = kh * p.W + kw;
        const bool first = ci < p.C1;
        const long coff = first ? (long)ci * HW : (long)(ci - p.C1) * HW;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool ok = kv && pv[nt] && (unsigned)(iy0[nt] + kh) < (unsigned)p.H && (unsigned)(ix0[nt] + kw) < (unsigned)p.W;
            const float* src = (first ? base1[nt] : base2[nt]) + coff + koff;
            bq[slot][nt] = ok ? *src : 0.f;
        }
        // advance by two k
        lk += 2;
        kw += 2;
        if (kw >= p.KW) { kw -= p.KW; ++kh; if (kw >= p.KW) { kw -= p.KW; ++kh; } }
        if (kh >= p.KH) { kh -= p.KH; ++ci; if (kh >= p.KH) { kh -= p.KH; ++ci; } }
    };

#pragma unroll
    for (int s = 0; s < PF; ++s) load_step(s);

    for (int k0 = 0; k0 < K; k0 += 2 * PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][mt], bq[s][nt], acc[mt][nt], 0, 0, 0);
            load_step(s);  // refill this slot for k0 + 2*(PF+s); zeros past K
        }
    }

    // ---- epilogue.  C/D layout: col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co >= p.Cout) continue;
            float bv = 0.f;
            long ochan;
            if (p.scatter2x2) {
                const int cr = co >> 2, dy = (co >> 1) & 1, dx = co & 1;
                if (p.bias) bv = p.bias[cr];
                ochan = (long)cr * (4L * HoWo) + (long)dy * (2 * p.Wo) + dx;
            } else {
                if (p.bias) bv = p.bias[co];
                ochan = (long)co * HoWo;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (!pv[nt]) continue;
                float v = act_apply(p.alpha * acc[mt][nt][r] + bv, p.act);
                if (p.res) v += p.res[rbase[nt] + (long)co * HoWo];
                p.out[obase[nt] + ochan] = v;
            }
        }
    }
}

template <int MT, int NT>
static int launch_t(const ConvParams& p, hipStream_t s) {
    long Ntot = (long)p.B * p.Ho * p.Wo;
    long ntiles = (Ntot + 32 * NT * 4 - 1) / (32 * NT * 4);
    dim3 grid((unsigned)ntiles, (unsigned)((p.Cout + 32 * MT - 1) / (32 * MT)));
    const double flops = 2.0 * (double)Ntot * p.Cout * (p.C1 + p.C2) * p.KH * p.KW;
    if (p.profile_kid >= 0) launch_profiled(p.profile_kid, p.profile_work, conv_igemm_f32_kernel<MT, NT>, grid, dim3(256), s, p);
    else launch_profiled(MT == 1 ? PK_CONV_MT1 : (MT == 2 ? PK_CONV_MT2 : PK_CONV_MT4), flops, conv_igemm_f32_kernel<MT, NT>, grid, dim3(256), s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("conv launch failed: ") + hipGetErrorString(e));
        return CF_ERR_LAUNCH;
    }
    return CF_OK;
}

// Tile choice.  MT=4 halves the input re-reads of wide layers but needs 276 registers (1 wave/SIMD); MT=2 runs at
// 3 waves/SIMD, which hides the global-load latency of this LDS-free design better.  CF_CONV_MT overrides (A/B runs).
static int conv_mt_override() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CF_CONV_MT");
        v = e ? atoi(e) : 0;
    }
    return v;
}

int launch_conv(const ConvParams& p, hipStream_t s) {
    int mt = conv_mt_override();
    if (mt == 0) mt = p.Cout <= 32 ? 1 : 2;
    if (p.Cout <= 32) mt = 1;
    if (mt == 1) return launch_t<1, 2>(p, s);
    if (mt == 4 && p.Cout > 64) return launch_t<4, 2>(p, s);
    return launch_t<2, 2>(p, s);
}

}  // namespace cf

using namespace cf;

extern "C" int cf_conv2d(const float* x1, int C1, const float* x2, int C2, const float* wt, long w_bstride, const float* bias,
                         const float* res, float* out, int out_ctotal, int out_coff, int B, int H, int W, int Cout, int KH,
                         int KW, int stride, int pad_h, int pad_w, int act, float alpha, void* stream) {
    CF_REQUIRE(x1 && wt && out, "null pointer");
    CF_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "bad channel split C1=%d C2=%d", C1, C2);
    CF_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0, "bad shape B=%d H=%d W=%d Cout=%d", B, H, W, Cout);
    CF_REQUIRE(KH >= 1 && KW >= 1 && KH <= 15 && KW <= 15 && stride >= 1 && stride <= 4 && pad_h >= 0 && pad_w >= 0, "bad kernel");
    CF_REQUIRE(out_coff >= 0 && out_coff + Cout <= out_ctotal, "output channel slice out of range");
    CF_REQUIRE(act >= CF_ACT_NONE && act <= CF_ACT_SIGMOID, "bad activation %d", act);
    int Ho = (H + 2 * pad_h - KH) / stride + 1, Wo = (W + 2 * pad_w - KW) / stride + 1;
    CF_REQUIRE(Ho > 0 && Wo > 0, "empty output");
    CF_REQUIRE((long)B * (C1 + C2) * H * W < (1L << 40) && (long)(C1 + C2) * KH * KW * Cout < (1L << 31), "tensor too large");
    if (w_bstride) CF_REQUIRE((Ho * Wo) % 64 == 0, "per-sample weights need Ho*Wo %% 64 == 0 (got %d)", Ho * Wo);
    ConvParams p;
    p.x1 = x1; p.x2 = C2 ? x2 : nullptr; p.wt = wt; p.bias = bias; p.res = res; p.out = out; p.w_bstride = w_bstride;
    p.C1 = C1; p.C2 = C2; p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride;
    p.pad_h = pad_h; p.pad_w = pad_w; p.Ho = Ho; p.Wo = Wo; p.out_ctotal = out_ctotal; p.out_coff = out_coff; p.act = act;
    p.alpha = alpha; p.scatter2x2 = 0; p.gn_ws = nullptr; p.gn_groups = 0;
    return launch_conv(p, as_stream(stream));
}

extern "C" int cf_conv_transpose2d_k2s2(const float* x, const float* w, const float* bias, float* out, int out_ctotal,
                                        int out_coff, int B, int Cin, int H, int W, int Cout, void* stream) {
    CF_REQUIRE(x && w && out, "null pointer");
    CF_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "bad shape");
    CF_REQUIRE(out_coff >= 0 && out_coff + Cout <= out_ctotal, "output channel slice out of range");
    // torch layout [Cin][Cout][2][2] is already Wt[k = ci][m = co*4 + dy*2 + dx]
    ConvParams p;
    p.x1 = x; p.x2 = nullptr; p.wt = w; p.bias = bias; p.res = nullptr; p.out = out; p.w_bstride = 0;
    p.C1 = Cin; p.C2 = 0; p.B = B; p.H = H; p.W = W; p.Cout = Cout * 4; p.KH = 1; p.KW = 1; p.stride = 1;
    p.pad_h = 0; p.pad_w = 0; p.Ho = H; p.Wo = W; p.out_ctotal = out_ctotal; p.out_coff = out_coff; p.act = CF_ACT_NONE;
    p.alpha = 1.f; p.scatter2x2 = 1; p.gn_ws = nullptr; p.gn_groups = 0;
    return launch_conv(p, as_stream(stream));
}
