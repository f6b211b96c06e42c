// 3x3 / stride 1 / pad 1 convolution as a ROW Winograd F(2,3) on the f16 MFMA with the 3-term hi/lo operand split of conv_f16s.hip.
//
// Why: the 128- and 256-channel layers run at the board's power limit with the matrix pipe busy (profiles/r03_power_probe.txt,
// r03_conv_sq_pmc.txt): what moves them is fewer matrix instructions per output, not a tighter schedule.  Along x, two output columns
// (2u, 2u + 1) of one row need the four input columns d0..d3 = x[2u - 1 .. 2u + 2]:
//
//     V = B^T d = (d0 - d2,  d1 + d2,  d2 - d1,  d1 - d3)                (fp32 additions, BEFORE the hi/lo split)
//     U = G g   = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)       (per (co, ci, ky), on the host in fp64, packed like the direct weights)
//     M_j = sum over (ci, ky) of U_j[co][ci][ky] * V_j[ci][row + ky][u]  (four GEMMs, j = position 0..3: the MFMA work)
//     y[2u] = M0 + M1 + M2,   y[2u + 1] = M1 - M2 - M3                   (two adds per output in the accumulator registers)
//
// i.e. 4 x 3 = 12 MFMA k-steps per pair of output columns and 16-channel chunk where the direct form needs 2 x 9 = 18: 1.5x fewer
// MFMAs for the same result.  ky stays a direct sum, so nothing is exchanged between waves and the output transform needs no LDS.
// Numerics (tools/winograd_eval.py --row): max |err| / max|y| 2-8e-7 against fp64 at 128 / 256 channels, the same as the direct
// 3-term form and as an fp32 convolution.
//
// Structure (one workgroup = 4 waves = 128 output channels x NTW x 64 output pixels):
//   * wave w owns m-tile w (32 output channels) and ALL four positions of NTW "unit tiles" (32 column pairs each): 4 x NTW x 16
//     accumulator registers.  NTW = 4: 256 accumulators, one wave per SIMD (512 registers), a weight fragment pair feeds 12 MFMAs
//     and 8 ds_read_b128 feed the same 12 -- exactly the operand traffic per MFMA of conv_f16s' best 128-channel shape;
//     NTW = 2: 128 accumulators, two workgroups per CU.
//   * LDS holds V (not the raw patch): [patch row][position][unit] records of 80 bytes ([hi c0..15 | lo c0..15 | 16 B pad]); the odd
//     multiple of 16 B keeps every ds_read_b128 lane group on 16 distinct slots.  Two buffers, one barrier per chunk.
//   * staging: a lane loads one aligned quad of columns for 4 channels (buffer_load_dwordx4), its two neighbour columns come from the
//     adjacent lanes by DPP row shifts (the tile's outermost columns from one extra dword load of the edge lanes), the transform is
//     7 additions per quad and channel, then the usual split and two ds_write_b64 per (position, unit).
//   * weights: packed on the host in fragment order [m-tile][chunk][ky][position][hi|lo], loaded L2 -> registers two steps ahead.
// Reference layers: DoubleConv / ConvBlocks (nnunet/lib/utils.py:1182-1215), ConvDropoutNormNonlin (generic_UNet.py:26-69), ConvGRUCell
// (convGRU.py:57-66) at >= 128 output channels.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "conv.h"
#include "profile.h"

namespace cf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct WinoGeom {
    int TW, UW, RT, TH, PH, QW;   // tile columns (32 | 16), column pairs ("units") per tile row, rows per 32-unit tile, tile rows, patch rows, aligned quads per tile row
    int LINE, ROWP;               // LDS bytes per (patch row, position) line and per patch row (UW = 8: + 128 B so that rows alternate bank halves)
    int tiles_x, tiles_y;
    int nchunk, c1_pad;
    int ntask;                    // PH * QW * 4 staging tasks (4 channels x 4 columns) per chunk
};
// Timing experiments (clock build, make clock ABL=n -> -DCF_WINO_ABLATE=n; wrong results by design): 1 = staging waves skip transform + LDS
// writes, 2 = MFMA waves skip the weight-fragment loads, 3 = MFMA waves skip the V-record reads, 4 = staging waves skip the global loads.
// Compile-time on purpose: a run-time knob made every load conditional and the compiler's counted waits collapsed to vmcnt(0).

constexpr int W_REC = 80, W_CK = 16, W_NSTEP = 12;
constexpr int PS_NBUF = 4;        // LDS V buffers of the persistent kernel (one barrier per pair of chunks)
CF_CLOCK_DECL(g_clock_wino)
#ifdef CF_CLOCK_STAMPS
// phase totals of the persistent kernel (shader clocks, lane 0 of MFMA wave 0 and of staging wave 4 of every workgroup):
// [0] MFMA: barrier wait, [1] MFMA: chunk compute, [2] MFMA: epilogue + item setup, [3] staging: transform + LDS write, [4] staging: load issue,
// [5] staging: barrier wait, [6] workgroups, [7] chunks; inside [2]: [8] output transform + bias + activation, [9] stores + sums, [10] statistics
__device__ unsigned long long g_wino_phase[12];
#define WINO_T() __builtin_amdgcn_s_memtime()
#define WINO_ADD(i, v) do { if (lane == 0 && (wave == 0 || wave == 4)) atomicAdd(&g_wino_phase[i], (unsigned long long)(v)); } while (0)
#else
#define WINO_T() 0ULL
#define WINO_ADD(i, v) do { } while (0)
#endif

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// value of the lane one below / above inside its row of 16 lanes (DPP row_shr:1 / row_shl:1); the row's first / last lane keeps `own`
__device__ __forceinline__ float from_lane_below(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_lane_above(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x101, 0xf, 0xf, false));
}

// activation over the accumulator registers: ONE workgroup-uniform switch with the element loops inside (with the switch inside the unrolled
// element loop the epilogue was tens of KB of instructions, conv_f16s.hip)
template <int NTW>
__device__ __forceinline__ void wino_activate(f32x16 (&acc)[4][NTW], int act) {
#define CF_WINO_ACT_LOOP(A)                                                     \
    _Pragma("unroll") for (int e = 0; e < 2; ++e)                               \
    _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt)                          \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                            \
        acc[e][nt][r] = act_apply(acc[e][nt][r], A);                            \
        __builtin_amdgcn_sched_barrier(0); /* one element at a time: these cold paths must not set the kernel's register count */ \
    }
    switch (act) {
        case CF_ACT_GELU: CF_WINO_ACT_LOOP(CF_ACT_GELU) break;
        case CF_ACT_RELU: CF_WINO_ACT_LOOP(CF_ACT_RELU) break;
        case CF_ACT_LRELU: CF_WINO_ACT_LOOP(CF_ACT_LRELU) break;
        case CF_ACT_TANH: CF_WINO_ACT_LOOP(CF_ACT_TANH) break;
        case CF_ACT_SIGMOID: CF_WINO_ACT_LOOP(CF_ACT_SIGMOID) break;
        default: break;
    }
#undef CF_WINO_ACT_LOOP
}

template <int NTW, int PRE>
__global__ void __launch_bounds__(256, NTW == 4 ? 1 : 2)
conv_wino_kernel(const ConvParams p, const WinoGeom g, const _Float16* __restrict__ wpk) {
    constexpr int REC = W_REC, CK = W_CK, NSTEP = W_NSTEP;
    constexpr int VT = NTW == 4 ? 2 : 1;          // staging tasks per thread (host: ntask <= 256 * VT)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    CF_CLOCK_BEGIN();

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int mt = blockIdx.y * 4 + wave;

    // ---- tile decode, XCD-banded: the workgroups of one XCD walk a contiguous band of tiles (vertically adjacent tiles share halo rows in its L2)
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x;
        const int xcd = bid & 7, qn = nb >> 3, rn = nb & 7;
        bid = ((xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
    }
    const int tx = bid % g.tiles_x;
    const int t2 = bid / g.tiles_x;
    const int ty = t2 % g.tiles_y;
    const int b = t2 / g.tiles_y;
    const int x0 = tx * g.TW, y0 = ty * g.TH;
    const int HW = p.H * p.W;
    const unsigned HW4 = (unsigned)HW * 4u;
    const int buf_bytes = g.PH * g.ROWP;
    const bool do_stats = p.gn_ws != nullptr;

    // ---- staging tasks: (channel group of 4, patch row, quad q): columns x0 + 4q .. x0 + 4q + 3 of input row y0 - 1 + py
    // per-sample buffer resources (one sample of each input < 2 GiB, host-checked): out-of-range offsets read as 0 = the zero padding
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1 + (long)b * p.C1 * HW), 0, (int)((long)p.C1 * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x2 ? p.x2 + (long)b * p.C2 * HW : p.x1), 0, p.x2 ? (int)((long)p.C2 * HW * 4) : 0, 0x00020000);
    unsigned v_off[VT], e_off[VT], v_c4[VT];
    int v_lds[VT];
    bool v_has[VT], v_first[VT], v_last[VT];
#pragma unroll
    for (int t = 0; t < VT; ++t) {
        const int task = tid + t * 256;
        v_has[t] = task < g.ntask;
        // lane order (q, channel group, patch row): 16 consecutive lanes write 8 units x 2 channel groups -> 2-way LDS bank conflicts on the
        // ds_write_b64 (with the patch row second it was 4-way: a row is 1280 banks = 0 mod 32; SQ_LDS_BANK_CONFLICT was 35 % of the LDS cycles)
        const int q = task % g.QW;
        const int r = task / g.QW;
        const int cg = r & 3, py = r >> 2;
        const int iy = y0 - 1 + py;
        const bool row_ok = v_has[t] && (unsigned)iy < (unsigned)p.H;
        v_first[t] = q == 0;
        v_last[t] = q == g.QW - 1;
        v_off[t] = row_ok ? (unsigned)(iy * p.W + x0 + 4 * q) * 4u : OOB;
        const int ecol = v_first[t] ? x0 - 1 : x0 + g.TW;
        e_off[t] = (row_ok && (v_first[t] || v_last[t]) && (unsigned)ecol < (unsigned)p.W) ? (unsigned)(iy * p.W + ecol) * 4u : OOB;
        v_c4[t] = (unsigned)cg * 4u;
        v_lds[t] = py * g.ROWP + 2 * q * REC + cg * 8;
    }

    auto issue_loads = [&](int chunk, f32x4v (&stg)[VT][4], float (&edg)[VT][4]) {
        const int c0 = chunk * CK;
        const bool in1 = c0 < g.c1_pad;                        // workgroup-uniform
        const unsigned cb = (unsigned)(in1 ? c0 : c0 - g.c1_pad);
        const unsigned clim = (unsigned)(in1 ? p.C1 : p.C2);   // channels at or past it are the zero-weight tail: never fetched
#pragma unroll
        for (int t = 0; t < VT; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ch = cb + v_c4[t] + (unsigned)j;
                const bool ok = ch < clim;
                const unsigned off = ok ? v_off[t] + ch * HW4 : OOB;
                const unsigned eo = ok ? e_off[t] + ch * HW4 : OOB;
                if (in1) {
                    stg[t][j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc1, off, 0, 0));
                    edg[t][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc1, eo, 0, 0));
                } else {
                    stg[t][j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc2, off, 0, 0));
                    edg[t][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc2, eo, 0, 0));
                }
            }
        }
    };

    // PRE: x1 is a raw convolution output; its GroupNorm / InstanceNorm + activation is applied between the load and the transform, from
    // the {mean, scale, shift} table of this sample staged once in LDS behind the two V buffers.  Padding stays exactly zero.
    const float* ctab = reinterpret_cast<const float*>(lds + 2 * buf_bytes);
    const int ctab_n = g.nchunk * CK;
    auto pre_apply = [&](float v, float cm, float ca, float cs, bool valid) -> float {
        v = (v - cm) * ca + cs;
        v = p.in_slope < 0.f ? gelu_as(v) : (v > 0.f ? v : v * p.in_slope);
        return valid ? v : 0.f;
    };
    auto write_stage = [&](int chunk, const f32x4v (&stg)[VT][4], const float (&edg)[VT][4]) {
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            unsigned char* base = lds + (chunk & 1) * buf_bytes + v_lds[t];
            f32x4v cm = {0.f, 0.f, 0.f, 0.f}, ca = cm, cs = cm;
            if (PRE) {
                const int c0 = chunk * CK + (int)v_c4[t];
                cm = *reinterpret_cast<const f32x4v*>(ctab + c0);
                ca = *reinterpret_cast<const f32x4v*>(ctab + ctab_n + c0);
                cs = *reinterpret_cast<const f32x4v*>(ctab + 2 * ctab_n + c0);
            }
            const bool valid = v_off[t] != OOB, evalid = e_off[t] != OOB;
            f16x4 hi[2][4], lo[2][4];      // [unit 2q | 2q + 1][position] over the 4 channels
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float c0 = stg[t][j][0], c1 = stg[t][j][1], c2 = stg[t][j][2], c3 = stg[t][j][3], e = edg[t][j];
                if (PRE) {
                    c0 = pre_apply(c0, cm[j], ca[j], cs[j], valid);
                    c1 = pre_apply(c1, cm[j], ca[j], cs[j], valid);
                    c2 = pre_apply(c2, cm[j], ca[j], cs[j], valid);
                    c3 = pre_apply(c3, cm[j], ca[j], cs[j], valid);
                    e = pre_apply(e, cm[j], ca[j], cs[j], evalid);
                }
                // every lane takes part in the shifts (uniform control flow): quads of one patch row sit in consecutive lanes of a DPP row
                float left = from_lane_below(c3), right = from_lane_above(c0);
                left = v_first[t] ? e : left;
                right = v_last[t] ? e : right;
                const float va[4] = {left - c1, c0 + c1, c1 - c0, c0 - c2};      // unit 2q:     d = (left, c0, c1, c2)
                const float vb[4] = {c1 - c3, c2 + c3, c3 - c2, c2 - right};     // unit 2q + 1: d = (c1, c2, c3, right)
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    _Float16 h, l;
                    split_f16(va[pos], h, l);
                    hi[0][pos][j] = h;
                    lo[0][pos][j] = l;
                    split_f16(vb[pos], h, l);
                    hi[1][pos][j] = h;
                    lo[1][pos][j] = l;
                }
            }
            if (v_has[t]) {
#pragma unroll
                for (int pos = 0; pos < 4; ++pos)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        *reinterpret_cast<f16x4*>(base + pos * g.LINE + u * REC) = hi[u][pos];
                        *reinterpret_cast<f16x4*>(base + pos * g.LINE + u * REC + CK * 2) = lo[u][pos];
                    }
            }
        }
    };

    f32x4v stg[VT][4];
    float edg[VT][4];
    issue_loads(0, stg, edg);

    // ---- per-lane B-fragment offsets: unit tile nt, lane l31 -> (tile row nt * RT + l31 / UW, unit l31 % UW)
    int b_off[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) b_off[nt] = (nt * g.RT + l31 / g.UW) * g.ROWP + (l31 % g.UW) * REC + half * 16;

    f32x16 acc[4][NTW];
#pragma unroll
    for (int pos = 0; pos < 4; ++pos)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pos][nt][r] = 0.f;

    // packed weights: fragment (mt, chunk, step = ky * 4 + position, part) = 64 lanes x 8 halves
    const f16x8* wfrag = reinterpret_cast<const f16x8*>(wpk) + (long)mt * g.nchunk * (NSTEP * 2) * 64 + lane;
    // fragment register ring, prefetched D steps ahead (NSTEP % R == 0 keeps the slots static).  NTW = 2: a step is only 6 MFMAs (~190
    // clocks), so three steps ahead to cover an L2 round trip under load; NTW = 4: 12 MFMAs per step, two steps
    constexpr int R = NTW == 2 ? 4 : 3, D = R - 1;
    f16x8 aH[R] = {}, aL[R] = {};
    auto load_a = [&](int chunk, int step, int slot) {
        const f16x8* wc = wfrag + ((long)chunk * NSTEP + step) * 2 * 64;
        aH[slot] = wc[0];
        aL[slot] = wc[64];
    };
#pragma unroll
    for (int s = 0; s < D; ++s) load_a(0, s, s % R);
    if (PRE) {
        float* ct = reinterpret_cast<float*>(lds + 2 * buf_bytes);
        for (int k = tid; k < 3 * ctab_n; k += 256) {
            const int which = k / ctab_n, c = k - which * ctab_n;
            ct[k] = c < p.C1 ? p.in_norm[((long)b * 3 + which) * p.C1 + c] : 0.f;
        }
        __syncthreads();
    }
    write_stage(0, stg, edg);
    __syncthreads();

    // B fragments (V records) one step ahead in a second register set (NTW = 2: 16 more registers; with two waves per SIMD a read issued
    // right in front of its MFMA left the pipe idle for the LDS latency twice per step).  NTW = 4 has no registers left for it.
    constexpr int BP = NTW == 2 ? 1 : 0;
    f16x8 bH[BP + 1][NTW], bL[BP + 1][NTW];
    auto load_b = [&](const unsigned char* xb, int step, int set) {
        const int toff = (step >> 2) * g.ROWP + (step & 3) * g.LINE;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const unsigned char* rp = xb + b_off[nt] + toff;
            bH[set][nt] = *reinterpret_cast<const f16x8*>(rp);
            bL[set][nt] = *reinterpret_cast<const f16x8*>(rp + CK * 2);
        }
    };

    for (int c = 0; c < g.nchunk; ++c) {
        const bool more = c + 1 < g.nchunk;
        const int cn = more ? c + 1 : c;
        const unsigned char* xb = lds + (c & 1) * buf_bytes;
        if (BP) load_b(xb, 0, 0);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            // every load is unconditional (the last chunk re-reads itself) and pinned here, so the waits come out counted (conv_f16s.hip)
            if (step + D < NSTEP) load_a(c, step + D, (step + D) % R);
            else load_a(cn, step + D - NSTEP, (step + D) % R);
            if (step == 0) issue_loads(cn, stg, edg);
            if (BP) { if (step + 1 < NSTEP) load_b(xb, step + 1, (step + 1) & 1); }
            else load_b(xb, step, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int pos = step & 3;       // compile-time after unrolling
            const int set = BP ? (step & 1) : 0;
            const f16x8 ah = aH[step % R], al = aL[step % R];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                acc[pos][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bH[set][nt], acc[pos][nt], 0, 0, 0);
                acc[pos][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bL[set][nt], acc[pos][nt], 0, 0, 0);
                acc[pos][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bH[set][nt], acc[pos][nt], 0, 0, 0);
            }
        }
        if (more) write_stage(c + 1, stg, edg);
        __syncthreads();
    }

    // ---- output transform in the accumulator registers: acc[0] <- y[2u], acc[1] <- y[2u + 1]
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
            acc[0][nt][r] = (m0 + m1) + m2;
            acc[1][nt][r] = (m1 - m2) - m3;
        }

    // ---- epilogue: value = act(alpha * y + bias) (+ res); optional fused GroupNorm statistics of the stored values (conv_f16s.hip)
    float ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[r] = 0.f; ssq[r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float bv = 0.f;
        if (p.bias && co < p.Cout) bv = p.bias[co];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            acc[0][nt][r] = p.alpha * acc[0][nt][r] + bv;
            acc[1][nt][r] = p.alpha * acc[1][nt][r] + bv;
        }
    }
    wino_activate<NTW>(acc, p.act);      // workgroup-uniform; the ConvGRU gates (sigmoid / tanh) are the main users on this path
    constexpr unsigned OOB_CH = 0x40000000u;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        p.out + ((long)b * p.out_ctotal + p.out_coff) * (long)HW, 0, (int)((long)(p.out_ctotal - p.out_coff) * HW * 4), 0x00020000);
    unsigned o_off[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        int li = l31;
        asm volatile("" : "+v"(li));                       // keep the address math below the main loop
        const int oy = y0 + nt * g.RT + li / g.UW, ox = x0 + 2 * (li % g.UW);
        o_off[nt] = (unsigned)(oy * p.W + ox) * 4u;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool co_ok = co < p.Cout;
        const unsigned ochan = co_ok ? (unsigned)co * HW4 : OOB_CH;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            float v0 = acc[0][nt][r], v1 = acc[1][nt][r];
            if (p.res) {                                    // rare (SingleConv residuals)
                if (co_ok) {
                    const float* rp = p.res + ((long)b * p.Cout + co) * HW + (o_off[nt] >> 2);
                    v0 += rp[0];
                    v1 += rp[1];
                }
            }
            u32x2 pk;
            pk[0] = __builtin_bit_cast(unsigned, v0);
            pk[1] = __builtin_bit_cast(unsigned, v1);
            __builtin_amdgcn_raw_buffer_store_b64(pk, rs_out, o_off[nt] + ochan, 0, 0);
            const float m0 = co_ok ? v0 : 0.f, m1 = co_ok ? v1 : 0.f;
            ssum[r] += m0 + m1;
            ssq[r] += m0 * m0 + m1 * m1;
        }
    }
    if (do_stats) {
        xreduce16(ssum, lane);      // common.h: DPP + v_permlane16_swap, no LDS traffic
        xreduce16(ssq, lane);
        // workgroup-level combine in LDS (the V buffers are free now), then ONE fp64 atomic pair per (group, workgroup)
        float* red = reinterpret_cast<float*>(lds);  // [128 channels][2]
        __syncthreads();
        red[tid] = 0.f;
        __syncthreads();
        if ((lane & 1) == 0) {
            const int r = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            const int cl = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            atomicAdd(&red[2 * cl], ssum[0]);
            atomicAdd(&red[2 * cl + 1], ssq[0]);
        }
        __syncthreads();
        if (tid < 128) {
            const int co = blockIdx.y * 128 + tid;
            const int cpg = p.Cout / p.gn_groups;
            if (co < p.Cout && (tid == 0 || co % cpg == 0)) {
                int n = cpg - co % cpg;               // channels of this group from co on
                if (n > 128 - tid) n = 128 - tid;
                if (n > p.Cout - co) n = p.Cout - co;
                float s1 = 0.f, s2 = 0.f;
                for (int j = 0; j < n; ++j) { s1 += red[2 * (tid + j)]; s2 += red[2 * (tid + j) + 1]; }
                double* w = p.gn_ws + 2L * ((long)b * p.gn_groups + co / cpg);
                atomicAdd(w, (double)s1);
                atomicAdd(w + 1, (double)s2);
            }
        }
    }
    CF_CLOCK_END(g_clock_wino);
}

// =====================================================================================================================================
// Persistent, wave-specialised form (the default where a layer has enough tiles): one workgroup per CU = 4 MFMA waves (one per SIMD,
// 128 accumulators each: m-tile w x 4 positions x 2 unit tiles) + 4 staging waves (one per SIMD), walking a band of (128-channel block,
// tile) items of its XCD.  Why: in the kernel above the transform / split / LDS-write stream (4.8 vector instructions per MFMA,
// SQ_INSTS_VALU / SQ_INSTS_MFMA in profiles/r04_pmc_conv_wino.txt) sits in the SAME waves as the MFMAs and the matrix pipe is busy 53 %
// of the wave cycles where the direct kernel reaches 84 %.  Here
//   * the MFMA waves issue nothing but weight-fragment loads (three steps ahead), V-record reads (one step ahead) and MFMAs, one barrier
//     per 16-channel chunk, and run the item's epilogue (output transform, bias, activation, 32 eight-byte stores per lane, statistics by
//     shuffles + one fp64 atomic pair per group -- no LDS, no barrier);
//   * the staging waves run the chunk stream ACROSS item boundaries, two chunks of loads in flight (two register sets), up to two LDS buffers
//     ahead of the MFMA waves (four V buffers);
//   * ONE barrier per PAIR of chunks: barrier j = "chunks 2j, 2j + 1 are in LDS" for the MFMA waves and "chunks 2j - 2, 2j - 1 have been read"
//     for the staging waves (which then overwrite those buffers with chunks 2j + 2, 2j + 3); both roles execute ceil(N / 2) barriers.  (With a
//     barrier per chunk and two buffers the MFMA waves waited 500-700 clocks per chunk; with pairs they wait 170-270 -- and compute 300 clocks
//     longer per chunk, because the staging wave of their SIMD is now busy beside them: the layer times moved by 0 ... +4 %.  The CU, and on data
//     the board's power limit, are what is shared; the synchronisation was not the cost.)
// Where its clocks go (make clock + tools/wino_phases.py, profiles/r04_wino_phases.txt), per 16-channel chunk of 72 MFMAs = 2 304 clocks:
// MFMA wave 3 270 in the chunk loop (2 560 without the weight-fragment loads, make clock ABL=2) + 1 110 of epilogue at 8 chunks per item
// (the stores are issue-bound: ~8 700 clocks per item) + 300-500 at the barrier; staging wave ~2 800 for ~200 vector instructions beside
// the MFMA wave of its SIMD (4 600-5 100 with the GELU of a deferred normalisation, which then sets the pace).  Two variants were built
// and measured against this one and lost: the epilogue handed to the staging waves through an LDS buffer (their stores sit in the same
// in-order vmcnt queue as the staged loads), and a 4 + 2 + 2 split with dedicated epilogue waves (two staging waves with two tasks each
// get ~6 vector-instruction slots per MFMA of their SIMD and fall behind): -8 ... -25 %.
// Wave priorities (s_setprio 3 on one role, whole kernel or epilogue only): staging waves high = -8 % on plain layers, +<= 1 % with a deferred
// normalisation; MFMA waves high = no change.  The epilogue (~8 000 clocks per item: output transform 1 500, stores + sums 3 200, statistics
// 2 200) is ~800 vector instructions on a SIMD shared with a staging wave; dwordx4 instead of dwordx2 stores did not move it.
// Deferred normalisation: the {mean, scale, shift} quads come straight from global memory with the chunk's loads (no LDS table: items of a
// band cross samples).
struct StageRegs {
    f32x4v stg[4];
    float edg[4];
    f32x4v cm, ca, cs;
    bool valid, evalid;
};

template <int PRE>
__global__ void __launch_bounds__(512, 2)
conv_wino_ps_kernel(const ConvParams p, const WinoGeom g, const _Float16* __restrict__ wpk) {
    constexpr int NTW = 2, REC = W_REC, CK = W_CK, NSTEP = W_NSTEP;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    CF_CLOCK_BEGIN();

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    // ---- this workgroup's items: band of its XCD (blocks b and b + 8 share an XCD: speed only), interleaved with the XCD's other workgroups
    const int xcd = blockIdx.x & 7, wgi = blockIdx.x >> 3, nwgx = gridDim.x >> 3;
    const int ntiles = g.tiles_x * g.tiles_y * p.B;
    const int nitems = ntiles * ((p.Cout + 127) / 128);
    const int band_lo = (int)((long)nitems * xcd / 8), band_hi = (int)((long)nitems * (xcd + 1) / 8);
    if (band_lo + wgi >= band_hi) return;
    const int n_my = (band_hi - band_lo - wgi + nwgx - 1) / nwgx;
    auto decode = [&](int i, int& cb, int& b, int& y0, int& x0) {
        const int item = band_lo + wgi + i * nwgx;
        cb = item / ntiles;
        int t = item - cb * ntiles;
        const int tx = t % g.tiles_x;
        t /= g.tiles_x;
        const int ty = t % g.tiles_y;
        b = t / g.tiles_y;
        y0 = ty * g.TH;
        x0 = tx * g.TW;
    };
    const int HW = p.H * p.W;
    const unsigned HW4 = (unsigned)HW * 4u;
    const int buf_bytes = g.PH * g.ROWP;
    const int N = n_my * g.nchunk;                        // chunks of this workgroup's stream

    if (wave >= 4) {
        // ============================================================================================================ staging waves (4 .. 7)
        constexpr int VT = 1;                             // tasks per thread: 256 threads cover the <= 256 (row, quad, channel group) tasks of a chunk
        bool has[VT], first[VT], last[VT];
        unsigned c4[VT];
        int v_lds[VT], t_py[VT], t_q[VT];
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            const int task = (tid - 256) + t * 256;
            has[t] = task < g.ntask;
            const int q = task % g.QW;                    // lane order (q, channel group, patch row): see conv_wino_kernel
            const int r = task / g.QW;
            const int cg = r & 3, py = r >> 2;
            first[t] = q == 0;
            last[t] = q == g.QW - 1;
            c4[t] = (unsigned)cg * 4u;
            v_lds[t] = py * g.ROWP + 2 * q * REC + cg * 8;
            t_py[t] = py;
            t_q[t] = q;
        }

        int ld_i = 0, ld_c = 0, ld_b = 0;                 // the next chunk to load: item, chunk, its sample
        unsigned v_off[VT], e_off[VT];
        __amdgpu_buffer_rsrc_t rsrc1, rsrc2;
        auto setup = [&](int i) __attribute__((always_inline)) {
            int cb, b, y0, x0;
            decode(i, cb, b, y0, x0);
#pragma unroll
            for (int t = 0; t < VT; ++t) {
                const int iy = y0 - 1 + t_py[t];
                const bool row_ok = has[t] && (unsigned)iy < (unsigned)p.H;
                v_off[t] = row_ok ? (unsigned)(iy * p.W + x0 + 4 * t_q[t]) * 4u + c4[t] * HW4 : OOB;
                const int ecol = first[t] ? x0 - 1 : x0 + g.TW;
                e_off[t] = (row_ok && (first[t] || last[t]) && (unsigned)ecol < (unsigned)p.W) ? (unsigned)(iy * p.W + ecol) * 4u + c4[t] * HW4 : OOB;
            }
            // per-sample resources: offsets at or past C * HW * 4 (the zero-weight channel tail of a chunk, parked lanes) read as 0
            rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1 + (long)b * p.C1 * HW), 0, (int)((long)p.C1 * HW * 4), 0x00020000);
            rsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x2 ? p.x2 + (long)b * p.C2 * HW : p.x1), 0,
                                                      p.x2 ? (int)((long)p.C2 * HW * 4) : 0, 0x00020000);
            ld_b = b;
        };
        auto issue = [&](StageRegs (&R)[VT]) __attribute__((always_inline)) {   // loads chunk (ld_i, ld_c) and advances the cursor (saturating: every call loads)
#if defined(CF_WINO_ABLATE) && CF_WINO_ABLATE == 4
            return;
#endif
            const int c0 = ld_c * CK;
            const bool in1 = c0 < g.c1_pad;
            const unsigned cb = (unsigned)(in1 ? c0 : c0 - g.c1_pad);
#pragma unroll
            for (int t = 0; t < VT; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned ch = (cb + (unsigned)j) * HW4;               // scalar
                    if (in1) {
                        R[t].stg[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc1, v_off[t] + ch, 0, 0));
                        R[t].edg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc1, e_off[t] + ch, 0, 0));
                    } else {
                        R[t].stg[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc2, v_off[t] + ch, 0, 0));
                        R[t].edg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc2, e_off[t] + ch, 0, 0));
                    }
                }
                if (PRE) {
                    const int c = c0 + (int)c4[t];                               // C1 % 4 == 0 (host): a quad is wholly inside or wholly the padded tail
                    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_norm + (long)ld_b * 3 * p.C1), 0,
                                                                                        3 * p.C1 * 4, 0x00020000);
                    const unsigned o = c < p.C1 ? (unsigned)c * 4u : OOB;
                    R[t].cm = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rn, o, 0, 0));
                    R[t].ca = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rn, o + (unsigned)p.C1 * 4u, 0, 0));
                    R[t].cs = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rn, o + (unsigned)p.C1 * 8u, 0, 0));
                }
                R[t].valid = v_off[t] < OOB;
                R[t].evalid = e_off[t] < OOB;
            }
            if (ld_c + 1 < g.nchunk) ++ld_c;
            else if (ld_i + 1 < n_my) { ld_c = 0; ++ld_i; setup(ld_i); }
        };
        auto pre_apply = [&](float v, float cm, float ca, float cs, bool valid) -> float {
            v = (v - cm) * ca + cs;
            v = p.in_slope < 0.f ? gelu_as(v) : (v > 0.f ? v : v * p.in_slope);
            return valid ? v : 0.f;
        };
        auto write = [&](const StageRegs (&RR)[VT], int gc) __attribute__((always_inline)) {
#if defined(CF_WINO_ABLATE) && CF_WINO_ABLATE == 1
            return;
#endif
#pragma unroll
            for (int t = 0; t < VT; ++t) {
                const StageRegs& R = RR[t];
                unsigned char* base = lds + (gc & (PS_NBUF - 1)) * buf_bytes + v_lds[t];
                f16x4 hi[2][4], lo[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float c0 = R.stg[j][0], c1 = R.stg[j][1], c2 = R.stg[j][2], c3 = R.stg[j][3], e = R.edg[j];
                    if (PRE) {
                        c0 = pre_apply(c0, R.cm[j], R.ca[j], R.cs[j], R.valid);
                        c1 = pre_apply(c1, R.cm[j], R.ca[j], R.cs[j], R.valid);
                        c2 = pre_apply(c2, R.cm[j], R.ca[j], R.cs[j], R.valid);
                        c3 = pre_apply(c3, R.cm[j], R.ca[j], R.cs[j], R.valid);
                        e = pre_apply(e, R.cm[j], R.ca[j], R.cs[j], R.evalid);
                    }
                    float left = from_lane_below(c3), right = from_lane_above(c0);
                    left = first[t] ? e : left;
                    right = last[t] ? e : right;
                    const float va[4] = {left - c1, c0 + c1, c1 - c0, c0 - c2};
                    const float vb[4] = {c1 - c3, c2 + c3, c3 - c2, c2 - right};
#pragma unroll
                    for (int pos = 0; pos < 4; ++pos) {
                        _Float16 h, l;
                        split_f16(va[pos], h, l);
                        hi[0][pos][j] = h;
                        lo[0][pos][j] = l;
                        split_f16(vb[pos], h, l);
                        hi[1][pos][j] = h;
                        lo[1][pos][j] = l;
                    }
                }
                if (has[t]) {
#pragma unroll
                    for (int pos = 0; pos < 4; ++pos)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            *reinterpret_cast<f16x4*>(base + pos * g.LINE + u * REC) = hi[u][pos];
                            *reinterpret_cast<f16x4*>(base + pos * g.LINE + u * REC + CK * 2) = lo[u][pos];
                        }
                }
            }
        };
        StageRegs A[VT], B[VT];
        setup(0);
        issue(A);
        issue(B);
        unsigned long long tw = 0, ti = 0, tb = 0;
        (void)tw; (void)ti; (void)tb;
        // Both halves of the body are UNCONDITIONAL (an odd stream is padded by one chunk: the cursor saturates and the extra chunk lands in a
        // buffer nobody reads any more).  With the second half under `if (gc + 1 < N)` the loop header had a predecessor on which set B had not
        // been re-issued, and the compiler's wait for set A there came out as vmcnt(0).
        // ONE barrier per PAIR of chunks, four LDS buffers: barrier j says "chunks 2j, 2j + 1 are in LDS" to the MFMA waves and "chunks 2j - 2,
        // 2j - 1 have been read" to these waves, which then overwrite those two buffers with chunks 2j + 2, 2j + 3.  With a barrier per chunk and
        // two buffers the roles ran in lockstep one chunk apart: the MFMA waves paid every chunk's jitter at a barrier (500-700 clocks per
        // chunk) and these waves could get only one chunk ahead during an item's epilogue (profiles/r04_wino_phases.txt).
        for (int gc = 0; gc < N; gc += 2) {
            unsigned long long t0 = WINO_T();
            write(A, gc);
            unsigned long long t1 = WINO_T();
            issue(A);                                     // chunk gc + 2 (or the stream's last chunk again)
            unsigned long long t2 = WINO_T();
            tw += t1 - t0; ti += t2 - t1;
            t0 = WINO_T();
            write(B, gc + 1);
            t1 = WINO_T();
            issue(B);
            t2 = WINO_T();
            __syncthreads();
            const unsigned long long t3 = WINO_T();
            tw += t1 - t0; ti += t2 - t1; tb += t3 - t2;
        }
        WINO_ADD(3, tw); WINO_ADD(4, ti); WINO_ADD(5, tb);
        return;
    }

    // =================================================================================================================== MFMA waves
    int b_off[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) b_off[nt] = (nt * g.RT + l31 / g.UW) * g.ROWP + (l31 % g.UW) * REC + half * 16;
    const long mt_stride = (long)g.nchunk * (NSTEP * 2) * 64;             // f16x8 units per m-tile
    const f16x8* const wbase = reinterpret_cast<const f16x8*>(wpk) + lane;
    // weight-fragment ring, three steps ahead (a six-slot ring -- five steps, ~960 MFMA clocks -- took the chunk from 3 270 to 2 890 clocks
    // in the variant whose epilogue ran in other waves, profiles/r04_wino_phases.txt, but does not fit beside this role's epilogue registers)
    constexpr int R = 4, D = R - 1;
    f16x8 aH[R] = {}, aL[R] = {};
    auto load_a = [&](const f16x8* wc, int slot) {
#if defined(CF_WINO_ABLATE) && CF_WINO_ABLATE == 2
        return;
#endif
        aH[slot] = wc[0];
        aL[slot] = wc[64];
    };
    f16x8 bH[2][NTW] = {}, bL[2][NTW] = {};
    auto load_b = [&](const unsigned char* xb, int step, int set) {
#if defined(CF_WINO_ABLATE) && CF_WINO_ABLATE == 3
        return;
#endif
        const int toff = (step >> 2) * g.ROWP + (step & 3) * g.LINE;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const unsigned char* rp = xb + b_off[nt] + toff;
            bH[set][nt] = *reinterpret_cast<const f16x8*>(rp);
            bL[set][nt] = *reinterpret_cast<const f16x8*>(rp + CK * 2);
        }
    };
    f32x16 acc[4][NTW];
    int cb, b, y0, x0;
    decode(0, cb, b, y0, x0);
    const f16x8* wf = wbase + (long)(cb * 4 + wave) * mt_stride;
#pragma unroll
    for (int s = 0; s < D; ++s) load_a(wf + (long)s * 2 * 64, s % R);

    unsigned long long tm_wait = 0, tm_comp = 0, tm_epi = 0, tm_e1 = 0, tm_e2 = 0, tm_e3 = 0, tm_mark = WINO_T();
    (void)tm_wait; (void)tm_comp; (void)tm_epi; (void)tm_mark;
    int cb_n = cb, b_n = b, y0_n = y0, x0_n = x0;          // the item after the current one (decoded once per item)
    for (int i = 0; i < n_my; ++i) {
        cb = cb_n; b = b_n; y0 = y0_n; x0 = x0_n;
        const int mt = cb * 4 + wave;
        wf = wbase + (long)mt * mt_stride;
        if (i + 1 < n_my) decode(i + 1, cb_n, b_n, y0_n, x0_n);
        const f16x8* wf_next = wbase + (long)(cb_n * 4 + wave) * mt_stride;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pos][nt][r] = 0.f;

        for (int c = 0; c < g.nchunk; ++c) {
            const int gc = i * g.nchunk + c;
            const f16x8* wc = wf + (long)c * NSTEP * 2 * 64;
            const f16x8* wn = (c + 1 < g.nchunk) ? wc + NSTEP * 2 * 64 : wf_next;      // the next chunk of the stream (the last one re-reads itself)
            {
                const unsigned long long ta = WINO_T();
                if (c == 0) tm_epi += ta - tm_mark; else tm_comp += ta - tm_mark;
                if ((gc & 1) == 0) __syncthreads();       // chunks gc and gc + 1 are in LDS (one barrier per pair: see the staging loop)
                tm_mark = WINO_T();
                tm_wait += tm_mark - ta;
            }
            const unsigned char* xb = lds + (gc & (PS_NBUF - 1)) * buf_bytes;
            load_b(xb, 0, 0);
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {
                if (step + D < NSTEP) load_a(wc + (long)(step + D) * 2 * 64, (step + D) % R);
                else load_a(wn + (long)(step + D - NSTEP) * 2 * 64, (step + D) % R);
                if (step + 1 < NSTEP) load_b(xb, step + 1, (step + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                const int pos = step & 3;
                const int set = step & 1;
                const f16x8 ah = aH[step % R], al = aL[step % R];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    acc[pos][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bH[set][nt], acc[pos][nt], 0, 0, 0);
                    acc[pos][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bL[set][nt], acc[pos][nt], 0, 0, 0);
                    acc[pos][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bH[set][nt], acc[pos][nt], 0, 0, 0);
                }
            }
        }

        {
            const unsigned long long ta = WINO_T();
            tm_comp += ta - tm_mark;
            tm_mark = ta;
        }
        // ---- epilogue of item i (the staging waves are already one chunk into the next item)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
                acc[0][nt][r] = (m0 + m1) + m2;
                acc[1][nt][r] = (m1 - m2) - m3;
            }
        float ssum[16], ssq[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ssum[r] = 0.f;
            ssq[r] = 0.f;
            const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            // branch-free (a clamped index and a select; sixteen `if (co < Cout)` loads were sixteen exec-mask branches per item)
            const float bv = p.bias ? p.bias[co < p.Cout ? co : p.Cout - 1] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                acc[0][nt][r] = p.alpha * acc[0][nt][r] + bv;
                acc[1][nt][r] = p.alpha * acc[1][nt][r] + bv;
            }
        }
        wino_activate<NTW>(acc, p.act);
        const unsigned long long te1 = WINO_T();
        constexpr unsigned OOB_CH = 0x40000000u;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
            p.out + ((long)b * p.out_ctotal + p.out_coff) * (long)HW, 0, (int)((long)(p.out_ctotal - p.out_coff) * HW * 4), 0x00020000);
        unsigned o_off[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int oy = y0 + nt * g.RT + l31 / g.UW, ox = x0 + 2 * (l31 % g.UW);
            o_off[nt] = (unsigned)(oy * p.W + ox) * 4u;
        }
        // Stores: a lane holds two adjacent columns of unit u for both unit tiles (nt = 0, 1).  Adjacent lanes (units u, u + 1 of one row: UW is
        // even) swap one pair through a DPP quad permute, after which the even lane owns four adjacent columns of tile 0 and the odd lane four of
        // tile 1: 16 sixteen-byte stores per lane instead of 32 eight-byte ones (the store path moves ~14 B per clock and CU with dwordx4, half
        // that with dwordx2, and the item's stores were its epilogue: profiles/r04_wino_phases.txt).
        const bool odd = (lane & 1) != 0;
        const unsigned st_off = odd ? o_off[1] - 8u : o_off[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool co_ok = co < p.Cout;
            const unsigned ochan = co_ok ? (unsigned)co * HW4 : OOB_CH;
            float v[NTW][2];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                float v0 = acc[0][nt][r], v1 = acc[1][nt][r];
                if (p.res) {
                    if (co_ok) {
                        const float* rp = p.res + ((long)b * p.Cout + co) * HW + (o_off[nt] >> 2);
                        v0 += rp[0];
                        v1 += rp[1];
                    }
                }
                v[nt][0] = v0;
                v[nt][1] = v1;
                const float m0 = co_ok ? v0 : 0.f, m1 = co_ok ? v1 : 0.f;
                ssum[r] += m0 + m1;
                ssq[r] += m0 * m0 + m1 * m1;
            }
            const float k0 = odd ? v[1][0] : v[0][0], k1 = odd ? v[1][1] : v[0][1];      // kept: the own tile's pair
            const int s0 = __builtin_bit_cast(int, odd ? v[0][0] : v[1][0]), s1 = __builtin_bit_cast(int, odd ? v[0][1] : v[1][1]);
            const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(s0, s0, 0xB1, 0xf, 0xf, false));   // quad_perm [1, 0, 3, 2]
            const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(s1, s1, 0xB1, 0xf, 0xf, false));
            u32x4 pk;
            pk[0] = __builtin_bit_cast(unsigned, odd ? r0 : k0);
            pk[1] = __builtin_bit_cast(unsigned, odd ? r1 : k1);
            pk[2] = __builtin_bit_cast(unsigned, odd ? k0 : r0);
            pk[3] = __builtin_bit_cast(unsigned, odd ? k1 : r1);
            __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, st_off + ochan, 0, 0);
        }
        const unsigned long long te2 = WINO_T();
        tm_e1 += te1 - tm_mark;
        tm_e2 += te2 - te1;
        if (p.gn_ws) {
            // per-channel totals by the transpose-reduce of conv_f16s.hip: afterwards the lane with bits (b5 b4 b3 b2 b1 x) holds channel
            // cl = b1 + 2 b2 + 4 b5 + 8 b3 + 16 b4 of this wave's 32; a group of cpg = 2^k consecutive channels is then summed by k more
            // xor-shuffles over the lane bits of cl's low k bits, and one lane per group adds the pair to the fp64 workspace.
            xreduce16(ssum, lane);      // common.h: DPP + v_permlane16_swap, no LDS traffic
            xreduce16(ssq, lane);
            float s1 = ssum[0], s2 = ssq[0];
            const int cpg = p.Cout / p.gn_groups;
            const int cl = ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + 4 * half + 8 * ((lane >> 3) & 1) + 16 * ((lane >> 4) & 1);
            int k = 0;
            const bool pow2 = (cpg & (cpg - 1)) == 0;
            // lane masks of cl's bits 0..4: 2, 4, 32, 8, 16 (wave-uniform conditions; the mask-4 level needs the mask-2 level before it: common.h)
            if (pow2 && cpg > 1) { s1 = xor_sum<2>(s1); s2 = xor_sum<2>(s2); k = 1; }
            if (pow2 && cpg > 2) { s1 = xor_sum<4>(s1); s2 = xor_sum<4>(s2); k = 2; }
            if (pow2 && cpg > 4) { s1 = xor_sum<32>(s1); s2 = xor_sum<32>(s2); k = 3; }
            if (pow2 && cpg > 8) { s1 = xor_sum<8>(s1); s2 = xor_sum<8>(s2); k = 4; }
            if (pow2 && cpg > 16) { s1 = xor_sum<16>(s1); s2 = xor_sum<16>(s2); k = 5; }
            const int span = pow2 ? (1 << k) : 1;                            // channels whose total this lane now holds
            const int co = mt * 32 + cl;
            if ((lane & 1) == 0 && (cl & (span - 1)) == 0 && co < p.Cout) {
                double* w = p.gn_ws + 2L * ((long)b * p.gn_groups + co / cpg);
                atomicAdd(w, (double)s1);
                atomicAdd(w + 1, (double)s2);
            }
        }
        tm_e3 += WINO_T() - te2;
    }
    tm_epi += WINO_T() - tm_mark;
    WINO_ADD(0, tm_wait); WINO_ADD(1, tm_comp); WINO_ADD(2, tm_epi); WINO_ADD(6, 1); WINO_ADD(7, N);
    WINO_ADD(8, tm_e1); WINO_ADD(9, tm_e2); WINO_ADD(10, tm_e3);
    CF_CLOCK_END(g_clock_wino);
}

// route level: 0 = off (every layer stays on the direct kernels), 1 = automatic (the persistent wave-specialised kernel where a layer has at
// least two items per CU, else the one-tile-per-workgroup kernel with 2 or 4 unit tiles per wave), 2 / 4 = force the one-tile kernel with
// NTW = 2 / 4, 8 = force the persistent kernel -- each where the geometry allows.  Initial value from CF_CONV_WINO (0 | 1 | 2 | 4 | 8);
// cf_conv_wino_enable changes it at run time (tests, A/B runs).
int g_wino_level = -1;
int wino_level() {
    if (g_wino_level < 0) {
        const char* e = getenv("CF_CONV_WINO");
        int v = e ? atoi(e) : 1;
        if (!(v == 0 || v == 1 || v == 2 || v == 4 || v == 8)) v = 1;
        g_wino_level = v;
    }
    return g_wino_level;
}
int wino_enabled() { return wino_level() != 0; }
int wino_force_ntw() { const int v = wino_level(); return (v == 2 || v == 4) ? v : 0; }

int wino_num_cus() { return device_cu_count(); }

bool wino_geometry(const ConvParams& p, int ntw, WinoGeom& g) {
    g.TW = p.W >= 32 ? 32 : 16;
    g.UW = g.TW / 2;
    g.RT = 32 / g.UW;
    g.TH = ntw * g.RT;
    g.PH = g.TH + 2;
    g.QW = g.TW / 4;
    g.LINE = g.UW * W_REC;
    g.ROWP = 4 * g.LINE + (g.UW == 8 ? 128 : 0);
    if (p.W % g.TW != 0 || p.H % g.TH != 0) return false;
    g.tiles_x = p.W / g.TW;
    g.tiles_y = p.H / g.TH;
    g.c1_pad = p.C2 > 0 ? ((p.C1 + W_CK - 1) / W_CK) * W_CK : (1 << 30);
    g.nchunk = p.C2 > 0 ? g.c1_pad / W_CK + (p.C2 + W_CK - 1) / W_CK : (p.C1 + W_CK - 1) / W_CK;
    g.ntask = g.PH * g.QW * 4;
    return g.ntask <= 256 * (ntw == 4 ? 2 : 1);
}

size_t wino_lds_bytes(const WinoGeom& g, bool pre) { return (size_t)2 * g.PH * g.ROWP + (pre ? (size_t)3 * g.nchunk * W_CK * sizeof(float) : 0); }
size_t wino_ps_lds_bytes(const WinoGeom& g) { return (size_t)PS_NBUF * g.PH * g.ROWP; }

template <int NTW, int PRE>
int launch_wino(const ConvParams& p, const WinoGeom& g, const _Float16* wpk, hipStream_t s) {
    const size_t lds_bytes = wino_lds_bytes(g, PRE);
    if (lds_bytes > 160 * 1024) { set_error("conv_wino: LDS tile too large"); return CF_ERR_ARG; }
    auto kern = conv_wino_kernel<NTW, PRE>;
    static bool attr_set[64] = {};
    const int dev = current_device_slot();
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev] = true;
    }
    dim3 grid((unsigned)(g.tiles_x * g.tiles_y * p.B), (unsigned)((p.Cout + 127) / 128));
    dim3 block(256);
    const double flops = 2.0 * (double)p.B * p.H * p.W * p.Cout * (p.C1 + p.C2) * 9.0;     // algorithmic (direct-form) flops
    hipEvent_t e0, e1;
    if (profile_on() && profile_events(PK_CONV_WINO, flops, &e0, &e1))
        hipExtLaunchKernelGGL(kern, grid, block, lds_bytes, s, e0, e1, 0, p, g, wpk);
    else
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, s, p, g, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("conv_wino launch failed: ") + hipGetErrorString(e)); return CF_ERR_LAUNCH; }
    return CF_OK;
}

template <int PRE>
int launch_wino_ps(const ConvParams& p, const WinoGeom& g, const _Float16* wpk, hipStream_t s) {
    const size_t lds_bytes = wino_ps_lds_bytes(g);
    if (lds_bytes > 160 * 1024) { set_error("conv_wino (persistent): LDS buffers too large"); return CF_ERR_ARG; }
    auto kern = conv_wino_ps_kernel<PRE>;
    static bool attr_set[64] = {};
    const int dev = current_device_slot();
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev] = true;
    }
    const long nitems = (long)g.tiles_x * g.tiles_y * p.B * ((p.Cout + 127) / 128);
    long nwg = wino_num_cus();                          // one workgroup per CU, a multiple of the 8 XCDs
    if (nwg > ((nitems + 7) & ~7L)) nwg = (nitems + 7) & ~7L;
    dim3 grid((unsigned)nwg), block(512);
    const double flops = 2.0 * (double)p.B * p.H * p.W * p.Cout * (p.C1 + p.C2) * 9.0;
    hipEvent_t e0, e1;
    if (profile_on() && profile_events(PK_CONV_WINO, flops, &e0, &e1))
        hipExtLaunchKernelGGL(kern, grid, block, lds_bytes, s, e0, e1, 0, p, g, wpk);
    else
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, s, p, g, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("conv_wino (persistent) launch failed: ") + hipGetErrorString(e)); return CF_ERR_LAUNCH; }
    return CF_OK;
}

// kernel shape for a layer: 0 = not a Winograd layer, 2 / 4 = one tile per workgroup with that many unit tiles per wave, 8 = persistent
int wino_pick_ntw(const ConvParams& p, WinoGeom& g) {
    const int lvl = wino_level();
    const long cblocks = (p.Cout + 127) / 128;
    if ((lvl == 1 || lvl == 8) && wino_geometry(p, 2, g) && g.ntask <= 256 && wino_ps_lds_bytes(g) <= 160 * 1024 &&
        (!p.in_norm || ((p.C1 & 3) == 0 && (reinterpret_cast<uintptr_t>(p.in_norm) & 15) == 0))) {
        const long nitems = (long)g.tiles_x * g.tiles_y * p.B * cblocks;
        if (lvl == 8 || nitems >= 2L * wino_num_cus()) return 8;
    }
    if (lvl == 8) return 0;
    const int force = wino_force_ntw();
    if (force != 2 && wino_geometry(p, 4, g)) {
        // one wave per SIMD wants the whole chip covered a few times over: >= 512 workgroups (two rounds of 256 CUs)
        const long nwg = (long)g.tiles_x * g.tiles_y * p.B * cblocks;
        if (force == 4) return 4;
        (void)nwg;
    }
    if (force != 4 && wino_geometry(p, 2, g)) return 2;
    return 0;
}

}  // namespace

// Cout: whole 128-channel blocks, or a last block at least three quarters full (the U-Net's 480), as conv_f16s' 128-channel shapes
bool conv_wino_applicable(const ConvParams& p) {
    if (!wino_enabled()) return false;
    if (conv_terms() != 3) return false;       // the one-term ("mixed precision") product mode is built in conv_f16s.hip only
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1 || p.w_bstride || p.scatter2x2) return false;
    if (!(p.Cout % 128 == 0 || (p.Cout > 128 && p.Cout % 128 >= 96))) return false;
    if (p.W < 16 || (p.W & 15) != 0) return false;
    if (((reinterpret_cast<uintptr_t>(p.x1) | reinterpret_cast<uintptr_t>(p.x2)) & 15) != 0) return false;
    const long HW = (long)p.H * p.W;
    if ((long)p.C1 * HW * 4 >= (1L << 31) || (long)p.C2 * HW * 4 >= (1L << 31)) return false;
    if ((long)p.out_ctotal * HW * 4 >= (1L << 30)) return false;          // 32-bit store offsets, two parking values (kernel epilogue)
    if (p.in_norm && (p.C2 != 0 || (reinterpret_cast<uintptr_t>(p.in_norm) & 3) != 0)) return false;
    if (p.gn_ws && (p.gn_groups <= 0 || p.Cout % p.gn_groups != 0)) return false;
    WinoGeom g;
    return wino_pick_ntw(p, g) != 0;
}

int launch_conv_wino(const ConvParams& p, const _Float16* wpk, hipStream_t s) {
    WinoGeom g;
    const int ntw = wino_pick_ntw(p, g);
    if (!ntw) { set_error("conv_wino: layer shape not supported"); return CF_ERR_ARG; }
    if (p.gn_ws && !p.gn_prezeroed &&
        hipMemsetAsync(p.gn_ws, 0, sizeof(double) * 2 * (size_t)p.B * p.gn_groups, s) != hipSuccess) { set_error("conv_wino: memset failed"); return CF_ERR_LAUNCH; }
    if (ntw == 8) return p.in_norm ? launch_wino_ps<1>(p, g, wpk, s) : launch_wino_ps<0>(p, g, wpk, s);
    if (p.in_norm) return ntw == 4 ? launch_wino<4, 1>(p, g, wpk, s) : launch_wino<2, 1>(p, g, wpk, s);
    return ntw == 4 ? launch_wino<4, 0>(p, g, wpk, s) : launch_wino<2, 0>(p, g, wpk, s);
}

}  // namespace cf

using namespace cf;

static void wino_params(ConvParams& p, const float* x1, int C1, const float* x2, int C2, const float* bias, const float* res, float* out,
                        int out_ctotal, int out_coff, int B, int H, int W, int Cout, int act, float alpha, double* gn_ws, int gn_groups) {
    p.x1 = x1; p.x2 = C2 ? x2 : nullptr; p.wt = nullptr; p.bias = bias; p.res = res; p.out = out; p.w_bstride = 0;
    p.C1 = C1; p.C2 = C2; p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.KH = 3; p.KW = 3; p.stride = 1;
    p.pad_h = 1; p.pad_w = 1; p.Ho = H; p.Wo = W; p.out_ctotal = out_ctotal; p.out_coff = out_coff; p.act = act; p.alpha = alpha; p.scatter2x2 = 0;
    p.gn_ws = gn_ws; p.gn_groups = gn_groups < 0 ? -gn_groups : gn_groups; p.gn_prezeroed = gn_groups < 0;
}

CF_CLOCK_READER(cf_debug_clock_wino, cf::g_clock_wino)
#ifdef CF_CLOCK_STAMPS
extern "C" int cf_debug_wino_phases(unsigned long long* out12) {
    unsigned long long* out8 = out12;
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return CF_ERR_LAUNCH;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(cf::g_wino_phase), sizeof(z)) != hipSuccess) return CF_ERR_LAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(cf::g_wino_phase), z, sizeof(z)) != hipSuccess) return CF_ERR_LAUNCH;
    return CF_OK;
}
#endif

extern "C" int cf_conv_wino_enable(int level) {
    const int prev = wino_level();
    if (level == 0 || level == 1 || level == 2 || level == 4 || level == 8) g_wino_level = level;
    return prev;
}

extern "C" int cf_conv2d_wino_ok(int B, int C1, int C2, int H, int W, int Cout, int prenorm) {
    if (B <= 0 || C1 <= 0 || C2 < 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    ConvParams p;
    float* dummy = reinterpret_cast<float*>(uintptr_t(256));      // never dereferenced
    wino_params(p, dummy, C1, C2 ? dummy : nullptr, C2, nullptr, nullptr, dummy, Cout, 0, B, H, W, Cout, CF_ACT_NONE, 1.f, nullptr, 0);
    if (prenorm) { p.in_norm = dummy; p.in_slope = 0.01f; }
    return conv_wino_applicable(p) ? 1 : 0;
}

extern "C" int cf_conv2d_wino(const float* x1, int C1, const float* x2, int C2, const void* wpk, const float* bias, const float* res, float* out,
                              int out_ctotal, int out_coff, int B, int H, int W, int Cout, int act, float alpha, double* gn_ws, int gn_groups,
                              void* stream) {
    CF_REQUIRE(x1 && wpk && out, "null pointer");
    CF_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "bad channel split C1=%d C2=%d", C1, C2);
    CF_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0, "bad shape B=%d H=%d W=%d Cout=%d", B, H, W, Cout);
    CF_REQUIRE(out_coff >= 0 && out_coff + Cout <= out_ctotal, "output channel slice out of range");
    CF_REQUIRE(act >= CF_ACT_NONE && act <= CF_ACT_SIGMOID, "bad activation %d", act);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(wpk) & 15) == 0, "packed weights must be 16-byte aligned");
    ConvParams p;
    wino_params(p, x1, C1, x2, C2, bias, res, out, out_ctotal, out_coff, B, H, W, Cout, act, alpha, gn_ws, gn_groups);
    CF_REQUIRE(!gn_ws || (p.gn_groups > 0 && Cout % p.gn_groups == 0 && out_coff == 0 && out_ctotal == Cout), "bad GroupNorm statistics request");
    CF_REQUIRE(conv_wino_applicable(p), "layer shape outside the Winograd kernel (3x3 / stride 1 / pad 1, Cout in whole 128-channel blocks or a last block >= 96, "
                                        "W %% 16 == 0, H a multiple of the tile rows, 16-byte aligned inputs; cf_conv2d_wino_ok answers per shape)");
    return launch_conv_wino(p, reinterpret_cast<const _Float16*>(wpk), as_stream(stream));
}

extern "C" int cf_conv2d_wino_prenorm(const float* x, int C, const float* in_norm, float in_slope, const void* wpk, const float* bias, float* out,
                                      int B, int H, int W, int Cout, float alpha, double* gn_ws, int gn_groups, void* stream) {
    CF_REQUIRE(x && wpk && out && in_norm, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Cout > 0, "bad shape B=%d C=%d H=%d W=%d Cout=%d", B, C, H, W, Cout);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(wpk) & 15) == 0, "packed weights must be 16-byte aligned");
    ConvParams p;
    wino_params(p, x, C, nullptr, 0, bias, nullptr, out, Cout, 0, B, H, W, Cout, CF_ACT_NONE, alpha, gn_ws, gn_groups);
    p.in_norm = in_norm;
    p.in_slope = in_slope;
    CF_REQUIRE(!gn_ws || (p.gn_groups > 0 && Cout % p.gn_groups == 0), "bad GroupNorm statistics request");
    CF_REQUIRE(conv_wino_applicable(p), "layer shape outside the Winograd kernel (cf_conv2d_wino_ok answers per shape)");
    return launch_conv_wino(p, reinterpret_cast<const _Float16*>(wpk), as_stream(stream));
}
