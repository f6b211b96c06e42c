// Implicit-GEMM convolution on the f16 MFMA (v_mfma_f32_32x32x16_f16) with a 3-term "hi/lo" operand split:
//
//     x = hi + lo,   hi = fp16(x),   lo = fp16(x - hi)                             (22 significant bits per operand)
//     a*b ~= ah*bh + ah*bl + al*bh                                                 (al*bl ~ 2^-22 |a*b| dropped)
//
// Products of fp16 values are exact in the fp32 accumulator, so the result carries ~2^-22 relative operand error --
// two bits short of fp32 -- at 3 MFMAs per 16-deep k-step, i.e. up to 16/3 = 5.3x the rate of the exact fp32 MFMA
// (v_mfma_f32_32x32x2_f32, conv.hip).  fp16's narrow exponent is handled where it matters: the weights are scaled by
// an exact power of two at pack time (max|w| -> ~2^10, undone through alpha), so their lo halves stay normal; the
// activations this path sees are GroupNorm/LayerNorm/GELU outputs of O(1), whose lo halves lose at most 2^-25
// absolute (fp16 subnormal spacing) -- below the fp32 rounding of the sum.  Measured operand-level error on
// conv-shaped data: 7e-8 relative to max|y| vs 3e-7 for an fp32 conv; the recurrent flow network keeps
// its 1e-4 px EPE bar with a wide margin (tests/test_gpu_models.py), which bf16 splits (8-bit pieces) would not.
//
// GEMM view:  D[co][pixel] = sum_{tap, ci} W[co][ci][tap] * X[b, ci, oy*s + kh - p, ox*s + kw - p]
//   * k order = (channel chunk, tap, channel in chunk): one MFMA k-step = 16 consecutive channels at one tap, so a
//     lane's B fragment is 8 consecutive channels of ONE input pixel.
//   * the input patch of a workgroup is staged through LDS transposed on the fly: NCHW fp32 rows are read coalesced
//     along x (8 channels per thread), split into hi/lo halves and written as one 80-byte (CK=16) / 144-byte (CK=32)
//     record per pixel ([hi c0..CK) | lo c0..CK) | 16 B pad]); the odd multiple of 16 B makes every ds_read_b128 /
//     ds_write_b128 lane group hit 16 distinct 16-byte slots (conflict free).  The 3x3 halo is staged once per
//     workgroup, so HBM sees each input element ~1.3x per 64-channel output block.
//   * weights are packed on the host in exact fragment order (hi and lo planes), 1 KiB per (m-tile, chunk, tap, kstep,
//     part): an A fragment is one coalesced global_load_dwordx4 per lane from L1/L2 (all workgroups share them).
//   * double-buffered LDS, one barrier per chunk; global loads of chunk c+1 are issued before the MFMAs of chunk c and
//     converted/written after them (issue-early / write-late).
//   * wave tile: one 32-channel m-tile x NTW n-tiles of 32 pixels, one fp32 accumulator per tile.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "conv.h"
#include "profile.h"

namespace cf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

CF_CLOCK_DECL(g_clock_f16s)

struct F16sGeom {
    int TW, TH, NIMG;        // output tile of a workgroup: NIMG images x TH rows x TW cols (NIMG*TH*TW <= NT_WG*32)
    int PH, PW;              // staged patch rows / cols per image
    int pstep;               // input step between patch pixels (1, or the stride for 1x1 convs)
    int ostep;               // patch step between output pixels (stride, or 1 for 1x1 convs)
    int tiles_x, tiles_y, bgroups;
    int nchunk;
    int ablate;              // timing experiments only (CF_F16S_ABLATE): 1 = every chunk reads chunk 0's weight fragments (L1-resident), 2 = no fragment loads, 3 = no output stores
    int c1_pad;              // C1 rounded up to a multiple of CK: chunks below it read x1, the others x2 (the packed weights follow the same split)
    int NQ;                  // vector staging: 16-byte column quads per patch row (0: scalar staging)
    // magic multipliers floor(2^32/d)+1 for the index decodes (exact for n < 2^32/d; d == 1 handled apart): the kernel's setup was
    // ~8k ticks of integer division sequences, a tenth of a 4-chunk workgroup's life
    int PWR, pwh, kx1, kx2;  // LDS row pitch in records and the record offsets of taps kx = 1, 2 (stride 2: columns de-interleaved by parity)
    unsigned m_tx, m_ty, m_percg, m_phnq, m_nq, m_nrec, m_phpw, m_pw, m_thtw, m_tw;
};

__host__ __device__ inline unsigned f16s_magic(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d + 1ull); }
__device__ __forceinline__ int fdiv(int n, int d, unsigned m) { return d <= 1 ? n : (int)__umulhi((unsigned)n, m); }

// No range clamp: an activation beyond fp16's range (|x| >= 65520) becomes hi = inf, lo = NaN and poisons the output loudly, NaN and
// Inf inputs propagate as they do through an fp32 convolution (a clamp here saturated them silently and cost one VALU per element).
// Supported activation range: |x| < 65504, stated in include/cineflow.h.
__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// NLW = number of dedicated loader waves.  NLW == 0: all four waves stage and compute (first design).  NLW > 0: waves
// 0-3 only run MFMAs (their vmcnt queue holds nothing but weight fragments), waves 4.. only stage the input patch, two
// chunks of loads in flight, always one LDS buffer ahead.  vmcnt retires in issue order, so in the mixed design every
// wait for a weight fragment also waited for the staging loads issued before it (ablating those loads made the kernel
// 19-33 % faster); with separate roles nothing in the MFMA waves ever waits for HBM.
// WL = 1: the weight fragments go through LDS instead of straight from L1/L2 into registers.  Ablations (profiles/r02_conv_weight_path.md)
// showed that the per-wave fragment loads -- 2 KiB per k-step and wave, L1-resident or not -- cost 30-40 % of the kernel on every shape:
// the vector-memory return path (64 B/clk/CU) is what the MFMAs wait for.  With WL the workgroup fetches each fragment ONCE by LDS-DMA
// (global_load_lds_dwordx4, no VGPRs) into a two-slot ring of tap groups (3x3: one kernel row = 3 taps per slot) and every wave reads
// its A operands with ds_read_b128 (256 B/clk/CU); one barrier per group, the DMA of group q + 1 is in flight while group q is consumed.
#ifdef CF_F16S_ABLATION_BUILD
// per-phase clock sums over all waves (lane 0 of each): prologue, MFMA steps, write_stage, barrier wait, epilogue, waves
__device__ unsigned long long g_f16s_phase[9];   // [6..8]: epilogue split: scale + activation, coordinates + stores, statistics
#define F16S_CLK() __builtin_readcyclecounter()
#endif
// TERMS = 3: the hi/lo split (al*bh + ah*bl + ah*bh); TERMS = 1: ah*bh only -- operands rounded to fp16, fp32 accumulation: what the reference's
// fp16 autocast computes on the segmentation path (mixed_precision=True, nnunet/network_architecture/neural_network.py:140-146), selected per
// thread by cf_conv_terms(1).  Same staging, same packed weights (the lo planes are simply not multiplied).
template <int KH, int KW, int CK, int WM, int NTW, int MAXT, int NLW, int NW, int VEC, int PRE, int WL, int TERMS>
__global__ void __launch_bounds__(64 * NW + 64 * NLW, NLW ? 5 : (NW == 8 ? 4 : ((NTW <= 2 && MAXT <= 3) ? 4 : 2)))
conv_f16s_kernel(const ConvParams p, const F16sGeom g, const _Float16* __restrict__ wpk) {
    constexpr int KHW = KH * KW;           // taps: 3x3, 1x1, and the separable 1x5 / 5x1 of RAFT's SepConvGRU
    constexpr int KS = CK / 16;            // MFMA k-steps per tap per chunk
    constexpr int REC = CK * 4 + 16;       // bytes per pixel record
    constexpr int NT_WG = NTW * (NW / WM);  // n-tiles per workgroup (NW MFMA waves = WM m-tiles x NW/WM n-tile groups)
    constexpr int NG = CK / 8;             // 8-channel groups per record
    constexpr int NSTAGE = NLW ? 64 * NLW : 64 * NW;   // threads that stage
    (void)NT_WG;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    CF_CLOCK_BEGIN();

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform by construction: keep it in an SGPR
#ifdef CF_F16S_ABLATION_BUILD
    const unsigned long long ph_t0 = F16S_CLK();
    unsigned long long ph_t1 = 0, ph_stage = 0, ph_bar = 0, ph_t2 = 0;
#endif
    const int half = lane >> 5, l31 = lane & 31;
    const bool loader = NLW > 0 && wave >= NW;     // wave-uniform
    const int cw = wave % NW;                       // index among the MFMA waves
    const int mt = blockIdx.y * WM + (cw % WM);     // 32-channel m-tile of this wave
    const int ngrp = cw / WM;                       // n-tile group of this wave

    // ---- workgroup tile decode (XCD-banded so that vertically adjacent tiles share one L2)
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x;
        int xcd = bid & 7, qn = nb >> 3, rn = nb & 7;
        bid = ((xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
    }
    int t2 = fdiv(bid, g.tiles_x, g.m_tx);
    const int tx = bid - t2 * g.tiles_x;
    const int bg = fdiv(t2, g.tiles_y, g.m_ty);
    const int ty = t2 - bg * g.tiles_y;
    const int x0 = tx * g.TW, y0 = ty * g.TH, b0 = bg * g.NIMG;
    const int HW = p.H * p.W;
    const int HoWo = p.Ho * p.Wo;
    const int iy_org = y0 * p.stride - p.pad_h, ix_org = x0 * p.stride - p.pad_w;
    const int nrec = g.NIMG * g.PH * g.PW;
    const int buf_bytes = g.NIMG * g.PH * g.PWR * REC;
    const bool do_stats = p.gn_ws != nullptr;

    // =================================================================================================================
    // staging role (all waves when NLW == 0, waves 4.. otherwise)
    // =================================================================================================================
    // Inputs are read with raw buffer loads: the per-lane 32-bit byte offset of (sample, channel 0, iy, ix) is computed
    // once per task, the channel term is added per load, and the descriptor's range check returns 0 for (a) pixels
    // outside the image / batch, whose offset is parked at 2 GiB, and (b) channel reads past the end of the tensor.
    // (c) the padded channel tail of each input -- channels [C1, c1_pad) of x1 and [C2, ...) of x2, whose packed weights are exactly
    // zero: their loads are parked out of range too (issue_loads*), so a NaN / Inf in the NEXT sample's first channels cannot reach this
    // sample as NaN * 0.  Host checks: one sample of each tensor < 2 GiB (split-aware packing handles any C1 when x2 is present).
    constexpr unsigned OOB = 0x80000000u;
    unsigned t_o1[MAXT], t_o2[MAXT];  // byte offsets into x1 / x2 (OOB when invalid)
    int t_lds[MAXT];                  // byte offset of the 16-byte hi slot inside a buffer, -1: no task
    unsigned t_g8[MAXT];              // (8-channel group inside the chunk) * 8
    const int stid = NLW ? tid - 64 * NW : tid;
    const bool stages = NLW ? loader : true;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        t_o1[t] = OOB;
        t_o2[t] = OOB;
        t_lds[t] = -1;
        t_g8[t] = 0;
        if (!stages) continue;
        int task = stid + t * NSTAGE;
        int grp = fdiv(task, nrec, g.m_nrec);
        int pr = task - grp * nrec;
        if (grp < NG) {
            int img = fdiv(pr, g.PH * g.PW, g.m_phpw);
            int q = pr - img * (g.PH * g.PW);
            int py = fdiv(q, g.PW, g.m_pw), px = q - py * g.PW;
            int iy = iy_org + py * g.pstep, ix = ix_org + px * g.pstep;
            int b = b0 + img;
            const int pxs = g.pwh ? (px >> 1) + (px & 1) * g.pwh : px;    // stride 2: even columns first, then the odd ones
            t_lds[t] = (((img * g.PH + py) * g.PWR) + pxs) * REC + grp * 16;
            t_g8[t] = grp * 8;
            if (b < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                const unsigned sp = (unsigned)(iy * p.W + ix);
                t_o1[t] = ((unsigned)b * p.C1 * HW + sp) * 4u;
                t_o2[t] = ((unsigned)b * p.C2 * HW + sp) * 4u;
            }
        }
    }
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1), 0, (int)((long)p.B * p.C1 * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x2 ? p.x2 : p.x1), 0, p.x2 ? (int)((long)p.B * p.C2 * HW * 4) : 0, 0x00020000);
    const unsigned HW4 = (unsigned)HW * 4u;

    // The loaded values stay RAW in registers until the LDS write (nothing consumes a load result early).
    auto issue_loads = [&](int chunk, float (&stg)[MAXT][8]) {
        const int c0 = chunk * CK;
        const bool in1 = c0 < g.c1_pad;                   // workgroup-uniform; c1_pad = C1 rounded up to whole chunks (split-aware packing)
        const unsigned cb = (unsigned)(in1 ? c0 : c0 - g.c1_pad);
        const unsigned clim = (unsigned)(in1 ? p.C1 : p.C2);   // channels at or past it are the zero-weight tail: never fetched (a select, not a branch)
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const unsigned v0 = (in1 ? t_o1[t] : t_o2[t]) + (cb + t_g8[t]) * HW4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned off = (cb + t_g8[t] + (unsigned)j < clim) ? v0 + (unsigned)j * HW4 : OOB;
                const unsigned raw = in1 ? __builtin_amdgcn_raw_buffer_load_b32(rsrc1, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b32(rsrc2, off, 0, 0);
                stg[t][j] = __builtin_bit_cast(float, raw);
            }
        }
    };
    auto write_stage = [&](int chunk, const float (&stg)[MAXT][8]) {
        unsigned char* base = lds + (chunk & 1) * buf_bytes;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t_lds[t] < 0) continue;
            f16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 h, l;
                split_f16(stg[t][j], h, l);
                hi[j] = h;
                lo[j] = l;
            }
            *reinterpret_cast<f16x8*>(base + t_lds[t]) = hi;
            *reinterpret_cast<f16x8*>(base + t_lds[t] + CK * 2) = lo;
        }
    };


    // ---- vector staging (VEC): task = 4 image columns aligned to 16 bytes x 4 channels -> 4 buffer_load_dwordx4 instead of 16
    // buffer_load_dword for the same 16 elements.  Time stamps in the correlation kernel showed ~150-240 cycles per vector-memory
    // instruction once the CU's address path back-pressures; the scalar staging issues 64 such instructions per workgroup and
    // chunk, this path 15.  Aligned quads never straddle the image border when W % 4 == 0, so a quad is either wholly inside
    // (one range-checked load) or wholly padding (offset parked at 2 GiB -> zeros); columns of a quad that lie outside the
    // patch are simply not written.  Host checks: stride 1, W % 4 == 0, 16-byte aligned tensors, one task per thread.
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    constexpr int VT = VEC ? VEC : 1;                                     // vector tasks per staging thread
    unsigned v_o1[VT], v_o2[VT], v_c4[VT], v_mask[VT];
    int v_lds[VT];
#pragma unroll
    for (int t = 0; t < VT; ++t) { v_o1[t] = OOB; v_o2[t] = OOB; v_c4[t] = 0; v_mask[t] = 0; v_lds[t] = 0; }
    if (VEC) {
        constexpr int NCG = CK / 4;
        const int xq0 = ix_org & ~3;
        const int per_cg = g.NIMG * g.PH * g.NQ;
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            const int task = tid + t * NSTAGE;
            const int cg = fdiv(task, per_cg, g.m_percg);
            int r = task - cg * per_cg;
            if (cg < NCG) {
                const int img = fdiv(r, g.PH * g.NQ, g.m_phnq);
                r -= img * (g.PH * g.NQ);
                const int py = fdiv(r, g.NQ, g.m_nq), q = r - py * g.NQ;
                const int iy = iy_org + py, xq = xq0 + 4 * q, b = b0 + img;
                const int pxp = xq - ix_org;                              // patch column of the quad's first pixel (may be < 0)
#pragma unroll
                for (int k = 0; k < 4; ++k) v_mask[t] |= ((unsigned)(pxp + k) < (unsigned)g.PW) ? (1u << k) : 0u;
                v_lds[t] = ((img * g.PH + py) * g.PWR + pxp) * REC + cg * 8;
                v_c4[t] = cg * 4;
                if (b < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)xq < (unsigned)p.W) {
                    const unsigned sp = (unsigned)(iy * p.W + xq);
                    v_o1[t] = ((unsigned)b * p.C1 * HW + sp) * 4u;
                    v_o2[t] = ((unsigned)b * p.C2 * HW + sp) * 4u;
                }
            }
        }
    }
    auto issue_loads_v = [&](int chunk, f32x4v (&stg)[VT][4]) {
        const int c0 = chunk * CK;
        const bool in1 = c0 < g.c1_pad;
        const unsigned cb = (unsigned)(in1 ? c0 : c0 - g.c1_pad);
        const unsigned clim = (unsigned)(in1 ? p.C1 : p.C2);
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            const unsigned v0 = (in1 ? v_o1[t] : v_o2[t]) + (cb + v_c4[t]) * HW4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned off = (cb + v_c4[t] + (unsigned)j < clim) ? v0 + (unsigned)j * HW4 : OOB;
                stg[t][j] = in1 ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc1, off, 0, 0))
                                : __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc2, off, 0, 0));
            }
        }
    };
    // PRE: the input is a raw convolution output; its InstanceNorm / GroupNorm + LeakyReLU is applied here, between the load and the
    // hi/lo split: v = lrelu((v - mean) * scale + shift) with the per-channel triples of this sample staged once in LDS behind the two
    // patch buffers ([3][nchunk * CK] floats; channels past C1 hold zeros).  Padding quads (offset parked out of range) stay zero.
    const float* ctab = reinterpret_cast<const float*>(lds + 2 * buf_bytes);
    const int ctab_n = g.nchunk * CK;
    auto write_stage_v = [&](int chunk, const f32x4v (&stg)[VT][4]) {
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            unsigned char* base = lds + (chunk & 1) * buf_bytes + v_lds[t];
            f32x4v cm = {0.f, 0.f, 0.f, 0.f}, ca = cm, cs = cm;
            const bool valid = v_o1[t] != OOB;
            if (PRE) {
                const int c0 = chunk * CK + (int)v_c4[t];
                cm = *reinterpret_cast<const f32x4v*>(ctab + c0);
                ca = *reinterpret_cast<const f32x4v*>(ctab + ctab_n + c0);
                cs = *reinterpret_cast<const f32x4v*>(ctab + 2 * ctab_n + c0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!((v_mask[t] >> k) & 1u)) continue;
                f16x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = stg[t][j][k];
                    if (PRE) {
                        v = (v - cm[j]) * ca[j] + cs[j];
                        v = p.in_slope < 0.f ? gelu_as(v) : (v > 0.f ? v : v * p.in_slope);
                        v = valid ? v : 0.f;
                    }
                    _Float16 h, l;
                    split_f16(v, h, l);
                    hi[j] = h;
                    lo[j] = l;
                }
                *reinterpret_cast<f16x4*>(base + k * REC) = hi;
                *reinterpret_cast<f16x4*>(base + k * REC + CK * 2) = lo;
            }
        }
    };

    if (NLW > 0 && loader) {
        // ---- loader waves: two register sets keep TWO chunks of loads in flight (loop unrolled by two so the sets are
        // statically indexed); the wait in front of each LDS write is for loads issued a whole chunk period earlier.
        // Barrier count must equal the MFMA branch: 1 + nchunk (+ 3 when statistics are fused).
        float sa[MAXT][8], sb[MAXT][8];
        const int n = g.nchunk;
        issue_loads(0, sa);
        if (n > 1) issue_loads(1, sb);
        write_stage(0, sa);
        __syncthreads();
        int c = 0;
        while (true) {
            if (c + 2 < n) issue_loads(c + 2, sa);
            if (c + 1 < n) write_stage(c + 1, sb);
            __syncthreads();
            if (++c >= n) break;
            if (c + 2 < n) issue_loads(c + 2, sb);
            if (c + 1 < n) write_stage(c + 1, sa);
            __syncthreads();
            if (++c >= n) break;
        }
        if (do_stats) { __syncthreads(); __syncthreads(); __syncthreads(); }
        return;
    }

    // =================================================================================================================
    // MFMA role
    // =================================================================================================================
    // the first chunk's loads leave before the (division-heavy) output-coordinate setup below, which then overlaps their latency
    float stg0[MAXT][8];   // dead (eliminated) in the VEC instantiations
    f32x4v stgv[VT][4];
    if (VEC) issue_loads_v(0, stgv);
    else if (NLW == 0) issue_loads(0, stg0);
    // ---- per-lane B-fragment record offsets and output coordinates of this wave's n-tiles
    int b_rec[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        int pidx = (ngrp * NTW + nt) * 32 + l31;
        int img = fdiv(pidx, g.TH * g.TW, g.m_thtw);
        int q = pidx - img * (g.TH * g.TW);
        int tyy = fdiv(q, g.TW, g.m_tw), txx = q - tyy * g.TW;
        if (img >= g.NIMG) { img = 0; tyy = 0; txx = 0; }
        b_rec[nt] = ((img * g.PH + tyy * g.ostep) * g.PWR + txx * (g.pwh ? 1 : g.ostep)) * REC + half * 16;
    }

    f32x16 acc1[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[nt][r] = 0.f;

    constexpr int NSTEP = KHW * KS;
    if constexpr (WL) {
        // ---- weights through LDS: ring of two slots, one slot = the fragments of G consecutive k-steps for the WM m-tiles of the workgroup
        constexpr int G = (KHW == 9) ? 3 : ((KHW == 1) ? NSTEP : 5);      // k-steps per group (3x3: one kernel row)
        constexpr int NGRP = NSTEP / G;
        static_assert(NGRP * G == NSTEP, "weight groups must tile the chunk");
        constexpr int WSLOT = WM * G * 2 * 1024;                          // bytes per slot
        unsigned char* wl = lds + 2 * buf_bytes + (PRE ? 3 * ctab_n * 4 : 0);
        const f16x8* wbase = reinterpret_cast<const f16x8*>(wpk) + lane;
        const int nq = g.nchunk * NGRP;
        // The DMA is issued from inline asm on purpose: hipcc tracks a __builtin_amdgcn_global_load_lds as a pending LDS write and puts
        // s_waitcnt vmcnt(0) in front of the next ds_read that may alias it -- i.e. it waited for group q + 1 before reading group q
        // (measured: 20 % of the kernel).  Hidden from the compiler, the DMAs only make ITS counted vmcnt waits stricter (vmcnt counts
        // every outstanding operation and retires in order), never weaker; their own completion is waited for explicitly below.
        auto issue_w = [&](int q) {       // group q = (chunk q / NGRP, steps (q % NGRP) * G ...): packed contiguously per m-tile
            const unsigned dst0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)(wl + (q & 1) * WSLOT));
            for (int f = wave; f < WM * G * 2; f += NW) {                 // wave-uniform
                const int m = f / (G * 2), j = f - m * (G * 2);
                const f16x8* src = wbase + (((long)(blockIdx.y * WM + m) * g.nchunk * NSTEP + (long)q * G) * 2 + j) * 64;
                const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)f * 1024u);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        };
        issue_w(0);
        if (PRE) {
            float* ct = reinterpret_cast<float*>(lds + 2 * buf_bytes);
            for (int k = tid; k < 3 * ctab_n; k += 64 * NW) {
                const int which = k / ctab_n, c = k - which * ctab_n;
                ct[k] = c < p.C1 ? p.in_norm[((long)b0 * 3 + which) * p.C1 + c] : 0.f;
            }
            __syncthreads();
        }
        if (VEC) write_stage_v(0, stgv);
        else write_stage(0, stg0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // group 0's fragments have landed
        __syncthreads();
        const unsigned char* wa = wl + ((cw % WM) * G * 2) * 1024 + lane * 16;
        for (int c = 0; c < g.nchunk; ++c) {
            const bool more = c + 1 < g.nchunk;
            const unsigned char* xb = lds + (c & 1) * buf_bytes;
#pragma unroll
            for (int grp = 0; grp < NGRP; ++grp) {
                const int q = c * NGRP + grp;
                if (q + 1 < nq) issue_w(q + 1);
                if (grp == 0) {                   // unconditional (the last chunk re-reads itself), younger than the DMAs: the counted wait at the
                    const int cn = more ? c + 1 : c;      // end of this group leaves them in flight
                    if (VEC) issue_loads_v(cn, stgv);
                    else issue_loads(cn, stg0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* ws = wa + (q & 1) * WSLOT;
#pragma unroll
                for (int sg = 0; sg < G; ++sg) {
                    const int step = grp * G + sg;
                    const int tap = step / KS, ks = step % KS;
                    const int ky = tap / KW, kx = tap % KW;
                    const int toff = (ky * g.PWR + (g.pwh ? (kx >> 1) + (kx & 1) * g.pwh : kx)) * REC;
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(ws + (sg * 2) * 1024);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(ws + (sg * 2 + 1) * 1024);
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const unsigned char* rp = xb + b_rec[nt] + toff + ks * 32;
                        const f16x8 bh = *reinterpret_cast<const f16x8*>(rp);
                        const f16x8 bl = *reinterpret_cast<const f16x8*>(rp + CK * 2);
                        if constexpr (TERMS == 3) {
                            acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[nt], 0, 0, 0);
                            acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[nt], 0, 0, 0);
                        }
                        acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1[nt], 0, 0, 0);
                    }
                }
                if (grp == NGRP - 1 && more) {
                    if (VEC) write_stage_v(c + 1, stgv);
                    else write_stage(c + 1, stg0);
                }
                // group q + 1's fragments must have landed before anyone reads them: vmcnt retires in order, so at the end of group 0 the
                // patch loads issued after the DMAs (VEC: 4 per task) may stay in flight
                if (VEC && grp == 0 && NGRP > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VEC * 4) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    } else {
    // packed weights: fragment (mt, chunk, tap, ks, part) = 64 lanes x 8 halves
    const f16x8* wfrag = reinterpret_cast<const f16x8*>(wpk) + (long)mt * g.nchunk * (KHW * KS * 2) * 64 + lane;

    // A-fragment register ring, prefetched D steps ahead (NSTEP % R == 0 keeps the slots static).
    constexpr int R = (NSTEP % 3 == 0) ? 3 : 2;
    constexpr int D = R - 1;
    f16x8 aH[R] = {}, aL[R] = {};
    auto load_a = [&](int chunk, int step, int slot) {
#ifdef CF_F16S_ABLATION_BUILD          // timing experiments of profiles/r02_conv_weight_path.md: the run-time knob made every fragment load conditional
        if (g.ablate == 2) return;
        const f16x8* wc = wfrag + ((long)(g.ablate == 1 ? 0 : chunk) * NSTEP + step) * 2 * 64;
#else
        const f16x8* wc = wfrag + ((long)chunk * NSTEP + step) * 2 * 64;
#endif
        aH[slot] = wc[0];
        aL[slot] = wc[64];
    };

#pragma unroll
    for (int sidx = 0; sidx < D; ++sidx) load_a(0, sidx, sidx % R);
    if (PRE) {
        float* ct = reinterpret_cast<float*>(lds + 2 * buf_bytes);
        for (int k = tid; k < 3 * ctab_n; k += 64 * NW) {
            const int which = k / ctab_n, c = k - which * ctab_n;
            ct[k] = c < p.C1 ? p.in_norm[((long)b0 * 3 + which) * p.C1 + c] : 0.f;
        }
        __syncthreads();
    }
    if (VEC) write_stage_v(0, stgv);
    else if (NLW == 0) write_stage(0, stg0);
    __syncthreads();
#ifdef CF_F16S_ABLATION_BUILD
    ph_t1 = F16S_CLK();
#endif

    for (int c = 0; c < g.nchunk; ++c) {
        const bool more = c + 1 < g.nchunk;
        const unsigned char* xb = lds + (c & 1) * buf_bytes;
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            const int tap = step / KS, ks = step % KS;
            // Prefetch the fragments of step + D (possibly the next chunk's first steps) and, at step 0, the next chunk's patch.  Every
            // load is UNCONDITIONAL (the last chunk re-reads itself: L1/L2 hits) and pinned here by a scheduling barrier.  Reading the
            // ISA of round 1's loop showed why the fragment loads cost 35-60 % (profiles/r02_conv_weight_path.md): with the loads
            // conditional (`if (more)`) hipcc's waitcnt pass merged the paths and waited vmcnt(0) -- for the patch loads it had just
            // issued -- and, unpinned, its scheduler sank each fragment load to one MFMA before its use; either way the two-step
            // prefetch did not exist.  Unconditional + pinned, the waits come out counted (vmcnt(N) leaves the younger loads in flight).
            const int cn = more ? c + 1 : c;
            if (step + D < NSTEP) load_a(c, step + D, (step + D) % R);
            else load_a(cn, step + D - NSTEP, (step + D) % R);
            if (NLW == 0 && step == 0) {
                if (VEC) issue_loads_v(cn, stgv);
                else issue_loads(cn, stg0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // record offset of tap (ky, kx); stride 2 stores the patch columns de-interleaved (even columns, then odd), so column
            // 2 * txx + kx sits at txx + (kx >> 1) + (kx & 1) * pwh
            const int ky = tap / KW, kx = tap % KW;      // compile-time after unrolling
            const int toff = (ky * g.PWR + (g.pwh ? (kx >> 1) + (kx & 1) * g.pwh : kx)) * REC;
            const f16x8 ah = aH[step % R], al = aL[step % R];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const unsigned char* rp = xb + b_rec[nt] + toff + ks * 32;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(rp);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(rp + CK * 2);
                // small terms first, then the main term, into one accumulator
                if constexpr (TERMS == 3) {
                    acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[nt], 0, 0, 0);
                    acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[nt], 0, 0, 0);
                }
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1[nt], 0, 0, 0);
            }
        }
#ifdef CF_F16S_ABLATION_BUILD
        const unsigned long long ph_a = F16S_CLK();
#endif
        if (NLW == 0 && more) {
            if (VEC) write_stage_v(c + 1, stgv);
            else write_stage(c + 1, stg0);
        }
#ifdef CF_F16S_ABLATION_BUILD
        const unsigned long long ph_b = F16S_CLK();
#endif
        __syncthreads();
#ifdef CF_F16S_ABLATION_BUILD
        ph_stage += ph_b - ph_a;
        ph_bar += F16S_CLK() - ph_b;
#endif
    }
#ifdef CF_F16S_ABLATION_BUILD
    ph_t2 = F16S_CLK();
#endif

    }

    // ---- epilogue (same fusion as conv.hip): value = act(alpha*acc + bias) + res  (alpha carries the weight scale 2^-s)
    // Optional fused GroupNorm / InstanceNorm statistics of the stored values (sum, sum of squares per (sample, group)):
    // per lane the NTW tiles are added per register (= channel), then a transpose-reduce over the 32 pixel lanes of the
    // half-wave (16 -> 8 -> 4 -> 2 -> 1 registers, 16 shuffles per quantity instead of 80) leaves each channel's total in
    // one lane pair, which adds it to the fp64 workspace.  Host guarantees one sample per workgroup (NIMG == 1).
    float ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[r] = 0.f; ssq[r] = 0.f; }
    // The epilogue is written as separate compact phases (scale+bias, ONE uniform switch over the activation, residual+store):
    // with the activation switch inside the per-element code the unrolled epilogue was ~40 KB of instructions, of which each
    // workgroup executes a sparse path once -- time stamps showed 28-34k ticks (a third of a 64-channel workgroup's life) spent
    // there, on instruction fetch.
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float bv = 0.f;
        if (p.bias && co < p.Cout) bv = p.bias[p.scatter2x2 ? (co >> 2) : co];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc1[nt][r] = p.alpha * acc1[nt][r] + bv;
    }
    switch (p.act) {   // workgroup-uniform
        case CF_ACT_GELU:
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc1[nt][r] = act_apply(acc1[nt][r], CF_ACT_GELU);
                    __builtin_amdgcn_sched_barrier(0);   // one element at a time: these cold paths must not set the kernel's register count
                }
            break;
        case CF_ACT_RELU:
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc1[nt][r] = act_apply(acc1[nt][r], CF_ACT_RELU);
                    __builtin_amdgcn_sched_barrier(0);   // one element at a time: these cold paths must not set the kernel's register count
                }
            break;
        case CF_ACT_LRELU:
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc1[nt][r] = act_apply(acc1[nt][r], CF_ACT_LRELU);
                    __builtin_amdgcn_sched_barrier(0);   // one element at a time: these cold paths must not set the kernel's register count
                }
            break;
        case CF_ACT_TANH:
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc1[nt][r] = act_apply(acc1[nt][r], CF_ACT_TANH);
                    __builtin_amdgcn_sched_barrier(0);   // one element at a time: these cold paths must not set the kernel's register count
                }
            break;
        case CF_ACT_SIGMOID:
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc1[nt][r] = act_apply(acc1[nt][r], CF_ACT_SIGMOID);
                    __builtin_amdgcn_sched_barrier(0);   // one element at a time: these cold paths must not set the kernel's register count
                }
            break;
        default: break;
    }
    // output coordinates are derived here rather than ahead of the main loop: 4*NTW fewer registers live across it.
    // Stores go through a buffer resource over the NIMG output samples of this workgroup: one 32-bit add per store, and a pixel outside the
    // tile / a channel beyond Cout is an out-of-range offset (the store is dropped by the range check) instead of a branch around the store
    // and a 64-bit address per element.  Two parking values so that "both invalid" cannot wrap back into range; the host keeps
    // NIMG x (output sample bytes) below 1 GiB (launch_f16s_v).
    constexpr unsigned OOB_PX = 0x80000000u, OOB_CH = 0x40000000u;
    const unsigned up4 = p.scatter2x2 ? 4u : 1u;
    const unsigned osample = (unsigned)p.out_ctotal * (unsigned)HoWo * up4;                    // elements per output sample
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        p.out + ((long)b0 * p.out_ctotal + p.out_coff) * (long)HoWo * up4, 0,
        (int)(((unsigned)(b0 + g.NIMG <= p.B ? g.NIMG : p.B - b0) * osample - (unsigned)p.out_coff * (unsigned)HoWo * up4) * 4u), 0x00020000);
    bool o_ok[NTW];
    unsigned o_off[NTW];
    long r_off[NTW];
    auto out_coords = [&]() {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            int pidx = (ngrp * NTW + nt) * 32 + l31;
            asm volatile("" : "+v"(pidx));                   // keep the address math below the main loop
            int img = fdiv(pidx, g.TH * g.TW, g.m_thtw);
            int q = pidx - img * (g.TH * g.TW);
            int tyy = fdiv(q, g.TW, g.m_tw), txx = q - tyy * g.TW;
            bool in_tile = img < g.NIMG;
            if (!in_tile) { img = 0; tyy = 0; txx = 0; }
            int b = b0 + img, oy = y0 + tyy, ox = x0 + txx;
            o_ok[nt] = in_tile && b < p.B && oy < p.Ho && ox < p.Wo;
            const unsigned px = p.scatter2x2 ? (unsigned)(2 * oy) * (unsigned)(2 * p.Wo) + 2u * ox : (unsigned)oy * p.Wo + ox;
            o_off[nt] = o_ok[nt] ? ((unsigned)img * osample + px) * 4u : OOB_PX;
            r_off[nt] = (long)b * p.Cout * HoWo + (long)oy * p.Wo + ox;
        }
    };
#ifdef CF_F16S_ABLATION_BUILD
    const unsigned long long ph_e1 = F16S_CLK();
#endif
    {
    out_coords();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool co_ok = co < p.Cout;
        unsigned ochan;
        if (p.scatter2x2) {
            const unsigned cr = (unsigned)co >> 2, dy = ((unsigned)co >> 1) & 1u, dx = (unsigned)co & 1u;
            ochan = (cr * (4u * HoWo) + dy * (2u * p.Wo) + dx) * 4u;
        } else {
            ochan = (unsigned)co * (unsigned)HoWo * 4u;
        }
        if (!co_ok) ochan = OOB_CH;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const bool ok = o_ok[nt] && co_ok;
            float v = acc1[nt][r];
            if (p.res) { if (ok) v += p.res[r_off[nt] + (long)co * HoWo]; }   // rare (SingleConv / Linear residuals)
#ifdef CF_F16S_ABLATION_BUILD
            if (g.ablate != 3 || v == 12345.678f)      // CF_F16S_ABLATE=3: no output stores (upper bound of what hiding them could give)
#endif
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, o_off[nt] + ochan, 0, 0);
            const float m = ok ? v : 0.f;
            ssum[r] += m;
            ssq[r] += m * m;
        }
    }
    }
#ifdef CF_F16S_ABLATION_BUILD
    const unsigned long long ph_e2 = F16S_CLK();
#endif
    if (do_stats) {
        xreduce16(ssum, lane);      // common.h: DPP + v_permlane16_swap, no LDS traffic
        xreduce16(ssq, lane);
        // workgroup-level combine in LDS (the staging buffers are free now), then ONE fp64 atomic pair per (group, workgroup):
        // thousands of workgroups adding to the same 8 groups of a sample would otherwise serialise at the memory side.
        float* red = reinterpret_cast<float*>(lds);  // [WM*32 channels][2]
        __syncthreads();                              // every wave is done reading the last LDS buffer
        if (tid < WM * 64) red[tid] = 0.f;
        __syncthreads();
        if ((lane & 1) == 0) {
            const int r = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            const int cl = (cw % WM) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // channel inside the workgroup's block
            atomicAdd(&red[2 * cl], ssum[0]);
            atomicAdd(&red[2 * cl + 1], ssq[0]);
        }
        __syncthreads();
        if (tid < WM * 32 && b0 < p.B) {
            const int cb = blockIdx.y * WM * 32;
            const int co = cb + tid;
            const int cpg = p.Cout / p.gn_groups;
            if (co < p.Cout && (tid == 0 || co % cpg == 0)) {
                int n = cpg - co % cpg;               // channels of this group from co on
                if (n > WM * 32 - tid) n = WM * 32 - tid;
                if (n > p.Cout - co) n = p.Cout - co;
                float s1 = 0.f, s2 = 0.f;
                for (int j = 0; j < n; ++j) { s1 += red[2 * (tid + j)]; s2 += red[2 * (tid + j) + 1]; }
                double* w = p.gn_ws + 2L * ((long)b0 * p.gn_groups + co / cpg);
                atomicAdd(w, (double)s1);
                atomicAdd(w + 1, (double)s2);
            }
        }
    }
    CF_CLOCK_END(g_clock_f16s);
#ifdef CF_F16S_ABLATION_BUILD
    if (lane == 0 && ph_t1 && (blockIdx.x & 63) == 5) {   // a sample of the workgroups: same-address atomics from every wave would dominate the run
        const unsigned long long t3 = F16S_CLK();
        atomicAdd(&g_f16s_phase[0], ph_t1 - ph_t0);
        atomicAdd(&g_f16s_phase[1], (ph_t2 - ph_t1) - ph_stage - ph_bar);
        atomicAdd(&g_f16s_phase[2], ph_stage);
        atomicAdd(&g_f16s_phase[3], ph_bar);
        atomicAdd(&g_f16s_phase[4], t3 - ph_t2);
        atomicAdd(&g_f16s_phase[5], 1ULL);
        atomicAdd(&g_f16s_phase[6], ph_e1 - ph_t2);
        atomicAdd(&g_f16s_phase[7], ph_e2 - ph_e1);
        atomicAdd(&g_f16s_phase[8], t3 - ph_e2);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// cf_conv_terms: per-thread product mode of the f16-split convolutions launched from this thread (3 = hi/lo split, 1 = hi x hi only)
static thread_local int t_conv_terms = 3;
int conv_terms() { return t_conv_terms; }

static int f16s_deint() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CF_F16S_DEINT"); v = e ? atoi(e) : 1; }
    return v;
}

// CF_F16S_VEC=0 keeps the scalar (one dword per lane and channel) staging everywhere
static int f16s_vec() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CF_F16S_VEC");
        v = e ? atoi(e) : 1;
    }
    return v;
}

template <int KH, int KW, int CK, int WM, int NTW, int MAXT, int NLW, int NW, int VEC, int PRE = 0, int WL = 0>
static int launch_f16s_v(const ConvParams& p, F16sGeom g, const _Float16* wpk, hipStream_t s);

// picks the vector-staging instantiation when the layer qualifies (stride 1, W % 4 == 0, 16-byte aligned inputs, VT 4x4 tasks per
// staging thread), else the scalar one.  VT = 1 for the 3x3 shapes; the 1x5 / 5x1 shapes (CK = 32, wider patches) take VT = 2.
template <int KH, int KW, int CK, int WM, int NTW, int MAXT, int NLW = 0, int NW = 4>
static int launch_f16s(const ConvParams& p, F16sGeom g, const _Float16* wpk, hipStream_t s) {
    constexpr int KHW = KH * KW;
    g.NQ = 0;
    // (1x1 layers measured 6-17 % slower with it -- their scalar loads are already whole rows -- so spatial kernels only)
    if (NLW == 0 && KHW > 1 && f16s_vec() && p.stride == 1 && (p.W & 3) == 0 && (g.TW & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.x1) | reinterpret_cast<uintptr_t>(p.x2)) & 15) == 0) {
        const int a = (-p.pad_w) & 3;                       // ix_org mod 4 (tile origins are multiples of 4)
        const int nq = ((a + g.PW - 1) >> 2) + 1;
        const int tasks = (CK / 4) * g.NIMG * g.PH * nq;
        g.NQ = nq;
        if constexpr (KHW == 9) {
            if (tasks <= 64 * NW) {
                // the stride-1 3x3 four-wave shapes are also built with the deferred input normalisation; not the 8-wave shape of the small
                // maps (< 1024 workgroups): its short workgroups lose more to the table fill + barrier than the apply pass costs
                // (256 channels at 32x32, B = 32: 192 us vs 120 + 15 us, tools/prenorm_ab.py)
                if constexpr (MAXT == 2 && NLW == 0 && NW == 4) {
                    if (p.in_norm && g.NIMG == 1 && p.C2 == 0 && (reinterpret_cast<uintptr_t>(p.in_norm) & 3) == 0)
                        return launch_f16s_v<KH, KW, CK, WM, NTW, MAXT, NLW, NW, 1, 1>(p, g, wpk, s);
                }
                if (p.in_norm) { set_error("conv_f16s: deferred input normalisation is not built for this layer shape"); return CF_ERR_ARG; }
                return launch_f16s_v<KH, KW, CK, WM, NTW, MAXT, NLW, NW, 1>(p, g, wpk, s);
            }
            // (two tasks per thread for the (2 TH + 1) x (2 TW + 1) patches of the stride-2 shapes: measured equal or 2 % slower, not built)
        } else if constexpr (KHW > 1) {
            if (!p.in_norm && tasks <= 2 * 64 * NW) return launch_f16s_v<KH, KW, CK, WM, NTW, MAXT, NLW, NW, 2>(p, g, wpk, s);
        }
        g.NQ = 0;
    }
    if (p.in_norm) { set_error("conv_f16s: deferred input normalisation needs the vector staging path (3x3, stride 1, W % 4 == 0)"); return CF_ERR_ARG; }
    return launch_f16s_v<KH, KW, CK, WM, NTW, MAXT, NLW, NW, 0>(p, g, wpk, s);
}

template <int KH, int KW, int CK, int WM, int NTW, int MAXT, int NLW, int NW, int VEC, int PRE, int WL>
static int launch_f16s_v(const ConvParams& p, F16sGeom g, const _Float16* wpk, hipStream_t s) {
    constexpr int REC = CK * 4 + 16;
    constexpr int WGRP = (KH * KW == 9) ? 3 : ((KH * KW == 1) ? CK / 16 : 5);      // k-steps per LDS weight slot (kernel: G)
    constexpr int NSTAGE = NLW ? 64 * NLW : 64 * NW;
    const int nrec = g.NIMG * g.PH * g.PW;
    { static int ab = -1; if (ab < 0) { const char* e = getenv("CF_F16S_ABLATE"); ab = e ? atoi(e) : 0; } g.ablate = ab; }
    g.m_tx = f16s_magic(g.tiles_x);
    g.m_ty = f16s_magic(g.tiles_y);
    g.m_percg = f16s_magic(g.NIMG * g.PH * g.NQ);
    g.m_phnq = f16s_magic(g.PH * g.NQ);
    g.m_nq = f16s_magic(g.NQ);
    g.m_nrec = f16s_magic(nrec);
    g.m_phpw = f16s_magic(g.PH * g.PW);
    g.m_pw = f16s_magic(g.PW);
    g.m_thtw = f16s_magic(g.TH * g.TW);
    g.m_tw = f16s_magic(g.TW);
    if ((long)g.tiles_x * g.tiles_y * g.bgroups >= (1L << 31) / (g.tiles_x > g.tiles_y ? g.tiles_x : g.tiles_y)) {
        set_error("conv_f16s: grid too large for the index decode");
        return CF_ERR_ARG;
    }
    if (!VEC && (nrec * (CK / 8) + NSTAGE - 1) / NSTAGE > MAXT) {
        set_error("conv_f16s: staging tasks exceed MAXT");
        return CF_ERR_ARG;
    }
    size_t lds_bytes = (size_t)2 * g.NIMG * g.PH * g.PWR * REC + (PRE ? (size_t)3 * g.nchunk * CK * sizeof(float) : 0) +
                             (WL ? (size_t)2 * WM * WGRP * 2 * 1024 : 0);
#ifdef CF_F16S_ABLATION_BUILD
    { static long pad = -1; if (pad < 0) { const char* e = getenv("CF_F16S_LDSPAD"); pad = e ? atol(e) : 0; }   // timing builds: fewer resident workgroups
      if (pad > (long)lds_bytes) lds_bytes = (size_t)pad; }
#endif
    if ((double)g.NIMG * p.out_ctotal * p.Ho * p.Wo * (p.scatter2x2 ? 4.0 : 1.0) * 4.0 >= 1073741824.0) {
        set_error("conv_f16s: one workgroup's output samples must stay below 1 GiB (32-bit store offsets)");
        return CF_ERR_ARG;
    }
    if (lds_bytes > 160 * 1024) {
        set_error("conv_f16s: LDS tile too large");
        return CF_ERR_ARG;
    }
    if (p.probe) return CF_OK;
    auto kern = p.terms == 1 ? conv_f16s_kernel<KH, KW, CK, WM, NTW, MAXT, NLW, NW, VEC, PRE, WL, 1> : conv_f16s_kernel<KH, KW, CK, WM, NTW, MAXT, NLW, NW, VEC, PRE, WL, 3>;
    static bool attr_set[64][2] = {};       // per device and variant
    const int dev = current_device_slot();
    if (!attr_set[dev][p.terms == 1]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev][p.terms == 1] = true;
    }
    dim3 grid((unsigned)(g.tiles_x * g.tiles_y * g.bgroups), (unsigned)((p.Cout + 32 * WM - 1) / (32 * WM)));
    dim3 block(64 * NW + 64 * NLW);
    const double flops = 2.0 * (double)p.B * p.Ho * p.Wo * p.Cout * (p.C1 + p.C2) * p.KH * p.KW;
    hipEvent_t e0, e1;
    if (profile_on() && profile_events(PK_CONV_F16S, flops, &e0, &e1))
        hipExtLaunchKernelGGL(kern, grid, block, lds_bytes, s, e0, e1, 0, p, g, wpk);
    else
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, s, p, g, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("conv_f16s launch failed: ") + hipGetErrorString(e));
        return CF_ERR_LAUNCH;
    }
    return CF_OK;
}

// The weights-through-LDS shapes (3x3, stride 1, vector staging, one sample per workgroup, 8 waves).  Returns -1 when the layer does not
// qualify (the caller then takes the register-fragment shapes).
template <int WM, int NTW>
static int launch_f16s_wl(const ConvParams& p, F16sGeom g, const _Float16* wpk, hipStream_t s, bool* probe_single) {
    constexpr int CK = 16, NW = 8;
    if (!(f16s_vec() && p.stride == 1 && (p.W & 3) == 0 && (g.TW & 3) == 0 && g.NIMG == 1 &&
          ((reinterpret_cast<uintptr_t>(p.x1) | reinterpret_cast<uintptr_t>(p.x2)) & 15) == 0))
        return -1;
    const int a = (-p.pad_w) & 3;
    const int nq = ((a + g.PW - 1) >> 2) + 1;
    if ((CK / 4) * g.NIMG * g.PH * nq > 64 * NW) return -1;
    g.NQ = nq;
    if (probe_single) { *probe_single = true; return CF_OK; }      // geometry probe: the layer takes this shape, one sample per workgroup
    if (p.in_norm) {
        if (p.C2 != 0 || (reinterpret_cast<uintptr_t>(p.in_norm) & 3) != 0) return -1;
        return launch_f16s_v<3, 3, CK, WM, NTW, 2, 0, NW, 1, 1, 1>(p, g, wpk, s);
    }
    return launch_f16s_v<3, 3, CK, WM, NTW, 2, 0, NW, 1, 0, 1>(p, g, wpk, s);
}

// CF_F16S_WL=0 keeps the weight fragments on the L1/L2 -> register path everywhere (A/B knob)
static int f16s_wl() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CF_F16S_WL"); v = e ? atoi(e) : 1; }
    return v;
}

// kernel shapes of the f16-split kernel: 3x3 pad 1 and 1x1 pad 0 at stride 1 / 2, and the separable 1x5 (pad 0,2) / 5x1 (pad 2,0)
// convolutions of RAFT's SepConvGRU at stride 1
static int f16s_kind(const ConvParams& p) {
    if (p.KH == 3 && p.KW == 3 && p.pad_h == 1 && p.pad_w == 1 && (p.stride == 1 || p.stride == 2)) return 33;
    if (p.KH == 1 && p.KW == 1 && p.pad_h == 0 && p.pad_w == 0 && (p.stride == 1 || p.stride == 2)) return 11;
    if (p.KH == 1 && p.KW == 5 && p.pad_h == 0 && p.pad_w == 2 && p.stride == 1) return 15;
    if (p.KH == 5 && p.KW == 1 && p.pad_h == 2 && p.pad_w == 0 && p.stride == 1) return 51;
    return 0;
}

// One sample of each input must stay below 2 GiB (32-bit buffer offsets); larger batches are split by launch_conv_f16s itself.
bool conv_f16s_supported(const ConvParams& p) {
    if (p.w_bstride) return false;
    if (!f16s_kind(p)) return false;
    const long HW = (long)p.H * p.W;
    if ((long)p.C1 * HW * 4 >= (1L << 31) || (long)p.C2 * HW * 4 >= (1L << 31)) return false;
    // the epilogue's buffer-resource stores address the output samples of one workgroup with 32-bit offsets below 1 GiB (launch_f16s_v):
    // up to 8 whole images per workgroup when an image fits one tile (<= 256 pixels), else one
    const double osample = (double)p.out_ctotal * p.Ho * p.Wo * (p.scatter2x2 ? 4.0 : 1.0) * 4.0;
    if (osample * ((long)p.Ho * p.Wo <= 256 ? 8.0 : 1.0) >= 1073741824.0) return false;
    return true;
}

// CF_F16S_LOADERS=1 selects the wave-specialised variants (4 MFMA waves + 2-3 loader waves).  Measured (profiles/,
// DESIGN.md 5.1): they win on stride-2 and on small maps (+10-24 %) but lose 15-25 % on the large 3x3 layers that dominate
// the step, where the kernel is limited by bytes in flight rather than by the vmcnt coupling; default off.
static int f16s_loader_waves() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CF_F16S_LOADERS");
        v = e ? atoi(e) : 0;
    }
    return v;
}

// CF_F16S_WIDE=0 disables the 128-channel workgroups used for Cout % 128 == 0 (the input patch is staged once per 128 output
// channels instead of once per 64); 1 = automatic choice between the two 128-channel shapes, 2 / 3 force the 4-wave / 8-wave one.
static int f16s_wide() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CF_F16S_WIDE");
        v = e ? atoi(e) : 1;
    }
    return v;
}

// CF_F16S_WIDE_PAD=0: only Cout % 128 == 0 takes the 128-channel workgroup shapes.  Default: also a Cout whose last 128-channel block is at
// least three quarters full (the U-Net's 480 = 3 x 128 + 96): the packed weights are padded to whole 128-channel blocks anyway
// (pack_conv_weight_f16s: an even number of 64-channel pairs), the rows past Cout are zero weights whose stores the epilogue drops.
static int f16s_wide_pad() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CF_F16S_WIDE_PAD"); v = e ? atoi(e) : 1; }
    return v;
}
static bool f16s_cout_wide(int cout) { return cout % 128 == 0 || (f16s_wide_pad() && cout > 128 && cout % 128 >= 96); }

static int f16s_small_tile() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CF_F16S_SMALL_TILE");
        v = e ? atoi(e) : 1;  // default: 128-pixel workgroup tiles (4 waves/SIMD hide the staging latency; A/B in DESIGN.md)
    }
    return v;
}

static int launch_conv_f16s_impl(const ConvParams& p, const _Float16* wpk, hipStream_t s, bool* one_sample_per_wg);

// Runs the convolution on a batch whose inputs fit 32-bit buffer offsets; when p.gn_ws is set the workspace ends up holding the
// GroupNorm statistics of the output -- from the fused epilogue when every workgroup covers a single sample, otherwise from the
// separate statistics pass.
static int launch_conv_f16s_part(const ConvParams& p, const _Float16* wpk, hipStream_t s) {
    // short-K layers on large maps (<= 64 output channels, 3x3, stride 1, >= 1024 tiles): the persistent pipelined kernel of conv_stream.hip,
    // same arithmetic and packed weights (CF_CONV_STREAM=0 keeps this file's shapes: A/B knob)
    if (conv_stream_applicable(p)) {
        if (p.probe) return CF_OK;
        if (p.gn_ws && !p.gn_prezeroed &&
            hipMemsetAsync(p.gn_ws, 0, sizeof(double) * 2 * (size_t)p.B * p.gn_groups, s) != hipSuccess) { set_error("conv_stream: memset failed"); return CF_ERR_LAUNCH; }
        return launch_conv_stream(p, wpk, s);
    }
    if (!p.gn_ws) return launch_conv_f16s_impl(p, wpk, s, nullptr);
    bool fusable = false;
    launch_conv_f16s_impl(p, nullptr, s, &fusable);  // geometry probe only
    ConvParams q = p;
    // (the transposed convolution's scatter qualifies too: its GEMM rows are co * 4 + dy * 2 + dx, so a group of Cout / groups output
    // channels is a run of 4 * Cout / groups consecutive rows -- the same `co / cpg` with cpg = p.Cout / groups = 4 * Cout / groups)
    if (fusable) {
        // the fused statistics accumulate with atomics: the workspace must start at zero (gn_prezeroed: the caller hands out slices
        // of a pool it zeroed with ONE memset -- the per-launch memsets were 1.2 % of the step)
        if (!p.gn_prezeroed &&
            hipMemsetAsync(p.gn_ws, 0, sizeof(double) * 2 * (size_t)p.B * p.gn_groups, s) != hipSuccess) { set_error("conv_f16s: memset failed"); return CF_ERR_LAUNCH; }
        return launch_conv_f16s_impl(q, wpk, s, nullptr);
    }
    q.gn_ws = nullptr;
    int rc = launch_conv_f16s_impl(q, wpk, s, nullptr);
    if (rc != CF_OK) return rc;
    const int up = p.scatter2x2 ? 2 : 1;
    const int cout = p.scatter2x2 ? p.Cout / 4 : p.Cout;
    if (p.out_coff != 0 || p.out_ctotal != cout) { set_error("conv_f16s: GroupNorm statistics need a dense output tensor"); return CF_ERR_ARG; }
    return launch_gn_stats(p.out, p.gn_ws, p.B, cout, p.Ho * up * p.Wo * up, p.gn_groups, s);
}

// The kernel addresses its inputs with 32-bit buffer offsets (< 2 GiB per descriptor).  A batch whose input tensors are larger is
// cut into sub-batches HERE (same kernel, same numbers, a few more launches) instead of being handed to another kernel: the batch
// axis is the outermost one of every operand, so a sub-batch is a pointer offset.
// Samples per sub-batch: the batch is cut into the FEWEST parts that fit and the parts are made equal (the last one at most parts - 1
// samples smaller) -- a greedy cut could leave a remainder of one sample, for which the dispatch picks another kernel shape than for the
// large parts (one that e.g. lacks the deferred input normalisation although the capability probe, asked with the whole batch, said yes).
static long f16s_sub_batch(const ConvParams& p) {
    const long HW = (long)p.H * p.W;
    const long per1 = (long)p.C1 * HW * 4, per2 = (long)p.C2 * HW * 4;
    const long lim = (1L << 31) - 1;
    const long nb = lim / (per1 > per2 ? per1 : per2);
    if (nb < 1) return 0;
    if (p.B <= nb) return p.B;
    const long parts = (p.B + nb - 1) / nb;
    return (p.B + parts - 1) / parts;
}

int launch_conv_f16s(const ConvParams& p, const _Float16* wpk, hipStream_t s) {
    if (conv_stream_applicable(p)) return launch_conv_f16s_part(p, wpk, s);      // per-sample buffer resources: no 2 GiB limit on the batch
    const long HW = (long)p.H * p.W;
    const long nb = f16s_sub_batch(p);
    if (nb < 1) { set_error("conv_f16s: one sample of the input exceeds 2 GiB"); return CF_ERR_ARG; }
    if (p.B <= nb) return launch_conv_f16s_part(p, wpk, s);
    const int up = p.scatter2x2 ? 2 : 1;
    const long HoWo = (long)p.Ho * p.Wo;
    for (long b0 = 0; b0 < p.B; b0 += nb) {
        ConvParams q = p;
        q.B = (int)(p.B - b0 < nb ? p.B - b0 : nb);
        q.x1 = p.x1 + b0 * p.C1 * HW;
        if (p.x2) q.x2 = p.x2 + b0 * p.C2 * HW;
        if (p.res) q.res = p.res + b0 * p.Cout * HoWo;
        q.out = p.out + b0 * p.out_ctotal * HoWo * up * up;
        if (p.gn_ws) q.gn_ws = p.gn_ws + 2 * b0 * p.gn_groups;
        if (p.in_norm) q.in_norm = p.in_norm + b0 * 3 * p.C1;
        int rc = launch_conv_f16s_part(q, wpk, s);
        if (rc != CF_OK) return rc;
    }
    return CF_OK;
}

static int launch_conv_f16s_impl(const ConvParams& p, const _Float16* wpk, hipStream_t s, bool* one_sample_per_wg) {
    const int kind = f16s_kind(p);
    const bool k3 = kind == 33;
    const bool sep = kind == 15 || kind == 51;           // 1x5 / 5x1
    const bool spatial = k3 || sep;
    const int CK = k3 ? 16 : 32;
    const bool narrow = p.Cout <= 32;
    const bool small = f16s_small_tile() != 0;
    const bool s2 = k3 && p.stride == 2;
    // Cout = 3 x 128 + 96 and the like ride on the 128-channel shapes with a partly empty last m-tile -- where the four-wave shape runs
    // (>= 1024 workgroups); a small launch keeps the 64-channel shapes (and with them the deferred input normalisation)
    bool cout_wide = f16s_cout_wide(p.Cout);
    if (cout_wide && p.Cout % 128 != 0 && !s2) {
        const int tw = p.Wo < 32 ? p.Wo : 32;
        int th = 128 / tw;
        if (th > p.Ho) th = p.Ho;
        const long nwg = (long)((p.Wo + tw - 1) / tw) * ((p.Ho + th - 1) / th) * p.B * ((p.Cout + 127) / 128);
        if (nwg < 1024) cout_wide = false;
    }
    const bool wide = small && !s2 && f16s_wide() && cout_wide && f16s_loader_waves() == 0;
    // n-tiles (of 32 output pixels) per workgroup
    const int NT_WG = s2 ? ((small && !narrow) ? 2 : 4) : ((small || sep) ? 4 : 8);
    F16sGeom g;
    auto geometry = [&](int npx) {
    g.TW = p.Wo < 32 ? p.Wo : 32;
    g.TH = npx / g.TW;
    if (g.TH > p.Ho) g.TH = p.Ho;
    g.NIMG = 1;
    if (g.TH == p.Ho && g.TW == p.Wo) {
        g.NIMG = npx / (g.TH * g.TW);
        if (g.NIMG > 8) g.NIMG = 8;
        if (g.NIMG > p.B) g.NIMG = p.B;
        if (g.NIMG < 1) g.NIMG = 1;
    }
    if (spatial) {
        g.pstep = 1;
        g.ostep = p.stride;
        g.PH = (g.TH - 1) * p.stride + p.KH;
        g.PW = (g.TW - 1) * p.stride + p.KW;
    } else {
        g.pstep = p.stride;
        g.ostep = 1;
        g.PH = g.TH;
        g.PW = g.TW;
    }
    // keep the staging work within the per-thread task budget of the variant (MAXT x 256 eight-channel tasks)
    {
        const bool ws = small && f16s_loader_waves() > 0 && !sep;   // 128 staging threads instead of 256
        const int maxt = sep ? 4 : (ws ? 4 : (s2 ? ((small && !narrow) ? 3 : 5) : (small ? 2 : 4)));
        const int nstage = ws ? ((s2 && !narrow) ? 192 : 128) : ((wide && !sep) ? 512 : 256);
        while (g.NIMG > 1 && (g.NIMG * g.PH * g.PW * (CK / 8) + nstage - 1) / nstage > maxt) --g.NIMG;
    }
    // Stride 2: output-pixel lanes read every second input column; with plain row-major records (80 B apart) their 160-byte lane
    // stride lands 16 lanes on 8 of the 16 LDS slots (2-way conflicts on every B-fragment read).  The patch rows are therefore stored
    // de-interleaved -- even columns, then odd columns -- so a tap's 32 lanes read consecutive records again.
    g.PWR = g.PW; g.pwh = 0; g.kx1 = 1; g.kx2 = 2;
    if (s2 && f16s_deint()) { g.pwh = (g.PW + 1) / 2; g.PWR = 2 * g.pwh; g.kx1 = g.pwh; g.kx2 = 1; }
    g.tiles_x = (p.Wo + g.TW - 1) / g.TW;
    g.tiles_y = (p.Ho + g.TH - 1) / g.TH;
    g.bgroups = (p.B + g.NIMG - 1) / g.NIMG;
    // split-aware chunking: x1's channels are padded to whole chunks (the packed weights carry zeros there), then x2's follow
    g.c1_pad = p.C2 > 0 ? ((p.C1 + CK - 1) / CK) * CK : (1 << 30);
    g.nchunk = p.C2 > 0 ? g.c1_pad / CK + (p.C2 + CK - 1) / CK : (p.C1 + CK - 1) / CK;
    };
    // The weights-through-LDS shapes come first (3x3, stride 1, Cout not a multiple of 128): 8 waves, 256 output pixels x 64 (or 32)
    // channels; they need one sample per workgroup and the vector staging path, else the register-fragment shapes below take the layer.
    // Measured (profiles/r02_conv_weight_path.md, B = 16): 64 -> 64 at 256x256 234 -> 263 TF, 32 -> 32 195 -> 205, 81 -> 64 208 -> 222; the
    // 128-channel form of it (4 m-tiles x 2 pixel tiles per wave, 8 waves) LOSES to the four-wave shape whose fragments feed 12 MFMAs
    // (128 -> 128 at 128x128: 294 vs 343 TF) and is not dispatched (CF_F16S_WL=2 forces it for A/B runs).
    if (k3 && p.stride == 1 && f16s_wl() && small && f16s_loader_waves() == 0 && (!cout_wide || f16s_wl() == 2)) {
        const bool w128 = cout_wide;
        geometry(w128 ? 128 : 256);
        if (g.NIMG == 1) {
            const int rc = w128 ? launch_f16s_wl<4, 2>(p, g, one_sample_per_wg ? nullptr : wpk, s, one_sample_per_wg)
                                : (narrow ? launch_f16s_wl<1, 1>(p, g, one_sample_per_wg ? nullptr : wpk, s, one_sample_per_wg)
                                          : launch_f16s_wl<2, 2>(p, g, one_sample_per_wg ? nullptr : wpk, s, one_sample_per_wg));
            if (rc != -1) return rc;
        }
    }
    geometry(NT_WG * 32);
    if (one_sample_per_wg) {  // geometry probe
        *one_sample_per_wg = g.NIMG == 1;
        return CF_OK;
    }
    if (sep) {
        // RAFT's SepConvGRU (384 -> 128 / 256 at 1/8 resolution): CK = 32 (10 k-steps per chunk keep the 2-slot fragment ring static);
        // 128-pixel tiles; 128-channel workgroups when Cout allows (8 waves: 4 m-tiles x 2 pixel groups), else 64-channel ones
        if (kind == 15) return (p.Cout % 128 == 0) ? launch_f16s<1, 5, 32, 4, 2, 4, 0, 8>(p, g, wpk, s) : launch_f16s<1, 5, 32, 2, 2, 4>(p, g, wpk, s);
        return (p.Cout % 128 == 0) ? launch_f16s<5, 1, 32, 4, 2, 4, 0, 8>(p, g, wpk, s) : launch_f16s<5, 1, 32, 2, 2, 4>(p, g, wpk, s);
    }
    if (wide && k3) {
        // Two 128-channel shapes.  Four waves, each ONE m-tile x FOUR pixel tiles: a weight fragment feeds 12 MFMAs instead of 6, which
        // halves the per-wave weight re-reads from L1/L2 -- the resource the time stamps and the persistent-kernel experiment pointed at
        // (+9-15 % on the 128- and 256-channel layers at 64x64 and above); 148 VGPRs (no scratch under a 2-workgroup launch bound; a bound of 3 made the allocator spill 168 B/lane for the same speed), 3 waves/SIMD.  With few workgroups (32x32 maps)
        // the 8-wave shape (4 m-tiles x 2 pixel groups, 4 waves/SIMD) keeps more of the chip busy.  CF_F16S_WIDE=2 / 3 force one.
        const long nwg = (long)g.tiles_x * g.tiles_y * g.bgroups * ((p.Cout + 127) / 128);
        const int mode = f16s_wide();
        const bool four = mode == 2 || (mode == 1 && nwg >= 1024 && g.NIMG == 1);
        if (four && (g.NIMG * g.PH * g.PW * (CK / 8) + 255) / 256 <= 2) return launch_f16s<3, 3, 16, 4, 4, 2, 0, 4>(p, g, wpk, s);
        return launch_f16s<3, 3, 16, 4, 2, 2, 0, 8>(p, g, wpk, s);
    }
    if (wide) {   // 1x1: same choice between the two 128-channel shapes
        const long nwg = (long)g.tiles_x * g.tiles_y * g.bgroups * ((p.Cout + 127) / 128);
        static int k1four = -1;
        if (k1four < 0) { const char* e = getenv("CF_F16S_K1FOUR"); k1four = e ? atoi(e) : 1; }
        if (k1four && nwg >= 1024 && g.NIMG == 1 && (g.NIMG * g.PH * g.PW * (CK / 8) + 255) / 256 <= 2) return launch_f16s<1, 1, 32, 4, 4, 2, 0, 4>(p, g, wpk, s);
        return launch_f16s<1, 1, 32, 4, 2, 2, 0, 8>(p, g, wpk, s);
    }
    if (small && f16s_loader_waves() > 0) {
        if (narrow && !s2) return k3 ? launch_f16s<3, 3, 16, 1, 1, 4, 2>(p, g, wpk, s) : launch_f16s<1, 1, 32, 1, 1, 4, 2>(p, g, wpk, s);
        if (s2 && !narrow) return launch_f16s<3, 3, 16, 2, 1, 4, 3>(p, g, wpk, s);
        if (!narrow) return k3 ? launch_f16s<3, 3, 16, 2, 2, 4, 2>(p, g, wpk, s) : launch_f16s<1, 1, 32, 2, 2, 4, 2>(p, g, wpk, s);
    }
    if (small && narrow && !s2) return k3 ? launch_f16s<3, 3, 16, 1, 1, 2>(p, g, wpk, s) : launch_f16s<1, 1, 32, 1, 1, 2>(p, g, wpk, s);
    // stride 2 with Cout % 128 == 0: four m-tiles x two pixel tiles per wave -- every weight fragment is loaded once per workgroup and
    // feeds 6 MFMAs (the 64-channel shape below: loaded twice, 3 MFMAs each)
    if (small && s2 && !narrow && f16s_cout_wide(p.Cout) && f16s_wide()) return launch_f16s<3, 3, 16, 4, 2, 3>(p, g, wpk, s);
    if (small && s2 && !narrow) return launch_f16s<3, 3, 16, 2, 1, 3>(p, g, wpk, s);
    if (small && !narrow) return k3 ? launch_f16s<3, 3, 16, 2, 2, 2>(p, g, wpk, s) : launch_f16s<1, 1, 32, 2, 2, 2>(p, g, wpk, s);
    if (k3) {
        if (p.stride == 1) return narrow ? launch_f16s<3, 3, 16, 1, 2, 4>(p, g, wpk, s) : launch_f16s<3, 3, 16, 2, 4, 4>(p, g, wpk, s);
        return narrow ? launch_f16s<3, 3, 16, 1, 1, 5>(p, g, wpk, s) : launch_f16s<3, 3, 16, 2, 2, 5>(p, g, wpk, s);
    }
    return narrow ? launch_f16s<1, 1, 32, 1, 2, 4>(p, g, wpk, s) : launch_f16s<1, 1, 32, 2, 4, 4>(p, g, wpk, s);
}

}  // namespace cf

using namespace cf;

CF_CLOCK_READER(cf_debug_clock_f16s, cf::g_clock_f16s)

extern "C" int cf_conv_terms(int terms) {
    const int prev = t_conv_terms;
    if (terms == 1 || terms == 3) t_conv_terms = terms;
    return prev;
}

extern "C" int cf_conv2d_f16s(const float* x1, int C1, const float* x2, int C2, const void* wpk, const float* bias, const float* res,
                              float* out, int out_ctotal, int out_coff, int B, int H, int W, int Cout, int KH, int KW, int stride,
                              int pad_h, int pad_w, int act, float alpha, double* gn_ws, int gn_groups, void* stream) {
    CF_REQUIRE(x1 && wpk && out, "null pointer");
    CF_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "bad channel split C1=%d C2=%d", C1, C2);
    CF_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0, "bad shape B=%d H=%d W=%d Cout=%d", B, H, W, Cout);
    CF_REQUIRE(out_coff >= 0 && out_coff + Cout <= out_ctotal, "output channel slice out of range");
    CF_REQUIRE(act >= CF_ACT_NONE && act <= CF_ACT_SIGMOID, "bad activation %d", act);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(wpk) & 15) == 0, "packed weights must be 16-byte aligned");
    ConvParams p;
    p.x1 = x1; p.x2 = C2 ? x2 : nullptr; p.wt = nullptr; p.bias = bias; p.res = res; p.out = out; p.w_bstride = 0;
    p.C1 = C1; p.C2 = C2; p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride;
    p.pad_h = pad_h; p.pad_w = pad_w; p.Ho = (H + 2 * pad_h - KH) / stride + 1; p.Wo = (W + 2 * pad_w - KW) / stride + 1;
    p.out_ctotal = out_ctotal; p.out_coff = out_coff; p.act = act; p.alpha = alpha; p.scatter2x2 = 0;
    p.gn_ws = gn_ws; p.gn_groups = gn_groups < 0 ? -gn_groups : gn_groups; p.gn_prezeroed = gn_groups < 0;
    p.terms = conv_terms();
    CF_REQUIRE(!gn_ws || (p.gn_groups > 0 && Cout % p.gn_groups == 0 && out_coff == 0 && out_ctotal == Cout), "bad GroupNorm statistics request");
    CF_REQUIRE(p.Ho > 0 && p.Wo > 0, "empty output");
    CF_REQUIRE(conv_f16s_supported(p), "unsupported configuration for the f16-split kernel (3x3 pad 1 or 1x1 pad 0 at stride 1/2, 1x5 pad (0,2) or 5x1 pad (2,0) at stride 1; one sample < 2 GiB)");
    return launch_conv_f16s(p, reinterpret_cast<const _Float16*>(wpk), as_stream(stream));
}

// cf_conv2d_f16s whose input x is a RAW convolution output with its InstanceNorm / GroupNorm + LeakyReLU deferred to this consumer
// (the second convolution of a Generic_UNet stage): in_norm = float [B][3][C] {mean, scale, shift} from cf_group_norm_coef, applied
// as lrelu((x - mean) * scale + shift, in_slope) while the tile is staged -- the producer's apply pass (8 B per element) disappears.
// Built for 3x3 / stride 1 / pad 1 layers on the vector staging path (W % 4 == 0, one sample per workgroup, single input);
// cf_conv2d_f16s_prenorm_ok answers whether a shape qualifies (1) or not (0).
static void prenorm_params(ConvParams& p, const float* x, int C, const float* bias, float* out, int B, int H, int W, int Cout, float alpha,
                           double* gn_ws, int gn_groups, const float* in_norm, float in_slope) {
    p.x1 = x; p.x2 = nullptr; p.wt = nullptr; p.bias = bias; p.res = nullptr; p.out = out; p.w_bstride = 0;
    p.C1 = C; p.C2 = 0; p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.KH = 3; p.KW = 3; p.stride = 1;
    p.pad_h = 1; p.pad_w = 1; p.Ho = H; p.Wo = W; p.out_ctotal = Cout; p.out_coff = 0; p.act = CF_ACT_NONE; p.alpha = alpha; p.scatter2x2 = 0;
    p.gn_ws = gn_ws; p.gn_groups = gn_groups < 0 ? -gn_groups : gn_groups; p.gn_prezeroed = gn_groups < 0;
    p.in_norm = in_norm; p.in_slope = in_slope;
    p.terms = conv_terms();
}

extern "C" int cf_conv2d_f16s_prenorm_ok(int B, int C, int H, int W, int Cout) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    ConvParams p;
    float* dummy = reinterpret_cast<float*>(uintptr_t(256));      // never dereferenced: probe mode launches nothing
    prenorm_params(p, dummy, C, nullptr, dummy, B, H, W, Cout, 1.f, nullptr, 0, dummy, 0.01f);
    p.probe = 1;
    if (!conv_f16s_supported(p)) return 0;
    // the shapes are chosen per sub-batch (launch_conv_f16s): ask for the sizes that will actually be launched
    const long nb = f16s_sub_batch(p);
    if (nb < 1) return 0;
    const long last = B - (B - 1) / nb * nb;
    p.B = (int)nb;
    if (!conv_stream_applicable(p) && launch_conv_f16s_impl(p, reinterpret_cast<const _Float16*>(dummy), nullptr, nullptr) != CF_OK) return 0;
    if (last != nb) {
        p.B = (int)last;
        if (!conv_stream_applicable(p) && launch_conv_f16s_impl(p, reinterpret_cast<const _Float16*>(dummy), nullptr, nullptr) != CF_OK) return 0;
    }
    return 1;
}

extern "C" int cf_conv2d_f16s_prenorm(const float* x, int C, const float* in_norm, float in_slope, const void* wpk, const float* bias, float* out,
                                      int B, int H, int W, int Cout, float alpha, double* gn_ws, int gn_groups, void* stream) {
    CF_REQUIRE(x && wpk && out && in_norm, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Cout > 0, "bad shape B=%d C=%d H=%d W=%d Cout=%d", B, C, H, W, Cout);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(wpk) & 15) == 0, "packed weights must be 16-byte aligned");
    ConvParams p;
    prenorm_params(p, x, C, bias, out, B, H, W, Cout, alpha, gn_ws, gn_groups, in_norm, in_slope);
    CF_REQUIRE(!gn_ws || (p.gn_groups > 0 && Cout % p.gn_groups == 0), "bad GroupNorm statistics request");
    CF_REQUIRE(conv_f16s_supported(p), "unsupported configuration for the f16-split kernel");
    return launch_conv_f16s(p, reinterpret_cast<const _Float16*>(wpk), as_stream(stream));
}

extern "C" int cf_conv_transpose2d_k2s2_f16s(const float* x, const void* wpk, const float* bias, float* out, int out_ctotal,
                                             int out_coff, int B, int Cin, int H, int W, int Cout, float alpha, double* gn_ws, int gn_groups,
                                             void* stream) {
    CF_REQUIRE(x && wpk && out, "null pointer");
    CF_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "bad shape");
    CF_REQUIRE(out_coff >= 0 && out_coff + Cout <= out_ctotal, "output channel slice out of range");
    CF_REQUIRE((reinterpret_cast<uintptr_t>(wpk) & 15) == 0, "packed weights must be 16-byte aligned");
    ConvParams p;
    p.x1 = x; p.x2 = nullptr; p.wt = nullptr; p.bias = bias; p.res = nullptr; p.out = out; p.w_bstride = 0;
    p.C1 = Cin; p.C2 = 0; p.B = B; p.H = H; p.W = W; p.Cout = Cout * 4; p.KH = 1; p.KW = 1; p.stride = 1;
    p.pad_h = 0; p.pad_w = 0; p.Ho = H; p.Wo = W; p.out_ctotal = out_ctotal; p.out_coff = out_coff; p.act = CF_ACT_NONE;
    p.alpha = alpha; p.scatter2x2 = 1; p.gn_ws = gn_ws; p.gn_groups = gn_groups < 0 ? -gn_groups : gn_groups; p.gn_prezeroed = gn_groups < 0;
    p.terms = conv_terms();
    CF_REQUIRE(!gn_ws || (p.gn_groups > 0 && Cout % p.gn_groups == 0 && out_coff == 0 && out_ctotal == Cout), "bad GroupNorm statistics request");
    CF_REQUIRE(conv_f16s_supported(p), "unsupported configuration for the f16-split kernel (one sample of the input must stay below 2 GiB)");
    return launch_conv_f16s(p, reinterpret_cast<const _Float16*>(wpk), as_stream(stream));
}

#ifdef CF_F16S_ABLATION_BUILD
// timing builds only: read and reset the per-phase clock sums
extern "C" int cf_debug_f16s_phases(unsigned long long* out9) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out9, HIP_SYMBOL(cf::g_f16s_phase), 9 * sizeof(unsigned long long));
    unsigned long long z[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(cf::g_f16s_phase), z, sizeof(z));
    return 0;
}
#endif
