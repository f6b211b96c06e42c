// Persistent, software-pipelined sibling of conv_f16s_kernel for the layers whose K loop is too short to hide anything: 3x3 / stride 1 / pad 1
// convolutions to <= 64 output channels on large maps (the flow net's 64-channel level at 256x256, the U-Net's 32- and 64-channel levels).
// Same arithmetic as conv_f16s.hip -- f16 hi/lo 3-term split on v_mfma_f32_32x32x16_f16, fp32 accumulation, k order (16-channel chunk, tap,
// channel), same packed weights, same 80-byte LDS pixel records -- so its outputs are bit-identical to that kernel's.
//
// Why another kernel.  For these layers one workgroup of conv_f16s moves 151 KB (64 -> 64: a 10 x 34 x 64-channel patch in, 64 x 256 pixels out)
// for 4 chunks of MFMA work; at the CU's share of HBM (~14 B/clk) that is 21.6k clocks of memory time next to 27.6k clocks of MFMA time per
// pair of resident workgroups, and the measured rate (266 TF = 55 % of what the MFMA pipe sustains at that clock) is what T_mfma + T_hbm
// predicts: the first chunk's loads and the output stores of a workgroup overlap with nothing of its own, and two resident workgroups per CU
// that start together stay in step (profiles/r03_conv_stream.md; CF_F16S_ABLATE=3 in round 2: +27.6 % without the stores).  Here:
//   * ONE workgroup per CU (8 waves) is persistent and walks a band of tiles of its XCD; the 16-channel chunks of consecutive tiles form one
//     stream, so there is no per-tile prologue: while chunk s is multiplied, chunk s + 1 is converted into LDS and chunk s + 2 is in flight
//     (two register sets -> two chunks = ~44 KB of loads in flight per CU at any time);
//   * two accumulator sets: tile i's epilogue (scale, bias, GroupNorm statistics, 32 stores per wave) is interleaved tap by tap with the MFMAs
//     of tile i + 1's first chunk, and its stores drain while the next chunks compute;
//   * weights go through LDS as in the WL shapes of conv_f16s (one LDS-DMA per fragment and workgroup, none of them in a VGPR), but a whole
//     chunk (9 taps) per slot: ONE barrier per chunk, raw s_barrier + lgkmcnt(0) so that nothing in flight on the VM counter is drained;
//   * the deferred input normalisation (PRE) reads its {mean, scale, shift} table from LDS, double buffered by tile parity and prefetched with
//     the first chunk of a tile, because consecutive tiles of a workgroup can belong to different samples.
// Roofline: f16 MFMA (3 MFMAs per 16-deep k-step), with HBM a close second (144 / 72 flop per byte at 64 / 32 channels).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "conv.h"
#include "profile.h"

namespace cf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct StreamGeom {
    int tiles_x, tiles_y, ntiles;   // tiles of TH x 32 output pixels: per sample tiles_y x tiles_x, ntiles = B * tiles_y * tiles_x
    int nchunk, c1_chunks;          // 16-channel chunks (even); chunks < c1_chunks read x1, the others x2
};

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

constexpr int ST_NW = 8, ST_TH = 8, ST_TW = 32, ST_PW = 34, ST_REC = 80, ST_CK = 16, ST_NQ = 10;

template <int WM, int PRE>
__global__ void __launch_bounds__(64 * ST_NW) conv_stream_kernel(const ConvParams p, const StreamGeom g, const _Float16* __restrict__ wpk) {
    constexpr int NW = ST_NW, PW = ST_PW, REC = ST_REC, CK = ST_CK, NQ = ST_NQ;
    constexpr int NG = NW / WM;                  // n-tile groups (one per wave of an m-tile)
    constexpr int NTW = WM;                      // n-tiles per wave: 64 channels -> 2 m-tiles x 4 groups x 2, 32 channels -> 8 groups x 1 (a second
                                                 // n-tile per wave would need 16-row tiles and two staging tasks per thread: 64 more registers, spills)
    constexpr int TH = NG * NTW;                 // output rows of a tile (8): one 32-pixel n-tile per row
    static_assert(TH == ST_TH, "tile height");
    constexpr int PH = TH + 2;
    constexpr int NTASK = 4 * PH * NQ;           // staging tasks per chunk: 4 channel groups x patch rows x aligned column quads
    constexpr int VT = (NTASK + 64 * NW - 1) / (64 * NW);
    constexpr int PATCH = PH * PW * REC;         // bytes per patch slot
    constexpr int WSLOT = WM * 18 * 1024;        // bytes per weight slot: 9 taps x (hi, lo) fragments per m-tile
    constexpr int CT_SLOT = 2 * 64 * NW;         // floats per coefficient-table slot (PRE): [3][ctab_n] image, padded so that every thread stores its two
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const patch = lds;
    unsigned char* const wl = lds + 2 * PATCH;
    float* const red = reinterpret_cast<float*>(lds + 2 * PATCH + 2 * WSLOT);      // [2][WM * 32 channels][2]
    float* const biasl = red + 2 * WM * 64;                                         // [WM * 32]: the bias (no global load inside the pipelined loop)
    float* const ctab = biasl + WM * 32;                                            // [2][CT_SLOT] (PRE only)
    const int ctab_n = g.nchunk * CK;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int mtl = wave % WM, ngrp = wave / WM;

    // ---- this workgroup's tiles: XCD x (= blockIdx % 8) owns the contiguous band [ntiles * x / 8, ntiles * (x + 1) / 8) of the tile list (x
    // fastest, then y, then sample), its workgroups take the band's tiles round robin: neighbouring tiles run at about the same time on
    // the same XCD and share their halos in its L2
    const int xcd = blockIdx.x & 7, wgi = blockIdx.x >> 3, nwgx = gridDim.x >> 3;
    const int band_lo = (int)((long)g.ntiles * xcd / 8), band_hi = (int)((long)g.ntiles * (xcd + 1) / 8);
    if (band_lo + wgi >= band_hi) return;
    const int n_my = (band_hi - band_lo - wgi + nwgx - 1) / nwgx;
    const int HW = p.H * p.W;
    const unsigned HW4 = (unsigned)HW * 4u;
    const int nsteps = n_my * g.nchunk;
    auto decode = [&](int i, int& b, int& y0, int& x0) {
        int id = band_lo + wgi + i * nwgx;
        const int tx = id % g.tiles_x;
        id /= g.tiles_x;
        const int ty = id % g.tiles_y;
        b = id / g.tiles_y;
        y0 = ty * TH;
        x0 = tx * ST_TW;
    };

    // ================================================================================================ staging (all threads)
    // task = 4 channels x one aligned quad of 4 image columns of one patch row (conv_f16s' vector staging): 4 buffer_load_dwordx4 per chunk
    int v_lds[VT], v_py[VT], v_q[VT];
    unsigned v_c4[VT], v_c4hw[VT], v_mask[VT];
    bool v_has[VT];
#pragma unroll
    for (int t = 0; t < VT; ++t) {
        const int task = tid + t * 64 * NW;
        v_has[t] = task < NTASK;
        const int cg = task / (PH * NQ), r = task - cg * (PH * NQ);
        v_py[t] = r / NQ;
        v_q[t] = r - v_py[t] * NQ;
        const int pxp = 4 * v_q[t] - 3;                        // patch column of the quad's first pixel: the patch starts at image column x0 - 1
        v_mask[t] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v_mask[t] |= (v_has[t] && (unsigned)(pxp + k) < (unsigned)PW) ? (1u << k) : 0u;
        v_lds[t] = (v_py[t] * PW + pxp) * REC + cg * 8;
        v_c4[t] = (unsigned)cg * 4u;
        v_c4hw[t] = (unsigned)cg * 4u * ((unsigned)p.H * (unsigned)p.W * 4u);
        asm volatile("" : "+v"(v_c4hw[t]));                    // opaque: the optimiser otherwise folds the additions in issue() back into one multiply-add
    }
    // issue cursor: the chunk that the next issue() fetches
    int is_tile = 0, is_c = 0;
    unsigned is_off[VT];               // byte offset of (channel 0, iy, xq) inside the sample (OOB: outside the image / no task / past the last tile)
    __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1), 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs2 = rs1;
    float ct_reg[2] = {0.f, 0.f};      // PRE: the issue tile's coefficient table on its way to LDS (fetched with EVERY chunk: unconditional loads)
    int is_b = 0;                      // sample of the issue tile
    auto setup_issue_tile = [&]() {    // no memory operation in here: it runs under a condition
        if (is_tile >= n_my) {
#pragma unroll
            for (int t = 0; t < VT; ++t) is_off[t] = OOB;
            return;
        }
        int b, y0, x0;
        decode(is_tile, b, y0, x0);
        is_b = b;
        rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1 + (long)b * p.C1 * HW), 0, (int)((long)p.C1 * HW4), 0x00020000);
        rs2 = p.x2 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x2 + (long)b * p.C2 * HW), 0, (int)((long)p.C2 * HW4), 0x00020000) : rs1;
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            const int iy = y0 - 1 + v_py[t], xq = x0 - 4 + 4 * v_q[t];
            is_off[t] = (v_has[t] && (unsigned)iy < (unsigned)p.H && (unsigned)xq < (unsigned)p.W) ? (unsigned)(iy * p.W + xq) * 4u : OOB;
        }
    };
    // PRE: element e of the [3][ctab_n] table image (channels past C1 hold zeros); per-thread constants
    int ct_src[2];
    bool ct_ok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = tid + k * 64 * NW;
        const int which = e / ctab_n, c = e - which * ctab_n;
        ct_ok[k] = PRE && e < 3 * ctab_n && c < p.C1;
        ct_src[k] = ct_ok[k] ? which * p.C1 + c : 0;
    }
    auto issue = [&](f32x4v (&stg)[VT][4], bool (&ok)[VT], int& par) {
        if (is_c == 0) setup_issue_tile();
        const bool in1 = is_c < g.c1_chunks;                                  // workgroup-uniform
        const unsigned cb = (unsigned)(in1 ? is_c : is_c - g.c1_chunks) * CK;
        const unsigned clim = (unsigned)(in1 ? p.C1 : p.C2);
        const __amdgpu_buffer_rsrc_t rs = in1 ? rs1 : rs2;
        par = is_tile & 1;
        // keep the registers about to be re-loaded alive up to this point: in the window between their last use (write_stage at the end of the
        // previous step) and this re-issue the compiler otherwise parks temporaries in them and fences each with a vmcnt wait (write-after-
        // write against loads it cannot prove retired), which drains the chunk in flight and the epilogue stores at the start of a step
#pragma unroll
        for (int t = 0; t < VT; ++t) asm volatile("" ::"v"(stg[t][0]), "v"(stg[t][1]), "v"(stg[t][2]), "v"(stg[t][3]));
        if (PRE) asm volatile("" ::"v"(ct_reg[0]), "v"(ct_reg[1]));
        if (PRE) {
            // {mean, scale, shift} of the issue tile's sample: two floats per thread cover [3][C1 <= 341]; issued AHEAD of the chunk's data loads, so
            // the counted wait at the end of the step retires them; the same values again for every chunk of a tile (L2 hits, 4 KB)
            const float* src = p.in_norm + (long)is_b * 3 * p.C1;
#pragma unroll
            for (int k = 0; k < 2; ++k) ct_reg[k] = src[ct_src[k]];      // raw: the `channel exists` select waits until write_ctab (a use here would wait for the load at once)
        }
        const unsigned cbhw = cb * HW4;                                        // wave-uniform
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            ok[t] = is_off[t] != OOB;
            // additions only (a 32-bit multiply-add is a v_mad_u64_u32 whose unused upper addend picked a register with a load in flight: a
            // wait for that load, i.e. for everything older, in the middle of the issue block)
            const unsigned base = is_off[t] + v_c4hw[t] + cbhw;                // an out-of-range pixel stays out of range (sample < 2 GiB)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ch = cb + v_c4[t] + (unsigned)j;
                const unsigned off = (ch < clim) ? base + (unsigned)j * HW4 : OOB;   // the zero-weight channel tail is never fetched
                stg[t][j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
        }
        const int c_of = is_c;
        if (++is_c == g.nchunk) { is_c = 0; ++is_tile; }
        return c_of;
    };
    auto write_stage = [&](int slot, int chunk, const f32x4v (&stg)[VT][4], const bool (&ok)[VT], int par) {
        const float* ct = ctab + par * CT_SLOT;
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            unsigned char* base = patch + slot * PATCH + v_lds[t];
            f32x4v cm = {0.f, 0.f, 0.f, 0.f}, ca = cm, cs = cm;
            if (PRE) {
                const int c0 = chunk * CK + (int)v_c4[t];
                cm = *reinterpret_cast<const f32x4v*>(ct + c0);
                ca = *reinterpret_cast<const f32x4v*>(ct + ctab_n + c0);
                cs = *reinterpret_cast<const f32x4v*>(ct + 2 * ctab_n + c0);
            }
            // every loaded register is consumed UNCONDITIONALLY (conversion of all four columns), only the LDS writes of columns outside the
            // patch are masked: a load whose only uses sit behind a branch stays "maybe pending" for the compiler's wait insertion, which
            // then fences every later reuse of that register with a vmcnt wait that drains the chunks in flight
            f16x4 hi[4], lo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = stg[t][j][k];
                    if (PRE) {
                        v = (v - cm[j]) * ca[j] + cs[j];
                        v = p.in_slope < 0.f ? gelu_as(v) : (v > 0.f ? v : v * p.in_slope);
                        v = ok[t] ? v : 0.f;
                    }
                    _Float16 h, l;
                    split_f16(v, h, l);
                    hi[k][j] = h;
                    lo[k][j] = l;
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(hi[k]), "v"(lo[k]));      // (pins the conversions here: the optimiser would sink them into the masked blocks)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!((v_mask[t] >> k) & 1u)) continue;
                *reinterpret_cast<f16x4*>(base + k * REC) = hi[k];
                *reinterpret_cast<f16x4*>(base + k * REC + CK * 2) = lo[k];
            }
        }
    };
    auto write_ctab = [&](int par) {   // the table fetched with this step's issue, into the slot of that chunk's tile (read one step later at the earliest)
        if (PRE) {
            float* ct = ctab + par * CT_SLOT;
#pragma unroll
            for (int k = 0; k < 2; ++k) ct[tid + k * 64 * NW] = ct_ok[k] ? ct_reg[k] : 0.f;      // unconditional store: a slot holds 2 x 512 floats whatever 3 * ctab_n is
        }
    };
    // ---- weights: chunk `chunk` of the WM m-tiles by LDS-DMA into slot `slot` (fragment f of the workgroup = (m-tile f / 18, tap x part f % 18));
    // issued from inline asm so that the compiler's wait insertion neither sees a pending LDS write nor drains the VM counter for it
    const f16x8* const wbase = reinterpret_cast<const f16x8*>(wpk) + lane;
    auto issue_w = [&](int chunk, int slot) {
        const unsigned dst0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)(wl + slot * WSLOT));
        for (int f = wave; f < WM * 18; f += NW) {                          // wave-uniform
            const int m = f / 18, j = f - m * 18;
            const f16x8* src = wbase + (((long)m * g.nchunk + chunk) * 18 + j) * 64;
            const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)f * 1024u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };

    // ================================================================================================ MFMA side
    int b_rec[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) b_rec[nt] = (((ngrp * NTW + nt) * PW) + l31) * REC + half * 16;     // n-tile = output row ngrp * NTW + nt
    // acc: the tile being multiplied; hold: the finished tile whose epilogue runs inside the next tile's first step (a 16 * NTW register copy
    // per tile instead of two accumulator sets addressed by tile parity: ONE loop body, every long-lived value keeps its registers)
    f32x16 acc[NTW], hold[NTW];
    const bool do_stats = p.gn_ws != nullptr;
    const int cpg = do_stats ? p.Cout / p.gn_groups : 1;

    // deferred epilogue of a finished tile (accumulator set q, tile coordinates eb / ey0 / ex0): value = alpha * acc + bias, stored through a
    // per-sample buffer resource (a pixel outside the image / a channel beyond Cout is an out-of-range offset), statistics per channel
    int eb = 0, ey0 = 0, ex0 = 0, eb2 = 0;      // eb2: sample of the tile before the one whose epilogue is running
    struct Epi {                                // state of one running epilogue: lives inside ONE step (kept out of the registers that cross steps)
        float ssum[16], ssq[16];
        __amdgpu_buffer_rsrc_t rs_out;
        unsigned e_off[NTW];
        bool e_ok[NTW];
    };
    auto epi_begin = [&](Epi& E) {
        E.rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out + (long)eb * p.Cout * HW, 0, (int)((long)p.Cout * HW4), 0x00020000);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int oy = ey0 + ngrp * NTW + nt, ox = ex0 + l31;
            E.e_ok[nt] = oy < p.H && ox < p.W;
            // per-lane part of the address: pixel + this lane's first channel (m-tile, half); the register's channel (r & 3) + 8 * (r >> 2) rides
            // in the instruction's scalar offset.  Cout == 32 * WM (host check): every channel exists, only pixels can be out of range
            E.e_off[nt] = E.e_ok[nt] ? (unsigned)(oy * p.W + ox) * 4u + (unsigned)(mtl * 32 + 4 * half) * HW4 : OOB;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { E.ssum[r] = 0.f; E.ssq[r] = 0.f; }
    };
    auto epi_regs = [&](Epi& E, int r) __attribute__((always_inline)) {       // accumulator register r of every n-tile of the wave
        const float bv = biasl[mtl * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
        const unsigned soff = (unsigned)((r & 3) + 8 * (r >> 2)) * HW4;      // wave-uniform: an SGPR
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float v = p.alpha * hold[nt][r] + bv;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), E.rs_out, E.e_off[nt], soff, 0);
            const float m = E.e_ok[nt] ? v : 0.f;
            E.ssum[r] += m;
            E.ssq[r] += m * m;
        }
    };
    auto xreduce = [&](float (&v)[16]) {       // as conv_f16s: lane bits (b4 b3 b2 b1) select register 8*b4 + 4*b3 + 2*b2 + b1, summed over the half-wave
#pragma unroll
        for (int i = 0; i < 8; ++i) { const bool up = (lane & 16) != 0; const float keep = up ? v[i + 8] : v[i], send = up ? v[i] : v[i + 8]; v[i] = keep + __shfl_xor(send, 16, 64); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { const bool up = (lane & 8) != 0; const float keep = up ? v[i + 4] : v[i], send = up ? v[i] : v[i + 4]; v[i] = keep + __shfl_xor(send, 8, 64); }
#pragma unroll
        for (int i = 0; i < 2; ++i) { const bool up = (lane & 4) != 0; const float keep = up ? v[i + 2] : v[i], send = up ? v[i] : v[i + 2]; v[i] = keep + __shfl_xor(send, 4, 64); }
        { const bool up = (lane & 2) != 0; const float keep = up ? v[1] : v[0], send = up ? v[0] : v[1]; v[0] = keep + __shfl_xor(send, 2, 64); }
        v[0] += __shfl_xor(v[0], 1, 64);
    };
    auto epi_stats_to_lds = [&](Epi& E, int q) {       // per-channel totals of this wave into red[q] (zeroed two tiles ago)
        if (!do_stats) return;
        xreduce(E.ssum);
        xreduce(E.ssq);
        if ((lane & 1) == 0) {
            const int r = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            const int cl = mtl * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            atomicAdd(&red[q * WM * 64 + 2 * cl], E.ssum[0]);
            atomicAdd(&red[q * WM * 64 + 2 * cl + 1], E.ssq[0]);
        }
    };
    auto epi_stats_to_global = [&](int q, int b) {     // one barrier after epi_stats_to_lds: one fp64 atomic pair per (group, tile)
        if (!do_stats) return;
        if (tid < WM * 32) {
            const int co = tid;
            if (co < p.Cout && (tid == 0 || co % cpg == 0)) {
                int n = cpg - co % cpg;
                if (n > WM * 32 - tid) n = WM * 32 - tid;
                if (n > p.Cout - co) n = p.Cout - co;
                float s1 = 0.f, s2 = 0.f;
                for (int j = 0; j < n; ++j) { s1 += red[q * WM * 64 + 2 * (tid + j)]; s2 += red[q * WM * 64 + 2 * (tid + j) + 1]; }
                // issued from inline asm: a conditional VM operation the compiler can see between a staging load and its use makes its wait
                // insertion fall back to vmcnt(0), which would drain the two chunks in flight every tile
                double* w = p.gn_ws + 2L * ((long)b * p.gn_groups + co / cpg);
                const double d1 = (double)s1, d2 = (double)s2;
                asm volatile("global_atomic_add_f64 %0, %1, off\n\tglobal_atomic_add_f64 %0, %2, off offset:8" ::"v"(w), "v"(d1), "v"(d2) : "memory");
            }
        }
    };
    auto lds_barrier = [&]() {      // raw barrier: LDS traffic of this wave retired, nothing on the VM counter drained
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ================================================================================================ prologue
    f32x4v sa[VT][4] = {}, sb[VT][4] = {};       // register sets: chunk s + 2 goes to set s & 1
    bool oka[VT], okb[VT];
    int para = 0, parb = 0, ca_ = 0, cb_ = 0;
    for (int k = tid; k < 2 * WM * 64; k += 64 * NW) red[k] = 0.f;
    if (tid < WM * 32) biasl[tid] = (p.bias && tid < p.Cout) ? p.bias[tid] : 0.f;
    issue_w(0, 0);
    ca_ = issue(sa, oka, para);        // chunk 0
    cb_ = issue(sb, okb, parb);        // chunk 1 (nchunk >= 2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    write_ctab(0);                     // chunks 0 and 1 belong to tile 0 (nchunk >= 2)
    lds_barrier();
    write_stage(0, ca_, sa, oka, para);
    lds_barrier();

    // ================================================================================================ main loop
    // One step = one chunk.  ps = step parity (patch / weight slot and register set), pt = tile parity (accumulator set); nchunk is even, so a
    // tile starts on an even step.  `epi` (first step of a tile that has a predecessor): the predecessor's epilogue rides between the taps.
    // `rp` = parity of the red[] buffer the running epilogue adds into (the finished tile's index & 1); `second`: the step after an `epi` step.
    auto step = [&](const int ps, const bool epi, const bool second, const int rp, const int sidx, f32x4v (&st_issue)[VT][4], bool (&ok_issue)[VT],
                    int& par_issue, int& c_issue, const f32x4v (&st_write)[VT][4], const bool (&ok_write)[VT], const int par_write,
                    const int c_write) __attribute__((always_inline)) {
        // statistics of the tile before the finished one (red[rp ^ 1], filled during the previous tile's first step) to global memory: ahead of
        // this step's DMA and loads in program order, so that the counted wait below is not disturbed; red[rp ^ 1] is cleared one barrier later
        if (epi) epi_stats_to_global(rp ^ 1, eb2);
        if (second && tid < WM * 64) red[(rp ^ 1) * WM * 64 + tid] = 0.f;
        if (sidx + 1 < nsteps) issue_w((sidx + 1) % g.nchunk, ps ^ 1);
        Epi E;
        if (epi) epi_begin(E);
        const unsigned char* xb = patch + ps * PATCH;
        const unsigned char* wb = wl + ps * WSLOT + (mtl * 18) * 1024 + lane * 16;
        // operands of tap t + 1 are requested before the MFMAs of tap t (explicit two-deep register buffers); the scheduling barrier per tap
        // keeps the compiler from hoisting all nine taps' reads to the top (216 registers: spills at two waves per SIMD)
        f16x8 A[2][2], Bf[2][NTW][2];
        auto ldtap = [&](const int tap, const int buf) __attribute__((always_inline)) {
            const int toff = ((tap / 3) * PW + (tap % 3)) * REC;
            A[buf][0] = *reinterpret_cast<const f16x8*>(wb + (tap * 2) * 1024);
            A[buf][1] = *reinterpret_cast<const f16x8*>(wb + (tap * 2 + 1) * 1024);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const unsigned char* rp_ = xb + b_rec[nt] + toff;
                Bf[buf][nt][0] = *reinterpret_cast<const f16x8*>(rp_);
                Bf[buf][nt][1] = *reinterpret_cast<const f16x8*>(rp_ + CK * 2);
            }
        };
        ldtap(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 8) ldtap(tap + 1, (tap + 1) & 1);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[tap & 1][1], Bf[tap & 1][nt][0], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[tap & 1][0], Bf[tap & 1][nt][1], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[tap & 1][0], Bf[tap & 1][nt][0], acc[nt], 0, 0, 0);
            }
            if (epi && tap < 4) {             // four accumulator registers of the finished tile per tap: their VALU work and stores sit in the MFMA shadow
#pragma unroll
                for (int k = 0; k < 4; ++k) epi_regs(E, 4 * tap + k);
            }
            // Chunk s + 2 leaves HERE, behind the finished tile's stores in program order, in every step.  The VM counter retires in order and the
            // compiler counts for the merged epi / non-epi path: with the stores YOUNGER than these loads its waits in write_stage below (and
            // any wait that leaves these 4 * VT loads in flight) had to drain all 32 stores at the end of the step in which they were issued;
            // older than the loads they have the five remaining taps (~1900 clocks) to retire, and the loads still fly for ~1.5 steps.
            if (tap == 3) c_issue = issue(st_issue, ok_issue, par_issue);   // unconditional (out-of-range offsets past the last chunk)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (epi) epi_stats_to_lds(E, rp);
        // the weights of step s + 1, the register set of chunk s + 1 and the finished tile's stores have retired; chunk s + 2 stays in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * VT) : "memory");
        write_ctab(par_issue);
        // unconditional (past the last step it converts the zeros of out-of-range loads into the free slot): a conditional consumer leaves the
        // register set "maybe pending" for the compiler, which then guards every reuse of those registers with vmcnt waits that drain the queue
        write_stage(ps ^ 1, c_write, st_write, ok_write, par_write);
        lds_barrier();
    };

    int s = 0;
    bool have_prev = false;
    for (int i = 0; i < n_my; ++i) {
        int tb, ty0, tx0;
        decode(i, tb, ty0, tx0);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const int rp = (i - 1) & 1;               // red[] buffer of the finished tile i - 1
        for (int c = 0; c < g.nchunk; c += 2, s += 2) {
            const bool first = c == 0;
            step(0, first && have_prev, false, rp, s, sa, oka, para, ca_, sb, okb, parb, cb_);
            step(1, false, first && have_prev, rp, s + 1, sb, okb, parb, cb_, sa, oka, para, ca_);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) hold[nt] = acc[nt];
        eb2 = eb;
        eb = tb; ey0 = ty0; ex0 = tx0;
        have_prev = true;
    }
    // ---- drain: the statistics of the tile before the last (red[q ^ 1]) and the whole epilogue of the last tile (hold, red[q])
    {
        const int q = (n_my - 1) & 1;
        if (n_my > 1) epi_stats_to_global(q ^ 1, eb2);
        Epi E;
        epi_begin(E);
#pragma unroll
        for (int r = 0; r < 16; ++r) epi_regs(E, r);
        epi_stats_to_lds(E, q);
        lds_barrier();
        epi_stats_to_global(q, eb);
    }
}

}  // namespace

int g_stream = -1;        // -1: read CF_CONV_STREAM on first use
int stream_enabled() {
    if (g_stream < 0) { const char* e = getenv("CF_CONV_STREAM"); g_stream = e ? atoi(e) : 1; }
    return g_stream;
}

// Is this layer one the persistent kernel is built for?  3x3 / stride 1 / pad 1, 32 or 64 output channels written densely without activation or
// residual, an even number (>= 2) of 16-channel chunks, image rows that are whole 16-byte quads, enough tiles to keep 256 persistent
// workgroups busy, one sample of every tensor below 2 GiB; the deferred input normalisation for a single input of <= 341 channels.
bool conv_stream_applicable(const ConvParams& p) {
    if (!stream_enabled()) return false;
    if (!(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_h == 1 && p.pad_w == 1)) return false;
    if ((p.Cout != 32 && p.Cout != 64) || p.scatter2x2 || p.res || p.act != CF_ACT_NONE || p.out_coff != 0 || p.out_ctotal != p.Cout || p.w_bstride) return false;
    if ((p.W & 3) || p.W < 32 || p.H < 8) return false;
    const int c1c = (p.C1 + 15) / 16, c2c = p.C2 > 0 ? (p.C2 + 15) / 16 : 0;
    const int nchunk = c1c + c2c;
    if (nchunk < 2 || (nchunk & 1)) return false;
    if (p.C2 > 0 && !p.x2) return false;
    if (((reinterpret_cast<uintptr_t>(p.x1) | reinterpret_cast<uintptr_t>(p.x2)) & 15) != 0) return false;
    const long HW = (long)p.H * p.W;
    if ((long)p.C1 * HW * 4 >= (1L << 31) || (long)p.C2 * HW * 4 >= (1L << 31) || (long)p.Cout * HW * 4 >= (1L << 30)) return false;
    if (p.in_norm && (p.C2 != 0 || 3 * nchunk * 16 > 2 * 512 || (reinterpret_cast<uintptr_t>(p.in_norm) & 3) != 0)) return false;
    if (p.gn_ws && (p.gn_groups <= 0 || p.Cout % p.gn_groups != 0)) return false;
    const long ntiles = (long)p.B * ((p.H + ST_TH - 1) / ST_TH) * ((p.W + 31) / 32);
    if (ntiles < 1024 || ntiles >= (1L << 31)) return false;      // short launches keep the one-tile-per-workgroup kernel (nothing to stream across)
    return true;
}

template <int WM, int PRE>
static int launch_stream(const ConvParams& p, const StreamGeom& g, const _Float16* wpk, hipStream_t s) {
    constexpr int PH = ST_TH + 2;
    const size_t lds_bytes = (size_t)2 * PH * ST_PW * ST_REC + (size_t)2 * WM * 18 * 1024 + sizeof(float) * (2 * WM * 64 + WM * 32 + (PRE ? 2 * 2 * 64 * ST_NW : 0));
    if (lds_bytes > 160 * 1024) { set_error("conv_stream: LDS budget exceeded"); return CF_ERR_ARG; }
    auto kern = conv_stream_kernel<WM, PRE>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const unsigned nwg = 256;            // one persistent workgroup per CU (g.ntiles >= 1024)
    const double flops = 2.0 * (double)p.B * p.H * p.W * p.Cout * (p.C1 + p.C2) * 9.0;
    hipEvent_t e0, e1;
    if (profile_on() && profile_events(PK_CONV_STREAM, flops, &e0, &e1))
        hipExtLaunchKernelGGL(kern, dim3(nwg), dim3(64 * ST_NW), lds_bytes, s, e0, e1, 0, p, g, wpk);
    else
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * ST_NW), lds_bytes, s, p, g, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("conv_stream launch failed: ") + hipGetErrorString(e)); return CF_ERR_LAUNCH; }
    return CF_OK;
}

// The caller (conv_f16s.hip) has checked conv_stream_applicable(p) and zeroed p.gn_ws.
int launch_conv_stream(const ConvParams& p, const _Float16* wpk, hipStream_t s) {
    StreamGeom g;
    const int WM = p.Cout <= 32 ? 1 : 2;
    g.tiles_x = (p.W + 31) / 32;
    g.tiles_y = (p.H + ST_TH - 1) / ST_TH;
    g.ntiles = p.B * g.tiles_y * g.tiles_x;
    g.c1_chunks = (p.C1 + 15) / 16;
    g.nchunk = g.c1_chunks + (p.C2 > 0 ? (p.C2 + 15) / 16 : 0);
    if (p.in_norm) return WM == 1 ? launch_stream<1, 1>(p, g, wpk, s) : launch_stream<2, 1>(p, g, wpk, s);
    return WM == 1 ? launch_stream<1, 0>(p, g, wpk, s) : launch_stream<2, 0>(p, g, wpk, s);
}

}  // namespace cf

// A/B knob (also CF_CONV_STREAM=0|1 in the environment): 0 keeps every layer on conv_f16s_kernel's one-tile-per-workgroup shapes.  Returns the previous
// setting.  The two kernels compute bit-identical outputs (tests/test_gpu_ops.py::test_conv_stream_*).
extern "C" int cf_conv_stream_enable(int on) {
    const int prev = cf::stream_enabled();
    cf::g_stream = on ? 1 : 0;
    return prev;
}
