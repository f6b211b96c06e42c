// Register-resident-weights kernel for the 3x3 / stride 1 / pad 1 layers with <= 64 input channels and 32 or 64 output channels on large maps
// (64 -> 64 with and without deferred normalisation, 32 -> 32, 32 + 32 -> 32).  Same arithmetic as conv_stream.hip / conv_f16s.hip: f16 hi/lo
// 3-term split on v_mfma_f32_32x32x16_f16, fp32 accumulation, same packed weights, same 80-byte LDS pixel records; taps of a chunk summed in
// (kx, ky) order like conv_stream.
//
// Why (profiles/r03_conv_stream.md): with two waves per SIMD these layers are bound by the instructions a wave issues per MFMA -- weight LDS-DMA
// (100-185 clocks of issue each), weight-fragment LDS reads, staging loads, conversions -- not by any pipe.  Here
//   * a workgroup is FOUR waves, one per SIMD, 512 registers each (accumulators and most weights in AGPRs);
//   * every wave keeps the weights of its m-tile for ALL k-steps in registers for the whole kernel (<= 36 k-steps x (hi, lo) x 4 = 288
//     registers): no weight traffic of any kind after the prologue -- no LDS-DMA, no A-fragment reads (18 of the 42 ds_read_b128 per chunk);
//   * a wave owns FOUR vertically adjacent output rows: a B fragment (input row, kx) feeds up to three ky: 36 ds_read_b128 per 108 MFMAs;
//   * persistent workgroups, chunk stream across tile boundaries, the previous tile's epilogue (from a copy of the accumulators) and the
//     conversion of the next chunk sliced between the 18 tap units of a step, one raw barrier per step -- as in conv_stream;
//   * one staging register set: the loads of chunk s + 2 leave as soon as chunk s + 1 has been converted, 12-14 units (~2 500 clocks) ahead
//     of their use.
// What is left per MFMA gap of one wave: 0.33 ds_read_b128, 0.07 loads, 0.15-0.3 stores, ~3 VALU -- inside the five fillers a gap hides at one
// wave per SIMD (MI355X_MICROARCH.md, constants table).
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

struct WregGeom {
    int tiles_x, tiles_y, ntiles;   // tiles of TH x 32 output pixels
    int c1_chunks;                  // chunks < c1_chunks read x1, the others x2
};

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* q) {
    const unsigned long long a = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const T*)(((unsigned long long)hi << 32) | lo);
}

constexpr int WR_NW = 4, WR_NTW = 4, WR_PW = 34, WR_REC = 80, WR_CK = 16, WR_NQ = 10;
constexpr int wr_tile_rows(int wm) { return (WR_NW / wm) * WR_NTW; }       // 64 channels: 2 wave groups x 4 rows = 8; 32 channels: 4 x 4 = 16

constexpr int wr_epi_unit(int rps, int k) { return rps == 4 ? 2 + 4 * k : 1 + 2 * k; }
constexpr int wr_n_younger(int rps, int nslu, int ntw) {
    int n = 0;
    for (int k = 0; k < rps; ++k) n += (wr_epi_unit(rps, k) > nslu) ? ntw : 0;
    return n;
}

template <int WM, int NCH, int PRE>
__global__ void __launch_bounds__(64 * WR_NW) conv_wreg_kernel(const ConvParams p, const WregGeom g, const _Float16* __restrict__ wpk) {
    constexpr int NW = WR_NW, NTW = WR_NTW, PW = WR_PW, REC = WR_REC, CK = WR_CK, NQ = WR_NQ;
    constexpr int NG = NW / WM;
    constexpr int TH = NG * NTW;
    constexpr int PH = TH + 2;
    constexpr int NTASK = 4 * PH * NQ;
    constexpr int VT = (NTASK + 64 * NW - 1) / (64 * NW);
    constexpr int PATCH = PH * PW * REC;
    constexpr int CT_SLOT = 64 * NW;             // floats per coefficient-table slot (PRE): [3][NCH * 16 <= 64] image, one entry per thread
    constexpr unsigned OOB = 0x80000000u;
    static_assert(NCH == 2 || NCH == 4, "two or four 16-channel chunks");
    constexpr int NREG = NCH == 4 ? (PRE ? 2 : 3) : NCH;     // chunks whose weights live in registers; the others' 18 fragments each stay in LDS for the whole kernel
    constexpr int WLDS = (NCH - NREG) * WM * 18 * 1024;      // bytes (with all 288 registers of four chunks resident the 256 + 256 file spilled)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const patch = lds;
    unsigned char* const wl = lds + 2 * PATCH;                                      // [WM][18 fragments][64 lanes][16 B]: the last chunk's weights
    float* const red = reinterpret_cast<float*>(lds + 2 * PATCH + WLDS);            // [2][WM * 32 channels][2]
    float* const biasl = red + 2 * WM * 64;
    float* const ctab = biasl + WM * 32;                                            // [2][CT_SLOT] (PRE only)
    constexpr int ctab_n = NCH * CK;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int mtl = wave % WM, ngrp = wave / WM;

    const int xcd = blockIdx.x & 7, wgi = blockIdx.x >> 3, nwgx = gridDim.x >> 3;
    const int band_lo = (int)((long)g.ntiles * xcd / 8), band_hi = (int)((long)g.ntiles * (xcd + 1) / 8);
    if (band_lo + wgi >= band_hi) return;
    const int n_my = (band_hi - band_lo - wgi + nwgx - 1) / nwgx;
    const int HW = p.H * p.W;
    const unsigned HW4 = (unsigned)HW * 4u;
    auto decode = [&](int i, int& b, int& y0, int& x0) {
        int id = band_lo + wgi + i * nwgx;
        const int tx = id % g.tiles_x;
        id /= g.tiles_x;
        const int ty = id % g.tiles_y;
        b = id / g.tiles_y;
        y0 = ty * TH;
        x0 = tx * 32;
    };

    // ================================================================================================ weights: resident for the whole kernel
    f16x8 W[NREG * 9][2];      // [chunk * 9 + tap][hi, lo] of this wave's m-tile
    {
        const f16x8* wsrc = reinterpret_cast<const f16x8*>(wpk) + (long)mtl * NCH * 18 * 64 + lane;
#pragma unroll
        for (int i = 0; i < NREG * 9; ++i) {
            W[i][0] = wsrc[(i * 2) * 64];
            W[i][1] = wsrc[(i * 2 + 1) * 64];
        }
        if (NCH > NREG && ngrp == 0) {        // one wave per m-tile parks the last chunk's fragments in LDS (read by every wave of that m-tile)
#pragma unroll
            for (int j = 0; j < (NCH - NREG) * 18; ++j)
                *reinterpret_cast<f16x8*>(wl + ((mtl * (NCH - NREG) * 18 + j) * 64 + lane) * 16) = wsrc[(NREG * 18 + j) * 64];
        }
    }

    // Weight fragment i (= chunk * 18 + tap * 2 + part) lives in an AGPR quad while i < NA, in a VGPR quad otherwise, and the MFMA names it
    // directly: left to the register allocator most fragments sat in AGPRs and were copied to VGPRs in front of every use (97
    // v_accvgpr_read per 108 MFMAs).  The accumulators are AGPRs in both forms.
    constexpr int NA = (256 - NTW * 16) / 4;                  // AGPR quads left beside the accumulators
    // (NCH = 4: the register allocator spills the pinned operands around the long loop body -- 880 B of scratch --, so there the builtin form
    // stays and the allocator moves fragments between the files itself)
    constexpr bool PIN = NCH == 2;
    auto mfma = [&](f32x16& a, const int i, const f16x8& b) __attribute__((always_inline)) {
        if (!PIN)
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[i >> 1][i & 1], b, a, 0, 0, 0);
        else if (i < NA)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(a) : "a"(W[i >> 1][i & 1]), "v"(b));
        else
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(a) : "v"(W[i >> 1][i & 1]), "v"(b));
    };
    auto mfma_v = [&](f32x16& a, const f16x8& w, const f16x8& b) __attribute__((always_inline)) {
        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, b, a, 0, 0, 0);
    };

    // ================================================================================================ staging (all threads)
    int v_lds[VT], v_py[VT], v_q[VT];
    unsigned v_c4[VT], v_c4hw[VT], v_mask[VT];
    bool v_has[VT];
#pragma unroll
    for (int t = 0; t < VT; ++t) {
        const int task = tid + t * 64 * NW;
        // lane bits (cg: 2, quad parity: 1, row parity: 1) first: conflict-free ds_write_b64 (see conv_stream.hip)
        static_assert(PH % 2 == 0 && NQ == 10, "task map below");
        constexpr int R0 = 64 * (PH / 2), R1 = 16 * (PH / 2);
        static_assert(R0 + R1 == NTASK && R0 % 64 == 0, "task map below");
        v_has[t] = task < NTASK;
        const int e = task - R0;
        const int cg = task & 3;
        v_q[t] = task < R0 ? (((task >> 2) & 1) | (((task >> 4) & 3) << 1)) : 8 + ((e >> 2) & 1);
        v_py[t] = task < R0 ? 2 * (task >> 6) + ((task >> 3) & 1) : 2 * (e >> 4) + ((e >> 3) & 1);
        if (!v_has[t]) { v_q[t] = 0; v_py[t] = 0; }
        const int pxp = 4 * v_q[t] - 3;
        v_mask[t] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v_mask[t] |= (v_has[t] && (unsigned)(pxp + k) < (unsigned)PW) ? (1u << k) : 0u;
        v_lds[t] = (v_py[t] * PW + pxp) * REC + cg * 8;
        v_c4[t] = (unsigned)cg * 4u;
        v_c4hw[t] = (unsigned)cg * 4u * ((unsigned)p.H * (unsigned)p.W * 4u);
        asm volatile("" : "+v"(v_c4hw[t]));
    }
    int is_tile = 0;                   // tile of the chunk the next issue() fetches (its chunk index is a compile-time constant at every call site)
    unsigned is_off[VT];
    __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1), 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs2 = rs1;
    auto setup_issue_tile = [&]() {
        if (is_tile >= n_my) {
#pragma unroll
            for (int t = 0; t < VT; ++t) is_off[t] = OOB;
            return;
        }
        int b, y0, x0;
        decode(is_tile, b, y0, x0);
        rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1 + (long)b * p.C1 * HW), 0, (int)((long)p.C1 * HW4), 0x00020000);
        rs2 = p.x2 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x2 + (long)b * p.C2 * HW), 0, (int)((long)p.C2 * HW4), 0x00020000) : rs1;
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            const int iy = y0 - 1 + v_py[t], xq = x0 - 4 + 4 * v_q[t];
            is_off[t] = (v_has[t] && (unsigned)iy < (unsigned)p.H && (unsigned)xq < (unsigned)p.W) ? (unsigned)(iy * p.W + xq) * 4u : OOB;
        }
    };
    // PRE: entry e of the [3][ctab_n] table image <- element ct_src of the sample's [3][C1] table, by LDS-DMA, one dword per thread
    unsigned ct_src;
    {
        const int which = tid / ctab_n, c = tid - which * ctab_n;
        ct_src = (PRE && tid < 3 * ctab_n && c < p.C1) ? (unsigned)(which * p.C1 + c) * 4u : 0u;
    }
    auto tile_sample = [&](int i) {
        if (i >= n_my) return 0;
        int b, y0, x0;
        decode(i, b, y0, x0);
        return b;
    };
    auto issue_ctab = [&](int par, int b) {      // the table of sample b -> ctab[par]
        if (PRE) {
            const float* src = uniform_ptr(p.in_norm + (long)b * 3 * p.C1);
            const unsigned dst0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)(ctab + par * CT_SLOT));
            const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)(wave * 64) * 4u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(ct_src), "s"(dst), "s"(src) : "memory");
        }
    };
    // chunk `ch` (compile-time) of the issue tile; the caller advances is_tile after the tile's last chunk
    auto issue = [&](f32x4v (&stg)[VT][4], const int ch) __attribute__((always_inline)) {
        const bool in1 = ch < g.c1_chunks;                                    // workgroup-uniform
        const unsigned cb = (unsigned)(in1 ? ch : ch - g.c1_chunks) * CK;
        const unsigned clim = (unsigned)(in1 ? p.C1 : p.C2);
        const __amdgpu_buffer_rsrc_t rs = in1 ? rs1 : rs2;
        const unsigned cbhw = cb * HW4;
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            const unsigned base = is_off[t] + v_c4hw[t] + cbhw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned c = cb + v_c4[t] + (unsigned)j;
                const unsigned off = (c < clim) ? base + (unsigned)j * HW4 : OOB;
                stg[t][j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
        }
    };
    // one slice = the four channels of image column k of staging task t: conversion + two 8-byte LDS writes
    auto write_stage_slice = [&](int slot, const int chunk, const f32x4v (&stg)[VT][4], int par, const int t, const int k) __attribute__((always_inline)) {
        unsigned char* base = patch + slot * PATCH + v_lds[t];
        f32x4v cm = {0.f, 0.f, 0.f, 0.f}, ca = cm, cs = cm;
        if (PRE) {
            const float* ct = ctab + par * CT_SLOT;
            const int c0 = chunk * CK + (int)v_c4[t];
            cm = *reinterpret_cast<const f32x4v*>(ct + c0);
            ca = *reinterpret_cast<const f32x4v*>(ct + ctab_n + c0);
            cs = *reinterpret_cast<const f32x4v*>(ct + 2 * ctab_n + c0);
        }
        const bool ok = is_off[t] != OOB;      // (is_off still belongs to the tile of the set being converted: the next tile is set up after the slices)
        f16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = stg[t][j][k];
            if (PRE) {
                v = (v - cm[j]) * ca[j] + cs[j];
                v = p.in_slope < 0.f ? gelu_as(v) : (v > 0.f ? v : v * p.in_slope);
                v = ok ? v : 0.f;
            }
            _Float16 h, l;
            split_f16(v, h, l);
            hi[j] = h;
            lo[j] = l;
        }
        asm volatile("" ::"v"(hi), "v"(lo));
        if ((v_mask[t] >> k) & 1u) {
            *reinterpret_cast<f16x4*>(base + k * REC) = hi;
            *reinterpret_cast<f16x4*>(base + k * REC + CK * 2) = lo;
        }
    };

    // ================================================================================================ MFMA side
    const int brow = ((ngrp * NTW) * PW + l31) * REC + half * 16;
    f32x16 acc[NTW], hold[NTW];
    const bool do_stats = p.gn_ws != nullptr;
    const int cpg = do_stats ? p.Cout / p.gn_groups : 1;
    int eb = 0, ey0 = 0, ex0 = 0, eb2 = 0;
    bool have_prev = false;
    struct Epi {
        bool full;                              // wave-uniform: every pixel of the finished tile exists
        float tsum, tsq;
        __amdgpu_buffer_rsrc_t rs_out;
        unsigned e_off[NTW];
        bool e_ok[NTW];
    };
    // the stores of an epilogue slice are UNCONDITIONAL (out-of-range offsets while there is no finished tile): every step then issues the
    // same number of vector-memory operations and the counted wait in front of the staging slices is exact
    auto epi_begin = [&](Epi& E) {
        E.rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out + (long)eb * p.Cout * HW, 0, (int)((long)p.Cout * HW4), 0x00020000);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int oy = ey0 + ngrp * NTW + nt, ox = ex0 + l31;
            E.e_ok[nt] = have_prev && oy < p.H && ox < p.W;
            E.e_off[nt] = E.e_ok[nt] ? (unsigned)(oy * p.W + ox) * 4u + (unsigned)(mtl * 32 + 4 * half) * HW4 : OOB;
        }
        E.full = have_prev && ey0 + TH <= p.H && ex0 + 32 <= p.W;
        E.tsum = 0.f;
        E.tsq = 0.f;
    };
    // sums over the 32 lanes of each half of the wave (-> lanes 16-31 / 48-63) of TWO values at once: four in-row butterflies and one row
    // broadcast as v_add_f32_dpp; the two chains interleave so that one s_nop covers the DPP read-after-write hazard of both (written by hand:
    // the compiler expanded each step into v_mov + v_mov_dpp + a packed add)
    auto half_sum2 = [&](float& a, float& b) __attribute__((always_inline)) {
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "s_nop 1"
            : "+v"(a), "+v"(b));
    };
    constexpr int RPS = 16 / NCH;       // accumulator registers (of all four n-tiles) whose epilogue rides in one step
    auto epi_reg = [&](Epi& E, const int r) __attribute__((always_inline)) {       // accumulator register r of every n-tile of the wave
        const float bv = biasl[mtl * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
        const unsigned soff = (unsigned)((r & 3) + 8 * (r >> 2)) * HW4;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float v = p.alpha * hold[nt][r] + bv;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), E.rs_out, E.e_off[nt], soff, 0);
            float m = v;
            if (!E.full) m = E.e_ok[nt] ? v : 0.f;
            s1 += m;
            s2 = fmaf(m, m, s2);
        }
        if (do_stats) {
            half_sum2(s1, s2);
            const bool mine = (lane & 15) == (r % RPS);
            E.tsum = mine ? s1 : E.tsum;
            E.tsq = mine ? s2 : E.tsq;
        }
    };
    auto epi_stats_to_lds = [&](Epi& E, int q, const int r0) {      // totals of registers r0 .. r0 + RPS - 1 (lanes 16 + i of each half) into red[q]
        if (!do_stats) return;
        if ((lane & 16) && (lane & 15) < RPS) {
            const int r = r0 + (lane & 15);
            const int cl = mtl * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            atomicAdd(&red[q * WM * 64 + 2 * cl], E.tsum);
            atomicAdd(&red[q * WM * 64 + 2 * cl + 1], E.tsq);
        }
    };
    auto epi_stats_to_global = [&](int q, int b) {
        if (!do_stats) return;
        if (tid < WM * 32) {
            const int co = tid;
            if (co < p.Cout && (tid == 0 || co % cpg == 0)) {
                int n = cpg - co % cpg;
                if (n > WM * 32 - tid) n = WM * 32 - tid;
                if (n > p.Cout - co) n = p.Cout - co;
                float s1 = 0.f, s2 = 0.f;
                for (int j = 0; j < n; ++j) { s1 += red[q * WM * 64 + 2 * (tid + j)]; s2 += red[q * WM * 64 + 2 * (tid + j) + 1]; }
                double* w = p.gn_ws + 2L * ((long)b * p.gn_groups + co / cpg);
                const double d1 = (double)s1, d2 = (double)s2;
                asm volatile("global_atomic_add_f64 %0, %1, off\n\tglobal_atomic_add_f64 %0, %2, off offset:8" ::"v"(w), "v"(d1), "v"(d2) : "memory");
            }
        }
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ================================================================================================ prologue
    f32x4v stg[VT][4] = {};
    for (int k = tid; k < 2 * WM * 64; k += 64 * NW) red[k] = 0.f;
    if (tid < WM * 32) biasl[tid] = (p.bias && tid < p.Cout) ? p.bias[tid] : 0.f;
    setup_issue_tile();
    issue_ctab(0, tile_sample(0));
    if (NCH == 2) issue_ctab(1, tile_sample(1));
    issue(stg, 0);                                              // chunk 0 of tile 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the weights and the tables as well)
    lds_barrier();
#pragma unroll
    for (int t = 0; t < VT; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) write_stage_slice(0, 0, stg, 0, t, k);
    issue(stg, 1);                                              // chunk 1 of tile 0 (NCH >= 2): in flight into step 0
    lds_barrier();

    // ================================================================================================ main loop
    // Step c of tile i multiplies chunk c from patch slot c & 1 (NCH is even).  Between its 18 tap units (kx, input row):
    //   unit  0              counted wait: the chunk loaded during the previous step has landed (only that step's later stores are younger)
    //   units 0 .. NSLU-1    conversion of the NEXT chunk (c + 1 of this tile / 0 of the next) into the other slot, two slices per unit
    //   unit  NSLU           the register set is free: the chunk after that leaves (18 - NSLU units + a barrier ahead of its use)
    //   epilogue units       RPS accumulator registers x 4 rows of the previous tile: stores + statistics, one register per unit
    // PRE: the coefficient table of a tile is fetched (LDS-DMA) early enough for one counted wait AND one barrier to lie between its landing and
    // its first reader: NCH = 4: table of tile i + 1 in step (i, 0); NCH = 2: table of tile i + 2 in step (i, 1).
    constexpr int NR = NTW + 2, NU = 3 * NR;
    constexpr int NSL = 4 * VT, NSLU = NSL / 2;
    // epilogue register k of a step (k < RPS) rides in unit wr_epi_unit(RPS, k); N_YOUNGER = vector-memory operations of a step that are
    // younger than its loads: the stores of the epilogue registers in later units
    constexpr int N_YOUNGER = wr_n_younger(RPS, NSLU, NTW);
    static_assert(N_YOUNGER < 60 && wr_epi_unit(RPS, RPS - 1) < NU, "schedule");
    for (int i = 0; i < n_my; ++i) {
        int tb, ty0, tx0;
        decode(i, tb, ty0, tx0);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const int rp = (i - 1) & 1;               // red[] buffer of the finished tile i - 1
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ps = c & 1;
            const int cw = (c + 1) % NCH;                       // chunk converted in this step (tile i, or i + 1 after the wrap)
            const int ci = (c + 2) % NCH;                       // chunk that leaves in this step (tile i + (c + 2) / NCH)
            const int par_w = (i + (c + 1) / NCH) & 1;          // table slot of the converted chunk's tile
            if (c == 0 && have_prev) epi_stats_to_global(rp ^ 1, eb2);
            if (c == 1 && have_prev && tid < WM * 64) red[(rp ^ 1) * WM * 64 + tid] = 0.f;
            Epi E;
            epi_begin(E);
            const unsigned char* xb = patch + ps * PATCH;
            f16x8 Bq[3][2];                                     // operand ring: requested two units (12 MFMAs) ahead of their use
            auto ldB = [&](const int u, const int buf) __attribute__((always_inline)) {
                const unsigned char* rp_ = xb + brow + ((u % NR) * PW + (u / NR)) * REC;
                Bq[buf][0] = *reinterpret_cast<const f16x8*>(rp_);
                Bq[buf][1] = *reinterpret_cast<const f16x8*>(rp_ + CK * 2);
            };
            // the LDS-resident chunk: A(kx, ky) is used in units ir = ky .. ky + 3, so one register set serves (A(kx + 1, 0) / (kx + 1, 1) are
            // requested in units ir = 4 / 5 of kx, A(kx, 2) in unit 0 of kx: two units ahead of their first use)
            f16x8 Aq[3][2];
            const unsigned char* wb = wl + (mtl * (NCH - NREG) * 18 + (c >= NREG ? (c - NREG) * 18 : 0)) * 1024 + lane * 16;
            auto ldA = [&](const int kx, const int ky) __attribute__((always_inline)) {
                Aq[ky][0] = *reinterpret_cast<const f16x8*>(wb + ((ky * 3 + kx) * 2) * 1024);
                Aq[ky][1] = *reinterpret_cast<const f16x8*>(wb + ((ky * 3 + kx) * 2 + 1) * 1024);
            };
            if (c >= NREG) { ldA(0, 0); ldA(0, 1); }
            ldB(0, 0);
            ldB(1, 1);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int kx = u / NR, ir = u % NR;
                if (u + 2 < NU) ldB(u + 2, (u + 2) % 3);
                if (c >= NREG) {
                    if (ir == 0) ldA(kx, 2);
                    if (ir == 4 && kx < 2) ldA(kx + 1, 0);
                    if (ir == 5 && kx < 2) ldA(kx + 1, 1);
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int nt = ir - ky;
                    if (nt >= 0 && nt < NTW) {
                        if (c < NREG) {
                            const int wi = (c * 9 + ky * 3 + kx) * 2;
                            mfma(acc[nt], wi + 1, Bq[u % 3][0]);
                            mfma(acc[nt], wi, Bq[u % 3][1]);
                            mfma(acc[nt], wi, Bq[u % 3][0]);
                        } else {
                            mfma_v(acc[nt], Aq[ky][1], Bq[u % 3][0]);
                            mfma_v(acc[nt], Aq[ky][0], Bq[u % 3][1]);
                            mfma_v(acc[nt], Aq[ky][0], Bq[u % 3][0]);
                        }
                    }
                }
                // ---- the unit's slice of everything else
                if (u == 0) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_YOUNGER) : "memory");
                    if (PRE && NCH == 4 && c == 0) issue_ctab((i + 1) & 1, tile_sample(i + 1));
                    if (PRE && NCH == 2 && c == 1) issue_ctab(i & 1, tile_sample(i + 2));
                }
                if (u < NSLU) {
                    write_stage_slice(ps ^ 1, cw, stg, par_w, (2 * u) / 4, (2 * u) % 4);
                    write_stage_slice(ps ^ 1, cw, stg, par_w, (2 * u + 1) / 4, (2 * u + 1) % 4);
                }
#pragma unroll
                for (int k = 0; k < RPS; ++k)
                    if (wr_epi_unit(RPS, k) == u) epi_reg(E, c * RPS + k);
                if (u == NSLU) {
                    if (ci == 0) {                              // the first chunk of the next tile: its offsets and descriptors
                        is_tile = i + (c + 2) / NCH;
                        setup_issue_tile();
                    }
                    issue(stg, ci);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            epi_stats_to_lds(E, rp, c * RPS);
            lds_barrier();
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
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            Epi E;
            epi_begin(E);
#pragma unroll
            for (int k = 0; k < RPS; ++k) epi_reg(E, c * RPS + k);
            epi_stats_to_lds(E, q, c * RPS);
        }
        lds_barrier();
        epi_stats_to_global(q, eb);
    }
}

}  // namespace

int g_wreg = -1;          // -1: read CF_CONV_WREG on first use
static int wreg_enabled() {
    if (g_wreg < 0) { const char* e = getenv("CF_CONV_WREG"); g_wreg = e ? atoi(e) : 1; }
    return g_wreg;
}

// 3x3 / stride 1 / pad 1 to 32 or 64 dense output channels without activation or residual, two or four 16-channel chunks of input (x1's and
// x2's chunks counted separately), rows of whole 16-byte quads, >= 1024 tiles, every sample below 2 GiB; the deferred input normalisation for a
// single input
bool conv_wreg_applicable(const ConvParams& p) {
    if (!wreg_enabled()) return false;
    if (!(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_h == 1 && p.pad_w == 1)) return false;
    if ((p.Cout != 32 && p.Cout != 64) || p.scatter2x2 || p.res || p.act != CF_ACT_NONE || p.out_coff != 0 || p.out_ctotal != p.Cout || p.w_bstride) return false;
    const int TH = wr_tile_rows(p.Cout <= 32 ? 1 : 2);
    if ((p.W & 3) || p.W < 32 || p.H < TH) return false;
    const int c1c = (p.C1 + 15) / 16, c2c = p.C2 > 0 ? (p.C2 + 15) / 16 : 0;
    const int nchunk = c1c + c2c;
    if (nchunk != 2 && nchunk != 4) return false;
    if (p.C2 > 0 && !p.x2) return false;
    if (((reinterpret_cast<uintptr_t>(p.x1) | reinterpret_cast<uintptr_t>(p.x2)) & 15) != 0) return false;
    const long HW = (long)p.H * p.W;
    if ((long)p.C1 * HW * 4 >= (1L << 31) || (long)p.C2 * HW * 4 >= (1L << 31) || (long)p.Cout * HW * 4 >= (1L << 30)) return false;
    if (p.in_norm && (p.C2 != 0 || (reinterpret_cast<uintptr_t>(p.in_norm) & 3) != 0)) return false;
    if (p.in_norm && nchunk == 4 && wreg_enabled() < 2) return false;      // (that instantiation spills its weights: 592 B of scratch per lane; level 2 runs it for the tests)
    if (p.gn_ws && (p.gn_groups <= 0 || p.Cout % p.gn_groups != 0)) return false;
    const long ntiles = (long)p.B * ((p.H + TH - 1) / TH) * ((p.W + 31) / 32);
    if (ntiles < 1024 || ntiles >= (1L << 31)) return false;
    return true;
}

template <int WM, int NCH, int PRE>
static int launch_wreg(const ConvParams& p, const WregGeom& g, const _Float16* wpk, hipStream_t s) {
    constexpr int PH = wr_tile_rows(WM) + 2;
    const size_t lds_bytes = (size_t)2 * PH * WR_PW * WR_REC + (NCH == 4 ? (size_t)(PRE ? 2 : 1) * WM * 18 * 1024 : 0) + sizeof(float) * (2 * WM * 64 + WM * 32 + (PRE ? 2 * 64 * WR_NW : 0));
    auto kern = conv_wreg_kernel<WM, NCH, PRE>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const unsigned nwg = 256;            // one persistent workgroup per CU
    const double flops = 2.0 * (double)p.B * p.H * p.W * p.Cout * (p.C1 + p.C2) * 9.0;
    hipEvent_t e0, e1;
    if (profile_on() && profile_events(PK_CONV_STREAM, flops, &e0, &e1))
        hipExtLaunchKernelGGL(kern, dim3(nwg), dim3(64 * WR_NW), lds_bytes, s, e0, e1, 0, p, g, wpk);
    else
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * WR_NW), lds_bytes, s, p, g, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("conv_wreg launch failed: ") + hipGetErrorString(e)); return CF_ERR_LAUNCH; }
    return CF_OK;
}

// The caller has checked conv_wreg_applicable(p) and zeroed p.gn_ws.
int launch_conv_wreg(const ConvParams& p, const _Float16* wpk, hipStream_t s) {
    WregGeom g;
    const int WM = p.Cout <= 32 ? 1 : 2;
    g.tiles_x = (p.W + 31) / 32;
    g.tiles_y = (p.H + wr_tile_rows(WM) - 1) / wr_tile_rows(WM);
    g.ntiles = p.B * g.tiles_y * g.tiles_x;
    g.c1_chunks = (p.C1 + 15) / 16;
    const int nchunk = g.c1_chunks + (p.C2 > 0 ? (p.C2 + 15) / 16 : 0);
    const int pre = p.in_norm ? 1 : 0;
    if (WM == 1) {
        if (nchunk == 2) return pre ? launch_wreg<1, 2, 1>(p, g, wpk, s) : launch_wreg<1, 2, 0>(p, g, wpk, s);
        return pre ? launch_wreg<1, 4, 1>(p, g, wpk, s) : launch_wreg<1, 4, 0>(p, g, wpk, s);
    }
    if (nchunk == 2) return pre ? launch_wreg<2, 2, 1>(p, g, wpk, s) : launch_wreg<2, 2, 0>(p, g, wpk, s);
    return pre ? launch_wreg<2, 4, 1>(p, g, wpk, s) : launch_wreg<2, 4, 0>(p, g, wpk, s);
}

}  // namespace cf

extern "C" int cf_conv_wreg_enable(int on) {
    const int prev = cf::wreg_enabled();
    cf::g_wreg = on < 0 ? 0 : (on > 2 ? 2 : on);
    return prev;
}
