// Persistent, software-pipelined sibling of conv_f16s_kernel for the layers whose K loop is too short to hide anything: 3x3 / stride 1 / pad 1
// convolutions to 32 or 64 output channels on large maps (the flow net's 64-channel level at 256x256, the U-Net's 32- and 64-channel levels).
// Same arithmetic as conv_f16s.hip -- f16 hi/lo 3-term split on v_mfma_f32_32x32x16_f16, fp32 accumulation, same packed weights, same 80-byte
// LDS pixel records; the taps of a chunk are summed in (kx, ky) instead of (ky, kx) order, so outputs agree with that kernel's to fp32
// summation-order noise (tests/test_gpu_ops.py: <= 4e-6 of the output scale), not bit for bit.
//
// Why another kernel (profiles/r03_conv_stream.md).  In conv_f16s these layers reach ~55 % of the MFMA rate the chip sustains, and shader-clock
// stamps inside the kernel (make trace, tools/stream_trace.py) show why: the eight waves of a workgroup go through the phases of a chunk --
// weight DMA issue, taps, convert-and-write of the next chunk, barrier -- at the same time, so the MFMA pipe only works during `taps`
// (3540 of 6390 clocks per chunk at 64 -> 64), and the epilogue of all 256 workgroups hits HBM in the same 1/nchunk of the time.  Here:
//   * ONE workgroup per CU (8 waves) is persistent and walks a band of tiles of its XCD; the 16-channel chunks of consecutive tiles form one
//     stream: while chunk s is multiplied, chunk s + 1 is converted into LDS and chunk s + 2 is in flight;
//   * a step (= chunk) is 12 tap units of 4-5 MFMAs and EVERYTHING else is cut into slices that ride between the units, in the MFMA shadow: the
//     weight DMA of the next step, the loads of chunk s + 2, the conversion of chunk s + 1, and -- in the first two steps of a tile -- the
//     previous tile's epilogue (scale, bias, GroupNorm statistics, 16 stores per wave and step).  The only serial part is one raw s_barrier
//     (+ lgkmcnt(0)) per step; one counted vmcnt per step, placed so that only the loads of chunk s + 2 are younger than it;
//   * row sharing: a wave owns two vertically adjacent output rows, so a B fragment (input row, kx) serves up to three ky: 0.78 KB of LDS
//     operand reads per MFMA instead of 1.0-1.33;
//   * weights go through LDS as in the WL shapes of conv_f16s (LDS-DMA, none of them in a VGPR), a whole chunk (9 taps) per slot;
//   * the deferred input normalisation (PRE) reads its {mean, scale, shift} table from LDS, double buffered by tile parity and fetched by
//     LDS-DMA with the weights, because consecutive tiles of a workgroup can belong to different samples.
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
#ifdef CF_STREAM_TRACE
    int ablate;                     // trace build: CF_STREAM_ABLATE bits switch parts of a step off (wrong results, timing only)
#endif
};

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Development build only (make trace -> libcineflow_hip_trace.so, read by tools/stream_trace.py): per-wave shader-clock totals of the phases of a
// step.  s_memtime shares the LGKM counter with the LDS, so a stamp waits for the wave's outstanding LDS traffic -- the stamps sit where that
// queue is empty or about to be drained anyway.
#ifdef CF_STREAM_TRACE
__device__ unsigned long long g_stream_trace[256 * 8 * 16];
#define TR(i)                                                     \
    {                                                             \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
        tr_acc[i] += n_ - tr_t;                                   \
        if (tr_epi_step) tr_epi[i] += n_ - tr_t;                  \
        tr_t = n_;                                                \
    }
#define ABL(bit) ((g.ablate >> (bit)) & 1)      // 0 stores, 1 statistics, 2 input loads, 3 weight DMA, 4 MFMAs, 5 staging LDS writes, 6 operand reads
#else
#define TR(i)
#define ABL(bit) false
#endif

// a wave-uniform pointer the compiler cannot prove uniform, into SGPRs (an "s" asm constraint alone is not enforced for a 64-bit value)
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* q) {
    const unsigned long long a = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const T*)(((unsigned long long)hi << 32) | lo);
}

constexpr int ST_NW = 8, ST_NTW = 2, ST_TW = 32, ST_PW = 34, ST_REC = 80, ST_CK = 16, ST_NQ = 10;
constexpr int st_tile_rows(int wm) { return (ST_NW / wm) * ST_NTW; }       // 64 channels: 2 m-tiles x 4 wave groups x 2 rows = 8; 32 channels: 8 groups x 2 rows = 16

template <int WM, int PRE>
__global__ void __launch_bounds__(64 * ST_NW) conv_stream_kernel(const ConvParams p, const StreamGeom g, const _Float16* __restrict__ wpk) {
    constexpr int NW = ST_NW, NTW = ST_NTW, PW = ST_PW, REC = ST_REC, CK = ST_CK, NQ = ST_NQ;
    constexpr int NG = NW / WM;                  // wave groups per m-tile; a wave owns NTW = 2 vertically adjacent output rows (32-pixel n-tiles)
    constexpr int TH = NG * NTW;                 // output rows of a tile
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
    (void)nsteps;
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
        // Lane bits (cg: 2, quad parity: 1, row parity: 1) first: the 16 contiguous lanes that share an LDS cycle of a ds_write_b64 then cover all
        // 32 banks (bank = 20 * pixel + 2 * cg mod 32; +4 pixels = +16, +1 row = +8) -- with the quads of a row on consecutive lanes they met on
        // two bank pairs, 5-way (20 instead of 6 LDS cycles per store, ~900 of a step's 3456 clocks; profiles/r03_conv_stream.md).  The main
        // region holds quads 0-7 (lane bits 4-5 = quad / 2: a wave-load still reads whole 128-byte row segments), the edge region quads 8-9.
        static_assert(PH % 2 == 0 && NQ == 10, "task map below");
        constexpr int R0 = 64 * (PH / 2), R1 = 16 * (PH / 2);
        static_assert(R0 + R1 == NTASK && R0 % 64 == 0, "task map below");
        v_has[t] = task < NTASK;
        const int e = task - R0;
        const int cg = task & 3;
        v_q[t] = task < R0 ? (((task >> 2) & 1) | (((task >> 4) & 3) << 1)) : 8 + ((e >> 2) & 1);
        v_py[t] = task < R0 ? 2 * (task >> 6) + ((task >> 3) & 1) : 2 * (e >> 4) + ((e >> 3) & 1);
        if (!v_has[t]) { v_q[t] = 0; v_py[t] = 0; }
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
            is_off[t] = (v_has[t] && (unsigned)iy < (unsigned)p.H && (unsigned)xq < (unsigned)p.W && !ABL(2)) ? (unsigned)(iy * p.W + xq) * 4u : OOB;
        }
    };
    // PRE: element e of the [3][ctab_n] table image comes from element ct_src of the sample's [3][C1] table, by LDS-DMA (one dword per lane, two
    // rounds cover <= 1024 entries).  Channels past C1 (the zero-weight tail of the last chunk) re-read channel 0: any finite coefficients do.
    unsigned ct_src[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = tid + k * 64 * NW;
        const int which = e / ctab_n, c = e - which * ctab_n;
        ct_src[k] = (PRE && e < 3 * ctab_n && c < p.C1) ? (which * p.C1 + c) * 4 : 0;      // byte offset
    }
    auto issue_ctab = [&](int par) {      // the issue tile's table -> ctab[par]; retired with the weights by the counted wait of the step
        if (PRE) {
            const float* src = uniform_ptr(p.in_norm + (long)is_b * 3 * p.C1);
            const unsigned dst0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)(ctab + par * CT_SLOT));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)(k * 64 * NW + wave * 64) * 4u);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(ct_src[k]), "s"(dst), "s"(src) : "memory");
            }
        }
    };
    auto issue = [&](f32x4v (&stg)[VT][4], bool (&ok)[VT], int& par) {
        const bool in1 = is_c < g.c1_chunks;                                  // workgroup-uniform
        const unsigned cb = (unsigned)(in1 ? is_c : is_c - g.c1_chunks) * CK;
        const unsigned clim = (unsigned)(in1 ? p.C1 : p.C2);
        const __amdgpu_buffer_rsrc_t rs = in1 ? rs1 : rs2;
        par = is_tile & 1;
        // keep the registers about to be re-loaded alive up to this point: in the window between their last use (the staging slices of the
        // previous step) and this re-issue the compiler otherwise parks temporaries in them and fences each with a vmcnt wait (write-after-
        // write against loads it cannot prove retired), which drains the chunk in flight and the epilogue stores
#pragma unroll
        for (int t = 0; t < VT; ++t) asm volatile("" ::"v"(stg[t][0]), "v"(stg[t][1]), "v"(stg[t][2]), "v"(stg[t][3]));
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
    // One slice = the four channels of image column k of staging task t: conversion + two 8-byte LDS writes.  The slices of a chunk are spread
    // over the tap units of the step before the one that multiplies it (in the MFMA shadow); write_stage = all of them at once (prologue).
    // PRE: the {mean, scale, shift} quads of a task's four channels, read from the table with the task's FIRST slice and kept for its other three
    // (three ds_read_b128 per task and chunk instead of twelve)
    f32x4v pcm[VT], pca[VT], pcs[VT];
#pragma unroll
    for (int t = 0; t < VT; ++t) pcm[t] = pca[t] = pcs[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    auto write_stage_slice = [&](int slot, int chunk, const f32x4v (&stg)[VT][4], const bool (&ok)[VT], int par, const int t, const int k)
                                 __attribute__((always_inline)) {
        unsigned char* base = patch + slot * PATCH + v_lds[t];
        if (PRE && k == 0) {
            const float* ct = ctab + par * CT_SLOT;
            const int c0 = chunk * CK + (int)v_c4[t];
            pcm[t] = *reinterpret_cast<const f32x4v*>(ct + c0);
            pca[t] = *reinterpret_cast<const f32x4v*>(ct + ctab_n + c0);
            pcs[t] = *reinterpret_cast<const f32x4v*>(ct + 2 * ctab_n + c0);
        }
        const f32x4v cm = pcm[t], ca = pca[t], cs = pcs[t];
        // every loaded register is consumed UNCONDITIONALLY (conversion of all four columns), only the LDS writes of columns outside the
        // patch are masked: a load whose only uses sit behind a branch stays "maybe pending" for the compiler's wait insertion, which
        // then fences every later reuse of that register with a vmcnt wait that drains the chunks in flight
        f16x4 hi, lo;
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
            hi[j] = h;
            lo[j] = l;
        }
        asm volatile("" ::"v"(hi), "v"(lo));      // (pins the conversions here: the optimiser would sink them into the masked block)
        if (((v_mask[t] >> k) & 1u) && !ABL(5)) {
            *reinterpret_cast<f16x4*>(base + k * REC) = hi;
            *reinterpret_cast<f16x4*>(base + k * REC + CK * 2) = lo;
        }
    };
    auto write_stage = [&](int slot, int chunk, const f32x4v (&stg)[VT][4], const bool (&ok)[VT], int par) {
#pragma unroll
        for (int t = 0; t < VT; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) write_stage_slice(slot, chunk, stg, ok, par, t, k);
    };
    // ---- weights: chunk `chunk` of the WM m-tiles by LDS-DMA into slot `slot` (fragment f of the workgroup = (m-tile f / 18, tap x part f % 18));
    // issued from inline asm so that the compiler's wait insertion neither sees a pending LDS write nor drains the VM counter for it
    const unsigned lane16 = (unsigned)lane * 16u;
    // The two waves of a SIMD (w and w + 4) alternate: in even steps waves 0-3 fetch the whole weight slot, in odd steps waves 4-7.  An LDS-DMA
    // costs its wave 100-185 clocks of issue beside MFMAs and LDS reads (MI355X_MICROARCH.md, constants table); with both partners fetching in
    // the same units the SIMD's matrix pipe waited on both, now the partner without the duty keeps it fed.
    constexpr int NDMA = (WM * 18 + 3) / 4;                                 // fragments per duty wave and chunk (the last round covers some of the waves)
    auto issue_w1 = [&](int chunk, int slot, const int k) __attribute__((always_inline)) {
        const int f = (wave & 3) + k * 4;                                   // wave-uniform
        if (f < WM * 18) {
            const unsigned dst0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)(wl + slot * WSLOT));
            const int m = f / 18, j = f - m * 18;
            // scalar base + per-lane 32-bit offset: no 64-bit vector arithmetic (a v_mad_u64 the compiler hoisted to the head of the tile loop took a
            // register with a load in flight for its unused addend: vmcnt(0) once per tile)
            const _Float16* src = uniform_ptr(wpk + (((long)m * g.nchunk + chunk) * 18 + j) * 512);
            const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)f * 1024u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(dst), "s"(src) : "memory");
        }
    };
    auto issue_w = [&](int chunk, int slot) {      // prologue: waves 0-3
        if (wave < 4) {
#pragma unroll
            for (int k = 0; k < NDMA; ++k) issue_w1(chunk, slot, k);
        }
    };

    // ================================================================================================ MFMA side
    const int brow = ((ngrp * NTW) * PW + l31) * REC + half * 16;      // this lane's record in patch row `first output row of the wave`, tap column 0
    // acc: the tile being multiplied; hold: the finished tile whose epilogue runs inside the next tile's first step (a 16 * NTW register copy
    // per tile instead of two accumulator sets addressed by tile parity: ONE loop body, every long-lived value keeps its registers)
    f32x16 acc[NTW], hold[NTW];
    const bool do_stats = p.gn_ws != nullptr;
    const int cpg = do_stats ? p.Cout / p.gn_groups : 1;

    // deferred epilogue of a finished tile (accumulator set q, tile coordinates eb / ey0 / ex0): value = alpha * acc + bias, stored through a
    // per-sample buffer resource (a pixel outside the image / a channel beyond Cout is an out-of-range offset), statistics per channel
    int eb = 0, ey0 = 0, ex0 = 0, eb2 = 0;      // eb2: sample of the tile before the one whose epilogue is running
    struct Epi {                                // state of one running half-epilogue (8 accumulator registers): lives inside ONE step
        float tsum, tsq;                        // lane 16 + i of each 32-lane half: the half's totals of register 8 * h + i (i < 8)
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
        E.tsum = 0.f;
        E.tsq = 0.f;
    };
    // sum over the 32 lanes of each half of the wave, in lanes 16-31 / 48-63: four in-row butterflies and one row broadcast, all DPP adds (plain
    // VALU in the MFMA shadow; the ds_bpermute chain this replaces was ~2000 clocks of exposed LDS latency at the end of every epilogue step)
    auto half_sum = [&](float v) __attribute__((always_inline)) {
#define CF_DPP_ADD(ctrl, rows) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rows, 0xf, false))
        CF_DPP_ADD(0xB1, 0xf);        // quad_perm [1,0,3,2]
        CF_DPP_ADD(0x4E, 0xf);        // quad_perm [2,3,0,1]
        CF_DPP_ADD(0x141, 0xf);       // row_half_mirror
        CF_DPP_ADD(0x140, 0xf);       // row_mirror
        CF_DPP_ADD(0x142, 0xa);       // row_bcast15 into rows 1 and 3
#undef CF_DPP_ADD
        return v;
    };
    auto epi_regs = [&](Epi& E, const int r) __attribute__((always_inline)) {       // accumulator register r of every n-tile of the wave
        const float bv = biasl[mtl * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
        const unsigned soff = (unsigned)((r & 3) + 8 * (r >> 2)) * HW4;      // wave-uniform: an SGPR
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float v = p.alpha * hold[nt][r] + bv;
            if (!ABL(0)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), E.rs_out, E.e_off[nt], soff, 0);
            const float m = E.e_ok[nt] ? v : 0.f;
            s1 += m;
            s2 += m * m;
        }
        if (do_stats && !ABL(1)) {
            s1 = half_sum(s1);
            s2 = half_sum(s2);
            const bool mine = (lane & 15) == (r & 7);
            E.tsum = mine ? s1 : E.tsum;
            E.tsq = mine ? s2 : E.tsq;
        }
    };
    auto epi_stats_to_lds = [&](Epi& E, int q, const int h) {       // per-channel totals of registers 8h .. 8h+7 of this wave into red[q] (zeroed two tiles ago)
        if (!do_stats || ABL(1)) return;
        if ((lane & 24) == 16) {
            const int r = 8 * h + (lane & 7);
            const int cl = mtl * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            atomicAdd(&red[q * WM * 64 + 2 * cl], E.tsum);
            atomicAdd(&red[q * WM * 64 + 2 * cl + 1], E.tsq);
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
    setup_issue_tile();
    issue_w(0, 0);
    issue_ctab(0);
    ca_ = issue(sa, oka, para);        // chunk 0
    cb_ = issue(sb, okb, parb);        // chunk 1 (nchunk >= 2): stays in flight into step 0
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * VT) : "memory");
    lds_barrier();
    write_stage(0, ca_, sa, oka, para);
    lds_barrier();

#ifdef CF_STREAM_TRACE
    unsigned long long tr_acc[5] = {}, tr_epi[5] = {}, tr_t = __builtin_amdgcn_s_memtime();
    const unsigned long long tr_start = tr_t;
    bool tr_epi_step = false;
#endif
    // ================================================================================================ main loop
    // One step = one chunk = NU tap units of 4-5 MFMAs; ps = step parity (patch / weight slot and register set); nchunk is even, so a tile
    // starts on an even step.  Everything that is not an MFMA rides BETWEEN the tap units of a step, a slice per unit, so that the only
    // serial part of a step is its closing barrier (profiles/r03_conv_stream.md: with staging, weight DMA and epilogue as phases of their own
    // the MFMA pipe idled 45 % of a step, all eight waves being in the same phase at the same time):
    //   units 0 .. U_ISSUE-1  the weights of step s + 1 (LDS-DMA; the duty half of the waves, see issue_w1) [PRE: + the coefficient table of chunk s + 2's tile]
    //   unit  U_ISSUE         chunk s + 2 leaves (4 * VT loads per thread)
    //   unit  U_WS            counted wait: everything older than those loads has landed -- chunk s + 1 and the weights of step s + 1
    //   units U_WS ..         chunk s + 1 -> LDS, one (task, column) slice per unit
    //   units 8 .. 11       `epi` (the first two steps of a tile that has a predecessor): half of the predecessor's epilogue per step, two
    //                       accumulator registers per unit; the stores are YOUNGER than the loads in flight, so no counted wait drains them
    //                       before the next step's, a full step later.
    // `rp` = parity of the red[] buffer the running epilogue adds into (the finished tile's index & 1).
    constexpr int NR = NTW + 2, NU = 3 * NR;
    constexpr int NSL = 4 * VT, U_EPI = NU - 4;
    // weight fragment k of the duty wave leaves in unit dma_unit(k): two per unit first, then one per unit; all of them ahead of chunk s + 2
    constexpr int U_WS = NU - NSL;
    constexpr int U_ISSUE = VT == 1 ? 6 : 3;
    auto dma_unit = [](const int k) constexpr { return VT == 1 ? (k < 6 ? k / 2 : k - 3) : (k < 4 ? k / 2 : k - 2); };
    static_assert(dma_unit(NDMA - 1) < U_ISSUE && U_ISSUE < U_WS, "the weights leave before chunk s + 2, the counted wait comes after it");
    auto step = [&](const int ps, const bool epi, const int rp, const int wchunk, f32x4v (&st_issue)[VT][4], bool (&ok_issue)[VT],
                    int& par_issue, int& c_issue, const f32x4v (&st_write)[VT][4], const bool (&ok_write)[VT], const int par_write,
                    const int c_write) __attribute__((always_inline)) {
        // statistics of the tile before the finished one (red[rp ^ 1], complete since that tile's second step) to global memory, red[rp ^ 1]
        // cleared one barrier later
        if (ps == 0 && epi) epi_stats_to_global(rp ^ 1, eb2);
        if (ps == 1 && epi && tid < WM * 64) red[(rp ^ 1) * WM * 64 + tid] = 0.f;
        const bool tab_due = is_c == 0;
        if (tab_due) setup_issue_tile();
        Epi E;
        if (epi) epi_begin(E);
#ifdef CF_STREAM_TRACE
        tr_epi_step = epi;
#endif
        TR(0)
        const unsigned char* xb = patch + ps * PATCH;
        const unsigned char* wb = wl + ps * WSLOT + (mtl * 18) * 1024 + lane * 16;
        // Row sharing.  The wave's NTW output rows are vertically adjacent, so the B fragment of patch row ir at tap column kx serves output row
        // ir - ky for every ky: NTW + 2 fragment pairs per kx instead of 3 * NTW.  With the A fragments (three ky per kx) that is 0.78 KB of LDS
        // operand reads per MFMA instead of 1.0.  k order per accumulator: (chunk, kx, ky, channel).
        // unit u = (kx, ir): one B pair; requested one unit ahead (two-deep buffers), the A triple of the next kx with the last unit of this one
        // A(kx, ky) is used in units ir = ky, ky + 1 only, so one register set serves: A(kx + 1, 0) / A(kx + 1, 1) are requested in units ir = 2 / 3
        // of kx and A(kx, 2) in unit ir = 0 of kx, each two units ahead of its first use
        static_assert(NTW == 2, "the A schedule below is written for two rows per wave");
        f16x8 Aq[3][2], Bq[2][2];
        auto ldA = [&](const int kx, const int ky) __attribute__((always_inline)) {
            Aq[ky][0] = *reinterpret_cast<const f16x8*>(wb + ((ky * 3 + kx) * 2) * 1024);
            Aq[ky][1] = *reinterpret_cast<const f16x8*>(wb + ((ky * 3 + kx) * 2 + 1) * 1024);
        };
        auto ldB = [&](const int u, const int buf) __attribute__((always_inline)) {
            const unsigned char* rp_ = xb + brow + ((u % NR) * PW + (u / NR)) * REC;
            Bq[buf][0] = *reinterpret_cast<const f16x8*>(rp_);
            Bq[buf][1] = *reinterpret_cast<const f16x8*>(rp_ + CK * 2);
        };
        ldA(0, 0);
        ldB(0, 0);
        ldA(0, 1);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int kx = u / NR, ir = u % NR;
            if (!ABL(6)) {
                if (u + 1 < NU) ldB(u + 1, (u + 1) & 1);
                if (ir == 0) ldA(kx, 2);
                if (ir == 2 && kx < 2) ldA(kx + 1, 0);
                if (ir == 3 && kx < 2) ldA(kx + 1, 1);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int nt = ir - ky;
                if (nt >= 0 && nt < NTW && !ABL(4)) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aq[ky][1], Bq[u & 1][0], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aq[ky][0], Bq[u & 1][1], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aq[ky][0], Bq[u & 1][0], acc[nt], 0, 0, 0);
                }
            }
            // ---- the unit's slice of everything else
            if (wchunk >= 0 && !ABL(3)) {
                if ((wave >> 2) == ps) {
#pragma unroll
                    for (int k = 0; k < NDMA; ++k)
                        if (dma_unit(k) == u) issue_w1(wchunk, ps ^ 1, k);
                }
            }
            if (u == 0 && tab_due) issue_ctab(is_tile & 1);      // once per tile: with the first chunk of the issue tile (older than that chunk's loads: the counted wait retires it)
            if (u == U_ISSUE) c_issue = issue(st_issue, ok_issue, par_issue);   // unconditional (out-of-range offsets past the last chunk)
            if (u == U_WS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * VT) : "memory");
            if (u >= U_WS && u - U_WS < NSL) write_stage_slice(ps ^ 1, c_write, st_write, ok_write, par_write, (u - U_WS) / 4, (u - U_WS) % 4);
            if (epi && u >= U_EPI) {
                epi_regs(E, 8 * ps + 2 * (u - U_EPI));
                epi_regs(E, 8 * ps + 2 * (u - U_EPI) + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (epi) epi_stats_to_lds(E, rp, ps);
        TR(1)
        lds_barrier();
        TR(4)
    };

    bool have_prev = false;
    for (int i = 0; i < n_my; ++i) {
        int tb, ty0, tx0;
        decode(i, tb, ty0, tx0);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const int rp = (i - 1) & 1;               // red[] buffer of the finished tile i - 1
        const bool last_tile = i + 1 == n_my;
        for (int c = 0; c < g.nchunk; c += 2) {
            const bool first = c == 0, tail = c + 2 == g.nchunk;
            step(0, first && have_prev, rp, c + 1, sa, oka, para, ca_, sb, okb, parb, cb_);
            step(1, first && have_prev, rp, tail ? (last_tile ? -1 : 0) : c + 2, sb, okb, parb, cb_, sa, oka, para, ca_);
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
        for (int h = 0; h < 2; ++h) {
            Epi E;
            epi_begin(E);
#pragma unroll
            for (int r = 0; r < 8; ++r) epi_regs(E, 8 * h + r);
            epi_stats_to_lds(E, q, h);
        }
        lds_barrier();
        epi_stats_to_global(q, eb);
    }
#ifdef CF_STREAM_TRACE
    if (lane == 0) {
        unsigned long long* t = g_stream_trace + ((size_t)blockIdx.x * NW + wave) * 16;
        for (int k = 0; k < 5; ++k) { t[k] = tr_acc[k]; t[5 + k] = tr_epi[k]; }
        t[10] = __builtin_amdgcn_s_memtime() - tr_start;
        t[11] = (unsigned long long)nsteps;
        t[12] = (unsigned long long)n_my;
    }
#endif
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
    if (p.terms != 3) return false;       // the one-term ("mixed precision") product mode is built in conv_f16s.hip only
    if (!(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad_h == 1 && p.pad_w == 1)) return false;
    if ((p.Cout != 32 && p.Cout != 64) || p.scatter2x2 || p.res || p.act != CF_ACT_NONE || p.out_coff != 0 || p.out_ctotal != p.Cout || p.w_bstride) return false;
    if ((p.W & 3) || p.W < 32 || p.H < 16) return false;
    const int c1c = (p.C1 + 15) / 16, c2c = p.C2 > 0 ? (p.C2 + 15) / 16 : 0;
    const int nchunk = c1c + c2c;
    if (nchunk < 2 || (nchunk & 1)) return false;
    if (p.C2 > 0 && !p.x2) return false;
    if (((reinterpret_cast<uintptr_t>(p.x1) | reinterpret_cast<uintptr_t>(p.x2)) & 15) != 0) return false;
    const long HW = (long)p.H * p.W;
    if ((long)p.C1 * HW * 4 >= (1L << 31) || (long)p.C2 * HW * 4 >= (1L << 31) || (long)p.Cout * HW * 4 >= (1L << 30)) return false;
    if (p.in_norm && (p.C2 != 0 || 3 * nchunk * 16 > 2 * 512 || (reinterpret_cast<uintptr_t>(p.in_norm) & 3) != 0)) return false;
    // measured routing (profiles/r03_conv_stream.md): with the deferred normalisation the 64-channel layers are 8 % faster in conv_f16s (the
    // GELU conversion is VALU-bound next to the MFMAs in either kernel and conv_f16s spreads it over two resident workgroups); level 2 of the
    // knob takes every shape the kernel can run (tests)
    if (p.in_norm && p.Cout > 32 && stream_enabled() < 2) return false;
    if (p.gn_ws && (p.gn_groups <= 0 || p.Cout % p.gn_groups != 0)) return false;
    const int TH = st_tile_rows(p.Cout <= 32 ? 1 : 2);
    const long ntiles = (long)p.B * ((p.H + TH - 1) / TH) * ((p.W + 31) / 32);
    if (ntiles < 1024 || ntiles >= (1L << 31)) return false;      // short launches keep the one-tile-per-workgroup kernel (nothing to stream across)
    return true;
}

template <int WM, int PRE>
static int launch_stream(const ConvParams& p, const StreamGeom& g, const _Float16* wpk, hipStream_t s) {
    constexpr int PH = st_tile_rows(WM) + 2;
    const size_t lds_bytes = (size_t)2 * PH * ST_PW * ST_REC + (size_t)2 * WM * 18 * 1024 + sizeof(float) * (2 * WM * 64 + WM * 32 + (PRE ? 2 * 2 * 64 * ST_NW : 0));
    if (lds_bytes > 160 * 1024) { set_error("conv_stream: LDS budget exceeded"); return CF_ERR_ARG; }
    auto kern = conv_stream_kernel<WM, PRE>;
    static bool attr_set[64] = {};       // per device: a second GPU driven by the same process needs the 160 KB opt-in too
    const int dev = current_device_slot();
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev] = true;
    }
    // one persistent workgroup per CU (g.ntiles >= 1024), in whole groups of 8 (the kernel bands the tiles by blockIdx % 8 = XCD under
    // round-robin placement: speed only, any grid is correct)
    const unsigned nwg = (unsigned)device_cu_count();
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
    g.tiles_y = (p.H + st_tile_rows(WM) - 1) / st_tile_rows(WM);
    g.ntiles = p.B * g.tiles_y * g.tiles_x;
    g.c1_chunks = (p.C1 + 15) / 16;
    g.nchunk = g.c1_chunks + (p.C2 > 0 ? (p.C2 + 15) / 16 : 0);
#ifdef CF_STREAM_TRACE
    { const char* e = getenv("CF_STREAM_ABLATE"); g.ablate = e ? atoi(e) : 0; }
#endif
    if (p.in_norm) return WM == 1 ? launch_stream<1, 1>(p, g, wpk, s) : launch_stream<2, 1>(p, g, wpk, s);
    return WM == 1 ? launch_stream<1, 0>(p, g, wpk, s) : launch_stream<2, 0>(p, g, wpk, s);
}

}  // namespace cf

// A/B knob (also CF_CONV_STREAM=0|1 in the environment): 0 keeps every layer on conv_f16s_kernel's one-tile-per-workgroup shapes.  Returns the previous
// setting.  The two kernels agree to fp32 summation-order noise (<= 4e-6 of the output scale: taps summed in (kx, ky) instead of (ky, kx) order,
// tests/test_gpu_ops.py::test_conv_stream_*), NOT bitwise: the default level 1 changes network outputs within that bound against level 0.
extern "C" int cf_conv_stream_enable(int on) {
    const int prev = cf::stream_enabled();
    cf::g_stream = on < 0 ? 0 : (on > 2 ? 2 : on);
    return prev;
}

#ifdef CF_STREAM_TRACE
// development hook of the trace build: the per-wave phase totals of the last conv_stream launch -> dst[256 * 8 * 16]
extern "C" int cf_conv_stream_trace_read(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(cf::g_stream_trace), sizeof(unsigned long long) * 256 * 8 * 16);
}
#endif
