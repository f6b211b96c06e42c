// Local cost volume (CorrVolume, radius 4, dilation S in {1, 2, 4}) on the f16 MFMA, in a 2-D banded form.
//
//   out[b, (dy+4)*9 + (dx+4), y, x] = (1/C) sum_c cur[b,c,y,x] * prev[b,c,y + dy*S, x + dx*S]            (zero outside; dy, dx in -4..4)
//
// Why: the fp32-VALU kernel (corr.hip) runs at the board's power limit at 3.3-3.4 TB/s of algorithmic traffic: 81 * C multiply-adds per pixel on the
// vector ALUs, each fed from LDS (~39 pJ per MAC, profiles/r03_power_probe.txt).  The matrix pipe does the same products for a fraction of the energy
// even when most of a tile is wasted, PROVIDED the waste is kept down and the LDS footprint of a K = 32 instruction is kept at the vector kernel's:
//   * 2-D patches instead of row bands.  A dilated problem splits into S x S independent residue classes; inside a class the window is 9 x 9 dense.
//     M = a 4 x 4 patch of output positions (16 rows of the MFMA), N = a 4 x 4 patch of prev positions: the windows of an M-patch are covered by
//     3 x 3 = 9 N-patches, 81 of every 144 products are wanted (56 %; row bands: 16 x 9 of 16 x 32 per dy = 28 %).
//   * ONE v_mfma_f32_16x16x32_f16 per (M-patch, N-patch, 8 channels) with the whole hi/lo split product inside its K = 32:
//         A k-groups = [ah | al | ah | al],  B k-groups = [bh | bh | bl | bl]   ->  sum = (ah + al)(bh + bl)        (8 channels each)
//     i.e. all four split terms (one more than the convolutions' three) and a staged chunk of only 8 channels: 32 bytes per position, the
//     LDS footprint of the vector kernel's fp32 chunk.  72 matrix clocks per pixel and SIMD at C = 64 against 162 vector clocks.
//   * workgroup = 16 waves (four per SIMD, <= 128 registers each) = one tile of 8 class rows x 64 real columns of one y-class (all S x-classes); a
//     wave owns two vertically adjacent M-patches (8 rows x 4 positions of one x-class): 2 x 9 accumulator tiles = 72 registers; per chunk 2 A
//     reads + 12 B reads feed 18 MFMAs.  (The first build had 8 waves with 2 x 2 M-patches = 144 accumulators: with the two staging register sets
//     beside them every B read was waited for on the spot -- 24 exposed LDS latencies per chunk -- and the kernel ran at 2.9 TB/s.)
//   * LDS records [x-class][row][position] of 32 bytes {hi c0..7 | lo c0..7}, the two halves SWAPPED on odd rows and row pitches = 64 (mod 256):
//     the sixteen 16-byte reads of a 4 x 4 patch fragment (hi parts of two k-groups, or lo parts) then land on sixteen distinct slots.
//   * persistent workgroups (one per CU) walk a band of tiles of their XCD; the chunk stream runs across tile boundaries, ONE chunk of loads in
//     flight (a single staging register set: 64 KB per CU), two LDS stages, one barrier per chunk; the 81 planes of a tile leave through LDS (the
//     stage that is free at that moment) in two halves of four rows as whole 256-byte row segments.
//     What bounds it (profiles/r04_corr_mfma.txt): the bytes in flight.  A second staging set (two chunks in flight) needs 16 more registers than
//     the 128 a 1024-thread workgroup has per lane; tried twice (with every fragment register given up: one A and one B fragment live) it still
//     spills 40-60 B per lane INSIDE the chunk loop, the reloads collapse the counted waits to vmcnt(0), and the loop gets slower, not faster.
// Shapes: C % 8 == 0, W % 64 == 0, H % (8 S) == 0 (the three levels of the flow network: (64, 256, 4), (128, 128, 2), (256, 64, 1)); anything else
// stays on corr.hip.  Numerics: exact products of the split operands, fp32 accumulation: within 1e-5 of the fp32 / fp64 oracle on O(1) features.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "conv.h"
#include "profile.h"

namespace cf {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef CF_CLOCK_STAMPS
// clock build: shader clocks of wave 0 / lane 0 of every workgroup: [0] chunk loops, [1] epilogues, [2] tiles, [3] chunks
__device__ unsigned long long g_corr_phase[4];
#define CM_T() __builtin_amdgcn_s_memtime()
#define CM_ADD(i, v) do { if (tid == 0) atomicAdd(&g_corr_phase[i], (unsigned long long)(v)); } while (0)
#else
#define CM_T() 0ULL
#define CM_ADD(i, v) do { } while (0)
#endif

constexpr int CM_THREADS = 1024;
constexpr int CM_OROW = 68, CM_OPLANE = 4 * CM_OROW + 1;      // epilogue planes: 4 tile rows x 68 floats (+1: planes of neighbouring dx start on neighbouring banks)
constexpr int CM_EPI_BYTES = 81 * CM_OPLANE * 4;               // all 81 (dy, dx) planes of one half tile (4 rows): 88 452 B

template <int S>
struct CmGeom {
    static constexpr int NPOS = 64 / S;                // class positions per tile row
    static constexpr int PPOS = NPOS + 8;              // prev positions per class row (halo 4 each side)
    static constexpr int PRP = PPOS * 32 + 64;         // prev row pitch in bytes (= 64 mod 256)
    static constexpr int CRP = NPOS * 32 + 64;         // cur row pitch
    static constexpr int PCLS = 16 * PRP;              // bytes per x-class: 16 prev rows
    static constexpr int CCLS = 8 * CRP;               // 8 cur rows
    static constexpr int PREV_BYTES = S * PCLS;
    static constexpr int STAGE = PREV_BYTES + S * CCLS;
    // LDS map: [stage 0][gap][stage 1].  In the epilogue of a tile one stage is free (the other already holds the next tile's first chunk): the 81
    // planes of a half tile live in that stage + the gap, which are adjacent whichever stage it is.
    static constexpr int GAP = ((CM_EPI_BYTES - STAGE + 1023) / 1024) * 1024;
    static constexpr int LDS_BYTES = 2 * STAGE + GAP;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static constexpr int NPQ = 16 + 2 * S;             // 16-byte column quads per staged prev row
    static constexpr int NPT = 16 * NPQ;               // prev staging tasks (then 128 cur tasks)
    static_assert(PRP % 256 == 64 && CRP % 256 == 64, "row pitches must be 64 mod 256");
    static_assert(NPT <= 384 && (2 * NPT) % 64 == 0, "prev staging half-tasks fill whole waves (0..11)");
};

__device__ __forceinline__ void split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)v[j];
        hi[j] = h;
        lo[j] = (_Float16)(v[j] - (float)h);
    }
}

template <int S>
__global__ void __launch_bounds__(CM_THREADS, 4)
corr_volume_mfma_kernel(const float* __restrict__ cur, const float* __restrict__ prev, float* __restrict__ out, int B, int C, int H, int W,
                        int tiles_x, int tiles_y, int ntiles) {
    typedef CmGeom<S> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr unsigned OOB = 0x80000000u;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = H * W;
    const unsigned HW4 = (unsigned)HW * 4u;
    const int nchunk = C >> 3;

    // ---- this workgroup's tiles: XCD band, interleaved (corr.hip); tile order: x fastest, then tile row, then y-class, then sample
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    const int band_lo = (int)((long)ntiles * xcd / 8), band_hi = (int)((long)ntiles * (xcd + 1) / 8);
    if (band_lo + wg >= band_hi) return;
    const int n_my = (band_hi - band_lo - wg + nwg - 1) / nwg;
    struct Tile { int b, ry, Y0, X0r; };
    auto decode = [&](int i) {
        int id = band_lo + wg + i * nwg;
        Tile t;
        const int tx = id % tiles_x;
        id /= tiles_x;
        const int ty = id % tiles_y;
        id /= tiles_y;
        t.ry = id % S;
        t.b = id / S;
        t.X0r = tx * 64;
        t.Y0 = ty * 8;          // class rows
        return t;
    };


    // ================================================================================================================ staging (all threads)
    // half-task = (row, aligned quad of 4 real columns) x 4 of the chunk's 8 channels: 4 buffer_load_dwordx4, then per column 8 bytes of the hi
    // and 8 bytes of the lo half of the position's 32-byte record (the thread pair 2 i, 2 i + 1 fills one record).
    // waves 0..11 stage prev rows (NPT <= 384 tasks = 768 half-tasks), waves 12..15 the 128 cur tasks: the role -- and with it the buffer
    // descriptor -- is wave-uniform
    const bool is_prev = wave < 12;
    const int hk = tid & 1;                                                   // channel half
    const int tk = (is_prev ? tid : tid - 768) >> 1;
    const bool has_task = is_prev ? tk < G::NPT : true;
    const int trow = is_prev ? tk / G::NPQ : tk >> 4;                         // LDS row: 0..15 prev (class row Y0 - 4 + trow), 0..7 cur
    const int tq = is_prev ? tk % G::NPQ : tk & 15;
    // byte offset inside a stage of the record of the quad's first column; column k adds a compile-time constant (its x-class k % S and its
    // position (4 tq + k) / S = 4 tq / S + k / S: 4 tq is a multiple of S)
    const int w_lds0 = (is_prev ? trow * G::PRP + (4 * tq / S) * 32 : G::PREV_BYTES + trow * G::CRP + (4 * tq / S) * 32) + hk * 8;
    const int hi_off = (trow & 1) * 16;                                       // odd rows: {lo | hi}
    int is_tile = 0, is_ch = 0;                                               // issue cursor
    unsigned t_off = OOB;
    __amdgpu_buffer_rsrc_t rs;
    auto setup = [&]() {
        if (is_tile >= n_my) { t_off = OOB; return; }
        const Tile t = decode(is_tile);
        const long sample = (long)t.b * C * HW;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((is_prev ? prev : cur) + sample), 0, (int)((long)C * HW4), 0x00020000);
        const int y = S * (t.Y0 + trow - (is_prev ? 4 : 0)) + t.ry;
        const int x = t.X0r + 4 * tq - (is_prev ? 4 * S : 0);
        t_off = (has_task && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? (unsigned)(y * W + x) * 4u : OOB;
    };
    // timing experiments (make clock CABL=n -> -DCF_CORR_ABLATE=n; wrong results by design): 1 = no global loads (the staged registers keep
    // their first contents), 2 = no MFMAs / fragment reads, 3 = no output stores
    auto issue = [&](f32x4 (&stg)[4]) {
        const unsigned c0 = (unsigned)(is_ch * 8 + hk * 4) * HW4;
#if !defined(CF_CORR_ABLATE) || CF_CORR_ABLATE != 1
#pragma unroll
        for (int c = 0; c < 4; ++c) stg[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, t_off + c0 + (unsigned)c * HW4, 0, 0));
#else
        (void)c0;
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(stg[c]));
#endif
        if (++is_ch == nchunk) {
            is_ch = 0;
            ++is_tile;
            setup();
        }
    };
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    auto write = [&](int step, const f32x4 (&stg)[4]) {
        if (!has_task) return;
        unsigned char* base = lds + (step & 1) * (G::STAGE + G::GAP);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f16x4 hi, lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v = stg[c][k];
                const _Float16 h = (_Float16)v;
                hi[c] = h;
                lo[c] = (_Float16)(v - (float)h);
            }
            const int wk = w_lds0 + (k % S) * (is_prev ? G::PCLS : G::CCLS) + (k / S) * 32;
            *reinterpret_cast<f16x4*>(base + wk + hi_off) = hi;
            *reinterpret_cast<f16x4*>(base + wk + 16 - hi_off) = lo;
        }
    };

    // ================================================================================================================ MFMA fragments
    // wave -> x-class rx and position quad pq (4 positions); lane -> (patch position py, px; k-group kg)
    constexpr int QPC = 16 / S;                                               // position quads per class row
    const int rx = wave / QPC, pq = wave % QPC;
    const int pos = lane & 15, py = pos >> 2, px = pos & 3, kg = lane >> 4;
    // A k-groups {hi, lo, hi, lo}, B k-groups {hi, hi, lo, lo}; on odd rows the record is {lo | hi}
    const int a_part = ((kg & 1) ^ (py & 1)) * 16, b_part = ((kg >> 1) ^ (py & 1)) * 16;
    const int a_base = G::PREV_BYTES + rx * G::CCLS + py * G::CRP + (4 * pq + px) * 32 + a_part;
    const int b_base = rx * G::PCLS + py * G::PRP + (4 * pq + px) * 32 + b_part;
    const float invC = 1.0f / (float)C;

    f32x4 acc[2][3][3];          // [M-patch mr][a = N-patch row - mr][b = N-patch column]

    // ONE staging register set (a second one does not fit 128 registers: it spilled, and a scratch reload inside the chunk loop waits for
    // vmcnt(0), i.e. drains the prefetch): the loads of chunk st + 1 leave at the top of step st and are converted at its end, behind the 18
    // MFMAs of this wave and the whole step of the three other waves of its SIMD
    f32x4 sa[4] = {};
    setup();
    issue(sa);
    write(0, sa);
    __syncthreads();

    int step = 0;
    for (int i = 0; i < n_my; ++i) {
#pragma unroll
        for (int mr = 0; mr < 2; ++mr)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) acc[mr][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto chunk = [&](int st) __attribute__((always_inline)) {
            // st: the chunk being multiplied (stage st & 1); chunk st + 1 is loaded now and converted behind the MFMAs
            issue(sa);
            const unsigned char* sb_ = lds + (st & 1) * (G::STAGE + G::GAP);
            const f16x8 A0 = *reinterpret_cast<const f16x8*>(sb_ + a_base);
            const f16x8 A1 = *reinterpret_cast<const f16x8*>(sb_ + a_base + 4 * G::CRP);
            // N-patch row nr serves M-patch 0 as a = nr and M-patch 1 as a = nr - 1.  The twelve B fragments stream through TWO registers, one read
            // ahead of the MFMAs (with a single one every read was waited for on the spot: twelve LDS round trips in a row per wave and chunk)
            auto rdb = [&](int i) -> f16x8 { return *reinterpret_cast<const f16x8*>(sb_ + b_base + (i / 3) * 4 * G::PRP + (i % 3) * 128); };
#if !defined(CF_CORR_ABLATE) || CF_CORR_ABLATE != 2
            f16x8 Bq[2];
            Bq[0] = rdb(0);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int nr = i / 3, nc = i % 3;
                if (i + 1 < 12) Bq[(i + 1) & 1] = rdb(i + 1);
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 Bf = Bq[i & 1];
                if (nr <= 2) acc[0][nr][nc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, Bf, acc[0][nr][nc], 0, 0, 0);
                if (nr >= 1) acc[1][nr - 1][nc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, Bf, acc[1][nr - 1][nc], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#else
            asm volatile("" :: "v"(A0), "v"(A1));
#endif
            write(st + 1, sa);
            __syncthreads();
        };
        const unsigned long long tm0 = CM_T();
        for (int ch = 0; ch < nchunk; ++ch, ++step) chunk(step);

        const unsigned long long tm1 = CM_T();
        // ---- epilogue: two half tiles (four rows each): the wanted (m, n) pairs of every accumulator tile -> LDS planes [dx][row][column] ->
        // whole row segments.  Lane (n = pos: prev patch position npy = py, npx = px; kg = output row inside the M-patch), register r = output column.
        const Tile tl = decode(i);
        float* const ob = out + (long)tl.b * 81 * HW;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)(81L * HW4), 0x00020000);
        // `step` chunks are done: the stage of chunk step - 1 is free, the other one holds the next tile's first chunk
        float* const obuf = reinterpret_cast<float*>(lds + (((step - 1) & 1) ? G::STAGE : 0));
        // plane of a pair = (dy + 4) * 9 + (dx + 4) with dy = 4 (a - 1) + py - kg, dx = 4 (b - 1) + px - r: the lane part of the float index is
        // ((py - kg) * 9 + px) * PLANE + kg * OROW + (first column of the wave); a, b, r add compile-time constants
        float* const eb = obuf + ((py - kg) * 9 + px) * CM_OPLANE + kg * CM_OROW + S * 4 * pq + rx;
#pragma unroll
        for (int mr = 0; mr < 2; ++mr) {
            // every lane writes its wanted pairs of M-patch mr at once (a = 0 needs py >= kg, a = 2 py <= kg; b = 0 needs px >= r, b = 2 px <= r):
            // three quarters of the lanes are active per write.  (The first version walked the nine dy planes one after the other with a
            // per-lane choice of a: a quarter of the lanes active, 18 barriers, 25 000 clocks per tile.)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const bool a_ok = a == 1 || (a == 0 ? py >= kg : py <= kg);
                float* const ea = eb + (4 * a * 9) * CM_OPLANE;             // (one base per a: the constants below stay inside the offset field)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = a_ok && (b == 1 || (b == 0 ? px >= r : px <= r));
                        if (ok) ea[(4 * b - r) * CM_OPLANE + S * r] = acc[mr][a][b][r] * invC;
                    }
            }
            __syncthreads();
            // 81 planes x 4 rows x 16 column quads = 5 184 sixteen-byte stores
#pragma unroll 1
            for (int e = tid; e < 81 * 64; e += CM_THREADS) {
                const int plane = e >> 6, rem = e & 63, row = rem >> 4, c4 = rem & 15;
                const float* src = obuf + plane * CM_OPLANE + row * CM_OROW + 4 * c4;
                const f32x4 v = {src[0], src[1], src[2], src[3]};
                const int y = S * (tl.Y0 + 4 * mr + row) + tl.ry, x = tl.X0r + 4 * c4;
#if !defined(CF_CORR_ABLATE) || CF_CORR_ABLATE != 3
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, (unsigned)((plane * H + y) * W + x) * 4u, 0, 0);
#else
                asm volatile("" :: "v"(v), "v"(y), "v"(x));
#endif
            }
            __syncthreads();
        }
        CM_ADD(0, tm1 - tm0); CM_ADD(1, CM_T() - tm1); CM_ADD(2, 1); CM_ADD(3, nchunk);
    }
}

int g_corr_mfma = -1;
int corr_mfma_enabled() {
    if (g_corr_mfma < 0) { const char* e = getenv("CF_CORR_MFMA"); g_corr_mfma = e ? atoi(e) : 1; }
    return g_corr_mfma;
}

template <int S>
int launch_corr_mfma(const float* cur, const float* prev, float* out, int B, int C, int H, int W, hipStream_t s) {
    typedef CmGeom<S> G;
    const size_t lds_bytes = (size_t)G::LDS_BYTES;
    auto kern = corr_volume_mfma_kernel<S>;
    static bool attr_set[64] = {};
    const int dev = current_device_slot();
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev] = true;
    }
    const int tiles_x = W / 64, tiles_y = (H / S) / 8;
    const long nt = (long)B * S * tiles_y * tiles_x;
    if (nt >= (1L << 31)) { set_error("corr_volume: too many tiles"); return CF_ERR_ARG; }
    const long ncu = device_cu_count();
    const unsigned nwg = (unsigned)(nt >= ncu ? ncu : ((nt + 7) / 8) * 8);
    const double bytes = 4.0 * (double)B * H * W * (2.0 * C + 81.0);
    const int kid = S == 1 ? PK_CORRVOL_S1 : (S == 2 ? PK_CORRVOL_S2 : PK_CORRVOL_S4);
    hipEvent_t e0, e1;
    if (profile_on() && profile_events(kid, bytes, &e0, &e1))
        hipExtLaunchKernelGGL(kern, dim3(nwg), dim3(CM_THREADS), lds_bytes, s, e0, e1, 0, cur, prev, out, B, C, H, W, tiles_x, tiles_y, (int)nt);
    else
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(CM_THREADS), lds_bytes, s, cur, prev, out, B, C, H, W, tiles_x, tiles_y, (int)nt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("corr_volume (MFMA) launch failed: ") + hipGetErrorString(e)); return CF_ERR_LAUNCH; }
    return CF_OK;
}

}  // namespace

// radius 4, dilation 1 / 2 / 4, C % 16 == 0, W % 64 == 0, H % (8 S) == 0, 16-byte aligned inputs, one sample of the inputs < 2 GiB
bool corr_mfma_applicable(int C, int H, int W, int stride, const float* cur, const float* prev) {
    if (!corr_mfma_enabled()) return false;
    if (!(stride == 1 || stride == 2 || stride == 4)) return false;
    if ((C & 15) != 0 || (W & 63) != 0 || H % (8 * stride) != 0) return false;
    if ((long)C * H * W * 4 >= (1L << 31) || 81L * H * W * 4 >= (1L << 31)) return false;
    return ((reinterpret_cast<uintptr_t>(cur) | reinterpret_cast<uintptr_t>(prev)) & 15) == 0;
}

int launch_corr_volume_mfma(const float* cur, const float* prev, float* out, int B, int C, int H, int W, int stride, hipStream_t s) {
    if (stride == 1) return launch_corr_mfma<1>(cur, prev, out, B, C, H, W, s);
    if (stride == 2) return launch_corr_mfma<2>(cur, prev, out, B, C, H, W, s);
    return launch_corr_mfma<4>(cur, prev, out, B, C, H, W, s);
}

}  // namespace cf

// A/B knob without a reference counterpart: 0 keeps every CorrVolume call on the fp32 vector kernel of corr.hip, 1 (default, also CF_CORR_MFMA)
// takes the MFMA kernel where the shape allows.  Returns the previous setting.
#ifdef CF_CLOCK_STAMPS
extern "C" int cf_debug_corr_phases(unsigned long long* out4) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return CF_ERR_LAUNCH;
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(cf::g_corr_phase), sizeof(z)) != hipSuccess) return CF_ERR_LAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(cf::g_corr_phase), z, sizeof(z)) != hipSuccess) return CF_ERR_LAUNCH;
    return CF_OK;
}
#endif

extern "C" int cf_corr_mfma_enable(int on) {
    const int prev = cf::corr_mfma_enabled();
    cf::g_corr_mfma = on ? 1 : 0;
    return prev;
}
