// Correlation kernels: the local cost volume (CorrVolume), the RAFT all-pairs pyramid + lookup, convex upsampling.
#include "conv.h"
#include "profile.h"

namespace cf {

// =====================================================================================================================
// Local cost volume, radius 4 (9x9 = 81 displacements), dilation s in {1,2,4}:
//   out[b,(dy+4)*9+(dx+4),y,x] = (1/C) sum_c cur[b,c,y,x] * prev[b,c,y+dy*s,x+dx*s]      (zero outside)
// Algorithmic traffic: read cur+prev once (2*C*H*W*4 B), write 81*H*W*4 B; 2*81*C flops per pixel -> 12-17 flop/B, at the
// fp32-VALU / HBM ridge of gfx950.  A dilated problem splits into independent residue classes (pixel (y,x) only meets prev
// pixels with the same (y mod s, x mod s)); thread = (dy, tile row, 8 same-class pixels) keeps 72 accumulators and per channel
// 6 ds_read_b128 feed 72 FMAs.  What the ablation of the first (non-persistent, 4-row-tile) version showed
// (profiles/r01_corr_ablation.md): its 4.5 FMA waves sat unevenly on the 4 SIMDs, every 4-channel chunk paid a barrier, the
// ds_read_b128 of the stride-4 layout ran at half rate on bank conflicts, each store instruction scattered 4-B pieces, and
// a per-lane choice between the cur/prev descriptors was compiled into a waterfall loop around every load.  Here:
//   * tile = 7 sub-rows x 64 real columns of one y-class: 9 dy x 7 rows x 8 octets = 504 FMA threads = 8 full waves (two per
//     SIMD), plus 4 staging waves (one per SIMD) -> 12 waves, ONE workgroup per CU, <=168 VGPRs;
//   * the workgroup is persistent: it walks a band of tiles of its XCD, the staging waves stream chunk after chunk (8
//     channels per chunk, two register sets = two chunks in flight, two LDS stages) straight across tile boundaries, so
//     the epilogue of tile i overlaps the loads of tile i+1;
//   * LDS rows are block-interleaved ([row][8-position block][x-class][8 floats], odd row stride in 16-B units) and the
//     lanes of a wave are permuted so that each 16-lane ds_read_b128 group covers 2 rows x 8 octets: conflict free for
//     every dilation;
//   * epilogue through LDS in 9 passes (one dx each, every thread writes 8 values), double buffered: one barrier per
//     pass, full 256-B row segments per store.
// =====================================================================================================================
constexpr int P7_ROWS = 7;
constexpr int P7_CC = 8;
constexpr int P7_FMA_THREADS = 512;
constexpr int P7_THREADS = 768;
constexpr int P7_OROW = 68, P7_OPLANE = P7_ROWS * P7_OROW;   // 476 = 28 (mod 64): 4 banks per (dy,row) step -> conflict-free scalar writes

// Stride 4: which (prev row, 8-position block) each 8-lane group of a staging wave-task handles.  The de-interleaving writes are
// scalar (ds_write2_b32: banks b..b+7 and b+8..b+15 per lane group); with lanes walking the groups in order they pile up to 5-deep
// on the same banks and the FMA waves' reads starve behind them (SQ_LDS_BANK_CONFLICT 42 % of LDS cycles).  In this assignment the
// eight 16-bank windows of every write instruction tile the 64 banks exactly twice (entry = row*4 + block, 255 = idle).
__device__ constexpr unsigned char P7_PREV_MAP4[6][8] = {{18, 37, 48, 0, 32, 53, 38, 54},  {9, 26, 8, 50, 33, 16, 25, 34},
                                                         {40, 58, 10, 42, 56, 46, 24, 255}, {2, 14, 52, 4, 22, 36, 49, 255},
                                                         {20, 21, 12, 13, 6, 30, 5, 29},    {45, 57, 1, 17, 28, 44, 41, 255}};
__device__ constexpr unsigned char P7_CUR_MAP4[2][8] = {{0, 1, 16, 17, 8, 9, 24, 25}, {4, 5, 12, 13, 20, 21, 255, 255}};

template <int S>
__global__ void __launch_bounds__(P7_THREADS, 3) corr_volume_p7_kernel(const float* __restrict__ cur, const float* __restrict__ prev,
                                                                    float* __restrict__ out, int B, int C, int H, int W, int tiles_x,
                                                                    int tiles_y, int ntiles) {
    constexpr int NI = 64 / S;                  // same-class positions per tile row
    constexpr int NBP = (NI + 8) / 8;           // 8-position blocks per prev class row (halo 4 each side)
    constexpr int PRS = NBP * S * 8 + 4;        // prev row stride (floats): 76 / 84 / 100 -> 19 / 21 / 25 sixteen-byte units (odd)
    constexpr int CRS = 68;                     // cur row stride: 17 units (odd)
    constexpr int PROWS = P7_ROWS + 8;
    constexpr int PREV_C = PROWS * PRS;
    constexpr int SLAB = PREV_C + P7_ROWS * CRS;   // floats per channel: [prev | cur]
    constexpr int STAGE = P7_CC * SLAB;
    constexpr int PGRP = (64 + 8 * S) / 4;      // float4 groups per staged prev row
    constexpr int NPG = PROWS * PGRP;           // 270 / 300 / 360 <= 6 wave-tasks
    constexpr int NCG = P7_ROWS * 16;           // 112 <= 2 wave-tasks
    static_assert(NPG <= 6 * 64 && NCG <= 2 * 64, "wave-task map");
    constexpr int OBUF = 9 * P7_OPLANE;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE + 2 * OBUF];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));

    // ---- this workgroup's tiles: XCD x (= blockIdx % 8) owns the contiguous band [ntiles*x/8, ntiles*(x+1)/8) of the tile list
    // (x fastest, then tile row, then y-class, then sample: neighbours share halos in that XCD's L2)
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    const int band_lo = (int)((long)ntiles * xcd / 8), band_hi = (int)((long)ntiles * (xcd + 1) / 8);
    const int n_my = (band_hi - band_lo - wg + nwg - 1) / nwg;   // tiles band_lo + wg + i*nwg
    if (band_lo + wg >= band_hi) return;
    const int HW = H * W;
    const unsigned HW4 = (unsigned)HW * 4u;
    const int nchunk = (C + P7_CC - 1) / P7_CC;
    const int nsteps = n_my * nchunk;
    const int tid = threadIdx.x;
    const bool stager = tid >= P7_FMA_THREADS;
    constexpr unsigned OOB = 0x80000000u;

    struct Tile { int b, ry, ytile, xtile; };
    auto decode = [&](int i) {
        int id = band_lo + wg + i * nwg;
        Tile t;
        const int tx = id % tiles_x;
        id /= tiles_x;
        const int ty = id % tiles_y;
        id /= tiles_y;
        t.ry = id % S;
        t.b = id / S;
        t.xtile = tx * 64;
        t.ytile = ty * P7_ROWS;   // in class rows
        return t;
    };
    // position p of class cl inside a block-interleaved row
    auto blk = [](int p, int cl) { return ((p >> 3) * S + cl) * 8 + (p & 7); };

    // FMA waves only.  Accumulators are paired along the anti-diagonal: a2[q][e] = (acc[2q][e+1], acc[2q+1][e]) -- both take the
    // SAME prev element pv[2q+e+1], so one v_pk_fma_f32 with a = (cv[2q], cv[2q+1]) (an aligned register pair) and b = that
    // element broadcast covers them: no shifted operand copies.  s0[q] = acc[2q][0], s8[q] = acc[2q+1][8] are the unpaired ends.
    f32x2 a2[4][8];
    float s0[4], s8[4];
    if (stager) {
        // =================================================================================================== staging waves
        const int swave = __builtin_amdgcn_readfirstlane((tid - P7_FMA_THREADS) >> 6);   // 0..3
        int t_lds[2];          // float index inside a channel slab of element 0 of the float4; -1: no task
        int t_row[2], t_col[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int wt = swave * 2 + k;
            const int grp = (tid & 63) + 64 * (wt < 6 ? wt : wt - 6);
            t_lds[k] = -1;
            t_row[k] = t_col[k] = 0;
            if (S == 4) {
                const int ent = wt < 6 ? P7_PREV_MAP4[wt][(tid & 63) >> 3] : P7_CUR_MAP4[wt - 6][(tid & 63) >> 3];
                if (ent != 255) {
                    const int row = ent >> 2, p = (ent & 3) * 8 + (tid & 7);   // class position; real column offset 4p
                    t_row[k] = wt < 6 ? row - 4 : row;
                    t_col[k] = wt < 6 ? 4 * p - 16 : 4 * p;
                    t_lds[k] = (wt < 6 ? row * PRS : PREV_C + row * CRS) + blk(p, 0);
                }
            } else if (wt < 6) {
                if (grp < NPG) {
                    const int pr = grp / PGRP, col = (grp - pr * PGRP) * 4;   // real column offset inside the staged row
                    t_row[k] = pr - 4;
                    t_col[k] = col - 4 * S;
                    t_lds[k] = pr * PRS + blk(col / S, col % S);
                }
            } else if (grp < NCG) {
                const int rr = grp >> 4, col = (grp & 15) * 4;
                t_row[k] = rr;
                t_col[k] = col;
                t_lds[k] = PREV_C + rr * CRS + blk(col / S, col % S);
            }
        }
        // issue-side cursor (runs three steps ahead of the FMA waves)
        int is_tile = 0, is_ch = 0;
        unsigned t_off[2];
        __amdgpu_buffer_rsrc_t rs_prev, rs_cur;
        // Past the last tile the cursor keeps running with out-of-range offsets (the buffer loads return zeros without touching memory):
        // every issue / write below is UNCONDITIONAL, which is what lets the compiler count the loads in flight -- with a conditional
        // issue its s_waitcnt pass fell back to vmcnt(0) in write(), i.e. the set issued a moment ago had to land before the barrier and
        // the three-step lead of this cursor collapsed to one exposed HBM latency per step
        auto setup = [&]() {
            if (is_tile >= n_my) {
                t_off[0] = t_off[1] = OOB;
                return;
            }
            const Tile t = decode(is_tile);
            const long sample = (long)t.b * C * HW;
            rs_prev = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(prev + sample), 0, (int)((long)C * HW4), 0x00020000);
            rs_cur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cur + sample), 0, (int)((long)C * HW4), 0x00020000);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int y = S * (t.ytile + t_row[k]) + t.ry, x = t.xtile + t_col[k];
                t_off[k] = (t_lds[k] >= 0 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? (unsigned)(y * W + x) * 4u : OOB;
            }
        };
        setup();
        f32x4 sa[2][P7_CC], sb[2][P7_CC];
        auto issue = [&](f32x4 (&stg)[2][P7_CC]) {
            const unsigned c0 = (unsigned)(is_ch * P7_CC) * HW4;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int c = 0; c < P7_CC; ++c) {
                    const unsigned off = t_off[k] + c0 + (unsigned)c * HW4;
                    stg[k][c] = (swave * 2 + k < 6) ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_prev, off, 0, 0))
                                                    : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_cur, off, 0, 0));
                }
            if (++is_ch == nchunk) {
                is_ch = 0;
                ++is_tile;
                setup();
            }
        };
        auto write = [&](int step, const f32x4 (&stg)[2][P7_CC]) {
            float* base = lds + (step & 1) * STAGE;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (t_lds[k] < 0) continue;
#pragma unroll
                for (int c = 0; c < P7_CC; ++c) {
                    float* d = base + c * SLAB + t_lds[k];
                    if (S == 1) {
                        *reinterpret_cast<f32x4*>(d) = stg[k][c];
                    } else if (S == 2) {   // columns col..col+3 = classes 0,1,0,1 at positions p0,p0,p0+1,p0+1 (p0 even: same block)
                        *reinterpret_cast<f32x2*>(d) = f32x2{stg[k][c][0], stg[k][c][2]};
                        *reinterpret_cast<f32x2*>(d + 8) = f32x2{stg[k][c][1], stg[k][c][3]};
                    } else {               // S == 4: one element per class, same position
                        d[0] = stg[k][c][0]; d[8] = stg[k][c][1]; d[16] = stg[k][c][2]; d[24] = stg[k][c][3];
                    }
                }
            }
        };
        auto tile_end = [&](int step) {   // the FMA waves' epilogue has 9 barriers
            if (step % nchunk == nchunk - 1)
                for (int d = 0; d < 9; ++d) __syncthreads();
        };
        issue(sa);
        issue(sb);
        write(0, sa);
        issue(sa);
        __syncthreads();
        // FMA waves work on `step` (stage step&1); set B holds step+1, set A step+2
        int step = 0;
        for (; step + 1 < nsteps; step += 2) {
            write(step + 1, sb);
            issue(sb);
            __syncthreads();
            tile_end(step);
            write(step + 2, sa);
            issue(sa);
            __syncthreads();
            tile_end(step + 1);
        }
        if (step < nsteps) {
            write(step + 1, sb);
            __syncthreads();
            tile_end(step);
        }
        return;
    }

    // ======================================================================================================= FMA waves
    // lane permutation: the hardware serves a ds_read_b128 in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32;
    // logical lanes 16g..16g+15 (= 2 rows x 8 octets) are mapped onto hardware group g
    const int lane = tid & 63, lo = lane & 31;
    const int m = lo < 4 ? lo : lo < 12 ? lo + 12 : lo < 16 ? lo - 8 : lo < 20 ? lo + 8 : lo < 28 ? lo - 12 : lo;
    const int t = (tid & ~63) + (lane & 32) + m;
    const bool compute = t < 9 * P7_ROWS * 8;
    const int dy = compute ? t / (P7_ROWS * 8) : 0;   // 0..8 (displacement dy-4)
    const int rem = t % (P7_ROWS * 8);
    const int r = rem >> 3;                     // tile row 0..6
    const int u = rem & 7;
    const int rx = u % S;                       // x residue class
    const int o = u / S;                        // octet of class positions
    const int prev_off = (r + dy) * PRS + (o * S + rx) * 8;       // + c*SLAB ; blocks o and o+1 (the second S*8 floats further)
    const int cur_off = PREV_C + r * CRS + (o * S + rx) * 8;      // + c*SLAB
    const float invC = 1.0f / (float)C;

    __syncthreads();
    int step = 0;
    for (int i = 0; i < n_my; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s0[q] = s8[q] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a2[q][e] = f32x2{0.f, 0.f};
        }
        for (int ch = 0; ch < nchunk; ++ch, ++step) {
            if (compute) {
                const float* xb = lds + (step & 1) * STAGE;
                // explicit software pipeline: the operands of channel c+1 are requested before the 72 FMAs of channel c (left to
                // itself the scheduler sinks the ds_reads next to their use and every channel pays the LDS latency: 2 waves per
                // SIMD cannot hide it)
                float4 pa[4], ca[2], pb[4], cb[2];
                auto ldch = [&](int c, float4 (&pq)[4], float4 (&cq)[2]) {
                    const float4* pp = reinterpret_cast<const float4*>(xb + c * SLAB + prev_off);
                    const float4* cp = reinterpret_cast<const float4*>(xb + c * SLAB + cur_off);
                    pq[0] = pp[0]; pq[1] = pp[1]; pq[2] = pp[2 * S]; pq[3] = pp[2 * S + 1];
                    cq[0] = cp[0]; cq[1] = cp[1];
                };
                // One row pair q: 8 v_pk_fma_f32 (a = (cv[2q], cv[2q+1]), b = pv[2q+e+1] broadcast by op_sel) + 2 v_fmac_f32 for the ends.
                // Written as asm so that the 40 instructions per channel stay exactly these (the scheduler otherwise re-pairs the
                // accumulators with register copies and spills) and, with the memory clobber, so that the operand reads of the NEXT
                // channel stay in front of them.
                auto fmach = [&](const float4 (&pq)[4], const float4 (&cq)[2]) {
                    const f32x2 pp[8] = {f32x2{pq[0].x, pq[0].y}, f32x2{pq[0].z, pq[0].w}, f32x2{pq[1].x, pq[1].y}, f32x2{pq[1].z, pq[1].w},
                                         f32x2{pq[2].x, pq[2].y}, f32x2{pq[2].z, pq[2].w}, f32x2{pq[3].x, pq[3].y}, f32x2{pq[3].z, pq[3].w}};
                    const f32x2 cv2[4] = {f32x2{cq[0].x, cq[0].y}, f32x2{cq[0].z, cq[0].w}, f32x2{cq[1].x, cq[1].y}, f32x2{cq[1].z, cq[1].w}};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        asm volatile(
                            "v_fmac_f32_e32 %8, %18, %16\n\t"
                            "v_pk_fma_f32 %0, %10, %11, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                            "v_pk_fma_f32 %1, %10, %12, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                            "v_pk_fma_f32 %2, %10, %12, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                            "v_pk_fma_f32 %3, %10, %13, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                            "v_pk_fma_f32 %4, %10, %13, %4 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                            "v_pk_fma_f32 %5, %10, %14, %5 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                            "v_pk_fma_f32 %6, %10, %14, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                            "v_pk_fma_f32 %7, %10, %15, %7 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                            "v_fmac_f32_e32 %9, %19, %17"
                            : "+v"(a2[q][0]), "+v"(a2[q][1]), "+v"(a2[q][2]), "+v"(a2[q][3]), "+v"(a2[q][4]), "+v"(a2[q][5]), "+v"(a2[q][6]),
                              "+v"(a2[q][7]), "+v"(s0[q]), "+v"(s8[q])
                            : "v"(cv2[q]), "v"(pp[q]), "v"(pp[q + 1]), "v"(pp[q + 2]), "v"(pp[q + 3]), "v"(pp[q + 4]), "v"(pp[q].x), "v"(pp[q + 4].y),
                              "v"(cv2[q].x), "v"(cv2[q].y)
                            : "memory");
                    }
                };
                ldch(0, pa, ca);
#pragma unroll
                for (int c = 0; c < P7_CC; c += 2) {
                    ldch(c + 1, pb, cb);
                    fmach(pa, ca);
                    if (c + 2 < P7_CC) ldch(c + 2, pa, ca);
                    fmach(pb, cb);
                }
            }
            __syncthreads();
        }
        // ---- epilogue: pass d moves the dx = d-4 planes (9 dy x 7 rows x 64 columns) through LDS and out as whole row segments
        const Tile tl = decode(i);
        float* ob = out + (long)tl.b * 81 * HW;
        int etid = tid;
        asm volatile("" : "+v"(etid));   // keeps the store-phase index arithmetic out of the registers that live across the FMA loop
#pragma unroll
        for (int d = 0; d < 9; ++d) {
            float* obuf = lds + 2 * STAGE + (d & 1) * OBUF;
            if (compute) {
                float* o0 = obuf + dy * P7_OPLANE + r * P7_OROW + S * 8 * o + rx;
                float v[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] = (d == 0 ? s0[q] : a2[q][d > 0 ? d - 1 : 0].x) * invC;
                    v[2 * q + 1] = (d == 8 ? s8[q] : a2[q][d < 8 ? d : 0].y) * invC;
                }
                if (S == 1) {
                    *reinterpret_cast<f32x4*>(o0) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(o0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o0[S * j] = v[j];
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = etid + q * P7_FMA_THREADS;
                if (e < 9 * P7_ROWS * 16) {
                    const int plane = e / (P7_ROWS * 16), rm = e % (P7_ROWS * 16), row = rm >> 4, c4 = rm & 15;
                    const int y = S * (tl.ytile + row) + tl.ry, x = tl.xtile + 4 * c4;
                    if (y < H && x < W)
                        *reinterpret_cast<f32x4*>(ob + (long)(plane * 9 + d) * HW + (long)y * W + x) =
                            *reinterpret_cast<const f32x4*>(obuf + plane * P7_OPLANE + row * P7_OROW + 4 * c4);
                }
            }
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

// The same lookup for radius 4 / N % 64 == 0, laid out for the memory system.  The flat kernel above gives every output its own
// thread: neighbouring lanes are neighbouring PIXELS, i.e. different correlation planes (4 KiB apart at level 0), so each of its four
// gathers touches 64 cache lines and each store scatters 4-byte pieces 4 KiB apart (measured 427 GB/s of algorithmic traffic, 5 % of
// the HBM peak).  Here a workgroup owns 64 consecutive pixels of one sample, wave l owns pyramid level l, and inside a wave the lanes
// are (pixel 0..6) x (x-offset i 0..8): the taps of one gather lie side by side in 7 planes, and the 324 x 64 outputs leave through
// LDS as whole 256-byte rows.  Arithmetic per output is unchanged (same rounding path as the flat kernel).
constexpr int LK_D = 9, LK_G = 7;      // offsets per axis, pixels per wave pass
template <int LK_PIX>                  // pixels per workgroup: 64 (256-byte rows, 84 KB of LDS) or 32 (128-byte rows, three workgroups per CU)
__global__ void __launch_bounds__(256) corr_lookup_tiled_kernel(const float* __restrict__ pyr, const float* __restrict__ coords,
                                                               float* __restrict__ out, int B, int H, int W, int levels) {
    extern __shared__ float stage[];                 // [levels * 81][LK_PIX + 1]
    const int N = H * W;
    const int tiles = N / LK_PIX;
    const int b = blockIdx.x / tiles, n0 = (blockIdx.x % tiles) * LK_PIX;
    const int lane = threadIdx.x & 63, l = threadIdx.x >> 6;       // wave = pyramid level (levels <= 4)
    const int pl = lane / LK_D, i = lane - pl * LK_D;              // pixel inside the pass, x-offset index
    if (l < levels) {
        long off = 0;
        for (int q = 0; q < l; ++q) off += (long)B * N * (H >> q) * (W >> q);
        const int Hl = H >> l, Wl = W >> l;
        const float inv = 1.0f / (float)(1 << l);
        for (int p0 = 0; p0 < LK_PIX; p0 += LK_G) {
            const int p = p0 + pl;
            if (lane < LK_G * LK_D && p < LK_PIX) {
                const int n = n0 + p;
                const float* plane = pyr + off + ((long)b * N + n) * Hl * Wl;
                const float cx = __fadd_rn(__fmul_rn(coords[(long)b * 2 * N + n], inv), (float)(i - 4));
                const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, cx), (float)(Wl - 1)), 1.0f);
                const float x = __fmul_rn(__fdiv_rn(__fadd_rn(gx, 1.0f), 2.0f), (float)(Wl - 1));
                const float ybase = __fmul_rn(coords[(long)b * 2 * N + N + n], inv);
#pragma unroll
                for (int j = 0; j < LK_D; ++j) {
                    const float cy = __fadd_rn(ybase, (float)(j - 4));
                    const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, cy), (float)(Hl - 1)), 1.0f);
                    const float y = __fmul_rn(__fdiv_rn(__fadd_rn(gy, 1.0f), 2.0f), (float)(Hl - 1));
                    Taps t = make_taps(y, x, Hl, Wl);
                    stage[(l * 81 + i * LK_D + j) * (LK_PIX + 1) + p] = sample_taps(plane, t, Wl);
                }
            }
        }
    }
    __syncthreads();
    const int CH = levels * 81;
    float* ob = out + (long)b * CH * N + n0;
    for (int e = threadIdx.x; e < CH * LK_PIX; e += 256) {
        const int ch = e / LK_PIX, p = e % LK_PIX;
        ob[(long)ch * N + p] = stage[ch * (LK_PIX + 1) + p];
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

// The same, laid out for the stores: thread = (b, y, x, i) owns the EIGHT horizontally adjacent outputs (8y + i, 8x .. 8x + 7) of every channel and
// writes them as two 16-byte stores; neighbouring lanes are neighbouring x, so a half wave writes 1 KB of one output row (the kernel above
// gives neighbouring lanes outputs 32 bytes apart: 2.4 TB/s).  The nine mask reads per (i, j) stay 128-byte row segments.  Same arithmetic in
// the same order as the kernel above (bit-identical results).  C <= 4 (the flow field has 2 channels).
template <int C>
__global__ void __launch_bounds__(256) convex_upsample_rows_kernel(const float* __restrict__ flow, const float* __restrict__ mask,
                                                                  float* __restrict__ out, int B, int h, int w) {
    const int hw = h * w;
    const int xblocks = (w + 31) >> 5;
    int bid = blockIdx.x;
    const int xb = bid % xblocks; bid /= xblocks;
    const int y = bid % h;
    const int b = bid / h;
    const int x = xb * 32 + (threadIdx.x & 31), i = threadIdx.x >> 5;
    if (x >= w) return;
    float f[C][9];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float* fb = flow + ((long)b * C + c) * hw;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            f[c][k] = ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) ? 8.0f * fb[yy * w + xx] : 0.f;
        }
    }
    float o[C][8];
    const int p = y * w + x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* mb = mask + ((long)b * 576 + i * 8 + j) * hw + p;
        float m[9];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) { m[k] = mb[(long)k * 64 * hw]; mx = fmaxf(mx, m[k]); }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); sum += m[k]; }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc += (m[k] / sum) * f[c][k];
            o[c][j] = acc;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float4* op = reinterpret_cast<float4*>(out + (((long)b * C + c) * (8 * h) + 8 * y + i) * (8 * w) + 8 * x);
        op[0] = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
        op[1] = make_float4(o[c][4], o[c][5], o[c][6], o[c][7]);
    }
}

}  // namespace cf

using namespace cf;

extern "C" int cf_corr_volume(const float* cur, const float* prev, float* out, int B, int C, int H, int W, int radius, int stride,
                              void* stream) {
    CF_REQUIRE(cur && prev && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && radius >= 0 && radius <= 8 && stride >= 1, "bad shape");
    hipStream_t s = as_stream(stream);
    if (radius == 4 && corr_mfma_applicable(C, H, W, stride, cur, prev)) return launch_corr_volume_mfma(cur, prev, out, B, C, H, W, stride, s);
    if (radius == 4 && (stride == 1 || stride == 2 || stride == 4) && (W & 3) == 0 && (long)(C + P7_CC) * H * W * 4 < (1L << 31) &&
        ((reinterpret_cast<uintptr_t>(cur) | reinterpret_cast<uintptr_t>(prev)) & 15) == 0) {
        // algorithmic bytes: read cur + prev once, write the 81-channel volume once
        const double bytes = 4.0 * (double)B * H * W * (2.0 * C + 81.0);
        // persistent kernel: one workgroup per CU (256 on MI355X), each walking its XCD's band of 7-row x 64-column tiles
        {
            const int tiles_x = (W + 63) / 64, tiles_y = ((H + stride - 1) / stride + P7_ROWS - 1) / P7_ROWS;
            const long nt = (long)B * stride * tiles_y * tiles_x;
            CF_REQUIRE(nt < (1L << 31), "too many tiles");
            const unsigned nwg = (unsigned)(nt >= 256 ? 256 : ((nt + 7) / 8) * 8);
            dim3 grid(nwg), block(P7_THREADS);
            if (stride == 1) launch_profiled(PK_CORRVOL_S1, bytes, corr_volume_p7_kernel<1>, grid, block, s, cur, prev, out, B, C, H, W, tiles_x, tiles_y, (int)nt);
            else if (stride == 2) launch_profiled(PK_CORRVOL_S2, bytes, corr_volume_p7_kernel<2>, grid, block, s, cur, prev, out, B, C, H, W, tiles_x, tiles_y, (int)nt);
            else launch_profiled(PK_CORRVOL_S4, bytes, corr_volume_p7_kernel<4>, grid, block, s, cur, prev, out, B, C, H, W, tiles_x, tiles_y, (int)nt);
        }
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
    {   // the fused f16-split kernel (level 0 and the pooled levels from one set of accumulators) when the shape allows; CF_ALLPAIRS_FUSED=0
        // keeps the fp32-MFMA GEMM + three pooling launches (A/B knob, and the path of other map widths)
        static int fused = -1;
        if (fused < 0) { const char* e = getenv("CF_ALLPAIRS_FUSED"); fused = e ? atoi(e) : 1; }
        if (fused) {
            const int rc = allpairs_pyramid_fused(f1, f2, pyr, B, C, H, W, levels, s);
            if (rc != 1) return rc;
        }
    }
    // level 0: out[b, n1, n2] = sum_c f1[b,c,n1] * f2[b,c,n2] / sqrt(C): a 1x1 "conv" of f2 with per-sample weights Wt[k=c][m=n1] = f1[b]
    ConvParams p;
    p.x1 = f2; p.x2 = nullptr; p.wt = f1; p.bias = nullptr; p.res = nullptr; p.out = pyr; p.w_bstride = (long)C * N;
    p.C1 = C; p.C2 = 0; p.B = B; p.H = H; p.W = W; p.Cout = N; p.KH = 1; p.KW = 1; p.stride = 1; p.pad_h = 0; p.pad_w = 0;
    p.Ho = H; p.Wo = W; p.out_ctotal = N; p.out_coff = 0; p.act = CF_ACT_NONE; p.alpha = (float)(1.0 / sqrt((double)C));
    p.scatter2x2 = 0; p.gn_ws = nullptr; p.gn_groups = 0;
    // algorithmic bytes: read f1 and f2 once, write the N x N volume once; each pooled level written once
    p.profile_kid = PK_ALLPAIRS; p.profile_work = 4.0 * B * (2.0 * C * N + (double)N * N);
    int rc = launch_conv(p, s);
    if (rc != CF_OK) return rc;
    long off = 0;
    for (int l = 1; l < levels; ++l) {
        int Hl = H >> (l - 1), Wl = W >> (l - 1);
        long planes = (long)B * N;
        const float* in = pyr + off;
        off += planes * Hl * Wl;
        long total = planes * (Hl >> 1) * (Wl >> 1);
        launch_profiled(PK_ALLPAIRS, 4.0 * total, avgpool2x2_kernel, dim3(flat_grid(total, 256)), dim3(256), s, in, pyr + off, planes, Hl, Wl);
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
    static int tiled = -1;
    if (tiled < 0) { const char* e = getenv("CF_LOOKUP_TILED"); tiled = e ? atoi(e) : 1; }      // 0: the flat kernel everywhere (A/B knob)
    if (tiled && radius == 4 && levels <= 4 && (H * W) % 64 == 0) {
        const int pix = tiled == 1 ? 16 : (tiled == 2 ? 32 : 64);       // measured at B = 64: 64 px 208 us (one workgroup per CU), 32 px 94 us, 16 px 76 us
        const size_t lds = sizeof(float) * (size_t)levels * 81 * (pix + 1);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(corr_lookup_tiled_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(corr_lookup_tiled_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(corr_lookup_tiled_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr_set = true;
        }
        hipStream_t st = as_stream(stream);
        const dim3 grid((unsigned)((long)B * (H * W / pix)));
        auto kern = pix == 64 ? corr_lookup_tiled_kernel<64> : (pix == 32 ? corr_lookup_tiled_kernel<32> : corr_lookup_tiled_kernel<16>);
        hipEvent_t e0, e1;
        if (profile_on() && profile_events(PK_CORR_LOOKUP, 4.0 * total + 8.0 * B * H * W, &e0, &e1))
            hipExtLaunchKernelGGL(kern, grid, dim3(256), lds, st, e0, e1, 0, pyr, coords, out, B, H, W, levels);
        else
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, pyr, coords, out, B, H, W, levels);
        CF_CHECK_LAUNCH();
        return CF_OK;
    }
    // algorithmic bytes: the written channels + coords once; the gathers (<= 4 per output) are cache traffic
    launch_profiled(PK_CORR_LOOKUP, 4.0 * total + 8.0 * B * H * W, corr_lookup_kernel, dim3(flat_grid(total, 256)), dim3(256), as_stream(stream), pyr,
                    coords, out, B, H, W, levels, radius);
    CF_CHECK_LAUNCH();
    return CF_OK;
}

extern "C" int cf_convex_upsample(const float* flow, const float* mask, float* out, int B, int C, int h, int w, void* stream) {
    CF_REQUIRE(flow && mask && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0, "bad shape");
    long total = (long)B * 64 * h * w;
    // algorithmic bytes: read the 576-channel mask and the coarse field once, write the 8x upsampled field once
    const double bytes = 4.0 * B * h * w * (576.0 + C + 64.0 * C);
    const long rows_wgs = (long)B * h * ((w + 31) / 32);
    if ((C == 2 || C == 1) && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && rows_wgs < (1L << 31)) {
        if (C == 2) launch_profiled(PK_CONVEX_UP, bytes, convex_upsample_rows_kernel<2>, dim3((unsigned)rows_wgs), dim3(256), as_stream(stream), flow, mask, out, B, h, w);
        else launch_profiled(PK_CONVEX_UP, bytes, convex_upsample_rows_kernel<1>, dim3((unsigned)rows_wgs), dim3(256), as_stream(stream), flow, mask, out, B, h, w);
    } else {
        launch_profiled(PK_CONVEX_UP, bytes, convex_upsample_kernel, dim3(flat_grid(total, 256)), dim3(256), as_stream(stream), flow, mask, out, B, C, h, w);
    }
    CF_CHECK_LAUNCH();
    return CF_OK;
}
