// Shared host/device helpers for libcineflow_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/cineflow.h"

#ifndef CF_GELU_EXACT_ERFF
#define CF_GELU_EXACT_ERFF 0      // build with -DCF_GELU_EXACT_ERFF=1 for the library erff in every GELU
#endif

namespace cf {

void set_error(const std::string& s);

#define CF_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            char _b[512];                                       \
            snprintf(_b, sizeof(_b), __VA_ARGS__);              \
            cf::set_error(std::string(__func__) + ": " + _b);   \
            return CF_ERR_ARG;                                  \
        }                                                       \
    } while (0)

#define CF_CHECK_LAUNCH()                                                                  \
    do {                                                                                   \
        hipError_t _e = hipGetLastError();                                                 \
        if (_e != hipSuccess) {                                                            \
            cf::set_error(std::string(__func__) + ": launch failed: " + hipGetErrorString(_e)); \
            return CF_ERR_LAUNCH;                                                          \
        }                                                                                  \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// grid size for a flat memory-bound kernel: enough blocks to fill 256 CUs, grid-stride the rest
static inline int flat_grid(long n, int block, int per_thread = 1) {
    long blocks = (n + (long)block * per_thread - 1) / ((long)block * per_thread);
    if (blocks < 1) blocks = 1;
    if (blocks > 256L * 16) blocks = 256L * 16;
    return (int)blocks;
}

// GELU = 0.5 x (1 + erf(x / sqrt 2)) with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute -- fp32 rounding level on the
// product): one rcp, one exp2 and eight FMAs, no branch; the library erff is ~3x the instructions with a divergent range split,
// which made the GELU apply pass VALU-bound next to its 8 bytes per element.
__device__ __forceinline__ float gelu_as(float v) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    return 0.5f * v * (1.0f + copysignf(fmaf(-poly, e, 1.0f), v));
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case CF_ACT_GELU: return CF_GELU_EXACT_ERFF ? 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)) : gelu_as(v);
        case CF_ACT_RELU: return v > 0.f ? v : 0.f;
        case CF_ACT_LRELU: return v > 0.f ? v : 0.01f * v;
        case CF_ACT_TANH: return tanhf(v);
        case CF_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// Sampling coordinate of SpatialTransformer.forward (integration.py:61-79) followed by
// grid_sample's align_corners=True un-normalisation, with the reference's fp32 rounding path:
//   g = 2 * ((i + f) / (S - 1) - 0.5);  pos = ((g + 1) / 2) * (S - 1)
// Contraction into FMAs is switched off (hipcc's default is -ffp-contract=fast and __fadd_rn / __fmul_rn are plain operators in its
// headers) so that every intermediate rounds as in PyTorch.
__device__ __forceinline__ float st_coord(float idx, float f, float size_m1) {
#pragma clang fp contract(off)
    float loc = __fadd_rn(idx, f);
    float g = __fmul_rn(2.0f, __fsub_rn(__fdiv_rn(loc, size_m1), 0.5f));
    return __fmul_rn(__fdiv_rn(__fadd_rn(g, 1.0f), 2.0f), size_m1);
}

// Bilinear tap set with zero padding (grid_sample padding_mode='zeros').
struct Taps {
    int y0, x0;
    float w00, w01, w10, w11;  // (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1)
    bool v00, v01, v10, v11;
};

__device__ __forceinline__ Taps make_taps(float y, float x, int H, int W) {
#pragma clang fp contract(off)
    Taps t;
    float yf = floorf(y), xf = floorf(x);
    float wy = __fsub_rn(y, yf), wx = __fsub_rn(x, xf);
    float ey = __fsub_rn(1.0f, wy), ex = __fsub_rn(1.0f, wx);
    // guard the float->int conversion against huge / NaN coordinates
    yf = fminf(fmaxf(yf, -2.0f), (float)H + 1.0f);
    xf = fminf(fmaxf(xf, -2.0f), (float)W + 1.0f);
    t.y0 = (int)yf;
    t.x0 = (int)xf;
    t.w00 = __fmul_rn(ex, ey);
    t.w01 = __fmul_rn(wx, ey);
    t.w10 = __fmul_rn(ex, wy);
    t.w11 = __fmul_rn(wx, wy);
    bool y0v = t.y0 >= 0 && t.y0 < H, y1v = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    bool x0v = t.x0 >= 0 && t.x0 < W, x1v = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    bool fin = (y == y) && (x == x) && fabsf(y) < 1e9f && fabsf(x) < 1e9f;
    t.v00 = fin && y0v && x0v;
    t.v01 = fin && y0v && x1v;
    t.v10 = fin && y1v && x0v;
    t.v11 = fin && y1v && x1v;
    return t;
}

__device__ __forceinline__ float sample_taps(const float* __restrict__ plane, const Taps& t, int W) {
#pragma clang fp contract(off)
    const float* p = plane + (long)t.y0 * W + t.x0;
    float a = t.v00 ? p[0] : 0.f;
    float b = t.v01 ? p[1] : 0.f;
    float c = t.v10 ? p[W] : 0.f;
    float d = t.v11 ? p[W + 1] : 0.f;
    // same association as ATen's CPU grid_sampler: nw*w + ne*w + sw*w + se*w
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a, t.w00), __fmul_rn(b, t.w01)), __fmul_rn(c, t.w10)),
                     __fmul_rn(d, t.w11));
}

// Per-device bookkeeping for the launchers (a process may drive several GPUs): the current device's index (0 when out of the table's range)
// and its compute-unit count rounded down to whole groups of 8 XCDs (256 on an MI355X in SPX mode; 256 when the query fails).
static inline int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return dev;
}
static inline int device_cu_count() {
    static int n[64] = {};
    const int dev = current_device_slot();
    if (!n[dev]) {
        hipDeviceProp_t prop;
        n[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8) ? (prop.multiProcessorCount & ~7) : 256;
    }
    return n[dev];
}

// Diagnostic build only (make clock -> libcineflow_hip_clock.so, never loaded by the package): every workgroup of the stamped kernels adds
// its shader-clock and real-time-clock spans to a per-file counter, so that tools/power_probe.py can report the clock the chip HOLDS inside the
// kernel (MI355X_MICROARCH.md, DVFS give-back item 6: delta s_memtime / delta s_memrealtime x 100 MHz).  The stamps go to a buffer no kernel reads.
#ifdef CF_CLOCK_STAMPS
#define CF_CLOCK_DECL(name) __device__ unsigned long long name[3];
#define CF_CLOCK_BEGIN() const unsigned long long ck_t0_ = __builtin_amdgcn_s_memtime(), ck_r0_ = __builtin_amdgcn_s_memrealtime()
#define CF_CLOCK_END(name)                                                                                      \
    do {                                                                                                        \
        if (threadIdx.x == 0) {                                                                                 \
            atomicAdd(&name[0], (unsigned long long)(__builtin_amdgcn_s_memtime() - ck_t0_));                   \
            atomicAdd(&name[1], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - ck_r0_));               \
            atomicAdd(&name[2], 1ULL);                                                                          \
        }                                                                                                       \
    } while (0)
#define CF_CLOCK_READER(fn, name)                                                                               \
    extern "C" int fn(double* ghz, long* workgroups) {                                                          \
        unsigned long long v[3] = {0, 0, 0}, z[3] = {0, 0, 0};                                                  \
        if (hipDeviceSynchronize() != hipSuccess) return CF_ERR_LAUNCH;                                         \
        if (hipMemcpyFromSymbol(v, HIP_SYMBOL(name), sizeof(v)) != hipSuccess) return CF_ERR_LAUNCH;            \
        if (hipMemcpyToSymbol(HIP_SYMBOL(name), z, sizeof(z)) != hipSuccess) return CF_ERR_LAUNCH;              \
        *ghz = v[1] ? 0.1 * (double)v[0] / (double)v[1] : 0.0;                                                  \
        *workgroups = (long)v[2];                                                                               \
        return CF_OK;                                                                                           \
    }
#else
#define CF_CLOCK_DECL(name)
#define CF_CLOCK_BEGIN() do { } while (0)
#define CF_CLOCK_END(name) do { } while (0)
#define CF_CLOCK_READER(fn, name)
#endif

// ---- cross-lane sums without LDS traffic (gfx950).  __shfl_xor is a ds_bpermute: an LDS-queue instruction with ~100 clocks of latency per level;
// partners inside a row of 16 lanes are reachable by DPP modifiers instead, and rows / wave halves are exchanged by v_permlane16_swap /
// v_permlane32_swap (new on gfx950): a' = [a.row0, b.row0, a.row2, b.row2], b' = [a.row1, b.row1, a.row3, b.row3].
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;       // quad_perm [1,0,3,2] / [2,3,0,1]
constexpr int DPP_ROR8 = 0x128;                       // row_ror:8 = lane ^ 8 inside the row
constexpr int DPP_HALF_MIRROR = 0x141;                // lane ^ 7 inside each 8 lanes
// Inline assembly: this hipcc's __builtin_amdgcn_permlane16_swap / 32_swap lose their second result (both members of the returned pair are the
// new vdst: tools/ubench/xreduce_check.hip).  The s_nop pairs stand in for the hazard recogniser, which does not see inside an asm statement
// (a VALU write of an operand needs wait states before the swap reads it, and so does a VALU read after it: four each way here, twice what
// the compiler inserts around its own builtin).
__device__ __forceinline__ void swap_rows16(float& a, float& b) {
    asm("s_nop 3\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap_halves32(float& a, float& b) {
    asm("s_nop 3\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
}
// Transpose-reduce of sixteen per-lane partial sums over the 32 lanes of a wave half: afterwards v[0] of the lane with bits (b4 b3 b2 b1 x)
// holds register 8 b4 + 4 b3 + 2 b2 + b1 summed over those 32 lanes (both lanes of a pair hold the same).  The level that pairs on lane bit 2
// uses the half mirror (partner lane ^ 7): the partner differs in bits 0 and 1 as well, which the two remaining levels sum over anyway.
__device__ __forceinline__ void xreduce16(float (&v)[16], int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = v[i], b = v[i + 8];
        swap_rows16(a, b);                                    // rows with bit 4 clear: own v[i] + the partner row's; set: own v[i + 8] + the partner row's
        v[i] = a + b;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool up = (lane & 8) != 0;
        const float keep = up ? v[i + 4] : v[i], send = up ? v[i] : v[i + 4];
        v[i] = keep + dpp_f32<DPP_ROR8>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool up = (lane & 4) != 0;
        const float keep = up ? v[i + 2] : v[i], send = up ? v[i] : v[i + 2];
        v[i] = keep + dpp_f32<DPP_HALF_MIRROR>(send);
    }
    {
        const bool up = (lane & 2) != 0;
        const float keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
        v[0] = keep + dpp_f32<DPP_XOR2>(send);
    }
    v[0] += dpp_f32<DPP_XOR1>(v[0]);
}
// s + s of lane ^ MASK for MASK in {2, 4, 8, 16, 32}.  MASK = 4 goes through the half mirror (lane ^ 7) and is only right when s is already the
// same on the four lanes of a quad (i.e. after the MASK = 2 level, as after xreduce16's pair sum).
template <int MASK>
__device__ __forceinline__ float xor_sum(float s) {
    static_assert(MASK == 2 || MASK == 4 || MASK == 8 || MASK == 16 || MASK == 32, "lane mask");
    if constexpr (MASK == 2) return s + dpp_f32<DPP_XOR2>(s);
    else if constexpr (MASK == 4) return s + dpp_f32<DPP_HALF_MIRROR>(s);
    else if constexpr (MASK == 8) return s + dpp_f32<DPP_ROR8>(s);
    else if constexpr (MASK == 16) { float a = s, b = s; swap_rows16(a, b); return a + b; }
    else { float a = s, b = s; swap_halves32(a, b); return a + b; }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace cf
