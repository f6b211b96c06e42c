// Micro-benchmark: how much of the sustained f16 MFMA rate do operand fetches cost?  The MFMA loop of mfma_rate.hip (random operands,
// 12 MFMAs per iteration into 4 accumulators) with, per iteration, NL ds_read_b128 from a 32 KB LDS image and NG global_load_dwordx4 from
// an L2-resident 1 MB buffer feeding the operands (so that the loads cannot be dropped).  The convolution kernel's four-wave shape has
// NL = 8, NG = 2 per 12 MFMAs.  Reports TFLOP/s (dense f16) and the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NL, int NG>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* ticks, int iters, const f16x8* __restrict__ gbuf) {
    __shared__ f16x8 lds[2048];   // 32 KB
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = gbuf[i];
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a0 = gbuf[lane], a1 = gbuf[64 + lane];
    f16x8 b[8];
    for (int i = 0; i < 8; ++i) b[i] = gbuf[128 + i * 64 + lane];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int base = (it * 131 + wave * 257) & 1023;
        if (NG >= 1) a0 = gbuf[((base * 64 + blockIdx.x * 64) & 65535) + lane];
        if (NG >= 2) a1 = gbuf[((base * 64 + 4096 + blockIdx.x * 64) & 65535) + lane];
#pragma unroll
        for (int i = 0; i < NL; ++i) b[i] = lds[((base + i * 64) & 1023) + lane + (i & 1) * 1024 - ((i & 1) ? 64 : 0)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[2 * t], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[2 * t + 1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[2 * t], acc[t], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NL, int NG>
static void run(float* d, unsigned long long* dt, const f16x8* g, hipEvent_t e0, hipEvent_t e1) {
    for (int wps = 1; wps <= 3; ++wps) {
        const int blocks = 256 * wps, iters = 60000 / wps;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<NL, NG>), dim3(blocks), dim3(256), 0, 0, d, dt, iters, g);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long ticks = 0;
        hipMemcpy(&ticks, dt, 8, hipMemcpyDeviceToHost);
        const double mfma = (double)blocks * 4 * iters * 12;
        const double tf = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) * 1e-12;
        printf("ds_read_b128 %d + global dwordx4 %d per 12 MFMAs, waves/SIMD %d: %7.2f ms  %7.1f TFLOP/s  clock %.3f GHz  %.1f clocks per MFMA per SIMD\n", NL, NG, wps,
               ms, tf, (double)ticks / (ms * 1e6), (double)ticks / ((double)iters * 12 * wps));
        fflush(stdout);
    }
}

int main() {
    float* d;
    unsigned long long* dt;
    f16x8* g;
    hipMalloc(&d, 256 * 3 * 256 * 4);
    hipMalloc(&dt, 8);
    hipMalloc(&g, 65536 * 64 * sizeof(f16x8) / 64 + 4096 * sizeof(f16x8));
    const size_t n = 65536 + 4096;
    _Float16* h = (_Float16*)malloc(n * 16);
    srand(1);
    for (size_t i = 0; i < n * 8; ++i) h[i] = (_Float16)(((rand() % 2001) - 1000) * 1e-3f);
    hipMemcpy(g, h, n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    run<0, 0>(d, dt, g, e0, e1);
    run<4, 0>(d, dt, g, e0, e1);
    run<8, 0>(d, dt, g, e0, e1);
    run<0, 2>(d, dt, g, e0, e1);
    run<4, 2>(d, dt, g, e0, e1);
    run<8, 2>(d, dt, g, e0, e1);
    return 0;
}
