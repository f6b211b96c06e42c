// Micro-benchmark: what v_mfma_f32_32x32x16_f16 rate does this MI355X sustain, and at what shader clock?  Pure register-operand MFMA loops
// (4 independent accumulators per wave) at 1, 2, 3 waves per SIMD, with all-zero, small-magnitude and random fp16 operands (switching activity
// is data dependent), for a short launch (~1 ms) and a long one (~100 ms: past the power-management time constant).  The clock is derived
// from s_memtime ticks per wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* ticks, int iters, const _Float16* opnd) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lane = threadIdx.x & 63;
    f16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = opnd[(lane * 16 + i * 8 + j) & 4095]; b[i][j] = opnd[(lane * 16 + i * 8 + j + 1024) & 4095]; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc[t], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
    float* d;
    unsigned long long* dt;
    _Float16 *dz, *ds, *dr;
    hipMalloc(&d, 256 * 3 * 256 * 4);
    hipMalloc(&dt, 8);
    _Float16 h[4096];
    hipMalloc(&dz, sizeof(h)); hipMalloc(&ds, sizeof(h)); hipMalloc(&dr, sizeof(h));
    for (int i = 0; i < 4096; ++i) h[i] = (_Float16)0.f;
    hipMemcpy(dz, h, sizeof(h), hipMemcpyHostToDevice);
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (_Float16)(((rand() % 2001) - 1000) * 1e-6f);   // |x| <= 1e-3: the 'lo' halves of a split look like this
    hipMemcpy(ds, h, sizeof(h), hipMemcpyHostToDevice);
    for (int i = 0; i < 4096; ++i) h[i] = (_Float16)(((rand() % 2001) - 1000) * 1e-3f);   // |x| <= 1
    hipMemcpy(dr, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"zeros ", "small ", "random"};
    const _Float16* ops[3] = {dz, ds, dr};
    for (int wps = 1; wps <= 3; ++wps)
        for (int o = 0; o < 3; ++o)
            for (int lng = 0; lng < 2; ++lng) {
                const int blocks = 256 * wps, iters = lng ? 200000 / wps : 4000;
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, dt, iters, ops[o]);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                unsigned long long ticks = 0;
                hipMemcpy(&ticks, dt, 8, hipMemcpyDeviceToHost);
                const double mfma = (double)blocks * 4 * iters * 12;
                const double tf = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) * 1e-12;
                printf("waves/SIMD %d  %s  %7.2f ms : %7.1f TFLOP/s f16 dense (%4.1f %% of 2500), s_memtime %.3f ticks/ns, %.1f ticks per MFMA per SIMD\n", wps,
                       names[o], ms, tf, tf / 25.0, (double)ticks / (ms * 1e6), (double)ticks / ((double)iters * 12 * wps));
                fflush(stdout);
            }
    return 0;
}
