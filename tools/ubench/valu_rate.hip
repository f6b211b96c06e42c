// Micro-benchmark: issue rate of v_pk_fma_f32 vs v_fma_f32 on gfx950 (1, 2, 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0) {
    f2 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f2{(float)i, (float)threadIdx.x};
    f2 a = f2{a0, a0 * 0.5f}, b = f2{1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (PK) acc[i] = __builtin_elementwise_fma(a, acc[i], b);
            else { acc[i].x = fmaf(a.x, acc[i].x, b.x); }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 64 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) for (int pk = 0; pk < 2; ++pk) {
        int blocks = 256 * wps, iters = 20000;   // 256 threads = 4 waves = one per SIMD
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (pk) k<1><<<blocks, 256>>>(d, iters, 0.999f); else k<0><<<blocks, 256>>>(d, iters, 0.999f);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst = (double)blocks * 4 * iters * 16;              // wave-instructions
        double cyc = ms * 1e-3 * 2.4e9;                              // at 2.4 GHz
        printf("waves/SIMD %d  %s: %.3f ms  -> %.2f cycles per wave-instruction per SIMD, %.1f TFLOP/s\n", wps, pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms,
               cyc / (inst / (256.0 * 4)), inst * 64 * (pk ? 4 : 2) / (ms * 1e-3) * 1e-12);
    }
    return 0;
}
