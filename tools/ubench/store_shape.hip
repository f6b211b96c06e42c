// Micro-benchmark: the convolution epilogue's store pattern.  Every wave writes a 32-channel x 128-pixel fp32 block (16 KB) of an
// [B][C][H*W] tensor: (a) as the kernel does today -- 64 dword stores, lanes 0-31 one 128-byte row segment of channel c, lanes 32-63 the
// same segment of channel c + 4 --, (b) 16 dwordx4 stores (each lane four consecutive pixels: 512 contiguous bytes per channel and half
// wave).  Same bytes, same addresses overall; 12 waves per CU, all CUs, ~268 MB per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VEC>
__global__ void __launch_bounds__(256) k(float* __restrict__ out, int HW, int C, int tiles_per_img, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.x % tiles_per_img, b = blockIdx.x / tiles_per_img;     // 128-pixel tile
    float* base = out + ((long)b * C + wave * 32) * HW + tile * 128;                  // this wave's 32 channels
    float v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = seed * (float)(i + lane);
    if (VEC == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) base[(long)co * HW + nt * 32 + l31] = v[nt * 16 + r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
            *reinterpret_cast<f32x4*>(base + (long)co * HW + l31 * 4) = f32x4{v[r], v[16 + r], v[32 + r], v[48 + r]};
        }
    }
}

int main() {
    const int B = 32, C = 128, HW = 128 * 128, tiles = HW / 128;
    float* d;
    hipMalloc(&d, (size_t)B * C * HW * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int vec = 1; vec <= 4; vec += 3) {
            float ms = 0;
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0);
                if (vec == 1) hipLaunchKernelGGL(k<1>, dim3(B * tiles), dim3(256), 0, 0, d, HW, C, tiles, 1.0f + it);
                else hipLaunchKernelGGL(k<4>, dim3(B * tiles), dim3(256), 0, 0, d, HW, C, tiles, 1.0f + it);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%s: %.1f us  %.0f GB/s\n", vec == 1 ? "64 x dword stores / wave " : "16 x dwordx4 stores / wave", ms * 1e3, (double)B * C * HW * 4 / (ms * 1e-3) * 1e-9);
        }
    return 0;
}
