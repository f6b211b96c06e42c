// Checks common.h's DPP / permlane reductions against the __shfl_xor forms, level by level (one wave, integers in floats: exact).
// Build: hipcc -O3 --offload-arch=gfx950 -I../../cardiac-segmentation-optical-flow_amd/csrc xreduce_check.hip -o xreduce_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "common.h"
using namespace cf;
__global__ void k(const float* in, float* out) {
    const int lane = threadIdx.x;
    float v[16], w[16];
    for (int i = 0; i < 16; ++i) v[i] = w[i] = in[lane * 16 + i];
    xreduce16(v, lane);
    // reference
    for (int i = 0; i < 8; ++i) { const bool up = lane & 16; const float keep = up ? w[i + 8] : w[i], send = up ? w[i] : w[i + 8]; w[i] = keep + __shfl_xor(send, 16, 64); }
    for (int i = 0; i < 4; ++i) { const bool up = lane & 8; const float keep = up ? w[i + 4] : w[i], send = up ? w[i] : w[i + 4]; w[i] = keep + __shfl_xor(send, 8, 64); }
    for (int i = 0; i < 2; ++i) { const bool up = lane & 4; const float keep = up ? w[i + 2] : w[i], send = up ? w[i] : w[i + 2]; w[i] = keep + __shfl_xor(send, 4, 64); }
    { const bool up = lane & 2; const float keep = up ? w[1] : w[0], send = up ? w[0] : w[1]; w[0] = keep + __shfl_xor(send, 2, 64); }
    w[0] += __shfl_xor(w[0], 1, 64);
    out[lane] = v[0];
    out[64 + lane] = w[0];
    const float s = w[0];
    float a = xor_sum<2>(s); out[128 + lane] = a; out[192 + lane] = s + __shfl_xor(s, 2, 64);
    float a4 = xor_sum<4>(a); out[256 + lane] = a4; out[320 + lane] = a + __shfl_xor(a, 4, 64);
    out[384 + lane] = xor_sum<8>(s); out[448 + lane] = s + __shfl_xor(s, 8, 64);
    out[512 + lane] = xor_sum<16>(s); out[576 + lane] = s + __shfl_xor(s, 16, 64);
    out[640 + lane] = xor_sum<32>(s); out[704 + lane] = s + __shfl_xor(s, 32, 64);
    // raw primitives
    float p = (float)lane, q = (float)(100 + lane);
    swap_rows16(p, q); out[768 + lane] = p; out[832 + lane] = q;
    out[896 + lane] = dpp_f32<DPP_ROR8>((float)lane);
    out[960 + lane] = dpp_f32<DPP_HALF_MIRROR>((float)lane);
}
int main() {
    float h[1024], *di, *dout, ho[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37 + 11) % 101);
    hipMalloc(&di, 4096); hipMalloc(&dout, 4096);
    hipMemcpy(di, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout);
    hipMemcpy(ho, dout, 4096, hipMemcpyDeviceToHost);
    const char* names[] = {"xreduce16", "xor2", "xor4", "xor8", "xor16", "xor32"};
    for (int t = 0; t < 6; ++t) {
        int bad = 0;
        for (int l = 0; l < 64; ++l) bad += ho[t * 128 + l] != ho[t * 128 + 64 + l];
        printf("%s: %d lanes differ\n", names[t], bad);
        if (bad) { for (int l = 0; l < 64; ++l) printf("%g/%g ", ho[t * 128 + l], ho[t * 128 + 64 + l]); printf("\n"); }
    }
    printf("swap_rows16 a': "); for (int l = 0; l < 64; ++l) printf("%g ", ho[768 + l]); printf("\n");
    printf("swap_rows16 b': "); for (int l = 0; l < 64; ++l) printf("%g ", ho[832 + l]); printf("\n");
    printf("ror8: "); for (int l = 0; l < 64; ++l) printf("%g ", ho[896 + l]); printf("\n");
    printf("half mirror: "); for (int l = 0; l < 64; ++l) printf("%g ", ho[960 + l]); printf("\n");
    return 0;
}
