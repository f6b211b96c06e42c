// Multi-head attention core on channel-first tokens, fp32 MFMA, flash-style (no N x N score matrix in HBM).
//
// Orientation is chosen so that nothing crosses lanes between the two products (cdna_hip_programming.md section 3,
// "An accumulator tile as the next MFMA's operand", restated for the f32 32x32x2 form):
//   S^T[key][q]  = sum_dd K[key][dd] * Q[q][dd]        A = K (row = key on the lane), B = Q^T (col = q on the lane)
//   O^T[dd][q]  += sum_key V[key][dd] * P[q][key]      A = V^T (row = dd), B = P^T : lane (q, half h) already holds
//                                                       P[q][key_h(r)] in accumulator register r, key_h(r) = (r&3)+8(r>>2)+4h
// so the softmax statistics (running max / sum) are per lane (one query per lane, its 32 keys split over the two
// half-waves: one __shfl_xor(.,32) per reduction), and O^T is rescaled by a per-lane scalar.
// A workgroup = 4 waves = 128 queries of one (sample, head); K/V stream through LDS in blocks of 32 keys.
#include "common.h"

namespace cf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int D>
__global__ void __launch_bounds__(256) attention_cf_kernel(const float* __restrict__ q, long q_bs, const float* __restrict__ k,
                                                           long k_bs, const float* __restrict__ v, long v_bs,
                                                           float* __restrict__ out, int heads, int Nq, int Nk, float scale) {
    constexpr int DT = (D + 31) / 32;
    constexpr int VP = 33;  // V row pitch (floats): lanes vary dd at fixed key -> conflict-free
    __shared__ float k_lds[D * 32];
    __shared__ float v_lds[D * VP];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int qblocks = (Nq + 127) / 128;
    const int qb = blockIdx.x % qblocks;
    const int bh = blockIdx.x / qblocks;
    const int h = bh % heads, b = bh / heads;
    const int q0 = qb * 128 + wave * 32;
    const bool active = q0 < Nq;  // wave-uniform (Nq % 32 == 0)

    const float* qp = q + (long)b * q_bs + (long)h * D * Nq;
    const float* kp = k + (long)b * k_bs + (long)h * D * Nk;
    const float* vp = v + (long)b * v_bs + (long)h * D * Nk;

    // Q^T fragments, pre-scaled (nn.MultiheadAttention scales q before q k^T)
    float qf[D / 2];
#pragma unroll
    for (int s = 0; s < D / 2; ++s) qf[s] = active ? qp[(long)(2 * s + half) * Nq + q0 + l31] * scale : 0.f;

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;  // l_run: this half-wave's partial sum

    for (int kb = 0; kb < Nk; kb += 32) {
        __syncthreads();  // previous block fully consumed
        for (int e = threadIdx.x; e < D * 32; e += 256) {
            int dd = e >> 5, key = e & 31;
            k_lds[dd * 32 + key] = kp[(long)dd * Nk + kb + key];
            v_lds[dd * VP + key] = vp[(long)dd * Nk + kb + key];
        }
        __syncthreads();
        if (!active) continue;

        f32x16 sT;
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[r] = 0.f;
#pragma unroll
        for (int s = 0; s < D / 2; ++s)
            sT = __builtin_amdgcn_mfma_f32_32x32x2f32(k_lds[(2 * s + half) * 32 + l31], qf[s], sT, 0, 0, 0);

        float mx = sT[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sT[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);  // 0 on the first block (m_run = -inf)
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sT[r] = expf(sT[r] - m_new);
            psum += sT[r];
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            const int dd = dt * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
                float a = (dd < D) ? v_lds[dd * VP + key] : 0.f;
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sT[r], o[dt], 0, 0, 0);
            }
        }
    }
    if (!active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    float* op = out + ((long)b * heads + h) * D * Nq;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dd = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (dd < D) op[(long)dd * Nq + q0 + l31] = o[dt][r] * inv;
        }
}

}  // namespace cf

using namespace cf;

extern "C" int cf_attention_cf(const float* q, long q_bs, const float* k, long k_bs, const float* v, long v_bs, float* out, int B,
                               int heads, int d, int Nq, int Nk, void* stream) {
    CF_REQUIRE(q && k && v && out, "null pointer");
    CF_REQUIRE(B > 0 && heads > 0 && Nq > 0 && Nk > 0, "bad shape");
    CF_REQUIRE(Nq % 32 == 0 && Nk % 32 == 0, "token counts must be multiples of 32 (Nq=%d Nk=%d)", Nq, Nk);
    CF_REQUIRE(d == 8 || d == 16 || d == 32 || d == 64, "head dim %d unsupported (8,16,32,64)", d);
    CF_REQUIRE(q_bs >= (long)heads * d * Nq && k_bs >= (long)heads * d * Nk && v_bs >= (long)heads * d * Nk, "batch stride too small");
    const float scale = (float)(1.0 / sqrt((double)d));
    dim3 grid((unsigned)(B * heads * ((Nq + 127) / 128)));
    hipStream_t s = as_stream(stream);
    switch (d) {
        case 8: hipLaunchKernelGGL((attention_cf_kernel<8>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale); break;
        case 16: hipLaunchKernelGGL((attention_cf_kernel<16>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale); break;
        case 32: hipLaunchKernelGGL((attention_cf_kernel<32>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale); break;
        default: hipLaunchKernelGGL((attention_cf_kernel<64>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale); break;
    }
    CF_CHECK_LAUNCH();
    return CF_OK;
}
