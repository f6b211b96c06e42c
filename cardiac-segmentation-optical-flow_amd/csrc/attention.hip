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
                                                           float* __restrict__ out, int heads, int Nq, int Nk, float scale, int nk_valid) {
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
        if (kb + 32 > nk_valid) {        // padded keys of a ragged sequence (host pads N to a multiple of 32): out of the softmax
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kb + (r & 3) + 8 * (r >> 2) + 4 * half >= nk_valid) sT[r] = -INFINITY;
        }

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


// =====================================================================================================================
// f16 hi/lo-split variant (head dims 16, 32, 64): the same orientation and online softmax, but both products run on
// v_mfma_f32_32x32x16_f16 with the 3-term split of conv_f16s.hip (x = hi + lo in fp16, a.b ~ al.bh + ah.bl + ah.bh, fp32
// accumulation): 24 f16 MFMAs of 32 cycles per 32-key block instead of 64 fp32 MFMAs of 64 cycles.  The accumulator-as-operand
// trick survives: for k-step s the f16 B operand of lane (q, half h) wants 8 consecutive k-indices, and the contraction order
// over the keys is free, so k-index j of step s is DEFINED as the key held by accumulator register 8s + j, i.e. key
// (j & 3) + 8 (j >> 2) + 16 s + 4 h; the V^T fragment of that step is then two 8-byte runs of 4 keys.  K and V tiles are
// converted once per workgroup while staging (K as [key][hi D | lo D], V transposed as [dd][hi 32 keys | lo 32 keys], record
// sizes odd multiples of 16 B: conflict-free b128/b64 reads), double buffered with the next block's loads in flight.
// =====================================================================================================================
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_h(float x, _Float16& hi, _Float16& lo) {
    x = __builtin_amdgcn_fmed3f(x, -60000.f, 60000.f);
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

template <int D>
__global__ void __launch_bounds__(256, 2) attention_cf_f16s_kernel(const float* __restrict__ q, long q_bs, const float* __restrict__ k,
                                                                  long k_bs, const float* __restrict__ v, long v_bs,
                                                                  float* __restrict__ out, int heads, int Nq, int Nk, float scale, int nk_valid) {
    constexpr int KS = D / 16;             // k-steps of the score product
    constexpr int DT = (D + 31) / 32;      // 32-row tiles of O^T
    constexpr int KREC = 4 * D + 16;       // bytes per key record  [hi D halves | lo D halves | pad]
    constexpr int VREC = 128 + 16;         // bytes per dd record   [hi 32 keys | lo 32 keys | pad]
    constexpr int KBUF = 32 * KREC, VBUF = DT * 32 * VREC;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (KBUF + VBUF)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // Q^T fragments (pre-scaled), hi / lo: lane (q, half) holds dd = 16 s + 8 half + j
    h8 qh[KS], ql[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = active ? qp[(long)(16 * s + 8 * half + j) * Nq + q0 + l31] * scale : 0.f;
            _Float16 a, c;
            split_h(x, a, c);
            qh[s][j] = a;
            ql[s][j] = c;
        }

    // staging tasks: K -- thread = (key, 8-dd group): 8 coalesced dword loads, one 16-B hi and one 16-B lo write;
    //                V -- thread = (dd, 8-key group): two dwordx4 loads, one 16-B hi and one 16-B lo write
    const int k_key = tid & 31, k_g = tid >> 5;             // groups 0..7 (D = 64 uses all)
    const bool k_task = k_g < D / 8;
    const int v_dd = tid >> 2, v_g = tid & 3;
    const bool v_task = v_dd < D;
    float kst[8];
    f4 vst[2];
    auto issue = [&](int kb) {
        if (k_task)
#pragma unroll
            for (int j = 0; j < 8; ++j) kst[j] = kp[(long)(8 * k_g + j) * Nk + kb + k_key];
        if (v_task) {
            const f4* src = reinterpret_cast<const f4*>(vp + (long)v_dd * Nk + kb + 8 * v_g);
            vst[0] = src[0];
            vst[1] = src[1];
        }
    };
    auto stage = [&](int buf) {
        unsigned char* kb_ = lds + buf * (KBUF + VBUF);
        unsigned char* vb_ = kb_ + KBUF;
        if (k_task) {
            h8 a, c;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 x, y;
                split_h(kst[j], x, y);
                a[j] = x;
                c[j] = y;
            }
            *reinterpret_cast<h8*>(kb_ + k_key * KREC + k_g * 16) = a;
            *reinterpret_cast<h8*>(kb_ + k_key * KREC + 2 * D + k_g * 16) = c;
        }
        if (v_task) {
            h8 a, c;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 x, y;
                split_h(vst[j >> 2][j & 3], x, y);
                a[j] = x;
                c[j] = y;
            }
            *reinterpret_cast<h8*>(vb_ + v_dd * VREC + v_g * 16) = a;
            *reinterpret_cast<h8*>(vb_ + v_dd * VREC + 64 + v_g * 16) = c;
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;  // l_run: this half-wave's partial sum

    issue(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int kb = 0; kb < Nk; kb += 32, buf ^= 1) {
        const bool more = kb + 32 < Nk;
        if (more) issue(kb + 32);
        if (active) {
            const unsigned char* kb_ = lds + buf * (KBUF + VBUF);
            const unsigned char* vb_ = kb_ + KBUF;
            f32x16 sT;
#pragma unroll
            for (int r = 0; r < 16; ++r) sT[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const h8 kh = *reinterpret_cast<const h8*>(kb_ + l31 * KREC + (16 * s + 8 * half) * 2);
                const h8 kl = *reinterpret_cast<const h8*>(kb_ + l31 * KREC + 2 * D + (16 * s + 8 * half) * 2);
                sT = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sT, 0, 0, 0);
                sT = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sT, 0, 0, 0);
                sT = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], sT, 0, 0, 0);
            }
            if (kb + 32 > nk_valid) {    // padded keys of a ragged sequence: out of the softmax
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb + (r & 3) + 8 * (r >> 2) + 4 * half >= nk_valid) sT[r] = -INFINITY;
            }
            float mx = sT[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sT[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = expf(m_run - m_new);  // 0 on the first block (m_run = -inf)
            m_run = m_new;
            float psum = 0.f;
            h8 ph[2], pl[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = expf(sT[r] - m_new);
                psum += pv;
                _Float16 a, c;
                split_h(pv, a, c);
                ph[r >> 3][r & 7] = a;
                pl[r >> 3][r & 7] = c;
            }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                const unsigned char* vr = vb_ + (dt * 32 + l31) * VREC;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // keys (j & 3) + 8 (j >> 2) + 16 s + 4 half, j = 0..7: two runs of 4
                    const h4 a0 = *reinterpret_cast<const h4*>(vr + (16 * s + 4 * half) * 2);
                    const h4 a1 = *reinterpret_cast<const h4*>(vr + (16 * s + 4 * half + 8) * 2);
                    const h4 c0 = *reinterpret_cast<const h4*>(vr + 64 + (16 * s + 4 * half) * 2);
                    const h4 c1 = *reinterpret_cast<const h4*>(vr + 64 + (16 * s + 4 * half + 8) * 2);
                    const h8 vh = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const h8 vl = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s], o[dt], 0, 0, 0);
                }
            }
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
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

extern "C" int cf_attention_cf_masked(const float* q, long q_bs, const float* k, long k_bs, const float* v, long v_bs, float* out, int B,
                                      int heads, int d, int Nq, int Nk, int nk_valid, void* stream);

extern "C" int cf_attention_cf(const float* q, long q_bs, const float* k, long k_bs, const float* v, long v_bs, float* out, int B,
                               int heads, int d, int Nq, int Nk, void* stream) {
    return cf_attention_cf_masked(q, q_bs, k, k_bs, v, v_bs, out, B, heads, d, Nq, Nk, Nk, stream);
}

// The same with only the first nk_valid keys taking part in the softmax: a sequence whose length is not a multiple of 32 is padded by
// the caller (q, k, v zero-filled up to Nq, Nk; the padded query rows of `out` are discarded).
extern "C" int cf_attention_cf_masked(const float* q, long q_bs, const float* k, long k_bs, const float* v, long v_bs, float* out, int B,
                                      int heads, int d, int Nq, int Nk, int nk_valid, void* stream) {
    CF_REQUIRE(q && k && v && out, "null pointer");
    CF_REQUIRE(nk_valid > 0 && nk_valid <= Nk && nk_valid > Nk - 32, "nk_valid must lie in the last 32-key block (Nk=%d nk_valid=%d)", Nk, nk_valid);
    CF_REQUIRE(B > 0 && heads > 0 && Nq > 0 && Nk > 0, "bad shape");
    CF_REQUIRE(Nq % 32 == 0 && Nk % 32 == 0, "token counts must be multiples of 32 (Nq=%d Nk=%d)", Nq, Nk);
    CF_REQUIRE(d == 8 || d == 16 || d == 32 || d == 64, "head dim %d unsupported (8,16,32,64)", d);
    CF_REQUIRE(q_bs >= (long)heads * d * Nq && k_bs >= (long)heads * d * Nk && v_bs >= (long)heads * d * Nk, "batch stride too small");
    const float scale = (float)(1.0 / sqrt((double)d));
    dim3 grid((unsigned)(B * heads * ((Nq + 127) / 128)));
    hipStream_t s = as_stream(stream);
    // head dims >= 16 run on the f16 hi/lo-split kernel (CF_ATTN_F32=1 keeps the exact fp32-MFMA kernel); it reads V with 16-byte
    // loads: 8-key groups must be 16-byte aligned
    static int f32_only = -1;
    if (f32_only < 0) { const char* e = getenv("CF_ATTN_F32"); f32_only = e ? atoi(e) : 0; }
    const bool valign = ((reinterpret_cast<uintptr_t>(v) | (uintptr_t)(v_bs * 4)) & 15) == 0 && (Nk & 7) == 0;
    if (!f32_only && d >= 16 && valign) {
        if (d == 16) hipLaunchKernelGGL((attention_cf_f16s_kernel<16>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale, nk_valid);
        else if (d == 32) hipLaunchKernelGGL((attention_cf_f16s_kernel<32>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale, nk_valid);
        else hipLaunchKernelGGL((attention_cf_f16s_kernel<64>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale, nk_valid);
        CF_CHECK_LAUNCH();
        return CF_OK;
    }
    switch (d) {
        case 8: hipLaunchKernelGGL((attention_cf_kernel<8>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale, nk_valid); break;
        case 16: hipLaunchKernelGGL((attention_cf_kernel<16>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale, nk_valid); break;
        case 32: hipLaunchKernelGGL((attention_cf_kernel<32>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale, nk_valid); break;
        default: hipLaunchKernelGGL((attention_cf_kernel<64>), grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, out, heads, Nq, Nk, scale, nk_valid); break;
    }
    CF_CHECK_LAUNCH();
    return CF_OK;
}

// =====================================================================================================================
// Windowed cross-attention of the Swin skip filter (nnunet/lib/swin_cross_attention.py:13-112, :292-380): for every window of
// ws x ws tokens and every head, out = softmax(scale * q k^T + relative-position bias + shift mask) v, with q and k taken from one
// projected map and v from another.  The cyclic shift, the window partition, window_reverse and the roll back are index arithmetic:
// token (ty, tx) of window (wy, wx) sits at shifted position (ys, xs) = (wy*ws + ty, wx*ws + tx), i.e. at (ys + shift) mod H in the
// un-shifted map, and the result is written back to that same place.  Tokens of different border regions of the shifted image do not
// see each other (-100 on the logits, as the reference's attn_mask).  One workgroup per (sample, window, head): the window's q, k, v
// (N <= 64 tokens x hd <= 64) sit in LDS, thread (row i, quarter) owns 16 columns of the score row; fp32 throughout -- this network is
// the cropping / segmentation front end, not the frames/s path.
namespace cf {

__global__ void __launch_bounds__(256) window_attention_kernel(const float* __restrict__ qk, const float* __restrict__ v, const float* __restrict__ bias_table,
                                                              float* __restrict__ out, int C, int H, int W, int heads, int ws, int shift, float scale) {
    extern __shared__ float sm[];
    const int hd = C / heads, N = ws * ws;
    float* sq = sm;                    // [N][hd + 1]
    float* sk = sq + N * (hd + 1);
    float* sv = sk + N * (hd + 1);
    float* sp = sv + N * (hd + 1);     // [N][N + 1]
    const int nwx = W / ws, nwy = H / ws;
    int bid = blockIdx.x;
    const int head = bid % heads;
    bid /= heads;
    const int wx = bid % nwx;
    bid /= nwx;
    const int wy = bid % nwy;
    const int b = bid / nwy;
    const long HW = (long)H * W;
    const float* qb = qk + ((long)b * 2 * C + head * hd) * HW;
    const float* kb = qb + (long)C * HW;
    const float* vb = v + ((long)b * C + head * hd) * HW;
    for (int e = threadIdx.x; e < N * hd; e += blockDim.x) {
        const int n = e % N, d = e / N;                       // token fastest: neighbouring lanes read neighbouring pixels
        const int ys = wy * ws + n / ws, xs = wx * ws + n % ws;
        const long pix = (long)((ys + shift) % H) * W + (xs + shift) % W;
        sq[n * (hd + 1) + d] = qb[d * HW + pix] * scale;
        sk[n * (hd + 1) + d] = kb[d * HW + pix];
        sv[n * (hd + 1) + d] = vb[d * HW + pix];
    }
    __syncthreads();
    auto region = [&](int p, int S) { return p < S - ws ? 0 : (p < S - shift ? 1 : 2); };
    // scores: thread -> (row i, column j) pairs
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
        const int i = e / N, j = e % N;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc = fmaf(sq[i * (hd + 1) + d], sk[j * (hd + 1) + d], acc);
        const int tyi = i / ws, txi = i % ws, tyj = j / ws, txj = j % ws;
        acc += bias_table[((tyi - tyj + ws - 1) * (2 * ws - 1) + (txi - txj + ws - 1)) * heads + head];
        if (shift > 0) {
            const int ri = 3 * region(wy * ws + tyi, H) + region(wx * ws + txi, W), rj = 3 * region(wy * ws + tyj, H) + region(wx * ws + txj, W);
            if (ri != rj) acc += -100.0f;
        }
        sp[i * (N + 1) + j] = acc;
    }
    __syncthreads();
    // softmax per row (one thread per row: N <= 64 rows, short rows)
    if ((int)threadIdx.x < N) {
        float* row = sp + threadIdx.x * (N + 1);
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, row[j]);
        float sum = 0.f;
        for (int j = 0; j < N; ++j) { row[j] = expf(row[j] - mx); sum += row[j]; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < N; ++j) row[j] *= inv;
    }
    __syncthreads();
    float* ob = out + ((long)b * C + head * hd) * HW;
    for (int e = threadIdx.x; e < N * hd; e += blockDim.x) {
        const int n = e % N, d = e / N;
        float acc = 0.f;
        for (int j = 0; j < N; ++j) acc = fmaf(sp[n * (N + 1) + j], sv[j * (hd + 1) + d], acc);
        const int ys = wy * ws + n / ws, xs = wx * ws + n % ws;
        ob[d * HW + (long)((ys + shift) % H) * W + (xs + shift) % W] = acc;
    }
}

}  // namespace cf

extern "C" int cf_window_attention(const float* qk, const float* v, const float* bias_table, float* out, int B, int C, int H, int W, int heads,
                                   int window, int shift, void* stream) {
    CF_REQUIRE(qk && v && bias_table && out, "null pointer");
    CF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && heads > 0 && C % heads == 0, "bad shape");
    CF_REQUIRE(window > 0 && window <= 8 && H % window == 0 && W % window == 0, "window %d must divide the map %dx%d (and be <= 8)", window, H, W);
    CF_REQUIRE(shift >= 0 && shift < window, "shift must be in [0, window)");
    const int hd = C / heads, N = window * window;
    CF_REQUIRE(hd <= 64, "head dim %d > 64", hd);
    const size_t lds = sizeof(float) * ((size_t)3 * N * (hd + 1) + (size_t)N * (N + 1));
    const long nblk = (long)B * (H / window) * (W / window) * heads;
    CF_REQUIRE(nblk < (1L << 31), "too many windows");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cf::window_attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(cf::window_attention_kernel, dim3((unsigned)nblk), dim3(256), lds, cf::as_stream(stream), qk, v, bias_table, out, C, H, W, heads, window,
                       shift, (float)(1.0 / sqrt((double)hd)));
    CF_CHECK_LAUNCH();
    return CF_OK;
}
