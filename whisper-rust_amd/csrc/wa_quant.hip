// wa_quant.hip - quantised weights (ggml Q5_0 / Q8_0 model files) in the reference's order.
//
// With a quantised weight matrix the reference CPU path quantises the F32 activation row to Q8_0 (quantize_row_q8_0,
// ggml-cpu/arch/x86/quants.c, AVX2: d = max|x| / 127, q = rint(x * (127 / max|x|)), d stored as F16) and forms every output as
//     acc[l] = fma( f32(d_w) * f32(d_x),  (float) sum_{e<4} w[4l+e] * x[4l+e],  acc[l] )        l = 0..7, block after block
//     out    = ((acc0 + acc4) + (acc2 + acc6)) + ((acc1 + acc5) + (acc3 + acc7))                  (hsum_float_8)
// (ggml_vec_dot_q5_0_q8_0 / ggml_vec_dot_q8_0_q8_0, same file; the integer sums are exact: |w| <= 16 or 127, |x| <= 127).
// Here: 8 lanes per output row, lane l owns elements 4l..4l+3 of every 32-element block (one v_dot4_i32_i8 per block and
// token), the three DPP exchanges reproduce hsum_float_8.  Up to 8 activation rows share one pass over the weights.
// Bit-identical to the reference engine on the Q5_0 / Q8_0 goldens (tests/test_parity_gpu.py).
#include "wa_device.h"

// -------------------------------------------------------------------------------------------------
// quantize_row_q8_0: one 32-lane half-wave per block
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_quantize_q8_0(const float * __restrict__ x, int ldx, int rows, int K, int8_t * __restrict__ qs,
                                                       float * __restrict__ qd) {
    const int nb = K >> 5;
    const long g = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const long gb = g >> 5;
    const int l = (int) (g & 31);
    if (gb >= (long) rows * nb) return;
    const int row = (int) (gb / nb), b = (int) (gb - (long) row * nb);
    const float v = x[(size_t) row * ldx + b * 32 + l];
    float a = fabsf(v);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 32));
    const float d = a / 127.f;
    const float id = a != 0.0f ? 127.f / a : 0.0f;
    const float r = rintf(v * id);                          // _mm256_round_ps(_MM_ROUND_NEAREST): to nearest, ties to even
    qs[(size_t) row * K + b * 32 + l] = (int8_t) (int) r;
    if (l == 0) qd[(size_t) row * nb + b] = h2f(f2h(d));   // the dot product reads the scale back from its F16 field
}
void wa_launch_quantize_q8_0(hipStream_t stream, const float * x, int ldx, int rows, int K, int8_t * qs, float * qd) {
    const long n = (long) rows * K;
    hipLaunchKernelGGL(k_quantize_q8_0, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, x, ldx, rows, K, qs, qd);
}

// -------------------------------------------------------------------------------------------------
// elements 4l..4l+3 of a weight block as four signed bytes
// -------------------------------------------------------------------------------------------------
template <int QT>
__device__ __forceinline__ int wa_q_quad(const uint8_t * __restrict__ wqs, const uint32_t * __restrict__ wqh, size_t blk, int l) {
    if (QT == 8) return *(const int *) (wqs + blk * 32 + 4 * l);
    // Q5_0 (ggml-common.h:187-193): element j < 16 = low nibble of qs[j], element j + 16 = high nibble; bit e of qh = fifth bit of element e;
    // value = (nibble | bit << 4) - 16, i.e. nibble if the bit is set, nibble | 0xF0 (as a signed byte) if not
    const uint32_t q = *(const uint32_t *) (wqs + blk * 16 + 4 * (l & 3));
    const uint32_t nib = (l < 4 ? q : q >> 4) & 0x0f0f0f0fu;
    const uint32_t nb = ~(wqh[blk] >> (4 * l)) & 0xfu;                                       // bits NOT set, elements 4l..4l+3
    const uint32_t spread = (nb & 1u) | ((nb & 2u) << 7) | ((nb & 4u) << 14) | ((nb & 8u) << 21);
    return (int) (nib | spread * 0xf0u);
}

// -------------------------------------------------------------------------------------------------
// C[M][N] = xq Wq^T; grid = (ceil(N / 32), ceil(M / 8)); 256 threads = 32 output rows x 8 lanes
// -------------------------------------------------------------------------------------------------
template <int QT, int EPI>
__global__ __launch_bounds__(256) void k_qgemm_exact(const int8_t * __restrict__ xq, const float * __restrict__ xd, int M, const uint8_t * __restrict__ wqs,
                                                     const uint32_t * __restrict__ wqh, const float * __restrict__ wqd, int N, int K, wa_epi e) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // xs int8 [8][K] | xds f32 [8][K/32]
    const int nb = K >> 5;
    int8_t * xs = (int8_t *) smem;
    float * xds = (float *) (smem + (size_t) 8 * K);
    const int tid = threadIdx.x, l = tid & 7;
    const int m0 = blockIdx.y * 8, mt = min(8, M - m0);
    for (int c = tid; c < 8 * (K >> 4); c += 256) {
        const int m = c / (K >> 4), cc = c - m * (K >> 4);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m < mt) v = *(const uint4 *) (xq + (size_t) (m0 + m) * K + cc * 16);
        *(uint4 *) (xs + (size_t) m * K + cc * 16) = v;
    }
    for (int c = tid; c < 8 * nb; c += 256) { const int m = c / nb, b = c - m * nb; xds[c] = m < mt ? xd[(size_t) (m0 + m) * nb + b] : 0.0f; }
    __syncthreads();
    const int n = blockIdx.x * 32 + (tid >> 3);
    const int nn = n < N ? n : N - 1;
    float acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = 0.0f;
    wa_epi_pre pre[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) if (l == 0 && m < mt) pre[m] = epi_preload<EPI>(e, m0 + m, nn);
#pragma unroll 4
    for (int b = 0; b < nb; ++b) {
        const size_t blk = (size_t) nn * nb + b;
        const int w4 = wa_q_quad<QT>(wqs, wqh, blk, l);
        const float dw = wqd[blk];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int x4 = *(const int *) (xs + (size_t) m * K + b * 32 + 4 * l);
            const int isum = __builtin_amdgcn_sdot4(w4, x4, 0, false);
            acc[m] = fmaf(dw * xds[m * nb + b], (float) isum, acc[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        float v = acc[m];
        v = v + dpp_f32<0x104>(v);          // row_shl:4  acc[l] + acc[l+4]
        v = v + dpp_f32<0x102>(v);          // row_shl:2  (a0+a4)+(a2+a6) | (a1+a5)+(a3+a7)
        v = v + dpp_f32<0x101>(v);          // row_shl:1  the two halves
        if (l == 0 && n < N && m < mt) epi_apply<EPI>(e, m0 + m, n, v, pre[m]);
    }
}

template <int QT>
static void qgemm_dispatch(hipStream_t s, wa_epi_mode mode, const int8_t * xq, const float * xd, int M, const uint8_t * wqs, const uint32_t * wqh,
                           const float * wqd, int N, int K, const wa_epi & e) {
    const dim3 grid((N + 31) / 32, (M + 7) / 8);
    const size_t lds = (size_t) 8 * K + (size_t) 8 * (K >> 5) * sizeof(float);
#define WA_CASE(E) case E: { \
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) k_qgemm_exact<QT, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
        hipLaunchKernelGGL((k_qgemm_exact<QT, E>), grid, dim3(256), lds, s, xq, xd, M, wqs, wqh, wqd, N, K, e); } break;
    switch (mode) {
        WA_CASE(WA_EPI_F16) WA_CASE(WA_EPI_ENC_QKV) WA_CASE(WA_EPI_GELU_F32) WA_CASE(WA_EPI_RESID) WA_CASE(WA_EPI_F32) WA_CASE(WA_EPI_CROSS_KV) WA_CASE(WA_EPI_DEC_QKV)
        default: break;
    }
#undef WA_CASE
}
void wa_launch_qgemm_exact(hipStream_t stream, wa_epi_mode mode, const int8_t * xq, const float * xd, int M, int wtype, const uint8_t * wqs,
                           const uint32_t * wqh, const float * wqd, int N, int K, const wa_epi & e) {
    if (wtype == 6) qgemm_dispatch<6>(stream, mode, xq, xd, M, wqs, wqh, wqd, N, K, e);
    else            qgemm_dispatch<8>(stream, mode, xq, xd, M, wqs, wqh, wqd, N, K, e);
}

// -------------------------------------------------------------------------------------------------
// ggml_get_rows on the quantised token embedding (dequantize_row_q5_0 / q8_0, ggml-quants.c) + positional embedding
// -------------------------------------------------------------------------------------------------
__global__ void k_dec_embed_q(const int32_t * __restrict__ tok, const int32_t * __restrict__ pos, int n_tokens, int d, int wtype,
                              const uint8_t * __restrict__ wqs, const uint32_t * __restrict__ wqh, const float * __restrict__ wqd,
                              const float * __restrict__ pe, float * __restrict__ x) {
    const int j = blockIdx.x;
    const int t = tok[j], p = pos[j], nb = d >> 5;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        const size_t blk = (size_t) t * nb + (i >> 5);
        const int el = i & 31;
        int q;
        if (wtype == 6) {
            const uint8_t byte = wqs[blk * 16 + (el & 15)];
            q = (int) ((el < 16 ? byte & 0x0f : byte >> 4) | (((wqh[blk] >> el) & 1u) << 4)) - 16;
        } else q = (int) (int8_t) wqs[blk * 32 + el];
        x[(size_t) j * d + i] = (float) q * wqd[blk] + pe[(size_t) p * d + i];
    }
}
void wa_launch_dec_embed_q(hipStream_t stream, const int32_t * tok, const int32_t * pos, int n_tokens, int d, int wtype, const uint8_t * wqs,
                           const uint32_t * wqh, const float * wqd, const float * pe, float * x) {
    hipLaunchKernelGGL(k_dec_embed_q, dim3(n_tokens), dim3(256), 0, stream, tok, pos, n_tokens, d, wtype, wqs, wqh, wqd, pe, x);
}
