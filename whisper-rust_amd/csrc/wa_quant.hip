// wa_quant.hip - quantised weights (ggml Q5_0 / Q8_0 model files) in the reference's order.
//
// With a quantised weight matrix the reference CPU path quantises the F32 activation row to Q8_0 (quantize_row_q8_0,
// ggml-cpu/arch/x86/quants.c, AVX2: d = max|x| / 127, q = rint(x * (127 / max|x|)), d stored as F16) and forms every output as
//     acc[l] = fma( f32(d_w) * f32(d_x),  (float) sum_{e<4} w[4l+e] * x[4l+e],  acc[l] )        l = 0..7, block after block
//     out    = ((acc0 + acc4) + (acc2 + acc6)) + ((acc1 + acc5) + (acc3 + acc7))                  (hsum_float_8)
// (ggml_vec_dot_q5_0_q8_0 / ggml_vec_dot_q8_0_q8_0, same file; the integer sums are exact: |w| <= 16 or 127, |x| <= 127).
// Here: 8 lanes per output row, lane l owns elements 4l..4l+3 of every 32-element block (one v_dot4_i32_i8 per block and
// token), the three DPP exchanges reproduce hsum_float_8.  Up to 8 activation rows share one pass over the weights.
// The weights are stored for this access pattern at load (wa_loader.cpp: signed bytes, [row][lane][block][4]).
// Bit-identical to the reference engine on the Q5_0 / Q8_0 goldens (tests/test_parity_gpu.py).
#include "wa_device.h"

// -------------------------------------------------------------------------------------------------
// quantize_row_q8_0: one 32-lane half-wave per block.  The quants are written in the kernel layout of the weights
// ([row][l = 0..7][block][4], wa_internal.h: wa_lin): lane l of a dot product reads four blocks with one 16-byte load.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_quantize_q8_0(const float * __restrict__ x, int ldx, int rows, int K, int8_t * __restrict__ qs,
                                                       float * __restrict__ qd) {
    const int nb = K >> 5;
    const long g = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const long gb = g >> 5;
    const int l = (int) (g & 31);
    if (gb >= (long) rows * nb) return;
    const int row = (int) (gb / nb), b = (int) (gb - (long) row * nb);
    wa_q8_store(x[(size_t) row * ldx + b * 32 + l], row, b, l, nb, qs, qd);
}
void wa_launch_quantize_q8_0(hipStream_t stream, const float * x, int ldx, int rows, int K, int8_t * qs, float * qd) {
    const long n = (long) rows * K;
    hipLaunchKernelGGL(k_quantize_q8_0, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, x, ldx, rows, K, qs, qd);
}

typedef int   wq_i4 __attribute__((ext_vector_type(4)));
typedef float wq_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wq_hsum8(float v) {
    v = v + dpp_f32<0x104>(v);          // row_shl:4  acc[l] + acc[l+4]
    v = v + dpp_f32<0x102>(v);          // row_shl:2  (a0+a4)+(a2+a6) | (a1+a5)+(a3+a7)
    v = v + dpp_f32<0x101>(v);          // row_shl:1  the two halves
    return v;
}
#define WQ_BLOCK(acc, w, dw, x, dx) acc = fmaf((dw) * (dx), (float) __builtin_amdgcn_sdot4((w), (x), 0, false), acc)
#define WQ_STEP4(acc, W, DW, X, DX) do { WQ_BLOCK(acc, (W).x, (DW).x, (X).x, (DX).x); WQ_BLOCK(acc, (W).y, (DW).y, (X).y, (DX).y); \
                                         WQ_BLOCK(acc, (W).z, (DW).z, (X).z, (DX).z); WQ_BLOCK(acc, (W).w, (DW).w, (X).w, (DX).w); } while (0)

// -------------------------------------------------------------------------------------------------
// C[M][N] = xq Wq^T, M > 1; grid = (ceil(N / 32), ceil(M / 8)); 256 threads = 32 output rows x 8 lanes; the 8 activation rows in LDS
// -------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void k_qgemm_exact(const int8_t * __restrict__ xq, const float * __restrict__ xd, int M, const int8_t * __restrict__ wq,
                                                     const float * __restrict__ wd, int N, int K, wa_epi e) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // xs int8 [8][K] (kernel layout) | xds f32 [8][K/32]
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
    const int * wl = (const int *) wq + ((size_t) nn * 8 + l) * nb;            // this lane's quads, block after block
    const float * dl = wd + (size_t) nn * nb;
    const int * xl = (const int *) xs + (size_t) l * nb;
    if ((nb & 3) == 0) {
        wq_i4 wn = *(const wq_i4 *) wl; wq_f4 dn = *(const wq_f4 *) dl;
        for (int b = 0; b < nb; b += 4) {
            const wq_i4 w = wn; const wq_f4 dw = dn;
            const int bn = min(b + 4, nb - 4);
            wn = *(const wq_i4 *) (wl + bn); dn = *(const wq_f4 *) (dl + bn);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const wq_i4 x = *(const wq_i4 *) (xl + (size_t) m * (K >> 2) + b);
                const wq_f4 dx = *(const wq_f4 *) (xds + m * nb + b);
                WQ_STEP4(acc[m], w, dw, x, dx);
            }
        }
    } else {
        for (int b = 0; b < nb; ++b) {
            const int w = wl[b]; const float dw = dl[b];
#pragma unroll
            for (int m = 0; m < 8; ++m) WQ_BLOCK(acc[m], w, dw, xl[(size_t) m * (K >> 2) + b], xds[m * nb + b]);
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float v = wq_hsum8(acc[m]);
        if (l == 0 && n < N && m < mt) epi_apply<EPI>(e, m0 + m, n, v, pre[m]);
    }
}

// -------------------------------------------------------------------------------------------------
// M == 1 (the decode step): grid = ceil(N / 8) single-wave workgroups, 8 output rows x 8 lanes each; no LDS, the activation row is
// read from L2 with the same 16-byte pattern as the weights; 16 blocks (4 loads of each kind) in flight ahead of the arithmetic
// -------------------------------------------------------------------------------------------------
// one output row per 8 lanes: the row's dot product with the activation row (valid in lane l == 0 of the group)
__device__ __forceinline__ float wq_row_dot(const int8_t * __restrict__ xq, const float * __restrict__ xd, const int8_t * __restrict__ wq,
                                            const float * __restrict__ wd, int nn, int nb, int l) {
    const int * wl = (const int *) wq + ((size_t) nn * 8 + l) * nb;
    const float * dl = wd + (size_t) nn * nb;
    const int * xl = (const int *) xq + (size_t) l * nb;
    float acc = 0.0f;
    if ((nb & 3) == 0) {
        wq_i4 wn[4], xn[4]; wq_f4 dn[4], en[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int bj = min(4 * j, nb - 4);
            wn[j] = *(const wq_i4 *) (wl + bj); dn[j] = *(const wq_f4 *) (dl + bj); xn[j] = *(const wq_i4 *) (xl + bj); en[j] = *(const wq_f4 *) (xd + bj);
        }
        for (int b = 0; b < nb; b += 16) {
            wq_i4 w[4], x[4]; wq_f4 dw[4], dx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { w[j] = wn[j]; x[j] = xn[j]; dw[j] = dn[j]; dx[j] = en[j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int bj = min(b + 16 + 4 * j, nb - 4);
                wn[j] = *(const wq_i4 *) (wl + bj); dn[j] = *(const wq_f4 *) (dl + bj); xn[j] = *(const wq_i4 *) (xl + bj); en[j] = *(const wq_f4 *) (xd + bj);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (b + 4 * j < nb) WQ_STEP4(acc, w[j], dw[j], x[j], dx[j]);
        }
    } else {
        for (int b = 0; b < nb; ++b) WQ_BLOCK(acc, wl[b], dl[b], xl[b], xd[b]);
    }
    return wq_hsum8(acc);
}

template <int EPI>
__global__ __launch_bounds__(64) void k_qgemv_exact(const int8_t * __restrict__ xq, const float * __restrict__ xd, const int8_t * __restrict__ wq,
                                                    const float * __restrict__ wd, int N, int K, wa_epi e) {
    const int tid = threadIdx.x, l = tid & 7;
    const int n = blockIdx.x * 8 + (tid >> 3);
    const int nn = n < N ? n : N - 1;
    wa_epi_pre pre;
    if (l == 0) pre = epi_preload<EPI>(e, 0, nn);
    const float v = wq_row_dot(xq, xd, wq, wd, nn, K >> 5, l);
    if (l == 0 && n < N) epi_apply<EPI>(e, 0, n, v, pre);
}

// the first MLP product of the decode step: 4 waves = 32 output rows = one Q8_0 block of the GELU output (N % 32 == 0)
__global__ __launch_bounds__(256) void k_qgemv_gelu_q8(const int8_t * __restrict__ xq, const float * __restrict__ xd, const int8_t * __restrict__ wq,
                                                       const float * __restrict__ wd, int N, int K, const float * __restrict__ bias,
                                                       const wa_f16 * __restrict__ gelu, int8_t * __restrict__ oq, float * __restrict__ oqd) {
    __shared__ float g[32];
    const int tid = threadIdx.x, l = tid & 7;
    const int n = blockIdx.x * 32 + (tid >> 3);
    const float bn = l == 0 ? bias[n] : 0.0f;
    const float v = wq_row_dot(xq, xd, wq, wd, n, K >> 5, l);
    if (l == 0) g[tid >> 3] = wa_gelu(v + bn, gelu);
    __syncthreads();
    if (tid < 32) wa_q8_store(g[tid], 0, blockIdx.x, tid, N >> 5, oq, oqd);
}
void wa_launch_qgemv_gelu_q8(hipStream_t s, const int8_t * xq, const float * xd, const int8_t * wq, const float * wd, int N, int K, const float * bias,
                             const wa_f16 * gelu, int8_t * oq, float * oqd) {
    hipLaunchKernelGGL(k_qgemv_gelu_q8, dim3(N / 32), dim3(256), 0, s, xq, xd, wq, wd, N, K, bias, gelu, oq, oqd);
}

void wa_launch_qgemm_exact(hipStream_t s, wa_epi_mode mode, const int8_t * xq, const float * xd, int M, const int8_t * wq, const float * wd, int N, int K,
                           const wa_epi & e) {
    const dim3 grid((N + 31) / 32, (M + 7) / 8);
    const size_t lds = (size_t) 8 * K + (size_t) 8 * (K >> 5) * sizeof(float);
#define WA_CASE(E) case E: { \
        if (M == 1) { hipLaunchKernelGGL((k_qgemv_exact<E>), dim3((N + 7) / 8), dim3(64), 0, s, xq, xd, wq, wd, N, K, e); break; } \
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) k_qgemm_exact<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
        hipLaunchKernelGGL((k_qgemm_exact<E>), grid, dim3(256), lds, s, xq, xd, M, wq, wd, N, K, e); } break;
    switch (mode) {
        WA_CASE(WA_EPI_F16) WA_CASE(WA_EPI_ENC_QKV) WA_CASE(WA_EPI_GELU_F32) WA_CASE(WA_EPI_RESID) WA_CASE(WA_EPI_F32) WA_CASE(WA_EPI_CROSS_KV) WA_CASE(WA_EPI_DEC_QKV)
        default: break;
    }
#undef WA_CASE
}

// -------------------------------------------------------------------------------------------------
// ggml_get_rows on the quantised token embedding (dequantize_row_q5_0 / q8_0, ggml-quants.c: q * d) + positional embedding
// -------------------------------------------------------------------------------------------------
__global__ void k_dec_embed_q(const int32_t * __restrict__ tok, const int32_t * __restrict__ pos, int n_tokens, int d,
                              const int8_t * __restrict__ wq, const float * __restrict__ wd, const float * __restrict__ pe, float * __restrict__ x) {
    const int j = blockIdx.x;
    const int t = tok[j], p = pos[j], nb = d >> 5;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        const int b = i >> 5, el = i & 31;
        const int q = (int) wq[(((size_t) t * 8 + (el >> 2)) * nb + b) * 4 + (el & 3)];
        x[(size_t) j * d + i] = (float) q * wd[(size_t) t * nb + b] + pe[(size_t) p * d + i];
    }
}
void wa_launch_dec_embed_q(hipStream_t stream, const int32_t * tok, const int32_t * pos, int n_tokens, int d, const int8_t * wq, const float * wd,
                           const float * pe, float * x) {
    hipLaunchKernelGGL(k_dec_embed_q, dim3(n_tokens), dim3(256), 0, stream, tok, pos, n_tokens, d, wq, wd, pe, x);
}
