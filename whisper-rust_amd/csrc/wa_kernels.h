// wa_kernels.h - launchers of the hand-written gfx950 kernels (wa_kernels.hip).
// Every launcher enqueues on `stream` and returns immediately; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

typedef uint16_t wa_f16;

#define WA_ATT_MAXKV 5120        // most KV cells one attention launch may see: 512 cells x (8 decoders + 2), whisper.cpp:7100-7106

// ---- GEMM epilogue descriptor -----------------------------------------------------------------
enum wa_epi_mode {
    WA_EPI_F16 = 0,     // out f16[m][n]            = f16((acc + bias[n]) * scale[n])
    WA_EPI_ENC_QKV,     // n <  split0: out f16[m][n]          (Q | K, ld = ldo)
                        // n >= split0: out2 f16[n-split0][m]  (V transposed, ld = ldo2)
    WA_EPI_GELU_F16,    // out f16[m][n]            = gelu_f16_table(acc + bias[n])
    WA_EPI_RESID,       // out f32[m][n]            = (acc + bias[n]) + resid[m][n]
    WA_EPI_CONV2,       // out f32[m][n]            = resid[m][n] + gelu(acc + bias[n]);  dbg[m][n] = gelu(...)
    WA_EPI_F32,         // out f32[m][n]            = acc (+ bias[n])
    WA_EPI_CROSS_KV,    // n -> (layer, k|v, head, c): out/out2 f16 [layer][head][aux0 = tpad][64]
    WA_EPI_GELU_F32,    // out f32[m][n]            = gelu_f16_table(acc + bias[n])      (quantised models: the next product quantises from F32)
    WA_EPI_DEC_QKV,     // n <  split0: out  f16[m][n]                        (scaled query)
                        // n <  split1: out2 f16[(row_off + m)][n - split0]   (scaled key  -> KV cell)
                        // else       : out3 f16[(row_off + m)][n - split1]   (value       -> KV cell)
    WA_EPI_ATTN_PV,     // the P V product of reference-order attention, one head per grid.y (wa_launch_attn_exact_mfma):
                        //   out f16[m][64 y + n] = f16(float(double(acc) + sum_{c < aux1} double(float(W[n][aux0 + c]) * float(pl[m][c]))))
                        //   (aux0 = np = n_kv & ~31, aux1 = n_kv - np leftover cells in F64 as vec.cpp:221-223; pl = out2, 32 per row)
};

// One row of a lock-step decode step over several independent chunks (wa_decode.cpp: wa_batcher): where THIS row's state keeps its
// self K/V cells and its encoder K/V (layer 0; the kernels add the layer offset, equal for all rows), and its cell range.
struct wa_rowptr { wa_f16 * kv_k; wa_f16 * kv_v; const wa_f16 * cross_k; const wa_f16 * cross_v; int n_kv; int kv_head; };

struct wa_epi {
    const float * bias  = nullptr;
    const float * scale = nullptr;
    void * out  = nullptr; int ldo  = 0;
    void * out2 = nullptr; int ldo2 = 0;
    void * out3 = nullptr; int ldo3 = 0;
    const float * resid = nullptr; int ldr = 0;
    float * dbg = nullptr;
    const wa_f16 * gelu = nullptr;
    int split0 = 0, split1 = 0;
    int row_off = 0;
    int aux0 = 0, aux1 = 0;
    const int * dyn = nullptr;   // device {n_kv, kv_head}: when set, row_off is read from dyn[1] (graph-replayed decode step)
    long long bs_a = 0, bs_w = 0, bs_o2 = 0;   // batched launch (grid.y = batch index y): element offsets y * bs_* added to A, W and out2
    const wa_rowptr * rowp = nullptr; long long rowp_off = 0;      // WA_EPI_DEC_QKV with rows of DIFFERENT states: row m's key / value go to its own cell
};

// C[M x N] = A[M x K] (f16, row stride lda) * W[N x K]^T (f16, row stride ldw); K % 32 == 0.
// MFMA path (any M); rows/cols beyond M/N are neither read out of bounds (clamped) nor stored.
void wa_launch_gemm(hipStream_t stream, wa_epi_mode mode, const wa_f16 * A, int lda, const wa_f16 * W, int ldw,
                    int M, int N, int K, const wa_epi & e);

// ---- log-mel ------------------------------------------------------------------------------------
// pcm: device f32[n_samples]. mel: device f32[n_mel][n_len]. mel_max: device scratch (1 uint).
void wa_launch_mel(hipStream_t stream, const float * pcm, int n_samples, const float * hann, const float * sincos,
                   const float * filters, int n_mel, int n_fft_bins, float * mel, int n_len, unsigned int * mel_max);

// mel window [seek, seek + 2*n_ctx) -> time-major f16 with one zero row in front and zero rows behind
void wa_launch_mel_window(hipStream_t stream, const float * mel, int n_mel, int n_len, int seek, int n_frames,
                          wa_f16 * melT, int rows_total);

// ---- LayerNorm of the tolerance path (F32 sums): y = ((x - mean) * rsqrt(var + eps)) * w + b;  out16 f16 and/or out32 f32; d % 4 == 0
void wa_launch_layernorm(hipStream_t stream, const float * x, int ldx, int rows, int d, const float * w, const float * b,
                         float eps, wa_f16 * out16, int ld16, float * out32, int ld32);

// ---- encoder self-attention (tolerance path, one sweep with a running maximum): qk [T][2d] (Q | K), vt [d][tpad] ----------
void wa_launch_enc_attn(hipStream_t stream, const wa_f16 * qk, int ldqk, const wa_f16 * vt, int ldvt, int T, int d,
                        int n_head, float scale, wa_f16 * out, int ldo);

// ---- decoder ------------------------------------------------------------------------------------
void wa_launch_dec_embed(hipStream_t stream, const int32_t * tok, const int32_t * pos, int n_tokens, int d,
                         const wa_f16 * te, const float * pe, float * x);

// ---- reference-order kernels (wa_exact.hip): bit-identical to the reference's ggml-cpu AVX2 path ----
void wa_launch_gemm_exact(hipStream_t stream, wa_epi_mode mode, const wa_f16 * A, int lda, const wa_f16 * W, int ldw, int M, int N, int K,
                          const wa_epi & e);
void wa_launch_gemv_exact(hipStream_t stream, wa_epi_mode mode, const wa_f16 * A, int lda, const int32_t * rows, const wa_f16 * W, int ldw,
                          int M, int N, int K, const wa_epi & e);
// Same GEMV with LayerNorm(x) fused in front (x F32 [M][K]): one launch instead of two on the decode step.
void wa_launch_ln_gemv_exact(hipStream_t stream, wa_epi_mode mode, const float * x, int ldx, const int32_t * rows, const float * ln_w,
                             const float * ln_b, float eps, const wa_f16 * W, int ldw, int M, int N, int K, const wa_epi & e);
void wa_launch_im2col3(hipStream_t stream, const wa_f16 * src, int src_ld, int row0, int stride, int IC, int OL, wa_f16 * dst, int dst_ld);
void wa_launch_layernorm_exact(hipStream_t stream, const float * x, int ldx, int rows, int d, const float * w, const float * b, float eps,
                               wa_f16 * out16, int ld16, float * out32, int ld32, int8_t * qs = nullptr, float * qd = nullptr);
// q [n_tokens][ldq] f16; K row c of head h at kbase + h*k_head_stride + c*k_row_stride (64 halfs), V likewise.
// partial: f32 [n_tokens][n_head][32][64], p_left: f16 [n_tokens][n_head][32] (used when n_tokens*n_head < 512).
// Reference-order self-attention of the encoder on the matrix cores (T queries = T keys of every head, no mask), bit-identical to
// wa_launch_attn_exact: scores + soft-max by k_attn_scores_mfma into the probability buffer `p` f16 [n_head][T][kvp] (+ the leftover
// cells' probabilities `p_left` f16 [n_head][T][32]), then P V as a batched reference-order GEMM against V^T.
// qk [T][ldqk] (Q | K), vt [d][ldvt] (V transposed, ldvt >= kvp, finite beyond T), kvp = T rounded up to 128.
void wa_launch_attn_exact_mfma(hipStream_t stream, const wa_f16 * qk, int ldqk, const wa_f16 * vt, int ldvt, int T, int d, int n_head, float scale,
                               wa_f16 * p, wa_f16 * p_left, int kvp, wa_f16 * out, int ldo,
                               float * out32 = nullptr /* when set: the result in F32 [T][ldo] instead (quantised models: the next product quantises it) */);
void wa_launch_attn_exact(hipStream_t stream, const wa_f16 * q, int ldq, const wa_f16 * kbase, size_t k_head_stride, int k_row_stride,
                          const wa_f16 * vbase, size_t v_head_stride, int v_row_stride, int n_head, int n_tokens, int n_kv, const int8_t * mask,
                          float scale, float * partial, wa_f16 * p_left, wa_f16 * out, int ldo, float * qk_out, const int * dyn_n_kv = nullptr,
                          float * out32 = nullptr /* when set: the result in F32 [n_tokens][ldo] instead of F16 (quantised models) */,
                          int8_t * q8 = nullptr, float * q8d = nullptr /* when set: the F32 result as Q8_0 rows in the layout of wa_launch_quantize_q8_0 */,
                          const wa_rowptr * rowp = nullptr, int rowp_cross = 0, long long rowp_off = 0
                          /* rowp: query row j reads ITS state's K / V (self cells [0, rowp[j].n_kv), or the encoder's when rowp_cross) at + rowp_off */);

// ---- quantised weights (wa_quant.hip): ggml's Q5_0 / Q8_0 x Q8_0 products in the reference's AVX2 order ----
// quantize_row_q8_0 (arch/x86/quants.c): x f32 [rows][ldx] -> qs int8 [rows][8][K/32][4] (kernel layout), qd f32 [rows][K/32] (block scale, rounded through F16)
void wa_launch_quantize_q8_0(hipStream_t stream, const float * x, int ldx, int rows, int K, int8_t * qs, float * qd);
// C[M][N] = xq . Wq^T, ggml_vec_dot_q5_0_q8_0 / q8_0_q8_0 order; wq int8 [N][8][K/32][4], wd f32 [N][K/32] (wa_internal.h: wa_lin); any M
void wa_launch_qgemm_exact(hipStream_t stream, wa_epi_mode mode, const int8_t * xq, const float * xd, int M, const int8_t * wq, const float * wd, int N, int K,
                           const wa_epi & e);
// one row (K <= 2048, K % 32 == 0): LayerNorm in reference order, quantised to Q8_0 row 0 of (qs, qd), by one 256-thread block
void wa_launch_ln_q8_row(hipStream_t stream, const float * x, int K, const float * w, const float * b, float eps, int8_t * qs, float * qd);
// M == 1: GELU(x Wq^T + bias) quantised to Q8_0 straight away (the operand of the second MLP product); N % 32 == 0
void wa_launch_qgemv_gelu_q8(hipStream_t stream, const int8_t * xq, const float * xd, const int8_t * wq, const float * wd, int N, int K, const float * bias,
                             const wa_f16 * gelu, int8_t * oq, float * oqd);
// token embedding rows of a quantised matrix (dequantize_row_q5_0 / q8_0) + positional embedding
void wa_launch_dec_embed_q(hipStream_t stream, const int32_t * tok, const int32_t * pos, int n_tokens, int d, const int8_t * wq, const float * wd,
                           const float * pe, float * x);
