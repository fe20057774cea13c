// wa_rows.hip - the decode step for 2..8 token rows (whisper.cpp:2474-2852, n_tokens = rows) as ONE launch.  Design: wa_rows.h.
//
// Phases of a layer (the edges are wa_mega.hip's, one run of granules per token row):
//   P1  LayerNorm(x) -> q|k|v rows (E_QKV; new key / value also to their KV cells)      P2  self-attention units (row, head)   -> E_AO
//   P3  out-projection + residual (E_X1)   P4  LayerNorm -> cross query (E_QC)           P5  cross-attention units (row, head, quarter) -> E_AO2
//   P6  out-projection + residual (E_X2)   P7  LayerNorm -> FC1 + GELU (E_HF)            P8  FC2 + residual (E_X3)
// then the final LayerNorm and the logits rows.  A product phase = [gather or LayerNorm into LDS] -> barrier -> products; the weight
// chunk of a phase was requested by wave 7 at the barrier of the phase before (LDS-DMA into the other slot) and is waited for by wave 7
// alone, right before the barrier that starts the products.
#include "wa_device.h"
#include "wa_rows.h"
#include <algorithm>

typedef unsigned long long u64;
#define GAS __attribute__((address_space(1)))
#define LAS __attribute__((address_space(3)))
typedef GAS u64 gu64;
typedef GAS unsigned gu32;
typedef const GAS wa_f16 * gch;
typedef const GAS float * gcf;
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef const __attribute__((address_space(4))) wa_rows_args * mb_kargs;
typedef const __attribute__((address_space(4))) wa_mega_layer * mb_layers;
__device__ __forceinline__ mb_kargs mb_uniform(mb_kargs p) {
    const unsigned long long v = (unsigned long long) p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v), hi = __builtin_amdgcn_readfirstlane((unsigned) (v >> 32));
    return (mb_kargs) (((unsigned long long) hi << 32) | lo);
}

#define MB_THREADS 512
#define MB_NW 8
#define MB_NCW 7                  // waves that compute products; wave 7 streams the weights
#define MB_SPIN_LIMIT 20000u      // polls (~0.5 us each, ~10 ms) before a hand-off is declared dead (the host pauses the form and tries again later)
#define MB_TRACE_LAYER 5
#define MB_PAD 64                 // bytes behind every weight row in LDS: consecutive rows start 16 banks apart

enum { E_QKV = 0, E_AO, E_X1, E_QC, E_AO2, E_X2, E_HF, E_X3 };

struct mb_ctl { gu32 * status; unsigned seq; bool dead; };

__device__ __forceinline__ void gr_store(gu64 * g, unsigned seq, unsigned v) {
    __hip_atomic_store(g, ((u64) seq << 32) | (u64) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gr_load(gu64 * g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) { return (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, 0xf, 0xf, true); }

__device__ __forceinline__ void mb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the barrier in front of a product phase: wave 7's LDS-DMA of this phase's weights has landed
__device__ __forceinline__ void mb_barrier_w(int wave) {
    if (wave == MB_NW - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mb_barrier();
}

// optional timeline (tools/rows_trace.py): 100 MHz wall-clock ticks of one workgroup, 32 stamps per layer
__device__ __forceinline__ void mb_trace(mb_kargs A, bool who, int slot) {
    if (A->dbg && who) ((GAS unsigned *) A->dbg)[slot] = (unsigned) wall_clock64();
}

// MB_CHAOS (the test build, libwhisper_chaos.so): waves and whole workgroups stall at random for ~25 us in front of products, units and gathers,
// so that the rest of the workgroup - and of the grid - runs far ahead of them.  Results must not change.
#ifdef MB_CHAOS
__device__ __forceinline__ void mb_chaos(unsigned a, unsigned b, unsigned c_, unsigned phase, unsigned seq) {
    unsigned h = (a * 2654435761u) ^ (b * 40503u) ^ (c_ * 2246822519u) ^ (phase * 3266489917u) ^ (seq * 668265263u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    if ((h & 7u) == 0u) for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);
}
#define MB_CHAOS_AT(phase) mb_chaos((unsigned) blockIdx.x, (unsigned) (threadIdx.x >> 6), (unsigned) l, (phase), seq)
#define MB_CHAOS_WG(phase) mb_chaos((unsigned) blockIdx.x, 99u, (unsigned) l, (phase), seq)
#else
#define MB_CHAOS_AT(phase) do { } while (0)
#define MB_CHAOS_WG(phase) do { } while (0)
#endif

__device__ __forceinline__ gu64 * mb_edge(mb_kargs A, int layer, int e) {
    return (gu64 *) A->granules + ((size_t) layer * WA_MEGA_EDGES + e) * ((size_t) A->B * A->row_gr);
}

// One wave polls the granules idx(0..NPL-1) (idx < 0: none) until every tag equals this launch's sequence number (as wa_mega.hip: mg_sweep).
template <int NPL, typename F>
__device__ __forceinline__ void mb_sweep(gu64 * g, F idx, mb_ctl & c, int lane, unsigned (&v)[NPL], unsigned code) {
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {      // unconditional loads (a lane without a granule reads granule 0): predicated ones are issued one round trip at a time
            const int i = idx(k);
            const u64 x = gr_load(g + (i >= 0 ? i : 0)); v[k] = (unsigned) x; ok &= i < 0 || (unsigned) (x >> 32) == c.seq;
        }
        if (__all(ok) || c.dead) return;
        if ((spins & 127u) == 127u) {
            const unsigned st = __builtin_amdgcn_readfirstlane(__hip_atomic_load(c.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (st != 0u) { c.dead = true; return; }
            if (spins >= MB_SPIN_LIMIT) {
                if (lane == 0) __hip_atomic_store(c.status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c.dead = true;
                return;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// all 512 threads: `per_row` granules of each of B token rows of `edge` (row b's run starts at b * row_gr), eight per thread and round,
// handed to store(b, j, value) once valid
template <int NPL, typename ST>
__device__ __forceinline__ void mb_gather(mb_ctl & c, gu64 * edge, int B, int per_row, int row_gr, int tid, int lane, ST store, unsigned code) {
    const int total = B * per_row;
    for (int base = 0; base < total; base += MB_THREADS * NPL) {
        unsigned v[NPL];
        mb_sweep<NPL>(edge, [&](int k) { const int i = base + tid + MB_THREADS * k; if (i >= total) return -1; const int b = i / per_row; return b * row_gr + (i - b * per_row); },
                    c, lane, v, code);
#pragma unroll
        for (int k = 0; k < NPL; ++k) { const int i = base + tid + MB_THREADS * k; if (i < total) { const int b = i / per_row; store(b, i - b * per_row, v[k]); } }
    }
}

// -------------------------------------------------------------------------------------------------
// Weight rows -> LDS by LDS-DMA (wave 7).  The LDS image is rows of `pitch` bytes + MB_PAD, lane-linear in 16-byte cells (an LDS-DMA
// instruction writes 64 consecutive cells): every lane works out which (row, column) its cell is and reads THAT address; cells of the
// padding (and behind the last row) read a valid dummy address.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mb_dma_rows(const GAS unsigned char * W, int pitch, int row0, int R, int n_rows_total, unsigned char * slot, int lane) {
    const int cpr = (pitch >> 4) + (MB_PAD >> 4), total = R * cpr, np = (total + 63) >> 6, cvalid = pitch >> 4;
    const float inv = 1.0f / (float) cpr;
    for (int p = 0; p < np; ++p) {
        const int i = p * 64 + lane;
        int ri = (int) ((float) i * inv);
        int cc = i - ri * cpr;
        if (cc < 0) { ri -= 1; cc += cpr; } else if (cc >= cpr) { ri += 1; cc -= cpr; }
        int gr = row0 + (ri < R ? ri : R - 1);
        gr = gr < n_rows_total ? gr : n_rows_total - 1;
        const GAS unsigned char * src = W + (size_t) gr * pitch + (size_t) (cc < cvalid ? cc : 0) * 16;
        __builtin_amdgcn_global_load_lds((const GAS void *) src, (LAS void *) (slot + (size_t) p * 1024), 16, 0, 0);
    }
}

// `n` consecutive floats src[i0 .. i0 + n) (clamped to n_total) -> LDS dst, one dword per lane (no alignment requirement on i0), n <= 64
__device__ __forceinline__ void mb_dma_f32(const float * src, int i0, int n_total, float * dst, int lane) {
    const int i = i0 + lane < n_total ? i0 + lane : n_total - 1;
    __builtin_amdgcn_global_load_lds((const GAS void *) ((gcf) src + i), (LAS void *) dst, 4, 0, 0);
}
// a whole F32 vector of n floats (n % 4 == 0, 16-byte aligned) -> LDS
__device__ __forceinline__ void mb_dma_vec(const float * src, int n, float * dst, int lane) {
    const int cells = n >> 2;
    for (int p = 0; p * 64 < cells; ++p) {
        const int i = p * 64 + lane;
        __builtin_amdgcn_global_load_lds((const GAS void *) ((gcf) src + 4 * (i < cells ? i : cells - 1)), (LAS void *) ((unsigned char *) dst + (size_t) p * 1024), 16, 0, 0);
    }
}

// -------------------------------------------------------------------------------------------------
// Products in ggml_vec_dot_f16 order (vec.cpp:191-231; as k_gemv_exact / wa_mega.hip: mg_dot8, mg_dot16), weights and activations in LDS,
// BC token rows per pass over the weight row.
//   8 lanes per weight row : lane u owns elements 4u..4u+3 of every 32-element step;  results in lanes with u == 0
//   16 lanes per weight row: lane u owns elements 2u, 2u+1 (the 4d-long rows);          results in lanes with u == 0
// -------------------------------------------------------------------------------------------------
template <int BC>
__device__ __forceinline__ void mb_dot8(const unsigned char * wrow, const wa_f16 * xs, int ldx, int b0, int B, int nsteps, float (&res)[BC]) {
    float acc[BC][4];
    const wa_f16 * xb[BC];
#pragma unroll
    for (int j = 0; j < BC; ++j) { acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.0f; xb[j] = xs + (size_t) (b0 + j < B ? b0 + j : B - 1) * ldx; }
#pragma unroll 4
    for (int s = 0; s < nsteps; ++s) {
        const half4v w4 = *(const half4v *) (wrow + s * 64);
#pragma unroll
        for (int j = 0; j < BC; ++j) {
            const half4v x4 = *(const half4v *) (xb[j] + s * 32);
            acc[j][0] = fmaf((float) w4[0], (float) x4[0], acc[j][0]);
            acc[j][1] = fmaf((float) w4[1], (float) x4[1], acc[j][1]);
            acc[j][2] = fmaf((float) w4[2], (float) x4[2], acc[j][2]);
            acc[j][3] = fmaf((float) w4[3], (float) x4[3], acc[j][3]);
        }
    }
#pragma unroll
    for (int j = 0; j < BC; ++j) {
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = acc[j][i];
            v = v + dpp_f32<0x104>(v);          // row_shl:4  s[j] + s[j+2]
            v = v + dpp_f32<0x102>(v);          // row_shl:2  (s0+s2) + (s1+s3)
            t[i] = v + dpp_f32<0x101>(v);       // row_shl:1  a[l] + a[l+4]
        }
        res[j] = (t[0] + t[1]) + (t[2] + t[3]);
    }
}
template <int BC>
__device__ __forceinline__ void mb_dot16(const unsigned char * wrow, const wa_f16 * xs, int ldx, int b0, int B, int nsteps, float (&res)[BC]) {
    float acc[BC][2];
    const wa_f16 * xb[BC];
#pragma unroll
    for (int j = 0; j < BC; ++j) { acc[j][0] = acc[j][1] = 0.0f; xb[j] = xs + (size_t) (b0 + j < B ? b0 + j : B - 1) * ldx; }
#pragma unroll 8
    for (int s = 0; s < nsteps; ++s) {
        const half2v w2 = *(const half2v *) (wrow + s * 64);
#pragma unroll
        for (int j = 0; j < BC; ++j) {
            const half2v x2 = *(const half2v *) (xb[j] + s * 32);
            acc[j][0] = fmaf((float) w2[0], (float) x2[0], acc[j][0]);
            acc[j][1] = fmaf((float) w2[1], (float) x2[1], acc[j][1]);
        }
    }
#pragma unroll
    for (int j = 0; j < BC; ++j) {
        float t[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v = acc[j][i];
            v = v + dpp_f32<0x108>(v);          // row_shl:8  s[j] + s[j+2]
            v = v + dpp_f32<0x104>(v);          // row_shl:4  (s0+s2) + (s1+s3)
            t[i] = v + dpp_f32<0x102>(v);       // row_shl:2  a[l] + a[l+4]   -> lane 0: t0,t1  lane 1: t2,t3
        }
        const float r = t[0] + t[1];
        res[j] = r + dpp_f32<0x101>(r);         // (t0+t1) + (t2+t3)
    }
}

// -------------------------------------------------------------------------------------------------
// Quantised models (Q5_0 / Q8_0 files; contract: wa_quant.hip - ggml_vec_dot_q5_0_q8_0 / q8_0_q8_0 in the AVX2 order).  A weight row is
// [lane u = 0..7][block][4] signed bytes + [block] F32 scales; an operand row sits in LDS quantised to Q8_0 in the same order
// (quads [u][block | 8 words apart][4], then the block scales).  Lane u chains  acc = fma(dw dx, (float) dot4(w, x), acc)  over the blocks in
// order; three DPP adds are hsum_float_8.
// -------------------------------------------------------------------------------------------------
typedef int   mq_i4 __attribute__((ext_vector_type(4)));
typedef float mq_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int mq_ld(int nb) { return nb | 8; }                                   // words between the quad rows u of an operand row
__host__ __device__ __forceinline__ size_t mq_row_bytes(int nb) { return (size_t) 32 * (nb | 8) + (size_t) 4 * nb; }
__device__ __forceinline__ unsigned * mq_quads(unsigned char * op, int nb, int u) { return (unsigned *) op + u * mq_ld(nb); }
__device__ __forceinline__ float * mq_scales(unsigned char * op, int nb) { return (float *) (op + (size_t) 32 * mq_ld(nb)); }
// quantize_row_q8_0 (arch/x86/quants.c) of a 32-element block held one value per lane of a half-wave (as wa_q8_store): the quad of lanes
// 4k..4k+3 packed over DPP into lane 4k; returns the block's scale (rounded through F16).  All lanes take part.
__device__ __forceinline__ unsigned mq_quant32(float y, float & dq) {
    float a = fabsf(y);
    a = fmaxf(a, dpp_f32<0x128>(a)); a = fmaxf(a, dpp_f32<0x124>(a)); a = fmaxf(a, dpp_f32<0x122>(a)); a = fmaxf(a, dpp_f32<0x121>(a));
    a = fmaxf(a, __shfl_xor(a, 16, 32));
    const float id = a != 0.0f ? 127.f / a : 0.0f;
    dq = h2f(f2h(a / 127.f));
    const unsigned q = (unsigned) (int) rintf(y * id) & 0xffu;
    return q | (dpp_u32<0x101>(q) << 8) | (dpp_u32<0x102>(q) << 16) | (dpp_u32<0x103>(q) << 24);      // row_shl:1..3
}
template <int BC>
__device__ __forceinline__ void mb_dotq8(const unsigned char * wq, const unsigned char * wd, const unsigned char * xop, size_t op_bytes, int u, int nb,
                                         int b0, int B, float (&res)[BC]) {
    float acc[BC];
    const unsigned char * xq[BC], * xd[BC];
#pragma unroll
    for (int j = 0; j < BC; ++j) {
        acc[j] = 0.0f;
        const unsigned char * r = xop + (size_t) (b0 + j < B ? b0 + j : B - 1) * op_bytes;
        xq[j] = r + (size_t) u * mq_ld(nb) * 4; xd[j] = r + (size_t) 32 * mq_ld(nb);
    }
#pragma unroll 2
    for (int c = 0; c < (nb >> 2); ++c) {
        const mq_i4 w = *(const mq_i4 *) (wq + 16 * c);
        const mq_f4 sd = *(const mq_f4 *) (wd + 16 * c);
#pragma unroll
        for (int j = 0; j < BC; ++j) {
            const mq_i4 x = *(const mq_i4 *) (xq[j] + 16 * c);
            const mq_f4 dx = *(const mq_f4 *) (xd[j] + 16 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j] = fmaf(sd[e] * dx[e], (float) __builtin_amdgcn_sdot4(w[e], x[e], 0, false), acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < BC; ++j) {
        float v = acc[j];
        v = v + dpp_f32<0x104>(v);          // row_shl:4  acc[l] + acc[l+4]
        v = v + dpp_f32<0x102>(v);
        res[j] = v + dpp_f32<0x101>(v);
    }
}
// product tasks of a quantised chunk: the slot holds the rows' quants (row stride K + MB_PAD), then - `soff` bytes in - their block scales
template <int BC, typename EPI>
__device__ __forceinline__ void mb_products_q(const unsigned char * slot, int soff, int Rc, int row_base, int N, const float * bias_l, const float * scale_l,
                                              const unsigned char * xop, size_t op_bytes, int K, int B, int wave, int lane, EPI epi) {
    const int nb = K >> 5, stride_q = K + MB_PAD, stride_s = 4 * nb + MB_PAD;
    const int groups = (Rc + 7) >> 3, nsub = (B + BC - 1) / BC, per = (B + nsub - 1) / nsub, ntasks = groups * nsub;
    const int u = lane & 7;
    for (int t = wave; t < ntasks; t += MB_NCW) {
        const int g = t / nsub, sb = t - g * nsub, b0 = sb * per, b1 = min(B, b0 + per);
        const int ri = g * 8 + (lane >> 3), rr = ri < Rc ? ri : Rc - 1;
        const bool has = ri < Rc && row_base + ri < N && u == 0;
        const float bv = bias_l[ri < 64 ? ri : 63], sv = scale_l ? scale_l[ri < 64 ? ri : 63] : 1.0f;
        float res[BC];
        mb_dotq8<BC>(slot + (size_t) rr * stride_q + (size_t) u * nb * 4, slot + soff + (size_t) rr * stride_s, xop, op_bytes, u, nb, b0, B, res);
        epi(res, b0, b1, row_base + ri, has, bv, sv);
    }
}

// -------------------------------------------------------------------------------------------------
// LayerNorm of ONE token row by ONE wave (ops.cpp:3225-3242 semantics as wa_exact.hip: wa_ln_stats - F64 sums in any order, accepted
// when certified order-independent, else redone in index order): the row comes from granules (or the embeddings), stays in registers,
// and leaves as F16 in `dst`.  The in-order fallback walks the registers lane by lane (no LDS copy of the row).
// -------------------------------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ double mb_seq_sum(const float (&xv)[NP], int d, bool squares, float mean) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma nounroll
        for (int j0 = 0; j0 < 64; j0 += 8) {
            double e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv[k]), j0 + j));
                const float a = x - mean;
                e[j] = 64 * k + j0 + j < d ? (squares ? (double) (a * a) : (double) x) : 0.0;      // (+0.0 leaves an IEEE sum unchanged)
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) t += e[j];
        }
    }
    return t;
}
template <int NP, bool Q = false>         // Q: the normalised row leaves as Q8_0 (the next product's operand), not as F16
__device__ __forceinline__ void mb_ln_row(mb_kargs A, mb_ctl & c, gu64 * edge_row /* null: embeddings */, const float * lnp /* LDS: gamma | beta, each d floats in whole KB */, const float * gw_g, const float * gb_g /* global, when non-null */,
                                          int b, int lane_, wa_f16 * dst, float * xres_b, int row_d, int r_d, unsigned code, int token = 0, bool tw = false, int tslot = 0) {
    // (nothing derived from the lane index may live across calls: hoisted out of the layer loop the per-element LDS addresses went to scratch, and every
    //  reload waits for the vector-memory queue)
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const int d = A->d;
    float xv[NP];
    if (edge_row) {
        unsigned v[NP];
        mb_sweep<NP>(edge_row, [&](int k) { const int i = lane + 64 * k; return i < d ? i : -1; }, c, lane, v, code);
#pragma unroll
        for (int k = 0; k < NP; ++k) xv[k] = (lane + 64 * k < d) ? __uint_as_float(v[k]) : 0.0f;
    } else {                    // k_dec_embed: token embedding + positional embedding
        const gcf pe = (gcf) A->pe + (size_t) A->rows[b].pos * d;
        if constexpr (!Q) {
            const gch te = (gch) A->te + (size_t) token * d;
#pragma unroll
            for (int k = 0; k < NP; ++k) { const int i = lane + 64 * k, ic = i < d ? i : d - 1; const float e = h2f(te[ic]) + pe[ic]; xv[k] = i < d ? e : 0.0f; }
        } else {                // k_dec_embed_q: the row dequantised (q * d, ggml-quants.c)
            const int nb = d >> 5;
            const GAS int8_t * tq = (const GAS int8_t *) A->te + (size_t) token * 8 * nb * 4;
            const gcf td = (gcf) A->te_d + (size_t) token * nb;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int i = lane + 64 * k, ic = i < d ? i : d - 1, bb = ic >> 5, el = ic & 31;
                const float e = (float) (int) tq[(((size_t) (el >> 2)) * nb + bb) * 4 + (el & 3)] * td[bb] + pe[ic];
                xv[k] = i < d ? e : 0.0f;
            }
        }
        if (xres_b) {           // the residual values of the rows this workgroup owns in the d-row products
#pragma unroll
            for (int k = 0; k < NP; ++k) { const int i = lane + 64 * k; if (i >= row_d && i < row_d + r_d && i < d) xres_b[i - row_d] = xv[k]; }
        }
    }
    mb_trace(A, tw, tslot);          // (timeline: the row is in)
    double s = 0.0, a = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) { s += (double) xv[k]; a += (double) fabsf(xv[k]); }
    s = wave_sum_d(s); a = wave_sum_d(a);
    float mean, mean_hi;
    if (!wa_sum_bounds(s, a, d, mean, mean_hi, A->rn_d)) {
        bool same = true;
#pragma unroll
        for (int k = 0; k < NP; ++k) if (lane + 64 * k < d) same &= wa_mean_indifferent(xv[k], mean, mean_hi);
        if (!__all(same)) {
            // Third level: an F64 sum of F32 values whose set bits span few enough binary places is EXACT in every order - no addition ever
            // rounds -, so the wave's sum IS the reference's.  Bits of an element: exponent e .. e - 23; of a partial sum of n <= 2048
            // elements: at most emax + 11 .. emin - 23; exact while that fits the 53 bits of a double.  Only wider rows go in order.
            int emax = 0, emin = 255;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int e = (int) ((__float_as_uint(xv[k]) >> 23) & 0xffu);
                if (e != 0) { emax = max(emax, e); emin = min(emin, e); }       // (zeros - and the padding - add nothing; denormals: e = 0 is left out, they only widen the span if present)
                else if ((__float_as_uint(xv[k]) & 0x7fffffu) != 0u) emin = 0;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { emax = max(emax, __shfl_xor(emax, o, WAVE)); emin = min(emin, __shfl_xor(emin, o, WAVE)); }
            if (emax - emin + 24 + 12 <= 53) mean = (float) (s / (double) d);
            else { s = mb_seq_sum<NP>(xv, d, false, 0.0f); mean = (float) (s / (double) d); }
        }
    }
    mb_trace(A, tw, tslot + 1);      // (the mean is certified)
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) if (lane + 64 * k < d) { const float t = xv[k] - mean; s2 += (double) (t * t); }
    s2 = wave_sum_d(s2);
    float variance;
    if (!wa_sum_certain(s2, s2, d, variance, A->rn_d)) { mb_trace(A, tw, tslot + 2); s2 = mb_seq_sum<NP>(xv, d, true, mean); variance = (float) (s2 / (double) d); }
    const float scale = 1.0f / sqrtf(variance + A->eps);
    // gamma | beta: from LDS (wave 7 brought them), or - the first LayerNorm of a launch - from global memory; one uniform branch around the loads
    float gam[NP], bet[NP];
    if (gw_g) {
#pragma unroll
        for (int k = 0; k < NP; ++k) { const int i = lane + 64 * k, ic = i < d ? i : 0; gam[k] = ((gcf) gw_g)[ic]; bet[k] = ((gcf) gb_g)[ic]; }
    } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) { const int i = lane + 64 * k, ic = i < d ? i : 0; gam[k] = lnp[ic]; bet[k] = lnp[((d + 255) & ~255) + ic]; }      // (beta: behind gamma's whole LDS-DMA pieces)
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int i = lane + 64 * k;
        float y = xv[k] - mean;
        y = y * scale;
        y = y * gam[k];
        y = y + bet[k];
        if constexpr (!Q) { if (i < d) dst[i] = f2h(y); }
        else {                  // a half-wave holds one 32-element block (d % 64 == 0: every block of k < d / 64 is whole)
            float dq;
            const unsigned w = mq_quant32(i < d ? y : 0.0f, dq);
            const int nb = d >> 5, blk = (lane >> 5) + 2 * k;
            if (i < d && (lane & 3) == 0) mq_quads((unsigned char *) dst, nb, (lane & 31) >> 2)[blk] = w;
            if (i < d && (lane & 31) == 0) mq_scales((unsigned char *) dst, nb)[blk] = dq;
        }
    }
}

// The product tasks of one weight chunk (Rc rows starting at matrix row `row_base`, in `slot`): task = (group of 64 / LPR weight rows, sub-batch of
// the token rows), dealt over the computing waves.  epi(values, first token row, end token row, matrix row, lane holds results, bias, scale) - called
// by every lane; bias / scale of the chunk's rows sit behind the weights in the slot (wave 7 brought them along).
template <int LPR, int BC, typename EPI>
__device__ __forceinline__ void mb_products(const unsigned char * slot, int stride, int Rc, int row_base, int N, const float * bias_l, const float * scale_l,
                                            const wa_f16 * xs, int ldx, int K, int B, int wave, int lane, EPI epi) {
    constexpr int GP = 64 / LPR;
    const int groups = (Rc + GP - 1) / GP, nsub = (B + BC - 1) / BC, per = (B + nsub - 1) / nsub, ntasks = groups * nsub;
    const int u = lane & (LPR - 1);
    for (int t = wave; t < ntasks; t += MB_NCW) {
        const int g = t / nsub, sb = t - g * nsub, b0 = sb * per, b1 = min(B, b0 + per);
        const int ri = g * GP + lane / LPR;
        const bool valid = ri < Rc && row_base + ri < N;
        const unsigned char * wrow = slot + (size_t) (ri < Rc ? ri : Rc - 1) * stride + u * (LPR == 8 ? 8 : 4);
        const bool has = valid && u == 0;
        const float bv = bias_l[ri < 64 ? ri : 63], sv = scale_l ? scale_l[ri < 64 ? ri : 63] : 1.0f;
        float res[BC];
        if constexpr (LPR == 8) mb_dot8<BC>(wrow, xs + 4 * u, ldx, b0, B, K >> 5, res);
        else                    mb_dot16<BC>(wrow, xs + 2 * u, ldx, b0, B, K >> 5, res);
        epi(res, b0, b1, row_base + ri, has, bv, sv);
    }
}

// two F16 results (matrix rows n, n + 1 in lanes 16 j and 16 j + 8; n even) as one granule
__device__ __forceinline__ unsigned mb_pack_h2(unsigned h) { return (h & 0xffffu) | (dpp_u32<0x108>(h) << 16); }

// -------------------------------------------------------------------------------------------------
// attention scratch in LDS (aliases the FC2-input area, which holds nothing during an attention phase)
// -------------------------------------------------------------------------------------------------
#define MB_ATT_BYTES 34816          /* self-attention unit: 28160; cross-attention unit: 17408 + 2 x 8192 of F64 for its finish */

__device__ __forceinline__ float mb_score(const u32x4 & ka, const u32x4 & kb, const float (&qa)[8], const float (&qb)[8], float scale) {
    const wa_f16 * k8a = (const wa_f16 *) &ka, * k8b = (const wa_f16 *) &kb;
    float v[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        float t = fmaf(h2f(k8a[l]), qa[l], 0.0f);
        t = fmaf(h2f(k8b[l]), qb[l], t);
        t = t + dpp_f32<0x4e>(t);                // quad_perm [2,3,0,1]: s[j] + s[j+2]
        v[l] = t + dpp_f32<0xb1>(t);             // quad_perm [1,0,3,2]: (s0+s2) + (s1+s3)
    }
    const float t0 = v[0] + v[4], t1 = v[1] + v[5], t2 = v[2] + v[6], t3 = v[3] + v[7];
    return ((t0 + t1) + (t2 + t3)) * scale;
}

// the head's 64 outputs from the 32 chain sums + the leftover cells in F64, index order (vec.cpp:221-223), by threads 0..63; published packed.
// Q: the 64 outputs are two Q8_0 blocks of the out-projection's operand: quantised HERE, once (8 quads + the scale = 9 granules per block).
template <bool Q = false>
__device__ __forceinline__ void mb_attn_finish(const float * part, const wa_f16 * vleft /* [nl][64] LDS */, const wa_f16 * pleft /* [nl] */, int nl,
                                               gu64 * edge_row, int h, unsigned seq, int tid) {
    if (tid < 64) {
        float s32[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) s32[r] = part[r * 64 + tid];
        double sumf = (double) wa_tree32(s32);
        for (int cc = 0; cc < nl; ++cc) sumf += (double) (h2f(vleft[cc * 64 + tid]) * h2f(pleft[cc]));
        if constexpr (Q) {
            float dq;
            const unsigned w = mq_quant32((float) sumf, dq);
            gu64 * eb = edge_row + (size_t) (2 * h + (tid >> 5)) * 9;
            if ((tid & 3) == 0) gr_store(eb + ((tid & 31) >> 2), seq, w);
            if ((tid & 31) == 0) gr_store(eb + 8, seq, __float_as_uint(dq));
        } else {
            const unsigned hv = (unsigned) f2h((float) sumf);
            const unsigned hi = dpp_u32<0x101>(hv);          // row_shl:1: lane i reads lane i + 1
            if ((tid & 1) == 0) gr_store(edge_row + ((h * 64 + tid) >> 1), seq, (hv & 0xffffu) | (hi << 16));
        }
    }
}

// -------------------------------------------------------------------------------------------------
// unit: self-attention of (token row b, head h) (whisper.cpp:2636-2651; arithmetic of k_attn_exact<1>).  Keys and values of earlier
// tokens come from the row's own cells in HBM; the cells written by THIS launch - the row's own and those of every other row that shares
// its cache (earlier tokens of a small batch; other beams, which the mask hides) - arrive as granules, like the query.
// -------------------------------------------------------------------------------------------------
template <bool Q = false>
__device__ __forceinline__ void mb_unit_self(mb_kargs A, mb_ctl & c, unsigned char * area, int l, int b, int h, int tid, bool tw = false) {
#define MB_TS(k) mb_trace(A, tw, 4096 + l * 16 + (k))
    MB_TS(0);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float  * part = (float *) area;                                  // [32][64]
    float  * sc   = (float *) (area + 8192);                         // [MAXKV]
    wa_f16 * p16  = (wa_f16 *) (area + 8192 + WA_ROWS_MAXKV * 4);    // [MAXKV]
    float  * gs   = (float *) (area + 8192 + WA_ROWS_MAXKV * 6);     // [MAXKV / 8]
    wa_f16 * vleft = (wa_f16 *) (area + 8192 + WA_ROWS_MAXKV * 6 + WA_ROWS_MAXKV / 2);      // [32][64]
    unsigned char * sm = area + 8192 + WA_ROWS_MAXKV * 6 + WA_ROWS_MAXKV / 2 + 4096;
    wa_f16 * knew = (wa_f16 *) sm;                                   // [8][64]
    wa_f16 * vnew = (wa_f16 *) (sm + 1024);                          // [8][64]
    wa_f16 * qs   = (wa_f16 *) (sm + 2048);                          // [64]
    float  * red  = (float *) (sm + 2048 + 128);                     // [8]
    double * redd = (double *) (sm + 2048 + 192);                    // [8]
    float  * s_inv = (float *) (sm + 2048 + 256);
    int    * ncl  = (int *) (sm + 2048 + 288);                       // [8] cell of row j's new key / value in THIS row's cache, or -1
    const int d = A->d, B = A->B, n_kv = A->rows[b].n_kv;
    const gch kp = (gch) A->rows[b].kv_k + (size_t) l * A->kv_layer_stride + h * 64;
    const gch vp = (gch) A->rows[b].kv_v + (size_t) l * A->kv_layer_stride + h * 64;
    const GAS int8_t * mrow = (const GAS int8_t *) A->rows[b].mask;
    // requested before the query is waited for: the keys of the first 512 cells (4 per 4-lane group) and the values of the first four P V steps
    const int a_ = tid & 3, kslot_ = tid >> 2;
    u32x4 ka0[4], kb0[4];
    wa_f16 vv0[4][4];
    {
        const int nst = (n_kv & ~31) >> 5;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int cc = p * (MB_THREADS / 4) + kslot_; cc = cc < n_kv ? cc : n_kv - 1;
            ka0[p] = *(const GAS u32x4 *) (kp + (size_t) cc * d + 8 * a_); kb0[p] = *(const GAS u32x4 *) (kp + (size_t) cc * d + 32 + 8 * a_);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sidx = q < nst ? q : (nst > 0 ? nst - 1 : 0);
                vv0[q][i] = *(const GAS wa_f16 *) (vp + (size_t) (sidx * 32 + wave * 4 + i) * d + lane);
            }
    }
    if (wave < B) {         // wave j: k | v of row j (two runs of 32 packed granules), and the query when j is this row
        const bool same = A->rows[wave].kv_k == A->rows[b].kv_k;
        if (same) {
            unsigned v[2];
            gu64 * e = mb_edge(A, l, E_QKV) + (size_t) wave * A->row_gr;
            mb_sweep<2>(e, [&](int k) {
                if (k == 0) return (lane < 32 ? (d >> 1) : d) + h * 32 + (lane & 31);
                return wave == b && lane < 32 ? h * 32 + lane : -1; }, c, lane, v, 1000u + l);
            if (lane < 32) ((unsigned *) (knew + wave * 64))[lane] = v[0]; else ((unsigned *) (vnew + wave * 64))[lane - 32] = v[0];
            if (wave == b && lane < 32) ((unsigned *) qs)[lane] = v[1];
        }
        if (lane == 0) ncl[wave] = same && A->rows[wave].kv_head < n_kv ? A->rows[wave].kv_head : -1;
    } else if (lane == 0) ncl[wave] = -1;
    MB_TS(1);
    mb_barrier();
    MB_TS(2);
    int nc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) nc[j] = __builtin_amdgcn_readfirstlane(ncl[j]);
    auto new_of = [&](int cc) { int r = -1;
#pragma unroll
        for (int j = 7; j >= 0; --j) if (cc == nc[j]) r = j;
        return r; };
    // ---- scores: 4 lanes per key ----
    float lmax = -INFINITY;
    {
        const int a = tid & 3, kslot = tid >> 2;
        float qa[8], qb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { qa[i] = h2f(qs[8 * a + i]); qb[i] = h2f(qs[32 + 8 * a + i]); }
        for (int c0 = 0; c0 < n_kv; c0 += (MB_THREADS / 4) * 4) {
            u32x4 ka[4], kb[4];
            int8_t mk[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                int cc = c0 + p * (MB_THREADS / 4) + kslot; cc = cc < n_kv ? cc : n_kv - 1;
                if (c0 == 0) { ka[p] = ka0[p]; kb[p] = kb0[p]; }
                else { ka[p] = *(const GAS u32x4 *) (kp + (size_t) cc * d + 8 * a); kb[p] = *(const GAS u32x4 *) (kp + (size_t) cc * d + 32 + 8 * a); }
            }
            if (mrow) {         // (ONE uniform branch around the four loads: as a select per element every load was issued and waited for on its own - also without a mask)
#pragma unroll
                for (int p = 0; p < 4; ++p) { int cc = c0 + p * (MB_THREADS / 4) + kslot; cc = cc < n_kv ? cc : n_kv - 1; mk[p] = mrow[cc]; }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (c0 + p * (MB_THREADS / 4) >= n_kv) break;       // (uniform: a block of 128 cells behind the row's last one - with 110 cells three of the four)
                const int cc = c0 + p * (MB_THREADS / 4) + kslot;
                float r = mb_score(ka[p], kb[p], qa, qb, 1.0f);
                if (cc < n_kv && new_of(cc) < 0) {       // (a cell written by this launch: below, from its granules)
                    if (mk[p]) r = -INFINITY;
                    if (a == 0) sc[cc] = r;
                    lmax = fmaxf(lmax, r);
                }
            }
        }
        if (tid < 32) {     // the cells of this launch: 4 lanes per row j
            const int j = tid >> 2, cc = nc[0] * (j == 0) + nc[1] * (j == 1) + nc[2] * (j == 2) + nc[3] * (j == 3) + nc[4] * (j == 4) + nc[5] * (j == 5) + nc[6] * (j == 6) + nc[7] * (j == 7);
            const bool act = cc >= 0 && new_of(cc) == j;
            const u32x4 kna = *(const u32x4 *) (knew + j * 64 + 8 * a), knb = *(const u32x4 *) (knew + j * 64 + 32 + 8 * a);
            float r = mb_score(kna, knb, qa, qb, 1.0f);
            if (act) {
                if (mrow && mrow[cc]) r = -INFINITY;
                if (a == 0) sc[cc] = r;
                lmax = fmaxf(lmax, r);
            }
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    MB_TS(3);
    mb_barrier();
    MB_TS(4);
    float mx = red[0];
#pragma unroll
    for (int k = 1; k < MB_NW; ++k) mx = fmaxf(mx, red[k]);
    // ---- soft_max exactly as ops.cpp:4792-4818 + vec.cpp:257-308 (k_attn_exact): a thread per cell - exp, the group sums by the reference's 8-lane tree over
    //      DPP (as the cross-attention unit), F64 partial sums; ONE barrier, then every thread forms the same total from the eight wave sums ----
    const int n8 = n_kv & ~7, ng = n8 >> 3;
    float inv;
    {
        double ps = 0.0;
        for (int c0 = 0; c0 < n_kv; c0 += MB_THREADS) {       // (uniform trip count: every lane takes part in the DPP exchanges)
            const int cc = c0 + tid;
            const bool in = cc < n_kv;
            const float e = !in ? 0.0f : cc < n8 ? wa_expf(sc[cc] - mx) : wa_expf_libm(sc[cc] - mx);
            if (in) sc[cc] = e;
            float t = e + dpp_f32<0x104>(e);        // lanes r = 0..3 of the group: e[r] + e[r+4]
            t = t + dpp_f32<0x102>(t);
            t = t + dpp_f32<0x101>(t);              // r = 0: ((e0+e4)+(e2+e6)) + ((e1+e5)+(e3+e7)), ops.cpp's tree
            if (in) {
                if (cc < n8) { if ((tid & 7) == 0) { gs[cc >> 3] = t; ps += (double) t; } }
                else ps += (double) e;              // (the n % 8 tail cells: any order, the total is certified below)
            }
        }
        ps = wave_sum_d(ps);
        if (lane == 0) redd[wave] = ps;
        mb_barrier();
        const double sum = ((redd[0] + redd[1]) + (redd[2] + redd[3])) + ((redd[4] + redd[5]) + (redd[6] + redd[7]));
        // (the reference adds the ng + (n % 8) addends one after the other in F64: error <= (ng + 7) u S; this sum is a tree of depth <= 16 over the SAME addends:
        //  error <= 16 u S; together (ng + 8 + 16) u S - not twice the reference's bound, which sent twice as many soft-maxes back to the launch sequence)
        const double delta = (double) (ng + 8 + 16) * 0x1p-53 * sum * 1.000001;
        const float ilo = (float) (1.0 / (sum + delta)), ihi = (float) (1.0 / (sum - delta));
        inv = ilo;
        if (ilo != ihi) {       // (~1e-9 per soft-max; the same for every thread) the reference's order, by one thread
            if (tid == 0) {
                double so = 0.0;
                for (int g = 0; g < ng; ++g) so += (double) gs[g];
                for (int cc = n8; cc < n_kv; ++cc) so += (double) sc[cc];
                *s_inv = (float) (1.0 / so);
            }
            mb_barrier();
            inv = *s_inv;
        }
    }
    MB_TS(5);
    for (int cc = tid; cc < n_kv; cc += MB_THREADS) p16[cc] = f2h(sc[cc] * inv);
    const int np = n_kv & ~31, nsteps = np >> 5, nl = n_kv - np;
    if (tid < 256) {        // the leftover cells' V rows -> LDS
        const int row = tid >> 3, cc = np + row;
        if (row < nl) {
            const int j = new_of(cc);
            *(u32x4 *) (vleft + (size_t) tid * 8) = j >= 0 ? *(const u32x4 *) (vnew + j * 64 + (tid & 7) * 8) : *(const GAS u32x4 *) (vp + (size_t) cc * d + (tid & 7) * 8);
        }
    }
    mb_barrier();
    MB_TS(6);
    // ---- P V: chains r = cell mod 32 (4 per wave), lane = d_head index ----
    {
        float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        const int r0 = wave * 4;
        for (int s0 = 0; s0 < nsteps; s0 += 4) {
            wa_f16 vv[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int sidx = s0 + q < nsteps ? s0 + q : nsteps - 1, cc = sidx * 32 + r0 + i;
                    if (s0 == 0) vv[q][i] = vv0[q][i]; else vv[q][i] = *(const GAS wa_f16 *) (vp + (size_t) cc * d + lane);
                    const int j = new_of(cc);       // (wave-uniform)
                    if (j >= 0) vv[q][i] = vnew[j * 64 + lane];
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (s0 + q < nsteps) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaf(h2f(vv[q][i]), h2f(p16[(s0 + q) * 32 + r0 + i]), acc[i]);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) part[(r0 + i) * 64 + lane] = acc[i];
    }
    MB_TS(7);
    mb_barrier();
    mb_attn_finish<Q>(part, vleft, p16 + np, nl, mb_edge(A, l, E_AO) + (size_t) b * A->row_gr, h, c.seq, tid);
    MB_TS(8);
    mb_barrier();
    MB_TS(9);
#undef MB_TS
}

// -------------------------------------------------------------------------------------------------
// unit: one of P parts (4 quarters, or 2 halves when the quarters of all rows would not fit the grid in one round) of the cross-attention of
// (token row b, head h) over the row's encoder K / V (whisper.cpp:2683-2758); arithmetic and split of wa_mega.hip: mg_role_cross - part w owns
// the cells c with (c mod 32) in [NCH w, NCH w + NCH), NCH = 32 / P: whole soft-max groups of 8 cells and whole P V chains; the parts
// exchange maxima, F64 partial sums and chain sums through the (layer, row, head) granule area.  Local index o = NCH s + r  <->  cell 32 s + NCH w + r.
// -------------------------------------------------------------------------------------------------
#define MB_CSTEPS 48
#define MB_CGR_MAX 0
#define MB_CGR_SUM 8
#define MB_CGR_PART 64               // chain sums of the parts 1 .. P - 1: [P - 1][NCH][64]; behind them their leftover probabilities [P - 1][NCH]
#define MB_CGR_INORD 1664            // (quarters, rare) the parts' group sums for the in-order total: [4][64] = 48 group sums | 8 tail cells | 8 unused

template <int P> struct mb_cross_regs { u32x4 ka[12 / P], kb[12 / P]; unsigned short vv[4 / P][MB_CSTEPS]; };
// own keys and own chain elements: unconditional, clamped loads - all in flight together
template <int P>
__device__ __forceinline__ void mb_cross_load(mb_kargs A, int l, int b, int h, int w, int tid, mb_cross_regs<P> & R) {
    constexpr int NCH = 32 / P, CPW = 4 / P;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = A->T, tpad = A->cross_tpad, a = tid & 3, ks = tid >> 2, nsteps = (T & ~31) >> 5;
    const gch kp = (gch) A->rows[b].cross_k + (size_t) l * A->cross_layer_stride + (size_t) h * tpad * 64;
    const gch vp = (gch) A->rows[b].cross_v + (size_t) l * A->cross_layer_stride + (size_t) h * tpad * 64;
#pragma unroll
    for (int p = 0; p < 12 / P; ++p) {
        const int o = p * 128 + ks, c0 = 32 * (o / NCH) + NCH * w + (o % NCH), cc = c0 < T ? c0 : T - 1;
        R.ka[p] = *(const GAS u32x4 *) (kp + (size_t) cc * 64 + 8 * a); R.kb[p] = *(const GAS u32x4 *) (kp + (size_t) cc * 64 + 32 + 8 * a);
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int s = 0; s < MB_CSTEPS; ++s) {
            const int sc_ = s < nsteps ? s : (nsteps > 0 ? nsteps - 1 : 0);
            R.vv[c][s] = *(const GAS unsigned short *) (vp + (size_t) (32 * sc_ + NCH * w + CPW * wave + c) * 64 + lane);
        }
}

template <bool Q, int P>
__device__ __forceinline__ void mb_unit_cross(mb_kargs A, mb_ctl & c, unsigned char * area, int l, int b, int h, int w, int tid, bool tw, const mb_cross_regs<P> & R) {
    constexpr int NCH = 32 / P, CPW = 4 / P, NK = NCH * MB_CSTEPS;          // chains of this part, chains per wave, local cell slots
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned seq = c.seq;
#define MB_TC(k) mb_trace(A, tw, l * 32 + 16 + (k))
    MB_TC(0);
    // LDS: part [32][64] f32 | vleft [32][64] f16 | sc [768] f32 | p16 [16][48] f16 | pleft [32] f16 | qs [64] f16 | redd [8] f64 | red [8] + bc [4] f32 |
    //      two F64 areas [16][64] for the finish
    float  * part  = (float *) area;
    wa_f16 * vleft = (wa_f16 *) (area + 8192);
    float  * sc    = (float *) (area + 12288);
    wa_f16 * p16   = (wa_f16 *) (area + 15360);
    wa_f16 * pleft = (wa_f16 *) (area + 16896);
    wa_f16 * qs    = (wa_f16 *) (area + 16960);
    double * redd  = (double *) (area + 17088);
    float  * red   = (float *) (area + 17152);
    float  * bc    = red + 8;
    double * dbl0  = (double *) (area + 17408), * dbl1 = (double *) (area + 17408 + 8192);
    const int T = A->T, tpad = A->cross_tpad, H = A->n_head;
    const float kq_scale = A->kq_scale;
    const int a = tid & 3, ks = tid >> 2;
    const int np = T & ~31, nsteps = np >> 5, nl = T - np, n8 = T & ~7, ng = n8 >> 3;
    gu64 * X = (gu64 *) A->cross_gr + (((size_t) l * A->B + b) * H + h) * WA_ROWS_CGR;
    const gch vp = (gch) A->rows[b].cross_v + (size_t) l * A->cross_layer_stride + (size_t) h * tpad * 64;
    if (w == 0 && tid < 256) {
        const int row = tid >> 3;
        if (row < nl) *(u32x4 *) (vleft + (size_t) tid * 8) = *(const GAS u32x4 *) (vp + (size_t) (np + row) * 64 + (tid & 7) * 8);
    }
    if (wave == 0) {
        unsigned v[1];
        mb_sweep<1>(mb_edge(A, l, E_QC) + (size_t) b * A->row_gr, [&](int) { return lane < 32 ? h * 32 + lane : -1; }, c, lane, v, 2000u + l);
        if (lane < 32) ((unsigned *) qs)[lane] = v[0];
        MB_TC(1);
    }
    mb_barrier();
    MB_TC(2);
    // ---- scores of the own cells ----
    float lmax = -INFINITY;
    {
        float qa[8], qb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { qa[i] = h2f(qs[8 * a + i]); qb[i] = h2f(qs[32 + 8 * a + i]); }
#pragma unroll
        for (int p = 0; p < 12 / P; ++p) {
            const int o = p * 128 + ks, cc = 32 * (o / NCH) + NCH * w + (o % NCH);
            const float r = mb_score(R.ka[p], R.kb[p], qa, qb, kq_scale);
            if (cc < T) { if (a == 0) sc[o] = r; lmax = fmaxf(lmax, r); }
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    MB_TC(3);
    mb_barrier();
    if (wave == 0) {        // (1) maxima of the parts
        float m = red[0];
#pragma unroll
        for (int k = 1; k < MB_NW; ++k) m = fmaxf(m, red[k]);
        if (lane == 0) gr_store(X + MB_CGR_MAX + w, seq, __float_as_uint(m));
        unsigned v[1];
        mb_sweep<1>(X + MB_CGR_MAX, [&](int) { return lane < P ? lane : -1; }, c, lane, v, 2100u + l);
        float g = lane < P ? __uint_as_float(v[0]) : -INFINITY;
        g = fmaxf(g, dpp_f32<0x4e>(g)); g = fmaxf(g, dpp_f32<0xb1>(g));      // max over lanes 0..3
        if (lane == 0) bc[0] = g;
    }
    mb_barrier();
    MB_TC(4);
    const float mx = bc[0];
    // ---- exp, group sums (8-lane tree = ops.cpp's), F64 partial sum: a thread per own cell (ops.cpp:4792-4818, vec.cpp:257-308) ----
    {
        double ps = 0.0;
#pragma unroll
        for (int k = 0; k < (NK + MB_THREADS - 1) / MB_THREADS; ++k) {
            const int o = tid + MB_THREADS * k;
            const bool in = o < NK;
            const int oc = in ? o : 0, s_ = oc / NCH, r_ = oc % NCH, g = 4 * s_ + ((NCH * w + r_) >> 3), cc = 32 * s_ + NCH * w + r_;
            const float e = !in ? 0.0f : cc < n8 ? wa_expf(sc[oc] - mx) : (cc < T ? wa_expf_libm(sc[oc] - mx) : 0.0f);
            if (in) sc[oc] = e;
            float t = e + dpp_f32<0x104>(e);        // lanes r = 0..3 of the group: e[r] + e[r+4]
            t = t + dpp_f32<0x102>(t);
            t = t + dpp_f32<0x101>(t);              // r = 0: the group sum, ops.cpp's tree
            if (in) ps += g < ng ? ((tid & 7) == 0 ? (double) t : 0.0) : (double) e;       // (the n % 8 tail cells: any order, the total is certified below)
        }
        ps = wave_sum_d(ps);
        if (lane == 0) redd[wave] = ps;
    }
    MB_TC(5);
    mb_barrier();
    if (wave == 0) {        // (2) partial sums -> total, certified
        const double ps = ((redd[0] + redd[1]) + (redd[2] + redd[3])) + ((redd[4] + redd[5]) + (redd[6] + redd[7]));
        const u64 pb = (u64) __double_as_longlong(ps);
        if (lane == 0) { gr_store(X + MB_CGR_SUM + 2 * w, seq, (unsigned) pb); gr_store(X + MB_CGR_SUM + 2 * w + 1, seq, (unsigned) (pb >> 32)); }
        unsigned v[1];
        mb_sweep<1>(X + MB_CGR_SUM, [&](int) { return lane < 2 * P ? lane : -1; }, c, lane, v, 2200u + l);
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const unsigned lo = __builtin_amdgcn_readlane(v[0], 2 * k), hi = __builtin_amdgcn_readlane(v[0], 2 * k + 1);
            tot += __longlong_as_double((long long) (((u64) hi << 32) | lo));
        }
        const double delta = (double) (ng + 8 + 16) * 0x1p-53 * tot * 1.000001;      // (as in the self-attention unit: reference (ng + 7) u S + this tree's 16 u S)
        const float ilo = (float) (1.0 / (tot + delta)), ihi = (float) (1.0 / (tot - delta));
        float inv_ = ilo;
        if (ilo != ihi || A->force_inorder) {       // the order could matter (~3e-5 per soft-max; the same decision in every part: they hold the same total)
            if constexpr (P == 4) {
                // The reference's order (vec.cpp:278-305: the group sums one after the other in F64, then the n % 8 tail cells), inside the launch: every quarter
                // publishes its 47-48 group sums (the tree of ops.cpp over its exponentials, as above) and the tail cells it owns, gathers all four quarters'
                // and lets one lane add them in index order - group g = 4 s + w is step s of quarter w.  (Until late in round 3 the row was marked for the launch
                // sequence instead: a 1-4 ms stall of a whole lock-step group, a few times per job.)
                float * io = part;                  // [4][64] scratch (the chain sums come later)
                gu64 * XI = X + MB_CGR_INORD;
                {
                    const int s_ = lane;            // slots 0..47: step s of this quarter; 48..55: the cells of the partial group, if this quarter owns it; the rest: 0
                    float pv = 0.0f;
                    if (s_ < MB_CSTEPS) {
                        const float * e8 = sc + 8 * s_;
                        if (4 * s_ + w < ng) pv = ((e8[0] + e8[4]) + (e8[2] + e8[6])) + ((e8[1] + e8[5]) + (e8[3] + e8[7]));
                    } else if (s_ < MB_CSTEPS + 8) {
                        const int j = s_ - MB_CSTEPS, st = ng >> 2;         // group ng = step ng / 4 of quarter ng % 4: the cells n8 + j
                        if ((ng & 3) == w && st < MB_CSTEPS && n8 + j < T) pv = sc[8 * st + j];
                    }
                    gr_store(XI + 64 * w + s_, seq, __float_as_uint(pv));
                }
                unsigned vi[4];
                mb_sweep<4>(XI, [&](int k) { return 64 * k + lane; }, c, lane, vi, 2250u + l);
#pragma unroll
                for (int k = 0; k < 4; ++k) io[64 * k + lane] = __uint_as_float(vi[k]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (this wave's own LDS writes, read back below)
                double so = 0.0;
                for (int s_ = 0; s_ < MB_CSTEPS; ++s_)
                    for (int q = 0; q < 4; ++q) if (4 * s_ + q < ng) so += (double) io[64 * q + s_];
                for (int j = 0; n8 + j < T; ++j) so += (double) io[64 * (ng & 3) + MB_CSTEPS + j];
                inv_ = (float) (1.0 / so);
            } else {
                if (lane == 0) ((GAS unsigned *) A->row_status)[b] = (unsigned) WA_MEGA_REDO;      // (halves: not instantiated - this row goes to the launch sequence, the pass goes on)
            }
        }
        if (lane == 0) bc[1] = inv_;
    }
    mb_barrier();
    MB_TC(6);
    const float inv = bc[1];
    gu64 * XP = X + MB_CGR_PART;
#pragma unroll
    for (int k = 0; k < (NK + MB_THREADS - 1) / MB_THREADS; ++k) {
        const int o = tid + MB_THREADS * k;
        if (o < NK) {
            const int s_ = o / NCH, r_ = o % NCH, cc = 32 * s_ + NCH * w + r_;
            if (cc < T) {
                const wa_f16 ph = f2h(sc[o] * inv);
                p16[r_ * MB_CSTEPS + s_] = ph;         // by chain: a P V wave reads a chain's 48 probabilities as 6 x 16 bytes
                if (cc >= np) { if (w == 0) pleft[cc - np] = ph; else gr_store(XP + (P - 1) * NCH * 64 + (w - 1) * NCH + r_, seq, (unsigned) ph); }
            }
        }
    }
    mb_barrier();
    MB_TC(7);
    // ---- P V: wave = own chains (cells 32 s + NCH w + CPW wave + c), lane = d_head index ----
#pragma unroll
    for (int cw = 0; cw < CPW; ++cw) {
        float acc = 0.0f;
        half8 pw[MB_CSTEPS / 8];
#pragma unroll
        for (int k = 0; k < MB_CSTEPS / 8; ++k) pw[k] = *(const half8 *) (p16 + (CPW * wave + cw) * MB_CSTEPS + 8 * k);
#pragma unroll
        for (int s_ = 0; s_ < MB_CSTEPS; ++s_) if (s_ < nsteps) acc = fmaf(h2f(R.vv[cw][s_]), (float) pw[s_ >> 3][s_ & 7], acc);
        if (w == 0) part[(CPW * wave + cw) * 64 + lane] = acc;
        else gr_store(XP + ((w - 1) * NCH + CPW * wave + cw) * 64 + lane, seq, __float_as_uint(acc));
    }
    MB_TC(8);
    if (w == 0) {           // (3) gather the other parts' chain sums and leftover probabilities, finish the head
        constexpr int NG_ = (P - 1) * NCH * 64;          // 1536 (quarters) or 1024 (halves) chain sums
        unsigned v[NG_ / MB_THREADS + 1];
        mb_sweep<NG_ / MB_THREADS + 1>(XP, [&](int k) {
            if (k < NG_ / MB_THREADS) return tid + MB_THREADS * k;
            return tid < (P - 1) * NCH && NCH + tid < nl ? NG_ + tid : -1; }, c, lane, v, 2300u + l);
#pragma unroll
        for (int k = 0; k < NG_ / MB_THREADS; ++k) part[NCH * 64 + tid + MB_THREADS * k] = __uint_as_float(v[k]);      // (part ww's chain r = global chain NCH ww + r)
        if (tid < (P - 1) * NCH && NCH + tid < nl) pleft[NCH + tid] = (wa_f16) v[NG_ / MB_THREADS];
        mb_barrier();
        MB_TC(9);
        // finish: the leftover cells (vec.cpp:221-223: F64, index order) spread over waves 1..7 - five cells each, for the 64 outputs, every product
        // ONE v_fma_mix_f32 (F16 x F16 is exact in F32; with a -0.0 addend it is the multiplication's float), parked in LDS as F64 - while wave 0 runs
        // the tree; what stays in series is wave 0's 32 F64 additions (wa_mega.hip: mg_attn_finish)
        int nlo = nl;
        asm volatile("" : "+s"(nlo));
        if (tid >= 64) {
            const int wv = tid >> 6, o = tid & 63;
            float nzero = -0.0f;
            asm volatile("" : "+v"(nzero));
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int cc = 5 * (wv - 1) + i;
                if (cc < 32) {
                    const float pr = fmaf(h2f(vleft[cc * 64 + o]), h2f(pleft[cc]), nzero);
                    (cc < 16 ? dbl0 : dbl1)[(cc & 15) * 64 + o] = (double) (cc < nlo ? pr : -0.0f);
                }
            }
        }
        double sumf = 0.0;
        if (tid < 64) {
            float s32[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) s32[r] = part[r * 64 + tid];
            sumf = (double) wa_tree32(s32);
        }
        mb_barrier();
        if (tid < 64) {
            double dv[32];
#pragma unroll
            for (int cc = 0; cc < 32; ++cc) dv[cc] = (cc < 16 ? dbl0 : dbl1)[(cc & 15) * 64 + tid];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cc = 0; cc < 32; ++cc) sumf += dv[cc];
            gu64 * edge_row = mb_edge(A, l, E_AO2) + (size_t) b * A->row_gr;
            if constexpr (Q) {
                float dq;
                const unsigned wq_ = mq_quant32((float) sumf, dq);
                gu64 * eb = edge_row + (size_t) (2 * h + (tid >> 5)) * 9;
                if ((tid & 3) == 0) gr_store(eb + ((tid & 31) >> 2), seq, wq_);
                if ((tid & 31) == 0) gr_store(eb + 8, seq, __float_as_uint(dq));
            } else {
                const unsigned hv = (unsigned) f2h((float) sumf);
                const unsigned hi = dpp_u32<0x101>(hv);
                if ((tid & 1) == 0) gr_store(edge_row + ((h * 64 + tid) >> 1), seq, (hv & 0xffffu) | (hi << 16));
            }
        }
        MB_TC(10);
    }
    mb_barrier();
}

// -------------------------------------------------------------------------------------------------
// final LayerNorm + logits = token_embedding . x for every token row (whisper.cpp:2820-2835): every workgroup, every wave; the embedding
// rows stream from HBM into registers (two buffers of 24 steps: the loads of the next piece fly during the current one)
// -------------------------------------------------------------------------------------------------
// A workgroup's share of the logits: a CONTIGUOUS run of 8-row groups of the token embedding (so that its results leave as whole 256-byte
// stores - also straight into pinned host memory, WHISPER_AMD_ROWS_HOST_OUT); wave w takes the groups g0 + w + 8 j.  Local row li = 8 (w + 8 j) + r.
__device__ __forceinline__ int mb_share(int n_vocab, int * g1) {
    const int NG = (n_vocab + 7) >> 3, gpw = (NG + (int) gridDim.x - 1) / (int) gridDim.x, g0 = (int) blockIdx.x * gpw;
    *g1 = g0 + gpw < NG ? g0 + gpw : NG;
    return g0;
}
// the staged rows (lg [BT][256], local rows < 256) -> the logits rows, 64 consecutive floats per store
__device__ __forceinline__ void mb_logits_out(mb_kargs A, const float * lg, int B, int tid) {
    int g1; const int g0 = mb_share(A->n_vocab, &g1), n_vocab = A->n_vocab;
    GAS float * logits = (GAS float *) A->logits;
    const int n_loc = (g1 - g0) * 8 < 256 ? (g1 - g0) * 8 : 256;
    for (int idx = tid; idx < B * 256; idx += MB_THREADS) {
        const int m = idx >> 8, t = idx & 255, row = g0 * 8 + t;
        if (t < n_loc && row < n_vocab) logits[(size_t) m * n_vocab + row] = lg[m * 256 + t];
    }
}

// the first piece of a wave's first embedding rows, asked for BEFORE the final LayerNorm waits for its row (piece 0 of mb_logits / mb_logits_q)
__device__ __forceinline__ void mb_te_first(mb_kargs A, bool quant, unsigned (&buf)[48], int lane, int wave) {
    const int d = A->d, ns = d >> 5, n_vocab = A->n_vocab, u = lane & 7;
    int g1_; const int g0_ = mb_share(n_vocab, &g1_);
    const int row = (g0_ + wave) * 8 + (lane >> 3), rc = row < n_vocab ? row : 0;
    if (!quant) {
        const gch wrow = (gch) A->te + (size_t) rc * d + 4 * u;
#pragma unroll
        for (int k = 0; k < 24; ++k) { const u32x2 t = *(const GAS u32x2 *) (wrow + (size_t) (k < ns ? k : ns - 1) * 32); buf[2 * k] = t.x; buf[2 * k + 1] = t.y; }
    } else {
        const GAS unsigned * wl = (const GAS unsigned *) A->te + ((size_t) rc * 8 + u) * ns;
        const GAS unsigned * dl = (const GAS unsigned *) A->te_d + (size_t) rc * ns;
#pragma unroll
        for (int k4 = 0; k4 < 6; ++k4) {         // 16 bytes = four blocks per load (a row's quads of a lane and its scales are contiguous; d / 32 is a multiple of 4):
            const int bc = 4 * k4 < ns ? 4 * k4 : ns - 4;         // as single dwords a piece was 48 load instructions of 64 scattered 4-byte accesses each
            const u32x4 q = *(const GAS u32x4 *) (wl + bc), sd = *(const GAS u32x4 *) (dl + bc);
            buf[4 * k4] = q.x; buf[4 * k4 + 1] = q.y; buf[4 * k4 + 2] = q.z; buf[4 * k4 + 3] = q.w;
            buf[24 + 4 * k4] = sd.x; buf[24 + 4 * k4 + 1] = sd.y; buf[24 + 4 * k4 + 2] = sd.z; buf[24 + 4 * k4 + 3] = sd.w;
        }
    }
}

template <int BT>
__device__ __forceinline__ void mb_logits(mb_kargs A, const wa_f16 * xs, int B /* rows of xs = logits rows */, int lane, int wave, unsigned (&pfa)[48], float * lg /* LDS [BT][256]: the staged share */) {
    const int d = A->d, ns = d >> 5, nbat = (ns + 23) / 24, n_vocab = A->n_vocab;
    const int u = lane & 7;
    GAS float * logits = (GAS float *) A->logits;
    unsigned pfb[48];
    int g1; const int g0 = mb_share(n_vocab, &g1);
    auto grp = [&](int j) { return g0 + wave + MB_NW * j; };
    int n_items = 0;
    for (int j = 0; grp(j) < g1; ++j) n_items += nbat;
    auto load = [&](int it, unsigned (&buf)[48]) {
        const int j = it / nbat, bt = it - j * nbat, row = grp(j) * 8 + (lane >> 3);
        const gch wrow = (gch) A->te + (size_t) (row < n_vocab ? row : 0) * d + 4 * u;
#pragma unroll
        for (int k = 0; k < 24; ++k) {       // (unconditional: every load of a piece in flight together)
            const int s = bt * 24 + k;
            const u32x2 t = *(const GAS u32x2 *) (wrow + (size_t) (s < ns ? s : ns - 1) * 32); buf[2 * k] = t.x; buf[2 * k + 1] = t.y;
        }
    };
    float acc[BT][4];
    auto one = [&](int it, unsigned (&buf)[48]) {
        const int j = it / nbat, bt = it - j * nbat;
        if (bt == 0) {
#pragma unroll
            for (int m = 0; m < BT; ++m) acc[m][0] = acc[m][1] = acc[m][2] = acc[m][3] = 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int s = bt * 24 + k;
            if (s < ns) {
                u32x2 t; t.x = buf[2 * k]; t.y = buf[2 * k + 1];
                const half4v w4 = __builtin_bit_cast(half4v, t);
#pragma unroll
                for (int m = 0; m < BT; ++m) {
                    const half4v x4 = *(const half4v *) (xs + (size_t) (m < B ? m : B - 1) * d + s * 32 + 4 * u);
                    acc[m][0] = fmaf((float) w4[0], (float) x4[0], acc[m][0]);
                    acc[m][1] = fmaf((float) w4[1], (float) x4[1], acc[m][1]);
                    acc[m][2] = fmaf((float) w4[2], (float) x4[2], acc[m][2]);
                    acc[m][3] = fmaf((float) w4[3], (float) x4[3], acc[m][3]);
                }
            }
        }
        if (bt == nbat - 1) {
            const int row = grp(j) * 8 + (lane >> 3);
#pragma unroll
            for (int m = 0; m < BT; ++m) {
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[m][i];
                    v = v + dpp_f32<0x104>(v);
                    v = v + dpp_f32<0x102>(v);
                    t[i] = v + dpp_f32<0x101>(v);
                }
                const float r = (t[0] + t[1]) + (t[2] + t[3]);
                if (u == 0 && j < 4) lg[m * 256 + (8 * j + wave) * 8 + (lane >> 3)] = r;         // staged: mb_logits_out
                else if (u == 0 && row < n_vocab && m < B) logits[(size_t) m * n_vocab + row] = r;   // (a share of more than 256 rows: the rest directly)
            }
        }
    };
    // (piece 0 is on its way since before the final LayerNorm: mb_te_first)
    for (int it = 0; it < n_items; it += 2) {
        if (it + 1 < n_items) load(it + 1, pfb);
        one(it, pfa);
        if (it + 1 >= n_items) break;
        if (it + 2 < n_items) load(it + 2, pfa);
        one(it + 1, pfb);
    }
}

// the same for a quantised token embedding: a row = 8 lanes x (quads [nb], F32 block scales [nb]) streamed in pieces of 24 blocks; the token
// rows' final LayerNorm outputs sit in `xop` quantised (mb_ln_row<.., true>)
template <int BT>
__device__ __forceinline__ void mb_logits_q(mb_kargs A, const unsigned char * xop, size_t op_bytes, int B, int lane, int wave, unsigned (&pfa)[48], float * lg) {
    const int d = A->d, nb = d >> 5, nbat = (nb + 23) / 24, n_vocab = A->n_vocab;
    const int u = lane & 7;
    GAS float * logits = (GAS float *) A->logits;
    unsigned pfb[48];
    int g1; const int g0 = mb_share(n_vocab, &g1);
    auto grp = [&](int j) { return g0 + wave + MB_NW * j; };
    int n_items = 0;
    for (int j = 0; grp(j) < g1; ++j) n_items += nbat;
    auto load = [&](int it, unsigned (&buf)[48]) {
        const int j = it / nbat, bt = it - j * nbat, row = grp(j) * 8 + (lane >> 3), rc = row < n_vocab ? row : 0;
        const GAS unsigned * wl = (const GAS unsigned *) A->te + ((size_t) rc * 8 + u) * nb;
        const GAS unsigned * dl = (const GAS unsigned *) A->te_d + (size_t) rc * nb;
#pragma unroll
        for (int k4 = 0; k4 < 6; ++k4) {         // (16-byte loads: mb_te_first)
            const int bb = bt * 24 + 4 * k4, bc = bb < nb ? bb : nb - 4;
            const u32x4 q = *(const GAS u32x4 *) (wl + bc), sd = *(const GAS u32x4 *) (dl + bc);
            buf[4 * k4] = q.x; buf[4 * k4 + 1] = q.y; buf[4 * k4 + 2] = q.z; buf[4 * k4 + 3] = q.w;
            buf[24 + 4 * k4] = sd.x; buf[24 + 4 * k4 + 1] = sd.y; buf[24 + 4 * k4 + 2] = sd.z; buf[24 + 4 * k4 + 3] = sd.w;
        }
    };
    float acc[BT];
    const unsigned char * xq[BT], * xd[BT];
#pragma unroll
    for (int m = 0; m < BT; ++m) { const unsigned char * r = xop + (size_t) (m < B ? m : B - 1) * op_bytes; xq[m] = r + (size_t) u * mq_ld(nb) * 4; xd[m] = r + (size_t) 32 * mq_ld(nb); }
    auto one = [&](int it, unsigned (&buf)[48]) {
        const int j = it / nbat, bt = it - j * nbat;
        if (bt == 0) {
#pragma unroll
            for (int m = 0; m < BT; ++m) acc[m] = 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int bb = bt * 24 + k;
            if (bb < nb) {
#pragma unroll
                for (int m = 0; m < BT; ++m)
                    acc[m] = fmaf(__uint_as_float(buf[24 + k]) * ((const float *) xd[m])[bb], (float) __builtin_amdgcn_sdot4((int) buf[k], ((const int *) xq[m])[bb], 0, false), acc[m]);
            }
        }
        if (bt == nbat - 1) {
            const int row = grp(j) * 8 + (lane >> 3);
#pragma unroll
            for (int m = 0; m < BT; ++m) {
                float v = acc[m];
                v = v + dpp_f32<0x104>(v); v = v + dpp_f32<0x102>(v); v = v + dpp_f32<0x101>(v);
                if (u == 0 && j < 4) lg[m * 256 + (8 * j + wave) * 8 + (lane >> 3)] = v;
                else if (u == 0 && row < n_vocab && m < B) logits[(size_t) m * n_vocab + row] = v;
            }
        }
    };
    // (piece 0 is on its way since before the final LayerNorm: mb_te_first)
    for (int it = 0; it < n_items; it += 2) {
        if (it + 1 < n_items) load(it + 1, pfb);
        one(it, pfa);
        if (it + 1 >= n_items) break;
        if (it + 2 < n_items) load(it + 2, pfa);
        one(it + 1, pfb);
    }
}

// -------------------------------------------------------------------------------------------------
// next-token prediction per row (wa_mega.hip: mg_pick / mg_final's records; the host re-derives every token from the logits with the reference's
// rules - this arithmetic is a prediction only).  mb_pick: one wave, the row's token and the sampling state after it -> pk[0..4].
// -------------------------------------------------------------------------------------------------
struct mb_best { float v; int i; };
__device__ __forceinline__ void mb_best_merge(mb_best & a, float v, int i) { if (v > a.v || (v == a.v && i < a.i)) { a.v = v; a.i = i; } }
__device__ __forceinline__ void mb_best_wave(mb_best & a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float v = __shfl_xor(a.v, o, WAVE); const int i = __shfl_xor(a.i, o, WAVE); mb_best_merge(a, v, i); }
}
__device__ __forceinline__ void mb_pick(mb_kargs A, int b, int lane, int * pk, int n_rec) {
    int token = A->rows[b].token, last = A->rows[b].s_last, penult = A->rows[b].s_penult, seek_delta = A->rows[b].s_seek_delta, has_ts = A->rows[b].s_has_ts;
    if (A->rows[b].spec) {
        const GAS int * ps = (const GAS int *) A->rows[b].ps_in;
        const GAS unsigned * rec = (const GAS unsigned *) A->rows[b].rec_in;
        penult = ps[0]; seek_delta = ps[2]; has_ts = ps[3];
        mb_best bt = { -INFINITY, 0x7fffffff }, bs = { -INFINITY, 0x7fffffff };
        u32x4 ra[4]; unsigned rb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {           // every record (n_rec <= 256: four per lane) in ONE round of loads
            const int g = lane + 64 * j, gg = g < n_rec ? g : 0;
            ra[j] = *(const GAS u32x4 *) (rec + gg * 8); rb[j] = rec[gg * 8 + 4];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (lane + 64 * j < n_rec) {
            mb_best_merge(bt, __uint_as_float(ra[j].x), (int) ra[j].y);
            mb_best_merge(bs, __uint_as_float(ra[j].z), (int) ra[j].w);
        }
        mb_best_wave(bt); mb_best_wave(bs);
        float sm = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (lane + 64 * j < n_rec) {
            const float m = __uint_as_float(ra[j].z);
            if (m > -INFINITY) sm += __uint_as_float(rb[j]) * __expf(m - bs.v);
        }
        sm = wave_sum(sm);
        // whisper.cpp:6309-6333: timestamp mass above every text token => a timestamp; else the arg-max of everything allowed
        if (!(bs.v > -INFINITY)) token = bt.v > -INFINITY ? bt.i : 0;
        else if (!(bt.v > -INFINITY)) token = bs.i;
        else if (__logf(sm) + bs.v > bt.v) token = bs.i;
        else token = bs.v > bt.v ? bs.i : bt.i;
        last = token;
        if (token > A->token_beg) { seek_delta = 2 * (token - A->token_beg); has_ts = 1; }
    }
    if (lane == 0) {
        pk[0] = token; pk[1] = last; pk[2] = penult; pk[3] = seek_delta; pk[4] = has_ts;
        if (blockIdx.x == 0 && A->rows[b].ps_out) {
            GAS int * po = (GAS int *) A->rows[b].ps_out;
            po[0] = last; po[1] = penult; po[2] = seek_delta; po[3] = has_ts; po[4] = token;
        }
        if (blockIdx.x == 0 && A->tok_out) ((GAS int *) A->tok_out)[b] = token;
    }
}
// candidate records of logits row m (token row br) from this workgroup's share of the logits, kept in LDS by mb_logits (lg [BT][256]); all threads
__device__ __forceinline__ void mb_record(mb_kargs A, int br, const float * lg_m, const int * pk, unsigned * scratch /* LDS [8][8] */, int tid) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x, n_vocab = A->n_vocab, beg = A->token_beg, eot = A->token_eot;
    const GAS unsigned * smask = (const GAS unsigned *) A->rows[br].smask;
    const int st_last = pk[1], st_penult = pk[2], st_seek = pk[3], st_has = pk[4];
    const bool last_ts = st_last >= beg, penult_ts = st_penult < 0 || st_penult >= beg;
    const bool no_ts = last_ts && penult_ts, no_text = last_ts && !penult_ts;
    const int ts_min = st_has ? beg + st_seek / 2 : beg;
    mb_best bt = { -INFINITY, 0x7fffffff }, bs = { -INFINITY, 0x7fffffff };
    float s_ts = 0.0f;
    int g1; const int g0 = mb_share(n_vocab, &g1);
    if (tid < 256) {        // local row li  <->  vocabulary row 8 g0 + li (mb_share)
        const int li = tid, row = 8 * g0 + li;
        if (row < n_vocab && row < 8 * g1) {
            const float r = lg_m[li];
            const unsigned mw = smask[row >> 5];
            if (!((mw >> (row & 31)) & 1u)) {
                if (row >= beg) { if (!no_ts && row >= ts_min) { bs.v = r; bs.i = row; s_ts = 1.0f; } }
                else if (!(no_text && row < eot)) { bt.v = r; bt.i = row; }
            }
        }
    }
    const float m_loc = bs.v;
    mb_best_wave(bt); mb_best_wave(bs);
    float sw = m_loc > -INFINITY ? s_ts * __expf(m_loc - bs.v) : 0.0f;
    sw = wave_sum(sw);
    if (lane == 0) { scratch[wave * 8 + 0] = __float_as_uint(bt.v); scratch[wave * 8 + 1] = (unsigned) bt.i; scratch[wave * 8 + 2] = __float_as_uint(bs.v);
                     scratch[wave * 8 + 3] = (unsigned) bs.i; scratch[wave * 8 + 4] = __float_as_uint(sw); }
    mb_barrier();
    if (wave == 0) {
        mb_best t2 = { -INFINITY, 0x7fffffff }, s2 = { -INFINITY, 0x7fffffff };
        float sl = 0.0f, ml = -INFINITY;
        if (lane < MB_NW) { t2.v = __uint_as_float(scratch[lane * 8 + 0]); t2.i = (int) scratch[lane * 8 + 1]; s2.v = __uint_as_float(scratch[lane * 8 + 2]); s2.i = (int) scratch[lane * 8 + 3];
                            sl = __uint_as_float(scratch[lane * 8 + 4]); ml = s2.v; }
        mb_best_wave(t2); mb_best_wave(s2);
        float sg = ml > -INFINITY ? sl * __expf(ml - s2.v) : 0.0f;
        sg = wave_sum(sg);
        if (lane == 0) {
            GAS unsigned * ro = (GAS unsigned *) A->rows[br].rec_out + (size_t) wg * 8;
            ro[0] = __float_as_uint(t2.v); ro[1] = (unsigned) t2.i; ro[2] = __float_as_uint(s2.v); ro[3] = (unsigned) s2.i; ro[4] = __float_as_uint(sg);
        }
    }
    mb_barrier();
}

// -------------------------------------------------------------------------------------------------
// the kernel
// -------------------------------------------------------------------------------------------------
struct mb_phase { const wa_f16 * W; const float * D; const float * bias; const float * scale; int N, K, r, rc, nck, soff; };

template <int NP, bool Q>
__device__ __forceinline__ void mb_body(mb_kargs A_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const mb_kargs A = mb_uniform(A_);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x, nwg = gridDim.x;
    mb_ctl c; c.status = (gu32 *) A->status; c.seq = A->seq; c.dead = false;
    const unsigned seq = c.seq;
    const int d = A->d, L = A->n_layer, B = A->B, H = A->n_head, d4 = 4 * d, RG = A->row_gr;
    const mb_layers Ly = (mb_layers) A->layers;

    // LDS: xres [8][8] f32 | fc1x [8][32] f32 (quantised models: a block of FC1 outputs) | pk, record scratch | lnp: gamma | beta of the next LayerNorm | xinB: B operand rows
    // of length d (LayerNorm outputs) | area: B operand rows of length 4d (gathered inputs) / attention scratch | slot 0 | slot 1 (weight rows, then - in
    // the last KB - the rows' bias and scale).  An operand row is [len] F16, or - quantised models - Q8_0 in the order of mb_dotq8.
    float  * xres = (float *) smem;
    float  * fc1x = (float *) (smem + 256);
    int    * pk   = (int *) (smem + 1280);            // [8][8] per token row: its token and the sampling state after it (mb_pick)
    unsigned * rscr = (unsigned *) (smem + 1536);     // [8][8] scratch of mb_record
    float  * lnp  = (float *) (smem + 1792);
    const size_t lnp_bytes = 2 * (((size_t) d * 4 + 1023) & ~(size_t) 1023);      // (an LDS-DMA instruction writes a whole KB: each vector ends on one)
    const size_t opB = Q ? mq_row_bytes(d >> 5) : (size_t) d * 2, op4 = Q ? mq_row_bytes(d4 >> 5) : (size_t) d4 * 2;      // bytes of an operand row
    unsigned char * xinB = smem + 1792 + lnp_bytes;
    const size_t xinB_bytes = ((size_t) B * opB + 255) & ~(size_t) 255;
    unsigned char * area = xinB + xinB_bytes;
    unsigned char * xin = area;
    size_t area_bytes = (size_t) B * op4;
    if (area_bytes < MB_ATT_BYTES) area_bytes = MB_ATT_BYTES;
    area_bytes = (area_bytes + 255) & ~(size_t) 255;
    unsigned char * slot0 = area + area_bytes;
    const int slot_bytes = A->slot_bytes;

    // rows of every matrix this workgroup owns (an even count, the same for every workgroup: mg_rpw).  Quantised models: the first MLP product by
    // WHOLE Q8_0 blocks - workgroup b < 4d / 32 owns rows 32 b .. 32 b + 31 - so that a block leaves already quantised (9 granules, not 32 F32 values)
    auto rpw = [&](int N) { const int r = (N + nwg - 1) / nwg; return (r + 1) & ~1; };
    const int r_qkv = rpw(3 * d), r_d = rpw(d), r_ff = Q ? 32 : rpw(d4);
    const int row_d = wg * r_d;
    auto phase_of = [&](int l, int p) {
        const __attribute__((address_space(4))) wa_mega_layer & Y = Ly[l];
        mb_phase ph;
        ph.scale = nullptr;
        if (p == 0)      { ph.W = Y.qkv_w; ph.D = Y.qkv_d; ph.bias = Y.qkv_b; ph.scale = Y.qkv_s; ph.N = 3 * d; ph.K = d; ph.r = r_qkv; }
        else if (p == 1) { ph.W = Y.out_w; ph.D = Y.out_d; ph.bias = Y.out_b; ph.N = d; ph.K = d; ph.r = r_d; }
        else if (p == 2) { ph.W = Y.cq_w;  ph.D = Y.cq_d;  ph.bias = Y.cq_b;  ph.N = d; ph.K = d; ph.r = r_d; }
        else if (p == 3) { ph.W = Y.co_w;  ph.D = Y.co_d;  ph.bias = Y.co_b;  ph.N = d; ph.K = d; ph.r = r_d; }
        else if (p == 4) { ph.W = Y.fc1_w; ph.D = Y.fc1_d; ph.bias = Y.fc1_b; ph.N = d4; ph.K = d; ph.r = r_ff; }
        else             { ph.W = Y.fc2_w; ph.D = Y.fc2_d; ph.bias = Y.fc2_b; ph.N = d; ph.K = d4; ph.r = r_d; }
        const int gp = (!Q && p == 5) ? 4 : 8;
        const int stride = Q ? ph.K + MB_PAD + (ph.K >> 3) + MB_PAD : 2 * ph.K + MB_PAD;          // LDS bytes per weight row (quantised: quants + block scales)
        int rc = ((slot_bytes - (Q ? 3072 : 1024)) / stride) / gp * gp;
        const int rmax = (ph.r + gp - 1) / gp * gp;
        ph.rc = rc > rmax ? rmax : rc;
        ph.nck = (ph.r + ph.rc - 1) / ph.rc;
        ph.soff = (ph.rc * (ph.K + MB_PAD) + 1023) & ~1023;                                       // (quantised) where the scales' image starts
        return ph;
    };
    // wave 7: request chunk ck of phase (l, p) into the slot of parity `par`
    auto request = [&](int l, int p, int ck, int par) {
        const mb_phase ph = phase_of(l, p);
        const int row0 = wg * ph.r + ck * ph.rc;
        if (row0 >= ph.N) return;
        const int R = min(ph.rc, ph.r - ck * ph.rc);
        unsigned char * slot = slot0 + (size_t) par * slot_bytes;
        if constexpr (Q) {
            mb_dma_rows((const GAS unsigned char *) ph.W, ph.K, row0, R, ph.N, slot, lane);
            mb_dma_rows((const GAS unsigned char *) ph.D, ph.K >> 3, row0, R, ph.N, slot + ph.soff, lane);
        } else mb_dma_rows((const GAS unsigned char *) ph.W, 2 * ph.K, row0, R, ph.N, slot, lane);
        mb_dma_f32(ph.bias, row0, ph.N, (float *) (slot + slot_bytes - 512), lane);
        if (ph.scale) mb_dma_f32(ph.scale, row0, ph.N, (float *) (slot + slot_bytes - 256), lane);
    };
    // wave 7: the parameters of the LayerNorm that follows the product phase p of layer l
    auto request_ln = [&](int l, int p) {
        const float * w, * b_;
        if (p == 0)      { w = Ly[l].ln2_w; b_ = Ly[l].ln2_b; }
        else if (p == 2) { w = Ly[l].ln3_w; b_ = Ly[l].ln3_b; }
        else if (l + 1 < L) { w = Ly[l + 1].ln1_w; b_ = Ly[l + 1].ln1_b; }
        else             { w = A->lnf_w; b_ = A->lnf_b; }
        mb_dma_vec(w, d, lnp, lane); mb_dma_vec(b_, d, lnp + ((d + 255) & ~255), lane);
    };
    const int twg = A->dbg ? ((const GAS int *) A->dbg)[4095] : 0;          // (trace: the workgroup that stamps)
    const bool tw = wg == twg && tid == 0;
    int par = 0;                 // parity of the chunk the NEXT product phase reads
    // One product phase: the operand rows are in LDS; per chunk: barrier (weights landed, operand visible, previous chunk's readers done) ->
    // wave 7 requests the next chunk of the whole sequence -> waves 0..6 run the tasks.
    auto run_phase = [&](int l, int p, auto tasks) {
        const mb_phase ph = phase_of(l, p);
        for (int ck = 0; ck < ph.nck; ++ck) {
            mb_barrier_w(wave);
            if (p == 1) mb_trace(A, tw, l * 32 + 14); else if (p == 4) mb_trace(A, tw, l * 32 + 15); else if (p == 2) mb_trace(A, tw, l * 32 + 27); else if (p == 5) mb_trace(A, tw, l * 32 + 28);
            if (wave == MB_NW - 1) {
                int nl_ = l, np_ = p, nc_ = ck + 1;
                if (nc_ >= ph.nck) { nc_ = 0; np_ = p + 1; if (np_ >= 6) { np_ = 0; nl_ = l + 1; } }
                if (nl_ < L) request(nl_, np_, nc_, par ^ 1);
                if (ck == 0 && (p == 0 || p == 2 || p == 4)) request_ln(l, p);      // (the LayerNorm before this phase has been through: its parameters may go)
            } else {
                const int Rc = min(ph.rc, ph.r - ck * ph.rc);
                MB_CHAOS_AT(10u + p);
                tasks(ph, slot0 + (size_t) par * slot_bytes, wg * ph.r + ck * ph.rc, Rc, ck * ph.rc);
            }
            par ^= 1;
        }
    };
    // the products of a chunk whose rows are 8 lanes wide, three token rows per pass (either weight format); epi as mb_products
    // Token rows per task: a phase's tasks = (groups of 8 weight rows) x (sub-batches of the token rows) run on the seven computing waves at once, so the
    // sub-batches are as narrow as still gives <= 7 tasks (one octet of rows: one token row per task up to B = 7; two octets: two per task up to B = 6).
    auto prod8 = [&](const mb_phase & ph, const unsigned char * slot, int row_base, int Rc, const float * scale_l, const unsigned char * xop, size_t opb, auto epi) {
        const float * bias_l = (const float *) (slot + slot_bytes - 512);
        const int groups = (Rc + 7) >> 3;
        const int bc = groups * B <= MB_NCW ? 1 : groups * ((B + 1) >> 1) <= MB_NCW ? 2 : 3;
        if constexpr (Q) {
            if (bc == 1)      mb_products_q<1>(slot, ph.soff, Rc, row_base, ph.N, bias_l, scale_l, xop, opb, ph.K, B, wave, lane, epi);
            else if (bc == 2) mb_products_q<2>(slot, ph.soff, Rc, row_base, ph.N, bias_l, scale_l, xop, opb, ph.K, B, wave, lane, epi);
            else              mb_products_q<3>(slot, ph.soff, Rc, row_base, ph.N, bias_l, scale_l, xop, opb, ph.K, B, wave, lane, epi);
        } else {
            if (bc == 1)      mb_products<8, 1>(slot, 2 * ph.K + MB_PAD, Rc, row_base, ph.N, bias_l, scale_l, (const wa_f16 *) xop, (int) (opb >> 1), ph.K, B, wave, lane, epi);
            else if (bc == 2) mb_products<8, 2>(slot, 2 * ph.K + MB_PAD, Rc, row_base, ph.N, bias_l, scale_l, (const wa_f16 *) xop, (int) (opb >> 1), ph.K, B, wave, lane, epi);
            else              mb_products<8, 3>(slot, 2 * ph.K + MB_PAD, Rc, row_base, ph.N, bias_l, scale_l, (const wa_f16 *) xop, (int) (opb >> 1), ph.K, B, wave, lane, epi);
        }
    };
    // gathered attention outputs / MLP activations -> operand rows in `xin`: packed F16 pairs, or - quantised - blocks of 9 granules (8 quads + scale)
    auto gather_rows = [&](gu64 * edge, int len, unsigned code) {
        if constexpr (Q) {
            const int nb = len >> 5;
            mb_gather<8>(c, edge, B, 9 * nb, RG, tid, lane, [&](int b, int j, unsigned v) {
                const int blk = j / 9, kq = j - 9 * blk;
                unsigned char * op = xin + (size_t) b * (len == d ? opB : op4);
                if (kq < 8) mq_quads(op, nb, kq)[blk] = v; else mq_scales(op, nb)[blk] = __uint_as_float(v); }, code);
        } else {
            const size_t opb = len == d ? (size_t) d4 * 2 : op4;          // (F16: the d-long rows keep the 4d pitch)
            if (len == d) mb_gather<8>(c, edge, B, len >> 1, RG, tid, lane, [&](int b, int j, unsigned v) { ((unsigned *) (xin + (size_t) b * opb))[j] = v; }, code);
            else          mb_gather<16>(c, edge, B, len >> 1, RG, tid, lane, [&](int b, int j, unsigned v) { ((unsigned *) (xin + (size_t) b * opb))[j] = v; }, code);
        }
    };
    const size_t opD = Q ? opB : (size_t) d4 * 2;      // pitch of the d-long operand rows in `xin`

    if (wave == MB_NW - 1 && L > 0) request(0, 0, 0, 0);
    if (wave < B) mb_pick(A, wave, lane, pk + 8 * wave, min(nwg, 256));      // the rows' tokens: given, or picked from the records their previous passes left
    mb_barrier();
#define MB_T(k) do { mb_trace(A, tw, l * 32 + (k)); mb_trace(A, tid == 0 && l == MB_TRACE_LAYER, 8192 + wg * 16 + (k)); } while (0)      /* every workgroup at ONE layer */
    for (int l = 0; l < L; ++l) {
        const __attribute__((address_space(4))) wa_mega_layer & Y = Ly[l];
        MB_T(0);
        MB_CHAOS_AT(27u);
        // ---------------- P1: LayerNorm + q|k|v ----------------
        // F16 kernels: two instances of the first LayerNorm - layer 0 (embedding rows, gamma | beta from global memory) and the others (granules, LDS).  As ONE
        // instance with run-time choices the layers >= 1 paid 4-8 us for code they never run (medium, 5 rows: 12.9 -> 5.1 us; a 5-row pass 1.80 -> 1.65 ms).  The
        // quantised kernels keep one instance: they are short of registers (99-198 spilled VGPRs) and a second copy cost more than it won (large-v3-q5_0: 3.22 -> 3.65 ms).
        if (wave < B) {
            if constexpr (Q) {
                mb_ln_row<NP, Q>(A, c, l == 0 ? nullptr : mb_edge(A, l - 1, E_X3) + (size_t) wave * RG, lnp, l == 0 ? Y.ln1_w : nullptr, l == 0 ? Y.ln1_b : nullptr,
                                 wave, lane, (wa_f16 *) (xinB + (size_t) wave * opB), xres + wave * 8, row_d, r_d, 100u + l, pk[8 * wave], tw, l * 32 + 29);
            } else {
                if (l == 0) mb_ln_row<NP, Q>(A, c, nullptr, lnp, Y.ln1_w, Y.ln1_b, wave, lane, (wa_f16 *) (xinB + (size_t) wave * opB), xres + wave * 8, row_d, r_d, 100u, pk[8 * wave]);
                else        mb_ln_row<NP, Q>(A, c, mb_edge(A, l - 1, E_X3) + (size_t) wave * RG, lnp, nullptr, nullptr, wave, lane, (wa_f16 *) (xinB + (size_t) wave * opB), nullptr, 0, 0, 100u + l, 0, tw, l * 32 + 29);
            }
        }
        MB_T(1);
        run_phase(l, 0, [&](const mb_phase & ph, const unsigned char * slot, int row_base, int Rc, int) {
            gu64 * eq = mb_edge(A, l, E_QKV);
            prod8(ph, slot, row_base, Rc, (const float *) (slot + slot_bytes - 256), xinB, opB, [&](auto & res, int b0, int b1, int n, bool has, float bias, float scale) {
#pragma unroll
                for (int j = 0; j < (int) (sizeof(res) / sizeof(float)); ++j) {
                    const int b = __builtin_amdgcn_readfirstlane(b0 + j < b1 ? b0 + j : b1 - 1);      // (wave-uniform: the row record comes by scalar loads)
                    float v = res[j] + bias;
                    v = v * scale;
                    const unsigned pk = mb_pack_h2((unsigned) f2h(v));
                    if (has && b0 + j < b1 && (lane & 15) == 0) {
                        gr_store(eq + (size_t) b * RG + (n >> 1), seq, pk);
                        if (n >= d) {       // new key / value also go to the row's KV cell for later tokens
                            GAS wa_f16 * cell = (n < 2 * d ? (GAS wa_f16 *) A->rows[b].kv_k + (n - d) : (GAS wa_f16 *) A->rows[b].kv_v + (n - 2 * d)) +
                                                (size_t) l * A->kv_layer_stride + (size_t) A->rows[b].kv_head * d;
                            *(GAS unsigned *) cell = pk;
                        }
                    }
                }
            });
        });
        MB_T(2);
        // ---------------- P2: self-attention ----------------
        MB_CHAOS_WG(20u); MB_CHAOS_AT(21u);
        for (int u = wg; u < B * H; u += nwg) mb_unit_self<Q>(A, c, area, l, u / H, u % H, tid, tw);
        MB_T(3);
        // ---------------- P3: out-projection + residual ----------------
        MB_CHAOS_AT(22u);
        gather_rows(mb_edge(A, l, E_AO), d, 200u + l);
        auto resid_epi = [&](gu64 * ex) {
            return [&, ex](auto & res, int b0, int b1, int n, bool has, float bias, float) {
#pragma unroll
                for (int j = 0; j < (int) (sizeof(res) / sizeof(float)); ++j) {
                    const int b = b0 + j;
                    if (has && b < b1) {
                        const float v = res[j] + bias;
                        const float xn = v + xres[b * 8 + (n - row_d)];
                        gr_store(ex + (size_t) b * RG + n, seq, __float_as_uint(xn));
                        xres[b * 8 + (n - row_d)] = xn;
                    }
                }
            };
        };
        MB_T(4);
        run_phase(l, 1, [&](const mb_phase & ph, const unsigned char * slot, int row_base, int Rc, int) { prod8(ph, slot, row_base, Rc, nullptr, xin, opD, resid_epi(mb_edge(A, l, E_X1))); });
        MB_T(5);
        // ---------------- P4: LayerNorm + cross query ----------------
        if (wave < B) mb_ln_row<NP, Q>(A, c, mb_edge(A, l, E_X1) + (size_t) wave * RG, lnp, nullptr, nullptr, wave, lane, (wa_f16 *) (xinB + (size_t) wave * opB), nullptr, 0, 0, 300u + l);
        MB_T(6);
        run_phase(l, 2, [&](const mb_phase & ph, const unsigned char * slot, int row_base, int Rc, int) {
            gu64 * eq = mb_edge(A, l, E_QC);
            prod8(ph, slot, row_base, Rc, nullptr, xinB, opB, [&](auto & res, int b0, int b1, int n, bool has, float bias, float) {
#pragma unroll
                for (int j = 0; j < (int) (sizeof(res) / sizeof(float)); ++j) {
                    const int b = b0 + j;
                    const float v = res[j] + bias;
                    const unsigned pk = mb_pack_h2((unsigned) f2h(v));
                    if (has && b < b1 && (lane & 15) == 0) gr_store(eq + (size_t) b * RG + (n >> 1), seq, pk);
                }
            });
        });
        MB_T(7);
        // ---------------- P5: cross-attention ----------------
        MB_CHAOS_WG(23u); MB_CHAOS_AT(24u);
        // a (row, head) in 4 quarters.  (Two halves - where the quarters of all rows need a second round on part of the grid, B H 4 > workgroups - were
        // built (mb_unit_cross<Q, 2>) and measured: the 144 registers of a half's keys / values push the whole kernel into scratch (463 spilled
        // registers; 5 rows 0.67 -> 0.78 ms, 8 rows 0.92 -> 1.06 ms per step), the half itself took 22 us against 2 x 9.8 for two rounds of quarters.  Tried again once
        // the kernel was free of scratch (mb_ln_row's opaque lane): ~30 spilled registers, a half takes 17 us = two quarters, 8 rows 0.863 -> 0.877 ms: no.)
        for (int u = wg; u < B * H * 4; u += nwg) {
            const int bh = u >> 2;
            mb_cross_regs<4> CR;    // (asked for here: earlier - across the cross-query products - the 72 registers cost those products 2 us and won 0.8)
            mb_cross_load<4>(A, l, bh / H, bh % H, u & 3, tid, CR);
            mb_unit_cross<Q, 4>(A, c, area, l, bh / H, bh % H, u & 3, tid, tw, CR);
        }
        MB_T(8);
        // ---------------- P6: out-projection + residual ----------------
        MB_CHAOS_AT(25u);
        gather_rows(mb_edge(A, l, E_AO2), d, 400u + l);
        MB_T(9);
        run_phase(l, 3, [&](const mb_phase & ph, const unsigned char * slot, int row_base, int Rc, int) { prod8(ph, slot, row_base, Rc, nullptr, xin, opD, resid_epi(mb_edge(A, l, E_X2))); });
        MB_T(10);
        // ---------------- P7: LayerNorm + FC1 + GELU ----------------
        if (wave < B) mb_ln_row<NP, Q>(A, c, mb_edge(A, l, E_X2) + (size_t) wave * RG, lnp, nullptr, nullptr, wave, lane, (wa_f16 *) (xinB + (size_t) wave * opB), nullptr, 0, 0, 500u + l);
        MB_T(11);
        run_phase(l, 4, [&](const mb_phase & ph, const unsigned char * slot, int row_base, int Rc, int ck_row) {
            gu64 * eh = mb_edge(A, l, E_HF);
            const GAS wa_f16 * gelu = (const GAS wa_f16 *) A->gelu;
            prod8(ph, slot, row_base, Rc, nullptr, xinB, opB, [&](auto & res, int b0, int b1, int n, bool has, float bias, float) {
                constexpr int NB_ = (int) (sizeof(res) / sizeof(float));
                float tv[NB_];                                         // wa_gelu (vec.h:571-585) through the F16 table: the look-ups of the token rows together
#pragma unroll
                for (int j = 0; j < NB_; ++j) { res[j] = res[j] + bias; tv[j] = h2f(gelu[has ? f2h(res[j]) : 0]); }
#pragma unroll
                for (int j = 0; j < NB_; ++j) {
                    const int b = b0 + j;
                    const float v = res[j];
                    float gl = v;
                    if (v <= -10.0f) gl = 0.0f; else if (v < 10.0f) gl = tv[j];
                    if constexpr (Q) { if (has && b < b1) fc1x[b * 32 + (n - wg * 32)] = gl; }      // F32: quantised below, once the block's 32 rows are through
                    else {
                        const unsigned pk = mb_pack_h2((unsigned) f2h(gl));
                        if (has && b < b1 && (lane & 15) == 0) gr_store(eh + (size_t) b * RG + (n >> 1), seq, pk);
                    }
                }
            });
            (void) ck_row;
        });
        if constexpr (Q) {      // this workgroup's block of every token row: quantize_row_q8_0, published as 9 granules (wave b: row b)
            mb_barrier();
            if (wave < B && 32 * wg < d4) {
                float dq;
                const unsigned w = mq_quant32(fc1x[wave * 32 + (lane & 31)], dq);
                gu64 * eb = mb_edge(A, l, E_HF) + (size_t) wave * RG + (size_t) wg * 9;
                if (lane < 32 && (lane & 3) == 0) gr_store(eb + (lane >> 2), seq, w);
                if (lane == 0) gr_store(eb + 8, seq, __float_as_uint(dq));
            }
        }
        MB_T(12);
        // ---------------- P8: FC2 + residual ----------------
        MB_CHAOS_AT(26u);
        gather_rows(mb_edge(A, l, E_HF), d4, 600u + l);
        MB_T(13);
        run_phase(l, 5, [&](const mb_phase & ph, const unsigned char * slot, int row_base, int Rc, int) {
            const float * bias_l = (const float *) (slot + slot_bytes - 512);
            const int groups = (Rc + (Q ? 7 : 3)) / (Q ? 8 : 4);
            if constexpr (Q) {
                if (groups * B <= MB_NCW) mb_products_q<1>(slot, ph.soff, Rc, row_base, ph.N, bias_l, nullptr, xin, op4, ph.K, B, wave, lane, resid_epi(mb_edge(A, l, E_X3)));
                else                      mb_products_q<2>(slot, ph.soff, Rc, row_base, ph.N, bias_l, nullptr, xin, op4, ph.K, B, wave, lane, resid_epi(mb_edge(A, l, E_X3)));
            } else {
                if (groups * B <= MB_NCW) mb_products<16, 1>(slot, 2 * ph.K + MB_PAD, Rc, row_base, ph.N, bias_l, nullptr, (const wa_f16 *) xin, d4, ph.K, B, wave, lane, resid_epi(mb_edge(A, l, E_X3)));
                else                      mb_products<16, 2>(slot, 2 * ph.K + MB_PAD, Rc, row_base, ph.N, bias_l, nullptr, (const wa_f16 *) xin, d4, ph.K, B, wave, lane, resid_epi(mb_edge(A, l, E_X3)));
            }
        });
    }
    { const int l = L; MB_T(0); }
    // ---------------- final LayerNorm + logits (of the token rows that want them) ----------------
    const int n_out = A->n_out;
    unsigned pf0[48];
    mb_te_first(A, Q, pf0, lane, wave);
    if (wave < n_out) {
        const int br = A->out_row[wave];
        if constexpr (Q) {
            mb_ln_row<NP, Q>(A, c, L > 0 ? mb_edge(A, L - 1, E_X3) + (size_t) br * RG : nullptr, lnp, L > 0 ? nullptr : A->lnf_w, L > 0 ? nullptr : A->lnf_b, br, lane,
                             (wa_f16 *) (xinB + (size_t) wave * opB), nullptr, 0, 0, 3000u, pk[8 * br]);
        } else {
            if (L > 0) mb_ln_row<NP, Q>(A, c, mb_edge(A, L - 1, E_X3) + (size_t) br * RG, lnp, nullptr, nullptr, br, lane, (wa_f16 *) (xinB + (size_t) wave * opB), nullptr, 0, 0, 3000u);
            else       mb_ln_row<NP, Q>(A, c, nullptr, lnp, A->lnf_w, A->lnf_b, br, lane, (wa_f16 *) (xinB + (size_t) wave * opB), nullptr, 0, 0, 3000u, pk[8 * br]);      // (a model without layers: tests)
        }
    }
    bool want_rec = false;
    for (int m = 0; m < n_out; ++m) want_rec = want_rec || A->rows[A->out_row[m]].smask != nullptr;
    float * lg = (float *) area;            // (the gathered-inputs area holds nothing any more) the workgroup's logits, staged for whole-line stores and the records
    mb_barrier();
    if constexpr (Q) {
        if (n_out <= 2) mb_logits_q<2>(A, xinB, opB, n_out, lane, wave, pf0, lg); else if (n_out <= 4) mb_logits_q<4>(A, xinB, opB, n_out, lane, wave, pf0, lg);
        else if (n_out <= 5) mb_logits_q<5>(A, xinB, opB, n_out, lane, wave, pf0, lg); else mb_logits_q<8>(A, xinB, opB, n_out, lane, wave, pf0, lg);
    } else {
        if (n_out <= 2) mb_logits<2>(A, (const wa_f16 *) xinB, n_out, lane, wave, pf0, lg); else if (n_out <= 4) mb_logits<4>(A, (const wa_f16 *) xinB, n_out, lane, wave, pf0, lg);
        else if (n_out <= 5) mb_logits<5>(A, (const wa_f16 *) xinB, n_out, lane, wave, pf0, lg); else mb_logits<8>(A, (const wa_f16 *) xinB, n_out, lane, wave, pf0, lg);
    }
    mb_barrier();
    mb_logits_out(A, lg, n_out, tid);
    if (want_rec) {          // candidate records of every row that asked for them (the next pass of its chunk picks its token from them)
        for (int m = 0; m < n_out; ++m) {
            const int br = A->out_row[m];
            if (A->rows[br].smask) mb_record(A, br, lg + m * 256, pk + 8 * br, rscr, tid);
        }
    }
    { const int l = L; MB_T(1); }
    if (wg == 0 && tid == 0) ((GAS unsigned *) A->status)[1] = seq;       // this launch ran (the host accepts a step only with its own number here)
}

// One kernel per LayerNorm width (elements per lane: 12 for d <= 768, 16 for <= 1024, 20 for <= 1280) and weight format, each in a translation unit
// of its own (the Makefile compiles this file three times, -DWA_ROWS_NP=12 / 16 / 20): with every width in ONE kernel the register allocation was
// the widest variant's for all of them (1525 spilled SGPRs and 32-113 VGPRs against 503 and 18 for the d <= 768 kernel alone).
#ifndef WA_ROWS_NP
#define WA_ROWS_NP 12
#endif
#define MB_CAT_(a, b) a##b
#define MB_CAT(a, b) MB_CAT_(a, b)
__global__ __launch_bounds__(MB_THREADS) void MB_CAT(k_decode_rows_np, WA_ROWS_NP)(const wa_rows_args A) {
    mb_body<WA_ROWS_NP, false>((mb_kargs) __builtin_amdgcn_kernarg_segment_ptr());
}
// the same step for a quantised model (Q5_0 / Q8_0 files): a kernel of its own, so that the F16 kernel's code and registers stay as they are
__global__ __launch_bounds__(MB_THREADS) void MB_CAT(k_decode_rows_q_np, WA_ROWS_NP)(const wa_rows_args A) {
    mb_body<WA_ROWS_NP, true>((mb_kargs) __builtin_amdgcn_kernarg_segment_ptr());
}
bool MB_CAT(wa_rows_launch_np, WA_ROWS_NP)(hipStream_t s, const wa_rows_args & a, int n_wg, size_t lds) {
    int dev = 0;
    (void) hipGetDevice(&dev);
    static bool attr_set[64] = {};
    if (!attr_set[dev & 63]) {
        if (hipFuncSetAttribute((const void *) MB_CAT(k_decode_rows_np, WA_ROWS_NP), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *) MB_CAT(k_decode_rows_q_np, WA_ROWS_NP), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
        attr_set[dev & 63] = true;
    }
    if (a.quant) hipLaunchKernelGGL(MB_CAT(k_decode_rows_q_np, WA_ROWS_NP), dim3(n_wg), dim3(MB_THREADS), lds, s, a);
    else         hipLaunchKernelGGL(MB_CAT(k_decode_rows_np, WA_ROWS_NP), dim3(n_wg), dim3(MB_THREADS), lds, s, a);
    return hipGetLastError() == hipSuccess;
}

#if WA_ROWS_NP == 12      // the host side once
bool wa_rows_launch_np16(hipStream_t s, const wa_rows_args & a, int n_wg, size_t lds);
bool wa_rows_launch_np20(hipStream_t s, const wa_rows_args & a, int n_wg, size_t lds);

size_t wa_rows_lds_bytes(int d, int B, int n_wg, int quant, int * slot_bytes) {
    if (B < 1 || B > WA_ROWS_MAX || d < 64 || d > WA_MEGA_MAX_D || (d & 127) != 0 || n_wg < 1) return 0;
    const size_t opB = quant ? mq_row_bytes(d >> 5) : (size_t) d * 2, op4 = quant ? mq_row_bytes((4 * d) >> 5) : (size_t) 4 * d * 2;
    const size_t xinB_bytes = ((size_t) B * opB + 255) & ~(size_t) 255;
    size_t area = (size_t) B * op4;
    if (area < MB_ATT_BYTES) area = MB_ATT_BYTES;
    area = (area + 255) & ~(size_t) 255;
    const size_t fixed = 1792 + 2 * (((size_t) d * 4 + 1023) & ~(size_t) 1023) + xinB_bytes + area;
    auto rpw = [&](int N) { const int r = (N + n_wg - 1) / n_wg; return (r + 1) & ~1; };
    if (rpw(d) > 8) return 0;                       // (xres holds 8 residual values per token row)
    if (quant && (4 * d) / 32 > n_wg) return 0;     // (whole-block ownership of the first MLP product: one workgroup per block)
    // a slot holds at least one task group of every phase; the whole chunk of a phase when there is room (fewer barriers)
    size_t need_min, want = 0;
    if (quant) {
        auto rowb = [&](int K) { return (size_t) K + MB_PAD + (size_t) (K >> 3) + MB_PAD; };
        need_min = 8 * rowb(4 * d) + 3072;
        want = std::max(want, (size_t) ((rpw(3 * d) + 7) / 8 * 8) * rowb(d));
        want = std::max(want, (size_t) 32 * rowb(d));
        want = std::max(want, (size_t) ((rpw(d) + 7) / 8 * 8) * rowb(4 * d));
        want = ((want + 1023) & ~(size_t) 1023) + 3072;
    } else {
        need_min = std::max((size_t) 8 * (2 * d + MB_PAD), (size_t) 4 * (8 * d + MB_PAD)) + 1024;      // (+ the KB of bias / scale behind the rows)
        want = std::max(want, (size_t) ((rpw(3 * d) + 7) / 8 * 8) * (2 * d + MB_PAD));
        want = std::max(want, (size_t) ((rpw(4 * d) + 7) / 8 * 8) * (2 * d + MB_PAD));
        want = std::max(want, (size_t) ((rpw(d) + 3) / 4 * 4) * (8 * d + MB_PAD));
        want = ((want + 1023) & ~(size_t) 1023) + 1024;
    }
    const size_t total_max = 160 * 1024;
    if (fixed + 2 * (need_min + 1024) > total_max) return 0;
    size_t slot = std::min(want, (total_max - fixed) / 2);
    slot &= ~(size_t) 1023;                         // whole LDS-DMA pieces
    if (slot < need_min) return 0;
    if (slot_bytes) *slot_bytes = (int) slot;
    return fixed + 2 * slot;
}

bool wa_launch_decode_rows(hipStream_t s, const wa_rows_args & a, int n_wg) {
    int slot = 0;
    const size_t lds = wa_rows_lds_bytes(a.d, a.B, n_wg, a.quant, &slot);
    if (lds == 0 || slot != a.slot_bytes) return false;
    if (a.d <= 768)  return wa_rows_launch_np12(s, a, n_wg, lds);
    if (a.d <= 1024) return wa_rows_launch_np16(s, a, n_wg, lds);
    return wa_rows_launch_np20(s, a, n_wg, lds);
}
#endif
