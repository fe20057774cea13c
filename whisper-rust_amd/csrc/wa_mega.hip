// wa_mega.hip - the single-token decode step (whisper.cpp:2474-2852, n_tokens == 1) as ONE launch.
//
// Why: the step streams 335 MB (ggml-small) = 53 us of HBM time, but as 122 dependent launches it took 760 us: every
// launch pays its own fill / drain and cannot fetch a byte before its predecessor has finished.  Here one grid of one
// 512-thread workgroup per CU stays resident for the whole step:
//   * work is partitioned statically: GEMV workgroups own a slice of the rows of EVERY matrix, H workgroups own one
//     self-attention head each, H more own one cross-attention head each;
//   * everything that does not depend on the token - weights, the self K/V cells of earlier tokens, the encoder K/V -
//     is loaded ahead of the hand-off that needs it (into VGPRs or LDS), so a phase costs one hand-off plus arithmetic;
//   * activations travel between workgroups as 8-byte {tag = launch sequence number, value} granules written with
//     write-through (sc1) stores and polled with sc1 loads: the data is its own flag, no fences, no barrier kernel
//     (MI355X guide, Guideline 16 form R2).  Every poll is bounded and reports a time-out through `status`.
// All arithmetic is the reference order of wa_exact.hip (ggml_vec_dot_f16 chains and tree, ops.cpp soft_max, certified
// F64 LayerNorm sums): the logits are bit-identical to the launch-sequence path and to whisper.cpp CPU.
#include "wa_device.h"
#include "wa_mega.h"

typedef unsigned long long u64;
#define GAS __attribute__((address_space(1)))
typedef GAS u64 gu64;
typedef GAS unsigned gu32;
typedef const GAS wa_f16 * gch;      // every global access is spelled global: a pointer read from the argument block or from
typedef const GAS float * gcf;       // the layer table is generic to the compiler, and a flat access also waits on the LDS counter
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // (HIP's u32x4 class cannot be read through an address-space pointer)

// The roles are inlined into the kernel (as separate functions they saved 112 callee-saved VGPRs per thread on entry: 58 MB of scratch
// per launch).  They read the launch arguments from the kernel-argument segment (scalar loads: every field stays wave-uniform); the
// kernel hands them its address and mg_uniform makes it provably uniform again.
typedef const __attribute__((address_space(4))) wa_mega_args * mg_kargs;
__device__ __forceinline__ mg_kargs mg_uniform(mg_kargs p) {
    const unsigned long long v = (unsigned long long) p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v), hi = __builtin_amdgcn_readfirstlane((unsigned) (v >> 32));
    return (mg_kargs) (((unsigned long long) hi << 32) | lo);
}

#define MG_THREADS 512
#define MG_NW (MG_THREADS / 64)
#define MG_NP3 4                  // LayerNorm elements per lane of one of the six gather waves: d <= 1536
#define MG_SPIN_LIMIT 20000u      // polls (~0.5 us each, ~10 ms) before a hand-off is declared dead: the host then pauses the one-launch step and tries again later
#ifndef MG_DEFER
#define MG_DEFER 1                // request a wave's next weights after the CU's next gather instead of right away (BIG assist re-fetches stay in place)
#endif

enum { E_QKV = 0, E_AO, E_X1, E_QC, E_AO2, E_X2, E_HF, E_X3 };

struct mg_ctl { gu32 * status; unsigned seq; bool dead; };

__device__ __forceinline__ void gr_store(gu64 * g, unsigned seq, unsigned v) {
    __hip_atomic_store(g, ((u64) seq << 32) | (u64) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gr_load(gu64 * g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A granule whose readers all sit on the WRITER'S XCD (`local`, established at run time: mg_role_cross): a plain store keeps the line
// in that XCD's L2, where the readers' L1-bypassing polls find it - an sc1 store drops it from L2 and every reader goes out to the
// fabric (MI355X_MICROARCH.md, inter-workgroup visibility).  Never for a granule that another XCD reads: its L2 would stay stale.
__device__ __forceinline__ void gr_store_l(gu64 * g, unsigned seq, unsigned v, bool local) {
    if (local) __hip_atomic_store(g, ((u64) seq << 32) | (u64) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else       __hip_atomic_store(g, ((u64) seq << 32) | (u64) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) { return (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, 0xf, 0xf, true); }
// The lane index behind an opaque move: addresses and masks derived from it are recomputed where they are used instead of being hoisted
// out of the layer loop - where they sat in scratch and came back behind an s_waitcnt vmcnt(0), i.e. behind the wave's weight prefetch.
__device__ __forceinline__ int mq_fresh(int) {
    unsigned z = 0; asm volatile("" : "+v"(z));         // an opaque zero: the two mbcnt below cannot be merged with an earlier pair (nor kept, nor spilled)
    return (int) __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}
// quantize_row_q8_0 (arch/x86/quants.c) of whole 32-element blocks held one value per lane (as wa_q8_store), the quads packed over DPP:
// one LDS word per four lanes at q32 (the lane's quad row and block), the scale rounded through F16 at dsc.  All lanes take part.
__device__ __forceinline__ void mq_put(float y, bool act, unsigned * q32, float * dsc, int lane) {
    float a = fabsf(y);
    a = fmaxf(a, dpp_f32<0x128>(a)); a = fmaxf(a, dpp_f32<0x124>(a)); a = fmaxf(a, dpp_f32<0x122>(a)); a = fmaxf(a, dpp_f32<0x121>(a));
    a = fmaxf(a, __shfl_xor(a, 16, 32));
    const float dq = a / 127.f, id = a != 0.0f ? 127.f / a : 0.0f;
    const unsigned q = (unsigned) (int) rintf(y * id) & 0xffu;
    const unsigned w = q | (dpp_u32<0x101>(q) << 8) | (dpp_u32<0x102>(q) << 16) | (dpp_u32<0x103>(q) << 24);      // row_shl:1..3
    if (act && (lane & 3) == 0) *q32 = w;
    if (act && (lane & 31) == 0) *dsc = h2f(f2h(dq));
}

__device__ __forceinline__ gu64 * mg_edge(mg_kargs A, int layer, int e) {
    return (gu64 *) A->granules + ((size_t) layer * WA_MEGA_EDGES + e) * A->edge_stride;
}

// Workgroup barrier that orders LDS only.  __syncthreads() also drains the vector-memory counter (s_waitcnt vmcnt(0)), i.e. it
// would make every wave wait here for the weights it has just started to prefetch - the opposite of what the prefetch is for.
__device__ __forceinline__ void mg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (quantised models) LDS of the quantised activation row inside the xin area: quants [4d] bytes, then the block scales
__device__ __forceinline__ int8_t * mq_xq(wa_f16 * xin) { return (int8_t *) xin; }
__device__ __forceinline__ float  * mq_xd(wa_f16 * xin) { return (float *) ((unsigned char *) xin + 6 * WA_MEGA_MAX_D); }
// words between the quad rows u = 0..7 of the activation row: nb | 8 puts the eight 16-byte reads of a product step on disjoint banks
// (u * nb alone: nb = 96 folds them onto two)
__device__ __forceinline__ int mq_ld(int nb) { return nb | 8; }

// optional timeline (tools/mega_debug.py): 100 MHz wall-clock ticks of one workgroup per role, behind the cross-attention dumps
__device__ __forceinline__ void mg_trace(mg_kargs A, bool who, int slot, unsigned v) {
    if (A->dbg && who) ((GAS unsigned *) A->dbg)[(size_t) A->n_layer * A->n_head * 5120 + slot] = v;
}
__device__ __forceinline__ unsigned mg_now() { return (unsigned) wall_clock64(); }
// MG_CHAOS (a test build, tools/chaos_check.sh): some product waves of some workgroups stall for ~25 us right before their product, so that
// the rest of the workgroup - and the rest of the grid - runs far ahead of them.  Results must not change: nothing in LDS may rely on how long
// a product or a hand-off takes.
#ifdef MG_CHAOS
__device__ __forceinline__ void mg_chaos(unsigned wg, unsigned wave, unsigned l, unsigned phase, unsigned seq) {
    unsigned h = (wg * 2654435761u) ^ (wave * 40503u) ^ (l * 2246822519u) ^ (phase * 3266489917u) ^ (seq * 668265263u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    if ((h & 7u) == 0u) for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);
}
#define MG_CHAOS_AT(phase) mg_chaos((unsigned) wg, (unsigned) wave, (unsigned) l, (phase), seq)
#define MG_CHAOS_ID(id, phase, sq) mg_chaos((unsigned) (id), (unsigned) wave, (unsigned) l, (phase), (sq))       /* attention roles: id = 1000 + head / 2000 + workgroup */
#else
#define MG_CHAOS_AT(phase) do { } while (0)
#define MG_CHAOS_ID(id, phase, sq) do { } while (0)
#endif
// (test build) a workgroup may also START late: whole workgroups stall at the head of their role
__device__ __forceinline__ void mg_chaos_start(unsigned seq) {
#ifdef MG_CHAOS
    mg_chaos(blockIdx.x, 0u, 99u, 41u, seq);
#endif
}

// One wave polls the granules idx(0..NPL-1) (idx < 0: none) until every tag equals this launch's sequence number.
// (Measured: a second, staggered poll in flight per wave makes every hand-off LONGER - 0.377 -> 0.401 ms per token -, longer pauses between
// polls too (s_sleep 6: 0.384, 14: 0.402), none at all changes nothing; a pause before the first poll of the gathers that follow an attention phase cuts their polls by 2-3 x and changes
// nothing either: the hand-off time is the store-to-load path itself, not contention by the polls.)
template <int NPL, typename F>
__device__ __forceinline__ unsigned mg_sweep(gu64 * g, F idx, mg_ctl & c, int lane, unsigned (&v)[NPL], unsigned code) {
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int i = idx(k);
            if (i >= 0) { const u64 x = gr_load(g + i); v[k] = (unsigned) x; ok &= (unsigned) (x >> 32) == c.seq; }
        }
        if (__all(ok) || c.dead) return spins;
        if ((spins & 127u) == 127u) {
            const unsigned st = __builtin_amdgcn_readfirstlane(__hip_atomic_load(c.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (st != 0u) { c.dead = true; return spins; }
            if (spins >= MG_SPIN_LIMIT) {
                if (lane == 0) __hip_atomic_store(c.status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c.dead = true;
                return spins;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// -------------------------------------------------------------------------------------------------
// LayerNorm phases.  SIX waves (slots q 0..5, mg_slot; the others pass q < 0 and only keep the barriers) each
// obtain a sixth of the F32 residual row - from granules, or from the embeddings for layer 0 -, keep it in LDS (xf) and
// write their part of LayerNorm(row) as F16 into xin.  ops.cpp:3225-3242 semantics as k_layernorm_exact: F64 sums in an
// arbitrary order, accepted when certified order-independent, else redone in index order (by each slot, identically).
// A single wave polling all d granules took ~2 us per pass and 1.6 us for the arithmetic; a sixth each is faster on both.
// -------------------------------------------------------------------------------------------------
// Gathering is shared by SIX waves: the six lowest-numbered waves not in `ex` (the waves that have only just issued a weight
// prefetch - a poll behind it would wait for those loads first, vmcnt being in order).  Returns the wave's slot or -1.
#define MG_WGTRACE_LAYER 4
#define MG_NQ 6
#define MG_EX_P1 (MG_DEFER ? 0x06u : 0x20u)      /* waves 1, 2 have just asked for the next QKV rows (without MG_DEFER: wave 5 for the next FC2 rows) */
/* With MG_DEFER the fresh requests sit elsewhere: before P4 waves 1, 2 (FC1 rows, asked for after the P3 barrier); before P7 wave 4 (next
   cross-query rows, after the P6 barrier) - or wave 3 in the quantised whole-block form (its FC1 rows, after its P6 product); before the
   FC2 gather wave 3 (next out-projection rows, after P7's LayerNorm). */
#define MG_EX_P4 (MG_DEFER ? 0x06u : 0x08u)      /* (without MG_DEFER wave 3: next out-projection rows) */
#define MG_EX_P7 (BIGP ? 0x18u : MG_DEFER && !QB ? 0x10u : 0x08u)      /* wide form: waves 3 and 4 have both just asked for their FC1 rows */
#define MG_EX_AO 0x06u      /* waves 1, 2: FC1 rows */
#define MG_EX_AO2 0x10u     /* wave 4: next cross-query rows */
#define MG_EX_HF (MG_DEFER ? 0x08u : 0x06u)      /* (without MG_DEFER waves 1, 2: next QKV rows) */
#define MG_EX_HFQ 0x18u     /* quantised, whole-block FC1: waves 3, 4 have just asked for the next out-projection / cross-query rows */
#define MG_EX_FINAL 0x20u   /* wave 5: first logits rows */
__device__ __forceinline__ int mg_slot(int wave, unsigned ex) {
    if ((ex >> wave) & 1u) return -1;
    const int sl = __builtin_popcount(~ex & ((1u << wave) - 1u) & 0xffu);
    return sl < MG_NQ ? sl : -1;
}
__device__ __forceinline__ int mg_seg(int n) { return ((n + MG_NQ - 1) / MG_NQ + 31) & ~31; }      // granules per slot
__device__ __forceinline__ int mg_ln_seg(int d) { return mg_seg(d); }
template <int NP3>
__device__ __forceinline__ void mg_ln_params(float (&gw)[NP3], float (&gb)[NP3], const float * lw, const float * lb, int d, int q, int lane) {
    const int seg = mg_ln_seg(d), i0 = q * seg, i1 = min(d, i0 + seg);
#pragma unroll
    for (int k = 0; k < NP3; ++k) {
        const int i = i0 + lane + 64 * k;
        const bool ok = q >= 0 && i < i1;
        gw[k] = ok ? ((gcf) lw)[i] : 0.0f; gb[k] = ok ? ((gcf) lb)[i] : 0.0f;
    }
}
// gw / gb: gamma and beta of THIS LayerNorm, loaded one phase ahead (a load issued here would sit, with its pointer fetch, in
// front of the polling loads: measured 5 us per LayerNorm phase)
template <int NP3, bool Q = false>        // Q: the normalised row leaves as Q8_0 (quantize_row_q8_0: the next product's operand), not as F16
__device__ __forceinline__ void mg_ln3(mg_kargs A, mg_ctl & c, gu64 * edge /* null: embeddings */, const float (&gw)[NP3], const float (&gb)[NP3], int q,
                                       int lane, float * xf, wa_f16 * xin, double * lnred, unsigned code, int tslot = -1, int token = 0) {
    const int d = A->d, seg = mg_ln_seg(d), i0 = q * seg, i1 = min(d, i0 + seg);
    float xv[NP3];
    if (q >= 0) {
        if (edge) {
            unsigned v[NP3];
            const unsigned sp = mg_sweep<NP3>(edge, [&](int k) { const int i = i0 + lane + 64 * k; return i < i1 ? i : -1; }, c, lane, v, code);
            if (tslot >= 0) { mg_trace(A, lane == 0, tslot, mg_now()); mg_trace(A, lane == 0, tslot + 1, sp); }
#pragma unroll
            for (int k = 0; k < NP3; ++k) xv[k] = (i0 + lane + 64 * k < i1) ? __uint_as_float(v[k]) : 0.0f;
        } else if constexpr (!Q) {                    // k_dec_embed: token embedding + positional embedding
            const gch te = (gch) A->te + (size_t) token * d;
            const gcf pe = (gcf) A->pe + (size_t) A->pos * d;
#pragma unroll
            for (int k = 0; k < NP3; ++k) { const int i = i0 + lane + 64 * k; xv[k] = i < i1 ? h2f(te[i]) + pe[i] : 0.0f; }
        } else {                                      // k_dec_embed_q: the row dequantised (q * d, ggml-quants.c) + positional embedding
            const int nb = d >> 5;
            const GAS int8_t * tq = (const GAS int8_t *) A->te + (size_t) token * 8 * nb * 4;
            const gcf td = (gcf) A->te_d + (size_t) token * nb;
            const gcf pe = (gcf) A->pe + (size_t) A->pos * d;
#pragma unroll
            for (int k = 0; k < NP3; ++k) {
                const int i = i0 + lane + 64 * k, b = i >> 5, el = i & 31;
                xv[k] = i < i1 ? (float) (int) tq[(((size_t) (el >> 2)) * nb + b) * 4 + (el & 3)] * td[b] + pe[i] : 0.0f;
            }
        }
        double s = 0.0, a = 0.0;
#pragma unroll
        for (int k = 0; k < NP3; ++k) { const int i = i0 + lane + 64 * k; if (i < i1) xf[i] = xv[k]; s += (double) xv[k]; a += (double) fabsf(xv[k]); }
        s = wave_sum_d(s); a = wave_sum_d(a);
        if (lane == 0) { lnred[q] = s; lnred[MG_NQ + q] = a; }
    }
    mg_barrier();
    if (tslot >= 0) mg_trace(A, lane == 0, tslot + 4, mg_now());
    float mean = 0.0f;
    // every wave takes the same decision from the same six partial sums (the gather waves need the mean, the others the barrier count)
    double s = ((lnred[0] + lnred[1]) + (lnred[2] + lnred[3])) + (lnred[4] + lnred[5]);
    const double a = ((lnred[6] + lnred[7]) + (lnred[8] + lnred[9])) + (lnred[10] + lnred[11]);
    float mean_hi;
    bool need_seq = false;
    int dd = d;
    if constexpr (Q) asm volatile("" : "+s"(dd));      // (the certificate's F64 constants of d are recomputed here rather than kept over the layer loop in scratch)
    if (!wa_sum_bounds(s, a, dd, mean, mean_hi, A->rn_d)) {       // rare (a mean near zero): second-level certificate over all elements
        if (q >= 0) {
            bool same = true;
#pragma unroll
            for (int k = 0; k < NP3; ++k) if (i0 + lane + 64 * k < i1) same &= wa_mean_indifferent(xv[k], mean, mean_hi);
            if (lane == 0) ((int *) (lnred + 3 * MG_NQ))[q] = __all(same) ? 1 : 0;
        }
        mg_barrier();
        const int * fl = (const int *) (lnred + 3 * MG_NQ);
        need_seq = !(fl[0] & fl[1] & fl[2] & fl[3] & fl[4] & fl[5]);
    }
    if (q >= 0) {
        if (need_seq) {
            if (lane == 0) s = wa_seq_sum_lds(xf, d, false, 0.0f);
            s = __shfl(s, 0, WAVE);
            mean = (float) (s / (double) d);
        }
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NP3; ++k) if (i0 + lane + 64 * k < i1) { const float t = xv[k] - mean; s2 += (double) (t * t); }
        s2 = wave_sum_d(s2);
        if (lane == 0) lnred[2 * MG_NQ + q] = s2;
    }
    mg_barrier();
    if (tslot >= 0) mg_trace(A, lane == 0, tslot + 5, mg_now());
    if (q >= 0) {
        double s2 = ((lnred[12] + lnred[13]) + (lnred[14] + lnred[15])) + (lnred[16] + lnred[17]);
        float variance;
        if (!wa_sum_certain(s2, s2, d, variance, A->rn_d)) {
            if (lane == 0) s2 = wa_seq_sum_lds(xf, d, true, mean);
            s2 = __shfl(s2, 0, WAVE);
            variance = (float) (s2 / (double) d);
        }
        const float scale = 1.0f / sqrtf(variance + A->eps);
        // (Q: a slot's part is whole 32-element blocks - i0, d % 32 == 0 -; lane's block at k is (i0 >> 5) + (lane >> 5) + 2 k)
        const int ln = mq_fresh(lane);
        unsigned * q32 = (unsigned *) mq_xq(xin) + ((ln & 31) >> 2) * mq_ld(d >> 5) + (i0 >> 5) + (ln >> 5);
        float * dsc = mq_xd(xin) + (i0 >> 5) + (ln >> 5);
#pragma unroll
        for (int k = 0; k < NP3; ++k) {
            const int i = i0 + ln + 64 * k;
            float y = xv[k] - mean;
            y = y * scale;
            y = y * gw[k];
            y = y + gb[k];
            if constexpr (Q) mq_put(y, i < i1, q32 + 2 * k, dsc + 2 * k, ln);
            else if (i < i1) xin[i] = f2h(y);
        }
        if (tslot >= 0) mg_trace(A, lane == 0, tslot + 2, mg_now());
    }
    mg_barrier();
}

// wave(s): copy the packed-F16 granules [i0, i1) (two halfs each) into LDS once they are all valid
template <int NPL>
__device__ __forceinline__ void mg_gather_h2(mg_ctl & c, gu64 * edge, int i0, int i1, int lane, unsigned * dst32, unsigned code, mg_kargs A = nullptr,
                                             int tslot = -1) {
    unsigned v[NPL];
    const unsigned sp = mg_sweep<NPL>(edge, [&](int k) { const int i = i0 + lane + 64 * k; return i < i1 ? i : -1; }, c, lane, v, code);
    if (tslot >= 0) { mg_trace(A, lane == 0, tslot, mg_now()); mg_trace(A, lane == 0, tslot + 1, sp); }
#pragma unroll
    for (int k = 0; k < NPL; ++k) { const int i = i0 + lane + 64 * k; if (i < i1) dst32[i] = v[k]; }
}

// wave(s): the F32 granules [i0, i1) (whole 32-element blocks) quantised to Q8_0 into the activation row in LDS once they are all valid
template <int NPL>
__device__ __forceinline__ void mg_gather_q8(mg_ctl & c, gu64 * edge, int i0, int i1, int lane, wa_f16 * xin, int nb, unsigned code, mg_kargs A = nullptr,
                                             int tslot = -1) {
    unsigned v[NPL];
    const unsigned sp = mg_sweep<NPL>(edge, [&](int k) { const int i = i0 + lane + 64 * k; return i < i1 ? i : -1; }, c, lane, v, code);
    if (tslot >= 0) { mg_trace(A, lane == 0, tslot, mg_now()); mg_trace(A, lane == 0, tslot + 1, sp); }
    const int ln = mq_fresh(lane);
    unsigned * q32 = (unsigned *) mq_xq(xin) + ((ln & 31) >> 2) * mq_ld(nb) + (i0 >> 5) + (ln >> 5);
    float * dsc = mq_xd(xin) + (i0 >> 5) + (ln >> 5);
#pragma unroll
    for (int k = 0; k < NPL; ++k) mq_put(__uint_as_float(v[k]), i0 + ln + 64 * k < i1, q32 + 2 * k, dsc + 2 * k, ln);
}

// wave(s): blocks that were quantised by their producer - 9 granules per block (quads 0..7, scale) - copied into the activation row in LDS
template <int NPL>
__device__ __forceinline__ void mg_gather_qb(mg_ctl & c, gu64 * edge, int i0, int i1, int lane, wa_f16 * xin, int nb, unsigned code, mg_kargs A = nullptr,
                                             int tslot = -1) {
    unsigned v[NPL];
    const unsigned sp = mg_sweep<NPL>(edge, [&](int k) { const int i = i0 + lane + 64 * k; return i < i1 ? i : -1; }, c, lane, v, code);
    if (tslot >= 0) { mg_trace(A, lane == 0, tslot, mg_now()); mg_trace(A, lane == 0, tslot + 1, sp); }
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int i = i0 + lane + 64 * k;
        if (i < i1) { const int b = i / 9, kq = i - 9 * b; if (kq < 8) ((unsigned *) mq_xq(xin))[kq * mq_ld(nb) + b] = v[k]; else mq_xd(xin)[b] = __uint_as_float(v[k]); }
    }
}

// -------------------------------------------------------------------------------------------------
// GEMV rows in ggml_vec_dot_f16 order (as k_gemv_exact).  pf[] holds the weights of the first <= 48 (LPR 8) or
// <= 96 (LPR 16) 32-element steps of a lane's row, loaded ahead of the hand-off.
//   LPR 8 : lane u owns elements 4u..4u+3 of every step (8-byte loads);  result in lanes with u == 0
//   LPR 16: lane u owns elements 2u, 2u+1 (4-byte loads; long rows, few of them); result in lanes with u == 0
// -------------------------------------------------------------------------------------------------
template <int NS = 0>
__device__ __forceinline__ void mg_pf8(unsigned (&pf)[96], gch wrow, bool valid, int nsteps_rt, int s0) {
    const int nsteps = NS > 0 ? NS : nsteps_rt;
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        if (s0 + 4 * c < nsteps) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int b = 4 * c + bb;
                if (valid) { const u32x2 t = *(const GAS u32x2 *) (wrow + (size_t) (s0 + b) * 32); pf[2 * b] = t.x; pf[2 * b + 1] = t.y; }
            }
        }
    }
}
template <int NS = 0>
__device__ __forceinline__ float mg_dot8(unsigned (&pf)[96], gch wrow, bool valid, int nsteps_rt, const wa_f16 * xin, int u, bool have_first) {
    const int nsteps = NS > 0 ? NS : nsteps_rt;
    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (int s0 = 0; s0 < nsteps; s0 += 48) {
        if (s0 > 0 || !have_first) mg_pf8<NS>(pf, wrow, valid, nsteps, s0);
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            if (s0 + 4 * c < nsteps) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int b = 4 * c + bb;
                    u32x2 t; t.x = pf[2 * b]; t.y = pf[2 * b + 1];
                    const half4v w4 = __builtin_bit_cast(half4v, t);
                    const half4v x4 = *(const half4v *) (xin + (size_t) (s0 + b) * 32 + 4 * u);
                    acc[0] = fmaf((float) w4[0], (float) x4[0], acc[0]);
                    acc[1] = fmaf((float) w4[1], (float) x4[1], acc[1]);
                    acc[2] = fmaf((float) w4[2], (float) x4[2], acc[2]);
                    acc[3] = fmaf((float) w4[3], (float) x4[3], acc[3]);
                }
            }
        }
    }
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = acc[i];
        v = v + dpp_f32<0x104>(v);          // row_shl:4  s[j] + s[j+2]
        v = v + dpp_f32<0x102>(v);          // row_shl:2  (s0+s2) + (s1+s3)
        t[i] = v + dpp_f32<0x101>(v);       // row_shl:1  a[l] + a[l+4]
    }
    return (t[0] + t[1]) + (t[2] + t[3]);
}
template <int NS = 0>
__device__ __forceinline__ void mg_pf16(unsigned (&pf)[96], gch wrow, bool valid, int nsteps_rt, int s0) {
    const int nsteps = NS > 0 ? NS : nsteps_rt;
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        if (s0 + 8 * c < nsteps) {
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) {
                const int b = 8 * c + bb;
                if (valid) pf[b] = *(const GAS unsigned *) (wrow + (size_t) (s0 + b) * 32);
            }
        }
    }
}
template <int NS = 0>
__device__ __forceinline__ float mg_dot16(unsigned (&pf)[96], gch wrow, bool valid, int nsteps_rt, const wa_f16 * xin, int u, bool have_first) {
    const int nsteps = NS > 0 ? NS : nsteps_rt;
    float acc[2] = { 0.0f, 0.0f };
    for (int s0 = 0; s0 < nsteps; s0 += 96) {
        if (s0 > 0 || !have_first) mg_pf16<NS>(pf, wrow, valid, nsteps, s0);
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            if (s0 + 8 * c < nsteps) {
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) {
                    const int b = 8 * c + bb;
                    const half2v w2 = __builtin_bit_cast(half2v, pf[b]);
                    const half2v x2 = *(const half2v *) (xin + (size_t) (s0 + b) * 32 + 2 * u);
                    acc[0] = fmaf((float) w2[0], (float) x2[0], acc[0]);
                    acc[1] = fmaf((float) w2[1], (float) x2[1], acc[1]);
                }
            }
        }
    }
    float t[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v = acc[i];
        v = v + dpp_f32<0x108>(v);          // row_shl:8  s[j] + s[j+2]
        v = v + dpp_f32<0x104>(v);          // row_shl:4  (s0+s2) + (s1+s3)
        t[i] = v + dpp_f32<0x102>(v);       // row_shl:2  a[l] + a[l+4]   -> lane 0: t0,t1  lane 1: t2,t3
    }
    const float r = t[0] + t[1];
    return r + dpp_f32<0x101>(r);           // (t0+t1) + (t2+t3)
}

// rows of one matrix owned by one GEMV workgroup: an even count, the same for every workgroup
__device__ __forceinline__ int mg_rpw(int N, int nG) { const int r = (N + nG - 1) / nG; return (r + 1) & ~1; }

struct mg_task { gch wrow; bool valid; int row; float bias, scale; const GAS int * wl; const GAS float * dl; };      // wl / dl: quantised rows (below)

template <int NS = 0>
__device__ __forceinline__ mg_task mg_task8(unsigned (&pf)[96], const wa_f16 * W, const float * bias, const float * scale, int N, int K,
                                            int row0, int rows_wg, int grp, int lane) {
    mg_task t;
    const int ri = grp * 8 + (lane >> 3);
    t.row = row0 + ri;
    t.valid = ri < rows_wg && t.row < N;
    t.wrow = (gch) W + (size_t) (t.valid ? t.row : 0) * K + 4 * (lane & 7);
    t.bias = 0.0f; t.scale = 1.0f;
    mg_pf8<NS>(pf, t.wrow, t.valid, K >> 5, 0);
    if (t.valid && (lane & 7) == 0) { if (bias) t.bias = ((gcf) bias)[t.row]; if (scale) t.scale = ((gcf) scale)[t.row]; }
    return t;
}
template <int NS = 0>
__device__ __forceinline__ mg_task mg_task16(unsigned (&pf)[96], const wa_f16 * W, const float * bias, int N, int K, int row0, int rows_wg,
                                             int grp, int lane) {
    mg_task t;
    const int ri = grp * 4 + (lane >> 4);
    t.row = row0 + ri;
    t.valid = ri < rows_wg && t.row < N;
    t.wrow = (gch) W + (size_t) (t.valid ? t.row : 0) * K + 2 * (lane & 15);
    t.bias = 0.0f; t.scale = 1.0f;
    mg_pf16<NS>(pf, t.wrow, t.valid, K >> 5, 0);
    if (t.valid && (lane & 15) == 0 && bias) t.bias = ((gcf) bias)[t.row];
    return t;
}

// -------------------------------------------------------------------------------------------------
// The same rows of a QUANTISED matrix (Q5_0 / Q8_0 files; wa_quant.hip has the contract: ggml_vec_dot_q5_0_q8_0 / q8_0_q8_0, AVX2 order).
// A row is [lane u = 0..7][block][4] signed bytes + [block] F32 scales; the activation row sits in LDS quantised to Q8_0 in the same
// order (xq [u][block][4], xd [block]).  Lane u chains  acc = fma(dw dx, (float) dot4(w, x), acc)  over the blocks in order; the three DPP
// adds are hsum_float_8.  pf[0..47] = the lane's quads of the first 48 blocks it owns, pf[48..95] = their scales, loaded ahead.
//   8 lanes per row : a lane owns every block (K = d: 24-40 blocks);  result in lanes with u == 0
//   16 lanes per row: lanes 0-7 own the first half of the blocks, lanes 8-15 continue the SAME chains over the second half
//                     (the running sums move over by one DPP row_shr:8): twice the registers for the long rows (K = 4d);  result in lane 8
// -------------------------------------------------------------------------------------------------
typedef int   mq_i4 __attribute__((ext_vector_type(4)));
typedef float mq_f4 __attribute__((ext_vector_type(4)));
#define MQ_PF 48
template <int NB = 0>
__device__ __forceinline__ void mq_pf(unsigned (&pf)[96], const GAS int * wl, const GAS float * dl, bool valid, int b0, int b1_rt) {
    const int b1 = NB > 0 ? NB : b1_rt;
#pragma unroll
    for (int c = 0; c < MQ_PF / 4; ++c) {
        if (b0 + 4 * c < b1 && valid) {
            const mq_i4 w = *(const GAS mq_i4 *) (wl + b0 + 4 * c);
            const mq_f4 sd = *(const GAS mq_f4 *) (dl + b0 + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { pf[4 * c + e] = (unsigned) w[e]; pf[MQ_PF + 4 * c + e] = __float_as_uint(sd[e]); }
        }
    }
}
// the lane's chains over its blocks [b0, b1): xq_u = this lane's quads in LDS (int per block), xd = the row's block scales in LDS
template <int NB = 0>
__device__ __forceinline__ float mq_chain(unsigned (&pf)[96], const GAS int * wl, const GAS float * dl, bool valid, int b0, int b1_rt, const int * xq_u,
                                          const float * xd, float acc, bool have_first) {
    const int b1 = NB > 0 ? NB : b1_rt;
    for (int s0 = b0; s0 < b1; s0 += MQ_PF) {
        if (s0 > b0 || !have_first) mq_pf<0>(pf, wl, dl, valid, s0, b1);
#pragma unroll
        for (int c = 0; c < MQ_PF / 4; ++c) {
            if (s0 + 4 * c < b1) {
                const mq_i4 x = *(const mq_i4 *) (xq_u + s0 + 4 * c);
                const mq_f4 dx = *(const mq_f4 *) (xd + s0 + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc = fmaf(__uint_as_float(pf[MQ_PF + 4 * c + e]) * dx[e], (float) __builtin_amdgcn_sdot4((int) pf[4 * c + e], x[e], 0, false), acc);
            }
        }
    }
    return acc;
}
__device__ __forceinline__ float mq_hsum8(float v) {
    v = v + dpp_f32<0x104>(v);          // row_shl:4  acc[l] + acc[l+4]
    v = v + dpp_f32<0x102>(v);          // row_shl:2
    return v + dpp_f32<0x101>(v);       // row_shl:1
}

// task / product of either kind behind one signature (Q: quantised rows; D = the matrix's block scales)
template <bool Q, int NS>
__device__ __forceinline__ mg_task mg_mk8(unsigned (&pf)[96], const wa_f16 * W, const float * D, const float * bias, const float * scale, int N, int K,
                                          int row0, int rows_wg, int grp, int lane) {
    if constexpr (!Q) { mg_task t = mg_task8<NS>(pf, W, bias, scale, N, K, row0, rows_wg, grp, lane); t.wl = nullptr; t.dl = nullptr; return t; }
    else {
        mg_task t;
        const int ri = grp * 8 + (lane >> 3), nb = K >> 5;
        t.row = row0 + ri;
        t.valid = ri < rows_wg && t.row < N;
        t.wrow = nullptr;
        t.wl = (const GAS int *) W + ((size_t) (t.valid ? t.row : 0) * 8 + (lane & 7)) * nb;
        t.dl = (const GAS float *) D + (size_t) (t.valid ? t.row : 0) * nb;
        t.bias = 0.0f; t.scale = 1.0f;
        mq_pf<NS>(pf, t.wl, t.dl, t.valid, 0, nb);
        if (t.valid && (lane & 7) == 0) { if (bias) t.bias = ((gcf) bias)[t.row]; if (scale) t.scale = ((gcf) scale)[t.row]; }
        return t;
    }
}
template <bool Q, int NS>
__device__ __forceinline__ float mg_do8(unsigned (&pf)[96], const mg_task & t, int nsteps, wa_f16 * xin, int lane) {
    if constexpr (!Q) return mg_dot8<NS>(pf, t.wrow, t.valid, nsteps, xin, lane & 7, true);
    else return mq_hsum8(mq_chain<NS>(pf, t.wl, t.dl, t.valid, 0, nsteps, (const int *) mq_xq(xin) + (lane & 7) * mq_ld(nsteps), mq_xd(xin), 0.0f, true));
}
#define MG_RES16(Q) ((Q) ? 8 : 0)       /* lane (of 16) that holds the result of a 16-lane row */
template <bool Q, int NS4>
__device__ __forceinline__ mg_task mg_mk16(unsigned (&pf)[96], const wa_f16 * W, const float * D, const float * bias, int N, int K, int row0, int rows_wg,
                                           int grp, int lane) {
    if constexpr (!Q) { mg_task t = mg_task16<NS4>(pf, W, bias, N, K, row0, rows_wg, grp, lane); t.wl = nullptr; t.dl = nullptr; return t; }
    else {
        mg_task t;
        const int ri = grp * 4 + (lane >> 4), nb = K >> 5, nbh = nb >> 1, half = (lane >> 3) & 1;
        t.row = row0 + ri;
        t.valid = ri < rows_wg && t.row < N;
        t.wrow = nullptr;
        t.wl = (const GAS int *) W + ((size_t) (t.valid ? t.row : 0) * 8 + (lane & 7)) * nb;
        t.dl = (const GAS float *) D + (size_t) (t.valid ? t.row : 0) * nb;
        t.bias = 0.0f; t.scale = 1.0f;
        mq_pf<0>(pf, t.wl, t.dl, t.valid, half * nbh, (half + 1) * nbh);
        if (t.valid && (lane & 15) == 8 && bias) t.bias = ((gcf) bias)[t.row];
        return t;
    }
}
template <bool Q, int NS4>
__device__ __forceinline__ float mg_do16(unsigned (&pf)[96], const mg_task & t, int nsteps, wa_f16 * xin, int lane) {
    if constexpr (!Q) return mg_dot16<NS4>(pf, t.wrow, t.valid, nsteps, xin, lane & 15, true);
    else {
        const int nbh = NS4 > 0 ? NS4 / 2 : nsteps >> 1, half = (lane >> 3) & 1;      // (known at compile time for ggml-small: no guards, the LDS reads in one batch)
        const int * xq_u = (const int *) mq_xq(xin) + (lane & 7) * mq_ld(nsteps);
        float acc = 0.0f;
        if (nbh <= MQ_PF) {
            // Everything but the chain itself for BOTH halves at once, in place (pf[j] = (float) dot4, pf[48 + j] = dw dx); what is left
            // in series is one fma per block: the lower half's chain, then the upper half's from its result.
#pragma unroll
            for (int c = 0; c < MQ_PF / 4; ++c) {
                if (4 * c < nbh) {
                    const mq_i4 x = *(const mq_i4 *) (xq_u + half * nbh + 4 * c);
                    const mq_f4 dx = *(const mq_f4 *) (mq_xd(xin) + half * nbh + 4 * c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pf[4 * c + e] = __float_as_uint((float) __builtin_amdgcn_sdot4((int) pf[4 * c + e], x[e], 0, false));
                        pf[MQ_PF + 4 * c + e] = __float_as_uint(__uint_as_float(pf[MQ_PF + 4 * c + e]) * dx[e]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < MQ_PF / 4; ++c)
                if (4 * c < nbh) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = fmaf(__uint_as_float(pf[MQ_PF + 4 * c + e]), __uint_as_float(pf[4 * c + e]), acc);
                }
            const float from_lower = dpp_f32<0x118>(acc);      // row_shr:8: lanes 8-15 take over the running sums of lanes 0-7
            acc = from_lower;                                  // (the lower half runs the second chain too; its result is not used)
#pragma unroll
            for (int c = 0; c < MQ_PF / 4; ++c)
                if (4 * c < nbh) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = fmaf(__uint_as_float(pf[MQ_PF + 4 * c + e]), __uint_as_float(pf[4 * c + e]), acc);
                }
            return mq_hsum8(acc);                              // valid in lane 8 of the 16
        }
        if (half == 0) acc = mq_chain<0>(pf, t.wl, t.dl, t.valid, 0, nbh, xq_u, mq_xd(xin), 0.0f, true);
        const float from_lower = dpp_f32<0x118>(acc);
        if (half == 1) acc = mq_chain<0>(pf, t.wl, t.dl, t.valid, nbh, nsteps, xq_u, mq_xd(xin), from_lower, true);
        return mq_hsum8(acc);                                  // valid in lane 8 of the 16
    }
}

// publish two F16 results (rows n, n+1 of lanes 16j and 16j+8) as one granule; returns the packed pair in lanes 16j
__device__ __forceinline__ unsigned mg_pub_h2(gu64 * edge, unsigned seq, bool valid, int n, unsigned h, int lane) {
    const unsigned hi = dpp_u32<0x108>(h);
    const unsigned pk = (h & 0xffffu) | (hi << 16);
    if (valid && (lane & 15) == 0) gr_store(edge + (n >> 1), seq, pk);
    return pk;
}

// -------------------------------------------------------------------------------------------------
// next-token prediction.  mg_pick (wave 0 of every workgroup, at the start): the input token of this launch - given, or
// (spec) merged from the candidate records the previous launch left - and the sampling state after it, into LDS.
// mg_final classifies every logit by that state and leaves this launch's records.  Device arithmetic here is a
// prediction only (fast exp, any order): the host re-derives every token from the logits with the reference's rules.
// -------------------------------------------------------------------------------------------------
#define MG_ATT_SMEM(maxkv) (8 * 8 + (maxkv) * 4 + ((maxkv) / 8) * 4 + 32 * 64 * 4 + 8 * 4 + 16 + (maxkv) * 2 + 64 * 2 + 64)      // LDS of the attention scratch (mg_att_carve)
// LDS of a GEMV workgroup: xf [d] f32 | xin [4d] f16 | LayerNorm partial sums | the F16 GELU table (vec.h:571-585; 128 KB: the
// FC1 epilogue's look-up is on the critical path of every layer, an L2 round trip there cost ~1 us)
#define MG_XINB_OFF  ((size_t) WA_MEGA_MAX_D * 4 + (size_t) 4 * WA_MEGA_MAX_D * 2)       // FC1's input: a GEMV-input area of its own (see mg_role_gemv)
#define MG_LNRED_OFF (MG_XINB_OFF + (size_t) 4 * WA_MEGA_MAX_D * 2)
#define MG_GELU_OFF  (MG_LNRED_OFF + 256)
#define MG_PICK_OFF  (MG_GELU_OFF + 131072)        // behind every role's LDS (the GEMV role's is the largest)
#define MG_PICK_BYTES 512
struct mg_best { float v; int i; };
__device__ __forceinline__ void mg_best_merge(mg_best & a, float v, int i) { if (v > a.v || (v == a.v && i < a.i)) { a.v = v; a.i = i; } }
__device__ __forceinline__ void mg_best_wave(mg_best & a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float v = __shfl_xor(a.v, o, WAVE); const int i = __shfl_xor(a.i, o, WAVE); mg_best_merge(a, v, i); }
}
__device__ __forceinline__ int mg_decide(const mg_best & bt, const mg_best & bs, float s_ts, int token_beg) {
    // whisper.cpp:6309-6333: timestamp mass above every text token => a timestamp; else the arg-max of everything allowed
    if (!(bs.v > -INFINITY)) return bt.v > -INFINITY ? bt.i : 0;
    if (!(bt.v > -INFINITY)) return bs.i;
    if (__logf(s_ts) + bs.v > bt.v) return bs.i;
    return bs.v > bt.v ? bs.i : bt.i;
}
__device__ __forceinline__ void mg_pick(mg_kargs A, int lane, int * pk) {
    int token = A->token, last = A->s_last, penult = A->s_penult, seek_delta = A->s_seek_delta, has_ts = A->s_has_ts;
    if (A->spec) {
        const GAS int * ps = (const GAS int *) A->ps_in;
        const GAS unsigned * rec = (const GAS unsigned *) A->rec_in;
        penult = ps[0]; seek_delta = ps[2]; has_ts = ps[3];
        mg_best bt = { -INFINITY, 0x7fffffff }, bs = { -INFINITY, 0x7fffffff };
        // every record (n_rec <= 256: four per lane) in ONE round of loads - a loop over them paid a cold global round trip per
        // iteration, 4.4 us at the head of every launch
        u32x4 ra[4]; unsigned rb[4];
        const int nr = A->n_rec;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = lane + 64 * j, gg = g < nr ? g : 0;
            ra[j] = *(const GAS u32x4 *) (rec + gg * 8); rb[j] = rec[gg * 8 + 4];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (lane + 64 * j < nr) {
            mg_best_merge(bt, __uint_as_float(ra[j].x), (int) ra[j].y);
            mg_best_merge(bs, __uint_as_float(ra[j].z), (int) ra[j].w);
        }
        mg_best_wave(bt); mg_best_wave(bs);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (lane + 64 * j < nr) {
            const float m = __uint_as_float(ra[j].z);
            if (m > -INFINITY) s += __uint_as_float(rb[j]) * __expf(m - bs.v);
        }
        s = wave_sum(s);
        token = mg_decide(bt, bs, s, A->token_beg);
        last = token;
        if (token > A->token_beg) { seek_delta = 2 * (token - A->token_beg); has_ts = 1; }
    }
    if (lane == 0) {
        pk[0] = token; pk[1] = last; pk[2] = penult; pk[3] = seek_delta; pk[4] = has_ts;
        if (blockIdx.x == 0) {
            GAS int * po = (GAS int *) A->ps_out;
            po[0] = last; po[1] = penult; po[2] = seek_delta; po[3] = has_ts;
            ((GAS int *) A->logits)[A->n_vocab + 1] = token;          // behind the logits and the status word: the token this launch decoded
        }
    }
}

// -------------------------------------------------------------------------------------------------
// final LayerNorm + logits = token_embedding . x (whisper.cpp:2820-2835): every workgroup, every wave
// -------------------------------------------------------------------------------------------------
template <int NS = 0, bool Q = false>
__device__ __forceinline__ void mg_prefetch_logits(mg_kargs A, unsigned (&pf)[96], bool & have_pf, int lane, int wave) {
    const int g = (int) blockIdx.x + (int) gridDim.x * wave;
    const int row = g * 8 + (lane >> 3);
    const bool valid = row < A->n_vocab;
    if constexpr (Q) {
        const int nb = A->d >> 5;
        mq_pf<NS>(pf, (const GAS int *) A->te + ((size_t) (valid ? row : 0) * 8 + (lane & 7)) * nb, (const GAS float *) A->te_d + (size_t) (valid ? row : 0) * nb, valid, 0, nb);
    } else mg_pf8<NS>(pf, (gch) A->te + (size_t) (valid ? row : 0) * A->d + 4 * (lane & 7), valid, A->d >> 5, 0);
    have_pf = true;
}
template <int NP3, int NS, bool Q = false>
__device__ __forceinline__ void mg_final(mg_kargs A, mg_ctl & c, unsigned char * smem, unsigned (&pf)[96], bool have_pf, const float (&gw)[NP3],
                                         const float (&gb)[NP3], int lane, int wave) {
    float  * xf  = (float *) smem;
    wa_f16 * xin = (wa_f16 *) (smem + WA_MEGA_MAX_D * 4);
    double * lnred = (double *) (smem + MG_LNRED_OFF);
    const int d = A->d, nwg = gridDim.x, wg = blockIdx.x, n_vocab = A->n_vocab;
    mg_ln3<NP3, Q>(A, c, A->n_layer > 0 ? mg_edge(A, A->n_layer - 1, E_X3) : nullptr, gw, gb, mg_slot(wave, MG_EX_FINAL), lane, xf, xin, lnred, 3000u,
                   blockIdx.x == 0 && wave == 0 ? (A->n_layer * 8) * 8 : -1, ((const int *) (smem + MG_PICK_OFF))[0]);
    const int NG = (n_vocab + 7) >> 3;
    GAS float * logits = (GAS float *) A->logits;
    // sampling state after this launch's token -> which logits the next pick may choose (whisper.cpp:6264-6302)
    const int * pk = (const int *) (smem + MG_PICK_OFF);
    const int beg = A->token_beg, eot = A->token_eot;
    const int st_last = pk[1], st_penult = pk[2], st_seek = pk[3], st_has = pk[4];
    const bool last_ts = st_last >= beg, penult_ts = st_penult < 0 || st_penult >= beg;
    const bool no_ts = last_ts && penult_ts, no_text = last_ts && !penult_ts;
    const int ts_min = st_has ? beg + st_seek / 2 : beg;
    const GAS unsigned * smask = (const GAS unsigned *) A->smask;
    mg_best bt = { -INFINITY, 0x7fffffff }, bs = { -INFINITY, 0x7fffffff };
    float s_ts = 0.0f;
    // row groups g(j) = wg + nwg (wave + 8 j) of this wave, two weight buffers: the loads of group j + 1 fly during group j
    unsigned pf2[96];
    const int ns = d >> 5;
    auto grp  = [&](int j) { return wg + nwg * (wave + MG_NW * j); };
    auto wrow = [&](int j) { const int row = grp(j) * 8 + (lane >> 3); return (gch) A->te + (size_t) (row < n_vocab ? row : 0) * d + 4 * (lane & 7); };
    auto qwl  = [&](int j) { const int row = grp(j) * 8 + (lane >> 3); return (const GAS int *) A->te + ((size_t) (row < n_vocab ? row : 0) * 8 + (lane & 7)) * ns; };
    auto qdl  = [&](int j) { const int row = grp(j) * 8 + (lane >> 3); return (const GAS float *) A->te_d + (size_t) (row < n_vocab ? row : 0) * ns; };
    auto vld  = [&](int j) { return grp(j) * 8 + (lane >> 3) < n_vocab; };
    auto load = [&](int j, unsigned (&buf)[96]) {
        if constexpr (Q) mq_pf<NS>(buf, qwl(j), qdl(j), vld(j), 0, ns); else mg_pf8<NS>(buf, wrow(j), vld(j), ns, 0);
    };
    auto one = [&](int j, unsigned (&buf)[96]) {
        const int row = grp(j) * 8 + (lane >> 3);
        const bool valid = row < n_vocab;
        const unsigned mw = valid && (lane & 7) == 0 ? smask[row >> 5] : 0xffffffffu;
        float r;
        if constexpr (Q) r = mq_hsum8(mq_chain<NS>(buf, qwl(j), qdl(j), valid, 0, ns, (const int *) mq_xq(xin) + (lane & 7) * mq_ld(ns), mq_xd(xin), 0.0f, ns <= MQ_PF));
        else r = mg_dot8<NS>(buf, wrow(j), valid, ns, xin, lane & 7, ns <= 48);
        if (valid && (lane & 7) == 0) {
            logits[row] = r;
            if (!((mw >> (row & 31)) & 1u)) {
                if (row >= beg) {
                    if (!no_ts && row >= ts_min) {
                        if (r > bs.v) { s_ts = s_ts * __expf(bs.v - r) + 1.0f; bs.v = r; bs.i = row; }
                        else s_ts += __expf(r - bs.v);
                    }
                } else if (!(no_text && row < eot)) mg_best_merge(bt, r, row);
            }
        }
    };
    if (!have_pf && grp(0) < NG) load(0, pf);
    for (int j = 0; grp(j) < NG; j += 2) {
        if (grp(j + 1) < NG) load(j + 1, pf2);
        one(j, pf);
        if (grp(j + 1) >= NG) break;
        if (grp(j + 2) < NG) load(j + 2, pf);
        one(j + 1, pf2);
    }
    {   // this workgroup's record
        unsigned * rb = (unsigned *) (smem + MG_PICK_OFF + 64);
        const float m_loc = bs.v;
        mg_best_wave(bt); mg_best_wave(bs);
        float sw = m_loc > -INFINITY ? s_ts * __expf(m_loc - bs.v) : 0.0f;
        sw = wave_sum(sw);
        if (lane == 0) { rb[wave * 8 + 0] = __float_as_uint(bt.v); rb[wave * 8 + 1] = (unsigned) bt.i; rb[wave * 8 + 2] = __float_as_uint(bs.v);
                         rb[wave * 8 + 3] = (unsigned) bs.i; rb[wave * 8 + 4] = __float_as_uint(sw); }
        mg_barrier();
        if (wave == 0) {
            mg_best t2 = { -INFINITY, 0x7fffffff }, s2 = { -INFINITY, 0x7fffffff };
            float sl = 0.0f, ml = -INFINITY;
            if (lane < MG_NW) { t2.v = __uint_as_float(rb[lane * 8 + 0]); t2.i = (int) rb[lane * 8 + 1]; s2.v = __uint_as_float(rb[lane * 8 + 2]); s2.i = (int) rb[lane * 8 + 3];
                                sl = __uint_as_float(rb[lane * 8 + 4]); ml = s2.v; }
            mg_best_wave(t2); mg_best_wave(s2);
            float sg = ml > -INFINITY ? sl * __expf(ml - s2.v) : 0.0f;
            sg = wave_sum(sg);
            if (lane == 0) {
                GAS unsigned * ro = (GAS unsigned *) A->rec_out + (size_t) wg * 8;
                ro[0] = __float_as_uint(t2.v); ro[1] = (unsigned) t2.i; ro[2] = __float_as_uint(s2.v); ro[3] = (unsigned) s2.i; ro[4] = __float_as_uint(sg);
            }
        }
    }
    mg_trace(A, blockIdx.x == 0 && wave == 0 && lane == 0, (A->n_layer * 8) * 8 + 3, mg_now());
    if (A->dbg) { mg_trace(A, blockIdx.x == 0 && wave == 0 && lane == 0, 3022, (unsigned) clock64()); mg_trace(A, blockIdx.x == 0 && wave == 0 && lane == 0, 3023, mg_now()); }
    // this launch ran: the host accepts a step only with its own number behind the logits (a launch that never started leaves the status word 0 too)
    if (blockIdx.x == 0 && wave == 0 && lane == 0) ((GAS unsigned *) A->logits)[A->n_vocab + 2] = c.seq;
}

// -------------------------------------------------------------------------------------------------
// role: GEMV workgroup.  wave 0 gathers (and normalises) the input of every phase; waves 1,2 own the QKV and FC1 rows,
// wave 3 the two out-projections, wave 4 the cross query, wave 5 FC2; waves 6,7 help gather the 4d-wide FC2 input.
// Every wave loads the weights of its NEXT task right after finishing the current one.
// -------------------------------------------------------------------------------------------------
template <int NP3, int NS, bool Q = false>
__device__ __forceinline__ void mg_role_gemv(mg_kargs A_, int idx_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const mg_kargs A = mg_uniform(A_);
    int lane = threadIdx.x & 63;
#define MG_FRESH() do { lane = mq_fresh(lane); } while (0)      /* see mq_fresh: nothing lane-derived lives across phases */
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = __builtin_amdgcn_readfirstlane(idx_), nG = (int) gridDim.x - 5 * A->n_head;
    mg_ctl c; c.status = (gu32 *) A->status; c.seq = A->seq; c.dead = false;
    mg_chaos_start(c.seq);
    mg_trace(A, wg == 0 && wave == 0 && (threadIdx.x & 63) == 0, (A->n_layer * 8) * 8 + 6, mg_now());        // entry
    if (A->dbg) { mg_trace(A, wg == 0 && wave == 0 && (threadIdx.x & 63) == 0, 3020, (unsigned) clock64()); mg_trace(A, wg == 0 && wave == 0 && (threadIdx.x & 63) == 0, 3021, mg_now()); }
    unsigned pf[96];
    bool have_pf = false;
    constexpr bool BIG = NP3 == MG_NP3;       // d > 768: more row groups per workgroup than prefetching waves - the idle waves assist
    // Quantised, d <= 768: the first MLP product is owned by WHOLE Q8_0 blocks - workgroup b < 4d / 32 has rows 32 b .. 32 b + 31, eight per
    // wave 1..4 -, so that the block leaves already quantised (8 quads + scale: 9 granules instead of 32 F32 values, and no consumer
    // quantises it again).  The four waves meet over an LDS counter; the last one to arrive quantises and publishes.
    constexpr bool QB = Q && !BIG && MG_DEFER;
    // Wide models (d > 768), second form of the schedule: FC1 has a prefetched owner for every group (waves 1, 2 | 3, 4 after their P6 products |
    // 6, 7, which hold no logits rows until the last layer), and waves 3, 4 ask for the first half of their FC2 rows right after FC1 - an assist
    // that fetches on demand exposes an HBM round trip per group (P7 5.3 -> ~1.5 us, P8 5.9 -> ~4 us per layer on large-v3).
    constexpr bool BIGP = BIG && MG_DEFER;
    // (fc1x / fc1cnt sit in the xin area at byte 7 MAX_D = 8960 .. 9092.  The whole-block form runs for d <= 768 only (QB = Q && !BIG): the widest thing
    //  ever written into xin there is the FC2 operand - quants [4d <= 3072] bytes at 0, block scales [4d / 32 <= 96] floats at 6 MAX_D = 7680 .. 8064 -, so
    //  no gather reaches byte 8960; the counter is never reset, only compared modulo 4.)
    float    * fc1x   = (float *) (smem + WA_MEGA_MAX_D * 4 + 7 * WA_MEGA_MAX_D);           // [32] the block's GELU outputs (behind xq / xd in the xin area)
    unsigned * fc1cnt = (unsigned *) (smem + WA_MEGA_MAX_D * 4 + 7 * WA_MEGA_MAX_D + 128);  // arrivals, never reset: a multiple of 4 after every layer

    float  * xf  = (float *) smem;                               // [d]   residual row (F32)
    wa_f16 * xin = (wa_f16 *) (smem + WA_MEGA_MAX_D * 4);        // [4d]  GEMV input (F16)
    // The LayerNorms write THEIR outputs (the inputs of q|k|v, the cross query and FC1) into an area of their own; xin takes the gathered
    // inputs (attention outputs, FC1's output).  A gather follows the previous product WITHOUT a barrier (its waves poll while the product
    // runs) and a gather wave writes as soon as ITS granules are in, which says nothing about this workgroup's own product waves: with one
    // area the only protection was timing (other workgroups' results take > 1 us to arrive, workgroups run within ~0.5 us of each other;
    // 3-6 us where an attention phase sits in between).  Now every writer of an area is separated from its previous readers by the
    // barriers of a LayerNorm.
    wa_f16 * xinB = (wa_f16 *) (smem + MG_XINB_OFF);
    const int d = A->d, L = A->n_layer, d4 = 4 * d;
    const int r_qkv = mg_rpw(3 * d, nG), r_d = mg_rpw(d, nG), r_ff = mg_rpw(d4, nG);
    const int row_qkv = wg * r_qkv, row_d = wg * r_d, row_ff = wg * r_ff;
    const int g_qkv = (r_qkv + 7) >> 3, g_ff = (r_ff + 7) >> 3, g_d8 = (r_d + 7) >> 3, g_d16 = (r_d + 3) >> 2;
    // the layer table is read-only for the launch: constant address space = scalar loads (a vector load of a pointer costs a
    // drained vmcnt in front of the barrier: measured 3 us per LayerNorm phase)
    const __attribute__((address_space(4))) wa_mega_layer * Ly = (const __attribute__((address_space(4))) wa_mega_layer *) A->layers;
    const unsigned seq = c.seq;
    const int kv_head = A->kv_head;

    double * lnred = (double *) (smem + MG_LNRED_OFF);
    wa_f16 * gelu_l = (wa_f16 *) (smem + MG_GELU_OFF);
    float gw[NP3], gb[NP3];          // gather waves: gamma / beta of their part of the next LayerNorm
    int * pk = (int *) (smem + MG_PICK_OFF);
    mg_ln_params<NP3>(gw, gb, Ly[0].ln1_w, Ly[0].ln1_b, d, mg_slot(wave, MG_EX_P1), lane);
    if (wave == 0) mg_pick(A, lane, pk);
    if (QB && wave == 0 && lane == 0) *fc1cnt = 0u;
    mg_task t; t.valid = false; t.row = 0; t.wrow = nullptr; t.bias = 0.f; t.scale = 1.f;
    if (wave == 1 || wave == 2) t = mg_mk8<Q, NS>(pf, Ly[0].qkv_w, Ly[0].qkv_d, Ly[0].qkv_b, Ly[0].qkv_s, 3 * d, d, row_qkv, r_qkv, wave - 1, lane);
    else if (wave == 3)         t = mg_mk8<Q, NS>(pf, Ly[0].out_w, Ly[0].out_d, Ly[0].out_b, nullptr, d, d, row_d, r_d, 0, lane);
    else if (wave == 4)         t = mg_mk8<Q, NS>(pf, Ly[0].cq_w, Ly[0].cq_d, Ly[0].cq_b, nullptr, d, d, row_d, r_d, 0, lane);
    else if (wave == 5)         t = mg_mk16<Q, 4 * NS>(pf, Ly[0].fc2_w, Ly[0].fc2_d, Ly[0].fc2_b, d, d4, row_d, r_d, 0, lane);
    else if (wave >= 6 && !(BIGP && L > 0)) mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);      // held until the final phase (wide form: they own FC1 rows, the logits rows come in the last layer)
    mg_barrier();                   // the picked token is in LDS for the three embedding waves
    mg_trace(A, wg == 0 && wave == 0 && lane == 0, (A->n_layer * 8) * 8 + 7, mg_now());
    if (wave >= 3 && wave <= 5) {   // GELU table -> LDS by LDS-DMA (no registers, nothing waits here); first needed by FC1 of layer 0
        const GAS u32x4 * src = (const GAS u32x4 *) A->gelu;
        for (int j = wave - 3; j < 128; j += 3)         // 128 wave-instructions of 64 x 16 bytes
            __builtin_amdgcn_global_load_lds((const GAS void *) (src + j * 64 + lane), (__attribute__((address_space(3))) void *) (smem + MG_GELU_OFF + (size_t) j * 1024), 16, 0, 0);
    }

    for (int l = 0; l < L; ++l) {
        const __attribute__((address_space(4))) wa_mega_layer & Y = Ly[l];
        // (wide form) what wave 3 / 4 asks for once its FC1 rows are used: the first half of its FC2 rows, or - no such group - the next layer's rows
        auto big_next = [&](int w) {
            if (w - 2 < g_d16) t = mg_mk16<Q, 4 * NS>(pf, Y.fc2_w, Y.fc2_d, Y.fc2_b, d, d4, row_d, r_d, w - 2, lane);
            else if (l + 1 >= L) mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, w);
            else t = w == 3 ? mg_mk8<Q, NS>(pf, Ly[l + 1].out_w, Ly[l + 1].out_d, Ly[l + 1].out_b, nullptr, d, d, row_d, r_d, 0, lane)
                            : mg_mk8<Q, NS>(pf, Ly[l + 1].cq_w, Ly[l + 1].cq_d, Ly[l + 1].cq_b, nullptr, d, d, row_d, r_d, 0, lane);
        };
        // ---------------- P1: LayerNorm + q|k|v ----------------
        MG_FRESH();
        mg_ln3<NP3, Q>(A, c, l == 0 ? nullptr : mg_edge(A, l - 1, E_X3), gw, gb, mg_slot(wave, MG_EX_P1), lane, xf, xinB, lnred, 100u + l, wg == 0 && wave == 0 ? (l * 8 + 0) * 8 : -1, pk[0]);
        MG_FRESH();
        mg_ln_params<NP3>(gw, gb, Y.ln2_w, Y.ln2_b, d, mg_slot(wave, MG_EX_P4), lane);
        if (MG_DEFER && l > 0 && wave == 5) t = mg_mk16<Q, 4 * NS>(pf, Y.fc2_w, Y.fc2_d, Y.fc2_b, d, d4, row_d, r_d, 0, lane);      // deferred from the previous layer's P8
        // (Wide models give a workgroup more row groups than the one per wave that is prefetched.  The waves idle in a phase then
        //  assist: they take the extra groups on demand - overwriting the rows they hold for a later phase - and fetch those again
        //  afterwards, long before that phase.  ggml-small and below: one group per wave, nothing changes.)
        if (wave == 1 || wave == 2 || (BIG && (wave == 3 || wave == 4))) {
            const bool own = !BIG || wave <= 2;
            bool assisted = false;
            gu64 * eq = mg_edge(A, l, E_QKV);
            for (int grp = wave - 1; grp < g_qkv; grp += BIG ? 4 : 2) {
                if (grp >= 2) { t = mg_mk8<Q, NS>(pf, Y.qkv_w, Y.qkv_d, Y.qkv_b, Y.qkv_s, 3 * d, d, row_qkv, r_qkv, grp, lane); assisted = true; }
                MG_CHAOS_AT(1u);
                float v = mg_do8<Q, NS>(pf, t, d >> 5, xinB, lane);
                v = v + t.bias;
                v = v * t.scale;
                const unsigned pk = mg_pub_h2(eq, seq, t.valid, t.row, (unsigned) f2h(v), lane);
                if (t.valid && (lane & 15) == 0 && t.row >= d) {      // new key / value also go to their KV cell for later tokens
                    GAS wa_f16 * cell = (t.row < 2 * d ? (GAS wa_f16 *) A->kv_k + (t.row - d) : (GAS wa_f16 *) A->kv_v + (t.row - 2 * d)) +
                                        (size_t) l * A->kv_layer_stride + (size_t) kv_head * d;
                    *(GAS unsigned *) cell = pk;
                }
            }
            mg_trace(A, wg == 0 && wave == 1 && lane == 0, (l * 8 + 0) * 8 + 3, mg_now());
            if (own && !MG_DEFER) t = mg_mk8<Q, NS>(pf, Y.fc1_w, Y.fc1_d, Y.fc1_b, nullptr, d4, d, row_ff, r_ff, wave - 1, lane);
            else if (assisted) t = wave == 3 ? mg_mk8<Q, NS>(pf, Y.out_w, Y.out_d, Y.out_b, nullptr, d, d, row_d, r_d, 0, lane)
                                             : mg_mk8<Q, NS>(pf, Y.cq_w, Y.cq_d, Y.cq_b, nullptr, d, d, row_d, r_d, 0, lane);
        }
        // ---------------- P3: self-attention out-projection + residual ----------------
        MG_FRESH();
        {
            const int qs = mg_slot(wave, MG_EX_AO), sg = mg_seg(Q ? 9 * (d >> 5) : d >> 1), j0 = qs * sg, j1 = min(Q ? 9 * (d >> 5) : d >> 1, j0 + sg);
            if constexpr (Q) { if (qs >= 0) mg_gather_qb<1>(c, mg_edge(A, l, E_AO), j0, j1, lane, xin, d >> 5, 200u + l, A, wg == 0 && wave == 0 ? (l * 8 + 1) * 8 : -1); }
            else if (qs >= 0) mg_gather_h2<2>(c, mg_edge(A, l, E_AO), j0, j1, lane, (unsigned *) xin, 200u + l, A, wg == 0 && wave == 0 ? (l * 8 + 1) * 8 : -1);
        }
        mg_barrier();
        MG_FRESH();
        // (MG_DEFER: a wave's next weights are requested only once the CU's NEXT gather is over - a poll queued behind 24-48 KB of
        //  weight loads waits for them: hand-off-1to1 costs 0.8 us with quiet endpoints, 2.3-3.5 behind 8-15 streaming waves.)
        if (MG_DEFER && (wave == 1 || wave == 2)) t = mg_mk8<Q, NS>(pf, Y.fc1_w, Y.fc1_d, Y.fc1_b, nullptr, d4, d, QB ? 32 * wg : row_ff, QB ? 32 : r_ff, wave - 1, lane);
        if (wave == 3 || (BIG && wave == 5)) {        // wave 5 assists (it holds this layer's FC2 rows, next needed in P8)
            const bool own = !BIG || wave == 3;
            bool assisted = false;
            gu64 * ex = mg_edge(A, l, E_X1);
            for (int grp = own ? 0 : 1; grp < g_d8; grp += BIG ? 2 : 1) {
                if (grp >= 1) { t = mg_mk8<Q, NS>(pf, Y.out_w, Y.out_d, Y.out_b, nullptr, d, d, row_d, r_d, grp, lane); assisted = true; }
                MG_CHAOS_AT(2u);
                float v = mg_do8<Q, NS>(pf, t, d >> 5, xin, lane);
                v = v + t.bias;
                if (t.valid && (lane & 7) == 0) gr_store(ex + t.row, seq, __float_as_uint(v + xf[t.row]));
            }
            mg_trace(A, wg == 0 && own && lane == 0, (l * 8 + 1) * 8 + 3, mg_now());
            if (A->dbg && l == MG_WGTRACE_LAYER) mg_trace(A, own && lane == 0, 1024 + wg * 8 + 3, mg_now());
            if (own && !MG_DEFER) t = mg_mk8<Q, NS>(pf, Y.co_w, Y.co_d, Y.co_b, nullptr, d, d, row_d, r_d, 0, lane);
            else if (assisted) t = mg_mk16<Q, 4 * NS>(pf, Y.fc2_w, Y.fc2_d, Y.fc2_b, d, d4, row_d, r_d, 0, lane);
        }
        // ---------------- P4: LayerNorm + cross query ----------------
        MG_FRESH();
        // (debug: at layer MG_WGTRACE_LAYER every workgroup stamps this phase - the spread over workgroups is what a hand-off waits for)
        mg_ln3<NP3, Q>(A, c, mg_edge(A, l, E_X1), gw, gb, mg_slot(wave, MG_EX_P4), lane, xf, xinB, lnred, 300u + l,
                       wave == 0 ? (wg == 0 ? (l * 8 + 2) * 8 : (A->dbg && l == MG_WGTRACE_LAYER ? 1024 + wg * 8 : -1)) : -1);
        MG_FRESH();
        if (MG_DEFER && wave == 3) t = mg_mk8<Q, NS>(pf, Y.co_w, Y.co_d, Y.co_b, nullptr, d, d, row_d, r_d, 0, lane);
        if (BIGP && wave >= 6 && wave - 2 < g_ff) t = mg_mk8<Q, NS>(pf, Y.fc1_w, Y.fc1_d, Y.fc1_b, nullptr, d4, d, row_ff, r_ff, wave - 2, lane);      // FC1 groups 4, 5 (the next gather is a cross-attention away)
        mg_ln_params<NP3>(gw, gb, Y.ln3_w, Y.ln3_b, d, mg_slot(wave, MG_EX_P7), lane);
        if (wave == 4 || (BIG && wave == 3)) {        // wave 3 assists (it holds this layer's cross-attention output rows, next needed in P6)
            const bool own = !BIG || wave == 4;
            bool assisted = false;
            gu64 * eq = mg_edge(A, l, E_QC);
            for (int grp = own ? 0 : 1; grp < g_d8; grp += BIG ? 2 : 1) {
                if (grp >= 1) { t = mg_mk8<Q, NS>(pf, Y.cq_w, Y.cq_d, Y.cq_b, nullptr, d, d, row_d, r_d, grp, lane); assisted = true; }
                MG_CHAOS_AT(3u);
                float v = mg_do8<Q, NS>(pf, t, d >> 5, xinB, lane);
                v = v + t.bias;
                mg_pub_h2(eq, seq, t.valid, t.row, (unsigned) f2h(v), lane);
            }
            mg_trace(A, wg == 0 && own && lane == 0, (l * 8 + 2) * 8 + 3, mg_now());
            if (A->dbg && l == MG_WGTRACE_LAYER) mg_trace(A, own && lane == 0, 1024 + wg * 8 + 6, mg_now());
            if (QB) t = mg_mk8<Q, NS>(pf, Y.fc1_w, Y.fc1_d, Y.fc1_b, nullptr, d4, d, 32 * wg, 32, 3, lane);      // its eight rows of the block; next needed in P7
            else if (own && !MG_DEFER) {
                if (l + 1 < L) t = mg_mk8<Q, NS>(pf, Ly[l + 1].cq_w, Ly[l + 1].cq_d, Ly[l + 1].cq_b, nullptr, d, d, row_d, r_d, 0, lane);
                else mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
            } else if (assisted) t = mg_mk8<Q, NS>(pf, Y.co_w, Y.co_d, Y.co_b, nullptr, d, d, row_d, r_d, 0, lane);
        }
        // ---------------- P6: cross-attention out-projection + residual ----------------
        MG_FRESH();
        {
            const int qs = mg_slot(wave, MG_EX_AO2), sg = mg_seg(Q ? 9 * (d >> 5) : d >> 1), j0 = qs * sg, j1 = min(Q ? 9 * (d >> 5) : d >> 1, j0 + sg);
            if constexpr (Q) { if (qs >= 0) mg_gather_qb<1>(c, mg_edge(A, l, E_AO2), j0, j1, lane, xin, d >> 5, 400u + l, A, wg == 0 && wave == 0 ? (l * 8 + 3) * 8 : -1); }
            else if (qs >= 0) mg_gather_h2<2>(c, mg_edge(A, l, E_AO2), j0, j1, lane, (unsigned *) xin, 400u + l, A, wg == 0 && wave == 0 ? (l * 8 + 3) * 8 : -1);
        }
        mg_barrier();
        MG_FRESH();
        if (MG_DEFER && !QB && !BIGP && wave == 4) {
            if (l + 1 < L) t = mg_mk8<Q, NS>(pf, Ly[l + 1].cq_w, Ly[l + 1].cq_d, Ly[l + 1].cq_b, nullptr, d, d, row_d, r_d, 0, lane);
            else mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
        }
        if (wave == 3 || (BIG && wave == 4)) {        // wave 4 assists (it holds the next layer's cross-query rows)
            const bool own = !BIG || wave == 3;
            bool assisted = false;
            gu64 * ex = mg_edge(A, l, E_X2);
            for (int grp = own ? 0 : 1; grp < g_d8; grp += BIG ? 2 : 1) {
                if (grp >= 1) { t = mg_mk8<Q, NS>(pf, Y.co_w, Y.co_d, Y.co_b, nullptr, d, d, row_d, r_d, grp, lane); assisted = true; }
                MG_CHAOS_AT(4u);
                float v = mg_do8<Q, NS>(pf, t, d >> 5, xin, lane);
                v = v + t.bias;
                if (t.valid && (lane & 7) == 0) gr_store(ex + t.row, seq, __float_as_uint(v + xf[t.row]));
            }
            mg_trace(A, wg == 0 && own && lane == 0, (l * 8 + 3) * 8 + 3, mg_now());
            if (A->dbg && l == MG_WGTRACE_LAYER) mg_trace(A, own && lane == 0, 4096 + wg * 8 + 3, mg_now());
            if (QB) t = mg_mk8<Q, NS>(pf, Y.fc1_w, Y.fc1_d, Y.fc1_b, nullptr, d4, d, 32 * wg, 32, 2, lane);
            else if (BIGP) {                 // FC1 group 2 (wave 3) / 3 (wave 4), used in P7
                if (wave - 1 < g_ff) t = mg_mk8<Q, NS>(pf, Y.fc1_w, Y.fc1_d, Y.fc1_b, nullptr, d4, d, row_ff, r_ff, wave - 1, lane);
                else big_next(wave);
            } else if (own && !MG_DEFER) {
                if (l + 1 < L) t = mg_mk8<Q, NS>(pf, Ly[l + 1].out_w, Ly[l + 1].out_d, Ly[l + 1].out_b, nullptr, d, d, row_d, r_d, 0, lane);
                else mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
            } else if (assisted) {
                if (l + 1 < L) t = mg_mk8<Q, NS>(pf, Ly[l + 1].cq_w, Ly[l + 1].cq_d, Ly[l + 1].cq_b, nullptr, d, d, row_d, r_d, 0, lane);
                else mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
            }
        }
        // ---------------- P7: LayerNorm + FC1 + GELU ----------------
        MG_FRESH();
        if (l == 0 && wave >= 3 && wave <= 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the GELU table has landed (barriers below publish it)
        mg_ln3<NP3, Q>(A, c, mg_edge(A, l, E_X2), gw, gb, mg_slot(wave, MG_EX_P7), lane, xf, xinB, lnred, 500u + l,
                       wave == 0 ? (wg == 0 ? (l * 8 + 4) * 8 : (A->dbg && l == MG_WGTRACE_LAYER ? 4096 + wg * 8 : -1)) : -1);
        MG_FRESH();
        if (MG_DEFER && !QB && !BIGP && wave == 3) {
            if (l + 1 < L) t = mg_mk8<Q, NS>(pf, Ly[l + 1].out_w, Ly[l + 1].out_d, Ly[l + 1].out_b, nullptr, d, d, row_d, r_d, 0, lane);
            else mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
        }
        if (l + 1 < L) mg_ln_params<NP3>(gw, gb, Ly[l + 1].ln1_w, Ly[l + 1].ln1_b, d, mg_slot(wave, MG_EX_P1), lane);
        else           mg_ln_params<NP3>(gw, gb, A->lnf_w, A->lnf_b, d, mg_slot(wave, MG_EX_FINAL), lane);
        if (QB) {
            if (wave >= 1 && wave <= 4) {
                MG_CHAOS_AT(5u);
                float v = mg_do8<Q, NS>(pf, t, d >> 5, xinB, lane);
                v = v + t.bias;
                float gl = v;                                      // wa_gelu (vec.h:571-585) through the F16 table (LDS copy)
                if (v <= -10.0f) gl = 0.0f; else if (v < 10.0f) gl = h2f(gelu_l[t.valid ? f2h(v) : 0]);
                if ((lane & 7) == 0) fc1x[8 * (wave - 1) + (lane >> 3)] = gl;
                MG_CHAOS_AT(51u);
                unsigned arrived = 0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (LDS operations of a wave execute in order: the values are in place before the count moves)
                if (lane == 0) arrived = __hip_atomic_fetch_add(fc1cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                arrived = (unsigned) __builtin_amdgcn_readfirstlane((int) arrived);
                asm volatile("" ::: "memory");
                if ((arrived & 3u) == 3u && 32 * wg < d4) {               // all 32 values are there: quantize_row_q8_0 of the block (as mq_put), by lanes 0..31
                    MG_CHAOS_AT(52u);
                    const float y = fc1x[lane & 31];
                    float a = fabsf(y);
                    a = fmaxf(a, dpp_f32<0x128>(a)); a = fmaxf(a, dpp_f32<0x124>(a)); a = fmaxf(a, dpp_f32<0x122>(a)); a = fmaxf(a, dpp_f32<0x121>(a));
                    a = fmaxf(a, __shfl_xor(a, 16, 32));
                    const float dq = a / 127.f, id = a != 0.0f ? 127.f / a : 0.0f;
                    const unsigned q = (unsigned) (int) rintf(y * id) & 0xffu;
                    const unsigned w = q | (dpp_u32<0x101>(q) << 8) | (dpp_u32<0x102>(q) << 16) | (dpp_u32<0x103>(q) << 24);
                    gu64 * eb = mg_edge(A, l, E_HF) + (size_t) wg * 9;
                    if (lane < 32 && (lane & 3) == 0) gr_store(eb + (lane >> 2), seq, w);
                    if (lane == 0) gr_store(eb + 8, seq, __float_as_uint(h2f(f2h(dq))));
                    mg_trace(A, wg == 0 && lane == 0, (l * 8 + 4) * 8 + 3, mg_now());
                }
                if (wave == 3) {
                    if (l + 1 < L) t = mg_mk8<Q, NS>(pf, Ly[l + 1].out_w, Ly[l + 1].out_d, Ly[l + 1].out_b, nullptr, d, d, row_d, r_d, 0, lane);
                    else mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
                } else if (wave == 4) {
                    if (l + 1 < L) t = mg_mk8<Q, NS>(pf, Ly[l + 1].cq_w, Ly[l + 1].cq_d, Ly[l + 1].cq_b, nullptr, d, d, row_d, r_d, 0, lane);
                    else mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
                }
            }
        } else if (BIGP) {
            if ((wave >= 1 && wave <= 4) || wave >= 6) {
                gu64 * eh = mg_edge(A, l, E_HF);
                for (int grp = wave <= 4 ? wave - 1 : wave - 2; grp < g_ff; grp += 6) {
                    if (grp >= 6) t = mg_mk8<Q, NS>(pf, Y.fc1_w, Y.fc1_d, Y.fc1_b, nullptr, d4, d, row_ff, r_ff, grp, lane);      // (no released shape has more than six groups)
                    MG_CHAOS_AT(6u);
                    float v = mg_do8<Q, NS>(pf, t, d >> 5, xinB, lane);
                    v = v + t.bias;
                    float gl = v;                                  // wa_gelu (vec.h:571-585) through the F16 table (LDS copy)
                    if (v <= -10.0f) gl = 0.0f; else if (v < 10.0f) gl = h2f(gelu_l[t.valid ? f2h(v) : 0]);
                    if constexpr (Q) { if (t.valid && (lane & 7) == 0) gr_store(eh + t.row, seq, __float_as_uint(gl)); }
                    else mg_pub_h2(eh, seq, t.valid, t.row, (unsigned) f2h(gl), lane);
                }
                mg_trace(A, wg == 0 && wave == 1 && lane == 0, (l * 8 + 4) * 8 + 3, mg_now());
                if ((wave == 3 || wave == 4) && wave - 1 < g_ff) big_next(wave);       // (without an FC1 group it asked in P6 already)
                else if (wave >= 6 && l + 1 >= L) mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
            }
        } else
        if (wave == 1 || wave == 2 || (BIG && (wave == 3 || wave == 4))) {        // waves 3, 4 assist (they hold the next layer's out-projection / cross-query rows)
            const bool own = !BIG || wave <= 2;
            bool assisted = false;
            gu64 * eh = mg_edge(A, l, E_HF);
            for (int grp = wave - 1; grp < g_ff; grp += BIG ? 4 : 2) {
                if (grp >= 2) { t = mg_mk8<Q, NS>(pf, Y.fc1_w, Y.fc1_d, Y.fc1_b, nullptr, d4, d, row_ff, r_ff, grp, lane); assisted = true; }
                MG_CHAOS_AT(7u);
                float v = mg_do8<Q, NS>(pf, t, d >> 5, xinB, lane);
                v = v + t.bias;
                float gl = v;                                      // wa_gelu (vec.h:571-585) through the F16 table (LDS copy)
                if (v <= -10.0f) gl = 0.0f; else if (v < 10.0f) gl = h2f(gelu_l[t.valid ? f2h(v) : 0]);
                if constexpr (Q) { if (t.valid && (lane & 7) == 0) gr_store(eh + t.row, seq, __float_as_uint(gl)); }      // F32: the second MLP product quantises from it
                else mg_pub_h2(eh, seq, t.valid, t.row, (unsigned) f2h(gl), lane);
            }
            mg_trace(A, wg == 0 && wave == 1 && lane == 0, (l * 8 + 4) * 8 + 3, mg_now());
            if (A->dbg && l == MG_WGTRACE_LAYER) mg_trace(A, wave == 1 && lane == 0, 4096 + wg * 8 + 6, mg_now());
            if (own && !MG_DEFER) {
                if (l + 1 < L) t = mg_mk8<Q, NS>(pf, Ly[l + 1].qkv_w, Ly[l + 1].qkv_d, Ly[l + 1].qkv_b, Ly[l + 1].qkv_s, 3 * d, d, row_qkv, r_qkv, wave - 1, lane);
                else mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
            } else if (assisted) {
                if (l + 1 >= L) mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
                else t = wave == 3 ? mg_mk8<Q, NS>(pf, Ly[l + 1].out_w, Ly[l + 1].out_d, Ly[l + 1].out_b, nullptr, d, d, row_d, r_d, 0, lane)
                                   : mg_mk8<Q, NS>(pf, Ly[l + 1].cq_w, Ly[l + 1].cq_d, Ly[l + 1].cq_b, nullptr, d, d, row_d, r_d, 0, lane);
            }
        }
        // ---------------- P8: FC2 + residual ----------------
        MG_FRESH();
        {   // the widest hand-off (2d granules)
            const int ng = QB ? 9 * (d4 >> 5) : Q ? d4 : 2 * d;
            const int qs = mg_slot(wave, QB || BIGP ? MG_EX_HFQ : MG_EX_HF), sg = mg_seg(ng), i0 = qs * sg, i1 = min(ng, i0 + sg);
            if constexpr (QB) { if (qs >= 0) mg_gather_qb<3>(c, mg_edge(A, l, E_HF), i0, i1, lane, xin, d4 >> 5, 600u + l, A, wg == 0 && wave == 0 ? (l * 8 + 5) * 8 : -1); }
            else if constexpr (Q) { if (qs >= 0) mg_gather_q8<(NP3 == MG_NP3 ? 14 : 8)>(c, mg_edge(A, l, E_HF), i0, i1, lane, xin, d4 >> 5, 600u + l, A, wg == 0 && wave == 0 ? (l * 8 + 5) * 8 : -1); }
            else if (qs >= 0) mg_gather_h2<7>(c, mg_edge(A, l, E_HF), i0, i1, lane, (unsigned *) xin, 600u + l, A, wg == 0 && wave == 0 ? (l * 8 + 5) * 8 : -1);
        }
        mg_barrier();
        MG_FRESH();
        mg_trace(A, wg == 0 && wave == 5 && lane == 0, (l * 8 + 5) * 8 + 4, mg_now());
        if (MG_DEFER && (wave == 1 || wave == 2)) {
            if (l + 1 < L) t = mg_mk8<Q, NS>(pf, Ly[l + 1].qkv_w, Ly[l + 1].qkv_d, Ly[l + 1].qkv_b, Ly[l + 1].qkv_s, 3 * d, d, row_qkv, r_qkv, wave - 1, lane);
            else mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
        }
        if (wave == 5 || (BIG && (wave == 3 || wave == 4))) {        // waves 3, 4 assist (as in P7)
            const bool own = !BIG || wave == 5;
            bool assisted = false;
            gu64 * ex = mg_edge(A, l, E_X3);
            for (int grp = own ? 0 : wave - 2; grp < g_d16; grp += BIG ? 3 : 1) {
                if (grp >= (BIGP ? 3 : 1)) t = mg_mk16<Q, 4 * NS>(pf, Y.fc2_w, Y.fc2_d, Y.fc2_b, d, d4, row_d, r_d, grp, lane);      // (wide form: groups 1, 2 were asked for after FC1)
                if (grp >= 1) assisted = true;
                MG_CHAOS_AT(8u);
                float v = mg_do16<Q, 4 * NS>(pf, t, d4 >> 5, xin, lane);
                v = v + t.bias;
                if (t.valid && (lane & 15) == MG_RES16(Q)) gr_store(ex + t.row, seq, __float_as_uint(v + xf[t.row]));
            }
            mg_trace(A, wg == 0 && own && lane == 0, (l * 8 + 5) * 8 + 3, mg_now());
            if (own && (!MG_DEFER || l + 1 >= L)) {
                if (l + 1 < L) t = mg_mk16<Q, 4 * NS>(pf, Ly[l + 1].fc2_w, Ly[l + 1].fc2_d, Ly[l + 1].fc2_b, d, d4, row_d, r_d, 0, lane);
                else mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
            } else if (assisted) {
                if (l + 1 >= L) mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
                else t = wave == 3 ? mg_mk8<Q, NS>(pf, Ly[l + 1].out_w, Ly[l + 1].out_d, Ly[l + 1].out_b, nullptr, d, d, row_d, r_d, 0, lane)
                                   : mg_mk8<Q, NS>(pf, Ly[l + 1].cq_w, Ly[l + 1].cq_d, Ly[l + 1].cq_b, nullptr, d, d, row_d, r_d, 0, lane);
            }
        }
    }
    if (L == 0 && wave >= 1 && wave <= 5) mg_prefetch_logits<NS, Q>(A, pf, have_pf, lane, wave);
    MG_FRESH();
    mg_final<NP3, NS, Q>(A, c, smem, pf, have_pf, gw, gb, lane, wave);
#undef MG_FRESH
}

// -------------------------------------------------------------------------------------------------
// soft_max over sc[0..n) exactly as ops.cpp:4792-4818 + vec.cpp:257-308 (k_attn_exact); leaves F16 probabilities in p16
// -------------------------------------------------------------------------------------------------
struct mg_att_smem { float * sc; wa_f16 * p16; float * gs; float * red; double * redd; float * s_inv; wa_f16 * qs; float * part; };
__device__ __forceinline__ mg_att_smem mg_att_carve(unsigned char * base, int maxkv) {
    mg_att_smem m;
    m.redd = (double *) base;                       base += 8 * sizeof(double);
    m.sc   = (float *) base;                        base += (size_t) maxkv * 4;
    m.gs   = (float *) base;                        base += (size_t) (maxkv / 8) * 4;
    m.part = (float *) base;                        base += 32 * 64 * 4;
    m.red  = (float *) base;                        base += 8 * 4;
    m.s_inv = (float *) base;                       base += 4 * 4;
    m.p16  = (wa_f16 *) base;                       base += (size_t) maxkv * 2;
    m.qs   = (wa_f16 *) base;
    return m;
}


// one key's score from its two 16-byte pieces (lane a of the key's 4-lane group): k_attn_exact's arithmetic
__device__ __forceinline__ float mg_score(const u32x4 & ka, const u32x4 & kb, const float (&qa)[8], const float (&qb)[8], float scale) {
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

// final tree over the 32 partial-sum chains + F64 leftovers, by threads 0..63 (tid = d_head index); then publish
template <bool Q = false>                 // Q: the result leaves in F32, one granule per element (the out-projection quantises it from F32)
__device__ __forceinline__ void mg_attn_finish(const float * part, const wa_f16 * vleft /* [nl][64] */, const wa_f16 * p16, int np, int nl,
                                               gu64 * edge, int h, unsigned seq, int tid, double * dbl0, double * dbl1 /* LDS, [16][64] each */,
                                               mg_kargs A = nullptr, int tslot = -1) {
    // All eight waves call.  The leftover cells (vec.cpp:221-223: F64, index order) are spread over waves 1..7 - five cells each, for the 64
    // outputs: the product as ONE v_fma_mix_f32 (F16 x F16 is exact in F32; with a -0.0 addend it is the multiplication's float), converted
    // to F64 and parked in LDS - while wave 0 runs the tree; what is left in series is wave 0's 32 F64 additions.  All 32 rows are read at
    // constant offsets - rows >= nl hold stale LDS and become -0.0, the neutral element of an IEEE sum - and nl is opaque: with
    // `cc < nl ? cc : 0` addresses the compiler kept 32 scalar selects and 32 masks over the layer loop, spilled them, and issued the LDS
    // reads one at a time.  (One wave doing all of it: 1.9 us per head; now 0.9.)
    asm volatile("" : "+s"(nl));
    if (tid >= 64) {
        const int wv = tid >> 6, o = tid & 63;
        float nzero = -0.0f;
        asm volatile("" : "+v"(nzero));          // (opaque, or the fma is folded back into conversions + a multiplication)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int cc = 5 * (wv - 1) + i;
            if (cc < 32) {
                const float pr = fmaf(h2f(vleft[cc * 64 + o]), h2f((p16 + np)[cc]), nzero);
                (cc < 16 ? dbl0 : dbl1)[(cc & 15) * 64 + o] = (double) (cc < nl ? pr : -0.0f);
            }
        }
    }
    double sumf = 0.0;
    if (tid < 64) {
        if (tslot >= 0) mg_trace(A, tid == 0, tslot, mg_now());
        float s32[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) s32[r] = part[r * 64 + tid];
        sumf = (double) wa_tree32(s32);
        if (tslot >= 0) mg_trace(A, tid == 0, tslot + 1, mg_now() + (sumf == 1e300 ? 1u : 0u));
    }
    mg_barrier();
    if (tid < 64) {
        double dv[32];
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) dv[cc] = (cc < 16 ? dbl0 : dbl1)[(cc & 15) * 64 + tid];
        __builtin_amdgcn_sched_barrier(0);       // all 32 reads in flight together
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) sumf += dv[cc];
        if (tslot >= 0) mg_trace(A, tid == 0, tslot + 2, mg_now() + (sumf == 1e300 ? 1u : 0u));
        if constexpr (Q) {
            // The head's 64 outputs are two Q8_0 blocks of the out-projection's operand (quantize_row_q8_0, arch/x86/quants.c): quantised
            // HERE, once, instead of by every consumer - a block leaves as 8 quads + its scale (9 granules instead of 32 F32 values).
            const float y = (float) sumf;
            float a = fabsf(y);
            a = fmaxf(a, dpp_f32<0x128>(a)); a = fmaxf(a, dpp_f32<0x124>(a)); a = fmaxf(a, dpp_f32<0x122>(a)); a = fmaxf(a, dpp_f32<0x121>(a));
            a = fmaxf(a, __shfl_xor(a, 16, 32));
            const float dsc = a / 127.f, id = a != 0.0f ? 127.f / a : 0.0f;
            const unsigned q = (unsigned) (int) rintf(y * id) & 0xffu;
            const unsigned w = q | (dpp_u32<0x101>(q) << 8) | (dpp_u32<0x102>(q) << 16) | (dpp_u32<0x103>(q) << 24);      // row_shl:1..3: the quad of lanes 4k..4k+3 in lane 4k
            gu64 * eb = edge + (size_t) (2 * h + (tid >> 5)) * 9;
            if ((tid & 3) == 0) gr_store(eb + ((tid & 31) >> 2), seq, w);
            if ((tid & 31) == 0) gr_store(eb + 8, seq, __float_as_uint(h2f(f2h(dsc))));
        } else {
        const unsigned hv = (unsigned) f2h((float) sumf);
        const unsigned hi = dpp_u32<0x101>(hv);          // row_shl:1: lane i reads lane i+1
        if ((tid & 1) == 0) gr_store(edge + ((h * 64 + tid) >> 1), seq, (hv & 0xffffu) | (hi << 16));
        }
    }
}

// -------------------------------------------------------------------------------------------------
// role: self-attention of head h (whisper.cpp:2636-2651), every layer.  The K/V cells of earlier tokens are copied
// into LDS while the GEMV workgroups are busy with the previous phases; the new cell arrives with the query.
// -------------------------------------------------------------------------------------------------
template <bool Q = false>
__device__ __forceinline__ void mg_role_self(mg_kargs A_, int idx_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const mg_kargs A = mg_uniform(A_);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = __builtin_amdgcn_readfirstlane(idx_);
    mg_ctl c; c.status = (gu32 *) A->status; c.seq = A->seq; c.dead = false;
    mg_chaos_start(c.seq);

    wa_f16 * Ks = (wa_f16 *) smem;                                  // [512][64]
    wa_f16 * Vs = Ks + WA_MEGA_MAX_KV * 64;                          // [512][64]
    const mg_att_smem M = mg_att_carve(smem + (size_t) WA_MEGA_MAX_KV * 64 * 2 * 2, WA_MEGA_MAX_KV);
    const int d = A->d, n_kv = A->n_kv, kv_head = A->kv_head, L = A->n_layer;
    const int a = tid & 3, kslot = tid >> 2;
    if (wave == 0) mg_pick(A, lane, (int *) (smem + MG_PICK_OFF));
    for (int l = 0; l < L; ++l) {
        {   // cells of earlier tokens -> LDS (row kv_head is stale here and replaced below)
            const gch kc = (gch) A->kv_k + (size_t) l * A->kv_layer_stride + h * 64;
            const gch vc = (gch) A->kv_v + (size_t) l * A->kv_layer_stride + h * 64;
            const int n16 = n_kv * 8;
            u32x4 tk[8], tv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = tid + MG_THREADS * j;
                if (idx < n16) {
                    tk[j] = *(const GAS u32x4 *) (kc + (size_t) (idx >> 3) * d + (idx & 7) * 8);
                    tv[j] = *(const GAS u32x4 *) (vc + (size_t) (idx >> 3) * d + (idx & 7) * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = tid + MG_THREADS * j;
                if (idx < n16) { *(u32x4 *) (Ks + (size_t) idx * 8) = tk[j]; *(u32x4 *) (Vs + (size_t) idx * 8) = tv[j]; }
            }
        }
        mg_barrier();
        if (wave == 0) {    // q | k | v of this head: three runs of 32 packed granules
            unsigned v[2];
            const unsigned sp = mg_sweep<2>(mg_edge(A, l, E_QKV), [&](int k) {
                if (k == 0) return (lane < 32 ? 0 : (d >> 1)) + h * 32 + (lane & 31);
                return lane < 32 ? d + h * 32 + lane : -1; }, c, lane, v, 1000u + l);
            mg_trace(A, h == 0 && lane == 0, (l * 8 + 6) * 8 + 0, mg_now()); mg_trace(A, h == 0 && lane == 0, (l * 8 + 6) * 8 + 1, sp);
            if (lane < 32) { ((unsigned *) M.qs)[lane] = v[0]; ((unsigned *) (Vs + (size_t) kv_head * 64))[lane] = v[1]; }
            else ((unsigned *) (Ks + (size_t) kv_head * 64))[lane - 32] = v[0];
        }
        mg_barrier();
        // ---- scores: 4 lanes per key ----
        float lmax = -INFINITY;
        {
            float qa[8], qb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { qa[i] = h2f(M.qs[8 * a + i]); qb[i] = h2f(M.qs[32 + 8 * a + i]); }
            for (int c0 = 0; c0 < n_kv; c0 += MG_THREADS / 4) {
                const int cc = c0 + kslot, cl = cc < n_kv ? cc : n_kv - 1;
                const u32x4 ka = *(const u32x4 *) (Ks + (size_t) cl * 64 + 8 * a), kb = *(const u32x4 *) (Ks + (size_t) cl * 64 + 32 + 8 * a);
                const float r = mg_score(ka, kb, qa, qb, 1.0f);
                if (cc < n_kv) { if (a == 0) M.sc[cc] = r; lmax = fmaxf(lmax, r); }
            }
        }
        mg_trace(A, h == 0 && tid == 0, (l * 8 + 6) * 8 + 4, mg_now());
        {   // soft_max (ops.cpp:4792-4818, vec.cpp:257-308) with one cell per thread (n_kv <= 512) and three barriers: maximum;
            // exp + the 8-lane group tree + F64 partial sums; every thread certifies the total and scales its own cell
            lmax = wave_max(lmax);
            if (lane == 0) M.red[wave] = lmax;
            mg_barrier();
            float mx = M.red[0];
#pragma unroll
            for (int k = 1; k < MG_NW; ++k) mx = fmaxf(mx, M.red[k]);
            const int n8 = n_kv & ~7, ng = n8 >> 3;
            float e = 0.0f;
            double ps = 0.0;
            if (tid < n_kv) e = tid < n8 ? wa_expf(M.sc[tid] - mx) : wa_expf_libm(M.sc[tid] - mx);
            {
                float t = e + dpp_f32<0x104>(e);        // lanes r = 0..3 of a group of 8 cells: e[r] + e[r+4]
                t = t + dpp_f32<0x102>(t);              // r = 0: (e0+e4)+(e2+e6)   r = 1: (e1+e5)+(e3+e7)
                t = t + dpp_f32<0x101>(t);              // r = 0: the group sum in ops.cpp's order
                if (tid < n8) ps = (tid & 7) == 0 ? (double) t : 0.0;
                else if (tid < n_kv) ps = (double) e;   // the n % 8 tail cells
            }
            ps = wave_sum_d(ps);
            if (lane == 0) M.redd[wave] = ps;
            mg_barrier();
            const double tot = ((M.redd[0] + M.redd[1]) + (M.redd[2] + M.redd[3])) + ((M.redd[4] + M.redd[5]) + (M.redd[6] + M.redd[7]));
            // (the reference adds the ng + (n % 8) addends one after the other in F64: error <= (ng + 7) u S; this sum is a tree of depth <= 16 over the SAME addends:
            //  error <= 16 u S; together (ng + 8 + 16) u S - not twice the reference's bound, which sent twice as many soft-maxes back to the launch sequence)
            const double delta = (double) (ng + 8 + 16) * 0x1p-53 * tot * 1.000001;
            const float ilo = (float) (1.0 / (tot + delta)), ihi = (float) (1.0 / (tot - delta));
            if (ilo != ihi && tid == 0 && !c.dead) __hip_atomic_store(c.status, (unsigned) WA_MEGA_REDO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < n_kv) M.p16[tid] = f2h(e * ilo);
            mg_barrier();
        }
        mg_trace(A, h == 0 && tid == 0, (l * 8 + 6) * 8 + 5, mg_now());
        MG_CHAOS_ID(1000 + h, 21u, c.seq);
        // ---- P V: chains r = cell mod 32 (4 per wave), lane = d_head index ----
        const int np = n_kv & ~31, nsteps = np >> 5;
        {
            float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            const int r0 = wave * 4;
            for (int s = 0; s < nsteps; ++s) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = fmaf(h2f(Vs[(size_t) (s * 32 + r0 + i) * 64 + lane]), h2f(M.p16[s * 32 + r0 + i]), acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) M.part[(r0 + i) * 64 + lane] = acc[i];
        }
        mg_barrier();
        // (the 8 KB behind cell WA_MEGA_KV_ROOM of the K and of the V copy are free: the host sends longer contexts through the launch sequence)
        mg_attn_finish<Q>(M.part, Vs + (size_t) np * 64, M.p16, np, n_kv - np, mg_edge(A, l, E_AO), h, c.seq, tid, (double *) (Ks + WA_MEGA_KV_ROOM * 64),
                          (double *) (Vs + WA_MEGA_KV_ROOM * 64), A, A->dbg && h == 0 && l == MG_WGTRACE_LAYER ? 3004 : -1);
        mg_trace(A, h == 0 && tid == 0, (l * 8 + 6) * 8 + 3, mg_now());
        mg_barrier();
    }
    unsigned pf[96];
    bool have_pf = false;
    float gw[MG_NP3], gb[MG_NP3];
    if (wave >= 1) mg_prefetch_logits<0, Q>(A, pf, have_pf, lane, wave);
    mg_ln_params<MG_NP3>(gw, gb, A->lnf_w, A->lnf_b, A->d, mg_slot(wave, MG_EX_FINAL), lane);
    mg_final<MG_NP3, 0, Q>(A, c, smem, pf, have_pf, gw, gb, lane, wave);
}

// -------------------------------------------------------------------------------------------------
// role: cross-attention over the encoder K/V (whisper.cpp:2683-2758), every layer; FOUR workgroups per head.
// The score and soft-max arithmetic of one head is VALU-bound on one CU (3 + 5 us measured), so the head's keys are
// split so that every piece of the reference's summation structure stays inside one workgroup: workgroup w owns the
// cells c with (c mod 32) in [8w, 8w + 8), i.e. the partial-sum chains 8w..8w+7 of the P V product AND the soft-max
// groups g = c / 8 with g mod 4 == w (whole groups of 8 consecutive cells).  Per workgroup: 376 keys (K: 24 VGPRs
// per lane, V: one 2-byte element per step and lane), both loaded a whole layer ahead.  The four exchange
//   (1) their local maxima (one granule each),  (2) their F64 partial sums of the group sums (two granules each),
//   (3) their 8 x 64 chain sums + the probabilities of their leftover cells, gathered by workgroup 0, which runs the
//       final tree, the F64 leftovers and publishes the head's output.
// The F64 sum is order-independent when certified (k_attn_exact); an uncertified sum (~1e-9 per soft-max) raises
// status WA_MEGA_REDO and the host recomputes the token with the launch sequence.
// -------------------------------------------------------------------------------------------------
#define MG_CSTEPS 48                                    // steps of a chain incl. the leftover step: T <= 1535
#define MG_CGR 2048                                     // granules per (layer, head) of the cross exchange area
#define MG_CGR_MAX 0
#define MG_CGR_SUM 8
#define MG_CGR_XCC 16                                   // (layer 0 area only) the four workgroups' XCC_IDs
#define MG_CGR_PART 64                                  // + (w - 1) * 576: 512 chain sums + 8 leftover probabilities

template <bool Q = false>
__device__ __forceinline__ void mg_role_cross(mg_kargs A_, int idx_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const mg_kargs A = mg_uniform(A_);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = A->n_head;
    const int ci = __builtin_amdgcn_readfirstlane(idx_);
    const int h = ci >> 2, w = ci & 3;
    mg_ctl c; c.status = (gu32 *) A->status; c.seq = A->seq; c.dead = false;
    mg_chaos_start(c.seq);
    const unsigned seq = c.seq;

    // LDS: part [32][64] f32 | vleft [32][64] f16 | sc [384] f32 | p16 [384] f16 | pleft [32] f16 | qs [64] f16 | red [8] | redd [8] | bc [4]
    float  * part  = (float *) smem;
    wa_f16 * vleft = (wa_f16 *) (smem + 8192);
    float  * sc    = (float *) (smem + 8192 + 4096);
    wa_f16 * p16   = (wa_f16 *) (smem + 8192 + 4096 + 1536);
    wa_f16 * pleft = (wa_f16 *) (smem + 8192 + 4096 + 1536 + 768);
    wa_f16 * qs    = (wa_f16 *) (smem + 8192 + 4096 + 1536 + 768 + 64);
    double * redd  = (double *) (smem + 8192 + 4096 + 1536 + 768 + 64 + 128);
    float  * red   = (float *) (smem + 8192 + 4096 + 1536 + 768 + 64 + 128 + 64);
    float  * bc    = red + 8;

    const int T = A->T, tpad = A->cross_tpad, L = A->n_layer;
    const float kq_scale = A->kq_scale;
    const int a = tid & 3, ks = tid >> 2;
    const int np = T & ~31, nsteps = np >> 5, nl = T - np, n8 = T & ~7, ng = n8 >> 3;
    if (wave == 0) mg_pick(A, lane, (int *) (smem + MG_PICK_OFF));
    // Do the head's four workgroups share an XCD (mg_role_of places them 8 apart, which is where the dispatcher has been observed to put
    // them - not a promise)?  Each publishes its XCC_ID, all read the four; only then do the exchanges among them use L2-resident stores.
    if (wave == 0) {
        const unsigned xcc = (unsigned) __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;        // HW_REG_XCC_ID[3:0]
        gu64 * X0 = (gu64 *) A->cross_gr + (size_t) h * MG_CGR + MG_CGR_XCC;
        if (lane == 0) gr_store(X0 + w, seq, xcc);
        unsigned v[1];
        mg_sweep<1>(X0, [&](int) { return lane < 4 ? lane : -1; }, c, lane, v, 2900u);
        const bool same = lane >= 4 || v[0] == xcc;
        if (lane == 0) bc[2] = __builtin_amdgcn_ballot_w64(same) == ~0ull && !c.dead ? 1.0f : 0.0f;
    }
    mg_barrier();
    const bool local = bc[2] != 0.0f;
    for (int l = 0; l < L; ++l) {
        gu64 * X = (gu64 *) A->cross_gr + ((size_t) l * H + h) * MG_CGR;
        const gch kp = (gch) A->cross_k + (size_t) l * A->cross_layer_stride + (size_t) h * tpad * 64;
        const gch vp = (gch) A->cross_v + (size_t) l * A->cross_layer_stride + (size_t) h * tpad * 64;
        // ---- prefetch (a whole layer ahead of the query): own keys, own chain elements, leftover V rows ----
        u32x4 ka[3], kb[3];
        unsigned short vv[MG_CSTEPS];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int o = p * 128 + ks, cc = 32 * (o >> 3) + 8 * w + (o & 7);
            if (cc < T) { ka[p] = *(const GAS u32x4 *) (kp + (size_t) cc * 64 + 8 * a); kb[p] = *(const GAS u32x4 *) (kp + (size_t) cc * 64 + 32 + 8 * a); }
        }
#pragma unroll
        for (int s = 0; s < MG_CSTEPS; ++s) if (s < nsteps) vv[s] = *(const GAS unsigned short *) (vp + (size_t) (32 * s + 8 * w + wave) * 64 + lane);
        if (w == 0 && tid < 256) {
            const int row = tid >> 3;
            if (row < nl) *(u32x4 *) (vleft + (size_t) tid * 8) = *(const GAS u32x4 *) (vp + (size_t) (np + row) * 64 + (tid & 7) * 8);
        }
        if (wave == 0) {
            unsigned v[1];
            mg_trace(A, ci == 0 && lane == 0, (l * 8 + 7) * 8 + 2, mg_now());
            const unsigned sp = mg_sweep<1>(mg_edge(A, l, E_QC), [&](int) { return lane < 32 ? h * 32 + lane : -1; }, c, lane, v, 2000u + l);
            mg_trace(A, ci == 0 && lane == 0, (l * 8 + 7) * 8 + 0, mg_now()); mg_trace(A, ci == 0 && lane == 0, (l * 8 + 7) * 8 + 1, sp);
            if (A->dbg && l == MG_WGTRACE_LAYER && w == 0) mg_trace(A, lane == 0, 3200 + h, mg_now());
            if (lane < 32) ((unsigned *) qs)[lane] = v[0];
        }
        mg_barrier();
#define MG_CX(k) do { if (A->dbg && l == MG_WGTRACE_LAYER) mg_trace(A, ci == 0 && tid == 0, 3010 + (k), mg_now()); } while (0)
        MG_CX(0);
        MG_CHAOS_ID(2000 + ci, 31u, seq);
        // ---- scores of the own cells (local index o = 8 s + r  <->  cell 32 s + 8 w + r) ----
        float lmax = -INFINITY;
        {
            float qa[8], qb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { qa[i] = h2f(qs[8 * a + i]); qb[i] = h2f(qs[32 + 8 * a + i]); }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int o = p * 128 + ks, cc = 32 * (o >> 3) + 8 * w + (o & 7);
                const float r = mg_score(ka[p], kb[p], qa, qb, kq_scale);
                if (cc < T) { if (a == 0) sc[o] = r; lmax = fmaxf(lmax, r); }
            }
        }
        lmax = wave_max(lmax);
        if (lane == 0) red[wave] = lmax;
        MG_CX(1);
        mg_barrier();
        MG_CX(2);
        if (wave == 0) {        // (1) maxima of the four workgroups
            float m = red[0];
#pragma unroll
            for (int k = 1; k < MG_NW; ++k) m = fmaxf(m, red[k]);
            if (lane == 0) gr_store_l(X + MG_CGR_MAX + w, seq, __float_as_uint(m), local);
            unsigned v[1];
            mg_sweep<1>(X + MG_CGR_MAX, [&](int) { return lane < 4 ? lane : -1; }, c, lane, v, 2100u + l);
            float g = lane < 4 ? __uint_as_float(v[0]) : -INFINITY;
            g = fmaxf(g, dpp_f32<0x4e>(g)); g = fmaxf(g, dpp_f32<0xb1>(g));      // max over lanes 0..3
            if (lane == 0) bc[0] = g;
            MG_CX(3);
        }
        mg_barrier();
        mg_trace(A, ci == 0 && tid == 0, (l * 8 + 7) * 8 + 4, mg_now());
        const float mx = bc[0];
        // ---- exp, group sums (8-lane tree = ops.cpp's), F64 partial sum: thread = one own cell (ops.cpp:4792-4818, vec.cpp:257-308) ----
        {
            double ps = 0.0;
            if (tid < 8 * MG_CSTEPS) {
                const int g = 4 * (tid >> 3) + w, cc = 8 * g + (tid & 7);          // global group / cell of local index tid
                const float e = cc < n8 ? wa_expf(sc[tid] - mx) : (cc < T ? wa_expf_libm(sc[tid] - mx) : 0.0f);
                sc[tid] = e;
                float t = e + dpp_f32<0x104>(e);        // lanes r = 0..3 of the group: e[r] + e[r+4]
                t = t + dpp_f32<0x102>(t);              // r = 0: (e0+e4)+(e2+e6)   r = 1: (e1+e5)+(e3+e7)
                t = t + dpp_f32<0x101>(t);              // r = 0: the group sum, ops.cpp's tree
                if (g < ng) ps = (tid & 7) == 0 ? (double) t : 0.0;
                else ps = (double) e;                   // the n % 8 tail cells (any order: the total is certified below)
            }
            ps = wave_sum_d(ps);
            if (lane == 0) redd[wave] = ps;
        }
        MG_CX(4);
        mg_barrier();
        if (wave == 0) {        // (2) partial sums -> total, certified
            const double ps = ((redd[0] + redd[1]) + (redd[2] + redd[3])) + ((redd[4] + redd[5]) + (redd[6] + redd[7]));
            const u64 pb = (u64) __double_as_longlong(ps);
            if (lane == 0) { gr_store_l(X + MG_CGR_SUM + 2 * w, seq, (unsigned) pb, local); gr_store_l(X + MG_CGR_SUM + 2 * w + 1, seq, (unsigned) (pb >> 32), local); }
            unsigned v[1];
            mg_sweep<1>(X + MG_CGR_SUM, [&](int) { return lane < 8 ? lane : -1; }, c, lane, v, 2200u + l);
            double tot = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned lo = __builtin_amdgcn_readlane(v[0], 2 * k), hi = __builtin_amdgcn_readlane(v[0], 2 * k + 1);
                tot += __longlong_as_double((long long) (((u64) hi << 32) | lo));
            }
            // (the reference adds the ng + (n % 8) addends one after the other in F64: error <= (ng + 7) u S; this sum is a tree of depth <= 16 over the SAME addends:
            //  error <= 16 u S; together (ng + 8 + 16) u S - not twice the reference's bound, which sent twice as many soft-maxes back to the launch sequence)
            const double delta = (double) (ng + 8 + 16) * 0x1p-53 * tot * 1.000001;
            const float ilo = (float) (1.0 / (tot + delta)), ihi = (float) (1.0 / (tot - delta));
            if (ilo != ihi && lane == 0 && !c.dead) __hip_atomic_store(c.status, (unsigned) WA_MEGA_REDO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) bc[1] = ilo;
            MG_CX(5);
        }
        mg_barrier();
        mg_trace(A, ci == 0 && tid == 0, (l * 8 + 7) * 8 + 5, mg_now());
        const float inv = bc[1];
        if (tid < 8 * MG_CSTEPS) {
            const int cc = 32 * (tid >> 3) + 8 * w + (tid & 7);
            if (cc < T) {
                const wa_f16 ph = f2h(sc[tid] * inv);
                p16[(tid & 7) * MG_CSTEPS + (tid >> 3)] = ph;         // by chain: the P V wave reads its 48 probabilities as 6 x 16 bytes
                if (cc >= np) { if (w == 0) pleft[cc - np] = ph; else gr_store_l(X + MG_CGR_PART + (w - 1) * 576 + 512 + (tid & 7), seq, (unsigned) ph, local); }
            }
        }
        mg_barrier();
        MG_CX(6);
        MG_CHAOS_ID(2000 + ci, 32u, seq);
        // ---- P V: wave = own chain (cells 32 s + 8 w + wave), lane = d_head index ----
        {
            float acc = 0.0f;
            half8 pw[MG_CSTEPS / 8];
#pragma unroll
            for (int k = 0; k < MG_CSTEPS / 8; ++k) pw[k] = *(const half8 *) (p16 + wave * MG_CSTEPS + 8 * k);
#pragma unroll
            for (int s = 0; s < MG_CSTEPS; ++s) if (s < nsteps) acc = fmaf(h2f(vv[s]), (float) pw[s >> 3][s & 7], acc);
            if (w == 0) part[wave * 64 + lane] = acc;
            else gr_store_l(X + MG_CGR_PART + (w - 1) * 576 + wave * 64 + lane, seq, __float_as_uint(acc), local);
        }
        mg_trace(A, ci == 0 && tid == 0, (l * 8 + 7) * 8 + 6, mg_now());
        if (w == 0) {           // (3) gather the other three workgroups' chain sums and leftover probabilities, finish the head
            if (wave >= 1 && wave <= 6) {
                const int ww = (wave - 1) >> 1, half = (wave - 1) & 1;       // source workgroup ww + 1, rows [4 half, 4 half + 4) of its 8 chains
                gu64 * src = X + MG_CGR_PART + ww * 576;
                unsigned v[5];
                mg_sweep<5>(src, [&](int k) { return k < 4 ? half * 256 + 64 * k + lane : (half == 0 && lane < 8 && 8 * (ww + 1) + lane < nl ? 512 + lane : -1); }, c, lane, v, 2300u + l);
#pragma unroll
                for (int k = 0; k < 4; ++k) part[(8 * (ww + 1) + 4 * half + k) * 64 + lane] = __uint_as_float(v[k]);
                if (half == 0 && lane < 8) { const int cc = 8 * (ww + 1) + lane; if (cc < nl) pleft[cc] = (wa_f16) v[4]; }
            }
            mg_barrier();
            mg_trace(A, ci == 0 && tid == 0, (l * 8 + 7) * 8 + 7, mg_now());
            mg_attn_finish<Q>(part, vleft, pleft - np, np, nl, mg_edge(A, l, E_AO2), h, seq, tid, (double *) (smem + 16384), (double *) (smem + 24576), A,
                              A->dbg && ci == 0 && l == MG_WGTRACE_LAYER ? 3000 : -1);
            mg_trace(A, ci == 0 && tid == 0, (l * 8 + 7) * 8 + 3, mg_now());
            if (A->dbg && l == MG_WGTRACE_LAYER) mg_trace(A, tid == 0, 3100 + h, mg_now());
        }
        mg_barrier();
    }
    unsigned pf[96];
    bool have_pf = false;
    float gw[MG_NP3], gb[MG_NP3];
    if (wave >= 1) mg_prefetch_logits<0, Q>(A, pf, have_pf, lane, wave);
    mg_ln_params<MG_NP3>(gw, gb, A->lnf_w, A->lnf_b, A->d, mg_slot(wave, MG_EX_FINAL), lane);
    mg_final<MG_NP3, 0, Q>(A, c, smem, pf, have_pf, gw, gb, lane, wave);
}

__global__ __launch_bounds__(MG_THREADS) void k_decode_mega(const wa_mega_args A) {
    int role, idx;                                       // H self-attention + 4 H cross-attention workgroups, the rest stream weights
    mg_role_of((int) gridDim.x, A.n_head, (int) blockIdx.x, role, idx);
    const mg_kargs Ap = (mg_kargs) __builtin_amdgcn_kernarg_segment_ptr();     // = &A (the struct is the only argument)
    if (role == 0) {        // compile-time row lengths for the shapes that matter (ggml-small / -base..: d = 768; large: d = 1280): no guards in the products
        if (A.d == 768) mg_role_gemv<2, 24>(Ap, idx); else if (A.d < 768) mg_role_gemv<2, 0>(Ap, idx);
        else mg_role_gemv<MG_NP3, 0>(Ap, idx);        // (a d = 1280 instantiation was tried: the 40-step products fully unrolled spill 2300 registers)
    }
    else if (role == 1) mg_role_self(Ap, idx);
    else                mg_role_cross(Ap, idx);
}

// the same step for a quantised model (Q5_0 / Q8_0 files): a kernel of its own, so that the F16 kernel's code and registers stay as tuned
__global__ __launch_bounds__(MG_THREADS) void k_decode_mega_q(const wa_mega_args A) {
    int role, idx;
    mg_role_of((int) gridDim.x, A.n_head, (int) blockIdx.x, role, idx);
    const mg_kargs Ap = (mg_kargs) __builtin_amdgcn_kernarg_segment_ptr();
    if (role == 0) {
        if (A.d == 768) mg_role_gemv<2, 24, true>(Ap, idx); else if (A.d < 768) mg_role_gemv<2, 0, true>(Ap, idx);
        else if (A.d == 1280) mg_role_gemv<MG_NP3, 40, true>(Ap, idx); else mg_role_gemv<MG_NP3, 0, true>(Ap, idx);
    }
    else if (role == 1) mg_role_self<true>(Ap, idx);
    else                mg_role_cross<true>(Ap, idx);
}

size_t wa_mega_lds_bytes() {
    const size_t s_self  = (size_t) WA_MEGA_MAX_KV * 64 * 2 * 2 + MG_ATT_SMEM(WA_MEGA_MAX_KV);
    const size_t s_cross = 32768;       // 14.9 KB of the role's arrays, then 2 x 8 KB at 16 KB for mg_attn_finish's split form
    const size_t s_gemv  = MG_PICK_OFF;
    size_t m = s_self > s_cross ? s_self : s_cross;
    m = m > s_gemv ? m : s_gemv;
    if (m > MG_PICK_OFF || (MG_PICK_OFF & 255) != 0) abort();      // the pick area sits behind the largest role
    return MG_PICK_OFF + MG_PICK_BYTES;
}

bool wa_launch_decode_mega(hipStream_t s, const wa_mega_args & a, int n_wg) {
    static bool attr_set[64] = {};          // per device (the attribute belongs to the device's copy of the code object)
    const size_t lds = wa_mega_lds_bytes();
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        if (hipFuncSetAttribute((const void *) k_decode_mega, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess ||
            hipFuncSetAttribute((const void *) k_decode_mega_q, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) return false;
        attr_set[dev & 63] = true;
    }
    if (a.quant) hipLaunchKernelGGL(k_decode_mega_q, dim3(n_wg), dim3(MG_THREADS), lds, s, a);
    else         hipLaunchKernelGGL(k_decode_mega, dim3(n_wg), dim3(MG_THREADS), lds, s, a);
    return hipGetLastError() == hipSuccess;
}
