// wa_device.h - device-side helpers shared by wa_kernels.hip (MFMA "flash" path) and wa_exact.hip
// (reference-order path): F16 conversion, wave reductions, ggml's expf polynomial, the GELU table
// lookup and the GEMM epilogues.
#pragma once
#include "wa_kernels.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 h16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

__device__ __forceinline__ float h2f(wa_f16 v) { union { wa_f16 u; h16 h; } c; c.u = v; return (float) c.h; }
// RNE from the F32 VALUE.  The empty asm makes `v` opaque: without it hipcc folds f16(a * b) / f16(a + b) into one
// v_fma_mixlo_f16, which rounds the exact result ONCE - not the reference's round-to-F32-then-to-F16 (seen as a 1-ulp
// different soft-max probability every few thousand values).
__device__ __forceinline__ wa_f16 f2h(float v) { asm("" : "+v"(v)); union { wa_f16 u; h16 h; } c; c.h = (h16) v; return c.u; }

// Workgroup barrier that orders LDS only.  __syncthreads() also drains the vector-memory counter (s_waitcnt vmcnt(0)), i.e. every
// wave would wait at it for global loads it has only just issued as a prefetch.
__device__ __forceinline__ void wa_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
// DPP cross-lane moves (one VALU op each; ds_bpermute-based __shfl costs a ~100-cycle LDS crossbar trip, which is
// what the latency-bound decode kernels were spending their time on).  ctrl: row_shl:n = 0x100+n, row_shr:n = 0x110+n,
// row_bcast15 = 0x142, row_bcast31 = 0x143, quad_perm = 0x00..0xff.  Out-of-range source lanes read 0 (bound_ctrl).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xffffffffll), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __builtin_bit_cast(double, ((long long) hi << 32) | (unsigned int) lo);
}
// wave-wide F64 sum, result broadcast to every lane (order of the additions is arbitrary: callers certify it)
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_f64<0x111>(v);          // row_shr:1
    v += dpp_f64<0x112>(v);          // row_shr:2
    v += dpp_f64<0x114>(v);          // row_shr:4
    v += dpp_f64<0x118>(v);          // row_shr:8  -> lane 15 of each row holds the row total
    v += dpp_f64<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v += dpp_f64<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int) (b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int) (b >> 32), 63);
    return __builtin_bit_cast(double, ((long long) hi << 32) | (unsigned int) lo);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}

// ggml's vectorised expf (vec.h:774-811, the AVX2+FMA flavour the reference CPU path runs for all
// but the last n%8 softmax elements), restated in scalar form with the same operation order.
__device__ __forceinline__ float wa_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = fmaf(-n, 0x1.7f7d1cp-20f, fmaf(-n, 0x1.62e4p-1f, x));
    const uint32_t e = __float_as_uint(z) << 23;
    const float k = __uint_as_float(e + __float_as_uint(1.0f));
    const float an = fabsf(n);
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u,
                         0x1.ffffecp-1f * b);
    if (!(an > 126.0f)) return fmaf(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    const float s1 = __uint_as_float(g + 0x7f000000u);
    const float s2 = __uint_as_float(e - g);
    if (an > 192.0f) return s1 * s1;
    return fmaf(s2, j, s2) * s1;
}

// GELU exactly as ggml_vec_gelu_f32 with GGML_GELU_FP16 (vec.h:571-585): table lookup on the F16
// bits of x, identity above 10, zero below -10.  Result is an F32 that is exactly F16-representable
// (or x itself for x >= 10).
// the same value with the table read unconditional (any F16 bit pattern is a valid index): a batch of these stays in flight
// together, where the branchy form costs one serial memory round trip per element
__device__ __forceinline__ float wa_gelu_nb(float x, const wa_f16 * __restrict__ table) {
    const float t = h2f(table[f2h(x)]);
    return x <= -10.0f ? 0.0f : (x >= 10.0f ? x : t);
}
__device__ __forceinline__ float wa_gelu(float x, const wa_f16 * __restrict__ table) {
    if (x <= -10.0f) return 0.0f;
    if (x >=  10.0f) return x;
    return h2f(table[f2h(x)]);
}

// quantize_row_q8_0 (ggml-cpu/arch/x86/quants.c, AVX2: d = max|x| / 127, q = rint(x * (127 / max|x|)), d kept as F16) of one 32-element
// block whose element `el` = lane & 31 is held by each lane of a half-wave; output in the kernel layout of wa_quant.hip
// (qs [row][el / 4][block][el % 4], qd [row][block]).  All 32 lanes of the half-wave must be active.
__device__ __forceinline__ void wa_q8_store(float y, int row, int blk, int el, int nb, int8_t * __restrict__ qs, float * __restrict__ qd) {
    float a = fabsf(y);
    a = fmaxf(a, dpp_f32<0x128>(a));        // row_ror:8, 4, 2, 1: the maximum over a 16-lane row, in every lane of it (one VALU op each;
    a = fmaxf(a, dpp_f32<0x124>(a));        //  five ds_bpermute exchanges here were half of a single-row LayerNorm launch)
    a = fmaxf(a, dpp_f32<0x122>(a));
    a = fmaxf(a, dpp_f32<0x121>(a));
    a = fmaxf(a, __shfl_xor(a, 16, 32));    // the block's two rows
    const float d = a / 127.f;
    const float id = a != 0.0f ? 127.f / a : 0.0f;
    qs[(((size_t) row * 8 + (el >> 2)) * nb + blk) * 4 + (el & 3)] = (int8_t) (int) rintf(y * id);      // to nearest, ties to even
    if (el == 0) qd[(size_t) row * nb + blk] = h2f(f2h(d));       // the dot product reads the scale back from its F16 field
}

// =================================================================================================
// GEMM epilogues (shared by the MFMA GEMM and the GEMV)
// =================================================================================================
// Operands an epilogue reads besides the accumulator; loading them early (before the K loop) takes their HBM/L2
// round trip off the critical path of the latency-bound decode kernels.
struct wa_epi_pre { float bias = 0.f, scale = 1.f, resid = 0.f; };

template <int EPI>
__device__ __forceinline__ wa_epi_pre epi_preload(const wa_epi & e, int m, int n) {
    wa_epi_pre p;
    if (e.bias) p.bias = e.bias[n];
    if ((EPI == WA_EPI_F16 || EPI == WA_EPI_CROSS_KV || EPI == WA_EPI_DEC_QKV) && e.scale) p.scale = e.scale[n];
    if (EPI == WA_EPI_RESID || EPI == WA_EPI_CONV2) p.resid = e.resid[(size_t) m * e.ldr + n];
    return p;
}

template <int EPI>
__device__ __forceinline__ void epi_apply(const wa_epi & e, int m, int n, float acc, const wa_epi_pre & p) {
    float v = acc;
    if (e.bias) v = v + p.bias;
    if (EPI == WA_EPI_F16) {
        if (e.scale) v = v * p.scale;
        ((wa_f16 *) e.out)[(size_t) m * e.ldo + n] = f2h(v);
    } else if (EPI == WA_EPI_ENC_QKV) {
        if (n < e.split0) ((wa_f16 *) e.out)[(size_t) m * e.ldo + n] = f2h(v);
        else              ((wa_f16 *) e.out2)[(size_t) (n - e.split0) * e.ldo2 + m] = f2h(v);
    } else if (EPI == WA_EPI_GELU_F16) {
        ((wa_f16 *) e.out)[(size_t) m * e.ldo + n] = f2h(wa_gelu(v, e.gelu));
    } else if (EPI == WA_EPI_GELU_F32) {
        ((float *) e.out)[(size_t) m * e.ldo + n] = wa_gelu(v, e.gelu);
    } else if (EPI == WA_EPI_RESID) {
        ((float *) e.out)[(size_t) m * e.ldo + n] = v + p.resid;
    } else if (EPI == WA_EPI_CONV2) {
        const float g = wa_gelu(v, e.gelu);
        if (e.dbg) e.dbg[(size_t) m * e.ldo + n] = g;
        ((float *) e.out)[(size_t) m * e.ldo + n] = p.resid + g;
    } else if (EPI == WA_EPI_F32) {
        ((float *) e.out)[(size_t) m * e.ldo + n] = v;
    } else if (EPI == WA_EPI_CROSS_KV) {
        // n = layer*2d + kv*d + head*64 + c ; aux0 = tpad, aux1 = d
        if (e.scale) v = v * p.scale;
        const int d = e.aux1, two_d = 2 * d;
        const int il = n / two_d, r = n - il * two_d;
        const int kv = r >= d, rr = kv ? r - d : r;
        const int n_head = d >> 6, hd = rr >> 6, c = rr & 63;
        wa_f16 * dst = (wa_f16 *) (kv ? e.out2 : e.out);
        dst[(((size_t) il * n_head + hd) * e.aux0 + m) * 64 + c] = f2h(v);
    } else if (EPI == WA_EPI_DEC_QKV) {
        if (e.scale) v = v * p.scale;
        const int row_off = e.dyn ? e.dyn[1] : e.row_off;
        if (n < e.split0)      ((wa_f16 *) e.out)[(size_t) m * e.ldo + n] = f2h(v);
        else if (e.rowp) {     // rows of different states (lock-step decode of several chunks): each into its own state's cell
            const wa_rowptr r = e.rowp[m];
            if (n < e.split1) (r.kv_k + e.rowp_off)[(size_t) r.kv_head * e.ldo2 + (n - e.split0)] = f2h(v);
            else              (r.kv_v + e.rowp_off)[(size_t) r.kv_head * e.ldo3 + (n - e.split1)] = f2h(v);
        }
        else if (n < e.split1) ((wa_f16 *) e.out2)[(size_t) (row_off + m) * e.ldo2 + (n - e.split0)] = f2h(v);
        else                   ((wa_f16 *) e.out3)[(size_t) (row_off + m) * e.ldo3 + (n - e.split1)] = f2h(v);
    }
}

template <int EPI>
__device__ __forceinline__ void epi_store(const wa_epi & e, int m, int n, float acc) {
    epi_apply<EPI>(e, m, n, acc, epi_preload<EPI>(e, m, n));
}

// -------------------------------------------------------------------------------------------------
// reference-order helpers (wa_exact.hip)
// -------------------------------------------------------------------------------------------------
// Final reduction of ggml_vec_dot_f16's 4x8 partial sums (simd-mappings.h:367-385): s[j*8 + l]
__device__ __forceinline__ float wa_tree32(const float (&s)[32]) {
    float a[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) a[l] = (s[l] + s[16 + l]) + (s[8 + l] + s[24 + l]);
    const float t0 = a[0] + a[4], t1 = a[1] + a[5], t2 = a[2] + a[6], t3 = a[3] + a[7];
    return (t0 + t1) + (t2 + t3);
}

// glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c, FMA ifunc variant) restated in F64: the reference's
// softmax calls libm expf for the last n%8 elements of every row (vec.cpp:301-305).  Validated on the
// build host against libm: 1 mismatch (1 ulp) in 4e7 arguments of [-110, 0].
static __device__ const unsigned long long WA_EXP2F_T[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL, 0x3fef72b83c7d517bULL,
    0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL, 0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL,
    0x3feedea64c123422ULL, 0x3feece086061892dULL, 0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL,
    0x3feea47eb03a5585ULL, 0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL, 0x3feee89f995ad3adULL,
    0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL, 0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL,
    0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};
__device__ __forceinline__ float wa_expf_libm(float x) {
    if (!(x >= -0x1.9fe368p6f)) return 0.0f;            // underflow and -inf (arguments here are always <= 0)
    const double InvLn2N = 0x1.71547652b82fep+0 * 32;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double SHIFT = 0x1.8p+52;
    double z = InvLn2N * (double) x;
    double kd = z + SHIFT;
    const unsigned long long ki = (unsigned long long) __double_as_longlong(kd);
    kd = kd - SHIFT;
    const double r = z - kd;
    const unsigned long long t = WA_EXP2F_T[ki & 31] + (ki << 47);
    const double s = __longlong_as_double((long long) t);
    z = fma(C0, r, C1);
    const double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(z, r2, y);
    y = y * s;
    return (float) y;
}

// -------------------------------------------------------------------------------------------------
// certified F64 sums (LayerNorm / soft-max denominators), shared by wa_exact.hip and wa_mega.hip
// -------------------------------------------------------------------------------------------------
// The reference sums a row in index order in F64 (ops.cpp:3225-3237).  A wave sums it in another order; the two
// F64 results can differ by at most delta = 2 n u sum|x| (u = 2^-53).  Rounding to F32 after the division is
// monotonic, so when (S - delta)/n and (S + delta)/n round to the SAME float, that float is the reference's
// value whatever its order was.  Otherwise (probability ~ n 2^-27 per row) one lane redoes the sum in index order.
// `rn` = 1.0 / (double) n, which a caller on a latency-critical path brings along (an F64 division is ~150 dependent cycles).
__device__ __forceinline__ bool wa_sum_bounds(double S, double A, int n, float & lo, float & hi, double rn) {
    // conservative bounds of (S -+ delta)/n by multiplication (1/n in F64 is within 2^-53; the 2^-48 slack covers it):
    // if both bounds round to the same float, the correctly rounded quotient of any sum in the interval does too.
    const double delta = (2.0 * (double) n * 0x1p-53 * A + fabs(S) * 0x1p-48) * rn * 1.000001;
    const double q = S * rn;
    lo = (float) (q - delta); hi = (float) (q + delta);
    return lo == hi;
}
__device__ __forceinline__ bool wa_sum_certain(double S, double A, int n, float & out, double rn) { float hi; return wa_sum_bounds(S, A, n, out, hi, rn); }
__device__ __forceinline__ bool wa_sum_certain(double S, double A, int n, float & out) { return wa_sum_certain(S, A, n, out, 1.0 / (double) n); }
// Second level, for a MEAN only.  A mean close to zero is where the first certificate fails (floats are dense there: |mean| below
// ~ n 2^-29 sum|x|, a few per cent of LayerNorm rows), yet the mean is read by nothing but t = x - mean.  If every element gives the same
// float t for mean = lo and for mean = hi, it does for every mean in between - the reference's included (x - m and the rounding to
// F32 are monotonic in m) -, and everything downstream is a function of those t.  Only a row with an element within ~2^-12 of its
// mean still needs the in-order sum.
__device__ __forceinline__ bool wa_mean_indifferent(float x, float lo, float hi) { return (x - lo) == (x - hi); }

// In-order F64 sum of an LDS-resident row by one lane (the certificate's fallback): b128 reads pipeline, the 8-cycle
// dependent F64 adds are all that is left (~3 us for 768 elements; the same loop over global memory took ~60 us).
__device__ __forceinline__ double wa_seq_sum_lds(const float * row, int d, bool squares, float mean) {
    double t = 0.0;
    int i = 0;
    for (; i + 4 <= d; i += 4) {
        const float4 v = *(const float4 *) (row + i);
        if (!squares) { t += (double) v.x; t += (double) v.y; t += (double) v.z; t += (double) v.w; }
        else {
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, e = v.w - mean;
            t += (double) (a * a); t += (double) (b * b); t += (double) (c * c); t += (double) (e * e);
        }
    }
    for (; i < d; ++i) { const float a = row[i] - mean; t += squares ? (double) (a * a) : (double) row[i]; }
    return t;
}

