// Eight expf at once for the host's 51 865-wide sampling passes (wa_full.cpp), bit-identical to glibc's expf for arguments <= 0.
//
// The reference calls libm expf per vocabulary entry (whisper.cpp:6134-6143: probabilities; 6115-6122: the log-sum-exp); glibc's expf
// (sysdeps/ieee754/flt-32/e_expf.c, 2.27 and later) is table + cubic in F64:  z = x N/ln2,  k = round(z),  r = z - k,
// s = 2^(k/N) from a 32-entry table,  result = (float) ((C0 r + C1) r^2 + (C2 r + 1)) s.  On an FMA machine the dispatcher picks the
// build of that file with contraction (the three a*b + c above are fused).  This header restates exactly that with AVX2 + FMA
// intrinsics, two 4-double halves per call.  tools/micro/expf_avx2_check.cpp compares it with libm over EVERY float in [-104, 0]: no
// difference (glibc 2.35; fusing k's or r's multiply-add as well makes no difference anywhere in that range either);
// wa_expf8_usable() repeats a sampled comparison at run time (another libm build, a CPU without FMA: the callers keep libm then).
#pragma once
#include <immintrin.h>
#include <cmath>
#include <cstdint>
#include <cstring>

static const uint64_t WA_EXP2F_TAB[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL, 0x3fef72b83c7d517bULL,
    0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL, 0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL,
    0x3feedea64c123422ULL, 0x3feece086061892dULL, 0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL,
    0x3feea47eb03a5585ULL, 0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL, 0x3feee89f995ad3adULL,
    0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL, 0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL,
    0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};

__attribute__((target("avx2,fma"))) static inline __m128 wa_expf4_core(__m128 xf) {
    const __m256d InvLn2N = _mm256_set1_pd(0x1.71547652b82fep+0 * 32), SHIFT = _mm256_set1_pd(0x1.8p+52);
    const __m256d C0 = _mm256_set1_pd(0x1.c6af84b912394p-5 / 32 / 32 / 32), C1 = _mm256_set1_pd(0x1.ebfce50fac4f3p-3 / 32 / 32), C2 = _mm256_set1_pd(0x1.62e42ff0c52d6p-1 / 32);
    const __m256d xd = _mm256_cvtps_pd(xf);
    __m256d z = _mm256_mul_pd(InvLn2N, xd);
    __m256d kd = _mm256_add_pd(z, SHIFT);
    const __m256i ki = _mm256_castpd_si256(kd);
    kd = _mm256_sub_pd(kd, SHIFT);
    const __m256d r = _mm256_sub_pd(z, kd);
    const __m256i idx = _mm256_and_si256(ki, _mm256_set1_epi64x(31));
    __m256i t = _mm256_i64gather_epi64((const long long *) WA_EXP2F_TAB, idx, 8);
    t = _mm256_add_epi64(t, _mm256_slli_epi64(ki, 47));
    const __m256d s = _mm256_castsi256_pd(t);
    z = _mm256_fmadd_pd(C0, r, C1);
    const __m256d r2 = _mm256_mul_pd(r, r);
    __m256d y = _mm256_fmadd_pd(C2, r, _mm256_set1_pd(1.0));
    y = _mm256_fmadd_pd(z, r2, y);
    y = _mm256_mul_pd(y, s);
    return _mm256_cvtpd_ps(y);
}

// y[i] = expf(x[i]) for x[i] <= 0: [-87, 0] by the restatement, below -104 (and -inf) the result is +0 (libm underflows to zero from
// -103.97 on), everything else - the subnormal results in between, NaN, positive arguments - is libm's call
__attribute__((target("avx2,fma"))) static inline void wa_expf8(const float * x, float * y) {
    const __m256 v = _mm256_loadu_ps(x);
    const __m128 a = wa_expf4_core(_mm256_castps256_ps128(v)), b = wa_expf4_core(_mm256_extractf128_ps(v, 1));
    const __m256 zero = _mm256_cmp_ps(v, _mm256_set1_ps(-104.0f), _CMP_LT_OQ);
    _mm256_storeu_ps(y, _mm256_andnot_ps(zero, _mm256_set_m128(b, a)));
    const __m256 in = _mm256_and_ps(_mm256_cmp_ps(v, _mm256_set1_ps(-87.0f), _CMP_GE_OQ), _mm256_cmp_ps(v, _mm256_setzero_ps(), _CMP_LE_OQ));
    int out = (~_mm256_movemask_ps(_mm256_or_ps(in, zero))) & 0xff;
    while (out) { const int k = __builtin_ctz(out); out &= out - 1; y[k] = expf(x[k]); }
}

// one-time self-check against this process's libm (a few thousand arguments spread over [-87, 0] incl. the table's break points)
static inline bool wa_expf8_usable() {
    static const bool ok = [] {
        if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("fma")) return false;
        uint32_t seed = 12345u;
        for (int it = 0; it < 4096; ++it) {
            float x[8], y[8];
            for (int k = 0; k < 8; ++k) {
                seed = seed * 1664525u + 1013904223u;
                const float u = (float) (seed >> 8) * (1.0f / 16777216.0f);
                x[k] = (it & 1) ? -87.0f * u : -20.0f * u * u;
            }
            wa_expf8(x, y);
            for (int k = 0; k < 8; ++k) { const float r = expf(x[k]); if (memcmp(&r, &y[k], 4) != 0) return false; }
        }
        return true;
    }();
    return ok;
}

// ---- the 51 865-wide passes of the sampling path on top of it (callers: wa_full.cpp; each has the libm loop beside it for machines that fail the self-check) ----

// probs[i] = logits[i] == -inf ? 0 : expf(logprobs[i])      (whisper.cpp:6134-6143)
__attribute__((target("avx2,fma"))) static inline void wa_probs_expf8(const float * logits, int n, const float * logprobs, float * probs) {
    int i = 0;
    const __m256 vninf = _mm256_set1_ps(-INFINITY);
    for (; i + 8 <= n; i += 8) {
        wa_expf8(logprobs + i, probs + i);
        const __m256 dead = _mm256_cmp_ps(_mm256_loadu_ps(logits + i), vninf, _CMP_EQ_OQ);
        _mm256_storeu_ps(probs + i, _mm256_andnot_ps(dead, _mm256_loadu_ps(probs + i)));
    }
    for (; i < n; ++i) probs[i] = logits[i] == -INFINITY ? 0.0f : expf(logprobs[i]);
}

// S = sum in index order, in F32, of expf(x[i] - mx) over the finite x[i]      (whisper.cpp:6115-6122, 6312-6320)
// Terms that provably leave the running sum unchanged are not evaluated: in round-to-nearest S + t == S whenever t < ulp(S)/2.
__attribute__((target("avx2,fma"))) static inline float wa_sum_expf8(const float * x, int n, float mx) {
    float S = 0.0f, thr = -INFINITY;
    // s in [2^e, 2^(e+1)): ulp(s)/2 = 2^(e-24); a term below it, i.e. with x[i] - mx < (e - 24) ln 2 (less a margin for expf's rounding), is skipped
    auto bound = [](float s) { int e; (void) frexpf(s, &e); const int k = e - 1 - 24; return (float) k * (k >= 0 ? 0.693147f : 0.693148f) - 0.01f; };
    int i = 0;
    const __m256 vmx = _mm256_set1_ps(mx), vninf = _mm256_set1_ps(-INFINITY);
    for (; i + 8 <= n; i += 8) {
        const __m256 v = _mm256_loadu_ps(x + i), d = _mm256_sub_ps(v, vmx);
        int mask = _mm256_movemask_ps(_mm256_and_ps(_mm256_cmp_ps(d, _mm256_set1_ps(thr), _CMP_GE_OQ), _mm256_cmp_ps(v, vninf, _CMP_GT_OQ)));
        if (!mask) continue;
        float dd[8], e[8];
        _mm256_storeu_ps(dd, d);
        if (mask & (mask - 1)) wa_expf8(dd, e);                       // several terms: all eight at once (the unused lanes cost nothing extra)
        else { const int k = __builtin_ctz(mask); e[k] = expf(dd[k]); }
        const float S0 = S;
        while (mask) { const int k = __builtin_ctz(mask); mask &= mask - 1; S += e[k]; }      // index order
        if (S != S0) thr = bound(S);
    }
    for (; i < n; ++i) if (x[i] > -INFINITY && x[i] - mx >= thr) S += expf(x[i] - mx);
    return S;
}
