// The vectorised passes of the sampling path (whisper-rust_amd/csrc/wa_expf8.h) against the plain libm loops they replace (the reference's:
// whisper.cpp:6115-6122 log-sum-exp, 6134-6143 probabilities, 6312-6320 timestamp mass): bit-identical on flat, peaked and masked rows of
// odd lengths, and wa_expf8 against libm on 2e7 sampled arguments (tools/micro/expf_avx2_check.cpp is the exhaustive form).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "wa_expf8.h"

static uint32_t rng_state = 2463534242u;
static float urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return (float) (rng_state >> 8) * (1.0f / 16777216.0f); }

int main() {
    if (!wa_expf8_usable()) { printf("wa_expf8 not usable on this machine (no AVX2 + FMA, or another libm): callers keep libm\n"); return 0; }
    long bad = 0;
    for (long it = 0; it < 2500000; ++it) {
        float x[8], y[8];
        for (int k = 0; k < 8; ++k) { const float u = urand(); x[k] = (it % 3 == 0) ? -104.5f * u : (it % 3 == 1) ? -30.0f * u * u : -1e-3f * u; }
        if (it % 1000 == 0) { x[3] = -INFINITY; x[5] = -0.0f; }
        wa_expf8(x, y);
        for (int k = 0; k < 8; ++k) { const float r = expf(x[k]); if (memcmp(&r, &y[k], 4) != 0) bad++; }
    }
    printf("expf8: %ld mismatches\n", bad);
    long bad_sum = 0, bad_probs = 0;
    for (int t = 0; t < 400; ++t) {
        const int n = 1 + (int) (urand() * 60000.0f);
        std::vector<float> x(n), lp(n), p0(n), p1(n);
        const int kind = t % 4;
        for (int i = 0; i < n; ++i) {
            float v = kind == 0 ? 8.0f * urand() : kind == 1 ? 40.0f * urand() * urand() * urand() : kind == 2 ? -50.0f * urand() : 3.0f * urand();
            if (kind == 1 && i == n / 3) v = 90.0f;                      // one dominant entry: almost everything behind it is skipped
            if (urand() < (kind == 3 ? 0.7f : 0.05f)) v = -INFINITY;     // masked entries
            x[i] = v;
        }
        if (t % 50 == 49) for (int i = 0; i < n; ++i) x[i] = -INFINITY;   // everything masked
        float mx = -INFINITY; for (int i = 0; i < n; ++i) if (x[i] > mx) mx = x[i];
        float S = 0.0f; for (int i = 0; i < n; ++i) if (x[i] > -INFINITY) S += expf(x[i] - mx);      // the reference's loop
        const float S8 = wa_sum_expf8(x.data(), n, mx);
        if (memcmp(&S, &S8, 4) != 0) { bad_sum++; printf("  sum: n %d kind %d: %a vs %a\n", n, kind, S, S8); }
        const float lse = logf(S) + mx;
        for (int i = 0; i < n; ++i) lp[i] = x[i] > -INFINITY ? x[i] - lse : -INFINITY;
        for (int i = 0; i < n; ++i) p0[i] = x[i] == -INFINITY ? 0.0f : expf(lp[i]);
        wa_probs_expf8(x.data(), n, lp.data(), p1.data());
        if (memcmp(p0.data(), p1.data(), (size_t) n * 4) != 0) bad_probs++;
    }
    printf("ordered sums: %ld mismatches; probabilities: %ld mismatching rows\n", bad_sum, bad_probs);
    return bad || bad_sum || bad_probs ? 1 : 0;
}
