// Exhaustive check of the 8-wide restatement of glibc's expf (wa_expf8.h) against the libm of this machine: every float in [-104, 0]
// (the arguments of the sampling path are log-probabilities and logit - max: never positive), plus -inf / NaN handling.
//   g++ -O2 -mavx2 -mfma -pthread tools/micro/expf_avx2_check.cpp -I whisper-rust_amd/csrc -o /tmp/expf_check && /tmp/expf_check [variant]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include "wa_expf8.h"

int main(int argc, char ** argv) {
    const uint32_t lo = 0x80000000u, hi = 0xC2D00000u;      // -0.0 .. -104.0
    const int nt = 8;
    std::atomic<uint64_t> bad{0};
    std::vector<std::thread> th;
    uint32_t first_bad[nt]; memset(first_bad, 0, sizeof(first_bad));
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() {
        const uint64_t n = (uint64_t) hi - lo + 1, a = lo + n * t / nt, b = lo + n * (t + 1) / nt;
        uint64_t nb = 0;
        for (uint64_t u = a; u + 8 <= b; u += 8) {
            float x[8], y[8];
            for (int k = 0; k < 8; ++k) { const uint32_t v = (uint32_t) (u + k); memcpy(&x[k], &v, 4); }
            wa_expf8(x, y);
            for (int k = 0; k < 8; ++k) {
                const float r = expf(x[k]);
                if (memcmp(&r, &y[k], 4) != 0) { if (!nb) first_bad[t] = (uint32_t) (u + k); nb++; }
            }
        }
        bad += nb;
    });
    for (auto & x : th) x.join();
    printf("mismatches against libm expf over [-104, 0]: %llu\n", (unsigned long long) bad.load());
    for (int t = 0; t < nt; ++t) if (first_bad[t]) { float x; memcpy(&x, &first_bad[t], 4); float y[8], xs[8] = {x,x,x,x,x,x,x,x}; wa_expf8(xs, y); printf("  e.g. x = %a: libm %a, here %a\n", x, expf(x), y[0]); }
    // specials
    float xs[8] = { -INFINITY, -0.0f, 0.0f, -103.9f, -150.0f, -1e30f, -87.5f, -88.5f }, ys[8];
    wa_expf8(xs, ys);
    int sb = 0;
    for (int k = 0; k < 8; ++k) { const float r = expf(xs[k]); if (memcmp(&r, &ys[k], 4) != 0) { sb++; printf("  special x = %a: libm %a, here %a\n", xs[k], r, ys[k]); } }
    printf("specials: %d mismatches\n", sb);
    return bad.load() || sb ? 1 : 0;
}
