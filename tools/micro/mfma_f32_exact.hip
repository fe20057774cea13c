// Microtest: is the f32-input MFMA a bitwise fmaf chain in ascending k?  (MI355X_MICROARCH.md: "exact f32 (= fmaf chain, bitwise)".)
// The reference-order GEMM (ggml_vec_dot_f16, vec.cpp:191-231) is 32 independent chains acc = fmaf(x, y, acc) in k order; if
// D = MFMA(A, B, C) equals fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c)))) bit for bit, a chain maps onto one accumulator tile.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/micro/mfma_f32_exact.hip -o gpurun_out/mfma_f32_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one wave: D[16][16] = A[16][4] B[4][16] + C ; nrep chained instructions (K = 4 nrep): the accumulator of one feeds the next
__global__ void k_mfma16(const float * A, const float * B, const float * C, float * D, int nrep) {
    const int l = threadIdx.x;
    f32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = C[(4 * (l / 16) + i) * 16 + (l % 16)];
    for (int r = 0; r < nrep; ++r) {
        const float a = A[(size_t) r * 64 + (l % 16) * 4 + (l / 16)];        // A[r][m][k]
        const float b = B[(size_t) r * 64 + (l / 16) * 16 + (l % 16)];       // B[r][k][n]
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) D[(4 * (l / 16) + i) * 16 + (l % 16)] = acc[i];
}
// D[32][32] = A[32][2] B[2][32] + C
__global__ void k_mfma32(const float * A, const float * B, const float * C, float * D, int nrep) {
    const int l = threadIdx.x;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = C[(8 * (i / 4) + 4 * (l / 32) + (i % 4)) * 32 + (l % 32)];
    for (int r = 0; r < nrep; ++r) {
        const float a = A[(size_t) r * 64 + (l % 32) * 2 + (l / 32)];        // A[r][m][k]
        const float b = B[(size_t) r * 64 + (l / 32) * 32 + (l % 32)];       // B[r][k][n]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) D[(8 * (i / 4) + 4 * (l / 32) + (i % 4)) * 32 + (l % 32)] = acc[i];
}

// v_mfma_f32_16x16x1_4b_f32: four independent 16 x 16 blocks, K = 1 (a rank-1 update each).  Checks the layout the score kernel
// assumes (lane l feeds block l / 16 with A row / B column l % 16; block b's result in registers 4 b .. 4 b + 3, row 4 (l / 16) + j,
// column l % 16) and that an update is one fmaf.  nrep chained updates per block.
typedef float f32x16v __attribute__((ext_vector_type(16)));
__global__ void k_mfma16x1(const float * A, const float * B, const float * C, float * D, int nrep) {
    const int l = threadIdx.x;
    f32x16v acc;
    for (int b = 0; b < 4; ++b) for (int j = 0; j < 4; ++j) acc[4 * b + j] = C[b * 256 + (4 * (l / 16) + j) * 16 + (l % 16)];
    for (int r = 0; r < nrep; ++r) {
        const float a = A[(size_t) r * 64 + l];        // A[r][block][row]
        const float b = B[(size_t) r * 64 + l];        // B[r][block][col]
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc, 0, 0, 0);
    }
    for (int b = 0; b < 4; ++b) for (int j = 0; j < 4; ++j) D[b * 256 + (4 * (l / 16) + j) * 16 + (l % 16)] = acc[4 * b + j];
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float half_like(int spread) {       // a value an F16 can hold: 11-bit significand, exponent within +-spread
    const int m = (int) (rnd() % 2048) + 1024;          // [1024, 3071]: up to 12 bits - keep 11
    const int e = (int) (rnd() % (2 * spread + 1)) - spread;
    const float v = ldexpf((float) (m & ~1), e - 11);
    return (rnd() & 1) ? -v : v;
}
static float any_f32(int spread) {
    const float v = ldexpf(1.0f + (float) (rnd() & 0x7fffff) / 8388608.0f, (int) (rnd() % (2 * spread + 1)) - spread);
    return (rnd() & 1) ? -v : v;
}

int main() {
    const int NREP = 24, TRIALS = 400;
    float * dA, * dB, * dC, * dD;
    hipMalloc(&dA, NREP * 64 * 4); hipMalloc(&dB, NREP * 64 * 4); hipMalloc(&dC, 1024 * 4); hipMalloc(&dD, 1024 * 4);
    for (int shape = 0; shape < 2; ++shape) {
        const int MN = shape == 0 ? 16 : 32, KK = shape == 0 ? 4 : 2;
        for (int mode = 0; mode < 2; ++mode) {       // 0: F16-representable operands (the use case), 1: arbitrary F32
            long n = 0, bad_asc = 0, bad_desc = 0, bad_nofma = 0, bad_pair = 0;
            for (int t = 0; t < TRIALS; ++t) {
                std::vector<float> A(NREP * 64), B(NREP * 64), Cc(MN * MN), D(MN * MN);
                const int spread = 1 + t % 12;
                for (auto & v : A) v = mode ? any_f32(spread) : half_like(spread);
                for (auto & v : B) v = mode ? any_f32(spread) : half_like(spread);
                for (auto & v : Cc) v = (t & 1) ? 0.0f : any_f32(spread);
                hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
                hipMemcpy(dC, Cc.data(), Cc.size() * 4, hipMemcpyHostToDevice);
                if (shape == 0) hipLaunchKernelGGL(k_mfma16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, NREP);
                else            hipLaunchKernelGGL(k_mfma32, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, NREP);
                hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
                for (int m = 0; m < MN; ++m) for (int nn = 0; nn < MN; ++nn) {
                    float asc = Cc[m * MN + nn], desc = asc, nofma = asc, pair = asc;
                    for (int r = 0; r < NREP; ++r) {
                        float p[4];
                        for (int k = 0; k < KK; ++k) p[k] = 0;
                        for (int k = 0; k < KK; ++k) asc = fmaf(A[r * 64 + m * KK + k], B[r * 64 + k * MN + nn], asc);
                        for (int k = KK - 1; k >= 0; --k) desc = fmaf(A[r * 64 + m * KK + k], B[r * 64 + k * MN + nn], desc);
                        for (int k = 0; k < KK; ++k) { volatile float pr = A[r * 64 + m * KK + k] * B[r * 64 + k * MN + nn]; nofma = nofma + pr; }
                        { double s = 0; for (int k = 0; k < KK; ++k) s += (double) A[r * 64 + m * KK + k] * (double) B[r * 64 + k * MN + nn]; pair = (float) ((double) pair + s); }
                    }
                    uint32_t ud, ua, ue, un, up; const float dv = D[m * MN + nn];
                    memcpy(&ud, &dv, 4); memcpy(&ua, &asc, 4); memcpy(&ue, &desc, 4); memcpy(&un, &nofma, 4); memcpy(&up, &pair, 4);
                    ++n; bad_asc += ud != ua; bad_desc += ud != ue; bad_nofma += ud != un; bad_pair += ud != up;
                }
            }
            printf("mfma_f32_%dx%dx%d %s operands, K = %d chained: %ld outputs; mismatches vs ascending fmaf chain %ld, descending %ld, "
                   "mul+add %ld, exact-dot-per-instruction %ld\n", MN, MN, KK, mode ? "arbitrary F32" : "F16-valued", NREP * KK, n, bad_asc, bad_desc,
                   bad_nofma, bad_pair);
        }
    }
    {   // the 4-block K = 1 form
        long n = 0, bad = 0;
        for (int t = 0; t < 200; ++t) {
            const int NR = 6;
            std::vector<float> A(NR * 64), B(NR * 64), Cc(1024), D(1024);
            const int spread = 1 + t % 12;
            for (auto & v : A) v = half_like(spread);
            for (auto & v : B) v = half_like(spread);
            for (auto & v : Cc) v = (t & 1) ? 0.0f : any_f32(spread);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dC, Cc.data(), Cc.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_mfma16x1, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, NR);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            for (int b = 0; b < 4; ++b) for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                float acc = Cc[b * 256 + i * 16 + j];
                for (int r = 0; r < NR; ++r) acc = fmaf(A[r * 64 + b * 16 + i], B[r * 64 + b * 16 + j], acc);
                uint32_t x, y; const float dv = D[b * 256 + i * 16 + j];
                memcpy(&x, &dv, 4); memcpy(&y, &acc, 4);
                ++n; bad += x != y;
            }
        }
        printf("mfma_f32_16x16x1 (4 blocks, K = 1) chained 6 deep: %ld outputs; mismatches vs fmaf chain per block %ld\n", n, bad);
    }
    return 0;
}
