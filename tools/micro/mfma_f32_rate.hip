// Microbenchmark: sustained rate of v_mfma_f32_16x16x4_f32 when its A / B operands are produced by F16 -> F32 conversions
// (the inner loop of k_gemm_exact_mfma), against the bare MFMA loop.  Prints SIMD cycles per MFMA for 1, 2 and 3 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_rate.hip -o tools/micro/mfma_f32_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int MODE>      // 0: bare MFMA (operands fixed F32)  1: two conversions per MFMA (low / high halves: plain + SDWA)  2: low halves only
                         // 3: conversions of the whole 8-chunk first, then 8 MFMAs   4: v_cvt via v_pk_mul? (not used)
__global__ __launch_bounds__(256, 3) void k_rate(const half8 * src, float * out, int iters, unsigned long long * cyc) {
    f32x4 acc[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    half8 a[4], b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { a[c] = src[threadIdx.x * 8 + c]; b[c] = src[threadIdx.x * 8 + 4 + c]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (MODE == 3) {
                float fa[8], fb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { fa[i] = (float) a[c][i]; fb[i] = (float) b[c][i]; }
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[c * 8 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[i], acc[c * 8 + i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float x, y;
                    if (MODE == 0) { x = __builtin_bit_cast(float, ((const unsigned *) &a[c])[i & 3]); y = __builtin_bit_cast(float, ((const unsigned *) &b[c])[i & 3]); }
                    else if (MODE == 2) { x = (float) a[c][i & 6]; y = (float) b[c][i & 6]; }
                    else if (MODE == 4) {      // conversion by v_fma_mix_f32 (h * 1.0 + -0.0: exact identity), low / high half by op_sel
                        const unsigned pa = ((const unsigned *) &a[c])[i >> 1], pb = ((const unsigned *) &b[c])[i >> 1];
                        if (i & 1) { asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(x) : "v"(pa), "s"(0x80000000u));
                                     asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(y) : "v"(pb), "s"(0x80000000u)); }
                        else       { asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(x) : "v"(pa), "s"(0x80000000u));
                                     asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(y) : "v"(pb), "s"(0x80000000u)); }
                    } else if (MODE == 5) {    // high half: shift + plain conversion
                        const unsigned pa = ((const unsigned *) &a[c])[i >> 1], pb = ((const unsigned *) &b[c])[i >> 1];
                        const unsigned qa = (i & 1) ? pa >> 16 : pa, qb = (i & 1) ? pb >> 16 : pb;
                        x = (float) __builtin_bit_cast(_Float16, (unsigned short) qa); y = (float) __builtin_bit_cast(_Float16, (unsigned short) qb);
                    }
                    else { x = (float) a[c][i]; y = (float) b[c][i]; }
                    acc[c * 8 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[c * 8 + i], 0, 0, 0);
                }
            }
        }
        // keep the operands loop-variant so that the conversions are not hoisted
#pragma unroll
        for (int c = 0; c < 4; ++c) { asm volatile("" : "+v"(a[c]), "+v"(b[c])); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// exactness of the v_fma_mix_f32 conversion over all 65536 F16 bit patterns (both halves of a dword): bitwise equal to v_cvt_f32_f16 (NaNs by class)
__global__ void k_cvt_check(unsigned * bad) {
    const unsigned h = blockIdx.x * blockDim.x + threadIdx.x;          // 0..65535
    const unsigned p = h | (h << 16);
    float lo, hi;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(p), "s"(0x80000000u));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(p), "s"(0x80000000u));
    const float ref = (float) __builtin_bit_cast(_Float16, (unsigned short) h);
    const unsigned r = __builtin_bit_cast(unsigned, ref), a = __builtin_bit_cast(unsigned, lo), b = __builtin_bit_cast(unsigned, hi);
    const bool nan = ref != ref;
    if (nan ? !(lo != lo && hi != hi) : (a != r || b != r)) atomicAdd(bad, 1u);
}

template <int MODE>
static void run(const char * name, const half8 * src, float * out, unsigned long long * cyc, int blocks_per_cu) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k_rate<MODE>), dim3(grid), dim3(256), 0, 0, src, out, 10, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_rate<MODE>), dim3(grid), dim3(256), 0, 0, src, out, iters, cyc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4096]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < grid; ++i) mean += (double) h[i]; mean /= grid;
    const double mfma_per_wave = 32.0 * iters;
    printf("%-44s %d wave(s)/SIMD: %7.1f shader cycles per MFMA and wave = %6.1f per MFMA and SIMD;  %.1f TFLOP/s\n", name, blocks_per_cu, mean / mfma_per_wave,
           mean / mfma_per_wave / blocks_per_cu, 2.0 * 16 * 16 * 4 * mfma_per_wave * 4 * grid / (ms * 1e-3) / 1e12);
}

int main() {
    half8 * src; float * out; unsigned long long * cyc;
    hipMalloc(&src, 256 * 8 * sizeof(half8)); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
    hipMemset(src, 0x3c, 256 * 8 * sizeof(half8));
    unsigned * bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k_cvt_check, dim3(256), dim3(256), 0, 0, bad);
    unsigned hb = 99; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("v_fma_mix_f32 conversion vs v_cvt_f32_f16 over all 65536 F16 patterns: %u mismatches\n", hb);
    for (int w = 1; w <= 2; ++w) {
        run<0>("bare MFMA", src, out, cyc, w);
        run<1>("2 cvt (plain + SDWA) per MFMA", src, out, cyc, w);
        run<2>("2 cvt (plain only) per MFMA", src, out, cyc, w);
        run<3>("16 cvt then 8 MFMA", src, out, cyc, w);
        run<4>("2 v_fma_mix_f32 conversions per MFMA", src, out, cyc, w);
        run<5>("2 cvt, high halves by shift", src, out, cyc, w);
    }
    return 0;
}
