// Microtest for a 32-lanes-per-row GEMV row in ggml_vec_dot_f16 order: chain-major weights + chain-major activations in LDS (rotated chunks),
// one chain per lane, wa_tree32 over the lanes.  Compares with a CPU restatement (fmaf chains + tree).  Build: hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off -fno-slp-vectorize tools/micro/dot32_check.hip -o gpurun_out/dot32_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef uint16_t wa_f16;
typedef _Float16 h16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ int xt_pos(int chunk, int u, int nch) { const int p = chunk + (u >> 2); return p >= nch ? p - nch : p; }

// one workgroup of 64 threads = two rows; granules = pairs of F16 activations (element 2i low, 2i + 1 high)
template <int NS>
__global__ void k_dot32(const wa_f16 * Wt, const unsigned * gran, float * out, int rows) {
    __shared__ __attribute__((aligned(16))) wa_f16 xT[32 * NS];
    const int lane = threadIdx.x, u = lane & 31, nch = NS >> 3;
    for (int i = lane; i < 16 * NS; i += 64) {            // gather
        const unsigned v = gran[i];
        const int u0 = (2 * i) & 31, j = i >> 4;
        const int off = 8 * xt_pos(j >> 3, u0, nch) + (j & 7);
        xT[u0 * NS + off] = (wa_f16) (v & 0xffffu);
        xT[(u0 + 1) * NS + off] = (wa_f16) (v >> 16);
    }
    __syncthreads();
    const int row = blockIdx.x * 2 + (lane >> 5);
    unsigned pf[96];
    const wa_f16 * wrow = Wt + ((size_t) (row < rows ? row : 0) * 32 + u) * NS;
#pragma unroll
    for (int c = 0; c < 24; ++c) if (8 * c < NS) { const u32x4 w = *(const u32x4 *) (wrow + 8 * c); pf[4 * c] = w.x; pf[4 * c + 1] = w.y; pf[4 * c + 2] = w.z; pf[4 * c + 3] = w.w; }
    const wa_f16 * xrow = xT + u * NS;
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 24; ++c) if (8 * c < NS) {
        const half8 x = *(const half8 *) (xrow + 8 * xt_pos(c, u, nch));
        u32x4 wq; wq.x = pf[4 * c]; wq.y = pf[4 * c + 1]; wq.z = pf[4 * c + 2]; wq.w = pf[4 * c + 3];
        const half8 w = __builtin_bit_cast(half8, wq);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float) w[e], (float) x[e], acc);
    }
    float v = acc + __shfl_xor(acc, 16, 64);
    v = v + dpp_f32<0x108>(v);
    v = v + dpp_f32<0x104>(v);
    v = v + dpp_f32<0x101>(v);
    v = v + dpp_f32<0x102>(v);
    if (u == 0 && row < rows) out[row] = v;
    out[rows + blockIdx.x * 64 + lane] = acc;           // the chains, for localisation
}
static float h2f(wa_f16 h) { h16 x; memcpy(&x, &h, 2); return (float) x; }
static wa_f16 f2h(float f) { h16 x = (h16) f; wa_f16 h; memcpy(&h, &x, 2); return h; }
int main() {
    const int NS = 128, K = NS * 32, rows = 6;
    std::vector<wa_f16> W(rows * K), Wt(rows * K), x(K);
    uint32_t s = 1234567u; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto & v : W) v = f2h(rnd()); for (auto & v : x) v = f2h(rnd() * 4.0f);
    for (int r = 0; r < rows; ++r) for (int k = 0; k < K; ++k) Wt[((size_t) r * 32 + (k & 31)) * NS + (k >> 5)] = W[(size_t) r * K + k];
    std::vector<unsigned> gran(K / 2); for (int i = 0; i < K / 2; ++i) gran[i] = x[2 * i] | ((unsigned) x[2 * i + 1] << 16);
    wa_f16 * dWt; unsigned * dg; float * dout;
    hipMalloc(&dWt, Wt.size() * 2); hipMalloc(&dg, gran.size() * 4); hipMalloc(&dout, (rows + 3 * 64) * 4);
    hipMemcpy(dWt, Wt.data(), Wt.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dg, gran.data(), gran.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dot32<NS>, dim3(3), dim3(64), 0, 0, dWt, dg, dout, rows);
    std::vector<float> out(rows + 3 * 64); hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0, badc = 0;
    for (int r = 0; r < rows; ++r) {
        float sp[32];
        for (int p = 0; p < 32; ++p) { float a = 0.0f; for (int j = 0; j < NS; ++j) a = fmaf(h2f(W[(size_t) r * K + p + 32 * j]), h2f(x[p + 32 * j]), a); sp[p] = a; }
        float a8[8]; for (int l = 0; l < 8; ++l) a8[l] = (sp[l] + sp[16 + l]) + (sp[8 + l] + sp[24 + l]);
        const float t0 = a8[0] + a8[4], t1 = a8[1] + a8[5], t2 = a8[2] + a8[6], t3 = a8[3] + a8[7];
        const float ref = (t0 + t1) + (t2 + t3);
        uint32_t ua, ub; memcpy(&ua, &ref, 4); memcpy(&ub, &out[r], 4);
        if (ua != ub) { ++bad; printf("row %d: gpu %.9g ref %.9g\n", r, out[r], ref); }
        for (int p = 0; p < 32; ++p) { const float g = out[rows + (r / 2) * 64 + (r & 1) * 32 + p]; uint32_t x1, x2; memcpy(&x1, &g, 4); memcpy(&x2, &sp[p], 4); if (x1 != x2) { if (badc < 8) printf("  row %d chain %d: gpu %.9g ref %.9g\n", r, p, g, sp[p]); ++badc; } }
    }
    printf("dot32_check: %d of %d rows differ, %d chains differ\n", bad, rows, badc);
    return bad != 0;
}
