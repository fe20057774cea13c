// Microbenchmark: what does a dependent kernel boundary / a dependent global round trip cost on this box?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty(int * p) { if (threadIdx.x == 9999) p[0] = 1; }
__global__ void k_chain(const int * __restrict__ idx, int n, int * out) {   // n dependent loads by one lane
    int i = 0;
    for (int k = 0; k < n; ++k) i = idx[i];
    if (threadIdx.x == 0) out[0] = i;
}
__global__ void k_clock(unsigned long long * out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0001f, 0.5f);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long) a; }
}
int main() {
    int * d; hipMalloc(&d, 1 << 26);
    std::vector<int> h(1 << 24);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (int) ((i * 1048583ull + 12345) % h.size());   // pseudo-random chain, 64 MB
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int * o; hipMalloc(&o, 64);
    unsigned long long * c; hipMalloc(&c, 64);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, s);
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, o);
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel chain (eager): %.2f us per kernel\n", ms);
    }
    {   // graph of 1000 empty kernels
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, o);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel chain (graph): %.2f us per kernel\n", ms);
        hipEventRecord(e0, s);
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(48), dim3(128), 0, s, o);
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("48x128 empty kernel chain (eager): %.2f us per kernel\n", ms);
    }
    for (int n : {1, 2, 4, 8, 64}) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s, d, n, o);
        hipEventRecord(e0, s);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s, d, n, o);
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("kernel with %2d dependent random loads (64 MB table): %.2f us per kernel\n", n, ms * 1000 / 200);
    }
    for (int it : {1000, 100000}) {
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, s, c, it);
        hipStreamSynchronize(s);
        unsigned long long hc[3]; hipMemcpy(hc, c, 24, hipMemcpyDeviceToHost);
        printf("clock probe iters=%d: shader cycles=%llu realtime ticks(100MHz)=%llu -> %.0f MHz\n", it, hc[0], hc[1], hc[1] ? 100.0 * hc[0] / hc[1] : 0.0);
    }
    return 0;
}
