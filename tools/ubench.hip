// Dev micro-benchmark: kernel floor, effective clock, dependent global-load latency on this box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty() {}
__global__ void k_fma_chain(float *out, int n) {
    float x = threadIdx.x * 1e-9f;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) x = fmaf(x, 1.0000001f, 1e-9f);
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = x; ((long long *)out)[1] = t1 - t0; }
}
__global__ void k_chase(const int *next, int steps, int *out) {
    int p = threadIdx.x == 0 ? 0 : 1;
    long long t0 = clock64();
    for (int i = 0; i < steps; ++i) p = next[p];
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = p; ((long long *)out)[1] = t1 - t0; }
}
__global__ void k_memtime(unsigned long long *o) {
    unsigned long long a = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    for (volatile int i = 0; i < 100000; ++i) {}
    unsigned long long b = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
    o[0] = b - a; o[1] = r2 - r;
}
int main() {
    float *out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); for (int i = 0; i < 1000; ++i) k_empty<<<1, 64>>>(); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("empty kernel eager back-to-back: %.2f us/launch\n", ms);
    }
    // graph of 1000 empty kernels
    hipStream_t s; hipStreamCreate(&s); hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 1000; ++i) k_empty<<<1, 64, 0, s>>>();
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("empty kernel in graph: %.2f us/kernel\n", ms);
    }
    int n = 1 << 20;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); k_fma_chain<<<1, 64>>>(out, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        long long cyc; hipMemcpy(&cyc, ((long long *)out) + 1, 8, hipMemcpyDeviceToHost);
        printf("fma chain: %d dependent fma in %.3f ms -> %.2f ns/fma ; clock64 ticks %lld (%.2f per fma)\n", n, ms, ms * 1e6 / n, cyc, (double)cyc / n);
    }
    unsigned long long *mt; hipMalloc(&mt, 16);
    k_memtime<<<1, 1>>>(mt); unsigned long long hm[2]; hipMemcpy(hm, mt, 16, hipMemcpyDeviceToHost);
    printf("memtime ticks %llu realtime ticks(100MHz) %llu -> shader clock %.0f MHz\n", hm[0], hm[1], (double)hm[0] / hm[1] * 100.0);
    // pointer chase over 64 MB (beyond L2, within MALL) and 1 GB (HBM)
    for (size_t bytes : {(size_t)1 << 16, (size_t)1 << 22, (size_t)1 << 26, (size_t)1 << 30}) {
        size_t cnt = bytes / 4; std::vector<int> h(cnt);
        size_t stride = 4099 * 16;  // hop far
        for (size_t i = 0; i < cnt; ++i) h[i] = (int)((i + stride) % cnt);
        int *d; hipMalloc(&d, bytes); hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        int *o; hipMalloc(&o, 64);
        int steps = 2000;
        k_chase<<<1, 1>>>(d, steps, o); hipDeviceSynchronize();
        hipEventRecord(e0); k_chase<<<1, 1>>>(d, steps, o); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        long long cyc; hipMemcpy(&cyc, ((long long *)o) + 1, 8, hipMemcpyDeviceToHost);
        printf("chase %zu MB: %.1f ns/hop (%.0f ticks/hop)\n", bytes >> 20, ms * 1e6 / steps, (double)cyc / steps);
        hipFree(d); hipFree(o);
    }
    return 0;
}
