// Dev micro-benchmark: cost of a device-wide barrier among co-resident workgroups (agent-scope atomics,
// sense reversal) vs the cost of a kernel boundary, on MI355X.  Bounded spins: a bug cannot hang the device.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_sync(unsigned *cnt, unsigned *flag, int nbar, float *sink, long long *fail) {
    unsigned sense = 0;
    float acc = threadIdx.x;
    for (int b = 0; b < nbar; ++b) {
        acc = acc * 1.0001f + 1.0f;  // a little work between barriers
        __syncthreads();
        if (threadIdx.x == 0) {
            sense ^= 1u;
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == gridDim.x - 1) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, sense, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                int spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sense && spins < (1 << 20)) { __builtin_amdgcn_s_sleep(2); ++spins; }
                if (spins >= (1 << 20)) atomicAdd((unsigned long long *)fail, 1ull);
            }
        }
        __syncthreads();
    }
    if (acc == -1.f) sink[0] = acc;
}
__global__ void k_small(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1] += 1.f; }
int main() {
    unsigned *cnt, *flag; float *sink; long long *fail;
    hipMalloc(&cnt, 4); hipMalloc(&flag, 4); hipMalloc(&sink, 64); hipMalloc(&fail, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {64, 256, 512, 768}) {
        hipMemset(cnt, 0, 4); hipMemset(flag, 0, 4); hipMemset(fail, 0, 8);
        k_sync<<<grid, 256>>>(cnt, flag, 10, sink, fail); hipDeviceSynchronize();
        hipMemset(cnt, 0, 4); hipMemset(flag, 0, 4);
        const int nbar = 200;
        hipEventRecord(e0); k_sync<<<grid, 256>>>(cnt, flag, nbar, sink, fail); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long f; hipMemcpy(&f, fail, 8, hipMemcpyDeviceToHost);
        printf("grid %4d: %.2f us per barrier (%d barriers, %lld timeouts)\n", grid, ms * 1000.f / nbar, nbar, f);
    }
    hipGraph_t g; hipGraphExec_t ge; hipStream_t s; hipStreamCreate(&s);
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 200; ++i) k_small<<<256, 256, 0, s>>>(sink);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel boundary in a graph (256 x 256 trivial kernel): %.2f us per kernel\n", ms * 1000.f / 200);
    return 0;
}
