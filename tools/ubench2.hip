// Dev micro-benchmark: cost of straight-line code size per kernel launch (instruction fetch after a
// kernel boundary) vs the same work in a rolled loop.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __global__ void k_unrolled(float *out, float a) {
    float x = threadIdx.x, y = a;
#pragma unroll
    for (int i = 0; i < N; ++i) { x = fmaf(x, y, (float)i); y = fmaf(y, x, 0.5f); }
    out[threadIdx.x] = x + y;
}
template <int N> __global__ void k_rolled(float *out, float a) {
    float x = threadIdx.x, y = a;
#pragma unroll 1
    for (int i = 0; i < N; ++i) { x = fmaf(x, y, (float)i); y = fmaf(y, x, 0.5f); }
    out[threadIdx.x] = x + y;
}
__global__ void k_other(float *out) { out[threadIdx.x + 64] = threadIdx.x; }
template <typename F> float time_graph(F launch, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 200; ++i) launch(s);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / 200.f;
}
int main() {
    float *out; hipMalloc(&out, 4096); hipStream_t s; hipStreamCreate(&s);
    printf("per-kernel time in a 200-kernel graph (1 block x 64 threads, 2N dependent FMAs):\n");
#define RUN(N) printf("N=%5d unrolled %.2f us   rolled %.2f us   unrolled+other alternating %.2f us/pair\n", N, \
        time_graph([&](hipStream_t st) { k_unrolled<N><<<1, 64, 0, st>>>(out, 1.0001f); }, s), \
        time_graph([&](hipStream_t st) { k_rolled<N><<<1, 64, 0, st>>>(out, 1.0001f); }, s), \
        time_graph([&](hipStream_t st) { k_unrolled<N><<<1, 64, 0, st>>>(out, 1.0001f); k_other<<<1, 64, 0, st>>>(out); }, s) * 1.0f)
    RUN(16); RUN(128); RUN(512); RUN(2048); RUN(8192);
    return 0;
}
