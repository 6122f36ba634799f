// Dev micro-benchmark: which XCD does block b of a launch land on?  (speed-only knowledge: MI355X guide says blocks are dealt
// round-robin over the 8 XCDs but the XCD of block 0 is not fixed.)  Graph of kernels with the grid sizes of one SAC1 update.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(int *out, int slot) {
    if (threadIdx.x == 0) {
        int x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[slot * 1024 + blockIdx.x] = x & 0xf;
    }
    // some work so that launches overlap as little / as much as real ones
    float v = threadIdx.x;
    for (int i = 0; i < 2000; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) out[0] = 1;
}
int main() {
    const int grids[5] = {400, 240, 312, 361, 280};
    const int REP = 6;
    int *out; hipMalloc(&out, 5 * REP * 1024 * 4); hipMemset(out, 0xff, 5 * REP * 1024 * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int r = 0; r < REP; ++r) for (int i = 0; i < 5; ++i) k<<<grids[i], 256, 0, s>>>(out, r * 5 + i);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int it = 0; it < 3; ++it) {
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        std::vector<int> h(5 * REP * 1024);
        hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        printf("graph launch %d: per kernel [xcc of block 0 | #blocks violating xcc(b) == (xcc(0) + b) %% 8]\n ", it);
        for (int sl = 0; sl < 5 * REP; ++sl) {
            const int gsz = grids[sl % 5], x0 = h[sl * 1024];
            int bad = 0;
            for (int b = 0; b < gsz; ++b) if (h[sl * 1024 + b] != (x0 + b) % 8) ++bad;
            printf(" %d|%d", x0, bad);
        }
        printf("\n");
    }
    return 0;
}
