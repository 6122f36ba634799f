// Where do a graph's kernel arguments live?  A kernel that chases one value out of a 2 KB by-value argument
// (s_load from the kernarg segment) vs the same struct read through a pointer to device memory.
//   hipcc --offload-arch=gfx950 -O3 tools/kernarg_bench.hip -o tools/kernarg_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { int v[512]; };
__global__ void __launch_bounds__(256) k_empty(int *out, int i) { if (i == -1) out[0] = 1; }
__global__ void __launch_bounds__(256) k_val(Big b, const int *tab, int *out, int i) {
    // two dependent scalar reads from the argument block, then one global load whose address depends on them
    const int a = b.v[(i * 7) & 511];
    const int c = b.v[(a + i) & 511];
    const int x = tab[(c + threadIdx.x) & 1023];
    if (x == 123456) out[threadIdx.x] = x;
}
__global__ void __launch_bounds__(256) k_ptr(const Big *b, const int *tab, int *out, int i) {
    const int a = b->v[(i * 7) & 511];
    const int c = b->v[(a + i) & 511];
    const int x = tab[(c + threadIdx.x) & 1023];
    if (x == 123456) out[threadIdx.x] = x;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    Big h; for (int i = 0; i < 512; ++i) h.v[i] = (i * 37) & 511;
    Big *d; int *tab, *out;
    CK(hipMalloc(&d, sizeof(Big))); CK(hipMemcpy(d, &h, sizeof(Big), hipMemcpyHostToDevice));
    CK(hipMalloc(&tab, 4096)); CK(hipMemset(tab, 0, 4096)); CK(hipMalloc(&out, 4096));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int mode = 0; mode < 3; ++mode) {
        const int N = 400;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            if (mode == 0) k_empty<<<256, 256, 0, s>>>(out, i);
            if (mode == 1) k_val<<<256, 256, 0, s>>>(h, tab, out, i);
            if (mode == 2) k_ptr<<<256, 256, 0, s>>>(d, tab, out, i);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-40s %7.3f us/kernel\n", mode == 0 ? "empty" : mode == 1 ? "2 KB by value (kernarg segment)" : "pointer to device memory", ms * 1e3 / (5 * N));
    }
    return 0;
}
