// Does a kernel that starts with a cold instruction cache pay for its code size?  Graph of launches of
// straight-line kernels (REPT scalar adds = 4 * REPT bytes of code), either the SAME kernel repeated
// (code stays in the 64 KB instruction cache shared by two CUs) or a rotation of NK distinct copies
// whose total size exceeds it.   hipcc --offload-arch=gfx950 -O3 tools/icache_bench.hip -o /tmp/icache_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ID, int REPT>
__global__ void __launch_bounds__(256) k_line(int *out, int v) {
    int x = v + ID;
    if (REPT == 1024) asm volatile(".rept 1024\n s_add_u32 %0, %0, 1\n .endr" : "+s"(x));
    if (REPT == 2048) asm volatile(".rept 2048\n s_add_u32 %0, %0, 1\n .endr" : "+s"(x));
    if (REPT == 4096) asm volatile(".rept 4096\n s_add_u32 %0, %0, 1\n .endr" : "+s"(x));
    if (REPT == 8192) asm volatile(".rept 8192\n s_add_u32 %0, %0, 1\n .endr" : "+s"(x));
    if (x == 12345) out[threadIdx.x] = x;
}
// same instruction count, tiny code: a loop
template <int REPT>
__global__ void __launch_bounds__(256) k_loop(int *out, int v) {
    int x = v;
    for (int i = 0; i < REPT / 64; ++i) asm volatile(".rept 64\n s_add_u32 %0, %0, 1\n .endr" : "+s"(x));
    if (x == 12345) out[threadIdx.x] = x;
}

typedef void (*kfn)(int *, int);
template <int REPT> static void fill(std::vector<kfn> &v) {
    v = {k_line<0, REPT>, k_line<1, REPT>, k_line<2, REPT>, k_line<3, REPT>, k_line<4, REPT>, k_line<5, REPT>, k_line<6, REPT>, k_line<7, REPT>};
}

static int run(const char *name, std::vector<kfn> ks, int nk, int *d, hipStream_t s) {
    const int N = 400;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) ks[i % nk]<<<256, 256, 0, s>>>(d, i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %7.3f us/kernel\n", name, ms * 1e3 / (5 * N));
    return 0;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int *d; CK(hipMalloc(&d, 4096));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<kfn> k;
    char nm[96];
#define CASE(R) fill<R>(k); \
    for (int nk : {1, 2, 4, 8}) { snprintf(nm, sizeof nm, "%5d B straight-line, rotation of %d", 4 * R, nk); if (run(nm, k, nk, d, s)) return 1; } \
    { std::vector<kfn> l = {k_loop<R>}; snprintf(nm, sizeof nm, "%5d instructions as a 64-long loop", R); if (run(nm, l, 1, d, s)) return 1; }
    CASE(1024) CASE(2048) CASE(4096) CASE(8192)
    return 0;
}
