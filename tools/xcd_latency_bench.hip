// Load-to-use latency of data the PREVIOUS kernel of a graph wrote on another XCD, by store flavour.
// Producer: workgroup b writes a 64 KB region (a pointer chain: element i holds the index of the next element, stride 128 B).
// Consumer (next graph node): workgroup b chases the chain of region (b + shift) % nwg with one lane: 16 dependent loads,
// cycles per load by s_memtime.  shift = 0: the consumer runs where the producer ran (same XCD: blockIdx % 8);  shift = 1:
// the neighbouring XCD wrote it.   hipcc --offload-arch=gfx950 -O3 tools/xcd_latency_bench.hip -o tools/xcd_latency_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int REG = 16384;  // floats per region (64 KB)
constexpr int HOPS = 16, STRIDE = 32 * 17;  // elements between hops (prime-ish multiple of a 128-B line)
template <int MODE>
__global__ void __launch_bounds__(256) k_prod(int *buf, int salt) {
    int *r = buf + (size_t)blockIdx.x * REG;
    for (int i = threadIdx.x; i < REG; i += 256) {
        const int nxt = (i + STRIDE + salt) % REG;
        if (MODE == 0) r[i] = nxt;
        if (MODE == 1) __builtin_nontemporal_store(nxt, r + i);
        if (MODE == 2) __hip_atomic_store(r + i, nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(r + i), "v"(nxt) : "memory");
        if (MODE == 4) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(r + i), "v"(nxt) : "memory");
        if (MODE == 5) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(r + i), "v"(nxt) : "memory");
    }
}
template <int LMODE>
__global__ void __launch_bounds__(64) k_cons(const int *buf, int shift, unsigned long long *out, int *sink) {
    const int *r = buf + (size_t)((blockIdx.x + shift) % gridDim.x) * REG;
    if (threadIdx.x != 0) return;
    int i = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int h = 0; h < HOPS; ++h) {
        if (LMODE == 0) i = r[i];
        if (LMODE == 1) i = __hip_atomic_load(r + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x] = (t1 - t0) / HOPS;
    if (i == -1) *sink = i;
}
template <int MODE, int LMODE>
static int run(const char *name, int shift, int *buf, unsigned long long *out, int *sink, hipStream_t s) {
    const int NWG = 256;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int it = 0; it < 20; ++it) {
        k_prod<MODE><<<NWG, 256, 0, s>>>(buf, it & 7);
        k_cons<LMODE><<<NWG, 64, 0, s>>>(buf, shift, out, sink);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(NWG);
    CK(hipMemcpy(h.data(), out, NWG * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-58s shift %d: cycles per dependent load  median %5llu  p10 %5llu  p90 %5llu\n", name, shift, h[NWG / 2], h[NWG / 10], h[NWG * 9 / 10]);
    return 0;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int *buf, *sink; unsigned long long *out;
    CK(hipMalloc(&buf, (size_t)256 * REG * 4)); CK(hipMalloc(&out, 256 * 8)); CK(hipMalloc(&sink, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int shift : {0, 1}) {
        if (run<0, 0>("plain stores, plain loads", shift, buf, out, sink, s)) return 1;
        if (run<1, 0>("nontemporal stores, plain loads", shift, buf, out, sink, s)) return 1;
        if (run<2, 0>("agent-scope atomic (write-through) stores, plain loads", shift, buf, out, sink, s)) return 1;
        if (run<3, 0>("sc1 stores (asm), plain loads", shift, buf, out, sink, s)) return 1;
        if (run<4, 0>("sc0 sc1 stores (asm), plain loads", shift, buf, out, sink, s)) return 1;
        if (run<5, 0>("sc0 stores (asm), plain loads", shift, buf, out, sink, s)) return 1;
        if (run<0, 1>("plain stores, agent-scope atomic loads", shift, buf, out, sink, s)) return 1;
        if (run<1, 1>("nontemporal stores, agent-scope atomic loads", shift, buf, out, sink, s)) return 1;
    }
    return 0;
}
