// What survives a kernel boundary in an XCD's L2?  (MI355X: 8 XCDs x 4 MiB L2, blocks dealt round-robin: block b runs on XCD b % 8.)
// Graph: touch -> [K tiny kernels] -> [filler: F bytes per XCD streamed through the L2] -> probe.
//   touch  : block b writes (W) or only reads (R) its 16 KB region (a pointer chain), 32 blocks = 512 KB per XCD
//   filler : block b streams its slice of a separate buffer (plain or nontemporal loads), F bytes per XCD in all
//   probe  : block b chases the chain of region (b + shift) % 256 with one lane: cycles per dependent load
//            (shift 0: the XCD that touched it; shift 1: a neighbour)   ~230 = L2 hit, ~550+ = Infinity Cache / HBM
// Also a staleness check: XCD y caches a region by reading it, XCD x rewrites it in the next kernel, y probes in the third.
// hipcc --offload-arch=gfx950 -O3 tools/l2_persist_bench.hip -o tools/l2_persist_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int REG = 4096;   // ints per region (16 KB)
constexpr int NWG = 256, HOPS = 16, STRIDE = 32 * 5 + 32;  // hop = a different 128-B line each time
__global__ void __launch_bounds__(256) k_touch(int *buf, int write, int salt, int shift, int *sink) {
    int *r = buf + (size_t)((blockIdx.x + shift) % NWG) * REG;
    int acc = 0;
    for (int i = threadIdx.x; i < REG; i += 256) {
        if (write) r[i] = (i + STRIDE + salt * 32) % REG;
        else acc += r[i];
    }
    if (acc == -12345) *sink = acc;
}
__global__ void k_tiny(int *sink) { if (threadIdx.x == 999) *sink = 1; }
__global__ void __launch_bounds__(256) k_fill(const int4 *big, size_t int4_per_block, int nt, int *sink) {
    const int4 *p = big + (size_t)blockIdx.x * int4_per_block;
    int acc = 0;
    for (size_t i = threadIdx.x; i < int4_per_block; i += 256) {
        typedef int i4v __attribute__((ext_vector_type(4)));
        i4v v;
        if (nt) v = __builtin_nontemporal_load(reinterpret_cast<const i4v *>(p + i));
        else v = *reinterpret_cast<const i4v *>(p + i);
        acc += v.x + v.w;
    }
    if (acc == -12345) *sink = acc;
}
__global__ void __launch_bounds__(64) k_probe(const int *buf, int shift, unsigned long long *out, int *last, int *sink) {
    const int *r = buf + (size_t)((blockIdx.x + shift) % NWG) * REG;
    if (threadIdx.x != 0) return;
    int i = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int h = 0; h < HOPS; ++h) i = r[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x] = (t1 - t0) / HOPS;
    last[blockIdx.x] = i;
    if (i == -1) *sink = i;
}
struct Ctx { int *buf, *sink, *last; int4 *big; unsigned long long *out; hipStream_t s; };
static int run(Ctx &c, const char *name, int write, int K, double fill_mb_per_xcd, int nt, int shift) {
    hipGraph_t g; hipGraphExec_t ge;
    const size_t per_block = (size_t)(fill_mb_per_xcd * 1048576.0 / 32.0 / 16.0);  // int4 per block (32 blocks per XCD)
    CK(hipStreamBeginCapture(c.s, hipStreamCaptureModeThreadLocal));
    for (int it = 0; it < 6; ++it) {
        k_touch<<<NWG, 256, 0, c.s>>>(c.buf, write, it, 0, c.sink);
        for (int k = 0; k < K; ++k) k_tiny<<<NWG, 64, 0, c.s>>>(c.sink);
        if (per_block) k_fill<<<NWG, 256, 0, c.s>>>(c.big, per_block, nt, c.sink);
        k_probe<<<NWG, 64, 0, c.s>>>(c.buf, shift, c.out, c.last, c.sink);
    }
    CK(hipStreamEndCapture(c.s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, c.s)); CK(hipGraphLaunch(ge, c.s)); CK(hipStreamSynchronize(c.s));
    std::vector<unsigned long long> h(NWG);
    CK(hipMemcpy(h.data(), c.out, NWG * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-34s touch %s, %d tiny kernels, fill %4.1f MB/XCD %-5s probe shift %d: median %4llu  p10 %4llu  p90 %4llu cycles/load\n", name, write ? "W" : "R", K,
           fill_mb_per_xcd, per_block ? (nt ? "(nt)" : "(ld)") : "", shift, h[NWG / 2], h[NWG / 10], h[NWG * 9 / 10]);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return 0;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    Ctx c{};
    CK(hipMalloc(&c.buf, (size_t)NWG * REG * 4)); CK(hipMalloc(&c.out, NWG * 8)); CK(hipMalloc(&c.sink, 4)); CK(hipMalloc(&c.last, NWG * 4));
    CK(hipMalloc(&c.big, (size_t)128 << 20)); CK(hipMemset(c.big, 1, (size_t)128 << 20));
    CK(hipStreamCreate(&c.s));
    k_touch<<<NWG, 256, 0, c.s>>>(c.buf, 1, 0, 0, c.sink);
    CK(hipStreamSynchronize(c.s));
    for (int shift : {0, 1}) {
        run(c, "written, next kernel", 1, 0, 0, 0, shift);
        run(c, "written, 4 boundaries later", 1, 4, 0, 0, shift);
        run(c, "read only, next kernel", 0, 0, 0, 0, shift);
        run(c, "read only, 4 boundaries later", 0, 4, 0, 0, shift);
    }
    for (double f : {0.5, 1.0, 2.0, 3.0, 4.0, 6.0, 8.0}) {
        run(c, "written, then plain loads", 1, 0, f, 0, 0);
        run(c, "written, then nt loads", 1, 0, f, 1, 0);
    }
    for (double f : {1.0, 2.0, 3.0, 4.0}) run(c, "read only, then plain loads", 0, 0, f, 0, 0);
    // staleness: every XCD caches its neighbour's region by reading it; the owner rewrites it (salt changes the chain); the
    // neighbour probes: the end of the chain must be the NEW one
    {
        std::vector<int> want(NWG), got(NWG);
        int bad = 0;
        for (int it = 0; it < 8; ++it) {
            k_touch<<<NWG, 256, 0, c.s>>>(c.buf, 0, 0, 1, c.sink);           // block b reads region b + 1 (another XCD's)
            k_touch<<<NWG, 256, 0, c.s>>>(c.buf, 1, 100 + it, 0, c.sink);    // owners rewrite
            k_probe<<<NWG, 64, 0, c.s>>>(c.buf, 1, c.out, c.last, c.sink);   // the neighbour reads again
            CK(hipStreamSynchronize(c.s));
            CK(hipMemcpy(got.data(), c.last, NWG * 4, hipMemcpyDeviceToHost));
            int i = 0;
            for (int h = 0; h < HOPS; ++h) i = (i + STRIDE + (100 + it) * 32) % REG;
            for (int b = 0; b < NWG; ++b) if (got[b] != i) ++bad;
        }
        std::vector<unsigned long long> h(NWG);
        CK(hipMemcpy(h.data(), c.out, NWG * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("staleness: neighbour cached the region, owner rewrote it, neighbour probed: %d stale chains of %d; probe median %llu cycles/load\n", bad, 8 * NWG, h[NWG / 2]);
    }
    return 0;
}
