// Dev micro-benchmark: does read-only data stay warm in the XCD L2s across dependent kernel boundaries inside a hipGraph?
// Each kernel: every workgroup reads `per_wg` bytes of a region (one round of float4 loads, all in flight) and writes 16 B.
//   same   : every launch reads the same region            (L2-warm if the boundary does not invalidate)
//   rotate : launch i reads region i % 4 (128 MB in all)  (L2-cold, Infinity-Cache-warm)
// hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 tools/l2warm_bench.hip -o tools/l2warm_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NL>
__global__ void __launch_bounds__(256) k(const float4 *in, float4 *out, int wg_stride4) {
    const float4 *p = in + (size_t)blockIdx.x * wg_stride4 + threadIdx.x;
    float4 v[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) v[q] = p[q * 256];
    float4 s = v[0];
#pragma unroll
    for (int q = 1; q < NL; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const size_t REG = 32u << 20;  // 32 MB per region (256 workgroups x up to 128 KB)
    float4 *buf, *out;
    hipMalloc(&buf, REG * 4); hipMemset(buf, 0, REG * 4); hipMalloc(&out, 256 * 256 * 16);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int CH = 512;
    auto run = [&](const char *name, int nl, bool rotate, bool shared) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < CH; ++i) {
            const float4 *in = buf + (rotate ? (size_t)(i % 4) * (REG / 16) : 0);
            const int stride = shared ? 0 : nl * 256;  // shared: every workgroup reads the same bytes
            if (nl == 2) k<2><<<256, 256, 0, s>>>(in, out, stride); else if (nl == 8) k<8><<<256, 256, 0, s>>>(in, out, stride); else k<32><<<256, 256, 0, s>>>(in, out, stride);
        }
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        float best = 1e9f, ms;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-34s %3d KB/wg: %6.2f us/kernel\n", name, nl * 4, best * 1000.f / CH);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    };
    for (int nl : {2, 8, 32}) {
        run("same region, private slices", nl, false, false);
        run("rotating regions, private slices", nl, true, false);
        run("same region, all wgs same bytes", nl, false, true);
        run("rotating, all wgs same bytes", nl, true, true);
    }
    return 0;
}
