// Dev micro-benchmark: what does one kernel of a DEPENDENT chain cost inside a hipGraph on this box, as a function of
// what it does first?  Each kernel reads what the previous one wrote (different buffer every time) and writes its own.
//   empty        : nothing
//   ld-scalar    : 1 round of loads whose addresses come from preloaded scalar arguments, then stores
//   ld-struct    : the same, but the pointer sits in a by-value struct (kernarg s_load first)
//   ld2          : two dependent rounds of loads (index -> data)
//   ld-mfma N    : one round of loads, then N dependent v_mfma_f32_32x32x2_f32, then stores
// hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 tools/chain_bench.hip -o tools/chain_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
struct P { const float *in; float *out; int n; int pad[29]; };

__global__ void k_empty() {}
template <int NL, int NM>
__global__ void __launch_bounds__(256) k_sc(const float *in, float *out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) v[q] = reinterpret_cast<const float4 *>(in)[(i + q * 65536) % n];
    float4 s = v[0];
#pragma unroll
    for (int q = 1; q < NL; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    if (NM > 0) {
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = s.x;
#pragma unroll 1
        for (int m = 0; m < NM; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s.y, s.z, acc, 0, 0, 0);
        s.x = acc[0] + acc[7];
    }
    reinterpret_cast<float4 *>(out)[i] = s;
}
template <int NL>
__global__ void __launch_bounds__(256) k_st(P p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) v[q] = reinterpret_cast<const float4 *>(p.in)[(i + q * 65536) % p.n];
    float4 s = v[0];
#pragma unroll
    for (int q = 1; q < NL; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    reinterpret_cast<float4 *>(p.out)[i] = s;
}
template <int NL>
__global__ void __launch_bounds__(256) k_l2(const float *in, float *out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float4 a = reinterpret_cast<const float4 *>(in)[i];
    const int j = ((int)(a.x * 0.f) + i * 7 + 13) % n;  // address depends on the first load
    float4 v[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) v[q] = reinterpret_cast<const float4 *>(in)[(j + q * 65536) % n];
    float4 s = a;
#pragma unroll
    for (int q = 0; q < NL; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    reinterpret_cast<float4 *>(out)[i] = s;
}

int main() {
    const int NB = 8, n4 = 1 << 20;  // 8 buffers of 16 MB (float4 count 1 M)
    float *buf[NB];
    for (int i = 0; i < NB; ++i) { hipMalloc(&buf[i], (size_t)n4 * 16); hipMemset(buf[i], 0, (size_t)n4 * 16); }
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int CH = 600;
    auto run = [&](const char *name, int grid, auto launch) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < CH; ++i) launch(buf[i % NB], buf[(i + 1) % NB], grid);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        float best = 1e9f, ms;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-34s grid %4d: %6.2f us/kernel  (%s)\n", name, grid, best * 1000.f / CH, hipGetErrorString(hipGetLastError()));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    };
    for (int grid : {4, 256, 512}) {
        run("empty", grid, [&](float *, float *, int gr) { k_empty<<<gr, 256, 0, s>>>(); });
        run("ld-scalar x1", grid, [&](float *a, float *b, int gr) { k_sc<1, 0><<<gr, 256, 0, s>>>(a, b, n4); });
        run("ld-scalar x16", grid, [&](float *a, float *b, int gr) { k_sc<16, 0><<<gr, 256, 0, s>>>(a, b, n4); });
        run("ld-struct x16", grid, [&](float *a, float *b, int gr) { P p{}; p.in = a; p.out = b; p.n = n4; k_st<16><<<gr, 256, 0, s>>>(p); });
        run("ld2 (dependent) x16", grid, [&](float *a, float *b, int gr) { k_l2<16><<<gr, 256, 0, s>>>(a, b, n4); });
        run("ld-scalar x16 + 36 mfma", grid, [&](float *a, float *b, int gr) { k_sc<16, 36><<<gr, 256, 0, s>>>(a, b, n4); });
        run("ld-scalar x16 + 72 mfma", grid, [&](float *a, float *b, int gr) { k_sc<16, 72><<<gr, 256, 0, s>>>(a, b, n4); });
        run("ld-scalar x16 + 144 mfma", grid, [&](float *a, float *b, int gr) { k_sc<16, 144><<<gr, 256, 0, s>>>(a, b, n4); });
    }
    return 0;
}
