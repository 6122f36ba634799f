// Dev micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate and shader clock as a function of waves per SIMD and
// accumulator chains per wave (every CU busy).  hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *st, int nm) {
    floatx16 acc[NACC];
    const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = a + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int m = 0; m < nm; ++m) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][9];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { st[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; st[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}
int main() {
    float *out; unsigned long long *st;
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&st, 2048 * 4 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, int grid, int nacc, int nm) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (nacc == 1) k<1><<<grid, 256>>>(out, st, nm); else if (nacc == 2) k<2><<<grid, 256>>>(out, st, nm); else k<4><<<grid, 256>>>(out, st, nm);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<unsigned long long> h((size_t)grid * 8);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        double tc = 0, tr = 0; for (int i = 0; i < grid * 4; ++i) { tc += h[2 * i]; tr += h[2 * i + 1]; }
        tc /= grid * 4; tr /= grid * 4;
        const double nmf = (double)nm * nacc;
        printf("%-28s grid %4d: kernel %.1f us | per wave: %.0f cycles (%.1f / mfma), %.2f us -> clock %.0f MHz | chip %.1f TFLOP/s\n", name, grid, ms * 1e3, tc, tc / nmf,
               tr / 100.0, tc / tr * 100.0, nmf * 4096.0 * grid * 4 / (tr / 100.0 * 1e-6) / 1e12);
    };
    for (int nm : {144, 2000}) {
        run("1 wave/SIMD, 1 chain", 256, 1, nm);
        run("1 wave/SIMD, 2 chains", 256, 2, nm / 2);
        run("2 waves/SIMD, 1 chain", 512, 1, nm);
        run("2 waves/SIMD, 2 chains", 512, 2, nm / 2);
        run("4 waves/SIMD, 1 chain", 1024, 1, nm);
        run("1/2 chip, 1 wave/SIMD", 128, 1, nm);
    }
    return 0;
}
