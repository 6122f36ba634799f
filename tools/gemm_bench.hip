// Dev harness: drives k_gemm of sac1.hip directly (fwd "A" stage shape: 5 nets x [256x400]x[400x300])
// with per-phase cycle stamps of wave 0 of every workgroup.  hipcc -DDDRL_STAMPS.
#include "../distributed-drl_amd/csrc/sac1.hip"
#include "../distributed-drl_amd/csrc/common.hip"
#include "../distributed-drl_amd/csrc/replay.hip"
#include <algorithm>
__global__ void k_touch(float *p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += 1e-9f; }
int main() {
    const int B = 256, h1 = 400, h2 = 300, ldh1 = 404, ldh2 = 304, NE = 5;
    float *H1, *H2, *W2, *b2; long long *stamps;
    hipMalloc(&H1, (size_t)NE * B * ldh1 * 4); hipMalloc(&H2, (size_t)NE * B * ldh2 * 4);
    hipMalloc(&W2, (size_t)NE * h1 * h2 * 4 + 4096); hipMalloc(&b2, NE * h2 * 4 + 64);
    hipMemset(H1, 0, (size_t)NE * B * ldh1 * 4); hipMemset(W2, 0, (size_t)NE * h1 * h2 * 4);
    GemmJobs js{};
    for (int e = 0; e < NE; ++e) gemm_add(js, gemm_fwd(H1 + (size_t)e * B * ldh1, ldh1, W2 + (size_t)e * h1 * h2, b2 + e * h2, H2 + (size_t)e * B * ldh2, ldh2, B, h1, h2));
    printf("tiles %d fast %d\n", js.total_tiles, js.job[0].fast);
    hipMalloc(&stamps, (size_t)js.total_tiles * 32 * 8); hipMemset(stamps, 0, (size_t)js.total_tiles * 32 * 8);
    js.stamps = nullptr;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) { k_touch<<<2048, 256>>>(H1, NE * B * ldh1); launch_gemm(js, nullptr); }
    hipDeviceSynchronize();
    float ms;
    hipEventRecord(e0); for (int i = 0; i < 100; ++i) launch_gemm(js, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("k_gemm back-to-back: %.2f us/launch\n", ms * 10.f);
    js.stamps = stamps;
    k_touch<<<2048, 256>>>(H1, NE * B * ldh1);
    launch_gemm(js, nullptr); hipDeviceSynchronize();
    std::vector<long long> hs((size_t)js.total_tiles * 32);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    long long tmin = hs[0];
    for (int b = 0; b < js.total_tiles; ++b) tmin = std::min(tmin, hs[(size_t)b * 32]);
    const char *names[30] = {"start", "jobloaded", "prologue-issued", "s0:bar1", "s0:st", "s0:bar2", "s0:rd+pf", "s0:mfma", "s1:bar1", "s1:st", "s1:bar2", "s1:rd+pf", "s1:mfma",
                             "s2:bar1", "s2:st", "s2:bar2", "s2:rd+pf", "s2:mfma", "s3:bar1", "s3:st", "s3:bar2", "s3:rd+pf", "s3:mfma", "", "", "", "", "", "loop-end", "end"};
    for (int b : {0, 1, 8, 100, 255, 256, 399}) {
        printf("block %3d: start +%6lld |", b, hs[(size_t)b * 32] - tmin);
        long long prev = hs[(size_t)b * 32];
        for (int i = 1; i < 30; ++i) { long long v = hs[(size_t)b * 32 + i]; if (!v) continue; printf(" %s %lld", names[i], v - prev); prev = v; }
        printf(" | total %lld\n", prev - hs[(size_t)b * 32]);
    }
    // average per phase over all blocks
    double avg[30] = {0}; int cnt = 0; double tot = 0, last_end = 0;
    for (int b = 0; b < js.total_tiles; ++b) { long long prev = hs[(size_t)b * 32]; for (int i = 1; i < 30; ++i) { long long v = hs[(size_t)b * 32 + i]; if (!v) continue; avg[i] += v - prev; prev = v; } tot += prev - hs[(size_t)b * 32]; last_end = std::max(last_end, (double)(prev - tmin)); ++cnt; }
    printf("mean over %d blocks (cycles of the 100MHz*? counter):", cnt);
    for (int i = 1; i < 30; ++i) if (avg[i] > 0) printf(" %s=%.0f", names[i], avg[i] / cnt);
    printf(" | mean total %.0f, last block end at +%.0f\n", tot / cnt, last_end);
    return 0;
}
