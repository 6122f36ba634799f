// Dev harness: the rollout's policy forward (k_actor_fwd) alone, timed; with -DDDRL_STAMPS the per-wave anatomy.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form
//       -Wno-unused-function [-DDDRL_STAMPS] tools/actor_bench.hip -o tools/actor_bench.bin
#include "../distributed-drl_amd/csrc/sac1.hip"
#include "../distributed-drl_amd/csrc/common.hip"
#include "../distributed-drl_amd/csrc/replay.hip"
#include <algorithm>
int main(int argc, char **argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 4096;
    ddrl_sac1_config_t c{};
    c.obs_dim = 8; c.act_dim = 2; c.hidden1 = 400; c.hidden2 = 300; c.batch = 256; c.variant = DDRL_SAC1;
    c.alpha = 0.1; c.gamma = 0.997; c.lr = 5e-5; c.polyak = 0.995; c.beta1 = 0.9; c.beta2 = 0.999; c.adam_eps = 1e-8; c.act_scale = 1.0;
    ddrl_actor_t *h = nullptr;
    if (ddrl_actor_create(&h, 0, &c, n) != DDRL_OK) { printf("create failed: %s\n", ddrl_last_error()); return 1; }
    int64_t n_pi, n_q;
    ddrl_sac1_param_counts(&c, &n_pi, &n_q);
    std::vector<float> wv(n_pi);
    srand(3);
    for (auto &v : wv) v = ((float)(rand() & 0xffff) / 65536.f - 0.5f) * 0.1f;
    float *wd; hipMalloc(&wd, n_pi * 4); hipMemcpy(wd, wv.data(), n_pi * 4, hipMemcpyHostToDevice);
    ddrl_actor_set_weights(h, wd, nullptr);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) ddrl_actor_internal_forward(h, n, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < 200; ++i) ddrl_actor_internal_forward(h, n, s);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k_actor_fwd, %lld rows: %.2f us per launch (back to back, %s)\n", n, ms * 1e3 / 200, hipGetErrorString(hipGetLastError()));
#ifdef DDRL_STAMPS
    const size_t NS = (size_t)1024 * 4 * 32;
    unsigned long long *st; hipMalloc(&st, NS * 8); hipMemset(st, 0, NS * 8);
    g_actor_st = st;
    ddrl_actor_internal_forward(h, n, s); hipStreamSynchronize(s);
    std::vector<unsigned long long> hs(NS);
    hipMemcpy(hs.data(), st, NS * 8, hipMemcpyDeviceToHost);
    const char *nm[8] = {"", "loads", "L1", "", "mfma", "redwr", "bar1", "comb"};
    for (int w = 0; w < 4; ++w) {
        double ph[32] = {0}; int cnt = 0; double tot = 0;
        for (int b = 0; b < 1024; ++b) {
            const unsigned long long *p = &hs[((size_t)b * 4 + w) * 32];
            if (!p[0] || !p[27]) continue;
            ++cnt; tot += (double)(p[27] - p[0]);
            for (int i = 1; i < 28; ++i) ph[i] += (double)(p[i] - p[i - 1]);
        }
        if (!cnt) continue;
        printf("wave %d (%d wgs) total %.0f | loads-issued %.0f  L1 %.0f |", w, cnt, tot / cnt, ph[1] / cnt, ph[2] / cnt);
        for (int t = 0; t < 5; ++t)
            printf(" t%d: mfma %.0f redwr %.0f bar1 %.0f comb %.0f bar2+dot %.0f |", t, ph[3 + 5 * t] / cnt, ph[4 + 5 * t] / cnt, ph[5 + 5 * t] / cnt, ph[6 + 5 * t] / cnt, ph[7 + 5 * t] / cnt);
        printf("\n");
    }
#endif
    return 0;
}
