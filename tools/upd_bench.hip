// Dev harness: the whole SAC1 update of the direct-operand path inside a hipGraph (config-2 shape), timed, plus — in a
// -DDDRL_STAMPS build — the per-workgroup anatomy of every launch of the LAST update (cycle stamps of thread 0 and the
// 100 MHz real-time counter for the launch's dispatch ramp / tail).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form
//       -Wno-unused-function [-DDDRL_STAMPS] tools/upd_bench.hip -o tools/upd_bench.bin
#include "../distributed-drl_amd/csrc/sac1.hip"
#include "../distributed-drl_amd/csrc/common.hip"
#include "../distributed-drl_amd/csrc/replay.hip"
#include <algorithm>

int main(int argc, char **argv) {
    const int per_graph = argc > 1 ? atoi(argv[1]) : 50;
    ddrl_sac1_config_t c{};
    c.obs_dim = 8; c.act_dim = 2; c.hidden1 = 400; c.hidden2 = 300; c.batch = 256; c.variant = DDRL_SAC1;
    c.alpha = 0.1; c.gamma = 0.997; c.lr = 5e-5; c.polyak = 0.995; c.beta1 = 0.9; c.beta2 = 0.999; c.adam_eps = 1e-8; c.act_scale = 1.0;
    ddrl_sac1_t *h = nullptr;
    if (ddrl_sac1_create(&h, 0, &c) != DDRL_OK) { printf("create failed: %s\n", ddrl_last_error()); return 1; }
    int64_t n_pi, n_q;
    ddrl_sac1_param_counts(&c, &n_pi, &n_q);
    const size_t n = (size_t)(n_pi + 2 * n_q);
    std::vector<float> w(n);
    srand(3);
    for (auto &v : w) v = ((float)(rand() & 0xffff) / 65536.f - 0.5f) * 0.1f;
    float *wd; hipMalloc(&wd, n * 4); hipMemcpy(wd, w.data(), n * 4, hipMemcpyHostToDevice);
    ddrl_sac1_set_weights(h, wd, nullptr);
    float *in[8];
    ddrl_sac1_input_buffers(h, 0, in);
    const int cnt[8] = {256 * 8, 256 * 8, 256 * 2, 256, 256, 256 * 2, 256 * 2, 256 * 2};
    for (int i = 0; i < 8; ++i) {
        std::vector<float> v(cnt[i]);
        for (auto &x : v) x = (float)(rand() & 0xffff) / 32768.f - 1.f;
        if (i == 4) for (auto &x : v) x = x > 0.9f ? 1.f : 0.f;
        hipMemcpy(in[i], v.data(), cnt[i] * 4, hipMemcpyHostToDevice);
    }
    hipStream_t s; hipStreamCreate(&s);
    auto one = [&]() { return ddrl_sac1_step(h, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], nullptr, nullptr, nullptr, nullptr, s); };
    for (int i = 0; i < 4; ++i) if (one() != DDRL_OK) { printf("step failed: %s\n", ddrl_last_error()); return 1; }
    hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < per_graph; ++i) one();
    ddrl_sac1_internal_opt_sync(h, s);
    hipStreamEndCapture(s, &g);
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms, best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, s);
        for (int k = 0; k < 4; ++k) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    printf("graph of %d updates: %.2f us/update (%s)\n", per_graph, best * 1000.f / (4 * per_graph), hipGetErrorString(hipGetLastError()));
#ifdef DDRL_STAMPS
    unsigned long long *st;
    const size_t NS = (size_t)5 * 1024 * 16;
    hipMalloc(&st, NS * 8); hipMemset(st, 0, NS * 8);
    g_st_host = st;   // the stamp pointer travels in the kernel arguments: capture the same graph again, now stamped
    hipGraph_t g2; hipGraphExec_t ge2;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < per_graph; ++i) one();
    ddrl_sac1_internal_opt_sync(h, s);
    hipStreamEndCapture(s, &g2);
    if (hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
    hipGraphLaunch(ge2, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int k = 0; k < 4; ++k) hipGraphLaunch(ge2, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("stamped graph: %.2f us/update\n", ms * 1000.f / (4 * per_graph));
    std::vector<unsigned long long> hs(NS);
    auto anatomy = [&](const char *title) {
    printf("---- %s\n", title);
    hipMemcpy(hs.data(), st, NS * 8, hipMemcpyDeviceToHost);
    const char *kn[5] = {"k_dfwd<0>", "k_dfwd<1>", "k_dg bq", "k_dg mid", "k_dg pi"};
    const char *pn[6] = {"", "loads-issued", "prologue", "k-loop", "combine-bar", "epilogue"};
    unsigned long long prev_end = 0;
    for (int k = 0; k < 5; ++k) {
        double ph[6] = {0}, tot = 0; int nwg = 0;
        unsigned long long rmin = ~0ull, rmax = 0, rstart_max = 0, lmax = 0;
        for (int b = 0; b < 1024; ++b) {
            const unsigned long long *p = &hs[((size_t)k * 1024 + b) * 16];
            if (!p[15]) continue;
            ++nwg;
            for (int i = 1; i < 6; ++i) if (p[i] && p[i - 1]) ph[i] += (double)(p[i] - p[i - 1]);
            tot += (double)(p[5] - p[0]);
            lmax = std::max(lmax, p[5] - p[0]);
            rmin = std::min(rmin, p[14]); rmax = std::max(rmax, p[15]); rstart_max = std::max(rstart_max, p[14]);
        }
        if (!nwg) continue;
        printf("%-10s %4d wgs | span %.2f us (first start -> last end), last start +%.2f us, gap from previous launch's end %.2f us | per-wg cycles:", kn[k], nwg,
               (rmax - rmin) / 100.0, (rstart_max - rmin) / 100.0, prev_end ? ((double)rmin - (double)prev_end) / 100.0 : 0.0);
        for (int i = 1; i < 6; ++i) printf(" %s=%.0f", pn[i], ph[i] / nwg);
        printf(" | mean total %.0f max %llu\n", tot / nwg, lmax);
        prev_end = rmax;
        // per job of a k_dg launch: when its workgroups start / end relative to the launch's first start (100 MHz real-time counter)
        const DGJobs *J = k == 2 ? &h->dg_bq[0] : (k == 3 ? &h->dg_mid : (k == 4 ? &h->dg_pi : nullptr));
        if (J) {
            const int nwgs = J->total_tiles;
            for (int ji = 0; ji < J->njobs; ++ji) {
                double ssum = 0, esum = 0, emax = 0, csum = 0, cmax = 0, psum[6] = {0}; int n = 0;
                for (int b = 0; b < nwgs; ++b) {
                    const int q = nwgs >> 3, r = nwgs & 7, x = b & 7;
                    const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
                    if (t < J->job[ji].tile_start || t >= J->job[ji].tile_start + J->job[ji].ntiles) continue;
                    const unsigned long long *p = &hs[((size_t)k * 1024 + b) * 16];
                    if (!p[15]) continue;
                    ++n; ssum += (p[14] - rmin) / 100.0; const double e = (p[15] - rmin) / 100.0; esum += e; emax = std::max(emax, e);
                    const double c = (double)(p[5] - p[0]); csum += c; cmax = std::max(cmax, c);
                    for (int i = 1; i < 6; ++i) if (p[i] && p[i - 1]) psum[i] += (double)(p[i] - p[i - 1]);
                }
                if (n) printf("      job %d type %d M %4d N %4d K %4d: %3d wgs  start +%.2f us  end mean +%.2f max +%.2f us | cycles mean %.0f max %.0f | loads %.0f prol %.0f kloop %.0f bar %.0f epi %.0f\n", ji, J->job[ji].type,
                              J->job[ji].M, J->job[ji].N, J->job[ji].K, n, ssum / n, esum / n, emax, csum / n, cmax, psum[1] / n, psum[2] / n, psum[3] / n, psum[4] / n, psum[5] / n);
            }
        }
    }
    };
    anatomy("inside the update sequence (graph)");
    // the same launches repeated back to back, WITHOUT the optimizer step in the epilogues (ddrl_sac1_stage_time): inputs stay in the L2s
    hipMemset(st, 0, NS * 8);
    for (int stg : {2, 5, 7, 8, 9}) { float ms2; ddrl_sac1_stage_time(h, stg, 40, &ms2, s); }
    hipStreamSynchronize(s);
    anatomy("each launch repeated back to back, no optimizer step");
#endif
    return 0;
}
