// Dev harness: drives k_fwd<PHASE> of sac1_fused.h directly (stage A shape: 5 nets, 256 x (8|10 -> 400) x 300)
// with per-phase cycle stamps of wave 0 of every workgroup.  hipcc -DDDRL_STAMPS.
#include "../distributed-drl_amd/csrc/sac1.hip"
#include "../distributed-drl_amd/csrc/common.hip"
#include "../distributed-drl_amd/csrc/replay.hip"
#include <algorithm>
#ifndef PHASE
#define PHASE 0
#endif
int main() {
    const int B = 256, h1 = 400, h2 = 300, ldh1 = 404, ldh2 = 304, NE = PHASE ? 3 : 5, nt2 = 10, o = 8, ac = 2;
    float *H1, *H2, *W, *in, *hp; long long *stamps; OptState *opt;
    hipMalloc(&H1, (size_t)NE * B * ldh1 * 4); hipMalloc(&H2, (size_t)NE * B * ldh2 * 4);
    const size_t per = (size_t)12 * h1 + h1 + (size_t)h1 * h2 + h2 + 4 * h2 + 64;
    hipMalloc(&W, NE * per * 4 + 65536); hipMemset(W, 0, NE * per * 4 + 65536);
    hipMalloc(&in, B * 16 * 4 * 4); hipMemset(in, 0, B * 16 * 4 * 4);
    hipMalloc(&hp, (size_t)8 * FH * nt2 * B * 4); hipMemset(hp, 0, (size_t)8 * FH * nt2 * B * 4); hipMalloc(&opt, sizeof(OptState)); hipMemset(opt, 0, sizeof(OptState));
    FwdArgs F{};
    F.njobs = NE; F.hd.tiles_m = B / 32; F.tiles_n = nt2; F.hd.tpj = (B / 32) * nt2; F.hd.h1 = h1; F.hd.h2 = h2;
    { const int chunk = ((h1 + 15) >> 4) << 2; F.ks_max = chunk < 64 ? (chunk < KS ? KS : chunk) : 64;
      const int ta = 32 * (F.ks_max + 2), tb = F.ks_max * 36; F.op_lds = ta > tb ? ta : tb; }
    F.B = B; F.ldh1 = ldh1; F.ldh2 = ldh2; F.act = ac; F.nt2 = nt2; F.scale = 1.f; F.opt = opt;
    for (int e = 0; e < NE; ++e) {
        float *P = W + e * per;
        FwdJob j{};
        j.in = FIn{in, e >= 3 ? in + B * o : nullptr, P, P + 12 * h1, o, e >= 3 ? ac : 0};
        j.W2 = P + 13 * h1; j.b2 = j.W2 + (size_t)h1 * h2; j.H2 = H2 + (size_t)e * B * ldh2; j.H1 = (e == 0 || e >= 3) ? H1 + (size_t)e * B * ldh1 : nullptr;
        j.wh0 = j.b2 + h2; j.wh1 = j.wh0 + 2 * h2; j.nh = e < 3 ? 4 : 1; j.hsplit = e < 3 ? 2 : 1; j.hstride = e < 3 ? 2 : 1;
        j.hp = hp + (size_t)e * FH * nt2 * B;
        if (PHASE) { j.in = FIn{in, nullptr, P, P + 12 * h1, o, 0}; j.nh = 1; j.hsplit = 1; j.hstride = 1; j.php = hp; j.pbmu = P; j.pbls = P; j.peps = in; j.side = e == 0 ? 1 : (e == 1 ? 2 : 0); }
        F.job[e] = j;
    }
    F.act0 = in + 4096; F.act2 = in + 5120; F.logp0 = in + 6144; F.logp1 = in + 6400; F.save0 = in + 8192; F.php1 = hp; F.pbmu1 = W; F.pbls1 = W; F.peps1 = in;
    F.hd.pbase = W; for (int e = 0; e < NE; ++e) { F.hd.w2_off[e] = (int)(F.job[e].W2 - W); F.hd.w1_off[e] = (int)(F.job[e].in.W1 - W); }
    const int grid = NE * F.hd.tpj;
    hipMalloc(&stamps, (size_t)grid * 32 * 8); hipMemset(stamps, 0, (size_t)grid * 32 * 8);
    F.stamps = nullptr;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch_fwd<PHASE>(F, nullptr);
    hipDeviceSynchronize();
    float ms;
    hipEventRecord(e0); for (int i = 0; i < 100; ++i) launch_fwd<PHASE>(F, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("k_fwd<PHASE> back-to-back: %.2f us/launch (grid %d, smem %zu)\n", ms * 10.f, grid, fwd_smem(F));
    F.stamps = stamps;
    launch_fwd<PHASE>(F, nullptr); hipDeviceSynchronize();
    std::vector<long long> hs((size_t)grid * 32);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    long long tmin = hs[0];
    for (int b = 0; b < grid; ++b) tmin = std::min(tmin, hs[(size_t)b * 32]);
    const char *names[12] = {"start", "loads-issued", "staged+bar", "gen0", "stB0", "mfma0", "gen1", "stB1", "mfma1", "combine+H2", "heads"};
    for (int b : {0, 1, 8, 100, 255, 256, 399}) {
        printf("block %3d: start +%6lld |", b, hs[(size_t)b * 32] - tmin);
        long long prev = hs[(size_t)b * 32];
        for (int i = 1; i < 11; ++i) { long long v = hs[(size_t)b * 32 + i]; if (!v) continue; printf(" %s %lld", names[i], v - prev); prev = v; }
        printf(" | total %lld\n", prev - hs[(size_t)b * 32]);
    }
    double avg[12] = {0}; int cnt = 0; double tot = 0, last_end = 0;
    for (int b = 0; b < grid; ++b) { long long prev = hs[(size_t)b * 32]; for (int i = 1; i < 11; ++i) { long long v = hs[(size_t)b * 32 + i]; if (!v) continue; avg[i] += v - prev; prev = v; } tot += prev - hs[(size_t)b * 32]; last_end = std::max(last_end, (double)(prev - tmin)); ++cnt; }
    { double e[6] = {0}; for (int b = 0; b < grid; ++b) for (int i = 0; i < 6; ++i) e[i] += hs[(size_t)b * 32 + 11 + i] - hs[(size_t)b * 32];
      printf("since start: indices %.0f  op-init %.0f  pb-issued %.0f  w1-issued %.0f  jb+b1-issued %.0f  in-issued %.0f\n", e[4] / grid, e[5] / grid, e[0] / grid, e[1] / grid, e[2] / grid, e[3] / grid); }
    printf("mean over %d blocks:", cnt);
    for (int i = 1; i < 11; ++i) if (avg[i] > 0) printf(" %s=%.0f", names[i], avg[i] / cnt);
    printf(" | mean total %.0f, last block end at +%.0f\n", tot / cnt, last_end);
    return 0;
}
