// Row-local policy math shared by the learner's forward stages (sac1_direct.h), the generic row kernels and the fused
// rollout step (env.hip): one row of one policy evaluation (algos/sac1/core.py:49-87, 104-106) and the counter-hash normal
// that stands in for tf.random_normal (identical integer arithmetic in oracle/noise_oracle.py).
#pragma once
#include "ddrl_common.h"

namespace ddrl_pol {

constexpr float LOG2PI = 1.8378770664093453f;  // float32(np.log(2*np.pi))
constexpr float STD_EPS = 1e-8f;               // core.py:5 EPS

__device__ __forceinline__ float normal_at(uint32_t seed, unsigned long long c) {  // == ddrl_normal_fill element c
    const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
    const uint32_t h1 = ddrl::hash3(seed, lo, 2u * hi), h2 = ddrl::hash3(seed, lo, 2u * hi + 1u);
    const float u1 = (float)((h1 >> 8) + 1u) * (1.0f / 16777216.0f);
    const float u2 = ddrl::u01(h2);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// One row of one policy evaluation, all action dims in one lane.  Same formulae and operation order as policy_head()
// of sac1.hip (core.py:49-87, 104-106); mu / log_std pre-activations arrive as sums of per-column-tile partials.
struct PolRow {
    float act[4], a[4], std[4], t[4];
    float logp;
};
__device__ __forceinline__ PolRow policy_row(const float (&mu)[4], const float (&lsr)[4], const float (&eps)[4], int act, float scale) {
    PolRow o;
    float sp = 0.f, sc = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        o.act[c] = 0.f; o.a[c] = 0.f; o.std[c] = 0.f; o.t[c] = 0.f;
        if (c < act) {
            const float t = tanhf(lsr[c]);
            const float log_std = -20.0f + 11.0f * (t + 1.0f);
            const float std = expf(log_std);
            const float e = eps[c];
            const float u = mu[c] + e * std;
            const float z = (e * std) / (std + STD_EPS);
            const float pre = -0.5f * ((z * z + 2.0f * log_std) + LOG2PI);
            const float a = tanhf(u);
            const float om = 1.0f - a * a;
            const float cl = fminf(fmaxf(om, 0.f), 1.f);
            const float corr = logf(cl + 1e-6f);
            sp += pre; sc += corr;
            o.act[c] = a * scale; o.a[c] = a; o.std[c] = std; o.t[c] = t;
        }
    }
    o.logp = sp - sc;
    return o;
}

}  // namespace ddrl_pol

// ---- the versioned policy forward's tables (written by k_version_plan in sac1.hip and by the env-step launch's tail in env.hip)
// One WORKGROUP of the versioned forward: `count` (<= 32) envs perm[base .. base + count) that all act on policy version `slot`, and the
// column tiles [cols & 255, + cols >> 8) of their row tile.
struct VerTile { int slot, base, count, cols; };
constexpr int VER_MAX_SLOTS = 2048;
// Device-side state of an actor's version store (exact per-env weight adoption, example/dsac.py:127-130).
struct VerState { int newest, target, n_tiles, err, live, n_wgs, pad[2]; };
// How the planning launch deals the row tiles' column tiles to workgroups, from the ACTUAL tile count.  With envs grouped by version
// the count sits a little over a whole number of rounds (8192 envs / 16 versions: 264 row tiles x 2 workgroups = 528 for 512 resident
// slots: sixteen workgroups ran a second round as long as the first).  A workgroup of `per` column tiles takes about 6.2 + 3.1 per
// microseconds with the chip full (tools/version_step_probe.py), so:
//   * everything fits one round: as many workgroups per row tile as still fit it (two, or three when 3 T <= slots);
//   * otherwise the first floor(g T / slots) FULL rounds' worth of row tiles take g = ceil(col_tiles / ANT) workgroups each (the
//     coarsest split, the least redundant layer-1 work), and the SURPLUS row tiles behind them are cut fine enough that their workgroups
//     are short and fit one more round of their own: 16 surplus tiles x 10 single-tile workgroups add 9 us behind the long round
//     instead of 17-22.  (A finer split of EVERY tile lost in every case tried: 1320 and 1410 workgroups, dispatch.)
// The long workgroups come first in launch order, the short ones fill the slots the long ones leave.
struct VerSplit { int g_long, n_long, g_short, n_wgs; };   // n_long row tiles x g_long workgroups, the rest x g_short
__host__ __device__ inline int ver_wg_dur(int col_tiles, int g) { return 62 + 31 * ((col_tiles + g - 1) / g); }
__host__ __device__ inline VerSplit ver_split(int n_tiles, int col_tiles, int slots, int cap) {
    const int ant = 5;   // = ANT (sac1_direct.h: column tiles one workgroup of the forward can walk)
    int g2 = (col_tiles + ant - 1) / ant;
    if (g2 < 2 && col_tiles >= 2) g2 = 2;
    if (g2 < 1) g2 = 1;
    VerSplit v{g2, n_tiles, g2, n_tiles * g2};
    if (n_tiles <= 0 || slots <= 0) return v;
    const int rounds = (n_tiles * g2) / slots;
    if (rounds == 0) {       // one round: a finer uniform split while it still fits
        if (g2 + 1 <= col_tiles && n_tiles * (g2 + 1) <= slots && n_tiles * (g2 + 1) <= cap &&
            (col_tiles + g2) / (g2 + 1) < (col_tiles + g2 - 1) / g2) { v.g_long = v.g_short = g2 + 1; v.n_wgs = n_tiles * (g2 + 1); }
        return v;
    }
    const int n_long = (rounds * slots) / g2 < n_tiles ? (rounds * slots) / g2 : n_tiles;
    const int u = n_tiles - n_long;
    if (u == 0) return v;
    int best = g2, best_cost = ((u * g2 + slots - 1) / slots) * ver_wg_dur(col_tiles, g2);
    const int cand[5] = {g2 + 1, 4, 5, (col_tiles + 1) / 2, col_tiles};
    for (int i = 0; i < 5; ++i) {
        const int g = cand[i];
        if (g <= g2 || g > col_tiles || u * g > slots || n_long * g2 + u * g > cap) continue;   // short workgroups: one round of their own
        const int cost = ver_wg_dur(col_tiles, g);
        if (cost < best_cost) { best_cost = cost; best = g; }
    }
    v.n_long = n_long; v.g_short = best; v.n_wgs = n_long * g2 + u * best;
    return v;
}

// What the fused rollout step (env.hip) needs from an actor handle (sac1.hip): internal to libddrl_hip.so.
struct ddrl_actor_rollout_view {
    int ok, device;
    float *obs;         // [max_rows][obs_dim]: the observations the next forward launch acts on
    const float *hp;    // head partials of the last forward launch [DFH = 8][n][DNT = 16]
    const float *bmu, *bls;
    int obs_dim, act, nt2;
    long long max_rows;
    float scale;
    // version store (n_slots > 0): the slot every env acts on, the head biases of slot 0 (slot s: + s * vstride), device state
    int n_slots;
    int *slot;
    const void *vs;     // VerState (sac1_direct.h): .newest is what an env adopts at its episode end
    long long vstride;
    const float *vbmu, *vbls;
    long long *steps_since_install;
    bool *plan_fresh;   // host flag of the actor: the forward's tile table matches the slots (cleared behind every launch that moves an env)
    // the forward's plan as the env-step launch writes it (fused: no planning launch between a vector step and the next forward)
    int *vcnt;          // [VER_MAX_SLOTS] envs per slot, all zero between launches
    int *perm;          // the forward's row lists; the env-step launch's own region starts at perm2d_off: [n_slots][max_rows]
    long long perm2d_off;
    VerTile *vtiles;
    int vt_cap, wg_slots;
};
int ddrl_actor_internal_forward(ddrl_actor_t *h, long long n, void *stream, int versioned);
ddrl_actor_rollout_view ddrl_actor_internal_view(ddrl_actor_t *h);
