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
};
int ddrl_actor_internal_forward(ddrl_actor_t *h, long long n, void *stream, int versioned);
ddrl_actor_rollout_view ddrl_actor_internal_view(ddrl_actor_t *h);
