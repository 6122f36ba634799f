// Batched lander environment: one thread per environment, state as struct-of-arrays
// [DDRL_ENV_STATE_FIELDS][n] in HBM (coalesced per-field loads/stores).  Stands where the
// reference calls gym's LunarLanderContinuous-v2 env.step / env.reset
// (example/dsac.py:78-79,102,127) and fuses the worker's per-step episode bookkeeping
// (example/dsac.py:102-127: ep_len/ep_ret, "time limit is not a terminal", reset at episode end).
//
// The dynamics are this build's own Box2D-style rigid-body model — see oracle/env_oracle.py for
// the specification; this kernel reproduces that restatement BIT-EXACTLY: float32, no FMA
// contraction (-ffp-contract=off), correctly rounded sqrt/divide, polynomial sin/cos shared with
// the oracle, counter-hash RNG.  Every arithmetic statement below mirrors one line of the oracle.
#include "ddrl_common.h"
#include "policy_row.h"
#include "replay_device.h"

namespace {

enum { X = 0, Y, VX, VY, ANG, OM, C1, C2, PREV, HASP, EPLEN, EPRET, SLEEP, EPI, T0, PSTEP = 25 };
constexpr int NF = DDRL_ENV_STATE_FIELDS;

constexpr float FPS = 50.0f, DT = 0.02f, SCALE = 30.0f, H_ = 13.333333f, HELIPAD_Y = 3.3333333f, LEG_DOWN = 0.6f;
constexpr float GRAV = -10.0f, INV_M = 0.20746888f, INV_I = 1.1904762f, MAIN_POWER = 13.0f, SIDE_POWER = 0.6f;
constexpr float SIDE_AWAY = 0.4f, SIDE_H = 0.46666667f, LEGX = 0.6666667f, LEGY = -0.8666667f;
constexpr int NIT = 4;
constexpr float SLOP = 0.005f, BAUM = 0.2f, MU = 0.5f, SLEEP_V2 = 0.0025f, SLEEP_W = 0.05f, SLEEP_STEPS = 25.0f;
constexpr uint32_t RESET_STREAM = 0xFFFFFFF0u;
__constant__ float HULLX[6] = {-0.46666667f, -0.56666666f, -0.56666666f, 0.56666666f, 0.56666666f, 0.46666667f};
__constant__ float HULLY[6] = {0.56666666f, 0.0f, -0.33333334f, -0.33333334f, 0.0f, 0.56666666f};

struct EnvStats {
    long long episodes, len_sum;
    double ret_sum;
};

__device__ __forceinline__ void sincos32(float x, float &sn, float &cs) {
    const float k = rintf(x * 0.63661975f);
    const float r = ((x - k * 1.5703125f) - k * 4.837513e-4f) - k * 7.54979e-8f;
    const float r2 = r * r;
    const float s = r + (r * r2) * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * -1.9515295891e-4f));
    const float c = (1.0f - 0.5f * r2) + (r2 * r2) * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f));
    const int q = ((int)k) & 3;
    sn = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
    cs = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

struct Env {
    float x, y, vx, vy, ang, om, c1, c2, prev, hasp, eplen, epret, sleep, epi, pstep;
    float terr[11];
    uint32_t seed, id;

    __device__ float rng(uint32_t step, uint32_t j) const {
        const uint32_t seed_e = seed ^ ddrl::mix32((uint32_t)epi);
        uint32_t h = ddrl::mix32(seed_e ^ 0x9E3779B9u);
        h = ddrl::mix32(h + id * 0x85EBCA6Bu + 0x27D4EB2Fu);
        h = ddrl::mix32(h ^ ((step * 16u + j) * 0xC2B2AE35u + 0x165667B1u));
        return ddrl::u01(h);
    }
    __device__ float ground(float px) const {
        const float fi = clampf(floorf(px * 0.5f), 0.0f, 9.0f);
        const int idx = (int)fi;
        // terrain lives in registers: select without dynamic indexing
        float h0 = terr[0], h1 = terr[1];
#pragma unroll
        for (int i = 1; i < 10; ++i)
            if (idx == i) { h0 = terr[i]; h1 = terr[i + 1]; }
        const float t = (px - 2.0f * fi) * 0.5f;
        return h0 + (h1 - h0) * t;
    }
    __device__ void obs(float *o) const {
        o[0] = (x - 10.0f) / 10.0f;
        o[1] = (y - (HELIPAD_Y + LEG_DOWN)) / 6.6666665f;
        o[2] = vx * 10.0f / FPS;
        o[3] = vy * 6.6666665f / FPS;
        o[4] = ang;
        o[5] = 20.0f * om / FPS;
        o[6] = c1;
        o[7] = c2;
    }
    // one physics step (oracle: LanderOracle._physics); returns reward, sets done_env and o[8]
    __device__ float physics(float act0, float act1, bool &done_env, float *o) {
        const float a0 = clampf(act0, -1.0f, 1.0f), a1 = clampf(act1, -1.0f, 1.0f);
        const uint32_t step = (uint32_t)pstep;
        const float d0 = (rng(step, 0) * 2.0f - 1.0f) / SCALE;
        const float d1 = (rng(step, 1) * 2.0f - 1.0f) / SCALE;
        float sn, cs;
        sincos32(ang, sn, cs);
        const float tip0 = sn, tip1 = cs, side0 = -tip1, side1 = tip0;
        // main engine
        const float m_power = a0 > 0.0f ? (clampf(a0, 0.0f, 1.0f) + 1.0f) * 0.5f : 0.0f;
        float ox = tip0 * (0.13333334f + 2.0f * d0) + side0 * d1;
        float oy = -tip1 * (0.13333334f + 2.0f * d0) - side1 * d1;
        vx = vx + (-ox * MAIN_POWER * m_power) * INV_M;
        vy = vy + (-oy * MAIN_POWER * m_power) * INV_M;
        // side engines
        const float direction = a1 < 0.0f ? -1.0f : 1.0f;
        const float s_power = fabsf(a1) > 0.5f ? clampf(fabsf(a1), 0.5f, 1.0f) : 0.0f;
        const float arm = 3.0f * d1 + direction * SIDE_AWAY;
        ox = tip0 * d0 + side0 * arm;
        oy = -tip1 * d0 - side1 * arm;
        const float px_ = -ox * SIDE_POWER * s_power, py_ = -oy * SIDE_POWER * s_power;
        float rx = ox - tip0 * 0.56666666f, ry = oy + tip1 * SIDE_H;
        vx = vx + px_ * INV_M;
        vy = vy + py_ * INV_M;
        om = om + (rx * py_ - ry * px_) * INV_I;
        // gravity
        vy = vy + GRAV * DT;
        // leg contacts
        float grx[2], gry[2], gvmin[2], gkn[2], gkt[2], accn[2] = {0.0f, 0.0f}, acct[2] = {0.0f, 0.0f};
#pragma unroll
        for (int leg = 0; leg < 2; ++leg) {
            const float pbx = leg == 0 ? -LEGX : LEGX;
            rx = cs * pbx - sn * LEGY;
            ry = sn * pbx + cs * LEGY;
            const float gap = (y + ry) - ground(x + rx);
            gvmin[leg] = gap >= 0.0f ? -gap / DT : BAUM * fmaxf(-gap - SLOP, 0.0f) / DT;
            gkn[leg] = INV_M + (rx * rx) * INV_I;
            gkt[leg] = INV_M + (ry * ry) * INV_I;
            grx[leg] = rx; gry[leg] = ry;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int leg = 0; leg < 2; ++leg) {
                const float vn = vy + om * grx[leg];
                float lam = -(vn - gvmin[leg]) / gkn[leg];
                float nw = fmaxf(accn[leg] + lam, 0.0f);
                float dl = nw - accn[leg];
                accn[leg] = nw;
                vy = vy + dl * INV_M;
                om = om + (grx[leg] * dl) * INV_I;
                const float vt = vx - om * gry[leg];
                lam = -vt / gkt[leg];
                const float lim = MU * accn[leg];
                nw = fminf(fmaxf(acct[leg] + lam, -lim), lim);
                dl = nw - acct[leg];
                acct[leg] = nw;
                vx = vx + dl * INV_M;
                om = om - (gry[leg] * dl) * INV_I;
            }
        }
        // integrate
        x = x + vx * DT;
        y = y + vy * DT;
        ang = ang + om * DT;
        c1 = accn[0] > 0.0f ? 1.0f : 0.0f;
        c2 = accn[1] > 0.0f ? 1.0f : 0.0f;
        pstep = pstep + 1.0f;
        // crash test
        sincos32(ang, sn, cs);
        bool crash = false;
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            const float hx = HULLX[v], hy = HULLY[v];
            const float wx = x + (cs * hx - sn * hy);
            const float wy = y + (sn * hx + cs * hy);
            crash = crash || (wy < ground(wx));
        }
        // rest detection
        const bool slow = ((vx * vx + vy * vy) < SLEEP_V2) && (fabsf(om) < SLEEP_W) && (c1 > 0.0f) && (c2 > 0.0f);
        sleep = slow ? sleep + 1.0f : 0.0f;
        const bool asleep = sleep >= SLEEP_STEPS;
        obs(o);
        const float shaping = ((-100.0f * sqrtf(o[0] * o[0] + o[1] * o[1]) - 100.0f * sqrtf(o[2] * o[2] + o[3] * o[3])) -
                               100.0f * fabsf(o[4])) + 10.0f * o[6] + 10.0f * o[7];
        float rew = hasp > 0.0f ? shaping - prev : 0.0f;
        prev = shaping;
        hasp = 1.0f;
        rew = (rew - m_power * 0.30f) - s_power * 0.03f;
        const bool out = crash || (fabsf(o[0]) >= 1.0f);
        rew = out ? -100.0f : (asleep ? 100.0f : rew);
        done_env = out || asleep;
        return rew;
    }
    // env.reset(): terrain, random initial impulse, one no-op step (oracle: LanderOracle.reset)
    __device__ void reset(float *o) {
        float hs[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) hs[j] = rng(RESET_STREAM, j) * 6.6666665f;
#pragma unroll
        for (int j = 3; j < 8; ++j) hs[j] = HELIPAD_Y;
#pragma unroll
        for (int i = 0; i < 11; ++i) terr[i] = 0.33f * ((hs[(i + 11) % 12] + hs[i]) + hs[i + 1]);
        const float fx = (rng(RESET_STREAM, 12) * 2.0f - 1.0f) * 1000.0f;
        const float fy = (rng(RESET_STREAM, 13) * 2.0f - 1.0f) * 1000.0f;
        x = 10.0f; y = H_;
        vx = (fx * DT) * INV_M; vy = (fy * DT) * INV_M;
        ang = om = c1 = c2 = prev = hasp = eplen = epret = sleep = pstep = 0.0f;
        bool d;
        (void)physics(0.0f, 0.0f, d, o);
    }
    __device__ void load(const float *S, long long n, long long i) {
        x = S[X * n + i]; y = S[Y * n + i]; vx = S[VX * n + i]; vy = S[VY * n + i]; ang = S[ANG * n + i];
        om = S[OM * n + i]; c1 = S[C1 * n + i]; c2 = S[C2 * n + i]; prev = S[PREV * n + i]; hasp = S[HASP * n + i];
        eplen = S[EPLEN * n + i]; epret = S[EPRET * n + i]; sleep = S[SLEEP * n + i]; epi = S[EPI * n + i];
        pstep = S[PSTEP * n + i];
#pragma unroll
        for (int t = 0; t < 11; ++t) terr[t] = S[(T0 + t) * n + i];
    }
    __device__ void store(float *S, long long n, long long i) const {
        S[X * n + i] = x; S[Y * n + i] = y; S[VX * n + i] = vx; S[VY * n + i] = vy; S[ANG * n + i] = ang;
        S[OM * n + i] = om; S[C1 * n + i] = c1; S[C2 * n + i] = c2; S[PREV * n + i] = prev; S[HASP * n + i] = hasp;
        S[EPLEN * n + i] = eplen; S[EPRET * n + i] = epret; S[SLEEP * n + i] = sleep; S[EPI * n + i] = epi;
        S[PSTEP * n + i] = pstep;
#pragma unroll
        for (int t = 0; t < 11; ++t) S[(T0 + t) * n + i] = terr[t];
    }
};

__global__ void __launch_bounds__(256) k_env_reset(float *S, long long n, uint32_t seed, const uint8_t *mask, float *obs_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Env e;
    e.seed = seed; e.id = (uint32_t)i;
    e.load(S, n, i);
    float o[8];
    if (!mask || mask[i]) {
        e.reset(o);
        e.store(S, n, i);
    } else {
        e.obs(o);
    }
    if (obs_out) {
        float4 *p = reinterpret_cast<float4 *>(obs_out + i * 8);
        p[0] = make_float4(o[0], o[1], o[2], o[3]);
        p[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

__global__ void __launch_bounds__(256) k_env_step(float *S, long long n, uint32_t seed, float max_ep_len, const float *act,
                                                  float *obs2, float *rew_out, float *done_out, float *next_obs,
                                                  uint8_t *ended_out, EnvStats *stats) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long n_end = 0, len_end = 0;
    double ret_end = 0.0;
    if (i < n) {
        Env e;
        e.seed = seed; e.id = (uint32_t)i;
        e.load(S, n, i);
        float o[8];
        bool done_env;
        const float2 a = *reinterpret_cast<const float2 *>(act + i * 2);
        const float rew = e.physics(a.x, a.y, done_env, o);
        e.eplen = e.eplen + 1.0f;                       // example/dsac.py:104
        e.epret = e.epret + rew;                        // :103
        const bool limit = e.eplen >= max_ep_len;
        const float done_store = limit ? 0.0f : (done_env ? 1.0f : 0.0f);  // :109
        const bool ended = done_env || limit;                              // :118
        if (obs2) {
            float4 *p = reinterpret_cast<float4 *>(obs2 + i * 8);
            p[0] = make_float4(o[0], o[1], o[2], o[3]);
            p[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (rew_out) rew_out[i] = rew;
        if (done_out) done_out[i] = done_store;
        if (ended_out) ended_out[i] = ended ? 1 : 0;
        if (ended) {
            n_end = 1; len_end = (long long)e.eplen; ret_end = (double)e.epret;
            e.epi = e.epi + 1.0f;
            e.reset(o);                                 // :127
        }
        if (next_obs) {
            float4 *p = reinterpret_cast<float4 *>(next_obs + i * 8);
            p[0] = make_float4(o[0], o[1], o[2], o[3]);
            p[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        e.store(S, n, i);
    }
    // episode statistics: wave reduction, one atomic per wave that saw an episode end
    for (int off = 32; off >= 1; off >>= 1) {
        n_end += __shfl_xor(n_end, off);
        len_end += __shfl_xor(len_end, off);
        ret_end += __shfl_xor(ret_end, off);
    }
    if ((threadIdx.x & 63) == 0 && n_end > 0) {
        atomicAdd((unsigned long long *)&stats->episodes, (unsigned long long)n_end);
        atomicAdd((unsigned long long *)&stats->len_sum, (unsigned long long)len_end);
        atomicAdd(&stats->ret_sum, ret_end);
    }
}

// `Wrapper(env, obs_noise, act_noise, reward_scale, action_repeat)` of algos/sac1/hyperparams.py:107-134
// around env.step, plus the n-step rollout's episode bookkeeping (algos/sac1/sac_ray.py:212-216,
// 238-258: ep_len counts WRAPPED steps, the episode ends on d or ep_len >= limit_steps, the stored
// done is the raw d).  The wrapper's np.random.random noise is the env's counter generator
// (slots 2.. of the per-step stream; a stream of its own for the reset observation).
constexpr uint32_t RESET_NOISE_STREAM = 0xFFFFFFEFu;
__global__ void __launch_bounds__(256) k_env_step_wrapped(float *S, long long n, uint32_t seed, float limit_steps, float *act,
                                                          float act_noise, float obs_noise, float reward_scale, int repeat,
                                                          float *obs2, float *rew_out, float *done_out, float *next_obs,
                                                          uint8_t *ended_out, EnvStats *stats) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long n_end = 0, len_end = 0;
    double ret_end = 0.0;
    if (i < n) {
        Env e;
        e.seed = seed; e.id = (uint32_t)i;
        e.load(S, n, i);
        float o[8];
        bool done_env = false;
        const uint32_t st0 = (uint32_t)e.pstep;
        const float2 a = *reinterpret_cast<const float2 *>(act + i * 2);
        const float a0 = a.x + act_noise * (-2.0f * e.rng(st0, 2) + 1.0f);   // hyperparams.py:124
        const float a1 = a.y + act_noise * (-2.0f * e.rng(st0, 3) + 1.0f);
        *reinterpret_cast<float2 *>(act + i * 2) = make_float2(a0, a1);       // `action += ...` mutates the caller's array: the rollout queues the NOISY action
        float r = 0.0f, rew = 0.0f;
        bool noisy = true;
        for (int k = 0; k < repeat; ++k) {                                    // :126-133
            const float rk = e.physics(a0, a1, done_env, o);
            r = r + rk;
            if (done_env && repeat != 1) { rew = 0.0f; break; }               // :130-131 (the reward is dropped)
            if (repeat == 1) { rew = r; noisy = false; break; }               // :132-133 (no noise, no scale)
            rew = reward_scale * r;                                           // :134 when the loop runs out
        }
        if (noisy) {
            const uint32_t st1 = (uint32_t)e.pstep;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = o[j] + obs_noise * (-2.0f * e.rng(st1, 4 + j) + 1.0f);
        }
        e.eplen = e.eplen + 1.0f;                                             // sac_ray.py:216
        e.epret = e.epret + rew;                                              // :215
        const bool ended = done_env || e.eplen >= limit_steps;                // :252
        if (obs2) {
            float4 *p = reinterpret_cast<float4 *>(obs2 + i * 8);
            p[0] = make_float4(o[0], o[1], o[2], o[3]);
            p[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (rew_out) rew_out[i] = rew;
        if (done_out) done_out[i] = done_env ? 1.0f : 0.0f;                   // raw d (the time-limit override is commented out, :221)
        if (ended_out) ended_out[i] = ended ? 1 : 0;
        if (ended) {
            n_end = 1; len_end = (long long)e.eplen; ret_end = (double)e.epret;
            e.epi = e.epi + 1.0f;
            e.reset(o);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = o[j] + obs_noise * (-2.0f * e.rng(RESET_NOISE_STREAM, j) + 1.0f);  // hyperparams.py:119-121
        }
        if (next_obs) {
            float4 *p = reinterpret_cast<float4 *>(next_obs + i * 8);
            p[0] = make_float4(o[0], o[1], o[2], o[3]);
            p[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        e.store(S, n, i);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        n_end += __shfl_xor(n_end, off);
        len_end += __shfl_xor(len_end, off);
        ret_end += __shfl_xor(ret_end, off);
    }
    if ((threadIdx.x & 63) == 0 && n_end > 0) {
        atomicAdd((unsigned long long *)&stats->episodes, (unsigned long long)n_end);
        atomicAdd((unsigned long long *)&stats->len_sum, (unsigned long long)len_end);
        atomicAdd(&stats->ret_sum, ret_end);
    }
}


// ------------------------------------------------------------------------------------------
// Fused rollout step (RolloutDevice.step = num_envs iterations of example/dsac.py:96-130 in the policy phase):
//   a = agent.get_action(o)      the policy-head partials of the forward launch in front of this kernel are summed in
//                                n-tile order, tanh-squashed with the counter-hash noise (ddrl_pol::policy_row)
//   o2, r, d, _ = env.step(a)    the physics above
//   replay_buffer.store(o, a, r, o2, d)   straight into ring row (ptr + i) % capacity — the n stores in env order,
//                                like ddrl_replay_store; the last block to finish advances the cursor
//   o = o2 (or env.reset())      written into the actor's observation buffer for the next forward launch
// One thread per env: no exchange between threads except the cursor ticket.
// ------------------------------------------------------------------------------------------
struct RolloutArgs {
    float *S;
    long long n;
    uint32_t seed;
    float max_ep_len;
    EnvStats *stats;
    // policy
    float *obs;            // [n][obs_dim] in / out (actor's buffer)
    const float *hp;       // [8][n][16]
    const float *bmu, *bls;
    int act, nt2;
    float scale;
    int deterministic;
    uint32_t noise_seed;
    unsigned long long noise_ctr;
    // version store of the actor (nullable): env i acts on slot[i] (bmu / bls are slot 0's then, slot s is vstride floats further)
    // and adopts the newest version where its episode ends — the reference worker's pull at episode end (example/dsac.py:127-130)
    int *slot;
    const int *newest;
    long long vstride;
    // ... and the NEXT forward's plan, written by this launch (nullable: vcnt): every env counts itself into its slot's group and files
    // itself in that group's row list; the last workgroup to finish turns the counts into the forward's workgroup table (what
    // k_version_plan does as a launch of its own, 7-11 us between this launch and the forward that waits for it)
    int *vcnt;             // [VER_MAX_SLOTS], zero on entry, zero again on exit
    int *perm;             // row lists: group s at perm[perm2d_off + s * n ..)
    long long perm2d_off;
    VerTile *vtiles;
    VerState *vs;
    int n_slots, col_tiles, wg_slots, vt_cap;
    // replay ring
    ddrl_replay_dev::RingState *rs;
    ddrl_replay_dev::RingPtrs ring;
    // optional mirrors for the host-side objects
    float *act_out, *next_obs_out;
};
template <int NH>  // head partial rows fetched per env: 4 (act_dim <= 2) or 8
__global__ void __launch_bounds__(64) k_env_step_pi(RolloutArgs a) {
    __shared__ long long s_ptr;
    __shared__ int s_last;
    if (threadIdx.x == 0) s_ptr = a.rs->ptr;
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, n = a.n;
    long long n_end = 0, len_end = 0;
    double ret_end = 0.0;
    int my_slot = 0;       // the version this env acts on at the NEXT step
    if (i < n) {
        // ---- loads first: head partials (mu heads 0..act-1, log_std heads act..2act-1), the acted-on observation, the env state
        const int nq = (a.nt2 + 3) >> 2;  // float4 groups of a partial row that hold tiles
        // (two statically indexed register arrays: one array indexed by `c + act` would live in scratch memory)
        constexpr int NA = NH / 2;
        float4 hm[NA][4], hl[NA][4];
#pragma unroll
        for (int c = 0; c < NA; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cm = c < a.act ? c : 0, qq = q < nq ? q : 0;
                hm[c][q] = *reinterpret_cast<const float4 *>(a.hp + ((long long)cm * n + i) * 16 + 4 * qq);
                hl[c][q] = *reinterpret_cast<const float4 *>(a.hp + ((long long)(a.act + cm) * n + i) * 16 + 4 * qq);
            }
        float o1[8];
        {
            const float4 *p = reinterpret_cast<const float4 *>(a.obs + i * 8);
            const float4 u = p[0], v = p[1];
            o1[0] = u.x; o1[1] = u.y; o1[2] = u.z; o1[3] = u.w; o1[4] = v.x; o1[5] = v.y; o1[6] = v.z; o1[7] = v.w;
        }
        Env e;
        e.seed = a.seed; e.id = (uint32_t)i;
        e.load(a.S, n, i);
        float mu[4], ls[4], ev[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float sm = 0.f, sl = 0.f;
            if (c < NA) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < nq) {  // n-tile order; slots beyond nt2 hold 0
                        const float4 m4 = hm[c < NA ? c : 0][q], l4 = hl[c < NA ? c : 0][q];
                        sm += m4.x; sm += m4.y; sm += m4.z; sm += m4.w;
                        sl += l4.x; sl += l4.y; sl += l4.z; sl += l4.w;
                    }
                }
            }
            const int cc = c < a.act ? c : 0;
            const long long voff = a.slot ? (long long)a.slot[i] * a.vstride : 0;
            mu[c] = sm + a.bmu[voff + cc];
            ls[c] = sl + a.bls[voff + cc];
            ev[c] = (a.deterministic || c >= a.act) ? 0.f : ddrl_pol::normal_at(a.noise_seed, a.noise_ctr + (unsigned long long)(i * a.act + c));
        }
        const ddrl_pol::PolRow pr = ddrl_pol::policy_row(mu, ls, ev, a.act, a.scale);
        const float a0 = a.deterministic ? tanhf(mu[0]) * a.scale : pr.act[0];
        const float a1 = a.act > 1 ? (a.deterministic ? tanhf(mu[1]) * a.scale : pr.act[1]) : 0.f;
        // ---- env.step + the worker's bookkeeping (as k_env_step)
        float o[8];
        bool done_env;
        const float rew = e.physics(a0, a1, done_env, o);
        e.eplen = e.eplen + 1.0f;                       // example/dsac.py:104
        e.epret = e.epret + rew;                        // :103
        const bool limit = e.eplen >= a.max_ep_len;
        const float done_store = limit ? 0.0f : (done_env ? 1.0f : 0.0f);  // :109
        const bool ended = done_env || limit;                              // :118
        // ---- replay_buffer.store(o, a, r, o2, d): ring arrays {obs1, obs2, acts, rews, done}
        const long long cap = a.ring.capacity;
        if (i >= n - cap) {  // rows that a later store of the same batch would overwrite are skipped (n > capacity)
            const long long row = (s_ptr + i) % cap;
            float4 *p1 = reinterpret_cast<float4 *>(a.ring.a[0] + row * 8), *p2 = reinterpret_cast<float4 *>(a.ring.a[1] + row * 8);
            p1[0] = make_float4(o1[0], o1[1], o1[2], o1[3]); p1[1] = make_float4(o1[4], o1[5], o1[6], o1[7]);
            p2[0] = make_float4(o[0], o[1], o[2], o[3]); p2[1] = make_float4(o[4], o[5], o[6], o[7]);
            *reinterpret_cast<float2 *>(a.ring.a[2] + row * 2) = make_float2(a0, a1);
            a.ring.a[3][row] = rew;
            a.ring.a[4][row] = done_store;
        }
        if (a.act_out) *reinterpret_cast<float2 *>(a.act_out + i * 2) = make_float2(a0, a1);
        if (ended) {
            n_end = 1; len_end = (long long)e.eplen; ret_end = (double)e.epret;
            e.epi = e.epi + 1.0f;
            e.reset(o);                                 // :127
            if (a.slot) a.slot[i] = *a.newest;          // :129-130  weights = ps.pull(keys); agent.set_weights(keys, weights)
        }
        if (a.vcnt) my_slot = ended ? *a.newest : a.slot[i];
        {
            float4 *p = reinterpret_cast<float4 *>(a.obs + i * 8);
            p[0] = make_float4(o[0], o[1], o[2], o[3]);
            p[1] = make_float4(o[4], o[5], o[6], o[7]);
            if (a.next_obs_out) {
                float4 *q = reinterpret_cast<float4 *>(a.next_obs_out + i * 8);
                q[0] = p[0]; q[1] = p[1];
            }
        }
        e.store(a.S, n, i);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        n_end += __shfl_xor(n_end, off);
        len_end += __shfl_xor(len_end, off);
        ret_end += __shfl_xor(ret_end, off);
    }
    if ((threadIdx.x & 63) == 0 && n_end > 0) {
        atomicAdd((unsigned long long *)&a.stats->episodes, (unsigned long long)n_end);
        atomicAdd((unsigned long long *)&a.stats->len_sum, (unsigned long long)len_end);
        atomicAdd(&a.stats->ret_sum, ret_end);
    }
    if (a.vcnt) {
        // this env's place in its group: the lanes of a wave that share a slot go as ONE atomic of their leader (a few thousand envs sit
        // on a handful of versions; all leaders' atomics leave in one instruction: one round trip whatever the number of groups)
        const int lane = threadIdx.x & 63;
        const bool valid = i < n;
        unsigned long long todo = __ballot(valid);
        int gsize = 0, rank = 0, leader = lane;
        while (todo) {
            const int ld = __ffsll((long long)todo) - 1;
            const int s0 = __builtin_amdgcn_readlane(my_slot, ld);   // (ld is wave-uniform: no trip through the LDS crossbar per group)
            const bool mine = valid && my_slot == s0;
            const unsigned long long m = __ballot(mine);
            if (mine) { gsize = __popcll(m); rank = __popcll(m & ((1ull << lane) - 1ull)); leader = ld; }
            todo &= ~m;
        }
        int base = 0;
        if (valid && lane == leader) base = atomicAdd(&a.vcnt[my_slot], gsize);
        base = __shfl(base, leader);
        if (valid) a.perm[a.perm2d_off + (long long)my_slot * n + base + rank] = (int)i;
        // (no fence: the counts are device-scope atomics whose results this wave has waited for — performed before its ticket below —
        // and nothing else crosses workgroups inside this launch: row lists, records and slots are read by the NEXT launch.  A
        // __threadfence() here writes back the L2's dirty lines — this launch's ring rows — in every workgroup: +8 us measured.)
    }
    // the last block to finish advances the ring cursor (every block has read rs->ptr before its ticket)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = atomicAdd(&a.rs->done_counter, 1u);
        s_last = ticket == gridDim.x - 1 ? 1 : 0;
        if (ticket == gridDim.x - 1) {
            const long long cap = a.ring.capacity;
            a.rs->ptr = (s_ptr + n) % cap;
            const long long sz = a.rs->size + n;
            a.rs->size = sz > cap ? cap : sz;
            a.rs->steps += n * a.ring.steps_inc;
            a.rs->done_counter = 0;
        }
    }
    if (!a.vcnt) return;
    __syncthreads();
    if (!s_last) return;   // block-uniform
    // ---- the last workgroup (one wave): the counts -> the next forward's workgroup table (the tables of k_version_plan, sac1.hip; the
    // row lists are this launch's own: group s at perm2d_off + s * n, so a record's row-list base is known without a scatter pass)
    __shared__ int t_cnt[VER_MAX_SLOTS], t_start[VER_MAX_SLOTS];
    __shared__ unsigned short t_slot[VER_MAX_SLOTS + 1024];   // slot of row tile ti (n / 32 + live groups <= 1024 + 2048 tiles)
    const int lane = threadIdx.x;   // (the counts are read with device-scope atomic loads, behind this workgroup's own ticket)
    constexpr int PER = VER_MAX_SLOTS / 64;
    const int per = (a.n_slots + 63) >> 6;   // slots per lane (<= PER): lane l holds slots [per l, per l + per)
    int c[PER], run = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int j = per * lane + q;
        c[q] = (q < per && j < a.n_slots) ? __hip_atomic_load(&a.vcnt[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        run += (c[q] + 31) >> 5;
    }
    int incl = run;
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    const int total = __shfl(incl, 63);
    // slot of every row tile: each group's first tile gets its slot number, a running maximum over the tiles fills the rest (groups lie
    // in slot order) — `chunk` tiles per lane, so the one big group of the newest version is not one lane's loop
    const int chunk = (total + 63) >> 6;
    for (int k = 0; k < chunk; ++k) t_slot[lane * chunk + k] = 0;
    __syncthreads();
    int ts = incl - run;   // first row tile of this lane's first slot
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (q < per) {   // wave-uniform
            const int j = per * lane + q, nt = (c[q] + 31) >> 5;
            if (j < VER_MAX_SLOTS) { t_cnt[j] = c[q]; t_start[j] = ts; }
            if (nt > 0) { t_slot[ts] = (unsigned short)j; a.vcnt[j] = 0; }   // (... and zero again for the next launch)
            ts += nt;
        }
    }
    const VerSplit sp = ver_split(total, a.col_tiles, a.wg_slots, a.vt_cap);
    if (lane == 0) { a.vs->n_tiles = total; a.vs->n_wgs = sp.n_wgs; }
    __syncthreads();
    {
        int mx = 0;
        for (int k = 0; k < chunk; ++k) mx = max(mx, (int)t_slot[lane * chunk + k]);
        int inc = mx;
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc = max(inc, u);
        }
        int runmx = __shfl_up(inc, 1);
        if (lane == 0) runmx = 0;
        for (int k = 0; k < chunk; ++k) {
            runmx = max(runmx, (int)t_slot[lane * chunk + k]);
            t_slot[lane * chunk + k] = (unsigned short)runmx;
        }
    }
    __syncthreads();
    // (one wave: every instruction counts — the divisions by the two group counts are done once, the per-record ones by reciprocal)
    const int n_long_wgs = sp.n_long * sp.g_long;
    const int gb_l = a.col_tiles / sp.g_long, ge_l = a.col_tiles % sp.g_long, gb_s = a.col_tiles / sp.g_short, ge_s = a.col_tiles % sp.g_short;
    const float inv_l = 1.0f / (float)sp.g_long, inv_s = 1.0f / (float)sp.g_short;
    const int base0 = (int)a.perm2d_off, ni = (int)n;
    for (int b = lane; b < sp.n_wgs; b += 64) {
        const bool lg = b < n_long_wgs;
        const int g = lg ? sp.g_long : sp.g_short, bb = lg ? b : b - n_long_wgs;
        const int q = (int)(((float)bb + 0.5f) * (lg ? inv_l : inv_s));   // bb / g: exact (bb < 2^20, g <= 16: the product is >= 0.03 off a whole number)
        const int ti = (lg ? 0 : sp.n_long) + q, grp = bb - q * g;
        const int j = t_slot[ti], k = ti - t_start[j], cj = t_cnt[j];
        const int gbase = lg ? gb_l : gb_s, gextra = lg ? ge_l : ge_s;
        const int ntl = gbase + (grp < gextra ? 1 : 0), nt0 = grp * gbase + (grp < gextra ? grp : gextra);
        a.vtiles[b] = VerTile{j, base0 + j * ni + 32 * k, cj - 32 * k < 32 ? cj - 32 * k : 32, nt0 | (ntl << 8)};
    }
}

__global__ void __launch_bounds__(256) k_env_obs(float *S, long long n, uint32_t seed, float *obs_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Env e;
    e.seed = seed; e.id = (uint32_t)i;
    e.load(S, n, i);
    float o[8];
    e.obs(o);
    float4 *p = reinterpret_cast<float4 *>(obs_out + i * 8);
    p[0] = make_float4(o[0], o[1], o[2], o[3]);
    p[1] = make_float4(o[4], o[5], o[6], o[7]);
}

}  // namespace

struct ddrl_env {
    int device;
    long long n;
    uint32_t seed;
    int max_ep_len;
    float *S;
    EnvStats *stats;
};

extern "C" {

int ddrl_env_create(ddrl_env_t **out, int device, int64_t n_envs, uint32_t seed, int32_t max_ep_len) {
    DDRL_REQUIRE(out != nullptr && n_envs > 0 && max_ep_len > 0, "bad out/n_envs/max_ep_len");
    DDRL_REQUIRE(max_ep_len < (1 << 24), "max_ep_len must stay exact in float32");
    ddrl::DeviceGuard g(device);
    if (!g.ok) { ddrl::set_error("cannot select device %d", device); return DDRL_ERR_HIP; }
    ddrl_env *h = new ddrl_env();
    h->device = device; h->n = n_envs; h->seed = seed; h->max_ep_len = max_ep_len; h->S = nullptr; h->stats = nullptr;
    if (hipMalloc((void **)&h->S, (size_t)NF * n_envs * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&h->stats, sizeof(EnvStats)) != hipSuccess) {
        ddrl::set_error("hipMalloc failed for %lld envs", (long long)n_envs);
        ddrl_env_destroy(h);
        return DDRL_ERR_NOMEM;
    }
    DDRL_HIP_CHECK(hipMemset(h->S, 0, (size_t)NF * n_envs * sizeof(float)));
    DDRL_HIP_CHECK(hipMemset(h->stats, 0, sizeof(EnvStats)));
    k_env_reset<<<(unsigned)((n_envs + 255) / 256), 256, 0, nullptr>>>(h->S, h->n, h->seed, nullptr, nullptr);
    DDRL_LAUNCH_CHECK();
    DDRL_HIP_CHECK(hipStreamSynchronize(nullptr));
    *out = h;
    return DDRL_OK;
}

int ddrl_env_destroy(ddrl_env_t *h) {
    if (!h) return DDRL_OK;
    ddrl::DeviceGuard g(h->device);
    (void)hipFree(h->S); (void)hipFree(h->stats);
    delete h;
    return DDRL_OK;
}

int ddrl_env_reset(ddrl_env_t *h, const uint8_t *mask_d, float *obs_d, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    ddrl::DeviceGuard g(h->device);
    k_env_reset<<<(unsigned)((h->n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(h->S, h->n, h->seed, mask_d, obs_d);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_env_step(ddrl_env_t *h, const float *act_d, float *obs2_d, float *rew_d, float *done_d, float *next_obs_d,
                  uint8_t *ended_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && act_d != nullptr, "NULL handle or action pointer");
    ddrl::DeviceGuard g(h->device);
    k_env_step<<<(unsigned)((h->n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(
        h->S, h->n, h->seed, (float)h->max_ep_len, act_d, obs2_d, rew_d, done_d, next_obs_d, ended_d, h->stats);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_env_step_wrapped(ddrl_env_t *h, float *act_d, float act_noise, float obs_noise, float reward_scale, int32_t action_repeat,
                          int32_t limit_steps, float *obs2_d, float *rew_d, float *done_d, float *next_obs_d, uint8_t *ended_d,
                          void *stream) {
    DDRL_REQUIRE(h != nullptr && act_d != nullptr, "NULL pointer");
    DDRL_REQUIRE(action_repeat >= 1 && limit_steps >= 1, "action_repeat and limit_steps must be >= 1");
    ddrl::DeviceGuard g(h->device);
    k_env_step_wrapped<<<(unsigned)((h->n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(
        h->S, h->n, h->seed, (float)limit_steps, act_d, act_noise, obs_noise, reward_scale, action_repeat, obs2_d, rew_d, done_d, next_obs_d,
        ended_d, h->stats);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_rollout_begin(ddrl_env_t *h, ddrl_actor_t *actor, void *stream) {
    DDRL_REQUIRE(h != nullptr && actor != nullptr, "NULL handle");
    const ddrl_actor_rollout_view v = ddrl_actor_internal_view(actor);
    DDRL_REQUIRE(v.ok, "actor has no direct-operand policy (shape outside the envelope): use ddrl_actor_act + ddrl_env_step + ddrl_replay_store");
    DDRL_REQUIRE(v.obs_dim == 8 && h->n <= v.max_rows && v.device == h->device, "actor / env mismatch (obs_dim 8, max_rows >= n_envs, same device)");
    ddrl::DeviceGuard g(h->device);
    k_env_obs<<<(unsigned)((h->n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(h->S, h->n, h->seed, v.obs);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_rollout_step(ddrl_env_t *h, ddrl_actor_t *actor, ddrl_replay_t *replay, int32_t n_steps, uint32_t noise_seed, uint64_t noise_ctr,
                      int deterministic, float *act_out_d, float *next_obs_out_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && actor != nullptr && replay != nullptr, "NULL handle");
    const ddrl_actor_rollout_view v = ddrl_actor_internal_view(actor);
    DDRL_REQUIRE(v.ok, "actor has no direct-operand policy (shape outside the envelope): use ddrl_actor_act + ddrl_env_step + ddrl_replay_store");
    DDRL_REQUIRE(v.obs_dim == 8 && v.act <= 2 && h->n <= v.max_rows && h->n % 32 == 0 && v.device == h->device,
                 "actor / env mismatch (obs_dim 8, act_dim <= 2, n_envs a multiple of 32 within max_rows, same device)");
    const ddrl_replay_dev::SamplerView rv = ddrl_replay_sampler_view(replay);
    DDRL_REQUIRE(rv.ring.n_arr == 5 && rv.ring.w[0] == 8 && rv.ring.w[1] == 8 && rv.ring.w[2] == 2 && rv.ring.w[3] == 1 && rv.ring.w[4] == 1,
                 "replay row shape must be (obs1[8], obs2[8], acts[2], rews, done)");
    DDRL_REQUIRE(!rv.ring.kind[0] && !rv.ring.kind[1], "the fused rollout step stores into float32 rings only (not a compact uint8 ring)");
    DDRL_REQUIRE(n_steps >= 1, "n_steps must be >= 1");
    ddrl::DeviceGuard g(h->device);
    RolloutArgs a{};
    a.S = h->S; a.n = h->n; a.seed = h->seed; a.max_ep_len = (float)h->max_ep_len; a.stats = h->stats;
    a.obs = v.obs; a.hp = v.hp; a.bmu = v.bmu; a.bls = v.bls; a.act = v.act; a.nt2 = v.nt2; a.scale = v.scale;
    a.deterministic = deterministic; a.noise_seed = noise_seed;
    a.rs = rv.state; a.ring = rv.ring; a.act_out = act_out_d; a.next_obs_out = next_obs_out_d;
    const bool store = v.n_slots > 0;
    DDRL_REQUIRE(!store || h->n == v.max_rows, "an actor with a version store steps exactly max_rows envs");
    for (int k = 0; k < n_steps; ++k) {  // the loop body of worker_rollout, n_steps times with the weights the actor holds
        // version store: once no install has happened for max_ep_len steps every env has been through an episode end and
        // acts on the newest version — the plain launch on the actor's current weights is then the same computation
        const bool versioned = store && *v.steps_since_install < (long long)h->max_ep_len;
        a.slot = versioned ? v.slot : nullptr;
        a.newest = versioned ? reinterpret_cast<const int *>(v.vs) : nullptr;   // VerState::newest is its first word
        a.vstride = v.vstride;
        a.bmu = versioned ? v.vbmu : v.bmu; a.bls = versioned ? v.vbls : v.bls;
        // the next forward's plan rides in this launch (base offsets of the row lists must fit the records' int)
        const bool fused_plan = versioned && v.vcnt != nullptr && v.perm2d_off + (long long)v.n_slots * h->n < (1ll << 31) && h->n / 32 <= 1024;
        a.vcnt = fused_plan ? v.vcnt : nullptr; a.perm = v.perm; a.perm2d_off = v.perm2d_off; a.vtiles = v.vtiles;
        a.vs = reinterpret_cast<VerState *>(const_cast<void *>(v.vs)); a.n_slots = v.n_slots; a.col_tiles = v.nt2; a.wg_slots = v.wg_slots; a.vt_cap = v.vt_cap;
        if (store) *v.steps_since_install += 1;
        const int rc = ddrl_actor_internal_forward(actor, h->n, stream, versioned ? 1 : 0);
        if (rc != DDRL_OK) return rc;
        a.noise_ctr = noise_ctr + (uint64_t)k * (uint64_t)h->n * (uint64_t)v.act;
        // one wave per workgroup: 4096 envs spread over 64 CUs instead of 16 (the kernel is a chain of dependent latencies)
        if (v.act <= 2) k_env_step_pi<4><<<(unsigned)((h->n + 63) / 64), 64, 0, ddrl::as_stream(stream)>>>(a);
        else k_env_step_pi<8><<<(unsigned)((h->n + 63) / 64), 64, 0, ddrl::as_stream(stream)>>>(a);
        DDRL_LAUNCH_CHECK();
        if (versioned) *v.plan_fresh = fused_plan;   // episode ends of this step moved envs to the newest version: the launch's own tail has planned for that, or the next forward plans
        ddrl_replay_note_store(replay, h->n);
    }
    return DDRL_OK;
}

int ddrl_env_stats(ddrl_env_t *h, int64_t *episodes_h, double *ret_sum_h, int64_t *len_sum_h, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    EnvStats st;
    DDRL_HIP_CHECK(hipMemcpyAsync(&st, h->stats, sizeof(st), hipMemcpyDeviceToHost, s));
    DDRL_HIP_CHECK(hipMemsetAsync(h->stats, 0, sizeof(EnvStats), s));
    DDRL_HIP_CHECK(hipStreamSynchronize(s));
    if (episodes_h) *episodes_h = st.episodes;
    if (ret_sum_h) *ret_sum_h = st.ret_sum;
    if (len_sum_h) *len_sum_h = st.len_sum;
    return DDRL_OK;
}

int ddrl_env_get_state(ddrl_env_t *h, float *state_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && state_d != nullptr, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    DDRL_HIP_CHECK(hipMemcpyAsync(state_d, h->S, (size_t)NF * h->n * sizeof(float), hipMemcpyDeviceToDevice, ddrl::as_stream(stream)));
    return DDRL_OK;
}

int ddrl_env_set_state(ddrl_env_t *h, const float *state_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && state_d != nullptr, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    DDRL_HIP_CHECK(hipMemcpyAsync(h->S, state_d, (size_t)NF * h->n * sizeof(float), hipMemcpyDeviceToDevice, ddrl::as_stream(stream)));
    return DDRL_OK;
}

}  // extern "C"
