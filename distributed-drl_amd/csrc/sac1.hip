// SAC1 learner update and batched policy forward on gfx950.
// Replaces Learner / Actor of algos/sac1/actor_learner.py:19-229 with the network of
// algos/sac1/core.py:91-121 (policy 8->400->300->(2,2), twin Q 10->400->300->1, ReLU).
//
// One update = 8 network evaluations (policy main@x, main@x2, target@x2; Q1,Q2 main@(x,a);
// Q1 main@(x,pi); Q1,Q2 target@(x2,pi_targ)), backward of pi_loss through Q1 and of value_loss,
// two TF1-Adam steps and the polyak update.  The dense fc work (hidden1 x hidden2 layers,
// forward / dgrad / wgrad) runs on the f32-input MFMA v_mfma_f32_32x32x2_f32, which is bit-exact
// f32 FMA arithmetic (no bf16: the 1e-5 loss parity forbids it); everything else is row-local
// VALU work with wavefront-shuffle reductions.  All stages of one update are launched
// back-to-back on one stream with no host synchronisation and no host-side state, so a whole
// update (or many) can be captured in a hipGraph.
//
// Numerics kept from the reference formulae (fp contraction is OFF for this file; FMAs are
// explicit): the only re-arrangement is z = eps*std/(std+EPS) for (pi-mu)/(std+EPS)
// (core.py:31 with pi = mu + eps*std, core.py:76-77) — algebraically identical, but free of the
// catastrophic cancellation that makes the literal float32 form noisy at the 1e-5 level
// (DESIGN.md §numerics).
#include "gemm_core.h"
#include "policy_row.h"

namespace {
struct L1Job {  // H1 = relu([in0 | in1] * W1 + b1); optionally also writes the concatenated input rows
    const float *in0, *in1, *W, *b;
    float *out, *aug_out;
    int d0, d1, rows, h1, ldo, aug_ld;
    int pre_only;  // 1: out = in0 * W1[0:d0] + b1 WITHOUT relu: the observation part of a Q layer 1 whose
                   //    action part is added by k_rows_a once the action exists (second-phase evaluations)
};
constexpr int MAX_L1_JOBS = 8;
struct OptState;
struct L1Jobs {
    int njobs;
    // tf.random_normal stand-in, generated here (job 0's row blocks) instead of by a kernel of its own
    int noise_on, act, n_each;
    uint32_t noise_seed;
    float *e0, *e1, *e2;
    const OptState *opt;
    L1Job job[MAX_L1_JOBS];
};

// ------------------------------------------------------------------------------------------
// K: layer 1 (K = obs_dim or obs_dim+act_dim: tiny) — VALU, threads along the output feature.
// Input rows and the W1 column block are staged in LDS with one coalesced burst each.
// ------------------------------------------------------------------------------------------
constexpr int L1_ROWS = 16;
constexpr int L1_MAXD = 40;
using ddrl_pol::normal_at;
using ddrl_pol::PolRow;
using ddrl_pol::policy_row;
__global__ void __launch_bounds__(256) k_l1(L1Jobs jobs) {
    const L1Job &jb = jobs.job[blockIdx.z];
    __shared__ float s_in[L1_ROWS][L1_MAXD];
    __shared__ float s_w[L1_MAXD][256];
    const int din = jb.pre_only ? jb.d0 : jb.d0 + jb.d1;
    const int r0 = blockIdx.y * L1_ROWS;
    if (r0 >= jb.rows) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int jc = j < jb.h1 ? j : jb.h1 - 1;
    const float bj = jb.b[jc];
    for (int k0 = 0; k0 < din; k0 += 16) {  // 16 loads in flight per round trip (din <= 16: one trip)
        float wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = jb.W[(long long)(k0 + u < din ? k0 + u : 0) * jb.h1 + jc];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (k0 + u < din) s_w[k0 + u][threadIdx.x] = wv[u];
    }
    for (int e = threadIdx.x; e < L1_ROWS * din; e += 256) {
        const int rr = e / din, k = e - rr * din;
        const int r = r0 + rr;
        const int rc = r < jb.rows ? r : jb.rows - 1;
        const float v = (k < jb.d0) ? jb.in0[(long long)rc * jb.d0 + k] : jb.in1[(long long)rc * jb.d1 + (k - jb.d0)];
        s_in[rr][k] = v;
        if (jb.aug_out && blockIdx.x == 0 && r < jb.rows) jb.aug_out[(long long)r * jb.aug_ld + k] = v;  // [x | a] rows for the layer-1 wgrad
    }
    if (jobs.noise_on && blockIdx.z == 0 && blockIdx.x == 0) {
        // eps_x, eps_x2, eps_t for this block's rows; element index as in one flat [3][B*act] fill
        const unsigned long long base = jobs.opt->noise_ctr;
        const int per_row = 3 * jobs.act;
        for (int e = threadIdx.x; e < L1_ROWS * per_row; e += 256) {
            const int rr = e / per_row, q = e - rr * per_row;
            const int wch = q / jobs.act, c = q - wch * jobs.act;
            const int r = r0 + rr;
            if (r < jb.rows) {
                const int k = r * jobs.act + c;
                (wch == 0 ? jobs.e0 : (wch == 1 ? jobs.e1 : jobs.e2))[k] =
                    normal_at(jobs.noise_seed, base + (unsigned long long)wch * jobs.n_each + k);
            }
        }
    }
    __syncthreads();
    float acc[L1_ROWS];
#pragma unroll
    for (int rr = 0; rr < L1_ROWS; ++rr) acc[rr] = 0.f;
    for (int k = 0; k < din; ++k) {
        const float w = s_w[k][threadIdx.x];
#pragma unroll
        for (int rr = 0; rr < L1_ROWS; ++rr) acc[rr] = fmaf(s_in[rr][k], w, acc[rr]);
    }
    if (j < jb.h1) {
#pragma unroll
        for (int rr = 0; rr < L1_ROWS; ++rr) {
            const int r = r0 + rr;
            const float v = acc[rr] + bj;
            if (r < jb.rows) jb.out[(long long)r * jb.ldo + j] = jb.pre_only ? v : fmaxf(v, 0.f);
        }
    }
}

struct NetPi { const float *W1, *b1, *W2, *b2, *Wmu, *bmu, *Wls, *bls; };
struct NetQ { const float *W1, *b1, *W2, *b2, *W3, *b3; };

// One policy head for one row held by one wave (hidden2 <= 512): per-dim quantities are valid in
// lanes c < act; every lane gets logp.  core.py:49-87,104-106.  `hrow` must be mask_row()ed.
// The head-kernel rows are fetched two action dims at a time (all of them at once for act <= 2).
struct HeadOut { float act, a, std, t, logp, eps; };
__device__ __forceinline__ HeadOut policy_head(const float (&hrow)[RV], int h2, int act, const NetPi &p, float eps_lane,
                                               float scale, int lane) {
    float mu = 0.f, lsr = 0.f;
    const float bm = p.bmu[lane < act ? lane : 0], bl = p.bls[lane < act ? lane : 0];
#pragma unroll
    for (int c0 = 0; c0 < MAXA; c0 += 2)
        if (c0 < act) {  // wave-uniform
            const int c1 = c0 + 1 < act ? c0 + 1 : c0;
            float wm0[RV], wl0[RV], wm1[RV], wl1[RV];
            load_row(p.Wmu, h2, lane, wm0, act, c0);
            load_row(p.Wls, h2, lane, wl0, act, c0);
            load_row(p.Wmu, h2, lane, wm1, act, c1);
            load_row(p.Wls, h2, lane, wl1, act, c1);
            const float s0 = wave_sum(dot_rv(hrow, wm0)), s1 = wave_sum(dot_rv(hrow, wl0));
            const float s2 = wave_sum(dot_rv(hrow, wm1)), s3 = wave_sum(dot_rv(hrow, wl1));
            if (lane == c0) { mu = s0 + bm; lsr = s1 + bl; }
            if (lane == c0 + 1 && c0 + 1 < act) { mu = s2 + bm; lsr = s3 + bl; }
        }
    HeadOut o;
    o.act = 0.f; o.a = 0.f; o.std = 0.f; o.t = 0.f; o.eps = eps_lane;
    float pre = 0.f, corr = 0.f;
    if (lane < act) {
        const float t = tanhf(lsr);
        const float log_std = -20.0f + 11.0f * (t + 1.0f);  // LOG_STD_MIN + 0.5*(MAX-MIN)*(ls+1), core.py:73-74
        const float std = expf(log_std);
        const float e = eps_lane;
        const float u = mu + e * std;                          // core.py:76-77
        const float z = (e * std) / (std + STD_EPS);           // == (pi - mu)/(std + EPS), core.py:31
        pre = -0.5f * ((z * z + 2.0f * log_std) + LOG2PI);
        const float a = tanhf(u);                              // core.py:83-84
        const float om = 1.0f - a * a;
        const float cl = fminf(fmaxf(om, 0.f), 1.f);           // clip_but_pass_gradient value, core.py:35-38,86
        corr = logf(cl + 1e-6f);
        o.act = a * scale; o.a = a; o.std = std; o.t = t;
    }
    float sp = 0.f, sc = 0.f;
#pragma unroll
    for (int c = 0; c < MAXA; ++c)
        if (c < act) { sp += __shfl(pre, c); sc += __shfl(corr, c); }
    o.logp = sp - sc;
    return o;
}

// ------------------------------------------------------------------------------------------
// K: heads, stage A — one wave per (row, eval in {pi@x, pi@x2, piT@x2, q1(x,a), q2(x,a)})
// ------------------------------------------------------------------------------------------
struct RowsA {
    const float *H2;  // [NEVAL][B][ldh2]
    NetPi pi_main, pi_targ;
    NetQ q1, q2;
    const float *e0, *e1, *e2;
    float *act0, *act2, *logp0, *logp1, *save0, *q1o, *q2o;
    // second-phase layer 1 (Q at the freshly sampled actions): slots 5, 6, 7 of H1 hold the
    // observation part x*W1[:obs] + b1 (k_l1, pre_only); this kernel adds act*W1[obs:] and the relu
    float *H1;                           // [NEVAL][B][ldh1]
    const float *Wa_q1, *Wa_q1t, *Wa_q2t;  // action rows W1[obs:obs+act, :] of main q1, target q1, target q2
    int B, h2, ldh2, act, h1, ldh1;
    float scale;
};
__device__ __forceinline__ void finish_l1(float *hx, int h1, int act, const float *__restrict__ Wa, float act_lane, int lane) {
    // hx[j] = relu(hx[j] + sum_c act[c] * Wa[c*h1 + j]); first two action dims fetched together
    float v[RV], w0[RV], w1[RV];
    const int cB = act > 1 ? 1 : 0;
    load_row(hx, h1, lane, v);
    load_row(Wa, h1, lane, w0);
    load_row(Wa + (long long)cB * h1, h1, lane, w1);
    const float a0 = __shfl(act_lane, 0), a1 = act > 1 ? __shfl(act_lane, 1) : 0.f;
#pragma unroll
    for (int i = 0; i < RV; ++i) v[i] = fmaf(a1, w1[i], fmaf(a0, w0[i], v[i]));
#pragma unroll
    for (int c = 2; c < MAXA; ++c)
        if (c < act) {
            float w[RV];
            load_row(Wa + (long long)c * h1, h1, lane, w);
            const float ac = __shfl(act_lane, c);
#pragma unroll
            for (int i = 0; i < RV; ++i) v[i] = fmaf(ac, w[i], v[i]);
        }
#pragma unroll
    for (int i = 0; i < RV; ++i) {
        const int j = lane + 64 * i;
        if (j < h1) hx[j] = fmaxf(v[i], 0.f);
    }
}
__global__ void __launch_bounds__(256) k_rows_a(RowsA a) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= a.B * 5) return;
    const int e = wid / a.B, r = wid - e * a.B;
    float hrow[RV];
    load_row(a.H2 + ((long long)e * a.B + r) * a.ldh2, a.h2, lane, hrow);
    if (e < 3) {
        const NetPi &p = (e == 2) ? a.pi_targ : a.pi_main;
        const float *eps = (e == 0 ? a.e0 : (e == 1 ? a.e1 : a.e2)) + (long long)r * a.act;
        const float el = eps[lane < a.act ? lane : 0];
        mask_row(hrow, a.h2, lane);
        const HeadOut o = policy_head(hrow, a.h2, a.act, p, el, a.scale, lane);
        const long long BH1 = (long long)a.B * a.ldh1;
        if (e == 0) {
            if (lane < a.act) {
                a.act0[r * a.act + lane] = o.act;
                *reinterpret_cast<float4 *>(a.save0 + ((long long)r * a.act + lane) * 4) = make_float4(o.a, o.std, o.t, o.eps);
            }
            if (lane == 0) a.logp0[r] = o.logp;
            finish_l1(a.H1 + 5 * BH1 + (long long)r * a.ldh1, a.h1, a.act, a.Wa_q1, o.act, lane);   // q1(x, pi)
        } else if (e == 1) {
            if (lane == 0) a.logp1[r] = o.logp;
        } else {
            if (lane < a.act) a.act2[r * a.act + lane] = o.act;
            finish_l1(a.H1 + 6 * BH1 + (long long)r * a.ldh1, a.h1, a.act, a.Wa_q1t, o.act, lane);  // q1T(x2, piT)
            finish_l1(a.H1 + 7 * BH1 + (long long)r * a.ldh1, a.h1, a.act, a.Wa_q2t, o.act, lane);  // q2T(x2, piT)
        }
    } else {
        const NetQ &q = (e == 3) ? a.q1 : a.q2;
        float w3[RV];
        load_row(q.W3, a.h2, lane, w3);
        const float b3 = q.b3[0];
        mask_row(hrow, a.h2, lane);
        const float v = wave_sum(dot_rv(hrow, w3)) + b3;
        if (lane == 0) (e == 3 ? a.q1o : a.q2o)[r] = v;
    }
}

// ------------------------------------------------------------------------------------------
// K: heads, stage B — Q(x,pi), target Qs, backup, losses, dq and dZ2 of the three Q paths.
// actor_learner.py:58-69.  One wave per row; deterministic block partials (summed by k_rows_c).
// ------------------------------------------------------------------------------------------
struct RowsB {
    const float *H2;
    NetQ q1, q2, q1t, q2t;
    const float *rew, *done, *logp0, *logp1, *q1o, *q2o;
    float *dZ2;  // [4][B][h2]  slots: 0 = q1(x,a), 1 = q2(x,a), 2 = q1(x,pi), 3 = pi
    float *dq4;  // [2][B][4]   (column 0 used; padded so that it is a 16-B aligned GEMM operand)
    float *loss_part;
    int B, h2, ldh2;
    float alpha, gamma;
};
__global__ void __launch_bounds__(64) k_rows_b(RowsB a) {
    // one wave = one row = one workgroup: 256 workgroups spread the row fetches over all CUs
    const int lane = threadIdx.x;
    const int r = blockIdx.x;
    float lpi = 0.f, l1 = 0.f, l2 = 0.f;
    if (r < a.B) {
        const long long BH = (long long)a.B * a.ldh2, BZ = (long long)a.B * a.h2;
        float h3[RV], h4[RV], h5[RV], h6[RV], h7[RV], w1[RV], w2[RV], w1t[RV], w2t[RV];
        // every load of the kernel up front: one memory round trip
        load_row(a.H2 + 3 * BH + (long long)r * a.ldh2, a.h2, lane, h3);
        load_row(a.H2 + 4 * BH + (long long)r * a.ldh2, a.h2, lane, h4);
        load_row(a.H2 + 5 * BH + (long long)r * a.ldh2, a.h2, lane, h5);
        load_row(a.H2 + 6 * BH + (long long)r * a.ldh2, a.h2, lane, h6);
        load_row(a.H2 + 7 * BH + (long long)r * a.ldh2, a.h2, lane, h7);
        load_row(a.q1.W3, a.h2, lane, w1);
        load_row(a.q2.W3, a.h2, lane, w2);
        load_row(a.q1t.W3, a.h2, lane, w1t);
        load_row(a.q2t.W3, a.h2, lane, w2t);
        const float rew = a.rew[r], done = a.done[r], lp0 = a.logp0[r], lp1 = a.logp1[r], q1v = a.q1o[r], q2v = a.q2o[r];
        const float b1 = a.q1.b3[0], b1t = a.q1t.b3[0], b2t = a.q2t.b3[0];
        mask_row(w1, a.h2, lane); mask_row(w1t, a.h2, lane); mask_row(w2t, a.h2, lane);
        const float q1pi = wave_sum(dot_rv(h5, w1)) + b1;
        const float q1t = wave_sum(dot_rv(h6, w1t)) + b1t;
        const float q2t = wave_sum(dot_rv(h7, w2t)) + b2t;
        const float minq = fminf(q1t, q2t);                          // actor_learner.py:59
        const float vb = minq - a.alpha * lp1;                       // :62
        const float backup = rew + (a.gamma * (1.0f - done)) * vb;   // :63
        const float e1 = backup - q1v, e2 = backup - q2v;
        lpi = a.alpha * lp0 - q1pi;                                  // :66
        l1 = e1 * e1; l2 = e2 * e2;                                  // :67-68
        const float inv_b = 1.0f / (float)a.B;
        const float dq1 = -e1 * inv_b, dq2 = -e2 * inv_b, dqp = -inv_b;
        if (lane == 0) {
            *reinterpret_cast<float4 *>(a.dq4 + (long long)r * 4) = make_float4(dq1, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(a.dq4 + ((long long)a.B + r) * 4) = make_float4(dq2, 0.f, 0.f, 0.f);
        }
        float *z0 = a.dZ2 + (long long)r * a.h2, *z1 = z0 + BZ, *z2 = z1 + BZ;
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int j = lane + 64 * i;
            if (j < a.h2) {
                z0[j] = h3[i] > 0.f ? dq1 * w1[i] : 0.f;
                z1[j] = h4[i] > 0.f ? dq2 * w2[i] : 0.f;
                z2[j] = h5[i] > 0.f ? dqp * w1[i] : 0.f;
            }
        }
    }
    // per-row loss terms; the sum over rows is done in a fixed order by the next row kernel
    // (k_rows_c), after the kernel boundary has made them visible — no in-kernel fence needed
    if (lane == 0 && r < a.B) {
        a.loss_part[r * 3 + 0] = lpi; a.loss_part[r * 3 + 1] = l1; a.loss_part[r * 3 + 2] = l2;
    }
}

// ------------------------------------------------------------------------------------------
// K: policy-head backward — d pi_loss / d (mu_raw, log_std_raw) and dZ2 of the policy trunk
// ------------------------------------------------------------------------------------------
struct RowsC {
    const float *H2;    // eval 0 rows (stride ldh2)
    const float *dZ1q;  // dZ1 of the q1(x,pi) path: [B][h1]
    const float *W1q1;  // main q1 layer-1 kernel [(obs+act)][h1]
    NetPi pi;
    const float *save0;
    float *dhead;  // [B][ldd]
    float *dZ2pi;  // [B][h2]
    const float *loss_part;  // [loss_blocks][3] from k_rows_b
    float *losses;           // [3] pi_loss, q1_loss, q2_loss (actor_learner.py:66-68)
    int B, h1, h2, ldh2, obs, act, ldd, loss_blocks;
    float alpha, scale;
    int nl;  // loss terms per row: 3 (SAC1) or 4 (SAC-v: + v_loss)
};
// Leading scalars (preloaded SGPRs): one base pointer + offsets and packed sizes — what the row workgroups'
// first (and only) burst of loads needs; the struct behind them arrives from the cold kernarg segment later.
__global__ void __launch_bounds__(64) k_rows_c(const float *base, int dz1_off, int h2_off, int w1q_off, int wmu_off, int wls_off, int save_off,
                                               int pk_h, int pk_l, int pk_a, RowsC a) {
    const int lane = threadIdx.x;
    const int nrows = pk_a >> 16;
    if ((int)blockIdx.x == nrows) {  // (gridDim is a hidden-argument load)
        // reduce_mean over the batch: the extra last workgroup sums the per-row terms of k_rows_b in
        // a fixed order (lane-strided partial sums, then the xor-shuffle tree)
        const int ln = lane;
        float s3[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b0 = 0; b0 < a.loss_blocks; b0 += 64) {
            const int b = b0 + ln;
            const bool ok = b < a.loss_blocks;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v = a.loss_part[(ok ? b : 0) * a.nl + (c < a.nl ? c : 0)];
                s3[c] += (ok && c < a.nl) ? v : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float tot = wave_sum(s3[c]);
            const float mean = tot / (float)a.B;
            if (ln == 0 && c < a.nl) a.losses[c] = c == 0 ? mean : 0.5f * mean;
        }
        return;
    }
    const int r = blockIdx.x;
    if (r >= nrows) return;
    // everything this wave needs for the first two action dims, in one burst
    float dz[RV], hrow[RV], wq0[RV], wq1[RV], wm0[RV], wl0[RV], wm1[RV], wl1[RV];
    const int ph1 = pk_h & 0xffff, ph2 = pk_h >> 16, pldh2 = pk_l & 0xffff, pobs = pk_l >> 16, pact = pk_a & 0xffff;
    const int cB = pact > 1 ? 1 : 0;
    load_row(base + dz1_off + (long long)r * ph1, ph1, lane, dz);
    load_row(base + h2_off + (long long)r * pldh2, ph2, lane, hrow);
    load_row(base + w1q_off + (long long)(pobs + 0) * ph1, ph1, lane, wq0);
    load_row(base + w1q_off + (long long)(pobs + cB) * ph1, ph1, lane, wq1);
    load_row(base + wmu_off, ph2, lane, wm0, pact, 0);
    load_row(base + wls_off, ph2, lane, wl0, pact, 0);
    load_row(base + wmu_off, ph2, lane, wm1, pact, cB);
    load_row(base + wls_off, ph2, lane, wl1, pact, cB);
    const float4 sv = *reinterpret_cast<const float4 *>(base + save_off + ((long long)r * pact + (lane < pact ? lane : 0)) * 4);
    mask_row(dz, a.h1, lane);
    float ga = 0.f;
    {
        const float s0 = wave_sum(dot_rv(dz, wq0)), s1 = wave_sum(dot_rv(dz, wq1));
        if (lane == 0) ga = s0;
        if (lane == 1 && a.act > 1) ga = s1;
    }
#pragma unroll
    for (int c = 2; c < MAXA; ++c)
        if (c < a.act) {  // act > 2: one more round trip per extra dim
            float w[RV];
            load_row(a.W1q1 + (long long)(a.obs + c) * a.h1, a.h1, lane, w);
            const float sdot = wave_sum(dot_rv(dz, w));
            if (lane == c) ga = sdot;
        }
    float dmu = 0.f, dls = 0.f;
    if (lane < a.act) {
        const float av = sv.x, std = sv.y, t = sv.z, e = sv.w;
        const float glp = a.alpha / (float)a.B;  // d pi_loss / d logp_pi
        const float om = 1.0f - av * av;
        const float cl = fminf(fmaxf(om, 0.f), 1.f);
        const float du = (ga * a.scale) * om + glp * ((2.0f * av * om) / (cl + 1e-6f));
        const float sd = std + STD_EPS;
        const float z = (e * std) / sd;
        const float dzdl = ((e * std) * STD_EPS) / (sd * sd);
        const float dl = du * (e * std) + glp * (-(z * dzdl) - 1.0f);
        dmu = du;
        dls = dl * (11.0f * (1.0f - t * t));
    }
    float gm[MAXA], gl[MAXA];
#pragma unroll
    for (int c = 0; c < MAXA; ++c) { gm[c] = __shfl(dmu, c); gl[c] = __shfl(dls, c); }
    if (lane < a.ldd) {  // dhead row = [dmu | dls | zero padding]
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < MAXA; ++c) {
            if (c < a.act && lane == c) v = gm[c];
            if (c < a.act && lane == a.act + c) v = gl[c];
        }
        a.dhead[(long long)r * a.ldd + lane] = v;
    }
    float acc[RV];
#pragma unroll
    for (int i = 0; i < RV; ++i) {
        acc[i] = fmaf(gl[0], wl0[i], gm[0] * wm0[i]);
        if (a.act > 1) { acc[i] = fmaf(gm[1], wm1[i], acc[i]); acc[i] = fmaf(gl[1], wl1[i], acc[i]); }
    }
#pragma unroll
    for (int c = 2; c < MAXA; ++c)
        if (c < a.act) {
            float wm[RV], wl[RV];
            load_row(a.pi.Wmu, a.h2, lane, wm, a.act, c);
            load_row(a.pi.Wls, a.h2, lane, wl, a.act, c);
#pragma unroll
            for (int i = 0; i < RV; ++i) {
                acc[i] = fmaf(gm[c], wm[i], acc[i]);
                acc[i] = fmaf(gl[c], wl[i], acc[i]);
            }
        }
    float *out = a.dZ2pi + (long long)r * a.h2;
#pragma unroll
    for (int i = 0; i < RV; ++i) {
        const int j = lane + 64 * i;
        if (j < a.h2) out[j] = hrow[i] > 0.f ? acc[i] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// SAC-v (example/model.py:17-76: policy + twin Q + V + target V).  Evaluations and their H1/H2 slots:
// 0 pi(x)  1 q1(x,a)  2 q2(x,a)  3 q1(x,pi)  4 q2(x,pi)  5 v(x)  6 v_targ(x2).
// ------------------------------------------------------------------------------------------
struct RowsAV {
    const float *H2;
    NetPi pi;
    NetQ q1, q2, v, vt;
    const float *e0;
    float *act0, *logp0, *save0, *q1o, *q2o, *vo, *vto;
    float *H1;
    const float *Wa_q1, *Wa_q2;  // action rows of the main q1 / q2 layer-1 kernels
    int B, h2, ldh2, act, h1, ldh1;
    float scale;
};
__global__ void __launch_bounds__(256) k_rows_a_v(RowsAV a) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= a.B * 5) return;
    const int k = wid / a.B, r = wid - k * a.B;
    const int e = k == 0 ? 0 : (k == 1 ? 1 : (k == 2 ? 2 : (k == 3 ? 5 : 6)));
    float hrow[RV];
    load_row(a.H2 + ((long long)e * a.B + r) * a.ldh2, a.h2, lane, hrow);
    if (k == 0) {
        const float el = a.e0[(long long)r * a.act + (lane < a.act ? lane : 0)];
        mask_row(hrow, a.h2, lane);
        const HeadOut o = policy_head(hrow, a.h2, a.act, a.pi, el, a.scale, lane);
        if (lane < a.act) {
            a.act0[r * a.act + lane] = o.act;
            *reinterpret_cast<float4 *>(a.save0 + ((long long)r * a.act + lane) * 4) = make_float4(o.a, o.std, o.t, o.eps);
        }
        if (lane == 0) a.logp0[r] = o.logp;
        const long long BH1 = (long long)a.B * a.ldh1;
        finish_l1(a.H1 + 3 * BH1 + (long long)r * a.ldh1, a.h1, a.act, a.Wa_q1, o.act, lane);  // q1(x, pi)
        finish_l1(a.H1 + 4 * BH1 + (long long)r * a.ldh1, a.h1, a.act, a.Wa_q2, o.act, lane);  // q2(x, pi)
    } else {
        const NetQ &q = k == 1 ? a.q1 : (k == 2 ? a.q2 : (k == 3 ? a.v : a.vt));
        float w3[RV];
        load_row(q.W3, a.h2, lane, w3);
        const float b3 = q.b3[0];
        mask_row(hrow, a.h2, lane);
        const float v = wave_sum(dot_rv(hrow, w3)) + b3;
        if (lane == 0) (k == 1 ? a.q1o : (k == 2 ? a.q2o : (k == 3 ? a.vo : a.vto)))[r] = v;
    }
}

// example/model.py:33-47: min double-Q, the Q and V regression targets, the four losses and their
// derivatives w.r.t. the head outputs; dZ2 slots: 0 q1(x,a)  1 q2(x,a)  2 q1(x,pi)  3 v(x)  (4 = pi)
struct RowsBV {
    const float *H2;
    NetQ q1, q2, v;
    const float *rew, *done, *logp0, *q1o, *q2o, *vo, *vto;
    float *dZ2, *dq4, *loss_part;
    int B, h2, ldh2;
    float alpha, gamma;
};
__global__ void __launch_bounds__(64) k_rows_b_v(RowsBV a) {
    const int lane = threadIdx.x;
    const int r = blockIdx.x;
    if (r >= a.B) return;
    const long long BH = (long long)a.B * a.ldh2, BZ = (long long)a.B * a.h2;
    float h1r[RV], h2r[RV], h3r[RV], h4r[RV], h5r[RV], w1[RV], w2[RV], wv[RV];
    load_row(a.H2 + 1 * BH + (long long)r * a.ldh2, a.h2, lane, h1r);
    load_row(a.H2 + 2 * BH + (long long)r * a.ldh2, a.h2, lane, h2r);
    load_row(a.H2 + 3 * BH + (long long)r * a.ldh2, a.h2, lane, h3r);
    load_row(a.H2 + 4 * BH + (long long)r * a.ldh2, a.h2, lane, h4r);
    load_row(a.H2 + 5 * BH + (long long)r * a.ldh2, a.h2, lane, h5r);
    load_row(a.q1.W3, a.h2, lane, w1);
    load_row(a.q2.W3, a.h2, lane, w2);
    load_row(a.v.W3, a.h2, lane, wv);
    const float rew = a.rew[r], done = a.done[r], lp0 = a.logp0[r], q1v = a.q1o[r], q2v = a.q2o[r], vv = a.vo[r], vt = a.vto[r];
    const float b1 = a.q1.b3[0], b2 = a.q2.b3[0];
    mask_row(w1, a.h2, lane); mask_row(w2, a.h2, lane);
    const float q1pi = wave_sum(dot_rv(h3r, w1)) + b1;
    const float q2pi = wave_sum(dot_rv(h4r, w2)) + b2;
    const float minq = fminf(q1pi, q2pi);                              // model.py:34
    const float q_backup = rew + (a.gamma * (1.0f - done)) * vt;       // :37
    const float v_backup = minq - a.alpha * lp0;                       // :38
    const float e1 = q_backup - q1v, e2 = q_backup - q2v, ev = v_backup - vv;
    const float inv_b = 1.0f / (float)a.B;
    const float dq1 = -e1 * inv_b, dq2 = -e2 * inv_b, dv = -ev * inv_b, dqp = -inv_b;
    if (lane == 0) {
        *reinterpret_cast<float4 *>(a.dq4 + (long long)r * 4) = make_float4(dq1, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(a.dq4 + ((long long)a.B + r) * 4) = make_float4(dq2, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(a.dq4 + ((long long)2 * a.B + r) * 4) = make_float4(dv, 0.f, 0.f, 0.f);
        a.loss_part[r * 4 + 0] = a.alpha * lp0 - q1pi;                 // :41
        a.loss_part[r * 4 + 1] = e1 * e1;                              // :42
        a.loss_part[r * 4 + 2] = e2 * e2;                              // :43
        a.loss_part[r * 4 + 3] = ev * ev;                              // :44
    }
    float *z0 = a.dZ2 + (long long)r * a.h2, *z1 = z0 + BZ, *z2 = z1 + BZ, *z3 = z2 + BZ;
#pragma unroll
    for (int i = 0; i < RV; ++i) {
        const int j = lane + 64 * i;
        if (j < a.h2) {
            z0[j] = h1r[i] > 0.f ? dq1 * w1[i] : 0.f;
            z1[j] = h2r[i] > 0.f ? dq2 * w2[i] : 0.f;
            z2[j] = h3r[i] > 0.f ? dqp * w1[i] : 0.f;
            z3[j] = h5r[i] > 0.f ? dv * wv[i] : 0.f;
        }
    }
}

#include "sac1_direct.h"

struct StageArgs {
    const float *src[8];
    float *dst[8];
    int n[8];
};
__global__ void __launch_bounds__(256) k_stage(StageArgs a) {
    const int w = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n[w]; i += gridDim.x * 256) a.dst[w][i] = a.src[w][i];
}

__global__ void k_fill_col4(float *p, long long groups, int ld, int col) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g < groups) *reinterpret_cast<float4 *>(p + (g * ld + col) * 4) = make_float4(1.f, 1.f, 1.f, 1.f);
}

__global__ void k_copy3(const float *a, const float *b, const float *c, float *oa, float *ob, float *oc, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        if (oa) oa[i] = a[i];
        if (ob) ob[i] = b[i];
        if (oc) oc[i] = c[i];
    }
}

// gradient of the fused pi layer-1 wgrad = sum of the row-tile partials (same order as in Adam)
__global__ void __launch_bounds__(256) k_reduce_parts(const float *__restrict__ part, float *__restrict__ g, long long n, long long stride,
                                                      int nparts) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float sacc = 0.f;  // same order of additions as k_adam_polyak
    for (int q = 0; q < nparts; ++q) sacc += part[(long long)q * stride + i];
    g[i] = sacc;
}

// the same for the direct path's layer-1 block layout: partial [q][k][j] -> gradient element w1y_index(k, j)
__global__ void __launch_bounds__(256) k_reduce_parts_w1y(const float *__restrict__ part, float *__restrict__ g, int nk, int N, int nparts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nk * N) return;
    const int k = i / N, j = i - k * N;
    float sacc = 0.f;  // tile order, as the last-arriver step of k_dg sums them
    for (int q = 0; q < nparts; ++q) sacc += part[(long long)q * nk * N + i];
    g[w1y_index(k, j)] = sacc;
}

// ------------------------------------------------------------------------------------------
// K: batched get_action — one wave per observation row (Actor.get_action, actor_learner.py:195-197)
// ------------------------------------------------------------------------------------------
struct ActArgs {
    const float *H2;
    NetPi pi;
    const float *eps;
    float *act_out;
    int rows, h2, ldh2, act, deterministic;
    float scale;
};
__global__ void __launch_bounds__(256) k_rows_act(ActArgs a) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.rows) return;
    float hrow[RV];
    load_row(a.H2 + r * a.ldh2, a.h2, lane, hrow);
    const float el = (a.deterministic || !a.eps) ? 0.f : a.eps[r * a.act + (lane < a.act ? lane : 0)];
    mask_row(hrow, a.h2, lane);
    const HeadOut o = policy_head(hrow, a.h2, a.act, a.pi, el, a.scale, lane);
    if (lane < a.act) a.act_out[r * a.act + lane] = o.act;
}

// get_action of ONE observation as one launch: every workgroup computes layer 1 (a few thousand products) and ITS 16 columns of layer 2
// — 256 threads = 16 columns x 16 slices of the contraction, combined in slice order — and leaves the head's partial sums of those
// columns; the last workgroup to finish (ticket) adds the partials in workgroup order, takes the noise elements from the counter
// (== ddrl_normal_fill) and squashes (ddrl_pol::policy_row).  One workgroup alone pulls layer 2's 480 KB through one CU (~25 us);
// the batched path is three launches behind a noise launch — a chain of four for one row, which is what a reference-style
// rollout worker pays per env step (actor_learner.py:195-197; example/dsac.py:96).
constexpr int A1_COLS = 16;
struct ActOneArgs {
    const float *obs;
    NetPi pi;
    float *act_out;
    float *part;             // [workgroups][16]: head partial sums (mu 0..act-1, log_std act..2act-1)
    unsigned int *ticket;
    int d0, h1, h2, act, deterministic;
    float scale;
    uint32_t seed;
    unsigned long long ctr;
};
__global__ void __launch_bounds__(256) k_act_one(ActOneArgs a) {
    __shared__ float xs[64];
    __shared__ float h1s[512];
    __shared__ float ps[16][A1_COLS + 1];
    __shared__ float v2[A1_COLS];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    if (tid < a.d0) xs[tid] = a.obs[tid];
    __syncthreads();
    const int col = tid & (A1_COLS - 1), slice = tid >> 4, c = blockIdx.x * A1_COLS + col;
    const int per = (a.h1 + 15) >> 4, k0 = slice * per, k1 = k0 + per < a.h1 ? k0 + per : a.h1;   // per <= 32
    // this thread's layer-2 operands are requested now (they do not depend on layer 1): one round trip under the layer-1 phase
    float w2[32];
    const int cc2 = c < a.h2 ? c : 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int k = k0 + q < k1 ? k0 + q : (k0 < a.h1 ? k0 : 0);
        w2[q] = a.pi.W2[(long long)k * a.h2 + cc2];
    }
    {   // layer 1: units tid and tid + 256, eight input rows per round trip for both
        const int j0 = tid < a.h1 ? tid : 0, j1 = tid + 256 < a.h1 ? tid + 256 : 0;
        float acc0 = a.pi.b1[j0], acc1 = a.pi.b1[j1];
        for (int d = 0; d < a.d0; d += 8) {
            float u0[8], u1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int dd = d + q < a.d0 ? d + q : 0;
                u0[q] = a.pi.W1[(long long)dd * a.h1 + j0]; u1[q] = a.pi.W1[(long long)dd * a.h1 + j1];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (d + q < a.d0) { acc0 += xs[d + q] * u0[q]; acc1 += xs[d + q] * u1[q]; }
        }
        if (tid < a.h1) h1s[tid] = fmaxf(acc0, 0.f);
        if (tid + 256 < a.h1) h1s[tid + 256] = fmaxf(acc1, 0.f);
    }
    __syncthreads();
    float acc = 0.f;
    if (c < a.h2) {
#pragma unroll
        for (int q = 0; q < 32; ++q)
            if (k0 + q < k1) acc += h1s[k0 + q] * w2[q];   // k order
    }
    ps[slice][col] = acc;
    __syncthreads();
    if (tid < A1_COLS) {
        const int cc = blockIdx.x * A1_COLS + tid;
        float sum = 0.f;
        if (cc < a.h2) {
            sum = a.pi.b2[cc];
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += ps[q][tid];
            sum = fmaxf(sum, 0.f);
        }
        v2[tid] = sum;
    }
    __syncthreads();
    if (tid < 2 * a.act) {   // head partials of this workgroup's columns: outputs 0..act-1 mu, act..2act-1 log_std
        const float *W = tid < a.act ? a.pi.Wmu : a.pi.Wls;
        const int o = tid < a.act ? tid : tid - a.act;
        float sum = 0.f;
        for (int q = 0; q < A1_COLS; ++q) {
            const int cc = blockIdx.x * A1_COLS + q;
            if (cc < a.h2) sum += v2[q] * W[(long long)cc * a.act + o];
        }
        a.part[blockIdx.x * 16 + tid] = sum;
    }
    __threadfence();   // the partials are out before the ticket (this launch has written nothing else)
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(a.ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    __shared__ float s_part[32 * 16];
    for (int i = tid; i < (int)gridDim.x * 16; i += 256)   // every partial in ONE round trip (a serial chain of device-scope loads by one thread: +7 us)
        s_part[i] = __hip_atomic_load(&a.part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, ls[4] = {0.f, 0.f, 0.f, 0.f}, ev[4] = {0.f, 0.f, 0.f, 0.f};
        for (int o = 0; o < a.act; ++o) {
            float sm = 0.f, sl = 0.f;
            for (unsigned b = 0; b < gridDim.x; ++b) {   // workgroup order
                sm += s_part[b * 16 + o];
                sl += s_part[b * 16 + a.act + o];
            }
            mu[o] = sm + a.pi.bmu[o]; ls[o] = sl + a.pi.bls[o];
            ev[o] = a.deterministic ? 0.f : ddrl_pol::normal_at(a.seed, a.ctr + (unsigned long long)o);
        }
        const ddrl_pol::PolRow pr = ddrl_pol::policy_row(mu, ls, ev, a.act, a.scale);
        for (int o = 0; o < a.act; ++o) a.act_out[o] = pr.act[o];
        *a.ticket = 0u;
    }
}

// ------------------------------------------------------------------------------------------
// host side: layout
// ------------------------------------------------------------------------------------------

struct Layout {
    // internal (padded) offsets
    long long pi_W1, pi_b1, pi_W2, pi_b2, pi_Wmu, pi_bmu, pi_Wls, pi_bls;
    long long q_W1[2], q_b1[2], q_W2[2], q_b2[2], q_W3[2], q_b3[2];
    long long v_W1, v_b1, v_W2, v_b2, v_W3, v_b3;  // SAC-v only (example/model.py: 'main/v')
    long long n_pi_int, total_int;
    long long n_pi, n_q, total_ext;
    std::vector<Seg> segs;  // 20 tensors in external order
    bool direct;            // layer-2 kernels k4-interleaved [Kp1/4][Np2][4] with a zero-padded bias [Np2] (sac1_direct.h)
    int Kp1, Np2;
};

static Layout make_layout(const ddrl_sac1_config_t &c, bool pi_only, bool direct = false) {
    // Internal layout: every kernel is followed IMMEDIATELY by its bias (so that a bias gradient
    // is just one more row of the kernel's wgrad GEMM); each (kernel, bias) pair starts on a
    // 16-byte boundary.  Direct path: the layer-2 kernels are stored k4-interleaved and padded to
    // whole 32-wide tiles instead (pads are zero and stay zero: their gradient is never written).
    Layout L;
    const long long o = c.obs_dim, a = c.act_dim, h1 = c.hidden1, h2 = c.hidden2;
    L.direct = direct;
    L.Kp1 = (int)((h1 + 31) & ~31ll); L.Np2 = (int)((h2 + 31) & ~31ll);
    long long in = 0, ext = 0;
    auto add = [&](long long &slot, long long n, bool pad_after) {
        slot = in;
        L.segs.push_back(Seg{ext, in, n});
        in += n;
        if (pad_after) in = pad4(in);
        ext += n;
    };
    auto add_w2 = [&](long long &wslot, long long &bslot) {
        if (!direct) { add(wslot, h1 * h2, false); add(bslot, h2, true); return; }
        in = pad4(in);
        wslot = in;
        L.segs.push_back(Seg{ext, in, h1 * h2, (int)h2, L.Np2, 1, 0});
        in += (long long)L.Kp1 * L.Np2; ext += h1 * h2;
        bslot = in;
        L.segs.push_back(Seg{ext, in, h2});
        in += L.Np2; ext += h2;
    };
    auto add_w1 = [&](long long &wslot, long long &bslot, long long D) {  // [W1 ; b1]
        if (!direct) { add(wslot, D * h1, false); add(bslot, h1, true); return; }
        in = pad4(in);
        wslot = bslot = in;   // one block array holds the kernel rows and, as input column D, the bias
        L.segs.push_back(Seg{ext, in, D * h1, (int)h1, 0, 2, 0});
        ext += D * h1;
        L.segs.push_back(Seg{ext, in, h1, (int)h1, 0, 2, (int)D});
        ext += h1;
        in += (long long)L.Kp1 * 16;
    };
    add_w1(L.pi_W1, L.pi_b1, o); add_w2(L.pi_W2, L.pi_b2);
    add(L.pi_Wmu, h2 * a, false); add(L.pi_bmu, a, true); add(L.pi_Wls, h2 * a, false); add(L.pi_bls, a, true);
    L.n_pi_int = in;
    L.n_pi = ext;
    for (int q = 0; q < 2 && !pi_only; ++q) {
        add_w1(L.q_W1[q], L.q_b1[q], o + a); add_w2(L.q_W2[q], L.q_b2[q]);
        add(L.q_W3[q], h2, false); add(L.q_b3[q], 1, true);
    }
    L.v_W1 = L.v_b1 = L.v_W2 = L.v_b2 = L.v_W3 = L.v_b3 = -1;
    if (!pi_only && c.variant == DDRL_SAC_V) {  // vf_mlp(x): obs -> h1 -> h2 -> 1 (example/core.py:112-113)
        add_w1(L.v_W1, L.v_b1, o); add_w2(L.v_W2, L.v_b2);
        add(L.v_W3, h2, false); add(L.v_b3, 1, true);
    }
    L.total_int = in;
    L.total_ext = ext;
    L.n_q = pi_only ? 0 : (ext - L.n_pi) / 2;
    return L;
}

static NetPi net_pi(const float *base, const Layout &L) {
    return NetPi{base + L.pi_W1, base + L.pi_b1, base + L.pi_W2, base + L.pi_b2,
                 base + L.pi_Wmu, base + L.pi_bmu, base + L.pi_Wls, base + L.pi_bls};
}
static NetQ net_q(const float *base, const Layout &L, int q) {
    return NetQ{base + L.q_W1[q], base + L.q_b1[q], base + L.q_W2[q], base + L.q_b2[q], base + L.q_W3[q], base + L.q_b3[q]};
}
static NetQ net_v(const float *base, const Layout &L) {
    return NetQ{base + L.v_W1, base + L.v_b1, base + L.v_W2, base + L.v_b2, base + L.v_W3, base + L.v_b3};
}

template <typename T>
static hipError_t dev_alloc(T **p, size_t count) {
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e == hipSuccess) e = hipMemset(*p, 0, count * sizeof(T));
    return e;
}

static int check_cfg(const ddrl_sac1_config_t *c) {
    DDRL_REQUIRE(c != nullptr, "config is NULL");
    DDRL_REQUIRE(c->obs_dim > 0 && c->act_dim > 0 && c->hidden1 > 0 && c->hidden2 > 0 && c->batch > 0, "dims must be positive");
    DDRL_REQUIRE(c->act_dim <= MAXA, "act_dim > 8 unsupported");
    DDRL_REQUIRE(c->variant == DDRL_SAC1 || c->variant == DDRL_SAC_V, "variant must be DDRL_SAC1 or DDRL_SAC_V");
    DDRL_REQUIRE(c->batch < 65536, "batch >= 65536 unsupported (packed kernel arguments)");
    DDRL_REQUIRE(c->obs_dim + c->act_dim <= L1_MAXD, "obs_dim + act_dim > 40 unsupported by the layer-1 kernel");
    DDRL_REQUIRE(c->hidden1 <= 64 * RV && c->hidden2 <= 64 * RV, "hidden sizes > 512 unsupported by the row kernels");
    return DDRL_OK;
}

// Envelope of the direct-operand path (sac1_direct.h); everything else takes the generic kernels.
static bool direct_ok(const ddrl_sac1_config_t &c) {
    return c.hidden1 % 4 == 0 && c.hidden2 % 4 == 0 && c.hidden1 <= 512 && c.hidden2 <= 32 * DNT &&
           c.obs_dim + c.act_dim <= 12 && 2 * c.act_dim <= DFH && c.act_dim <= 4 && c.batch <= 32768 &&
           getenv("DDRL_SAC1_GENERIC") == nullptr;
}

}  // namespace

// ==========================================================================================
struct ddrl_sac1 {
    int device;
    ddrl_sac1_config_t cfg;
    Layout L;
    float *slab;
    float *main_p, *target_p, *m, *v, *grad;
    // two sets of input buffers: the sampler may fill set s^1 while an update reads set s
    // (order inside a set: obs1 obs2 acts rews done eps_x eps_x2 eps_t)
    float *in[2][8];
    float *H1, *H2, *dZ2, *dZ1, *xa, *xp, *part;
    float *act0, *act2, *logp0, *logp1, *save0, *q1o, *q2o, *dq4, *dhead, *loss_part, *losses;
    int ldh1, ldh2, ldxa, ldxp, ldd;
    OptState *opt;       // two copies; opt + opt_cur is current, every optimizer step advances into the other one
    int opt_cur;
    Seg *segs_d;
    L1Jobs l1a[2];
    GemmJobs g_fa, g_fb, g_bq, g_bpi, g_last;
    RowsA ra[2];
    RowsB rb[2];
    RowsC rc;
    RowsAV rav[2];       // SAC-v
    RowsBV rbv[2];
    float *vo, *vto;
    AdamArgs ad;
    int rows_b_blocks;
    bool fused;          // direct-operand path (sac1_direct.h) instead of the generic kernels
    DFHead fh_a[2], fh_b[2];
    DFArgs f_a[2], f_b[2];
    DGJobs dg_bq[2], dg_mid, dg_pi;
    int bq_cols;         // column tiles of the q2(x, a) dgrad that run in launch "bq" (the rest: bq_rest, a job of launch "mid")
    DGJob bq_rest;
    // direct-path activations (x4 images, see sac1_direct.h) and the dgrad images of the main layer-2 kernels
    int Lp1, Lp2;
    float *H1r4, *H2c4, *H2r4, *dZ1r4, *dzpi_c4, *dzpi_r4, *dhead_r4, *xa_r4, *da_part, *dq, *w3snap, *xp_r4;
    float *c4_pi[2], *c4_q[3];   // c4_q[2]: V (SAC-v)
    float *c4_q2b;               // second copy of c4_q[1]: double-buffered like the policy's when part of the q2(x, a) dgrad runs in launch "mid",
                                 // where the optimizer epilogue of q2's layer-2 wgrad writes the NEXT image (copy sh_cur = current, sh_cur ^ 1 = next)
    int mid_rest_job;            // index of that dgrad's job in dg_mid (-1: none)
    float *xv_r4;                // SAC-v: the [x | 1] image of V's layer-1 wgrad
    int *part_cnt;       // arrival counters of the policy layer-1 partials (one per column tile)
    int sh_cur;          // which copy of the policy dgrad image is current (the optimizer epilogue writes the other one)
    bool fuse_apply;     // this launch_grads also applies the optimizer (Adam in the wgrad epilogues)
    // sampler riding in k_fwd<1> (ddrl_sac1_step_and_sample)
    bool sample_armed;
    ddrl_replay_dev::RingState *smp_rs;
    ddrl_replay_dev::RingPtrs smp_ring;
    int smp_set;
    float *hp;           // head partials [NEVAL][DFH][B][DNT]
    bool fused_l1_wgrad;  // pi layer-1 wgrad via dgrad-epilogue partials + Adam (needs hidden1 % 4 == 0)
    // set by ddrl_sac1_fill_noise, consumed by the next compute_grads / apply_grads
    bool noise_armed;
    uint32_t noise_seed;
    unsigned int noise_pending;
    bool grad_imported;  // the gradient buffer was overwritten by import(GRAD): Adam must not re-sum partials
    // ddrl_sac1_capture_begin / _abort: the host-side launch state above as it stood before a caller's stream capture (every
    // launch function toggles some of it at LAUNCH time, so an aborted capture — whose launches never ran — must put it back)
    // ddrl_sac1_step_host: the host-batch update as captured graphs — on this surface the ~8 HIP calls of an eager update (two copies, the
    // noise fill, five launches) cost more host time than the device needs for the update.  One graph per (host block, copy parity).
    struct HostGraph { const float *block; float *losses; uint32_t seed; int opt_cur, sh_cur; hipGraphExec_t exec; };
    std::vector<HostGraph> host_graphs;
    uint32_t *host_ctr_d;          // device copy of the noise counter a host block carries up (two words)
    struct HostSnap { bool valid; int opt_cur, sh_cur; bool fuse_apply, sample_armed, noise_armed, grad_imported; uint32_t noise_seed; unsigned int noise_pending; } snap;
};

static void refresh_shadows(ddrl_sac1 *h, hipStream_t s);
int ddrl_internal_normal_fill_ctr(float *out_d, int64_t n, uint32_t seed, const uint32_t *ctr_d, uint64_t base, void *stream);   // common.hip
int ddrl_internal_host_block_up(const float *src, float *dst_d, int64_t n, float *e0, float *e1, float *e2, int64_t m, uint32_t seed,
                                const uint32_t *ctr, void *stream);                                                              // common.hip

static int sac1_free(ddrl_sac1 *h) {
    for (auto &g : h->host_graphs) (void)hipGraphExecDestroy(g.exec);
    if (h->host_ctr_d) (void)hipFree(h->host_ctr_d);
    (void)hipFree(h->slab);
    delete h;
    return DDRL_OK;
}

static int reset_opt(ddrl_sac1 *h, hipStream_t s) {
    OptState o{};
    o.b1p_pi = o.b1p_q = (float)h->cfg.beta1;
    o.b2p_pi = o.b2p_q = (float)h->cfg.beta2;
    h->opt_cur = 0;
    DDRL_HIP_CHECK(hipMemcpyAsync(h->opt, &o, sizeof(o), hipMemcpyHostToDevice, s));
    DDRL_HIP_CHECK(hipStreamSynchronize(s));
    return DDRL_OK;
}

// Job tables of the SAC-v update (example/model.py:17-76) on the generic kernels.
static int build_sacv(ddrl_sac1 *h) {
    const ddrl_sac1_config_t *cfg = &h->cfg;
    const Layout &L = h->L;
    const int B = cfg->batch, o = cfg->obs_dim, a = cfg->act_dim, h1 = cfg->hidden1, h2 = cfg->hidden2;
    const float *Pm = h->main_p, *Pt = h->target_p;
    const int ldh1 = h->ldh1, ldh2 = h->ldh2;
    const long long BH1 = (long long)B * ldh1, BH2 = (long long)B * ldh2, BZ1 = (long long)B * h1, BZ2 = (long long)B * h2;
    h->fused = false;
    // evaluations: 0 pi(x) 1 q1(x,a) 2 q2(x,a) | 3 q1(x,pi) 4 q2(x,pi) (observation part here, finished by k_rows_a_v) | 5 v(x) 6 v_targ(x2)
    for (int st = 0; st < 2; ++st) {
        float *x = h->in[st][0], *x2 = h->in[st][1], *ac = h->in[st][2];
        auto l1 = [&](const float *in0, int d0, const float *in1, int d1, const float *W, const float *b, int ev, int pre) {
            return L1Job{in0, in1, W, b, h->H1 + ev * BH1, nullptr, d0, d1, B, h1, ldh1, 0, pre};
        };
        L1Jobs &J = h->l1a[st];
        J.njobs = 7;
        J.noise_on = 0; J.act = a; J.n_each = B * a; J.noise_seed = 0;
        J.e0 = h->in[st][5]; J.e1 = h->in[st][6]; J.e2 = h->in[st][7]; J.opt = h->opt;
        J.job[0] = l1(x, o, nullptr, 0, Pm + L.pi_W1, Pm + L.pi_b1, 0, 0);
        J.job[1] = l1(x, o, ac, a, Pm + L.q_W1[0], Pm + L.q_b1[0], 1, 0);
        J.job[2] = l1(x, o, ac, a, Pm + L.q_W1[1], Pm + L.q_b1[1], 2, 0);
        J.job[3] = l1(x, o, nullptr, a, Pm + L.q_W1[0], Pm + L.q_b1[0], 3, 1);
        J.job[4] = l1(x, o, nullptr, a, Pm + L.q_W1[1], Pm + L.q_b1[1], 4, 1);
        J.job[5] = l1(x, o, nullptr, 0, Pm + L.v_W1, Pm + L.v_b1, 5, 0);
        J.job[6] = l1(x2, o, nullptr, 0, Pt + L.v_W1, Pt + L.v_b1, 6, 0);
        J.job[0].aug_out = h->xp; J.job[0].aug_ld = h->ldxp;  // [x | 1]     : pi and v layer-1 wgrads
        J.job[1].aug_out = h->xa; J.job[1].aug_ld = h->ldxa;  // [x | a | 1] : Q layer-1 wgrads
        h->rav[st] = RowsAV{h->H2, net_pi(Pm, L), net_q(Pm, L, 0), net_q(Pm, L, 1), net_v(Pm, L), net_v(Pt, L), h->in[st][5],
                            h->act0, h->logp0, h->save0, h->q1o, h->q2o, h->vo, h->vto, h->H1,
                            Pm + L.q_W1[0] + (long long)o * h1, Pm + L.q_W1[1] + (long long)o * h1, B, h2, ldh2, a, h1, ldh1,
                            (float)cfg->act_scale};
        h->rbv[st] = RowsBV{h->H2, net_q(Pm, L, 0), net_q(Pm, L, 1), net_v(Pm, L), h->in[st][3], h->in[st][4], h->logp0, h->q1o, h->q2o,
                            h->vo, h->vto, h->dZ2, h->dq4, h->loss_part, B, h2, ldh2, (float)cfg->alpha, (float)cfg->gamma};
    }
    auto fwd = [&](const float *P, long long W2, long long b2, int ev) {
        return gemm_fwd(h->H1 + ev * BH1, ldh1, P + W2, P + b2, h->H2 + ev * BH2, ldh2, B, h1, h2);
    };
    gemm_add(h->g_fa, fwd(Pm, L.pi_W2, L.pi_b2, 0));
    gemm_add(h->g_fa, fwd(Pm, L.q_W2[0], L.q_b2[0], 1));
    gemm_add(h->g_fa, fwd(Pm, L.q_W2[1], L.q_b2[1], 2));
    gemm_add(h->g_fa, fwd(Pm, L.v_W2, L.v_b2, 5));
    gemm_add(h->g_fa, fwd(Pt, L.v_W2, L.v_b2, 6));
    gemm_add(h->g_fb, fwd(Pm, L.q_W2[0], L.q_b2[0], 3));
    gemm_add(h->g_fb, fwd(Pm, L.q_W2[1], L.q_b2[1], 4));
    float *G = h->grad;
    // value backward: dZ2 / dZ1 slots 0 q1(x,a) 1 q2(x,a) 2 q1(x,pi) 3 v(x)
    gemm_add(h->g_bq, gemm_dgrad(h->dZ2 + 2 * BZ2, Pm + L.q_W2[0], h->H1 + 3 * BH1, ldh1, h->dZ1 + 2 * BZ1, B, h1, h2));
    gemm_add(h->g_bq, gemm_dgrad(h->dZ2 + 0 * BZ2, Pm + L.q_W2[0], h->H1 + 1 * BH1, ldh1, h->dZ1 + 0 * BZ1, B, h1, h2));
    gemm_add(h->g_bq, gemm_dgrad(h->dZ2 + 1 * BZ2, Pm + L.q_W2[1], h->H1 + 2 * BH1, ldh1, h->dZ1 + 1 * BZ1, B, h1, h2));
    gemm_add(h->g_bq, gemm_dgrad(h->dZ2 + 3 * BZ2, Pm + L.v_W2, h->H1 + 5 * BH1, ldh1, h->dZ1 + 3 * BZ1, B, h1, h2));
    for (int q = 0; q < 2; ++q) {
        gemm_add(h->g_bq, gemm_wgrad(h->H1 + (1 + q) * BH1, ldh1, h1, h->dZ2 + q * BZ2, h2, h2, G + L.q_W2[q], h2, B));
        gemm_add(h->g_bq, gemm_wgrad(h->H2 + (1 + q) * BH2, ldh2, h2, h->dq4 + (long long)q * B * 4, 4, 1, G + L.q_W3[q], 1, B));
    }
    gemm_add(h->g_bq, gemm_wgrad(h->H1 + 5 * BH1, ldh1, h1, h->dZ2 + 3 * BZ2, h2, h2, G + L.v_W2, h2, B));
    gemm_add(h->g_bq, gemm_wgrad(h->H2 + 5 * BH2, ldh2, h2, h->dq4 + (long long)2 * B * 4, 4, 1, G + L.v_W3, 1, B));
    // policy backward (dZ2 / dZ1 slot 4) + the layer-1 wgrads of the value networks
    {
        GemmJob j = gemm_dgrad(h->dZ2 + 4 * BZ2, Pm + L.pi_W2, h->H1 + 0 * BH1, ldh1, h->dZ1 + 4 * BZ1, B, h1, h2);
        h->fused_l1_wgrad = (h1 % 4 == 0) && (o + 1 <= 12) && (L.pi_W1 % 4 == 0);
        if (h->fused_l1_wgrad) { j.part_x = h->xp; j.part = h->part; j.part_nk = o + 1; j.part_ldx = h->ldxp; }
        gemm_add(h->g_bpi, j);
    }
    gemm_add(h->g_bpi, gemm_wgrad(h->H1 + 0 * BH1, ldh1, h1, h->dZ2 + 4 * BZ2, h2, h2, G + L.pi_W2, h2, B));
    gemm_add(h->g_bpi, gemm_wgrad(h->H2, ldh2, h2, h->dhead, h->ldd, a, G + L.pi_Wmu, a, B));
    gemm_add(h->g_bpi, gemm_wgrad(h->H2, ldh2, h2, h->dhead + a, h->ldd, a, G + L.pi_Wls, a, B));
    for (int q = 0; q < 2; ++q)
        gemm_add(h->g_bpi, gemm_wgrad(h->xa, h->ldxa, o + a, h->dZ1 + q * BZ1, h1, h1, G + L.q_W1[q], h1, B));
    gemm_add(h->g_bpi, gemm_wgrad(h->xp, h->ldxp, o, h->dZ1 + 3 * BZ1, h1, h1, G + L.v_W1, h1, B));
    if (!h->fused_l1_wgrad) gemm_add(h->g_last, gemm_wgrad(h->xp, h->ldxp, o, h->dZ1 + 4 * BZ1, h1, h1, G + L.pi_W1, h1, B));
    h->rc = RowsC{h->H2, h->dZ1 + 2 * BZ1, Pm + L.q_W1[0], net_pi(Pm, L), h->save0, h->dhead, h->dZ2 + 4 * BZ2,
                  h->loss_part, h->losses, B, h1, h2, ldh2, o, a, h->ldd, h->rows_b_blocks, (float)cfg->alpha,
                  (float)cfg->act_scale, 4};
    h->ad = AdamArgs{h->main_p, h->target_p, h->m, h->v, h->grad, h->opt, h->opt + 1, L.total_int, L.n_pi_int, 0,
                     (float)cfg->lr, (float)cfg->beta1, (float)cfg->beta2, (float)cfg->adam_eps,
                     (float)cfg->polyak, (float)(1.0 - cfg->polyak),
                     h->part, L.pi_W1 / 4, (long long)(o + 1) * h1 / 4, (long long)(o + 1) * h1 / 4,
                     h->fused_l1_wgrad ? (B + 31) / 32 : 0, 0u};
    h->noise_armed = false; h->noise_seed = 0; h->noise_pending = 0; h->grad_imported = false;
    h->fuse_apply = false; h->sample_armed = false;
    return DDRL_OK;
}

// Job tables of the SAC-v update on the direct-operand kernels (sac1_direct.h).  B = rows of every image (whole 32-row
// tiles), Bv = the batch.  Evaluations (head-partial slots): 0 pi(x) 1 q1(x,a) 2 q2(x,a) 3 v(x) 4 v_targ(x2) |
// phase 1: 5 q1(x,pi) 6 q2(x,pi).  Image slots — H1r4 / H2c4: 0 pi 1 q1(x,a) 2 q2(x,a) 3 q1(x,pi) 4 v;  H2r4: 0 pi 1 q1 2 q2 3 v;
// dZ1r4: 0 q1 1 q2 2 v.  Network ids of the forward pack field: 0 policy, 1 q1, 2 q2, 3 v (the value networks follow the
// policy at equal distances in a parameter buffer).
static int build_sacv_direct(ddrl_sac1 *h, int Bv, int B) {
    const ddrl_sac1_config_t *cfg = &h->cfg;
    const Layout &L = h->L;
    const int o = cfg->obs_dim, a = cfg->act_dim, h1 = cfg->hidden1, h2 = cfg->hidden2;
    const int Kp1 = L.Kp1, Np2 = L.Np2, nt2 = (h2 + 31) / 32;
    const float *Pm = h->main_p, *Pt = h->target_p, *S = h->slab;
    DDRL_REQUIRE(L.v_W1 - L.q_W1[1] == L.q_W1[1] - L.q_W1[0], "internal: the value networks must sit at equal distances");
    h->fused_l1_wgrad = false;   // (direct path: the policy's layer-1 wgrad is a job of its own, no row-tile partials)
    h->sh_cur = 0;
    const long long HP = (long long)DFH * B * DNT;
    const long long H1I = (long long)B * h->Lp1, H2C = (long long)Np2 * B, H2R = (long long)B * h->Lp2;
    auto steps = [](int D) { return D + 1 <= 8 ? 4 : 4 + (D + 1 - 8 + 1) / 2; };
    const long long vW1[3] = {L.q_W1[0], L.q_W1[1], L.v_W1}, vW2[3] = {L.q_W2[0], L.q_W2[1], L.v_W2}, vb2[3] = {L.q_b2[0], L.q_b2[1], L.v_b2};
    const long long vW3[3] = {L.q_W3[0], L.q_W3[1], L.v_W3};
    for (int st = 0; st < 2; ++st) {
        auto head = [&](DFHead &d) {
            d = DFHead{};
            d.base = S; d.tiles_m = B / 32; d.tpj = (B / 32) * nt2; d.K = h1; d.Np = Np2; d.B = B; d.d0 = o;
            d.x_off = (int)(h->in[st][0] - S);
            d.main_off = (int)(Pm - S); d.targ_off = (int)(Pt - S); d.npi = (int)L.q_W1[0]; d.perq = (int)(L.q_W1[1] - L.q_W1[0]);
            d.hp_off = (int)(h->hp - S);
        };
        auto args = [&](DFArgs &F, int njobs) {
            F = DFArgs{};
            F.njobs = njobs; F.tiles_n = nt2; F.act = a; F.Lp1 = h->Lp1; F.Lp2 = h->Lp2; F.h2 = h2;
            F.scale = (float)cfg->act_scale;
            F.act0 = h->act0; F.act2 = h->act2; F.logp0 = h->logp0; F.logp1 = h->logp1; F.save0 = h->save0;
            F.php1 = nullptr; F.pbmu1 = Pm + L.pi_bmu; F.pbls1 = Pm + L.pi_bls; F.peps1 = h->in[st][6];
            F.pev_pack = 0;   // both phase-1 jobs take the action sampled by pi(x)
            F.noise_on = 0; F.n_each = Bv * a; F.Bv = Bv; F.noise_seed = 0;
            F.e0 = h->in[st][5]; F.e1 = h->in[st][6]; F.e2 = h->in[st][7]; F.opt = h->opt;
        };
        auto vj = [&](const float *P, int q, int ev) {   // a value network (q = 0, 1: Q; 2: V)
            DFJob j{};
            j.b2 = P + vb2[q]; j.wh0 = P + vW3[q]; j.wh1 = j.wh0; j.nh = 1; j.hsplit = 1; j.hstride = 1;
            j.hp = h->hp + ev * HP;
            return j;
        };
        const int ns_pi = steps(o) - 4, ns_q = steps(o + a) - 4;
        DFHead &HA = h->fh_a[st], &HB = h->fh_b[st];
        DFArgs &FA = h->f_a[st], &FB = h->f_b[st];
        head(HA); args(FA, 5);
        {
            DFJob j{};
            j.b2 = Pm + L.pi_b2; j.wh0 = Pm + L.pi_Wmu; j.wh1 = Pm + L.pi_Wls; j.nh = 2 * a; j.hsplit = a; j.hstride = a; j.hp = h->hp;
            j.H2c4 = h->H2c4; j.H2r4 = h->H2r4; j.H1r4 = h->H1r4; j.xr4 = h->xp_r4;
            FA.job[0] = j;
        }
        FA.job[1] = vj(Pm, 0, 1); FA.job[1].H2c4 = h->H2c4 + 1 * H2C; FA.job[1].H2r4 = h->H2r4 + 1 * H2R; FA.job[1].H1r4 = h->H1r4 + 1 * H1I;
        FA.job[1].xr4 = h->xa_r4;
        FA.job[2] = vj(Pm, 1, 2); FA.job[2].H2c4 = h->H2c4 + 2 * H2C; FA.job[2].H2r4 = h->H2r4 + 2 * H2R; FA.job[2].H1r4 = h->H1r4 + 2 * H1I;
        FA.job[3] = vj(Pm, 2, 3); FA.job[3].H2c4 = h->H2c4 + 4 * H2C; FA.job[3].H2r4 = h->H2r4 + 3 * H2R; FA.job[3].H1r4 = h->H1r4 + 4 * H1I;
        FA.job[3].xr4 = h->xv_r4;
        FA.job[4] = vj(Pt, 2, 4);
        head(HB); args(FB, 2);
        {
            auto pk = [](int ns, int x2, int targ, int net) { return ns | (x2 << 2) | (targ << 3) | (net << 4); };
            HA.pack = pk(ns_pi, 0, 0, 0) | (pk(ns_q, 0, 0, 1) << 6) | (pk(ns_q, 0, 0, 2) << 12) | (pk(ns_pi, 0, 0, 3) << 18) | (pk(ns_pi, 1, 1, 3) << 24);
            HB.pack = pk(ns_q, 0, 0, 1) | (pk(ns_q, 0, 0, 2) << 6);
        }
        auto from_pi = [&](DFJob &j, int side) {
            j.php = h->hp; j.pbmu = Pm + L.pi_bmu; j.pbls = Pm + L.pi_bls; j.peps = h->in[st][5]; j.side = side;
        };
        FB.job[0] = vj(Pm, 0, 5); from_pi(FB.job[0], 1);
        FB.job[0].H2c4 = h->H2c4 + 3 * H2C; FB.job[0].H1r4 = h->H1r4 + 3 * H1I;
        FB.job[1] = vj(Pm, 1, 6); from_pi(FB.job[1], 0);
        // ---- backward launch 1: the four value dgrads (slot 2 first: its dQ/da partials are what the next launch waits for)
        DGJobs &Q = h->dg_bq[st];
        Q = DGJobs{};
        Q.hp = h->hp; Q.B = B; Q.Bv = Bv; Q.sacv = 1; Q.q_ev0 = 1; Q.q_nev = 6;
        Q.b3q1 = Pm + L.q_b3[0]; Q.b3q2 = Pm + L.q_b3[1]; Q.b3q1t = Pm + L.v_b3; Q.b3q2t = Pt + L.v_b3;
        Q.rew = h->in[st][3]; Q.done = h->in[st][4]; Q.logp0 = h->logp0; Q.logp1 = h->logp1;
        Q.q1o = h->q1o; Q.q2o = h->q2o; Q.vo = h->vo; Q.vto = h->vto; Q.dq = h->dq; Q.loss_part = h->loss_part;
        Q.alpha = (float)cfg->alpha; Q.gamma = (float)cfg->gamma;
        auto dq_job = [&](int slot, int img, int q, float *C) {
            DGJob j{};
            j.type = DG_DGRAD_Q; j.M = B; j.N = h1; j.K = h2; j.slot = slot;
            j.A = h->H2c4 + img * H2C; j.lda = B; j.B = h->c4_q[q]; j.ldb = Kp1;
            j.gw = Pm + vW3[q]; j.gdq = nullptr; j.gconst = -1.0f / (float)Bv;
            j.mask = h->H1r4 + img * H1I; j.ldmask = h->Lp1; j.C = C; j.ldc = h->Lp1; j.adam_off = -1;
            return j;
        };
        {
            DGJob j = dq_job(2, 3, 0, nullptr);
            j.wa = Pm + L.q_W1[0]; j.wa_d0 = o; j.da_part = h->da_part; j.nact = a;
            dg_add(Q, j);
        }
        // (the W3 snapshots for the next launch's generated wgrad operands: one per value network)
        { DGJob j = dq_job(0, 1, 0, h->dZ1r4); j.gw_snap = h->w3snap; dg_add(Q, j); }
        { DGJob j = dq_job(1, 2, 1, h->dZ1r4 + H1I); j.gw_snap = h->w3snap + 512; dg_add(Q, j); }
        { DGJob j = dq_job(3, 4, 2, h->dZ1r4 + 2 * H1I); j.gw_snap = h->w3snap + 1024; dg_add(Q, j); }
    }
    float *G = h->grad;
    const AdamCtx ctx{0, h->main_p, h->target_p, h->m, h->v, G, h->opt, nullptr, L.n_pi_int,
                      (float)cfg->lr, (float)cfg->beta1, (float)cfg->beta2, (float)cfg->adam_eps,
                      (float)cfg->polyak, (float)(1.0 - cfg->polyak), 0u};
    auto wgrad_j4 = [&](const float *A, const float *Bm, long long w_off, long long b_off, float *shadow) {
        DGJob j{};
        j.type = DG_WGRAD_J4; j.M = h1 + 1; j.N = h2; j.K = B;
        j.A = A; j.lda = h->Lp1; j.B = Bm; j.ldb = h->Lp2;
        j.adam_off = w_off; j.ldc = Np2; j.bias_off = b_off; j.bias_row = h1; j.shadow = shadow; j.ld_sh = Kp1;
        return j;
    };
    auto wgrad_rm = [&](const float *A, int lda, int M, const float *Bm, int ldb, int N, long long off) {
        DGJob j{};
        j.type = DG_WGRAD_RM; j.M = M; j.N = N; j.K = B; j.A = A; j.lda = lda; j.B = Bm; j.ldb = ldb; j.adam_off = off; j.ldc = N;
        return j;
    };
    const int img1[3] = {1, 2, 4}, img2r[3] = {1, 2, 3};   // H1r4 / H2r4 slots of q1(x,a), q2(x,a), v(x)
    {   // ---- backward launch 2: the policy dgrad with its A operand generated in the tile (see the SAC1 tables), the policy-head backward
        // tiles beside it, layer-2 + head wgrads of the three value networks
        DGJobs &M = h->dg_mid;
        M = DGJobs{};
        M.B = B; M.Bv = Bv; M.ad = ctx;
        DGJob d{};
        d.type = DG_DGRAD; d.M = B; d.N = h1; d.K = h2; d.A = h->H2c4; d.lda = B; d.B = h->c4_pi[0]; d.ldb = Kp1;
        d.bgen = 3; d.dap = h->da_part; d.nparts = (h1 + 31) / 32; d.save0 = h->save0; d.wmu = Pm + L.pi_Wmu; d.wls = Pm + L.pi_Wls;
        d.nact = a; d.alpha = (float)cfg->alpha; d.scale = (float)cfg->act_scale;
        d.mask = h->H1r4; d.ldmask = h->Lp1; d.C = h->dZ1r4 + 3 * H1I; d.ldc = h->Lp1; d.adam_off = -1;
        dg_add(M, d);
        DGJob rc{};
        rc.type = DG_ROWS_C; rc.M = B; rc.N = h2; rc.K = 0; rc.nact = a; rc.adam_off = -1;
        rc.h2c4 = h->H2c4; rc.dap = h->da_part; rc.nparts = (h1 + 31) / 32; rc.save0 = h->save0;
        rc.wmu = Pm + L.pi_Wmu; rc.wls = Pm + L.pi_Wls; rc.dz_c4 = h->dzpi_c4; rc.dz_r4 = h->dzpi_r4; rc.dhead_r4 = h->dhead_r4;
        rc.ld_r4 = h->Lp2; rc.alpha = (float)cfg->alpha; rc.scale = (float)cfg->act_scale;
        dg_add(M, rc);
        for (int q = 0; q < 3; ++q) {
            DGJob j = wgrad_j4(h->H1r4 + img1[q] * H1I, h->H2r4 + img2r[q] * H2R, vW2[q], vb2[q], h->c4_q[q]);
            j.bgen = 1; j.gw = h->w3snap + 512 * q; j.gdq = h->dq + (long long)q * B;   // (W3 as the previous launch saw it)
            dg_add(M, j);
        }
        for (int q = 0; q < 3; ++q) dg_add(M, wgrad_rm(h->H2r4 + img2r[q] * H2R, h->Lp2, h2 + 1, h->dq + (long long)q * B, 1, 1, vW3[q]));
    }
    {   // ---- backward launch 3: policy wgrads (layer 2, heads, layer 1), layer-1 wgrads of the value networks, loss means, optimizer bookkeeping
        DGJobs &P = h->dg_pi;
        P = DGJobs{};
        P.B = B; P.Bv = Bv; P.ad = ctx;
        dg_add(P, wgrad_j4(h->H1r4, h->dzpi_r4, L.pi_W2, L.pi_b2, h->c4_pi[1]));
        dg_add(P, wgrad_rm(h->H2r4, h->Lp2, h2 + 1, h->dhead_r4, 32, a, L.pi_Wmu));
        dg_add(P, wgrad_rm(h->H2r4, h->Lp2, h2 + 1, h->dhead_r4 + (long long)a * 4, 32, a, L.pi_Wls));
        for (int q = 0; q < 3; ++q) {
            DGJob j = wgrad_rm(q < 2 ? h->xa_r4 : h->xv_r4, 32, (q < 2 ? o + a : o) + 1, h->dZ1r4 + q * H1I, h->Lp1, h1, vW1[q]);
            j.type = DG_WGRAD_W1Y;
            dg_add(P, j);
        }
        {
            DGJob j = wgrad_rm(h->xp_r4, 32, o + 1, h->dZ1r4 + 3 * H1I, h->Lp1, h1, L.pi_W1);
            j.type = DG_WGRAD_W1Y;
            dg_add(P, j);
        }
        DGJob ls{};
        ls.type = DG_LOSS; ls.M = 1; ls.N = 1; ls.K = 0; ls.adam_off = -1; ls.loss_part = h->loss_part; ls.losses = h->losses; ls.nl = 4;
        ls.nparts = -1;   // + the optimizer's books
        dg_add(P, ls);
    }
    h->rc = RowsC{};
    h->rc.nl = 4;
    h->ad = AdamArgs{h->main_p, h->target_p, h->m, h->v, h->grad, h->opt, h->opt + 1, L.total_int, L.n_pi_int, 0,
                     (float)cfg->lr, (float)cfg->beta1, (float)cfg->beta2, (float)cfg->adam_eps,
                     (float)cfg->polyak, (float)(1.0 - cfg->polyak),
                     h->part, L.pi_W1 / 4, (long long)(o + 1) * h1 / 4, (long long)(o + 1) * h1 / 4, 0, 0u};
    h->noise_armed = false; h->noise_seed = 0; h->noise_pending = 0; h->grad_imported = false;
    h->fuse_apply = false; h->sample_armed = false;
    return DDRL_OK;
}

extern "C" {

int ddrl_sac1_param_counts(const ddrl_sac1_config_t *cfg, int64_t *n_pi, int64_t *n_q) {
    int rc = check_cfg(cfg);
    if (rc != DDRL_OK) return rc;
    Layout L = make_layout(*cfg, false);
    if (n_pi) *n_pi = L.n_pi;
    if (n_q) *n_q = L.n_q;
    return DDRL_OK;
}

int ddrl_sac1_create(ddrl_sac1_t **out, int device, const ddrl_sac1_config_t *cfg) {
    DDRL_REQUIRE(out != nullptr, "out is NULL");
    int rc = check_cfg(cfg);
    if (rc != DDRL_OK) return rc;
    ddrl::DeviceGuard g(device);
    if (!g.ok) { ddrl::set_error("cannot select device %d", device); return DDRL_ERR_HIP; }
    ddrl_sac1 *h = new ddrl_sac1();  // value-initialised: every pointer/job table starts zeroed
    h->mid_rest_job = -1;            // (no Q dgrad columns in launch "mid" unless the SAC1 direct build below splits them off)
    h->device = device;
    h->cfg = *cfg;
    h->fused = direct_ok(*cfg);
    h->L = make_layout(*cfg, false, h->fused);
    const Layout &L = h->L;
    // direct-operand path: every row-indexed buffer holds a whole number of 32-row tiles; rows past the batch are padding
    // (zero inputs, no loss terms, zero upstream gradients) and means run over the Bv valid rows
    const int Bv = cfg->batch, B = h->fused ? (int)rup32(Bv) : Bv;
    const int o = cfg->obs_dim, a = cfg->act_dim, h1 = cfg->hidden1, h2 = cfg->hidden2;
    const size_t NT = (size_t)L.total_int;
    const int Kp1 = L.Kp1, Np2 = L.Np2;
    h->Lp1 = rup32(h1 + 1); h->Lp2 = rup32(h2 + 1);  // activation images keep room for the ones column
    // ONE slab for every buffer of the learner (parameters, optimizer state, activations, job
    // tables): a single large allocation is mapped with large page fragments, so the ~30 buffers a
    // stage touches share a handful of TLB entries instead of missing on one 4 KB page each.
    h->rows_b_blocks = B;  // per-row loss terms
    size_t slab_floats = 0;
    auto reserve = [&](size_t cnt) { size_t off = slab_floats; slab_floats += (cnt + 63) & ~(size_t)63; return off; };  // 256-B aligned
    struct Item { float **p; size_t off; };
    std::vector<Item> items;
#define ALLOC(ptr, cnt) items.push_back(Item{&h->ptr, reserve((size_t)(cnt))})
    // activations carry one extra physical column of ones (the bias-gradient row of the wgrads)
    h->ldh1 = (int)pad4(h1 + 1); h->ldh2 = (int)pad4(h2 + 1);
    h->ldxa = (int)pad4(o + a + 1); h->ldxp = (int)pad4(o + 1); h->ldd = (int)pad4(2 * a);
    ALLOC(main_p, NT); ALLOC(target_p, NT); ALLOC(m, NT); ALLOC(v, NT); ALLOC(grad, NT);
    for (int st = 0; st < 2; ++st) {
        const size_t cnt[8] = {(size_t)B * o, (size_t)B * o, (size_t)B * a, (size_t)B, (size_t)B, (size_t)B * a, (size_t)B * a, (size_t)B * a};
        for (int i = 0; i < 8; ++i) items.push_back(Item{&h->in[st][i], reserve(cnt[i])});
    }
    ALLOC(part, (size_t)((B + 31) / 32) * (o + 1) * h1);
    if (!h->fused) {
        ALLOC(H1, (size_t)NEVAL * B * h->ldh1); ALLOC(H2, (size_t)NEVAL * B * h->ldh2);
        ALLOC(dZ2, (size_t)5 * B * h2); ALLOC(dZ1, (size_t)5 * B * h1);
    } else {
        // image slots: SAC1 — H1r4 / H2c4: pi(x) q1(x,a) q2(x,a) q1(x,pi); H2r4: pi q1 q2; dZ1r4: q1 q2.  SAC-v adds V to each.
        const int xv = cfg->variant == DDRL_SAC_V ? 1 : 0;
        ALLOC(H1r4, (size_t)(4 + xv) * B * h->Lp1); ALLOC(H2c4, (size_t)(4 + xv) * Np2 * B); ALLOC(H2r4, (size_t)(3 + xv) * B * h->Lp2);
        ALLOC(dZ1r4, (size_t)(3 + xv) * B * h->Lp1) /* q1 q2 (v) + the policy's (last slot) */; ALLOC(dzpi_c4, (size_t)Np2 * B); ALLOC(dzpi_r4, (size_t)B * h->Lp2);
        ALLOC(dhead_r4, (size_t)B * 32); ALLOC(xa_r4, (size_t)B * 32); ALLOC(xv_r4, (size_t)B * 32); ALLOC(xp_r4, (size_t)B * 32); ALLOC(da_part, (size_t)16 * B * 4);
        ALLOC(dq, (size_t)3 * B + 256); ALLOC(w3snap, (size_t)3 * 512);
        for (int i = 0; i < 2; ++i) items.push_back(Item{&h->c4_pi[i], reserve((size_t)Np2 * Kp1)});
        for (int i = 0; i < 2 + xv; ++i) items.push_back(Item{&h->c4_q[i], reserve((size_t)Np2 * Kp1)});
        items.push_back(Item{&h->c4_q2b, reserve((size_t)Np2 * Kp1)});
    }
    ALLOC(xa, (size_t)B * h->ldxa); ALLOC(xp, (size_t)B * h->ldxp);
    ALLOC(act0, B * a); ALLOC(act2, B * a); ALLOC(logp0, B); ALLOC(logp1, B); ALLOC(save0, (size_t)B * a * 4);
    ALLOC(q1o, B); ALLOC(q2o, B); ALLOC(vo, B); ALLOC(vto, B); ALLOC(dq4, (size_t)3 * B * 4); ALLOC(dhead, (size_t)B * h->ldd);
    ALLOC(loss_part, (size_t)h->rows_b_blocks * 4); ALLOC(losses, 4);
    const int nt2 = (h2 + 31) / 32;
    ALLOC(hp, (size_t)NEVAL * DFH * B * DNT);
#undef ALLOC
    const size_t cnt_off = reserve(64);
    const size_t opt_off = reserve((2 * sizeof(OptState) + 3) / 4);
    const size_t segs_off = reserve((L.segs.size() * sizeof(Seg) + 3) / 4);
    (void)reserve(2048);  // readable guard behind the last buffer (unclamped tile loads, see OpPre)
    hipError_t e = hipMalloc((void **)&h->slab, slab_floats * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->slab, 0, slab_floats * sizeof(float));
    if (e != hipSuccess) {
        ddrl::set_error("hipMalloc of %zu bytes failed in ddrl_sac1_create: %s", slab_floats * sizeof(float), hipGetErrorString(e));
        sac1_free(h);
        return DDRL_ERR_NOMEM;
    }
    for (auto &it : items) *it.p = h->slab + it.off;
    h->opt = reinterpret_cast<OptState *>(h->slab + opt_off);
    h->part_cnt = reinterpret_cast<int *>(h->slab + cnt_off);
    h->segs_d = reinterpret_cast<Seg *>(h->slab + segs_off);
    DDRL_HIP_CHECK(hipMemcpy(h->segs_d, L.segs.data(), L.segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
    rc = reset_opt(h, nullptr);
    if (rc != DDRL_OK) { sac1_free(h); return rc; }

    // the physical ones columns (never overwritten: kernels write columns < h1 / h2 / obs(+act) only)
    if (!h->fused) {
        k_fill_col<<<(NEVAL * B + 255) / 256, 256>>>(h->H1, (long long)NEVAL * B, h->ldh1, h1, 1.0f);
        k_fill_col<<<(NEVAL * B + 255) / 256, 256>>>(h->H2, (long long)NEVAL * B, h->ldh2, h2, 1.0f);
    } else {  // x4 images [row/4][ld][4]: "column" c of every row = four consecutive floats per row group
        const int xv = cfg->variant == DDRL_SAC_V ? 1 : 0;
        k_fill_col4<<<((4 + xv) * B / 4 + 255) / 256, 256>>>(h->H1r4, (long long)(4 + xv) * B / 4, h->Lp1, h1);
        k_fill_col4<<<((3 + xv) * B / 4 + 255) / 256, 256>>>(h->H2r4, (long long)(3 + xv) * B / 4, h->Lp2, h2);
        k_fill_col4<<<(B / 4 + 255) / 256, 256>>>(h->xa_r4, (long long)B / 4, 32, o + a);
        k_fill_col4<<<(B / 4 + 255) / 256, 256>>>(h->xv_r4, (long long)B / 4, 32, o);
        k_fill_col4<<<(B / 4 + 255) / 256, 256>>>(h->xp_r4, (long long)B / 4, 32, o);
    }
    k_fill_col<<<(B + 255) / 256, 256>>>(h->xa, B, h->ldxa, o + a, 1.0f);
    k_fill_col<<<(B + 255) / 256, 256>>>(h->xp, B, h->ldxp, o, 1.0f);
    DDRL_LAUNCH_CHECK();
    DDRL_HIP_CHECK(hipDeviceSynchronize());

    if (cfg->variant == DDRL_SAC_V) {
        rc = h->fused ? build_sacv_direct(h, Bv, B) : build_sacv(h);
        if (rc != DDRL_OK) { sac1_free(h); return rc; }
        *out = h;
        return DDRL_OK;
    }
    const float *Pm = h->main_p, *Pt = h->target_p;
    const int ldh1 = h->ldh1, ldh2 = h->ldh2;
    const long long BH1 = (long long)B * ldh1, BH2 = (long long)B * ldh2, BZ1 = (long long)B * h1, BZ2 = (long long)B * h2;
    if (!h->fused) {
    // ---- layer-1 jobs (one table per input set).  evals: 0 pi(x) 1 pi(x2) 2 piT(x2) 3 q1(x,a) 4 q2(x,a) |
    // 5 q1(x,pi) 6 q1T(x2,piT) 7 q2T(x2,piT): observation part only here, finished by k_rows_a
    for (int st = 0; st < 2; ++st) {
        float *x = h->in[st][0], *x2 = h->in[st][1], *ac = h->in[st][2];
        auto l1 = [&](const float *in0, int d0, const float *in1, int d1, const float *W, const float *b, int ev, int pre) {
            return L1Job{in0, in1, W, b, h->H1 + ev * BH1, nullptr, d0, d1, B, h1, ldh1, 0, pre};
        };
        L1Jobs &J = h->l1a[st];
        J.njobs = 8;
        J.noise_on = 0; J.act = a; J.n_each = B * a; J.noise_seed = 0;
        J.e0 = h->in[st][5]; J.e1 = h->in[st][6]; J.e2 = h->in[st][7]; J.opt = h->opt;
        J.job[0] = l1(x, o, nullptr, 0, Pm + L.pi_W1, Pm + L.pi_b1, 0, 0);
        J.job[1] = l1(x2, o, nullptr, 0, Pm + L.pi_W1, Pm + L.pi_b1, 1, 0);
        J.job[2] = l1(x2, o, nullptr, 0, Pt + L.pi_W1, Pt + L.pi_b1, 2, 0);
        J.job[3] = l1(x, o, ac, a, Pm + L.q_W1[0], Pm + L.q_b1[0], 3, 0);
        J.job[4] = l1(x, o, ac, a, Pm + L.q_W1[1], Pm + L.q_b1[1], 4, 0);
        J.job[5] = l1(x, o, nullptr, a, Pm + L.q_W1[0], Pm + L.q_b1[0], 5, 1);
        J.job[6] = l1(x2, o, nullptr, a, Pt + L.q_W1[0], Pt + L.q_b1[0], 6, 1);
        J.job[7] = l1(x2, o, nullptr, a, Pt + L.q_W1[1], Pt + L.q_b1[1], 7, 1);
        J.job[0].aug_out = h->xp; J.job[0].aug_ld = h->ldxp;  // [x | 1]      : A operand of the pi layer-1 wgrad
        J.job[3].aug_out = h->xa; J.job[3].aug_ld = h->ldxa;  // [x | a | 1]  : A operand of the Q layer-1 wgrads
    }
    // ---- forward layer-2 GEMMs
    auto fwd = [&](const float *P, long long W2, long long b2, int ev) {
        return gemm_fwd(h->H1 + ev * BH1, ldh1, P + W2, P + b2, h->H2 + ev * BH2, ldh2, B, h1, h2);
    };
    gemm_add(h->g_fa, fwd(Pm, L.pi_W2, L.pi_b2, 0));
    gemm_add(h->g_fa, fwd(Pm, L.pi_W2, L.pi_b2, 1));
    gemm_add(h->g_fa, fwd(Pt, L.pi_W2, L.pi_b2, 2));
    gemm_add(h->g_fa, fwd(Pm, L.q_W2[0], L.q_b2[0], 3));
    gemm_add(h->g_fa, fwd(Pm, L.q_W2[1], L.q_b2[1], 4));
    gemm_add(h->g_fb, fwd(Pm, L.q_W2[0], L.q_b2[0], 5));
    gemm_add(h->g_fb, fwd(Pt, L.q_W2[0], L.q_b2[0], 6));
    gemm_add(h->g_fb, fwd(Pt, L.q_W2[1], L.q_b2[1], 7));
    // ---- backward GEMM launches.  dZ2 / dZ1 slots: 0 q1(x,a), 1 q2(x,a), 2 q1(x,pi), 3 pi.
    // Every bias gradient rides as the "ones column" row of its kernel's wgrad.
    float *G = h->grad;
    // launch "bwd Q": needs dZ2[0..2], dq4
    gemm_add(h->g_bq, gemm_dgrad(h->dZ2 + 2 * BZ2, Pm + L.q_W2[0], h->H1 + 5 * BH1, ldh1, h->dZ1 + 2 * BZ1, B, h1, h2));
    gemm_add(h->g_bq, gemm_dgrad(h->dZ2 + 0 * BZ2, Pm + L.q_W2[0], h->H1 + 3 * BH1, ldh1, h->dZ1 + 0 * BZ1, B, h1, h2));
    gemm_add(h->g_bq, gemm_dgrad(h->dZ2 + 1 * BZ2, Pm + L.q_W2[1], h->H1 + 4 * BH1, ldh1, h->dZ1 + 1 * BZ1, B, h1, h2));
    for (int q = 0; q < 2; ++q) {
        gemm_add(h->g_bq, gemm_wgrad(h->H1 + (3 + q) * BH1, ldh1, h1, h->dZ2 + q * BZ2, h2, h2, G + L.q_W2[q], h2, B));       // W2, b2
        gemm_add(h->g_bq, gemm_wgrad(h->H2 + (3 + q) * BH2, ldh2, h2, h->dq4 + (long long)q * B * 4, 4, 1, G + L.q_W3[q], 1, B));  // W3, b3
    }
    // launch "bwd pi": needs dZ2[3], dhead (k_rows_c) and dZ1[0..1] (launch above)
    {
        GemmJob j = gemm_dgrad(h->dZ2 + 3 * BZ2, Pm + L.pi_W2, h->H1 + 0 * BH1, ldh1, h->dZ1 + 3 * BZ1, B, h1, h2);
        h->fused_l1_wgrad = (h1 % 4 == 0) && (o + 1 <= 12) && (L.pi_W1 % 4 == 0);
        if (h->fused_l1_wgrad) { j.part_x = h->xp; j.part = h->part; j.part_nk = o + 1; j.part_ldx = h->ldxp; }
        gemm_add(h->g_bpi, j);
    }
    gemm_add(h->g_bpi, gemm_wgrad(h->H1 + 0 * BH1, ldh1, h1, h->dZ2 + 3 * BZ2, h2, h2, G + L.pi_W2, h2, B));     // W2, b2
    gemm_add(h->g_bpi, gemm_wgrad(h->H2, ldh2, h2, h->dhead, h->ldd, a, G + L.pi_Wmu, a, B));                      // Wmu, bmu
    gemm_add(h->g_bpi, gemm_wgrad(h->H2, ldh2, h2, h->dhead + a, h->ldd, a, G + L.pi_Wls, a, B));                  // Wls, bls
    for (int q = 0; q < 2; ++q)
        gemm_add(h->g_bpi, gemm_wgrad(h->xa, h->ldxa, o + a, h->dZ1 + q * BZ1, h1, h1, G + L.q_W1[q], h1, B));    // Q W1, b1
    // launch "last" (only when the fused form is unavailable): needs dZ1[3]
    if (!h->fused_l1_wgrad) gemm_add(h->g_last, gemm_wgrad(h->xp, h->ldxp, o, h->dZ1 + 3 * BZ1, h1, h1, G + L.pi_W1, h1, B));               // pi W1, b1

    }
    // ---- direct-operand path (sac1_direct.h)
    if (h->fused) {
        h->fused_l1_wgrad = false;   // (direct path: the policy's layer-1 wgrad is a job of its own, no row-tile partials)
        h->sh_cur = 0;
        const float *S = h->slab;
        const long long HP = (long long)DFH * B * DNT;
        const long long H1I = (long long)B * h->Lp1, H2C = (long long)Np2 * B, H2R = (long long)B * h->Lp2;
        // image slots — H1r4: 0 pi(x) 1 q1(x,a) 2 q2(x,a) 3 q1(x,pi);  H2c4: same;  H2r4: 0 pi(x) 1 q1(x,a) 2 q2(x,a)
        auto steps = [](int D) { return D + 1 <= 8 ? 4 : 4 + (D + 1 - 8 + 1) / 2; };  // input columns + the bias column, two per MFMA step
        for (int st = 0; st < 2; ++st) {
            auto head = [&](DFHead &d) {
                d = DFHead{};
                d.base = S; d.tiles_m = B / 32; d.tpj = (B / 32) * nt2; d.K = h1; d.Np = Np2; d.B = B; d.d0 = o;
                d.x_off = (int)(h->in[st][0] - S);
                d.main_off = (int)(Pm - S); d.targ_off = (int)(Pt - S); d.npi = (int)L.q_W1[0]; d.perq = (int)(L.q_W1[1] - L.q_W1[0]);
                d.hp_off = (int)(h->hp - S);
            };
            auto args = [&](DFArgs &F, int njobs) {
                F = DFArgs{};
                F.njobs = njobs; F.tiles_n = nt2; F.act = a; F.Lp1 = h->Lp1; F.Lp2 = h->Lp2; F.h2 = h2;
                F.scale = (float)cfg->act_scale;
                F.act0 = h->act0; F.act2 = h->act2; F.logp0 = h->logp0; F.logp1 = h->logp1; F.save0 = h->save0;
                F.php1 = h->hp + 1 * HP; F.pbmu1 = Pm + L.pi_bmu; F.pbls1 = Pm + L.pi_bls; F.peps1 = h->in[st][6];
                F.pev_pack = 0 | (2 << 2) | (2 << 4);   // q1(x, pi(x)) <- evaluation 0; the target Qs <- evaluation 2 (pi_targ(x2))
                F.noise_on = 0; F.n_each = Bv * a; F.Bv = Bv; F.noise_seed = 0;
                F.e0 = h->in[st][5]; F.e1 = h->in[st][6]; F.e2 = h->in[st][7]; F.opt = h->opt;
            };
            auto pij = [&](const float *P, int ev) {
                DFJob j{};
                j.b2 = P + L.pi_b2; j.wh0 = P + L.pi_Wmu; j.wh1 = P + L.pi_Wls; j.nh = 2 * a; j.hsplit = a; j.hstride = a;
                j.hp = h->hp + ev * HP;
                return j;
            };
            auto qj = [&](const float *P, int q, int ev) {
                DFJob j{};
                j.b2 = P + L.q_b2[q]; j.wh0 = P + L.q_W3[q]; j.wh1 = j.wh0; j.nh = 1; j.hsplit = 1; j.hstride = 1;
                j.hp = h->hp + ev * HP;
                return j;
            };
            const int ns_pi = steps(o) - 4, ns_q = steps(o + a) - 4;
            DFHead &HA = h->fh_a[st], &HB = h->fh_b[st];
            DFArgs &FA = h->f_a[st], &FB = h->f_b[st];
            // Q1(x, a), Q2(x, a) need nothing from the policy: they run in phase 1 beside the policy-dependent evaluations (k_dfwd:
            // "stored") when the tile counts allow the per-XCD split, else in phase 0 as before.  DDRL_QXA_PHASE=0 forces phase 0.
            const int tpj_f = (B / 32) * nt2;
            static const int qxa_phase = getenv("DDRL_QXA_PHASE") ? atoi(getenv("DDRL_QXA_PHASE")) : 1;
            const bool q_late = qxa_phase == 1 && (3 * tpj_f) % 8 == 0 && (2 * tpj_f) % 8 == 0;
            head(HA); args(FA, q_late ? 3 : 5);
            FA.job[0] = pij(Pm, 0); FA.job[0].H2c4 = h->H2c4; FA.job[0].H2r4 = h->H2r4; FA.job[0].H1r4 = h->H1r4;
            FA.job[0].xr4 = h->xp_r4;   // [x | 1] as an x4 image: A operand of the policy's layer-1 wgrad
            FA.job[1] = pij(Pm, 1);
            FA.job[2] = pij(Pt, 2);
            DFJob qa1 = qj(Pm, 0, 3); qa1.H2c4 = h->H2c4 + 1 * H2C; qa1.H2r4 = h->H2r4 + 1 * H2R; qa1.H1r4 = h->H1r4 + 1 * H1I;
            qa1.xr4 = h->xa_r4;
            DFJob qa2 = qj(Pm, 1, 4); qa2.H2c4 = h->H2c4 + 2 * H2C; qa2.H2r4 = h->H2r4 + 2 * H2R; qa2.H1r4 = h->H1r4 + 2 * H1I;
            if (!q_late) { FA.job[3] = qa1; FA.job[4] = qa2; }
            head(HB); args(FB, q_late ? 5 : 3);
            if (q_late) { FB.job[3] = qa1; FB.job[4] = qa2; }
            {   // pack field of a job: steps | obs2 << 2 | target << 3 | network << 4
                auto pk = [](int ns, int x2, int targ, int net) { return ns | (x2 << 2) | (targ << 3) | (net << 4); };
                const int qxa = (pk(ns_q, 0, 0, 1) << 18) | (pk(ns_q, 0, 0, 2) << 24);
                HA.pack = pk(ns_pi, 0, 0, 0) | (pk(ns_pi, 1, 0, 0) << 6) | (pk(ns_pi, 1, 1, 0) << 12) | (q_late ? 0 : qxa);
                HB.pack = pk(ns_q, 0, 0, 1) | (pk(ns_q, 1, 1, 1) << 6) | (pk(ns_q, 1, 1, 2) << 12) | (q_late ? (qxa | (1 << 30)) : 0);
            }
            auto from_pi = [&](DFJob &j, int pev, const float *Ppi, const float *eps, int side) {
                j.php = h->hp + pev * HP; j.pbmu = Ppi + L.pi_bmu; j.pbls = Ppi + L.pi_bls; j.peps = eps; j.side = side;
            };
            FB.job[0] = qj(Pm, 0, 5); from_pi(FB.job[0], 0, Pm, h->in[st][5], 1);
            FB.job[0].H2c4 = h->H2c4 + 3 * H2C; FB.job[0].H1r4 = h->H1r4 + 3 * H1I;
            FB.job[1] = qj(Pt, 0, 6); from_pi(FB.job[1], 2, Pt, h->in[st][7], 2);
            FB.job[2] = qj(Pt, 1, 7); from_pi(FB.job[2], 2, Pt, h->in[st][7], 0);
            // ---- backward launch 1: the three Q dgrads (slot 2 first: its dQ/da partials are what the next launch waits for)
            DGJobs &Q = h->dg_bq[st];
            Q = DGJobs{};
            Q.hp = h->hp; Q.B = B; Q.Bv = Bv; Q.sacv = 0; Q.q_ev0 = 3; Q.q_nev = 5;
            Q.b3q1 = Pm + L.q_b3[0]; Q.b3q2 = Pm + L.q_b3[1]; Q.b3q1t = Pt + L.q_b3[0]; Q.b3q2t = Pt + L.q_b3[1];
            Q.rew = h->in[st][3]; Q.done = h->in[st][4]; Q.logp0 = h->logp0; Q.logp1 = h->logp1;
            Q.q1o = h->q1o; Q.q2o = h->q2o; Q.dq = h->dq; Q.loss_part = h->loss_part;
            Q.alpha = (float)cfg->alpha; Q.gamma = (float)cfg->gamma;
            auto dq_job = [&](int slot, int img, int q, float *C) {
                DGJob j{};
                j.type = DG_DGRAD_Q; j.M = B; j.N = h1; j.K = h2; j.slot = slot;
                j.A = h->H2c4 + img * H2C; j.lda = B; j.B = h->c4_q[q]; j.ldb = Kp1;
                j.gw = Pm + L.q_W3[q]; j.gdq = nullptr; j.gconst = -1.0f / (float)Bv;
                j.mask = h->H1r4 + img * H1I; j.ldmask = h->Lp1; j.C = C; j.ldc = h->Lp1; j.adam_off = -1;
                return j;
            };
            {
                DGJob j = dq_job(2, 3, 0, nullptr);
                j.wa = Pm + L.q_W1[0]; j.wa_d0 = o; j.da_part = h->da_part; j.nact = a;
                dg_add(Q, j);
            }
            // (the W3 snapshots for the next launch's generated wgrad operands)
            { DGJob j = dq_job(0, 1, 0, h->dZ1r4); j.gw_snap = h->w3snap; dg_add(Q, j); }
            // Tile counts against the 256 CUs (profiles/r05_update_experiments.txt: a launch pays +1.2 ... +2.1 us where its tile count
            // crosses a multiple of 256 and is flat in between): the three Q dgrads are 3 x 104 = 312 tiles at config 2 — 56 over.  Only the
            // last launch reads the dZ1 of the stored-action dgrads (the Q layer-1 wgrads), so the q2(x, a) dgrad is cut by COLUMN tiles: the
            // first `bq_cols` stay here (launch = 256 tiles), the rest run in launch "mid" (which the two Q-head wgrads leave for launch
            // "pi": 464 - 20 + 56 = 500 <= 512; "pi": 190 + 20 = 210 <= 256).  A column sub-range of a dgrad is a job of its own — operand B,
            // the relu mask and the output image start n_off columns further — with the same arithmetic per tile: bit-identical results.
            {
                const int tm = B / 32, ct = (h1 + 31) / 32, T = tm * ct, ncu = 256;
                static const int want = getenv("DDRL_BQ_SPLIT") ? atoi(getenv("DDRL_BQ_SPLIT")) : 1;
                int keep = ct;   // column tiles of the q2(x, a) dgrad that stay in this launch
                const int rm_tiles = 2 * ((h2 + 1 + 31) / 32);                                  // the two Q-head wgrads (one column tile each)
                const int mid_now = T + tm * ((h2 + 31) / 32) + 2 * ((h1 + 1 + 31) / 32) * ((h2 + 31) / 32) + rm_tiles;
                const int pi_now = ((h1 + 1 + 31) / 32) * ((h2 + 31) / 32) + rm_tiles + 3 * ct + 1;
                if (want && 3 * T > ncu && 2 * T < ncu) {
                    const int k = (ncu - 2 * T) / tm, moved = (ct - k) * tm;
                    if (k >= 1 && k < ct && mid_now - rm_tiles + moved <= 2 * ncu && pi_now + rm_tiles <= ncu) keep = k;
                }
                h->bq_cols = keep;
                DGJob j = dq_job(1, 2, 1, h->dZ1r4 + H1I);
                j.gw_snap = h->w3snap + 512;
                if (keep < ct) j.N = keep * 32;
                dg_add(Q, j);
                if (keep < ct && st == 0) {   // the rest of the columns: a job of launch "mid" (rew / done of the input set patched at launch)
                    const long long noff = (long long)keep * 32 * 4;
                    DGJob r = dq_job(1, 2, 1, h->dZ1r4 + H1I + noff);
                    r.N = h1 - keep * 32;
                    r.B = h->c4_q[1] + noff;
                    r.mask = h->H1r4 + 2 * H1I + noff;
                    h->bq_rest = r;
                }
            }
        }
        float *G = h->grad;
        const AdamCtx ctx{0, h->main_p, h->target_p, h->m, h->v, G, h->opt, nullptr, L.n_pi_int,
                          (float)cfg->lr, (float)cfg->beta1, (float)cfg->beta2, (float)cfg->adam_eps,
                          (float)cfg->polyak, (float)(1.0 - cfg->polyak), 0u};
        auto wgrad_j4 = [&](const float *A, const float *Bm, long long w_off, long long b_off, float *shadow) {
            DGJob j{};
            j.type = DG_WGRAD_J4; j.M = h1 + 1; j.N = h2; j.K = B;
            j.A = A; j.lda = h->Lp1; j.B = Bm; j.ldb = h->Lp2;
            j.adam_off = w_off; j.ldc = Np2; j.bias_off = b_off; j.bias_row = h1; j.shadow = shadow; j.ld_sh = Kp1;
            return j;
        };
        auto wgrad_rm = [&](const float *A, int lda, int M, const float *Bm, int ldb, int N, long long off) {
            DGJob j{};
            j.type = DG_WGRAD_RM; j.M = M; j.N = N; j.K = B; j.A = A; j.lda = lda; j.B = Bm; j.ldb = ldb; j.adam_off = off; j.ldc = N;
            return j;
        };
        {   // ---- backward launch 2: the policy dgrad — its A operand (dZ2 of the policy trunk) generated in the tile from the dQ/da
            // partials (k_dg, bgen = 3), so it does not wait for the policy-head backward tiles, which run beside it and write the
            // images the policy wgrads of launch 3 contract over —, the Q layer-2 + head wgrads (optimizer in the epilogue)
            DGJobs &M = h->dg_mid;
            M = DGJobs{};
            M.B = B; M.Bv = Bv; M.ad = ctx;
            DGJob d{};
            d.type = DG_DGRAD; d.M = B; d.N = h1; d.K = h2; d.A = h->H2c4; d.lda = B; d.B = h->c4_pi[0]; d.ldb = Kp1;   // (B: launch_stage picks the copy)
            d.bgen = 3; d.dap = h->da_part; d.nparts = (h1 + 31) / 32; d.save0 = h->save0; d.wmu = Pm + L.pi_Wmu; d.wls = Pm + L.pi_Wls;
            d.nact = a; d.alpha = (float)cfg->alpha; d.scale = (float)cfg->act_scale;
            d.mask = h->H1r4; d.ldmask = h->Lp1; d.C = h->dZ1r4 + 2 * H1I; d.ldc = h->Lp1; d.adam_off = -1;
            dg_add(M, d);
            DGJob rc{};
            rc.type = DG_ROWS_C; rc.M = B; rc.N = h2; rc.K = 0; rc.nact = a; rc.adam_off = -1;
            rc.h2c4 = h->H2c4; rc.dap = h->da_part; rc.nparts = (h1 + 31) / 32; rc.save0 = h->save0;
            rc.wmu = Pm + L.pi_Wmu; rc.wls = Pm + L.pi_Wls; rc.dz_c4 = h->dzpi_c4; rc.dz_r4 = h->dzpi_r4; rc.dhead_r4 = h->dhead_r4;
            rc.ld_r4 = h->Lp2; rc.alpha = (float)cfg->alpha; rc.scale = (float)cfg->act_scale;
            dg_add(M, rc);
            for (int q = 0; q < 2; ++q) {
                DGJob j = wgrad_j4(h->H1r4 + (1 + q) * H1I, h->H2r4 + (1 + q) * H2R, L.q_W2[q], L.q_b2[q], h->c4_q[q]);
                j.bgen = 1; j.gw = h->w3snap + 512 * q; j.gdq = h->dq + (long long)q * B;   // (W3 as the previous launch saw it)
                dg_add(M, j);
            }
            const bool split = h->bq_cols < (h1 + 31) / 32;
            h->mid_rest_job = -1;
            if (!split) {
                for (int q = 0; q < 2; ++q) dg_add(M, wgrad_rm(h->H2r4 + (1 + q) * H2R, h->Lp2, h2 + 1, h->dq + (long long)q * B, 1, 1, L.q_W3[q]));
            } else {
                // the remaining column tiles of the q2(x, a) dgrad (see launch 1): its prologue needs the launch header of the Q dgrads
                const DGJobs &Q0 = h->dg_bq[0];
                M.hp = Q0.hp; M.sacv = 0; M.q_ev0 = Q0.q_ev0; M.q_nev = Q0.q_nev;
                M.b3q1 = Q0.b3q1; M.b3q2 = Q0.b3q2; M.b3q1t = Q0.b3q1t; M.b3q2t = Q0.b3q2t;
                M.rew = Q0.rew; M.done = Q0.done; M.logp0 = Q0.logp0; M.logp1 = Q0.logp1;   // (rew / done: launch_stage sets the input set's)
                M.q1o = Q0.q1o; M.q2o = Q0.q2o; M.dq = Q0.dq; M.loss_part = Q0.loss_part;
                M.alpha = Q0.alpha; M.gamma = Q0.gamma;
                h->mid_rest_job = M.njobs;
                dg_add(M, h->bq_rest);
            }
        }
        {   // ---- backward launch 3: the wgrads that need launch 2's outputs — policy layer 2 / heads / layer 1, Q layer 1 —, loss means,
            // optimizer bookkeeping.  No hand-off inside the launch any more: every tile is a plain GEMM tile with its Adam epilogue.
            DGJobs &P = h->dg_pi;
            P = DGJobs{};
            P.B = B; P.Bv = Bv; P.ad = ctx;
            dg_add(P, wgrad_j4(h->H1r4, h->dzpi_r4, L.pi_W2, L.pi_b2, h->c4_pi[1]));   // job 0: launch_stage picks the shadow copy
            dg_add(P, wgrad_rm(h->H2r4, h->Lp2, h2 + 1, h->dhead_r4, 32, a, L.pi_Wmu));
            dg_add(P, wgrad_rm(h->H2r4, h->Lp2, h2 + 1, h->dhead_r4 + (long long)a * 4, 32, a, L.pi_Wls));
            for (int q = 0; q < 2; ++q) {
                DGJob j = wgrad_rm(h->xa_r4, 32, o + a + 1, h->dZ1r4 + q * H1I, h->Lp1, h1, L.q_W1[q]);
                j.type = DG_WGRAD_W1Y;   // [W1 ; b1] lives in the layer-1 block layout
                dg_add(P, j);
            }
            {
                DGJob j = wgrad_rm(h->xp_r4, 32, o + 1, h->dZ1r4 + 2 * H1I, h->Lp1, h1, L.pi_W1);
                j.type = DG_WGRAD_W1Y;
                dg_add(P, j);
            }
            if (h->bq_cols < (h1 + 31) / 32)   // the Q-head wgrads (+ Adam + polyak of W3, b3), moved here from launch "mid": nothing reads the head
                                               // kernels between the two launches (the Q layer-2 wgrads of "mid" use the W3 snapshot)
                for (int q = 0; q < 2; ++q) dg_add(P, wgrad_rm(h->H2r4 + (1 + q) * H2R, h->Lp2, h2 + 1, h->dq + (long long)q * B, 1, 1, L.q_W3[q]));
            DGJob ls{};
            ls.type = DG_LOSS; ls.M = 1; ls.N = 1; ls.K = 0; ls.adam_off = -1; ls.loss_part = h->loss_part; ls.losses = h->losses; ls.nl = 3;
            ls.nparts = -1;   // + the optimizer's books
            dg_add(P, ls);
        }
    }

    // ---- row kernels
    for (int st = 0; st < 2; ++st) {
        h->ra[st] = RowsA{h->H2, net_pi(Pm, L), net_pi(Pt, L), net_q(Pm, L, 0), net_q(Pm, L, 1),
                          h->in[st][5], h->in[st][6], h->in[st][7],
                          h->act0, h->act2, h->logp0, h->logp1, h->save0, h->q1o, h->q2o,
                          h->H1, Pm + L.q_W1[0] + (long long)o * h1, Pt + L.q_W1[0] + (long long)o * h1,
                          Pt + L.q_W1[1] + (long long)o * h1, B, h2, ldh2, a, h1, ldh1, (float)cfg->act_scale};
        h->rb[st] = RowsB{h->H2, net_q(Pm, L, 0), net_q(Pm, L, 1), net_q(Pt, L, 0), net_q(Pt, L, 1), h->in[st][3], h->in[st][4],
                          h->logp0, h->logp1, h->q1o, h->q2o, h->dZ2, h->dq4, h->loss_part, B, h2, ldh2,
                          (float)cfg->alpha, (float)cfg->gamma};
    }
    h->rc = RowsC{h->H2, h->dZ1 + 2 * BZ1, Pm + L.q_W1[0], net_pi(Pm, L), h->save0, h->dhead, h->dZ2 + 3 * BZ2,
                  h->loss_part, h->losses, B, h1, h2, ldh2, o, a, h->ldd, h->rows_b_blocks, (float)cfg->alpha,
                  (float)cfg->act_scale, 3};
    h->ad = AdamArgs{h->main_p, h->target_p, h->m, h->v, h->grad, h->opt, h->opt + 1, L.total_int, L.n_pi_int, 0,
                     (float)cfg->lr, (float)cfg->beta1, (float)cfg->beta2, (float)cfg->adam_eps,
                     (float)cfg->polyak, (float)(1.0 - cfg->polyak),
                     h->part, L.pi_W1 / 4, (long long)(o + 1) * h1 / 4, (long long)(o + 1) * h1 / 4,
                     h->fused_l1_wgrad ? (B + 31) / 32 : 0, 0u};
    h->noise_armed = false; h->noise_seed = 0; h->noise_pending = 0; h->grad_imported = false;
    h->fuse_apply = false; h->sample_armed = false;
    *out = h;
    return DDRL_OK;
}

int ddrl_sac1_destroy(ddrl_sac1_t *h) {
    if (!h) return DDRL_OK;
    ddrl::DeviceGuard g(h->device);
    return sac1_free(h);
}

static float *which_buf(ddrl_sac1 *h, int which) {
    switch (which) {
        case DDRL_SAC1_MAIN: return h->main_p;
        case DDRL_SAC1_TARGET: return h->target_p;
        case DDRL_SAC1_ADAM_M: return h->m;
        case DDRL_SAC1_ADAM_V: return h->v;
        case DDRL_SAC1_GRAD: return h->grad;
        default: return nullptr;
    }
}

int ddrl_sac1_export(ddrl_sac1_t *h, int which, float *flat_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_d != nullptr, "NULL pointer");
    float *buf = which_buf(h, which);
    DDRL_REQUIRE(buf != nullptr, "unknown buffer id");
    ddrl::DeviceGuard g(h->device);
    if (which == DDRL_SAC1_GRAD && h->fused_l1_wgrad && !h->grad_imported) {
        // the pi layer-1 gradient exists only as row-tile partials until Adam runs: materialise it
        const long long n = (long long)(h->cfg.obs_dim + 1) * h->cfg.hidden1;
        if (h->fused)
            k_reduce_parts_w1y<<<(unsigned)((n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(h->part, h->grad + h->L.pi_W1, h->cfg.obs_dim + 1,
                                                                                               h->cfg.hidden1, h->ad.nparts);
        else
            k_reduce_parts<<<(unsigned)((n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(h->part, h->grad + h->L.pi_W1, n, n, h->ad.nparts);
    }
    k_pack<<<dim3(64, (unsigned)h->L.segs.size()), 256, 0, ddrl::as_stream(stream)>>>(h->segs_d, buf, flat_d, nullptr, 0);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_sac1_grad_buffer(ddrl_sac1_t *h, float **grad_d, int64_t *n) {
    DDRL_REQUIRE(h != nullptr && grad_d != nullptr && n != nullptr, "NULL pointer");
    *grad_d = h->grad;
    *n = h->L.total_int;
    return DDRL_OK;
}

int ddrl_sac1_grad_finalize(ddrl_sac1_t *h, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    ddrl::DeviceGuard g(h->device);
    if (h->fused_l1_wgrad && !h->grad_imported) {  // the pi layer-1 gradient exists only as row-tile partials: sum them into the buffer
        const long long n = (long long)(h->cfg.obs_dim + 1) * h->cfg.hidden1;
        if (h->fused)
            k_reduce_parts_w1y<<<(unsigned)((n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(h->part, h->grad + h->L.pi_W1, h->cfg.obs_dim + 1,
                                                                                               h->cfg.hidden1, h->ad.nparts);
        else
            k_reduce_parts<<<(unsigned)((n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(h->part, h->grad + h->L.pi_W1, n, n, h->ad.nparts);
        DDRL_LAUNCH_CHECK();
    }
    h->grad_imported = true;  // apply_grads takes the buffer as it is (e.g. after an in-place all-reduce)
    return DDRL_OK;
}

int ddrl_sac1_import(ddrl_sac1_t *h, int which, const float *flat_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_d != nullptr, "NULL pointer");
    float *buf = which_buf(h, which);
    DDRL_REQUIRE(buf != nullptr, "unknown buffer id");
    ddrl::DeviceGuard g(h->device);
    k_pack<<<dim3(64, (unsigned)h->L.segs.size()), 256, 0, ddrl::as_stream(stream)>>>(h->segs_d, flat_d, buf, nullptr, 1);
    if (which == DDRL_SAC1_MAIN) refresh_shadows(h, ddrl::as_stream(stream));
    DDRL_LAUNCH_CHECK();
    if (which == DDRL_SAC1_GRAD) h->grad_imported = true;
    return DDRL_OK;
}

int ddrl_sac1_set_weights(ddrl_sac1_t *h, const float *flat_main_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_main_d != nullptr, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    // copy into main AND target: Learner.set_weights runs target_init (actor_learner.py:125-127)
    k_pack<<<dim3(64, (unsigned)h->L.segs.size()), 256, 0, ddrl::as_stream(stream)>>>(h->segs_d, flat_main_d, h->main_p,
                                                                                    h->target_p, 1);
    refresh_shadows(h, ddrl::as_stream(stream));
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_sac1_get_weights(ddrl_sac1_t *h, float *flat_main_d, void *stream) {
    return ddrl_sac1_export(h, DDRL_SAC1_MAIN, flat_main_d, stream);
}

int ddrl_sac1_opt_steps(ddrl_sac1_t *h, int64_t *t_pi_h, int64_t *t_q_h, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    ddrl::DeviceGuard g(h->device);
    OptState o;
    hipStream_t s = ddrl::as_stream(stream);
    DDRL_HIP_CHECK(hipMemcpyAsync(&o, h->opt + h->opt_cur, sizeof(o), hipMemcpyDeviceToHost, s));
    DDRL_HIP_CHECK(hipStreamSynchronize(s));
    if (t_pi_h) *t_pi_h = o.t_pi;
    if (t_q_h) *t_q_h = o.t_q;
    return DDRL_OK;
}

int ddrl_sac1_opt_state_get(ddrl_sac1_t *h, int64_t *t_pi_h, int64_t *t_q_h, uint64_t *noise_ctr_h, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    ddrl::DeviceGuard g(h->device);
    OptState o;
    hipStream_t s = ddrl::as_stream(stream);
    DDRL_HIP_CHECK(hipMemcpyAsync(&o, h->opt + h->opt_cur, sizeof(o), hipMemcpyDeviceToHost, s));
    DDRL_HIP_CHECK(hipStreamSynchronize(s));
    if (t_pi_h) *t_pi_h = o.t_pi;
    if (t_q_h) *t_q_h = o.t_q;
    if (noise_ctr_h) *noise_ctr_h = o.noise_ctr;
    return DDRL_OK;
}

int ddrl_sac1_opt_state_set(ddrl_sac1_t *h, int64_t t_pi, int64_t t_q, uint64_t noise_ctr, void *stream) {
    DDRL_REQUIRE(h != nullptr && t_pi >= 0 && t_q >= 0, "NULL handle or negative step count");
    ddrl::DeviceGuard g(h->device);
    // beta powers as TF keeps them: a running float32 product, one multiply per applied step
    OptState o{};
    const float b1 = (float)h->cfg.beta1, b2 = (float)h->cfg.beta2;
    o.b1p_pi = b1; o.b2p_pi = b2; o.b1p_q = b1; o.b2p_q = b2;
    for (int64_t i = 0; i < t_pi; ++i) { o.b1p_pi *= b1; o.b2p_pi *= b2; }
    for (int64_t i = 0; i < t_q; ++i) { o.b1p_q *= b1; o.b2p_q *= b2; }
    o.t_pi = t_pi; o.t_q = t_q; o.noise_ctr = noise_ctr;
    hipStream_t s = ddrl::as_stream(stream);
    h->opt_cur = 0;
    DDRL_HIP_CHECK(hipMemcpyAsync(h->opt, &o, sizeof(o), hipMemcpyHostToDevice, s));
    DDRL_HIP_CHECK(hipStreamSynchronize(s));
    return DDRL_OK;
}

// Direct path: the dgrad images [h2/4][h1][4] of the main layer-2 kernels, regenerated from the k4-interleaved parameters
// whenever those change outside the optimizer epilogues (set_weights / import / the flat Adam kernel).
static void refresh_shadows(ddrl_sac1 *h, hipStream_t s) {
    if (!h->fused) return;
    const Layout &L = h->L;
    const int h1 = h->cfg.hidden1, h2 = h->cfg.hidden2;
    ShadowJobs sj{};
    int n = 0;
    sj.j4[n] = h->main_p + L.pi_W2; sj.c4[n++] = h->c4_pi[h->sh_cur];
    for (int q = 0; q < 2; ++q) { sj.j4[n] = h->main_p + L.q_W2[q]; sj.c4[n++] = (q == 1 && h->mid_rest_job >= 0 && h->sh_cur) ? h->c4_q2b : h->c4_q[q]; }
    if (h->cfg.variant == DDRL_SAC_V) { sj.j4[n] = h->main_p + L.v_W2; sj.c4[n++] = h->c4_q[2]; }
    k_shadow<<<dim3((h1 + 31) / 32, (h2 / 4 + 7) / 8, n), 256, 0, s>>>(sj, h1, h2, L.Np2, L.Kp1);   // one launch for all networks
}

static void launch_rows_c(ddrl_sac1 *h, hipStream_t s) {
    const RowsC &c = h->rc;
    const float *b = h->slab;
    k_rows_c<<<c.B + 1, 64, 0, s>>>(b, (int)(c.dZ1q - b), (int)(c.H2 - b), (int)(c.W1q1 - b), (int)(c.pi.Wmu - b), (int)(c.pi.Wls - b),
                                    (int)(c.save0 - b), c.h1 | (c.h2 << 16), c.ldh2 | (c.obs << 16), c.act | (c.B << 16), c);
}

// One stage of the update (input set `st`).  Stage ids as documented for ddrl_sac1_stage_time.
static void launch_stage(ddrl_sac1 *h, int stage, int st, hipStream_t s) {
    const ddrl_sac1_config_t &c = h->cfg;
    const int B = c.batch;
    const dim3 l1grid((c.hidden1 + 255) / 256, (B + L1_ROWS - 1) / L1_ROWS, 1);
    if (h->fused) {
        switch (stage) {
            case 2: h->f_a[st].opt = h->opt + h->opt_cur; launch_dfwd<0>(h->fh_a[st], h->f_a[st], s); break;
            case 5: {
                DFArgs &F = h->f_b[st];
                F.do_sample = h->sample_armed ? 1 : 0;
                if (h->sample_armed) {
                    F.rs = h->smp_rs; F.ring = h->smp_ring; F.sample_batch = h->cfg.batch;
                    float **b = h->in[h->smp_set];
                    F.sout = ddrl_replay_dev::BatchPtrs{{b[0], b[1], b[2], b[3], b[4], nullptr}};
                }
                launch_dfwd<1>(h->fh_b[st], F, s);
                break;
            }
            case 7: {
                DGJobs &J = h->dg_bq[st];
                if (h->mid_rest_job >= 0) J.job[2].B = h->sh_cur ? h->c4_q2b : h->c4_q[1];   // q2(x, a) dgrad: the current image of q2's layer-2 kernel
                launch_dg(J, s, 2);
                break;
            }
            case 8: {
                DGJobs &J = h->dg_mid;
                J.ad.on = h->fuse_apply ? 1 : 0;
                J.ad.opt = h->opt + h->opt_cur;
                J.job[0].B = h->c4_pi[h->sh_cur];           // the policy dgrad reads this update's image of the policy's layer-2 kernel ...
                J.rew = h->dg_bq[st].rew; J.done = h->dg_bq[st].done;   // (read by the Q dgrad columns that run here, if any)
                if (h->mid_rest_job >= 0) {
                    // those columns read the CURRENT image of q2's layer-2 kernel while the optimizer epilogue of q2's layer-2 wgrad (job 3 of
                    // this launch) writes the next one: two copies, like the policy's (found by the graph == eager test: with one copy the
                    // dgrad saw a half-stepped kernel)
                    float *cur = h->sh_cur ? h->c4_q2b : h->c4_q[1], *nxt = h->sh_cur ? h->c4_q[1] : h->c4_q2b;
                    J.job[h->mid_rest_job].B = cur + (long long)h->bq_cols * 32 * 4;
                    J.job[3].shadow = J.ad.on ? nxt : nullptr;
                }
                launch_dg(J, s, 3);
                break;
            }
            case 9: {
                DGJobs &J = h->dg_pi;
                J.ad.on = h->fuse_apply ? 1 : 0;
                J.ad.opt = h->opt + h->opt_cur;
                J.job[0].shadow = h->c4_pi[h->sh_cur ^ 1];  // ... the optimizer epilogue of the policy's layer-2 wgrad writes the next one
                J.ad.opt_next = h->opt + (h->opt_cur ^ 1);  // the optimizer bookkeeping rides in this launch (its loss tile)
                J.ad.noise_adv = h->noise_pending;
                launch_dg(J, s, 4);
                if (J.ad.on) { h->opt_cur ^= 1; h->sh_cur ^= 1; }
                break;
            }
            case 11: {
                const long long blocks = (h->L.total_int / 4 + 255) / 256;
                h->ad.adam_blocks = (int)blocks;
                h->ad.opt = h->opt + h->opt_cur; h->ad.opt_next = h->opt + (h->opt_cur ^ 1);
                h->opt_cur ^= 1;
                const int nparts = h->ad.nparts;
                if (nparts > 0) {  // the policy's layer-1 gradient exists as row-tile partials: sum them into the (block-layout) gradient first
                    const int nk = h->cfg.obs_dim + 1, N = h->cfg.hidden1;
                    k_reduce_parts_w1y<<<(unsigned)((nk * N + 255) / 256), 256, 0, s>>>(h->part, h->grad + h->L.pi_W1, nk, N, nparts);
                }
                h->ad.nparts = 0;
                {   // the dgrad images of the main layer-2 kernels (what refresh_shadows would rewrite in a launch of its own) ride in the flat step
                    const Layout &L = h->L;
                    int n = 0;
                    h->ad.sh_off4[n] = L.pi_W2 >> 2; h->ad.sh_dst[n++] = h->c4_pi[h->sh_cur];
                    for (int q = 0; q < 2; ++q) { h->ad.sh_off4[n] = L.q_W2[q] >> 2; h->ad.sh_dst[n++] = (q == 1 && h->mid_rest_job >= 0 && h->sh_cur) ? h->c4_q2b : h->c4_q[q]; }
                    if (h->cfg.variant == DDRL_SAC_V) { h->ad.sh_off4[n] = L.v_W2 >> 2; h->ad.sh_dst[n++] = h->c4_q[2]; }
                    h->ad.n_sh = n; h->ad.sh_K = h->cfg.hidden1; h->ad.sh_N = h->cfg.hidden2; h->ad.sh_Np = L.Np2; h->ad.sh_ld = L.Kp1;
                }
                k_adam_polyak<<<(unsigned)(blocks + (h->ad.do_sample ? 1 : 0)), 256, 0, s>>>(h->ad);
                h->ad.nparts = nparts; h->ad.n_sh = 0;
                break;
            }
            default: break;  // 1, 3, 4, 6, 10: folded into the launches above
        }
        return;
    }
    switch (stage) {
        case 1: h->l1a[st].opt = h->opt + h->opt_cur; k_l1<<<dim3(l1grid.x, l1grid.y, h->l1a[st].njobs), 256, 0, s>>>(h->l1a[st]); break;
        case 2: launch_gemm(h->g_fa, s); break;
        case 3:
            if (c.variant == DDRL_SAC_V) k_rows_a_v<<<(B * 5 + 3) / 4, 256, 0, s>>>(h->rav[st]);
            else k_rows_a<<<(B * 5 + 3) / 4, 256, 0, s>>>(h->ra[st]);
            break;
        case 5: launch_gemm(h->g_fb, s); break;
        case 6:
            if (c.variant == DDRL_SAC_V) k_rows_b_v<<<B, 64, 0, s>>>(h->rbv[st]);
            else k_rows_b<<<B, 64, 0, s>>>(h->rb[st]);
            break;
        case 7: launch_gemm(h->g_bq, s); break;
        case 8: launch_rows_c(h, s); break;  // +1: the loss-reduction workgroup
        case 9: launch_gemm(h->g_bpi, s); break;
        case 10: if (h->g_last.total_tiles > 0) launch_gemm(h->g_last, s); break;
        case 11: {
            const long long blocks = (h->L.total_int / 4 + 255) / 256;
            h->ad.adam_blocks = (int)blocks;
            h->ad.opt = h->opt + h->opt_cur; h->ad.opt_next = h->opt + (h->opt_cur ^ 1);
            h->opt_cur ^= 1;
            k_adam_polyak<<<(unsigned)(blocks + (h->ad.do_sample ? 1 : 0)), 256, 0, s>>>(h->ad);
            break;
        }
        default: break;  // 4: folded into stage 3
    }
}

static int launch_grads(ddrl_sac1 *h, const float *obs1, const float *obs2, const float *acts, const float *rews,
                        const float *done, const float *e0, const float *e1, const float *e2, float *losses_d, float *q1_d,
                        float *q2_d, float *logp_d, hipStream_t s) {
    const ddrl_sac1_config_t &c = h->cfg;
    const int B = c.batch;
    const float *src[8] = {obs1, obs2, acts, rews, done, e0, e1, e2};
    int st = -1;  // which input set the caller's pointers are (zero-copy), if any
    for (int cand = 0; cand < 2 && st < 0; ++cand) {
        bool same = true;
        for (int i = 0; i < 8; ++i) same = same && (src[i] == h->in[cand][i]);
        if (same) st = cand;
    }
    if (st < 0) {  // stage 0: copy the caller's batch into set 0
        st = 0;
        StageArgs sa{};
        const int n[8] = {B * c.obs_dim, B * c.obs_dim, B * c.act_dim, B, B, B * c.act_dim, B * c.act_dim, B * c.act_dim};
        for (int i = 0; i < 8; ++i) { sa.src[i] = src[i]; sa.dst[i] = h->in[0][i]; sa.n[i] = n[i]; }
        k_stage<<<dim3((unsigned)((B * c.obs_dim + 255) / 256), 8), 256, 0, s>>>(sa);
    }
    h->l1a[st].noise_on = h->noise_armed ? 1 : 0;
    h->l1a[st].noise_seed = h->noise_seed;
    h->f_a[st].noise_on = h->noise_armed ? 1 : 0;
    h->f_a[st].noise_seed = h->noise_seed;
    h->noise_pending = h->noise_armed ? (unsigned)(3 * B * c.act_dim) : 0u;
    h->noise_armed = false;
    h->grad_imported = false;
    for (int stage = 1; stage <= 10; ++stage) launch_stage(h, stage, st, s);
    DDRL_LAUNCH_CHECK();
    if (h->fuse_apply) h->noise_pending = 0;  // consumed by the optimizer bookkeeping of the last stage
    if (losses_d) DDRL_HIP_CHECK(hipMemcpyAsync(losses_d, h->losses, (size_t)h->rc.nl * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (q1_d || q2_d || logp_d) {
        k_copy3<<<(B + 255) / 256, 256, 0, s>>>(h->q1o, h->q2o, h->logp0, q1_d, q2_d, logp_d, B);
        DDRL_LAUNCH_CHECK();
    }
    return DDRL_OK;
}

static int launch_apply(ddrl_sac1 *h, hipStream_t s) {
    h->ad.noise_adv = h->noise_pending;
    h->noise_pending = 0;
    const int nparts = h->ad.nparts;
    if (h->grad_imported) h->ad.nparts = 0;  // use the imported (e.g. all-reduced) gradient as is
    launch_stage(h, 11, 0, s);
    h->ad.nparts = nparts;
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_sac1_input_buffers(ddrl_sac1_t *h, int set, float **bufs_h) {
    DDRL_REQUIRE(h != nullptr && bufs_h != nullptr && (set == 0 || set == 1), "NULL pointer or set not in {0,1}");
    for (int i = 0; i < 8; ++i) bufs_h[i] = h->in[set][i];
    return DDRL_OK;
}

int ddrl_sac1_batch(ddrl_sac1_t *h) { return h ? h->cfg.batch : DDRL_ERR_BAD_ARG; }

int ddrl_sac1_is_fused(ddrl_sac1_t *h) { return h ? (h->fused ? 1 : 0) : DDRL_ERR_BAD_ARG; }

int ddrl_sac1_fill_noise(ddrl_sac1_t *h, uint32_t seed, void *stream) {
    // No kernel of its own: the next compute_grads / step generates eps_x, eps_x2, eps_t inside its
    // first kernel (k_l1) from hash(seed, device counter + i) and its Adam kernel advances the counter.
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    (void)stream;
    h->noise_armed = true;
    h->noise_seed = seed;
    return DDRL_OK;
}

int ddrl_sac1_stage_time(ddrl_sac1_t *h, int stage, int reps, float *ms_per_launch_h, void *stream) {
    DDRL_REQUIRE(h != nullptr && ms_per_launch_h != nullptr, "NULL pointer");
    DDRL_REQUIRE(stage >= 1 && stage <= 10 && reps > 0, "stage must be in [1,10] (idempotent stages), reps > 0");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    hipEvent_t e0, e1;
    DDRL_HIP_CHECK(hipEventCreate(&e0));
    DDRL_HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_stage(h, stage, 0, s);
    DDRL_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch_stage(h, stage, 0, s);
    DDRL_HIP_CHECK(hipEventRecord(e1, s));
    DDRL_LAUNCH_CHECK();
    DDRL_HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    DDRL_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch_h = ms / (float)reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return DDRL_OK;
}

int ddrl_sac1_compute_grads(ddrl_sac1_t *h, const float *obs1_d, const float *obs2_d, const float *acts_d,
                            const float *rews_d, const float *done_d, const float *eps_x_d, const float *eps_x2_d,
                            const float *eps_t_d, float *losses_d, float *q1_d, float *q2_d, float *logp_pi_d,
                            void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    DDRL_REQUIRE(obs1_d && obs2_d && acts_d && rews_d && done_d && eps_x_d && eps_x2_d && eps_t_d, "NULL batch/noise pointer");
    ddrl::DeviceGuard g(h->device);
    return launch_grads(h, obs1_d, obs2_d, acts_d, rews_d, done_d, eps_x_d, eps_x2_d, eps_t_d, losses_d, q1_d, q2_d,
                        logp_pi_d, ddrl::as_stream(stream));
}

int ddrl_sac1_apply_grads(ddrl_sac1_t *h, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    ddrl::DeviceGuard g(h->device);
    return launch_apply(h, ddrl::as_stream(stream));
}

int ddrl_sac1_apply_grads_and_sample(ddrl_sac1_t *h, ddrl_replay_t *replay, int set, void *stream) {
    DDRL_REQUIRE(h != nullptr && replay != nullptr && (set == 0 || set == 1), "NULL pointer or set not in {0,1}");
    ddrl::DeviceGuard g(h->device);
    const ddrl_replay_dev::SamplerView v = ddrl_replay_sampler_view(replay);
    DDRL_REQUIRE(v.ring.n_arr == 5 && v.ring.w[0] == h->cfg.obs_dim && v.ring.w[1] == h->cfg.obs_dim && v.ring.w[2] == h->cfg.act_dim &&
                     v.ring.w[3] == 1 && v.ring.w[4] == 1,
                 "replay row shape differs from the learner's (obs1, obs2, acts, rews, done)");
    if (!ddrl_replay_can_fuse(replay, h->cfg.batch)) {  // empty ring (host view) or rows too large for the one-workgroup sampler
        float **b = h->in[set];
        int rc = launch_apply(h, ddrl::as_stream(stream));
        if (rc != DDRL_OK) return rc;
        return ddrl_replay_sample(replay, h->cfg.batch, b[0], b[1], b[2], b[3], b[4], nullptr, stream);
    }
    h->ad.do_sample = 1;
    h->ad.sample_batch = h->cfg.batch;
    h->ad.rs = v.state;
    h->ad.ring = v.ring;
    h->ad.sout = ddrl_replay_dev::BatchPtrs{{h->in[set][0], h->in[set][1], h->in[set][2], h->in[set][3], h->in[set][4], nullptr}};
    const int rc = launch_apply(h, ddrl::as_stream(stream));
    h->ad.do_sample = 0;
    if (rc == DDRL_OK) ddrl_replay_note_sample(replay);
    return rc;
}

int ddrl_sac1_step(ddrl_sac1_t *h, const float *obs1_d, const float *obs2_d, const float *acts_d, const float *rews_d,
                   const float *done_d, const float *eps_x_d, const float *eps_x2_d, const float *eps_t_d,
                   float *losses_d, float *q1_d, float *q2_d, float *logp_pi_d, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    // fused path: the optimizer runs in the epilogues of the last GEMM launch — no Adam kernel
    h->fuse_apply = h->fused;
    int rc = ddrl_sac1_compute_grads(h, obs1_d, obs2_d, acts_d, rews_d, done_d, eps_x_d, eps_x2_d, eps_t_d, losses_d,
                                     q1_d, q2_d, logp_pi_d, stream);
    const bool applied = h->fuse_apply;
    h->fuse_apply = false;
    if (rc != DDRL_OK || applied) return rc;
    return ddrl_sac1_apply_grads(h, stream);
}

// train(replay_buffer.sample_batch()) with HOST arrays (example/dsac.py:142-144; algos/sac1/sac1.py:146-148) as ONE call: the caller's
// page-locked block [obs1 | obs2 | acts | rews | done], laid out like input set 0, goes up with one asynchronous copy, the three noise
// tensors are generated in place behind it (counter-based, ddrl_normal_fill), and the update reads the input set it owns.
int ddrl_sac1_step_host(ddrl_sac1_t *h, float *block_h, int64_t n_floats, uint32_t noise_seed, uint64_t noise_ctr, float *losses_d,
                        void *stream) {
    DDRL_REQUIRE(h != nullptr && block_h != nullptr, "NULL pointer");
    const ddrl_sac1_config_t &c = h->cfg;
    const long long B = c.batch, m = B * c.act_dim;
    float **in0 = h->in[0];
    // the five items of set 0 follow one another exactly as ddrl_sac1_create lays them out: rup32(batch) rows each on the direct path
    // (the caller's block holds zeros in the padding rows), every item rounded up to 64 floats — no other buffer inside the span
    const long long Bp = h->fused ? rup32(B) : B;
    const long long w[5] = {c.obs_dim, c.obs_dim, c.act_dim, 1, 1};
    for (int j = 0; j < 4; ++j)
        DDRL_REQUIRE(in0[j + 1] - in0[j] == ((Bp * w[j] + 63) & ~63ll), "input set 0 is not one contiguous span: use ddrl_sac1_step with device pointers");
    DDRL_REQUIRE(n_floats == (in0[4] - in0[0]) + B, "block length differs from input set 0's span");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    // the noise counter travels in the two words behind the batch: a captured replay reads it from there
    uint32_t *tail = reinterpret_cast<uint32_t *>(block_h + n_floats);
    tail[0] = (uint32_t)noise_ctr; tail[1] = (uint32_t)(noise_ctr >> 32);
    if (!h->host_ctr_d) DDRL_HIP_CHECK(hipMalloc((void **)&h->host_ctr_d, 2 * sizeof(uint32_t)));
    const bool eps_contig = in0[6] - in0[5] == m && in0[7] - in0[6] == m;
    // a page-locked block is read by the device itself (one launch: batch up + noise); pageable memory goes through copies
    const float *block_dev = nullptr;
    bool resolved = false;
    auto resolve = [&]() {   // (not on the replay path: a recorded graph holds the address)
        if (resolved) return;
        resolved = true;
        hipPointerAttribute_t pa{};
        if (hipPointerGetAttributes(&pa, block_h) == hipSuccess && pa.type == hipMemoryTypeHost && pa.devicePointer != nullptr)
            block_dev = static_cast<const float *>(pa.devicePointer);
        else (void)hipGetLastError();
    };
    auto issue = [&]() -> int {
        resolve();
        if (block_dev) {
            const int rc = ddrl_internal_host_block_up(block_dev, in0[0], n_floats, in0[5], in0[6], in0[7], m, noise_seed,
                                                       reinterpret_cast<const uint32_t *>(block_dev + n_floats), stream);
            if (rc != DDRL_OK) return rc;
            return ddrl_sac1_step(h, in0[0], in0[1], in0[2], in0[3], in0[4], in0[5], in0[6], in0[7], losses_d, nullptr, nullptr, nullptr, stream);
        }
        DDRL_HIP_CHECK(hipMemcpyAsync(in0[0], block_h, (size_t)n_floats * sizeof(float), hipMemcpyHostToDevice, s));
        DDRL_HIP_CHECK(hipMemcpyAsync(h->host_ctr_d, tail, 2 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        int rc = DDRL_OK;
        if (eps_contig) rc = ddrl_internal_normal_fill_ctr(in0[5], 3 * m, noise_seed, h->host_ctr_d, 0, stream);
        else
            for (int i = 0; i < 3 && rc == DDRL_OK; ++i) rc = ddrl_internal_normal_fill_ctr(in0[5 + i], m, noise_seed, h->host_ctr_d, (uint64_t)(i * m), stream);
        if (rc != DDRL_OK) return rc;
        return ddrl_sac1_step(h, in0[0], in0[1], in0[2], in0[3], in0[4], in0[5], in0[6], in0[7], losses_d, nullptr, nullptr, nullptr, stream);
    };
    static const bool graphs_on = !(getenv("DDRL_HOST_GRAPH") && atoi(getenv("DDRL_HOST_GRAPH")) == 0);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (!graphs_on || !h->fused || s == nullptr || cap != hipStreamCaptureStatusNone) return issue();   // (the legacy null stream cannot be captured)
    if (h->noise_armed || h->sample_armed) return issue();   // a request armed through another call belongs to THIS update only: not into a graph
    for (auto &hg : h->host_graphs)
        if (hg.block == block_h && hg.losses == losses_d && hg.seed == noise_seed && hg.opt_cur == h->opt_cur && hg.sh_cur == h->sh_cur) {
            DDRL_HIP_CHECK(hipGraphLaunch(hg.exec, s));
            h->opt_cur ^= 1; h->sh_cur ^= 1;   // what the recorded launches did to the host-side state when they were recorded
            return DDRL_OK;
        }
    if (h->host_graphs.size() >= 16) return issue();   // (a caller that keeps changing blocks: eager)
    resolve();
    if (!block_dev) return issue();                    // pageable memory: the copies are not capturable
    ddrl_sac1::HostGraph hg{block_h, losses_d, noise_seed, h->opt_cur, h->sh_cur, nullptr};
    hipGraph_t graph = nullptr;
    // the recorded launches advance the host-side launch state without running: kept if the graph then runs once, put back if not
    const ddrl_sac1::HostSnap before{true, h->opt_cur, h->sh_cur, h->fuse_apply, h->sample_armed, h->noise_armed, h->grad_imported, h->noise_seed, h->noise_pending};
    auto put_back = [&]() {
        h->opt_cur = before.opt_cur; h->sh_cur = before.sh_cur; h->fuse_apply = before.fuse_apply; h->sample_armed = before.sample_armed;
        h->noise_armed = before.noise_armed; h->grad_imported = before.grad_imported; h->noise_seed = before.noise_seed; h->noise_pending = before.noise_pending;
    };
    DDRL_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = issue();
    const hipError_t e2 = hipStreamEndCapture(s, &graph);
    if (rc != DDRL_OK || e2 != hipSuccess || graph == nullptr) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        put_back();
        if (rc != DDRL_OK) return rc;                  // the update itself was refused (arguments): the same eagerly
        return issue();                                // the capture was refused: nothing ran — this update goes eagerly
    }
    const hipError_t e3 = hipGraphInstantiate(&hg.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e3 != hipSuccess) { (void)hipGetLastError(); put_back(); return issue(); }
    h->host_graphs.push_back(hg);
    DDRL_HIP_CHECK(hipGraphLaunch(hg.exec, s));   // (the capture recorded the update — host state advanced once — now it runs once)
    return DDRL_OK;
}

}  // extern "C"

__global__ void k_opt_copy(const OptState *src, OptState *dst) { *dst = *src; }

// Internal (loop.hip): make copy 0 of the double-buffered optimizer state the current one, so that a
// captured graph starts and ends on the same copy whatever the number of updates it holds.
int ddrl_sac1_internal_opt_sync(ddrl_sac1 *h, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    if (h->opt_cur == 0 && !(h->fused && h->sh_cur != 0)) return DDRL_OK;
    ddrl::DeviceGuard g(h->device);
    if (h->opt_cur != 0) {
        k_opt_copy<<<1, 1, 0, ddrl::as_stream(stream)>>>(h->opt + 1, h->opt);
        DDRL_LAUNCH_CHECK();
        h->opt_cur = 0;
    }
    if (h->fused && h->sh_cur != 0) {  // same for the double-buffered dgrad image of the policy's layer-2 kernel
        DDRL_HIP_CHECK(hipMemcpyAsync(h->c4_pi[0], h->c4_pi[1], (size_t)h->L.Np2 * h->L.Kp1 * sizeof(float), hipMemcpyDeviceToDevice,
                                      ddrl::as_stream(stream)));
        if (h->mid_rest_job >= 0)      // ... and of q2's, when it is double-buffered too
            DDRL_HIP_CHECK(hipMemcpyAsync(h->c4_q[1], h->c4_q2b, (size_t)h->L.Np2 * h->L.Kp1 * sizeof(float), hipMemcpyDeviceToDevice,
                                          ddrl::as_stream(stream)));
        h->sh_cur = 0;
    }
    return DDRL_OK;
}

extern "C" {

int ddrl_sac1_graph_sync(ddrl_sac1_t *h, void *stream) { return ddrl_sac1_internal_opt_sync(h, stream); }

int ddrl_sac1_capture_begin(ddrl_sac1_t *h) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    h->snap = ddrl_sac1::HostSnap{true, h->opt_cur, h->sh_cur, h->fuse_apply, h->sample_armed, h->noise_armed, h->grad_imported, h->noise_seed, h->noise_pending};
    return DDRL_OK;
}

int ddrl_sac1_capture_abort(ddrl_sac1_t *h) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    DDRL_REQUIRE(h->snap.valid, "ddrl_sac1_capture_abort without ddrl_sac1_capture_begin");
    const ddrl_sac1::HostSnap &s = h->snap;
    h->opt_cur = s.opt_cur; h->sh_cur = s.sh_cur; h->fuse_apply = s.fuse_apply; h->sample_armed = s.sample_armed;
    h->noise_armed = s.noise_armed; h->grad_imported = s.grad_imported; h->noise_seed = s.noise_seed; h->noise_pending = s.noise_pending;
    h->snap.valid = false;
    return DDRL_OK;
}

int ddrl_sac1_compute_grads_and_sample(ddrl_sac1_t *h, int set_in, ddrl_replay_t *replay, int set_out, void *stream) {
    DDRL_REQUIRE(h != nullptr && replay != nullptr && (set_in == 0 || set_in == 1) && (set_out == 0 || set_out == 1) && set_in != set_out,
                 "NULL pointer, or input sets not {0,1} / not distinct");
    ddrl::DeviceGuard g(h->device);
    float **b = h->in[set_in];
    if (!h->fused || !ddrl_replay_can_fuse(replay, h->cfg.batch)) {  // generic kernels: the sampler as its own launch
        int rc = ddrl_sac1_compute_grads(h, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], nullptr, nullptr, nullptr, nullptr, stream);
        if (rc != DDRL_OK) return rc;
        float **o = h->in[set_out];
        return ddrl_replay_sample(replay, h->cfg.batch, o[0], o[1], o[2], o[3], o[4], nullptr, stream);
    }
    const ddrl_replay_dev::SamplerView v = ddrl_replay_sampler_view(replay);
    DDRL_REQUIRE(v.ring.n_arr == 5 && v.ring.w[0] == h->cfg.obs_dim && v.ring.w[1] == h->cfg.obs_dim && v.ring.w[2] == h->cfg.act_dim &&
                     v.ring.w[3] == 1 && v.ring.w[4] == 1,
                 "replay row shape differs from the learner's (obs1, obs2, acts, rews, done)");
    h->sample_armed = true; h->smp_rs = v.state; h->smp_ring = v.ring; h->smp_set = set_out;
    const int rc = ddrl_sac1_compute_grads(h, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], nullptr, nullptr, nullptr, nullptr, stream);
    h->sample_armed = false;
    if (rc == DDRL_OK) ddrl_replay_note_sample(replay);
    return rc;
}

int ddrl_sac1_step_and_sample(ddrl_sac1_t *h, int set_in, ddrl_replay_t *replay, int set_out, void *stream) {
    DDRL_REQUIRE(h != nullptr && replay != nullptr && (set_in == 0 || set_in == 1) && (set_out == 0 || set_out == 1) && set_in != set_out,
                 "NULL pointer, or input sets not {0,1} / not distinct");
    ddrl::DeviceGuard g(h->device);
    float **b = h->in[set_in];
    if (!h->fused || !ddrl_replay_can_fuse(replay, h->cfg.batch)) {  // generic kernels: sampler beside the Adam kernel (or on its own)
        int rc = ddrl_sac1_compute_grads(h, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], nullptr, nullptr, nullptr, nullptr, stream);
        if (rc != DDRL_OK) return rc;
        return ddrl_sac1_apply_grads_and_sample(h, replay, set_out, stream);
    }
    const ddrl_replay_dev::SamplerView v = ddrl_replay_sampler_view(replay);
    DDRL_REQUIRE(v.ring.n_arr == 5 && v.ring.w[0] == h->cfg.obs_dim && v.ring.w[1] == h->cfg.obs_dim && v.ring.w[2] == h->cfg.act_dim &&
                     v.ring.w[3] == 1 && v.ring.w[4] == 1,
                 "replay row shape differs from the learner's (obs1, obs2, acts, rews, done)");
    h->sample_armed = true; h->smp_rs = v.state; h->smp_ring = v.ring; h->smp_set = set_out;
    const int rc = ddrl_sac1_step(h, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], nullptr, nullptr, nullptr, nullptr, stream);
    h->sample_armed = false;
    if (rc == DDRL_OK) ddrl_replay_note_sample(replay);
    return rc;
}

}  // extern "C"

// ==========================================================================================
// Actor: batched Actor.get_action
// ==========================================================================================
struct ddrl_actor {
    int device;
    ddrl_sac1_config_t cfg;
    Layout L;
    float *pi_p;  // padded internal
    float *H1, *H2;
    Seg *segs_d;
    long long max_rows;
    int ldh1, ldh2;
    // fused rollout step (ddrl_rollout_step): the policy in the direct-operand layout + the observation rows it acts on +
    // the head partials of the forward launch, in ONE slab (k_dfwd addresses everything as base + 32-bit offset)
    bool direct;
    Layout Ld;
    float *dslab, *pi_d, *obs_d, *hp_d;
    Seg *segs_dd;
    // version store (ddrl_actor_versions_enable): n_slots copies of the direct-layout policy, the slot every env acts on, the
    // grouping of the envs by slot for the forward launch.  pi_d stays the newest weights.
    int n_slots;
    long long vstride;
    float *vslab;
    int *slot_d, *perm_d;          // perm_d: the row list of every tile, [tiles][32] (+ parking space of k_version_plan)
    bool plan_fresh;               // the tile table matches the envs' slots (set by a plan launch, cleared by whatever moves an env)
    bool pi_p_stale;               // direct actors: the row-major copy (ddrl_actor_act / get_weights) is rebuilt from pi_d on demand
    float *flat_tmp;               // ... through this dense staging vector
    VerTile *vtiles_d;
    VerState *vs_d;
    int vt_cap;                    // records of the forward's workgroup table (= the versioned forward's grid)
    float *act1_part;              // ddrl_actor_act_one: head partials + ticket (allocated at the first call)
    int *vcnt_d;                   // envs per slot while an env-step launch counts them; all zero between launches
    long long perm2d_off;          // perm_d: where the env-step launch's [n_slots][max_rows] row lists start
    int wg_slots;                  // resident workgroups of the two-per-CU forward: 2 x CUs (the planning launch sizes the column split for it)
    long long steps_since_install;   // host-side: >= the envs' max_ep_len <=> every env has adopted the newest version
};

// ---- version store kernels -------------------------------------------------------------------------------------------
// One workgroup.  Picks the slot the incoming weights go to: the newest slot itself when no env has adopted it yet (nobody
// can ever act on it again once a newer version exists: a worker pulls whatever the server holds at ITS episode end), else
// the lowest slot no env acts on.  n_slots >= min(n_envs, max_ep_len) + 2 always leaves one.
__global__ void __launch_bounds__(256) k_version_copy(const float *__restrict__ src, float *__restrict__ vslab, long long vstride,
                                                      const VerState *__restrict__ vs, long long n4) {
    float4 *dst = reinterpret_cast<float4 *>(vslab + (long long)vs->target * vstride);
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) dst[i] = s4[i];
}

// One workgroup: envs grouped by slot (counting sort; the order inside a group is immaterial — rows are computed
// independently), one VerTile per 32 envs of a group.
// arr[s] += 1 for every valid lane, the lane's position in its group returned.  A few thousand envs sit on a handful of versions most
// of the time: 64 lanes on ONE LDS address are 64 serialised atomics, so the lanes that share the slot of the first pending lane go
// as one atomic (four such rounds: the four largest groups of the wave, typically all of it); what is left takes its own.
__device__ __forceinline__ int ver_group_add(int *arr, int s, bool valid, int lane) {
    unsigned long long todo = __ballot(valid);
    int pos = 0;
    for (int round = 0; round < 4 && todo; ++round) {
        const int leader = __ffsll((long long)todo) - 1;
        const int s0 = __shfl(s, leader);
        const bool mine = valid && s == s0;
        const unsigned long long m = __ballot(mine);
        int base = 0;
        if (lane == leader) base = atomicAdd(&arr[s0], __popcll(m));
        base = __shfl(base, leader);
        if (mine) { pos = base + __popcll(m & ((1ull << lane) - 1ull)); valid = false; }
        todo &= ~m;
    }
    if (valid) pos = atomicAdd(&arr[s], 1);
    return pos;
}
// ONE launch (one workgroup) plans a versioned forward — and, with `install`, first picks the slot incoming weights go to.
//   pick   the newest slot itself while no env has adopted it (nobody can ever act on a superseded version that nobody holds: a worker
//          pulls whatever the server holds at ITS episode end), else the lowest slot no env acts on (n_slots >= min(n_envs, max_ep_len) + 2
//          always leaves one); sticky vs->err when none is free.  The grouping does not depend on the pick: no env sits on the target.
//   group  envs by slot (counting sort; the order inside a group is immaterial — rows are computed independently), one VerTile per 32
//          envs of a group, and the envs of tile ti at rows[32 ti .. 32 ti + count): the forward reads its tile record and its row list
//          with two INDEPENDENT loads (round 4 chained n_tiles -> tile -> dense permutation -> observation rows: four round trips before
//          the first operand).  An env's position in its group is the value its histogram atomic returned — one atomic pass, not two.
constexpr int VER_REG_ENVS = 8;   // envs per thread kept in registers across the passes (8192 envs: config 4's rollout ranks)
__global__ void __launch_bounds__(1024) k_version_plan(const int *__restrict__ slot, long long n, int n_slots, int *__restrict__ rows,
                                                       VerTile *__restrict__ tiles, VerState *vs, int install, int col_tiles, int wg_slots,
                                                       int tiles_cap) {
    __shared__ int cnt[VER_MAX_SLOTS], tstart[VER_MAX_SLOTS];
    __shared__ int wsum_t[16], s_free, s_live;
    __shared__ VerSplit s_split;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int sv[VER_REG_ENVS], pv[VER_REG_ENVS];
#pragma unroll
    for (int e = 0; e < VER_REG_ENVS; ++e) sv[e] = t + 1024 * e < n ? slot[t + 1024 * e] : 0;
    for (int j = t; j < VER_MAX_SLOTS; j += 1024) cnt[j] = 0;
    if (t == 0) { s_free = n_slots; s_live = 0; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VER_REG_ENVS; ++e) pv[e] = ver_group_add(cnt, sv[e], t + 1024 * e < n, lane);
    for (long long i0 = 1024ll * VER_REG_ENVS; i0 < n; i0 += 1024) {   // (more envs than the registers hold: positions parked in `rows`' tail)
        const bool ok = i0 + t < n;
        const int p = ver_group_add(cnt, ok ? slot[i0 + t] : 0, ok, lane);
        if (ok) rows[32ll * (n / 32 + n_slots) + (i0 + t)] = p;
    }
    __syncthreads();
    const int newest = vs->newest;
    // exclusive scan of the groups' tile counts: two slots per thread, wave scan, wave totals
    const int c0 = cnt[2 * t], c1 = cnt[2 * t + 1];
    const int t0 = (c0 + 31) >> 5, t1 = (c1 + 31) >> 5;
    int st = t0 + t1;
    for (int o = 1; o < 64; o <<= 1) {
        const int ut = __shfl_up(st, o);
        if (lane >= o) st += ut;
    }
    if (lane == 63) wsum_t[w] = st;
    if (install) {
        int live = (c0 > 0) + (c1 > 0);
        int fr = n_slots;
        if (2 * t + 1 < n_slots && c1 == 0 && 2 * t + 1 != newest) fr = 2 * t + 1;
        if (2 * t < n_slots && c0 == 0 && 2 * t != newest) fr = 2 * t;
        for (int o = 32; o >= 1; o >>= 1) { live += __shfl_xor(live, o); fr = min(fr, __shfl_xor(fr, o)); }
        if (lane == 0) { atomicAdd(&s_live, live); atomicMin(&s_free, fr); }
    }
    __syncthreads();
    int bt = 0;
    for (int k = 0; k < w; ++k) bt += wsum_t[k];
    const int et = bt + st - (t0 + t1);   // exclusive prefix of this thread's pair
    tstart[2 * t] = et; tstart[2 * t + 1] = et + t0;
    if (t == 1023) {
        const VerSplit sp = ver_split(bt + st, col_tiles, wg_slots, tiles_cap);
        vs->n_tiles = bt + st; vs->n_wgs = sp.n_wgs; s_split = sp;
    }
    if (t == 0 && install) {
        int target = newest;
        if (cnt[newest] > 0) {
            if (s_free < n_slots) target = s_free;
            else vs->err = 1;              // sticky: no free slot (the newest one is overwritten: its envs act on fresher weights)
        }
        vs->target = target;
        vs->newest = target;
        vs->live = s_live;
    }
    __syncthreads();
    // one record per WORKGROUP of the forward (ver_split: the long workgroups of the full rounds first, the surplus tiles' short ones
    // behind them).  Tile ti belongs to the LAST slot whose first tile is <= ti (the empty slots behind it start where it ends)
    const VerSplit sp = s_split;
    const int n_long_wgs = sp.n_long * sp.g_long;
    for (int b = t; b < sp.n_wgs; b += 1024) {
        const bool lg = b < n_long_wgs;
        const int g = lg ? sp.g_long : sp.g_short, bb = lg ? b : b - n_long_wgs;
        const int ti = (lg ? 0 : sp.n_long) + bb / g, grp = bb % g;
        int lo = 0, hi = VER_MAX_SLOTS;            // first index with tstart > ti
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tstart[mid] > ti) hi = mid; else lo = mid + 1;
        }
        const int j = lo - 1, k = ti - tstart[j], c = cnt[j];
        const int gbase = col_tiles / g, gextra = col_tiles % g;   // column tiles dealt to the g workgroups as evenly as they go
        const int ntl = gbase + (grp < gextra ? 1 : 0), nt0 = grp * gbase + (grp < gextra ? grp : gextra);
        tiles[b] = VerTile{j, 32 * ti, c - 32 * k < 32 ? c - 32 * k : 32, nt0 | (ntl << 8)};
    }
#pragma unroll
    for (int e = 0; e < VER_REG_ENVS; ++e)
        if (t + 1024 * e < n) rows[32 * (tstart[sv[e]] + (pv[e] >> 5)) + (pv[e] & 31)] = t + 1024 * e;
    for (long long i0 = 1024ll * VER_REG_ENVS; i0 < n; i0 += 1024) {
        if (i0 + t < n) {
            const int p = rows[32ll * (n / 32 + n_slots) + (i0 + t)];
            rows[32 * (tstart[slot[i0 + t]] + (p >> 5)) + (p & 31)] = (int)(i0 + t);
        }
    }
}

// k_pack into the actor's current direct-layout policy AND into the version slot k_version_plan picked (one launch instead of a pack and
// a 0.5 MB copy; the slots' pads stay as allocated: zero, like the current copy's)
__global__ void __launch_bounds__(256) k_pack_version(const Seg *__restrict__ segs, const float *__restrict__ src, float *__restrict__ dst,
                                                      float *__restrict__ vslab, long long vstride, const VerState *__restrict__ vs) {
    const Seg s = segs[blockIdx.y];
    float *dst2 = vslab + (long long)vs->target * vstride;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.n; i += (long long)gridDim.x * 256) {
        long long ii = i;
        if (s.cols > 0) {
            const long long k = i / s.cols, j = i - k * s.cols;
            ii = s.mode == 2 ? w1y_index(s.d0 + (int)k, (int)j) : ((k >> 2) * s.ld + j) * 4 + (k & 3);
        }
        const float v = src[s.ext + i];
        dst[s.in + ii] = v;
        dst2[s.in + ii] = v;
    }
}

// get_action from the head partials of a (versioned) forward launch — what k_env_step_pi does before it steps the physics — for callers
// that step their envs themselves (the n-step rollout: ddrl_actor_act_versioned).  One lane per row, same summation order.
struct FinishArgs {
    const float *hp;          // [8][n][16]
    const float *bmu, *bls;   // slot 0's head biases; slot s: + s * vstride (slot == nullptr: the current weights')
    const int *slot;
    long long vstride, n;
    const float *eps;         // [n][act] (nullptr when deterministic)
    float *act_out;           // [n][act]
    int act, nt2, deterministic;
    float scale;
};
__global__ void __launch_bounds__(64) k_actor_finish(FinishArgs a) {
    const long long i = (long long)blockIdx.x * 64 + threadIdx.x, n = a.n;
    if (i >= n) return;
    const int nq = (a.nt2 + 3) >> 2;
    const long long voff = a.slot ? (long long)a.slot[i] * a.vstride : 0;
    float mu[4], ls[4], ev[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float sm = 0.f, sl = 0.f;
        if (c < a.act) {
            for (int q = 0; q < nq; ++q) {   // n-tile order; slots beyond nt2 hold 0
                const float4 m4 = *reinterpret_cast<const float4 *>(a.hp + ((long long)c * n + i) * 16 + 4 * q);
                const float4 l4 = *reinterpret_cast<const float4 *>(a.hp + ((long long)(a.act + c) * n + i) * 16 + 4 * q);
                sm += m4.x; sm += m4.y; sm += m4.z; sm += m4.w;
                sl += l4.x; sl += l4.y; sl += l4.z; sl += l4.w;
            }
        }
        const int cc = c < a.act ? c : 0;
        mu[c] = sm + a.bmu[voff + cc];
        ls[c] = sl + a.bls[voff + cc];
        ev[c] = (a.deterministic || c >= a.act || !a.eps) ? 0.f : a.eps[i * a.act + c];
    }
    const ddrl_pol::PolRow pr = ddrl_pol::policy_row(mu, ls, ev, a.act, a.scale);
    for (int c = 0; c < a.act; ++c) a.act_out[i * a.act + c] = a.deterministic ? tanhf(mu[c]) * a.scale : pr.act[c];
}

__global__ void __launch_bounds__(256) k_version_adopt(int *__restrict__ slot, const uint8_t *__restrict__ ended, long long n, const VerState *vs) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && ended[i]) slot[i] = vs->newest;
}

extern "C" {

int ddrl_actor_create(ddrl_actor_t **out, int device, const ddrl_sac1_config_t *cfg, int64_t max_rows) {
    DDRL_REQUIRE(out != nullptr && max_rows > 0, "bad out/max_rows");
    int rc = check_cfg(cfg);
    if (rc != DDRL_OK) return rc;
    ddrl::DeviceGuard g(device);
    if (!g.ok) { ddrl::set_error("cannot select device %d", device); return DDRL_ERR_HIP; }
    ddrl_actor *h = new ddrl_actor();
    h->device = device; h->cfg = *cfg; h->max_rows = max_rows;
    h->L = make_layout(*cfg, true);
    h->pi_p = h->H1 = h->H2 = nullptr; h->segs_d = nullptr;
    const size_t guard = 2048;  // floats of readable memory behind every GEMM operand (unclamped tile loads)
    hipError_t e = dev_alloc(&h->pi_p, (size_t)h->L.total_int + guard);
    h->ldh1 = (int)pad4(cfg->hidden1); h->ldh2 = (int)pad4(cfg->hidden2);
    if (e == hipSuccess) e = dev_alloc(&h->H1, (size_t)max_rows * h->ldh1 + guard);
    if (e == hipSuccess) e = dev_alloc(&h->H2, (size_t)max_rows * h->ldh2 + guard);
    if (e == hipSuccess) e = dev_alloc(&h->segs_d, h->L.segs.size());
    if (e != hipSuccess) {
        ddrl::set_error("hipMalloc failed in ddrl_actor_create: %s", hipGetErrorString(e));
        ddrl_actor_destroy(h);
        return DDRL_ERR_NOMEM;
    }
    DDRL_HIP_CHECK(hipMemcpy(h->segs_d, h->L.segs.data(), h->L.segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
    {
        ddrl_sac1_config_t c2 = *cfg;
        c2.batch = 32;
        h->direct = direct_ok(c2) && max_rows % 32 == 0 && max_rows <= 32 * 4095 && cfg->obs_dim + 1 <= 13;
        h->dslab = nullptr; h->segs_dd = nullptr;
        h->n_slots = 0; h->vslab = nullptr; h->slot_d = h->perm_d = nullptr; h->vtiles_d = nullptr; h->vs_d = nullptr; h->vcnt_d = nullptr; h->act1_part = nullptr;
        h->plan_fresh = false; h->pi_p_stale = false; h->flat_tmp = nullptr;
        h->steps_since_install = 1ll << 40;
    }
    if (h->direct) {
        h->Ld = make_layout(*cfg, true, true);
        const size_t np = ((size_t)h->Ld.total_int + 2048 + 63) & ~(size_t)63, no = ((size_t)max_rows * cfg->obs_dim + 63) & ~(size_t)63;
        const size_t nh = (size_t)DFH * max_rows * DNT;
        e = dev_alloc(&h->dslab, np + no + nh + 2048);
        if (e == hipSuccess) e = dev_alloc(&h->segs_dd, h->Ld.segs.size());
        if (e == hipSuccess) e = dev_alloc(&h->flat_tmp, (size_t)h->L.total_ext + 64);
        if (e != hipSuccess) {
            ddrl::set_error("hipMalloc failed in ddrl_actor_create: %s", hipGetErrorString(e));
            ddrl_actor_destroy(h);
            return DDRL_ERR_NOMEM;
        }
        h->pi_d = h->dslab; h->obs_d = h->dslab + np; h->hp_d = h->obs_d + no;
        DDRL_HIP_CHECK(hipMemcpy(h->segs_dd, h->Ld.segs.data(), h->Ld.segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
    }
    *out = h;
    return DDRL_OK;
}

int ddrl_actor_destroy(ddrl_actor_t *h) {
    if (!h) return DDRL_OK;
    ddrl::DeviceGuard g(h->device);
    (void)hipFree(h->dslab); (void)hipFree(h->segs_dd); (void)hipFree(h->flat_tmp);
    (void)hipFree(h->vslab); (void)hipFree(h->slot_d); (void)hipFree(h->perm_d); (void)hipFree(h->vtiles_d); (void)hipFree(h->vs_d); (void)hipFree(h->vcnt_d); (void)hipFree(h->act1_part);
    (void)hipFree(h->pi_p); (void)hipFree(h->H1); (void)hipFree(h->H2); (void)hipFree(h->segs_d);
    delete h;
    return DDRL_OK;
}

int ddrl_actor_set_weights(ddrl_actor_t *h, const float *flat_pi_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_pi_d != nullptr, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    if (!h->direct) {
        k_pack<<<dim3(64, (unsigned)h->L.segs.size()), 256, 0, s>>>(h->segs_d, flat_pi_d, h->pi_p, nullptr, 1);
    } else if (h->n_slots > 0) {
        // version store: the new weights become the newest version, in a slot no env acts on — ONE planning launch (slot pick + the tile
        // table of the next forward) and ONE pack into the current copy and that slot (round 4: two packs, the pick, a 0.5 MB copy
        // and, in the step, the grouping: five launches)
        k_version_plan<<<1, 1024, 0, s>>>(h->slot_d, h->max_rows, h->n_slots, h->perm_d, h->vtiles_d, h->vs_d, 1, (h->cfg.hidden2 + 31) / 32, h->wg_slots, h->vt_cap);
        k_pack_version<<<dim3(64, (unsigned)h->Ld.segs.size()), 256, 0, s>>>(h->segs_dd, flat_pi_d, h->pi_d, h->vslab, h->vstride, h->vs_d);
        h->plan_fresh = true;
        h->steps_since_install = 0;
        h->pi_p_stale = true;
    } else {
        k_pack<<<dim3(64, (unsigned)h->Ld.segs.size()), 256, 0, s>>>(h->segs_dd, flat_pi_d, h->pi_d, nullptr, 1);
        h->pi_p_stale = true;   // the row-major copy only serves ddrl_actor_act / get_weights: rebuilt there
    }
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

// direct actors: the row-major policy (generic kernels of ddrl_actor_act, ddrl_actor_get_weights) from the direct-layout one, on demand
static void actor_refresh_row_major(ddrl_actor *h, hipStream_t s) {
    if (!h->pi_p_stale) return;
    k_pack<<<dim3(64, (unsigned)h->Ld.segs.size()), 256, 0, s>>>(h->segs_dd, h->pi_d, h->flat_tmp, nullptr, 0);
    k_pack<<<dim3(64, (unsigned)h->L.segs.size()), 256, 0, s>>>(h->segs_d, h->flat_tmp, h->pi_p, nullptr, 1);
    h->pi_p_stale = false;
}

int ddrl_actor_versions_enable(ddrl_actor_t *h, int32_t n_slots, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    DDRL_REQUIRE(h->direct, "version store needs the direct-operand policy (shape outside the envelope)");
    DDRL_REQUIRE(n_slots >= 2 && n_slots <= VER_MAX_SLOTS, "n_slots outside [2, 2048]");
    DDRL_REQUIRE(h->n_slots == 0, "version store already enabled");
    ddrl::DeviceGuard g(h->device);
    const long long np = ((long long)h->Ld.total_int + 2048 + 63) & ~63ll;   // as the policy part of dslab (incl. the guard)
    hipError_t e = dev_alloc(&h->vslab, (size_t)np * n_slots + 2048);
    if (e == hipSuccess) e = dev_alloc(&h->slot_d, (size_t)h->max_rows + 64);
    // row lists: the planning launch's dense [tiles][32] (+ its parking space), then the env-step launch's own [n_slots][max_rows]
    const long long perm_dense = 32ll * (h->max_rows / 32 + n_slots) + h->max_rows + 64;
    if (e == hipSuccess) e = dev_alloc(&h->perm_d, (size_t)perm_dense + (size_t)n_slots * (size_t)h->max_rows);
    if (e == hipSuccess) e = dev_alloc(&h->vcnt_d, (size_t)VER_MAX_SLOTS);
    int ncu = 256;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    }
    // the forward's workgroup table: the coarsest split of the worst-case tile count (every live version leaves one partial tile) plus
    // one round of short workgroups (ver_split)
    const int col_tiles = (h->cfg.hidden2 + 31) / 32, g2 = col_tiles < 2 ? 1 : ((col_tiles + ANT - 1) / ANT < 2 ? 2 : (col_tiles + ANT - 1) / ANT);
    const long long vt_worst = h->max_rows / 32 + (n_slots < h->max_rows ? n_slots : h->max_rows);
    const long long vt_cap = vt_worst * g2 + 2 * ncu;
    if (e == hipSuccess) e = dev_alloc(&h->vtiles_d, (size_t)vt_cap + 16);
    if (e == hipSuccess) e = dev_alloc(&h->vs_d, 2);
    if (e != hipSuccess) {
        ddrl::set_error("hipMalloc failed in ddrl_actor_versions_enable (%d slots of %lld floats): %s", n_slots, np, hipGetErrorString(e));
        return DDRL_ERR_NOMEM;
    }
    hipStream_t s = ddrl::as_stream(stream);
    // (dev_alloc zero-fills: every env on slot 0, newest = 0) slot 0 = the weights the actor holds now
    h->vstride = np;
    h->n_slots = n_slots;
    h->wg_slots = 2 * ncu;
    if (const char *ev = getenv("DDRL_VER_WG_SLOTS")) {   // tests: a small "chip", so that small env counts walk the multi-round splits
        const int v = atoi(ev);
        if (v >= 1 && v <= 2 * ncu) h->wg_slots = v;
    }
    h->vt_cap = (int)vt_cap;
    h->perm2d_off = perm_dense;
    k_version_copy<<<256, 256, 0, s>>>(h->pi_d, h->vslab, h->vstride, h->vs_d, h->vstride / 4);
    h->steps_since_install = 1ll << 40;
    h->plan_fresh = false;
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_actor_versions_state(ddrl_actor_t *h, int32_t *slot_of_env_d, int32_t *state_h, void *stream) {
    DDRL_REQUIRE(h != nullptr && h->n_slots > 0, "version store not enabled");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    if (slot_of_env_d) DDRL_HIP_CHECK(hipMemcpyAsync(slot_of_env_d, h->slot_d, (size_t)h->max_rows * sizeof(int), hipMemcpyDeviceToDevice, s));
    if (state_h) {
        VerState vs;
        DDRL_HIP_CHECK(hipMemcpyAsync(&vs, h->vs_d, sizeof(vs), hipMemcpyDeviceToHost, s));
        DDRL_HIP_CHECK(hipStreamSynchronize(s));
        state_h[0] = vs.newest; state_h[1] = vs.live; state_h[2] = vs.n_tiles; state_h[3] = vs.err;
    }
    return DDRL_OK;
}

int ddrl_actor_act_versioned(ddrl_actor_t *h, const float *obs_d, const float *eps_d, int64_t n, int deterministic, int32_t horizon_steps,
                             float *act_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && obs_d != nullptr && act_d != nullptr, "NULL pointer");
    DDRL_REQUIRE(h->n_slots > 0 && n == h->max_rows, "version store not enabled, or n != max_rows");
    DDRL_REQUIRE(deterministic || eps_d != nullptr, "eps is required for stochastic actions");
    DDRL_REQUIRE(h->cfg.act_dim <= 4 && horizon_steps >= 1, "act_dim <= 4 and horizon_steps >= 1");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    const ddrl_sac1_config_t &c = h->cfg;
    DDRL_HIP_CHECK(hipMemcpyAsync(h->obs_d, obs_d, (size_t)n * c.obs_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    // `horizon_steps` after the last set_weights every env has been through an episode end: one version, the plain launch
    const bool versioned = h->steps_since_install < (long long)horizon_steps;
    h->steps_since_install += 1;
    const int rc = ddrl_actor_internal_forward(h, n, stream, versioned ? 1 : 0);
    if (rc != DDRL_OK) return rc;
    FinishArgs f{};
    f.hp = h->hp_d; f.n = n; f.eps = deterministic ? nullptr : eps_d; f.act_out = act_d; f.act = c.act_dim; f.nt2 = (c.hidden2 + 31) / 32;
    f.deterministic = deterministic; f.scale = (float)c.act_scale;
    f.slot = versioned ? h->slot_d : nullptr; f.vstride = h->vstride;
    f.bmu = (versioned ? h->vslab : h->pi_d) + h->Ld.pi_bmu; f.bls = (versioned ? h->vslab : h->pi_d) + h->Ld.pi_bls;
    k_actor_finish<<<(unsigned)((n + 63) / 64), 64, 0, s>>>(f);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_actor_versions_adopt(ddrl_actor_t *h, const uint8_t *ended_d, int64_t n, void *stream) {
    DDRL_REQUIRE(h != nullptr && h->n_slots > 0 && ended_d != nullptr, "version store not enabled / NULL mask");
    DDRL_REQUIRE(n > 0 && n <= h->max_rows, "n outside [1, max_rows]");
    ddrl::DeviceGuard g(h->device);
    k_version_adopt<<<(unsigned)((n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(h->slot_d, ended_d, n, h->vs_d);
    h->plan_fresh = false;
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_actor_get_weights(ddrl_actor_t *h, float *flat_pi_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_pi_d != nullptr, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    if (h->direct) {   // straight from the direct-layout policy
        k_pack<<<dim3(64, (unsigned)h->Ld.segs.size()), 256, 0, ddrl::as_stream(stream)>>>(h->segs_dd, h->pi_d, flat_pi_d, nullptr, 0);
        DDRL_LAUNCH_CHECK();
        return DDRL_OK;
    }
    k_pack<<<dim3(64, (unsigned)h->L.segs.size()), 256, 0, ddrl::as_stream(stream)>>>(h->segs_d, h->pi_p, flat_pi_d, nullptr, 0);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_actor_act(ddrl_actor_t *h, const float *obs_d, const float *eps_d, int64_t n, int deterministic, float *act_d,
                   void *stream) {
    DDRL_REQUIRE(h != nullptr && obs_d != nullptr && act_d != nullptr, "NULL pointer");
    DDRL_REQUIRE(n > 0 && n <= h->max_rows, "n outside [1, max_rows]");
    DDRL_REQUIRE(deterministic || eps_d != nullptr, "eps is required for stochastic actions");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    const ddrl_sac1_config_t &c = h->cfg;
    const Layout &L = h->L;
    actor_refresh_row_major(h, s);
    L1Jobs l1{};
    l1.njobs = 1;
    l1.job[0] = L1Job{obs_d, nullptr, h->pi_p + L.pi_W1, h->pi_p + L.pi_b1, h->H1, nullptr, c.obs_dim, 0, (int)n, c.hidden1, h->ldh1, 0, 0};
    k_l1<<<dim3((c.hidden1 + 255) / 256, (unsigned)((n + L1_ROWS - 1) / L1_ROWS), 1), 256, 0, s>>>(l1);
    GemmJobs gj{};
    gemm_add(gj, gemm_fwd(h->H1, h->ldh1, h->pi_p + L.pi_W2, h->pi_p + L.pi_b2, h->H2, h->ldh2, (int)n, c.hidden1, c.hidden2));
    launch_gemm(gj, s);
    ActArgs aa{h->H2, net_pi(h->pi_p, L), eps_d, act_d, (int)n, c.hidden2, h->ldh2, c.act_dim, deterministic, (float)c.act_scale};
    k_rows_act<<<(unsigned)((n + 3) / 4), 256, 0, s>>>(aa);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_actor_act_one(ddrl_actor_t *h, const float *obs_d, uint32_t noise_seed, uint64_t noise_ctr, int deterministic, float *act_d,
                       void *stream) {
    DDRL_REQUIRE(h != nullptr && obs_d != nullptr && act_d != nullptr, "NULL pointer");
    const ddrl_sac1_config_t &c = h->cfg;
    if (c.obs_dim > 64 || c.hidden1 > 512 || c.hidden2 > 512 || c.act_dim > 4) {
        ddrl::set_error("ddrl_actor_act_one: shape outside (obs 64, hidden 512, act 4): use ddrl_actor_act");
        return DDRL_ERR_UNSUPPORTED;
    }
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    if (!h->act1_part) {   // head partials of up to 32 workgroups + the ticket
        hipError_t e = dev_alloc(&h->act1_part, (size_t)32 * 16 + 16);
        if (e != hipSuccess) { ddrl::set_error("hipMalloc failed in ddrl_actor_act_one: %s", hipGetErrorString(e)); return DDRL_ERR_NOMEM; }
    }
    actor_refresh_row_major(h, s);
    ActOneArgs a{obs_d, net_pi(h->pi_p, h->L), act_d, h->act1_part, reinterpret_cast<unsigned int *>(h->act1_part + 32 * 16),
                 c.obs_dim, c.hidden1, c.hidden2, c.act_dim, deterministic, (float)c.act_scale, noise_seed, (unsigned long long)noise_ctr};
    k_act_one<<<(unsigned)((c.hidden2 + A1_COLS - 1) / A1_COLS), 256, 0, s>>>(a);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

}  // extern "C"

// ---- internal (env.hip: ddrl_rollout_step) ------------------------------------------------------------------------
// The policy forward of the fused rollout step: layer 1 + layer 2 + head partials of `n` observation rows that sit in
// the actor's own observation buffer, as one k_dfwd launch (n / 32 x ceil(h2 / 32) tiles).  The env-step kernel behind it
// turns the partials into actions (ddrl_pol::policy_row), steps the physics and appends the transitions to the ring.
#ifdef DDRL_STAMPS
unsigned long long *g_actor_st = nullptr;
#endif
int ddrl_actor_internal_forward(ddrl_actor *h, long long n, void *stream, int versioned) {
    DDRL_REQUIRE(h != nullptr && h->direct, "actor has no direct-operand policy (shape outside the envelope)");
    DDRL_REQUIRE(n > 0 && n % 32 == 0 && n <= h->max_rows, "n must be a positive multiple of 32 within max_rows");
    ddrl::DeviceGuard g(h->device);
    const ddrl_sac1_config_t &c = h->cfg;
    const Layout &L = h->Ld;
    const int nt2 = (c.hidden2 + 31) / 32;
    ActFwdArgs A{};
    A.W1 = h->pi_d + L.pi_W1; A.W2p = A.W1 + ((c.hidden1 + 31) & ~31) * 16;   // the k4-interleaved W2 follows the layer-1 block array
    A.b2 = h->pi_d + L.pi_b2; A.wmu = h->pi_d + L.pi_Wmu; A.wls = h->pi_d + L.pi_Wls; A.obs = h->obs_d; A.hp = h->hp_d;
    A.rows = (int)n; A.K = c.hidden1; A.Np = L.Np2; A.h2 = c.hidden2; A.d0 = c.obs_dim; A.act = c.act_dim; A.tiles_n = nt2;
    A.ngroups = (nt2 + ANT - 1) / ANT;   // 10 column tiles -> 2 workgroups of 5 per row tile
#ifdef DDRL_STAMPS
    A.st = g_actor_st;
#endif
    const int D1 = c.obs_dim + 1, ns = D1 <= 8 ? 4 : 4 + (D1 - 8 + 1) / 2;
    hipStream_t s = ddrl::as_stream(stream);
    if (versioned) {
        // envs grouped by the policy version they act on; a row tile = up to 32 envs of one version.  The launch covers the worst
        // case (every live version leaves one partial tile), surplus workgroups leave at once.
        DDRL_REQUIRE(h->n_slots > 0 && n == h->max_rows, "versioned forward: store not enabled, or n != max_rows");
        if (!h->plan_fresh) k_version_plan<<<1, 1024, 0, s>>>(h->slot_d, n, h->n_slots, h->perm_d, h->vtiles_d, h->vs_d, 0, nt2, h->wg_slots, h->vt_cap);
        h->plan_fresh = true;   // (ddrl_rollout_step clears it behind the env-step launch, ddrl_actor_versions_adopt behind its own)
        A.W1 = h->vslab + L.pi_W1; A.W2p = A.W1 + ((c.hidden1 + 31) & ~31) * 16;
        A.b2 = h->vslab + L.pi_b2; A.wmu = h->vslab + L.pi_Wmu; A.wls = h->vslab + L.pi_Wls;
        A.vtiles = h->vtiles_d; A.perm = h->perm_d; A.vs = h->vs_d; A.vstride = h->vstride; A.vt_max = h->vt_cap;
        const unsigned vgrid = (unsigned)h->vt_cap;   // (the plan's workgroup count is at most that; the rest leave at once)
        if (ns == 4) k_actor_fwd<4, 2, true><<<vgrid, 256, 0, s>>>(A);
        else if (ns == 5) k_actor_fwd<5, 2, true><<<vgrid, 256, 0, s>>>(A);
        else if (ns == 6) k_actor_fwd<6, 2, true><<<vgrid, 256, 0, s>>>(A);
        else k_actor_fwd<7, 2, true><<<vgrid, 256, 0, s>>>(A);
        DDRL_LAUNCH_CHECK();
        return DDRL_OK;
    }
    const unsigned grid = (unsigned)(n / 32) * A.ngroups;
    if (grid > 256) {   // more than one workgroup per CU: the two-per-CU register budget
        if (ns == 4) k_actor_fwd<4, 2><<<grid, 256, 0, s>>>(A);
        else if (ns == 5) k_actor_fwd<5, 2><<<grid, 256, 0, s>>>(A);
        else if (ns == 6) k_actor_fwd<6, 2><<<grid, 256, 0, s>>>(A);
        else k_actor_fwd<7, 2><<<grid, 256, 0, s>>>(A);
    } else {
        if (ns == 4) k_actor_fwd<4, 1><<<grid, 256, 0, s>>>(A);
        else if (ns == 5) k_actor_fwd<5, 1><<<grid, 256, 0, s>>>(A);
        else if (ns == 6) k_actor_fwd<6, 1><<<grid, 256, 0, s>>>(A);
        else k_actor_fwd<7, 1><<<grid, 256, 0, s>>>(A);
    }
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

ddrl_actor_rollout_view ddrl_actor_internal_view(ddrl_actor *h) {
    ddrl_actor_rollout_view v{};
    if (!h || !h->direct) return v;
    v.ok = 1; v.obs = h->obs_d; v.hp = h->hp_d; v.bmu = h->pi_d + h->Ld.pi_bmu; v.bls = h->pi_d + h->Ld.pi_bls;
    v.n_slots = h->n_slots; v.slot = h->slot_d; v.vs = h->vs_d; v.vstride = h->vstride;
    v.vbmu = h->n_slots ? h->vslab + h->Ld.pi_bmu : nullptr; v.vbls = h->n_slots ? h->vslab + h->Ld.pi_bls : nullptr;
    v.steps_since_install = &h->steps_since_install;
    v.plan_fresh = &h->plan_fresh;
    v.vcnt = h->vcnt_d; v.perm = h->perm_d; v.perm2d_off = h->perm2d_off; v.vtiles = h->vtiles_d; v.vt_cap = h->vt_cap; v.wg_slots = h->wg_slots;
    v.obs_dim = h->cfg.obs_dim; v.act = h->cfg.act_dim; v.nt2 = (h->cfg.hidden2 + 31) / 32; v.max_rows = h->max_rows;
    v.scale = (float)h->cfg.act_scale; v.device = h->device;
    return v;
}
