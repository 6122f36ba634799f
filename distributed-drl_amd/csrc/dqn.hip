// Discrete-action learners on the shared GEMM core: Double-DQN (algos/dqn) and soft-Q (algos/sqn).
#include <mutex>
#include "gemm_core.h"
#include "wide_l1.h"

// ==========================================================================================
// Double-DQN learner (algos/dqn/actor_learner.py:19-107 on algos/dqn/core.py:40-50):
// q = mlp(obs -> h1 -> h2 -> n_actions), q_x2 = the same variables at obs2, q_next = target(obs2);
// q_target = q_next[argmax q_x2]; q_loss = 0.5 mean((r + gamma (1-d) q_target - q[a])^2);
// one Adam over main/q1, polyak over all.  Layers 1 and 2 (obs_dim is arbitrary here) are jobs of the generic MFMA GEMM kernel,
// the head (its forward for every evaluation, the reference's row logic, its dgrad) is ONE launch (k_dqn_head), Adam + polyak one
// flat launch that also finishes the loss mean: 6 launches per update (+ a staging launch when the rows cannot be read in place).
// Wide observations (obs_dim >= 1024, config 5's 28 224) take layer 1 through the tiled kernels of wide_l1.h instead (forward split
// over K + reduce; the weight gradient as equal shares of the stage sequence, k_wide_sk), reading the caller's observation rows in place.
// variant DDRL_SQN = the soft-Q learner of algos/sqn/actor_learner.py:19-78 on algos/sqn/core.py:30-79:
// twin networks main/q1, main/q2; evaluations q1(x), q1(x2) (its softmax policy's sum p log p at x2),
// q2(x) and the targets q1_(x2), q2_(x2); v_backup = min(max q1_, max q2_) - alpha * sum p log p;
// q_loss = 0.5 mean((q_backup - q1[a])^2) + 0.5 mean((q_backup - q2[a])^2); one Adam over main/q1, main/q2.
// ==========================================================================================
namespace {

// ------------------------------------------------------------------------------------------
// k_dqn_head: everything between the layer-2 forward and the layer-2 backward in ONE launch (was: head forward as three
// k_gemm jobs, the rows kernel, the head dgrad as a k_gemm job — 31 us of launch-bound kernels at batch 512):
//   Q[ev] = [H2[ev] | 1] [W3 ; b3]              every evaluation's n_actions outputs (K = hidden2 + 1: the bias is the last row)
//   DDQN rows (actor_learner.py:40-55)           a_one_hot, q_value, argmax of the online q(x2), q_target, q_backup, q_loss, dQ
//   SQN rows (sqn/actor_learner.py:40-58)        log-softmax entropy of q1(x2), min of the targets' greedy values, q_backup, dQ1, dQ2
//   dZ2[n] = (dQ[n] W3[n]^T) .* (H2[gev[n]] > 0)
// One workgroup = HEAD_ROWS batch rows; the loss is summed over the workgroups' partials by the last one to finish, in
// workgroup order (deterministic).  The head kernels' gradient [H2 | 1]^T dQ stays a k_gemm job (now of the layer-2 backward launch).
// ------------------------------------------------------------------------------------------
constexpr int HEAD_ROWS = 4, HEAD_MAXA = 16, HEAD_MAXEV = 5;
struct HeadArgs {
    const float *H2;                 // [nev][B][ldh2], ones column at h2
    const float *W3b[HEAD_MAXEV];    // [W3 ; b3] of evaluation ev: (h2 + 1) rows of A
    const float *W3n[2];             // ... of the differentiated network n (main)
    const float *acts, *rew, *done;
    float *Q, *dQ, *dZ2, *part, *qsel;
    int *zero_words;                 // polled words of a LATER launch of this update (k_wide_sk's flags), zeroed here: n_zero ints
    int n_zero;
    int B, A, ldq, h2, ldh2, nev, nnet, sqn;
    int gev[2], nev_of_net[2];       // evaluation whose staged [W3 ; b3] is network n's main copy
    int ldw;                         // floats per staged [W3 ; b3]: (h2 + 1) * A rounded up to 4
    float gamma, alpha;
};
static size_t head_lds_bytes(const HeadArgs &a) { return ((size_t)a.nev * HEAD_ROWS * a.ldh2 + (size_t)a.nev * a.ldw) * sizeof(float); }
// the Q phase of k_dqn_head for four actions and NEV evaluations (Double-DQN: 3, SQN: 5): one wave = one batch row, the lanes
// split k; every evaluation's loads are issued together and every sum goes through the butterfly together (a cross-lane exchange
// is an LDS round trip: one chain per (evaluation, action) after the other was 5.3 us of the kernel)
template <int NEV>
__device__ __forceinline__ void head_q4(const HeadArgs &a, const float *sH, const float *sW, float *sQ, int lr, int row, int lane, int K, long long BQ) {
    float acc[NEV][4];
#pragma unroll
    for (int ev = 0; ev < NEV; ++ev) acc[ev][0] = acc[ev][1] = acc[ev][2] = acc[ev][3] = 0.f;
    for (int k = lane; k < K; k += 64) {
        float xv[NEV];
        float4 wv[NEV];
#pragma unroll
        for (int ev = 0; ev < NEV; ++ev) {
            xv[ev] = sH[((size_t)ev * HEAD_ROWS + lr) * a.ldh2 + k];
            wv[ev] = *reinterpret_cast<const float4 *>(sW + (size_t)ev * a.ldw + 4 * k);
        }
#pragma unroll
        for (int ev = 0; ev < NEV; ++ev) {
            acc[ev][0] = fmaf(xv[ev], wv[ev].x, acc[ev][0]); acc[ev][1] = fmaf(xv[ev], wv[ev].y, acc[ev][1]);
            acc[ev][2] = fmaf(xv[ev], wv[ev].z, acc[ev][2]); acc[ev][3] = fmaf(xv[ev], wv[ev].w, acc[ev][3]);
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
        for (int ev = 0; ev < NEV; ++ev)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[ev][c] += __shfl_xor(acc[ev][c], o);
    }
    if (lane == 0) {
#pragma unroll
        for (int ev = 0; ev < NEV; ++ev) {
#pragma unroll
            for (int c = 0; c < 4; ++c) sQ[(ev * HEAD_ROWS + lr) * HEAD_MAXA + c] = acc[ev][c];
            if (row < a.B) *reinterpret_cast<float4 *>(a.Q + ev * BQ + (long long)row * a.ldq) = make_float4(acc[ev][0], acc[ev][1], acc[ev][2], acc[ev][3]);
        }
    }
}
__global__ void __launch_bounds__(256) k_dqn_head(HeadArgs a) {
    // the workgroup's H2 rows and every evaluation's [W3 ; b3] are staged into LDS with ALL loads in flight at once (the first cut
    // read them inside the dot-product loops: 30 dependent round trips per wave, 30 us for 3.7 MFLOP)
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    float *sH = hsm;                                        // [nev][HEAD_ROWS][ldh2]
    float *sW = hsm + (size_t)a.nev * HEAD_ROWS * a.ldh2;   // [nev][ldw]
    __shared__ float sQ[HEAD_MAXEV][HEAD_ROWS][HEAD_MAXA];
    __shared__ float sdQ[2][HEAD_ROWS][HEAD_MAXA];
    __shared__ float s_loss[HEAD_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r0 = blockIdx.x * HEAD_ROWS;
    const long long BH = (long long)a.B * a.ldh2, BQ = (long long)a.B * a.ldq;
    const int K = a.h2 + 1;
    // the row scalars the "rows" phase needs, requested now (a cold round trip otherwise sits between two barriers)
    float p_act = 0.f, p_rew = 0.f, p_done = 0.f;
    if (tid < HEAD_ROWS && r0 + tid < a.B) { p_act = a.acts[r0 + tid]; p_rew = a.rew[r0 + tid]; p_done = a.done[r0 + tid]; }
    {
        // four loads per lane in flight per round (a plain load-store loop waits for every load before it issues the next one)
        const int l4 = a.ldh2 >> 2, n4 = a.nev * HEAD_ROWS * l4;
        for (int e0 = tid; e0 < n4; e0 += 4 * 256) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 256 * u;
                const int ec = e < n4 ? e : 0;
                const int ev = ec / (HEAD_ROWS * l4), rem = ec - ev * (HEAD_ROWS * l4), lr = rem / l4, c4 = rem - lr * l4;
                const int row = r0 + lr < a.B ? r0 + lr : a.B - 1;
                v[u] = *reinterpret_cast<const float4 *>(a.H2 + ev * BH + (long long)row * a.ldh2 + 4 * c4);
                if (r0 + lr >= a.B) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + 256 * u < n4) reinterpret_cast<float4 *>(sH)[e0 + 256 * u] = v[u];
        }
        const int nw = K * a.A, nw4 = (nw + 3) / 4;
        for (int ev = 0; ev < a.nev; ++ev) {
            const float *src = a.W3b[ev];
            if ((((unsigned long long)src) & 15ull) == 0) {   // (reads up to three floats past the bias: inside the parameter slab)
                for (int e0 = tid; e0 < nw4; e0 += 4 * 256) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4 *>(src)[e0 + 256 * u < nw4 ? e0 + 256 * u : 0];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (e0 + 256 * u < nw4) reinterpret_cast<float4 *>(sW + (size_t)ev * a.ldw)[e0 + 256 * u] = v[u];
                }
            } else {
                for (int e = tid; e < nw; e += 256) sW[(size_t)ev * a.ldw + e] = src[e];
            }
        }
    }
    __syncthreads();
    // ---- Q: wave w takes row w; the lanes split k, fixed-order butterfly
    {
        const int lr = w, row = r0 + lr;
        if (a.A == 4 && a.nev == 3) head_q4<3>(a, sH, sW, &sQ[0][0][0], lr, row, lane, K, BQ);
        else if (a.A == 4 && a.nev == 5) head_q4<5>(a, sH, sW, &sQ[0][0][0], lr, row, lane, K, BQ);
        else {
            for (int ev = 0; ev < a.nev; ++ev) {
                float acc[HEAD_MAXA];
#pragma unroll
                for (int c = 0; c < HEAD_MAXA; ++c) acc[c] = 0.f;
                const float *x = sH + ((size_t)ev * HEAD_ROWS + lr) * a.ldh2;
                const float *wk = sW + (size_t)ev * a.ldw;
                for (int k = lane; k < K; k += 64) {
                    const float xv = x[k];
#pragma unroll
                    for (int c = 0; c < HEAD_MAXA; ++c)
                        if (c < a.A) acc[c] = fmaf(xv, wk[k * a.A + c], acc[c]);
                }
                for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
                    for (int c = 0; c < HEAD_MAXA; ++c)
                        if (c < a.A) acc[c] += __shfl_xor(acc[c], o);
                }
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < HEAD_MAXA; ++c)
                        if (c < a.A) {
                            sQ[ev][lr][c] = acc[c];
                            if (row < a.B) a.Q[ev * BQ + (long long)row * a.ldq + c] = acc[c];
                        }
                }
            }
        }
    }
    __syncthreads();
    // ---- rows
    if (tid < HEAD_ROWS) {
        const int row = r0 + tid;
        float l = 0.f;
        if (row < a.B) {
            const int act = (int)p_act;                             // tf.cast(a_ph, tf.int32)
            const bool valid = act >= 0 && act < a.A;               // one_hot of an out-of-range index is all zeros
            if (!a.sqn) {
                const float *q = sQ[0][tid], *qx2 = sQ[1][tid], *qn = sQ[2][tid];
                int best = 0;
                float bv = qx2[0];
                for (int c = 1; c < a.A; ++c) { const float v = qx2[c]; if (v > bv) { bv = v; best = c; } }  // tf.argmax: first maximum
                const float q_value = valid ? q[act] : 0.f;
                const float backup = p_rew + (a.gamma * (1.0f - p_done)) * qn[best];
                const float e = backup - q_value;
                l = e * e;
                const float g = -e / (float)a.B;
                for (int c = 0; c < a.ldq; ++c) {
                    const float v = (valid && c == act) ? g : 0.f;
                    if (c < HEAD_MAXA) sdQ[0][tid][c] = v;
                    a.dQ[(long long)row * a.ldq + c] = v;
                }
                if (a.qsel) a.qsel[row] = q_value;
            } else {
                const float *q1 = sQ[0][tid], *q1x2 = sQ[1][tid], *q2 = sQ[2][tid], *q1t = sQ[3][tid], *q2t = sQ[4][tid];
                // pi_log = log_softmax(q1(x2) / alpha); "entropy_x2" = sum exp(pi_log) * pi_log  (core.py:32-42)
                float zmax = q1x2[0] / a.alpha;
                for (int c = 1; c < a.A; ++c) zmax = fmaxf(zmax, q1x2[c] / a.alpha);
                float se = 0.f;
                for (int c = 0; c < a.A; ++c) se += expf(q1x2[c] / a.alpha - zmax);
                const float lse = logf(se);
                float plogp = 0.f;
                for (int c = 0; c < a.A; ++c) {
                    const float pl = (q1x2[c] / a.alpha - zmax) - lse;
                    plogp += expf(pl) * pl;
                }
                float m1 = q1t[0], m2 = q2t[0];   // q_mu_ = q_[argmax q_] = max q_ (each target network's own greedy value)
                for (int c = 1; c < a.A; ++c) { m1 = fmaxf(m1, q1t[c]); m2 = fmaxf(m2, q2t[c]); }
                const float v_backup = fminf(m1, m2) - a.alpha * plogp;                 // actor_learner.py:47-50
                const float q_backup = p_rew + (a.gamma * (1.0f - p_done)) * v_backup;
                const float e1 = q_backup - (valid ? q1[act] : 0.f), e2 = q_backup - (valid ? q2[act] : 0.f);
                l = e1 * e1 + e2 * e2;
                const float g1 = -e1 / (float)a.B, g2 = -e2 / (float)a.B;
                for (int c = 0; c < a.ldq; ++c) {
                    const bool hit = valid && c == act;
                    if (c < HEAD_MAXA) { sdQ[0][tid][c] = hit ? g1 : 0.f; sdQ[1][tid][c] = hit ? g2 : 0.f; }
                    a.dQ[(long long)row * a.ldq + c] = hit ? g1 : 0.f;
                    a.dQ[BQ + (long long)row * a.ldq + c] = hit ? g2 : 0.f;
                }
            }
        }
        s_loss[tid] = l;
    }
    __syncthreads();
    // ---- dZ2
    for (int nr = 0; nr < a.nnet * HEAD_ROWS; ++nr) {
        const int n = nr / HEAD_ROWS, lr = nr % HEAD_ROWS, row = r0 + lr;
        if (row >= a.B) continue;
        const float *wn = sW + (size_t)a.nev_of_net[n] * a.ldw;
        const float *hrow = sH + ((size_t)a.gev[n] * HEAD_ROWS + lr) * a.ldh2;
        float *out = a.dZ2 + ((long long)n * a.B + row) * a.h2;
        for (int j = tid; j < a.h2; j += 256) {
            const float *wj = wn + j * a.A;
            float v = 0.f;
            for (int c = 0; c < a.A; ++c) v = fmaf(sdQ[n][lr][c], wj[c], v);
            out[j] = hrow[j] > 0.f ? v : 0.f;
        }
    }
    if (a.zero_words)
        for (int i = blockIdx.x * 256 + tid; i < a.n_zero; i += gridDim.x * 256) a.zero_words[i] = 0;
    // ---- loss: this workgroup's rows in order; the flat Adam launch at the end of the update sums the partials in workgroup order
    // (an in-kernel last-arriver needs an L2 write-back per workgroup on this multi-XCD part: 19 us for the whole kernel)
    if (tid == 0) {
        float p = 0.f;
        for (int r = 0; r < HEAD_ROWS; ++r) p += s_loss[r];
        a.part[blockIdx.x] = p;
    }
}

// acts / rews / done of the sampled rows (ddrl_dqn_step_ring: the observation rows stay in the ring)
__global__ void __launch_bounds__(256) k_dqn_gather3(const float *__restrict__ ra, const float *__restrict__ rr, const float *__restrict__ rd,
                                                     const long long *__restrict__ idx, float *acts, float *rew, float *done, long long *idx_out, int B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) {
        const long long r = idx[i];
        acts[i] = ra[r]; rew[i] = rr[r]; done[i] = rd[r];
        if (idx_out) idx_out[i] = r;
    }
}

// rows ridx[b] of a ring array -> the staged image x1 [B][ldx] (its ones column is physical and untouched): four 16-byte loads in flight per lane
__global__ void __launch_bounds__(256) k_dqn_gather_rows(const float *__restrict__ ring, const long long *__restrict__ ridx, float *__restrict__ x1, int width, int ldx) {
    const float4 *s4 = reinterpret_cast<const float4 *>(ring + ridx[blockIdx.x] * (long long)width);
    float4 *d4 = reinterpret_cast<float4 *>(x1 + (long long)blockIdx.x * ldx);
    const int w4 = width >> 2, tid = threadIdx.x;
    int e = tid;
    for (; e + 3 * 256 < w4; e += 4 * 256) {
        const float4 a0 = s4[e], a1 = s4[e + 256], a2 = s4[e + 512], a3 = s4[e + 768];
        d4[e] = a0; d4[e + 256] = a1; d4[e + 512] = a2; d4[e + 768] = a3;
    }
    for (; e < w4; e += 256) d4[e] = s4[e];
}

__global__ void __launch_bounds__(256) k_dqn_stage(const float *o1, const float *o2, const float *ac, const float *r, const float *d,
                                                   float *x1, float *x2, float *acts, float *rew, float *done, int B, int obs, int ldx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * obs) {
        const int row = i / obs, c = i - row * obs;
        x1[(long long)row * ldx + c] = o1[i];
        x2[(long long)row * ldx + c] = o2[i];
    }
    if (i < B) { acts[i] = ac[i]; rew[i] = r[i]; done[i] = d[i]; }
}

}  // namespace

struct ddrl_dqn {
    int device;
    ddrl_dqn_config_t cfg;
    int nnet;  // 1 (DDQN) or 2 (SQN: q1, q2)
    long long W1[2], b1[2], W2[2], b2[2], W3[2], b3[2], total_int, total_ext;
    std::vector<Seg> segs;
    float *slab;
    float *main_p, *target_p, *m, *v, *grad;
    float *x1, *x2, *acts, *rew, *done, *H1, *H2, *Q, *dQ, *dZ2, *dZ1, *loss, *qsel;
    int ldx, ldh1, ldh2, ldq;
    OptState *opt;
    int opt_cur;
    Seg *segs_d;
    GemmJobs g_f1, g_f2, g_f3, g_b2, g_b1;   // g_f3: the head forward as GEMM jobs, ddrl_dqn_q only (the update's head lives in k_dqn_head)
    HeadArgs head;
    float *hpart;
    long long *ring_idx;  // the sampled indices of ddrl_dqn_step_ring
    AdamArgs ad;
    bool wide;            // layer 1 on wide_l1.h
    WideArgs wf, ww[2];   // forward (all evaluations), wgrad per network
    int wf_x2[WD_MAXEV];  // which input an evaluation reads: 0 obs1, 1 obs2
    float *wpart, *wconsts;
    // stream-K layer-1 wgrad (k_wide_sk): per network the fragment table; one slab / flag array / epoch shared (the launches are stream-ordered)
    SkArgs sk[2];
    bool sk_on;
    SkFrag *sk_frags_d;
    float *sk_slab;
    int *sk_flag;
    int *sk_err_h;                   // pinned, device-visible: k_wide_sk's sticky "combine timed out" word
    long long steps_launched, poison_after;   // DDRL_SK_POISON_AFTER=n (tests): the host raises the word itself behind the n-th step's launches
    bool poisoned;                   // that word was seen: the parameters carry at least one step from a wrong layer-1 gradient.  Every step / export
                                     // fails until fresh parameters arrive (set_weights, import of MAIN), so they cannot be pushed or checkpointed
};

// The stream-K combine's verdict, looked at wherever the host is about to trust the parameters: at the next step (unsynchronised: the
// steps already queued behind the failing one have run by then) and, with the stream drained, before anything is exported.
static int dqn_poison_check(ddrl_dqn *h, hipStream_t s, bool drain) {
    if (h->sk_err_h && !h->poisoned) {
        if (drain && h->sk_on) (void)hipStreamSynchronize(s);
        if (*h->sk_err_h) {
            *h->sk_err_h = 0;
            h->poisoned = true;
            h->sk_on = false;                      // from here on: the tile-per-workgroup kernel
            h->head.zero_words = nullptr; h->head.n_zero = 0;
        }
    }
    if (!h->poisoned) return DDRL_OK;
    ddrl::set_error("k_wide_sk: a split tile's partner did not publish within ~1 s (workgroups of the launch not co-resident: shared GPU or "
                    "serialising profiler?) — the layer-1 gradient of an earlier update was wrong and every later step built on it.  This learner "
                    "refuses to step or export until it is given fresh parameters (set_weights / import of the main parameters, e.g. from a "
                    "checkpoint); it then runs the tile-per-workgroup kernel (DDRL_WIDE_SK=0 selects that from the start).");
    return DDRL_ERR_HIP;
}

extern "C" {

int ddrl_dqn_destroy(ddrl_dqn_t *h) {
    if (!h) return DDRL_OK;
    ddrl::DeviceGuard g(h->device);
    (void)hipFree(h->slab);
    (void)hipFree(h->sk_frags_d); (void)hipFree(h->sk_slab); (void)hipFree(h->sk_flag);
    if (h->sk_err_h) (void)hipHostFree(h->sk_err_h);
    delete h;
    return DDRL_OK;
}

int ddrl_dqn_param_count(const ddrl_dqn_config_t *c, int64_t *n_h) {
    DDRL_REQUIRE(c != nullptr && n_h != nullptr, "NULL pointer");
    *n_h = (int64_t)c->obs_dim * c->hidden1 + c->hidden1 + (int64_t)c->hidden1 * c->hidden2 + c->hidden2 + (int64_t)c->hidden2 * c->n_actions +
           c->n_actions;
    if (c->variant == DDRL_SQN) *n_h *= 2;
    return DDRL_OK;
}

int ddrl_dqn_create(ddrl_dqn_t **out, int device, const ddrl_dqn_config_t *cfg) {
    DDRL_REQUIRE(out != nullptr && cfg != nullptr, "NULL pointer");
    DDRL_REQUIRE(cfg->obs_dim > 0 && cfg->n_actions > 0 && cfg->hidden1 > 0 && cfg->hidden2 > 0 && cfg->batch > 0, "dims must be positive");
    DDRL_REQUIRE(cfg->variant == DDRL_DDQN || (cfg->variant == DDRL_SQN && cfg->alpha > 0.0), "variant must be DDRL_DDQN, or DDRL_SQN with alpha > 0");
    DDRL_REQUIRE(cfg->n_actions <= HEAD_MAXA, "n_actions > 16 unsupported by the head kernel");
    ddrl::DeviceGuard g(device);
    if (!g.ok) { ddrl::set_error("cannot select device %d", device); return DDRL_ERR_HIP; }
    ddrl_dqn *h = new ddrl_dqn();
    h->device = device;
    h->cfg = *cfg;
    const int B = cfg->batch, o = cfg->obs_dim, A = cfg->n_actions, h1 = cfg->hidden1, h2 = cfg->hidden2;
    {   // internal layout: every kernel immediately followed by its bias, pairs 16-B aligned (as Layout)
        long long in = 0, ext = 0;
        auto add = [&](long long &slot, long long n, bool pad_after) {
            slot = in;
            h->segs.push_back(Seg{ext, in, n});
            in += n;
            if (pad_after) in = pad4(in);
            ext += n;
        };
        h->nnet = cfg->variant == DDRL_SQN ? 2 : 1;
        for (int n = 0; n < h->nnet; ++n) {
            add(h->W1[n], (long long)o * h1, false); add(h->b1[n], h1, true); add(h->W2[n], (long long)h1 * h2, false); add(h->b2[n], h2, true);
            add(h->W3[n], (long long)h2 * A, false); add(h->b3[n], A, true);
        }
        h->total_int = in; h->total_ext = ext;
    }
    h->ldx = (int)pad4(o + 1); h->ldh1 = (int)pad4(h1 + 1); h->ldh2 = (int)pad4(h2 + 1); h->ldq = (int)pad4(A);
    size_t slab_floats = 0;
    auto reserve = [&](size_t cnt) { size_t off = slab_floats; slab_floats += (cnt + 63) & ~(size_t)63; return off; };
    struct Item { float **p; size_t off; };
    std::vector<Item> items;
#define ALLOC(ptr, cnt) items.push_back(Item{&h->ptr, reserve((size_t)(cnt))})
    const size_t NT = (size_t)h->total_int;
    ALLOC(main_p, NT); ALLOC(target_p, NT); ALLOC(m, NT); ALLOC(v, NT); ALLOC(grad, NT);
    ALLOC(x1, (size_t)B * h->ldx); ALLOC(x2, (size_t)B * h->ldx); ALLOC(acts, B); ALLOC(rew, B); ALLOC(done, B);
    ALLOC(H1, (size_t)5 * B * h->ldh1); ALLOC(H2, (size_t)5 * B * h->ldh2); ALLOC(Q, (size_t)5 * B * h->ldq); ALLOC(dQ, (size_t)2 * B * h->ldq);
    ALLOC(dZ2, (size_t)2 * B * h2); ALLOC(dZ1, (size_t)2 * B * h1); ALLOC(loss, 4); ALLOC(qsel, B);
    ALLOC(hpart, (size_t)(B + HEAD_ROWS - 1) / HEAD_ROWS + 64);
    float *ring_idx_f = nullptr;
    items.push_back(Item{&ring_idx_f, reserve((size_t)2 * B + 64)});
    h->wide = wide_applies(o, h1);
    const int nev_all = cfg->variant == DDRL_SQN ? 5 : 3;
    if (h->wide) {
        wide_plan(h->wf, nev_all, B, h1, o, true, 4, 512);
        ALLOC(wpart, wide_part_floats(h->wf)); ALLOC(wconsts, 8);
    }
#undef ALLOC
    const size_t opt_off = reserve((2 * sizeof(OptState) + 3) / 4);
    const size_t segs_off = reserve((h->segs.size() * sizeof(Seg) + 3) / 4);
    (void)reserve(2048);
    hipError_t e = hipMalloc((void **)&h->slab, slab_floats * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->slab, 0, slab_floats * sizeof(float));
    if (e != hipSuccess) {
        ddrl::set_error("hipMalloc of %zu bytes failed in ddrl_dqn_create: %s", slab_floats * sizeof(float), hipGetErrorString(e));
        delete h;
        return DDRL_ERR_NOMEM;
    }
    for (auto &it : items) *it.p = h->slab + it.off;
    h->ring_idx = reinterpret_cast<long long *>(ring_idx_f);   // (slab offsets are multiples of 64 floats: 8-byte aligned)
    h->opt = reinterpret_cast<OptState *>(h->slab + opt_off);
    h->segs_d = reinterpret_cast<Seg *>(h->slab + segs_off);
    DDRL_HIP_CHECK(hipMemcpy(h->segs_d, h->segs.data(), h->segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
    {
        OptState os{};
        os.b1p_pi = os.b1p_q = (float)cfg->beta1;
        os.b2p_pi = os.b2p_q = (float)cfg->beta2;
        DDRL_HIP_CHECK(hipMemcpy(h->opt, &os, sizeof(os), hipMemcpyHostToDevice));
        h->opt_cur = 0;
    }
    if (h->wide) {
        const float c8[8] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        DDRL_HIP_CHECK(hipMemcpy(h->wconsts, c8, sizeof(c8), hipMemcpyHostToDevice));
        DDRL_HIP_CHECK(wide_prepare());
    }
    k_fill_col<<<(B + 255) / 256, 256>>>(h->x1, B, h->ldx, o, 1.0f);
    k_fill_col<<<(B + 255) / 256, 256>>>(h->x2, B, h->ldx, o, 1.0f);
    k_fill_col<<<(5 * B + 255) / 256, 256>>>(h->H1, 5ll * B, h->ldh1, h1, 1.0f);
    k_fill_col<<<(5 * B + 255) / 256, 256>>>(h->H2, 5ll * B, h->ldh2, h2, 1.0f);
    DDRL_LAUNCH_CHECK();
    DDRL_HIP_CHECK(hipDeviceSynchronize());
    const float *Pm = h->main_p, *Pt = h->target_p;
    const long long BH1 = (long long)B * h->ldh1, BH2 = (long long)B * h->ldh2, BQ = (long long)B * h->ldq;
    // evaluations (input, parameters, network): DDQN q(x), q(x2), q_target(x2); SQN q1(x), q1(x2), q2(x), q1_target(x2), q2_target(x2)
    const bool sqn = cfg->variant == DDRL_SQN;
    const int nev = sqn ? 5 : 3;
    const float *xin[5] = {h->x1, h->x2, sqn ? h->x1 : h->x2, h->x2, h->x2};
    const float *par[5] = {Pm, Pm, sqn ? Pm : Pt, Pt, Pt};
    const int net[5] = {0, 0, sqn ? 1 : 0, 0, 1};
    for (int ev = 0; ev < nev; ++ev) {
        const int n = net[ev];
        if (h->wide) {
            h->wf.ev[ev] = WideEval{xin[ev], par[ev] + h->W1[n], par[ev] + h->b1[n], h->H1 + ev * BH1, h->ldx};
            h->wf_x2[ev] = xin[ev] == h->x2;
        }
        gemm_add(h->g_f1, gemm_fwd(xin[ev], h->ldx, par[ev] + h->W1[n], par[ev] + h->b1[n], h->H1 + ev * BH1, h->ldh1, B, o, h1));
        gemm_add(h->g_f2, gemm_fwd(h->H1 + ev * BH1, h->ldh1, par[ev] + h->W2[n], par[ev] + h->b2[n], h->H2 + ev * BH2, h->ldh2, B, h1, h2));
        GemmJob j = gemm_fwd(h->H2 + ev * BH2, h->ldh2, par[ev] + h->W3[n], par[ev] + h->b3[n], h->Q + ev * BQ, h->ldq, B, h2, A);
        j.relu = 0;
        gemm_add(h->g_f3, j);
    }
    float *G = h->grad;
    const int gev[2] = {0, 2};  // differentiated evaluation of network n: q(x) / q1(x), q2(x)
    for (int n = 0; n < h->nnet; ++n) {
        const int ev = gev[n];
        float *dQ = h->dQ + (long long)n * BQ, *dZ2 = h->dZ2 + (long long)n * B * h2, *dZ1 = h->dZ1 + (long long)n * B * h1;
        // (dZ2 = (dQ * W3^T) .* (H2 > 0) comes out of k_dqn_head) the head kernels' gradient [H2 | 1]^T dQ rides in the layer-2 backward launch
        gemm_add(h->g_b2, gemm_wgrad(h->H2 + ev * BH2, h->ldh2, h2, dQ, h->ldq, A, G + h->W3[n], A, B));
        gemm_add(h->g_b2, gemm_dgrad(dZ2, Pm + h->W2[n], h->H1 + ev * BH1, h->ldh1, dZ1, B, h1, h2));
        gemm_add(h->g_b2, gemm_wgrad(h->H1 + ev * BH1, h->ldh1, h1, dZ2, h2, h2, G + h->W2[n], h2, B));
        gemm_add(h->g_b1, gemm_wgrad(h->x1, h->ldx, o, dZ1, h1, h1, G + h->W1[n], h1, B));
        if (h->wide) {   // [dW1 ; db1] = [x | 1]^T dZ1: the bias row follows the kernel in the flat gradient
            wide_plan(h->ww[n], 1, o + 1, h1, B, false, 4, 0);
            h->ww[n].ev[0] = WideEval{h->x1, dZ1, nullptr, G + h->W1[n], h->ldx};
            h->ww[n].a_rows = o; h->ww[n].consts = h->wconsts;
        }
    }
    if (h->wide) { h->wf.part = h->wpart; h->wf.consts = h->wconsts; h->wf.a_rows = B; h->wf.ldo = h->ldh1; }
    h->sk_on = false; h->sk_frags_d = nullptr; h->sk_slab = nullptr; h->sk_flag = nullptr; h->sk_err_h = nullptr; h->poisoned = false;
    h->steps_launched = 0; h->poison_after = getenv("DDRL_SK_POISON_AFTER") ? atoll(getenv("DDRL_SK_POISON_AFTER")) : -1;
    if (h->wide && !(getenv("DDRL_WIDE_SK") && atoi(getenv("DDRL_WIDE_SK")) == 0)) {
        // the wgrad as equal shares of the stage sequence over <= 512 resident workgroups (two per CU) when the shape splits that way
        int ncu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        std::vector<SkFrag> fr;
        const int nwg = wide_plan_sk(h->ww[0], 2 * ncu, fr);
        if (nwg > 1) {
            hipError_t e2 = hipMalloc((void **)&h->sk_frags_d, fr.size() * sizeof(SkFrag));
            if (e2 == hipSuccess) e2 = hipMalloc((void **)&h->sk_slab, (size_t)nwg * 4 * WD_NB * 16 * 64 * sizeof(float));
            if (e2 == hipSuccess) e2 = hipMalloc((void **)&h->sk_flag, (size_t)(nwg + 1) * sizeof(int));
            if (e2 == hipSuccess) e2 = hipMemset(h->sk_flag, 0, (size_t)(nwg + 1) * sizeof(int));
            if (e2 == hipSuccess) e2 = hipMemcpy(h->sk_frags_d, fr.data(), fr.size() * sizeof(SkFrag), hipMemcpyHostToDevice);
            if (e2 == hipSuccess) e2 = hipHostMalloc((void **)&h->sk_err_h, sizeof(int), hipHostMallocMapped);
            if (e2 == hipSuccess) {
                *h->sk_err_h = 0;
                for (int n = 0; n < h->nnet; ++n) {
                    h->sk[n].w = h->ww[n]; h->sk[n].frags = h->sk_frags_d; h->sk[n].slab = h->sk_slab; h->sk[n].flag = h->sk_flag;
                    h->sk[n].nwg = nwg; h->sk[n].epoch = n + 1;   // (the networks' launches of one update share the flags: one epoch each)
                    h->sk[n].err = h->sk_err_h;
                }
                h->sk_on = true;
            }
        }
    }
    {
        HeadArgs &H = h->head;
        H = HeadArgs{};
        H.H2 = h->H2;
        for (int ev = 0; ev < nev; ++ev) H.W3b[ev] = par[ev] + h->W3[net[ev]];
        for (int n = 0; n < h->nnet; ++n) { H.W3n[n] = Pm + h->W3[n]; H.gev[n] = gev[n]; H.nev_of_net[n] = gev[n]; }   // evaluation gev[n] runs on network n's main copy
        H.ldw = (int)pad4((long long)(h2 + 1) * A);
        H.acts = h->acts; H.rew = h->rew; H.done = h->done;
        H.Q = h->Q; H.dQ = h->dQ; H.dZ2 = h->dZ2; H.part = h->hpart; H.qsel = sqn ? nullptr : h->qsel;
        H.B = B; H.A = A; H.ldq = h->ldq; H.h2 = h2; H.ldh2 = h->ldh2; H.nev = nev; H.nnet = h->nnet; H.sqn = sqn ? 1 : 0;
        H.gamma = (float)cfg->gamma; H.alpha = (float)cfg->alpha;
        if (h->sk_on) { H.zero_words = h->sk_flag; H.n_zero = h->sk[0].nwg + 1; }   // this update's k_wide_sk polls them
    }
    h->ad = AdamArgs{h->main_p, h->target_p, h->m, h->v, h->grad, h->opt, h->opt + 1, h->total_int, 0, 0,
                     (float)cfg->lr, (float)cfg->beta1, (float)cfg->beta2, (float)cfg->adam_eps,
                     (float)cfg->polyak, (float)(1.0 - cfg->polyak), nullptr, 0, 0, 0, 0, 0u};
    // every exit below this point owns the slab and the stream-K buffers: failures go through ddrl_dqn_destroy
    if (head_lds_bytes(h->head) > 150 * 1024) {
        ddrl::set_error("hidden2 too wide for the head kernel's LDS staging");
        ddrl_dqn_destroy(h);
        return DDRL_ERR_BAD_ARG;
    }
    {   // the attribute belongs to the function on this device, not to the handle: it only ever grows
        static size_t granted[64] = {0};
        static std::mutex granted_mu;
        std::lock_guard<std::mutex> lk(granted_mu);
        const size_t need = head_lds_bytes(h->head);
        if (device >= 0 && device < 64 && need > granted[device]) {
            const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dqn_head), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
            if (ea != hipSuccess) {
                ddrl::set_error("hipFuncSetAttribute(k_dqn_head, %zu bytes of LDS): %s", need, hipGetErrorString(ea));
                ddrl_dqn_destroy(h);
                return DDRL_ERR_HIP;
            }
            granted[device] = need;
        }
    }
    *out = h;
    return DDRL_OK;
}

int ddrl_dqn_set_weights(ddrl_dqn_t *h, const float *flat_main_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_main_d != nullptr, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    h->poisoned = false;   // fresh parameters (main and target)
    // main AND target: Learner.set_weights runs target_init (algos/dqn/actor_learner.py:99-101)
    k_pack<<<dim3(64, (unsigned)h->segs.size()), 256, 0, ddrl::as_stream(stream)>>>(h->segs_d, flat_main_d, h->main_p, h->target_p, 1);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_dqn_export(ddrl_dqn_t *h, int which, float *flat_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_d != nullptr, "NULL pointer");
    float *buf = which == DDRL_SAC1_MAIN ? h->main_p : which == DDRL_SAC1_TARGET ? h->target_p : which == DDRL_SAC1_ADAM_M ? h->m :
                 which == DDRL_SAC1_ADAM_V ? h->v : which == DDRL_SAC1_GRAD ? h->grad : nullptr;
    DDRL_REQUIRE(buf != nullptr, "unknown buffer id");
    ddrl::DeviceGuard g(h->device);
    if (int prc = dqn_poison_check(h, ddrl::as_stream(stream), true)) return prc;   // (drains the stream while the stream-K kernel is in use)
    k_pack<<<dim3(64, (unsigned)h->segs.size()), 256, 0, ddrl::as_stream(stream)>>>(h->segs_d, buf, flat_d, nullptr, 0);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_dqn_import(ddrl_dqn_t *h, int which, const float *flat_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_d != nullptr, "NULL pointer");
    // MAIN alone (no target_init), TARGET, or an Adam slot: what a resumed learner / a test with its own targets sets
    float *buf = which == DDRL_SAC1_MAIN ? h->main_p : which == DDRL_SAC1_TARGET ? h->target_p : which == DDRL_SAC1_ADAM_M ? h->m :
                 which == DDRL_SAC1_ADAM_V ? h->v : nullptr;
    DDRL_REQUIRE(buf != nullptr, "unknown or read-only buffer id");
    ddrl::DeviceGuard g(h->device);
    if (which == DDRL_SAC1_MAIN) h->poisoned = false;   // fresh main parameters (a resumed learner imports target and moments next)
    k_pack<<<dim3(64, (unsigned)h->segs.size()), 256, 0, ddrl::as_stream(stream)>>>(h->segs_d, flat_d, buf, nullptr, 1);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

}  // extern "C"

// The launches of one update.  ev != nullptr: an event after every stage (DDRL_DQN_STAGES + 1 events, ev[0] first) for
// ddrl_dqn_step_timed; the update itself is the same either way.
// ridx != nullptr: obs1_d / obs2_d are the replay ring's observation arrays and batch row r is their row ridx[r] (ddrl_dqn_step_ring).
static int dqn_step_launch(ddrl_dqn_t *h, const float *obs1_d, const float *obs2_d, const float *acts_d, const float *rews_d, const float *done_d,
                           float *loss_d, float *q_d, hipStream_t s, hipEvent_t *ev, const long long *ridx = nullptr) {
    const int B = h->cfg.batch, o = h->cfg.obs_dim;
    int e = 0;
    if (int prc = dqn_poison_check(h, s, false)) return prc;   // a stream-K combine of an EARLIER step gave up waiting for its partner
#define STAGE_MARK() do { if (ev) DDRL_HIP_CHECK(hipEventRecord(ev[e++], s)); } while (0)
    STAGE_MARK();
    // wide layer 1 reads the caller's observation rows in place (16-byte aligned rows: obs_dim % 4 == 0 there) and the head kernel
    // the caller's acts / rews / done: no staging launch at all; otherwise everything is staged into the padded images
    const bool in_place = h->wide && al16(obs1_d) && al16(obs2_d);
    if (!in_place) {
        const int n = B * o > B ? B * o : B;
        k_dqn_stage<<<(n + 255) / 256, 256, 0, s>>>(obs1_d, obs2_d, acts_d, rews_d, done_d, h->x1, h->x2, h->acts, h->rew, h->done, B, o, h->ldx);
    }
    STAGE_MARK();   // 0 stage
    if (h->wide) {
        WideArgs f = h->wf;
        for (int k = 0; k < f.nev; ++k) {
            f.ev[k].A = in_place ? (h->wf_x2[k] ? obs2_d : obs1_d) : (h->wf_x2[k] ? h->x2 : h->x1);
            f.ev[k].lda = in_place ? o : h->ldx;
            f.ev[k].ridx = ridx;
        }
        launch_wide_fwd(f, s);
    } else {
        launch_gemm(h->g_f1, s);
    }
    STAGE_MARK();   // 1 layer-1 forward (+ split-K reduce)
    launch_gemm(h->g_f2, s);
    STAGE_MARK();   // 2 layer-2 forward
    {
        HeadArgs H = h->head;
        if (in_place) { H.acts = acts_d; H.rew = rews_d; H.done = done_d; }
        k_dqn_head<<<(unsigned)((B + HEAD_ROWS - 1) / HEAD_ROWS), 256, head_lds_bytes(H), s>>>(H);
    }
    STAGE_MARK();   // 3 head: Q of every evaluation, backup / loss / dQ, dZ2
    STAGE_MARK();   // 4 (folded into 3)
    STAGE_MARK();   // 5 (folded into 3 and 6)
    launch_gemm(h->g_b2, s);
    STAGE_MARK();   // 6 layer-2 dgrad + wgrad
    if (h->wide) {
        // ring rows (ridx): the gradient contracts over the batch rows, so it reads a gathered copy of obs1 (the one array of the
        // batch that is still materialised: 58 MB of the 231)
        const bool direct = in_place && !ridx;
        if (ridx) k_dqn_gather_rows<<<B, 256, 0, s>>>(obs1_d, ridx, h->x1, o, h->ldx);
        for (int nn = 0; nn < h->nnet; ++nn) {
            if (h->sk_on) {
                SkArgs g = h->sk[nn];
                g.w.ev[0].A = direct ? obs1_d : h->x1;
                g.w.ev[0].lda = direct ? o : h->ldx;
                launch_wide_sk(g, s);
            } else {
                WideArgs g = h->ww[nn];
                g.ev[0].A = direct ? obs1_d : h->x1;
                g.ev[0].lda = direct ? o : h->ldx;
                launch_wide_wgrad(g, s);
            }
        }
    } else {
        launch_gemm(h->g_b1, s);
    }
    STAGE_MARK();   // 7 layer-1 wgrad
    {
        const long long blocks = (h->total_int / 4 + 255) / 256;
        h->ad.adam_blocks = (int)blocks;
        h->ad.opt = h->opt + h->opt_cur; h->ad.opt_next = h->opt + (h->opt_cur ^ 1);
        h->opt_cur ^= 1;
        h->ad.loss_part = h->hpart; h->ad.loss_out = h->loss; h->ad.loss_out2 = loss_d; h->ad.loss_n = (B + HEAD_ROWS - 1) / HEAD_ROWS; h->ad.loss_scale = 0.5f / (float)B;
        k_adam_polyak<<<(unsigned)blocks + 1, 256, 0, s>>>(h->ad);   // + one workgroup: the loss mean from k_dqn_head's partials
    }
    STAGE_MARK();   // 8 flat Adam + polyak
#undef STAGE_MARK
    DDRL_LAUNCH_CHECK();
    if (++h->steps_launched == h->poison_after && h->sk_on && h->sk_err_h) *h->sk_err_h = 1;   // (test hook: what a timed-out combine stores)
    if (q_d) DDRL_HIP_CHECK(hipMemcpy2DAsync(q_d, (size_t)h->cfg.n_actions * sizeof(float), h->Q, (size_t)h->ldq * sizeof(float),
                                             (size_t)h->cfg.n_actions * sizeof(float), (size_t)B, hipMemcpyDeviceToDevice, s));
    return DDRL_OK;
}

extern "C" {

int ddrl_dqn_step(ddrl_dqn_t *h, const float *obs1_d, const float *obs2_d, const float *acts_d, const float *rews_d, const float *done_d,
                  float *loss_d, float *q_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && obs1_d && obs2_d && acts_d && rews_d && done_d, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    return dqn_step_launch(h, obs1_d, obs2_d, acts_d, rews_d, done_d, loss_d, q_d, ddrl::as_stream(stream), nullptr);
}

int ddrl_dqn_step_ring(ddrl_dqn_t *h, ddrl_replay_t *replay, float *loss_d, float *q_d, int64_t *idx_out_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && replay != nullptr, "NULL handle");
    const int B = h->cfg.batch, o = h->cfg.obs_dim;
    const ddrl_replay_dev::SamplerView rv = ddrl_replay_sampler_view(replay);
    DDRL_REQUIRE(rv.device == h->device, "the replay ring lives on another device than the learner");
    if (!h->wide) {
        ddrl::set_error("ddrl_dqn_step_ring needs the wide layer-1 path (obs_dim >= 1024): use ddrl_replay_sample + ddrl_dqn_step");
        return DDRL_ERR_UNSUPPORTED;
    }
    DDRL_REQUIRE(rv.ring.n_arr == 5 && rv.ring.w[0] == o && rv.ring.w[1] == o && rv.ring.w[2] == 1 && rv.ring.w[3] == 1 && rv.ring.w[4] == 1,
                 "the ring must be (obs1[obs_dim], obs2[obs_dim], acts, rews, done) of this learner's observation width");
    if (rv.ring.kind[0] || rv.ring.kind[1]) {
        ddrl::set_error("ddrl_dqn_step_ring reads float32 observation rows: a compact (uint8) ring goes through ddrl_replay_sample + ddrl_dqn_step");
        return DDRL_ERR_UNSUPPORTED;
    }
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    long long *idx = h->ring_idx;
    int rc = ddrl_replay_sample_indices(replay, B, reinterpret_cast<int64_t *>(idx), stream);   // np.random.randint(0, size, B); sample_times += 1
    if (rc != DDRL_OK) return rc;
    k_dqn_gather3<<<(B + 255) / 256, 256, 0, s>>>(rv.ring.a[2], rv.ring.a[3], rv.ring.a[4], idx, h->acts, h->rew, h->done, reinterpret_cast<long long *>(idx_out_d), B);
    return dqn_step_launch(h, rv.ring.a[0], rv.ring.a[1], h->acts, h->rew, h->done, loss_d, q_d, s, nullptr, idx);
}

int ddrl_dqn_step_timed(ddrl_dqn_t *h, const float *obs1_d, const float *obs2_d, const float *acts_d, const float *rews_d, const float *done_d,
                        int reps, float *stage_ms_h, void *stream) {
    DDRL_REQUIRE(h != nullptr && obs1_d && obs2_d && acts_d && rews_d && done_d && stage_ms_h, "NULL pointer");
    DDRL_REQUIRE(reps > 0 && reps <= 64, "reps outside [1, 64]");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    constexpr int NE = DDRL_DQN_STAGES + 1;
    std::vector<hipEvent_t> ev((size_t)reps * NE);
    for (auto &e : ev) DDRL_HIP_CHECK(hipEventCreate(&e));
    int rc = DDRL_OK;
    for (int r = 0; r < reps && rc == DDRL_OK; ++r) rc = dqn_step_launch(h, obs1_d, obs2_d, acts_d, rews_d, done_d, nullptr, nullptr, s, ev.data() + (size_t)r * NE);
    if (rc == DDRL_OK) {
        DDRL_HIP_CHECK(hipStreamSynchronize(s));
        for (int k = 0; k < DDRL_DQN_STAGES; ++k) stage_ms_h[k] = 0.f;
        for (int r = 0; r < reps; ++r)
            for (int k = 0; k < DDRL_DQN_STAGES; ++k) {
                float ms = 0.f;
                DDRL_HIP_CHECK(hipEventElapsedTime(&ms, ev[(size_t)r * NE + k], ev[(size_t)r * NE + k + 1]));
                stage_ms_h[k] += ms / (float)reps;
            }
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

// self.q for n <= batch observations (Actor.get_action / the learner's q output): rows of q(x) main
int ddrl_dqn_q(ddrl_dqn_t *h, const float *obs_d, int64_t n, float *q_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && obs_d != nullptr && q_d != nullptr, "NULL pointer");
    DDRL_REQUIRE(n > 0 && n <= h->cfg.batch, "n outside [1, batch]");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    const int o = h->cfg.obs_dim;
    DDRL_HIP_CHECK(hipMemcpy2DAsync(h->x1, (size_t)h->ldx * sizeof(float), obs_d, (size_t)o * sizeof(float), (size_t)o * sizeof(float), (size_t)n,
                                    hipMemcpyDeviceToDevice, s));
    if (h->wide) launch_wide_fwd(h->wf, s);   // (evaluations read the staged images here)
    else launch_gemm(h->g_f1, s);   // all three evaluations run (rows beyond n hold the previous batch): simple, off the hot path
    launch_gemm(h->g_f2, s);
    launch_gemm(h->g_f3, s);
    DDRL_LAUNCH_CHECK();
    DDRL_HIP_CHECK(hipMemcpy2DAsync(q_d, (size_t)h->cfg.n_actions * sizeof(float), h->Q, (size_t)h->ldq * sizeof(float),
                                    (size_t)h->cfg.n_actions * sizeof(float), (size_t)n, hipMemcpyDeviceToDevice, s));
    return DDRL_OK;
}

}  // extern "C"
