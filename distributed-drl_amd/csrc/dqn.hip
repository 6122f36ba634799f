// Discrete-action learners on the shared GEMM core: Double-DQN (algos/dqn) and soft-Q (algos/sqn).
#include "gemm_core.h"
#include "wide_l1.h"

// ==========================================================================================
// Double-DQN learner (algos/dqn/actor_learner.py:19-107 on algos/dqn/core.py:40-50):
// q = mlp(obs -> h1 -> h2 -> n_actions), q_x2 = the same variables at obs2, q_next = target(obs2);
// q_target = q_next[argmax q_x2]; q_loss = 0.5 mean((r + gamma (1-d) q_target - q[a])^2);
// one Adam over main/q1, polyak over all.  Every layer (also layer 1: obs_dim is arbitrary here) is a
// job of the generic MFMA GEMM kernel; 8 launches per update.  Wide observations (obs_dim >= 1024, config 5's 28 224) take
// layer 1 through the tiled kernels of wide_l1.h instead (forward split over K + reduce; wgrad), reading the caller's
// observation rows in place.
// variant DDRL_SQN = the soft-Q learner of algos/sqn/actor_learner.py:19-78 on algos/sqn/core.py:30-79:
// twin networks main/q1, main/q2; evaluations q1(x), q1(x2) (its softmax policy's sum p log p at x2),
// q2(x) and the targets q1_(x2), q2_(x2); v_backup = min(max q1_, max q2_) - alpha * sum p log p;
// q_loss = 0.5 mean((q_backup - q1[a])^2) + 0.5 mean((q_backup - q2[a])^2); one Adam over main/q1, main/q2.
// ==========================================================================================
namespace {

struct DqnRows {
    const float *Q;      // [3][B][ldq]: q(x) main, q(x2) main, q_next(x2) target
    const float *acts, *rew, *done;
    float *dQ;           // [B][ldq]
    float *loss;         // [1]
    float *qsel;         // [B] q(x)[a] (diagnostic output)
    int B, A, ldq;
    float gamma;
};
// one workgroup; thread r handles row r (B <= 1024 per pass), then a fixed-order tree reduction
__global__ void __launch_bounds__(256) k_dqn_rows(DqnRows a) {
    __shared__ float s_red[256];
    float acc = 0.f;
    const long long BQ = (long long)a.B * a.ldq;
    for (int r0 = 0; r0 < a.B; r0 += 256) {
        const int r = r0 + threadIdx.x;
        if (r < a.B) {
            const float *q = a.Q + (long long)r * a.ldq, *qx2 = q + BQ, *qn = qx2 + BQ;
            const int act = (int)a.acts[r];                       // tf.cast(a_ph, tf.int32)
            int best = 0;
            float bv = qx2[0];
            for (int c = 1; c < a.A; ++c) { const float v = qx2[c]; if (v > bv) { bv = v; best = c; } }  // tf.argmax: first maximum
            const float q_value = q[(act >= 0 && act < a.A) ? act : 0];
            const float valid = (act >= 0 && act < a.A) ? 1.0f : 0.0f;  // one_hot of an out-of-range index is all zeros
            const float backup = a.rew[r] + (a.gamma * (1.0f - a.done[r])) * qn[best];
            const float e = backup - q_value * valid;
            acc += e * e;
            const float g = -e / (float)a.B;
            for (int c = 0; c < a.ldq; ++c) a.dQ[(long long)r * a.ldq + c] = (c == act && c < a.A) ? g : 0.f;
            if (a.qsel) a.qsel[r] = q_value * valid;
        }
    }
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.loss[0] = 0.5f * (s_red[0] / (float)a.B);
}

// SQN rows.  Q slots: 0 q1(x)  1 q1(x2)  2 q2(x)  3 q1_target(x2)  4 q2_target(x2);  dQ slots: 0 q1, 1 q2.
struct SqnRows {
    const float *Q;
    const float *acts, *rew, *done;
    float *dQ;     // [2][B][ldq]
    float *loss;   // [1] q_loss = q1_loss + q2_loss
    int B, A, ldq;
    float gamma, alpha;
};
__global__ void __launch_bounds__(256) k_sqn_rows(SqnRows a) {
    __shared__ float s_red[256];
    float acc = 0.f;
    const long long BQ = (long long)a.B * a.ldq;
    for (int r0 = 0; r0 < a.B; r0 += 256) {
        const int r = r0 + threadIdx.x;
        if (r < a.B) {
            const float *q1 = a.Q + (long long)r * a.ldq, *q1x2 = q1 + BQ, *q2 = q1x2 + BQ, *q1t = q2 + BQ, *q2t = q1t + BQ;
            const int act = (int)a.acts[r];
            const bool valid = act >= 0 && act < a.A;
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
            const float q_backup = a.rew[r] + (a.gamma * (1.0f - a.done[r])) * v_backup;
            const float e1 = q_backup - (valid ? q1[act] : 0.f), e2 = q_backup - (valid ? q2[act] : 0.f);
            acc += e1 * e1 + e2 * e2;
            const float g1 = -e1 / (float)a.B, g2 = -e2 / (float)a.B;
            for (int c = 0; c < a.ldq; ++c) {
                const bool hit = valid && c == act;
                a.dQ[(long long)r * a.ldq + c] = hit ? g1 : 0.f;
                a.dQ[BQ + (long long)r * a.ldq + c] = hit ? g2 : 0.f;
            }
        }
    }
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.loss[0] = 0.5f * (s_red[0] / (float)a.B);
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
    GemmJobs g_f1, g_f2, g_f3, g_b3, g_b2, g_b1;
    DqnRows rows;
    SqnRows srows;
    AdamArgs ad;
    bool wide;            // layer 1 on wide_l1.h
    WideArgs wf, ww[2];   // forward (all evaluations), wgrad per network
    int wf_x2[WD_MAXEV];  // which input an evaluation reads: 0 obs1, 1 obs2
    float *wpart, *wconsts;
};

extern "C" {

int ddrl_dqn_destroy(ddrl_dqn_t *h) {
    if (!h) return DDRL_OK;
    ddrl::DeviceGuard g(h->device);
    (void)hipFree(h->slab);
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
        {   // dZ2 = (dQ * W3^T) .* (H2 > 0): A = dQ [B x A] (row stride ldq), B(k, j) = W3[j * A + k]
            GemmJob j{};
            j.adam_off = -1;
            j.A = dQ; j.B = Pm + h->W3[n]; j.C = dZ2; j.bias = nullptr; j.mask = h->H2 + ev * BH2;
            j.M = B; j.N = h2; j.K = A; j.lda = h->ldq; j.ldb = A; j.ldc = h2; j.ldmask = h->ldh2; j.a_kc = 1; j.b_kc = 1; j.relu = 0;
            set_fast(j);
            gemm_add(h->g_b3, j);
        }
        gemm_add(h->g_b3, gemm_wgrad(h->H2 + ev * BH2, h->ldh2, h2, dQ, h->ldq, A, G + h->W3[n], A, B));
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
    h->srows = SqnRows{h->Q, h->acts, h->rew, h->done, h->dQ, h->loss, B, A, h->ldq, (float)cfg->gamma, (float)cfg->alpha};
    h->rows = DqnRows{h->Q, h->acts, h->rew, h->done, h->dQ, h->loss, h->qsel, B, A, h->ldq, (float)cfg->gamma};
    h->ad = AdamArgs{h->main_p, h->target_p, h->m, h->v, h->grad, h->opt, h->opt + 1, h->total_int, 0, 0,
                     (float)cfg->lr, (float)cfg->beta1, (float)cfg->beta2, (float)cfg->adam_eps,
                     (float)cfg->polyak, (float)(1.0 - cfg->polyak), nullptr, 0, 0, 0, 0, 0u};
    *out = h;
    return DDRL_OK;
}

int ddrl_dqn_set_weights(ddrl_dqn_t *h, const float *flat_main_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_main_d != nullptr, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
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
    k_pack<<<dim3(64, (unsigned)h->segs.size()), 256, 0, ddrl::as_stream(stream)>>>(h->segs_d, flat_d, buf, nullptr, 1);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

}  // extern "C"

// The launches of one update.  ev != nullptr: an event after every stage (DDRL_DQN_STAGES + 1 events, ev[0] first) for
// ddrl_dqn_step_timed; the update itself is the same either way.
static int dqn_step_launch(ddrl_dqn_t *h, const float *obs1_d, const float *obs2_d, const float *acts_d, const float *rews_d, const float *done_d,
                           float *loss_d, float *q_d, hipStream_t s, hipEvent_t *ev) {
    const int B = h->cfg.batch, o = h->cfg.obs_dim;
    int e = 0;
#define STAGE_MARK() do { if (ev) DDRL_HIP_CHECK(hipEventRecord(ev[e++], s)); } while (0)
    STAGE_MARK();
    // wide layer 1 reads the caller's observation rows in place (16-byte aligned rows: obs_dim % 4 == 0 there); otherwise
    // they are staged into the padded images with the ones column
    const bool in_place = h->wide && al16(obs1_d) && al16(obs2_d);
    const int n = in_place ? B : (B * o > B ? B * o : B);
    k_dqn_stage<<<(n + 255) / 256, 256, 0, s>>>(obs1_d, obs2_d, acts_d, rews_d, done_d, h->x1, h->x2, h->acts, h->rew, h->done, B, in_place ? 0 : o, h->ldx);
    STAGE_MARK();   // 0 stage
    if (h->wide) {
        WideArgs f = h->wf;
        for (int k = 0; k < f.nev; ++k) {
            f.ev[k].A = in_place ? (h->wf_x2[k] ? obs2_d : obs1_d) : (h->wf_x2[k] ? h->x2 : h->x1);
            f.ev[k].lda = in_place ? o : h->ldx;
        }
        launch_wide_fwd(f, s);
    } else {
        launch_gemm(h->g_f1, s);
    }
    STAGE_MARK();   // 1 layer-1 forward (+ split-K reduce)
    launch_gemm(h->g_f2, s);
    STAGE_MARK();   // 2 layer-2 forward
    launch_gemm(h->g_f3, s);
    STAGE_MARK();   // 3 head forward
    if (h->cfg.variant == DDRL_SQN) k_sqn_rows<<<1, 256, 0, s>>>(h->srows);
    else k_dqn_rows<<<1, 256, 0, s>>>(h->rows);
    STAGE_MARK();   // 4 rows: backup, loss, dQ
    launch_gemm(h->g_b3, s);
    STAGE_MARK();   // 5 head dgrad + wgrad
    launch_gemm(h->g_b2, s);
    STAGE_MARK();   // 6 layer-2 dgrad + wgrad
    if (h->wide) {
        for (int nn = 0; nn < h->nnet; ++nn) {
            WideArgs g = h->ww[nn];
            g.ev[0].A = in_place ? obs1_d : h->x1;
            g.ev[0].lda = in_place ? o : h->ldx;
            launch_wide_wgrad(g, s);
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
        k_adam_polyak<<<(unsigned)blocks, 256, 0, s>>>(h->ad);
    }
    STAGE_MARK();   // 8 flat Adam + polyak
#undef STAGE_MARK
    DDRL_LAUNCH_CHECK();
    if (loss_d) DDRL_HIP_CHECK(hipMemcpyAsync(loss_d, h->loss, sizeof(float), hipMemcpyDeviceToDevice, s));
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
