// Rollout-side window queues of the n-step SAC1 driver, one per env, on the device:
//   o_queue     = deque(maxlen = Ln + 1) of observations,
//   a_r_d_queue = deque(maxlen = Ln)     of (action, reward, done)
// (algos/sac1/sac_ray.py:192-248).  Kept as sliding windows in exactly the row layout of the n-step
// ring (ReplayBuffer of sac_ray.py:34-82: [Ln+1][obs], [Ln][act], [Ln], [Ln]), so that
// `replay_buffer.store(o_queue, a_r_d_queue)` for every env whose queues are full is one masked row
// store (ddrl_replay_store_masked_ex) straight from these arrays.
#include "ddrl_common.h"

namespace {

struct WinQ {
    float *o, *a, *r, *d;  // [n][(Ln+1)*obs], [n][Ln*act], [n][Ln], [n][Ln]
    int *t;                // t_queue per env
    long long n;
    int Ln, obs, act, save_freq;
};

// episode start (sac_ray.py:199-207): o_queue = [o], t_queue = 1.  Slots older than the newest are
// never stored before Ln further pushes have overwritten them.
__global__ void __launch_bounds__(256) k_winq_begin(WinQ q, const uint8_t *__restrict__ mask, const float *__restrict__ obs) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= q.n || (mask && !mask[e])) return;
    float *o = q.o + e * (long long)(q.Ln + 1) * q.obs + (long long)q.Ln * q.obs;
    for (int k = 0; k < q.obs; ++k) o[k] = obs[e * q.obs + k];
    q.t[e] = 1;
}

// one env.step later (sac_ray.py:229-248): a_r_d_queue.append((a, r, d)); o_queue.append((o2,));
// ready = t_queue >= Ln and t_queue % save_freq == 0; t_queue += 1.  One thread per env slides its
// own rows forward in place (ascending order: no hazard).
__global__ void __launch_bounds__(256) k_winq_push(WinQ q, const float *__restrict__ obs2, const float *__restrict__ act,
                                                   const float *__restrict__ rew, const float *__restrict__ done,
                                                   uint8_t *__restrict__ ready) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= q.n) return;
    float *o = q.o + e * (long long)(q.Ln + 1) * q.obs;
    for (int k = 0; k < q.Ln * q.obs; ++k) o[k] = o[k + q.obs];
    for (int k = 0; k < q.obs; ++k) o[q.Ln * q.obs + k] = obs2[e * q.obs + k];
    float *a = q.a + e * (long long)q.Ln * q.act;
    for (int k = 0; k < (q.Ln - 1) * q.act; ++k) a[k] = a[k + q.act];
    for (int k = 0; k < q.act; ++k) a[(q.Ln - 1) * q.act + k] = act[e * q.act + k];
    float *r = q.r + e * (long long)q.Ln, *d = q.d + e * (long long)q.Ln;
    for (int k = 0; k < q.Ln - 1; ++k) { r[k] = r[k + 1]; d[k] = d[k + 1]; }
    r[q.Ln - 1] = rew[e];
    d[q.Ln - 1] = done[e];
    const int t = q.t[e];
    if (ready) ready[e] = (t >= q.Ln && t % q.save_freq == 0) ? 1 : 0;
    q.t[e] = t + 1;
}

}  // namespace

struct ddrl_winq {
    int device;
    WinQ q;
};

extern "C" {

int ddrl_winq_destroy(ddrl_winq_t *h) {
    if (!h) return DDRL_OK;
    ddrl::DeviceGuard g(h->device);
    (void)hipFree(h->q.o); (void)hipFree(h->q.a); (void)hipFree(h->q.r); (void)hipFree(h->q.d); (void)hipFree(h->q.t);
    delete h;
    return DDRL_OK;
}

int ddrl_winq_create(ddrl_winq_t **out, int device, int64_t n_envs, int32_t Ln, int32_t obs_dim, int32_t act_dim, int32_t save_freq) {
    DDRL_REQUIRE(out != nullptr, "out is NULL");
    DDRL_REQUIRE(n_envs > 0 && Ln >= 1 && obs_dim > 0 && act_dim > 0 && save_freq >= 1, "n_envs, Ln, obs_dim, act_dim, save_freq must be positive");
    ddrl::DeviceGuard g(device);
    if (!g.ok) { ddrl::set_error("cannot select device %d", device); return DDRL_ERR_HIP; }
    ddrl_winq *h = new ddrl_winq();
    h->device = device;
    h->q = WinQ{nullptr, nullptr, nullptr, nullptr, nullptr, n_envs, Ln, obs_dim, act_dim, save_freq};
    const size_t n = (size_t)n_envs;
    const size_t bytes[5] = {n * (Ln + 1) * obs_dim * 4, n * Ln * act_dim * 4, n * Ln * 4, n * Ln * 4, n * sizeof(int)};
    void **ptr[5] = {(void **)&h->q.o, (void **)&h->q.a, (void **)&h->q.r, (void **)&h->q.d, (void **)&h->q.t};
    for (int i = 0; i < 5; ++i) {
        hipError_t e = hipMalloc(ptr[i], bytes[i]);
        if (e == hipSuccess) e = hipMemset(*ptr[i], 0, bytes[i]);
        if (e != hipSuccess) {
            ddrl::set_error("hipMalloc of %zu bytes failed in ddrl_winq_create: %s", bytes[i], hipGetErrorString(e));
            ddrl_winq_destroy(h);
            return DDRL_ERR_NOMEM;
        }
    }
    *out = h;
    return DDRL_OK;
}

int ddrl_winq_begin(ddrl_winq_t *h, const uint8_t *mask_d, const float *obs_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && obs_d != nullptr, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    k_winq_begin<<<(unsigned)((h->q.n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(h->q, mask_d, obs_d);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_winq_push(ddrl_winq_t *h, const float *obs2_d, const float *act_d, const float *rew_d, const float *done_d, uint8_t *ready_d,
                   void *stream) {
    DDRL_REQUIRE(h != nullptr && obs2_d && act_d && rew_d && done_d, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    k_winq_push<<<(unsigned)((h->q.n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(h->q, obs2_d, act_d, rew_d, done_d, ready_d);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_winq_buffers(ddrl_winq_t *h, float **arrays_h, int32_t **t_queue_h) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    if (arrays_h) { arrays_h[0] = h->q.o; arrays_h[1] = h->q.a; arrays_h[2] = h->q.r; arrays_h[3] = h->q.d; }
    if (t_queue_h) *t_queue_h = h->q.t;
    return DDRL_OK;
}

}  // extern "C"
