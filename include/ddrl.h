/*
 * ddrl.h — C-ABI of libddrl_hip.so: the MI355X-native (gfx950) actor–learner hot path of
 * createamind/Distributed-DRL (replay ring store + uniform-sample gather, SAC1 learner update,
 * batched policy forward, batched env.step, parameter-server buffer).
 *
 * The reference has no FFI on this path: its boundary is Ray's remote-callable surface
 * (`Class.remote(...)`, `handle.method.remote(...)`, `ray.get`) over plain-Python/NumPy classes.
 * Each entry point below cites the reference interface it replaces (paths relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes binding a maintainer of the reference
 * would add; `distributed-drl_amd/` holds that binding plus the Python classes with the
 * reference's names and signatures.
 *
 * Conventions
 *   - plain C types only; every pointer named *_d / documented "device" is a device pointer
 *     (HBM) to float32 unless stated otherwise; host pointers are named *_h / documented "host".
 *   - every function returns 0 (DDRL_OK) or a negative ddrl_status; no exception or abort
 *     crosses the boundary; `ddrl_last_error()` returns a thread-local message.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All device work is
 *     enqueued on it; no call synchronises the device except where documented (counts /
 *     *_to_host helpers).  Inputs are borrowed until the stream reaches the enqueued work;
 *     outputs are caller-allocated; handles own all persistent device memory.
 *   - handles are not thread-safe (the reference's actors execute methods serially,
 *     algos/sac1/sac_ray.py:316-317); the Python actor shim serialises calls per handle.
 */
#ifndef DDRL_H
#define DDRL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDRL_VERSION 100 /* 0.1.0 */

typedef enum {
    DDRL_OK = 0,
    DDRL_ERR_BAD_ARG = -1,
    DDRL_ERR_EMPTY_BUFFER = -2, /* sample from an empty ring: reference raises ValueError("high <= 0") */
    DDRL_ERR_HIP = -3,
    DDRL_ERR_NOMEM = -4,
    DDRL_ERR_UNSUPPORTED = -5,
    DDRL_ERR_NOT_REPRESENTABLE = -6, /* a value stored into a compact (uint8) ring array was not an integer in [0, 255] */
    DDRL_ERR_RCCL = -7 /* RCCL could not be loaded, or a collective / point-to-point call failed (ddrl_last_error has RCCL's text) */
} ddrl_status;

int ddrl_version(void);
const char *ddrl_last_error(void);
/* Name of the device the library sees as `device` ("gfx950…"), for fail-loud checks. host buf. */
int ddrl_device_arch(int device, char *buf_h, int buflen);
/* The address under which the DEVICE sees a page-locked host buffer (hipHostMalloc / hipHostRegister memory: what torch's pin_memory()
 * hands out) — usable wherever this header takes a device pointer for small, latency-bound transfers: ddrl_replay_sample then gathers a
 * batch straight into the caller's host block (posted PCIe writes, no copy behind the launch), as ReplayBuffer.prefetch does for the
 * reference's host-array surface (example/dsac.py:39-45).  DDRL_ERR_BAD_ARG for pageable memory. */
int ddrl_host_device_pointer(const void *host_ptr, void **dev_ptr_out);

/* ===================================================================================== */
/* Replay ring buffer — replaces class ReplayBuffer                                       */
/*   example/dsac.py:14-48, algos/sac1/sac1.py:28-63, algos/dqn/train.py:37-76            */
/* ===================================================================================== */
typedef struct ddrl_replay ddrl_replay_t;

#define DDRL_REPLAY_ACTS_1D 1u /* algos/dqn/train.py:48: acts_buf is [N] (act_dim must be 1) */
/* Opt-in COMPACT ring for integer-valued pixel observations (config 5: 84x84x4 frames, algos/dqn/train.py:43-52 at 4 M transitions =
 * 903 GB as float32): obs1 / obs2 are stored as uint8 — 4 M transitions = 226 GB, one MI355X — behind the unchanged float32
 * surface.  store converts; a value that is not an integer in [0, 255] raises the sticky DDRL_ERR_NOT_REPRESENTABLE (reported by
 * the next ddrl_replay_counts / ddrl_replay_sample that looks at the ring, like the sampler's errors); sample / gather / rows_export
 * convert back, so every result is bit-identical to the float32 ring's on such data. */
#define DDRL_REPLAY_U8_OBS 2u

/* ReplayBuffer.__init__(obs_dim, act_dim, size)  (example/dsac.py:20-27): five zero-filled
 * float32 struct-of-arrays rings obs1[N,obs] obs2[N,obs] acts[N,act] rews[N] done[N];
 * ptr = size = 0; counters steps / sample_times = 0; MT19937 stream seeded with 0. */
int ddrl_replay_create(ddrl_replay_t **out, int device, int64_t capacity, int obs_dim, int act_dim,
                       uint32_t flags);
int ddrl_replay_destroy(ddrl_replay_t *h);

/* np.random.seed(seed) in the reference's ReplayBuffer process (legacy global MT19937,
 * init_genrand).  Enqueued on `stream`. */
int ddrl_replay_seed(ddrl_replay_t *h, uint32_t seed, void *stream);

/* ReplayBuffer.store(obs, act, rew, next_obs, done) x n  (example/dsac.py:29-37): exactly n
 * sequential store() calls in row order, including wrap-around inside the batch
 * (ptr=(ptr+1)%N, size=min(size+1,N), steps+=1 per row).  Device float32 inputs:
 * obs_d[n,obs_dim] act_d[n,act_dim] rew_d[n] obs2_d[n,obs_dim] done_d[n] (done as 0.0/1.0). */
int ddrl_replay_store(ddrl_replay_t *h, const float *obs_d, const float *act_d, const float *rew_d,
                      const float *obs2_d, const float *done_d, int64_t n, void *stream);

/* ReplayBuffer.sample_batch(batch_size)  (example/dsac.py:39-45; algos/sac1/sac1.py:53-60 also
 * bumps sample_times): idxs = np.random.randint(0, size, batch_size) on the ring's MT19937
 * stream (bit-exact, the stream position advances by exactly the words NumPy would consume),
 * then the five fancy-index gathers into caller buffers obs1_d[B,obs] obs2_d[B,obs]
 * acts_d[B,act] rews_d[B] done_d[B].  idx_d (device int64[B]) may be NULL.
 * Returns DDRL_ERR_EMPTY_BUFFER when size == 0 (reference: ValueError "high <= 0"). */
int ddrl_replay_sample(ddrl_replay_t *h, int64_t batch, float *obs1_d, float *obs2_d, float *acts_d,
                       float *rews_d, float *done_d, int64_t *idx_d, void *stream);

/* Same gather for caller-provided indices (device int64[B]); does not touch the MT stream or
 * sample_times.  Used by the sharded sampler and by size-independent property tests. */
int ddrl_replay_gather(ddrl_replay_t *h, const int64_t *idx_d, int64_t batch, float *obs1_d,
                       float *obs2_d, float *acts_d, float *rews_d, float *done_d, void *stream);

/* ReplayBuffer.get_counts()  (example/dsac.py:47-48 -> steps; algos/sac1/sac1.py:62-63 ->
 * (sample_times, steps, size)).  Synchronises `stream` (the only sync on the replay path).
 * Any out pointer may be NULL.  Host outputs. */
int ddrl_replay_counts(ddrl_replay_t *h, int64_t *ptr_h, int64_t *size_h, int64_t *steps_h,
                       int64_t *sample_times_h, void *stream);

/* ---- generalised ring: any set of 1..6 float32 struct-of-arrays rings sharing one cursor ----
 * Replaces the n-step window buffer of algos/sac1/sac_ray.py:34-82 — class ReplayBuffer(opt) with
 * buffer_o[N,(Ln+1),obs] buffer_a[N,Ln(,act)] buffer_r[N,Ln] buffer_d[N,Ln]; store() writes one
 * (Ln+1)-frame window per slot and sample_batch() gathers whole windows; its counters advance by
 * opt.num_buffers per call (sac_ray.py:68,75) — and any other row layout.  The five-array
 * functions above are the instance widths = {obs, obs, act, 1, 1}.
 * widths_h / src_h / out_h / arrays_h are HOST arrays (of ints / of device pointers), in ring order. */
int ddrl_replay_create_ex(ddrl_replay_t **out, int device, int64_t capacity, int32_t n_arrays,
                          const int32_t *widths_h, int64_t steps_inc, int64_t samples_inc);
/* The same with a storage kind per array: kinds_h[j] = 0 float32, 1 uint8 (see DDRL_REPLAY_U8_OBS; NULL = all float32). */
int ddrl_replay_create_typed(ddrl_replay_t **out, int device, int64_t capacity, int32_t n_arrays, const int32_t *widths_h,
                             const uint8_t *kinds_h, int64_t steps_inc, int64_t samples_inc);
int ddrl_replay_store_ex(ddrl_replay_t *h, const float *const *src_h, int64_t n, void *stream);
/* store() for the rows whose mask_d[i] != 0 (device uint8[n]), in row order: the n-step rollout stores
 * a window only for the envs whose deque is full (algos/sac1/sac_ray.py:243-246).  The row count
 * stays on the device; host-side counters are refreshed by the next ddrl_replay_counts. */
int ddrl_replay_store_masked_ex(ddrl_replay_t *h, const float *const *src_h, const uint8_t *mask_d, int64_t n, void *stream);
int ddrl_replay_sample_ex(ddrl_replay_t *h, int64_t batch, float *const *out_h, int64_t *idx_d, void *stream);
int ddrl_replay_gather_ex(ddrl_replay_t *h, const int64_t *idx_d, int64_t batch, float *const *out_h, void *stream);
/* The index draw of sample_batch alone: idx_d[batch] = np.random.randint(0, size, batch) on the ring's stream (advances it and
 * sample_times exactly like ddrl_replay_sample); the rows stay where they are — for a consumer that reads them out of the ring itself
 * (ddrl_dqn_step_ring). */
int ddrl_replay_sample_indices(ddrl_replay_t *h, int64_t batch, int64_t *idx_d, void *stream);
/* `count` consecutive sample_batch(batch) calls in one launch sequence: out_h[j] is [count*batch, w_j], batch i = rows
 * [i*batch, (i+1)*batch).  Consumes the index stream exactly like `count` separate calls and advances sample_times
 * by count * samples_inc.  A shard owner draws the block of batches it owes a remote learner for one step with it
 * (the reference learner's `replay_buffer[i].sample_batch.remote()`, algos/sac1/sac_ray.py:137-141, batched). */
int ddrl_replay_sample_many(ddrl_replay_t *h, int64_t batch, int64_t count, float *const *out_h, void *stream);
/* Attach a feed plan to the ring's sampler (learner side of the sharded replay): the next plan_len calls of the
 * small-batch sampler (ddrl_replay_sample* with `batch`, alone or inside the learner's launches / a captured loop)
 * follow plan_d[0..plan_len) (DEVICE int32): -1 = draw from this ring as usual; (r << 24 | i) = copy batch i of
 * region r instead, consuming no local draw.  Region r is a block that ddrl_replay_sample_many produced on the
 * owning rank: region_base_h[r] (device) -> [obs1 | obs2 | ...] with each array [region_count_h[r]*batch, w_j].
 * The plan position restarts at 0 on every call; plan_d and the regions must stay valid while attached.
 * plan_d == NULL detaches.  n_regions <= 16.  A plan entry that is out of range (wrong batch, region or batch index)
 * or a local draw from an empty ring cannot fail the call that launched it (it may run inside a captured graph): it
 * leaves the output untouched and sets the ring's sticky device-side error, which the next ddrl_replay_counts
 * returns (DDRL_ERR_BAD_ARG / DDRL_ERR_EMPTY_BUFFER) and clears. */
int ddrl_replay_set_feed(ddrl_replay_t *h, const int32_t *plan_d, int32_t plan_len, int32_t batch, int32_t n_regions,
                         const float *const *region_base_h, const int32_t *region_count_h, void *stream);
/* Move the ring's sticky device-side error (0 = none) to out_d[0] (DEVICE int32) and clear it, in stream order and
 * without synchronising: a caller that keeps the device running ahead (partition.py) copies the word back behind an
 * event and looks at it one step later. */
int ddrl_replay_take_error(ddrl_replay_t *h, int32_t *out_d, void *stream);
int ddrl_replay_buffers_ex(ddrl_replay_t *h, float **arrays_h, int32_t *widths_h, int32_t *n_arrays_h);

/* Raw ring pointers (device) for checkpointing / inspection (algos/dqn/train.py:82-90 saves
 * exactly these five arrays + (ptr,size,max_size,steps,sample_times)). */
int ddrl_replay_buffers(ddrl_replay_t *h, float **obs1_d, float **obs2_d, float **acts_d,
                        float **rews_d, float **done_d);
/* Ring rows [row0, row0 + nrows) of array `array` as float32 into out_d[nrows, w] / from src_d — whatever the storage kind: how the
 * .npy checkpoint of algos/dqn/train.py:82-108 (float32 arrays) is streamed out of / into a compact ring, chunk by chunk.  (The
 * raw pointers of ddrl_replay_buffers* address BYTES for a uint8 array.)  A non-representable value on import: sticky error as in store. */
int ddrl_replay_rows_export(ddrl_replay_t *h, int32_t array, int64_t row0, int64_t nrows, float *out_d, void *stream);
int ddrl_replay_rows_import(ddrl_replay_t *h, int32_t array, int64_t row0, int64_t nrows, const float *src_d, void *stream);
/* Restore counters after loading the arrays (algos/dqn/train.py:92-108). */
int ddrl_replay_set_counts(ddrl_replay_t *h, int64_t ptr, int64_t size, int64_t steps,
                           int64_t sample_times, void *stream);
/* Copy the MT19937 state (624 words + position) to the host; synchronises `stream`. */
int ddrl_replay_mt_state(ddrl_replay_t *h, uint32_t *key_h, int32_t *pos_h, void *stream);

/* ===================================================================================== */
/* Parameter server — replaces class ParameterServer (example/dsac.py:51-73)              */
/*   one flat float32 device buffer; the name -> (offset, shape) table lives in Python    */
/* ===================================================================================== */
typedef struct ddrl_ps ddrl_ps_t;

int ddrl_ps_create(ddrl_ps_t **out, int device, int64_t count);
int ddrl_ps_destroy(ddrl_ps_t *h);
/* push: snapshot-by-copy of src_d[count] into the server at `offset` (dsac.py:59-62);
 * bumps the version.  pull: copy out (dsac.py:64-65). Device<->device, on `stream`. */
int ddrl_ps_push(ddrl_ps_t *h, const float *src_d, int64_t offset, int64_t count, void *stream);
int ddrl_ps_pull(ddrl_ps_t *h, float *dst_d, int64_t offset, int64_t count, void *stream);
int ddrl_ps_buffer(ddrl_ps_t *h, float **buf_d, int64_t *count);
int64_t ddrl_ps_version(ddrl_ps_t *h);

/* ===================================================================================== */
/* SAC1 learner — replaces class Learner (algos/sac1/actor_learner.py:19-148) with the    */
/* network of algos/sac1/core.py:91-121                                                   */
/* ===================================================================================== */
typedef struct ddrl_sac1 ddrl_sac1_t;
#define DDRL_SAC1 0
#define DDRL_SAC_V 1

typedef struct {
    int32_t obs_dim;   /* 8  LunarLanderContinuous-v2 */
    int32_t act_dim;   /* 2 */
    int32_t hidden1;   /* 400  core.py:91 hidden_sizes=(400,300) */
    int32_t hidden2;   /* 300 */
    int32_t batch;     /* 256  hyperparams.py:82 */
    int32_t variant;   /* DDRL_SAC1 (0): algos/sac1 ; DDRL_SAC_V (1): example/model.py (policy + twin Q + V + target V) */
    /* Python floats in the reference (doubles): each derived constant is rounded to float32
     * exactly where TensorFlow would round it (e.g. float32(1 - polyak), float32(alpha)). */
    double alpha;      /* 0.1    hyperparams.py:60 (fixed, not 'auto') */
    double gamma;      /* 0.997  hyperparams.py:67 */
    double lr;         /* 5e-5   hyperparams.py:78 */
    double polyak;     /* 0.995  hyperparams.py:79 */
    double beta1;      /* 0.9    tf.train.AdamOptimizer defaults */
    double beta2;      /* 0.999 */
    double adam_eps;   /* 1e-8 */
    double act_scale;  /* action_space.high[0] = 1.0  core.py:104-106 */
} ddrl_sac1_config_t;

/* Parameter layout (flat float32, TF variable creation order, kernels [in,out] row-major):
 *   pi : dense/kernel[obs,h1] dense/bias[h1] dense_1/kernel[h1,h2] dense_1/bias[h2]
 *        dense_2/kernel[h2,act] dense_2/bias[act] (mu)  dense_3/kernel[h2,act] dense_3/bias[act] (log_std)
 *   q1 : dense/kernel[obs+act,h1] dense/bias[h1] dense_1/kernel[h1,h2] dense_1/bias[h2]
 *        dense_2/kernel[h2,1] dense_2/bias[1]
 *   q2 : as q1
 * flat = [pi | q1 | q2]; n_pi = 125104, n_q = 125001 at the default sizes. */
int ddrl_sac1_param_counts(const ddrl_sac1_config_t *cfg, int64_t *n_pi, int64_t *n_q);

int ddrl_sac1_create(ddrl_sac1_t **out, int device, const ddrl_sac1_config_t *cfg);
int ddrl_sac1_destroy(ddrl_sac1_t *h);

/* Learner.set_weights (actor_learner.py:125-127): copy flat main weights in, then target_init
 * (target <- main).  Does not touch Adam state.  flat_main_d: device float32[n_pi+2*n_q]. */
int ddrl_sac1_set_weights(ddrl_sac1_t *h, const float *flat_main_d, void *stream);
/* Learner.get_weights (actor_learner.py:129-133): the "main" variables. */
int ddrl_sac1_get_weights(ddrl_sac1_t *h, float *flat_main_d, void *stream);
/* Dense (flat layout above) export / import of the learner's state, on `stream`: main and
 * target parameters, Adam first/second moments, and the gradient of the last compute_grads.
 * Used for checkpoint/resume and for the multi-learner gradient all-reduce.
 * set_weights == import(MAIN) + target_init;  get_weights == export(MAIN). */
typedef enum {
    DDRL_SAC1_MAIN = 0,
    DDRL_SAC1_TARGET = 1,
    DDRL_SAC1_ADAM_M = 2,
    DDRL_SAC1_ADAM_V = 3,
    DDRL_SAC1_GRAD = 4
} ddrl_sac1_buffer;
int ddrl_sac1_export(ddrl_sac1_t *h, int which, float *flat_d, void *stream);
int ddrl_sac1_import(ddrl_sac1_t *h, int which, const float *flat_d, void *stream);
/* Data-parallel learners without the export / import copies: after ddrl_sac1_compute_grads, ddrl_sac1_grad_finalize makes
 * the learner's own gradient buffer complete (sums the layer-1 partials) and marks it as final; ddrl_sac1_grad_buffer
 * returns that buffer (device pointer, *n floats, the learner's INTERNAL parameter-shaped layout, identical on every learner
 * of the same config; padding elements are zero) for an in-place all-reduce; ddrl_sac1_apply_grads then steps with it. */
int ddrl_sac1_grad_buffer(ddrl_sac1_t *h, float **grad_d, int64_t *n);
int ddrl_sac1_grad_finalize(ddrl_sac1_t *h, void *stream);
/* Adam step counters of the two optimizers (host outputs); synchronises `stream`. */
int ddrl_sac1_opt_steps(ddrl_sac1_t *h, int64_t *t_pi_h, int64_t *t_q_h, void *stream);
/* Optimizer bookkeeping for checkpoint / resume: Adam step counts of the two optimizers and the
 * device noise counter.  set() rebuilds the beta powers as TF keeps them (running float32 products,
 * one multiply per applied step), so that a learner restored from export()ed MAIN / TARGET / ADAM_M /
 * ADAM_V continues bit-identically.  Both synchronise `stream`. */
int ddrl_sac1_opt_state_get(ddrl_sac1_t *h, int64_t *t_pi_h, int64_t *t_q_h, uint64_t *noise_ctr_h, void *stream);
int ddrl_sac1_opt_state_set(ddrl_sac1_t *h, int64_t t_pi, int64_t t_q, uint64_t noise_ctr, void *stream);

/* Learner.train(batch) == sess.run(step_ops) (actor_learner.py:58-101,135-142): forward of main
 * and target nets, pi_loss / q1_loss / q2_loss from the PRE-update parameters, Adam(pi) then
 * Adam(q1,q2) (TF1 formula, epsilon outside the sqrt, lr_t = lr*sqrt(1-b2^t)/(1-b1^t)), then
 * polyak target update with the post-update main.  tf.random_normal (core.py:77) is replaced by
 * explicit noise inputs eps_x_d (main policy at x), eps_x2_d (main policy at x2), eps_t_d
 * (target policy at x2), each device float32[B,act].  losses_d: device float32[3]
 * = (pi_loss, q1_loss, q2_loss) (step_ops[0:3], actor_learner.py:97).  Optional per-row outputs
 * (may be NULL): q1_d[B], q2_d[B], logp_pi_d[B]  (step_ops[3:6]). */
int ddrl_sac1_step(ddrl_sac1_t *h, const float *obs1_d, const float *obs2_d, const float *acts_d,
                   const float *rews_d, const float *done_d, const float *eps_x_d,
                   const float *eps_x2_d, const float *eps_t_d, float *losses_d, float *q1_d,
                   float *q2_d, float *logp_pi_d, void *stream);

/* Two-phase form of ddrl_sac1_step for data-parallel learners: (1) forward+backward only,
 * gradients left in the grad buffer; (2) Adam(pi), Adam(q), polyak using the (possibly
 * all-reduced, via export/import of DDRL_SAC1_GRAD) grad buffer.  step == compute followed by apply. */
int ddrl_sac1_compute_grads(ddrl_sac1_t *h, const float *obs1_d, const float *obs2_d,
                            const float *acts_d, const float *rews_d, const float *done_d,
                            const float *eps_x_d, const float *eps_x2_d, const float *eps_t_d,
                            float *losses_d, float *q1_d, float *q2_d, float *logp_pi_d,
                            void *stream);
int ddrl_sac1_apply_grads(ddrl_sac1_t *h, void *stream);
/* apply_grads fused with the sampler of the NEXT update: equivalent to ddrl_sac1_apply_grads(h)
 * followed by ddrl_replay_sample(replay, B, <input set `set` of h>...), but the sampler runs as one
 * extra workgroup of the Adam/polyak kernel instead of as a kernel of its own (it writes an input
 * set the Adam kernel does not touch).  Same results, same MT19937 stream, same counters. */
int ddrl_sac1_apply_grads_and_sample(ddrl_sac1_t *h, ddrl_replay_t *replay, int set, void *stream);
/* One whole update from input set `set_in` (noise as armed by ddrl_sac1_fill_noise) AND the next
 * update's sample_batch (example/dsac.py:39-45: np.random.randint + five gathers) into set `set_out`,
 * i.e. Learner.train of update u plus `ray.get(replay_buffer.sample_batch.remote())` of update u+1
 * (actor_learner.py:135-142).  On the fused path the sampler rides in a forward launch and the
 * optimizer in the last backward launch; otherwise == compute_grads + apply_grads_and_sample. */
int ddrl_sac1_step_and_sample(ddrl_sac1_t *h, int set_in, ddrl_replay_t *replay, int set_out, void *stream);
/* The same without the optimizer step (data-parallel learners: all-reduce, then ddrl_sac1_apply_grads): forward + backward on
 * input set set_in while the sampler of the NEXT batch rides in a forward launch and gathers into set_out. */
int ddrl_sac1_compute_grads_and_sample(ddrl_sac1_t *h, int set_in, ddrl_replay_t *replay, int set_out, void *stream);

/* The reference's learner loop body with HOST arrays — `agent.train(replay_buffer.sample_batch())`, example/dsac.py:142-144,
 * algos/sac1/sac1.py:146-148 — as one call: block_h (page-locked host memory; n_floats floats) holds obs1, obs2, acts, rews, done at
 * the offsets the five buffers of ddrl_sac1_input_buffers(h, 0, .) have from the first one (DDRL_ERR_BAD_ARG if those do not form one
 * span; where an item is allocated with more rows than the batch — the direct path pads to whole 32-row tiles — the block must hold
 * zeros there); the first launch of the call reads it straight over PCIe into the input set (page-locked memory is device-addressable:
 * no copy node; pageable memory goes through copies and is never captured), eps_x / eps_x2 / eps_t are generated by the same launch
 * exactly as three ddrl_normal_fill calls at noise_ctr, noise_ctr + B*act, noise_ctr + 2*B*act would, then one ddrl_sac1_step on that
 * set.  The block must have room for TWO more floats behind the n_floats: the call writes the noise counter there and the device reads
 * it from there — which is what lets the whole sequence (batch up + noise, the update's launches) replay as ONE captured graph of
 * kernel launches per (block, state-copy parity) from the second use of a block on: on this surface the host's launch calls, not the
 * device, set the rate (DDRL_HOST_GRAPH=0: always eager).  The block must stay untouched until the work issued by the call has run.
 * The block may be rewritten once work queued on `stream` behind this call has started (record an event after the call and wait for
 * it). */
int ddrl_sac1_step_host(ddrl_sac1_t *h, float *block_h, int64_t n_floats, uint32_t noise_seed, uint64_t noise_ctr, float *losses_d,
                        void *stream);

/* For a caller that captures update sequences into a graph of its own (partition.py captures `gradients -> RCCL all-reduce ->
 * apply` of the data-parallel learners with torch.cuda.graph; the reference has no counterpart: its learners never
 * synchronise, example/dsac.py:59-62,233): puts the learner's double-buffered optimizer state / dgrad image on copy 0.  Call
 * it on the stream right before the capture begins and again as the LAST captured call, so that the graph starts and ends
 * on the same copy whatever the number of updates it holds (a copy node when that number is odd). */
int ddrl_sac1_graph_sync(ddrl_sac1_t *h, void *stream);
/* A stream capture that is ABORTED (the runtime or a collective refuses a node) leaves the device untouched, but every
 * ddrl_sac1_* launch call made while recording has already advanced this handle's host-side launch state (which copy of the
 * double-buffered optimizer state / dgrad images the next launch reads, the armed noise request, pending counter advances).
 * ddrl_sac1_capture_begin snapshots that state right before the capture starts (after the leading ddrl_sac1_graph_sync);
 * ddrl_sac1_capture_abort puts it back, so that eager calls continue exactly as if the capture had never been tried
 * (partition.py's fallback for the data-parallel step; no counterpart in the reference).  A capture that ends normally needs
 * neither.  One snapshot per handle; abort without begin is DDRL_ERR_BAD_ARG. */
int ddrl_sac1_capture_begin(ddrl_sac1_t *h);
int ddrl_sac1_capture_abort(ddrl_sac1_t *h);

/* The learner's internal input buffers (device): obs1[B,obs] obs2[B,obs] acts[B,act] rews[B]
 * done[B] eps_x[B,act] eps_x2[B,act] eps_t[B,act], in this order in bufs_h[8] (host array of
 * device pointers).  There are two sets (set = 0 or 1) so that a sampler can fill one set while an
 * update reads the other.  Passing exactly the eight pointers of one set to ddrl_sac1_step /
 * compute_grads skips the staging copy: ddrl_replay_sample gathers straight into the learner. */
int ddrl_sac1_input_buffers(ddrl_sac1_t *h, int set, float **bufs_h);
/* The batch size the learner was created with. */
int ddrl_sac1_batch(ddrl_sac1_t *h);
/* 1 when this handle's shape runs the five-launch direct-operand update (csrc/sac1_direct.h: hidden sizes % 4 == 0 and
 * <= 512, obs + act <= 12, act <= 4), 0 on the generic 9-launch path (k_l1 / k_gemm / k_rows_* / k_adam_polyak). */
int ddrl_sac1_is_fused(ddrl_sac1_t *h);
/* Arm noise generation for the NEXT ddrl_sac1_step / compute_grads: its first kernel fills the
 * three noise buffers of the input set in use with N(0,1) from the counter generator
 * (hash(seed, counter+i), same values as ddrl_normal_fill over a flat [3][B*act] buffer); the
 * counter lives on the device and advances by 3*B*act in that update's Adam kernel, so the
 * sequence can be captured in a hipGraph.  Stand-in for tf.random_normal (core.py:77).  No device
 * work is enqueued by this call itself. */
int ddrl_sac1_fill_noise(ddrl_sac1_t *h, uint32_t seed, void *stream);
/* Roofline accounting (bench.py): launch one stage of the update `reps` times back to back on
 * `stream` between two HIP events (after 3 untimed launches) and return the mean milliseconds
 * per launch (host output); synchronises `stream`.  The stages read the buffers left by the last
 * ddrl_sac1_step and are idempotent.  Stage ids:
 *   direct-operand path (ddrl_sac1_is_fused == 1), the five launches of one update in launch order:
 *     2 k_dfwd<0> (evaluations 0-4: layer 1 + layer 2 + head partials)   5 k_dfwd<1> (evaluations 5-7; + the next
 *     batch's sampler when one is armed)   7 k_dg "bq" (Q heads / losses / the three Q dgrads)   8 k_dg "mid" (policy dgrad with
 *     its A operand generated in the tile + policy-head backward tiles + Q layer-2 / head wgrads)   9 k_dg "pi" (policy wgrads of
 *     all three layers + Q layer-1 wgrads + loss means / optimizer bookkeeping); 8 and 9 run with their optimizer epilogues exactly
 *     as the last step left them armed; 1, 3, 4, 6, 10 are no-ops there;
 *   generic path: 1 layer 1 (8 nets; + noise) 2 gemm fwd (A: 5 nets) 3 heads (A) + 2nd-phase layer 1 5 gemm fwd (B: 3
 *     nets) 6 heads (B) + losses 7 gemm bwd (Q: 3 dgrad + 4 wgrad) 8 policy-head bwd 9 gemm bwd (pi: 1 dgrad + 5 wgrad)
 *     10 pi layer-1 wgrad.
 *   [0 = input staging copy, 11 = adam + polyak: not idempotent, not timed here] */
#define DDRL_SAC1_STAGES 12
int ddrl_sac1_stage_time(ddrl_sac1_t *h, int stage, int reps, float *ms_per_launch_h, void *stream);

/* ===================================================================================== */
/* Learner hot loop — replaces the body of worker_train                                    */
/*   `while True: batch = sample_batch(B); agent.train(batch)`                              */
/*   (algos/sac1/sac1.py:146-148; example/dsac.py:142-144 + example/model.py:92-101)        */
/* sample (MT19937 indices + gather, straight into the learner's buffers) -> update, n times, */
/* with no host work per update: the sequence for `updates_per_graph` updates is captured     */
/* once into a hipGraph (all cursors, RNG state and Adam state are device-resident) and       */
/* replayed.  Consecutive updates alternate between the learner's two input sets; with         */
/* DDRL_LOOP_FORK=1 the sampler runs on a forked graph branch so that sample(u+1) overlaps      */
/* update(u) — the role of the reference's `Cache` prefetch process (sac1.py:103-130); measured */
/* slower than inline on MI355X, hence opt-in.                                      */
/* ===================================================================================== */
typedef struct ddrl_loop ddrl_loop_t;
int ddrl_loop_create(ddrl_loop_t **out, ddrl_sac1_t *learner, ddrl_replay_t *replay, int32_t updates_per_graph,
                     uint32_t noise_seed);
int ddrl_loop_destroy(ddrl_loop_t *h);
/* Enqueue n_updates sample+train iterations on `stream` (graph replays + an eager remainder).
 * Returns DDRL_ERR_EMPTY_BUFFER if the ring is empty. */
int ddrl_loop_run(ddrl_loop_t *h, int64_t n_updates, void *stream);

/* ===================================================================================== */
/* Batched policy forward — replaces Actor.get_action (actor_learner.py:195-197) called    */
/* once per env step in worker_rollout (example/dsac.py:96-97)                             */
/* ===================================================================================== */
typedef struct ddrl_actor ddrl_actor_t;

int ddrl_actor_create(ddrl_actor_t **out, int device, const ddrl_sac1_config_t *cfg, int64_t max_rows);
int ddrl_actor_destroy(ddrl_actor_t *h);
/* Actor.set_weights / get_weights (actor_learner.py:186-193): the n_pi floats of the main/pi
 * variables (device, dense flat layout). */
int ddrl_actor_set_weights(ddrl_actor_t *h, const float *flat_pi_d, void *stream);
int ddrl_actor_get_weights(ddrl_actor_t *h, float *flat_pi_d, void *stream);
/* get_action for n observations: act = tanh(mu + eps*exp(log_std))*act_scale, or
 * tanh(mu)*act_scale when deterministic (core.py:49-87,104-106).  obs_d[n,obs], eps_d[n,act]
 * (ignored when deterministic; may be NULL then), act_d[n,act].  n <= max_rows.
 * Threading / capture: an actor handle belongs to ONE host thread and ONE stream at a time.  For policies inside the direct-operand
 * envelope ddrl_actor_set_weights only packs the direct-layout copy; the row-major copy this call (and get_weights) reads is rebuilt
 * lazily, by a launch this call issues the first time after a set_weights (a host-side flag).  A caller that captures
 * ddrl_actor_act into a graph of its own must therefore call it once eagerly after every set_weights before capturing or
 * replaying — a captured call replays "no rebuild" and would act on the weights of the capture. */
int ddrl_actor_act(ddrl_actor_t *h, const float *obs_d, const float *eps_d, int64_t n,
                   int deterministic, float *act_d, void *stream);
/* Actor.get_action(o) for ONE observation as one launch (actor_learner.py:195-197, the call a reference-style rollout worker makes
 * per env step, example/dsac.py:96): the same function as ddrl_actor_act with n = 1 and eps = the act_dim elements ddrl_normal_fill
 * yields at (noise_seed, noise_ctr) — within float32 summation order of it.  obs_d[obs_dim], act_d[act_dim] (device pointers, or
 * device-side addresses of page-locked host rows: ddrl_host_device_pointer).  DDRL_ERR_UNSUPPORTED for obs_dim > 64, a hidden
 * width > 512 or act_dim > 4.  Same threading / capture rules as ddrl_actor_act. */
int ddrl_actor_act_one(ddrl_actor_t *h, const float *obs_d, uint32_t noise_seed, uint64_t noise_ctr, int deterministic, float *act_d,
                       void *stream);

/* ===================================================================================== */
/* Batched lander environment — stands where gym's LunarLanderContinuous-v2 env.step /     */
/* env.reset are called (example/dsac.py:78-79,102,127).  gym/Box2D are third-party and    */
/* absent; the dynamics are this build's own Box2D-style rigid-body model, */
/* validated against oracle/env_oracle.py, not against Box2D ("parity unpinned").          */
/* ===================================================================================== */
typedef struct ddrl_env ddrl_env_t;

int ddrl_env_create(ddrl_env_t **out, int device, int64_t n_envs, uint32_t seed, int32_t max_ep_len);
int ddrl_env_destroy(ddrl_env_t *h);
/* env.reset() for every env (mask_d == NULL) or for envs with mask_d[i] != 0 (device uint8[n]).
 * obs_d[n,8] receives the current observation of every env. */
int ddrl_env_reset(ddrl_env_t *h, const uint8_t *mask_d, float *obs_d, void *stream);
/* One env.step(a) per env followed by the reference's episode bookkeeping
 * (example/dsac.py:102-127): outputs the transition as it is stored —
 *   obs2_d[n,8]  next observation o2,
 *   rew_d[n]     reward,
 *   done_d[n]    d with the time-limit override "d = False if ep_len == max_ep_len" (dsac.py:109),
 * then resets every env whose episode ended (d or ep_len == max_ep_len) and writes
 *   next_obs_d[n,8]  the observation to act on next (o2, or the reset observation),
 *   ended_d[n]       (uint8, may be NULL) 1 where an episode ended this step.
 * act_d[n,2] device float32. */
int ddrl_env_step(ddrl_env_t *h, const float *act_d, float *obs2_d, float *rew_d, float *done_d,
                  float *next_obs_d, uint8_t *ended_d, void *stream);
/* env.step through `Wrapper(env, obs_noise, act_noise, reward_scale, action_repeat)`
 * (algos/sac1/hyperparams.py:107-134: uniform action noise, `action_repeat` physics steps with the
 * rewards summed — dropped to 0.0 when the episode ends inside the repeat —, uniform observation
 * noise, reward scale; action_repeat == 1 returns the bare step) followed by the n-step rollout's
 * bookkeeping (algos/sac1/sac_ray.py:212-258): an episode ends on d or after `limit_steps` wrapped
 * steps (= ceil(max_ep_len / opt.action_repeat)), done_d is the raw d.  Same outputs as ddrl_env_step.
 * act_d is updated IN PLACE with the action noise, as the reference's `action += ...` mutates the caller's
 * array (hyperparams.py:124): what the rollout appends to its a_r_d_queue afterwards is the noisy action. */
int ddrl_env_step_wrapped(ddrl_env_t *h, float *act_d, float act_noise, float obs_noise, float reward_scale,
                          int32_t action_repeat, int32_t limit_steps, float *obs2_d, float *rew_d, float *done_d,
                          float *next_obs_d, uint8_t *ended_d, void *stream);
/* RolloutDevice.step fused: ONE vector step of worker_rollout's policy phase (example/dsac.py:96-130; SAC1 flavour
 * algos/sac1/sac1.py:177-213) for all n envs, as two launches with no host work in between:
 *   a = agent.get_action(o)  (Actor.get_action, actor_learner.py:195-197; noise element i*act+c of the counter stream
 *                             (noise_seed, noise_ctr) = what ddrl_normal_fill would produce; tanh(mu) when deterministic)
 *   o2, r, d, _ = env.step(a);  replay_buffer.store(o, a, r, o2, d)  (n stores in env order, like ddrl_replay_store);
 *   o = o2, or the reset observation where the episode ended (d or ep_len == max_ep_len; the stored d is 0 at the limit).
 * The acted-on observations live inside the actor handle: ddrl_rollout_begin copies the envs' current observations there
 * (call it once, and again after any ddrl_env_step / ddrl_env_reset issued outside this path).  Requires an actor whose
 * shape the direct-operand policy supports (LunarLander: obs 8, act 2, hidden % 4 == 0, n % 32 == 0) — else
 * DDRL_ERR_BAD_ARG; the unfused sequence is ddrl_actor_act + ddrl_env_step + ddrl_replay_store.
 * n_steps >= 1 vector steps are issued back to back (step k draws noise elements noise_ctr + k*n*act ...).
 * act_out_d[n,2], next_obs_out_d[n,8]: optional mirrors of the LAST step (may be NULL). */
/* Exact per-env weight adoption for a vectorised rollout worker.  A reference worker pulls the server's weights at ITS OWN episode end
 * (example/dsac.py:127-130; algos/sac1/sac1.py:209-213) and acts on them until its next one, so n workers hold up to
 * min(n, max_ep_len) + 1 different versions at a time.  ddrl_actor_versions_enable gives the actor `n_slots` resident copies of the
 * policy (n_slots >= min(n_envs, max_ep_len) + 2 never runs out; [2, 2048]) and a slot word per env (all on slot 0 = the current
 * weights).  From then on
 *   ddrl_actor_set_weights  stores the incoming vector as the NEWEST version in a slot no env acts on (the newest slot itself while
 *                           no env has adopted it) — a pull that some env's future episode end will see;
 *   ddrl_rollout_step       evaluates every env against the version in its slot (envs grouped by slot on the device, one row tile
 *                           per 32 envs of a version) and moves an env to the newest version where its episode ends;
 *   ddrl_actor_versions_adopt  does that move for steps taken outside the fused path (ended_d[n]: uint8 mask of ddrl_env_step).
 * ddrl_actor_act keeps evaluating the newest weights.  ddrl_actor_versions_state: slot_of_env_d[max_rows] (device, nullable) and
 * state_h[4] = {newest slot, slots in use at the last install, row tiles of the last versioned launch, sticky out-of-slots flag}
 * (host, nullable; synchronises `stream`). */
int ddrl_actor_versions_enable(ddrl_actor_t *h, int32_t n_slots, void *stream);
int ddrl_actor_versions_state(ddrl_actor_t *h, int32_t *slot_of_env_d, int32_t *state_h, void *stream);
int ddrl_actor_versions_adopt(ddrl_actor_t *h, const uint8_t *ended_d, int64_t n, void *stream);
/* get_action for all max_rows envs, each against the version in its slot (obs_d[n, obs_dim], eps_d[n, act] explicit normals, act_d[n, act]):
 * the versioned form of ddrl_actor_act for a caller that steps its envs itself (the n-step rollout of algos/sac1/sac_ray.py:208-262, followed
 * by ddrl_actor_versions_adopt with the step's episode ends).  horizon_steps = the envs' episode limit in calls: that many calls after the
 * last set_weights every env has adopted the newest version and the plain single-version launch is used. */
int ddrl_actor_act_versioned(ddrl_actor_t *h, const float *obs_d, const float *eps_d, int64_t n, int deterministic, int32_t horizon_steps,
                             float *act_d, void *stream);
int ddrl_rollout_begin(ddrl_env_t *h, ddrl_actor_t *actor, void *stream);
int ddrl_rollout_step(ddrl_env_t *h, ddrl_actor_t *actor, ddrl_replay_t *replay, int32_t n_steps, uint32_t noise_seed,
                      uint64_t noise_ctr, int deterministic, float *act_out_d, float *next_obs_out_d, void *stream);
/* Episode statistics accumulated on device since the last call: number of finished episodes,
 * sum of their returns and lengths.  Synchronises `stream`; resets the accumulators. Host outs. */
int ddrl_env_stats(ddrl_env_t *h, int64_t *episodes_h, double *ret_sum_h, int64_t *len_sum_h,
                   void *stream);
/* Raw state access for parity tests: copies the [DDRL_ENV_STATE_FIELDS, n] float32 state block
 * to / from a device buffer. */
#define DDRL_ENV_STATE_FIELDS 32
int ddrl_env_get_state(ddrl_env_t *h, float *state_d, void *stream);
int ddrl_env_set_state(ddrl_env_t *h, const float *state_d, void *stream);

/* ===================================================================================== */
/* Double-DQN learner — algos/dqn/actor_learner.py:19-107 (network: algos/dqn/core.py:40-50: */
/* mlp(obs -> hidden1 -> hidden2 -> n_actions), variables main/q1/dense{,_1,_2}/{kernel,bias}) */
/* ===================================================================================== */
typedef struct ddrl_dqn ddrl_dqn_t;
#define DDRL_DDQN 0
#define DDRL_SQN 1   /* algos/sqn/actor_learner.py:19-78 on algos/sqn/core.py:30-79: twin soft-Q networks main/q1, main/q2 */
typedef struct {
    int32_t obs_dim, n_actions, hidden1, hidden2, batch, variant;
    double gamma;      /* 0.99   algos/dqn/hyperparams.py:26 */
    double lr;         /* 1e-3   :57 */
    double polyak;     /* 0.995  :58 */
    double beta1, beta2, adam_eps;  /* tf.train.AdamOptimizer defaults */
    double alpha;      /* SQN only: 0.1  algos/sqn/hyperparams.py:27 (temperature of the softmax policy, > 0) */
} ddrl_dqn_config_t;
int ddrl_dqn_param_count(const ddrl_dqn_config_t *cfg, int64_t *n_h);
int ddrl_dqn_create(ddrl_dqn_t **out, int device, const ddrl_dqn_config_t *cfg);
int ddrl_dqn_destroy(ddrl_dqn_t *h);
/* Learner.set_weights (flat "main" vector in variable order) incl. target_init (actor_learner.py:99-101). */
int ddrl_dqn_set_weights(ddrl_dqn_t *h, const float *flat_main_d, void *stream);
/* Flat copies of the learner's buffers; `which` = DDRL_SAC1_MAIN / TARGET / ADAM_M / ADAM_V / GRAD. */
int ddrl_dqn_export(ddrl_dqn_t *h, int which, float *flat_d, void *stream);
/* The inverse for MAIN (without target_init), TARGET, ADAM_M, ADAM_V: a learner resumed from its own state, or one whose
 * target network differs from main (any state after the first update: actor_learner.py:62-66). GRAD is read-only. */
int ddrl_dqn_import(ddrl_dqn_t *h, int which, const float *flat_d, void *stream);
/* Learner.train(batch, cnt) == sess.run([q_loss, q, train_value_op, target_update]) (actor_learner.py:110-119):
 * acts_d holds the action indices as float32 (the buffer's acts_buf), loss_d[1] and q_d[batch, n_actions]
 * (either may be NULL) receive q_loss and self.q from the pre-update variables.
 * Wide observations (obs_dim >= 1024, a multiple of 4; config 5's 28 224) whose obs1_d / obs2_d are 16-byte aligned are read IN PLACE by the
 * layer-1 kernels, obs1_d until the last GEMM of the step (its layer-1 weight gradient): the rows must stay unchanged until `stream`
 * has passed the step (the ordinary stream-order contract of every borrowed input here); other inputs are copied by the first launch. */
int ddrl_dqn_step(ddrl_dqn_t *h, const float *obs1_d, const float *obs2_d, const float *acts_d, const float *rews_d,
                  const float *done_d, float *loss_d, float *q_d, void *stream);
/* One whole iteration of the learner's loop — `batch = replay_buffer.sample_batch(); agent.train(batch, cnt)` (algos/dqn/train.py:66-76 +
 * actor_learner.py:110-119) — without materialising the batch: the indices are drawn on the ring's own stream (== ddrl_replay_sample_indices),
 * acts / rews / done are gathered (3 x batch floats), the layer-1 FORWARD of every evaluation reads its observation rows straight out of
 * the ring through the index list (the LDS-DMA loads take per-lane source addresses), and only obs1 — which the layer-1 weight gradient
 * contracts over the batch — is gathered (58 MB of config 5's 231 MB).  Results are bit-identical to ddrl_replay_sample + ddrl_dqn_step.
 * Needs the wide layer-1 path (obs_dim >= 1024) and a float32 five-array ring of this observation width on the same device; else
 * DDRL_ERR_UNSUPPORTED (use the two calls; a ring on another device: DDRL_ERR_BAD_ARG).  idx_out_d[batch] (nullable) receives the
 * indices.  The forward and the weight gradient read the ring at two different points of the call, so the ring rows must stay unchanged
 * until `stream` has passed the call: stores into this ring must be ordered with it on `stream` (or by an event), never concurrent. */
int ddrl_dqn_step_ring(ddrl_dqn_t *h, ddrl_replay_t *replay, float *loss_d, float *q_d, int64_t *idx_out_d, void *stream);
/* `reps` updates (each exactly ddrl_dqn_step) with a HIP event between the launch groups on `stream`; stage_ms_h[DDRL_DQN_STAGES] receives
 * the mean milliseconds of: 0 input staging (nothing when the rows are read in place), 1 layer-1 forward of all evaluations (+ split-K
 * reduce), 2 layer-2 forward, 3 the head launch (Q of every evaluation, backup / loss / dQ, head dgrad), 4 and 5 unused (0),
 * 6 layer-2 dgrad + wgrad + head wgrad, 7 layer-1 wgrad, 8 flat Adam + polyak.  Synchronises
 * `stream`.  Measurement aid of bench.py's config-5 roofline block (the reference has no counterpart). */
#define DDRL_DQN_STAGES 9
int ddrl_dqn_step_timed(ddrl_dqn_t *h, const float *obs1_d, const float *obs2_d, const float *acts_d, const float *rews_d,
                        const float *done_d, int reps, float *stage_ms_h, void *stream);
/* self.q of the main network (SQN: q1) for n <= batch observations (Actor.get_action, actor_learner.py:193-198). */
int ddrl_dqn_q(ddrl_dqn_t *h, const float *obs_d, int64_t n, float *q_d, void *stream);

/* ===================================================================================== */
/* Rollout-side window queues of the n-step driver: per env, o_queue = deque(maxlen=Ln+1) of  */
/* observations and a_r_d_queue = deque(maxlen=Ln) of (a, r, d) — algos/sac1/sac_ray.py:192-248 */
/* ===================================================================================== */
typedef struct ddrl_winq ddrl_winq_t;
int ddrl_winq_create(ddrl_winq_t **out, int device, int64_t n_envs, int32_t Ln, int32_t obs_dim, int32_t act_dim,
                     int32_t save_freq);
int ddrl_winq_destroy(ddrl_winq_t *h);
/* Episode start for every env (mask_d NULL) or the envs with mask_d[i] != 0 (sac_ray.py:199-207):
 * o_queue = [obs_d[i]], t_queue = 1. */
int ddrl_winq_begin(ddrl_winq_t *h, const uint8_t *mask_d, const float *obs_d, void *stream);
/* After env.step (sac_ray.py:229-248): a_r_d_queue.append((a, r, d)); o_queue.append((o2,));
 * ready_d[i] = (t_queue >= Ln and t_queue % save_freq == 0); t_queue += 1.  ready_d (device uint8[n])
 * is the mask of `replay_buffer.store(o_queue, a_r_d_queue)`: pass it with ddrl_winq_buffers' arrays to
 * ddrl_replay_store_masked_ex of a ring created with widths {(Ln+1)*obs, Ln*act, Ln, Ln}. */
int ddrl_winq_push(ddrl_winq_t *h, const float *obs2_d, const float *act_d, const float *rew_d, const float *done_d,
                   uint8_t *ready_d, void *stream);
/* Device pointers of the window arrays (o, a, r, d; row = env) and of t_queue (int32[n]). */
int ddrl_winq_buffers(ddrl_winq_t *h, float **arrays_h, int32_t **t_queue_h);

/* ===================================================================================== */
/* Counter-based noise (stands in for tf.random_normal, core.py:77, and                    */
/* env.action_space.sample(), example/dsac.py:99) — same generator as oracle/noise_oracle  */
/* ===================================================================================== */
/* out_d[i] = N(0,1) (Box–Muller over hash(seed, counter+i)); uniform: lo + (hi-lo)*U[0,1). */
int ddrl_normal_fill(float *out_d, int64_t n, uint32_t seed, uint64_t counter, void *stream);
int ddrl_uniform_fill(float *out_d, int64_t n, float lo, float hi, uint32_t seed, uint64_t counter,
                      void *stream);

/* ===================================================================================== */
/* comm_*: the path's cross-process traffic as RCCL calls (one process per GPU, xGMI), for  */
/* hosts that bind this library without PyTorch — SURVEY 8(b), last row.  The Python package */
/* issues the same collectives through torch.distributed (backend "nccl" = RCCL).           */
/* RCCL is bound at run time: a copy the process already holds is shared, else librccl.so.1  */
/* (DDRL_RCCL_PATH overrides).  All buffers are DEVICE float32; calls are asynchronous on    */
/* `stream`; every rank of the communicator makes the matching call.                          */
/* ===================================================================================== */
typedef struct ddrl_comm ddrl_comm_t;
#define DDRL_COMM_ID_BYTES 128
/* Rank 0 creates the communicator id and hands the 128 bytes to every other rank out of band (the launcher's
 * rendezvous: a file, an environment variable, MPI ...); then EVERY rank calls ddrl_comm_init with it (collective). */
int ddrl_comm_unique_id(uint8_t *id_h);
int ddrl_comm_init(ddrl_comm_t **out, int device, int32_t rank, int32_t world, const uint8_t *id_h);
int ddrl_comm_destroy(ddrl_comm_t *h);
/* ps.push(keys, values) on the learner + ps.pull(keys) on every worker (example/dsac.py:59-65; every 300 updates,
 * algos/sac1/sac1.py:149): ONE broadcast of the flat parameter vector (ddrl_sac1_get_weights / ddrl_ps layout), in place. */
int ddrl_comm_bcast_params(ddrl_comm_t *h, float *flat_d, int64_t n, int32_t root, void *stream);
/* Data-parallel learners (example/dsac.py:233 runs num_learners of them unsynchronised; here synchronous): mean of the flat
 * gradient over the communicator's ranks, in place (ddrl_sac1_grad_buffer's vector between compute_grads and apply_grads). */
int ddrl_comm_allreduce_grads(ddrl_comm_t *h, float *flat_d, int64_t n, void *stream);
/* The block of batches a shard owner drew for a learner's step (ddrl_replay_sample_many -> ddrl_replay_set_feed): the reply of
 * `replay_buffer[i].sample_batch.remote()` (algos/sac1/sac_ray.py:137-141), one point-to-point message per (owner, learner). */
int ddrl_comm_send_batch(ddrl_comm_t *h, const float *block_d, int64_t n, int32_t peer, void *stream);
int ddrl_comm_recv_batch(ddrl_comm_t *h, float *block_d, int64_t n, int32_t peer, void *stream);
/* ncclGroupStart / ncclGroupEnd: several sends / receives as one operation (a rank that both serves and learns; a send to self). */
int ddrl_comm_group_start(void);
int ddrl_comm_group_end(void);

#ifdef __cplusplus
}
#endif
#endif /* DDRL_H */
