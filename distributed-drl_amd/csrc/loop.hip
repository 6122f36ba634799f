// Learner hot loop: `while True: batch = replay_buffer.sample_batch(B); agent.train(batch)`
// (algos/sac1/sac1.py:146-148; example/dsac.py:142-144 with example/model.py:92-101), n iterations
// per call with no host work per update.  Built on the public C-ABI only: the replay gathers
// straight into one of the learner's two input sets, the noise comes from the learner's device
// counter, and because every cursor / RNG state / optimizer state lives on the device the
// sequence for `updates_per_graph` updates is captured ONCE into a hipGraph and replayed (an
// eager launch stream would be host-bound at ~3.5 us per kernel, MI355X guide
// "graph-replay-floor").  Consecutive updates alternate between the learner's two input sets, so
// the sampler of update u+1 never overwrites what update u still reads; optionally
// (DDRL_LOOP_FORK=1) the sampler runs on a forked graph branch and overlaps the previous update —
// the job of the reference's `Cache` prefetch process (algos/sac1/sac1.py:103-130).
#include "ddrl_common.h"

#include <cstdlib>
#include <cstring>
#include <vector>

// internal (sac1.hip): put the learner's double-buffered optimizer state on copy 0
int ddrl_sac1_internal_opt_sync(ddrl_sac1_t *h, void *stream);

struct ddrl_loop {
    ddrl_sac1_t *learner;
    ddrl_replay_t *replay;
    int per_graph;
    uint32_t seed;
    float *buf[2][8];
    int batch;
    hipGraphExec_t exec;
    bool captured;
    int parity;  // input set of the next eager update
};

static int sample_into(ddrl_loop *h, int set, void *stream) {
    float **b = h->buf[set];
    return ddrl_replay_sample(h->replay, h->batch, b[0], b[1], b[2], b[3], b[4], nullptr, stream);
}

static int update_from(ddrl_loop *h, int set, void *stream) {
    float **b = h->buf[set];
    int rc = ddrl_sac1_fill_noise(h->learner, h->seed, stream);  // arms in-kernel noise generation
    if (rc != DDRL_OK) return rc;
    return ddrl_sac1_step(h->learner, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], nullptr, nullptr, nullptr, nullptr, stream);
}

static int one_update(ddrl_loop *h, void *stream) {
    const int set = h->parity;
    h->parity ^= 1;
    int rc = sample_into(h, set, stream);
    if (rc != DDRL_OK) return rc;
    return update_from(h, set, stream);
}

static int grads_from(ddrl_loop *h, int set, void *stream) {
    float **b = h->buf[set];
    int rc = ddrl_sac1_fill_noise(h->learner, h->seed, stream);
    if (rc != DDRL_OK) return rc;
    return ddrl_sac1_compute_grads(h->learner, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], nullptr, nullptr, nullptr, nullptr,
                                   stream);
}

// Capture `per_graph` updates.  Default: one branch; the sampler of update u+1 rides inside update u
// (ddrl_sac1_step_and_sample: an extra workgroup of a forward launch on the fused path, of the Adam
// kernel otherwise).
// Experimental overlap modes (bit-identical results, both measured SLOWER on MI355X — a kernel
// starting or ending on another branch costs the kernel running beside it more than it hides):
//   DDRL_LOOP_FORK=adam  sampler of update u+1 beside the Adam/polyak kernel of update u (105 us)
//   DDRL_LOOP_FORK=all   sampler of update u+1 anywhere beside update u (99 us)
static int capture(ddrl_loop *h, hipStream_t main_s) {
    const char *fk = getenv("DDRL_LOOP_FORK");
    const int mode = !fk ? 0 : (strcmp(fk, "all") == 0 ? 2 : 1);
    hipStream_t side = nullptr;
    DDRL_HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    const int n = h->per_graph;
    std::vector<hipEvent_t> e_smp(n + 1), e_upd(n + 1), e_grad(n + 1);
    hipEvent_t e_fork;
    DDRL_HIP_CHECK(hipEventCreateWithFlags(&e_fork, hipEventDisableTiming));
    for (int i = 0; i <= n; ++i) {
        DDRL_HIP_CHECK(hipEventCreateWithFlags(&e_smp[i], hipEventDisableTiming));
        DDRL_HIP_CHECK(hipEventCreateWithFlags(&e_upd[i], hipEventDisableTiming));
        DDRL_HIP_CHECK(hipEventCreateWithFlags(&e_grad[i], hipEventDisableTiming));
    }
    hipGraph_t graph = nullptr;
    int rc = ddrl_sac1_internal_opt_sync(h->learner, (void *)main_s);  // the graph starts on copy 0 of the optimizer state ...
    if (rc != DDRL_OK) return rc;
    DDRL_HIP_CHECK(hipStreamSynchronize(main_s));  // one-time: the capture stream may not be the caller's
    DDRL_HIP_CHECK(hipStreamBeginCapture(main_s, hipStreamCaptureModeThreadLocal));
    hipError_t e = hipSuccess;
#define HE(x) do { if (e == hipSuccess && rc == DDRL_OK) e = (x); } while (0)
#define RC(x) do { if (e == hipSuccess && rc == DDRL_OK) rc = (x); } while (0)
    if (mode == 0) {
        // sample(0) as a kernel; sample(i+1) rides inside the Adam kernel of update i.  No store can
        // interleave inside one graph, so this equals the sequential sample -> update order.
        RC(sample_into(h, 0, (void *)main_s));
        for (int i = 0; i < n; ++i) {
            RC(ddrl_sac1_fill_noise(h->learner, h->seed, (void *)main_s));
            if (i + 1 < n) {
                RC(ddrl_sac1_step_and_sample(h->learner, i & 1, h->replay, (i + 1) & 1, (void *)main_s));
            } else {
                float **b = h->buf[i & 1];
                RC(ddrl_sac1_step(h->learner, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], nullptr, nullptr, nullptr, nullptr,
                                  (void *)main_s));
            }
        }
    } else if (mode == 2) {
        HE(hipEventRecord(e_fork, main_s));
        HE(hipStreamWaitEvent(side, e_fork, 0));
        for (int i = 0; i < n; ++i) {
            if (i >= 2) HE(hipStreamWaitEvent(side, e_upd[i - 2], 0));  // set i&1 was last read by update i-2
            RC(sample_into(h, i & 1, (void *)side));
            HE(hipEventRecord(e_smp[i], side));
            HE(hipStreamWaitEvent(main_s, e_smp[i], 0));
            RC(update_from(h, i & 1, (void *)main_s));
            HE(hipEventRecord(e_upd[i], main_s));
        }
    } else {
        // sample(0) inline; then for each update: grads(i) -> [fork: sample(i+1)] || apply(i) -> join
        RC(sample_into(h, 0, (void *)main_s));
        for (int i = 0; i < n; ++i) {
            RC(grads_from(h, i & 1, (void *)main_s));
            if (i + 1 < n) {
                HE(hipEventRecord(e_grad[i], main_s));
                HE(hipStreamWaitEvent(side, e_grad[i], 0));
                RC(sample_into(h, (i + 1) & 1, (void *)side));
                HE(hipEventRecord(e_smp[i + 1], side));
            }
            RC(ddrl_sac1_apply_grads(h->learner, (void *)main_s));
            if (i + 1 < n) HE(hipStreamWaitEvent(main_s, e_smp[i + 1], 0));
        }
    }
    RC(ddrl_sac1_internal_opt_sync(h->learner, (void *)main_s));  // ... and ends on it (a copy node when n is odd)
#undef HE
#undef RC
    // every side-branch node is an ancestor of a main-stream node: the branch is joined
    hipError_t e2 = hipStreamEndCapture(main_s, &graph);
    for (int i = 0; i <= n; ++i) { (void)hipEventDestroy(e_smp[i]); (void)hipEventDestroy(e_upd[i]); (void)hipEventDestroy(e_grad[i]); }
    (void)hipEventDestroy(e_fork);
    (void)hipStreamDestroy(side);
    if (rc != DDRL_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || e2 != hipSuccess) {
        ddrl::set_error("graph capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        if (graph) (void)hipGraphDestroy(graph);
        return DDRL_ERR_HIP;
    }
    DDRL_HIP_CHECK(hipGraphInstantiate(&h->exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    h->captured = true;
    return DDRL_OK;
}

extern "C" {

int ddrl_loop_create(ddrl_loop_t **out, ddrl_sac1_t *learner, ddrl_replay_t *replay, int32_t updates_per_graph,
                     uint32_t noise_seed) {
    DDRL_REQUIRE(out && learner && replay, "NULL pointer");
    DDRL_REQUIRE(updates_per_graph >= 0 && updates_per_graph <= 4096, "updates_per_graph must be in [0, 4096]");
    ddrl_loop *h = new ddrl_loop();
    h->learner = learner; h->replay = replay; h->per_graph = updates_per_graph; h->seed = noise_seed;
    h->exec = nullptr; h->captured = false; h->parity = 0;
    int rc = ddrl_sac1_input_buffers(learner, 0, h->buf[0]);
    if (rc == DDRL_OK) rc = ddrl_sac1_input_buffers(learner, 1, h->buf[1]);
    if (rc != DDRL_OK) { delete h; return rc; }
    h->batch = ddrl_sac1_batch(learner);
    *out = h;
    return DDRL_OK;
}

int ddrl_loop_destroy(ddrl_loop_t *h) {
    if (!h) return DDRL_OK;
    if (h->exec) (void)hipGraphExecDestroy(h->exec);
    delete h;
    return DDRL_OK;
}

int ddrl_loop_run(ddrl_loop_t *h, int64_t n_updates, void *stream) {
    DDRL_REQUIRE(h != nullptr && n_updates >= 0, "bad handle / n_updates");
    hipStream_t s = ddrl::as_stream(stream);
    int64_t left = n_updates;
    if (h->per_graph > 0 && left >= h->per_graph) {
        if (!h->captured) {
            // one eager update first: surfaces EMPTY_BUFFER / argument errors outside the capture
            int rc = one_update(h, stream);
            if (rc != DDRL_OK) return rc;
            left -= 1;
            hipStream_t cs = s, own = nullptr;
            if (cs == nullptr) {  // the legacy null stream cannot be captured
                DDRL_HIP_CHECK(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
                DDRL_HIP_CHECK(hipDeviceSynchronize());
                cs = own;
            }
            rc = capture(h, cs);
            if (own) (void)hipStreamDestroy(own);
            if (rc != DDRL_OK) return rc;
        }
        if (left >= h->per_graph) {  // eager updates since the last replay may have left the optimizer state on copy 1
            const int rc2 = ddrl_sac1_internal_opt_sync(h->learner, stream);
            if (rc2 != DDRL_OK) return rc2;
        }
        while (left >= h->per_graph) {
            DDRL_HIP_CHECK(hipGraphLaunch(h->exec, s));
            left -= h->per_graph;
        }
    }
    for (; left > 0; --left) {
        int rc = one_update(h, stream);
        if (rc != DDRL_OK) return rc;
    }
    return DDRL_OK;
}

}  // extern "C"
