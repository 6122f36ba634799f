// Learner hot loop: `while True: batch = replay_buffer.sample_batch(B); agent.train(batch)`
// (algos/sac1/sac1.py:146-148; example/dsac.py:142-144 with example/model.py:92-101), n iterations
// per call with no host work per update.  Built on the public C-ABI only: the replay gathers
// straight into the learner's input buffers, the noise comes from the learner's device counter,
// and because every cursor / RNG state / optimizer state lives on the device the sequence for
// `updates_per_graph` updates is captured ONCE into a hipGraph and replayed (an eager launch
// stream would be host-bound at ~3.5 us per kernel, MI355X guide "graph-replay-floor").
#include "ddrl_common.h"

struct ddrl_loop {
    ddrl_sac1_t *learner;
    ddrl_replay_t *replay;
    int per_graph;
    uint32_t seed;
    float *buf[8];
    int batch;
    hipGraphExec_t exec;
    bool captured;
};

static int one_update(ddrl_loop *h, void *stream) {
    int rc = ddrl_replay_sample(h->replay, h->batch, h->buf[0], h->buf[1], h->buf[2], h->buf[3], h->buf[4], nullptr, stream);
    if (rc != DDRL_OK) return rc;
    rc = ddrl_sac1_fill_noise(h->learner, h->seed, stream);
    if (rc != DDRL_OK) return rc;
    return ddrl_sac1_step(h->learner, h->buf[0], h->buf[1], h->buf[2], h->buf[3], h->buf[4], h->buf[5], h->buf[6], h->buf[7],
                          nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" {

int ddrl_loop_create(ddrl_loop_t **out, ddrl_sac1_t *learner, ddrl_replay_t *replay, int32_t updates_per_graph,
                     uint32_t noise_seed) {
    DDRL_REQUIRE(out && learner && replay, "NULL pointer");
    DDRL_REQUIRE(updates_per_graph >= 0 && updates_per_graph <= 4096, "updates_per_graph must be in [0, 4096]");
    ddrl_loop *h = new ddrl_loop();
    h->learner = learner; h->replay = replay; h->per_graph = updates_per_graph; h->seed = noise_seed;
    h->exec = nullptr; h->captured = false;
    int rc = ddrl_sac1_input_buffers(learner, h->buf);
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

}  // extern "C"

extern "C" int ddrl_loop_run(ddrl_loop_t *h, int64_t n_updates, void *stream) {
    DDRL_REQUIRE(h != nullptr && n_updates >= 0, "bad handle / n_updates");
    hipStream_t s = ddrl::as_stream(stream);
    int64_t left = n_updates;
    if (h->per_graph > 0 && left >= h->per_graph) {
        if (!h->captured) {
            // one eager update first: surfaces EMPTY_BUFFER / argument errors outside the capture
            int rc = one_update(h, stream);
            if (rc != DDRL_OK) return rc;
            left -= 1;
            hipStream_t cs = s;
            hipStream_t own = nullptr;
            if (cs == nullptr) {  // the legacy null stream cannot be captured
                DDRL_HIP_CHECK(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
                DDRL_HIP_CHECK(hipDeviceSynchronize());
                cs = own;
            }
            hipGraph_t graph = nullptr;
            DDRL_HIP_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            int rc2 = DDRL_OK;
            for (int i = 0; i < h->per_graph && rc2 == DDRL_OK; ++i) rc2 = one_update(h, (void *)cs);
            hipError_t e = hipStreamEndCapture(cs, &graph);
            if (own) (void)hipStreamDestroy(own);
            if (rc2 != DDRL_OK) { if (graph) (void)hipGraphDestroy(graph); return rc2; }
            if (e != hipSuccess) { ddrl::set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return DDRL_ERR_HIP; }
            DDRL_HIP_CHECK(hipGraphInstantiate(&h->exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            h->captured = true;
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
