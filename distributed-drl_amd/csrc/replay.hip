// Replay ring buffer on HBM: store (ring append with wrap), bit-exact NumPy-compatible MT19937
// uniform index sampler, row gather.  Replaces class ReplayBuffer of the reference
// (example/dsac.py:14-48, algos/sac1/sac1.py:28-63, algos/dqn/train.py:37-76).
//
// Data layout: the reference's own struct-of-arrays — obs1[N,obs] obs2[N,obs] acts[N,act] (or
// [N]) rews[N] done[N], float32, row-major — so the rings are byte-compatible with the .npy
// checkpoint of algos/dqn/train.py:82-90.  Cursor/counters and the MT19937 state live in a
// device-side RingState so that store/sample kernels can be captured in a hipGraph and replayed
// without host-side argument patching.
#include "replay_device.h"

using namespace ddrl_replay_dev;

namespace {

// ------------------------------------------------------------------------------------------
// store: n sequential ReplayBuffer.store() calls (example/dsac.py:29-37)
// grid = (blocks, 5 arrays).  Rows i < n - capacity would be overwritten later in the same
// batch, so they are skipped; every remaining destination row is unique.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_store(RingState *st, RingPtrs ring, const float *obs, const float *act,
                                               const float *rew, const float *obs2, const float *done,
                                               long long n) {
    __shared__ long long s_ptr;
    if (threadIdx.x == 0) s_ptr = st->ptr;
    __syncthreads();
    const long long ptr = s_ptr, cap = ring.capacity;
    const long long skip = n > cap ? n - cap : 0;
    const int which = blockIdx.y;
    const float *src;
    float *dst;
    int width;
    switch (which) {
        case 0: src = obs; dst = ring.obs1; width = ring.obs_dim; break;
        case 1: src = obs2; dst = ring.obs2; width = ring.obs_dim; break;
        case 2: src = act; dst = ring.acts; width = ring.act_dim; break;
        case 3: src = rew; dst = ring.rews; width = 1; break;
        default: src = done; dst = ring.done; width = 1; break;
    }
    // rows [skip, n) land on distinct ring rows; ptr < cap and i - skip < cap, so one conditional
    // subtraction replaces the modulo.  32-bit index arithmetic whenever the batch allows it
    // (64-bit div/mod per element made this kernel 10x slower than the copy it is).
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool v4 = (width & 3) == 0;
    const int wv = v4 ? width >> 2 : width;  // row width in vector elements
    const long long total = (n - skip) * wv;
    const long long base_row = ptr + skip;   // ring row of source row `skip` before wrapping (< 2 * cap)
    if (total < 0x7fffffffll) {
        const unsigned tot = (unsigned)total, uw = (unsigned)wv, ustride = (unsigned)stride;
        for (unsigned e = (unsigned)t0; e < tot; e += ustride) {
            const unsigned ri = e / uw, c = e - ri * uw;
            long long row = base_row + ri;
            if (row >= cap) row -= cap;
            if (row >= cap) row -= cap;
            const long long so = (long long)(skip + ri) * wv + c, dof = row * wv + c;
            if (v4) reinterpret_cast<float4 *>(dst)[dof] = reinterpret_cast<const float4 *>(src)[so];
            else dst[dof] = src[so];
        }
    } else {
        for (long long e = t0; e < total; e += stride) {
            const long long ri = e / wv;
            const int c = (int)(e - ri * wv);
            const long long row = (base_row + ri) % cap;
            const long long so = (skip + ri) * wv + c, dof = row * wv + c;
            if (v4) reinterpret_cast<float4 *>(dst)[dof] = reinterpret_cast<const float4 *>(src)[so];
            else dst[dof] = src[so];
        }
    }
    // last block to finish advances the cursor (every block has read st->ptr before its ticket)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total_blocks = gridDim.x * gridDim.y;
        const unsigned ticket = atomicAdd(&st->done_counter, 1u);
        if (ticket == total_blocks - 1) {
            st->ptr = (ptr + n) % cap;
            const long long sz = st->size + n;
            st->size = sz > cap ? cap : sz;
            st->steps += n;
            st->done_counter = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// MT19937 (NumPy legacy RandomState): seed == init_genrand
// ------------------------------------------------------------------------------------------
__global__ void k_mt_seed(RingState *st, uint32_t seed) {
    // sequential recurrence; 624 steps on one lane (a few microseconds, off the hot path)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t x = seed;
        st->mt_key[0] = x;
        for (int i = 1; i < MT_N; ++i) {
            x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
            st->mt_key[i] = x;
        }
        st->mt_pos = MT_N;
    }
}

__global__ void __launch_bounds__(SAMPLE_THREADS) k_sample(RingState *st, RingPtrs ring, BatchPtrs out, int B,
                                                           long long *idx_out, int fuse_gather) {
    sample_block(st, ring, out, B, idx_out, fuse_gather);
}

// Stand-alone gather for large rows (the dqn pixel shape: 2 x 112 896 B per index): one
// workgroup per (row, array-slot) so that >= B*3 workgroups fill the chip; each lane keeps four
// independent 16-B loads in flight (HBM-bound random-row gather).
__global__ void __launch_bounds__(256) k_gather(RingPtrs ring, BatchPtrs out, const long long *__restrict__ idx,
                                                int B) {
    const int b = blockIdx.x;
    const long long row = idx[b];
    const int tid = threadIdx.x;
    if (blockIdx.y < 2) {
        const float *src = (blockIdx.y == 0 ? ring.obs1 : ring.obs2) + row * ring.obs_dim;
        float *dst = (blockIdx.y == 0 ? out.obs1 : out.obs2) + (long long)b * ring.obs_dim;
        const int width = ring.obs_dim;
        if ((width & 3) == 0) {
            const int w4 = width >> 2;
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(dst);
            int e = tid;
            for (; e + 3 * 256 < w4; e += 4 * 256) {
                const float4 a0 = s4[e], a1 = s4[e + 256], a2 = s4[e + 512], a3 = s4[e + 768];
                d4[e] = a0; d4[e + 256] = a1; d4[e + 512] = a2; d4[e + 768] = a3;
            }
            for (; e < w4; e += 256) d4[e] = s4[e];
        } else {
            for (int e = tid; e < width; e += 256) dst[e] = src[e];
        }
    } else {
        for (int e = tid; e < ring.act_dim; e += 256)
            out.acts[(long long)b * ring.act_dim + e] = ring.acts[row * ring.act_dim + e];
        if (tid == 0) {
            out.rews[b] = ring.rews[row];
            out.done[b] = ring.done[row];
        }
    }
}

__global__ void k_set_counts(RingState *st, long long ptr, long long size, long long steps, long long samples) {
    st->ptr = ptr; st->size = size; st->steps = steps; st->sample_times = samples;
    st->done_counter = 0; st->error = 0;
}

}  // namespace

struct ddrl_replay {
    int device;
    RingPtrs ring;
    RingState *state;    // device
    long long *idx_buf;  // device scratch for indices when the caller passes none
    long long idx_cap;
    // host mirror of the counters: exact while every store/sample goes through this API
    // eagerly; refreshed from the device by ddrl_replay_counts (graph replays run ahead of it)
    long long h_ptr, h_size, h_steps, h_samples;
    uint32_t flags;
};

static int refresh_counts(ddrl_replay *h, hipStream_t s) {
    RingState tmp;
    DDRL_HIP_CHECK(hipMemcpyAsync(&tmp, h->state, offsetof(RingState, mt_key), hipMemcpyDeviceToHost, s));
    DDRL_HIP_CHECK(hipStreamSynchronize(s));
    h->h_ptr = tmp.ptr; h->h_size = tmp.size; h->h_steps = tmp.steps; h->h_samples = tmp.sample_times;
    return DDRL_OK;
}

ddrl_replay_dev::SamplerView ddrl_replay_sampler_view(ddrl_replay_t *h) { return ddrl_replay_dev::SamplerView{h->state, h->ring}; }
bool ddrl_replay_can_fuse(ddrl_replay_t *h, int64_t batch) {
    const long long bytes = batch * (2ll * h->ring.obs_dim + h->ring.act_dim + 2) * (long long)sizeof(float);
    return batch <= MAX_FUSED_BATCH && bytes <= MAX_FUSED_BYTES && h->h_size > 0;
}
void ddrl_replay_note_sample(ddrl_replay_t *h) { h->h_samples += 1; }

extern "C" {

int ddrl_replay_create(ddrl_replay_t **out, int device, int64_t capacity, int obs_dim, int act_dim,
                       uint32_t flags) {
    DDRL_REQUIRE(out != nullptr, "out is NULL");
    DDRL_REQUIRE(capacity > 0 && capacity <= 0xFFFFFFFFll, "capacity must be in [1, 2^32-1]");
    DDRL_REQUIRE(obs_dim > 0 && act_dim > 0, "obs_dim/act_dim must be positive");
    DDRL_REQUIRE(!(flags & DDRL_REPLAY_ACTS_1D) || act_dim == 1, "ACTS_1D needs act_dim == 1");
    ddrl::DeviceGuard g(device);
    if (!g.ok) { ddrl::set_error("cannot select device %d", device); return DDRL_ERR_HIP; }
    ddrl_replay *h = new ddrl_replay();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->flags = flags;
    h->ring.capacity = capacity;
    h->ring.obs_dim = obs_dim;
    h->ring.act_dim = act_dim;
    const size_t no = (size_t)capacity * obs_dim * sizeof(float), na = (size_t)capacity * act_dim * sizeof(float),
                 n1 = (size_t)capacity * sizeof(float);
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc(&h->ring.obs1, no);
    if (e == hipSuccess) e = hipMalloc(&h->ring.obs2, no);
    if (e == hipSuccess) e = hipMalloc(&h->ring.acts, na);
    if (e == hipSuccess) e = hipMalloc(&h->ring.rews, n1);
    if (e == hipSuccess) e = hipMalloc(&h->ring.done, n1);
    if (e == hipSuccess) e = hipMalloc(&h->state, sizeof(RingState));
    h->idx_cap = 1 << 16;
    if (e == hipSuccess) e = hipMalloc(&h->idx_buf, h->idx_cap * sizeof(long long));
    if (e != hipSuccess) {
        ddrl::set_error("hipMalloc failed for replay of %lld x (%d,%d): %s", (long long)capacity, obs_dim, act_dim,
                        hipGetErrorString(e));
        ddrl_replay_destroy(h);
        return DDRL_ERR_NOMEM;
    }
    // np.zeros for the five rings (example/dsac.py:21-25)
    DDRL_HIP_CHECK(hipMemsetAsync(h->ring.obs1, 0, no, nullptr));
    DDRL_HIP_CHECK(hipMemsetAsync(h->ring.obs2, 0, no, nullptr));
    DDRL_HIP_CHECK(hipMemsetAsync(h->ring.acts, 0, na, nullptr));
    DDRL_HIP_CHECK(hipMemsetAsync(h->ring.rews, 0, n1, nullptr));
    DDRL_HIP_CHECK(hipMemsetAsync(h->ring.done, 0, n1, nullptr));
    DDRL_HIP_CHECK(hipMemsetAsync(h->state, 0, sizeof(RingState), nullptr));
    k_mt_seed<<<1, 64, 0, nullptr>>>(h->state, 0u);
    DDRL_LAUNCH_CHECK();
    DDRL_HIP_CHECK(hipStreamSynchronize(nullptr));
    *out = h;
    return DDRL_OK;
}

int ddrl_replay_destroy(ddrl_replay_t *h) {
    if (!h) return DDRL_OK;
    ddrl::DeviceGuard g(h->device);
    (void)hipFree(h->ring.obs1); (void)hipFree(h->ring.obs2); (void)hipFree(h->ring.acts);
    (void)hipFree(h->ring.rews); (void)hipFree(h->ring.done); (void)hipFree(h->state); (void)hipFree(h->idx_buf);
    delete h;
    return DDRL_OK;
}

int ddrl_replay_seed(ddrl_replay_t *h, uint32_t seed, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    ddrl::DeviceGuard g(h->device);
    k_mt_seed<<<1, 64, 0, ddrl::as_stream(stream)>>>(h->state, seed);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_replay_store(ddrl_replay_t *h, const float *obs_d, const float *act_d, const float *rew_d,
                      const float *obs2_d, const float *done_d, int64_t n, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    DDRL_REQUIRE(n >= 0, "n must be >= 0");
    if (n == 0) return DDRL_OK;
    DDRL_REQUIRE(obs_d && act_d && rew_d && obs2_d && done_d, "NULL transition pointer");
    ddrl::DeviceGuard g(h->device);
    const long long rows = n > h->ring.capacity ? h->ring.capacity : n;
    const long long widest = rows * ((h->ring.obs_dim & 3) == 0 ? h->ring.obs_dim / 4 : h->ring.obs_dim);
    long long blocks = (widest + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    k_store<<<dim3((unsigned)blocks, 5), 256, 0, ddrl::as_stream(stream)>>>(h->state, h->ring, obs_d, act_d, rew_d,
                                                                            obs2_d, done_d, n);
    DDRL_LAUNCH_CHECK();
    h->h_ptr = (h->h_ptr + n) % h->ring.capacity;
    h->h_size = (h->h_size + n > h->ring.capacity) ? h->ring.capacity : h->h_size + n;
    h->h_steps += n;
    return DDRL_OK;
}

static int launch_gather(ddrl_replay *h, const long long *idx, int64_t B, BatchPtrs out, hipStream_t s) {
    k_gather<<<dim3((unsigned)B, 3), 256, 0, s>>>(h->ring, out, idx, (int)B);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_replay_sample(ddrl_replay_t *h, int64_t batch, float *obs1_d, float *obs2_d, float *acts_d,
                       float *rews_d, float *done_d, int64_t *idx_d, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    DDRL_REQUIRE(batch > 0 && batch <= (1 << 24), "batch must be in [1, 2^24]");
    DDRL_REQUIRE(obs1_d && obs2_d && acts_d && rews_d && done_d, "NULL output pointer");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    if (h->h_size <= 0) {
        // the mirror may lag behind graph replays: ask the device before reporting empty
        int rc = refresh_counts(h, s);
        if (rc != DDRL_OK) return rc;
        if (h->h_size <= 0) {
            ddrl::set_error("high <= 0");  // message of the reference's ValueError
            return DDRL_ERR_EMPTY_BUFFER;
        }
    }
    BatchPtrs out{obs1_d, obs2_d, acts_d, rews_d, done_d};
    const long long bytes = batch * (2ll * h->ring.obs_dim + h->ring.act_dim + 2) * (long long)sizeof(float);
    const int fuse = (batch <= MAX_FUSED_BATCH && bytes <= MAX_FUSED_BYTES) ? 1 : 0;
    long long *idx = reinterpret_cast<long long *>(idx_d);
    if (!fuse && !idx) {
        if (batch > h->idx_cap) {
            DDRL_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(h->idx_buf);
            h->idx_cap = batch;
            DDRL_HIP_CHECK(hipMalloc(&h->idx_buf, h->idx_cap * sizeof(long long)));
        }
        idx = h->idx_buf;
    }
    k_sample<<<1, SAMPLE_THREADS, 0, s>>>(h->state, h->ring, out, (int)batch, idx, fuse);
    DDRL_LAUNCH_CHECK();
    h->h_samples += 1;
    if (!fuse) return launch_gather(h, idx, batch, out, s);
    return DDRL_OK;
}

int ddrl_replay_gather(ddrl_replay_t *h, const int64_t *idx_d, int64_t batch, float *obs1_d, float *obs2_d,
                       float *acts_d, float *rews_d, float *done_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && idx_d != nullptr, "NULL handle or index pointer");
    DDRL_REQUIRE(batch > 0 && batch <= (1 << 24), "batch must be in [1, 2^24]");
    DDRL_REQUIRE(obs1_d && obs2_d && acts_d && rews_d && done_d, "NULL output pointer");
    ddrl::DeviceGuard g(h->device);
    BatchPtrs out{obs1_d, obs2_d, acts_d, rews_d, done_d};
    return launch_gather(h, reinterpret_cast<const long long *>(idx_d), batch, out, ddrl::as_stream(stream));
}

int ddrl_replay_counts(ddrl_replay_t *h, int64_t *ptr_h, int64_t *size_h, int64_t *steps_h,
                       int64_t *sample_times_h, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    ddrl::DeviceGuard g(h->device);
    int rc = refresh_counts(h, ddrl::as_stream(stream));
    if (rc != DDRL_OK) return rc;
    if (ptr_h) *ptr_h = h->h_ptr;
    if (size_h) *size_h = h->h_size;
    if (steps_h) *steps_h = h->h_steps;
    if (sample_times_h) *sample_times_h = h->h_samples;
    return DDRL_OK;
}

int ddrl_replay_buffers(ddrl_replay_t *h, float **obs1_d, float **obs2_d, float **acts_d, float **rews_d,
                        float **done_d) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    if (obs1_d) *obs1_d = h->ring.obs1;
    if (obs2_d) *obs2_d = h->ring.obs2;
    if (acts_d) *acts_d = h->ring.acts;
    if (rews_d) *rews_d = h->ring.rews;
    if (done_d) *done_d = h->ring.done;
    return DDRL_OK;
}

int ddrl_replay_set_counts(ddrl_replay_t *h, int64_t ptr, int64_t size, int64_t steps, int64_t sample_times,
                           void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    DDRL_REQUIRE(ptr >= 0 && ptr < h->ring.capacity && size >= 0 && size <= h->ring.capacity, "ptr/size out of range");
    ddrl::DeviceGuard g(h->device);
    k_set_counts<<<1, 1, 0, ddrl::as_stream(stream)>>>(h->state, ptr, size, steps, sample_times);
    DDRL_LAUNCH_CHECK();
    h->h_ptr = ptr; h->h_size = size; h->h_steps = steps; h->h_samples = sample_times;
    return DDRL_OK;
}

int ddrl_replay_mt_state(ddrl_replay_t *h, uint32_t *key_h, int32_t *pos_h, void *stream) {
    DDRL_REQUIRE(h != nullptr && key_h != nullptr && pos_h != nullptr, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    RingState tmp;
    hipStream_t s = ddrl::as_stream(stream);
    DDRL_HIP_CHECK(hipMemcpyAsync(&tmp, h->state, sizeof(RingState), hipMemcpyDeviceToHost, s));
    DDRL_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(key_h, tmp.mt_key, sizeof(tmp.mt_key));
    *pos_h = tmp.mt_pos;
    return DDRL_OK;
}

}  // extern "C"
