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
struct StoreSrc {
    const float *a[MAX_ARRAYS];
};
__global__ void __launch_bounds__(256) k_store(RingState *st, RingPtrs ring, StoreSrc srcs, long long n) {
    __shared__ long long s_ptr;
    if (threadIdx.x == 0) s_ptr = st->ptr;
    __syncthreads();
    const long long ptr = s_ptr, cap = ring.capacity;
    const long long skip = n > cap ? n - cap : 0;
    const int which = blockIdx.y;
    const float *src = srcs.a[which];
    float *dst = ring.a[which];
    const int width = ring.w[which];
    // rows [skip, n) land on distinct ring rows; ptr < cap and i - skip < cap, so one conditional
    // subtraction replaces the modulo.  32-bit index arithmetic whenever the batch allows it
    // (64-bit div/mod per element made this kernel 10x slower than the copy it is).
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ring.kind[which] == 1) {
        // compact array: 16 float32 in -> 16 bytes out per lane (4 when the row is not a whole number of 16-element groups);
        // a value that does not survive the round trip (not an integer in [0, 255], NaN) raises the sticky error — the ring
        // is opt-in for integer-valued pixels and must never change a value silently
        unsigned char *d8 = reinterpret_cast<unsigned char *>(dst);
        const int G = ((width & 15) == 0 && aligned16(dst, src)) ? 16 : 1;
        const int wg = width / G;
        const long long total = (n - skip) * wg;
        const long long base_row = (ptr + skip) % cap;
        bool bad = false;
        for (long long e = t0; e < total; e += stride) {
            const long long ri = e / wg;
            const int c = (int)(e - ri * wg);
            long long row = base_row + ri;
            if (row >= cap) row -= cap;
            const long long so = (skip + ri) * width + (long long)c * G, dof = row * width + (long long)c * G;
            if (G == 16) {
                const float4 *s4 = reinterpret_cast<const float4 *>(src + so);
                const float4 v0 = s4[0], v1 = s4[1], v2 = s4[2], v3 = s4[3];
                const float f[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
                unsigned u[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const unsigned b = (unsigned)(int)fminf(fmaxf(f[k], 0.f), 255.f);
                    bad |= !((float)b == f[k]);
                    u[k >> 2] |= b << (8 * (k & 3));
                }
                *reinterpret_cast<uint4 *>(d8 + dof) = make_uint4(u[0], u[1], u[2], u[3]);
            } else {
                const float f = src[so];
                const unsigned b = (unsigned)(int)fminf(fmaxf(f, 0.f), 255.f);
                bad |= !((float)b == f);
                d8[dof] = (unsigned char)b;
            }
        }
        if (bad) st->error = DDRL_ERR_NOT_REPRESENTABLE;
    } else {
    const bool v4 = (width & 3) == 0 && aligned16(dst, src);
    const int wv = v4 ? width >> 2 : width;  // row width in vector elements
    const long long total = (n - skip) * wv;
    const long long base_row = (ptr + skip) % cap;  // ring row of source row `skip` (skip = n - cap is unbounded: reduce it once per thread)
    if (total < 0x7fffffffll) {
        const unsigned tot = (unsigned)total, uw = (unsigned)wv, ustride = (unsigned)stride;
        for (unsigned e = (unsigned)t0; e < tot; e += ustride) {
            const unsigned ri = e / uw, c = e - ri * uw;
            long long row = base_row + ri;  // base_row < cap and ri < cap: one conditional subtraction is the modulo
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
            st->steps += n * ring.steps_inc;
            st->done_counter = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// masked store: store() for the rows with mask != 0, in row order — the n-step rollout stores a
// window only for the envs whose deque is full (algos/sac1/sac_ray.py:243-246), so the number of
// rows is known on the device only.  Pass 1 (one workgroup) ranks the selected rows, pass 2 copies.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mask_scan(const uint8_t *__restrict__ mask, long long n, int *__restrict__ rank) {
    __shared__ int s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (long long i0 = 0; i0 < n; i0 += 256) {
        const long long i = i0 + tid;
        const bool f = i < n && mask[i] != 0;
        const unsigned long long b = __ballot(f);
        const int pre = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wave] = __popcll(b);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { woff += (w < wave) ? s_w[w] : 0; tot += s_w[w]; }
        if (i < n) rank[i] = f ? base + woff + pre : -1;
        base += tot;
        __syncthreads();
    }
    if (tid == 0) rank[n] = base;
}
__global__ void __launch_bounds__(256) k_store_masked(RingState *st, RingPtrs ring, StoreSrc srcs, const int *__restrict__ rank, long long n) {
    __shared__ long long s_ptr;
    if (threadIdx.x == 0) s_ptr = st->ptr;
    __syncthreads();
    const long long ptr = s_ptr, cap = ring.capacity;
    const long long total = rank[n];
    const long long skip = total > cap ? total - cap : 0;  // ranks below `skip` would be overwritten later in this batch
    const int which = blockIdx.y;
    const float *src = srcs.a[which];
    float *dst = ring.a[which];
    const int width = ring.w[which];
    const bool v4 = (width & 3) == 0 && aligned16(dst, src);
    const int wv = v4 ? width >> 2 : width;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n * wv; e += stride) {
        const long long ri = e / wv;
        const int c = (int)(e - ri * wv);
        const long long r = rank[ri];
        if (r < skip) continue;  // not selected (-1) or overwritten within the batch
        const long long row = (ptr + r) % cap;
        if (v4) reinterpret_cast<float4 *>(dst)[row * wv + c] = reinterpret_cast<const float4 *>(src)[e];
        else dst[row * wv + c] = src[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total_blocks = gridDim.x * gridDim.y;
        const unsigned ticket = atomicAdd(&st->done_counter, 1u);
        if (ticket == total_blocks - 1) {
            st->ptr = (ptr + total) % cap;
            const long long sz = st->size + total;
            st->size = sz > cap ? cap : sz;
            st->steps += total * ring.steps_inc;
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

// large draws that are gathered by a launch of their own: the 640-thread form of the sampler (>= 624: a whole state block per pass)
__global__ void __launch_bounds__(640) k_sample_wide(RingState *st, RingPtrs ring, BatchPtrs out, int B, long long *idx_out) {
    sample_block<640>(st, ring, out, B, idx_out, 0);
}

// Stand-alone gather for large rows (the dqn pixel shape: 2 x 112 896 B per index): one
// workgroup per (row, array) so that >= B * n_arr workgroups fill the chip; each lane keeps four
// independent 16-B loads in flight (HBM-bound random-row gather).
__global__ void __launch_bounds__(256) k_gather(RingPtrs ring, BatchPtrs out, const long long *__restrict__ idx,
                                                int B) {
    const int b = blockIdx.x, j = blockIdx.y;
    const long long row = idx[b];
    const int tid = threadIdx.x;
    const int width = ring.w[j];
    const float *src = ring.a[j] + row * width;
    float *dst = out.a[j] + (long long)b * width;
    if (ring.kind[j] == 1) {
        // compact array: one dword (four pixels) in -> one float4 out per lane, so that a wave's store instruction writes 1 KB of
        // consecutive bytes (the first cut — a 16-byte load and four float4 stores at a 64-byte lane stride — wrote partial lines:
        // 2.9 TB/s); eight loads in flight per lane.  The batch written is 4x the bytes read: the stores set the rate.
        const unsigned char *s8 = reinterpret_cast<const unsigned char *>(ring.a[j]) + row * width;
        if ((width & 3) == 0 && aligned16(out.a[j], ring.a[j])) {
            const int w4 = width >> 2;
            const unsigned *s32 = reinterpret_cast<const unsigned *>(s8);
            float4 *d4 = reinterpret_cast<float4 *>(dst);
            auto cvt = [](unsigned u) { return make_float4((float)(u & 255u), (float)((u >> 8) & 255u), (float)((u >> 16) & 255u), (float)(u >> 24)); };
            int e = tid;
            for (; e + 7 * 256 < w4; e += 8 * 256) {
                unsigned u[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) u[k] = s32[e + 256 * k];
#pragma unroll
                for (int k = 0; k < 8; ++k) d4[e + 256 * k] = cvt(u[k]);
            }
            for (; e < w4; e += 256) d4[e] = cvt(s32[e]);
        } else {
            for (int e = tid; e < width; e += 256) dst[e] = (float)s8[e];
        }
        return;
    }
    if ((width & 3) == 0 && aligned16(out.a[j], ring.a[j])) {
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
}

// Gather for many SMALL rows (the block a shard owner draws for a whole step: 2^18 rows of 32 B): one lane per 16-B
// (or 4-B) element, all rows of an array in one grid-stride sweep — k_gather's workgroup-per-row would be launch-bound.
__global__ void __launch_bounds__(256) k_gather_small(RingPtrs ring, BatchPtrs out, const long long *__restrict__ idx, int B) {
    const int j = blockIdx.y, width = ring.w[j];
    if (ring.kind[j] == 1) {
        const unsigned char *r8 = reinterpret_cast<const unsigned char *>(ring.a[j]);
        const unsigned total = (unsigned)B * (unsigned)width;
        for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
            const unsigned b = e / (unsigned)width, c = e - b * (unsigned)width;
            out.a[j][e] = (float)r8[idx[b] * width + c];
        }
        return;
    }
    const bool v4 = (width & 3) == 0 && aligned16(out.a[j], ring.a[j]);  // packed blocks start arrays at any float offset
    const unsigned wv = v4 ? width >> 2 : width, total = (unsigned)B * wv;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const unsigned b = e / wv, c = e - b * wv;
        const long long so = idx[b] * wv + c;
        if (v4) reinterpret_cast<float4 *>(out.a[j])[e] = reinterpret_cast<const float4 *>(ring.a[j])[so];
        else out.a[j][e] = ring.a[j][so];
    }
}

// Ring rows [row0, row0 + nrows) of one array as float32 (export) or from float32 (import): the .npy checkpoint of
// algos/dqn/train.py:82-108 is float32 whatever the storage kind.
__global__ void __launch_bounds__(256) k_rows_export(RingPtrs ring, int j, long long row0, long long nrows, float *__restrict__ out) {
    const long long total = nrows * ring.w[j], off = row0 * ring.w[j];
    const unsigned char *r8 = reinterpret_cast<const unsigned char *>(ring.a[j]);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256)
        out[e] = ring.kind[j] == 1 ? (float)r8[off + e] : ring.a[j][off + e];
}
__global__ void __launch_bounds__(256) k_rows_import(RingState *st, RingPtrs ring, int j, long long row0, long long nrows, const float *__restrict__ src) {
    const long long total = nrows * ring.w[j], off = row0 * ring.w[j];
    unsigned char *r8 = reinterpret_cast<unsigned char *>(ring.a[j]);
    bool bad = false;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const float f = src[e];
        if (ring.kind[j] == 1) {
            const unsigned b = (unsigned)(int)fminf(fmaxf(f, 0.f), 255.f);
            bad |= !((float)b == f);
            r8[off + e] = (unsigned char)b;
        } else {
            ring.a[j][off + e] = f;
        }
    }
    if (bad) st->error = DDRL_ERR_NOT_REPRESENTABLE;
}

__global__ void k_set_feed(RingState *st, Feed f) { st->feed = f; }
__global__ void k_take_error(RingState *st, int *out) { *out = st->error; st->error = 0; }
__global__ void k_add_samples(RingState *st, long long inc) { st->sample_times += inc; }

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
    bool h_dirty;        // a masked store (row count known on the device only) has run since the last refresh
    int *rank_buf;       // device scratch of the masked store
    long long rank_cap;
    uint32_t flags;
    bool feed_on;        // a feed plan is attached (ddrl_replay_set_feed): the sampler may be handed remote batches
};

static int refresh_counts(ddrl_replay *h, hipStream_t s) {
    RingState tmp;
    DDRL_HIP_CHECK(hipMemcpyAsync(&tmp, h->state, offsetof(RingState, mt_key), hipMemcpyDeviceToHost, s));
    DDRL_HIP_CHECK(hipStreamSynchronize(s));
    h->h_ptr = tmp.ptr; h->h_size = tmp.size; h->h_steps = tmp.steps; h->h_samples = tmp.sample_times;
    h->h_dirty = false;
    if (tmp.error != 0) {
        // a sampler that ran inside a graph / another kernel could not report to its caller: the first host call that
        // looks at the ring does (and clears the flag, so that the caller can repair the plan and go on)
        DDRL_HIP_CHECK(hipMemsetAsync(reinterpret_cast<char *>(h->state) + offsetof(RingState, error), 0, sizeof(int), s));
        DDRL_HIP_CHECK(hipStreamSynchronize(s));
        if (tmp.error == DDRL_ERR_EMPTY_BUFFER) {
            ddrl::set_error("high <= 0");  // the sampler drew from an empty ring (the reference's ValueError)
            return DDRL_ERR_EMPTY_BUFFER;
        }
        if (tmp.error == DDRL_ERR_NOT_REPRESENTABLE) {
            ddrl::set_error("a value stored into a compact (uint8) ring array was not an integer in [0, 255]: the ring holds a clamped/truncated value there");
            return DDRL_ERR_NOT_REPRESENTABLE;
        }
        ddrl::set_error("a feed-plan entry was out of range (wrong batch size, region or batch index): that update trained on a stale input set");
        return tmp.error;
    }
    return DDRL_OK;
}

ddrl_replay_dev::SamplerView ddrl_replay_sampler_view(ddrl_replay_t *h) { return ddrl_replay_dev::SamplerView{h->state, h->ring, h->device}; }
static long long row_floats(const ddrl_replay *h) {
    long long t = 0;
    for (int j = 0; j < h->ring.n_arr; ++j) t += h->ring.w[j];
    return t;
}
static bool has_compact(const ddrl_replay *h) {
    for (int j = 0; j < h->ring.n_arr; ++j) if (h->ring.kind[j]) return true;
    return false;
}
bool ddrl_replay_can_fuse(ddrl_replay_t *h, int64_t batch) {
    const long long bytes = batch * row_floats(h) * (long long)sizeof(float);
    return batch <= MAX_FUSED_BATCH && bytes <= MAX_FUSED_BYTES && (h->h_size > 0 || h->feed_on);
}
void ddrl_replay_note_sample(ddrl_replay_t *h) { h->h_samples += h->ring.samples_inc; }
void ddrl_replay_note_store(ddrl_replay_t *h, long long n) {  // host mirror bookkeeping for n stores issued by another kernel (ddrl_rollout_step)
    h->h_ptr = (h->h_ptr + n) % h->ring.capacity;
    h->h_size = (h->h_size + n > h->ring.capacity) ? h->ring.capacity : h->h_size + n;
    h->h_steps += n * h->ring.steps_inc;
}

extern "C" {

int ddrl_replay_create_ex(ddrl_replay_t **out, int device, int64_t capacity, int32_t n_arrays, const int32_t *widths_h,
                          int64_t steps_inc, int64_t samples_inc) {
    return ddrl_replay_create_typed(out, device, capacity, n_arrays, widths_h, nullptr, steps_inc, samples_inc);
}

int ddrl_replay_create_typed(ddrl_replay_t **out, int device, int64_t capacity, int32_t n_arrays, const int32_t *widths_h,
                             const uint8_t *kinds_h, int64_t steps_inc, int64_t samples_inc) {
    DDRL_REQUIRE(out != nullptr && widths_h != nullptr, "NULL pointer");
    for (int j = 0; kinds_h && j < n_arrays && j < MAX_ARRAYS; ++j) DDRL_REQUIRE(kinds_h[j] <= 1, "array kind must be 0 (float32) or 1 (uint8)");
    DDRL_REQUIRE(capacity > 0 && capacity <= 0xFFFFFFFFll, "capacity must be in [1, 2^32-1]");
    DDRL_REQUIRE(n_arrays >= 1 && n_arrays <= MAX_ARRAYS, "n_arrays must be in [1, 6]");
    for (int j = 0; j < n_arrays; ++j) DDRL_REQUIRE(widths_h[j] > 0, "row widths must be positive");
    ddrl::DeviceGuard g(device);
    if (!g.ok) { ddrl::set_error("cannot select device %d", device); return DDRL_ERR_HIP; }
    ddrl_replay *h = new ddrl_replay();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->ring.capacity = capacity;
    h->ring.n_arr = n_arrays;
    h->ring.steps_inc = steps_inc > 0 ? steps_inc : 1;
    h->ring.samples_inc = samples_inc > 0 ? samples_inc : 1;
    hipError_t e = hipSuccess;
    for (int j = 0; j < n_arrays && e == hipSuccess; ++j) {
        h->ring.w[j] = widths_h[j];
        h->ring.kind[j] = kinds_h ? kinds_h[j] : 0;
        e = hipMalloc(&h->ring.a[j], (size_t)capacity * widths_h[j] * (h->ring.kind[j] ? 1 : sizeof(float)));
    }
    if (e == hipSuccess) e = hipMalloc(&h->state, sizeof(RingState));
    h->idx_cap = 1 << 16;
    if (e == hipSuccess) e = hipMalloc(&h->idx_buf, h->idx_cap * sizeof(long long));
    if (e != hipSuccess) {
        ddrl::set_error("hipMalloc failed for a ring of %lld rows x %lld floats: %s", (long long)capacity, row_floats(h),
                        hipGetErrorString(e));
        ddrl_replay_destroy(h);
        return DDRL_ERR_NOMEM;
    }
    // np.zeros for every ring (example/dsac.py:21-25)
    for (int j = 0; j < n_arrays; ++j)
        DDRL_HIP_CHECK(hipMemsetAsync(h->ring.a[j], 0, (size_t)capacity * widths_h[j] * (h->ring.kind[j] ? 1 : sizeof(float)), nullptr));
    DDRL_HIP_CHECK(hipMemsetAsync(h->state, 0, sizeof(RingState), nullptr));
    k_mt_seed<<<1, 64, 0, nullptr>>>(h->state, 0u);
    DDRL_LAUNCH_CHECK();
    DDRL_HIP_CHECK(hipStreamSynchronize(nullptr));
    *out = h;
    return DDRL_OK;
}

int ddrl_replay_create(ddrl_replay_t **out, int device, int64_t capacity, int obs_dim, int act_dim,
                       uint32_t flags) {
    DDRL_REQUIRE(obs_dim > 0 && act_dim > 0, "obs_dim/act_dim must be positive");
    DDRL_REQUIRE(!(flags & DDRL_REPLAY_ACTS_1D) || act_dim == 1, "ACTS_1D needs act_dim == 1");
    const int32_t widths[5] = {obs_dim, obs_dim, act_dim, 1, 1};  // obs1 obs2 acts rews done
    const uint8_t u8 = (flags & DDRL_REPLAY_U8_OBS) ? 1 : 0;
    const uint8_t kinds[5] = {u8, u8, 0, 0, 0};
    const int rc = ddrl_replay_create_typed(out, device, capacity, 5, widths, kinds, 1, 1);
    if (rc == DDRL_OK) (*out)->flags = flags;
    return rc;
}

int ddrl_replay_destroy(ddrl_replay_t *h) {
    if (!h) return DDRL_OK;
    ddrl::DeviceGuard g(h->device);
    for (int j = 0; j < MAX_ARRAYS; ++j) (void)hipFree(h->ring.a[j]);
    (void)hipFree(h->state); (void)hipFree(h->idx_buf); (void)hipFree(h->rank_buf);
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

int ddrl_replay_store_ex(ddrl_replay_t *h, const float *const *src_h, int64_t n, void *stream) {
    DDRL_REQUIRE(h != nullptr && src_h != nullptr, "NULL pointer");
    DDRL_REQUIRE(n >= 0, "n must be >= 0");
    if (n == 0) return DDRL_OK;
    StoreSrc srcs{};
    int widest = 1;
    for (int j = 0; j < h->ring.n_arr; ++j) {
        DDRL_REQUIRE(src_h[j] != nullptr, "NULL source array");
        srcs.a[j] = src_h[j];
        const int wv = (h->ring.w[j] & 3) == 0 ? h->ring.w[j] / 4 : h->ring.w[j];
        if (wv > widest) widest = wv;
    }
    ddrl::DeviceGuard g(h->device);
    const long long rows = n > h->ring.capacity ? h->ring.capacity : n;
    long long blocks = (rows * widest + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    k_store<<<dim3((unsigned)blocks, (unsigned)h->ring.n_arr), 256, 0, ddrl::as_stream(stream)>>>(h->state, h->ring, srcs, n);
    DDRL_LAUNCH_CHECK();
    h->h_ptr = (h->h_ptr + n) % h->ring.capacity;
    h->h_size = (h->h_size + n > h->ring.capacity) ? h->ring.capacity : h->h_size + n;
    h->h_steps += n * h->ring.steps_inc;
    return DDRL_OK;
}

int ddrl_replay_store_masked_ex(ddrl_replay_t *h, const float *const *src_h, const uint8_t *mask_d, int64_t n, void *stream) {
    DDRL_REQUIRE(h != nullptr && src_h != nullptr && mask_d != nullptr, "NULL pointer");
    DDRL_REQUIRE(n >= 0 && n < 0x7fffffffll, "n must be in [0, 2^31)");
    DDRL_REQUIRE(!has_compact(h), "masked store into a compact (uint8) ring is not built");
    if (n == 0) return DDRL_OK;
    StoreSrc srcs{};
    int widest = 1;
    for (int j = 0; j < h->ring.n_arr; ++j) {
        DDRL_REQUIRE(src_h[j] != nullptr, "NULL source array");
        srcs.a[j] = src_h[j];
        const int wv = (h->ring.w[j] & 3) == 0 ? h->ring.w[j] / 4 : h->ring.w[j];
        if (wv > widest) widest = wv;
    }
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    if (n + 1 > h->rank_cap) {
        DDRL_HIP_CHECK(hipStreamSynchronize(s));
        (void)hipFree(h->rank_buf);
        h->rank_cap = n + 1;
        DDRL_HIP_CHECK(hipMalloc(&h->rank_buf, h->rank_cap * sizeof(int)));
    }
    k_mask_scan<<<1, 256, 0, s>>>(mask_d, n, h->rank_buf);
    long long blocks = (n * widest + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    k_store_masked<<<dim3((unsigned)blocks, (unsigned)h->ring.n_arr), 256, 0, s>>>(h->state, h->ring, srcs, h->rank_buf, n);
    DDRL_LAUNCH_CHECK();
    h->h_dirty = true;
    return DDRL_OK;
}

int ddrl_replay_store(ddrl_replay_t *h, const float *obs_d, const float *act_d, const float *rew_d,
                      const float *obs2_d, const float *done_d, int64_t n, void *stream) {
    DDRL_REQUIRE(h != nullptr && h->ring.n_arr == 5, "not a 5-array (obs1, obs2, acts, rews, done) ring");
    const float *src[5] = {obs_d, obs2_d, act_d, rew_d, done_d};
    return ddrl_replay_store_ex(h, src, n, stream);
}

static int launch_gather(ddrl_replay *h, const long long *idx, int64_t B, BatchPtrs out, hipStream_t s) {
    int widest = 1;
    for (int j = 0; j < h->ring.n_arr; ++j) widest = h->ring.w[j] > widest ? h->ring.w[j] : widest;
    if (widest <= 64 && B >= 4096) {  // many small rows
        long long blocks = (B * ((widest & 3) == 0 ? widest / 4 : widest) + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        k_gather_small<<<dim3((unsigned)blocks, (unsigned)h->ring.n_arr), 256, 0, s>>>(h->ring, out, idx, (int)B);
        DDRL_LAUNCH_CHECK();
        return DDRL_OK;
    }
    k_gather<<<dim3((unsigned)B, (unsigned)h->ring.n_arr), 256, 0, s>>>(h->ring, out, idx, (int)B);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_replay_sample_ex(ddrl_replay_t *h, int64_t batch, float *const *out_h, int64_t *idx_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && out_h != nullptr, "NULL pointer");
    DDRL_REQUIRE(batch > 0 && batch <= (1 << 24), "batch must be in [1, 2^24]");
    BatchPtrs out{};
    for (int j = 0; j < h->ring.n_arr; ++j) {
        DDRL_REQUIRE(out_h[j] != nullptr, "NULL output pointer");
        out.a[j] = out_h[j];
    }
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    if (h->feed_on) {
        DDRL_REQUIRE(ddrl_replay_can_fuse(h, batch), "a feed plan is attached: the batch must fit the one-workgroup sampler");
    } else if (h->h_size <= 0 || h->h_dirty) {
        // the mirror may lag behind graph replays / masked stores: ask the device before reporting empty
        int rc = refresh_counts(h, s);
        if (rc != DDRL_OK) return rc;
        if (h->h_size <= 0) {
            ddrl::set_error("high <= 0");  // message of the reference's ValueError
            return DDRL_ERR_EMPTY_BUFFER;
        }
    }
    const int fuse = ddrl_replay_can_fuse(h, batch) ? 1 : 0;
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
    if (!fuse && batch >= 2048) k_sample_wide<<<1, 640, 0, s>>>(h->state, h->ring, out, (int)batch, idx);
    else k_sample<<<1, SAMPLE_THREADS, 0, s>>>(h->state, h->ring, out, (int)batch, idx, fuse);
    DDRL_LAUNCH_CHECK();
    h->h_samples += h->ring.samples_inc;
    if (!fuse) return launch_gather(h, idx, batch, out, s);
    return DDRL_OK;
}

int ddrl_replay_sample_indices(ddrl_replay_t *h, int64_t batch, int64_t *idx_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && idx_d != nullptr, "NULL pointer");
    DDRL_REQUIRE(batch > 0 && batch <= (1 << 24), "batch must be in [1, 2^24]");
    DDRL_REQUIRE(!h->feed_on, "a feed plan is attached to this ring");
    ddrl::DeviceGuard g(h->device);
    hipStream_t s = ddrl::as_stream(stream);
    if (h->h_size <= 0 || h->h_dirty) {
        int rc = refresh_counts(h, s);
        if (rc != DDRL_OK) return rc;
        if (h->h_size <= 0) {
            ddrl::set_error("high <= 0");  // message of the reference's ValueError
            return DDRL_ERR_EMPTY_BUFFER;
        }
    }
    BatchPtrs none{};
    if (batch >= 2048) k_sample_wide<<<1, 640, 0, s>>>(h->state, h->ring, none, (int)batch, reinterpret_cast<long long *>(idx_d));
    else k_sample<<<1, SAMPLE_THREADS, 0, s>>>(h->state, h->ring, none, (int)batch, reinterpret_cast<long long *>(idx_d), 0);
    DDRL_LAUNCH_CHECK();
    h->h_samples += h->ring.samples_inc;
    return DDRL_OK;
}

int ddrl_replay_sample(ddrl_replay_t *h, int64_t batch, float *obs1_d, float *obs2_d, float *acts_d,
                       float *rews_d, float *done_d, int64_t *idx_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && h->ring.n_arr == 5, "not a 5-array (obs1, obs2, acts, rews, done) ring");
    float *out[5] = {obs1_d, obs2_d, acts_d, rews_d, done_d};
    return ddrl_replay_sample_ex(h, batch, out, idx_d, stream);
}

int ddrl_replay_sample_many(ddrl_replay_t *h, int64_t batch, int64_t count, float *const *out_h, void *stream) {
    DDRL_REQUIRE(h != nullptr && out_h != nullptr, "NULL pointer");
    DDRL_REQUIRE(batch > 0 && count > 0 && batch * count <= (1 << 24), "batch, count must be positive with batch * count <= 2^24");
    DDRL_REQUIRE(!h->feed_on, "a feed plan is attached to this ring");
    // `count` consecutive sample_batch(batch) calls consume the index stream exactly like one draw of batch * count
    // (the ring size cannot change in between: nothing else is ordered between them on this stream)
    const int rc = ddrl_replay_sample_ex(h, batch * count, out_h, nullptr, stream);
    if (rc != DDRL_OK || count == 1) return rc;
    ddrl::DeviceGuard g(h->device);
    k_add_samples<<<1, 1, 0, ddrl::as_stream(stream)>>>(h->state, (count - 1) * h->ring.samples_inc);
    DDRL_LAUNCH_CHECK();
    h->h_samples += (count - 1) * h->ring.samples_inc;
    return DDRL_OK;
}

int ddrl_replay_set_feed(ddrl_replay_t *h, const int32_t *plan_d, int32_t plan_len, int32_t batch, int32_t n_regions,
                         const float *const *region_base_h, const int32_t *region_count_h, void *stream) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    DDRL_REQUIRE(plan_d == nullptr || (plan_len >= 0 && batch > 0 && n_regions >= 0 && n_regions <= MAX_FEED), "bad plan length / batch / region count");
    DDRL_REQUIRE(plan_d == nullptr || n_regions == 0 || (region_base_h && region_count_h), "NULL region table");
    Feed f{};
    if (plan_d) {
        f.plan = plan_d; f.pos = 0; f.len = plan_len; f.batch = batch; f.n_regions = n_regions;
        for (int r = 0; r < n_regions; ++r) {
            DDRL_REQUIRE(region_count_h[r] >= 0 && region_count_h[r] < (1 << 24) && (region_count_h[r] == 0 || region_base_h[r]), "bad region");
            f.base[r] = region_base_h[r]; f.count[r] = region_count_h[r];
        }
    }
    ddrl::DeviceGuard g(h->device);
    k_set_feed<<<1, 1, 0, ddrl::as_stream(stream)>>>(h->state, f);
    DDRL_LAUNCH_CHECK();
    h->feed_on = plan_d != nullptr;
    h->h_dirty = true;  // fed updates do not count as local samples: the mirror re-reads the device counters
    return DDRL_OK;
}

int ddrl_replay_gather_ex(ddrl_replay_t *h, const int64_t *idx_d, int64_t batch, float *const *out_h, void *stream) {
    DDRL_REQUIRE(h != nullptr && idx_d != nullptr && out_h != nullptr, "NULL handle, index or output pointer");
    DDRL_REQUIRE(batch > 0 && batch <= (1 << 24), "batch must be in [1, 2^24]");
    BatchPtrs out{};
    for (int j = 0; j < h->ring.n_arr; ++j) {
        DDRL_REQUIRE(out_h[j] != nullptr, "NULL output pointer");
        out.a[j] = out_h[j];
    }
    ddrl::DeviceGuard g(h->device);
    return launch_gather(h, reinterpret_cast<const long long *>(idx_d), batch, out, ddrl::as_stream(stream));
}

int ddrl_replay_gather(ddrl_replay_t *h, const int64_t *idx_d, int64_t batch, float *obs1_d, float *obs2_d,
                       float *acts_d, float *rews_d, float *done_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && h->ring.n_arr == 5, "not a 5-array (obs1, obs2, acts, rews, done) ring");
    float *out[5] = {obs1_d, obs2_d, acts_d, rews_d, done_d};
    return ddrl_replay_gather_ex(h, idx_d, batch, out, stream);
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

int ddrl_replay_take_error(ddrl_replay_t *h, int32_t *out_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && out_d != nullptr, "NULL pointer");
    ddrl::DeviceGuard g(h->device);
    k_take_error<<<1, 1, 0, ddrl::as_stream(stream)>>>(h->state, out_d);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_replay_buffers_ex(ddrl_replay_t *h, float **arrays_h, int32_t *widths_h, int32_t *n_arrays_h) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    for (int j = 0; j < h->ring.n_arr; ++j) {
        if (arrays_h) arrays_h[j] = h->ring.a[j];
        if (widths_h) widths_h[j] = h->ring.w[j];
    }
    if (n_arrays_h) *n_arrays_h = h->ring.n_arr;
    return DDRL_OK;
}

int ddrl_replay_buffers(ddrl_replay_t *h, float **obs1_d, float **obs2_d, float **acts_d, float **rews_d,
                        float **done_d) {
    DDRL_REQUIRE(h != nullptr && h->ring.n_arr == 5, "not a 5-array (obs1, obs2, acts, rews, done) ring");
    if (obs1_d) *obs1_d = h->ring.a[0];
    if (obs2_d) *obs2_d = h->ring.a[1];
    if (acts_d) *acts_d = h->ring.a[2];
    if (rews_d) *rews_d = h->ring.a[3];
    if (done_d) *done_d = h->ring.a[4];
    return DDRL_OK;
}

int ddrl_replay_rows_export(ddrl_replay_t *h, int32_t array, int64_t row0, int64_t nrows, float *out_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && out_d != nullptr, "NULL pointer");
    DDRL_REQUIRE(array >= 0 && array < h->ring.n_arr && row0 >= 0 && nrows > 0 && row0 + nrows <= h->ring.capacity, "array / row range out of bounds");
    ddrl::DeviceGuard g(h->device);
    long long blocks = (nrows * h->ring.w[array] + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    k_rows_export<<<(unsigned)blocks, 256, 0, ddrl::as_stream(stream)>>>(h->ring, array, row0, nrows, out_d);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_replay_rows_import(ddrl_replay_t *h, int32_t array, int64_t row0, int64_t nrows, const float *src_d, void *stream) {
    DDRL_REQUIRE(h != nullptr && src_d != nullptr, "NULL pointer");
    DDRL_REQUIRE(array >= 0 && array < h->ring.n_arr && row0 >= 0 && nrows > 0 && row0 + nrows <= h->ring.capacity, "array / row range out of bounds");
    ddrl::DeviceGuard g(h->device);
    long long blocks = (nrows * h->ring.w[array] + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    k_rows_import<<<(unsigned)blocks, 256, 0, ddrl::as_stream(stream)>>>(h->state, h->ring, array, row0, nrows, src_d);
    DDRL_LAUNCH_CHECK();
    h->h_dirty = true;   // a value outside uint8 leaves the sticky error: the next host look at the ring reports it
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
