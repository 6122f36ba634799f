// Device-side pieces of the replay sampler shared by replay.hip (k_sample) and sac1.hip (the
// Adam/polyak kernel can run the NEXT update's sampler as one extra workgroup, see
// ddrl_sac1_apply_grads_and_sample).  Internal to libddrl_hip.so — not part of the C-ABI.
#pragma once
#include "ddrl_common.h"

namespace ddrl_replay_dev {

constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr int SAMPLE_THREADS = 256;
constexpr int MAX_FUSED_BATCH = 4096;          // idx staged in LDS for the fused sample+gather
constexpr long long MAX_FUSED_BYTES = 1 << 18;  // fuse the gather when the whole batch is <= 256 KiB

constexpr int MAX_FEED = 16;

// Batches drawn on OTHER ranks' shards (partition.py): the learner's sampler follows a per-update plan held in
// device memory — entry -1 = draw from the local ring, entry (r << 24 | i) = take batch i of region r, a block of
// `count[r]` batches [obs1 | obs2 | acts | rews | done] (each array [count*B, w]) that shard owner r drew with its own
// index stream and sent in one message.  Lives in the ring state so that a captured graph follows a new plan
// without re-capture.
struct Feed {
    const int *plan;
    int pos, len, batch, n_regions;
    const float *base[MAX_FEED];
    int count[MAX_FEED];
};

struct RingState {
    long long ptr, size, steps, sample_times;
    unsigned int done_counter;  // last-block-done ticket for the store kernel
    int error;                  // sticky device-side error (sample from empty ring)
    int mt_pos;
    int pad;
    uint32_t mt_key[MT_N];
    Feed feed;                  // after the MT state: refresh_counts copies the head of the struct only
};

constexpr int MAX_ARRAYS = 6;

// A ring is a set of float32 struct-of-arrays rings that share one cursor: array j is
// [capacity][w[j]] row-major.  The SAC layouts (example/dsac.py:20-27) are the instance
// {obs1[obs], obs2[obs], acts[act], rews[1], done[1]}; the n-step window buffer of
// algos/sac1/sac_ray.py:40-51 is {buffer_o[(Ln+1)*obs], buffer_a[Ln*act], buffer_r[Ln], buffer_d[Ln]}.
// kind[j] = 1: array j is stored as uint8 (the opt-in COMPACT ring for integer-valued pixel observations, DDRL_REPLAY_U8_OBS):
// a[j] then points at capacity * w[j] BYTES.  The surface stays float32 — store converts (a value that is not an integer in
// [0, 255] sets the sticky DDRL_ERR_NOT_REPRESENTABLE), every gather converts back: bit-identical to the float32 ring on such data.
struct RingPtrs {
    float *a[MAX_ARRAYS];
    int w[MAX_ARRAYS];
    unsigned char kind[MAX_ARRAYS];
    int n_arr;
    long long capacity;
    long long steps_inc, samples_inc;  // counter increments per store / per sample (sac_ray.py:68,75: num_buffers)
};

struct BatchPtrs {
    float *a[MAX_ARRAYS];
};

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__device__ __forceinline__ uint32_t mt_mix(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// Regenerate all 624 words: `mt` -> `nw` (a second LDS array: no barrier between a segment's reads and its writes).  The recurrence
// mt[k] <- f(mt[k], mt[k+1], mt[(k+397)%624]) has lag 397, so it splits into three segments whose inputs are all available in parallel:
// [0,227) reads only old words; [227,454) reads new words of segment 1; [454,624) reads new words of segment 2 (and new mt[0]
// for k = 623).  Three barriers; the caller swaps the two arrays.
__device__ __forceinline__ void mt_twist_lds(const uint32_t *mt, uint32_t *nw, int tid) {
    constexpr int S = MT_N - MT_M;   // 227
    if (tid < S) nw[tid] = mt_mix(mt[tid], mt[tid + 1], mt[tid + MT_M]);
    __syncthreads();
    const int k2 = S + tid;          // 227..453
    if (tid < S) nw[k2] = mt_mix(mt[k2], mt[k2 + 1], nw[k2 - S]);
    __syncthreads();
    const int k3 = 2 * S + tid;      // 454..623
    if (k3 < MT_N) nw[k3] = mt_mix(mt[k3], k3 == MT_N - 1 ? nw[0] : mt[k3 + 1], nw[k3 - S]);
    __syncthreads();
}

__device__ __forceinline__ bool aligned16(const void *a, const void *b) {
    return ((reinterpret_cast<unsigned long long>(a) | reinterpret_cast<unsigned long long>(b)) & 15ull) == 0;
}

__device__ __forceinline__ void gather_rows(const float *__restrict__ ring, float *__restrict__ out,
                                            const unsigned *s_idx, int B, int width, int tid, int nthreads, int kind = 0) {
    if (kind == 1) {   // compact array: bytes in, float32 out
        const unsigned char *r8 = reinterpret_cast<const unsigned char *>(ring);
        for (int e = tid; e < B * width; e += nthreads) {
            const int b = e / width, c = e - b * width;
            out[e] = (float)r8[(long long)s_idx[b] * width + c];
        }
        return;
    }
    if ((width & 3) == 0 && aligned16(ring, out)) {
        const int w4 = width >> 2;
        const float4 *r4 = reinterpret_cast<const float4 *>(ring);
        float4 *o4 = reinterpret_cast<float4 *>(out);
        for (int e = tid; e < B * w4; e += nthreads) {
            const int b = e / w4, c = e - b * w4;
            o4[e] = r4[(long long)s_idx[b] * w4 + c];
        }
    } else {
        for (int e = tid; e < B * width; e += nthreads) {
            const int b = e / width, c = e - b * width;
            out[e] = ring[(long long)s_idx[b] * width + c];
        }
    }
}

// idxs = np.random.randint(0, size, B) (masked rejection on 32-bit draws), optionally fused with
// the five gathers when the batch is small (the SAC1 shape: 256 x 80 B).  One workgroup: the
// accept/reject compaction is a wave ballot + prefix count, the stream position advances by
// exactly the number of words NumPy would have consumed.
// NT = threads of the workgroup: SAMPLE_THREADS where the sampler rides in another kernel's launch and for batches that are
// gathered in the same workgroup; 640 for large stand-alone draws (a shard owner's block of batches): a whole 624-word state
// block per pass — one accept / compact pass and one regeneration per block instead of three passes (the pass, not the words, is
// what a one-workgroup draw pays for: 0.60 -> 0.2 ms for 262 144 indices).
template <int NT = SAMPLE_THREADS>
__device__ __forceinline__ void sample_block(RingState *st, const RingPtrs &ring, const BatchPtrs &out, int B,
                                             long long *idx_out, int fuse_gather) {
    constexpr int SAMPLE_THREADS = NT;   // (shadows the namespace constant inside this function)
    __shared__ uint32_t mt_a[MT_N], mt_b[MT_N];
    uint32_t *mt = mt_a, *mt_other = mt_b;   // (block-uniform pointers: the regeneration writes the other array, then they swap)
    __shared__ int s_wave_tot[SAMPLE_THREADS / 64];
    __shared__ int s_consumed;
    __shared__ unsigned s_idx[MAX_FUSED_BATCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (fuse_gather && st->feed.plan != nullptr) {
        const int pos = st->feed.pos;
        const int p = pos < st->feed.len ? st->feed.plan[pos] : -1;
        __syncthreads();  // every lane has read the position
        if (tid == 0) st->feed.pos = pos + 1;
        if (p >= 0) {     // a batch another rank drew: copy it, consume no local draw
            const int r = p >> 24, i = p & 0xffffff;
            if (B != st->feed.batch || r >= st->feed.n_regions || i >= st->feed.count[r]) {
                if (tid == 0) st->error = DDRL_ERR_BAD_ARG;
                return;
            }
            const float *src = st->feed.base[r];
            const long long rows = (long long)st->feed.count[r] * B;
#pragma unroll
            for (int j = 0; j < MAX_ARRAYS; ++j)
                if (j < ring.n_arr) {
                    const int w = ring.w[j], n = B * w;
                    const float *sj = src + (long long)i * n;
                    if ((n & 3) == 0 && ((rows * w) & 3) == 0 && ((((unsigned long long)out.a[j]) | ((unsigned long long)src)) & 15ull) == 0) {
                        for (int e = tid; e < (n >> 2); e += SAMPLE_THREADS)
                            reinterpret_cast<float4 *>(out.a[j])[e] = reinterpret_cast<const float4 *>(sj)[e];
                    } else {
                        for (int e = tid; e < n; e += SAMPLE_THREADS) out.a[j][e] = sj[e];
                    }
                    src += rows * w;
                }
            return;
        }
    }
    const long long size = st->size;
    if (size <= 0) {  // reference: ValueError("high <= 0"); the host wrapper reports it
        if (tid == 0) st->error = DDRL_ERR_EMPTY_BUFFER;
        return;
    }
    const uint32_t rng = (uint32_t)(size - 1);
    if (rng == 0) {
        // NumPy fills with `low` and consumes no draw
        for (int i = tid; i < B; i += SAMPLE_THREADS) {
            if (idx_out) idx_out[i] = 0;
            if (fuse_gather) s_idx[i] = 0;
        }
    } else {
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        for (int i = tid; i < MT_N; i += SAMPLE_THREADS) mt[i] = st->mt_key[i];
        int pos = st->mt_pos;
        bool dirty = false;
        int produced = 0;
        __syncthreads();
        while (produced < B) {
            if (pos >= MT_N) {
                mt_twist_lds(mt, mt_other, tid);
                uint32_t *t = mt; mt = mt_other; mt_other = t;
                pos = 0;
                dirty = true;
            }
            const int w = pos + tid;
            const bool valid = w < MT_N;
            const uint32_t v = valid ? (mt_temper(mt[w]) & mask) : 0xffffffffu;
            const bool acc = valid && v <= rng;
            const unsigned long long bal = __ballot(acc);
            const int rank_in_wave = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) s_wave_tot[wave] = __popcll(bal);
            __syncthreads();
            int off = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < SAMPLE_THREADS / 64; ++i) {
                const int t = s_wave_tot[i];
                if (i < wave) off += t;
                tot += t;
            }
            const int rank = off + rank_in_wave;
            const int need = B - produced;
            if (acc && rank < need) {
                if (idx_out) idx_out[produced + rank] = (long long)v;
                if (fuse_gather) s_idx[produced + rank] = v;
                if (rank == need - 1) s_consumed = tid + 1;  // words consumed up to the last accepted draw
            }
            __syncthreads();
            if (tot >= need) {
                pos += s_consumed;
                produced = B;
            } else {
                pos += (MT_N - pos < SAMPLE_THREADS) ? (MT_N - pos) : SAMPLE_THREADS;
                produced += tot;
            }
            __syncthreads();
        }
        if (dirty)
            for (int i = tid; i < MT_N; i += SAMPLE_THREADS) st->mt_key[i] = mt[i];
        if (tid == 0) st->mt_pos = pos;
    }
    if (tid == 0) st->sample_times += ring.samples_inc;
    if (fuse_gather) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MAX_ARRAYS; ++j)
            if (j < ring.n_arr) gather_rows(ring.a[j], out.a[j], s_idx, B, ring.w[j], tid, SAMPLE_THREADS, ring.kind[j]);
    }
}


// What another kernel needs to run sample_block for a replay handle.
struct SamplerView {
    RingState *state;
    RingPtrs ring;
    int device;
};

}  // namespace ddrl_replay_dev

// defined in replay.hip
ddrl_replay_dev::SamplerView ddrl_replay_sampler_view(ddrl_replay_t *h);
bool ddrl_replay_can_fuse(ddrl_replay_t *h, int64_t batch);
void ddrl_replay_note_sample(ddrl_replay_t *h);  // host mirror bookkeeping for a sample issued by another kernel
void ddrl_replay_note_store(ddrl_replay_t *h, long long n);  // ... for n stores issued by another kernel
