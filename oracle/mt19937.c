/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * Plain-C restatement of the index stream behind the reference's
 *     idxs = np.random.randint(0, self.size, size=batch_size)
 * (example/dsac.py:40, algos/sac1/sac1.py:54, algos/dqn/train.py:67).
 *
 * The algorithm lives in a third-party dependency that is NOT under /root/reference:
 * NumPy's legacy global RandomState (MT19937; version unpinned by the reference, verified
 * here against NumPy 2.2.6).  Restated from the published algorithm:
 *   - np.random.seed(s)  == Matsumoto/Nishimura init_genrand(s), position := 624
 *   - one draw           == genrand_int32() (twist every 624 words, then tempering)
 *   - randint(0, n, B)   == for each index: rng = n-1; if rng == 0 emit 0 and consume NO draw;
 *                           else mask = smallest (2^k - 1) >= rng, repeat v = draw & mask
 *                           until v <= rng; emit v.    (legacy "masked rejection", 32-bit draws
 *                           because rng <= 0xFFFFFFFF)
 * Pinned by tests/test_oracle_replay.py against (a) the golden index streams produced by the
 * reference's own ReplayBuffer.sample_batch (tests/golden/index_streams.*) and (b) NumPy itself.
 */
#include <stdint.h>
#include <stddef.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t key[MT_N];
    int32_t pos;
} oracle_mt_t;

void oracle_mt_seed(oracle_mt_t *st, uint32_t seed)
{
    st->key[0] = seed;
    for (int i = 1; i < MT_N; ++i)
        st->key[i] = 1812433253u * (st->key[i - 1] ^ (st->key[i - 1] >> 30)) + (uint32_t)i;
    st->pos = MT_N;
}

static void mt_twist(oracle_mt_t *st)
{
    uint32_t *mt = st->key;
    int kk;
    uint32_t y;
    for (kk = 0; kk < MT_N - MT_M; ++kk) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < MT_N - 1; ++kk) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    st->pos = 0;
}

uint32_t oracle_mt_next(oracle_mt_t *st)
{
    if (st->pos >= MT_N)
        mt_twist(st);
    uint32_t y = st->key[st->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* randint(0, high, n) -> out[n] (int64 like NumPy's default).  Returns number of 32-bit
 * draws consumed, or -1 when high <= 0 (NumPy raises ValueError("high <= 0")). */
int64_t oracle_randint(oracle_mt_t *st, int64_t high, int64_t n, int64_t *out)
{
    if (high <= 0)
        return -1;
    uint32_t rng = (uint32_t)(high - 1);
    int64_t draws = 0;
    if (rng == 0) {
        for (int64_t i = 0; i < n; ++i)
            out[i] = 0;
        return 0;
    }
    uint32_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t v;
        do {
            v = oracle_mt_next(st) & mask;
            ++draws;
        } while (v > rng);
        out[i] = (int64_t)v;
    }
    return draws;
}

size_t oracle_mt_sizeof(void) { return sizeof(oracle_mt_t); }
