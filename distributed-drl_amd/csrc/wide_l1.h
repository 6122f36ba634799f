// Layer 1 of the discrete-action learners when the observation is WIDE (config 5: flat 84x84x4 = 28 224 floats,
// algos/dqn/train.py:43-52 feeding algos/dqn/core.py:40-50's first dense layer).  The generic k_gemm (one 32x32 tile per
// workgroup: 8 FLOP per operand byte) was built for the SAC shapes; at K = 28 224 it pulls 1.2 GB through the fabric for
// 206 MB of operands and runs at a third of the fp32 MFMA rate (profiles/r03_ddqn_cfg5_before_summary.txt).  One tiled kernel
// replaces it for both GEMMs of the layer:
//
//   k_wide<true>   forward, split K: partial pre-activations P[ev][s] = X[ev][:, Ks] * W1[ev][Ks, :]
//   k_wide_reduce  H1[ev] = relu(sum_s P[ev][s] + b1) in split order (deterministic)
//   k_wide<false>  [dW1 ; db1] = [X | 1]^T dZ1, the whole batch as K, one tile per workgroup (shapes that do not split evenly)
//   k_wide_sk      the same product as equal shares of the stage sequence over all resident workgroups (config 5: below)
//
// One workgroup = 128 output rows x NU <= 5 column units of 32 (32-36 FLOP per operand byte): 4 waves x (32 rows x NU units),
// both operands staged through LDS 32 k at a time (double-buffered 2 x 36 KB, so two workgroups share a CU; whole 128-byte
// lines per row and stage); per 8-deep k group a wave reads 1 + 4 NU operands and issues 4 NU v_mfma_f32_32x32x2_f32.
// hidden = 400 is 13 units: column tiles of 5, 4, 4 units (with 4, 3, 3, 3 the 3-unit tiles needed 9.8 K ranges for the
// 4-unit tiles' 13: 9 % imbalance); a tile's number of K ranges is proportional to its units (16 / 13) so that every
// workgroup carries the same MFMA work, and the total is chosen so that all workgroups are resident at once.
// Workgroups are dealt to the XCDs in contiguous runs of the order (evaluation, K range, row tile, column tile): the
// tiles that read the same X rows / W1 rows at the same time share an L2.
// Exact fp32 like the rest of the learner; the k order inside an 8-deep group differs from k_gemm's (lane half h takes
// k = 8g + 4h + {0..3}), which only reorders the fp32 sums.
#pragma once
#include "gemm_core.h"
#include <type_traits>
#include <vector>

namespace {

constexpr int WD_MAXEV = 5;
constexpr int WD_NB = 5;        // column units of the B image (160 columns)
constexpr int WD_BW = 32 * WD_NB;

struct WideEval {
    const float *A;     // forward: X [M][lda] (k contiguous)     wgrad: X [K][lda] (output row i = input column i contiguous)
    const float *B;     // forward: W1 [K][N]                      wgrad: dZ1 [K][N]
    const float *bias;  // forward only
    float *out;         // forward: H1 [M][ldo] (written by k_wide_reduce)   wgrad: G [(a_rows + 1)][N]
    long long lda;
    // optional row indirection of the forward's M rows: row r of the operand is row ridx[r] of A.  The observation rows of a sampled
    // batch are then read straight out of the replay ring — the LDS-DMA loads take per-lane source addresses anyway — and three
    // quarters of sample_batch(512)'s 231 MB gather never happen (ddrl_dqn_step_ring).  (The same for the K rows of the weight
    // gradient was built and measured: ring-row numbers in LDS, one lookup per 1 KB piece — the kernel, already at 256 VGPRs, went
    // from 134 to 163 us with 189 spilled SGPRs; the gradient contracts a gathered copy of obs1 instead.)
    const long long *ridx;
};
struct WideArgs {
    WideEval ev[WD_MAXEV];
    float *part;        // forward: [nev][Smax][Mp][Np]
    const float *consts; // {1, 0, 0, 0, 0, 0, 0, 0}: sources of what lies outside the matrices
    int nev, M, N, K;   // M: output rows (wgrad: a_rows + 1, the last one contracts the constant 1)
    int a_rows;         // rows of A that exist in memory (forward: M; wgrad: obs_dim)
    int m_tiles, rows, kb;  // rows per tile = 32 x waves of the kernel instance; kb = its k per stage
    // column tiles come in two classes: cnt[0] tiles of nu[0] units, then cnt[1] tiles of nu[1] units; S[c] K ranges each
    int nu[2], cnt[2], S[2], wg0[3];
    int Smax, Mp, Np, ldo, total;
#ifdef WD_STAMPS
    long long *stamps;  // diagnostic builds (tools/wide_bench.hip): [workgroup][wave][8] accumulated phase cycles
#endif
};

// 16 bytes per lane, global -> LDS without a register stop: lane l's bytes land at LDS byte address lds + 16 l (lds wave-uniform).
// Inline asm, not __builtin_amdgcn_global_load_lds: with the builtin hipcc waits vmcnt(0) before the first ds_read that follows
// (it cannot tell the two LDS buffers apart), which serialises the stage being loaded with the stage being computed.  The
// compiler does not count these loads: the kernel waits for them by hand (wide_dma_wait) before the barrier that publishes them.
// M0 (the LDS-DMA base) is compiler-reserved: saved, set and restored inside the one statement.
__device__ __forceinline__ void glds16(const float *src, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
}
__device__ __forceinline__ void wide_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const float *p) {
    return (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const float *)p;
}

template <bool FWD, int WV, int WD_KB>   // the product instance is <., 4, 32>: 4 waves, 32 k per LDS stage (8-wave 256-row tiles and 16-k stages measured no better)
__global__ void __launch_bounds__(64 * WV, 2) k_wide(WideArgs a) {
    static_assert(WD_KB == 32 || !FWD, "the forward A image is cut for 128-byte rows");
    constexpr int ROWS = 32 * WV, AOP = ROWS * WD_KB, WD_BOP = WD_KB * WD_BW, NG = WD_KB / 8;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float *sA = wsm;             // [2][AOP]  forward: [rows][8 float4 slots], chunk q of row r in slot q ^ ((r >> 1) & 7)   wgrad: [32 k][rows]
    float *sB = wsm + 2 * AOP;   // [2][32 k][160 columns]
    int L;
    {   // contiguous run of the workgroup order per XCD (block b runs on XCD b % 8)
        const int b = blockIdx.x, q = a.total >> 3, r = a.total & 7, x = b & 7;
        L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int c = L >= a.wg0[1] ? 1 : 0;
    L -= a.wg0[c];
    const int NU = a.nu[c], S = a.S[c];
    const int ct = L % a.cnt[c]; L /= a.cnt[c];
    const int mt = L % a.m_tiles; L /= a.m_tiles;
    const int s = L % S, e = L / S;
    const WideEval E = a.ev[e];
    const int unit0 = c ? a.cnt[0] * a.nu[0] + ct * NU : ct * NU;
    const int m0 = mt * ROWS, n0 = unit0 * 32;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = (a.K + WD_KB - 1) / WD_KB;
    const int st0 = (int)((long long)s * T / S), st1 = (int)((long long)(s + 1) * T / S);
    // global -> LDS staging by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass).  One wave-instruction
    // fills 1 KB of LDS in lane order from per-lane source addresses, so the images are cut into 1 KB pieces of 64 float4:
    //   forward A   piece I = rows 8I..8I+7 x 8 slots; the lane that fills slot p of row r loads chunk p ^ ((r >> 1) & 7) (the
    //               swizzle sits on the SOURCE address and on the read; the destination is linear)
    //   wgrad A     piece I = 1 KB of the [32 k][rows] image (4 waves: k rows 2I, 2I + 1)
    //   B           piece I = float4 64 I .. 64 I + 63 of the [32 k][160 columns] image (1.6 k rows)
    // What lies outside the matrices comes from a 16-byte block of zeros, the gradient's bias row from {1, 0, 0, 0}.
    constexpr int APW = (AOP / 256) / WV, BPW = (WD_BOP / 256) / WV;   // pieces per wave and stage: 4 of A, 5 of B
    static_assert(APW * WV * 256 == AOP && BPW * WV * 256 == WD_BOP, "the LDS images are dealt to the waves in whole 1 KB pieces");
    static_assert(WV == 4 && WD_KB == 32, "a B image of 32 k x 32 NU columns is 4 NU pieces: NU per wave of four");
    const float *const ones_blk = a.consts, *const zero_blk = a.consts + 4;
    const float *pa[APW];
    int ak[APW];            // k offset inside a stage of what this lane loads (forward A: of its chunk; else of its k row)
    int amode = 1;          // wgrad: 1 data, 2 the bias row's constant, 0 beyond it
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int I = APW * w + i;
        if (FWD) {
            const int row = 8 * I + (lane >> 3), q = (lane & 7) ^ ((row >> 1) & 7);
            ak[i] = 4 * q;
            const long long rr = (E.ridx && m0 + row < a.a_rows) ? E.ridx[m0 + row] : m0 + row;
            pa[i] = m0 + row < a.a_rows ? E.A + rr * E.lda + 4 * q : nullptr;
        } else {
            const int kk = WV == 8 ? I : 2 * I + (lane >> 5), i0 = m0 + 4 * (WV == 8 ? lane : lane & 31);
            ak[i] = kk;
            amode = i0 < a.a_rows ? 1 : i0 == a.a_rows ? 2 : 0;   // a_rows % 4 == 0: four input columns exist or not as a whole
            pa[i] = E.A + (long long)kk * E.lda + (amode == 1 ? i0 : 0);
        }
    }
    // The B image of a tile is as wide as the tile: [32 k][32 NU columns] = NU pieces per wave and stage (round 4 staged 160 columns for
    // every tile: a 4-unit tile then issued one DMA instruction of nine per wave and stage for nothing but zeros — and the issue of a
    // stage's pieces, ~1-2 k cycles per wave, is what the matrix pipe waits for).  The pointer table and the issue are therefore part of
    // the per-NU instance of the stage loop (run()).
    // a stage's pieces, all issued at the top of the stage before (dealing them between that stage's MFMA groups instead measured
    // the same: 359 vs 345 us at config 5 — the texture path takes ~1 k cycles per CU and stage either way)
    auto issue_a = [&](int st, int buf) {
        const int k = st * WD_KB;
        const unsigned dA = lds_addr(sA + buf * AOP + 256 * APW * w);
#pragma unroll
        for (int g = 0; g < APW; ++g) {
            const float *src;
            if (FWD) src = (pa[g] && k + ak[g] < a.K) ? pa[g] + k : zero_blk;   // K % 4 == 0
            else src = amode == 2 ? ones_blk : (amode == 1 && k + ak[g] < a.K) ? pa[g] + (long long)k * E.lda : zero_blk;
            glds16(src, dA + 1024 * g);
        }
    };
    floatx16 acc[WD_NB];
#pragma unroll
    for (int u = 0; u < WD_NB; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    const bool active = m0 + 32 * w < a.M;   // wave-uniform: this wave's 32 rows hold output
    const int sw = (l31 >> 1) & 7;
    auto compute = [&](const float *cA, const float *cB, auto nu_c) {
        constexpr int NUC = decltype(nu_c)::value;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float av[4], bv[NUC][4];
            if (FWD) {
                const float4 t = *reinterpret_cast<const float4 *>(cA + (32 * w + l31) * WD_KB + (((2 * g + h) ^ sw) << 2));
                av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = cA[(8 * g + 4 * h + j) * ROWS + 32 * w + l31];
            }
#pragma unroll
            for (int u = 0; u < NUC; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[u][j] = cB[(8 * g + 4 * h + j) * (32 * NUC) + 32 * u + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < NUC; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[u][j], acc[u], 0, 0, 0);
        }
    };
    // the stage loop, one instance per NU (dispatching inside the loop made the compiler shuffle the accumulators between variants)
    auto run = [&](auto nu_c) {
        constexpr int NUC = decltype(nu_c)::value;
        const float *pb[NUC];
        int bk[NUC];
#pragma unroll
        for (int j = 0; j < NUC; ++j) {
            const int f4 = 64 * (NUC * w + j) + lane, kk = f4 / (8 * NUC), c4 = f4 - kk * (8 * NUC);
            bk[j] = kk;
            pb[j] = n0 + 4 * c4 < a.N ? E.B + (long long)kk * a.N + n0 + 4 * c4 : nullptr;   // N % 4 == 0
        }
        auto issue = [&](int st, int buf) {
            issue_a(st, buf);
            const int k = st * WD_KB;
            const unsigned dB = lds_addr(sB + buf * WD_BOP + 256 * NUC * w);
#pragma unroll
            for (int j = 0; j < NUC; ++j) glds16((pb[j] && k + bk[j] < a.K) ? pb[j] + (long long)k * a.N : zero_blk, dB + 1024 * j);
        };
        issue(st0, 0);
        wide_dma_wait();
        __syncthreads();
#ifdef WD_STAMPS
        long long ph[6] = {0, 0, 0, 0, 0, 0}, t_prev = (long long)__builtin_readcyclecounter();
#define WD_PHN(i) do { const long long t_now = (long long)__builtin_readcyclecounter(); ph[i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define WD_PHN(i) do { } while (0)
#endif
        for (int st = st0; st < st1; ++st) {
            const int cur = (st - st0) & 1;
            WD_PHN(0);
            if (st + 1 < st1) issue(st + 1, cur ^ 1);   // that buffer's last reads ended before the previous barrier
            WD_PHN(1);   // issue of the next stage's loads
            if (active) compute(sA + cur * AOP, sB + cur * WD_BOP, nu_c);
            WD_PHN(2);   // compute
            wide_dma_wait();   // the next stage has landed (this wave's share; the barrier covers the others')
            WD_PHN(3);
            __syncthreads();
            WD_PHN(4);   // barrier
        }
#ifdef WD_STAMPS
        if (a.stamps && lane == 0) {
            long long *o = a.stamps + ((long long)blockIdx.x * WV + w) * 8;
            for (int i = 0; i < 6; ++i) o[i] = ph[i];
            o[6] = st1 - st0; o[7] = NU;
        }
#endif
    };
    if (st0 < st1) {
        if (NU == 5) run(std::integral_constant<int, 5>{});
        else if (NU == 4) run(std::integral_constant<int, 4>{});
        else if (NU == 3) run(std::integral_constant<int, 3>{});
        else if (NU == 2) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 1>{});
    }
    // D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h
    if (FWD) {
        float *P = a.part + (((long long)e * a.Smax + s) * a.Mp + m0 + 32 * w) * a.Np + n0 + l31;
#pragma unroll
        for (int u = 0; u < WD_NB; ++u)
            if (u < NU)
#pragma unroll
                for (int r = 0; r < 16; ++r) P[(long long)((r & 3) + 8 * (r >> 2) + 4 * h) * a.Np + 32 * u] = acc[u][r];
    } else {
#pragma unroll
        for (int u = 0; u < WD_NB; ++u)
            if (u < NU) {
                const int col = n0 + 32 * u + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < a.M && col < a.N) E.out[(long long)row * a.N + col] = acc[u][r];
                }
            }
    }
}

// H1 = relu(sum of the column's partials + b1); one thread per four columns
__global__ void __launch_bounds__(256) k_wide_reduce(WideArgs a) {
    const int n4s = a.N >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)a.M * n4s;
    if (i >= per * a.nev) return;
    const int e = (int)(i / per);
    const long long rc = i - e * per;
    const int row = (int)(rc / n4s), c4 = (int)(rc - (long long)row * n4s) * 4;
    const WideEval E = a.ev[e];
    const int S = (c4 >> 5) < a.cnt[0] * a.nu[0] ? a.S[0] : a.S[1];
    const float4 *P = reinterpret_cast<const float4 *>(a.part + (((long long)e * a.Smax) * a.Mp + row) * a.Np + c4);
    const long long sstride = (long long)a.Mp * a.Np / 4;
    float4 acc = P[0];
    for (int s = 1; s < S; ++s) {
        const float4 v = P[s * sstride];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float4 bz = *reinterpret_cast<const float4 *>(E.bias + c4);
    acc.x = fmaxf(acc.x + bz.x, 0.f); acc.y = fmaxf(acc.y + bz.y, 0.f); acc.z = fmaxf(acc.z + bz.z, 0.f); acc.w = fmaxf(acc.w + bz.w, 0.f);
    *reinterpret_cast<float4 *>(E.out + (long long)row * a.ldo + c4) = acc;
}

// ==========================================================================================
// k_wide_sk: the layer-1 weight gradient with the STAGE SEQUENCE, not the tile, as the unit of work ("stream-K").
// [dW1 ; db1] = [X | 1]^T dZ1 has 221 row tiles x 3 column tiles (5, 4, 4 units) of 16 stages at config 5: 663 workgroups for
// 512 resident slots — 151 slots run a second tile while 361 idle (k_wide<false>: 1.7 rounds, MFMA busy 0.48).  Here exactly
// `nwg` <= 512 resident workgroups each take an equal share of the weighted stage sequence (a stage of an NU-unit tile weighs NU;
// tiles in (row tile, column tile) order, so a workgroup's tiles share their X rows): a tail fragment of the tile its
// predecessor started, whole tiles, a head fragment of the tile its successor finishes.  A split tile is combined in-launch:
// the successor computes its (later-stage) fragment FIRST and publishes the accumulators as a slab (sc1 write-through stores, vmcnt
// drain, barrier, relaxed agent-scope flag; the flags are zeroed by an earlier launch of the same update: the guide's hand-off recipe R1); the workgroup
// that holds the head fragment computes it LAST, by which time the slab has long arrived, adds it in the fixed order
// head + tail and writes the tile.  Every sum has one fixed order for a given shape: bit-identical run to run.  A share is at
// least one tile long, so a tile never spans three workgroups and a workgroup only ever waits for its successor's first action.
// ==========================================================================================
struct SkFrag { int tile, s0, s1, role; };   // role 0: whole tile -> output; 1: tail part -> slab; 2: head part + successor's slab -> output; -1: none
struct SkArgs {
    WideArgs w;
    const SkFrag *frags;   // [nwg][3]
    float *slab;           // [nwg][4 waves][WD_NB][16 regs][64 lanes]
    int *flag;             // [nwg]
    int nwg, epoch;
    int *err;              // host-mapped sticky word: 1 = a combine ran out of patience (read by the host at the next step)
#ifdef WD_STAMPS
    long long *stamps;     // diagnostic builds (tools/wide_bench.hip): [workgroup][wave][16] — start, end, stages, then accumulated phase cycles
#endif
};
__device__ __forceinline__ void sk_store_sc1(float *p, float4 v) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
// (Round 6) The head fragment of a split tile STARTS from the successor's slab — its 20 loads go straight into the accumulator registers —
// instead of adding it at the end out of a second set of 80 registers.  The poll then sits in front of the head fragment: by then this
// workgroup has run its tail fragment and a whole tile, at least as many weighted stages as the successor needed to publish (a share is
// at least one largest tile long).  Sum order of a split tile: slab + head stages in stage order; fixed, bit-identical run to run.
__global__ void __launch_bounds__(256, 2) k_wide_sk(SkArgs sa) {
    constexpr int WV = 4, WD_KB = 32, ROWS = 32 * WV, AOP = ROWS * WD_KB, WD_BOP = WD_KB * WD_BW, NG = WD_KB / 8;
    constexpr int APW = (AOP / 256) / WV;
    const WideArgs &a = sa.w;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float *sA = wsm, *sB = wsm + 2 * AOP;
    const int wg = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WideEval E = a.ev[0];
    const int T = (a.K + WD_KB - 1) / WD_KB, ntc = a.cnt[0] + a.cnt[1];
    const float *const ones_blk = a.consts, *const zero_blk = a.consts + 4;
#ifdef WD_STAMPS
    long long sk_ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sk_prev = (long long)__builtin_amdgcn_s_memtime();
    const long long sk_t0 = sk_prev;
#define SK_PH(i) do { const long long t_now = (long long)__builtin_amdgcn_s_memtime(); sk_ph[i] += t_now - sk_prev; sk_prev = t_now; } while (0)
#else
#define SK_PH(i) do { } while (0)
#endif
    for (int f = 0; f < 3; ++f) {
        const SkFrag fr = sa.frags[3 * wg + f];
        if (fr.role < 0) continue;   // block-uniform
        const int mt = fr.tile / ntc, ct = fr.tile - mt * ntc;
        const int c = ct >= a.cnt[0] ? 1 : 0;
        const int NU = a.nu[c];
        const int unit0 = c ? a.cnt[0] * a.nu[0] + (ct - a.cnt[0]) * NU : ct * NU;
        const int m0 = mt * ROWS, n0 = unit0 * 32;
        const float *pa[APW];
        int ak[APW], amode = 1;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int I = APW * w + i;
            const int kk = 2 * I + (lane >> 5), i0 = m0 + 4 * (lane & 31);
            ak[i] = kk;
            amode = i0 < a.a_rows ? 1 : i0 == a.a_rows ? 2 : 0;
            pa[i] = E.A + (long long)kk * E.lda + (amode == 1 ? i0 : 0);
        }
        const bool active = m0 + 32 * w < a.M;
        // The whole fragment — accumulator start, stage loop, publish / combine / store — is ONE instance per tile width.  (With only the
        // stage loop instantiated per width and the accumulators declared outside, the five loops' results met in a merge and the
        // register allocator gave every width its own accumulator tuples: (5 + 4 + 3 + 2 + 1) x 16 registers, the whole file — that,
        // not the combine's second register set, is where the 47-53 spilled registers of rounds 4-5 came from.)
        auto body = [&](auto nu_c) {
            constexpr int NUC = decltype(nu_c)::value;
            // the B image of a tile is as wide as the tile — [32 k][32 NUC columns] = NUC pieces per wave and stage, k_wide's form (round 5
            // kept 160 columns here: with the accumulators of every width alive that cost 54 more spilled registers)
            const float *pb[NUC];
            int bk[NUC];
#pragma unroll
            for (int j = 0; j < NUC; ++j) {
                const int f4 = 64 * (NUC * w + j) + lane, kk = f4 / (8 * NUC), c4 = f4 - kk * (8 * NUC);
                bk[j] = kk;
                pb[j] = n0 + 4 * c4 < a.N ? E.B + (long long)kk * a.N + n0 + 4 * c4 : nullptr;
            }
            auto issue = [&](int st, int buf) {
                const int k = st * WD_KB;
                const unsigned dA = lds_addr(sA + buf * AOP + 256 * APW * w), dB = lds_addr(sB + buf * WD_BOP + 256 * NUC * w);
#pragma unroll
                for (int g = 0; g < APW; ++g) {
                    const float *src = amode == 2 ? ones_blk : (amode == 1 && k + ak[g] < a.K) ? pa[g] + (long long)k * E.lda : zero_blk;
                    glds16(src, dA + 1024 * g);
                }
#pragma unroll
                for (int j = 0; j < NUC; ++j) glds16((pb[j] && k + bk[j] < a.K) ? pb[j] + (long long)k * a.N : zero_blk, dB + 1024 * j);
            };
            floatx16 acc[NUC];
            if (fr.role == 2) {
                // consume (guide, Guideline 16 recipe R1): ONE lane polls the ONE word relaxed, ONE agent-scope acquire drops this CU's
                // stale L1 lines, its vmcnt drain holds the barrier for the invalidate, then every wave reads its part of the slab with
                // plain loads, all in flight at once, straight into the accumulators.
                // The spin is bounded: dispatch order is no contract (a shared GPU, a profiler serialising workgroups), so after ~1 s
                // the workgroup gives up: it raises the sticky host-mapped error word and goes on with whatever the slab holds — this
                // update's layer-1 gradient is then WRONG, and the next ddrl_dqn_step* call returns DDRL_ERR_HIP saying so and switches
                // this learner to the tile-per-workgroup kernel (round 4 trapped here, which took the whole HIP context down).
                if (tid == 0) {
                    int spin = 0;
                    while (__hip_atomic_load(sa.flag + wg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sa.epoch) {
                        __builtin_amdgcn_s_sleep(16);
                        if (++spin > (1 << 21)) { __hip_atomic_store(sa.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                const float *S = sa.slab + ((((long long)(wg + 1) * WV + w) * WD_NB) * 16 * 64 + lane * 4);
#pragma unroll
                for (int u = 0; u < NUC; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4 *>(S + (u * 4 + q) * 256);
                        acc[u][4 * q] = v.x; acc[u][4 * q + 1] = v.y; acc[u][4 * q + 2] = v.z; acc[u][4 * q + 3] = v.w;
                    }
                // a use of the loaded registers HERE: the compiler then waits for the slab in front of the stage loop.  Without it its
                // wait lands in front of the loop's first MFMA that reads them — a vmcnt(0) executed in EVERY stage, which also
                // waits for the next stage's LDS-DMA loads (they are invisible to the compiler's counter) just issued: no overlap.
#pragma unroll
                for (int u = 0; u < NUC; ++u) asm volatile("" : "+v"(acc[u]));
            } else {
#pragma unroll
                for (int u = 0; u < NUC; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
            }
            SK_PH(0);   // fragment set-up (pointers; role 2: poll + slab into the accumulators)
            if (fr.s0 < fr.s1 && fr.s1 <= T) {
                issue(fr.s0, 0);
                wide_dma_wait();
                __syncthreads();
                SK_PH(1);   // pipeline fill: the fragment's first stage, nothing to compute meanwhile
                for (int st = fr.s0; st < fr.s1; ++st) {
                    const int cur = (st - fr.s0) & 1;
                    if (st + 1 < fr.s1) issue(st + 1, cur ^ 1);
                    SK_PH(2);   // issue of the next stage
                    if (active) {
                        const float *cA = sA + cur * AOP, *cB = sB + cur * WD_BOP;
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            float av[4], bv[NUC][4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) av[j] = cA[(8 * g + 4 * h + j) * ROWS + 32 * w + l31];
#pragma unroll
                            for (int u = 0; u < NUC; ++u)
#pragma unroll
                                for (int j = 0; j < 4; ++j) bv[u][j] = cB[(8 * g + 4 * h + j) * (32 * NUC) + 32 * u + l31];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int u = 0; u < NUC; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[u][j], acc[u], 0, 0, 0);
                        }
                    }
                    SK_PH(3);   // compute
                    wide_dma_wait();
                    SK_PH(4);   // next stage landed
                    __syncthreads();
                    SK_PH(5);   // barrier
#ifdef WD_STAMPS
                    sk_ph[10] += 1;
#endif
                }
            }
            if (fr.role == 1) {
                // publish: [wave][unit][four-register group][lane] float4 — one fully coalesced 1 KB line per wave-instruction
                float *S = sa.slab + ((((long long)wg * WV + w) * WD_NB) * 16 * 64 + lane * 4);
#pragma unroll
                for (int u = 0; u < NUC; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        sk_store_sc1(S + (u * 4 + q) * 256, make_float4(acc[u][4 * q], acc[u][4 * q + 1], acc[u][4 * q + 2], acc[u][4 * q + 3]));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(sa.flag + wg, sa.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                SK_PH(6);   // publish
                return;
            }
#pragma unroll
            for (int u = 0; u < NUC; ++u) {
                const int col = n0 + 32 * u + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < a.M && col < a.N) E.out[(long long)row * a.N + col] = acc[u][r];
                }
            }
            SK_PH(7);   // tile stores
        };
        if (NU == 5) body(std::integral_constant<int, 5>{});
        else if (NU == 4) body(std::integral_constant<int, 4>{});
        else if (NU == 3) body(std::integral_constant<int, 3>{});
        else if (NU == 2) body(std::integral_constant<int, 2>{});
        else body(std::integral_constant<int, 1>{});
    }
#ifdef WD_STAMPS
    if (sa.stamps && lane == 0) {
        long long *o = sa.stamps + ((long long)wg * WV + w) * 16;
        o[0] = sk_t0; o[1] = (long long)__builtin_amdgcn_s_memtime(); o[2] = sk_ph[10];
        for (int i = 0; i < 8; ++i) o[3 + i] = sk_ph[i];
    }
#endif
}

// Host side of k_wide_sk: the fragments of every workgroup.  `a` is a one-evaluation, one-K-range plan (wide_plan(..., split = false)).
static int wide_plan_sk(const WideArgs &a, int slots, std::vector<SkFrag> &frags) {
    const int ntc = a.cnt[0] + a.cnt[1], ntile = a.m_tiles * ntc, T = (a.K + 31) / 32;
    auto nu_of = [&](int t) { const int ct = t % ntc; return ct < a.cnt[0] ? a.nu[0] : a.nu[1]; };
    long long W = 0, wmax = 0;
    for (int t = 0; t < ntile; ++t) { W += (long long)T * nu_of(t); if ((long long)T * nu_of(t) > wmax) wmax = (long long)T * nu_of(t); }
    int nwg = slots;
    if ((long long)nwg * wmax > W) nwg = (int)(W / wmax);   // a share is at least one (largest) tile long
    if (nwg < 1) nwg = 1;
    // boundary i in weighted units -> (tile, stage): the stage a boundary falls into starts the successor's fragment
    std::vector<int> bt(nwg + 1), bs(nwg + 1);
    {
        int t = 0;
        long long t_start = 0;
        for (int i = 0; i <= nwg; ++i) {
            const long long b = i == nwg ? W : (W * i) / nwg;
            while (t < ntile && t_start + (long long)T * nu_of(t) <= b) { t_start += (long long)T * nu_of(t); ++t; }
            bt[i] = t;
            bs[i] = t < ntile ? (int)((b - t_start) / nu_of(t)) : 0;
        }
    }
    frags.assign((size_t)nwg * 3, SkFrag{0, 0, 0, -1});
    for (int i = 0; i < nwg; ++i) {
        SkFrag *f = &frags[(size_t)3 * i];
        int t0 = bt[i];
        if (bs[i] > 0) { f[0] = SkFrag{t0, bs[i], T, 1}; ++t0; }             // the tail of the tile the predecessor finishes: first, published
        const int t1 = bt[i + 1];                                            // whole tiles [t0, t1)
        // whole tiles are run as one "fragment" each; the table holds three entries, so more than one whole tile per workgroup
        // (small problems) is expressed by shrinking nwg: here a share is < 2 tiles + change by construction only when nwg == slots
        if (t1 - t0 > 1) return -1;
        if (t1 - t0 == 1) f[1] = SkFrag{t0, 0, T, 0};
        if (bs[i + 1] > 0) f[2] = SkFrag{t1, 0, bs[i + 1], 2};               // the head of the tile the successor's slab completes: last
    }
    return nwg;
}

// the wide path takes layer 1 when the observation is wide enough for the split-K forward to fill the chip
static bool wide_applies(int obs, int h1) { return obs >= 1024 && obs % 4 == 0 && h1 % 4 == 0; }

// Tiling of an [M x N] output over K: column tiles in two classes, K ranges per class proportional to the class's units,
// at most `slots` workgroups (forward: all resident, 2 per CU; split = false: one K range, any number of workgroups).
static void wide_plan(WideArgs &a, int nev, int M, int N, int K, bool split, int waves, int slots, int kb = 32, int max_nu = WD_NB) {
    a.nev = nev; a.M = M; a.N = N; a.K = K; a.rows = 32 * waves; a.kb = kb;
    const int units = (N + 31) / 32, ntiles = (units + max_nu - 1) / max_nu, base = units / ntiles, rem = units % ntiles;
    a.nu[0] = base + 1; a.cnt[0] = rem; a.nu[1] = base; a.cnt[1] = ntiles - rem;
    a.m_tiles = (M + a.rows - 1) / a.rows;
    a.Mp = a.m_tiles * a.rows; a.Np = units * 32;
    const int T = (K + kb - 1) / kb;
    for (int c = 0; c < 2; ++c) {
        int S = 1;
        if (split) {
            S = (int)((long long)a.nu[c] * slots / ((long long)nev * a.m_tiles * units));
            if (S < 1) S = 1;
            if (S > T) S = T;
        }
        a.S[c] = a.cnt[c] ? S : 1;
    }
    a.Smax = a.S[0] > a.S[1] ? a.S[0] : a.S[1];
    a.wg0[0] = 0;
    a.wg0[1] = nev * a.m_tiles * a.S[0] * a.cnt[0];
    a.wg0[2] = a.wg0[1] + nev * a.m_tiles * a.S[1] * a.cnt[1];
    a.total = a.wg0[2];
}
static size_t wide_part_floats(const WideArgs &a) { return (size_t)a.nev * a.Smax * a.Mp * a.Np; }
template <bool FWD, int WV, int KB>
static hipError_t wide_launch(const WideArgs &a, hipStream_t s, bool prepare_only) {
    constexpr size_t lds = (size_t)2 * (32 * WV * KB + KB * WD_BW) * sizeof(float);   // 4 waves, 32-k stages: 72 KB (two workgroups per CU)
    if (prepare_only) return hipFuncSetAttribute(reinterpret_cast<const void *>(k_wide<FWD, WV, KB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_wide<FWD, WV, KB><<<a.total, 64 * WV, lds, s>>>(a);
    return hipSuccess;
}
// once per device, outside any stream capture (ddrl_dqn_create)
static hipError_t wide_prepare() {
    WideArgs z{};
    hipError_t e = wide_launch<true, 4, 32>(z, nullptr, true);
    if (e == hipSuccess) e = wide_launch<false, 4, 32>(z, nullptr, true);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_wide_sk), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)((size_t)2 * (128 * 32 + 32 * WD_BW) * sizeof(float)));
    return e;
}
static void launch_wide_fwd(const WideArgs &a, hipStream_t s) {
    (void)wide_launch<true, 4, 32>(a, s, false);
    const long long n = (long long)a.nev * a.M * (a.N >> 2);
    k_wide_reduce<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a);
}
static void launch_wide_wgrad(const WideArgs &a, hipStream_t s) { (void)wide_launch<false, 4, 32>(a, s, false); }
static void launch_wide_sk(const SkArgs &a, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * (128 * 32 + 32 * WD_BW) * sizeof(float);
    k_wide_sk<<<a.nwg, 256, lds, s>>>(a);
}

}  // namespace
