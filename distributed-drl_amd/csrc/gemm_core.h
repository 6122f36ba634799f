// Shared device core of the learners (included by sac1.hip and dqn.hip; every definition has internal linkage):
//   * the batched small-GEMM kernel on v_mfma_f32_32x32x2_f32 (exact fp32) with its job tables and builders
//     (forward / dgrad / wgrad incl. bias rows, optional Adam + polyak in the wgrad epilogue),
//   * the TF1-style Adam + polyak kernel over a flat padded parameter buffer,
//   * the dense <-> padded layout copy, wave-level row helpers, the double-buffered optimizer state.
#pragma once
#include "ddrl_common.h"
#include "replay_device.h"

#include <vector>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int MAXA = 8;        // act_dim supported by the row kernels
constexpr int NEVAL = 8;       // network evaluations per update
constexpr float LOG2PI = 1.8378770664093453f;  // float32(np.log(2*np.pi))
constexpr float STD_EPS = 1e-8f;               // core.py:5 EPS

// Performance rule for every kernel in this file (measured: a kernel's duration here is set by its
// chain of dependent global round trips, not by arithmetic): loads are issued branch-free and all
// at once — clamped addresses + selects instead of per-lane `if`, fixed-trip unrolled loops instead
// of runtime-trip loops — so that one round trip covers a whole phase.

// ------------------------------------------------------------------------------------------
// job descriptors
// ------------------------------------------------------------------------------------------
struct OptState;
struct GemmJob {
    const float *A, *B;
    float *C;
    const float *bias;  // forward: C = relu(A*B + bias)
    const float *mask;  // dgrad:   C = (A*B) where mask > 0 else 0   (row stride ldmask)
    int M, N, K, lda, ldb, ldc, ldmask;
    int a_kc;  // 1: A(i,k) = A[i*lda + k]   0: A(i,k) = A[k*lda + i]
    int b_kc;  // 1: B(k,j) = B[j*ldb + k]   0: B(k,j) = B[k*ldb + j]
    int relu;
    int fast;  // operands 16-B aligned with row strides % 4 == 0 (and K % 4 == 0 for K-contiguous ones)
    // optional fused layer-1 wgrad of the SAME network (dgrad jobs): the tile's rows of C = dZ1 are
    // contracted with the layer-1 input rows [X | 1] and written as per-row-tile partials
    //   part[mt][k][j] = sum_{r in row tile mt} X1[r][k] * C[r][j],  k < part_nk
    // which the Adam kernel sums over mt in a fixed order (deterministic, no extra launch).
    const float *part_x;
    float *part;
    int part_nk, part_ldx;
    int tiles_n, tile_start, ntiles;
    // optional optimizer step in the epilogue (wgrad jobs; GemmJobs::ad.on): element (i, j) of C is
    // parameter adam_off + i*ldc + j of the flat buffers (-1: none).  For the job with the fused
    // layer-1 partials, adam_off addresses row 0 of that layer-1 kernel (its bias row follows it).
    long long adam_off;
    int vec_epi;  // plain wgrad-style epilogue (no bias / relu / mask / partials) with 16-B aligned rows: each thread owns four
                  // consecutive columns of one row — C and the optimizer state move as float4 (5 + 4 wide accesses instead of 20 + 16)
};
// Adam + polyak applied by the workgroup that produced a gradient tile (every wgrad tile is complete
// inside one workgroup: in-workgroup split-K).  The one gradient that is NOT complete inside a
// workgroup — the policy's layer 1, summed over the row tiles' partials — is stepped by a 16-block
// k_adam_polyak launch behind it.  (Stepping it inside this launch was measured: a last-arriver among
// the producers +3.5 us, polling consumer workgroups +7 us — store ack, counter, partial loads and
// the parameter stores are four dependent memory round trips behind the slowest producer; a
// __threadfence per producer, i.e. L2 write-back + invalidate, made the launch 5x slower.)
struct AdamCtx {
    int on;
    float *p, *t, *m, *v; // flat buffers in the internal layout, indexed like the gradient buffer
    float *g;
    const OptState *opt;  // optimizer state read by this step ...
    OptState *opt_next;   // ... and the copy it is advanced into (by ONE workgroup of the last launch of the step; nobody reads it there)
    long long n_pi;       // elements below n_pi belong to the policy optimizer
    float lr, b1, b2, eps, pk, pk1;
    unsigned int noise_adv;
};
constexpr int MAX_GEMM_JOBS = 12;
#ifdef DDRL_STAMPS
#define STAMP(i) do { if (st) st[(i)] = (long long)__builtin_readcyclecounter(); } while (0)
#define STAMP_ARG , long long *st
#define STAMP_PASS , st
#else
#define STAMP(i) do { } while (0)
#define STAMP_ARG
#define STAMP_PASS
#endif
struct GemmJobs {
#ifdef DDRL_STAMPS
    long long *stamps;  // diagnostic builds only (tools/gemm_bench.hip): [grid][32] cycle stamps of wave 0
#endif
    int njobs, total_tiles;
    int ks_max;   // deepest sub-chunk of any job in this launch (two-chunk loop): sizes the LDS tiles
    int op_lds;   // floats per operand tile = max(32 * (ks_max + 2), ks_max * 36, 32 * 36)
    int tile_start[MAX_GEMM_JOBS];  // flat copy: the job lookup is one scalar load, not a pointer chase
    AdamCtx ad;
    GemmJob job[MAX_GEMM_JOBS];
};

struct Seg {  // one tensor: dense external offset <-> internal offset
    long long ext, in, n;
    // cols == 0: contiguous.  cols > 0: a [n / cols][cols] kernel (row k, column j) stored
    //   mode 1  k4-interleaved (layer 2, sac1_direct.h): in + ((k / 4) * ld + j) * 4 + k % 4
    //   mode 2  as layer-1 MFMA operand blocks: in + w1y_index(d0 + k, j) — d0 = 0 for the kernel, = its row count for the bias
    int cols, ld, mode, d0;
};
// Layer-1 [W1 ; b1] in the order the transposed layer-1 MFMAs read it (sac1_direct.h): per block of 32 hidden units two
// float4 per lane — input column d sits in MFMA step s of lane half h (d < 8: h = d / 4, s = d % 4; else h = (d - 8) % 2,
// s = 4 + (d - 8) / 2), steps 0-3 in the first float4 and 4-7 in the second; within a (block, float4, half) the 32 units
// are contiguous: one fully coalesced 512-byte segment per half-wave load.  16 floats per hidden unit, unused slots zero.
__host__ __device__ __forceinline__ long long w1y_index(int d, int j) {
    const int h = d < 8 ? d >> 2 : (d - 8) & 1, s = d < 8 ? d & 3 : 4 + ((d - 8) >> 1);
    return ((((long long)(j >> 5) * 2 + (s >> 2)) * 2 + h) * 32 + (j & 31)) * 4 + (s & 3);
}

struct OptState {  // device-resident Adam bookkeeping (running beta powers like TF's beta*_power)
    float b1p_pi, b2p_pi, b1p_q, b2p_q;
    long long t_pi, t_q;
    unsigned int pad0, pad1;
    unsigned long long noise_ctr;
};

__device__ __forceinline__ void adam1(float g, float &m, float &v, float &p, float &t, float omb1, float omb2, float al,
                                      float eps, float pk, float pk1) {
    m = m + (g - m) * omb1;
    v = v + (g * g - v) * omb2;
    p = p - (m * al) / (sqrtf(v) + eps);
    t = pk * t + pk1 * p;  // polyak with the post-update main (actor_learner.py:85-87)
}

// ------------------------------------------------------------------------------------------
// K: batched small GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).
// One workgroup = one 32x32 output tile; its 4 waves split K four ways (in-workgroup split-K: at
// M = batch = 256 a stage has only ~80-130 tiles per network, so K is what fills the 1024 SIMDs)
// and the four partial tiles are combined through LDS in a fixed order (deterministic).
// Operand fetch, per 32-deep K sub-chunk and per wave: BOTH operands are read from global memory
// as float4 (full 128-B row segments: 8 lanes per row, 8 rows per instruction, 4 instructions per
// operand, no branches: clamped addresses + selects) and handed to the MFMA lanes through a
// wave-private LDS tile:
//   * operand contiguous along K in memory (activations of fwd/dgrad, W2 rows of dgrad):
//       LDS image [idx][k] (row stride 34), each lane picks float2 (k, k+1) of its row/col;
//   * operand contiguous along M/N (weights of fwd, both operands of wgrad):
//       LDS image [k][idx] (row stride 36), each lane picks two b32 of its column.
// The next sub-chunk's global loads are issued before the current sub-chunk's 16 MFMAs.
// Bias gradients need no special case: activations carry a physical column of ones
// (H1[:, h1] = H2[:, h2] = XA[:, obs+act] = 1), so [dW ; db] = [X | 1]^T dZ is one wgrad job.
// Jobs whose operands are not 16-B aligned / stride % 4 (odd layer sizes) take a slow path with
// per-element guards.
// ------------------------------------------------------------------------------------------
constexpr int KS = 32;             // sub-chunk depth of the streaming (slow-path) loop
constexpr int KS2 = 64;            // maximum sub-chunk depth of the two-chunk loop
constexpr int RED_LDS = 4 * 32 * 33; // floats of the split-K combine buffer (aliases the operand tiles)

__device__ __forceinline__ float4 sel4(bool c0, bool c1, bool c2, bool c3, float4 t) {
    return make_float4(c0 ? t.x : 0.f, c1 ? t.y : 0.f, c2 ? t.z : 0.f, c3 ? t.w : 0.f);
}

// Fast path, global -> registers: unconditional aligned float4 loads from clamped addresses; the
// values of out-of-range elements are discarded later (st_tile_fast), NOT here, so that all eight
// loads of a sub-chunk are in flight together.  KC: rows = idx, columns = k.  !KC: rows = k, cols = idx.
template <bool KC>
__device__ __forceinline__ void ld_tile_fast(const float *__restrict__ base, int ld, int idx0, int nidx, int kb, int k1, int lane,
                                             float4 (&v)[4]) {
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (KC ? idx0 : kb) + q * 8 + r8, col = (KC ? kb : idx0) + c4;
        const int nrow = KC ? nidx : k1, ncol = KC ? k1 : nidx;
        const bool ok = row < nrow && col < ncol;
        v[q] = *reinterpret_cast<const float4 *>(base + (long long)(ok ? row : 0) * ld + (ok ? col : 0));
    }
}

// registers -> wave-private LDS image, zeroing what lies outside the matrix / this wave's K range
template <bool KC>
__device__ __forceinline__ void st_tile_fast(float *__restrict__ s, int idx0, int nidx, int kb, int k1, int lane,
                                             const float4 (&v)[4]) {
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int lrow = q * 8 + r8;
        const int row = (KC ? idx0 : kb) + lrow, col = (KC ? kb : idx0) + c4;
        const int nrow = KC ? nidx : k1, ncol = KC ? k1 : nidx;
        const bool ok = row < nrow;
        if (KC) {  // K % 4 == 0: the four k of a float4 are valid or not as a whole
            const bool okk = ok && col < ncol;
            *reinterpret_cast<float2 *>(s + lrow * 34 + c4) = make_float2(okk ? v[q].x : 0.f, okk ? v[q].y : 0.f);
            *reinterpret_cast<float2 *>(s + lrow * 34 + c4 + 2) = make_float2(okk ? v[q].z : 0.f, okk ? v[q].w : 0.f);
        } else {
            *reinterpret_cast<float4 *>(s + lrow * 36 + c4) =
                sel4(ok && col < ncol, ok && col + 1 < ncol, ok && col + 2 < ncol, ok && col + 3 < ncol, v[q]);
        }
    }
}

// Slow path (unaligned / odd strides): per-element guarded scalar loads, values already masked.
template <bool KC>
__device__ __forceinline__ void ld_tile_slow(const float *__restrict__ base, int ld, int idx0, int nidx, int kb, int k1, int lane,
                                             float4 (&v)[4]) {
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gi = KC ? idx0 + q * 8 + r8 : idx0 + c4 + u;
            const int k = KC ? kb + c4 + u : kb + q * 8 + r8;
            const bool ok = gi < nidx && k < k1;
            const long long off = KC ? (long long)(ok ? gi : 0) * ld + (ok ? k : 0) : (long long)(ok ? k : 0) * ld + (ok ? gi : 0);
            const float t = base[off];
            e[u] = ok ? t : 0.f;
        }
        v[q] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

template <bool KC>
__device__ __forceinline__ void st_tile_slow(float *__restrict__ s, int lane, const float4 (&v)[4]) {
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = q * 8 + r8;
        if (KC) {
            *reinterpret_cast<float2 *>(s + row * 34 + c4) = make_float2(v[q].x, v[q].y);
            *reinterpret_cast<float2 *>(s + row * 34 + c4 + 2) = make_float2(v[q].z, v[q].w);
        } else {
            *reinterpret_cast<float4 *>(s + row * 36 + c4) = v[q];
        }
    }
}

template <bool KC>
__device__ __forceinline__ void rd_tile(const float *__restrict__ s, int l31, int h, float (&x)[8], float (&y)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int k = 4 * c + 2 * h;
        if (KC) {
            const float2 t = *reinterpret_cast<const float2 *>(s + l31 * 34 + k);
            x[c] = t.x; y[c] = t.y;
        } else {
            x[c] = s[k * 36 + l31];
            y[c] = s[(k + 1) * 36 + l31];
        }
    }
}

// ---- two-chunk K loop (fast path) ----------------------------------------------------------------
// A wave's K range (<= 128) is processed as exactly TWO balanced sub-chunks of depth ks <= 64
// (K/4 = 100 -> 52 + 48, 76 -> 40 + 36, 64 -> 32 + 32): cutting it into fixed 32-deep pieces left a
// nearly empty last piece that still paid the full stage/read overhead.  The LDS tiles are
// wave-private, so no workgroup barrier is needed inside the loop — only wave-level ordering of the
// wave's own LDS writes and reads (the LDS queue of a wave is in-order; the fence keeps the compiler
// and the counters honest).  The second chunk's global loads are issued as soon as the first chunk
// has been staged, i.e. they fly during the first chunk's LDS reads and 2 x 13 MFMAs.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // s_waitcnt lgkmcnt(0): this wave's LDS ops are done
    __builtin_amdgcn_wave_barrier();
}

template <bool KC>
struct Op2 {
    const float4 *p[8];  // per-lane source of instruction i in chunk 0
    bool okr[8];         // KC: row (idx) valid            NC: unused
    bool okc[4];         // NC: column idx + e valid        KC: unused
    int kk;              // KC: k offset of this lane's float4 inside a chunk (c4)   NC: unused
    int krow[8];         // NC: k row of instruction i inside a chunk               KC: unused
    __device__ __forceinline__ void init(const float *base, int ld, int idx0, int nidx, int k0, int lane) {
        if (KC) {  // image [32 idx][64 k]: 16 float4 per row, 4 rows per instruction
            const int r4 = lane >> 4;
            kk = (lane & 15) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = idx0 + i * 4 + r4;
                okr[i] = row < nidx;
                p[i] = reinterpret_cast<const float4 *>(base + ((okr[i] ? row : 0) * ld + k0 + kk));
            }
        } else {   // image [64 k][32 idx]: 8 float4 per k row, 8 rows per instruction
            const int r8 = lane >> 3, c4 = (lane & 7) * 4, col = idx0 + c4;
#pragma unroll
            for (int e = 0; e < 4; ++e) okc[e] = col + e < nidx;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                krow[i] = i * 8 + r8;
                p[i] = reinterpret_cast<const float4 *>(base + ((k0 + krow[i]) * ld + (okc[0] ? col : 0)));
            }
        }
    }
    // chunk `ch` (0/1) starts ks0 elements after k0 and is `ks` deep; loads outside it are skipped
    __device__ __forceinline__ void load(int ch, int ks0, int ks, int ld, float4 (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KC) {
                const bool ok = kk < ks;
                v[i] = ok ? p[i][(ch ? ks0 : 0) >> 2] : make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (i * 8 < ks) {  // uniform skip of instructions wholly beyond the chunk
                const bool ok = krow[i] < ks;
                v[i] = ok ? p[i][(long long)(ch ? ks0 : 0) * ld >> 2] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    // kcs = row stride of the K-contiguous image (ks_max + 2, even)
    __device__ __forceinline__ void store(float *__restrict__ s, int ks, int kcs, int lane, const float4 (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KC) {
                if (kk < ks) {  // lanes beyond the chunk have nothing to stage (their columns are never read)
                    const bool ok = okr[i];
                    float *d = s + (i * 4 + (lane >> 4)) * kcs + kk;
                    *reinterpret_cast<float2 *>(d) = make_float2(ok ? v[i].x : 0.f, ok ? v[i].y : 0.f);
                    *reinterpret_cast<float2 *>(d + 2) = make_float2(ok ? v[i].z : 0.f, ok ? v[i].w : 0.f);
                }
            } else if (krow[i] < ((ks + 3) & ~3)) {
                // rows beyond the chunk's last 4-k group are never read (and would overflow the tile);
                // rows inside that group but beyond ks ARE read by the MFMAs (K = batch need not be a
                // multiple of 4 for the wgrads) and must be zero, not whatever the LDS held before
                const bool in = krow[i] < ks;
                *reinterpret_cast<float4 *>(s + krow[i] * 36 + (lane & 7) * 4) =
                    make_float4(in && okc[0] ? v[i].x : 0.f, in && okc[1] ? v[i].y : 0.f, in && okc[2] ? v[i].z : 0.f,
                                in && okc[3] ? v[i].w : 0.f);
            }
        }
    }
};

template <bool AKC, bool BKC>
__device__ __forceinline__ void mfma_chunk(const float *__restrict__ sA, const float *__restrict__ sB, int ks, int kcs, int l31, int h,
                                           floatx16 &acc) {
    const int ng = (ks + 3) >> 2;  // 4-k groups in this chunk (scalar, <= 16)
#pragma unroll
    for (int c0 = 0; c0 < 16; c0 += 4) {
        if (c0 < ng) {
            float ax[4], ay[4], bx[4], by[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * (c0 + u) + 2 * h;
                if (AKC) { const float2 t = *reinterpret_cast<const float2 *>(sA + l31 * kcs + k); ax[u] = t.x; ay[u] = t.y; }
                else { ax[u] = sA[k * 36 + l31]; ay[u] = sA[(k + 1) * 36 + l31]; }
                if (BKC) { const float2 t = *reinterpret_cast<const float2 *>(sB + l31 * kcs + k); bx[u] = t.x; by[u] = t.y; }
                else { bx[u] = sB[k * 36 + l31]; by[u] = sB[(k + 1) * 36 + l31]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c0 + u < ng) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[u], bx[u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[u], by[u], acc, 0, 0, 0);
                }
            }
        }
    }
}

template <bool AKC, bool BKC>
__device__ __forceinline__ void gemm_kloop2(const GemmJob &jb, float *sA, float *sB, int kcs, int m0, int n0, int k0, int k1, int lane,
                                            floatx16 &acc STAMP_ARG) {
    const int l31 = lane & 31, h = lane >> 5;
    const int kw = k1 > k0 ? k1 - k0 : 0;                 // this wave's K range (multiple of 4 on the fast path)
    const int half = ((kw + 7) >> 3) << 2;
    const int ks_a = half < kw ? half : kw;               // first chunk: half, rounded up to a multiple of 4
    const int ks_b = kw - ks_a;                           // second chunk
    Op2<AKC> oa;
    Op2<BKC> ob;
    oa.init(jb.A, jb.lda, m0, jb.M, k0, lane);
    ob.init(jb.B, jb.ldb, n0, jb.N, k0, lane);
    float4 pa[8], pb[8];
    oa.load(0, 0, ks_a, jb.lda, pa);
    ob.load(0, 0, ks_a, jb.ldb, pb);
    STAMP(2);
    oa.store(sA, ks_a, kcs, lane, pa);
    ob.store(sB, ks_a, kcs, lane, pb);
    STAMP(3);
    if (ks_b > 0) {  // wave-uniform
        oa.load(1, ks_a, ks_b, jb.lda, pa);
        ob.load(1, ks_a, ks_b, jb.ldb, pb);
    }
    wave_lds_sync();
    STAMP(4);
    mfma_chunk<AKC, BKC>(sA, sB, ks_a, kcs, l31, h, acc);
    STAMP(5);
    if (ks_b > 0) {
        wave_lds_sync();  // chunk 0's LDS reads are complete before the tile is overwritten
        oa.store(sA, ks_b, kcs, lane, pa);
        ob.store(sB, ks_b, kcs, lane, pb);
        wave_lds_sync();
        STAMP(6);
        mfma_chunk<AKC, BKC>(sA, sB, ks_b, kcs, l31, h, acc);
        STAMP(7);
    }
}

// K loop of one wave for one tile; specialised on the operand layouts so that every variant is
// straight-line code (fwd: A along K, B along N; dgrad: both along K; wgrad: both along M/N).
template <bool AKC, bool BKC, bool FAST>
__device__ __forceinline__ void gemm_kloop(const GemmJob &jb, float *sA, float *sB, int m0, int n0, int k0, int k1, int nsub,
                                           int lane, floatx16 &acc STAMP_ARG) {
    const int l31 = lane & 31, h = lane >> 5;
    float4 pa[4], pb[4];
    if (FAST) {
        ld_tile_fast<AKC>(jb.A, jb.lda, m0, jb.M, k0, k1, lane, pa);
        ld_tile_fast<BKC>(jb.B, jb.ldb, n0, jb.N, k0, k1, lane, pb);
    } else {
        ld_tile_slow<AKC>(jb.A, jb.lda, m0, jb.M, k0, k1, lane, pa);
        ld_tile_slow<BKC>(jb.B, jb.ldb, n0, jb.N, k0, k1, lane, pb);
    }
    STAMP(2);
    for (int sub = 0; sub < nsub; ++sub) {
        const int kb = k0 + sub * KS;
        __syncthreads();  // WAR: the previous sub-chunk's LDS reads are done
        STAMP(3 + sub * 5);
        if (FAST) {
            st_tile_fast<AKC>(sA, m0, jb.M, kb, k1, lane, pa);
            st_tile_fast<BKC>(sB, n0, jb.N, kb, k1, lane, pb);
        } else {
            st_tile_slow<AKC>(sA, lane, pa);
            st_tile_slow<BKC>(sB, lane, pb);
        }
        STAMP(4 + sub * 5);
        __syncthreads();
        STAMP(5 + sub * 5);
        float ax[8], ay[8], bx[8], by[8];
        rd_tile<AKC>(sA, l31, h, ax, ay);
        rd_tile<BKC>(sB, l31, h, bx, by);
        if (sub + 1 < nsub) {  // prefetch the next sub-chunk behind the MFMAs
            if (FAST) {
                ld_tile_fast<AKC>(jb.A, jb.lda, m0, jb.M, kb + KS, k1, lane, pa);
                ld_tile_fast<BKC>(jb.B, jb.ldb, n0, jb.N, kb + KS, k1, lane, pb);
            } else {
                ld_tile_slow<AKC>(jb.A, jb.lda, m0, jb.M, kb + KS, k1, lane, pa);
                ld_tile_slow<BKC>(jb.B, jb.ldb, n0, jb.N, kb + KS, k1, lane, pb);
            }
        }
        STAMP(6 + sub * 5);
        const int nc = (k1 - kb + 3) >> 2;  // valid 4-k groups of this sub-chunk (scalar)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < nc) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[c], bx[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[c], by[c], acc, 0, 0, 0);
            }
        }
        STAMP(7 + sub * 5);
    }
}

// Leading scalar arguments (total tiles, the flat tile_start table) are preloaded into SGPRs at wave
// launch; fields of the by-value struct behind them are s_load'ed from the kernarg segment, cold after
// every kernel boundary — with everything in the struct the ISA showed three dependent round trips
// (gridDim, tile_start, job record) before the first vector load; now it is one (the job record).
__global__ void __launch_bounds__(256) k_gemm(int total_tiles, int ts1, int ts2, int ts3, int ts4, int ts5, int ts6, int ts7, int ts8, int ts9,
                                              int ts10, int ts11, GemmJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 4 waves x (A tile, B tile) of jobs.op_lds floats; reused for the split-K combine
    // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs, whose L2s
    // are private and cold after every kernel boundary.  Give each XCD a CONTIGUOUS run of tiles,
#ifdef DDRL_STAMPS
    const long long t0_stamp = (long long)__builtin_readcyclecounter();
#endif
    // and order tiles panel-major (all M-tiles of one N-panel of one job are consecutive), so that a
    // W2 / dZ panel and an activation matrix are fetched by one or two L2s instead of all eight.
    int t, ji = 0;
    {
        const int nwg = total_tiles, b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    static_assert(MAX_GEMM_JOBS == 12, "k_gemm takes tile_start[1..11] as scalar arguments");
    ji = (t >= ts1) + (t >= ts2) + (t >= ts3) + (t >= ts4) + (t >= ts5) + (t >= ts6) + (t >= ts7) + (t >= ts8) + (t >= ts9) + (t >= ts10) +
         (t >= ts11);
    const GemmJob &jb = jobs.job[ji];
    t -= jb.tile_start;
    const int tiles_m = jb.ntiles / jb.tiles_n;
    const int m0 = (t % tiles_m) * 32, n0 = (t / tiles_m) * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: K range and loop guards stay scalar
    const int l31 = lane & 31, h = lane >> 5;
    const int K = jb.K;
    const int chunk = ((K + 15) >> 4) << 2;
    const int k0 = w * chunk;
    const int k1 = (k0 + chunk < K) ? (k0 + chunk) : K;
    const int nsub = (chunk + KS - 1) / KS;  // same for the four waves (barriers in the K loop)
    float *sA = smem + (w * 2 + 0) * jobs.op_lds, *sB = smem + (w * 2 + 1) * jobs.op_lds;
    const int kcs = jobs.ks_max + 2;
#ifdef DDRL_STAMPS
    long long *st = (jobs.stamps && lane == 0 && w == 0) ? jobs.stamps + (long long)blockIdx.x * 32 : nullptr;
    if (st) st[0] = t0_stamp;
#endif
    STAMP(1);
    // epilogue operands first: their latency hides behind the whole K loop
    __shared__ float s_px[32][13];
    if (jb.part) {
        for (int idx = tid; idx < 32 * 12; idx += 256) {
            const int rr = idx / 12, k = idx - rr * 12;
            const int gi = m0 + rr;
            const bool ok = gi < jb.M && k < jb.part_nk;
            const float v = jb.part_x[(long long)(ok ? gi : 0) * jb.part_ldx + (ok ? k : 0)];
            s_px[rr][k] = ok ? v : 0.f;
        }
    }
    float biasv[4], maskv[4];
    bool okv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = tid + 256 * q;
        const int gi = m0 + (o >> 5), gj = n0 + (o & 31);
        okv[q] = gi < jb.M && gj < jb.N;
        const int gic = okv[q] ? gi : 0, gjc = okv[q] ? gj : 0;
        biasv[q] = jb.bias ? jb.bias[gjc] : 0.f;
        maskv[q] = jb.mask ? jb.mask[(long long)gic * jb.ldmask + gjc] : 1.f;
    }
    // optimizer state of this tile's parameters (fused step): fetched now, used in the epilogue
    const bool do_adam = jobs.ad.on && jb.adam_off >= 0 && !jb.part;  // block-uniform
    float am[4], av[4], ap[4], at[4];
    float al_pi = 0.f, al_q = 0.f;
    if (jobs.ad.on) {
        const float b1p_pi = jobs.ad.opt->b1p_pi, b2p_pi = jobs.ad.opt->b2p_pi, b1p_q = jobs.ad.opt->b1p_q, b2p_q = jobs.ad.opt->b2p_q;
        al_pi = jobs.ad.lr * sqrtf(1.0f - b2p_pi) / (1.0f - b1p_pi);
        al_q = jobs.ad.lr * sqrtf(1.0f - b2p_q) / (1.0f - b1p_q);
    }
    const bool vec = jb.vec_epi != 0;  // block-uniform
    if (do_adam && vec) {
        const int vrow = tid >> 3, vc4 = (tid & 7) * 4;
        const bool vok = m0 + vrow < jb.M && n0 + vc4 < jb.N;
        const long long ic = vok ? jb.adam_off + (long long)(m0 + vrow) * jb.ldc + n0 + vc4 : jb.adam_off;
        const float4 m4 = *reinterpret_cast<const float4 *>(jobs.ad.m + ic), v4 = *reinterpret_cast<const float4 *>(jobs.ad.v + ic);
        const float4 p4 = *reinterpret_cast<const float4 *>(jobs.ad.p + ic), t4 = *reinterpret_cast<const float4 *>(jobs.ad.t + ic);
        am[0] = m4.x; am[1] = m4.y; am[2] = m4.z; am[3] = m4.w; av[0] = v4.x; av[1] = v4.y; av[2] = v4.z; av[3] = v4.w;
        ap[0] = p4.x; ap[1] = p4.y; ap[2] = p4.z; ap[3] = p4.w; at[0] = t4.x; at[1] = t4.y; at[2] = t4.z; at[3] = t4.w;
    } else if (do_adam) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = tid + 256 * q;
            const int gi = m0 + (o >> 5), gj = n0 + (o & 31);
            const long long idx = okv[q] ? jb.adam_off + (long long)gi * jb.ldc + gj : 0;
            am[q] = jobs.ad.m[idx]; av[q] = jobs.ad.v[idx]; ap[q] = jobs.ad.p[idx]; at[q] = jobs.ad.t[idx];
        }
    }
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // (A variant that fetched a wave's whole K range up front was measured: no faster for the
    // forward launches, 40 % slower for the 592-tile backward launch — the CU's fetch rate, not the
    // prefetch distance, is the limit; the one-sub-chunk-ahead streaming loop stays.)
    const int variant = (jb.fast ? (chunk <= 2 * KS2 ? 8 : 4) : 0) + (jb.a_kc ? 2 : 0) + (jb.b_kc ? 1 : 0);
    switch (variant) {
        case 8 + 2 + 0: gemm_kloop2<true, false>(jb, sA, sB, kcs, m0, n0, k0, k1, lane, acc STAMP_PASS); break;   // fwd
        case 8 + 2 + 1: gemm_kloop2<true, true>(jb, sA, sB, kcs, m0, n0, k0, k1, lane, acc STAMP_PASS); break;    // dgrad
        case 8 + 0 + 0: gemm_kloop2<false, false>(jb, sA, sB, kcs, m0, n0, k0, k1, lane, acc STAMP_PASS); break;  // wgrad
        case 4 + 2 + 0: gemm_kloop<true, false, true>(jb, sA, sB, m0, n0, k0, k1, nsub, lane, acc STAMP_PASS); break;   // K > 512
        case 4 + 2 + 1: gemm_kloop<true, true, true>(jb, sA, sB, m0, n0, k0, k1, nsub, lane, acc STAMP_PASS); break;
        case 4 + 0 + 0: gemm_kloop<false, false, true>(jb, sA, sB, m0, n0, k0, k1, nsub, lane, acc STAMP_PASS); break;
        case 2 + 0: gemm_kloop<true, false, false>(jb, sA, sB, m0, n0, k0, k1, nsub, lane, acc STAMP_PASS); break;
        case 2 + 1: gemm_kloop<true, true, false>(jb, sA, sB, m0, n0, k0, k1, nsub, lane, acc STAMP_PASS); break;
        default: gemm_kloop<false, false, false>(jb, sA, sB, m0, n0, k0, k1, nsub, lane, acc STAMP_PASS); break;
    }
    // split-K combine.  D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    STAMP(28);
    __syncthreads();
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(smem);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][(r & 3) + 8 * (r >> 2) + 4 * h][l31] = acc[r];
    __syncthreads();
    float outv[4];
    if (vec) {
        const int vrow = tid >> 3, vc4 = (tid & 7) * 4;
        const bool vok = m0 + vrow < jb.M && n0 + vc4 < jb.N;  // N % 4 == 0: the four columns are valid together
        const long long vidx = jb.adam_off + (long long)(m0 + vrow) * jb.ldc + n0 + vc4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            outv[e] = ((red[0][vrow][vc4 + e] + red[1][vrow][vc4 + e]) + red[2][vrow][vc4 + e]) + red[3][vrow][vc4 + e];
        if (vok) {
            *reinterpret_cast<float4 *>(jb.C + (long long)(m0 + vrow) * jb.ldc + n0 + vc4) = make_float4(outv[0], outv[1], outv[2], outv[3]);
            if (do_adam) {
                const float al = vidx < jobs.ad.n_pi ? al_pi : al_q;  // a tile never straddles the two optimizers (pairs are 16-B aligned)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    adam1(outv[e], am[e], av[e], ap[e], at[e], 1.0f - jobs.ad.b1, 1.0f - jobs.ad.b2, al, jobs.ad.eps, jobs.ad.pk, jobs.ad.pk1);
                *reinterpret_cast<float4 *>(jobs.ad.m + vidx) = make_float4(am[0], am[1], am[2], am[3]);
                *reinterpret_cast<float4 *>(jobs.ad.v + vidx) = make_float4(av[0], av[1], av[2], av[3]);
                *reinterpret_cast<float4 *>(jobs.ad.p + vidx) = make_float4(ap[0], ap[1], ap[2], ap[3]);
                *reinterpret_cast<float4 *>(jobs.ad.t + vidx) = make_float4(at[0], at[1], at[2], at[3]);
            }
        }
    } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = tid + 256 * q;
        const int row = o >> 5, col = o & 31;
        outv[q] = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = tid + 256 * q;
        const int gi = m0 + (o >> 5), gj = n0 + (o & 31);
        float v = outv[q];
        if (jb.bias) v += biasv[q];
        if (jb.relu) v = fmaxf(v, 0.f);
        v = maskv[q] > 0.f ? v : 0.f;
        if (okv[q]) jb.C[(long long)gi * jb.ldc + gj] = v;
        outv[q] = okv[q] ? v : 0.f;
        if (do_adam && okv[q]) {
            const long long idx = jb.adam_off + (long long)gi * jb.ldc + gj;
            const float al = idx < jobs.ad.n_pi ? al_pi : al_q;
            adam1(v, am[q], av[q], ap[q], at[q], 1.0f - jobs.ad.b1, 1.0f - jobs.ad.b2, al, jobs.ad.eps, jobs.ad.pk, jobs.ad.pk1);
            jobs.ad.m[idx] = am[q]; jobs.ad.v[idx] = av[q]; jobs.ad.p[idx] = ap[q]; jobs.ad.t[idx] = at[q];
        }
    }
    }
    if (jb.part) {  // block-uniform
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = tid + 256 * q;
            red[0][o >> 5][o & 31] = outv[q];
        }
        __syncthreads();
        const int mt = m0 >> 5;
        for (int idx = tid; idx < jb.part_nk * 32; idx += 256) {
            const int k = idx >> 5, col = idx & 31;
            float sacc = 0.f;
#pragma unroll
            for (int rr = 0; rr < 32; ++rr) sacc = fmaf(s_px[rr][k], red[0][rr][col], sacc);
            if (n0 + col < jb.N) jb.part[((long long)mt * jb.part_nk + k) * jb.N + n0 + col] = sacc;
        }
    }
    STAMP(29);
}

// ------------------------------------------------------------------------------------------
// wave-level helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int RV = 8;  // row vectors of up to 512 elements live in 8 registers per lane

// v[i] = p[lane + 64 i] (address clamped for lane + 64 i >= n: the value there is unspecified —
// mask_row() one operand of a product before use).  Eight independent loads, no branch, no select:
// a kernel issues ALL its load_row calls first so that they share one memory round trip.
__device__ __forceinline__ void load_row(const float *__restrict__ p, int n, int lane, float (&v)[RV], int stride = 1, int off = 0) {
#pragma unroll
    for (int i = 0; i < RV; ++i) {
        const int j = lane + 64 * i;
        v[i] = p[(long long)(j < n ? j : 0) * stride + off];
    }
}
__device__ __forceinline__ void mask_row(float (&v)[RV], int n, int lane) {
#pragma unroll
    for (int i = 0; i < RV; ++i) v[i] = (lane + 64 * i < n) ? v[i] : 0.f;
}
__device__ __forceinline__ float dot_rv(const float (&a)[RV], const float (&b)[RV]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RV; ++i) s = fmaf(a[i], b[i], s);
    return s;
}

// ------------------------------------------------------------------------------------------
// K: Adam(pi) + Adam(q1,q2) + polyak, fused over the flat (padded) parameter buffer.
// tf.train.AdamOptimizer's ApplyAdam update form; actor_learner.py:73-87.
// ------------------------------------------------------------------------------------------
struct AdamArgs {
    float *p, *t, *m, *v;
    float *g;
    const OptState *opt;  // read ...
    OptState *opt_next;   // ... and advanced into (by thread 0 of block 0)
    long long n, n_pi;
    long long base4;      // first float4 element of this launch (0: the whole buffer)
    float lr, b1, b2, eps, pk, pk1;
    // fused pi layer-1 wgrad: gradient of float4 elements [part_off4, part_off4 + part_n4) is the sum of
    // `nparts` row-tile partials written by the pi dgrad tiles (k_gemm epilogue), summed in tile order
    const float *part;
    long long part_off4, part_n4, part_stride4;
    int nparts;
    unsigned int noise_adv;  // noise elements consumed by this update (advances the device counter)
    // optional: one extra workgroup samples the NEXT update's batch (ddrl_sac1_apply_grads_and_sample)
    int adam_blocks, do_sample, sample_batch;
    ddrl_replay_dev::RingState *rs;
    ddrl_replay_dev::RingPtrs ring;
    ddrl_replay_dev::BatchPtrs sout;
    // optional: one extra workgroup (block adam_blocks, do_sample == 0) finishes a loss mean from per-workgroup partials that an
    // earlier launch left (k_dqn_head): loss_out[0] = loss_scale * sum_b loss_part[b], b in order
    const float *loss_part;
    float *loss_out;
    float *loss_out2;        // optional second destination (the caller's buffer: a copy launch per update otherwise)
    int loss_n;
    float loss_scale;
    // optional (direct path, split API): the dgrad images [Np/4][ld][4] of up to four k4-interleaved layer-2 kernels [K/4][Np][4] are
    // rewritten from the stepped values by the threads that step them — the flat step of a data-parallel learner then needs no
    // k_shadow launch behind it (sh_off4[s]: first float4 of kernel s in the flat buffer; it holds (ld / 4) * Np float4)
    int n_sh, sh_K, sh_N, sh_Np, sh_ld;
    long long sh_off4[4];
    float *sh_dst[4];
};
__global__ void __launch_bounds__(256) k_adam_polyak(AdamArgs a) {
    if (a.loss_part && !a.do_sample && (int)blockIdx.x == a.adam_blocks) {
        __shared__ float s_part[256];
        float tot = 0.f;
        for (int b0 = 0; b0 < a.loss_n; b0 += 256) {
            s_part[threadIdx.x] = b0 + (int)threadIdx.x < a.loss_n ? a.loss_part[b0 + threadIdx.x] : 0.f;
            __syncthreads();
            if (threadIdx.x == 0) {
                const int n = a.loss_n - b0 < 256 ? a.loss_n - b0 : 256;
                for (int b = 0; b < n; ++b) tot += s_part[b];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            a.loss_out[0] = a.loss_scale * tot;
            if (a.loss_out2) a.loss_out2[0] = a.loss_scale * tot;
        }
        return;
    }
    if (a.do_sample && (int)blockIdx.x == a.adam_blocks) {
        // Rides along: `idxs = np.random.randint(0, size, B)` + the five gathers of the NEXT update
        // (example/dsac.py:39-45) into the learner's other input set.  Adam touches no input set and
        // the sampler touches no parameter, so the two are independent; this takes the 6.6 us
        // single-workgroup sampler kernel off the dependent chain.
        ddrl_replay_dev::sample_block(a.rs, a.ring, a.sout, a.sample_batch, nullptr, 1);
        return;
    }
    const long long n4 = a.n >> 2, npi4 = a.n_pi >> 2;  // both buffers are padded to multiples of 4
    const long long i = a.base4 + (long long)blockIdx.x * blockDim.x + threadIdx.x;  // grid covers [base4, n4) exactly once
    float4 *P = reinterpret_cast<float4 *>(a.p), *T = reinterpret_cast<float4 *>(a.t);
    float4 *M = reinterpret_cast<float4 *>(a.m), *V = reinterpret_cast<float4 *>(a.v);
    const float4 *G = reinterpret_cast<const float4 *>(a.g);
    const long long ic = i < n4 ? i : a.base4;
    const bool from_parts = a.nparts > 0 && i >= a.part_off4 && i < a.part_off4 + a.part_n4;
    float4 g = G[ic];  // all loads of the kernel issued together
    float4 m = M[ic], v = V[ic], p = P[ic], t = T[ic];
    if (from_parts) {
        const float4 *PP = reinterpret_cast<const float4 *>(a.part) + (i - a.part_off4);
        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q0 = 0; q0 < a.nparts; q0 += 8) {  // 8 partials per round trip, summed in tile order
            float4 u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = PP[(long long)(q0 + q < a.nparts ? q0 + q : 0) * a.part_stride4];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q0 + q < a.nparts) { sacc.x += u[q].x; sacc.y += u[q].y; sacc.z += u[q].z; sacc.w += u[q].w; }
        }
        g = sacc;
        reinterpret_cast<float4 *>(a.g)[i] = g;  // keep the gradient buffer complete (export / all-reduce)
    }
    const float b1p_pi = a.opt->b1p_pi, b2p_pi = a.opt->b2p_pi, b1p_q = a.opt->b1p_q, b2p_q = a.opt->b2p_q;
    const float one = 1.0f;
    const float al_pi = a.lr * sqrtf(one - b2p_pi) / (one - b1p_pi);
    const float al_q = a.lr * sqrtf(one - b2p_q) / (one - b1p_q);
    const float omb1 = one - a.b1, omb2 = one - a.b2;
    if (i < n4) {
        const float al = i < npi4 ? al_pi : al_q;
        adam1(g.x, m.x, v.x, p.x, t.x, omb1, omb2, al, a.eps, a.pk, a.pk1);
        adam1(g.y, m.y, v.y, p.y, t.y, omb1, omb2, al, a.eps, a.pk, a.pk1);
        adam1(g.z, m.z, v.z, p.z, t.z, omb1, omb2, al, a.eps, a.pk, a.pk1);
        adam1(g.w, m.w, v.w, p.w, t.w, omb1, omb2, al, a.eps, a.pk, a.pk1);
        M[i] = m; V[i] = v; P[i] = p; T[i] = t;
        for (int sh = 0; sh < a.n_sh; ++sh) {
            const long long e = i - a.sh_off4[sh];
            if (e < 0 || e >= (long long)(a.sh_ld >> 2) * a.sh_Np) continue;
            // float4 e = rows k = 4 kq .. 4 kq + 3 of column j; the image holds column group j / 4 of row k at ((j / 4) * ld + k) * 4 + j % 4
            const int kq = (int)(e / a.sh_Np), j = (int)(e - (long long)kq * a.sh_Np);
            if (4 * (j >> 2) >= a.sh_N) break;
            float *c = a.sh_dst[sh] + ((long long)(j >> 2) * a.sh_ld + 4 * kq) * 4 + (j & 3);
            const float pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * kq + q < a.sh_K) c[4 * q] = pv[q];
            break;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // nobody reads the copy this writes
        OptState n = *a.opt;
        n.b1p_pi = b1p_pi * a.b1; n.b2p_pi = b2p_pi * a.b2; n.b1p_q = b1p_q * a.b1; n.b2p_q = b2p_q * a.b2;
        n.t_pi += 1; n.t_q += 1;
        n.noise_ctr += a.noise_adv;
        *a.opt_next = n;
    }
}

// ------------------------------------------------------------------------------------------
// K: dense external layout <-> padded internal layout; staging of the caller's batch
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack(const Seg *__restrict__ segs, const float *__restrict__ src, float *__restrict__ dst,
                                              float *__restrict__ dst2, int to_internal) {
    const Seg s = segs[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.n; i += (long long)gridDim.x * 256) {
        long long ii = i;
        if (s.cols > 0) {
            const long long k = i / s.cols, j = i - k * s.cols;
            ii = s.mode == 2 ? w1y_index(s.d0 + (int)k, (int)j) : ((k >> 2) * s.ld + j) * 4 + (k & 3);
        }
        if (to_internal) {
            const float v = src[s.ext + i];
            dst[s.in + ii] = v;
            if (dst2) dst2[s.in + ii] = v;
        } else {
            dst[s.ext + i] = src[s.in + ii];
        }
    }
}

__global__ void k_fill_col(float *p, long long rows, int ld, int col, float v) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r < rows) p[r * ld + col] = v;
}

static inline long long pad4(long long n) { return (n + 3) & ~3ll; }

static size_t gemm_smem(const GemmJobs &js);
static void launch_gemm(const GemmJobs &J, hipStream_t s) {
    const int *ts = J.tile_start;
    k_gemm<<<J.total_tiles, 256, gemm_smem(J), s>>>(J.total_tiles, ts[1], ts[2], ts[3], ts[4], ts[5], ts[6], ts[7], ts[8], ts[9], ts[10], ts[11], J);
}
static size_t gemm_smem(const GemmJobs &js) {
    const size_t a = (size_t)4 * 2 * js.op_lds * sizeof(float), b = (size_t)RED_LDS * sizeof(float);
    return a > b ? a : b;
}
static void gemm_add(GemmJobs &js, GemmJob j) {
    j.vec_epi = (!j.bias && !j.mask && !j.relu && !j.part && j.N % 4 == 0 && j.ldc % 4 == 0 && (((uintptr_t)j.C) & 15) == 0) ? 1 : 0;
    j.tiles_n = (j.N + 31) / 32;
    j.ntiles = ((j.M + 31) / 32) * j.tiles_n;
    j.tile_start = js.total_tiles;
    js.total_tiles += j.ntiles;
    for (int i = js.njobs; i < MAX_GEMM_JOBS; ++i) js.tile_start[i] = 0x7fffffff;
    js.tile_start[js.njobs] = j.tile_start;
    js.job[js.njobs++] = j;
    {   // LDS tile geometry of the launch
        const int chunk = ((j.K + 15) >> 4) << 2;
        int half = ((chunk + 7) >> 3) << 2;
        if (half > chunk) half = chunk;
        if (!j.fast || chunk > 2 * KS2) half = KS;  // streaming loop: 32-deep tiles
        if (half > js.ks_max) js.ks_max = half;
        if (js.ks_max < KS) js.ks_max = KS;
        const int a = 32 * (js.ks_max + 2), b = js.ks_max * 36;
        js.op_lds = a > b ? a : b;
    }
}
static bool al16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
static void set_fast(GemmJob &j) {
    bool ok = al16(j.A) && al16(j.B) && (j.lda % 4 == 0) && (j.ldb % 4 == 0);
    if (j.a_kc || j.b_kc) ok = ok && (j.K % 4 == 0);
    if (!j.a_kc) ok = ok && j.M <= j.lda;  // float4 reads stay inside the row stride
    if (!j.b_kc) ok = ok && j.N <= j.ldb;
    j.fast = ok ? 1 : 0;
}
// H2 = relu(H1 * W2 + b2)
static GemmJob gemm_fwd(const float *H1, int ldh1, const float *W2, const float *b2, float *H2, int ldh2, int rows, int h1, int h2) {
    GemmJob j{};
    j.adam_off = -1;
    j.A = H1; j.B = W2; j.C = H2; j.bias = b2; j.mask = nullptr;
    j.M = rows; j.N = h2; j.K = h1; j.lda = ldh1; j.ldb = h2; j.ldc = ldh2; j.ldmask = 0; j.a_kc = 1; j.b_kc = 0; j.relu = 1;
    set_fast(j);
    return j;
}
// dZ1 = (dZ2 * W2^T) .* (H1 > 0)
static GemmJob gemm_dgrad(const float *dZ2, const float *W2, const float *H1mask, int ldh1, float *dZ1, int rows, int h1, int h2) {
    GemmJob j{};
    j.adam_off = -1;
    j.A = dZ2; j.B = W2; j.C = dZ1; j.bias = nullptr; j.mask = H1mask;
    j.M = rows; j.N = h1; j.K = h2; j.lda = h2; j.ldb = h2; j.ldc = h1; j.ldmask = ldh1; j.a_kc = 1; j.b_kc = 1; j.relu = 0;
    set_fast(j);
    return j;
}
// [dW ; db] = [X | 1]^T * dZ : X[rows, nin | 1] (row stride ldx, physical ones column at nin), dZ[rows, nout]
// (row stride ldz) -> C[(nin+1), nout] (row stride ldc); the kernel's bias lives right behind it.
static GemmJob gemm_wgrad(const float *X, int ldx, int nin, const float *dZ, int ldz, int nout, float *C, int ldc, int rows) {
    GemmJob j{};
    j.adam_off = -1;
    j.A = X; j.B = dZ; j.C = C; j.bias = nullptr; j.mask = nullptr;
    j.M = nin + 1; j.N = nout; j.K = rows; j.lda = ldx; j.ldb = ldz; j.ldc = ldc; j.ldmask = 0; j.a_kc = 0; j.b_kc = 0; j.relu = 0;
    set_fast(j);
    return j;
}

}  // namespace
