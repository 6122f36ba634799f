// Direct-operand stages of the SAC1 update (included by sac1.hip, inside its anonymous namespace).
//
// Round-1's fused path staged every GEMM operand through LDS (coalesced float4 loads -> wave-private
// transposition tile -> MFMA lane layout).  Measured on MI355X the matrix cores run at ~1.6 GHz under
// this load and a stage is bounded by (MFMA issue) + (load -> LDS -> MFMA latency chain), so this path
// removes the LDS hop altogether:
//
//   * every GEMM operand lives in memory in an "x4" layout — the CONTRACTION index interleaved by 4,
//     the free index contiguous:  X4[k/4][i][k%4]  (ld = row stride of the free index).  One float4
//     load per lane is then the operand of FOUR v_mfma_f32_32x32x2_f32 steps: in step e of group g
//     the lanes of half h supply contraction index 8g + 4h + e (the order of a sum's terms is a free
//     choice; both operands use the same one).  32 lanes x 16 B = 512 contiguous bytes per half-wave.
//   * layer 1 is generated TRANSPOSED on the matrix cores: D = X1^T = W1^T x^T puts the batch row in
//     the lane and the hidden unit in the register — exactly the A-operand layout of the layer-2
//     MFMAs (unit (r&3) + 8(r>>2) + 4h in step r), so the accumulator registers feed the next MFMA
//     after one v_max each; the matching W2 image is the k4-interleaved parameter layout itself.
//   * a matrix that is contracted along different axes by different GEMMs is kept in both layouts
//     (W2 of the main networks: [h1/4][h2][4] for the forward + [h2/4][h1][4] for the dgrad, both
//     written by the optimizer epilogue; H1 / H2 / dZ: written in the layouts their consumers need
//     by the epilogue that has the tile in LDS anyway).
//
// Launches of one update (all captured in the learner's hipGraph; tiles at config 2):
//   k_dfwd<0>  evaluations 0-2: pi(x), pi(x2), pi_targ(x2) (layer 1 + layer 2 + head partials)                         240 tiles
//   k_dfwd<1>  evaluations 5-7, action from the policy-head partials, + the stored-action evaluations 3-4 (q1(x,a),
//              q2(x,a)) beside them (+ the next update's sampler workgroup)                                       240 + 160 + 1
//   k_dg "bq"  the three Q dgrads, dZ2 generated from H2 on the fly; losses, dq; dQ/da partials; W3 snapshot              312
//   k_dg "mid" the policy dgrad with its A operand generated in the tile, the policy-head backward tiles (images for the
//              wgrads), the Q layer-2 / head wgrads with Adam + polyak                                                      464
//   k_dg "pi"  policy layer-2 / head / layer-1 wgrads, Q layer-1 wgrads, Adam + polyak, loss means + optimizer bookkeeping  190

#ifdef DDRL_STAMPS  // diagnostic builds only (tools/upd_bench.hip): per-workgroup cycle stamps of thread 0
// the stamp buffer travels in the kernel arguments ([kernel id][1024 workgroups][16]): a __device__ pointer variable would
// have to be LOADED in front of every stamp, and the compiler drains every outstanding load before it uses that value
static unsigned long long *g_st_host = nullptr;
#define DST(kid, i) do { if (st_ && threadIdx.x == 0) st_[((kid) * 1024 + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define DRT(kid, i) do { if (st_ && threadIdx.x == 0) st_[((kid) * 1024 + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DST(kid, i) do { } while (0)
#define DRT(kid, i) do { } while (0)
#endif

constexpr int DFH = 8;    // head slots per evaluation (2 * act_dim <= 8)
constexpr int DNT = 16;   // n-tile slots of a head-partial row (hidden2 <= 512)
constexpr int DGMAX = 16; // 8-deep contraction groups per wave (contraction <= 512)

__device__ __forceinline__ float f4e(const float4 &v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// Activation images are written once and read by later launches on other XCDs: streaming (nontemporal) stores leave no
// dirty lines for the end-of-kernel write-back and do not push the optimizer state out of the L2s (50.8 -> 49.5 us per
// update; the same hint on the optimizer-state stores themselves: no change).
// (`tools/xcd_latency_bench.hip`: a dependent load of data the previous launch wrote costs 237 cycles on the writer's own XCD,
// 550 on another XCD after plain or sc1 (write-through) stores, 620-870 after nontemporal ones.  In the update: images
// nontemporal 49.6 us, sc1 50.8; the smaller dgrad outputs sc1 49.4.)
__device__ __forceinline__ void st_img(float *p, const float4 &v) {   // activation images: streaming
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(p));
}
__device__ __forceinline__ void st_dg(float *p, const float4 &v) {    // dgrad outputs: write-through
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_opt(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }   // optimizer state: plain
__device__ __forceinline__ int d_slot(int s, int h) { return s < 4 ? 4 * h + s : 8 + 2 * (s - 4) + h; }
__device__ __forceinline__ float relu1(float x) {
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));  // fmaxf() costs a canonicalising v_max in front
    return y;
}
static inline int rup32(int x) { return (x + 31) & ~31; }

// Kernel-argument records are fetched with scalar loads from the kernarg segment, cold after every kernel boundary: a
// field first touched deep inside a kernel costs a full miss (~0.3 us) at that point, and the compiler issues such loads
// one by one where the fields are used.  kline() starts one load per 64-byte line of a record at the top of the kernel;
// ktouch() (placed behind the first vector loads) waits for them, so that every later field load hits the scalar cache.
template <int BYTES>
__device__ __forceinline__ int kline(const void *rec) {
    const int *q = reinterpret_cast<const int *>(rec);
    int acc = q[(BYTES - 4) / 4];
#pragma unroll
    for (int o = 0; o < BYTES - 4; o += 64) acc |= q[o / 4];
    return acc;
}
__device__ __forceinline__ void ktouch(int v) { asm volatile("" ::"s"(v)); }

// ==========================================================================================
// forward stages
// ==========================================================================================
struct DFJob {
    const float *b2;         // [Np2] (pads are zero)
    const float *wh0, *wh1;  // head kernels — policy: Wmu, Wls ([h2][act]); Q: W3 ([h2])
    int nh, hsplit, hstride;
    float *hp;               // head partials [DFH][B][DNT]
    float *H2c4;             // [Np2/4][B][4]   (nullable)  A operand of the dgrads / relu mask of the policy head backward
    float *H2r4;             // [B/4][Lp2][4]   (nullable)  A operand of the head wgrads, relu mask of the generated Q wgrad operand
    float *H1r4;             // [B/4][Lp1][4]   (nullable, n-tile-0 workgroups)  A operand of the W2 wgrads, relu mask of the dgrads
    float *aug;              // [B][aug_ld] row-major [in0 | in1] rows (nullable, n-tile-0): layer-1 wgrad partial operand (policy)
    float *xr4;              // [B/4][32][4] augmented input rows (nullable, n-tile-0): A operand of the Q layer-1 wgrads
    int aug_ld;
    // phase 1: the policy evaluation whose sampled action is this job's second input
    const float *php, *pbmu, *pbls, *peps;
    int side;                // n-tile-0 workgroups: 1 -> act0, logp0, save0   2 -> act2, and logp1 from php1
};
struct DFArgs {
    int njobs, tiles_n, act, Lp1, Lp2, h2;
    float scale;
    float *act0, *act2, *logp0, *logp1, *save0;
    const float *php1, *pbmu1, *pbls1, *peps1;  // pi_main @ x2: only its log-prob is needed (actor_learner.py:62); php1 == nullptr: no such evaluation (SAC-v)
#ifdef DDRL_STAMPS
    unsigned long long *st;
#endif
    int pev_pack;            // phase 1: 2 bits per job = the policy evaluation (head-partial slot) its sampled action comes from
    int noise_on, n_each;    // n_each = valid rows * act (element index of a flat [3][rows * act] fill)
    int Bv;                  // valid rows: the batch; rows up to the next multiple of 32 are padding (zero inputs, no loss terms)
    uint32_t noise_seed;
    float *e0, *e1, *e2;
    const OptState *opt;
    int do_sample, sample_batch;
    ddrl_replay_dev::RingState *rs;
    ddrl_replay_dev::RingPtrs ring;
    ddrl_replay_dev::BatchPtrs sout;
    DFJob job[5];
};
// leading scalars of k_dfwd (preloaded into SGPRs at wave launch: the first loads need nothing else)
struct DFHead {
    const float *base;  // every offset below is relative to the slab
    int tpj, tiles_m;   // tiles per job, row tiles
    int K, Np;          // hidden1, padded hidden2 (row stride of the k4-interleaved W2)
    int B, d0;          // batch, obs_dim
    int main_off, targ_off;  // the main / target parameter buffers
    int npi, perq;           // inside a buffer: the policy, then q1 at npi, q2 at npi + perq
    int hp_off;              // head-partial buffer (phase 1 reads the policy evaluations' partials)
    int x_off;               // input set: obs1; obs2 and acts follow as consecutive 256-byte aligned items
    int pack;                // 6 bits per job: [1:0] layer-1 MFMA steps - 4 (input columns + the bias column, in pairs), [2] input is obs2, [3] target copy, [5:4] network; bit 30 (phase 1): jobs 3, 4 take the stored action
};

// One 32-unit block of the K loop: layer 1 (NS MFMA steps) -> relu -> 4 * nrq layer-2 MFMA steps.
template <int NS>
__device__ __forceinline__ void dblock(const float (&w1)[8], const float (&xin)[7], const float4 (&bq)[4], int nrq, floatx16 &x1, floatx16 &acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x1[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[s], xin[s], x1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) x1[r] = relu1(x1[r]);
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        if (rq < nrq) {  // wave-uniform
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[4 * rq + 0], bq[rq].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[4 * rq + 1], bq[rq].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[4 * rq + 2], bq[rq].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[4 * rq + 3], bq[rq].w, acc, 0, 0, 0);
        }
    }
}

// The operand streams of one wave for its blocks [b0, b0 + nb) of 32 hidden-1 units: W2 groups and [W1 ; b1] columns.
// The bias is simply input column D of the layer-1 MFMAs (its input is the constant 1) — no bias loads: a per-lane
// broadcast load costs the fetch path as much as a tile load.  A CU retires about one wave-load per 15 cycles WHATEVER its
// width, so every load is a 16-byte one: the layer-1 parameters are stored in the order these MFMAs read them (two float4
// per lane and block instead of seven dwords: 51 -> 30 loads per wave).  Issued first thing in the kernel from preloaded
// scalars only; loads beyond nb re-read the last block (no branches in front of a load).
struct DOps {
    float4 bq[4][4];
    float w1[4][8];
};
// Issue order = order of need (a wave's loads return in order): the layer-1 columns of a block, then its W2 groups.
// Only the first DPRE blocks are requested before the K loop; block bi + DPRE is requested when block bi's MFMAs are
// issued, so the fetch path (one wave-load per ~15 cycles and CU: ~3.6k cycles for the 8 waves of two workgroups) works
// beside the matrix pipe instead of in front of it.
// Measured at config 2 (us per update): DPRE 1 / 2 / 3 / 4 (= everything up front) 54.1 / 54.5 / 55.2 / 55.3; the input rows
// requested in front of the operand blocks 54.0, the epilogue operands from inside the loop 53.7; DGP 1 / 2 / 4 / 8 / 16
// 52.4 / 52.35 / 52.6 / 53.2 / 53.7; DGMID 1 / 2 / 4 / 6 / 8 52.3 / 52.3 / 52.0 / 52.1 / 52.3.
#ifndef DDRL_DPRE0
#define DDRL_DPRE0 1
#endif
#ifndef DDRL_DPRE1
#define DDRL_DPRE1 1
#endif
constexpr int DPRE = 1;    // operand blocks of k_dfwd requested before its K loop (per phase: DDRL_DPRE0 / DDRL_DPRE1)
#ifndef DDRL_DGP0
#define DDRL_DGP0 2
#endif
#ifndef DDRL_DGMID
#define DDRL_DGMID 4
#endif
constexpr int DGP0 = DDRL_DGP0;    // 8-deep operand groups of k_dg requested before its K loop
constexpr int DGMID = DDRL_DGMID;   // k_dg: the group behind whose MFMAs the epilogue operands are requested
struct DSrc {
    const float *W1, *W2p;
    int Np, b0, nb, n0;
};
__device__ __forceinline__ void dops_load_block(DOps &o, const DSrc &s, int bi, int lane) {
    const int l31 = lane & 31, h = lane >> 5;
    const int u0 = (s.b0 + (bi < s.nb ? bi : (s.nb > 0 ? s.nb - 1 : 0))) * 32;
#pragma unroll
    for (int q = 0; q < 2; ++q) {  // [W1 ; b1] in the layer-1 block layout (gemm_core.h, w1y_index): two float4 per lane and block
        const float4 v = *reinterpret_cast<const float4 *>(s.W1 + ((((long long)(u0 >> 5) * 2 + q) * 2 + h) * 32 + l31) * 4);
        o.w1[bi][4 * q + 0] = v.x; o.w1[bi][4 * q + 1] = v.y; o.w1[bi][4 * q + 2] = v.z; o.w1[bi][4 * q + 3] = v.w;
    }
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) o.bq[bi][rq] = *reinterpret_cast<const float4 *>(s.W2p + ((long long)(u0 / 4 + 2 * rq + h) * s.Np + s.n0 + l31) * 4);
}

// K loop of one wave over MT row tiles that share the wave's W2 / W1 registers.  h1r4 != nullptr (n-tile-0 workgroups of
// a differentiated evaluation): X1 also goes to memory as [row/4][unit][4] through a wave-private LDS tile (registers hold
// row-in-lane / unit-in-register; the image wants 4 rows per float4).
template <int NS, int MT, int PRE, class Mid>
__device__ __forceinline__ void dkloop(DOps &o, const DSrc &src, int K, int lane, const float (&xin)[MT][7], floatx16 (&acc)[MT],
                                       float *__restrict__ h1r4, int Lp1, int m0, float *__restrict__ tr, Mid &&mid) {
    const int l31 = lane & 31, h = lane >> 5, b0 = src.b0, nb = src.nb;
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
        if (bi + PRE < 4) {
            dops_load_block(o, src, bi + PRE, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (bi < nb) {
            const int u0 = (b0 + bi) * 32;
            const int nrq = (K - u0 >= 32) ? 4 : ((K - u0 + 7) >> 3);
#pragma unroll
            for (int tt = 0; tt < MT; ++tt) {
                floatx16 x1;
                dblock<NS>(o.w1[bi], xin[tt], o.bq[bi], nrq, x1, acc[tt]);
                if (h1r4) {  // block-uniform
#pragma unroll
                    for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + l31] = x1[r];
                    wave_lds_sync();
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int rg = h + 2 * p;
                        const float4 v = *reinterpret_cast<const float4 *>(tr + l31 * 36 + 4 * rg);
                        if (u0 + l31 < K) st_img(h1r4 + ((long long)((m0 + 32 * tt) / 4 + rg) * Lp1 + u0 + l31) * 4, v);
                    }
                    wave_lds_sync();
                }
            }
        }
        if (bi == 0) {  // whatever the epilogue needs and nothing in the K loop does: requested behind block 0's MFMAs
            mid();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// MT = 2 (phase 0: 400 tiles of 32 rows would put two workgroups on most CUs, each fetching its own copy of a W2 tile —
// the fetch phase, not the MFMAs, is what a stage waits for): one workgroup = 64 rows x 32 columns, the wave's W2 / W1
// registers serve both row tiles.
template <int PH, int MT>
__global__ void __launch_bounds__(256) k_dfwd(const float *base, int tpj_tm, int K_Np, int B_d0, int pack, int x_off, int main_off, int targ_off, int npi, int perq,
                                              int hp_off, DFArgs a) {  // 12 dwords: what the hardware preloads into SGPRs at wave launch
    static_assert(PH == 0 || MT == 1, "phase 1 computes one policy row per lane");
    __shared__ __attribute__((aligned(16))) float red[4][32][33];
    __shared__ __attribute__((aligned(16))) float tr[4][32 * 36];
    __shared__ float s_wh[DFH][32];
    const int tpj = tpj_tm & 0xffff, tiles_m = (tpj_tm >> 16) & 0xfff, njobs = (unsigned)tpj_tm >> 28;
    const int K = K_Np & 0xfff, Np = (K_Np >> 12) & 0xfff, B = B_d0 & 0xffff, d0 = (B_d0 >> 16) & 0xff, act = B_d0 >> 24;
#ifdef DDRL_STAMPS
    unsigned long long *const st_ = a.st;
#endif
    DRT(PH, 14); DST(PH, 0);
    if (PH == 1 && (int)blockIdx.x == njobs * tpj) {  // only launched when a.do_sample
        ddrl_replay_dev::sample_block(a.rs, a.ring, a.sout, a.sample_batch, nullptr, 1);
        return;
    }
    int t;
    // Phase 1 with pack bit 30: jobs 3, 4 are STORED-action evaluations (Q1(x, a), Q2(x, a)) moved here from phase 0 — they need no
    // policy output, so their K loops run while the three policy-dependent jobs of the same CUs still wait for head partials ->
    // action -> layer-1 input (4.4 k cycles), and phase 0 drops to 240 tiles, one per CU.  Each XCD gets a run of dependent tiles
    // first (dispatched first: one per CU) and then a run of stored ones (the second workgroup of a CU); host: 3 tpj % 8 == 0 == 2 tpj % 8.
    bool stored = false;
    if (PH == 1 && ((pack >> 30) & 1)) {
        const int nd = 3 * tpj, b = blockIdx.x, x = b & 7, slot = b >> 3, qd = nd >> 3, qs = (2 * tpj) >> 3;
        stored = slot >= qd;
        t = stored ? nd + x * qs + (slot - qd) : x * qd + slot;
    } else {   // XCD-aware, panel-major tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s); give each XCD a
        // contiguous run of tiles so that a W2 panel is fetched by one or two L2s instead of all eight (speed only)
        const int nwg = njobs * tpj, b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int ji = (t >= tpj) + (t >= 2 * tpj) + (t >= 3 * tpj) + (t >= 4 * tpj);
    t -= ji * tpj;
    const int nt = t / tiles_m;
    const int mchunk = t - nt * tiles_m, n0 = nt * 32;  // tiles_m counts workgroup-level row chunks of MT row tiles
    const int m0 = mchunk * (32 * MT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int jp = (pack >> (6 * ji)) & 63;
    const int ns = 4 + (jp & 3);
    // this job's network: bits [5:4] of its pack field = 0 policy / 1 q1 / 2 q2, bit [3] = target copy; the three networks sit
    // at fixed distances inside a parameter buffer (policy first, then the two Q networks)
    const int net = (jp >> 4) & 3;
    const int d1 = PH == 1 ? act : ((net == 1 || net == 2) ? act : 0);  // phase 0: a Q network's second input is the stored action (network 3 = V: none)
    const int w1o = ((jp & 8) ? targ_off : main_off) + (net == 0 ? 0 : npi + (net - 1) * perq);
    const int x2_off = x_off + ((B * d0 + 63) & ~63), a_off = x2_off + ((B * d0 + 63) & ~63);  // the input set's buffers are consecutive 256-B aligned slab items
    const float *W1 = base + w1o, *W2p = W1 + ((K + 31) & ~31) * 16;  // the k4-interleaved W2 follows the 16-floats-per-unit layer-1 block array
    const bool first_n = nt == 0;
    // this wave's 32-unit blocks: the waves at the END get the extra (possibly partial) block
    const int nblk = (K + 31) >> 5, bs = nblk >> 2, rem = nblk & 3;
    const int nb = bs + (w >= 4 - rem ? 1 : 0);
    const int b0 = w * bs + (w > 4 - rem ? w - (4 - rem) : 0);

    // ---- every load whose address needs only the preloaded scalars, in the order of need.
    // Phase 1: what its longest chain starts with — the policy-head partials of this tile's rows (-> action -> layer-1 input);
    // lanes of half 0 fetch the mu heads, half 1 the log_std heads (head index clamped: no branch in front of a load).
    const int D = d0 + d1;
    const bool two = PH == 1 && ji == 1 && first_n && a.php1 != nullptr;  // block-uniform: pi_main @ x2 rides here (log-prob only)
    float4 hv[2][4][DNT / 4];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < DNT / 4; ++q) hv[e][c][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PH == 1 && !stored) {   // block-uniform
        const long long HPq = (long long)DFH * B * DNT;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e == 0 || two) {  // block-uniform
                const float *hp = base + hp_off + (e == 0 ? ((a.pev_pack >> (2 * ji)) & 3) : 1) * HPq;
                const float4 *p0 = reinterpret_cast<const float4 *>(hp + ((long long)(h * d1) * B + m0 + l31) * DNT);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < DNT / 4; ++q) hv[e][c][q] = p0[(long long)(c < d1 ? c : 0) * B * (DNT / 4) + q];
                if (d1 > 2) {  // block-uniform
#pragma unroll
                    for (int c = 2; c < 4; ++c)
#pragma unroll
                        for (int q = 0; q < DNT / 4; ++q) hv[e][c][q] = p0[(long long)(c < d1 ? c : 2) * B * (DNT / 4) + q];
                }
            }
        }
    }
    DOps ops;
    const DSrc src{W1, W2p, Np, b0, nb, n0};
    constexpr int PRE = PH == 0 ? DDRL_DPRE0 : DDRL_DPRE1;
    {
        // ---- the observation part of the layer-1 input: lane (row, h) holds input column d_slot(s, h) for step s
        float xin[MT][7];
        {
            const float *in0 = base + ((jp & 4) ? x2_off : x_off), *in1 = base + a_off;
    #pragma unroll
            for (int tt = 0; tt < MT; ++tt) {
                const long long row = m0 + 32 * tt + l31;
    #pragma unroll
                for (int s = 0; s < 7; ++s) {
                    const int d = d_slot(s, h);
                    const bool f0 = d < d0, f1 = (PH == 0 || stored) && !f0 && d < D;
                    const float *p = f1 ? in1 + row * d1 + (d - d0) : in0 + row * d0 + (f0 ? d : 0);
                    // (no select on the loaded value: the compiler would sink the load into a branch.)  Column D: the bias row
                    const float v = *p;   // (masking by multiplication instead of the select: +1.5 us per update)
                    xin[tt][s] = (f0 || f1) ? v : (d == D ? 1.0f : 0.f);
                }
            }
        }
        {   // behind the input rows: a wave's loads return in order and the first MFMA needs those first
#pragma unroll
            for (int bi = 0; bi < PRE; ++bi) dops_load_block(ops, src, bi, lane);
        }
        DST(PH, 1);
        const DFJob &jb = a.job[ji];
        ktouch(kline<sizeof(DFJob)>(&jb) | kline<offsetof(DFArgs, do_sample)>(&a));
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float whv = 0.f;
        auto epilogue_operands = [&]() {
            b4 = *reinterpret_cast<const float4 *>(jb.b2 + n0 + 4 * (tid >> 5));
            const int c = tid >> 5, col = n0 + (tid & 31);
            const bool ok = c < jb.nh && col < a.h2;
            const float *p = c < jb.hsplit ? jb.wh0 : jb.wh1;
            const int cc = c < jb.hsplit ? c : c - jb.hsplit;
            whv = p[ok ? (long long)col * jb.hstride + cc : 0] * (ok ? 1.0f : 0.f);
        };
        if (PH == 1 && !stored) {  // the sampled action of this tile's rows, in every lane of every wave (no LDS, no barrier)
            float hs[2][4];
    #pragma unroll
            for (int e = 0; e < 2; ++e)
    #pragma unroll
                for (int c = 0; c < 4; ++c) {  // n-tile order; unused slots are 0
                    float s = 0.f;
    #pragma unroll
                    for (int q = 0; q < DNT / 4; ++q) { s += hv[e][c][q].x; s += hv[e][c][q].y; s += hv[e][c][q].z; s += hv[e][c][q].w; }
                    hs[e][c] = s;
                }
            float mu[4], ls[4], ev[4];
    #pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float o = __shfl_xor(hs[0][c], 32);
                const int cc = c < d1 ? c : 0;
                mu[c] = (h ? o : hs[0][c]) + jb.pbmu[cc];
                ls[c] = (h ? hs[0][c] : o) + jb.pbls[cc];
                ev[c] = jb.peps[(long long)(m0 + l31) * d1 + cc];
            }
            const PolRow o = policy_row(mu, ls, ev, d1, a.scale);
    #pragma unroll
            for (int s = 0; s < 7; ++s) {
                const int d = d_slot(s, h), c = d - d0;
                if (d >= d0 && d < D) xin[0][s] = c == 0 ? o.act[0] : (c == 1 ? o.act[1] : (c == 2 ? o.act[2] : o.act[3]));
            }
            if (first_n && w == 0 && lane < 32) {
                const int row = m0 + l31;
                if (jb.side == 1) {
                    a.logp0[row] = o.logp;
    #pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < d1) {
                            a.act0[row * d1 + c] = o.act[c];
                            *reinterpret_cast<float4 *>(a.save0 + ((long long)row * d1 + c) * 4) = make_float4(o.a[c], o.std[c], o.t[c], ev[c]);
                        }
                } else if (jb.side == 2) {
    #pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < d1) a.act2[row * d1 + c] = o.act[c];
                }
            }
            if (two && w == 1) {  // wave-uniform
    #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float o2 = __shfl_xor(hs[1][c], 32);
                    const int cc = c < d1 ? c : 0;
                    mu[c] = (h ? o2 : hs[1][c]) + a.pbmu1[cc];
                    ls[c] = (h ? hs[1][c] : o2) + a.pbls1[cc];
                    ev[c] = a.peps1[(long long)(m0 + l31) * d1 + cc];
                }
                const PolRow o1 = policy_row(mu, ls, ev, d1, a.scale);
                if (lane < 32) a.logp1[m0 + l31] = o1.logp;
            }
        }
        // ---- K loop
        DST(PH, 2);
        floatx16 acc[MT];
    #pragma unroll
        for (int tt = 0; tt < MT; ++tt)
    #pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
        float *h1r4 = (jb.H1r4 && first_n) ? jb.H1r4 : nullptr;
        {
            if (ns == 4) dkloop<4, MT, PRE>(ops, src, K, lane, xin, acc, h1r4, a.Lp1, m0, tr[w], epilogue_operands);
            else if (ns == 5) dkloop<5, MT, PRE>(ops, src, K, lane, xin, acc, h1r4, a.Lp1, m0, tr[w], epilogue_operands);
            else if (ns == 6) dkloop<6, MT, PRE>(ops, src, K, lane, xin, acc, h1r4, a.Lp1, m0, tr[w], epilogue_operands);
            else dkloop<7, MT, PRE>(ops, src, K, lane, xin, acc, h1r4, a.Lp1, m0, tr[w], epilogue_operands);
        }

        DST(PH, 3);
        // ---- side outputs of the n-tile-0 workgroups (off the critical path: nothing in this launch reads them)
        s_wh[tid >> 5][tid & 31] = whv;
        if (first_n && (jb.aug || jb.xr4)) {  // augmented input rows of the layer-1 wgrads (their ones column is set once at create)
            for (int idx = tid; idx < 32 * MT * D; idx += 256) {
                const int r = idx / D, d = idx - r * D;
                const long long row = m0 + r;
                const float v = d < d0 ? (base + ((jp & 4) ? x2_off : x_off))[row * d0 + d] : (base + a_off)[row * d1 + (d - d0)];
                if (jb.aug) jb.aug[row * jb.aug_ld + d] = v;
                if (jb.xr4) jb.xr4[((row >> 2) * 32 + d) * 4 + (row & 3)] = v;
            }
        }
        if (PH == 0 && a.noise_on && ji == 0 && first_n) {
            // eps_x, eps_x2, eps_t of this tile's rows; element index as in one flat [3][B*act] fill
            const unsigned long long nbase = a.opt->noise_ctr;
            const int per_row = 3 * act;
            for (int e = tid; e < 32 * MT * per_row; e += 256) {
                const int rr = e / per_row, q = e - rr * per_row;
                const int wch = q / act, c = q - wch * act;
                const int k = (m0 + rr) * act + c;
                if (m0 + rr < a.Bv) (wch == 0 ? a.e0 : (wch == 1 ? a.e1 : a.e2))[k] = normal_at(a.noise_seed, nbase + (unsigned long long)wch * a.n_each + k);
            }
        }

        // ---- split-K combine, bias + relu, H2 in the layouts its consumers read, head partials
        const int r = tid & 31, cg = tid >> 5;
    #pragma unroll
        for (int tt = 0; tt < MT; ++tt) {
            const int mb = m0 + 32 * tt;
            if (tt > 0) __syncthreads();
    #pragma unroll
            for (int q = 0; q < 16; ++q) red[w][(q & 3) + 8 * (q >> 2) + 4 * h][l31] = acc[tt][q];
            __syncthreads();
            if (tt == 0) DST(PH, 4);
            float v[4];
    #pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * cg + e;
                const float s = ((red[0][r][c] + red[1][r][c]) + red[2][r][c]) + red[3][r][c];
                v[e] = fmaxf(s + (e == 0 ? b4.x : (e == 1 ? b4.y : (e == 2 ? b4.z : b4.w))), 0.f);
            }
            if (jb.H2c4) st_img(jb.H2c4 + ((long long)(n0 / 4 + cg) * B + mb + r) * 4, make_float4(v[0], v[1], v[2], v[3]));
            __syncthreads();
    #pragma unroll
            for (int e = 0; e < 4; ++e) red[0][r][4 * cg + e] = v[e];
            __syncthreads();
            if (jb.H2r4) {  // (column hidden2 of the image is the ones column: never written here)
                const int c = tid & 31, rg = tid >> 5;
                if (n0 + c < a.h2)
                    st_img(jb.H2r4 + ((long long)(mb / 4 + rg) * a.Lp2 + n0 + c) * 4,
                           make_float4(red[0][4 * rg][c], red[0][4 * rg + 1][c], red[0][4 * rg + 2][c], red[0][4 * rg + 3][c]));
            }
            {
                const int c = tid >> 5;
                if (c < jb.nh) {
                    float s = 0.f;
    #pragma unroll
                    for (int col = 0; col < 32; ++col) s = fmaf(red[0][r][col], s_wh[c][col], s);
                    jb.hp[((long long)c * B + mb + r) * DNT + nt] = s;
                }
            }
        }
    }
    DST(PH, 5); DRT(PH, 15);
}

// Row tiles per workgroup of a phase-0 launch.  Measured at the config-2 shape (400 tiles of 32 rows): MT = 2 halves the
// W2 bytes a CU fetches, but leaves ONE wave per SIMD whose layer-1 -> relu -> layer-2 chain of dependent MFMAs has nobody
// to fill its issue bubbles: 15.9 us per launch against 12.0 us with two 32-row workgroups per CU.  DDRL_FWD_MT=2 selects it.
static int dfwd_mt(int B, int tiles_n) {
    static const int want = getenv("DDRL_FWD_MT") ? atoi(getenv("DDRL_FWD_MT")) : 1;
    return (want == 2 && B % 64 == 0 && 5 * (B / 32) * tiles_n > 256) ? 2 : 1;
}

template <int PH>
static void launch_dfwd(const DFHead &d, const DFArgs &F_, hipStream_t s) {
#ifdef DDRL_STAMPS
    DFArgs F = F_;
    F.st = g_st_host;
#else
    const DFArgs &F = F_;
#endif
    const int mt = PH == 0 ? dfwd_mt(d.B, F.tiles_n) : 1;
    const int tiles_m = d.B / (32 * mt), tpj = tiles_m * F.tiles_n;
    const int grid = F.njobs * tpj + ((PH == 1 && F.do_sample) ? 1 : 0);
    const int a1 = tpj | (tiles_m << 16) | (F.njobs << 28), a2 = d.K | (d.Np << 12), a3 = d.B | (d.d0 << 16) | (F.act << 24);
    if (PH == 0 && mt == 2)
        k_dfwd<0, 2><<<grid, 256, 0, s>>>(d.base, a1, a2, a3, d.pack, d.x_off, d.main_off, d.targ_off, d.npi, d.perq, d.hp_off, F);
    else
        k_dfwd<PH, 1><<<grid, 256, 0, s>>>(d.base, a1, a2, a3, d.pack, d.x_off, d.main_off, d.targ_off, d.npi, d.perq, d.hp_off, F);
}

// ==========================================================================================
// k_actor_fwd: the policy forward of the rollout (thousands of rows, one network, heads only).
// One workgroup = one 32-row tile x up to ANT consecutive column tiles of layer 2.  Each wave computes layer 1 for ITS
// quarter of the hidden-1 units ONCE and keeps it in registers (k_dfwd regenerates it per column tile), then walks the
// column tiles: every W2 group of tile t + 1 is requested between the MFMAs of the same group of tile t, so the only
// latency a wave ever waits for is that of its first loads.  4096 rows x 10 column tiles = 128 x 2 workgroups of 5 tiles:
// one per CU, 16.2 us per launch.  (Measured alternatives: 4 workgroups of 3, 3, 2, 2 tiles per row tile, two per CU, so that
// a wave busy issuing loads — ~60 cycles per 16-byte-per-lane load — leaves the matrix pipe to the other workgroup's wave:
// 17.9 us, layer 1 is generated 4x instead of 2x and the register budget halves; skewing the four waves by one MFMA each
// after every barrier: no change — the issue cost is per wave, not a queue at the CU's fetch path.)
// Same summation orders as k_dfwd (block order inside a wave, waves 0..3 in the combine, columns 0..31 in the head dot).
// ==========================================================================================
constexpr int ANT = 5;
// (VerTile / VerState / VerSplit / ver_split: policy_row.h — the rollout's env-step launch writes the same tables)
struct ActFwdArgs {
    const float *W1, *W2p, *b2, *wmu, *wls, *obs;
    float *hp;
    int rows, K, Np, h2, d0, act, tiles_n, ngroups;
    // VER: the weight pointers above are those of version slot 0; slot s lives vstride floats further per slot
    const VerTile *vtiles;
    const int *perm;
    const VerState *vs;
    long long vstride;
    int vt_max;              // VER: records the workgroup table is allocated for (= the launch's grid: the worst case; the plan says how many run)
#ifdef DDRL_STAMPS
    unsigned long long *st;  // dev harness (tools/actor_bench.hip): [workgroup][wave][32] cycle stamps
#endif
};
#ifdef DDRL_STAMPS
#define AST(i) do { if (a.st && lane == 0) a.st[((long long)blockIdx.x * 4 + w) * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AST(i) do { } while (0)
#endif
// OCC = workgroups per CU the register budget is cut for: 1 -> 268 registers, one wave per SIMD: the fastest single round (<= 256
// workgroups: 4096 rows, 16.4 us); 2 -> 256 registers with 64 spilled to scratch, two workgroups per CU: +1.8 us on a single round, but
// 8192 rows (config 4's rollout ranks: 512 workgroups) take one round instead of two (rollout-only 195 -> 208 M env-steps/s, 16 384
// rows 227 -> 253 M).
template <int NS, int OCC, bool VER = false>
__global__ void __launch_bounds__(256, OCC) k_actor_fwd(ActFwdArgs a) {
    // rows of 36 floats: 16-byte aligned (b128 LDS reads) and conflict-free for 16 lanes reading 4 words each
    __shared__ __attribute__((aligned(16))) float red[2][4][32][36];
    __shared__ __attribute__((aligned(16))) float red2[2][32][36];
    __shared__ __attribute__((aligned(16))) float s_wh[ANT][DFH][32];
    __shared__ __attribute__((aligned(16))) float s_b2[ANT][32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // column tiles are dealt to `ngroups` workgroups per row tile as evenly as they go (VER: the planning launch's table, one record per workgroup)
    int m0, ntiles, nt0;
    int vcount = 32, vrow = 0;
    if (!VER) {
        const int ngroups = a.ngroups;
        const int rt = blockIdx.x / ngroups, grp = blockIdx.x - rt * ngroups;
        const int gbase = a.tiles_n / ngroups, gextra = a.tiles_n % ngroups;
        ntiles = gbase + (grp < gextra ? 1 : 0);
        nt0 = grp * gbase + (grp < gextra ? grp : gextra);
        m0 = rt * 32;
    } else {
        // two independent loads (the launch's workgroup count, this workgroup's record), then this lane's env out of the record's row list;
        // entries beyond the tile's count hold whatever an earlier plan left — valid env numbers or zero, never read as data
        const int n_wgs = a.vs->n_wgs;
        const VerTile vt = a.vtiles[(int)blockIdx.x < a.vt_max ? blockIdx.x : 0];
        if ((int)blockIdx.x >= n_wgs) return;   // the launch covers the worst case; block-uniform
        vrow = a.perm[vt.base + l31];
        const long long off = (long long)vt.slot * a.vstride;
        a.W1 += off; a.W2p += off; a.b2 += off; a.wmu += off; a.wls += off;
        m0 = vt.base; vcount = vt.count; nt0 = vt.cols & 255; ntiles = vt.cols >> 8;
        vrow = __shfl(vrow, (lane & 32) + (l31 < vcount ? l31 : vcount - 1));   // rows beyond the count repeat the tile's last env
    }
    const int K = a.K, Np = a.Np, d0 = a.d0;
    const int nblk = (K + 31) >> 5, bs = nblk >> 2, rem = nblk & 3;
    const int nb = bs + (w >= 4 - rem ? 1 : 0);
    const int b0 = w * bs + (w > 4 - rem ? w - (4 - rem) : 0);
    AST(0);
    // ---- loads in the order of need: the input rows, the layer-1 columns of this wave's blocks, the W2 groups of tile 0
    float xin[7];
    {
        long long row = m0 + l31 < a.rows ? m0 + l31 : a.rows - 1;
        if (VER) row = vrow;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int d = d_slot(s, h);
            const float v = a.obs[row * d0 + (d < d0 ? d : 0)];
            xin[s] = d < d0 ? v : (d == d0 ? 1.0f : 0.f);  // column d0: the bias row
        }
    }
    float w1[4][8];
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
        const int blk = b0 + (bi < nb ? bi : (nb > 0 ? nb - 1 : 0));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(a.W1 + ((((long long)blk * 2 + q) * 2 + h) * 32 + l31) * 4);
            w1[bi][4 * q + 0] = v.x; w1[bi][4 * q + 1] = v.y; w1[bi][4 * q + 2] = v.z; w1[bi][4 * q + 3] = v.w;
        }
    }
    // OCC = 1: the W2 groups double-buffered by tile parity.  OCC = 2 (two workgroups per CU, 256 registers): ONE set — the group of tile
    // t + 1 is requested into the registers of the same group of tile t right behind the MFMAs that consumed it (the same one-tile prefetch
    // distance with 64 registers less: the double-buffered form spilled 64 registers to scratch at this budget)
    constexpr int NBUF = OCC == 2 ? 1 : 2;
    float4 bq[NBUF][4][4];
    const int lane_off = (h * Np + l31) * 4;
    auto fetch_group = [&](int buf, int t, int bi, int rq) {  // one W2 group (8 hidden-1 units x 32 columns) of column tile t
        // wave-uniform base (scalar registers) + one per-lane offset shared by every group: no vector address arithmetic per load
        const int nt = nt0 + t < a.tiles_n ? nt0 + t : a.tiles_n - 1;
        const int u0 = (b0 + (bi < nb ? bi : (nb > 0 ? nb - 1 : 0))) * 32;
        const float *gp = a.W2p + ((long long)(u0 / 4 + 2 * rq) * Np + nt * 32) * 4;
        bq[buf][bi][rq] = *reinterpret_cast<const float4 *>(gp + lane_off);
    };
    // head kernels and biases of this workgroup's column tiles: requested now, staged into LDS behind tile 0's MFMAs
    float whv[(ANT * DFH * 32 + 255) / 256], b2v = 0.f;
#pragma unroll
    for (int i = 0; i < (ANT * DFH * 32 + 255) / 256; ++i) {
        const int e = tid + 256 * i;
        const int t = e / (DFH * 32), c = (e / 32) % DFH, col = (nt0 + t) * 32 + (e & 31);
        const bool ok = e < ANT * DFH * 32 && c < 2 * a.act && col < a.h2 && t < ntiles;
        const float *p = c < a.act ? a.wmu : a.wls;
        const float v = p[ok ? (long long)col * a.act + (c < a.act ? c : c - a.act) : 0];
        whv[i] = ok ? v : 0.f;
    }
    {
        const int t = tid >> 5, col = (nt0 + t) * 32 + (tid & 31);
        const bool ok = tid < ANT * 32 && col < Np && t < ntiles;
        const float v = a.b2[ok ? col : 0];  // (pads of b2 are zero)
        b2v = ok ? v : 0.f;
    }
    AST(1);
    // ---- layer 1 of this wave's blocks, once
    floatx16 x1[4];
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x1[bi][r] = 0.f;
        if (bi < nb) {  // wave-uniform
#pragma unroll
            for (int s = 0; s < NS; ++s) x1[bi] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[bi][s], xin[s], x1[bi], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) x1[bi][r] = relu1(x1[bi][r]);
        }
    }
    // tile 0's W2 groups: requested behind the layer-1 MFMAs (which wait for the input rows and layer-1 columns only)
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) fetch_group(0, 0, bi, rq);
    AST(2);
    auto stage_heads = [&]() {
#pragma unroll
        for (int i = 0; i < (ANT * DFH * 32 + 255) / 256; ++i) {
            const int e = tid + 256 * i;
            if (e < ANT * DFH * 32) (&s_wh[0][0][0])[e] = whv[i];
        }
        if (tid < ANT * 32) (&s_b2[0][0])[tid] = b2v;
    };
    // ---- the column tiles
    const int r = tid & 31, cg = tid >> 5;
#pragma unroll
    for (int t = 0; t < ANT; ++t) {
        if (t < ntiles) {  // block-uniform
            floatx16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
                if (bi < nb) {
                    const int u0 = (b0 + bi) * 32;
                    const int nrq = (K - u0 >= 32) ? 4 : ((K - u0 + 7) >> 3);
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        if (rq < nrq) {
                            const float4 b = bq[t & (NBUF - 1)][bi][rq];
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[bi][4 * rq + 0], b.x, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[bi][4 * rq + 1], b.y, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[bi][4 * rq + 2], b.z, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[bi][4 * rq + 3], b.w, acc, 0, 0, 0);
                        }
                        if (t + 1 < ANT) {  // the same group of the next column tile, requested between this group's MFMAs and the next's
                            __builtin_amdgcn_sched_barrier(0);
                            fetch_group((t + 1) & (NBUF - 1), t + 1, bi, rq);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else if (t + 1 < ANT) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) fetch_group((t + 1) & (NBUF - 1), t + 1, bi, rq);
                }
            }
            AST(3 + 5 * t);
            if (t == 0) stage_heads();  // requested behind tile 0's MFMAs, stored before the first barrier
            // split-K combine, bias + relu, head partials (LDS tiles double-buffered by tile parity: two barriers per tile)
#pragma unroll
            for (int q = 0; q < 16; ++q) red[t & 1][w][(q & 3) + 8 * (q >> 2) + 4 * h][l31] = acc[q];
            AST(4 + 5 * t);
            __syncthreads();
            AST(5 + 5 * t);
            {
                const float4 p0 = *reinterpret_cast<const float4 *>(&red[t & 1][0][r][4 * cg]), p1 = *reinterpret_cast<const float4 *>(&red[t & 1][1][r][4 * cg]);
                const float4 p2 = *reinterpret_cast<const float4 *>(&red[t & 1][2][r][4 * cg]), p3 = *reinterpret_cast<const float4 *>(&red[t & 1][3][r][4 * cg]);
                const float4 bb = *reinterpret_cast<const float4 *>(&s_b2[t][4 * cg]);
                float4 v;
                v.x = fmaxf((((p0.x + p1.x) + p2.x) + p3.x) + bb.x, 0.f);
                v.y = fmaxf((((p0.y + p1.y) + p2.y) + p3.y) + bb.y, 0.f);
                v.z = fmaxf((((p0.z + p1.z) + p2.z) + p3.z) + bb.z, 0.f);
                v.w = fmaxf((((p0.w + p1.w) + p2.w) + p3.w) + bb.w, 0.f);
                *reinterpret_cast<float4 *>(&red2[t & 1][r][4 * cg]) = v;
            }
            AST(6 + 5 * t);
            __syncthreads();
            if (cg < 2 * a.act) {
                float sum = 0.f;
#pragma unroll
                for (int c4 = 0; c4 < 8; ++c4) {
                    const float4 x = *reinterpret_cast<const float4 *>(&red2[t & 1][r][4 * c4]), y = *reinterpret_cast<const float4 *>(&s_wh[t][cg][4 * c4]);
                    sum = fmaf(x.x, y.x, sum); sum = fmaf(x.y, y.y, sum); sum = fmaf(x.z, y.z, sum); sum = fmaf(x.w, y.w, sum);
                }
                if (VER) {
                    if (r < vcount) a.hp[((long long)cg * a.rows + a.perm[m0 + r]) * DNT + nt0 + t] = sum;
                } else if (m0 + r < a.rows) a.hp[((long long)cg * a.rows + m0 + r) * DNT + nt0 + t] = sum;
            }
            AST(7 + 5 * t);
        }
    }
}

// ==========================================================================================
// k_dg: the backward launches.  One workgroup = one 32x32 output tile of one job; the four waves split the contraction
// in 8-deep groups; both operands are x4 images read straight into the MFMA lane layout.
// ==========================================================================================
enum { DG_DGRAD_Q = 0, DG_DGRAD = 1, DG_WGRAD_J4 = 2, DG_WGRAD_RM = 3, DG_ROWS_C = 4, DG_LOSS = 5, DG_WGRAD_W1Y = 6 };
// The first 64 bytes are the "hot line": every scalar the first operand loads of a tile need.  k_dg fetches it with ONE
// s_load_dwordx16 (and starts one load per remaining line of the record and of the launch header at the same time), waits ONCE,
// and issues its operand loads; the compiler's own field loads — issued where the fields are used, each behind its own
// s_waitcnt — made the prologue a chain of five to nine cold scalar-cache misses (2.7-4.1 k cycles from wave start to the first
// operand load requested, tools/upd_bench.hip -DDDRL_STAMPS).
struct alignas(64) DGJob {
    const float *A, *B;      // x4 images: [K/4][lda][4], [K/4][ldb][4]
    const float *gp;         // the generated operand's vector along the contraction: DGRAD_Q: gw (W3[k]);  bgen: gdq (dq[r]) — set by dg_add
    int lda, ldb;
    int K, type;             // contraction length; DG_*
    int tiles_m, tile_start;
    int slot;                // DGRAD_Q: 0 q1(x,a)  1 q2(x,a)  2 q1(x,pi)
    int bgen;
    int N, ntiles;           // output columns; tiles of the job
    // ---- end of the hot line
    int M;                   // output rows
    // generated operands
    const float *gw;         // DGRAD_Q: W3[k] (A = dq[row] * W3[k] * (H2 > 0));  WGRAD with bgen: W3[n] (B = dq[r] * W3[n] * (H2 > 0))
    const float *gdq;        // DGRAD_Q: nullptr -> the constant gconst;  bgen: dq[r] (flat [B])
    float *gw_snap;          // DGRAD_Q (nullable): tile (0, 0) copies W3 here for the NEXT launch's generated wgrad operand — that launch
                             // also steps W3 (Adam in the head-wgrad epilogue), and a gradient must see the pre-update kernel whichever
                             // workgroup runs first
    float gconst;
    // dgrad epilogue
    const float *mask;       // H1r4 [M/4][ldmask][4]
    int ldmask;
    float *C;                // dgrad: dZ1r4 [M/4][ldc][4] (nullable);  WGRAD_RM: gradient, row-major (row stride ldc)
    int ldc;
    const float *wa;         // DGRAD_Q slot 2: the main q1 layer-1 block array (action rows = input columns wa_d0 ..)
    int wa_d0;
    float *da_part;          // ... dQ/da partials [tiles_n][B][4]
    int nact;
    // wgrad epilogue (optimizer): offsets into the flat parameter-shaped buffers
    long long adam_off;      // J4: the W2 image;  RM: element (0, 0)
    long long bias_off;      // J4: b2
    int bias_row;            // J4: contraction-side row that holds the bias gradient (= hidden1: the ones row of H1r4)
    float *shadow;           // J4: the dgrad image [N/4][ld_sh][4] of the updated kernel (nullable)
    int ld_sh;
    // DG_ROWS_C (M = batch, N = hidden2): policy-head backward + dZ2 of the policy trunk as tiles
    const float *h2c4;       // relu mask [N/4][B][4]
    const float *dap;        // dQ/da partials [nparts][B][4]
    int nparts;
    const float *save0, *wmu, *wls;
    float *dz_c4, *dz_r4, *dhead_r4;
    int ld_r4;
    float alpha, scale;
    // DG_LOSS
    const float *loss_part;
    float *losses;
    int nl;
};
constexpr int MAX_DG_JOBS = 16;
struct DGJobs {
    int njobs, total_tiles;
    int tile_start[MAX_DG_JOBS];
    AdamCtx ad;
    // DGRAD_Q prologue (actor_learner.py:58-69)
    const float *hp;   // head partials [NEVAL][DFH][B][DNT]
    const float *b3q1, *b3q2, *b3q1t, *b3q2t;   // SAC-v: b3q1t = main V's b3, b3q2t = target V's b3
    const float *rew, *done, *logp0, *logp1;
    float *q1o, *q2o, *dq, *loss_part;
    float *vo, *vto;   // SAC-v outputs v(x), v_targ(x2)
#ifdef DDRL_STAMPS
    unsigned long long *st;
#endif
    int sacv;          // 0: SAC1 losses (actor_learner.py:58-69);  1: SAC-v (example/model.py:38-50)
    int q_ev0, q_nev;  // the head-partial slots the prologue sums: evaluations q_ev0 .. q_ev0 + q_nev - 1
    float alpha, gamma;
    int B, Bv;         // rows of every image (a multiple of 32), valid rows (the batch: means and loss terms run over these)
    DGJob job[MAX_DG_JOBS];
};


template <int GMAX>
__global__ void __launch_bounds__(256) k_dg(int total_tiles, int tsA, int tsB, int tsC, int tsD, int tsE, int tsF, int tsG, int tsH, int kid, DGJobs jobs) {
    __shared__ __attribute__((aligned(16))) float red[4][32][33];
    __shared__ float s_q[8][32];
    __shared__ float s_g[32];
    __shared__ float s_wa[4][32];
    __shared__ __attribute__((aligned(16))) float s_gw[512];
    __shared__ __attribute__((aligned(16))) float s_w8[512 * 8];   // agen: the policy's head kernels, [k][Wmu 0..3 | Wls 0..3]
    int t, ji;
#ifdef DDRL_STAMPS
    unsigned long long *const st_ = jobs.st;
#endif
    DRT(kid, 14); DST(kid, 0);
    {   // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8, private L2s); each XCD
        // takes a contiguous run of the launch's panel-major tile list, so that an operand panel is fetched by one or two L2s
        // instead of all eight (speed only).  Measured alternatives (us per update, tools/patches/): every job spread over all
        // XCDs with its own runs 52.3, panel-cyclic ownership (XCD x owns column panel x of every job in every launch) 52.0,
        // this 50.2: what an XCD saves in re-fetched operand panels outweighs both a job's tail and inter-launch L2 reuse
        // (which does not happen anyway: an XCD's footprint per update is the size of its L2).
        const int nwg = total_tiles, b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    static_assert(MAX_DG_JOBS == 16, "k_dg takes tile_start[1..15] as eight scalar arguments, two 16-bit starts each (0xffff: no such job)");
    {
        auto ge2 = [&](int p) { return (t >= (p & 0xffff)) + (t >= (int)((unsigned)p >> 16)); };
        ji = (t >= (int)((unsigned)tsA >> 16)) + ge2(tsB) + ge2(tsC) + ge2(tsD) + ge2(tsE) + ge2(tsF) + ge2(tsG) + ge2(tsH);
    }
    const DGJob &jb = jobs.job[ji];
    // ONE round trip to the kernarg segment: the hot line of the job record into SGPRs + one touch per other line of the record and
    // of the launch header (so that every later field load hits the scalar cache), one wait
    typedef int v16i __attribute__((ext_vector_type(16)));
    static_assert(sizeof(DGJob) == 5 * 64 && offsetof(DGJob, M) == 64, "DGJob: hot line = the first 64 bytes, 5 lines in all");
    static_assert(offsetof(DGJobs, job) == 6 * 64, "DGJobs: header of 6 lines, records 64-byte aligned");
    v16i hot;
    int kl;
    {
        // (addresses from the kernarg segment pointer: taking &jobs would make the compiler copy the by-value struct to scratch.)
        // Kernarg layout: ten ints, then DGJobs at the next multiple of its 64-byte alignment.
        constexpr int JOBS_OFF = 64;
        static_assert(alignof(DGJobs) == 64 && 10 * sizeof(int) <= JOBS_OFF, "kernarg offset of k_dg's DGJobs argument");
        const char *ka = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
        const int jboff = __builtin_amdgcn_readfirstlane(JOBS_OFF + (int)offsetof(DGJobs, job) + ji * (int)sizeof(DGJob));  // (wave-uniform: an SGPR address)
        const void *hdp = ka + JOBS_OFF, *jbp = ka + jboff;
        int k1, k2, k3, k4, k5, k6, k7, k8, k9, k10;
        asm volatile("s_load_dwordx16 %0, %11, 0x0\n\t"
                     "s_load_dword %1, %11, 0x40\n\t"
                     "s_load_dword %2, %11, 0x80\n\t"
                     "s_load_dword %3, %11, 0xc0\n\t"
                     "s_load_dword %4, %11, 0x100\n\t"
                     "s_load_dword %5, %12, 0x0\n\t"
                     "s_load_dword %6, %12, 0x40\n\t"
                     "s_load_dword %7, %12, 0x80\n\t"
                     "s_load_dword %8, %12, 0xc0\n\t"
                     "s_load_dword %9, %12, 0x100\n\t"
                     "s_load_dword %10, %12, 0x140\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(hot), "=&s"(k1), "=&s"(k2), "=&s"(k3), "=&s"(k4), "=&s"(k5), "=&s"(k6), "=&s"(k7), "=&s"(k8), "=&s"(k9), "=&s"(k10)
                     : "s"(jbp), "s"(hdp)
                     : "memory");
        kl = k1 | k2 | k3 | k4 | k5 | k6 | k7 | k8 | k9 | k10;
    }
    // (the pointers are rebuilt from SGPR halves: the cast goes through the GLOBAL address space, or every load through them
    // would be a flat_load — counted out of order, the compiler then waits for vmcnt(0) in front of every MFMA group)
    typedef const float __attribute__((address_space(1))) *gfp;
    auto mkptr = [](int lo, int hi) { return (const float *)(gfp)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo); };
    const float *const hot_A = mkptr(hot[0], hot[1]), *const hot_B = mkptr(hot[2], hot[3]), *const hot_gp = mkptr(hot[4], hot[5]);
    const int hot_lda = hot[6], hot_ldb = hot[7], hot_K = hot[8], hot_type = hot[9], hot_tiles_m = hot[10], hot_tile_start = hot[11];
    const int hot_slot = hot[12], hot_bgen = hot[13], hot_N = hot[14];
    t -= hot_tile_start;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int type = hot_type;
    const int Bn = jobs.B, Bv = jobs.Bv;

    if (type == DG_LOSS) {
        // reduce_mean over the batch: per-row terms summed in a fixed order (lane-strided partial sums, then the xor tree)
        if (w != 0) return;
        float s3[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b0 = 0; b0 < Bv; b0 += 64) {
            const int b = b0 + lane;
            const bool ok = b < Bv;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v = jb.loss_part[(ok ? b : 0) * jb.nl + (c < jb.nl ? c : 0)];
                s3[c] += (ok && c < jb.nl) ? v : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float tot = wave_sum(s3[c]);
            const float mean = tot / (float)Bv;
            if (lane == 0 && c < jb.nl) jb.losses[c] = c == 0 ? mean : 0.5f * mean;
        }
        if (lane == 0 && jobs.ad.on && jb.nparts == -1) {   // (nparts = -1: this loss tile also keeps the optimizer's books, once per update)
            OptState nx = *jobs.ad.opt;
            nx.b1p_pi *= jobs.ad.b1; nx.b2p_pi *= jobs.ad.b2; nx.b1p_q *= jobs.ad.b1; nx.b2p_q *= jobs.ad.b2;
            nx.t_pi += 1; nx.t_q += 1;
            nx.noise_ctr += jobs.ad.noise_adv;
            *jobs.ad.opt_next = nx;
        }
        return;
    }
    const int tiles_m = hot_tiles_m;
    const int mt = t % tiles_m, nt = t / tiles_m;
    const int m0 = mt * 32, n0 = nt * 32;
    const int r = tid & 31, cg = tid >> 5;

    if (type == DG_ROWS_C) {
        // Policy-head backward for the 32 rows of this tile (d pi_loss / d (mu_raw, log_std_raw), through tanh-squash,
        // reparameterisation and clip-pass-gradient) — every thread for its row, eight-fold redundant, no exchange —
        // then dZ2[r][c] = (H2[r][c] > 0) * sum_a (dmu[a] Wmu[c][a] + dls[a] Wls[c][a]) for four consecutive c.
        const int act = jb.nact, row = m0 + r;
        float4 dp[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) dp[q] = *reinterpret_cast<const float4 *>(jb.dap + ((long long)(q < jb.nparts ? q : 0) * Bn + row) * 4);
        float4 sv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) sv[c] = *reinterpret_cast<const float4 *>(jb.save0 + ((long long)row * act + (c < act ? c : 0)) * 4);
        const float4 hm = *reinterpret_cast<const float4 *>(jb.h2c4 + ((long long)(n0 / 4 + cg) * Bn + row) * 4);
        float wm[4][4], wl[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = n0 + 4 * cg + e;
            const bool okc = c < jb.N;
#pragma unroll
            for (int a2 = 0; a2 < 4; ++a2) {
                const long long o = (long long)(okc ? c : 0) * act + (a2 < act ? a2 : 0);
                // (masked by multiplication: behind a select the compiler sinks each of the 32 loads into a branch of its own and
                // waits for it there — sixteen dependent round trips: 14.3 k cycles per tile against 9.1 k)
                const float vm = jb.wmu[o], vl = jb.wls[o], keep = (okc && a2 < act) ? 1.0f : 0.f;
                wm[e][a2] = vm * keep;
                wl[e][a2] = vl * keep;
            }
        }
        float ga[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (q < jb.nparts) { ga[0] += dp[q].x; ga[1] += dp[q].y; ga[2] += dp[q].z; ga[3] += dp[q].w; }
        float dmu[4], dls[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float av = sv[c].x, std = sv[c].y, tt = sv[c].z, e = sv[c].w;
            const float glp = row < Bv ? jb.alpha / (float)Bv : 0.f;  // d pi_loss / d logp_pi (padding rows carry no loss)
            const float om = 1.0f - av * av;
            const float cl = fminf(fmaxf(om, 0.f), 1.f);
            const float du = (ga[c] * jb.scale) * om + glp * ((2.0f * av * om) / (cl + 1e-6f));
            const float sd = std + STD_EPS;
            const float z = (e * std) / sd;
            const float dzdl = ((e * std) * STD_EPS) / (sd * sd);
            const float dl = du * (e * std) + glp * (-(z * dzdl) - 1.0f);
            dmu[c] = c < act ? du : 0.f;
            dls[c] = c < act ? dl * (11.0f * (1.0f - tt * tt)) : 0.f;
        }
        float z4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float acc = fmaf(dls[0], wl[e][0], dmu[0] * wm[e][0]);
#pragma unroll
            for (int c = 1; c < 4; ++c) { acc = fmaf(dmu[c], wm[e][c], acc); acc = fmaf(dls[c], wl[e][c], acc); }
            z4[e] = f4e(hm, e) > 0.f ? acc : 0.f;
        }
        st_dg(jb.dz_c4 + ((long long)(n0 / 4 + cg) * Bn + row) * 4, make_float4(z4[0], z4[1], z4[2], z4[3]));
#pragma unroll
        for (int e = 0; e < 4; ++e) red[0][r][4 * cg + e] = z4[e];
        if (nt == 0 && cg == 0) {  // dhead image [row/4][32][4]: columns [dmu | dls]
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < act) {
                    jb.dhead_r4[((long long)(row >> 2) * 32 + c) * 4 + (row & 3)] = dmu[c];
                    jb.dhead_r4[((long long)(row >> 2) * 32 + act + c) * 4 + (row & 3)] = dls[c];
                }
        }
        __syncthreads();
        {
            const int c = tid & 31, rg = tid >> 5;
            st_dg(jb.dz_r4 + ((long long)(m0 / 4 + rg) * jb.ld_r4 + n0 + c) * 4,
                  make_float4(red[0][4 * rg][c], red[0][4 * rg + 1][c], red[0][4 * rg + 2][c], red[0][4 * rg + 3][c]));
        }
        DST(kid, 5); DRT(kid, 15);
        return;
    }

    // ---- GEMM tiles.  This wave's 8-deep contraction groups: the waves at the END get the extra ones
    const int Kc = hot_K, lda = hot_lda, ldb = hot_ldb;
    const float *Aop = hot_A, *Bop = hot_B;
    const int G = (Kc + 7) >> 3, gs = G >> 2, grem = G & 3;
    const int ng = gs + (w >= 4 - grem ? 1 : 0);
    const int g0 = w * gs + (w > 4 - grem ? w - (4 - grem) : 0);
    const bool first_n = nt == 0;

    // DGRAD_Q: the Q-head partials of this tile's rows (-> dq -> the generated A operand); c = 0..4 <-> evaluations 3..7.
    const bool need_q = type == DG_DGRAD_Q && hot_slot != 2;
    float4 qv[DNT / 4];
    {
        const int c = tid >> 5;
        const long long HP = (long long)DFH * Bn * DNT;
        const float4 *p4 = reinterpret_cast<const float4 *>(jobs.hp + (jobs.q_ev0 + (c < jobs.q_nev ? c : 0)) * HP + (long long)(m0 + r) * DNT);
#pragma unroll
        for (int q = 0; q < DNT / 4; ++q) qv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (need_q) {  // block-uniform
#pragma unroll
            for (int q = 0; q < DNT / 4; ++q) qv[q] = p4[q];
        }
    }
    // generated-operand inputs: W3 along the contraction (DGRAD_Q) or dq along it (generated wgrad operand).  The same 16
    // bytes for all 32 lanes of a half-wave: a per-lane load of them would cost the fetch path as much as a tile load,
    // so the vector is staged once per workgroup in LDS (two coalesced loads per thread) and read from there in the K loop.
    const bool has_gen = type == DG_DGRAD_Q || hot_bgen;  // block-uniform
    const bool agen = hot_bgen == 3;                       // block-uniform: DG_DGRAD of the policy, A = dZ2 generated
    float gv0 = 0.f, gv1 = 0.f, gwn = 0.f;
    if (has_gen) {
        const float *gp = hot_gp;
        const int glen = 8 * G;  // (W3 / dq are followed by readable memory up to the next multiple of 8)
        if (hot_bgen != 3) {   // (3: the policy dgrad generates its A operand from per-row gradients and the head kernels: below)
            gv0 = gp[tid < glen ? tid : 0];
            gv1 = gp[tid + 256 < glen ? tid + 256 : 0];
            if (type != DG_DGRAD_Q) gwn = jb.gw[(n0 + l31 < hot_N) ? n0 + l31 : 0];
        }
        if (type == DG_DGRAD_Q && jb.gw_snap && t == 0) {  // block-uniform
            if (tid < glen) jb.gw_snap[tid] = gv0;
            if (tid + 256 < glen) jb.gw_snap[tid + 256] = gv1;
        }
    }
    // both operand streams of this wave: unconditional loads, groups beyond ng re-read the last one (a branch or a
    // select in front of a load makes the compiler wait for the previous load before issuing the next)
    float4 a4[GMAX], b4[GMAX];
    const float *Ap = Aop + (long long)(m0 + l31) * 4, *Bp = Bop + (long long)(n0 + l31) * 4;
    const int glast = ng > 0 ? g0 + ng - 1 : (G > 0 ? G - 1 : 0);
    auto fetch_group = [&](int g) {
        const int gg = g0 + g < glast ? g0 + g : glast;
        const long long kg = 2 * gg + h;
        a4[g] = *reinterpret_cast<const float4 *>(Ap + kg * lda * 4);  // (nontemporal operand LOADS: 50.4-50.6 vs 49.5 us per update)
        b4[g] = *reinterpret_cast<const float4 *>(Bp + kg * ldb * 4);
    };
    // only the first DGP groups are requested up front; group g + DGP is requested when group g's MFMAs are issued (the
    // fetch path then works beside the matrix pipe, and the first MFMA does not queue behind the whole stream)
    constexpr int DGP = DGP0 < GMAX ? DGP0 : GMAX;
#pragma unroll
    for (int g = 0; g < DGP; ++g) fetch_group(g);
    // ---- policy dgrad with a GENERATED A operand (bgen = 3): A[row][k] = dZ2 of the policy trunk = (H2[row][k] > 0) * sum_a (dmu[row][a]
    // Wmu[k][a] + dls[row][a] Wls[k][a]) — the policy-head backward of this tile's 32 rows, every lane for its own row (the lane's row IS
    // the row of its A elements), with the arithmetic of the DG_ROWS_C tiles (which still write the images the policy wgrads contract
    // over, off this chain).  The dgrad no longer waits for a launch between the dQ/da partials and itself, and the layer-1 wgrad of
    // the policy becomes a plain job of the last launch instead of a hand-off at the end of the update's longest chain.
    float pdm[4] = {0.f, 0.f, 0.f, 0.f}, pdl[4] = {0.f, 0.f, 0.f, 0.f};
    if (agen) {
        __builtin_amdgcn_sched_barrier(0);
        const int act = jb.nact, row = m0 + l31;
        float4 dp[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) dp[q] = *reinterpret_cast<const float4 *>(jb.dap + ((long long)(q < jb.nparts ? q : 0) * Bn + row) * 4);
        float4 sv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) sv[c] = *reinterpret_cast<const float4 *>(jb.save0 + ((long long)row * act + (c < act ? c : 0)) * 4);
        // head kernels: thread = contraction index k (and k + 256): Wmu[k][0..act), Wls[k][0..act) -> one [k][8] row of s_w8 (unused action
        // slots and the rows K .. 8G - 1 are zero), two float2 loads per row when act == 2
        float4 wm4[2], wl4[2];
        const int k0 = tid, k1 = tid + 256;
        const bool ok0 = k0 < Kc, ok1 = k1 < Kc;
        const long long kb0 = (long long)(ok0 ? k0 : 0) * act, kb1 = (long long)(ok1 ? k1 : 0) * act;
        if (act == 2) {  // block-uniform; every load of a side is issued before the first use (one wait, not one per row)
            const float2 a0 = *reinterpret_cast<const float2 *>(jb.wmu + kb0), b0 = *reinterpret_cast<const float2 *>(jb.wls + kb0);
            const float2 a1 = *reinterpret_cast<const float2 *>(jb.wmu + kb1), b1 = *reinterpret_cast<const float2 *>(jb.wls + kb1);
            const float e0 = ok0 ? 1.0f : 0.f, e1 = ok1 ? 1.0f : 0.f;
            wm4[0] = make_float4(a0.x * e0, a0.y * e0, 0.f, 0.f); wl4[0] = make_float4(b0.x * e0, b0.y * e0, 0.f, 0.f);
            wm4[1] = make_float4(a1.x * e1, a1.y * e1, 0.f, 0.f); wl4[1] = make_float4(b1.x * e1, b1.y * e1, 0.f, 0.f);
        } else {
            float am[2][4], al[2][4];
#pragma unroll
            for (int a2 = 0; a2 < 4; ++a2) {
                const int ao = a2 < act ? a2 : 0;
                am[0][a2] = jb.wmu[kb0 + ao]; al[0][a2] = jb.wls[kb0 + ao]; am[1][a2] = jb.wmu[kb1 + ao]; al[1][a2] = jb.wls[kb1 + ao];
            }
#pragma unroll
            for (int a2 = 0; a2 < 4; ++a2) {
                const float e0 = (ok0 && a2 < act) ? 1.0f : 0.f, e1 = (ok1 && a2 < act) ? 1.0f : 0.f;
                am[0][a2] *= e0; al[0][a2] *= e0; am[1][a2] *= e1; al[1][a2] *= e1;
            }
            wm4[0] = make_float4(am[0][0], am[0][1], am[0][2], am[0][3]); wl4[0] = make_float4(al[0][0], al[0][1], al[0][2], al[0][3]);
            wm4[1] = make_float4(am[1][0], am[1][1], am[1][2], am[1][3]); wl4[1] = make_float4(al[1][0], al[1][1], al[1][2], al[1][3]);
        }
        float ga[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (q < jb.nparts) { ga[0] += dp[q].x; ga[1] += dp[q].y; ga[2] += dp[q].z; ga[3] += dp[q].w; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float av = sv[c].x, std = sv[c].y, tt = sv[c].z, e = sv[c].w;
            const float glp = row < Bv ? jb.alpha / (float)Bv : 0.f;  // d pi_loss / d logp_pi (padding rows carry no loss)
            const float om = 1.0f - av * av;
            const float cl = fminf(fmaxf(om, 0.f), 1.f);
            const float du = (ga[c] * jb.scale) * om + glp * ((2.0f * av * om) / (cl + 1e-6f));
            const float sd = std + STD_EPS;
            const float z = (e * std) / sd;
            const float dzdl = ((e * std) * STD_EPS) / (sd * sd);
            const float dl = du * (e * std) + glp * (-(z * dzdl) - 1.0f);
            pdm[c] = c < act ? du : 0.f;
            pdl[c] = c < act ? dl * (11.0f * (1.0f - tt * tt)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = tid + 256 * u;
            if (k < 8 * G) {
                if (act <= 2) *reinterpret_cast<float4 *>(&s_w8[k * 4]) = make_float4(wm4[u].x, wm4[u].y, wl4[u].x, wl4[u].y);   // packed rows: half the LDS reads
                else { *reinterpret_cast<float4 *>(&s_w8[k * 8]) = wm4[u]; *reinterpret_cast<float4 *>(&s_w8[k * 8 + 4]) = wl4[u]; }
            }
        }
    }
    const bool agen2 = agen && jb.nact <= 2;   // block-uniform
    ktouch(kl);
    DST(kid, 1);
    // ---- epilogue operands: requested from inside the K loop (behind the MFMAs of the first groups), so that the first
    // MFMA does not queue behind up to 16 more wave-loads per wave.  (A wave's loads return in order, so the optimizer state —
    // cold, from the fabric — still holds up the operand groups requested behind it: +2.8 k / +4.1 k cycles of K loop in the
    // two optimizer launches against the same launches without the step.  Measured alternative: a fifth "helper" wave per
    // workgroup that fetches the state into LDS before the combine barrier — the dispatcher then fits one 5-wave workgroup
    // per CU at 131 VGPRs (59.5 us per update), and capping the kernel at 128 VGPRs spills (55.7 us) against 52.5 us.)
    const bool is_dgrad = type == DG_DGRAD_Q || type == DG_DGRAD;
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    float wa_v = 0.f;  // staged into LDS after the K loop
    const bool has_da = is_dgrad && jb.da_part;  // block-uniform
    const bool rm_like = type == DG_WGRAD_RM || type == DG_WGRAD_W1Y;  // element-addressed gradient: row-major, or the layer-1 block layout
    const bool do_adam = jobs.ad.on && (type == DG_WGRAD_J4 || rm_like) && jb.adam_off >= 0;
    const bool narrow = rm_like && hot_N <= 8;   // block-uniform
    float al_pi = 0.f, al_q = 0.f;
    // optimizer state of this tile: J4 — thread (col r, row group cg) owns rows 4cg..4cg+3 of column r as one float4;
    // RM — thread owns elements (o >> 5, o & 31), o = tid + 256 q
    float4 j_m, j_v, j_p, j_t;
    float bm = 0.f, bv = 0.f, bp = 0.f, bt = 0.f;
    float am[4], av[4], ap[4], at[4];
    bool okv[4];
    long long j_idx = 0, b_idx = 0;
    bool j_ok = false, b_ok = false;
    auto epilogue_operands = [&]() {
        if (is_dgrad) mk = *reinterpret_cast<const float4 *>(jb.mask + ((long long)(m0 / 4 + cg) * jb.ldmask + n0 + r) * 4);  // mapping C: (col r, row group cg)
        if (has_da) {
            const int c = tid >> 5, col = n0 + (tid & 31);
            const bool ok = c < jb.nact && col < jb.N;
            const float v = jb.wa[ok ? w1y_index(jb.wa_d0 + c, col) : 0];
            wa_v = ok ? v : 0.f;
        }
        if (jobs.ad.on) {
            const float b1p_pi = jobs.ad.opt->b1p_pi, b2p_pi = jobs.ad.opt->b2p_pi, b1p_q = jobs.ad.opt->b1p_q, b2p_q = jobs.ad.opt->b2p_q;
            al_pi = jobs.ad.lr * sqrtf(1.0f - b2p_pi) / (1.0f - b1p_pi);
            al_q = jobs.ad.lr * sqrtf(1.0f - b2p_q) / (1.0f - b1p_q);
        }
        if (type == DG_WGRAD_J4) {
            j_ok = m0 + 4 * cg < jb.bias_row && n0 + r < jb.N;
            j_idx = jb.adam_off + ((long long)(m0 / 4 + cg) * jb.ldc + n0 + r) * 4;
            b_ok = m0 + 4 * cg == jb.bias_row && n0 + r < jb.N;  // (hidden1 % 4 == 0: the bias row opens a group)
            b_idx = jb.bias_off + n0 + r;
            if (do_adam) {
                const long long ic = j_ok ? j_idx : jb.adam_off;
                j_m = *reinterpret_cast<const float4 *>(jobs.ad.m + ic); j_v = *reinterpret_cast<const float4 *>(jobs.ad.v + ic);
                j_p = *reinterpret_cast<const float4 *>(jobs.ad.p + ic); j_t = *reinterpret_cast<const float4 *>(jobs.ad.t + ic);
                const long long bc = b_ok ? b_idx : jb.bias_off;
                bm = jobs.ad.m[bc]; bv = jobs.ad.v[bc]; bp = jobs.ad.p[bc]; bt = jobs.ad.t[bc];
            }
        } else if (rm_like) {
            // narrow outputs (the head kernels: N = act or 1 column): 32 rows x 8 columns = one element per thread instead of four —
            // a quarter of the optimizer-state loads queued in front of the operand groups that follow them
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (narrow && q > 0) { okv[q] = false; continue; }   // block-uniform
                const int o = tid + 256 * q;
                const int gi = m0 + (narrow ? (o >> 3) : (o >> 5)), gj = n0 + (narrow ? (o & 7) : (o & 31));
                okv[q] = gi < jb.M && gj < jb.N;
                if (do_adam) {
                    const long long idx = !okv[q] ? jb.adam_off : jb.adam_off + (type == DG_WGRAD_W1Y ? w1y_index(gi, gj) : (long long)gi * jb.ldc + gj);
                    am[q] = jobs.ad.m[idx]; av[q] = jobs.ad.v[idx]; ap[q] = jobs.ad.p[idx]; at[q] = jobs.ad.t[idx];
                }
            }
        }
    };
    // ---- DGRAD_Q prologue: q1, q2, q1(x,pi), the target backup, the per-row loss terms and dq = dLoss/dq (actor_learner.py:58-69)
    float dqr = m0 + l31 < Bv ? jb.gconst : 0.f;
    if (has_gen && !agen) { s_gw[tid] = gv0; s_gw[tid + 256] = gv1; }
    if (has_gen && !need_q) __syncthreads();
    if (type == DG_DGRAD_Q) {
        if (need_q) {
            float rew = 0.f, done = 0.f, lp0 = 0.f, lp1 = 0.f;
            if (w == 0) { rew = jobs.rew[m0 + l31]; done = jobs.done[m0 + l31]; lp0 = jobs.logp0[m0 + l31]; lp1 = jobs.logp1[m0 + l31]; }
            const float b3_1 = jobs.b3q1[0], b3_2 = jobs.b3q2[0], b3_1t = jobs.b3q1t[0], b3_2t = jobs.b3q2t[0];
            float qsum = 0.f;
#pragma unroll
            for (int q = 0; q < DNT / 4; ++q) { qsum += qv[q].x; qsum += qv[q].y; qsum += qv[q].z; qsum += qv[q].w; }
            s_q[tid >> 5][tid & 31] = qsum;
            __syncthreads();
            if (w == 0 && !jobs.sacv) {
                const int row = m0 + l31;
                const float q1v = s_q[0][l31] + b3_1, q2v = s_q[1][l31] + b3_2, q1pi = s_q[2][l31] + b3_1;
                const float q1t = s_q[3][l31] + b3_1t, q2t = s_q[4][l31] + b3_2t;
                const float minq = fminf(q1t, q2t);                             // actor_learner.py:59
                const float vb = minq - jobs.alpha * lp1;                       // :62
                const float backup = rew + (jobs.gamma * (1.0f - done)) * vb;   // :63
                const float e1 = backup - q1v, e2 = backup - q2v;
                const float inv_b = row < Bv ? 1.0f / (float)Bv : 0.f;  // padding rows: no loss terms, zero upstream gradients
                const float dq1 = -e1 * inv_b, dq2 = -e2 * inv_b;
                if (lane < 32) {
                    s_g[l31] = jb.slot == 0 ? dq1 : dq2;
                    if (first_n && jb.slot == 0) {
                        jobs.q1o[row] = q1v; jobs.q2o[row] = q2v;
                        jobs.loss_part[row * 3 + 0] = jobs.alpha * lp0 - q1pi;  // :66
                        jobs.loss_part[row * 3 + 1] = e1 * e1;                  // :67
                        jobs.loss_part[row * 3 + 2] = e2 * e2;                  // :68
                        jobs.dq[row] = dq1;
                        jobs.dq[Bn + row] = dq2;
                    }
                }
            } else if (w == 0) {
                // SAC-v (example/model.py:38-50); summed evaluations: q1(x,a) q2(x,a) v(x) v_targ(x2) q1(x,pi) q2(x,pi)
                const int row = m0 + l31;
                const float q1v = s_q[0][l31] + b3_1, q2v = s_q[1][l31] + b3_2, vv = s_q[2][l31] + b3_1t, vt = s_q[3][l31] + b3_2t;
                const float q1pi = s_q[4][l31] + b3_1, q2pi = s_q[5][l31] + b3_2;
                const float q_backup = rew + (jobs.gamma * (1.0f - done)) * vt;        // model.py:41
                const float v_backup = fminf(q1pi, q2pi) - jobs.alpha * lp0;           // :38,42
                const float e1 = q_backup - q1v, e2 = q_backup - q2v, ev = v_backup - vv;
                const float inv_b = row < Bv ? 1.0f / (float)Bv : 0.f;
                const float dq1 = -e1 * inv_b, dq2 = -e2 * inv_b, dv = -ev * inv_b;
                if (lane < 32) {
                    s_g[l31] = jb.slot == 0 ? dq1 : (jb.slot == 1 ? dq2 : dv);
                    if (first_n && jb.slot == 0) {
                        jobs.q1o[row] = q1v; jobs.q2o[row] = q2v; jobs.vo[row] = vv; jobs.vto[row] = vt;
                        jobs.loss_part[row * 4 + 0] = jobs.alpha * lp0 - q1pi;  // :45
                        jobs.loss_part[row * 4 + 1] = e1 * e1;                  // :46
                        jobs.loss_part[row * 4 + 2] = e2 * e2;                  // :47
                        jobs.loss_part[row * 4 + 3] = ev * ev;                  // :48
                        jobs.dq[row] = dq1;
                        jobs.dq[Bn + row] = dq2;
                        jobs.dq[2 * Bn + row] = dv;
                    }
                }
            }
            __syncthreads();
            dqr = s_g[l31];
        }
    }
    // ---- K loop
    DST(kid, 2);
    floatx16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        if (g + DGP < GMAX) {
            fetch_group(g + DGP);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g == DGMID) {
            epilogue_operands();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g < ng) {
            float4 av4 = a4[g], bv4 = b4[g];
            if (type == DG_DGRAD_Q) {
                const float4 w3 = *reinterpret_cast<const float4 *>(&s_gw[8 * (g0 + g) + 4 * h]);
                av4 = make_float4(av4.x > 0.f ? dqr * w3.x : 0.f, av4.y > 0.f ? dqr * w3.y : 0.f, av4.z > 0.f ? dqr * w3.z : 0.f, av4.w > 0.f ? dqr * w3.w : 0.f);
            } else if (agen2) {
                // act <= 2: rows [Wmu0 Wmu1 Wls0 Wls1]; the terms of the action slots 2, 3 are exact zeros and add nothing (fma(0, 0, acc) == acc)
                const float *wk = &s_w8[(8 * (g0 + g) + 4 * h) * 4];
                float z4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wk + 4 * e);
                    float acc = fmaf(pdl[0], w4.z, pdm[0] * w4.x);
                    acc = fmaf(pdm[1], w4.y, acc); acc = fmaf(pdl[1], w4.w, acc);
                    z4[e] = acc;
                }
                av4 = make_float4(av4.x > 0.f ? z4[0] : 0.f, av4.y > 0.f ? z4[1] : 0.f, av4.z > 0.f ? z4[2] : 0.f, av4.w > 0.f ? z4[3] : 0.f);
            } else if (agen) {
                const float *wk = &s_w8[(8 * (g0 + g) + 4 * h) * 8];
                float z4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float4 m4 = *reinterpret_cast<const float4 *>(wk + 8 * e), l4 = *reinterpret_cast<const float4 *>(wk + 8 * e + 4);
                    float acc = fmaf(pdl[0], l4.x, pdm[0] * m4.x);     // (the summation order of the DG_ROWS_C tiles)
                    acc = fmaf(pdm[1], m4.y, acc); acc = fmaf(pdl[1], l4.y, acc);
                    acc = fmaf(pdm[2], m4.z, acc); acc = fmaf(pdl[2], l4.z, acc);
                    acc = fmaf(pdm[3], m4.w, acc); acc = fmaf(pdl[3], l4.w, acc);
                    z4[e] = acc;
                }
                av4 = make_float4(av4.x > 0.f ? z4[0] : 0.f, av4.y > 0.f ? z4[1] : 0.f, av4.z > 0.f ? z4[2] : 0.f, av4.w > 0.f ? z4[3] : 0.f);
            } else if (jb.bgen) {
                const float4 d4 = *reinterpret_cast<const float4 *>(&s_gw[8 * (g0 + g) + 4 * h]);
                bv4 = make_float4(bv4.x > 0.f ? d4.x * gwn : 0.f, bv4.y > 0.f ? d4.y * gwn : 0.f, bv4.z > 0.f ? d4.z * gwn : 0.f, bv4.w > 0.f ? d4.w * gwn : 0.f);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av4.x, bv4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av4.y, bv4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av4.z, bv4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av4.w, bv4.w, acc, 0, 0, 0);
        }
    }
    // ---- split-K combine.  D layout: col = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
    DST(kid, 3);
    if (has_da && tid < 128) s_wa[tid >> 5][tid & 31] = wa_v;
#pragma unroll
    for (int q = 0; q < 16; ++q) red[w][(q & 3) + 8 * (q >> 2) + 4 * h][l31] = acc[q];
    __syncthreads();
    DST(kid, 4);

    if (is_dgrad) {
        // mapping C: thread (col r, row group cg): rows m0 + 4cg .. + 3 of column n0 + r
        float o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = ((red[0][4 * cg + i][r] + red[1][4 * cg + i][r]) + red[2][4 * cg + i][r]) + red[3][4 * cg + i][r];
            o4[i] = f4e(mk, i) > 0.f ? s : 0.f;
        }
        const bool colok = n0 + r < jb.N;
        if (jb.C && colok) st_dg(jb.C + ((long long)(m0 / 4 + cg) * jb.ldc + n0 + r) * 4, make_float4(o4[0], o4[1], o4[2], o4[3]));
        if (has_da) {  // block-uniform: dQ/da partial of this column tile (rows x action dims)
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) red[0][4 * cg + i][r] = colok ? o4[i] : 0.f;
            __syncthreads();
            const int c = tid >> 5;
            if (c < 4) {
                float s = 0.f;
#pragma unroll
                for (int col = 0; col < 32; ++col) s = fmaf(red[0][r][col], s_wa[c][col], s);
                jb.da_part[((long long)nt * Bn + m0 + r) * 4 + c] = s;
            }
        }
        DST(kid, 5); DRT(kid, 15);
        return;
    }
    if (type == DG_WGRAD_J4) {
        // gradient tile rows = hidden-1 index (the contraction axis of the forward), cols = hidden-2 index
        float g4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g4[i] = ((red[0][4 * cg + i][r] + red[1][4 * cg + i][r]) + red[2][4 * cg + i][r]) + red[3][4 * cg + i][r];
        const float omb1 = 1.0f - jobs.ad.b1, omb2 = 1.0f - jobs.ad.b2;
        const float al = jb.adam_off < jobs.ad.n_pi ? al_pi : al_q;
        if (j_ok) {
            *reinterpret_cast<float4 *>(jobs.ad.g + j_idx) = make_float4(g4[0], g4[1], g4[2], g4[3]);
            if (do_adam) {
                adam1(g4[0], j_m.x, j_v.x, j_p.x, j_t.x, omb1, omb2, al, jobs.ad.eps, jobs.ad.pk, jobs.ad.pk1);
                adam1(g4[1], j_m.y, j_v.y, j_p.y, j_t.y, omb1, omb2, al, jobs.ad.eps, jobs.ad.pk, jobs.ad.pk1);
                adam1(g4[2], j_m.z, j_v.z, j_p.z, j_t.z, omb1, omb2, al, jobs.ad.eps, jobs.ad.pk, jobs.ad.pk1);
                adam1(g4[3], j_m.w, j_v.w, j_p.w, j_t.w, omb1, omb2, al, jobs.ad.eps, jobs.ad.pk, jobs.ad.pk1);
                st_opt(jobs.ad.m + j_idx, j_m); st_opt(jobs.ad.v + j_idx, j_v);
                st_opt(jobs.ad.p + j_idx, j_p); st_opt(jobs.ad.t + j_idx, j_t);
            }
        }
        if (b_ok) {
            jobs.ad.g[b_idx] = g4[0];
            if (do_adam) {
                adam1(g4[0], bm, bv, bp, bt, omb1, omb2, al, jobs.ad.eps, jobs.ad.pk, jobs.ad.pk1);
                jobs.ad.m[b_idx] = bm; jobs.ad.v[b_idx] = bv; jobs.ad.p[b_idx] = bp; jobs.ad.t[b_idx] = bt;
            }
        }
        if (do_adam && jb.shadow) {  // block-uniform: the updated kernel in the dgrad layout [n/4][k][4]
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) red[0][4 * cg + i][r] = f4e(j_p, i);
            __syncthreads();
            // mapping R: thread (row r, col group cg): columns n0 + 4cg .. + 3 of row m0 + r
            if (m0 + r < jb.bias_row && n0 + 4 * cg < jb.N)
                *reinterpret_cast<float4 *>(jb.shadow + ((long long)(n0 / 4 + cg) * jb.ld_sh + m0 + r) * 4) =
                    make_float4(red[0][r][4 * cg], red[0][r][4 * cg + 1], red[0][r][4 * cg + 2], red[0][r][4 * cg + 3]);
        }
        DST(kid, 5); DRT(kid, 15);
        return;
    }
    // DG_WGRAD_RM / DG_WGRAD_W1Y: element-addressed gradient (head kernels row-major; layer-1 kernels with their bias in the block layout)
    {
        const float omb1 = 1.0f - jobs.ad.b1, omb2 = 1.0f - jobs.ad.b2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (narrow && q > 0) continue;   // block-uniform
            const int o = tid + 256 * q;
            const int row = narrow ? (o >> 3) : (o >> 5), col = narrow ? (o & 7) : (o & 31);
            const float gv = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
            if (okv[q]) {
                const long long idx = jb.adam_off + (type == DG_WGRAD_W1Y ? w1y_index(m0 + row, n0 + col) : (long long)(m0 + row) * jb.ldc + n0 + col);
                jobs.ad.g[idx] = gv;
                if (do_adam) {
                    const float al = idx < jobs.ad.n_pi ? al_pi : al_q;
                    adam1(gv, am[q], av[q], ap[q], at[q], omb1, omb2, al, jobs.ad.eps, jobs.ad.pk, jobs.ad.pk1);
                    jobs.ad.m[idx] = am[q]; jobs.ad.v[idx] = av[q]; jobs.ad.p[idx] = ap[q]; jobs.ad.t[idx] = at[q];
                }
            }
        }
    }
    DST(kid, 5); DRT(kid, 15);
}

static void dg_add(DGJobs &js, DGJob j) {
    j.gp = j.type == DG_DGRAD_Q ? j.gw : j.gdq;
    j.tiles_m = (j.M + 31) / 32;
    j.ntiles = j.type == DG_LOSS ? 1 : j.tiles_m * ((j.N + 31) / 32);
    j.tile_start = js.total_tiles;
    js.total_tiles += j.ntiles;
    for (int i = js.njobs; i < MAX_DG_JOBS; ++i) js.tile_start[i] = 0xffff;   // (k_dg compares 16-bit starts)
    js.tile_start[js.njobs] = j.tile_start;
    js.job[js.njobs++] = j;
}
static void launch_dg(const DGJobs &J_, hipStream_t s, int kid = 0) {
#ifdef DDRL_STAMPS
    DGJobs J = J_;
    J.st = g_st_host;
#else
    const DGJobs &J = J_;
#endif
    const int *ts = J.tile_start;
    int per_wave = 0;  // deepest 8-group count of a wave over the launch's GEMM jobs
    for (int i = 0; i < J.njobs; ++i) {
        const DGJob &j = J.job[i];
        if (j.type == DG_ROWS_C || j.type == DG_LOSS) continue;
        const int G = (j.K + 7) / 8, pw = (G + 3) / 4;
        if (pw > per_wave) per_wave = pw;
    }
    int pk[8];
    for (int i = 0; i < 8; ++i) pk[i] = (ts[2 * i] & 0xffff) | (int)((unsigned)(ts[2 * i + 1] & 0xffff) << 16);   // pk[0]'s low half (job 0 starts at 0) is unused
    if (per_wave <= 10) k_dg<10><<<J.total_tiles, 256, 0, s>>>(J.total_tiles, pk[0], pk[1], pk[2], pk[3], pk[4], pk[5], pk[6], pk[7], kid, J);
    else k_dg<DGMAX><<<J.total_tiles, 256, 0, s>>>(J.total_tiles, pk[0], pk[1], pk[2], pk[3], pk[4], pk[5], pk[6], pk[7], kid, J);
}

// The dgrad image [N/4][ld][4] of a k4-interleaved kernel [K/4][Np][4] (after a set_weights / import / flat Adam step)
struct ShadowJobs {
    const float *j4[4];
    float *c4[4];
};
__global__ void __launch_bounds__(256) k_shadow(ShadowJobs sj, int K, int N, int Np, int ld_sh) {
    const float *__restrict__ j4 = sj.j4[blockIdx.z];
    float *__restrict__ c4 = sj.c4[blockIdx.z];
    const int k = blockIdx.x * 32 + (threadIdx.x & 31), ng = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (k >= K || 4 * ng >= N) return;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (4 * ng + e < N) ? j4[((long long)(k >> 2) * Np + 4 * ng + e) * 4 + (k & 3)] : 0.f;
    *reinterpret_cast<float4 *>(c4 + ((long long)ng * ld_sh + k) * 4) = make_float4(v[0], v[1], v[2], v[3]);
}
