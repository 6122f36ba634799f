// Fused forward stages of the SAC1 update (included by sac1.hip, inside its anonymous namespace).
//
// Why: at batch 256 an update is a chain of dependent kernels and every kernel boundary costs
// ~2.5 us (launch + cold L2 + the first round trips), more than the arithmetic of the small row
// kernels between the GEMMs.  The fused path removes the boundaries instead of polishing them:
//
//   k_fwd<0>  = layer 1 of evaluations 0-4 generated straight into the GEMM's A operand in LDS
//               (K = obs(+act) <= 12: 8-12 FMAs per element, cheaper than fetching the tile),
//               the layer-2 GEMM, and the head dot products as per-column-tile partials
//               hp[head][n-tile][row] (the 32x32 output tile is in LDS anyway);
//   k_fwd<1>  = the same for evaluations 5-7, whose action input is computed in the prologue from
//               the policy-head partials of k_fwd<0> (sum over n-tiles in a fixed order, tanh-squash,
//               log-prob: core.py:49-87) — this is what k_rows_a + the second half of k_l1 did.
//
// The first-column-tile workgroups also write what later stages read from memory: X1 of the
// evaluations that are differentiated (H1 slots 0,3,4,5), the augmented input rows [x|1], [x|a|1]
// of the layer-1 wgrads, act0/act2/logp0/logp1/save0, and (phase 0) the generated noise.
// Shapes outside the fused path's envelope (see fused_ok()) use the generic kernels of sac1.hip.

constexpr int FD = 12;      // layer-1 input width handled by the generator (obs_dim + act_dim <= 12)
constexpr int FH = 8;       // heads per evaluation: 2 * act_dim <= 8
constexpr int F_MAXNT = 16; // column tiles of hidden2 (<= 512 / 32)

struct FIn {  // X1 = relu([in0 | in1] * W1 + b1)
    const float *in0, *in1;  // [B][d0], [B][d1]; phase 1: in1 = nullptr, the action comes from the policy heads
    const float *W1, *b1;
    int d0, d1;
};

struct FwdJob {
    FIn in;
    const float *W2, *b2;
    float *H2;               // [B][ldh2]
    float *H1;               // side output of the n-tile-0 workgroups: X1 rows [B][ldh1] (nullptr: not needed later)
    float *aug;              // side output of the n-tile-0 workgroups: [in0 | in1] rows with row stride aug_ld (ones column is physical)
    int aug_ld;
    float *w2snap;           // side output of the m-tile-0 workgroups: a copy of W2 as read by this update (see k_gemm's fused
                             // optimizer step: the policy dgrad must not read W2 while the wgrad tiles update it in place)
    const float *wh0, *wh1;  // head kernels — policy: Wmu, Wls ([h2][act]); Q: W3 ([h2]), unused
    int nh, hsplit, hstride; // heads; heads < hsplit come from wh0, the rest from wh1; element stride between rows of a head kernel
    float *hp;               // head partials [nh][B][F_MAXNT] (n-tile innermost: a consumer reads its row's partials as four float4; slots >= nt2 stay 0)
    // phase 1: the policy evaluation whose sampled action is this job's second input
    const float *php;        // [2*act][B][F_MAXNT]
    const float *pbmu, *pbls;
    const float *peps;       // [B][act]
    int side;                // n-tile-0 workgroups: 1 -> act0, logp0, save0   2 -> act2, and logp1 from the evaluation below
};

// What the first loads of k_fwd need (the W2 tile and the W1 columns) travels as leading SCALAR kernel
// arguments: those are preloaded into SGPRs at wave launch (kernarg preload), whereas fields of a
// by-value struct are fetched with s_load from the kernarg segment, which is cold after every kernel
// boundary — the ISA showed three serialized round trips (gridDim, tile counts, pointers) = 2.5 k
// cycles before the first vector load was issued.
struct FwdHead {
    const float *pbase;        // parameters are addressed as pbase + offset (main and target live in one slab)
    int tiles_m, tpj, h1, h2;  // tpj = tiles_m * tiles_n: tiles per job
    int w2_off[5], w1_off[5];  // per job
    int hp_off;                // head partial buffer (phase 1 passes it in the scalar slot of job 3)
};
struct FwdArgs {
    FwdHead hd;
    int tiles_n, njobs, ks_max, op_lds;
    int B, ldh1, ldh2, act, nt2;
    float scale;
    float *act0, *act2, *logp0, *logp1, *save0;
    const float *php1, *pbmu1, *pbls1, *peps1;  // pi_main @ x2: only its log-prob is needed (actor_learner.py:62)
    int noise_on, n_each;
    uint32_t noise_seed;
    float *e0, *e1, *e2;
    const OptState *opt;
    // phase 1, optional: one extra workgroup draws the NEXT update's batch (np.random.randint + the five
    // gathers, example/dsac.py:39-45) into the learner's other input set — phase 1 has idle CUs
    // (240 tiles on 256) and nothing in this update touches that set or the sampler state
    int do_sample, sample_batch;
    ddrl_replay_dev::RingState *rs;
    ddrl_replay_dev::RingPtrs ring;
    ddrl_replay_dev::BatchPtrs sout;
    FwdJob job[5];
#ifdef DDRL_STAMPS
    long long *stamps;
#endif
};

// One row of one policy evaluation, all action dims in one lane.  Same formulae and operation order
// as policy_head() (core.py:49-87, 104-106); mu / log_std pre-activations arrive as sums of partials.
struct PolRow {
    float act[4], a[4], std[4], t[4];
    float logp;
};
__device__ __forceinline__ PolRow policy_row(const float (&mu)[4], const float (&lsr)[4], const float (&eps)[4], int act, float scale) {
    PolRow o;
    float sp = 0.f, sc = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        o.act[c] = 0.f; o.a[c] = 0.f; o.std[c] = 0.f; o.t[c] = 0.f;
        if (c < act) {
            const float t = tanhf(lsr[c]);
            const float log_std = -20.0f + 11.0f * (t + 1.0f);
            const float std = expf(log_std);
            const float e = eps[c];
            const float u = mu[c] + e * std;
            const float z = (e * std) / (std + STD_EPS);
            const float pre = -0.5f * ((z * z + 2.0f * log_std) + LOG2PI);
            const float a = tanhf(u);
            const float om = 1.0f - a * a;
            const float cl = fminf(fmaxf(om, 0.f), 1.f);
            const float corr = logf(cl + 1e-6f);
            sp += pre; sc += corr;
            o.act[c] = a * scale; o.a[c] = a; o.std[c] = std; o.t[c] = t;
        }
    }
    o.logp = sp - sc;
    return o;
}

// Layer 1 on the matrix cores: X1[32 rows][32 units] = relu(in[32][D] * W1[D][32] + b1) is D/2
// v_mfma_f32_32x32x2_f32 (exact fp32 FMAs, input column ascending) per 32-unit block; the result
// goes from the accumulator layout straight into the K-contiguous A image of the layer-2 GEMM.
// (A VALU generator with the input tile broadcast from LDS was LDS-bandwidth bound: 8 waves x 1 KB
// per ds_read_b128 -> 4.6 k cycles per sub-chunk; this form takes ~0.7 k.)
struct L1Blk { float w[FD / 2]; float b; };  // lane (l31, h): W1[2j + h][unit], b1[unit] of one 32-unit block
__device__ __forceinline__ void l1_block(const float (&av)[FD / 2], const L1Blk &wb, int D, floatx16 &x) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = 0.f;
#pragma unroll
    for (int j = 0; j < FD / 2; ++j)
        if (2 * j < D) x = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wb.w[j], x, 0, 0, 0);  // wave-uniform guard
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = fmaxf(x[r] + wb.b, 0.f);
}

template <int PH>
__global__ void __launch_bounds__(256) k_fwd(const float *pbase, int tiles_m, int tpj, int h1_, int h2_, int w2o0, int w2o1, int w2o2, int w2o3,
                                             int w2o4, int w1o0, int w1o1, int w1o2, int w1o3, int w1o4, FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ __attribute__((aligned(16))) float s_in[FD][32];  // transposed input tile: [input column][row]
    __shared__ float s_wh[FH][32];
    __shared__ float s_hd[2][FH][32];
    if (PH == 1 && (int)blockIdx.x == 3 * tpj) {  // only launched when a.do_sample
        ddrl_replay_dev::sample_block(a.rs, a.ring, a.sout, a.sample_batch, nullptr, 1);
        return;
    }
#ifdef DDRL_STAMPS
    long long *st = (a.stamps && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 0) ? a.stamps + (long long)blockIdx.x * 32 : nullptr;
    STAMP(0);
#endif
    int t;
    {   // XCD-aware, panel-major tile order (see k_gemm)
        const int nwg = (PH == 0 ? 5 : 3) * tpj, b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;  // == gridDim.x (a hidden-argument load)
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int ji = t / tpj;
    t -= ji * tpj;
    const int m0 = (t % tiles_m) * 32, nt = t / tiles_m, n0 = nt * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int K = h1_, B = a.B, H2N = h2_;
    const int chunk = ((K + 15) >> 4) << 2;
    const int k0 = w * chunk;
    const int k1 = (k0 + chunk < K) ? (k0 + chunk) : K;
    const int kw = k1 > k0 ? k1 - k0 : 0;
    const int ks_a = kw < 64 ? kw : 64, ks_b = kw - ks_a;  // sub-chunks aligned to the 32-unit layer-1 blocks
    float *sA = smem + (w * 2 + 0) * a.op_lds, *sB = smem + (w * 2 + 1) * a.op_lds;
    const int kcs = a.ks_max + 2;
    const bool first_n = nt == 0;
    STAMP(15);

    // ---- every independent load of the kernel, up front --------------------------------------
    // Phase 1 first issues what its longest chain starts with: the policy-head partials of this tile's rows
    // (-> action -> layer-1 input).  Three jobs only, so the scalar slots of jobs 3 / 4 carry the offset of the
    // partial buffer and batch | act << 16; the policy evaluation is fixed by the job order (job 0: pi(x) = eval 0;
    // jobs 1, 2: pi_targ(x2) = eval 2; log-prob of pi(x2) = eval 1 for job 1's first column tile).
    float hsum[2] = {0.f, 0.f};
    if (PH == 1) {
        const int Bq = w2o4 & 0xffff, actq = w2o4 >> 16;
        const long long HPq = (long long)FH * Bq * F_MAXNT;
        const int c = tid >> 5, r = tid & 31;
        const bool okc = c < 2 * actq;
        const bool two_q = ji == 1 && first_n;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e == 0 || two_q) {
                const float *hp = pbase + w2o3 + (e == 0 ? (ji == 0 ? 0 : 2) : 1) * HPq;
                const float4 *p4 = reinterpret_cast<const float4 *>(hp + ((long long)(okc ? c : 0) * Bq + m0 + r) * F_MAXNT);
                float4 v[F_MAXNT / 4];
#pragma unroll
                for (int q = 0; q < F_MAXNT / 4; ++q) v[q] = p4[q];
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < F_MAXNT / 4; ++q) { s += v[q].x; s += v[q].y; s += v[q].z; s += v[q].w; }  // n-tile order; unused slots are 0
                hsum[e] = s;
            }
        }
    }
    // (1) + (2) need only the preloaded part of the arguments
    const int w2o = ji == 0 ? w2o0 : (ji == 1 ? w2o1 : (ji == 2 ? w2o2 : (ji == 3 ? w2o3 : w2o4)));
    const int w1o = ji == 0 ? w1o0 : (ji == 1 ? w1o1 : (ji == 2 ? w1o2 : (ji == 3 ? w1o3 : w1o4)));
    const float *W2 = pbase + w2o, *W1 = pbase + w1o;
    // (1) B operand = W2 rows of this wave's K range
    Op2<false> ob;
    ob.init(W2, H2N, n0, H2N, k0, lane);
    float4 pb[8];
    STAMP(16);
    ob.load(0, 0, ks_a, H2N, pb);
    STAMP(11);
    // (2) W1 / b1 of the wave's (up to four) 32-unit blocks, directly in the MFMA B-operand layout;
    // rows beyond the job's input width are fetched (they lie inside the parameter slab) and zeroed
    // once the width is known.
    // (Fetching [W1 ; b1] as six 1-KB row loads and redistributing through LDS was measured slower:
    // 17.2 vs 14.0 us per launch — the extra LDS hop sits on the critical path before the first MFMA.)
    L1Blk wb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int kc = k0 + q * 32 + l31;
        const int kcc = kc < k1 ? kc : 0;
#pragma unroll
        for (int j = 0; j < FD / 2; ++j) wb[q].w[j] = W1[(long long)(2 * j + h) * K + kcc];
    }
    STAMP(12);
    const FwdJob &jb = a.job[ji];
    const int d0 = jb.in.d0, d1 = PH == 0 ? jb.in.d1 : a.act, D = d0 + d1;
#pragma unroll
    for (int q = 0; q < 4; ++q) wb[q].b = jb.in.b1[(k0 + q * 32 + l31 < k1) ? k0 + q * 32 + l31 : 0];
    STAMP(13);
    // (3) epilogue operands
    const float biasv = jb.b2[(n0 + l31 < H2N) ? n0 + l31 : 0];
    float whv = 0.f;
    {
        const int c = tid >> 5, col = n0 + (tid & 31);
        const bool ok = c < jb.nh && col < H2N;
        const float *p = c < jb.hsplit ? jb.wh0 : jb.wh1;
        const int cc = c < jb.hsplit ? c : c - jb.hsplit;
        const float v = p[ok ? (long long)col * jb.hstride + cc : 0];
        whv = ok ? v : 0.f;
    }
    // (4) the 32 input rows of this tile
    float inv[2];
    bool inok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = tid + 256 * u;           // 32 * FD = 384 slots
        const int d = idx >> 5, r = idx & 31;
        const int row = m0 + r;
        const int lim = PH == 0 ? D : d0;        // phase 1: the action columns are computed below
        inok[u] = idx < 32 * FD && d < lim;
        const bool from0 = d < d0;
        const float *p = (from0 || !inok[u]) ? jb.in.in0 : jb.in.in1;
        const long long off = from0 ? (long long)row * d0 + d : (long long)row * d1 + (d - d0);
        inv[u] = p[inok[u] ? off : 0];
    }
    STAMP(14);
    // (5) phase 1: the noise of the policy evaluation(s)
    float epsv[4] = {0.f, 0.f, 0.f, 0.f};
    const bool two = PH == 1 && jb.side == 2 && first_n;  // block-uniform
    if (PH == 1) {
        if (w < 2) {  // wave 0: the action-giving evaluation; wave 1: pi_main @ x2 (side == 2)
            const float *pe = w == 0 ? jb.peps : a.peps1;
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) epsv[c2] = (w == 0 || two) ? pe[(long long)(m0 + l31) * a.act + (c2 < a.act ? c2 : 0)] : 0.f;
        }
    }

    STAMP(1);
    // ---- stage the shared inputs ------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = tid + 256 * u;
        if (idx < 32 * FD) {
            const int d = idx >> 5, r = idx & 31;
            const float v = inok[u] ? inv[u] : 0.f;
            if (PH == 0 || d < d0 || d >= D) s_in[d][r] = v;
            if (jb.aug && first_n && inok[u]) jb.aug[(long long)(m0 + r) * jb.aug_ld + d] = v;
        }
    }
    s_wh[tid >> 5][tid & 31] = whv;

    if (PH == 1) {
        s_hd[0][tid >> 5][tid & 31] = hsum[0];
        if (two) s_hd[1][tid >> 5][tid & 31] = hsum[1];
    }
    if (PH == 0 && a.noise_on && ji == 0 && first_n) {
        // eps_x, eps_x2, eps_t of this tile's rows; element index as in one flat [3][B*act] fill
        const unsigned long long base = a.opt->noise_ctr;
        const int per_row = 3 * a.act;
        for (int e = tid; e < 32 * per_row; e += 256) {
            const int rr = e / per_row, q = e - rr * per_row;
            const int wch = q / a.act, c = q - wch * a.act;
            const int k = (m0 + rr) * a.act + c;
            (wch == 0 ? a.e0 : (wch == 1 ? a.e1 : a.e2))[k] = normal_at(a.noise_seed, base + (unsigned long long)wch * a.n_each + k);
        }
    }
    __syncthreads();
    if (PH == 1) {
        if (w == 0 || (w == 1 && two)) {  // wave-uniform
            const int e = w;               // 0: action-giving evaluation, 1: pi_main @ x2
            const float *bmu = e == 0 ? jb.pbmu : a.pbmu1, *bls = e == 0 ? jb.pbls : a.pbls1;
            float mu[4], ls[4], ev[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cc = c < a.act ? c : 0;
                mu[c] = s_hd[e][cc][l31] + bmu[cc];
                ls[c] = s_hd[e][a.act + cc][l31] + bls[cc];
                ev[c] = epsv[c];
            }
            const PolRow o = policy_row(mu, ls, ev, a.act, a.scale);
            const int row = m0 + l31;
            if (e == 0 && lane < 32) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < a.act) s_in[d0 + c][l31] = o.act[c];
            }
            if (first_n && lane < 32) {
                if (e == 0 && jb.side == 1) {
                    a.logp0[row] = o.logp;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < a.act) {
                            a.act0[row * a.act + c] = o.act[c];
                            *reinterpret_cast<float4 *>(a.save0 + ((long long)row * a.act + c) * 4) = make_float4(o.a[c], o.std[c], o.t[c], ev[c]);
                        }
                } else if (e == 0 && jb.side == 2) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < a.act) a.act2[row * a.act + c] = o.act[c];
                } else if (e == 1) {
                    a.logp1[row] = o.logp;
                }
            }
        }
        __syncthreads();
    }

    STAMP(2);
    // ---- K loop: A generated, B loaded -------------------------------------------------------
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float *h1row = (jb.H1 && first_n) ? jb.H1 + (long long)m0 * a.ldh1 : nullptr;
    float av[FD / 2];  // A operand of the layer-1 MFMAs: in[row l31][2j + h]
#pragma unroll
    for (int j = 0; j < FD / 2; ++j) av[j] = s_in[2 * j + h][l31];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool okk = k0 + q * 32 + l31 < k1;
#pragma unroll
        for (int j = 0; j < FD / 2; ++j) wb[q].w[j] = (okk && 2 * j + h < D) ? wb[q].w[j] : 0.f;
    }
    // X1 of a sub-chunk (two 32-unit blocks) from the accumulator layout into the A image
    auto put = [&](const floatx16 &x, int bb, int ks, int kbase) {
        const int col = bb * 32 + l31;
        if (bb * 32 < ks && col < ks) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sA[((r & 3) + 8 * (r >> 2) + 4 * h) * kcs + col] = x[r];
            if (h1row) {
#pragma unroll
                for (int r = 0; r < 16; ++r) h1row[(long long)((r & 3) + 8 * (r >> 2) + 4 * h) * a.ldh1 + kbase + col] = x[r];
            }
        }
    };
    floatx16 xa0, xa1, xb0, xb1;
    l1_block(av, wb[0], D, xa0);
    if (32 < ks_a) l1_block(av, wb[1], D, xa1);
    put(xa0, 0, ks_a, k0);
    put(xa1, 1, ks_a, k0);
    // the second sub-chunk's layer 1 runs on the matrix cores while the W2 tile is still in flight;
    // only its LDS writes have to wait until the first sub-chunk's A image has been consumed
    if (0 < ks_b) l1_block(av, wb[2], D, xb0);
    if (32 < ks_b) l1_block(av, wb[3], D, xb1);
    STAMP(3);
    ob.store(sB, ks_a, kcs, lane, pb);
    float *snap = (jb.w2snap && m0 == 0 && ob.okc[0]) ? jb.w2snap + (long long)k0 * H2N + n0 + (lane & 7) * 4 : nullptr;
    if (snap) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (ob.krow[i] < ks_a) *reinterpret_cast<float4 *>(snap + (long long)ob.krow[i] * H2N) = pb[i];
    }
    if (ks_b > 0) ob.load(1, ks_a, ks_b, H2N, pb);
    wave_lds_sync();
    STAMP(4);
    mfma_chunk<true, false>(sA, sB, ks_a, kcs, l31, h, acc);
    STAMP(5);
    if (ks_b > 0) {
        wave_lds_sync();
        put(xb0, 0, ks_b, k0 + 64);
        put(xb1, 1, ks_b, k0 + 64);
        STAMP(6);
        ob.store(sB, ks_b, kcs, lane, pb);
        if (snap) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (ob.krow[i] < ks_b) *reinterpret_cast<float4 *>(snap + (long long)(ks_a + ob.krow[i]) * H2N) = pb[i];
        }
        wave_lds_sync();
        STAMP(7);
        mfma_chunk<true, false>(sA, sB, ks_b, kcs, l31, h, acc);
        STAMP(8);
    }

    // ---- split-K combine, bias + relu, H2 store, head partials -------------------------------
    __syncthreads();
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(smem);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][(r & 3) + 8 * (r >> 2) + 4 * h][l31] = acc[r];
    __syncthreads();
    float outv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = tid + 256 * q;
        const int row = o >> 5, col = o & 31;
        outv[q] = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
    }
    const bool colok = n0 + (tid & 31) < H2N;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = tid + 256 * q;
        const int gi = m0 + (o >> 5), gj = n0 + (o & 31);
        float v = fmaxf(outv[q] + biasv, 0.f);
        v = colok ? v : 0.f;
        if (colok) jb.H2[(long long)gi * a.ldh2 + gj] = v;
        outv[q] = v;
    }
    STAMP(9);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = tid + 256 * q;
        red[0][o >> 5][o & 31] = outv[q];
    }
    __syncthreads();
    {
        const int c = tid >> 5, r = tid & 31;
        if (c < jb.nh) {
            float s = 0.f;
#pragma unroll
            for (int col = 0; col < 32; ++col) s = fmaf(red[0][r][col], s_wh[c][col], s);
            jb.hp[((long long)c * B + m0 + r) * F_MAXNT + nt] = s;
        }
    }
    STAMP(10);
}

static size_t fwd_smem(const FwdArgs &a) {
    const size_t x = (size_t)4 * 2 * a.op_lds * sizeof(float), y = (size_t)RED_LDS * sizeof(float);
    return x > y ? x : y;
}

template <int PH>
static void launch_fwd(const FwdArgs &F, hipStream_t s) {
    const FwdHead &d = F.hd;
    const int s3 = PH == 1 ? d.hp_off : d.w2_off[3], s4 = PH == 1 ? (F.B | (F.act << 16)) : d.w2_off[4];
    k_fwd<PH><<<F.njobs * d.tpj + ((PH == 1 && F.do_sample) ? 1 : 0), 256, fwd_smem(F), s>>>(d.pbase, d.tiles_m, d.tpj, d.h1, d.h2, d.w2_off[0], d.w2_off[1], d.w2_off[2], s3, s4,
                                                        d.w1_off[0], d.w1_off[1], d.w1_off[2], d.w1_off[3], d.w1_off[4], F);
}

// ==========================================================================================
// k_bwdq: backward of the three Q evaluations through layer 2 (the dgrad GEMMs), with what
// k_rows_b did folded in: the prologue turns the Q-head partials of the forward stages into
// q1(x,a), q2(x,a), q1(x,pi), the target backup, the per-row loss terms and dq = dLoss/dq
// (actor_learner.py:58-69) for the 32 rows of the tile, and the A operand
//     dZ2[i][k] = dq[i] * W3[k] * (H2[i][k] > 0)
// is produced from the H2 tile while it is staged into LDS — dZ2 never makes a round trip through
// memory on this path.  The n-tile-0 workgroups also write dZ2 / dq (B operands of the Q wgrads,
// which run in the next launch), q1/q2 and the loss terms.
// ==========================================================================================
struct BqJob {
    const float *H2;   // [B][ldh2] of the differentiated evaluation
    const float *W3;   // [h2]
    const float *W2;   // [h1][h2]
    const float *H1;   // [B][ldh1]: relu mask of the dgrad output
    float *dZ1;        // [B][h1]
    float *dZ2;        // side output [B][h2] (nullptr: not needed)
    int slot;          // 0: q1(x,a)   1: q2(x,a)   2: q1(x,pi)
};
struct BqHead {
    const float *pbase;
    int tiles_m, tpj, h1, h2;
    int h2_off[3], w2_off[3];
};
struct BqArgs {
    BqHead hd;
    int tiles_n, ks_max, op_lds, B, ldh1, ldh2, nt2;
    const float *hp;   // head partials [NEVAL][FH][B][F_MAXNT]
    const float *b3q1, *b3q2, *b3q1t, *b3q2t;
    const float *rew, *done, *logp0, *logp1;
    float *q1o, *q2o, *dq4, *loss_part;
    float alpha, gamma;
    BqJob job[3];
};

__global__ void __launch_bounds__(256) k_bwdq(const float *pbase, int tiles_m, int tpj, int h1_, int h2_, int h2o0, int h2o1, int h2o2, int w2o0,
                                              int w2o1, int w2o2, int hp_off, int batch, BqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float s_q[8][32];
    __shared__ float s_g[32];
    int t;
    {
        const int nwg = 3 * tpj, b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int ji = t / tpj;
    t -= ji * tpj;
    const int m0 = (t % tiles_m) * 32, nt = t / tiles_m, n0 = nt * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int K = h2_, N = h1_, ldh2 = ((h2_ + 1) + 3) & ~3;
    const int chunk = ((K + 15) >> 4) << 2;
    const int k0 = w * chunk;
    const int k1 = (k0 + chunk < K) ? (k0 + chunk) : K;
    const int kw = k1 > k0 ? k1 - k0 : 0;
    const int half = ((kw + 7) >> 3) << 2;
    const int ks_a = half < kw ? half : kw, ks_b = kw - ks_a;
    const bool first_n = nt == 0;

    // ---- loads that need only the preloaded scalars.  First what the longest chain starts with: the Q-head partials of
    // this tile's rows (-> dq -> the generated A operand); job order is fixed (0: q1(x,pi), which needs none; 1, 2: q1, q2 at (x,a)).
    float qsum = 0.f;
    if (ji != 0) {  // c = 0..4 <-> evaluations 3,4,5,6,7
        const int c = tid >> 5, r = tid & 31;
        const long long HP = (long long)FH * batch * F_MAXNT;
        const float4 *p4 = reinterpret_cast<const float4 *>(pbase + hp_off + (3 + (c < 5 ? c : 0)) * HP + (long long)(m0 + r) * F_MAXNT);
        float4 v[F_MAXNT / 4];
#pragma unroll
        for (int q = 0; q < F_MAXNT / 4; ++q) v[q] = p4[q];
#pragma unroll
        for (int q = 0; q < F_MAXNT / 4; ++q) { qsum += v[q].x; qsum += v[q].y; qsum += v[q].z; qsum += v[q].w; }
    }
    // both operand tiles of the first sub-chunk
    const float *H2 = pbase + (ji == 0 ? h2o0 : (ji == 1 ? h2o1 : h2o2));
    const float *W2 = pbase + (ji == 0 ? w2o0 : (ji == 1 ? w2o1 : w2o2));
    Op2<true> oa, ob;
    oa.init(H2, ldh2, m0, 0x7fffffff, k0, lane);  // rows are always valid (B % 32 == 0)
    ob.init(W2, K, n0, N, k0, lane);
    float4 pa[8], pb[8];
    oa.load(0, 0, ks_a, ldh2, pa);
    ob.load(0, 0, ks_a, K, pb);
    // ---- everything else (needs the argument record)
    const BqJob &jb = a.job[ji];
    const int slot = jb.slot, B = a.B;
    float *sA = smem + (w * 2 + 0) * a.op_lds, *sB = smem + (w * 2 + 1) * a.op_lds;
    const int kcs = a.ks_max + 2;
    const int kk = (lane & 15) * 4;
    // W3 of this lane's four k of each sub-chunk
    const float4 w3a = *reinterpret_cast<const float4 *>(jb.W3 + ((kk < ks_a) ? k0 + kk : 0));
    const float4 w3b = *reinterpret_cast<const float4 *>(jb.W3 + ((kk < ks_b) ? k0 + ks_a + kk : 0));
    // epilogue mask
    float maskv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = tid + 256 * q;
        const int gi = m0 + (o >> 5), gj = n0 + (o & 31);
        maskv[q] = jb.H1[(long long)gi * a.ldh1 + (gj < N ? gj : 0)];
    }
    // Q-head partials of this tile's rows: c = 0..4 <-> evaluations 3,4,5,6,7
    const bool need_q = slot != 2;  // block-uniform: the q1(x,pi) path has dq = -1/B
    float rew = 0.f, done = 0.f, lp0 = 0.f, lp1 = 0.f;
    if (need_q) {
        if (w == 0) { rew = a.rew[m0 + l31]; done = a.done[m0 + l31]; lp0 = a.logp0[m0 + l31]; lp1 = a.logp1[m0 + l31]; }
    }
    const float b3_1 = a.b3q1[0], b3_2 = a.b3q2[0], b3_1t = a.b3q1t[0], b3_2t = a.b3q2t[0];
    if (need_q) {
        s_q[tid >> 5][tid & 31] = qsum;
        __syncthreads();
        if (w == 0) {
            const int row = m0 + l31;
            const float q1v = s_q[0][l31] + b3_1, q2v = s_q[1][l31] + b3_2, q1pi = s_q[2][l31] + b3_1;
            const float q1t = s_q[3][l31] + b3_1t, q2t = s_q[4][l31] + b3_2t;
            const float minq = fminf(q1t, q2t);                          // actor_learner.py:59
            const float vb = minq - a.alpha * lp1;                       // :62
            const float backup = rew + (a.gamma * (1.0f - done)) * vb;   // :63
            const float e1 = backup - q1v, e2 = backup - q2v;
            const float inv_b = 1.0f / (float)B;
            const float dq1 = -e1 * inv_b, dq2 = -e2 * inv_b;
            if (lane < 32) {
                s_g[l31] = slot == 0 ? dq1 : dq2;
                if (first_n && slot == 0) {
                    a.q1o[row] = q1v; a.q2o[row] = q2v;
                    a.loss_part[row * 3 + 0] = a.alpha * lp0 - q1pi;     // :66
                    a.loss_part[row * 3 + 1] = e1 * e1;                  // :67
                    a.loss_part[row * 3 + 2] = e2 * e2;                  // :68
                    *reinterpret_cast<float4 *>(a.dq4 + (long long)row * 4) = make_float4(dq1, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4 *>(a.dq4 + ((long long)B + row) * 4) = make_float4(dq2, 0.f, 0.f, 0.f);
                }
            }
        }
    } else if (tid < 32) {
        s_g[tid] = -1.0f / (float)B;
    }
    __syncthreads();
    float gr[8];  // dq of this lane's eight A-tile rows
#pragma unroll
    for (int i = 0; i < 8; ++i) gr[i] = s_g[i * 4 + (lane >> 4)];
    float *z2row = (jb.dZ2 && first_n) ? jb.dZ2 + (long long)m0 * K : nullptr;

    // A operand: dZ2 tile made from the H2 tile on its way into LDS
    auto stage_a = [&](const float4 (&v)[8], const float4 &w3, int ks, int kbase) {
        if (kk < ks) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = i * 4 + (lane >> 4);
                const float g = gr[i];
                const float4 z = make_float4(v[i].x > 0.f ? g * w3.x : 0.f, v[i].y > 0.f ? g * w3.y : 0.f,
                                             v[i].z > 0.f ? g * w3.z : 0.f, v[i].w > 0.f ? g * w3.w : 0.f);
                float *d = sA + r * kcs + kk;
                *reinterpret_cast<float2 *>(d) = make_float2(z.x, z.y);
                *reinterpret_cast<float2 *>(d + 2) = make_float2(z.z, z.w);
                if (z2row) *reinterpret_cast<float4 *>(z2row + (long long)r * K + kbase + kk) = z;
            }
        }
    };
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    stage_a(pa, w3a, ks_a, k0);
    ob.store(sB, ks_a, kcs, lane, pb);
    if (ks_b > 0) {
        oa.load(1, ks_a, ks_b, ldh2, pa);
        ob.load(1, ks_a, ks_b, K, pb);
    }
    wave_lds_sync();
    mfma_chunk<true, true>(sA, sB, ks_a, kcs, l31, h, acc);
    if (ks_b > 0) {
        wave_lds_sync();
        stage_a(pa, w3b, ks_b, k0 + ks_a);
        ob.store(sB, ks_b, kcs, lane, pb);
        wave_lds_sync();
        mfma_chunk<true, true>(sA, sB, ks_b, kcs, l31, h, acc);
    }
    // split-K combine, relu mask, dZ1 store
    __syncthreads();
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(smem);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][(r & 3) + 8 * (r >> 2) + 4 * h][l31] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = tid + 256 * q;
        const int row = o >> 5, col = o & 31;
        const float s = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
        const int gj = n0 + col;
        if (gj < N) jb.dZ1[(long long)(m0 + row) * N + gj] = maskv[q] > 0.f ? s : 0.f;
    }
}

static size_t bq_smem(const BqArgs &a) {
    const size_t x = (size_t)4 * 2 * a.op_lds * sizeof(float), y = (size_t)RED_LDS * sizeof(float);
    return x > y ? x : y;
}
static void launch_bwdq(const BqArgs &A, hipStream_t s) {
    const BqHead &d = A.hd;
    k_bwdq<<<3 * d.tpj, 256, bq_smem(A), s>>>(d.pbase, d.tiles_m, d.tpj, d.h1, d.h2, d.h2_off[0], d.h2_off[1], d.h2_off[2], d.w2_off[0], d.w2_off[1],
                                              d.w2_off[2], (int)(A.hp - d.pbase), A.B, A);
}
