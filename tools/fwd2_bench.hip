// Dev harness for the "direct operand" forward stage (phase-0 shape: 5 evaluations, 256 x (8|10 -> 400) -> 300):
// layer 1 generated TRANSPOSED on the matrix cores (D = X1^T: lane <-> batch row, register <-> unit), so that its
// accumulator registers ARE the A operand of the layer-2 MFMAs; W2 is stored k4-interleaved ([K/4][N][4]) so that
// one float4 load per lane is the B operand of four MFMA steps.  No LDS staging of operands at all.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form tools/fwd2_bench.hip -o tools/fwd2_bench.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct F2Job {
    const float *b2;   // [Np]
    const float *x;    // [B][8]
    const float *a;    // [B][2] or nullptr
    const float *wh;   // [8][Np] head kernels (rows >= nh are zero)
    float *H2c4;       // [Np/4][B][4] or nullptr
    float *H2r4;       // [B/4][Np][4] or nullptr
    float *hp;         // [8][B][16]
    int nh;
};
struct F2Args {
    int B, K, N, tiles_n, njobs;
    F2Job job[5];
};

__device__ __forceinline__ int slot_of(int s, int h) { return s < 4 ? 4 * h + s : 8 + 2 * (s - 4) + h; }
__device__ __forceinline__ float relu1(float x) {
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
}

// One 32-unit block of the K loop: layer 1 (NS MFMA steps) -> relu -> 4 * nrq layer-2 MFMA steps.
template <int NS>
__device__ __forceinline__ void block_step(const float (&w1)[6], const float4 (&bia)[4], const float (&xin)[6], const float4 (&bq)[4], int nrq,
                                           floatx16 &acc) {
    floatx16 x1;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) { x1[4 * rq + 0] = bia[rq].x; x1[4 * rq + 1] = bia[rq].y; x1[4 * rq + 2] = bia[rq].z; x1[4 * rq + 3] = bia[rq].w; }
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

template <int NS, int NBMAX>
__device__ __forceinline__ void kloop(const float *__restrict__ W1, const float *__restrict__ W2p, int Kp, int Np, int K, int b0, int nb, int n0, int l31,
                                      int h, const float (&xin)[6], floatx16 &acc) {
    float4 bq[NBMAX][4], bia[NBMAX][4];
    float w1[NBMAX][6];
#pragma unroll
    for (int bi = 0; bi < NBMAX; ++bi) {  // loads beyond nb are clamped to the last block (harmless re-reads, no branches)
        const int u0 = (b0 + (bi < nb ? bi : nb - 1)) * 32;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) bq[bi][rq] = *reinterpret_cast<const float4 *>(W2p + ((long long)(u0 / 4 + 2 * rq + h) * Np + n0 + l31) * 4);
    }
#pragma unroll
    for (int bi = 0; bi < NBMAX; ++bi) {
        const int u0 = (b0 + (bi < nb ? bi : nb - 1)) * 32;
#pragma unroll
        for (int s = 0; s < NS; ++s) w1[bi][s] = W1[(long long)slot_of(s, h) * Kp + u0 + l31];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) bia[bi][rq] = *reinterpret_cast<const float4 *>(W1 + (long long)12 * Kp + u0 + 8 * rq + 4 * h);
    }
#pragma unroll
    for (int bi = 0; bi < NBMAX; ++bi) {
        if (bi < nb) {
            const int u0 = (b0 + bi) * 32;
            const int nrq = (K - u0 >= 32) ? 4 : ((K - u0 + 7) >> 3);
            block_step<NS>(w1[bi], bia[bi], xin, bq[bi], nrq, acc);
        }
    }
}

// NW waves per workgroup split the K blocks (4: two workgroups per CU when tiles > CUs; 8: one workgroup per CU, two waves per SIMD)
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_fwd2(const float *base, int tpj, int tiles_m, int K, int Kp, int Np, int B, int w2o0, int w2o1, int w2o2, int w2o3,
                                                    int w2o4, int w1o0, int w1o1, int w1o2, int w1o3, int w1o4, int ns_pack, F2Args a) {
    __shared__ float red[NW][32][33];
    __shared__ float s_wh[8][32];
    int t;
    {
        const int nwg = a.njobs * tpj, b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int ji = (t >= tpj) + (t >= 2 * tpj) + (t >= 3 * tpj) + (t >= 4 * tpj);
    t -= ji * tpj;
    const int nt = (t * (65536 / tiles_m + 1)) >> 16;  // t / tiles_m for t < 65536 / ... (tiles_m <= 64)
    const int m0 = (t - nt * tiles_m) * 32, n0 = nt * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int w2o = ji == 0 ? w2o0 : (ji == 1 ? w2o1 : (ji == 2 ? w2o2 : (ji == 3 ? w2o3 : w2o4)));
    const int w1o = ji == 0 ? w1o0 : (ji == 1 ? w1o1 : (ji == 2 ? w1o2 : (ji == 3 ? w1o3 : w1o4)));
    const int ns = (ns_pack >> (4 * ji)) & 15;
    const float *W2p = base + w2o, *W1 = base + w1o;
    // this wave's 32-unit blocks: the waves at the END get the extra (possibly partial) block
    const int nblk = (K + 31) >> 5, bs = nblk / NW, rem = nblk - bs * NW;
    const int nb = bs + (w >= NW - rem ? 1 : 0);
    const int b0 = w * bs + (w > NW - rem ? w - (NW - rem) : 0);
    const F2Job &jb = a.job[ji];
    float xin[6];
    {
        const int row = m0 + l31;
        const float4 x4 = *reinterpret_cast<const float4 *>(jb.x + (long long)row * 8 + 4 * h);
        xin[0] = x4.x; xin[1] = x4.y; xin[2] = x4.z; xin[3] = x4.w;
        float av = 0.f;
        if (jb.a) av = jb.a[(long long)row * 2 + h];
        xin[4] = av; xin[5] = 0.f;
    }
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NBMAX = NW == 4 ? 4 : 2;
    if (nb > 0) {
        if (ns == 4) kloop<4, NBMAX>(W1, W2p, Kp, Np, K, b0, nb, n0, l31, h, xin, acc);
        else if (ns == 5) kloop<5, NBMAX>(W1, W2p, Kp, Np, K, b0, nb, n0, l31, h, xin, acc);
        else kloop<6, NBMAX>(W1, W2p, Kp, Np, K, b0, nb, n0, l31, h, xin, acc);
    }
    const float4 b4 = *reinterpret_cast<const float4 *>(jb.b2 + n0 + 4 * ((tid & 255) >> 5));
    if (tid < 256) s_wh[tid >> 5][tid & 31] = jb.wh[(long long)(tid >> 5) * Np + n0 + (tid & 31)];
    // ---- split-K combine
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][(r & 3) + 8 * (r >> 2) + 4 * h][l31] = acc[r];
    __syncthreads();
    const int r = tid & 31, cg = (tid & 255) >> 5;
    float v[4];
    if (tid < 256) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * cg + e;
            float s = red[0][r][c];
#pragma unroll
            for (int q = 1; q < NW; ++q) s += red[q][r][c];
            v[e] = fmaxf(s + (e == 0 ? b4.x : (e == 1 ? b4.y : (e == 2 ? b4.z : b4.w))), 0.f);
        }
        if (jb.H2c4) *reinterpret_cast<float4 *>(jb.H2c4 + ((long long)(n0 / 4 + cg) * B + m0 + r) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    if (tid < 256) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[0][r][4 * cg + e] = v[e];
    }
    __syncthreads();
    if (tid < 256) {
        if (jb.H2r4) {
            const int c = tid & 31, rg = tid >> 5;
            *reinterpret_cast<float4 *>(jb.H2r4 + ((long long)(m0 / 4 + rg) * Np + n0 + c) * 4) =
                make_float4(red[0][4 * rg][c], red[0][4 * rg + 1][c], red[0][4 * rg + 2][c], red[0][4 * rg + 3][c]);
        }
        const int c = tid >> 5;
        if (c < jb.nh) {
            float s = 0.f;
#pragma unroll
            for (int col = 0; col < 32; ++col) s = fmaf(red[0][r][col], s_wh[c][col], s);
            jb.hp[((long long)c * B + m0 + r) * 16 + nt] = s;
        }
    }
}

static float frand() { return (float)(rand() & 0xffffff) / (float)0x1000000 * 2.f - 1.f; }

struct Offs { const float *base; int w2o[5], w1o[5], ns_pack; int Kp, Np; };

template <int NW>
static void run(const F2Args &a, const Offs &o, const char *name) {
    const int tiles_m = a.B / 32, tpj = tiles_m * a.tiles_n;
    const int grid = a.njobs * tpj;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() {
        k_fwd2<NW><<<grid, 64 * NW>>>(o.base, tpj, tiles_m, a.K, o.Kp, o.Np, a.B, o.w2o[0], o.w2o[1], o.w2o[2], o.w2o[3], o.w2o[4], o.w1o[0], o.w1o[1], o.w1o[2],
                                      o.w1o[3], o.w1o[4], o.ns_pack, a);
    };
    for (int i = 0; i < 5; ++i) go();
    hipDeviceSynchronize();
    float ms;
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) go();
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.2f us/launch (grid %d x %d) err=%s\n", name, ms * 5.f, grid, 64 * NW, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int B = 256, K = 400, Kp = 416, N = 300, Np = 320, NE = 5;
    F2Args a{};
    a.B = B; a.K = K; a.N = N; a.tiles_n = (N + 31) / 32; a.njobs = NE;
    const size_t per = (size_t)13 * Kp + (size_t)Kp * Np;  // [W1 | bias row | W2p]
    std::vector<float> hP(NE * per, 0.f), hW2((size_t)NE * K * N), hb2((size_t)NE * Np, 0.f), hx((size_t)B * 8), hac((size_t)B * 2), hwh((size_t)NE * 8 * Np, 0.f);
    srand(1);
    for (auto &v : hx) v = frand();
    for (auto &v : hac) v = frand();
    for (int e = 0; e < NE; ++e) {
        float *W1 = hP.data() + e * per, *W2p = W1 + (size_t)13 * Kp;
        const int D = e < 3 ? 8 : 10;
        for (int d = 0; d < D; ++d)
            for (int k = 0; k < K; ++k) W1[(size_t)d * Kp + k] = frand() * 0.3f;
        for (int k = 0; k < K; ++k) W1[(size_t)12 * Kp + k] = frand() * 0.1f;
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < N; ++n) {
                const float v = frand() * 0.1f;
                hW2[((size_t)e * K + k) * N + n] = v;
                W2p[((size_t)(k / 4) * Np + n) * 4 + (k & 3)] = v;
            }
        for (int n = 0; n < N; ++n) hb2[(size_t)e * Np + n] = frand() * 0.1f;
        for (int c = 0; c < (e < 3 ? 4 : 1); ++c)
            for (int n = 0; n < N; ++n) hwh[((size_t)e * 8 + c) * Np + n] = frand();
    }
    float *dP, *db2, *dx, *dac, *dwh, *dH2c4, *dH2r4, *dhp;
    hipMalloc(&dP, hP.size() * 4); hipMalloc(&db2, hb2.size() * 4); hipMalloc(&dx, hx.size() * 4);
    hipMalloc(&dac, hac.size() * 4); hipMalloc(&dwh, hwh.size() * 4);
    hipMalloc(&dH2c4, (size_t)NE * Np * B * 4); hipMalloc(&dH2r4, (size_t)NE * Np * B * 4); hipMalloc(&dhp, (size_t)NE * 8 * B * 16 * 4);
    hipMemset(dhp, 0, (size_t)NE * 8 * B * 16 * 4);
    hipMemcpy(dP, hP.data(), hP.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db2, hb2.data(), hb2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dac, hac.data(), hac.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dwh, hwh.data(), hwh.size() * 4, hipMemcpyHostToDevice);
    Offs o{}; o.base = dP; o.Kp = Kp; o.Np = Np;
    for (int e = 0; e < NE; ++e) {
        F2Job j{};
        o.w1o[e] = (int)(e * per); o.w2o[e] = (int)(e * per + (size_t)13 * Kp);
        o.ns_pack |= (e < 3 ? 4 : 5) << (4 * e);
        j.b2 = db2 + (size_t)e * Np; j.x = dx; j.a = e < 3 ? nullptr : dac;
        j.wh = dwh + (size_t)e * 8 * Np; j.nh = e < 3 ? 4 : 1;
        const bool keep = (e == 0 || e >= 3);
        j.H2c4 = keep ? dH2c4 + (size_t)e * Np * B : nullptr; j.H2r4 = keep ? dH2r4 + (size_t)e * Np * B : nullptr;
        j.hp = dhp + (size_t)e * 8 * B * 16;
        a.job[e] = j;
    }
    run<8>(a, o, "phase-0 shape, NW=8");
    run<4>(a, o, "phase-0 shape, NW=4");
    {   // phase-1-like: three Q-shaped jobs (240 tiles)
        F2Args a3 = a; Offs o3 = o;
        a3.njobs = 3;
        for (int e = 0; e < 3; ++e) { a3.job[e] = a.job[3 + (e & 1)]; o3.w1o[e] = o.w1o[3 + (e & 1)]; o3.w2o[e] = o.w2o[3 + (e & 1)]; }
        o3.ns_pack = 0x555;
        run<4>(a3, o3, "phase-1 shape (3 Q jobs), NW=4");
        run<8>(a3, o3, "phase-1 shape (3 Q jobs), NW=8");
    }
    run<4>(a, o, "phase-0 shape, NW=4 (again, for the check)");
    // ---- correctness: eval 0, 3 (H2 + heads), eval 1 (heads only) vs fp64
    std::vector<float> gH2c4((size_t)NE * Np * B), gH2r4((size_t)NE * Np * B), ghp((size_t)NE * 8 * B * 16);
    hipMemcpy(gH2c4.data(), dH2c4, gH2c4.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(gH2r4.data(), dH2r4, gH2r4.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ghp.data(), dhp, ghp.size() * 4, hipMemcpyDeviceToHost);
    for (int e : {0, 3, 1}) {
        const float *W1 = hP.data() + e * per;
        const int D = e < 3 ? 8 : 10;
        double maxerr = 0, maxerr_r4 = 0, maxhp = 0, maxv = 0;
        for (int r = 0; r < B; ++r) {
            std::vector<double> x1(K), h2(N);
            for (int k = 0; k < K; ++k) {
                double s = W1[(size_t)12 * Kp + k];
                for (int d = 0; d < D; ++d) s += (double)(d < 8 ? hx[(size_t)r * 8 + d] : hac[(size_t)r * 2 + d - 8]) * W1[(size_t)d * Kp + k];
                x1[k] = s > 0 ? s : 0;
            }
            for (int n = 0; n < N; ++n) {
                double s = hb2[(size_t)e * Np + n];
                for (int k = 0; k < K; ++k) s += x1[k] * hW2[((size_t)e * K + k) * N + n];
                h2[n] = s > 0 ? s : 0;
                maxv = std::max(maxv, h2[n]);
                if (e != 1) {
                    maxerr = std::max(maxerr, std::fabs(h2[n] - gH2c4[(size_t)e * Np * B + ((size_t)(n / 4) * B + r) * 4 + (n & 3)]));
                    maxerr_r4 = std::max(maxerr_r4, std::fabs(h2[n] - gH2r4[(size_t)e * Np * B + ((size_t)(r / 4) * Np + n) * 4 + (r & 3)]));
                }
            }
            for (int c = 0; c < (e < 3 ? 4 : 1); ++c) {
                double s = 0, g = 0;
                for (int n = 0; n < N; ++n) s += h2[n] * hwh[((size_t)e * 8 + c) * Np + n];
                for (int q = 0; q < 16; ++q) g += ghp[(size_t)e * 8 * B * 16 + ((size_t)c * B + r) * 16 + q];
                maxhp = std::max(maxhp, std::fabs(s - g));
            }
        }
        printf("eval %d: max|H2| %.3f  err c4 %.3g  err r4 %.3g  err heads %.3g\n", e, maxv, maxerr, maxerr_r4, maxhp);
    }
    return 0;
}
