// One-pass backward of a [Conv 1x1 -> BatchNorm -> ReLU] layer for N, K <= 128:
//   dz   = BatchNorm/ReLU backward of this layer from (g, z)            (prologue 2 of pn2_mlp_gemm)
//   dX   = dz * W, masked by the ReLU of the layer below, + its BatchNorm-backward column sums
//   dW  += dz^T * act(x),  db += column sums of dz
// The two-kernel path (pn2_mlp_gemm prologue 2 + pn2_mlp_dw) reads g and z twice and the layer input
// twice; these layers are HBM-bound (M up to 524288 rows of 32..128 floats), so this kernel stages
// every 64-row tile ONCE into LDS and feeds both products from it.  Workgroups are persistent
// (grid-stride over tiles): W lives in LDS, the dW block and the column sums live in registers until
// the end, where each workgroup writes one partial slab (summed in fixed order afterwards).
// autograd of reference models/pointnet2_utils.py:196-200, :312-314.
#include "pn2_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FB_THREADS = 512, FB_WAVES = 8, FB_ROWS = 64;

struct BwdArgs {
    const float *g, *z;                 // g [M][N] (or pooled [M/pool_k][N]), z [M][N]
    int ldg, ldz;
    const float *scale, *shift, *mean, *invstd, *c1, *c2;   // this layer (N channels)
    const unsigned char *argk;
    int pool_k;
    const float *w;                     // [N][K]
    int ldw;
    const float *x;                     // layer input [M][K]: raw z of the layer below, or an activation
    int ldx;
    const float *ascale, *ashift, *amean, *ainvstd;          // layer below (null: x is an activation)
    float *gp;                          // [M][K] dX (nullable)
    int ldgp;
    float *stat_partial;                // [grid][2][K] (nullable)
    float *dw_partial;                  // [grid][N][K+1]
    int M, N, K;
};

// The weight tile [NP][KP] on its way from global memory to LDS: every load of a thread is issued before the first
// store (a `load; store` loop waits for one memory round trip per pass: 6-8 of them at the head of every launch), the
// addresses are clamped and the values masked when stored, and the caller may issue its first tile's loads in between.
template <int NP, int KP, int THREADS>
struct WeightTile {
    static constexpr int TOTAL = NP * (KP / 4), PASSES = (TOTAL + THREADS - 1) / THREADS;
    float4 v[PASSES];
    __device__ __forceinline__ void load(const float *w, int ldw, int N, int K, int tid)
    {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int e = tid + i * THREADS;
            const int n = e / (KP / 4), k4 = (e - n * (KP / 4)) * 4;
            v[i] = *reinterpret_cast<const float4 *>(w + (size_t)min(n, N - 1) * ldw + (k4 < K ? k4 : 0));
        }
    }
    __device__ __forceinline__ void store(float *sW, int N, int K, int tid) const
    {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int e = tid + i * THREADS;
            const int n = e / (KP / 4), k4 = (e - n * (KP / 4)) * 4;
            if (TOTAL % THREADS == 0 || e < TOTAL)
                *reinterpret_cast<float4 *>(&sW[n * KP + k4]) = (n < N && k4 < K) ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
};

template <int NBLK, int KBLK, bool POOLED>
__global__ __launch_bounds__(FB_THREADS, (NBLK * KBLK <= 4) ? 4 : 2) void mlp_bwd_fused_kernel(BwdArgs p)
{
    PN2_MAIN_BRANCH_PRIORITY();
    constexpr int NP = 32 * NBLK, KP = 32 * KBLK;
    constexpr int LDD = NP + 4, LDZ = KP + 4;
    constexpr int NDX = 2 * KBLK, NDW = NBLK * KBLK;
    constexpr int DXPW = (NDX + FB_WAVES - 1) / FB_WAVES, DWPW = (NDW + FB_WAVES - 1) / FB_WAVES;
    constexpr int DROWS = FB_THREADS / (NP / 4);      // rows staged per pass of the dz tile (x NBLK passes = 64)
    constexpr int XROWS = FB_THREADS / (KP / 4);
    constexpr int RED = (4 * KP > DROWS * NP) ? 4 * KP : DROWS * NP;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sD = smem;                                 // [64][LDD]  dz tile
    float *sZ = sD + FB_ROWS * LDD;                   // [64][LDZ]  raw layer input tile
    float *sW = sZ + FB_ROWS * LDZ;                   // [NP][KP]   weights, zero outside [N][K]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int ntiles = (p.M + FB_ROWS - 1) / FB_ROWS;
    const bool masked = p.ascale != nullptr;

    WeightTile<NP, KP, FB_THREADS> wt;
    wt.load(p.w, p.ldw, p.N, p.K, tid);

    // staging ownership: dz columns dc4..dc4+3, rows dr + DROWS*i; input columns xc4.., rows xr + XROWS*i
    const int dc4 = (tid % (NP / 4)) * 4, dr = tid / (NP / 4);
    const int xc4 = (tid % (KP / 4)) * 4, xr = tid / (KP / 4);
    const bool n_ok = dc4 < p.N, k_ok = xc4 < p.K;
    float4 sc, sh, mu, is, a1, a2;
    sc = sh = mu = is = a1 = a2 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dbs = sc;
    if (n_ok) {
        sc = *reinterpret_cast<const float4 *>(p.scale + dc4);
        sh = *reinterpret_cast<const float4 *>(p.shift + dc4);
        mu = *reinterpret_cast<const float4 *>(p.mean + dc4);
        is = *reinterpret_cast<const float4 *>(p.invstd + dc4);
        a1 = *reinterpret_cast<const float4 *>(p.c1 + dc4);
        a2 = *reinterpret_cast<const float4 *>(p.c2 + dc4);
    }

    // MFMA block ownership.  dX blocks (rb, cb): wave + 8 i.  dW blocks (nb, kb): from the last wave down.
    float xsc[DXPW], xsh[DXPW], xmu[DXPW], xis[DXPW], csum[DXPW], csq[DXPW];
#pragma unroll
    for (int i = 0; i < DXPW; ++i) {
        const int b = wave + FB_WAVES * i;
        const int col = (b % KBLK) * 32 + l31;
        xsc[i] = xsh[i] = xmu[i] = xis[i] = csum[i] = csq[i] = 0.f;
        if (masked && b < NDX && col < p.K) {
            xsc[i] = p.ascale[col]; xsh[i] = p.ashift[col]; xmu[i] = p.amean[col]; xis[i] = p.ainvstd[col];
        }
    }
    float wsc[DWPW], wsh[DWPW];
    f32x16 accW[DWPW];
#pragma unroll
    for (int i = 0; i < DWPW; ++i) {
        const int b = (FB_WAVES - 1 - wave) + FB_WAVES * i;
        const int col = (b % KBLK) * 32 + l31;
        wsc[i] = wsh[i] = 0.f;
        if (masked && b < NDW && col < p.K) { wsc[i] = p.ascale[col]; wsh[i] = p.ashift[col]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[i][r] = 0.f;
    }

    float4 gv[NBLK], zv[NBLK], xv[KBLK];
    uchar4 av[NBLK];
    // Loads are unconditional (clamped addresses, masked when consumed): a load under a divergent
    // branch makes the compiler wait for it at the end of the branch, which serialises the passes.
    const int dcc = n_ok ? dc4 : 0, xcc = k_ok ? xc4 : 0;
    auto issue = [&](int tile) {
        const int row0 = tile * FB_ROWS;
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            const int row = min(row0 + dr + DROWS * i, p.M - 1);
            if (POOLED) {                             // g / argk are per centroid
                const int cent = row / p.pool_k;
                gv[i] = *reinterpret_cast<const float4 *>(p.g + (size_t)cent * p.ldg + dcc);
                av[i] = *reinterpret_cast<const uchar4 *>(p.argk + (size_t)cent * p.N + dcc);
            } else {
                gv[i] = *reinterpret_cast<const float4 *>(p.g + (size_t)row * p.ldg + dcc);
            }
            zv[i] = *reinterpret_cast<const float4 *>(p.z + (size_t)row * p.ldz + dcc);
        }
#pragma unroll
        for (int i = 0; i < KBLK; ++i) {
            const int row = min(row0 + xr + XROWS * i, p.M - 1);
            xv[i] = *reinterpret_cast<const float4 *>(p.x + (size_t)row * p.ldx + xcc);
        }
    };
    if ((int)blockIdx.x < ntiles) issue(blockIdx.x);
    wt.store(sW, p.N, p.K, tid);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * FB_ROWS;
        __syncthreads();                              // readers of the previous tile are done (first pass: sW written)
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            const int r = dr + DROWS * i;
            const int row = row0 + r;
            float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.M && n_ok) {
                float4 g = gv[i];
                if (POOLED) {
                    const int kk = row % p.pool_k;
                    g.x = av[i].x == kk ? g.x : 0.f;
                    g.y = av[i].y == kk ? g.y : 0.f;
                    g.z = av[i].z == kk ? g.z : 0.f;
                    g.w = av[i].w == kk ? g.w : 0.f;
                }
                const float4 z = zv[i];
#define PN2_DZ(f) dv.f = sc.f * (((sc.f * z.f + sh.f) > 0.f ? g.f : 0.f) - a1.f - (z.f - mu.f) * is.f * a2.f)
                PN2_DZ(x); PN2_DZ(y); PN2_DZ(z); PN2_DZ(w);
#undef PN2_DZ
                dbs.x += dv.x; dbs.y += dv.y; dbs.z += dv.z; dbs.w += dv.w;
            }
            *reinterpret_cast<float4 *>(&sD[r * LDD + dc4]) = dv;
        }
#pragma unroll
        for (int i = 0; i < KBLK; ++i) {
            const bool ok = k_ok && row0 + xr + XROWS * i < p.M;
            *reinterpret_cast<float4 *>(&sZ[(xr + XROWS * i) * LDZ + xc4]) = ok ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);

        // ---- dX = dz * W : lane half h reduces n in [h NP/2, (h+1) NP/2)
        if (p.gp) {
#pragma unroll
            for (int i = 0; i < DXPW; ++i) {
                const int b = wave + FB_WAVES * i;
                if (b < NDX) {                        // uniform per wave
                    const int rb = b / KBLK, cb = b - rb * KBLK;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    const float *aRow = &sD[(rb * 32 + l31) * LDD + half * (NP / 2)];
                    const float *bCol = &sW[(half * (NP / 2)) * KP + cb * 32 + l31];
                    // operands of group q+1 are read before the four MFMAs of group q are issued (the compiler
                    // otherwise places every LDS read right in front of its consumer: read -> wait -> 2 MFMAs)
                    float4 a4 = *reinterpret_cast<const float4 *>(aRow);
                    float b0 = bCol[0], b1 = bCol[KP], b2 = bCol[2 * KP], b3 = bCol[3 * KP];
#pragma unroll
                    for (int q = 0; q < NP / 8; ++q) {
                        float4 a4n = a4;
                        float b0n = b0, b1n = b1, b2n = b2, b3n = b3;
                        if (q + 1 < NP / 8) {
                            a4n = *reinterpret_cast<const float4 *>(aRow + 4 * (q + 1));
                            b0n = bCol[(4 * q + 4) * KP]; b1n = bCol[(4 * q + 5) * KP];
                            b2n = bCol[(4 * q + 6) * KP]; b3n = bCol[(4 * q + 7) * KP];
                        }
                        __builtin_amdgcn_sched_barrier(0);         // keep the reads above the MFMAs
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b2, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b3, acc, 0, 0, 0);
                        a4 = a4n; b0 = b0n; b1 = b1n; b2 = b2n; b3 = b3n;
                    }
                    const int col = cb * 32 + l31;
                    float cs = 0.f, cq = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float v = acc[r];
                        if (masked) {
                            const float zp = sZ[rl * LDZ + col];
                            v = (xsc[i] * zp + xsh[i]) > 0.f ? v : 0.f;
                            cs += v;
                            cq += v * ((zp - xmu[i]) * xis[i]);
                        }
                        if (row0 + rl < p.M && col < p.K) pn2::store_rows(&p.gp[(size_t)(row0 + rl) * p.ldgp + col], v);
                    }
                    csum[i] += cs;
                    csq[i] += cq;
                }
            }
        }
        // ---- dW += dz^T * act(x) : reduction over the tile's rows, lane half h takes rows 32h .. 32h+31
#pragma unroll
        for (int i = 0; i < DWPW; ++i) {
            const int b = (FB_WAVES - 1 - wave) + FB_WAVES * i;
            if (b < NDW) {
                const int nb = b / KBLK, kb = b - nb * KBLK;
                const float *dBase = &sD[(32 * half) * LDD + nb * 32 + l31];
                const float *xBase = &sZ[(32 * half) * LDZ + kb * 32 + l31];
                // same read-ahead: the operands of steps t+4..t+7 are in flight while steps t..t+3 multiply
                float a[4], x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[u] = dBase[u * LDD]; x[u] = xBase[u * LDZ]; }
#pragma unroll
                for (int t = 0; t < 32; t += 4) {
                    float an[4], xn[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        an[u] = a[u]; xn[u] = x[u];
                        if (t + 4 < 32) { an[u] = dBase[(t + 4 + u) * LDD]; xn[u] = xBase[(t + 4 + u) * LDZ]; }
                    }
                    __builtin_amdgcn_sched_barrier(0);             // keep the reads above the MFMAs
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float xv = masked ? fmaxf(wsc[i] * x[u] + wsh[i], 0.f) : x[u];
                        accW[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], xv, accW[i], 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a[u] = an[u]; x[u] = xn[u]; }
                }
            }
        }
    }

    // ---- per-workgroup partials
    const int Kout = p.K + 1;
    float *slab = p.dw_partial + (size_t)blockIdx.x * p.N * Kout;
#pragma unroll
    for (int i = 0; i < DWPW; ++i) {
        const int b = (FB_WAVES - 1 - wave) + FB_WAVES * i;
        if (b < NDW) {
            const int nb = b / KBLK, kb = b - nb * KBLK;
            const int k = kb * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n < p.N && k < p.K) pn2::store_rows(&slab[(size_t)n * Kout + k], accW[i][r]);
            }
        }
    }
    __syncthreads();
    float *red = smem;                                // reuse the tile area: max(DROWS*NP, 4*KP) floats
    static_assert(RED <= FB_ROWS * (NP + 4) + FB_ROWS * (KP + 4), "reduction scratch does not fit");
    *reinterpret_cast<float4 *>(&red[dr * NP + dc4]) = dbs;
    __syncthreads();
    if (tid < p.N) {
        float t = 0.f;
        for (int i = 0; i < DROWS; ++i) t += red[i * NP + tid];
        slab[(size_t)tid * Kout + p.K] = t;
    }
    if (p.stat_partial) {
        __syncthreads();
        // dX block b = rb*KBLK + cb lives on wave b % 8, slot b / 8: combine rb = 0, 1 per column
#pragma unroll
        for (int i = 0; i < DXPW; ++i) {
            const int b = wave + FB_WAVES * i;
            if (b < NDX) {
                const int rb = b / KBLK, cb = b - rb * KBLK;
                float s = csum[i] + __shfl_xor(csum[i], 32);
                float q = csq[i] + __shfl_xor(csq[i], 32);
                if (half == 0) { red[(rb * 2 + 0) * KP + cb * 32 + l31] = s; red[(rb * 2 + 1) * KP + cb * 32 + l31] = q; }
            }
        }
        __syncthreads();
        for (int e = tid; e < 2 * p.K; e += FB_THREADS) {
            const int which = e / p.K, c = e - which * p.K;
            p.stat_partial[((size_t)blockIdx.x * 2 + which) * p.K + c] = red[(0 * 2 + which) * KP + c] + red[(1 * 2 + which) * KP + c];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Split-role form for the narrowest layers (K <= 32, N <= 64: the three layers of SA1, M = 524 288 rows).
// There a tile's MFMA work fits four waves (2 dX blocks + NBLK dW blocks) and the kernel above runs at
// memory time + MFMA time, because staging and multiplying alternate inside every workgroup.  Here the
// roles are split: waves 2..5 only STAGE (global loads -> dz / input tile of step t+1 into the other LDS
// buffer, prefetching t+2), waves 0,1,6,7 only MULTIPLY tile t; one barrier per tile.
// The roles are separate code paths, so each gets its own register allocation (78-104 VGPRs, 2-3 workgroups per CU).
template <int NBLK, bool POOLED>
__global__ __launch_bounds__(FB_THREADS, 4) void mlp_bwd_split_kernel(BwdArgs p)
{
    PN2_MAIN_BRANCH_PRIORITY();
    constexpr int NP = 32 * NBLK, KP = 32;
    constexpr int LDD = NP + 4, LDZ = KP + 4;
    constexpr int ST = 256;                           // staging threads (waves 2..5)
    constexpr int DROWS = ST / (NP / 4), XROWS = ST / (KP / 4);
    constexpr int DPASS = FB_ROWS / DROWS, XPASS = FB_ROWS / XROWS;
    constexpr int TILE = FB_ROWS * LDD + FB_ROWS * LDZ;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sW = smem + 2 * TILE;                      // [NP][KP]
    float *red = smem;                                // after the loop: [DROWS][NP] bias-gradient partials ...
    float *red2 = smem + DROWS * NP;                  // ... and [2][2][KP] statistics of the two dX waves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int ntiles = (p.M + FB_ROWS - 1) / FB_ROWS;
    const bool masked = p.ascale != nullptr;
    const int G = gridDim.x;
    const int t0 = blockIdx.x;                        // < ntiles (the grid never exceeds the tile count)
    const int Kout = p.K + 1;
    float *slab = p.dw_partial + (size_t)blockIdx.x * p.N * Kout;

    WeightTile<NP, KP, FB_THREADS> wt;
    wt.load(p.w, p.ldw, p.N, p.K, tid);

    // The two roles are two separate code paths (wave-uniform branch), each with its own registers; both
    // execute exactly the same sequence of barriers: A (weights in LDS), B (first tile staged), one per tile.
    if (wave >= 2 && wave < 6) {
        // ================================ stagers =================================================
        const int stid = tid - 128;
        const int dc4 = (stid % (NP / 4)) * 4, dr = stid / (NP / 4);
        const int xc4 = (stid % (KP / 4)) * 4, xr = stid / (KP / 4);
        const bool n_ok = dc4 < p.N, k_ok = xc4 < p.K;
        const int dcc = n_ok ? dc4 : 0, xcc = k_ok ? xc4 : 0;
        float4 sc, sh, mu, is, a1, a2;
        sc = sh = mu = is = a1 = a2 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 dbs = sc;
        if (n_ok) {
            sc = *reinterpret_cast<const float4 *>(p.scale + dc4);
            sh = *reinterpret_cast<const float4 *>(p.shift + dc4);
            mu = *reinterpret_cast<const float4 *>(p.mean + dc4);
            is = *reinterpret_cast<const float4 *>(p.invstd + dc4);
            a1 = *reinterpret_cast<const float4 *>(p.c1 + dc4);
            a2 = *reinterpret_cast<const float4 *>(p.c2 + dc4);
        }
        float4 gv[DPASS], zv[DPASS], xv[XPASS];
        uchar4 av[DPASS];
        auto issue = [&](int tile) {
            const int row0 = tile * FB_ROWS;
#pragma unroll
            for (int i = 0; i < DPASS; ++i) {
                const int row = min(row0 + dr + DROWS * i, p.M - 1);
                if (POOLED) {
                    const int cent = row / p.pool_k;
                    gv[i] = *reinterpret_cast<const float4 *>(p.g + (size_t)cent * p.ldg + dcc);
                    av[i] = *reinterpret_cast<const uchar4 *>(p.argk + (size_t)cent * p.N + dcc);
                } else {
                    gv[i] = *reinterpret_cast<const float4 *>(p.g + (size_t)row * p.ldg + dcc);
                }
                zv[i] = *reinterpret_cast<const float4 *>(p.z + (size_t)row * p.ldz + dcc);
            }
#pragma unroll
            for (int i = 0; i < XPASS; ++i) {
                const int row = min(row0 + xr + XROWS * i, p.M - 1);
                xv[i] = *reinterpret_cast<const float4 *>(p.x + (size_t)row * p.ldx + xcc);
            }
        };
        auto commit = [&](int tile, int buf) {
            float *sD = smem + buf * TILE, *sZ = sD + FB_ROWS * LDD;
            const int row0 = tile * FB_ROWS;
#pragma unroll
            for (int i = 0; i < DPASS; ++i) {
                const int r = dr + DROWS * i;
                const int row = row0 + r;
                float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < p.M && n_ok) {
                    float4 g = gv[i];
                    if (POOLED) {
                        const int kk = row % p.pool_k;
                        g.x = av[i].x == kk ? g.x : 0.f;
                        g.y = av[i].y == kk ? g.y : 0.f;
                        g.z = av[i].z == kk ? g.z : 0.f;
                        g.w = av[i].w == kk ? g.w : 0.f;
                    }
                    const float4 z = zv[i];
#define PN2_DZ(f) dv.f = sc.f * (((sc.f * z.f + sh.f) > 0.f ? g.f : 0.f) - a1.f - (z.f - mu.f) * is.f * a2.f)
                    PN2_DZ(x); PN2_DZ(y); PN2_DZ(z); PN2_DZ(w);
#undef PN2_DZ
                    dbs.x += dv.x; dbs.y += dv.y; dbs.z += dv.z; dbs.w += dv.w;
                }
                *reinterpret_cast<float4 *>(&sD[r * LDD + dc4]) = dv;
            }
#pragma unroll
            for (int i = 0; i < XPASS; ++i) {
                const bool ok = k_ok && row0 + xr + XROWS * i < p.M;
                *reinterpret_cast<float4 *>(&sZ[(xr + XROWS * i) * LDZ + xc4]) = ok ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        issue(t0);
        wt.store(sW, p.N, p.K, tid);
        __syncthreads();                              // A
        commit(t0, 0);
        if (t0 + G < ntiles) issue(t0 + G);
        __syncthreads();                              // B
        int buf = 0;
        for (int tile = t0; tile < ntiles; tile += G, buf ^= 1) {
            if (tile + G < ntiles) {
                commit(tile + G, buf ^ 1);
                if (tile + 2 * G < ntiles) issue(tile + 2 * G);
            }
            __syncthreads();                          // tile
        }
        *reinterpret_cast<float4 *>(&red[dr * NP + dc4]) = dbs;      // tile buffers are free now
    } else {
        // ================================ multipliers ================================================
        // waves 0,1: the dX blocks (rows 32*wave..); waves 7 (and 6): the dW blocks (columns 32*(7-wave)..)
        const bool dx_wave = wave < 2, dw_wave = wave >= 8 - NBLK;
        const int col = l31;
        float xsc = 0.f, xsh = 0.f, xmu = 0.f, xis = 0.f, csum = 0.f, csq = 0.f;
        if (masked && col < p.K) { xsc = p.ascale[col]; xsh = p.ashift[col]; xmu = p.amean[col]; xis = p.ainvstd[col]; }
        f32x16 accW;
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[r] = 0.f;
        asm volatile("" ::"v"(xsc), "v"(xsh), "v"(xmu), "v"(xis));   // arrived before the loop (in-order vmcnt)
        wt.store(sW, p.N, p.K, tid);
        __syncthreads();                              // A
        __syncthreads();                              // B
        int buf = 0;
        for (int tile = t0; tile < ntiles; tile += G, buf ^= 1) {
            const float *sD = smem + buf * TILE, *sZ = sD + FB_ROWS * LDD;
            const int row0 = tile * FB_ROWS;
            if (dx_wave && p.gp) {
                const int rb = wave;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float *aRow = &sD[(rb * 32 + l31) * LDD + half * (NP / 2)];
                const float *bCol = &sW[(half * (NP / 2)) * KP + l31];
                float4 a4 = *reinterpret_cast<const float4 *>(aRow);
                float b0 = bCol[0], b1 = bCol[KP], b2 = bCol[2 * KP], b3 = bCol[3 * KP];
#pragma unroll
                for (int q = 0; q < NP / 8; ++q) {
                    float4 a4n = a4;
                    float b0n = b0, b1n = b1, b2n = b2, b3n = b3;
                    if (q + 1 < NP / 8) {
                        a4n = *reinterpret_cast<const float4 *>(aRow + 4 * (q + 1));
                        b0n = bCol[(4 * q + 4) * KP]; b1n = bCol[(4 * q + 5) * KP];
                        b2n = bCol[(4 * q + 6) * KP]; b3n = bCol[(4 * q + 7) * KP];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b3, acc, 0, 0, 0);
                    a4 = a4n; b0 = b0n; b1 = b1n; b2 = b2n; b3 = b3n;
                }
                float cs = 0.f, cq = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = acc[r];
                    if (masked) {
                        const float zp = sZ[rl * LDZ + col];
                        v = (xsc * zp + xsh) > 0.f ? v : 0.f;
                        cs += v;
                        cq += v * ((zp - xmu) * xis);
                    }
                    if (row0 + rl < p.M && col < p.K) pn2::store_rows(&p.gp[(size_t)(row0 + rl) * p.ldgp + col], v);
                }
                csum += cs;
                csq += cq;
            }
            if (dw_wave) {
                const int nb = 7 - wave;
                const float *dBase = &sD[(32 * half) * LDD + nb * 32 + l31];
                const float *xBase = &sZ[(32 * half) * LDZ + l31];
                float a[4], x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[u] = dBase[u * LDD]; x[u] = xBase[u * LDZ]; }
#pragma unroll
                for (int t = 0; t < 32; t += 4) {
                    float an[4], xn[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        an[u] = a[u]; xn[u] = x[u];
                        if (t + 4 < 32) { an[u] = dBase[(t + 4 + u) * LDD]; xn[u] = xBase[(t + 4 + u) * LDZ]; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float xv2 = masked ? fmaxf(xsc * x[u] + xsh, 0.f) : x[u];
                        accW = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], xv2, accW, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a[u] = an[u]; x[u] = xn[u]; }
                }
            }
            __syncthreads();                          // tile
        }
        if (dw_wave) {
            const int nb = 7 - wave;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n < p.N && col < p.K) pn2::store_rows(&slab[(size_t)n * Kout + col], accW[r]);
            }
        }
        if (dx_wave) {
            const float s = csum + __shfl_xor(csum, 32), q = csq + __shfl_xor(csq, 32);
            if (half == 0) { red2[(wave * 2 + 0) * KP + l31] = s; red2[(wave * 2 + 1) * KP + l31] = q; }
        }
    }
    __syncthreads();
    if (tid < p.N) {
        float t = 0.f;
        for (int i = 0; i < DROWS; ++i) t += red[i * NP + tid];
        slab[(size_t)tid * Kout + p.K] = t;
    }
    if (p.stat_partial) {
        for (int e = tid; e < 2 * p.K; e += FB_THREADS) {
            const int which = e / p.K, c = e - which * p.K;
            p.stat_partial[((size_t)blockIdx.x * 2 + which) * p.K + c] = red2[(0 * 2 + which) * KP + c] + red2[(1 * 2 + which) * KP + c];
        }
    }
}

template <int NBLK>
constexpr int fs_lds_bytes() { return (2 * (FB_ROWS * (32 * NBLK + 4) + FB_ROWS * 36) + 32 * NBLK * 32) * 4; }

// ---------------------------------------------------------------------------------------------------------
// General split-role form: NMW multiplier waves (all dX / dW blocks of a TR-row tile spread over them) and NSW
// staging waves, double-buffered tiles, one barrier per tile.  Used for the 128 x 128 layers (TR = 32: the weight
// tile plus two 32-row tile buffers fit the LDS; 8 + 4 waves): the multiplier waves issue MFMAs back to back
// instead of alternating with the staging code.
template <int NBLK, int KBLK, bool POOLED, int TR, int NMW, int NSW>
__global__ __launch_bounds__(64 * (NMW + NSW), 2) void mlp_bwd_split2_kernel(BwdArgs p)
{
    PN2_MAIN_BRANCH_PRIORITY();
    constexpr int THREADS = 64 * (NMW + NSW), ST = 64 * NSW;
    constexpr int NP = 32 * NBLK, KP = 32 * KBLK;
    constexpr int LDD = NP + 4, LDZ = KP + 4;
    constexpr int RB = TR / 32;
    constexpr int NDX = RB * KBLK, NDW = NBLK * KBLK;
    constexpr int DXPW = (NDX + NMW - 1) / NMW;
    // dW blocks: the waves without a dX block (NUP of them) take DWHI each, the dX waves share the rest (DWLO each),
    // so that the two multiplier waves of a SIMD carry about the same number of MFMAs per tile
    constexpr int NUP = NMW > NDX ? NMW - NDX : 0;
    constexpr int UNITS = NDW + NDX * (NP / TR);       // work in dW-block units (a dX block costs NP/TR of them)
    constexpr int TARGET = (UNITS + NMW - 1) / NMW;
    constexpr int DWHI = NUP ? (TARGET < NDW / NUP ? TARGET : NDW / NUP) : 0;
    constexpr int DWLO = NUP ? (NDW - NUP * DWHI + NDX - 1) / (NDX ? NDX : 1) : (NDW + NMW - 1) / NMW;
    constexpr int DWPW = DWHI > DWLO ? DWHI : DWLO;
    constexpr int DROWS = ST / (NP / 4), XROWS = ST / (KP / 4);
    constexpr int DPASS = TR / DROWS, XPASS = TR / XROWS;
    static_assert(DPASS * DROWS == TR && XPASS * XROWS == TR, "staging rows must divide the tile");
    constexpr int TILE = TR * LDD + TR * LDZ;
    static_assert(ST * 4 + RB * 2 * KP <= 2 * TILE, "reduction scratch does not fit");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sW = smem + 2 * TILE;                      // [NP][KP]
    float *red = smem;                                // after the loop: [DROWS][NP] bias-gradient partials ...
    float *red2 = smem + ST * 4;                      // ... and [RB][2][KP] statistics of the dX blocks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int ntiles = (p.M + TR - 1) / TR;
    const bool masked = p.ascale != nullptr;
    const int G = gridDim.x;
    const int t0 = blockIdx.x;                        // < ntiles (the grid never exceeds the tile count)
    // Layers with more than KP inputs: blockIdx.y selects a block of KP input columns; the workgroup then solves the
    // same problem on the column slice (x, W, dX, dW and the statistics of the layer below are all per input column;
    // g and z are read once per block).
    const int k0 = (int)blockIdx.y * KP, Ktot = p.K;
    p.K = min(KP, Ktot - k0);
    p.x += k0;
    p.w += k0;
    if (p.gp) p.gp += k0;
    if (masked) { p.ascale += k0; p.ashift += k0; p.amean += k0; p.ainvstd += k0; }
    const int Kout = Ktot + 1;
    float *slab = p.dw_partial + (size_t)blockIdx.x * p.N * Kout + k0;

    WeightTile<NP, KP, THREADS> wt;
    wt.load(p.w, p.ldw, p.N, p.K, tid);

    if (wave >= NMW) {
        // ================================ stagers =================================================
        const int stid = tid - 64 * NMW;
        const int dc4 = (stid % (NP / 4)) * 4, dr = stid / (NP / 4);
        const int xc4 = (stid % (KP / 4)) * 4, xr = stid / (KP / 4);
        const bool n_ok = dc4 < p.N, k_ok = xc4 < p.K;
        const int dcc = n_ok ? dc4 : 0, xcc = k_ok ? xc4 : 0;
        float4 sc, sh, mu, is, a1, a2;
        sc = sh = mu = is = a1 = a2 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 dbs = sc;
        if (n_ok) {
            sc = *reinterpret_cast<const float4 *>(p.scale + dc4);
            sh = *reinterpret_cast<const float4 *>(p.shift + dc4);
            mu = *reinterpret_cast<const float4 *>(p.mean + dc4);
            is = *reinterpret_cast<const float4 *>(p.invstd + dc4);
            a1 = *reinterpret_cast<const float4 *>(p.c1 + dc4);
            a2 = *reinterpret_cast<const float4 *>(p.c2 + dc4);
        }
        float4 gv[DPASS], zv[DPASS], xv[XPASS];
        uchar4 av[DPASS];
        auto issue = [&](int tile) {
            const int row0 = tile * TR;
#pragma unroll
            for (int i = 0; i < DPASS; ++i) {
                const int row = min(row0 + dr + DROWS * i, p.M - 1);
                if (POOLED) {
                    const int cent = row / p.pool_k;
                    gv[i] = *reinterpret_cast<const float4 *>(p.g + (size_t)cent * p.ldg + dcc);
                    av[i] = *reinterpret_cast<const uchar4 *>(p.argk + (size_t)cent * p.N + dcc);
                } else {
                    gv[i] = *reinterpret_cast<const float4 *>(p.g + (size_t)row * p.ldg + dcc);
                }
                zv[i] = *reinterpret_cast<const float4 *>(p.z + (size_t)row * p.ldz + dcc);
            }
#pragma unroll
            for (int i = 0; i < XPASS; ++i) {
                const int row = min(row0 + xr + XROWS * i, p.M - 1);
                xv[i] = *reinterpret_cast<const float4 *>(p.x + (size_t)row * p.ldx + xcc);
            }
        };
        auto commit = [&](int tile, int buf) {
            float *sD = smem + buf * TILE, *sZ = sD + TR * LDD;
            const int row0 = tile * TR;
#pragma unroll
            for (int i = 0; i < DPASS; ++i) {
                const int r = dr + DROWS * i;
                const int row = row0 + r;
                float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < p.M && n_ok) {
                    float4 g = gv[i];
                    if (POOLED) {
                        const int kk = row % p.pool_k;
                        g.x = av[i].x == kk ? g.x : 0.f;
                        g.y = av[i].y == kk ? g.y : 0.f;
                        g.z = av[i].z == kk ? g.z : 0.f;
                        g.w = av[i].w == kk ? g.w : 0.f;
                    }
                    const float4 z = zv[i];
#define PN2_DZ(f) dv.f = sc.f * (((sc.f * z.f + sh.f) > 0.f ? g.f : 0.f) - a1.f - (z.f - mu.f) * is.f * a2.f)
                    PN2_DZ(x); PN2_DZ(y); PN2_DZ(z); PN2_DZ(w);
#undef PN2_DZ
                    dbs.x += dv.x; dbs.y += dv.y; dbs.z += dv.z; dbs.w += dv.w;
                }
                *reinterpret_cast<float4 *>(&sD[r * LDD + dc4]) = dv;
            }
#pragma unroll
            for (int i = 0; i < XPASS; ++i) {
                const bool ok = k_ok && row0 + xr + XROWS * i < p.M;
                *reinterpret_cast<float4 *>(&sZ[(xr + XROWS * i) * LDZ + xc4]) = ok ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        issue(t0);
        wt.store(sW, p.N, p.K, tid);
        __syncthreads();                              // A
        commit(t0, 0);
        if (t0 + G < ntiles) issue(t0 + G);
        __syncthreads();                              // B
        int buf = 0;
        for (int tile = t0; tile < ntiles; tile += G, buf ^= 1) {
            if (tile + G < ntiles) {
                commit(tile + G, buf ^ 1);
                if (tile + 2 * G < ntiles) issue(tile + 2 * G);
            }
            __syncthreads();                          // tile
        }
        *reinterpret_cast<float4 *>(&red[dr * NP + dc4]) = dbs;      // tile buffers are free now
    } else {
        // ================================ multipliers ================================================
        // dX block b = rb*KBLK + cb on wave b % NMW (slot b / NMW); dW blocks: see DWHI / DWLO
        // SHARED (the 128 x 128 form: four waves without a dX block, four dW blocks each): a wave takes ONE block of input
        // columns (kb) and all NBLK blocks of dz columns, so the activated input operand is read and activated once per
        // row and shared by its four MFMAs (before: one dz block per wave, the input re-read and re-activated per block
        // and by all four waves -- every vector instruction of a multiplier adds to the SIMD's time)
        constexpr bool SHARED = NUP > 0 && NUP == KBLK && DWHI == NBLK && DWLO == 0;
        auto dw_block = [&](int i) -> int {
            if (NUP == 0) { const int b = (NMW - 1 - wave) + NMW * i; return b < NDW ? b : -1; }
            if (SHARED) return (wave >= NDX && i < NBLK) ? i * KBLK + (wave - NDX) : -1;
            if (wave >= NDX) return i < DWHI ? (wave - NDX) * DWHI + i : -1;
            const int b = NUP * DWHI + wave * DWLO + i;
            return (i < DWLO && b < NDW) ? b : -1;
        };
        float xsc[DXPW], xsh[DXPW], xmu[DXPW], xis[DXPW], csum[DXPW], csq[DXPW];
#pragma unroll
        for (int i = 0; i < DXPW; ++i) {
            const int b = wave + NMW * i;
            const int col = (b % KBLK) * 32 + l31;
            xsc[i] = xsh[i] = xmu[i] = xis[i] = csum[i] = csq[i] = 0.f;
            if (masked && b < NDX && col < p.K) {
                xsc[i] = p.ascale[col]; xsh[i] = p.ashift[col]; xmu[i] = p.amean[col]; xis[i] = p.ainvstd[col];
            }
        }
        float wsc[DWPW], wsh[DWPW];
        f32x16 accW[DWPW];
#pragma unroll
        for (int i = 0; i < DWPW; ++i) {
            const int b = dw_block(i);
            const int col = ((b < 0 ? 0 : b) % KBLK) * 32 + l31;
            wsc[i] = wsh[i] = 0.f;
            if (masked && b >= 0 && col < p.K) { wsc[i] = p.ascale[col]; wsh[i] = p.ashift[col]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[i][r] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < DXPW; ++i) asm volatile("" ::"v"(xsc[i]), "v"(xsh[i]), "v"(xmu[i]), "v"(xis[i]));
#pragma unroll
        for (int i = 0; i < DWPW; ++i) asm volatile("" ::"v"(wsc[i]), "v"(wsh[i]));
        wt.store(sW, p.N, p.K, tid);
        __syncthreads();                              // A
        __syncthreads();                              // B
        int buf = 0;
        for (int tile = t0; tile < ntiles; tile += G, buf ^= 1) {
            const float *sD = smem + buf * TILE, *sZ = sD + TR * LDD;
            const int row0 = tile * TR;
            if (p.gp) {
#pragma unroll
                for (int i = 0; i < DXPW; ++i) {
                    const int b = wave + NMW * i;
                    if (b < NDX) {
                        const int rb = b / KBLK, cb = b - rb * KBLK;
                        f32x16 acc;
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                        const float *aRow = &sD[(rb * 32 + l31) * LDD + half * (NP / 2)];
                        const float *bCol = &sW[(half * (NP / 2)) * KP + cb * 32 + l31];
                        float4 a4 = *reinterpret_cast<const float4 *>(aRow);
                        float b0 = bCol[0], b1 = bCol[KP], b2 = bCol[2 * KP], b3 = bCol[3 * KP];
#pragma unroll
                        for (int q = 0; q < NP / 8; ++q) {
                            float4 a4n = a4;
                            float b0n = b0, b1n = b1, b2n = b2, b3n = b3;
                            if (q + 1 < NP / 8) {
                                a4n = *reinterpret_cast<const float4 *>(aRow + 4 * (q + 1));
                                b0n = bCol[(4 * q + 4) * KP]; b1n = bCol[(4 * q + 5) * KP];
                                b2n = bCol[(4 * q + 6) * KP]; b3n = bCol[(4 * q + 7) * KP];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b2, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b3, acc, 0, 0, 0);
                            a4 = a4n; b0 = b0n; b1 = b1n; b2 = b2n; b3 = b3n;
                        }
                        const int col = cb * 32 + l31;
                        float cs = 0.f, cq = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            float v = acc[r];
                            if (masked) {
                                const float zp = sZ[rl * LDZ + col];
                                v = (xsc[i] * zp + xsh[i]) > 0.f ? v : 0.f;
                                cs += v;
                                cq += v * ((zp - xmu[i]) * xis[i]);
                            }
                            if (row0 + rl < p.M && col < p.K) pn2::store_rows(&p.gp[(size_t)(row0 + rl) * p.ldgp + col], v);
                        }
                        csum[i] += cs;
                        csq[i] += cq;
                    }
                }
            }
            if (SHARED) {
                if (wave >= NDX) {
                    constexpr int HR = TR / 2;                       // rows per lane half
                    const float *dBase = &sD[(HR * half) * LDD + l31];
                    const float *xBase = &sZ[(HR * half) * LDZ + (wave - NDX) * 32 + l31];
                    float xr = xBase[0], a[DWPW];
#pragma unroll
                    for (int i = 0; i < DWPW; ++i) a[i] = dBase[i * 32];
#pragma unroll
                    for (int t = 0; t < HR; ++t) {
                        float xn = xr, an[DWPW];
#pragma unroll
                        for (int i = 0; i < DWPW; ++i) an[i] = a[i];
                        if (t + 1 < HR) {
                            xn = xBase[(t + 1) * LDZ];
#pragma unroll
                            for (int i = 0; i < DWPW; ++i) an[i] = dBase[(t + 1) * LDD + i * 32];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const float xv2 = masked ? fmaxf(wsc[0] * xr + wsh[0], 0.f) : xr;
#pragma unroll
                        for (int i = 0; i < DWPW; ++i) accW[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], xv2, accW[i], 0, 0, 0);
                        xr = xn;
#pragma unroll
                        for (int i = 0; i < DWPW; ++i) a[i] = an[i];
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < DWPW; ++i) {
                const int b = dw_block(i);
                if (b >= 0) {
                    const int nb = b / KBLK, kb = b - nb * KBLK;
                    constexpr int HR = TR / 2;                       // rows per lane half
                    const float *dBase = &sD[(HR * half) * LDD + nb * 32 + l31];
                    const float *xBase = &sZ[(HR * half) * LDZ + kb * 32 + l31];
                    float a[4], x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a[u] = dBase[u * LDD]; x[u] = xBase[u * LDZ]; }
#pragma unroll
                    for (int t = 0; t < HR; t += 4) {
                        float an[4], xn[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            an[u] = a[u]; xn[u] = x[u];
                            if (t + 4 < HR) { an[u] = dBase[(t + 4 + u) * LDD]; xn[u] = xBase[(t + 4 + u) * LDZ]; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float xv2 = masked ? fmaxf(wsc[i] * x[u] + wsh[i], 0.f) : x[u];
                            accW[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], xv2, accW[i], 0, 0, 0);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { a[u] = an[u]; x[u] = xn[u]; }
                    }
                }
            }
            __syncthreads();                          // tile
        }
#pragma unroll
        for (int i = 0; i < DWPW; ++i) {
            const int b = dw_block(i);
            if (b >= 0) {
                const int nb = b / KBLK, kb = b - nb * KBLK;
                const int k = kb * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (n < p.N && k < p.K) pn2::store_rows(&slab[(size_t)n * Kout + k], accW[i][r]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < DXPW; ++i) {
            const int b = wave + NMW * i;
            if (b < NDX) {
                const int rb = b / KBLK, cb = b - rb * KBLK;
                const float s = csum[i] + __shfl_xor(csum[i], 32), q = csq[i] + __shfl_xor(csq[i], 32);
                if (half == 0) { red2[(rb * 2 + 0) * KP + cb * 32 + l31] = s; red2[(rb * 2 + 1) * KP + cb * 32 + l31] = q; }
            }
        }
    }
    __syncthreads();
    if (tid < p.N && blockIdx.y == 0) {
        float t = 0.f;
        for (int i = 0; i < DROWS; ++i) t += red[i * NP + tid];
        slab[(size_t)tid * Kout + Ktot] = t;
    }
    if (p.stat_partial) {
        for (int e = tid; e < 2 * p.K; e += THREADS) {
            const int which = e / p.K, c = e - which * p.K;
            float v = 0.f;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) v += red2[(rb * 2 + which) * KP + c];
            p.stat_partial[((size_t)blockIdx.x * 2 + which) * Ktot + k0 + c] = v;
        }
    }
}

template <int NBLK, int KBLK, int TR>
constexpr int fs2_lds_bytes() { return (2 * (TR * (32 * NBLK + 4) + TR * (32 * KBLK + 4)) + 32 * NBLK * 32 * KBLK) * 4; }

template <int NBLK, int KBLK>
constexpr int fb_lds_bytes() { return (FB_ROWS * (32 * NBLK + 4) + FB_ROWS * (32 * KBLK + 4) + 32 * NBLK * 32 * KBLK) * 4; }

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int fb_blocks(int n) { return n <= 32 ? 1 : (n <= 64 ? 2 : 4); }

// Resident workgroups per CU of one instantiation (registers and LDS both count), asked of the runtime
// once; also raises the dynamic-LDS limit of the kernel (> 64 KB has to be requested).
template <int NBLK, int KBLK, bool POOLED>
int fb_resident_of();

template <int NBLK, int KBLK>
int fb_resident()
{
    return fb_resident_of<NBLK, KBLK, false>() < fb_resident_of<NBLK, KBLK, true>() ? fb_resident_of<NBLK, KBLK, false>()
                                                                                   : fb_resident_of<NBLK, KBLK, true>();
}

template <int NBLK, int KBLK, bool POOLED>
int fb_resident_of()
{
    static pn2::PerDevice cached;                     // resident workgroups per CU on this device (attribute set with it)
    if (cached.get() >= 0) return cached.get();
    constexpr int lds = fb_lds_bytes<NBLK, KBLK>();
    const void *fn = reinterpret_cast<const void *>(&mlp_bwd_fused_kernel<NBLK, KBLK, POOLED>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mlp_bwd_fused_kernel<NBLK, KBLK, POOLED>, FB_THREADS, lds) != hipSuccess) return 0;
    cached.set(n < 1 ? 0 : (n > 4 ? 4 : n));
    return cached.get();
}


template <int NBLK, bool POOLED>
int fs_resident_of()
{
    static pn2::PerDevice cached;
    if (cached.get() >= 0) return cached.get();
    constexpr int lds = fs_lds_bytes<NBLK>();
    const void *fn = reinterpret_cast<const void *>(&mlp_bwd_split_kernel<NBLK, POOLED>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mlp_bwd_split_kernel<NBLK, POOLED>, FB_THREADS, lds) != hipSuccess) return 0;
    cached.set(n < 1 ? 0 : (n > 4 ? 4 : n));
    return cached.get();
}

// General split-role kernel: which shapes use it and with which tile (PN2_TUNE_FB_SPLIT2=0: off; bit 1 = the
// 128 x 128 layers with 32-row tiles, bit 2 = the 64..128-channel layers of SA2 with 64-row tiles).
int fs2_tile_rows(int nblk, int kblk)
{
    const int m = pn2::tune_get("fb_split2", 1);     // SA2 shapes measured: 72 / 40 / 57 us against 70 / 42 / 60 us: off
    if (nblk == 4 && kblk == 4) return (m & 1) ? 32 : 0;
    if ((nblk == 2 && kblk == 2) || (nblk == 4 && kblk == 2) || (nblk == 2 && kblk == 4)) return (m & 2) ? 64 : 0;
    return 0;
}

template <int NBLK, int KBLK, bool POOLED, int TR>
int fs2_prepare()
{
    static pn2::PerDevice cached;
    if (cached.get() >= 0) return cached.get();
    constexpr int lds = fs2_lds_bytes<NBLK, KBLK, TR>();
    const void *fn = reinterpret_cast<const void *>(&mlp_bwd_split2_kernel<NBLK, KBLK, POOLED, TR, 8, 4>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mlp_bwd_split2_kernel<NBLK, KBLK, POOLED, TR, 8, 4>, 768, lds) != hipSuccess) return 0;
    cached.set(n < 1 ? 0 : 1);                        // one 12-wave workgroup per CU
    return cached.get();
}

template <int NBLK, int KBLK, int TR>
int fs2_resident() { const int a = fs2_prepare<NBLK, KBLK, false, TR>(), b = fs2_prepare<NBLK, KBLK, true, TR>(); return a < b ? a : b; }

int fs2_resident_rt(int nblk, int kblk)
{
    if (nblk == 4 && kblk == 4) return fs2_resident<4, 4, 32>();
    if (nblk == 2 && kblk == 2) return fs2_resident<2, 2, 64>();
    if (nblk == 4 && kblk == 2) return fs2_resident<4, 2, 64>();
    if (nblk == 2 && kblk == 4) return fs2_resident<2, 4, 64>();
    return 0;
}

template <int NBLK, int KBLK, int TR>
int fs2_launch(const BwdArgs &a, int grid, hipStream_t stream, int kblocks = 1)
{
    if (a.argk)
        hipLaunchKernelGGL((mlp_bwd_split2_kernel<NBLK, KBLK, true, TR, 8, 4>), dim3(grid, kblocks), dim3(768), (fs2_lds_bytes<NBLK, KBLK, TR>()), stream, a);
    else
        hipLaunchKernelGGL((mlp_bwd_split2_kernel<NBLK, KBLK, false, TR, 8, 4>), dim3(grid, kblocks), dim3(768), (fs2_lds_bytes<NBLK, KBLK, TR>()), stream, a);
    return PN2_LAUNCH_RC();
}

// split-role kernel: K <= 32 and N <= 64 (PN2_TUNE_FB_SPLIT=0 switches it off)
bool fs_applies(int nblk, int kblk) { return kblk == 1 && nblk <= 2 && pn2::tune_get("fb_split", 1) != 0; }

int fs_resident_rt(int nblk)
{
    const int a = nblk == 1 ? fs_resident_of<1, false>() : fs_resident_of<2, false>();
    const int b = nblk == 1 ? fs_resident_of<1, true>() : fs_resident_of<2, true>();
    return a < b ? a : b;
}

int fb_resident_rt(int nblk, int kblk)
{
#define PN2_FB(NB, KB) if (nblk == NB && kblk == KB) return fb_resident<NB, KB>()
    PN2_FB(1, 1); PN2_FB(1, 2); PN2_FB(1, 4);
    PN2_FB(2, 1); PN2_FB(2, 2); PN2_FB(2, 4);
    PN2_FB(4, 1); PN2_FB(4, 2); PN2_FB(4, 4);
#undef PN2_FB
    return 0;
}

template <int NBLK, int KBLK>
int fb_launch(const BwdArgs &a, int grid, hipStream_t stream)
{
    if (a.argk)
        hipLaunchKernelGGL((mlp_bwd_fused_kernel<NBLK, KBLK, true>), dim3(grid), dim3(FB_THREADS), (fb_lds_bytes<NBLK, KBLK>()), stream, a);
    else
        hipLaunchKernelGGL((mlp_bwd_fused_kernel<NBLK, KBLK, false>), dim3(grid), dim3(FB_THREADS), (fb_lds_bytes<NBLK, KBLK>()), stream, a);
    return PN2_LAUNCH_RC();
}

}  // namespace

// Number of partial slabs (= workgroups) pn2_mlp_bwd_layer uses, 0 if the shape is not covered.
PN2_EXPORT int pn2_mlp_bwd_layer_partials(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0 || N > 128 || (N & 3) || (K & 3)) return 0;
    // more than 128 inputs: 128-column blocks of the 128 x 128 split-role kernel (65..128 outputs only)
    if (K > 128 && !(N > 64 && fs2_tile_rows(4, 4) && pn2::tune_get("fb_kblocks", 1))) return 0;
    const int nblk_ = fb_blocks(N), kblk_ = fb_blocks(K > 128 ? 128 : K);
    int per_cu = fs_applies(nblk_, kblk_) ? fs_resident_rt(nblk_) : fb_resident_rt(nblk_, kblk_);
    int tile_rows = FB_ROWS;
    if (fs2_tile_rows(nblk_, kblk_)) {
        per_cu = fs2_resident_rt(nblk_, kblk_);
        tile_rows = fs2_tile_rows(nblk_, kblk_);
    }
    if (per_cu < 1) return 0;
    int cus = 256;
    {
        static pn2::PerDevice cached_cus;
        if (cached_cus.get() < 0) {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) == hipSuccess &&
                hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
                cached_cus.set(n);
            else
                cached_cus.set(256);
        }
        cus = cached_cus.get();
    }
    const int tiles = (M + tile_rows - 1) / tile_rows;
    int wgs = cus * pn2::tune_get("mlp_fb_wgs", per_cu);
    if (K > 128) {
        // column-blocked launch: the blocks' workgroups share the CUs, and every workgroup stages its weight slice
        // (64 KB) before its first tile -- keep at least two tiles per workgroup
        const int kb = (K + 127) / 128;
        int per_block = wgs / kb;
        const int by_tiles = tiles / pn2::tune_get("fb_kb_min_tiles", 2);
        if (per_block > by_tiles) per_block = by_tiles;
        if (per_block < 1) per_block = 1;
        wgs = per_block;
    }
    return tiles < wgs ? tiles : wgs;
}

PN2_EXPORT int pn2_mlp_bwd_layer(const float *g, int ldg, const float *z, int ldz, const unsigned char *argk, int pool_k,
                                 const float *scale, const float *shift, const float *mean, const float *invstd,
                                 const float *c1, const float *c2, const float *w, int ldw, const float *x, int ldx,
                                 const float *ascale, const float *ashift, const float *amean, const float *ainvstd,
                                 float *gp, int ldgp, float *stat_partial, float *dw_partial, float *dw, float *db,
                                 float *dgamma_below, float *dbeta_below, float *c1_below, float *c2_below, int M, int N,
                                 int K, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(g); PN2_REQUIRE_PTR(z); PN2_REQUIRE_PTR(scale); PN2_REQUIRE_PTR(shift); PN2_REQUIRE_PTR(mean);
    PN2_REQUIRE_PTR(invstd); PN2_REQUIRE_PTR(c1); PN2_REQUIRE_PTR(c2); PN2_REQUIRE_PTR(w); PN2_REQUIRE_PTR(x);
    PN2_REQUIRE_PTR(dw_partial);
    if (!dw && c1_below) return PN2_ERR_NULL;        // dw NULL: slabs only (summed later, pn2_mlp_dw_reduce_many)
    if (M <= 0 || N <= 0 || K <= 0 || (argk && pool_k <= 0) || ldw < K || ldx < K || ldz < N) return PN2_ERR_SHAPE;
    const bool masked = ascale != nullptr;
    if (masked && (!ashift || !amean || !ainvstd)) return PN2_ERR_NULL;
    if (stat_partial && (!masked || !gp)) return PN2_ERR_NULL;
    if (c1_below && (!stat_partial || !c2_below)) return PN2_ERR_NULL;
    const int P = pn2_mlp_bwd_layer_partials(M, N, K);
    if (P == 0) return PN2_ERR_UNSUPPORTED;
    bool ok = (ldg % 4 == 0) && (ldz % 4 == 0) && (ldw % 4 == 0) && (ldx % 4 == 0) && aligned16(g) && aligned16(z) &&
              aligned16(w) && aligned16(x) && aligned16(scale) && aligned16(shift) && aligned16(mean) &&
              aligned16(invstd) && aligned16(c1) && aligned16(c2);
    if (argk) ok = ok && ((reinterpret_cast<uintptr_t>(argk) & 3) == 0);
    if (!ok) return PN2_ERR_UNSUPPORTED;
    BwdArgs a;
    a.g = g; a.z = z; a.ldg = ldg; a.ldz = ldz; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd;
    a.c1 = c1; a.c2 = c2; a.argk = argk; a.pool_k = pool_k; a.w = w; a.ldw = ldw; a.x = x; a.ldx = ldx;
    a.ascale = ascale; a.ashift = ashift; a.amean = amean; a.ainvstd = ainvstd; a.gp = gp; a.ldgp = ldgp;
    a.stat_partial = stat_partial; a.dw_partial = dw_partial; a.M = M; a.N = N; a.K = K;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int nblk = fb_blocks(N), kblk = fb_blocks(K > 128 ? 128 : K);
    int rc = PN2_ERR_UNSUPPORTED;
    if (fs2_tile_rows(nblk, kblk)) {
        if (nblk == 4 && kblk == 4) rc = fs2_launch<4, 4, 32>(a, P, stream, (K + 127) / 128);
        else if (nblk == 2 && kblk == 2) rc = fs2_launch<2, 2, 64>(a, P, stream);
        else if (nblk == 4 && kblk == 2) rc = fs2_launch<4, 2, 64>(a, P, stream);
        else rc = fs2_launch<2, 4, 64>(a, P, stream);
    } else if (fs_applies(nblk, kblk)) {
#define PN2_FS(NB, PO) hipLaunchKernelGGL((mlp_bwd_split_kernel<NB, PO>), dim3(P), dim3(FB_THREADS), (fs_lds_bytes<NB>()), stream, a)
        if (nblk == 1) { if (argk) PN2_FS(1, true); else PN2_FS(1, false); }
        else { if (argk) PN2_FS(2, true); else PN2_FS(2, false); }
#undef PN2_FS
        rc = PN2_LAUNCH_RC();
    } else {
#define PN2_FB(NB, KB) if (nblk == NB && kblk == KB) rc = fb_launch<NB, KB>(a, P, stream)
    PN2_FB(1, 1); PN2_FB(1, 2); PN2_FB(1, 4);
    PN2_FB(2, 1); PN2_FB(2, 2); PN2_FB(2, 4);
    PN2_FB(4, 1); PN2_FB(4, 2); PN2_FB(4, 4);
#undef PN2_FB
    }
    if (rc != PN2_OK) return rc;
    if (c1_below)                                    // BatchNorm-backward constants of the layer below, same launch
        return pn2::launch_bwd_post(dw_partial, P, N, K, dw, db, stat_partial, P, K, (double)M, dgamma_below, dbeta_below, c1_below,
                                    c2_below, stream);
    if (!dw) return PN2_OK;
    return pn2::launch_dw_reduce(dw_partial, P, N, K, dw, db, stream);
}
