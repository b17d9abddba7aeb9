// Grouped / pointwise MLP of PointNetSetAbstraction and PointNetFeaturePropagation on gfx950:
// the [Conv 1x1 -> BatchNorm -> ReLU] x n (+ max over nsample) stacks of
// models/pointnet2_utils.py:196-200 and :312-314, as exact-fp32 MFMA GEMMs over channel-last
// rows (M = B*S*K or B*N rows, Ci -> Co channels).
//
//   Z = act(X) * W^T + bias          v_mfma_f32_32x32x2_f32, 128-row x BN-column tiles
//   act(x) = max(scale[c]*x + shift[c], 0)   the PREVIOUS layer's BatchNorm+ReLU, applied while
//                                            the tile is staged (the normalised activation is
//                                            never written to HBM)
//   epilogue: per-channel sum(z), sum(z^2) of this layer (train-mode batch statistics), kept in
//             registers across a workgroup's tiles -> one partial per workgroup (deterministic)
//
// These layers are HBM-bound (Ci, Co <= 512 against M up to 524288): the design goal is one
// read of the producer's raw output and one write of ours per layer.  MFMA operand mapping
// (cdna_hip_programming.md section 3): lane l supplies A[i=l&31][k] and B[k][j=l&31] with k taken from
// its half (l>>5) of the staged K chunk, so each lane reads CONSECUTIVE k with ds_read_b128.
#include <math.h>

#include "pn2_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MLP_BM = 128;       // rows per tile (4 waves x 32)
constexpr int MLP_BK = 32;        // K chunk staged per step
constexpr int MLP_LD = MLP_BK + 4;  // LDS row pitch in floats: 144 B keeps ds_read_b128 conflict-free
constexpr int MLP_THREADS = 256;

enum { PRO_NONE = 0, PRO_BN_RELU = 1, PRO_BN_BWD = 2 };
// PRO_BN_BWD with argk != null: g[row,c] = (argk[row/pool_k, c] == row%pool_k) ? x1[row/pool_k, c] : 0
// (the gradient of torch.max over nsample, models/pointnet2_utils.py:200, never materialised).

struct GemmArgs {
    // A = [X1 (K1 cols) | X2 (K2 cols)], row-major with row pitches ld1 / ld2 (X2 may be null)
    const float *x1, *x2;
    int ld1, ld2, K1, K2;
    // PRO_BN_RELU: a = max(scale[k]*x + shift[k], 0) over the concatenated K axis
    // PRO_BN_BWD : a = dz computed from (g = x1, z = x2, both [M,K1]):
    //              gh = (scale*z+shift > 0) ? g : 0; xh = (z-mean)*invstd; dz = scale*(gh - c1 - xh*c2)
    const float *scale, *shift, *mean, *invstd, *c1, *c2;
    const unsigned char *argk;    // [M/pool_k][K1] or null
    int pool_k;
    // B: W is [N][K] row-major (wt == 0) or [K][N] row-major (wt == 1)
    const float *w;
    int ldw, wt;
    const float *bias;            // [N] or null
    float *out;                   // [M][N], pitch ldo
    int ldo, M, N, K;
    float *out2;                  // optional: columns >= nsplit go to out2[M][N-nsplit] (pitch ldo2) instead
    int ldo2, nsplit;
    float *stat_partial;          // [gridDim.x][2][N] or null: column sums of out and out^2
    // backward epilogue (mask_z != null): out = (mscale*zprev+mshift > 0) ? out : 0 ; partials of
    // sum(out) and sum(out * (zprev-mmean)*minvstd) go to stat_partial instead
    const float *mask_z, *mscale, *mshift, *mmean, *minvstd;
    int ldm;
    // pooled forward epilogue (pool_max != null; rows come in groups of 32 = one 32-row accumulator block): per
    // group and column the largest and the smallest z with the first row that holds it.  max over the group of
    // relu(scale*z+shift) is relu(scale*zmax+shift) for scale >= 0 and relu(scale*zmin+shift) otherwise (rounding is
    // monotone), so the max-pool (models/pointnet2_utils.py:200) needs no second pass over z once the batch
    // statistics are known (bn_finalize_out_kernel).
    float *pool_max, *pool_min;
    unsigned char *pool_amax, *pool_amin;
};

// Epilogue of one 32x32 accumulator block (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*half): bias,
// optionally the ReLU mask of the layer below with its BatchNorm-backward sums, column sums, stores.
// Every value the rows need is loaded and waited for BEFORE the first store, and a block whose 32 rows are all
// < M (FULL) stores in straight-line code: vmcnt counts stores too on gfx9, and with a load result first used
// inside per-row branches the compiler puts s_waitcnt vmcnt(0) in front of every store, i.e. one store
// acknowledgement round trip per row (16 per block).
template <bool FULL, bool BWD, int POOL = 0>       // POOL 1: record the group extrema; 2: ... and do not store z (inference)
__device__ __forceinline__ void gemm_store_block(const GemmArgs &p, const f32x16 &acc, int row0, int half, int col,
                                                 float &csum, float &csq)
{
    float vmax = -INFINITY, vmin = INFINITY;
    int imax = 0, imin = 0;
    float bv = 0.f;
    if (p.bias) bv = p.bias[col];
    const bool second = p.out2 && col >= p.nsplit;
    float *ob = second ? p.out2 + (col - p.nsplit) : p.out + col;
    const size_t ld = second ? (size_t)p.ldo2 : (size_t)p.ldo;
    float ms = 0.f, mh = 0.f, mm = 0.f, mi = 0.f;
    float zp[16];
    if (BWD) {
        ms = p.mscale[col]; mh = p.mshift[col]; mm = p.mmean[col]; mi = p.minvstd[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            zp[r] = p.mask_z[(size_t)(FULL ? row : min(row, p.M - 1)) * p.ldm + col];
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // addresses four at a time, not 16 pairs
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(zp[r]));
        asm volatile("" :: "v"(ms), "v"(mh), "v"(mm), "v"(mi));
    }
    asm volatile("" :: "v"(bv));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (FULL || row < p.M) {
            float z = acc[r] + bv;
            if (BWD) {
                z = (ms * zp[r] + mh) > 0.f ? z : 0.f;
                csum += z;
                csq += z * ((zp[r] - mm) * mi);
            } else {
                csum += z;
                csq += z * z;
            }
            if (POOL != 2) {                                         // compile-time: the stores stay straight-line code
                // write-through: the rows are read next by another kernel (on any XCD) and the launch would otherwise
                // end with the write-back of up to 32 MB of dirty L2 lines
                pn2::store_rows(&ob[(size_t)row * ld], z);
            }
            if (POOL) {                                              // rows ascend with r: '>' keeps the first
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (z > vmax) { vmax = z; imax = rr; }
                if (z < vmin) { vmin = z; imin = rr; }
            }
        }
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the next block's loads (16 more registers) behind these stores
    if (POOL) {
        // the other 16 rows of this column live in lane ^ 32
        const float omax = __shfl_xor(vmax, 32), omin = __shfl_xor(vmin, 32);
        const int oimax = __shfl_xor(imax, 32), oimin = __shfl_xor(imin, 32);
        if (omax > vmax || (omax == vmax && oimax < imax)) { vmax = omax; imax = oimax; }
        if (omin < vmin || (omin == vmin && oimin < imin)) { vmin = omin; imin = oimin; }
        const size_t o = (size_t)(row0 >> 5) * p.N + col;
        if (half == 0) { p.pool_max[o] = vmax; p.pool_amax[o] = (unsigned char)imax; }
        else { p.pool_min[o] = vmin; p.pool_amin[o] = (unsigned char)imin; }
    }
}

template <bool BWD_POSSIBLE>
__device__ __forceinline__ void gemm_store_block_any(const GemmArgs &p, const f32x16 &acc, int row0, int half, int col,
                                                     bool bwd_epi, float &csum, float &csq)
{
    const bool full = row0 + 32 <= p.M;                       // wave-uniform
    if (BWD_POSSIBLE && bwd_epi) {
        if (full) gemm_store_block<true, true>(p, acc, row0, half, col, csum, csq);
        else gemm_store_block<false, true>(p, acc, row0, half, col, csum, csq);
    } else {
        if (full && p.pool_max && p.out) gemm_store_block<true, false, 1>(p, acc, row0, half, col, csum, csq);
        else if (full && p.pool_max) gemm_store_block<true, false, 2>(p, acc, row0, half, col, csum, csq);
        else if (full) gemm_store_block<true, false>(p, acc, row0, half, col, csum, csq);
        else gemm_store_block<false, false>(p, acc, row0, half, col, csum, csq);
    }
}

template <int BN, int PRO, bool VEC4>
__global__ __launch_bounds__(MLP_THREADS) void mlp_gemm_kernel(GemmArgs p)
{
    constexpr int NB = BN / 32;                       // 32-column accumulator blocks per wave
    __shared__ __attribute__((aligned(16))) float sA[MLP_BM * MLP_LD];
    __shared__ __attribute__((aligned(16))) float sB[BN * MLP_LD];
    __shared__ float sRed[4][2][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int col0 = blockIdx.y * BN;
    const int ntiles = (p.M + MLP_BM - 1) / MLP_BM;
    const bool bwd_epi = p.mask_z != nullptr;

    float csum[NB], csq[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) { csum[cb] = 0.0f; csq[cb] = 0.0f; }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * MLP_BM;
        f32x16 acc[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;

        for (int k0 = 0; k0 < p.K; k0 += MLP_BK) {
            // ---- stage A chunk [128 x 32] ------------------------------------------------
            if (VEC4) {
#pragma unroll
                for (int i = 0; i < (MLP_BM * MLP_BK / 4) / MLP_THREADS; ++i) {
                    const int e = tid + i * MLP_THREADS;
                    const int r = e >> 3, c4 = (e & 7) * 4;
                    const int row = row0 + r, k = k0 + c4;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < p.M && k < p.K) {
                        if (PRO == PRO_BN_BWD) {
                            float4 g;
                            if (p.argk) {
                                const int cent = row / p.pool_k, kk = row - cent * p.pool_k;
                                const uchar4 ak = *reinterpret_cast<const uchar4 *>(p.argk + (size_t)cent * p.K1 + k);
                                const float4 go = *reinterpret_cast<const float4 *>(p.x1 + (size_t)cent * p.ld1 + k);
                                g = make_float4(ak.x == kk ? go.x : 0.f, ak.y == kk ? go.y : 0.f, ak.z == kk ? go.z : 0.f,
                                                ak.w == kk ? go.w : 0.f);
                            } else {
                                g = *reinterpret_cast<const float4 *>(p.x1 + (size_t)row * p.ld1 + k);
                            }
                            const float4 z = *reinterpret_cast<const float4 *>(p.x2 + (size_t)row * p.ld2 + k);
                            const float4 sc = *reinterpret_cast<const float4 *>(p.scale + k);
                            const float4 sh = *reinterpret_cast<const float4 *>(p.shift + k);
                            const float4 mu = *reinterpret_cast<const float4 *>(p.mean + k);
                            const float4 is = *reinterpret_cast<const float4 *>(p.invstd + k);
                            const float4 a1 = *reinterpret_cast<const float4 *>(p.c1 + k);
                            const float4 a2 = *reinterpret_cast<const float4 *>(p.c2 + k);
#define PN2_DZ(f) v.f = sc.f * (((sc.f * z.f + sh.f) > 0.f ? g.f : 0.f) - a1.f - (z.f - mu.f) * is.f * a2.f)
                            PN2_DZ(x); PN2_DZ(y); PN2_DZ(z); PN2_DZ(w);
#undef PN2_DZ
                        } else {
                            v = k < p.K1 ? *reinterpret_cast<const float4 *>(p.x1 + (size_t)row * p.ld1 + k)
                                         : *reinterpret_cast<const float4 *>(p.x2 + (size_t)row * p.ld2 + (k - p.K1));
                            if (PRO == PRO_BN_RELU) {
                                const float4 sc = *reinterpret_cast<const float4 *>(p.scale + k);
                                const float4 sh = *reinterpret_cast<const float4 *>(p.shift + k);
                                v.x = fmaxf(sc.x * v.x + sh.x, 0.f);
                                v.y = fmaxf(sc.y * v.y + sh.y, 0.f);
                                v.z = fmaxf(sc.z * v.z + sh.z, 0.f);
                                v.w = fmaxf(sc.w * v.w + sh.w, 0.f);
                            }
                        }
                    }
                    *reinterpret_cast<float4 *>(&sA[r * MLP_LD + c4]) = v;
                }
            } else {
#pragma unroll 4
                for (int i = 0; i < (MLP_BM * MLP_BK) / MLP_THREADS; ++i) {
                    const int e = tid + i * MLP_THREADS;
                    const int r = e >> 5, c = e & 31;
                    const int row = row0 + r, k = k0 + c;
                    float v = 0.f;
                    if (row < p.M && k < p.K) {
                        if (PRO == PRO_BN_BWD) {
                            float g;
                            if (p.argk) {
                                const int cent = row / p.pool_k, kk = row - cent * p.pool_k;
                                g = p.argk[(size_t)cent * p.K1 + k] == kk ? p.x1[(size_t)cent * p.ld1 + k] : 0.f;
                            } else {
                                g = p.x1[(size_t)row * p.ld1 + k];
                            }
                            const float z = p.x2[(size_t)row * p.ld2 + k];
                            const float sc = p.scale[k];
                            v = sc * (((sc * z + p.shift[k]) > 0.f ? g : 0.f) - p.c1[k] - (z - p.mean[k]) * p.invstd[k] * p.c2[k]);
                        } else {
                            v = k < p.K1 ? p.x1[(size_t)row * p.ld1 + k] : p.x2[(size_t)row * p.ld2 + (k - p.K1)];
                            if (PRO == PRO_BN_RELU) v = fmaxf(p.scale[k] * v + p.shift[k], 0.f);
                        }
                    }
                    sA[r * MLP_LD + c] = v;
                }
            }
            // ---- stage B chunk as [BN cols][32 k] ------------------------------------------
            if (p.wt == 0) {
                if (VEC4 && (p.ldw & 3) == 0) {
#pragma unroll
                    for (int i = 0; i < (BN * MLP_BK / 4 + MLP_THREADS - 1) / MLP_THREADS; ++i) {
                        const int e = tid + i * MLP_THREADS;
                        if (e < BN * MLP_BK / 4) {
                            const int r = e >> 3, c4 = (e & 7) * 4;
                            const int n = col0 + r, k = k0 + c4;
                            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (n < p.N && k < p.K) v = *reinterpret_cast<const float4 *>(p.w + (size_t)n * p.ldw + k);
                            *reinterpret_cast<float4 *>(&sB[r * MLP_LD + c4]) = v;
                        }
                    }
                } else {
                    for (int e = tid; e < BN * MLP_BK; e += MLP_THREADS) {
                        const int r = e >> 5, c = e & 31;
                        const int n = col0 + r, k = k0 + c;
                        sB[r * MLP_LD + c] = (n < p.N && k < p.K) ? p.w[(size_t)n * p.ldw + k] : 0.f;
                    }
                }
            } else {   // W given as [K][N]: read along N (coalesced), transpose into LDS
                for (int e = tid; e < BN * MLP_BK; e += MLP_THREADS) {
                    const int c = e / BN, r = e - c * BN;              // c: k within chunk, r: column
                    const int n = col0 + r, k = k0 + c;
                    sB[r * MLP_LD + c] = (n < p.N && k < p.K) ? p.w[(size_t)k * p.ldw + n] : 0.f;
                }
            }
            __syncthreads();
            // ---- MFMA: each lane owns k = 16*half + t of the chunk -------------------------
            const float *aRow = &sA[(wave * 32 + l31) * MLP_LD + 16 * half];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a4 = *reinterpret_cast<const float4 *>(aRow + 4 * q);
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    const float4 b4 = *reinterpret_cast<const float4 *>(&sB[(cb * 32 + l31) * MLP_LD + 16 * half + 4 * q]);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[cb], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*half ----------
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int col = col0 + cb * 32 + l31;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
            float ms = 0.f, mh = 0.f, mm = 0.f, mi = 0.f;
            if (bwd_epi) { ms = p.mscale[col]; mh = p.mshift[col]; mm = p.mmean[col]; mi = p.minvstd[col]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= p.M) continue;
                float z = acc[cb][r] + bv;
                if (bwd_epi) {
                    const float zp = p.mask_z[(size_t)row * p.ldm + col];
                    z = (ms * zp + mh) > 0.f ? z : 0.f;
                    csum[cb] += z;
                    csq[cb] += z * ((zp - mm) * mi);
                } else {
                    csum[cb] += z;
                    csq[cb] += z * z;
                }
                pn2::store_rows((p.out2 && col >= p.nsplit) ? &p.out2[(size_t)row * p.ldo2 + (col - p.nsplit)] : &p.out[(size_t)row * p.ldo + col], z);
            }
        }
    }
    if (!p.stat_partial) return;
    // ---- per-workgroup partial statistics ---------------------------------------------------
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        float s = csum[cb] + __shfl_xor(csum[cb], 32);
        float q = csq[cb] + __shfl_xor(csq[cb], 32);
        if (half == 0) { sRed[wave][0][cb * 32 + l31] = s; sRed[wave][1][cb * 32 + l31] = q; }
    }
    __syncthreads();
    for (int e = tid; e < 2 * BN; e += MLP_THREADS) {
        const int which = e / BN, c = e - which * BN;
        if (col0 + c < p.N) {
            const float v = (sRed[0][which][c] + sRed[1][which][c]) + (sRed[2][which][c] + sRed[3][which][c]);
            p.stat_partial[((size_t)blockIdx.x * 2 + which) * p.N + col0 + c] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined form of mlp_gemm_kernel for the all-float4 case.  A workgroup's (tile,
// K-chunk) steps form one flat sequence; while the MFMAs of step s run, the global loads of step
// s+1 (across tile boundaries too) are already in flight into registers, and are transformed
// (BatchNorm/ReLU prologue or the dz formula) and written to LDS after the barrier.
// WT: the weight is given [K][N] (dX = dZ * W); its tile is staged k-major and read with
// ds_read_b32 (lanes along N: conflict-free) instead of being transposed.
// NW = 1: 4 waves, each 32 rows x all BN columns.  NW = 2 (BN = 128): 8 waves, each 32 rows x 64 columns
// -- twice the waves per CU to cover LDS / barrier latency, half the accumulator registers.
template <int BN, int PRO, bool WT, int NW>
__global__ __launch_bounds__(MLP_THREADS * NW, 2 * NW) void mlp_gemm_pipe_kernel(GemmArgs p, int pool_shift)
{
    PN2_MAIN_BRANCH_PRIORITY();
    constexpr int T = MLP_THREADS * NW;
    constexpr int NB = BN / 32;
    constexpr int NBW = NB / NW;                      // 32-column blocks per wave
    constexpr int AI = 4 / NW;                        // A float4 per thread per chunk
    constexpr int BI = (NB + NW - 1) / NW;            // B float4 per thread per chunk
    static_assert(NB % NW == 0 && 4 % NW == 0, "bad wave split");
    constexpr int LDBT = BN + 4;
    constexpr int SB_ELEMS = WT ? MLP_BK * LDBT : BN * MLP_LD;
    constexpr int SA_ELEMS = MLP_BM * MLP_LD;
    __shared__ __attribute__((aligned(16))) float sAbuf[2 * SA_ELEMS];     // double-buffered: step s reads
    __shared__ __attribute__((aligned(16))) float sBbuf[2 * SB_ELEMS];     // half s&1 while s+1 is written
    __shared__ float sRed[4][2][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int rw = wave & 3, cw = wave >> 2;         // row slice / column group of this wave
    const int col0 = blockIdx.y * BN;
    const int ntiles = (p.M + MLP_BM - 1) / MLP_BM;
    const int nk = (p.K + MLP_BK - 1) / MLP_BK;
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int nsteps = my_tiles * nk;
    const bool bwd_epi = p.mask_z != nullptr;
    const int ar = tid >> 3, ac4 = (tid & 7) * 4;           // A staging: rows ar + (T/8)*i, 4 columns at ac4
    constexpr int ARS = T / 8;

    float4 ra[AI], rz[AI], rb[BI];
    uchar4 rk[AI];
    float4 cs, ch, cm, ci, cc1, cc2;
    cs = ch = cm = ci = cc1 = cc2 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto issue = [&](int step) {
        const int t = step / nk, kc = step - t * nk;
        const int row0 = ((int)blockIdx.x + t * (int)gridDim.x) * MLP_BM, k0 = kc * MLP_BK;
        const int k = k0 + ac4;
        const bool kok = k < p.K;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = row0 + ar + ARS * i;
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (PRO == PRO_BN_BWD) { rz[i] = ra[i]; rk[i] = make_uchar4(255, 255, 255, 255); }
            if (row < p.M && kok) {
                if (PRO == PRO_BN_BWD) {
                    if (p.argk) {
                        const int cent = pool_shift >= 0 ? (row >> pool_shift) : row / p.pool_k;
                        ra[i] = *reinterpret_cast<const float4 *>(p.x1 + (size_t)cent * p.ld1 + k);
                        rk[i] = *reinterpret_cast<const uchar4 *>(p.argk + (size_t)cent * p.K1 + k);
                    } else {
                        ra[i] = *reinterpret_cast<const float4 *>(p.x1 + (size_t)row * p.ld1 + k);
                    }
                    rz[i] = *reinterpret_cast<const float4 *>(p.x2 + (size_t)row * p.ld2 + k);
                } else {
                    ra[i] = k < p.K1 ? *reinterpret_cast<const float4 *>(p.x1 + (size_t)row * p.ld1 + k)
                                     : *reinterpret_cast<const float4 *>(p.x2 + (size_t)row * p.ld2 + (k - p.K1));
                }
            }
        }
        if (PRO != PRO_NONE && kok) {
            cs = *reinterpret_cast<const float4 *>(p.scale + k);
            ch = *reinterpret_cast<const float4 *>(p.shift + k);
            if (PRO == PRO_BN_BWD) {
                cm = *reinterpret_cast<const float4 *>(p.mean + k);
                ci = *reinterpret_cast<const float4 *>(p.invstd + k);
                cc1 = *reinterpret_cast<const float4 *>(p.c1 + k);
                cc2 = *reinterpret_cast<const float4 *>(p.c2 + k);
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int e = tid + i * T;
            rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e >= BN * 8) continue;
            if (!WT) {
                const int r = e >> 3, c4 = (e & 7) * 4;              // [BN cols][32 k]
                if (col0 + r < p.N && k0 + c4 < p.K)
                    rb[i] = *reinterpret_cast<const float4 *>(p.w + (size_t)(col0 + r) * p.ldw + k0 + c4);
            } else {
                const int kk = e / (BN / 4), c4 = (e - kk * (BN / 4)) * 4;   // [32 k][BN cols]
                if (k0 + kk < p.K && col0 + c4 < p.N)
                    rb[i] = *reinterpret_cast<const float4 *>(p.w + (size_t)(k0 + kk) * p.ldw + col0 + c4);
            }
        }
    };

    auto commit = [&](int step) {
        const int t = step / nk;
        const int row0 = ((int)blockIdx.x + t * (int)gridDim.x) * MLP_BM;
        float *sA = sAbuf + (step & 1) * SA_ELEMS;
        float *sB = sBbuf + (step & 1) * SB_ELEMS;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            float4 v = ra[i];
            if (PRO == PRO_BN_RELU) {
                v.x = fmaxf(cs.x * v.x + ch.x, 0.f);
                v.y = fmaxf(cs.y * v.y + ch.y, 0.f);
                v.z = fmaxf(cs.z * v.z + ch.z, 0.f);
                v.w = fmaxf(cs.w * v.w + ch.w, 0.f);
                // rows past M / columns past K were loaded as 0 and must stay 0 (shift may be > 0)
                const int row = row0 + ar + ARS * i;
                if (row >= p.M) v = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (PRO == PRO_BN_BWD) {
                float4 g = ra[i];
                const float4 z = rz[i];
                if (p.argk) {
                    const int row = row0 + ar + ARS * i;
                    const int kk = pool_shift >= 0 ? (row & ((1 << pool_shift) - 1)) : row % p.pool_k;
                    g.x = rk[i].x == kk ? g.x : 0.f;
                    g.y = rk[i].y == kk ? g.y : 0.f;
                    g.z = rk[i].z == kk ? g.z : 0.f;
                    g.w = rk[i].w == kk ? g.w : 0.f;
                }
#define PN2_DZ(f) v.f = cs.f * (((cs.f * z.f + ch.f) > 0.f ? g.f : 0.f) - cc1.f - (z.f - cm.f) * ci.f * cc2.f)
                PN2_DZ(x); PN2_DZ(y); PN2_DZ(z); PN2_DZ(w);
#undef PN2_DZ
                const int row = row0 + ar + ARS * i;
                if (row >= p.M) v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            *reinterpret_cast<float4 *>(&sA[(ar + ARS * i) * MLP_LD + ac4]) = v;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int e = tid + i * T;
            if (e >= BN * 8) continue;
            if (!WT) {
                const int r = e >> 3, c4 = (e & 7) * 4;
                *reinterpret_cast<float4 *>(&sB[r * MLP_LD + c4]) = rb[i];
            } else {
                const int kk = e / (BN / 4), c4 = (e - kk * (BN / 4)) * 4;
                *reinterpret_cast<float4 *>(&sB[kk * LDBT + c4]) = rb[i];
            }
        }
    };

    float csum[NBW], csq[NBW];
    f32x16 acc[NBW];
#pragma unroll
    for (int cb = 0; cb < NBW; ++cb) {
        csum[cb] = 0.0f;
        csq[cb] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
    }

    if (nsteps > 0) {
        issue(0);
        commit(0);
        __syncthreads();
    }
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) issue(step + 1);                  // next step's loads fly under the MFMAs
        const float *sA = sAbuf + (step & 1) * SA_ELEMS;
        const float *sB = sBbuf + (step & 1) * SB_ELEMS;
        const float *aRow = &sA[(rw * 32 + l31) * MLP_LD + 16 * half];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a4 = *reinterpret_cast<const float4 *>(aRow + 4 * q);
#pragma unroll
            for (int cb = 0; cb < NBW; ++cb) {
                float4 b4;
                if (!WT) {
                    b4 = *reinterpret_cast<const float4 *>(&sB[((cw * NBW + cb) * 32 + l31) * MLP_LD + 16 * half + 4 * q]);
                } else {
                    const float *bp = &sB[(16 * half + 4 * q) * LDBT + (cw * NBW + cb) * 32 + l31];
                    b4 = make_float4(bp[0], bp[LDBT], bp[2 * LDBT], bp[3 * LDBT]);
                }
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[cb], 0, 0, 0);
            }
        }
        const int t = step / nk, kc = step - t * nk;
        if (kc == nk - 1) {
            if constexpr (PRO == PRO_BN_BWD) {
            // dX form: the per-row form below keeps this kernel's register count (the straight-line one costs
            // 40 more on top of the dz prologue's prefetch registers and spills)
            // ---- epilogue of this tile: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*half
            const int row0 = ((int)blockIdx.x + t * (int)gridDim.x) * MLP_BM;
#pragma unroll
            for (int cb = 0; cb < NBW; ++cb) {
                const int col = col0 + (cw * NBW + cb) * 32 + l31;
                if (col < p.N) {
                    const float bv = p.bias ? p.bias[col] : 0.f;
                    float ms = 0.f, mh = 0.f, mm = 0.f, mi = 0.f;
                    float zp[16];
                    if (bwd_epi) {
                        ms = p.mscale[col]; mh = p.mshift[col]; mm = p.mmean[col]; mi = p.minvstd[col];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = row0 + rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            zp[r] = row < p.M ? p.mask_z[(size_t)row * p.ldm + col] : 0.f;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (row < p.M) {
                            float z = acc[cb][r] + bv;
                            if (bwd_epi) {
                                z = (ms * zp[r] + mh) > 0.f ? z : 0.f;
                                csum[cb] += z;
                                csq[cb] += z * ((zp[r] - mm) * mi);
                            } else {
                                csum[cb] += z;
                                csq[cb] += z * z;
                            }
                            pn2::store_rows((p.out2 && col >= p.nsplit) ? &p.out2[(size_t)row * p.ldo2 + (col - p.nsplit)] : &p.out[(size_t)row * p.ldo + col], z);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
            }
            } else {
            // ---- epilogue of this tile
            const int row0 = ((int)blockIdx.x + t * (int)gridDim.x) * MLP_BM + rw * 32;
#pragma unroll
            for (int cb = 0; cb < NBW; ++cb) {
                const int col = col0 + (cw * NBW + cb) * 32 + l31;
                if (col < p.N)
                    gemm_store_block_any<PRO == PRO_BN_BWD>(p, acc[cb], row0, half, col, bwd_epi, csum[cb], csq[cb]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
            }
            }
        }
        if (step + 1 < nsteps) commit(step + 1);                 // into the other LDS half
        __syncthreads();                                         // the only barrier of the step
    }
    if (!p.stat_partial) return;
#pragma unroll
    for (int cb = 0; cb < NBW; ++cb) {
        float s = csum[cb] + __shfl_xor(csum[cb], 32);
        float q = csq[cb] + __shfl_xor(csq[cb], 32);
        if (half == 0) { sRed[rw][0][(cw * NBW + cb) * 32 + l31] = s; sRed[rw][1][(cw * NBW + cb) * 32 + l31] = q; }
    }
    __syncthreads();
    for (int e = tid; e < 2 * BN; e += T) {
        const int which = e / BN, c = e - which * BN;
        if (col0 + c < p.N) {
            const float v = (sRed[0][which][c] + sRed[1][which][c]) + (sRed[2][which][c] + sRed[3][which][c]);
            p.stat_partial[((size_t)blockIdx.x * 2 + which) * p.N + col0 + c] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Few-rows form of the pipelined GEMM: 32-row x 128-column tiles, the four waves of a workgroup
// take one 32-column block each and share the A rows.  Used when M is so small (deep levels:
// M = 1024..8192) that 128-row tiles would leave most of the 256 CUs idle.  Same prologues,
// epilogues, statistics and double-buffered staging as mlp_gemm_pipe_kernel.
// BM = 64: eight waves, 2 row slices x 4 column blocks: half the weight-tile traffic per flop, for shapes
// that still give every CU a workgroup with 64-row tiles.
// BK = 64 (BM = 64 only, dynamic LDS): half as many steps, each with twice the MFMA work -- with one 8-wave workgroup
// per CU (the dz prologue's registers) a 32-wide step (0.85 us of MFMAs per SIMD) is shorter than the global-load
// latency of the prefetch it is supposed to hide.
template <int PRO, bool WT, int BM, int BK>
__global__ __launch_bounds__(8 * BM) void mlp_gemm_rows32_kernel(GemmArgs p, int pool_shift)
{
    PN2_MAIN_BRANCH_PRIORITY();
    constexpr int BN = 128;
    constexpr int T = 8 * BM;                         // threads
    constexpr int RS = BM / 32;                       // 32-row slices
    constexpr int Q = BK / 4;                         // float4 per staged row
    constexpr int AI = (BM * Q) / T;                  // A float4 per thread and chunk (1; 2 for BK = 64)
    constexpr int RPP = T / Q;                        // rows staged per pass
    constexpr int NBI = (BN * Q) / T;                 // B float4 per thread and chunk
    constexpr int LD = BK + 4;                        // LDS row pitch (BK + 4: ds_read_b128 stays conflict-free)
    constexpr int HK = BK / 2;                        // k range of one lane half
    constexpr int LDBT = BN + 4;
    constexpr int SB_ELEMS = WT ? BK * LDBT : BN * LD;
    constexpr int SA_ELEMS = BM * LD;
    static_assert(BK == 32 || BK == 64, "tile");
    extern __shared__ __attribute__((aligned(16))) float rows_lds[];
    float *sAbuf = rows_lds;                          // [2][SA_ELEMS]
    float *sBbuf = rows_lds + 2 * SA_ELEMS;           // [2][SB_ELEMS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int col0 = blockIdx.y * BN;
    const int ntiles = (p.M + BM - 1) / BM;
    const int nk = (p.K + BK - 1) / BK;
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int nsteps = my_tiles * nk;
    const bool bwd_epi = p.mask_z != nullptr;
    const int ar = tid / Q, ac4 = (tid % Q) * 4;            // A staging: rows ar + i*RPP, 4 columns at ac4
    const int rw = wave % RS, cw = wave / RS;               // this wave's row slice / 32-column block

    float4 ra[AI], rz[AI], rb[NBI];
    uchar4 rk[AI];
    float4 cs, ch, cm, ci, cc1, cc2;
    cs = ch = cm = ci = cc1 = cc2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < AI; ++i) { ra[i] = rz[i] = cs; rk[i] = make_uchar4(255, 255, 255, 255); }

    auto issue = [&](int step) {
        const int t = step / nk, kc = step - t * nk;
        const int row0 = ((int)blockIdx.x + t * (int)gridDim.x) * BM, k0 = kc * BK;
        const int k = k0 + ac4;
        const bool kok = k < p.K;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = row0 + ar + i * RPP;
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (PRO == PRO_BN_BWD) { rz[i] = ra[i]; rk[i] = make_uchar4(255, 255, 255, 255); }
            if (row < p.M && kok) {
                if (PRO == PRO_BN_BWD) {
                    if (p.argk) {
                        const int cent = pool_shift >= 0 ? (row >> pool_shift) : row / p.pool_k;
                        ra[i] = *reinterpret_cast<const float4 *>(p.x1 + (size_t)cent * p.ld1 + k);
                        rk[i] = *reinterpret_cast<const uchar4 *>(p.argk + (size_t)cent * p.K1 + k);
                    } else {
                        ra[i] = *reinterpret_cast<const float4 *>(p.x1 + (size_t)row * p.ld1 + k);
                    }
                    rz[i] = *reinterpret_cast<const float4 *>(p.x2 + (size_t)row * p.ld2 + k);
                } else {
                    ra[i] = k < p.K1 ? *reinterpret_cast<const float4 *>(p.x1 + (size_t)row * p.ld1 + k)
                                     : *reinterpret_cast<const float4 *>(p.x2 + (size_t)row * p.ld2 + (k - p.K1));
                }
            }
        }
        if (PRO != PRO_NONE && kok) {
            cs = *reinterpret_cast<const float4 *>(p.scale + k);
            ch = *reinterpret_cast<const float4 *>(p.shift + k);
            if (PRO == PRO_BN_BWD) {
                cm = *reinterpret_cast<const float4 *>(p.mean + k);
                ci = *reinterpret_cast<const float4 *>(p.invstd + k);
                cc1 = *reinterpret_cast<const float4 *>(p.c1 + k);
                cc2 = *reinterpret_cast<const float4 *>(p.c2 + k);
            }
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int e = tid + i * T;
            rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!WT) {
                const int r = e / Q, c4 = (e % Q) * 4;
                if (col0 + r < p.N && k0 + c4 < p.K)
                    rb[i] = *reinterpret_cast<const float4 *>(p.w + (size_t)(col0 + r) * p.ldw + k0 + c4);
            } else {
                const int kk = e / (BN / 4), c4 = (e - kk * (BN / 4)) * 4;
                if (k0 + kk < p.K && col0 + c4 < p.N)
                    rb[i] = *reinterpret_cast<const float4 *>(p.w + (size_t)(k0 + kk) * p.ldw + col0 + c4);
            }
        }
    };
    auto commit = [&](int step) {
        const int t = step / nk;
        const int row0 = ((int)blockIdx.x + t * (int)gridDim.x) * BM + ar;
        float *sA = sAbuf + (step & 1) * SA_ELEMS;
        float *sB = sBbuf + (step & 1) * SB_ELEMS;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = row0 + i * RPP;
            float4 v = ra[i];
            if (PRO == PRO_BN_RELU) {
                v.x = fmaxf(cs.x * v.x + ch.x, 0.f);
                v.y = fmaxf(cs.y * v.y + ch.y, 0.f);
                v.z = fmaxf(cs.z * v.z + ch.z, 0.f);
                v.w = fmaxf(cs.w * v.w + ch.w, 0.f);
                if (row >= p.M) v = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (PRO == PRO_BN_BWD) {
                float4 g = ra[i];
                const float4 z = rz[i];
                if (p.argk) {
                    const int kk = pool_shift >= 0 ? (row & ((1 << pool_shift) - 1)) : row % p.pool_k;
                    g.x = rk[i].x == kk ? g.x : 0.f;
                    g.y = rk[i].y == kk ? g.y : 0.f;
                    g.z = rk[i].z == kk ? g.z : 0.f;
                    g.w = rk[i].w == kk ? g.w : 0.f;
                }
#define PN2_DZ(f) v.f = cs.f * (((cs.f * z.f + ch.f) > 0.f ? g.f : 0.f) - cc1.f - (z.f - cm.f) * ci.f * cc2.f)
                PN2_DZ(x); PN2_DZ(y); PN2_DZ(z); PN2_DZ(w);
#undef PN2_DZ
                if (row >= p.M) v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            *reinterpret_cast<float4 *>(&sA[(ar + i * RPP) * LD + ac4]) = v;
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int e = tid + i * T;
            if (!WT) {
                const int r = e / Q, c4 = (e % Q) * 4;
                *reinterpret_cast<float4 *>(&sB[r * LD + c4]) = rb[i];
            } else {
                const int kk = e / (BN / 4), c4 = (e - kk * (BN / 4)) * 4;
                *reinterpret_cast<float4 *>(&sB[kk * LDBT + c4]) = rb[i];
            }
        }
    };

    float csum = 0.f, csq = 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int col = col0 + cw * 32 + l31;                    // this lane's output column

    if (nsteps > 0) {
        issue(0);
        commit(0);
        __syncthreads();
    }
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) issue(step + 1);
        const float *sA = sAbuf + (step & 1) * SA_ELEMS;
        const float *sB = sBbuf + (step & 1) * SB_ELEMS;
        const float *aRow = &sA[(rw * 32 + l31) * LD + HK * half];
        // Operand read-ahead (left alone, the compiler puts every LDS read right in front of its consumer:
        // read -> wait -> MFMAs).  Not in mlp_gemm_pipe_kernel: its 8-wave form has no registers to spare.
        auto load_b = [&](int q) -> float4 {
            if (!WT) return *reinterpret_cast<const float4 *>(&sB[(cw * 32 + l31) * LD + HK * half + 4 * q]);
            const float *bp = &sB[(HK * half + 4 * q) * LDBT + cw * 32 + l31];
            return make_float4(bp[0], bp[LDBT], bp[2 * LDBT], bp[3 * LDBT]);
        };
        float4 a4 = *reinterpret_cast<const float4 *>(aRow);
        float4 b4 = load_b(0);
#pragma unroll
        for (int q = 0; q < HK / 4; ++q) {
            float4 an = a4, bn = b4;
            if (q + 1 < HK / 4) { an = *reinterpret_cast<const float4 *>(aRow + 4 * (q + 1)); bn = load_b(q + 1); }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            a4 = an; b4 = bn;
        }
        const int t = step / nk, kc = step - t * nk;
        if (kc == nk - 1) {
            const int row0 = ((int)blockIdx.x + t * (int)gridDim.x) * BM + rw * 32;
            if (col < p.N) gemm_store_block_any<PRO == PRO_BN_BWD>(p, acc, row0, half, col, bwd_epi, csum, csq);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        }
        if (step + 1 < nsteps) commit(step + 1);
        __syncthreads();
    }
    if (!p.stat_partial) return;
    const float s = csum + __shfl_xor(csum, 32);
    const float q = csq + __shfl_xor(csq, 32);
    if (half == 0 && col < p.N) {                             // every wave owns its own 32 rows x 32 columns
        const size_t prow = (size_t)blockIdx.x * RS + rw;     // RS partial rows per workgroup
        p.stat_partial[(prow * 2 + 0) * p.N + col] = s;
        p.stat_partial[(prow * 2 + 1) * p.N + col] = q;
    }
}

// ---------------------------------------------------------------------------------------------
// Very-few-rows form (fp4: M <= 1024 rows against K, N of 256 ... 768): a 32 x 128 tile per workgroup leaves three
// quarters of the chip idle and runs its K loop as 12 ... 24 dependent load -> multiply steps (25 us for 0.4 GFLOP).  Here a
// workgroup owns ONE 32 x 32 output block and its eight waves split K: a wave issues every load of its K slice at once,
// straight into the MFMA operand layout (lane (i = l & 31, h = l >> 5) holds four consecutive k of row i / column i: the
// four k of lane half 0 and the four of half 1 feed four MFMAs, any pairing of k values is a valid contraction), multiplies
// as the loads land, and wave 0 sums the eight partial accumulators in wave order and runs the shared epilogue (bias,
// ReLU mask of the layer below, statistics, stores).  No LDS staging: every operand element is used by exactly one MFMA
// lane, the re-reads across workgroups (A by N/32 of them, W by M/32) come from L2.
// Measured (tools/mlpbench.py): fp4 (1024 rows) forward 43.0 -> 24.9 us, forward + backward 116 -> 97 us; fp3 (4096 rows:
// 1024 workgroups, every operand re-read 8 ... 128 times from L2) 36.9 -> 45.8 and 134 -> 159 us: M <= 1024 only.
// NS = 8-wide k steps per wave (K <= 64 NS).  Prologues keep their per-k constants in registers beside the operands:
// PRO_BN_RELU up to NS = 8, PRO_BN_BWD (dz from g and z; explicit g only) at NS = 4.
template <int PRO, bool WT, int NS>
__global__ __launch_bounds__(512) void mlp_gemm_tiny_kernel(GemmArgs p)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ float sAcc[7][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
    const int row = row0 + l31, col = col0 + l31;
    const int rowc = min(row, p.M - 1), colc = min(col, p.N - 1);
    const bool row_ok = row < p.M, col_ok = col < p.N;
    const int kb = wave * (NS * 8) + 4 * half;               // this lane's first k
    float4 a[NS], b[NS], z[PRO == PRO_BN_BWD ? NS : 1], cs[PRO != PRO_NONE ? NS : 1], ch[PRO != PRO_NONE ? NS : 1];
    float4 cm[PRO == PRO_BN_BWD ? NS : 1], ci[PRO == PRO_BN_BWD ? NS : 1], cc1[PRO == PRO_BN_BWD ? NS : 1],
        cc2[PRO == PRO_BN_BWD ? NS : 1];
    // ---- every load of the slice (clamped addresses: a k past the end is masked when consumed)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = kb + 8 * s;
        const int kc = min(k, p.K - 4);
        if (PRO == PRO_BN_BWD) {
            a[s] = *reinterpret_cast<const float4 *>(p.x1 + (size_t)rowc * p.ld1 + kc);
            z[s] = *reinterpret_cast<const float4 *>(p.x2 + (size_t)rowc * p.ld2 + kc);
        } else {
            a[s] = kc < p.K1 ? *reinterpret_cast<const float4 *>(p.x1 + (size_t)rowc * p.ld1 + kc)
                             : *reinterpret_cast<const float4 *>(p.x2 + (size_t)rowc * p.ld2 + (kc - p.K1));
        }
        if (!WT) {
            b[s] = *reinterpret_cast<const float4 *>(p.w + (size_t)colc * p.ldw + kc);
        } else {
            const float *wp = p.w + (size_t)kc * p.ldw + colc;
            b[s] = make_float4(wp[0], wp[p.ldw], wp[2 * (size_t)p.ldw], wp[3 * (size_t)p.ldw]);
        }
        if (PRO != PRO_NONE) {
            cs[s] = *reinterpret_cast<const float4 *>(p.scale + kc);
            ch[s] = *reinterpret_cast<const float4 *>(p.shift + kc);
        }
        if (PRO == PRO_BN_BWD) {
            cm[s] = *reinterpret_cast<const float4 *>(p.mean + kc);
            ci[s] = *reinterpret_cast<const float4 *>(p.invstd + kc);
            cc1[s] = *reinterpret_cast<const float4 *>(p.c1 + kc);
            cc2[s] = *reinterpret_cast<const float4 *>(p.c2 + kc);
        }
    }
    __builtin_amdgcn_sched_barrier(0);       // all of them in flight before the first is waited for (left alone, the
                                             // scheduler sinks each load to its use and the slice runs step by step)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // ---- multiply in the order the loads were issued
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool k_ok = kb + 8 * s < p.K;
        float4 av = a[s];
        if (PRO == PRO_BN_RELU) {
            av.x = fmaxf(cs[s].x * av.x + ch[s].x, 0.f);
            av.y = fmaxf(cs[s].y * av.y + ch[s].y, 0.f);
            av.z = fmaxf(cs[s].z * av.z + ch[s].z, 0.f);
            av.w = fmaxf(cs[s].w * av.w + ch[s].w, 0.f);
        } else if (PRO == PRO_BN_BWD) {
            const float4 g = a[s], zz = z[s];
#define PN2_DZ(f) av.f = cs[s].f * (((cs[s].f * zz.f + ch[s].f) > 0.f ? g.f : 0.f) - cc1[s].f - (zz.f - cm[s].f) * ci[s].f * cc2[s].f)
            PN2_DZ(x); PN2_DZ(y); PN2_DZ(z); PN2_DZ(w);
#undef PN2_DZ
        }
        if (!(k_ok && row_ok)) av = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 bv = b[s];
        if (!(k_ok && col_ok)) bv = make_float4(0.f, 0.f, 0.f, 0.f);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    // ---- the eight K slices, summed in wave order by wave 0
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sAcc[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 7; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += sAcc[w][r][lane];
    float csum = 0.f, csq = 0.f;
    if (col_ok) gemm_store_block_any<PRO == PRO_BN_BWD>(p, acc, row0, half, col, p.mask_z != nullptr, csum, csq);
    if (!p.stat_partial) return;
    const float ssum = csum + __shfl_xor(csum, 32);
    const float sq = csq + __shfl_xor(csq, 32);
    if (half == 0 && col_ok) {
        p.stat_partial[((size_t)blockIdx.x * 2 + 0) * p.N + col] = ssum;
        p.stat_partial[((size_t)blockIdx.x * 2 + 1) * p.N + col] = sq;
    }
}

// Fixed-order partial sums with all loads issued before the first add: thread slice py takes partials py,
// py+32, ... (even ones into s0, odd ones into s1).  NJ = compile-time bound on the number per thread.
template <int NJ>
__device__ __forceinline__ void strided_sum(const float *__restrict__ base, size_t stride, int py, int P, float &s0, float &s1)
{
    float v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = base[(size_t)min(py + 32 * j, P - 1) * stride];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (py + 32 * j < P) { if (j & 1) s1 += v[j]; else s0 += v[j]; }
}
__device__ __forceinline__ void strided_sum_any(const float *__restrict__ base, size_t stride, int py, int P, float &s0, float &s1)
{
    if (P <= 32) strided_sum<1>(base, stride, py, P, s0, s1);
    else if (P <= 64) strided_sum<2>(base, stride, py, P, s0, s1);
    else if (P <= 128) strided_sum<4>(base, stride, py, P, s0, s1);
    else if (P <= 256) strided_sum<8>(base, stride, py, P, s0, s1);
    else if (P <= 512) strided_sum<16>(base, stride, py, P, s0, s1);
    else {
        int i = py;
        for (; i + 32 < P; i += 64) { s0 += base[(size_t)i * stride]; s1 += base[(size_t)(i + 32) * stride]; }
        if (i < P) s0 += base[(size_t)i * stride];
    }
}

// the same for two interleaved series (sum and sum of squares of one channel) in ONE memory round trip
template <int NJ>
__device__ __forceinline__ void strided_sum_pair(const float *__restrict__ a, const float *__restrict__ b, size_t stride, int py,
                                                 int P, float &a0, float &a1, float &b0, float &b1)
{
    float va[NJ], vb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const size_t o = (size_t)min(py + 32 * j, P - 1) * stride;
        va[j] = a[o];
        vb[j] = b[o];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (py + 32 * j < P) { if (j & 1) { a1 += va[j]; b1 += vb[j]; } else { a0 += va[j]; b0 += vb[j]; } }
}
__device__ __forceinline__ void strided_sum_pair_any(const float *__restrict__ a, const float *__restrict__ b, size_t stride,
                                                     int py, int P, float &a0, float &a1, float &b0, float &b1)
{
    if (P <= 32) strided_sum_pair<1>(a, b, stride, py, P, a0, a1, b0, b1);
    else if (P <= 64) strided_sum_pair<2>(a, b, stride, py, P, a0, a1, b0, b1);
    else if (P <= 128) strided_sum_pair<4>(a, b, stride, py, P, a0, a1, b0, b1);
    else if (P <= 256) strided_sum_pair<8>(a, b, stride, py, P, a0, a1, b0, b1);
    else if (P <= 512) strided_sum_pair<16>(a, b, stride, py, P, a0, a1, b0, b1);
    else { strided_sum_any(a, stride, py, P, a0, a1); strided_sum_any(b, stride, py, P, b0, b1); }
}

struct BnFinArgs {
    const float *partial;           // [P][2][C]; null: scale / shift are given (eval mode), nothing to finalize
    int P, C;
    double count;
    const float *gamma, *beta;
    float eps, momentum;
    const float *momentum_dev;
    float *running_mean, *running_var, *scale, *shift, *mean_out, *invstd_out;
    long long *num_batches_tracked;
};

// The 32 channels of column block `cb` by one 1024-thread workgroup (32 channels x 32 partial slices).  Every caller
// gets the same scale / shift bit for bit (in sScale / sShift after the barrier the CALLER places); only a `writer`
// stores them and updates the running estimates.
__device__ __forceinline__ void bn_finalize_block(const BnFinArgs &a, int cb, bool writer, float *sScale, float *sShift)
{
    const int cl = threadIdx.x & 31, py = threadIdx.x >> 5;
    const int c = cb * 32 + cl;
    const int C = a.C, P = a.P;
    if (!a.partial) {
        if (py == 0) { sScale[cl] = c < C ? a.scale[c] : 0.f; sShift[cl] = c < C ? a.shift[c] : 0.f; }
        return;
    }
    // momentum: a device word when given (so that a captured launch follows the per-epoch schedule of the reference
    // loop, localfunctions.py:191-195), else the host value; negative = nn.BatchNorm(momentum=None): the cumulative
    // average 1 / num_batches_tracked, the counter then being incremented by the CALLER beforehand (all workgroups
    // read it here)
    float momentum = a.momentum;
    if (a.momentum_dev) momentum = *a.momentum_dev;
    const bool cumulative = momentum < 0.0f;
    if (cumulative) momentum = (a.num_batches_tracked && *a.num_batches_tracked > 0) ? 1.0f / (float)*a.num_batches_tracked : 0.0f;
    if (!cumulative && a.num_batches_tracked && writer && cb == 0 && threadIdx.x == 0) *a.num_batches_tracked += 1;
    __shared__ double sS[32][33], sQ[32][33];
    double s = 0.0, q = 0.0;
    // the layer parameters travel with the partials (one memory round trip for the whole kernel, which runs 22
    // times per step), not after the reduction
    const int cc = min(c, C - 1);
    const float g = a.gamma ? a.gamma[cc] : 1.0f, b = a.beta ? a.beta[cc] : 0.0f;
    const float rm = (writer && a.running_mean) ? a.running_mean[cc] : 0.f, rv = (writer && a.running_var) ? a.running_var[cc] : 0.f;
    if (c < C) {
        // a thread's partials are all loaded before the first add: the loads are independent, a load-add loop
        // would pay one memory round trip per iteration
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;            // <= 8 terms each: fp32 is exact enough here
        strided_sum_pair_any(a.partial + c, a.partial + C + c, (size_t)2 * C, py, P, s0, s1, q0, q1);
        s = (double)s0 + (double)s1;
        q = (double)q0 + (double)q1;
    }
    sS[py][cl] = s;
    sQ[py][cl] = q;
    __syncthreads();
    if (py == 0) {
        for (int i = 1; i < 32; ++i) { s += sS[i][cl]; q += sQ[i][cl]; }
        const double mean = s / a.count;
        double var = q / a.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
        const float sc = g * invstd, sh = b - (float)mean * g * invstd;
        sScale[cl] = sc;
        sShift[cl] = sh;
        if (writer && c < C) {
            a.scale[c] = sc;
            a.shift[c] = sh;
            if (a.mean_out) a.mean_out[c] = (float)mean;
            if (a.invstd_out) a.invstd_out[c] = invstd;
            if (a.running_mean) a.running_mean[c] = (1.0f - momentum) * rm + momentum * (float)mean;
            if (a.running_var) {
                const double unbiased = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
                a.running_var[c] = (1.0f - momentum) * rv + momentum * (float)unbiased;
            }
        }
    }
}

// partial[P][2][C] -> BatchNorm coefficients of a train-mode layer (models/pointnet2_utils.py:198 /
// :314 with nn.BatchNorm semantics: biased variance for normalisation, unbiased for the running
// estimate, running = (1-m)*running + m*batch).  One thread per channel, partials summed in
// double in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void bn_finalize_kernel(BnFinArgs a)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ float sScale[32], sShift[32];
    bn_finalize_block(a, blockIdx.x, true, sScale, sShift);
}

// The same followed, in the same launch, by the output of the stack: workgroup (cb, slice) finalizes column block cb
// (redundantly over the slices: a few hundred KB of L2 hits; slice 0 is the writer) and then emits its slice of the
// rows for those 32 channels --
//   apply  (pool_max == null): y[r][c] = max(scale*z[r][c] + shift, 0)
//   select (pool_max != null): from the per-group extrema the GEMM epilogue left (GemmArgs::pool_max ...):
//          y = max(scale*(scale >= 0 ? zmax : zmin) + shift, 0), argk = the row that held it (0 when y == 0 or
//          scale == 0: every row of the group ties and torch.max returns the first).
// One launch instead of bn_finalize + bn_relu_out, and the pooled form never re-reads z.
struct BnOutArgs {
    const float *z;
    int ldz;
    const float *pool_max, *pool_min;
    const unsigned char *pool_amax, *pool_amin;
    long long rows_out;
    float *y;                       // [rows_out][C]
    unsigned char *argk;            // [rows_out][C], select only
};

__global__ __launch_bounds__(1024) void bn_finalize_out_kernel(BnFinArgs a, BnOutArgs o)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ float sScale[32], sShift[32];
    bn_finalize_block(a, blockIdx.x, blockIdx.y == 0, sScale, sShift);
    __syncthreads();
    const int cq = (threadIdx.x & 7) * 4, c = blockIdx.x * 32 + cq;
    if (c >= a.C) return;                                       // C % 4 == 0
    const float4 sc = make_float4(sScale[cq], sScale[cq + 1], sScale[cq + 2], sScale[cq + 3]);
    const float4 sh = make_float4(sShift[cq], sShift[cq + 1], sShift[cq + 2], sShift[cq + 3]);
    const long long per = (o.rows_out + gridDim.y - 1) / gridDim.y;
    const long long r1 = min(o.rows_out, per * (blockIdx.y + 1));
    const int C = a.C;
    if (!o.pool_max) {
#pragma unroll 4
        for (long long r = per * blockIdx.y + (threadIdx.x >> 3); r < r1; r += 128) {
            const float4 v = *reinterpret_cast<const float4 *>(o.z + (size_t)r * o.ldz + c);
            float4 y;
            y.x = fmaxf(sc.x * v.x + sh.x, 0.f);
            y.y = fmaxf(sc.y * v.y + sh.y, 0.f);
            y.z = fmaxf(sc.z * v.z + sh.z, 0.f);
            y.w = fmaxf(sc.w * v.w + sh.w, 0.f);
            pn2::store_rows4(o.y, (size_t)r * C + c, y, (size_t)o.rows_out * C * sizeof(float));
        }
    } else {
#pragma unroll 2
        for (long long r = per * blockIdx.y + (threadIdx.x >> 3); r < r1; r += 128) {
            const size_t e = (size_t)r * C + c;
            const float4 hi = *reinterpret_cast<const float4 *>(o.pool_max + e);
            const float4 lo = *reinterpret_cast<const float4 *>(o.pool_min + e);
            const uchar4 ah = *reinterpret_cast<const uchar4 *>(o.pool_amax + e);
            const uchar4 al = *reinterpret_cast<const uchar4 *>(o.pool_amin + e);
            float4 y;
            uchar4 k;
#define PN2_SEL(f) do { const bool up = sc.f >= 0.f; y.f = fmaxf(sc.f * (up ? hi.f : lo.f) + sh.f, 0.f); \
                        k.f = (y.f > 0.f && sc.f != 0.f) ? (up ? ah.f : al.f) : (unsigned char)0; } while (0)
            PN2_SEL(x); PN2_SEL(y); PN2_SEL(z); PN2_SEL(w);
#undef PN2_SEL
            pn2::store_rows4(o.y, e, y, (size_t)o.rows_out * C * sizeof(float));
            *reinterpret_cast<uchar4 *>(o.argk + e) = k;
        }
    }
}

// eval-mode coefficients from the running estimates
__global__ __launch_bounds__(256) void bn_eval_coeff_kernel(int C, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta,
                                                            const float *__restrict__ running_mean,
                                                            const float *__restrict__ running_var, float eps,
                                                            float *__restrict__ scale, float *__restrict__ shift)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(running_var[c] + eps);
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    scale[c] = g * invstd;
    shift[c] = b - running_mean[c] * g * invstd;
}

// y[r, c] = max(scale[c]*z[r,c] + shift[c], 0), optionally max-pooled over groups of K rows
// (torch.max(new_points, 2)[0], models/pointnet2_utils.py:200) with the winning k recorded.
template <bool POOL>
__global__ __launch_bounds__(256) void bn_relu_out_kernel(const float *__restrict__ z, long long rows_out, int C, int K,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, float *__restrict__ y,
                                                          unsigned char *__restrict__ argk)
{
    const int c4n = C >> 2;
    const long long total = rows_out * c4n;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long ro = t / c4n;
        const int c = (int)(t - ro * c4n) * 4;
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
        if (!POOL) {
            const float4 v = *reinterpret_cast<const float4 *>(z + (size_t)ro * C + c);
            float4 o;
            o.x = fmaxf(sc.x * v.x + sh.x, 0.f);
            o.y = fmaxf(sc.y * v.y + sh.y, 0.f);
            o.z = fmaxf(sc.z * v.z + sh.z, 0.f);
            o.w = fmaxf(sc.w * v.w + sh.w, 0.f);
            *reinterpret_cast<float4 *>(y + (size_t)ro * C + c) = o;
        } else {
            float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            uchar4 bk = make_uchar4(0, 0, 0, 0);
            const float *base = z + (size_t)ro * K * C + c;
            for (int k = 0; k < K; ++k) {
                const float4 v = *reinterpret_cast<const float4 *>(base + (size_t)k * C);
                const float ox = fmaxf(sc.x * v.x + sh.x, 0.f), oy = fmaxf(sc.y * v.y + sh.y, 0.f);
                const float oz = fmaxf(sc.z * v.z + sh.z, 0.f), ow = fmaxf(sc.w * v.w + sh.w, 0.f);
                if (ox > best.x) { best.x = ox; bk.x = (unsigned char)k; }      // first maximum wins, like torch.max
                if (oy > best.y) { best.y = oy; bk.y = (unsigned char)k; }
                if (oz > best.z) { best.z = oz; bk.z = (unsigned char)k; }
                if (ow > best.w) { best.w = ow; bk.w = (unsigned char)k; }
            }
            *reinterpret_cast<float4 *>(y + (size_t)ro * C + c) = best;
            if (argk) *reinterpret_cast<uchar4 *>(argk + (size_t)ro * C + c) = bk;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// dW[n][k] = sum_m dz[m][n] * a[m][k]   (+ column Kact == bias gradient: a[m][Kact] := 1)
// dz comes from the PRO_BN_BWD formula (explicit g or pooled g), a = act(x) like the forward A
// operand.  Output block 64(n) x 64(k) per workgroup; each wave reduces its own 32 rows of every
// 128-row tile (MFMA k-axis = m), the 4 waves are summed through LDS at the end, and every
// workgroup writes one partial [Co][Kact+1] slab (summed in fixed order by dw_reduce_kernel).
struct DwArgs {
    const float *g, *z;             // dz sources: g [M][N] (or pooled [M/pool_k][N]), z [M][N]
    int ldg, ldz;
    const float *scale, *shift, *mean, *invstd, *c1, *c2;   // of THIS layer (N channels)
    const unsigned char *argk;
    int pool_k;
    const float *x1, *x2;           // activation sources of the layer input, [M][K1] | [M][K2]
    int ld1, ld2, K1, K2;
    const float *ascale, *ashift;   // previous layer's BN coefficients (null: input is already an activation)
    float *partial;                 // [gridDim.x][N][K1+K2+1]
    int M, N;
};

constexpr int DW_BN = 64, DW_BK = 64, DW_LD = 68;

// VEC4: every row pitch / column split is a multiple of 4 floats and 16-B aligned, so a thread
// owns 4 fixed columns (its BatchNorm constants live in registers for the whole kernel) and
// stages 8 rows of them per tile with float4 loads, all issued before the first is consumed.
template <bool VEC4, bool POOLED>
__global__ __launch_bounds__(MLP_THREADS) void mlp_dw_kernel(DwArgs p)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ __attribute__((aligned(16))) float sD[MLP_BM * DW_LD];     // dz tile   [128 m][64 n]
    __shared__ __attribute__((aligned(16))) float sX[MLP_BM * DW_LD];     // act tile  [128 m][64 k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.y * DW_BN, k0 = blockIdx.z * DW_BK;
    const int Kact = p.K1 + p.K2, Kout = Kact + 1;
    const int ntiles = (p.M + MLP_BM - 1) / MLP_BM;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // per-thread column constants (VEC4 path)
    const int c4 = (tid & 15) * 4, rb = tid >> 4;
    const int n4 = n0 + c4, k4 = k0 + c4;
    const bool n_ok = n4 < p.N;                      // N % 4 == 0 on this path
    const bool k_act = k4 < Kact, k_one = !VEC4 && k4 == Kact;
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);      // VEC4: running column sums of dz (bias gradient)
    const bool from1 = k4 < p.K1;
    float4 sc = make_float4(0, 0, 0, 0), sh = sc, mu = sc, is = sc, a1 = sc, a2 = sc, asc = sc, ash = sc;
    if (VEC4) {
        if (n_ok) {
            sc = *reinterpret_cast<const float4 *>(p.scale + n4);
            sh = *reinterpret_cast<const float4 *>(p.shift + n4);
            mu = *reinterpret_cast<const float4 *>(p.mean + n4);
            is = *reinterpret_cast<const float4 *>(p.invstd + n4);
            a1 = *reinterpret_cast<const float4 *>(p.c1 + n4);
            a2 = *reinterpret_cast<const float4 *>(p.c2 + n4);
        }
        if (k_act && p.ascale) {
            asc = *reinterpret_cast<const float4 *>(p.ascale + k4);
            ash = *reinterpret_cast<const float4 *>(p.ashift + k4);
        }
    }

    float4 gv[8], zv[8], xv[8];
    uchar4 av[8];
    // VEC4: global loads of a tile into registers (issued one tile ahead of their use).  Unconditional
    // (clamped addresses; rows / columns outside the problem are masked when the values are consumed): a
    // load under a divergent branch is waited for at the end of the branch, which serialised the 8 passes.
    const int n4c = n_ok ? n4 : 0, k4c = k_act ? k4 : 0;
    const float *xsrc = from1 ? p.x1 : (p.x2 ? p.x2 : p.x1);
    const int xld = from1 ? p.ld1 : (p.x2 ? p.ld2 : p.ld1), xcol = k_act ? (from1 ? k4 : k4 - p.K1) : 0;
    auto issue = [&](int tile) {
        const int row0 = tile * MLP_BM;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = min(row0 + rb + 16 * i, p.M - 1);
            if (POOLED) {                             // g / argk per centroid
                const int cent = row / p.pool_k;
                gv[i] = *reinterpret_cast<const float4 *>(p.g + (size_t)cent * p.ldg + n4c);
                av[i] = *reinterpret_cast<const uchar4 *>(p.argk + (size_t)cent * p.N + n4c);
            } else {
                gv[i] = *reinterpret_cast<const float4 *>(p.g + (size_t)row * p.ldg + n4c);
            }
            zv[i] = *reinterpret_cast<const float4 *>(p.z + (size_t)row * p.ldz + n4c);
            xv[i] = *reinterpret_cast<const float4 *>(xsrc + (size_t)row * xld + xcol);
        }
    };
    if (VEC4 && (int)blockIdx.x < ntiles) issue(blockIdx.x);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * MLP_BM;
        if (VEC4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = rb + 16 * i;
                const int row = row0 + r;
                float4 dv = make_float4(0, 0, 0, 0), av4 = make_float4(0, 0, 0, 0);
                if (row < p.M) {
                    if (n_ok) {
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
                    if (k_act) {
                        av4 = xv[i];
                        if (p.ascale) {
                            av4.x = fmaxf(asc.x * av4.x + ash.x, 0.f);
                            av4.y = fmaxf(asc.y * av4.y + ash.y, 0.f);
                            av4.z = fmaxf(asc.z * av4.z + ash.z, 0.f);
                            av4.w = fmaxf(asc.w * av4.w + ash.w, 0.f);
                        }
                    } else if (k_one) {
                        av4.x = 1.0f;                            // bias-gradient column
                    }
                }
                *reinterpret_cast<float4 *>(&sD[r * DW_LD + c4]) = dv;
                *reinterpret_cast<float4 *>(&sX[r * DW_LD + c4]) = av4;
            }
            __syncthreads();
            if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);   // next tile's loads fly under the MFMAs
        } else {
            for (int e = tid; e < MLP_BM * 64; e += MLP_THREADS) {
                const int r = e >> 6, c = e & 63;
                const int row = row0 + r;
                float dv = 0.f, xv = 0.f;
                if (row < p.M) {
                    const int n = n0 + c;
                    if (n < p.N) {
                        float g;
                        if (p.argk) {
                            const int cent = row / p.pool_k, kk = row - cent * p.pool_k;
                            g = p.argk[(size_t)cent * p.N + n] == kk ? p.g[(size_t)cent * p.ldg + n] : 0.f;
                        } else {
                            g = p.g[(size_t)row * p.ldg + n];
                        }
                        const float z = p.z[(size_t)row * p.ldz + n];
                        const float s1 = p.scale[n];
                        dv = s1 * (((s1 * z + p.shift[n]) > 0.f ? g : 0.f) - p.c1[n] - (z - p.mean[n]) * p.invstd[n] * p.c2[n]);
                    }
                    const int k = k0 + c;
                    if (k < Kact) {
                        xv = k < p.K1 ? p.x1[(size_t)row * p.ld1 + k] : p.x2[(size_t)row * p.ld2 + (k - p.K1)];
                        if (p.ascale) xv = fmaxf(p.ascale[k] * xv + p.ashift[k], 0.f);
                    } else if (k == Kact) {
                        xv = 1.0f;                               // bias-gradient column
                    }
                }
                sD[r * DW_LD + c] = dv;
                sX[r * DW_LD + c] = xv;
            }
            __syncthreads();
        }
        // MFMA: i = n (dz column), j = k (act column), reduction over this wave's 32 rows
        const float *dBase = &sD[(wave * 32 + 16 * half) * DW_LD];
        const float *xBase = &sX[(wave * 32 + 16 * half) * DW_LD];
        // operand read-ahead (one step) as in mlp_gemm_rows32_kernel
        float a0 = dBase[l31], a1v = dBase[32 + l31], b0 = xBase[l31], b1 = xBase[32 + l31];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float a0n = a0, a1n = a1v, b0n = b0, b1n = b1;
            if (t + 1 < 16) {
                a0n = dBase[(t + 1) * DW_LD + l31]; a1n = dBase[(t + 1) * DW_LD + 32 + l31];
                b0n = xBase[(t + 1) * DW_LD + l31]; b1n = xBase[(t + 1) * DW_LD + 32 + l31];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b1, acc[1][1], 0, 0, 0);
            a0 = a0n; a1v = a1n; b0 = b0n; b1 = b1n;
        }
        __syncthreads();
    }
    // cross-wave sum through LDS
    float *red = sD;                                             // [64][DW_LD] accumulator image
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const int k = b * 32 + l31;
                        if (w == 0) red[n * DW_LD + k] = acc[a][b][r];
                        else red[n * DW_LD + k] += acc[a][b][r];
                    }
        }
        __syncthreads();
    }
    float *out = p.partial + (size_t)blockIdx.x * p.N * Kout;
    const int klim = VEC4 ? Kact : Kout;                         // VEC4: the bias column comes from dbs below
    for (int e = tid; e < 64 * 64; e += MLP_THREADS) {
        const int n = e >> 6, k = e & 63;
        if (n0 + n < p.N && k0 + k < klim) pn2::store_rows(&out[(size_t)(n0 + n) * Kout + k0 + k], red[n * DW_LD + k]);
    }
    if (VEC4 && blockIdx.z == 0) {
        // 16 threads (rb = 0..15) hold partial sums for the same 4 columns: combine through LDS
        __syncthreads();
        float *cs = sX;                                           // [16][64]
        *reinterpret_cast<float4 *>(&cs[rb * 64 + c4]) = dbs;
        __syncthreads();
        if (tid < 64 && n0 + tid < p.N) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += cs[i * 64 + tid];
            out[(size_t)(n0 + tid) * Kout + Kact] = t;
        }
    }
}

// Split-role form of mlp_dw_kernel<true, POOLED> (the VEC4 staging, the same tiles, the same slabs): in the form above
// the four waves of a workgroup stage a tile (loads, dz from g and z, BatchNorm + ReLU of the input: ~800 vector
// instructions a wave), meet at a barrier, multiply (64 MFMAs a wave), meet again -- the matrix pipe idles while the tile
// is staged (25-31 % busy).  Here waves 4..11 only stage (tile i+1 into the other LDS buffer; a thread owns four rows of a
// tile and keeps the loads of TWO later tiles in flight in two register sets, so a load has a whole tile period to land)
// and waves 0..3 only multiply (tile i), one barrier per tile; 139 KB of LDS, one 12-wave workgroup per CU.  (With four
// staging waves and one register set the multipliers waited two thirds of the time: rocprofv3 SQ_WAIT_ANY 43 %.)
constexpr int DWS_THREADS = 768;
constexpr int DWS_TILE = MLP_BM * DW_LD;                   // floats of one staged operand tile
constexpr size_t DWS_LDS = (size_t)4 * DWS_TILE * sizeof(float);

template <bool POOLED>
__global__ __launch_bounds__(DWS_THREADS) void mlp_dw_split_kernel(DwArgs p)
{
    PN2_MAIN_BRANCH_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) float dws_lds[];        // [2 buffers][dz tile | act tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.y * DW_BN, k0 = blockIdx.z * DW_BK;
    const int Kact = p.K1 + p.K2, Kout = Kact + 1;
    const int ntiles = (p.M + MLP_BM - 1) / MLP_BM;
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    float *out = p.partial + (size_t)blockIdx.x * p.N * Kout;

    if (wave >= 4) {
        // ---------------- stagers: thread st owns 4 dz columns and 4 act columns of rows rb, rb + 32, rb + 64, rb + 96
        const int st = tid - 256;
        const int c4 = (st & 15) * 4, rb = st >> 4;
        const int n4 = n0 + c4, k4 = k0 + c4;
        const bool n_ok = n4 < p.N, k_act = k4 < Kact, from1 = k4 < p.K1;
        float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 sc = make_float4(0, 0, 0, 0), sh = sc, mu = sc, is = sc, a1 = sc, a2 = sc, asc = sc, ash = sc;
        if (n_ok) {
            sc = *reinterpret_cast<const float4 *>(p.scale + n4);
            sh = *reinterpret_cast<const float4 *>(p.shift + n4);
            mu = *reinterpret_cast<const float4 *>(p.mean + n4);
            is = *reinterpret_cast<const float4 *>(p.invstd + n4);
            a1 = *reinterpret_cast<const float4 *>(p.c1 + n4);
            a2 = *reinterpret_cast<const float4 *>(p.c2 + n4);
        }
        if (k_act && p.ascale) {
            asc = *reinterpret_cast<const float4 *>(p.ascale + k4);
            ash = *reinterpret_cast<const float4 *>(p.ashift + k4);
        }
        const int n4c = n_ok ? n4 : 0;
        const float *xsrc = from1 ? p.x1 : (p.x2 ? p.x2 : p.x1);
        const int xld = from1 ? p.ld1 : (p.x2 ? p.ld2 : p.ld1), xcol = k_act ? (from1 ? k4 : k4 - p.K1) : 0;
        struct Regs { float4 g[4], z[4], x[4]; uchar4 a[4]; };
        Regs R0, R1;
        auto issue = [&](Regs &R, int i) {                     // unconditional loads (clamped rows), masked on use
            const int row0 = ((int)blockIdx.x + i * (int)gridDim.x) * MLP_BM;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = min(row0 + rb + 32 * u, p.M - 1);
                if (POOLED) {
                    const int cent = row / p.pool_k;
                    R.g[u] = *reinterpret_cast<const float4 *>(p.g + (size_t)cent * p.ldg + n4c);
                    R.a[u] = *reinterpret_cast<const uchar4 *>(p.argk + (size_t)cent * p.N + n4c);
                } else {
                    R.g[u] = *reinterpret_cast<const float4 *>(p.g + (size_t)row * p.ldg + n4c);
                }
                R.z[u] = *reinterpret_cast<const float4 *>(p.z + (size_t)row * p.ldz + n4c);
                R.x[u] = *reinterpret_cast<const float4 *>(xsrc + (size_t)row * xld + xcol);
            }
        };
        auto commit = [&](const Regs &R, int i) {
            const int row0 = ((int)blockIdx.x + i * (int)gridDim.x) * MLP_BM;
            float *sD = dws_lds + (i & 1) * 2 * DWS_TILE, *sX = sD + DWS_TILE;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rb + 32 * u;
                const int row = row0 + r;
                float4 dv = make_float4(0, 0, 0, 0), av4 = make_float4(0, 0, 0, 0);
                if (row < p.M) {
                    if (n_ok) {
                        float4 g = R.g[u];
                        if (POOLED) {
                            const int kk = row % p.pool_k;
                            g.x = R.a[u].x == kk ? g.x : 0.f;
                            g.y = R.a[u].y == kk ? g.y : 0.f;
                            g.z = R.a[u].z == kk ? g.z : 0.f;
                            g.w = R.a[u].w == kk ? g.w : 0.f;
                        }
                        const float4 z = R.z[u];
#define PN2_DZ(f) dv.f = sc.f * (((sc.f * z.f + sh.f) > 0.f ? g.f : 0.f) - a1.f - (z.f - mu.f) * is.f * a2.f)
                        PN2_DZ(x); PN2_DZ(y); PN2_DZ(z); PN2_DZ(w);
#undef PN2_DZ
                        dbs.x += dv.x; dbs.y += dv.y; dbs.z += dv.z; dbs.w += dv.w;
                    }
                    if (k_act) {
                        av4 = R.x[u];
                        if (p.ascale) {
                            av4.x = fmaxf(asc.x * av4.x + ash.x, 0.f);
                            av4.y = fmaxf(asc.y * av4.y + ash.y, 0.f);
                            av4.z = fmaxf(asc.z * av4.z + ash.z, 0.f);
                            av4.w = fmaxf(asc.w * av4.w + ash.w, 0.f);
                        }
                    }
                }
                *reinterpret_cast<float4 *>(&sD[r * DW_LD + c4]) = dv;
                *reinterpret_cast<float4 *>(&sX[r * DW_LD + c4]) = av4;
            }
        };
        // tile j lives in register set j & 1 from its issue to its commit
        if (my_tiles > 0) issue(R0, 0);
        if (my_tiles > 1) issue(R1, 1);
        if (my_tiles > 0) commit(R0, 0);
        if (my_tiles > 2) issue(R0, 2);
        __syncthreads();                                          // tile 0 is staged
        for (int i = 0; i < my_tiles; i += 2) {
            if (i + 1 < my_tiles) commit(R1, i + 1);              // into the buffer the multipliers left at the last barrier
            if (i + 3 < my_tiles) issue(R1, i + 3);
            __syncthreads();
            if (i + 1 < my_tiles) {
                if (i + 2 < my_tiles) commit(R0, i + 2);
                if (i + 4 < my_tiles) issue(R0, i + 4);
                __syncthreads();
            }
        }
        // bias gradient: 32 stager threads hold partial sums of the same 4 columns; the buffers are free now
        __syncthreads();                                          // (the multipliers' accumulator images are written)
        float *cs = dws_lds + 4 * 64 * DW_LD;                     // [32][64] behind the four accumulator images
        *reinterpret_cast<float4 *>(&cs[rb * 64 + c4]) = dbs;
        __syncthreads();
        if (blockIdx.z == 0 && st < 64 && n0 + st < p.N) {
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < 32; ++u) t += cs[u * 64 + st];
            out[(size_t)(n0 + st) * Kout + Kact] = t;
        }
    } else {
        // ---------------- multipliers: wave w reduces over rows 32 w .. 32 w + 31 of every tile, all 64 x 64 outputs
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        __syncthreads();
        for (int i = 0; i < my_tiles; ++i) {
            const float *sD = dws_lds + (i & 1) * 2 * DWS_TILE, *sX = sD + DWS_TILE;
            const float *dBase = &sD[(wave * 32 + 16 * half) * DW_LD];
            const float *xBase = &sX[(wave * 32 + 16 * half) * DW_LD];
            float a0 = dBase[l31], a1v = dBase[32 + l31], b0 = xBase[l31], b1 = xBase[32 + l31];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float a0n = a0, a1n = a1v, b0n = b0, b1n = b1;
                if (t + 1 < 16) {
                    a0n = dBase[(t + 1) * DW_LD + l31]; a1n = dBase[(t + 1) * DW_LD + 32 + l31];
                    b0n = xBase[(t + 1) * DW_LD + l31]; b1n = xBase[(t + 1) * DW_LD + 32 + l31];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b1, acc[1][1], 0, 0, 0);
                a0 = a0n; a1v = a1n; b0 = b0n; b1 = b1n;
            }
            __syncthreads();
        }
        // the four waves' accumulators as four images in LDS (the tile buffers are free: everybody passed the last barrier)
        float *img = dws_lds + wave * 64 * DW_LD;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    img[n * DW_LD + b * 32 + l31] = acc[a][b][r];
                }
        __syncthreads();
        __syncthreads();                                          // (the stagers' column sums are written)
    }
    // every thread: sum the four images in wave order (the order of the one-role form) and store the slab block
    for (int e = tid; e < 64 * 64; e += DWS_THREADS) {
        const int n = e >> 6, k = e & 63;
        if (n0 + n < p.N && k0 + k < Kact) {
            const float *q = dws_lds + n * DW_LD + k;
            float v = q[0];
            v += q[64 * DW_LD]; v += q[2 * 64 * DW_LD]; v += q[3 * 64 * DW_LD];
            pn2::store_rows(&out[(size_t)(n0 + n) * Kout + k0 + k], v);
        }
    }
}

// The same roles on 128 (n) x 128 (k) output blocks and 64-row tiles (N, K >= 128): a staged byte feeds twice the MFMAs
// of the 64 x 64 form -- that form needs 98 KB per CU every 1.7 us to keep the matrix pipe busy (14.7 TB/s over the chip;
// it gets ~3.5 out of L2 / MALL, which is what its 25-31 % busy fraction is).  Multiplier wave (wn, wk) owns one 64 x 64
// quadrant over all 64 rows of a tile (128 MFMAs a tile, no cross-wave sum at the end); the eight staging waves stage
// 64 rows x (128 dz + 128 act) columns: the same bytes per tile as above in twice the time.
// Measured with the roles switched off one at a time (fp2's first layer, M = 16384, 256 x 384, 47.5 us): neither role
// 13.9 us (launch, first tile, 16.5 MB of slabs), staging alone 23.7, multiplying alone 48.4: whole = fixed + max of the
// two, the stagers' vector instructions fit into the gaps of this multiplier loop (an MFMA every ~103 clocks: a wave can
// issue nothing else while its MFMA holds the issue slot, and four ds_read_b32 + moves per four MFMAs cost ~40 clocks
// apiece).  A loop with one ds_read_b128 + one ds_read_b64 per EIGHT MFMAs multiplies alone in 38.3 us -- and the whole
// kernel still takes 47.5: with the matrix pipe saturated the staging no longer overlaps, the times ADD (fixed 15.0 +
// staging 9.7 + multiplying 23.3).  The clock stays at 2.39 GHz throughout (tools/bqlab/lab6.hip).  The simple loop
// stays (profiles/r03/dw_split128_roles_switched_off.log, dw_split128_wide_operand_reads_withdrawn.patch).
constexpr int DWW_ROWS = 64, DWW_LD = 132;
constexpr int DWW_TILE = DWW_ROWS * DWW_LD;
constexpr size_t DWW_LDS = (size_t)4 * DWW_TILE * sizeof(float);

template <bool POOLED>
__global__ __launch_bounds__(DWS_THREADS) void mlp_dw_split128_kernel(DwArgs p)
{
    PN2_MAIN_BRANCH_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) float dws_lds[];        // [2 buffers][dz tile | act tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.y * 128, k0 = blockIdx.z * 128;
    const int Kact = p.K1 + p.K2, Kout = Kact + 1;
    const int ntiles = (p.M + DWW_ROWS - 1) / DWW_ROWS;
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    float *out = p.partial + (size_t)blockIdx.x * p.N * Kout;

    if (wave >= 4) {
        // ---------------- stagers: thread st owns 4 dz columns and 4 act columns of rows rb, rb + 16, rb + 32, rb + 48
        const int st = tid - 256;
        const int c4 = (st & 31) * 4, rb = st >> 5;
        const int n4 = n0 + c4, k4 = k0 + c4;
        const bool n_ok = n4 < p.N, k_act = k4 < Kact, from1 = k4 < p.K1;
        float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 sc = make_float4(0, 0, 0, 0), sh = sc, mu = sc, is = sc, a1 = sc, a2 = sc, asc = sc, ash = sc;
        if (n_ok) {
            sc = *reinterpret_cast<const float4 *>(p.scale + n4);
            sh = *reinterpret_cast<const float4 *>(p.shift + n4);
            mu = *reinterpret_cast<const float4 *>(p.mean + n4);
            is = *reinterpret_cast<const float4 *>(p.invstd + n4);
            a1 = *reinterpret_cast<const float4 *>(p.c1 + n4);
            a2 = *reinterpret_cast<const float4 *>(p.c2 + n4);
        }
        if (k_act && p.ascale) {
            asc = *reinterpret_cast<const float4 *>(p.ascale + k4);
            ash = *reinterpret_cast<const float4 *>(p.ashift + k4);
        }
        const int n4c = n_ok ? n4 : 0;
        const float *xsrc = from1 ? p.x1 : (p.x2 ? p.x2 : p.x1);
        const int xld = from1 ? p.ld1 : (p.x2 ? p.ld2 : p.ld1), xcol = k_act ? (from1 ? k4 : k4 - p.K1) : 0;
        struct Regs { float4 g[4], z[4], x[4]; uchar4 a[4]; };
        Regs R0, R1;
        auto issue = [&](Regs &R, int i) {
            const int row0 = ((int)blockIdx.x + i * (int)gridDim.x) * DWW_ROWS;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = min(row0 + rb + 16 * u, p.M - 1);
                if (POOLED) {
                    const int cent = row / p.pool_k;
                    R.g[u] = *reinterpret_cast<const float4 *>(p.g + (size_t)cent * p.ldg + n4c);
                    R.a[u] = *reinterpret_cast<const uchar4 *>(p.argk + (size_t)cent * p.N + n4c);
                } else {
                    R.g[u] = *reinterpret_cast<const float4 *>(p.g + (size_t)row * p.ldg + n4c);
                }
                R.z[u] = *reinterpret_cast<const float4 *>(p.z + (size_t)row * p.ldz + n4c);
                R.x[u] = *reinterpret_cast<const float4 *>(xsrc + (size_t)row * xld + xcol);
            }
        };
        auto commit = [&](const Regs &R, int i) {
            const int row0 = ((int)blockIdx.x + i * (int)gridDim.x) * DWW_ROWS;
            float *sD = dws_lds + (i & 1) * 2 * DWW_TILE, *sX = sD + DWW_TILE;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rb + 16 * u;
                const int row = row0 + r;
                float4 dv = make_float4(0, 0, 0, 0), av4 = make_float4(0, 0, 0, 0);
                if (row < p.M) {
                    if (n_ok) {
                        float4 g = R.g[u];
                        if (POOLED) {
                            const int kk = row % p.pool_k;
                            g.x = R.a[u].x == kk ? g.x : 0.f;
                            g.y = R.a[u].y == kk ? g.y : 0.f;
                            g.z = R.a[u].z == kk ? g.z : 0.f;
                            g.w = R.a[u].w == kk ? g.w : 0.f;
                        }
                        const float4 z = R.z[u];
#define PN2_DZ(f) dv.f = sc.f * (((sc.f * z.f + sh.f) > 0.f ? g.f : 0.f) - a1.f - (z.f - mu.f) * is.f * a2.f)
                        PN2_DZ(x); PN2_DZ(y); PN2_DZ(z); PN2_DZ(w);
#undef PN2_DZ
                        dbs.x += dv.x; dbs.y += dv.y; dbs.z += dv.z; dbs.w += dv.w;
                    }
                    if (k_act) {
                        av4 = R.x[u];
                        if (p.ascale) {
                            av4.x = fmaxf(asc.x * av4.x + ash.x, 0.f);
                            av4.y = fmaxf(asc.y * av4.y + ash.y, 0.f);
                            av4.z = fmaxf(asc.z * av4.z + ash.z, 0.f);
                            av4.w = fmaxf(asc.w * av4.w + ash.w, 0.f);
                        }
                    }
                }
                *reinterpret_cast<float4 *>(&sD[r * DWW_LD + c4]) = dv;
                *reinterpret_cast<float4 *>(&sX[r * DWW_LD + c4]) = av4;
            }
        };
        if (my_tiles > 0) issue(R0, 0);
        if (my_tiles > 1) issue(R1, 1);
        if (my_tiles > 0) commit(R0, 0);
        if (my_tiles > 2) issue(R0, 2);
        __syncthreads();                                          // tile 0 is staged
        for (int i = 0; i < my_tiles; i += 2) {
            if (i + 1 < my_tiles) commit(R1, i + 1);
            if (i + 3 < my_tiles) issue(R1, i + 3);
            __syncthreads();
            if (i + 1 < my_tiles) {
                if (i + 2 < my_tiles) commit(R0, i + 2);
                if (i + 4 < my_tiles) issue(R0, i + 4);
                __syncthreads();
            }
        }
        // bias gradient: 16 stager threads hold partial sums of the same 4 columns; the tile buffers are free now
        float *cs = dws_lds;                                      // [16][128]
        *reinterpret_cast<float4 *>(&cs[rb * 128 + c4]) = dbs;
        __syncthreads();
        if (blockIdx.z == 0 && st < 128 && n0 + st < p.N) {
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) t += cs[u * 128 + st];
            out[(size_t)(n0 + st) * Kout + Kact] = t;
        }
    } else {
        // ---------------- multipliers: wave (wn, wk) owns the 64 x 64 quadrant at (64 wn, 64 wk) over all rows
        const int wn = wave & 1, wk = wave >> 1;
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        __syncthreads();
        for (int i = 0; i < my_tiles; ++i) {
            const float *sD = dws_lds + (i & 1) * 2 * DWW_TILE, *sX = sD + DWW_TILE;
            const float *dBase = &sD[(32 * half) * DWW_LD + wn * 64 + l31];      // lane half h: rows 32 h .. 32 h + 31
            const float *xBase = &sX[(32 * half) * DWW_LD + wk * 64 + l31];
            float a0 = dBase[0], a1v = dBase[32], b0 = xBase[0], b1 = xBase[32];
#pragma unroll 8
            for (int t = 0; t < 32; ++t) {
                float a0n = a0, a1n = a1v, b0n = b0, b1n = b1;
                if (t + 1 < 32) {
                    a0n = dBase[(t + 1) * DWW_LD]; a1n = dBase[(t + 1) * DWW_LD + 32];
                    b0n = xBase[(t + 1) * DWW_LD]; b1n = xBase[(t + 1) * DWW_LD + 32];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b1, acc[1][1], 0, 0, 0);
                a0 = a0n; a1v = a1n; b0 = b0n; b1 = b1n;
            }
            __syncthreads();
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int k = k0 + wk * 64 + b * 32 + l31;
                    if (n < p.N && k < Kact) pn2::store_rows(&out[(size_t)n * Kout + k], acc[a][b][r]);
                }
        __syncthreads();                                          // (the stagers' column sums)
    }
}

// Wide-layer form of mlp_dw_kernel (N >= 128 and K >= 128, all-float4): a workgroup owns a
// 128(n) x 128(k) block of dW, each wave a 64 x 64 quarter of it over ALL rows of every 64-row
// tile (no cross-wave reduction), so a staged element feeds twice as many MFMAs and the dz / act
// tiles are re-staged only N/128 resp. K/128 times.
constexpr int DW2_ROWS = 64, DW2_LD = 132;

template <bool POOLED>
__global__ __launch_bounds__(MLP_THREADS) void mlp_dw128_kernel(DwArgs p)
{
    __shared__ __attribute__((aligned(16))) float sD[DW2_ROWS * DW2_LD];   // dz tile  [64 m][128 n]
    __shared__ __attribute__((aligned(16))) float sX[DW2_ROWS * DW2_LD];   // act tile [64 m][128 k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wn = wave & 1, wk = wave >> 1;
    const int n0 = blockIdx.y * 128, k0 = blockIdx.z * 128;
    const int Kact = p.K1 + p.K2, Kout = Kact + 1;
    const int ntiles = (p.M + DW2_ROWS - 1) / DW2_ROWS;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int c4 = (tid & 31) * 4, rb = tid >> 5;           // staging: 4 columns at c4, rows rb + 8*i
    const int n4 = n0 + c4, k4 = k0 + c4;
    const bool n_ok = n4 < p.N, k_act = k4 < Kact, from1 = k4 < p.K1;
    float4 sc, sh, mu, is, a1, a2, asc, ash;
    sc = sh = mu = is = a1 = a2 = asc = ash = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dbs = sc;
    if (n_ok) {
        sc = *reinterpret_cast<const float4 *>(p.scale + n4);
        sh = *reinterpret_cast<const float4 *>(p.shift + n4);
        mu = *reinterpret_cast<const float4 *>(p.mean + n4);
        is = *reinterpret_cast<const float4 *>(p.invstd + n4);
        a1 = *reinterpret_cast<const float4 *>(p.c1 + n4);
        a2 = *reinterpret_cast<const float4 *>(p.c2 + n4);
    }
    if (k_act && p.ascale) {
        asc = *reinterpret_cast<const float4 *>(p.ascale + k4);
        ash = *reinterpret_cast<const float4 *>(p.ashift + k4);
    }
    float4 gv[8], zv[8], xv[8];
    uchar4 av[8];
    // unconditional loads (clamped addresses, masked when consumed), see mlp_dw_kernel
    const int n4c = n_ok ? n4 : 0;
    const float *xsrc = from1 ? p.x1 : (p.x2 ? p.x2 : p.x1);
    const int xld = from1 ? p.ld1 : (p.x2 ? p.ld2 : p.ld1), xcol = k_act ? (from1 ? k4 : k4 - p.K1) : 0;
    auto issue = [&](int tile) {
        const int row0 = tile * DW2_ROWS;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = min(row0 + rb + 8 * i, p.M - 1);
            if (POOLED) {
                const int cent = row / p.pool_k;
                gv[i] = *reinterpret_cast<const float4 *>(p.g + (size_t)cent * p.ldg + n4c);
                av[i] = *reinterpret_cast<const uchar4 *>(p.argk + (size_t)cent * p.N + n4c);
            } else {
                gv[i] = *reinterpret_cast<const float4 *>(p.g + (size_t)row * p.ldg + n4c);
            }
            zv[i] = *reinterpret_cast<const float4 *>(p.z + (size_t)row * p.ldz + n4c);
            xv[i] = *reinterpret_cast<const float4 *>(xsrc + (size_t)row * xld + xcol);
        }
    };
    if ((int)blockIdx.x < ntiles) issue(blockIdx.x);
    // sub-blocks of this wave that lie inside [N] x [Kact]
    const int na = min(2, max(0, (p.N - (n0 + wn * 64) + 31) / 32));
    const int nb = min(2, max(0, (Kact - (k0 + wk * 64) + 31) / 32));

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * DW2_ROWS;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = rb + 8 * i;
            const int row = row0 + r;
            float4 dv = make_float4(0.f, 0.f, 0.f, 0.f), av4 = dv;
            if (row < p.M) {
                if (n_ok) {
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
                if (k_act) {
                    av4 = xv[i];
                    if (p.ascale) {
                        av4.x = fmaxf(asc.x * av4.x + ash.x, 0.f);
                        av4.y = fmaxf(asc.y * av4.y + ash.y, 0.f);
                        av4.z = fmaxf(asc.z * av4.z + ash.z, 0.f);
                        av4.w = fmaxf(asc.w * av4.w + ash.w, 0.f);
                    }
                }
            }
            *reinterpret_cast<float4 *>(&sD[r * DW2_LD + c4]) = dv;
            *reinterpret_cast<float4 *>(&sX[r * DW2_LD + c4]) = av4;
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        // MFMA: i = n, j = k, reduction over the tile's 64 rows (lane half h takes rows 32h .. 32h+31)
        const float *dBase = &sD[(32 * half) * DW2_LD + wn * 64 + l31];
        const float *xBase = &sX[(32 * half) * DW2_LD + wk * 64 + l31];
        if (na > 0 && nb > 0) {
#pragma unroll 8
            for (int t = 0; t < 32; ++t) {
                const float a0 = dBase[t * DW2_LD], a1v = dBase[t * DW2_LD + 32];
                const float b0 = xBase[t * DW2_LD], b1 = xBase[t * DW2_LD + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                if (nb > 1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                if (na > 1) {
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b0, acc[1][0], 0, 0, 0);
                    if (nb > 1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b1, acc[1][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float *out = p.partial + (size_t)blockIdx.x * p.N * Kout;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int k = k0 + wk * 64 + b * 32 + l31;
                if (n < p.N && k < Kact) pn2::store_rows(&out[(size_t)n * Kout + k], acc[a][b][r]);
            }
    if (blockIdx.z == 0) {
        // bias gradient: 8 threads (rb = 0..7) hold partial column sums for the same 4 columns
        float *cs = sX;                                           // [8][128]
        *reinterpret_cast<float4 *>(&cs[rb * 128 + c4]) = dbs;
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += cs[i * 128 + tid];
            out[(size_t)(n0 + tid) * Kout + Kact] = t;
        }
    }
}

// dW[n][k] = sum_p partial[p][n][k] (k < Kst <= K), db[n] = sum_p partial[p][n][K]; fixed order.  A 1024-thread workgroup
// is SL = 1 << sl_shift slices x 1024/SL lanes; a lane owns four consecutive elements of the flattened [N][K+1] slab (one:
// `vec` false, slabs whose size or address is not a multiple of four floats), slice s sums partials s, s + SL, ... with up to
// eight loads in flight, the slices are combined in order through LDS.  SL grows with P (dw_reduce_shape): few partials
// (the deep levels' 8-16 slabs of up to 200 000 elements) need lanes, many partials (512 slabs of a 32 x 13 layer) need
// slices -- the first form of this routine (32 elements x 32 slices whatever P) spent 13 us on 8 slabs.
// Kst = columns of dw that exist ([N][Kst]): a first layer whose input rows carry zero pad columns has no gradient
// entries for them.
struct DwReduceShape { int sl_shift, vec, blocks; };
inline DwReduceShape dw_reduce_shape(const float *partial, int P, int N, int K)
{
    DwReduceShape r;
    const long long total = (long long)N * (K + 1);
    r.vec = (total % 4 == 0) && ((reinterpret_cast<uintptr_t>(partial) & 15) == 0);
    r.sl_shift = 0;
    while (r.sl_shift < 5 && (P >> r.sl_shift) > 8) ++r.sl_shift;
    const long long per_wg = (long long)(1024 >> r.sl_shift) * (r.vec ? 4 : 1);
    r.blocks = (int)((total + per_wg - 1) / per_wg);
    return r;
}

__device__ __forceinline__ void dw_reduce_block(int block, const float *__restrict__ partial, int P, int N, int K, int Kst,
                                                int sl_shift, int vec, float *__restrict__ dw, float *__restrict__ db, float4 *sbuf)
{
    const int Kout = K + 1;
    const int total = N * Kout;
    const int SL = 1 << sl_shift, lanes = 1024 >> sl_shift;
    const int q = threadIdx.x & (lanes - 1), sl = threadIdx.x >> (10 - sl_shift);
    const int e0 = (block * lanes + q) * (vec ? 4 : 1);
    const size_t stride = (size_t)total;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e0 < total) {
        const float *src = partial + e0;
        for (int p0 = sl; p0 < P; p0 += 8 * SL) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t o = (size_t)min(p0 + u * SL, P - 1) * stride;
                if (vec) v[u] = *reinterpret_cast<const float4 *>(src + o);
                else v[u] = make_float4(src[o], 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + u * SL < P) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    if (SL > 1) {
        sbuf[sl * lanes + q] = acc;
        __syncthreads();
        if (sl != 0) return;
        for (int i = 1; i < SL; ++i) {
            const float4 t = sbuf[i * lanes + q];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
    }
    if (e0 >= total) return;
    if (Kst < 0) Kst = K;
    const float out[4] = {acc.x, acc.y, acc.z, acc.w};
    int n = e0 / Kout, k = e0 - n * Kout;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (u == 0 || vec) {
            if (k < K) { if (k < Kst) dw[(size_t)n * Kst + k] = out[u]; }
            else if (db) db[n] = out[u];
            if (++k == Kout) { k = 0; ++n; }
        }
    }
}

__global__ __launch_bounds__(1024) void dw_reduce_kernel(const float *__restrict__ partial, int P, int N, int K, int sl_shift, int vec,
                                                        float *__restrict__ dw, float *__restrict__ db)
{
    __shared__ __attribute__((aligned(16))) float4 sbuf[1024];
    dw_reduce_block(blockIdx.x, partial, P, N, K, K, sl_shift, vec, dw, db, sbuf);
}

// The slab sums of SEVERAL layers in one launch (the bottom layers of all stacks, whose sums nothing but the optimizer
// waits for: pn2_mlp_dw_reduce_many).  Jobs by value: a captured launch keeps them.
constexpr int DW_MANY_MAX = 16;
struct DwReduceMany {
    int n;
    int first[DW_MANY_MAX + 1];                 // first workgroup of job j; first[n] = grid size
    const float *partial[DW_MANY_MAX];
    float *dw[DW_MANY_MAX], *db[DW_MANY_MAX];
    int P[DW_MANY_MAX], N[DW_MANY_MAX], K[DW_MANY_MAX], Kst[DW_MANY_MAX];
    unsigned char sl_shift[DW_MANY_MAX], vec[DW_MANY_MAX];
};

__global__ __launch_bounds__(1024) void dw_reduce_many_kernel(DwReduceMany m)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ __attribute__((aligned(16))) float4 sbuf[1024];
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.first[j + 1]) ++j;
    dw_reduce_block((int)blockIdx.x - m.first[j], m.partial[j], m.P[j], m.N[j], m.K[j], m.Kst[j], m.sl_shift[j], m.vec[j], m.dw[j],
                    m.db[j], sbuf);
}

// Column partial sums of gh and gh*xh over rows for the TOP layer of a stack (the inner layers
// get theirs from the GEMM's backward epilogue):  gh = (scale*z+shift > 0) ? g : 0,
// xh = (z-mean)*invstd.  Explicit g [M][C], or pooled: only the arg-max row of every centroid
// carries gradient, so the sum runs over centroids with a gather of z.
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *__restrict__ g, int ldg,
                                                            const float *__restrict__ z, int ldz, long long rows,
                                                            int C, const unsigned char *__restrict__ argk, int pool_k,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ shift,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd,
                                                            float *__restrict__ partial)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ float sS[8][32], sQ[8][32];
    const int cl = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.y * 32 + cl;
    float s = 0.f, q = 0.f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        // four rows per pass, every load of a stage issued before the first use: g / argk, then the gathered z -- two
        // memory round trips per four rows instead of three per row (the launch is nothing but these round trips)
        const long long stride = (long long)gridDim.x * 8;
        for (long long r0 = (long long)blockIdx.x * 8 + ry; r0 < rows; r0 += 4 * stride) {
            float gv[4], zv[4];
            long long zr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long r = r0 + u * stride < rows ? r0 + u * stride : rows - 1;
                gv[u] = g[(size_t)r * ldg + c];
                zr[u] = argk ? r * pool_k + argk[(size_t)r * C + c] : r;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) zv[u] = z[(size_t)zr[u] * ldz + c];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r0 + u * stride < rows) {
                    const float gh = (sc * zv[u] + sh) > 0.f ? gv[u] : 0.f;
                    s += gh;
                    q += gh * ((zv[u] - mu) * is);
                }
            }
        }
    }
    sS[ry][cl] = s;
    sQ[ry][cl] = q;
    __syncthreads();
    if (ry == 0 && c < C) {
        for (int i = 1; i < 8; ++i) { s += sS[i][cl]; q += sQ[i][cl]; }
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = s;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = q;
    }
}

// Dense form (no pooling, C % 4 == 0): a thread owns 4 columns, 256 threads cover 256 / (C/4 or 32) rows per pass,
// 4 passes (8 float4 loads) are issued before the first add -- the scalar kernel above pays one memory round trip
// per row and thread.  blockIdx.y = 128-column slab.
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec4_kernel(const float *__restrict__ g, int ldg, const float *__restrict__ z,
                                                                 int ldz, long long rows, int C, const float *__restrict__ scale,
                                                                 const float *__restrict__ shift, const float *__restrict__ mean,
                                                                 const float *__restrict__ invstd, float *__restrict__ partial)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ float4 sS[8][32], sQ[8][32];
    const int cl = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c4 = blockIdx.y * 128 + cl * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (c4 < C) {
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c4), sh = *reinterpret_cast<const float4 *>(shift + c4);
        const float4 mu = *reinterpret_cast<const float4 *>(mean + c4), is = *reinterpret_cast<const float4 *>(invstd + c4);
        const long long stride = (long long)gridDim.x * 8;
        for (long long r0 = (long long)blockIdx.x * 8 + ry; r0 < rows; r0 += 4 * stride) {
            float4 gv[4], zv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long r = r0 + u * stride < rows ? r0 + u * stride : rows - 1;
                gv[u] = *reinterpret_cast<const float4 *>(g + (size_t)r * ldg + c4);
                zv[u] = *reinterpret_cast<const float4 *>(z + (size_t)r * ldz + c4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r0 + u * stride < rows) {
#define PN2_ACC(f) { const float gh = (sc.f * zv[u].f + sh.f) > 0.f ? gv[u].f : 0.f; s.f += gh; q.f += gh * ((zv[u].f - mu.f) * is.f); }
                    PN2_ACC(x) PN2_ACC(y) PN2_ACC(z) PN2_ACC(w)
#undef PN2_ACC
                }
            }
        }
    }
    sS[ry][cl] = s;
    sQ[ry][cl] = q;
    __syncthreads();
    if (ry == 0 && c4 < C) {
        for (int i = 1; i < 8; ++i) {
            s.x += sS[i][cl].x; s.y += sS[i][cl].y; s.z += sS[i][cl].z; s.w += sS[i][cl].w;
            q.x += sQ[i][cl].x; q.y += sQ[i][cl].y; q.z += sQ[i][cl].z; q.w += sQ[i][cl].w;
        }
        *reinterpret_cast<float4 *>(partial + ((size_t)blockIdx.x * 2 + 0) * C + c4) = s;
        *reinterpret_cast<float4 *>(partial + ((size_t)blockIdx.x * 2 + 1) * C + c4) = q;
    }
}

// partial[P][2][C] -> dbeta = sum gh, dgamma = sum gh*xh, c1 = dbeta/count, c2 = dgamma/count
__device__ __forceinline__ void bn_bwd_finalize_block(int block, const float *__restrict__ partial, int P, int C, double count,
                                                      float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                      float *__restrict__ c1, float *__restrict__ c2, double (*sS)[33],
                                                      double (*sQ)[33])
{
    const int cl = threadIdx.x & 31, py = threadIdx.x >> 5;
    const int c = block * 32 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
        strided_sum_pair_any(partial + c, partial + C + c, (size_t)2 * C, py, P, s0, s1, q0, q1);
        s = (double)s0 + (double)s1;
        q = (double)q0 + (double)q1;
    }
    sS[py][cl] = s;
    sQ[py][cl] = q;
    __syncthreads();
    if (py != 0 || c >= C) return;
    for (int i = 1; i < 32; ++i) { s += sS[i][cl]; q += sQ[i][cl]; }
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
    c1[c] = (float)(s / count);
    c2[c] = (float)(q / count);
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int P, int C,
                                                              double count, float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta, float *__restrict__ c1,
                                                              float *__restrict__ c2)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ double sS[32][33], sQ[32][33];
    bn_bwd_finalize_block(blockIdx.x, partial, P, C, count, dgamma, dbeta, c1, c2, sS, sQ);
}

// Both reductions that follow a layer's backward kernel in ONE launch (every launch costs a few
// microseconds of fixed time): blocks [0, nred) sum the dW slabs, the rest finalize the BatchNorm
// statistics of the layer below.
__global__ __launch_bounds__(1024) void bwd_post_kernel(const float *__restrict__ dw_partial, int P, int N, int K, int sl_shift,
                                                       int vec, float *__restrict__ dw, float *__restrict__ db, int nred,
                                                       const float *__restrict__ stat_partial, int Ps, int C, double count,
                                                       float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                       float *__restrict__ c1, float *__restrict__ c2)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ __attribute__((aligned(16))) double sbuf[2 * 32 * 33];
    static_assert(sizeof(sbuf) >= 1024 * sizeof(float4), "slice combine buffer");
    if ((int)blockIdx.x < nred)
        dw_reduce_block(blockIdx.x, dw_partial, P, N, K, K, sl_shift, vec, dw, db, reinterpret_cast<float4 *>(sbuf));
    else
        bn_bwd_finalize_block(blockIdx.x - nred, stat_partial, Ps, C, count, dgamma, dbeta, c1, c2,
                              reinterpret_cast<double (*)[33]>(sbuf), reinterpret_cast<double (*)[33]>(sbuf + 32 * 33));
}

inline unsigned grid_for(long long total, int threads)
{
    long long blocks = (total + threads - 1) / threads;
    const long long cap = 256LL * 16;
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

template <int BN>
int launch_gemm(const GemmArgs &a, int pro, bool vec4, bool pipe, int gx, hipStream_t stream)
{
    dim3 grid((unsigned)gx, (unsigned)((a.N + BN - 1) / BN));
    if (pipe) {
        int pool_shift = -1;
        if (a.argk && (a.pool_k & (a.pool_k - 1)) == 0) { pool_shift = 0; while ((1 << pool_shift) < a.pool_k) ++pool_shift; }
        constexpr int NWmax = BN == 128 ? 2 : 1;
        // bit per prologue kind; the BN-backward prologue needs too many registers for the 8-wave form
        const bool wide = NWmax == 2 && ((pn2::tune_get("mlp_nw2", 3) >> pro) & 1);
#define PN2_PIPE(P, W) do { if (wide) hipLaunchKernelGGL((mlp_gemm_pipe_kernel<BN, P, W, NWmax>), grid, dim3(MLP_THREADS * NWmax), 0, stream, a, pool_shift); \
                            else hipLaunchKernelGGL((mlp_gemm_pipe_kernel<BN, P, W, 1>), grid, dim3(MLP_THREADS), 0, stream, a, pool_shift); } while (0)
        if (pro == PRO_NONE) { if (a.wt) PN2_PIPE(PRO_NONE, true); else PN2_PIPE(PRO_NONE, false); }
        else if (pro == PRO_BN_RELU) { if (a.wt) PN2_PIPE(PRO_BN_RELU, true); else PN2_PIPE(PRO_BN_RELU, false); }
        else { if (a.wt) PN2_PIPE(PRO_BN_BWD, true); else PN2_PIPE(PRO_BN_BWD, false); }
#undef PN2_PIPE
        return PN2_LAUNCH_RC();
    }
#define PN2_GEMM(P, V) hipLaunchKernelGGL((mlp_gemm_kernel<BN, P, V>), grid, dim3(MLP_THREADS), 0, stream, a)
    if (pro == PRO_NONE) { if (vec4) PN2_GEMM(PRO_NONE, true); else PN2_GEMM(PRO_NONE, false); }
    else if (pro == PRO_BN_RELU) { if (vec4) PN2_GEMM(PRO_BN_RELU, true); else PN2_GEMM(PRO_BN_RELU, false); }
    else { if (vec4) PN2_GEMM(PRO_BN_BWD, true); else PN2_GEMM(PRO_BN_BWD, false); }
#undef PN2_GEMM
    return PN2_LAUNCH_RC();
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// 32-row tiles are used when 128-row tiles would give fewer workgroups than CUs
static bool pn2_mlp_gemm_uses_rows32(int M, int N)
{
    const long long wgs128 = (long long)((M + MLP_BM - 1) / MLP_BM) * ((N + 127) / 128);
    return wgs128 < pn2::tune_get("mlp_rows32_thresh", 512);
}

PN2_EXPORT int pn2_mlp_gemm_max_partials(int M)
{
    // upper bound over both tilings (the statistics workspace is sized by the caller from this)
    const int ntiles = (M + 31) / 32;
    return ntiles < 512 ? (ntiles < 1 ? 1 : ntiles) : 512;
}

static int mlp_gemm_impl(const float *x1, int ld1, int K1, const float *x2, int ld2, int K2, int prologue,
                         const float *scale, const float *shift, const float *mean, const float *invstd,
                         const float *c1, const float *c2, const unsigned char *argk, int pool_k, const float *w, int ldw,
                         int w_is_kn,
                         const float *bias, float *out, int ldo, float *out2, int ldo2, int nsplit, int M, int N,
                         float *stat_partial, const float *mask_z, int ldm, const float *mscale, const float *mshift,
                         const float *mmean, const float *minvstd, float *pool_max, float *pool_min,
                         unsigned char *pool_amax, unsigned char *pool_amin, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(x1);
    PN2_REQUIRE_PTR(w);
    if (!out && !pool_max) return PN2_ERR_NULL;                 // the pooled epilogue may run without storing z
    if (M < 0 || N <= 0 || K1 <= 0 || K2 < 0 || ld1 < K1) return PN2_ERR_SHAPE;
    if (out2 ? (nsplit <= 0 || nsplit >= N || ldo < nsplit || ldo2 < N - nsplit) : (ldo < N)) return PN2_ERR_SHAPE;
    if (prologue < PRO_NONE || prologue > PRO_BN_BWD) return PN2_ERR_SHAPE;
    if (prologue == PRO_BN_BWD) {
        if (!x2 || !scale || !shift || !mean || !invstd || !c1 || !c2) return PN2_ERR_NULL;
        if (K2 != K1 || ld2 < K1 || (argk && pool_k <= 0)) return PN2_ERR_SHAPE;
    } else {
        if (K2 > 0 && (!x2 || ld2 < K2)) return PN2_ERR_NULL;
        if (prologue == PRO_BN_RELU && (!scale || !shift)) return PN2_ERR_NULL;
    }
    if (mask_z && (!mscale || !mshift || !mmean || !minvstd || ldm < N)) return PN2_ERR_NULL;
    if (mask_z && prologue != PRO_BN_BWD) return PN2_ERR_SHAPE;      // the masked epilogue belongs to the dX GEMM
    if (M == 0) return PN2_OK;
    GemmArgs a;
    a.x1 = x1; a.x2 = x2; a.ld1 = ld1; a.ld2 = ld2; a.K1 = K1; a.K2 = K2;
    a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.c1 = c1; a.c2 = c2;
    a.argk = prologue == PRO_BN_BWD ? argk : nullptr; a.pool_k = pool_k;
    a.w = w; a.ldw = ldw; a.wt = w_is_kn; a.bias = bias; a.out = out; a.ldo = ldo; a.M = M; a.N = N;
    a.out2 = out2; a.ldo2 = ldo2; a.nsplit = nsplit;
    a.K = prologue == PRO_BN_BWD ? K1 : K1 + K2;
    a.stat_partial = stat_partial;
    a.mask_z = mask_z; a.mscale = mscale; a.mshift = mshift; a.mmean = mmean; a.minvstd = minvstd; a.ldm = ldm;
    a.pool_max = pool_max; a.pool_min = pool_min; a.pool_amax = pool_amax; a.pool_amin = pool_amin;
    // float4 staging needs 16-B aligned rows and a concat boundary on a multiple of 4
    bool vec4 = (ld1 % 4 == 0) && aligned16(x1) && (K1 % 4 == 0) && (a.K % 4 == 0);
    if (x2) vec4 = vec4 && (ld2 % 4 == 0) && aligned16(x2);
    if (prologue != PRO_NONE) vec4 = vec4 && aligned16(scale) && aligned16(shift);
    if (prologue == PRO_BN_BWD) vec4 = vec4 && aligned16(mean) && aligned16(invstd) && aligned16(c1) && aligned16(c2);
    const int gx = pn2_mlp_gemm_max_partials(M);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    {
        bool pipe_ok = vec4 && (ldw % 4 == 0) && aligned16(w) && (w_is_kn ? (N % 4 == 0) : true);
        if (prologue == PRO_BN_BWD && argk) pipe_ok = pipe_ok && ((reinterpret_cast<uintptr_t>(argk) & 3) == 0);
        // very few rows: one 32 x 32 block per workgroup, K split over its eight waves (statistics: one partial row per
        // 32-row tile = pn2_mlp_gemm_max_partials(M) of them)
        // (the pooled epilogue is the shared one: a 32-row block is a group, so pooled and plain layers keep the same sums)
        if (pipe_ok && !(pool_max && M % 32) && !a.argk && M <= pn2::tune_get("mlp_tiny_rows", 1024) && (M + 31) / 32 == gx) {
            const int ns = (a.K + 63) / 64;                        // 8-wide steps per wave
            dim3 grid((unsigned)gx, (unsigned)((N + 31) / 32));
            hipStream_t st = stream;
#define PN2_TINY(P, W, NS) do { hipLaunchKernelGGL((mlp_gemm_tiny_kernel<P, W, NS>), grid, dim3(512), 0, st, a); return PN2_LAUNCH_RC(); } while (0)
            if (prologue == PRO_NONE && !a.wt) {
                if (ns <= 4) PN2_TINY(PRO_NONE, false, 4);
                if (ns <= 8) PN2_TINY(PRO_NONE, false, 8);
                if (ns <= 12) PN2_TINY(PRO_NONE, false, 12);
                if (ns <= 16) PN2_TINY(PRO_NONE, false, 16);
            } else if (prologue == PRO_BN_RELU && !a.wt) {
                if (ns <= 4) PN2_TINY(PRO_BN_RELU, false, 4);
                if (ns <= 8) PN2_TINY(PRO_BN_RELU, false, 8);
            } else if (prologue == PRO_BN_BWD && a.wt && (N % 4 == 0)) {
                if (ns <= 4) PN2_TINY(PRO_BN_BWD, true, 4);
            }
#undef PN2_TINY
        }
        // few rows: 32-row tiles so that (row tiles) x (128-column blocks) still covers the 256 CUs
        if (pipe_ok && pn2_mlp_gemm_uses_rows32(M, N) && pn2::tune_get("mlp_rows32", 1)) {
            int pool_shift = -1;
            if (a.argk && (a.pool_k & (a.pool_k - 1)) == 0) { pool_shift = 0; while ((1 << pool_shift) < a.pool_k) ++pool_shift; }
            dim3 grid((unsigned)gx, (unsigned)((N + 127) / 128));
            // 64-row tiles (statistics: two partial rows per workgroup, so half as many workgroups) when that
            // still covers the CUs
            const long long wgs64 = (long long)((M + 63) / 64) * ((N + 127) / 128);
            const bool rows64 = (gx % 2 == 0) && wgs64 >= pn2::tune_get("mlp_rows64_min_wgs", 256) && pn2::tune_get("mlp_rows64", 1);
            if (rows64) grid.x = (unsigned)(gx / 2);
            // 64-wide K steps for the long reductions of the 64-row form (dynamic LDS: 100 KB)
            const bool bk64 = rows64 && a.K >= 128 && ((pn2::tune_get("mlp_rows_bk64", 4) >> prologue) & 1);
            // 32-row tiles (M <= 4096: at most one workgroup per CU, every step pays the full load latency): 64-wide
            // steps as well when the reduction is long enough
            const bool bk64s = !rows64 && a.K >= 256 && ((pn2::tune_get("mlp_rows32_bk64", 7) >> prologue) & 1);
            constexpr size_t LDS32_64 = 2 * (64 * 36 + 128 * 36) * sizeof(float), LDS32_32 = 2 * (32 * 36 + 128 * 36) * sizeof(float);
            constexpr size_t LDS64_N = 2 * (64 * 68 + 128 * 68) * sizeof(float), LDS64_T = 2 * (64 * 68 + 64 * 132) * sizeof(float);
            constexpr size_t LDS64S_N = 2 * (32 * 68 + 128 * 68) * sizeof(float), LDS64S_T = 2 * (32 * 68 + 64 * 132) * sizeof(float);
#define PN2_R32_ATTR(KERNEL) do { \
            static pn2::PerDevice lds_memo; \
            if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(KERNEL), 160 * 1024, lds_memo)) return e; \
            } while (0)
#define PN2_R32(P, W) do { \
        if (bk64) { \
            PN2_R32_ATTR((mlp_gemm_rows32_kernel<P, W, 64, 64>)); \
            hipLaunchKernelGGL((mlp_gemm_rows32_kernel<P, W, 64, 64>), grid, dim3(512), (W) ? LDS64_T : LDS64_N, stream, a, pool_shift); \
        } else if (rows64) hipLaunchKernelGGL((mlp_gemm_rows32_kernel<P, W, 64, 32>), grid, dim3(512), LDS32_64, stream, a, pool_shift); \
        else if (bk64s) { \
            PN2_R32_ATTR((mlp_gemm_rows32_kernel<P, W, 32, 64>)); \
            hipLaunchKernelGGL((mlp_gemm_rows32_kernel<P, W, 32, 64>), grid, dim3(256), (W) ? LDS64S_T : LDS64S_N, stream, a, pool_shift); \
        } else hipLaunchKernelGGL((mlp_gemm_rows32_kernel<P, W, 32, 32>), grid, dim3(256), LDS32_32, stream, a, pool_shift); } while (0)
            if (prologue == PRO_NONE) { if (a.wt) PN2_R32(PRO_NONE, true); else PN2_R32(PRO_NONE, false); }
            else if (prologue == PRO_BN_RELU) { if (a.wt) PN2_R32(PRO_BN_RELU, true); else PN2_R32(PRO_BN_RELU, false); }
            else { if (a.wt) PN2_R32(PRO_BN_BWD, true); else PN2_R32(PRO_BN_BWD, false); }
#undef PN2_R32
#undef PN2_R32_ATTR
            return PN2_LAUNCH_RC();
        }
    }
    // the pipelined kernel also stages the weight tile with float4 loads
    bool pipe = vec4 && (ldw % 4 == 0) && aligned16(w) && (w_is_kn ? (N % 4 == 0) : true) && pn2::tune_get("mlp_pipe", 1);
    if (prologue == PRO_BN_BWD && argk) pipe = pipe && ((reinterpret_cast<uintptr_t>(argk) & 3) == 0);
    if (pool_max && !pipe) return PN2_ERR_UNSUPPORTED;          // only the pipelined epilogues pool
    if (N <= 32) return launch_gemm<32>(a, prologue, vec4, pipe, gx, stream);
    if (N <= 64) return launch_gemm<64>(a, prologue, vec4, pipe, gx, stream);
    return launch_gemm<128>(a, prologue, vec4, pipe, gx, stream);
}

PN2_EXPORT int pn2_mlp_gemm(const float *x1, int ld1, int K1, const float *x2, int ld2, int K2, int prologue,
                            const float *scale, const float *shift, const float *mean, const float *invstd,
                            const float *c1, const float *c2, const unsigned char *argk, int pool_k, const float *w, int ldw,
                            int w_is_kn,
                            const float *bias, float *out, int ldo, float *out2, int ldo2, int nsplit, int M, int N,
                            float *stat_partial, const float *mask_z, int ldm, const float *mscale, const float *mshift,
                            const float *mmean, const float *minvstd, pn2_stream_t stream_)
{
    return mlp_gemm_impl(x1, ld1, K1, x2, ld2, K2, prologue, scale, shift, mean, invstd, c1, c2, argk, pool_k, w, ldw, w_is_kn,
                         bias, out, ldo, out2, ldo2, nsplit, M, N, stat_partial, mask_z, ldm, mscale, mshift, mmean, minvstd,
                         nullptr, nullptr, nullptr, nullptr, stream_);
}

PN2_EXPORT int pn2_mlp_gemm_pool32(const float *x1, int ld1, int K1, const float *x2, int ld2, int K2, int prologue,
                                   const float *scale, const float *shift, const float *w, int ldw, const float *bias,
                                   float *out, int ldo, int M, int N, float *stat_partial, float *pool_max, float *pool_min,
                                   unsigned char *pool_amax, unsigned char *pool_amin, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(pool_max);
    PN2_REQUIRE_PTR(pool_min);
    PN2_REQUIRE_PTR(pool_amax);
    PN2_REQUIRE_PTR(pool_amin);
    if (prologue != PRO_NONE && prologue != PRO_BN_RELU) return PN2_ERR_SHAPE;
    if (M % 32 != 0) return PN2_ERR_SHAPE;                      // groups of 32 rows = whole accumulator blocks
    return mlp_gemm_impl(x1, ld1, K1, x2, ld2, K2, prologue, scale, shift, nullptr, nullptr, nullptr, nullptr, nullptr, 0, w,
                         ldw, 0, bias, out, ldo, nullptr, 0, 0, M, N, stat_partial, nullptr, 0, nullptr, nullptr, nullptr,
                         nullptr, pool_max, pool_min, pool_amax, pool_amin, stream_);
}

PN2_EXPORT int pn2_bn_finalize(const float *partial, int P, int C, double count, const float *gamma,
                               const float *beta, float eps, float momentum, const float *momentum_dev, float *running_mean,
                               float *running_var, float *scale, float *shift, float *mean_out, float *invstd_out,
                               long long *num_batches_tracked, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(partial);
    PN2_REQUIRE_PTR(scale);
    PN2_REQUIRE_PTR(shift);
    if (P <= 0 || C <= 0 || count <= 0) return PN2_ERR_SHAPE;
    BnFinArgs a;
    a.partial = partial; a.P = P; a.C = C; a.count = count; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.momentum = momentum; a.momentum_dev = momentum_dev; a.running_mean = running_mean; a.running_var = running_var;
    a.scale = scale; a.shift = shift; a.mean_out = mean_out; a.invstd_out = invstd_out;
    a.num_batches_tracked = num_batches_tracked;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, static_cast<hipStream_t>(stream_), a);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_bn_finalize_out(const float *partial, int P, int C, double count, const float *gamma,
                                   const float *beta, float eps, float momentum, const float *momentum_dev,
                                   float *running_mean, float *running_var, float *scale, float *shift, float *mean_out,
                                   float *invstd_out, long long *num_batches_tracked, const float *z, int ldz,
                                   const float *pool_max, const float *pool_min, const unsigned char *pool_amax,
                                   const unsigned char *pool_amin, long long rows_out, float *y, unsigned char *argk,
                                   pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(scale);
    PN2_REQUIRE_PTR(shift);
    PN2_REQUIRE_PTR(y);
    if (C <= 0 || rows_out < 0 || (partial && (P <= 0 || count <= 0))) return PN2_ERR_SHAPE;
    if (pool_max) {
        if (!pool_min || !pool_amax || !pool_amin || !argk) return PN2_ERR_NULL;
        if (!aligned16(pool_max) || !aligned16(pool_min) || (reinterpret_cast<uintptr_t>(pool_amax) & 3) ||
            (reinterpret_cast<uintptr_t>(pool_amin) & 3) || (reinterpret_cast<uintptr_t>(argk) & 3))
            return PN2_ERR_UNSUPPORTED;
    } else {
        PN2_REQUIRE_PTR(z);
        if (ldz < C) return PN2_ERR_SHAPE;
        if (ldz % 4 != 0 || !aligned16(z)) return PN2_ERR_UNSUPPORTED;
    }
    if (C % 4 != 0 || !aligned16(y)) return PN2_ERR_UNSUPPORTED;
    BnFinArgs a;
    a.partial = partial; a.P = P; a.C = C; a.count = count; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.momentum = momentum; a.momentum_dev = momentum_dev; a.running_mean = running_mean; a.running_var = running_var;
    a.scale = scale; a.shift = shift; a.mean_out = mean_out; a.invstd_out = invstd_out;
    a.num_batches_tracked = num_batches_tracked;
    BnOutArgs o;
    o.z = z; o.ldz = ldz; o.pool_max = pool_max; o.pool_min = pool_min; o.pool_amax = pool_amax; o.pool_amin = pool_amin;
    o.rows_out = rows_out; o.y = y; o.argk = argk;
    // row slices: 128 rows per pass of a workgroup; at most 64 slices (every slice repeats the finalize)
    long long slices = (rows_out + 127) / 128;
    const long long cap = pn2::tune_get("bn_out_slices", 64);
    if (slices > cap) slices = cap;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(bn_finalize_out_kernel, dim3((C + 31) / 32, (unsigned)slices), dim3(1024), 0,
                       static_cast<hipStream_t>(stream_), a, o);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_bn_eval_coeff(int C, const float *gamma, const float *beta, const float *running_mean,
                                 const float *running_var, float eps, float *scale, float *shift,
                                 pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(running_mean);
    PN2_REQUIRE_PTR(running_var);
    PN2_REQUIRE_PTR(scale);
    PN2_REQUIRE_PTR(shift);
    if (C <= 0) return PN2_ERR_SHAPE;
    hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), C,
                       gamma, beta, running_mean, running_var, eps, scale, shift);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_bn_relu_out(const float *z, long long rows_out, int C, int pool_k, const float *scale,
                               const float *shift, float *y, unsigned char *argk, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(z);
    PN2_REQUIRE_PTR(scale);
    PN2_REQUIRE_PTR(shift);
    PN2_REQUIRE_PTR(y);
    if (rows_out < 0 || C <= 0 || pool_k < 0 || pool_k > 255) return PN2_ERR_SHAPE;
    if (C % 4 != 0 || !aligned16(z) || !aligned16(y) || !aligned16(scale) || !aligned16(shift)) return PN2_ERR_UNSUPPORTED;
    if (rows_out == 0) return PN2_OK;
    const long long total = rows_out * (C / 4);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (pool_k > 0)
        hipLaunchKernelGGL(bn_relu_out_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, stream, z, rows_out, C,
                           pool_k, scale, shift, y, argk);
    else
        hipLaunchKernelGGL(bn_relu_out_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, stream, z, rows_out, C, 1,
                           scale, shift, y, argk);
    return PN2_LAUNCH_RC();
}

// 128 x 128 output blocks pay when little of the last k block is padding (sa4's 260 inputs would pad to 384)
static bool dw_split128_applies(int N, int K)
{
    if (N < 128 || K < 128 || !pn2::tune_get("mlp_dw_split", 1) || !pn2::tune_get("mlp_dw_split128", 1)) return false;
    const int kp = (K + 127) / 128 * 128, np = (N + 127) / 128 * 128;
    return (long long)kp * np * 4 <= (long long)K * N * 5;             // at most 25 % more products than the layer has
}

PN2_EXPORT int pn2_mlp_dw_partials(int M, int N, int K)
{
    // ~512 workgroups in flight (2 per CU): the M axis is split so that (M slabs) x (64x64 output blocks) ~ 512
    if (dw_split128_applies(N, K)) {
        // one 12-wave workgroup per CU: (M slabs) x (128 x 128 output blocks) ~ 256, 64-row tiles
        const int nt = (M + DWW_ROWS - 1) / DWW_ROWS;
        int q = pn2::tune_get("dw_split_wgs", 256) / (((N + 127) / 128) * ((K + 127) / 128));
        q = q < 8 ? 8 : q;
        return q > nt ? nt : q;
    }
    const bool wide = N >= 128 && K >= 128 && pn2::tune_get("mlp_dw128", 0);
    const int ntiles = wide ? (M + DW2_ROWS - 1) / DW2_ROWS : (M + MLP_BM - 1) / MLP_BM;
    const int blocks = wide ? ((N + 127) / 128) * ((K + 127) / 128) : ((N + DW_BN - 1) / DW_BN) * ((K + DW_BK - 1) / DW_BK);
    // the split-role form holds one 8-wave workgroup per CU
    const int wgs = (!wide && pn2::tune_get("mlp_dw_split", 1)) ? pn2::tune_get("dw_split_wgs", 256) : pn2::tune_get("dw_wgs", 512);
    int p = wgs / (blocks < 1 ? 1 : blocks);                                // one-role form: 256/384/640/1024 measured slower
    // (a cap tying the partial-slab traffic to the layer's input traffic was measured: no gain)
    const int div = pn2::tune_get("dw_p_div", 0);
    if (div > 0) {
        const long long cap = (long long)M * (N + K) / ((long long)div * N * (K + 1));
        if (p > cap) p = (int)cap;
    }
    if (p < 8) p = 8;
    if (p > ntiles) p = ntiles;
    return p < 1 ? 1 : p;
}

int pn2::launch_dw_reduce(const float *partial, int P, int N, int K, float *dw, float *db, hipStream_t stream)
{
    const DwReduceShape r = dw_reduce_shape(partial, P, N, K);
    hipLaunchKernelGGL(dw_reduce_kernel, dim3(r.blocks), dim3(1024), 0, stream, partial, P, N, K, r.sl_shift, r.vec, dw, db);
    return PN2_LAUNCH_RC();
}

int pn2::launch_bwd_post(const float *dw_partial, int P, int N, int K, float *dw, float *db, const float *stat_partial, int Ps,
                         int C, double count, float *dgamma, float *dbeta, float *c1, float *c2, hipStream_t stream)
{
    const DwReduceShape r = dw_reduce_shape(dw_partial, P, N, K);
    const int nred = r.blocks, nfin = (C + 31) / 32;
    hipLaunchKernelGGL(bwd_post_kernel, dim3(nred + nfin), dim3(1024), 0, stream, dw_partial, P, N, K, r.sl_shift, r.vec, dw, db, nred,
                       stat_partial, Ps, C, count, dgamma, dbeta, c1, c2);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_mlp_dw_reduce_many(int n, const float *const *partial, const int *P, const int *N, const int *K,
                                      const int *Kstore, float *const *dw, float *const *db, pn2_stream_t stream)
{
    PN2_REQUIRE_PTR(partial); PN2_REQUIRE_PTR(P); PN2_REQUIRE_PTR(N); PN2_REQUIRE_PTR(K); PN2_REQUIRE_PTR(Kstore);
    PN2_REQUIRE_PTR(dw); PN2_REQUIRE_PTR(db);
    if (n < 0) return PN2_ERR_SHAPE;
    for (int j0 = 0; j0 < n; j0 += DW_MANY_MAX) {
        DwReduceMany m;
        m.n = n - j0 < DW_MANY_MAX ? n - j0 : DW_MANY_MAX;
        long long blocks = 0;
        for (int j = 0; j < DW_MANY_MAX; ++j) {
            const int i = j0 + (j < m.n ? j : m.n - 1);
            if (!partial[i] || !dw[i]) return PN2_ERR_NULL;
            if (P[i] <= 0 || N[i] <= 0 || K[i] <= 0 || Kstore[i] <= 0 || Kstore[i] > K[i]) return PN2_ERR_SHAPE;
            m.partial[j] = partial[i]; m.dw[j] = dw[i]; m.db[j] = db[i];
            m.P[j] = P[i]; m.N[j] = N[i]; m.K[j] = K[i]; m.Kst[j] = Kstore[i];
            const DwReduceShape r = dw_reduce_shape(partial[i], P[i], N[i], K[i]);
            m.sl_shift[j] = (unsigned char)r.sl_shift; m.vec[j] = (unsigned char)r.vec;
            m.first[j] = (int)blocks;
            if (j < m.n) blocks += r.blocks;
            if (blocks > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
        }
        m.first[DW_MANY_MAX] = (int)blocks;
        for (int j = m.n; j < DW_MANY_MAX; ++j) m.first[j] = (int)blocks;
        hipLaunchKernelGGL(dw_reduce_many_kernel, dim3((unsigned)blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), m);
        if (const int rc = PN2_LAUNCH_RC()) return rc;
    }
    return PN2_OK;
}

PN2_EXPORT int pn2_mlp_bwd_post(const float *dw_partial, int P, int N, int K, float *dw, float *db, const float *stat_partial,
                                int Ps, int C, double count, float *dgamma, float *dbeta, float *c1, float *c2,
                                pn2_stream_t stream)
{
    PN2_REQUIRE_PTR(dw_partial); PN2_REQUIRE_PTR(dw); PN2_REQUIRE_PTR(stat_partial); PN2_REQUIRE_PTR(c1); PN2_REQUIRE_PTR(c2);
    if (P <= 0 || N <= 0 || K <= 0 || Ps <= 0 || C <= 0 || count <= 0) return PN2_ERR_SHAPE;
    return pn2::launch_bwd_post(dw_partial, P, N, K, dw, db, stat_partial, Ps, C, count, dgamma, dbeta, c1, c2,
                                static_cast<hipStream_t>(stream));
}

PN2_EXPORT int pn2_mlp_dw(const float *g, int ldg, const float *z, int ldz, const unsigned char *argk, int pool_k,
                          const float *scale, const float *shift, const float *mean, const float *invstd,
                          const float *c1, const float *c2, const float *x1, int ld1, int K1, const float *x2,
                          int ld2, int K2, const float *ascale, const float *ashift, int M, int N,
                          float *partial, float *dw, float *db, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(g);
    PN2_REQUIRE_PTR(z);
    PN2_REQUIRE_PTR(scale);
    PN2_REQUIRE_PTR(shift);
    PN2_REQUIRE_PTR(mean);
    PN2_REQUIRE_PTR(invstd);
    PN2_REQUIRE_PTR(c1);
    PN2_REQUIRE_PTR(c2);
    PN2_REQUIRE_PTR(x1);
    PN2_REQUIRE_PTR(partial);
    if (M <= 0 || N <= 0 || K1 <= 0 || K2 < 0 || (K2 > 0 && !x2) || (argk && pool_k <= 0)) return PN2_ERR_SHAPE;
    if ((ascale == nullptr) != (ashift == nullptr)) return PN2_ERR_NULL;
    DwArgs a;
    a.g = g; a.z = z; a.ldg = ldg; a.ldz = ldz; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd;
    a.c1 = c1; a.c2 = c2; a.argk = argk; a.pool_k = pool_k; a.x1 = x1; a.x2 = x2; a.ld1 = ld1; a.ld2 = ld2;
    a.K1 = K1; a.K2 = K2; a.ascale = ascale; a.ashift = ashift; a.partial = partial; a.M = M; a.N = N;
    const int K = K1 + K2, P = pn2_mlp_dw_partials(M, N, K);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    dim3 grid((unsigned)P, (unsigned)((N + DW_BN - 1) / DW_BN), (unsigned)((K + 1 + DW_BK - 1) / DW_BK));
    bool vec4 = (N % 4 == 0) && (ldg % 4 == 0) && (ldz % 4 == 0) && (K1 % 4 == 0) && (ld1 % 4 == 0) && (K % 4 == 0) &&
                aligned16(g) && aligned16(z) && aligned16(x1) && aligned16(scale) && aligned16(shift) &&
                aligned16(mean) && aligned16(invstd) && aligned16(c1) && aligned16(c2);
    if (x2) vec4 = vec4 && (ld2 % 4 == 0) && aligned16(x2);
    if (ascale) vec4 = vec4 && aligned16(ascale) && aligned16(ashift);
    if (argk) vec4 = vec4 && ((reinterpret_cast<uintptr_t>(argk) & 3) == 0);
    if (vec4 && N >= 128 && K >= 128 && pn2::tune_get("mlp_dw128", 0)) {   // measured: no gain over the 64x64 form
        grid.y = (unsigned)((N + 127) / 128);
        grid.z = (unsigned)((K + 127) / 128);
        if (argk) hipLaunchKernelGGL(mlp_dw128_kernel<true>, grid, dim3(MLP_THREADS), 0, stream, a);
        else hipLaunchKernelGGL(mlp_dw128_kernel<false>, grid, dim3(MLP_THREADS), 0, stream, a);
    } else if (vec4 && dw_split128_applies(N, K)) {
        grid.y = (unsigned)((N + 127) / 128);
        grid.z = (unsigned)((K + 127) / 128);
        static pn2::PerDevice memo_p, memo_d;
        if (argk) {
            if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_dw_split128_kernel<true>), (int)DWW_LDS, memo_p)) return e;
            hipLaunchKernelGGL(mlp_dw_split128_kernel<true>, grid, dim3(DWS_THREADS), DWW_LDS, stream, a);
        } else {
            if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_dw_split128_kernel<false>), (int)DWW_LDS, memo_d)) return e;
            hipLaunchKernelGGL(mlp_dw_split128_kernel<false>, grid, dim3(DWS_THREADS), DWW_LDS, stream, a);
        }
    } else if (vec4 && pn2::tune_get("mlp_dw_split", 1)) {
        grid.z = (unsigned)((K + DW_BK - 1) / DW_BK);             // the bias column is summed on the side
        static pn2::PerDevice memo_p, memo_d;
        if (argk) {
            if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_dw_split_kernel<true>), (int)DWS_LDS, memo_p)) return e;
            hipLaunchKernelGGL(mlp_dw_split_kernel<true>, grid, dim3(DWS_THREADS), DWS_LDS, stream, a);
        } else {
            if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_dw_split_kernel<false>), (int)DWS_LDS, memo_d)) return e;
            hipLaunchKernelGGL(mlp_dw_split_kernel<false>, grid, dim3(DWS_THREADS), DWS_LDS, stream, a);
        }
    } else if (vec4) {
        grid.z = (unsigned)((K + DW_BK - 1) / DW_BK);             // the bias column is summed on the side
        if (argk) hipLaunchKernelGGL((mlp_dw_kernel<true, true>), grid, dim3(MLP_THREADS), 0, stream, a);
        else hipLaunchKernelGGL((mlp_dw_kernel<true, false>), grid, dim3(MLP_THREADS), 0, stream, a);
    } else {
        hipLaunchKernelGGL((mlp_dw_kernel<false, false>), grid, dim3(MLP_THREADS), 0, stream, a);
    }
    int rc = PN2_LAUNCH_RC();
    if (rc != PN2_OK || dw == nullptr) return rc;         // dw NULL: slabs only, reduced later by pn2_mlp_bwd_post
    return pn2::launch_dw_reduce(partial, P, N, K, dw, db, stream);
}

PN2_EXPORT int pn2_bn_bwd_reduce_partials(long long rows)
{
    // 8 row slots per workgroup x 4 rows per pass: one pass per thread up to 16 384 rows; bn_bwd_finalize sums up to 512
    // partials in one round trip
    const long long b = (rows + 31) / 32;
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

PN2_EXPORT int pn2_bn_bwd_reduce(const float *g, int ldg, const float *z, int ldz, long long rows, int C,
                                 const unsigned char *argk, int pool_k, const float *scale, const float *shift,
                                 const float *mean, const float *invstd, float *partial, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(g);
    PN2_REQUIRE_PTR(z);
    PN2_REQUIRE_PTR(scale);
    PN2_REQUIRE_PTR(shift);
    PN2_REQUIRE_PTR(mean);
    PN2_REQUIRE_PTR(invstd);
    PN2_REQUIRE_PTR(partial);
    if (rows <= 0 || C <= 0 || (argk && pool_k <= 0)) return PN2_ERR_SHAPE;
    dim3 grid((unsigned)pn2_bn_bwd_reduce_partials(rows), (unsigned)((C + 31) / 32));
    if (!argk && C % 4 == 0 && ldg % 4 == 0 && ldz % 4 == 0 && aligned16(g) && aligned16(z) && aligned16(scale) &&
        aligned16(shift) && aligned16(mean) && aligned16(invstd) && aligned16(partial)) {
        grid.y = (unsigned)((C + 127) / 128);
        hipLaunchKernelGGL(bn_bwd_reduce_vec4_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), g, ldg, z, ldz, rows,
                           C, scale, shift, mean, invstd, partial);
        return PN2_LAUNCH_RC();
    }
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), g, ldg, z, ldz, rows,
                       C, argk, pool_k, scale, shift, mean, invstd, partial);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_bn_bwd_finalize(const float *partial, int P, int C, double count, float *dgamma, float *dbeta,
                                   float *c1, float *c2, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(partial);
    PN2_REQUIRE_PTR(c1);
    PN2_REQUIRE_PTR(c2);
    if (P <= 0 || C <= 0 || count <= 0) return PN2_ERR_SHAPE;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, static_cast<hipStream_t>(stream_),
                       partial, P, C, count, dgamma, dbeta, c1, c2);
    return PN2_LAUNCH_RC();
}
