// query_ball_point + grouping, matrix-core form of the pair test (gfx950).
//
// The reference computes the [S,N] distance matrix with a matmul (models/pointnet2_utils.py:37)
// and adds the norms (:38-39); the VALU kernel in pn2_ball_group.hip spends 6 vector
// instructions per 64 pairs on that expression and is bound by vector-instruction issue
// (rocprofv3 PMC).  Here the exact expression is evaluated by the matrix pipe:
//
//     f = fma(1, r2, fma(1, -|p|^2, fma(-|c|^2, 1, fma(2cz, pz, fma(2cy, py, fma(2cx, px, 0))))))
//                                                              three v_mfma_f32_32x32x2_f32
//
// v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fp32 fma chain (cdna_hip_programming.md
// section 3); scaling by 2 and negation commute with round-to-nearest, so the chain holds, step by
// step, 2*dot, -((-2*dot) + |c|^2), -dist and finally round(r2 - dist) with `dist` exactly the
// reference's value (:37-39).  A point is inside the ball (not masked at :102) iff dist <= r2 iff
// the sign bit of f is clear (r2 - dist == +0 when equal).  32 centroids x 32 points per MFMA
// triple; the vector unit only collects 16 sign bits per lane (v_alignbit_b32).
//
// Work split: a 1024-thread workgroup = 2 centroid groups (32 each) x 8 point slices; every
// wave scans its slice for its 32 centroids and appends hits to a per-(centroid, slice)
// sub-list with LDS atomics; sub-lists are verified ascending (re-ranked if an atomic pair landed
// out of order), concatenated in quarter order and truncated to nsample, which is exactly "the
// nsample lowest indices" (:103).  Grouping as in pn2_ball_group.hip.
#include <math.h>

#include "pn2_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MF_THREADS = 1024;
constexpr int MF_WAVES = 16;
constexpr int MF_SLICES = 8;                  // point slices per centroid group (one wave each)
constexpr int MF_CENT = 64;                   // centroids per workgroup (2 groups of 32)
constexpr int MF_MAXN = 4096;                 // whole block resident in LDS as SoA x,y,z,|p|^2

template <bool ORDERED>
__global__ __launch_bounds__(MF_THREADS) void ball_query_group_mfma_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ points,
    int B, int N, int S, int K, int D, int ldg, float r2, int tiles_per_block, unsigned ldg_magic,
    int64_t *__restrict__ idx, float *__restrict__ grouped, int32_t *err_count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NT = (N + 32 * MF_SLICES - 1) / (32 * MF_SLICES) * (32 * MF_SLICES);   // slices of a multiple of 32
    const int NQ = NT / MF_SLICES;
    const int cap = ORDERED ? K : K + 32;                         // sub-list capacity (atomic appends may overshoot)
    float *sX = reinterpret_cast<float *>(smem);
    float *sY = sX + NT;
    float *sZ = sY + NT;
    float *sP = sZ + NT;                                          // -|p|^2 (-inf on padding)
    unsigned *cnt = reinterpret_cast<unsigned *>(sP + NT);        // [64][MF_SLICES]
    int *merged = reinterpret_cast<int *>(cnt + MF_CENT * MF_SLICES);     // [64][K] ushort merged lists (MF_CENT*K/2 ints)
    unsigned short *lists = reinterpret_cast<unsigned short *>(merged + MF_CENT * K / 2 + 2);   // [64][MF_SLICES][cap]

    const unsigned logical = pn2::xcd_remap(blockIdx.x, gridDim.x);
    const int b = (int)(logical / (unsigned)tiles_per_block);
    const int tile = (int)(logical % (unsigned)tiles_per_block);
    const int tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / MF_SLICES, quarter = wave % MF_SLICES;   // `quarter` = this wave's point slice
    const int s_base = tile * MF_CENT;

    const float *bx = xyz + (size_t)b * N * 3;
    const float *bc = new_xyz + (size_t)b * S * 3;

    // ---- stage the block (all loads of a thread issued first) ----------------------------------
    {
        constexpr int PT = MF_MAXN / MF_THREADS;          // 4 points per thread
        float sx[PT], sy[PT], sz[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int j = tid + i * MF_THREADS;
            const int jj = j < N ? j : 0;
            sx[i] = bx[jj * 3 + 0];
            sy[i] = bx[jj * 3 + 1];
            sz[i] = bx[jj * 3 + 2];
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int j = tid + i * MF_THREADS;
            if (j < NT) {
                const bool in = j < N;
                sX[j] = in ? sx[i] : 0.0f;
                sY[j] = in ? sy[i] : 0.0f;
                sZ[j] = in ? sz[i] : 0.0f;
                sP[j] = in ? -pn2::norm3(sx[i], sy[i], sz[i]) : -INFINITY;     // padding: f = -inf, never a hit
            }
        }
        if (tid < MF_CENT * MF_SLICES) cnt[tid] = 0;
    }

    // ---- A operands: lane l -> centroid i = l&31 of this wave's group, k = l>>5 -----------------
    const int my_s = s_base + grp * 32 + l31;
    const bool s_ok = my_s < S;
    const int sc = s_ok ? my_s : S - 1;
    const float cx = bc[sc * 3 + 0], cy = bc[sc * 3 + 1], cz = bc[sc * 3 + 2];
    const float a1 = half ? 2.0f * cy : 2.0f * cx;               // MFMA 1: k0 = 2cx*px, k1 = 2cy*py
    const float a2 = half ? -pn2::norm3(cx, cy, cz) : 2.0f * cz; // MFMA 2: k0 = 2cz*pz, k1 = -|c|^2 * 1
                                                                 // MFMA 3: k0 = 1 * -|p|^2, k1 = 1 * r2
    __syncthreads();

    // ---- scan this wave's quarter: 32 points per step; the MFMAs of step t+1 are issued before the
    //      vector work of step t so the matrix pipe's latency hides under it ---------------------
    const float *pxy = (half ? sY : sX) + quarter * NQ + l31;
    const float *pz = sZ + quarter * NQ + l31;
    const float *pn = sP + quarter * NQ + l31;
    const int nsteps = NQ >> 5;
    const unsigned rows_valid = (unsigned)max(0, min(32, S - (s_base + grp * 32)));   // rows < rows_valid exist
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;

    // one step = 32 points: issue(t) puts the three MFMAs of step t in flight, consume() turns an
    // accumulator into hit bits and appends; the loop is unrolled by two so that the MFMAs of the
    // next step are issued before the vector work of the current one (no register copies).
    auto issue = [&](int t) -> f32x16 {
        const float bxy = pxy[t * 32], bz = pz[t * 32], nnp = pn[t * 32];
        f32x16 a = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bxy, zero16, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, half ? 1.0f : bz, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, half ? r2 : nnp, a, 0, 0, 0);
        return a;
    };
    // ORDERED: hit counters of the wave's 32 centroid rows live in one VGPR (lane i = row i)
    int cntreg = 0;
    const unsigned lane_lt = (1u << l31) - 1u;                  // bits of the lower lanes of this half
    auto consume = [&](const f32x16 &f, int t) {
        // sign bit of f = round(r2 - dist): set <=> dist > r2 <=> masked out (:102); reg r -> bit 15-r
        unsigned bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(f[r]), 31);
        unsigned hits = ~bits & 0xffffu;
        if (!ORDERED) {
            if (__ballot(hits != 0)) {
                const int j = quarter * NQ + t * 32 + l31;
                while (hits) {                                  // this lane's point is inside >= 1 ball
                    const int pbit = __builtin_ctz(hits);
                    hits &= hits - 1;
                    const int r = 15 - pbit;
                    const unsigned row = (unsigned)((r & 3) + 8 * (r >> 2) + 4 * half);   // C/D row of register r
                    if (row < rows_valid) {
                        const int c = grp * 32 + (int)row;
                        const unsigned pos = atomicAdd(&cnt[c * MF_SLICES + quarter], 1u);
                        if (pos < (unsigned)cap) lists[(c * MF_SLICES + quarter) * cap + pos] = (unsigned short)j;
                    }
                }
            }
        } else {
            // Which registers hold a hit anywhere in the wave?  (OR over the 64 lanes, uniform result)
            int u = (int)hits;
            u |= pn2::dpp_i32<0xB1>(u);
            u |= pn2::dpp_i32<0x4E>(u);
            u |= pn2::dpp_i32<0x141>(u);
            u |= pn2::dpp_i32<0x140>(u);
            unsigned U = (unsigned)(__builtin_amdgcn_readlane(u, 0) | __builtin_amdgcn_readlane(u, 16) |
                                    __builtin_amdgcn_readlane(u, 32) | __builtin_amdgcn_readlane(u, 48));
            const int j = quarter * NQ + t * 32 + l31;
            while (U) {                                         // uniform loop: one register (= 2 rows) per turn
                const int pbit = __builtin_ctz(U);
                U &= U - 1;
                const int r = 15 - pbit;
                const int rowA = (r & 3) + 8 * (r >> 2), rowB = rowA + 4;          // lanes < 32 / >= 32
                const bool hit = (hits >> pbit) & 1u;
                const unsigned long long m = __ballot(hit);
                const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
                const int cA = __builtin_amdgcn_readlane(cntreg, rowA);
                const int cB = __builtin_amdgcn_readlane(cntreg, rowB);
                // ascending point order inside the step: position = hits so far + hits in lower lanes
                const int pos = (half ? cB : cA) + __builtin_popcount((half ? mhi : mlo) & lane_lt);
                const int row = half ? rowB : rowA;
                if (hit && pos < K && (unsigned)row < rows_valid)
                    lists[((grp * 32 + row) * MF_SLICES + quarter) * cap + pos] = (unsigned short)j;
                const int nA = cA + __builtin_popcount(mlo), nB = cB + __builtin_popcount(mhi);
                cntreg = lane == rowA ? nA : (lane == rowB ? nB : cntreg);
            }
        }
    };
    if (nsteps > 0) {
        f32x16 fa = issue(0), fb = zero16;
        int t = 0;
        for (; t + 2 <= nsteps; t += 2) {                       // nsteps = NQ/32 is even (NQ % 64 == 0)? no: handle tail below
            fb = issue(t + 1);
            consume(fa, t);
            if (t + 2 < nsteps) fa = issue(t + 2);
            consume(fb, t + 1);
        }
        if (t < nsteps) consume(fa, t);
    }
    if (ORDERED && lane < 32) cnt[(grp * 32 + lane) * MF_SLICES + quarter] = (unsigned)cntreg;
    __syncthreads();

    // ---- (1) every sub-list must be ascending: arrival order is, except that two hits of one
    //      32-point step may have landed swapped.  One thread per (centroid, quarter) checks and,
    //      if needed, insertion-sorts its sub-list (rare, short). ---------------------------------
    if (!ORDERED && tid < MF_CENT * MF_SLICES) {
        const int nq = (int)min(cnt[tid], (unsigned)cap);
        unsigned short *L = lists + tid * cap;
        bool bad = false;
        for (int k = 0; k + 1 < nq; ++k) bad = bad || (L[k] > L[k + 1]);
        if (bad) {
            for (int k = 1; k < nq; ++k) {
                const unsigned short v = L[k];
                int m = k - 1;
                while (m >= 0 && L[m] > v) { L[m + 1] = L[m]; --m; }
                L[m + 1] = v;
            }
        }
    }
    if (!ORDERED) __syncthreads();

    // ---- (2) merged list = sub-lists in quarter order, first K, padded with the first (:103-106);
    //      one thread per output slot -> idx (int64, coalesced) and an LDS copy for grouping ------
    unsigned short *mlist = reinterpret_cast<unsigned short *>(merged);        // [64][K], 0xffff = empty ball
    for (int e = tid; e < MF_CENT * K; e += MF_THREADS) {
        const int c = e / K, k = e - c * K;
        const int s = s_base + c;
        if (s >= S) continue;
        int total = 0;
#pragma unroll
        for (int q = 0; q < MF_SLICES; ++q) total += (int)min(cnt[c * MF_SLICES + q], (unsigned)cap);
        const int kk = k < total ? k : 0;                       // pad with the first hit
        int q = 0, off = kk;
#pragma unroll
        for (int qq = 0; qq < MF_SLICES - 1; ++qq) {            // walk the slices in index order
            const int nq = (int)min(cnt[c * MF_SLICES + qq], (unsigned)cap);
            if (q == qq && off >= nq) { off -= nq; q = qq + 1; }
        }
        int64_t v = N;                                          // reference: IndexError at :59
        unsigned short vs = 0xffff;
        if (total > 0) { vs = lists[(c * MF_SLICES + q) * cap + off]; v = vs; }
        else if (k == 0 && err_count) atomicAdd(err_count, 1);
        idx[((size_t)b * S + s) * K + k] = v;
        mlist[e] = vs;
    }
    if (!grouped) return;
    __syncthreads();

    // ---- (3) grouped rows, one element group per thread over the whole workgroup ---------------
    const int Cg = 3 + D;
    const int row_elems = K * ldg;
    const float *bp = points ? points + (size_t)b * N * D : nullptr;
    const int ncent = min(MF_CENT, S - s_base);
    float *gbase = grouped + ((size_t)b * S + s_base) * row_elems;
    if (ldg == Cg && (Cg & 3) == 0 && ((reinterpret_cast<uintptr_t>(grouped) & 15) == 0)) {
        // rows of 4*qpr floats: one float4 per thread and step; quad 0 = [xyz - centroid, feat0]
        const int qpr = Cg >> 2;
        const int per_c = K * qpr;
        float4 *g4 = reinterpret_cast<float4 *>(gbase);
        for (int e = tid; e < ncent * per_c; e += MF_THREADS) {
            const int rowk = qpr == 1 ? e : (int)__umulhi((unsigned)e, ldg_magic);      // e / qpr = c*K + k
            const int part = e - rowk * qpr;
            const unsigned short js = mlist[rowk];
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (js != 0xffff) {
                const int j = js;
                const float *row = bp + (size_t)j * D;
                if (part == 0) {
                    const int c = rowk / K;
                    const float *cc = bc + (size_t)(s_base + c) * 3;
                    v = make_float4(sX[j] - cc[0], sY[j] - cc[1], sZ[j] - cc[2], row[0]);       // :128, :131
                } else {
                    const float *src = row + (4 * part - 3);
                    v = make_float4(src[0], src[1], src[2], src[3]);
                }
            }
            g4[e] = v;
        }
    } else {
        // generic pitch: element (row = c*K + k, col); col < 3 xyz, col < Cg feats, else zero pad
        const long long total = (long long)ncent * K * ldg;
        for (long long e = tid; e < total; e += MF_THREADS) {
            const int rowk = (int)(e / ldg), col = (int)(e - (long long)rowk * ldg);
            const unsigned short js = mlist[rowk];
            float v = 0.0f;
            if (js != 0xffff && col < Cg) {
                const int j = js;
                if (col < 3) {
                    const int c = rowk / K;
                    const float pv = col == 0 ? sX[j] : (col == 1 ? sY[j] : sZ[j]);
                    v = pv - bc[(size_t)(s_base + c) * 3 + col];                               // :128
                } else {
                    v = bp[(size_t)j * D + (col - 3)];                                         // :131-132
                }
            }
            gbase[e] = v;
        }
    }
}

}  // namespace

namespace pn2 {

// Returns PN2_ERR_UNSUPPORTED when the shape is outside what this kernel is built for (the caller
// then uses the vector-unit kernel).
int launch_ball_query_mfma(const float *xyz, const float *new_xyz, const float *points, int B, int N, int S, int K,
                           int D, int ldg, float r2, int64_t *idx, float *grouped, int32_t *err_count,
                           hipStream_t stream)
{
    if (N > MF_MAXN || K > 64) return PN2_ERR_UNSUPPORTED;
    // the workgroup-wide grouping pass is built for rows of 4*q floats (SA1: 3+9 = 12); other row
    // shapes go through the vector-unit kernel / the element-wise grouping kernel
    if (grouped && !(ldg == 3 + D && ((3 + D) & 3) == 0)) return PN2_ERR_UNSUPPORTED;
    const int NT = (N + 32 * MF_SLICES - 1) / (32 * MF_SLICES) * (32 * MF_SLICES);
    // ballot-ordered appends (no atomics, no re-check) were measured at 40.6 us against 26.4 us for the
    // LDS-atomic form: kept as an A/B switch only
    const bool ordered = pn2::tune_get("bq_ordered", 0) != 0;
    const int cap = ordered ? K : K + 32;
    const size_t lds = (size_t)NT * 4 * sizeof(float) + MF_CENT * MF_SLICES * sizeof(unsigned) + ((size_t)MF_CENT * K / 2 + 2) * sizeof(int) +
                       (size_t)MF_CENT * MF_SLICES * cap * sizeof(unsigned short);
    if (lds > 160 * 1024) return PN2_ERR_UNSUPPORTED;
    const int tiles = (S + MF_CENT - 1) / MF_CENT;
    const long long nwg = (long long)B * tiles;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    const int Cg = 3 + D;
    const int qpr = (ldg == Cg && (Cg & 3) == 0) ? (Cg >> 2) : 1;
    const unsigned magic = qpr > 1 ? (unsigned)((1ULL << 32) / (unsigned)qpr) + 1u : 0u;     // e/qpr exact for e*qpr < 2^32
    auto kern = ordered ? ball_query_group_mfma_kernel<true> : ball_query_group_mfma_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(MF_THREADS), lds, stream, xyz, new_xyz,
                       points, B, N, S, K, D, ldg, r2, tiles, magic, idx, grouped, err_count);
    return PN2_LAUNCH_RC();
}

}  // namespace pn2
