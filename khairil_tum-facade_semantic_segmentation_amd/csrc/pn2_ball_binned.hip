// query_ball_point + grouping over a prebuilt query plan (gfx950).
//
// Reference: models/pointnet2_utils.py:87-107 (query_ball_point) and :127-132 (the grouping half of
// sample_and_group).  The plan of a block (pn2_ball_bin.h: cell-sorted points, per-centroid candidate runs,
// packed rows) is built once -- by the tail of the farthest-point-sampling kernel or by ball_bin_kernel +
// ball_pack_rows_kernel below -- and ball_query_binned_kernel only does the per-centroid work: 16 lanes per
// centroid test the candidates of the 27 neighbouring cells with the reference's exact expression
// (pn2::pair_sqdist), members set their bit in a bitmap indexed by the ORIGINAL point index (so "the nsample
// lowest indices, padded with the lowest" are the first set bits), the lane that found a member ranks it by
// a popcount below its bit, then the same lanes write idx and gather / store the grouped rows
// [xyz - centroid, feats].  There is no workgroup barrier and no LDS staging of the block: candidates are
// 16-byte loads from the L2-resident plan, and the row stores are write-through, 256 contiguous bytes per
// centroid and instruction.  Same indices as the reference, bit for bit.
#include <math.h>

#include "pn2_ball_bin.h"

namespace {

using pn2::BinHeader;

constexpr int BQ_THREADS = 256;
constexpr int BQ_MAXK = 64;
constexpr int BQ_ROW = BQ_MAXK + 2;         // result slots per centroid in LDS: K indices + a dump slot

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v3i __attribute__((ext_vector_type(3)));

// rows [b][j] = [x, y, z, feats(D), 0...] at pitch rp: one float4 of one row per call
__device__ __forceinline__ void pack_row_quarter(long long t, const float *__restrict__ xyz, const float *__restrict__ points, int N, int D,
                                                 int rp, char *__restrict__ tables, size_t table_stride, size_t rows_off)
{
    const int qpr = rp >> 4;                       // float4 per packed row
    const long long row = t / qpr;
    const int part = (int)(t - row * qpr);
    const int b = (int)(row / N), j = (int)(row - (long long)b * N);
    const float *x = xyz + (size_t)row * 3;
    const float *f = points ? points + (size_t)row * D : nullptr;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = part * 4 + c;              // column of the grouped row: 0..2 xyz, 3.. feats
        v[c] = col < 3 ? x[col] : (col < 3 + D ? f[col - 3] : 0.0f);
    }
    *reinterpret_cast<float4 *>(tables + (size_t)b * table_stride + rows_off + (size_t)j * rp + (size_t)part * 16) =
        make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void ball_pack_rows_kernel(const float *__restrict__ xyz, const float *__restrict__ points, int N, int D,
                                                             int rp, char *__restrict__ tables, size_t table_stride, size_t rows_off,
                                                             long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < total) pack_row_quarter(t, xyz, points, N, D, rp, tables, table_stride, rows_off);
}

// ---- stand-alone producers of the plan: one workgroup per block bins and plans, many pack the rows ---------------
template <int P>
__global__ __launch_bounds__(1024) void ball_bin_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz, int N, int S, int D,
                                                        float r2, char *__restrict__ tables, size_t table_stride, int B,
                                                        const float *__restrict__ points, long long pack_total)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= B) {
        // workgroups past the B binning ones pack the gather rows in the same launch (they run on the CUs the
        // one-workgroup-per-block binning leaves idle)
        const long long t = ((long long)blockIdx.x - B) * 1024 + threadIdx.x;
        if (t < pack_total)
            pack_row_quarter(t, xyz, points, N, D, pn2::bin_row_pitch(D), tables, table_stride, pn2::bin_rows_off(N, S));
        return;
    }
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *bx = xyz + (size_t)b * N * 3;
    PN2_STAMP(0);
    float px[P], py[P], pz[P];
    if (P == 4 && (N & 3) == 0 && (reinterpret_cast<uintptr_t>(xyz) & 15) == 0) {
        const int j0 = tid * 4;
        const float4 *src = reinterpret_cast<const float4 *>(bx + (size_t)(j0 < N ? j0 : 0) * 3);
        const float4 q0 = src[0], q1 = src[1], q2 = src[2];
        px[0] = q0.x; py[0] = q0.y; pz[0] = q0.z;
        px[1] = q0.w; py[1] = q1.x; pz[1] = q1.y;
        px[2] = q1.z; py[2] = q1.w; pz[2] = q2.x;
        px[3] = q2.y; py[3] = q2.z; pz[3] = q2.w;
    } else {
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int j = tid * P + k;
            const int jj = j < N ? j : 0;
            px[k] = bx[jj * 3 + 0];
            py[k] = bx[jj * 3 + 1];
            pz[k] = bx[jj * 3 + 2];
        }
    }
    PN2_STAMP(1);
    pn2::bin_block<1024, P>(px, py, pz, N, r2, D, new_xyz + (size_t)b * S * 3, S, smem, tables + (size_t)b * table_stride);
}

// ---- the query ---------------------------------------------------------------------------------------------------
// 16 lanes per centroid, NW bitmap words per centroid (N <= 32 * NW).  A workgroup of 256 threads handles 16 centroids;
// nothing in it synchronises across waves.  The kernel is bound by the vector instructions it issues (rocprofv3:
// SQ_ACTIVE_INST_VALU, 4 clocks per instruction with four waves on a SIMD), so every phase is written for few
// instructions: addresses are 32-bit offsets into buffer descriptors, members are ranked by the lane that found them
// (a popcount below their bit) instead of being extracted from the bitmap word by word.
constexpr int BQ_LPC = 16;
constexpr int BQ_CENT = BQ_THREADS / BQ_LPC;
constexpr int BQ_STACK = 24;                   // members a lane can hold before the workgroup's wave falls back to extraction

template <int NW>
__device__ __forceinline__ void ball_query_tile(unsigned logical, unsigned *bm, unsigned short *pre, unsigned short *mIdx, unsigned short *stack,
    const char *__restrict__ tables, size_t table_stride, int sorted_off, int rows_off, int rp, const float *__restrict__ new_xyz,
    int N, int S, int K, int D, float r2, int tiles_per_block, unsigned qpr_magic, int64_t *__restrict__ idx,
    float *__restrict__ grouped, int32_t *err_count)
{
    constexpr int LPC = BQ_LPC, CENT = BQ_CENT;
    constexpr int WPL = NW / LPC;                // bitmap words per lane
    static_assert(WPL >= 4 && WPL % 4 == 0, "a lane's bitmap words are moved as 16-byte vectors");
    const int b = (int)(logical / (unsigned)tiles_per_block);
    const int tile = (int)(logical - (unsigned)b * (unsigned)tiles_per_block);
    const int tid = threadIdx.x, cl = tid >> 4, lg = tid & (LPC - 1);
    const int my_s = tile * CENT + cl;
    if (my_s >= S) return;                           // no workgroup barrier anywhere below
    PN2_STAMP(0);
    const char *tb = tables + (size_t)b * table_stride;
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(tb), 0, (int)table_stride, 0x00020000);
    const __amdgpu_buffer_rsrc_t crs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(new_xyz + (size_t)b * S * 3), 0, S * 12, 0x00020000);
    const v3i cbits = __builtin_amdgcn_raw_buffer_load_b96(crs, my_s * 12, 0, 0);
    // the centroid's plan: nine candidate runs and the test-everything flag
    const v4i pl0 = __builtin_amdgcn_raw_buffer_load_b128(trs, my_s * 64, (int)pn2::BIN_PLAN_OFF, 0);
    const v4i pl1 = __builtin_amdgcn_raw_buffer_load_b128(trs, my_s * 64 + 16, (int)pn2::BIN_PLAN_OFF, 0);
    const v4i pl2 = __builtin_amdgcn_raw_buffer_load_b128(trs, my_s * 64 + 32, (int)pn2::BIN_PLAN_OFF, 0);
    const float cx = __int_as_float(cbits.x), cy = __int_as_float(cbits.y), cz = __int_as_float(cbits.z);
    const float cn = pn2::norm3(cx, cy, cz);

    unsigned *mybm = bm + cl * NW;
#pragma unroll
    for (int k = 0; k < WPL / 4; ++k) reinterpret_cast<uint4 *>(mybm + lg * WPL)[k] = make_uint4(0u, 0u, 0u, 0u);

    const int SORT0 = sorted_off;
    int cnt = 0;                                     // members this lane found (its stack fill; may exceed BQ_STACK)
    unsigned short *mystack = stack + tid;
    // one candidate: the reference's exact expression (:37-39), then its mask (:102)
    auto test = [&](const v4i &p, bool valid) {
        const float x = __int_as_float(p.x), y = __int_as_float(p.y), z = __int_as_float(p.z);
        const float d = pn2::pair_sqdist(cx, cy, cz, cn, x, y, z, pn2::norm3(x, y, z));
        const bool hit = valid & !(d > r2);          // both sides evaluated: nothing for the compiler to sink under `valid`
        if (hit) {
            const unsigned i = (unsigned)p.w;
            atomicOr(&mybm[i >> 5], 1u << (i & 31u));
            mystack[min(cnt, BQ_STACK - 1) * BQ_THREADS] = (unsigned short)i;
            ++cnt;
        }
    };
    // The loaded words are only used under the member test; without this the compiler sinks the load itself into
    // that branch (load, wait, use -- once per candidate).  An empty asm that "modifies" them pins the loads where
    // they are issued.
    auto pin = [](v4i &p) { asm volatile("" : "+v"(p)); };
    // ---- candidates ------------------------------------------------------------------------------------
    const bool full = pl2.y != 0;
    if (!full) {
        const unsigned rng[9] = {(unsigned)pl0.x, (unsigned)pl0.y, (unsigned)pl0.z, (unsigned)pl0.w, (unsigned)pl1.x,
                                 (unsigned)pl1.y, (unsigned)pl1.z, (unsigned)pl1.w, (unsigned)pl2.x};
        PN2_STAMP(1);
        // Two loads per run cover its first 32 slots unmasked: a slot past the run's end holds a point of another
        // cell (or the far-away padding), and a point that passes the exact test IS a member -- it is found through
        // its own run as well, sets the same bit and takes the same rank twice.
        constexpr int FIRST = 2;
        unsigned o0[9];                              // byte offset of this lane's first slot, per run
        v4i p[9][FIRST];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            o0[i] = ((rng[i] & 0xffffu) + (unsigned)lg) << 4;
#pragma unroll
            for (int u = 0; u < FIRST; ++u) p[i][u] = __builtin_amdgcn_raw_buffer_load_b128(trs, (int)o0[i] + u * LPC * 16, SORT0, 0);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
#pragma unroll
            for (int u = 0; u < FIRST; ++u) pin(p[i][u]);
#pragma unroll
            for (int u = 0; u < FIRST; ++u) test(p[i][u], true);
            const unsigned oe = (rng[i] >> 16) << 4;
            for (unsigned o = o0[i] + FIRST * LPC * 16; o < oe; o += 2 * LPC * 16) {     // long runs (dense blocks)
                v4i q0 = __builtin_amdgcn_raw_buffer_load_b128(trs, (int)o, SORT0, 0);
                v4i q1 = __builtin_amdgcn_raw_buffer_load_b128(trs, (int)o + LPC * 16, SORT0, 0);
                pin(q0);
                pin(q1);
                test(q0, true);
                test(q1, true);
            }
            // (four masked loads per round trip here: facade 14.4 -> 13.8 us, but the sparse case 9.35 -> 9.6 us)
        }
    } else {
        for (int j = lg; j < N; j += LPC) {
            v4i q = __builtin_amdgcn_raw_buffer_load_b128(trs, j * 16, SORT0, 0);
            pin(q);
            test(q, true);
        }
    }
    // LDS operations of one wave complete in order; this keeps the compiler from moving the reads up
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    PN2_STAMP(2);

    // ---- rank of every member = members below it: lane lg owns words WPL*lg .. WPL*lg + WPL-1 and publishes how many
    //      members precede each of them; the lane that found a member then places it -----------------------------
    unsigned short *oi = mIdx + cl * BQ_ROW;
    int n;
    {
        unsigned w[WPL];
#pragma unroll
        for (int k = 0; k < WPL / 4; ++k) {
            const uint4 t = reinterpret_cast<const uint4 *>(mybm + lg * WPL)[k];
            w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
        }
        int run[WPL + 1];                            // members in this lane's words before word k
        run[0] = 0;
#pragma unroll
        for (int k = 0; k < WPL; ++k) run[k + 1] = run[k] + __builtin_popcount(w[k]);
        const int mine = run[WPL];
        int inc = mine;                              // inclusive prefix over the centroid's 16 lanes
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);      // row_shr:1
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);      // row_shr:2
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);      // row_shr:4
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);      // row_shr:8
        int total = mine;                            // the centroid's member count, in every lane of its row
        total += pn2::dpp_i32<0xB1>(total);          // quad_perm [1,0,3,2]
        total += pn2::dpp_i32<0x4E>(total);          // quad_perm [2,3,0,1]
        total += pn2::dpp_i32<0x141>(total);         // row_half_mirror
        total += pn2::dpp_i32<0x140>(total);         // row_mirror
        const int base = inc - mine;
        unsigned short *mypre = pre + cl * NW;
#pragma unroll
        for (int k = 0; k < WPL / 4; ++k) {
            uint2 t;
            t.x = (unsigned)(base + run[4 * k]) | ((unsigned)(base + run[4 * k + 1]) << 16);
            t.y = (unsigned)(base + run[4 * k + 2]) | ((unsigned)(base + run[4 * k + 3]) << 16);
            reinterpret_cast<uint2 *>(mypre + lg * WPL)[k] = t;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool overflow = __builtin_amdgcn_ballot_w64(cnt > BQ_STACK) != 0ull;
        if (!overflow) {
            for (int sidx = 0; sidx < cnt; ++sidx) {
                const unsigned i = mystack[sidx * BQ_THREADS];
                const unsigned word = mybm[i >> 5];
                const int below = mypre[i >> 5];
                const int rank = below + __builtin_popcount(word & ((1u << (i & 31u)) - 1u));
                if (rank < K) oi[rank] = (unsigned short)i;
            }
        } else {
            // a lane found more members than its stack holds (dense block, or a centroid that tests every point):
            // extract this lane's bits word by word
#pragma unroll 1
            for (int k = 0; k < WPL; ++k) {
                unsigned bits = mybm[lg * WPL + k];
                int pos = mypre[lg * WPL + k];
                while (bits && pos < K) {
                    oi[pos++] = (unsigned short)((lg * WPL + k) * 32 + __builtin_ctz(bits));
                    bits &= bits - 1u;
                }
            }
        }
        n = min(total, K);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    PN2_STAMP(3);
    // ---- idx [b, s, 0..K) (padded with the first member, :104-106); write-through stores -----------------
    {
        if ((K & 1) == 0) {
            int64_t *iblock = idx + (size_t)b * S * K;
            const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(iblock, 0, (int)((unsigned)S * (unsigned)K * 8u), 0x00020000);
            const unsigned ibase = (unsigned)my_s * (unsigned)K * 8u;
            for (int k = 2 * lg; k < K; k += 2 * BQ_LPC) {
                int a0 = N, a1 = N;                                           // empty: IndexError at :59
                if (n > 0) { a0 = oi[k < n ? k : 0]; a1 = oi[k + 1 < n ? k + 1 : 0]; }
                v4i v;
                v.x = a0; v.y = 0; v.z = a1; v.w = 0;
                __builtin_amdgcn_raw_buffer_store_b128(v, irsrc, (int)(ibase + (unsigned)k * 8u), 0, 16);
            }
        } else {
            int64_t *orow = idx + ((size_t)b * S + my_s) * K;
            for (int k = lg; k < K; k += BQ_LPC) orow[k] = n > 0 ? (int64_t)oi[k < n ? k : 0] : (int64_t)N;
        }
        if (n == 0 && lg == 0 && err_count) atomicAdd(err_count, 1);
    }
    PN2_STAMP(4);
    if (!grouped) return;
    // ---- grouped rows [xyz - centroid, feats] of this centroid: K rows of qpr float4, contiguous; lane lg writes
    //      float4 number lg + 16 i, gathered from the table's packed rows (one cache line per neighbour); an offset
    //      past the descriptor's range reads zeros and touches no memory.
    const int Cg = 3 + D;
    const int qpr = Cg >> 2;
    const int E = K * qpr;
    // one descriptor per block: the rows of this block's centroids (S*K*Cg floats < 4 GB, checked by the launcher)
    float *gblock = grouped + (size_t)b * S * K * Cg;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(gblock, 0, (int)((unsigned)S * (unsigned)K * (unsigned)Cg * 4u), 0x00020000);
    const unsigned gbase = ((unsigned)my_s * (unsigned)K * (unsigned)Cg + (unsigned)lg * 4u) * 4u;
    const int ROWS0 = rows_off;
    const bool some = n > 0;
    const float gx = some ? cx : 0.0f, gy = some ? cy : 0.0f, gz = some ? cz : 0.0f;   // an empty ball stores zero rows
    constexpr int U = 6;
    for (int e0 = lg; e0 < E; e0 += U * LPC) {
        v4i f[U];
        bool q0[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u * LPC, E - 1);
            const int k = qpr == 1 ? e : (int)__umulhi((unsigned)e, qpr_magic);             // e / qpr
            const int part = e - k * qpr;
            q0[u] = part == 0;
            const unsigned j = some ? (unsigned)oi[k < n ? k : 0] : 0x00ffffffu;            // padded with the first member
            f[u] = __builtin_amdgcn_raw_buffer_load_b128(trs, (int)(j * (unsigned)rp + (unsigned)part * 16u), ROWS0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e0 + u * LPC < E) {
                v4i v = f[u];
                if (q0[u]) {                                                              // :128
                    v.x = __float_as_int(__int_as_float(v.x) - gx);
                    v.y = __float_as_int(__int_as_float(v.y) - gy);
                    v.z = __float_as_int(__int_as_float(v.z) - gz);
                }
                __builtin_amdgcn_raw_buffer_store_b128(v, grs, (int)(gbase + (unsigned)(e0 - lg + u * LPC) * 16u), 0, 16);   // aux 16 = sc1: write-through
            }
        }
    }
    PN2_STAMP(5);
}

// One tile of 16 centroids per workgroup.  (Several tiles per workgroup one after the other, so that a tile's row
// stores drain under the next tile's tests, measured slower -- 14.3 us for two, 21.9 us for four against 10.6 us: a
// wave needs ~5.5 us for a tile even with a SIMD to itself, the launch is bound by that dependent chain and by the
// vector instructions of the four waves sharing a SIMD, not by the stores.)
template <int NW>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_binned_kernel(
    const char *__restrict__ tables, size_t table_stride, int sorted_off, int rows_off, int rp, const float *__restrict__ new_xyz,
    int N, int S, int K, int D, float r2, int tiles_per_block, unsigned qpr_magic, int64_t *__restrict__ idx,
    float *__restrict__ grouped, int32_t *err_count)
{
    __shared__ __attribute__((aligned(16))) unsigned bm[BQ_CENT * NW];                 // member bitmaps (bit = original index)
    __shared__ __attribute__((aligned(16))) unsigned short pre[BQ_CENT * NW];          // members below each bitmap word
    __shared__ __attribute__((aligned(16))) unsigned short mIdx[BQ_CENT * BQ_ROW];     // result indices, ascending (+ a dump slot)
    __shared__ unsigned short stack[BQ_STACK * BQ_THREADS];                            // [slot][thread]: members a lane found
    ball_query_tile<NW>(pn2::xcd_remap(blockIdx.x, gridDim.x), bm, pre, mIdx, stack, tables, table_stride, sorted_off, rows_off, rp, new_xyz,
                        N, S, K, D, r2, tiles_per_block, qpr_magic, idx, grouped, err_count);
}

}  // namespace

namespace pn2 {
// the binning / planning kernel alone (also the second launch of pn2_farthest_point_sample_plan)
int launch_ball_bin(const float *xyz, const float *new_xyz, int B, int N, int S, int D, float r2, char *plans, hipStream_t stream,
                    const float *pack_points = nullptr, bool pack = false)
{
    const size_t lds = bin_lds_bytes<1024>();
    const size_t stride = bin_block_bytes(N, S, D);
    const long long pack_total = pack ? (long long)B * N * (bin_row_pitch(D) >> 4) : 0;
    const long long grid = (long long)B + (pack_total + 1023) / 1024;
    if (grid > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
#define PN2_BIN(P) hipLaunchKernelGGL(ball_bin_kernel<P>, dim3((unsigned)grid), dim3(1024), lds, stream, xyz, new_xyz, N, S, D, r2, plans, stride, B, pack_points, pack_total)
    if (N <= 1024) PN2_BIN(1);
    else if (N <= 2048) PN2_BIN(2);
    else if (N <= 4096) PN2_BIN(4);
    else PN2_BIN(8);
#undef PN2_BIN
    return PN2_LAUNCH_RC();
}
}  // namespace pn2

// ---- C ABI ---------------------------------------------------------------------------------------------------
PN2_EXPORT long long pn2_ball_plan_bytes(int N, int S, int D)
{
    if (N <= 0 || N > 8192 || S <= 0 || D < 0) return 0;
    return (long long)pn2::bin_block_bytes(N, S, D);
}

PN2_EXPORT int pn2_ball_pack_rows(const float *xyz, const float *points, int B, int N, int S, int D, void *plans, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz);
    PN2_REQUIRE_PTR(plans);
    if (B < 0 || N <= 0 || S <= 0 || D < 0) return PN2_ERR_SHAPE;
    if (D > 0 && points == nullptr) return PN2_ERR_NULL;
    if (N > 8192) return PN2_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(plans) & 127) != 0) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    const int rp = pn2::bin_row_pitch(D);
    const long long total = (long long)B * N * (rp >> 4);
    const long long nwg = (total + 255) / 256;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ball_pack_rows_kernel, dim3((unsigned)nwg), dim3(256), 0, static_cast<hipStream_t>(stream_), xyz, points, N, D, rp,
                       static_cast<char *>(plans), pn2::bin_block_bytes(N, S, D), pn2::bin_rows_off(N, S), total);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_ball_plan(double radius, const float *xyz, const float *new_xyz, const float *points, int B, int N, int S, int D,
                             void *plans, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz);
    PN2_REQUIRE_PTR(new_xyz);
    PN2_REQUIRE_PTR(plans);
    if (B < 0 || N <= 0 || S <= 0 || D < 0) return PN2_ERR_SHAPE;
    if (D > 0 && points == nullptr) return PN2_ERR_NULL;
    if (N > 8192) return PN2_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(plans) & 127) != 0) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    const float r2 = (float)(radius * radius);          // python `radius ** 2` (double), compared in fp32
    // binning workgroups and row-packing workgroups in ONE launch
    return pn2::launch_ball_bin(xyz, new_xyz, B, N, S, D, r2, static_cast<char *>(plans), static_cast<hipStream_t>(stream_), points, true);
}

PN2_EXPORT int pn2_ball_query_group_planned(double radius, int nsample, const void *plans, const float *xyz, const float *new_xyz,
                                            const float *points, int B, int N, int S, int D, int64_t *idx, float *grouped, int ldg,
                                            int32_t *err_count, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(plans);
    PN2_REQUIRE_PTR(xyz);
    PN2_REQUIRE_PTR(new_xyz);
    PN2_REQUIRE_PTR(idx);
    if (B < 0 || N <= 0 || S <= 0 || D < 0 || nsample <= 0) return PN2_ERR_SHAPE;
    if (nsample > BQ_MAXK || N > 8192) return PN2_ERR_UNSUPPORTED;
    if (ldg == 0) ldg = 3 + D;
    if (ldg < 3 + D) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const float r2 = (float)(radius * radius);
    const size_t stride = pn2::bin_block_bytes(N, S, D);
    if (stride >= 0x7fffffffull || (unsigned long long)S * nsample * 8ull >= 0xffffffffull) return PN2_ERR_UNSUPPORTED;
    // rows the fused store path handles: dense pitch, a whole number of float4 per row, 16-byte aligned, < 4 GB per block
    const bool fused = grouped && ldg == 3 + D && ((3 + D) & 3) == 0 && (reinterpret_cast<uintptr_t>(grouped) & 15) == 0 &&
                       (unsigned long long)S * nsample * (3 + D) * 4ull < 0xffffffffull;
    if (grouped && !fused && D > 0 && points == nullptr) return PN2_ERR_NULL;     // only the separate grouping pass reads `points`
    const int tiles = (S + BQ_CENT - 1) / BQ_CENT;
    const long long nwg = (long long)B * tiles;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    const int qpr = (3 + D) >> 2;
    const unsigned magic = qpr > 1 ? (unsigned)((1ULL << 32) / (unsigned)qpr) + 1u : 0u;     // e/qpr exact for e*qpr < 2^32
    const char *t = static_cast<const char *>(plans);
    float *g = fused ? grouped : nullptr;
#define PN2_BQB(NW)                                                                                                           \
    hipLaunchKernelGGL((ball_query_binned_kernel<NW>), dim3((unsigned)nwg), dim3(BQ_THREADS), 0, stream, t, stride,          \
                       (int)pn2::bin_sorted_off(S), (int)pn2::bin_rows_off(N, S), pn2::bin_row_pitch(D), new_xyz, N, S, nsample, D, r2, \
                       tiles, magic, idx, g, err_count)
    if (N <= 2048) PN2_BQB(64);
    else if (N <= 4096) PN2_BQB(128);
    else PN2_BQB(256);
#undef PN2_BQB
    int rc = PN2_LAUNCH_RC();
    if (rc != PN2_OK || !grouped || fused) return rc;
    return pn2_group_points(xyz, new_xyz, points, idx, B, N, S, nsample, D, grouped, ldg, nullptr, stream_);
}
