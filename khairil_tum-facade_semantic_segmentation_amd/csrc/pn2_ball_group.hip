// query_ball_point fused with the grouping half of sample_and_group, on gfx950.
//
// Reference (models/pointnet2_utils.py:87-107, 127-132) materialises a [B,S,N] distance
// matrix, masks it, fully sorts a [B,S,N] int64 tensor and gathers three times.  Here a
// workgroup stages one block's xyz (+|p|^2) in LDS once; every wave owns CPW centroids and
// scans the points 64 at a time in ascending index order (lane = point), two centroids per
// packed-fp32 instruction, appending hits with ballot + mbcnt so the "nsample lowest indices"
// rule holds by construction and a centroid stops as soon as it has nsample hits.  The same
// wave then writes idx (int64) and the grouped rows [xyz-centroid, feats] with coalesced
// stores.  Measured (rocprofv3 PMC): the scan is VALU-issue bound (one wave-instruction per
// SIMD per 4 clocks), so the kernel is written to minimise issued instructions; the roof it is
// priced against is HBM (the grouped-tensor write).  Algorithmic bytes: DESIGN.md.
#include <math.h>

#include "pn2_common.h"

namespace pn2 {
int launch_ball_query_grid(const float *xyz, const float *new_xyz, const float *points, int B, int N, int S, int K,
                           int D, int ldg, float r2, int64_t *idx, float *grouped, int32_t *err_count,
                           hipStream_t stream, bool sampled_box);
}

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int BQ_TILE = 4096;                 // points staged in LDS per pass (64 KiB as float4)
constexpr int BQ_UNROLL = 8;                  // independent gathers in flight per lane (generic group phase)

// Packed fp32 (two centroids per instruction).  Each half is an IEEE op identical to the scalar
// one: mul / fma / add round exactly like v_mul_f32 / v_fma_f32 / v_add_f32.
// p broadcast from the LOW half of `p`: r = (p.lo*c.lo, p.lo*c.hi)
__device__ __forceinline__ v2f pk_mul_plo(v2f p, v2f c)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(p), "v"(c));
    return r;
}
// r = (p.hi*c.lo + a.lo, p.hi*c.hi + a.hi)
__device__ __forceinline__ v2f pk_fma_phi(v2f p, v2f c, v2f a)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(p), "v"(c), "v"(a));
    return r;
}
// r = (p.lo*c.lo + a.lo, p.lo*c.hi + a.hi)
__device__ __forceinline__ v2f pk_fma_plo(v2f p, v2f c, v2f a)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(p), "v"(c), "v"(a));
    return r;
}
// r = (x.lo*y.lo + a.lo, x.hi*y.hi + a.hi)
__device__ __forceinline__ v2f pk_fma(v2f x, v2f y, v2f a)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(a));
    return r;
}
// r = (t.lo + p.hi, t.hi + p.hi)
__device__ __forceinline__ v2f pk_add_phi(v2f t, v2f p)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(t), "v"(p));
    return r;
}

// Two independent chains (centroid pairs A and B) of the five packed ops, interleaved:
//   t = p.x*cx ; t = p.y*cy + t ; t = p.z*cz + t ; t = -2*t + |c|^2 ; t = t + |p|^2
__device__ __forceinline__ void pk_dist_x2(v2f &ta, v2f &tb, v2f pxy, v2f pzw, v2f ax, v2f ay, v2f az, v2f an,
                                           v2f bx, v2f by, v2f bz, v2f bn, v2f m2)
{
    asm("v_pk_mul_f32 %0, %2, %4 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %2, %8 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %5, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %1, %2, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %0, %3, %6, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %3, %10, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %0, %12, %7\n\t"
        "v_pk_fma_f32 %1, %1, %12, %11\n\t"
        "v_pk_add_f32 %0, %0, %3 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %1, %1, %3 op_sel:[0,1] op_sel_hi:[1,1]"
        : "=&v"(ta), "=&v"(tb)
        : "v"(pxy), "v"(pzw), "v"(ax), "v"(ay), "v"(az), "v"(an), "v"(bx), "v"(by), "v"(bz), "v"(bn), "v"(m2));
}

// THREADS per workgroup, CPW (even) centroids per wave.
template <int THREADS, int CPW>
__global__ __launch_bounds__(THREADS) void ball_query_group_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ points,
    int B, int N, int S, int K, int D, int ldg, float r2, int tiles_per_block, unsigned cg_magic,
    int64_t *__restrict__ idx, float *__restrict__ grouped, int32_t *err_count)
{
    static_assert(CPW % 2 == 0, "centroids are processed in packed pairs");
    constexpr int WAVES = THREADS / PN2_WAVE;
    constexpr int PAIRS = CPW / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *pts = reinterpret_cast<float4 *>(smem);                                   // [tile_pts]
    const int tile_pts = min(BQ_TILE, (N + PN2_WAVE - 1) & ~(PN2_WAVE - 1));
    const int lcap = K + PN2_WAVE;                       // a chunk may overshoot K by < 64 entries
    int *lists = reinterpret_cast<int *>(smem + (size_t)tile_pts * sizeof(float4));   // [WAVES][CPW][lcap]

    const unsigned logical = pn2::xcd_remap(blockIdx.x, gridDim.x);
    const int b = (int)(logical / (unsigned)tiles_per_block);
    const int tile = (int)(logical % (unsigned)tiles_per_block);
    const int tid = threadIdx.x;
    const int lane = tid & (PN2_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / PN2_WAVE);
    const int s0 = (tile * WAVES + wave) * CPW;
    int *mylist = lists + (size_t)wave * CPW * lcap;

    const float *bx = xyz + (size_t)b * N * 3;
    const float *bc = new_xyz + (size_t)b * S * 3;

    v2f cx2[PAIRS], cy2[PAIRS], cz2[PAIRS], cn2[PAIRS];
    int cnt[CPW];                                         // hits so far; >= K means "done"
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int s = min(s0 + c, S - 1);
        const float x = bc[s * 3 + 0], y = bc[s * 3 + 1], z = bc[s * 3 + 2];
        cx2[c / 2][c & 1] = x;
        cy2[c / 2][c & 1] = y;
        cz2[c / 2][c & 1] = z;
        cn2[c / 2][c & 1] = pn2::norm3(x, y, z);
        cnt[c] = (s0 + c < S) ? 0 : K;                    // centroids past S never collect
    }
    v2f minus2 = {-2.0f, -2.0f};
    // Pin the per-centroid constants in VGPR pairs: left alone, hipcc keeps these wave-uniform
    // values in SGPRs and re-materialises a VGPR copy (v_mov_b64) in front of every packed op.
    asm volatile("" : "+v"(minus2));
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        asm volatile("" : "+v"(cx2[q]));
        asm volatile("" : "+v"(cy2[q]));
        asm volatile("" : "+v"(cz2[q]));
        asm volatile("" : "+v"(cn2[q]));
    }

    for (int n0 = 0; n0 < N; n0 += BQ_TILE) {
        if (n0) __syncthreads();
        const int npad = min(tile_pts, (N - n0 + PN2_WAVE - 1) & ~(PN2_WAVE - 1));
        // Stage xyz (+|p|^2) of this tile: all of a thread's global loads are issued before the
        // first is consumed (one round of memory latency per tile, not one per point).
        {
            constexpr int PT = BQ_TILE / THREADS;                 // points per thread per tile
            float sx[PT], sy[PT], sz[PT];
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int j = tid + i * THREADS;
                const int g = n0 + j;
                const bool in = j < npad && g < N;
                const int gg = in ? g : 0;
                sx[i] = bx[gg * 3 + 0];
                sy[i] = bx[gg * 3 + 1];
                sz[i] = bx[gg * 3 + 2];
            }
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int j = tid + i * THREADS;
                const int g = n0 + j;
                if (j < npad) {
                    const bool in = g < N;
                    // padding: d = +inf, never a hit
                    pts[j] = in ? make_float4(sx[i], sy[i], sz[i], pn2::norm3(sx[i], sy[i], sz[i]))
                                : make_float4(0.0f, 0.0f, 0.0f, INFINITY);
                }
            }
        }
        __syncthreads();
        const int nchunks = npad / PN2_WAVE;
        float4 pnext = pts[lane];
        for (int ch = 0; ch < nchunks; ++ch) {
            const float4 p = pnext;
            if (ch + 1 < nchunks) pnext = pts[(ch + 1) * PN2_WAVE + lane];      // prefetch: LDS latency off the chain
            const v2f pxy = {p.x, p.y};
            const v2f pzw = {p.z, p.w};
            unsigned long long m[CPW];
            bool hit[CPW];
            unsigned long long any = 0;
            // src = new_xyz (centroid), dst = xyz (point): pointnet2_utils.py:101, 37-39.
            // The five packed ops of two centroid pairs are interleaved inside ONE asm statement:
            // dependent ops never issue back to back and hipcc adds no pad between statements.
            v2f t[PAIRS];
#pragma unroll
            for (int q = 0; q + 1 < PAIRS; q += 2) pk_dist_x2(t[q], t[q + 1], pxy, pzw, cx2[q], cy2[q], cz2[q], cn2[q],
                                                              cx2[q + 1], cy2[q + 1], cz2[q + 1], cn2[q + 1], minus2);
            if (PAIRS & 1) {
                const int q = PAIRS - 1;
                t[q] = pk_mul_plo(pxy, cx2[q]);                  // a0*b0
                t[q] = pk_fma_phi(pxy, cy2[q], t[q]);            // fma(a1,b1,.)
                t[q] = pk_fma_plo(pzw, cz2[q], t[q]);            // fma(a2,b2,.) = dot
                t[q] = pk_fma(t[q], minus2, cn2[q]);             // (-2*dot) + |centroid|^2, one rounding
                t[q] = pk_add_phi(t[q], pzw);                    // + |point|^2
            }
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) {
                hit[2 * q] = !(t[q].x > r2);                     // :102  (hit = not masked out)
                hit[2 * q + 1] = !(t[q].y > r2);
                m[2 * q] = __ballot(hit[2 * q]);
                m[2 * q + 1] = __ballot(hit[2 * q + 1]);
                any |= m[2 * q] | m[2 * q + 1];
            }
            if (any) {
                const int pidx = n0 + ch * PN2_WAVE + lane;
                bool still = false;
#pragma unroll
                for (int c = 0; c < CPW; ++c) {
                    if (m[c] && cnt[c] < K) {                    // a full centroid ignores later hits
                        const int pos = cnt[c] + pn2::mbcnt(m[c]);
                        if (hit[c]) mylist[c * lcap + pos] = pidx;
                        cnt[c] += __builtin_popcountll(m[c]);
                    }
                    still = still || (cnt[c] < K);
                }
                if (!still) break;                               // every centroid of this wave is full
            }
        }
    }

    // ---- pad + idx (int64) -------------------------------------------------------------
    const int Cg = 3 + D;                    // logical row width; rows are stored with pitch ldg >= Cg,
    const int row_elems = K * ldg;           // pad columns [Cg, ldg) zero-filled
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int s = s0 + c;
        if (s >= S) break;
        int *lst = mylist + c * lcap;
        const int n = min(cnt[c], K);
        int64_t *orow = idx + ((size_t)b * S + s) * K;
        if (n == 0) {                                       // reference: IndexError at :59
            if (lane == 0 && err_count) atomicAdd(err_count, 1);
            for (int k = lane; k < K; k += PN2_WAVE) { orow[k] = N; lst[k] = -1; }
        } else {
            const int first = lst[0];
            for (int k = lane; k < K; k += PN2_WAVE) {      // :104-106 pad with the first hit
                const int v = k < n ? lst[k] : first;
                lst[k] = v;
                orow[k] = v;
            }
        }
    }
    if (!grouped) return;

    // ---- grouped rows: [xyz[j]-centroid (3), points[j] (D)] for the wave's CPW centroids ----
    const float *bp = points ? points + (size_t)b * N * D : nullptr;
    if (Cg >= 32) {
        // Wide rows: one row per step, lanes along the channel axis (row base is wave-uniform).
#pragma unroll 1
        for (int c = 0; c < CPW; ++c) {
            const int s = s0 + c;
            if (s >= S) break;
            const int *lst = mylist + c * lcap;
            float *g = grouped + ((size_t)b * S + s) * row_elems;
            const float ctr[3] = {cx2[c / 2][c & 1], cy2[c / 2][c & 1], cz2[c / 2][c & 1]};
            // xyz part of all K rows: element e = k*3 + col
            for (int e = lane; e < K * 3; e += PN2_WAVE) {
                const int k = e / 3, col = e - k * 3;
                const int j = lst[k];
                const float cc = col == 0 ? ctr[0] : (col == 1 ? ctr[1] : ctr[2]);
                g[k * ldg + col] = j >= 0 ? bx[(size_t)j * 3 + col] - cc : 0.0f;   // :128
            }
#pragma unroll 4
            for (int k = 0; k < K; ++k) {
                const int j = __builtin_amdgcn_readfirstlane(lst[k]);
                const float *src = bp + (size_t)max(j, 0) * D;
                float *dst = g + (size_t)k * ldg + 3;
                for (int col = lane; col < ldg - 3; col += PN2_WAVE)
                    dst[col] = (j >= 0 && col < D) ? src[col] : 0.0f;           // :131-132 (+ zero pad)
            }
        }
        return;
    }
    if (ldg == Cg && (Cg & 3) == 0 && ((reinterpret_cast<uintptr_t>(grouped) & 15) == 0)) {
        // Narrow rows whose width is a multiple of 4 floats (SA1: 3+9 = 12): one float4 of the
        // output per lane and step; quad 0 of a row is [xyz - centroid, feat0], quad p > 0 is
        // feats[4p-3 .. 4p] (one dword-aligned 16-B load).  16-B aligned coalesced stores.
        const int qpr = Cg >> 2;                                   // quads per row
        const int nq = K * qpr;                                    // quads per centroid
        const unsigned qmagic = (unsigned)((1ULL << 32) / (unsigned)qpr) + 1u;
#pragma unroll 1
        for (int c = 0; c < CPW; ++c) {
            const int s = s0 + c;
            if (s >= S) break;
            const int *lst = mylist + c * lcap;
            float4 *g4 = reinterpret_cast<float4 *>(grouped + ((size_t)b * S + s) * row_elems);
            const float ccx = cx2[c / 2][c & 1], ccy = cy2[c / 2][c & 1], ccz = cz2[c / 2][c & 1];
#pragma unroll 2
            for (int q = lane; q < nq; q += PN2_WAVE) {
                const int k = qpr == 1 ? q : (int)__umulhi((unsigned)q, qmagic);   // q / qpr
                const int part = q - k * qpr;
                const int j = lst[k];
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (j >= 0) {
                    const float *row = bp + (size_t)j * D;
                    if (part == 0) {
                        const float *pj = bx + (size_t)j * 3;
                        v = make_float4(pj[0] - ccx, pj[1] - ccy, pj[2] - ccz, row[0]);   // :128, :131
                    } else {
                        const float *src = row + (4 * part - 3);
                        v = make_float4(src[0], src[1], src[2], src[3]);
                    }
                }
                g4[q] = v;
            }
        }
        return;
    }
    // Narrow rows: elements of one centroid's [K, 3+D] block are walked 64 at a time (256-B
    // coalesced stores); BQ_UNROLL independent gathers are issued before the first is consumed.
#pragma unroll 1
    for (int c = 0; c < CPW; ++c) {
        const int s = s0 + c;
        if (s >= S) break;
        const int *lst = mylist + c * lcap;
        float *g = grouped + ((size_t)b * S + s) * row_elems;
        const float ccx = cx2[c / 2][c & 1], ccy = cy2[c / 2][c & 1], ccz = cz2[c / 2][c & 1];
#pragma unroll 1
        for (int f0 = 0; f0 < row_elems; f0 += PN2_WAVE * BQ_UNROLL) {
            float v[BQ_UNROLL];
#pragma unroll
            for (int u = 0; u < BQ_UNROLL; ++u) {
                const int f = f0 + u * PN2_WAVE + lane;
                v[u] = 0.0f;
                if (f < row_elems) {
                    const int k = (int)__umulhi((unsigned)f, cg_magic);          // f / ldg (exact: host checks the range)
                    const int col = f - k * ldg;
                    const int j = lst[k];
                    if (j >= 0 && col < Cg) {
                        if (col < 3) {
                            const float ctr = col == 0 ? ccx : (col == 1 ? ccy : ccz);
                            v[u] = bx[(size_t)j * 3 + col] - ctr;                // :128
                        } else {
                            v[u] = bp[(size_t)j * D + (col - 3)];                // :131-132
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < BQ_UNROLL; ++u) {
                const int f = f0 + u * PN2_WAVE + lane;
                if (f < row_elems) g[f] = v[u];
            }
        }
    }
}

template <int THREADS, int CPW>
int launch_ball_query_group(const float *xyz, const float *new_xyz, const float *points, int B, int N, int S, int K,
                            int D, int ldg, float r2, int64_t *idx, float *grouped, int32_t *err_count, hipStream_t stream)
{
    constexpr int WAVES = THREADS / PN2_WAVE;
    const int per_wg = WAVES * CPW;
    const int tiles = (S + per_wg - 1) / per_wg;
    const int tile_pts = N < BQ_TILE ? ((N + PN2_WAVE - 1) & ~(PN2_WAVE - 1)) : BQ_TILE;
    const size_t lds = (size_t)tile_pts * sizeof(float4) + (size_t)WAVES * CPW * (K + PN2_WAVE) * sizeof(int);
    const long long nwg = (long long)B * tiles;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    if ((long long)K * ldg + PN2_WAVE * BQ_UNROLL >= (1LL << 32) / ldg) return PN2_ERR_UNSUPPORTED;
    const unsigned magic = (unsigned)((1ULL << 32) / (unsigned)ldg) + 1u;   // umulhi(f, magic) == f / ldg for f*ldg < 2^32
    auto kern = ball_query_group_kernel<THREADS, CPW>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(THREADS), lds, stream, xyz, new_xyz, points, B, N, S, K, D,
                       ldg, r2, tiles, magic, idx, grouped, err_count);
    return PN2_LAUNCH_RC();
}

}  // namespace

// which: 0 = the library's choice; 1 = cell-pruned (pn2_ball_grid.hip), the grid over the block's bounding box; 3 = the same
// with the grid over the box of the block's first 64 centroids (no workgroup-wide reduction); 2 = the vector-unit scan of this file.
// (A matrix-core kernel -- the exact expression as a k-ordered fp32 fma chain on v_mfma_f32_32x32x2_f32 -- was
// withdrawn in round 3: at the one shape the dispatch still gave it, SA2's 1024-point blocks, the scan below is faster
// (7.1 against 12.3 us), and its sign-bit membership test dropped NaN points, which the reference keeps.)  A kernel that does not take the operands
// returns PN2_ERR_UNSUPPORTED (nothing launched) when it was asked for by number.
static int ball_query_group_dispatch(int which, double radius, int nsample, const float *xyz, const float *new_xyz,
                                     const float *points, int B, int N, int S, int D, int64_t *idx, float *grouped, int ldg,
                                     int32_t *err_count, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz);
    PN2_REQUIRE_PTR(new_xyz);
    PN2_REQUIRE_PTR(idx);
    if (B < 0 || N <= 0 || S <= 0 || D < 0 || nsample <= 0) return PN2_ERR_SHAPE;
    if (D > 0 && points == nullptr) return PN2_ERR_NULL;
    if (nsample > 64) return PN2_ERR_UNSUPPORTED;
    if (ldg == 0) ldg = 3 + D;
    if (ldg < 3 + D) return PN2_ERR_SHAPE;
    if (which < 0 || which > 3) return PN2_ERR_UNSUPPORTED;
    if (B == 0) return PN2_OK;
    const float r2 = (float)(radius * radius);          // python `radius ** 2` (double), compared in fp32
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const long long total = (long long)B * S;
    // Many centroids over a large block that fits LDS: cell-pruned candidates (pn2_ball_grid.hip).
    if (which == 1 || which == 3 || (which == 0 && total >= 4096 && N >= pn2::tune_get("bq_grid_minn", 2048) && pn2::tune_get("bq_grid", 1))) {
        // the library's choice: the grid over the box of the block's first centroids (no block-wide reduction: 19.4 against
        // 19.5 us cube, 20.8 against 21.6 us facade, profiles/r04/ballbench_sampled_box.log); which = 1 keeps the round-1 form
        const bool sampled = which == 3 || (which == 0 && pn2::tune_get("bq_grid_sampled", 1));
        const int rc = pn2::launch_ball_query_grid(xyz, new_xyz, points, B, N, S, nsample, D, ldg, r2, idx, grouped,
                                                   err_count, stream, sampled);
        if (rc != PN2_ERR_UNSUPPORTED || which != 0) return rc;
    }
    const int cfg = pn2::tune_get("bq_cfg", total >= 8192 ? 3 : (total >= 2048 ? 1 : 2));
    // Few centroids with wide rows (deep levels): the gather is latency-bound inside the scan
    // waves, so the scan kernel writes idx only and a fully parallel elementwise kernel groups.
    if (grouped && total < 2048 && pn2::tune_get("bq_split", 1)) {
        int rc = ball_query_group_dispatch(2, radius, nsample, xyz, new_xyz, nullptr, B, N, S, 0, idx, nullptr, 0, err_count, stream_);
        if (rc != PN2_OK) return rc;
        return pn2_group_points(xyz, new_xyz, points, idx, B, N, S, nsample, D, grouped, ldg, nullptr, stream_);
    }
#define PN2_BQ(T, C) return launch_ball_query_group<T, C>(xyz, new_xyz, points, B, N, S, nsample, D, ldg, r2, idx, grouped, err_count, stream)
    switch (cfg) {
        case 0: PN2_BQ(512, 8);
        case 1: PN2_BQ(256, 2);
        case 2: PN2_BQ(128, 2);
        case 3: PN2_BQ(512, 4);
        case 4: PN2_BQ(1024, 4);
        case 5: PN2_BQ(256, 8);
        case 6: PN2_BQ(1024, 8);
        case 7: PN2_BQ(256, 4);
        default: return PN2_ERR_UNSUPPORTED;
    }
#undef PN2_BQ
}

PN2_EXPORT int pn2_ball_query_group(double radius, int nsample, const float *xyz, const float *new_xyz,
                                    const float *points, int B, int N, int S, int D, int64_t *idx,
                                    float *grouped, int ldg, int32_t *err_count, pn2_stream_t stream_)
{
    return ball_query_group_dispatch(0, radius, nsample, xyz, new_xyz, points, B, N, S, D, idx, grouped, ldg, err_count, stream_);
}

PN2_EXPORT int pn2_ball_query_group_select(int which, double radius, int nsample, const float *xyz, const float *new_xyz,
                                           const float *points, int B, int N, int S, int D, int64_t *idx,
                                           float *grouped, int ldg, int32_t *err_count, pn2_stream_t stream_)
{
    return ball_query_group_dispatch(which, radius, nsample, xyz, new_xyz, points, B, N, S, D, idx, grouped, ldg, err_count, stream_);
}
