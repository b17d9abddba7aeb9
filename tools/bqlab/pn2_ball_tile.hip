// LAB KERNEL (tools/bqlab/lab4.hip includes it; the library does not build it).
// query_ball_point + grouping in ONE launch, no workspace: every workgroup owns a spatial TILE of its block (gfx950).
//
// Reference: models/pointnet2_utils.py:87-107 (query_ball_point) and :127-132 (the grouping half of
// sample_and_group): per centroid the nsample lowest indices among the points with dist <= r^2, tail padded with
// the first; grouped rows [xyz[idx] - centroid, points[idx]].
//
// The planned kernel (pn2_ball_binned.hip) needs a per-block plan built by launches of its own, which a caller that
// queries once pays in full.  Here nothing is shared between workgroups, so nothing has to be built first:
//
//   1. every workgroup of a block reads the block's coordinates (coalesced 12-byte loads; L2 hits for all but the
//      first) and reduces the SAME bounding box / largest norm (max and min are order-independent, so all
//      workgroups agree bit for bit);
//   2. the box is cut into `tmax` tiles (axis with the widest tile halved until tmax tiles exist); workgroup t owns
//      tile t: the centroids that fall into it, and as CANDIDATES the points of the tile widened by R' (the radius
//      plus the worst-case rounding of the reference's fp32 expression, pn2::bin_cell_width) -- a point the
//      reference accepts for one of the tile's centroids can not lie outside;
//   3. the candidates are compacted into LDS IN INDEX ORDER (wave w holds a contiguous index range, ballots give
//      the ranks), with the rows the grouping will need ([x, y, z, |p|^2] and, for narrow rows, the features);
//   4. a wave takes a centroid and scans the tile's candidates 64 at a time with the reference's exact expression
//      (pn2::pair_sqdist): ballot + mbcnt append the members in ascending index, so "the nsample lowest, padded with
//      the first" holds by construction and the scan stops at nsample members;
//   5. the same wave writes idx (int64) and the grouped rows from LDS with write-through 16-byte stores.
//
// Centroids the tile argument does not cover (non-finite coordinates anywhere in the block; a centroid whose norm
// exceeds the block's largest, i.e. new_xyz is not a subset of xyz) and tiles with more candidates than the LDS
// holds scan every point of the block from memory instead: the same result, slower.  Same indices as the
// reference, bit for bit.
#include <math.h>

#include "pn2_ball_bin.h"

namespace {

constexpr int BT_MAXK = 64;
constexpr int BT_BPW = 8;               // centroids per wave and batch (result rows kept in LDS)
constexpr int BT_CSTASH = 64;           // centroids of a tile whose coordinates are kept in LDS
constexpr int BT_MAXNQ = 4;             // feature rows staged in LDS are at most 4 x 16 bytes (D <= 13 after the xyz part)
constexpr float BT_UPAD = 1.0e-4f;      // absolute slack of a halo, in tile units (rounding of the tile coordinate)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v3i __attribute__((ext_vector_type(3)));

struct BtSmem {
    int red, wtot, misc, clist, ccoord, res, nres, cand4, cidx, frow, total;
};

// LDS layout shared by launcher and kernel.  cap = candidates a tile can stage; dp = floats per staged feature row
// (0: features are not staged); K = nsample.
__host__ __device__ inline BtSmem bt_layout(int threads, int cap, int dp, int K)
{
    const int W = threads / 64;
    BtSmem s;
    int o = 0;
    s.red = o;    o += W * 8 * 4;
    s.wtot = o;   o += ((W * 4 + 15) & ~15);
    s.misc = o;   o += 16;
    s.clist = o;  o += threads * 4 * 2;                  // ushort per centroid of a scan round
    s.ccoord = o; o += BT_CSTASH * 16;
    s.res = o;    o += W * BT_BPW * K * 4;
    s.nres = o;   o += ((W * BT_BPW * 4 + 15) & ~15);
    s.cand4 = o;  o += cap * 16;
    s.cidx = o;   o += cap * 4;
    s.frow = o;   o += ((cap * dp * 4 + 15) & ~15);
    s.total = o;
    return s;
}

// NQ = 16-byte pieces of a feature row staged in LDS (0: rows are not staged)
// PPT = points a thread holds per segment of the block (a multiple of 4: four consecutive points are three 16-byte
// loads); MULTI: the block has more points than the threads hold at once (N > PPT * THREADS)
template <int THREADS, int PPT, int NQ, bool MULTI>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ball_tile_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ points, int N, int S, int K, int D,
    float r2, int tmax, int cap, int dp, unsigned qpr_magic, int64_t *__restrict__ idx, float *__restrict__ grouped,
    int32_t *err_count)
{
    constexpr int W = THREADS / 64;
    constexpr int SEG = THREADS * PPT;
    constexpr int G = PPT / 4;                            // groups of four consecutive points per thread
    static_assert(PPT % 4 == 0 && PPT <= 16, "points per thread");
    constexpr int RS = THREADS * 4;                       // centroids examined per scan round
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BtSmem L = bt_layout(THREADS, cap, dp, K);
    float *red = reinterpret_cast<float *>(smem + L.red);
    unsigned *wtot = reinterpret_cast<unsigned *>(smem + L.wtot);
    unsigned *misc = reinterpret_cast<unsigned *>(smem + L.misc);     // [0] centroids of this tile in the round
    unsigned short *clist = reinterpret_cast<unsigned short *>(smem + L.clist);
    float4 *ccoord = reinterpret_cast<float4 *>(smem + L.ccoord);
    unsigned *res = reinterpret_cast<unsigned *>(smem + L.res);
    unsigned *nres = reinterpret_cast<unsigned *>(smem + L.nres);
    float4 *cand4 = reinterpret_cast<float4 *>(smem + L.cand4);
    unsigned *cidx = reinterpret_cast<unsigned *>(smem + L.cidx);
    float *frow = reinterpret_cast<float *>(smem + L.frow);

    const unsigned logical = pn2::xcd_remap(blockIdx.x, gridDim.x);
    const int b = (int)(logical / (unsigned)tmax);
    const int tile = (int)(logical - (unsigned)b * (unsigned)tmax);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xyz + (size_t)b * N * 3), 0, N * 12, 0x00020000);
    const __amdgpu_buffer_rsrc_t crs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(new_xyz + (size_t)b * S * 3), 0, S * 12, 0x00020000);

    PN2_STAMP(0);
    if (tid == 0) misc[0] = 0u;
    const int nseg = MULTI ? (N + SEG - 1) / SEG : 1;
    float px[PPT], py[PPT], pz[PPT];
    // Wave w of a segment holds the contiguous index range [w*PPT*64, (w+1)*PPT*64); point (g, k) of a lane is index
    // w*PPT*64 + g*256 + lane*4 + k, so (w, g, lane, k) order = index order.  One address register: the loads of a
    // thread differ by an immediate; an index past N reads zeros (or whatever: it is masked where it matters).
    auto load_segment = [&](int seg) {
        const int voff = (seg * SEG + wave * PPT * 64 + lane * 4) * 12;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const v4i a = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, g * 256 * 12, 0);
            const v4i c = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, g * 256 * 12 + 16, 0);
            const v4i d = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, g * 256 * 12 + 32, 0);
            px[4 * g + 0] = __int_as_float(a.x); py[4 * g + 0] = __int_as_float(a.y); pz[4 * g + 0] = __int_as_float(a.z);
            px[4 * g + 1] = __int_as_float(a.w); py[4 * g + 1] = __int_as_float(c.x); pz[4 * g + 1] = __int_as_float(c.y);
            px[4 * g + 2] = __int_as_float(c.z); py[4 * g + 2] = __int_as_float(c.w); pz[4 * g + 2] = __int_as_float(d.x);
            px[4 * g + 3] = __int_as_float(d.y); py[4 * g + 3] = __int_as_float(d.z); pz[4 * g + 3] = __int_as_float(d.w);
        }
    };
    auto index_of = [&](int seg, int i) { return seg * SEG + wave * PPT * 64 + (i >> 2) * 256 + lane * 4 + (i & 3); };
    // the first round's centroids travel with the first segment's loads (one round trip)
    v3i cq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cq[r] = __builtin_amdgcn_raw_buffer_load_b96(crs, tid * 12, r * THREADS * 12, 0);
    load_segment(0);

    // ---- 1. bounding box (as maxima of +-coordinate) and a non-finite flag: identical in every workgroup.  x * 0 is NaN
    //         for an infinite or NaN x and (+-)0 otherwise; max() alone would skip a NaN.
    float q[7] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, 0.0f};
    float nanacc = 0.0f;
    for (int seg = 0; seg < nseg; ++seg) {
        if (seg > 0) load_segment(seg);
        const bool whole = (seg + 1) * SEG <= N;        // uniform: no index of the segment is past N
        if (whole) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const float x = px[i], y = py[i], z = pz[i];
                q[0] = fmaxf(q[0], -x); q[1] = fmaxf(q[1], -y); q[2] = fmaxf(q[2], -z);
                q[3] = fmaxf(q[3], x);  q[4] = fmaxf(q[4], y);  q[5] = fmaxf(q[5], z);
                nanacc = __builtin_fmaf(x, 0.0f, nanacc); nanacc = __builtin_fmaf(y, 0.0f, nanacc); nanacc = __builtin_fmaf(z, 0.0f, nanacc);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const bool ok = index_of(seg, i) < N;
                const float x = ok ? px[i] : px[0], y = ok ? py[i] : py[0], z = ok ? pz[i] : pz[0];
                const bool ok0 = index_of(seg, 0) < N;  // (a lane whose first index is past N holds no point at all)
                q[0] = ok0 ? fmaxf(q[0], -x) : q[0]; q[1] = ok0 ? fmaxf(q[1], -y) : q[1]; q[2] = ok0 ? fmaxf(q[2], -z) : q[2];
                q[3] = ok0 ? fmaxf(q[3], x) : q[3];  q[4] = ok0 ? fmaxf(q[4], y) : q[4];  q[5] = ok0 ? fmaxf(q[5], z) : q[5];
                if (ok0) { nanacc = __builtin_fmaf(x, 0.0f, nanacc); nanacc = __builtin_fmaf(y, 0.0f, nanacc); nanacc = __builtin_fmaf(z, 0.0f, nanacc); }
            }
        }
    }
    q[6] = nanacc == 0.0f ? 0.0f : 1.0f;                // NaN compares unequal
#pragma unroll
    for (int k = 0; k < 7; ++k) q[k] = pn2::bin_wave_max(q[k]);
    PN2_STAMP(1);
    if (W > 1) {
        if (lane < 7) {
            float v = q[0];
#pragma unroll
            for (int k = 1; k < 7; ++k) v = lane == k ? q[k] : v;
            red[wave * 8 + lane] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            float v = red[k];
#pragma unroll
            for (int w = 1; w < W; ++w) v = fmaxf(v, red[w * 8 + k]);
            q[k] = v;
        }
    }
    PN2_STAMP(2);
    const float mn0 = -q[0], mn1 = -q[1], mn2 = -q[2];
    const float ext0 = q[3] - mn0, ext1 = q[4] - mn1, ext2 = q[5] - mn2;
    // an upper bound of every |p|^2 of the block, rounded like the reference rounds a norm (fp32 rounding is monotone)
    const float m2 = pn2::norm3(fmaxf(q[0], q[3]), fmaxf(q[1], q[4]), fmaxf(q[2], q[5]));
    const float Rp = pn2::bin_cell_width(r2, m2);
    const bool finite = q[6] == 0.0f && Rp > 0.0f && Rp < INFINITY && m2 < INFINITY;

    // ---- 2. tiles: halve the axis whose tiles are widest until there are tmax of them (tmax is a power of two)
    int t0 = 1, t1 = 1, t2 = 1;
    float w0 = ext0, w1 = ext1, w2 = ext2;              // tile widths; halving is exact
    if (finite) {
        while (t0 * t1 * t2 < tmax) {
            if (w0 >= w1 && w0 >= w2) {
                if (!(w0 > 0.0f)) break;                // every point of the block is the same point
                t0 *= 2; w0 *= 0.5f;
            } else if (w1 >= w2) {
                t1 *= 2; w1 *= 0.5f;
            } else {
                t2 *= 2; w2 *= 0.5f;
            }
        }
    }
    const int ntile = finite ? t0 * t1 * t2 : tmax;     // a power of two
    if (tile >= ntile) return;                          // uniform for the workgroup
    // this tile's box [olo, ohi) (the outermost tiles reach to infinity: a centroid outside the cloud belongs to the
    // tile next to it, and clamping it onto the box moves it no farther from any point) and the candidates' box
    // [blo, bhi]: widened by R' plus the rounding of the tile borders
    const int tc0 = tile % t0, tc1 = (tile / t0) % t1, tc2 = tile / (t0 * t1);
    const float olo0 = tc0 == 0 ? -INFINITY : mn0 + (float)tc0 * w0, ohi0 = tc0 == t0 - 1 ? INFINITY : mn0 + (float)(tc0 + 1) * w0;
    const float olo1 = tc1 == 0 ? -INFINITY : mn1 + (float)tc1 * w1, ohi1 = tc1 == t1 - 1 ? INFINITY : mn1 + (float)(tc1 + 1) * w1;
    const float olo2 = tc2 == 0 ? -INFINITY : mn2 + (float)tc2 * w2, ohi2 = tc2 == t2 - 1 ? INFINITY : mn2 + (float)(tc2 + 1) * w2;
    const float h0 = Rp + BT_UPAD * w0, h1 = Rp + BT_UPAD * w1, h2 = Rp + BT_UPAD * w2;
    const float blo0 = olo0 - h0, bhi0 = ohi0 + h0, blo1 = olo1 - h1, bhi1 = ohi1 + h1, blo2 = olo2 - h2, bhi2 = ohi2 + h2;

    // ---- 3. candidates of the tile, compacted in index order (done once; centroid rounds below reuse them)
    unsigned ncand_total = 0;                           // uniform
    if (finite) {
        for (int seg = 0; seg < nseg; ++seg) {
            if (nseg > 1) { __syncthreads(); load_segment(seg); }
            unsigned inmask = 0, mine = 0;              // bit i: this lane's point i is a candidate
            const bool whole = (seg + 1) * SEG <= N;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const float x = px[i], y = py[i], z = pz[i];
                bool in = x >= blo0 && x <= bhi0 && y >= blo1 && y <= bhi1 && z >= blo2 && z <= bhi2;
                if (!whole) in = in && index_of(seg, i) < N;
                inmask |= in ? (1u << i) : 0u;
                mine += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
            }
            if (lane == 0) wtot[wave] = mine;
            __syncthreads();
            unsigned pre = ncand_total, all = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const unsigned c = wtot[w];
                pre += w < wave ? c : 0u;
                all += c;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                // rank of point (g, k) of this lane = candidates before the group + those of lower lanes in the group
                // + this lane's own lower k
                unsigned lower = 0, gsum = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64((inmask >> (4 * g + k)) & 1u);
                    lower += (unsigned)pn2::mbcnt(bal);
                    gsum += (unsigned)__builtin_popcountll(bal);
                }
                const unsigned own = (inmask >> (4 * g)) & 15u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = 4 * g + k;
                    const unsigned pos = pre + lower + (unsigned)__builtin_popcount(own & ((1u << k) - 1u));
                    if (((own >> k) & 1u) && pos < (unsigned)cap) {
                        cand4[pos] = make_float4(px[i], py[i], pz[i], 0.0f);        // |p|^2 follows below
                        cidx[pos] = (unsigned)index_of(seg, i);
                    }
                }
                pre += gsum;
            }
            ncand_total += all;
        }
    }
    PN2_STAMP(3);
    const bool overloaded = ncand_total > (unsigned)cap;
    const int ncand = overloaded ? 0 : (int)ncand_total;
    const int Cg = 3 + D;
    const int qpr = Cg >> 2;
    const bool rows = grouped != nullptr;
    const bool staged = rows && NQ > 0 && !overloaded;   // narrow rows: features of the candidates live in LDS
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(points ? points + (size_t)b * N * D : xyz), 0, points ? (int)((unsigned)N * (unsigned)D * 4u) : 0, 0x00020000);
    __syncthreads();
    // the candidates' squared norms, by the threads that will also fetch their features
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int pos = s2 * THREADS + tid;
        if (pos < ncand) {
            const float4 p = cand4[pos];
            reinterpret_cast<float *>(cand4 + pos)[3] = pn2::norm3(p.x, p.y, p.z);
        }
    }
    // per-lane constants of the row phase: lane e, e + 64, ... writes float4 number e of a centroid's K x qpr block
    int ekk[4], epart[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = lane + 64 * u;
        ekk[u] = qpr <= 1 ? e : (int)__umulhi((unsigned)e, qpr_magic);             // e / qpr
        epart[u] = e - ekk[u] * qpr;
    }

    const int nround = (S + RS - 1) / RS;
    bool features_staged = false;
    for (int round = 0; round < nround; ++round) {
        const int cbase = round * RS;
        // ---- 4. the centroids of this tile among centroids cbase .. cbase + RS
        if (round > 0) {
            __syncthreads();                            // everybody is done with the previous round's list
            if (tid == 0) misc[0] = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) cq[r] = __builtin_amdgcn_raw_buffer_load_b96(crs, (cbase + tid) * 12, r * THREADS * 12, 0);
            __syncthreads();
        }
        const bool all_in = cbase + RS <= S;            // uniform
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = cbase + r * THREADS + tid;
            const float cx = __int_as_float(cq[r].x), cy = __int_as_float(cq[r].y), cz = __int_as_float(cq[r].z);
            const float cn = pn2::norm3(cx, cy, cz);
            // outside what R' was sized for (or the block has non-finite coordinates): test every point, like the reference;
            // such centroids are dealt round robin
            const bool full = !finite || !(cn <= m2);
            bool mine = full ? (c & (ntile - 1)) == tile
                             : (cx >= olo0 && cx < ohi0 && cy >= olo1 && cy < ohi1 && cz >= olo2 && cz < ohi2);
            if (!all_in) mine = mine && c < S;
            const unsigned long long bm = __builtin_amdgcn_ballot_w64(mine);
            if (bm != 0ull) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&misc[0], (unsigned)__builtin_popcountll(bm));
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (mine) {
                    const unsigned pos = base + (unsigned)pn2::mbcnt(bm);
                    clist[pos] = (unsigned short)((r * THREADS + tid) | (full ? 0x8000 : 0));
                    if (pos < (unsigned)BT_CSTASH) ccoord[pos] = make_float4(cx, cy, cz, cn);
                }
            }
        }
        __syncthreads();
        const int ncent = (int)misc[0];
        PN2_STAMP(4);
        if (ncent == 0) continue;

        // ---- 5. (first round with centroids) features of the candidates: loads issued now, landed after the first tests
        v4i fq[NQ > 0 ? NQ : 1][2];                      // candidates tid and tid + THREADS (cap <= 2 * THREADS)
        const bool stage_now = staged && !features_staged;
        if (stage_now) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int pos = s2 * THREADS + tid;
                const unsigned o = pos < ncand ? cidx[pos] * (unsigned)D * 4u : 0xfffffff0u;       // out of range reads zeros
#pragma unroll
                for (int u = 0; u < NQ; ++u) fq[u][s2] = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)(o + 16u * (unsigned)u), 0, 0);
            }
        }

        for (int batch0 = 0; batch0 < ncent; batch0 += BT_BPW * W) {
            const int bend = min(ncent, batch0 + BT_BPW * W);
            // ---- 6. tests of this wave's centroids of the batch: members in ascending index into res[slot][0..K)
            for (int k = batch0 + wave; k < bend; k += W) {
                const int slot = wave * BT_BPW + (k - batch0) / W;
                unsigned *myres = res + slot * K;
                const unsigned entry = clist[k];
                float cx, cy, cz, cn;
                if (k < BT_CSTASH) {
                    const float4 cc = ccoord[k];
                    cx = cc.x; cy = cc.y; cz = cc.z; cn = cc.w;
                } else {
                    const v3i cc = __builtin_amdgcn_raw_buffer_load_b96(crs, (cbase + (int)(entry & 0x7fffu)) * 12, 0, 0);
                    cx = __int_as_float(cc.x); cy = __int_as_float(cc.y); cz = __int_as_float(cc.z);
                    cn = pn2::norm3(cx, cy, cz);
                }
                const bool brute = (entry & 0x8000u) != 0u || overloaded;
                int cnt = 0;
                if (!brute) {
                    const int nfull = ncand & ~63;
                    int base = 0;
                    for (; base < nfull && cnt < K; base += 64) {
                        const float4 p = cand4[base + lane];
                        const float d = pn2::pair_sqdist(cx, cy, cz, cn, p.x, p.y, p.z, p.w);
                        const bool hit = !(d > r2);
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)cnt));
                        if (hit && rank < K) myres[rank] = (unsigned)(base + lane);
                        cnt += __builtin_popcountll(m);
                    }
                    if (base == nfull && base < ncand && cnt < K) {
                        const int j = base + lane;
                        const float4 p = cand4[min(j, ncand - 1)];
                        const float d = pn2::pair_sqdist(cx, cy, cz, cn, p.x, p.y, p.z, p.w);
                        const bool hit = j < ncand && !(d > r2);
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)cnt));
                        if (hit && rank < K) myres[rank] = (unsigned)j;
                        cnt += __builtin_popcountll(m);
                    }
                } else {
                    for (int base = 0; base < N && cnt < K; base += 128) {
                        const int j0 = base + lane, j1 = base + 64 + lane;
                        const v3i p0 = __builtin_amdgcn_raw_buffer_load_b96(xrs, min(j0, N - 1) * 12, 0, 0);
                        const v3i p1 = __builtin_amdgcn_raw_buffer_load_b96(xrs, min(j1, N - 1) * 12, 0, 0);
                        const float x0 = __int_as_float(p0.x), y0 = __int_as_float(p0.y), z0 = __int_as_float(p0.z);
                        const float x1 = __int_as_float(p1.x), y1 = __int_as_float(p1.y), z1 = __int_as_float(p1.z);
                        const float d0 = pn2::pair_sqdist(cx, cy, cz, cn, x0, y0, z0, pn2::norm3(x0, y0, z0));
                        const float d1 = pn2::pair_sqdist(cx, cy, cz, cn, x1, y1, z1, pn2::norm3(x1, y1, z1));
                        const bool hit0 = j0 < N && !(d0 > r2), hit1 = j1 < N && !(d1 > r2);
                        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(hit0), m1 = __builtin_amdgcn_ballot_w64(hit1);
                        const int r0 = cnt + pn2::mbcnt(m0);
                        if (hit0 && r0 < K) myres[r0] = (unsigned)j0;
                        cnt += __builtin_popcountll(m0);
                        const int r1 = cnt + pn2::mbcnt(m1);
                        if (hit1 && r1 < K) myres[r1] = (unsigned)j1;
                        cnt += __builtin_popcountll(m1);
                    }
                }
                const int n = min(cnt, K);
                if (lane == 0) {
                    nres[slot] = (unsigned)n | (brute ? 0x8000u : 0u);
                    if (n == 0 && err_count) atomicAdd(err_count, 1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- idx [b, c, 0..K) of the wave's centroids, padded with the first member (:104-106); an empty ball stores
            //      N (IndexError at :59).  Even K: 64 / (K/2) centroids per store instruction.
            const int mine_n = bend > batch0 + wave ? (bend - batch0 - wave + W - 1) / W : 0;      // this wave's centroids in the batch
            if ((K & 1) == 0) {
                const int hk = K >> 1;
                const int per = 64 / hk;                 // centroids per pass (K = 32: 4)
                for (int s0 = 0; s0 < mine_n; s0 += per) {
                    const int sl = s0 + lane / hk, pr = lane % hk;
                    if (lane < per * hk && sl < mine_n) {
                        const int slot = wave * BT_BPW + sl;
                        const unsigned nr = nres[slot];
                        const int n = (int)(nr & 0x7fffu);
                        const bool brute = (nr & 0x8000u) != 0u;
                        const int c = cbase + (int)(clist[batch0 + wave + sl * W] & 0x7fffu);
                        int a0 = N, a1 = N;
                        if (n > 0) {
                            const unsigned *myres = res + slot * K;
                            const unsigned e0 = myres[2 * pr < n ? 2 * pr : 0], e1 = myres[2 * pr + 1 < n ? 2 * pr + 1 : 0];
                            a0 = (int)(brute ? e0 : cidx[e0]);
                            a1 = (int)(brute ? e1 : cidx[e1]);
                        }
                        v4i v;
                        v.x = a0; v.y = 0; v.z = a1; v.w = 0;
                        *reinterpret_cast<v4i *>(idx + ((size_t)b * S + c) * K + 2 * pr) = v;
                    }
                }
            } else {
                for (int sl = 0; sl < mine_n; ++sl) {
                    const int slot = wave * BT_BPW + sl;
                    const unsigned nr = nres[slot];
                    const int n = (int)(nr & 0x7fffu);
                    const bool brute = (nr & 0x8000u) != 0u;
                    const int c = cbase + (int)(clist[batch0 + wave + sl * W] & 0x7fffu);
                    const unsigned *myres = res + slot * K;
                    for (int kk = lane; kk < K; kk += 64) {
                        int a = N;
                        if (n > 0) { const unsigned e = myres[kk < n ? kk : 0]; a = (int)(brute ? e : cidx[e]); }
                        idx[((size_t)b * S + c) * K + kk] = (int64_t)a;
                    }
                }
            }
            PN2_STAMP(5);
            if (!rows) continue;
            // ---- 7. the staged features land in LDS (once)
            if (stage_now && !features_staged) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int pos = s2 * THREADS + tid;
                    if (pos < ncand) {
#pragma unroll
                        for (int u = 0; u < NQ; ++u) {
                            const int f0 = 4 * u;
                            if (f0 + 0 < D) frow[pos * dp + f0 + 0] = __int_as_float(fq[u][s2].x);
                            if (f0 + 1 < D) frow[pos * dp + f0 + 1] = __int_as_float(fq[u][s2].y);
                            if (f0 + 2 < D) frow[pos * dp + f0 + 2] = __int_as_float(fq[u][s2].z);
                            if (f0 + 3 < D) frow[pos * dp + f0 + 3] = __int_as_float(fq[u][s2].w);
                        }
                    }
                }
                features_staged = true;
                __syncthreads();
            }
            PN2_STAMP(6);
            // ---- 8. grouped rows [xyz - centroid, feats] of this wave's centroids: K rows of qpr float4, contiguous
            const int E = K * qpr;
            for (int sl = 0; sl < mine_n; ++sl) {
                const int slot = wave * BT_BPW + sl;
                const int k = batch0 + wave + sl * W;
                const unsigned *myres = res + slot * K;
                const int c = cbase + (int)(clist[k] & 0x7fffu);
                const unsigned nr = nres[slot];
                const int n = (int)(nr & 0x7fffu);
                const bool brute = (nr & 0x8000u) != 0u;
                float cx = 0.0f, cy = 0.0f, cz = 0.0f;
                if (n > 0) {
                    if (k < BT_CSTASH) {
                        const float4 cc = ccoord[k];
                        cx = cc.x; cy = cc.y; cz = cc.z;
                    } else {
                        const v3i cc = __builtin_amdgcn_raw_buffer_load_b96(crs, c * 12, 0, 0);
                        cx = __int_as_float(cc.x); cy = __int_as_float(cc.y); cz = __int_as_float(cc.z);
                    }
                }
                const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(grouped + ((size_t)b * S + c) * K * Cg, 0, E * 16, 0x00020000);
                if (staged && !brute && qpr <= 4) {
                    // gathered from LDS; lane constants hoisted (E <= 256: four passes at most)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = lane + 64 * u;
                        if (e < E) {
                            v4i v;
                            v.x = v.y = v.z = v.w = 0;                                            // an empty ball stores zero rows
                            if (n > 0) {
                                const unsigned m = myres[ekk[u] < n ? ekk[u] : 0];
                                if (epart[u] == 0) {
                                    const float4 p = cand4[m];
                                    v.x = __float_as_int(p.x - cx); v.y = __float_as_int(p.y - cy); v.z = __float_as_int(p.z - cz);   // :128
                                    v.w = __float_as_int(frow[m * dp]);
                                } else {
                                    const float *f = frow + m * dp + 4 * epart[u] - 3;
                                    v.x = __float_as_int(f[0]); v.y = __float_as_int(f[1]); v.z = __float_as_int(f[2]); v.w = __float_as_int(f[3]);
                                }
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(v, grs, e * 16, 0, 16);         // aux 16 = sc1: write-through
                        }
                    }
                } else {
                    for (int e = lane; e < E; e += 64) {
                        const int kk = qpr == 1 ? e : (int)__umulhi((unsigned)e, qpr_magic);      // e / qpr
                        const int part = e - kk * qpr;
                        v4i v;
                        v.x = v.y = v.z = v.w = 0;
                        if (n > 0) {
                            const unsigned m = myres[kk < n ? kk : 0];
                            const unsigned j = brute ? m : cidx[m];
                            float col[4];
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) {
                                const int colno = 4 * part + cc;
                                if (colno < 3) {
                                    const float x = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(j * 12u) + colno * 4, 0, 0));
                                    col[cc] = x - (colno == 0 ? cx : (colno == 1 ? cy : cz));
                                } else {
                                    col[cc] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(prs, (int)((j * (unsigned)D + (unsigned)(colno - 3)) * 4u), 0, 0));
                                }
                            }
                            v.x = __float_as_int(col[0]); v.y = __float_as_int(col[1]); v.z = __float_as_int(col[2]); v.w = __float_as_int(col[3]);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(v, grs, e * 16, 0, 16);
                    }
                }
            }
            PN2_STAMP(7);
        }
    }
    PN2_STAMP_DRAIN();
    PN2_STAMP(8);
}

}  // namespace

namespace pn2 {
// tiles per block: a power of two, about one tile per 16 centroids, at most 64
static int bt_tiles(int S, int per_tile)
{
    int t = 1;
    while (t < 64 && t * 2 * per_tile <= S) t *= 2;
    return t;
}

// rc PN2_ERR_UNSUPPORTED: nothing was launched (the caller falls back to another kernel)
int launch_ball_query_tile(const float *xyz, const float *new_xyz, const float *points, int B, int N, int S, int K, int D, int ldg,
                           float r2, int64_t *idx, float *grouped, int32_t *err_count, hipStream_t stream)
{
    if (K > BT_MAXK || N > 32768 || S > (1 << 24)) return PN2_ERR_UNSUPPORTED;
    const int Cg = 3 + D;
    // rows this kernel writes itself: dense pitch, a whole number of float4 per row, 16-byte aligned
    const bool fused = grouped && ldg == Cg && (Cg & 3) == 0 && (reinterpret_cast<uintptr_t>(grouped) & 15) == 0 &&
                       (long long)K * Cg * 4 < 0x7fffffffLL && (long long)N * (D > 0 ? D : 1) * 4 < 0x7fffffffLL;
    if (grouped && !fused) return PN2_ERR_UNSUPPORTED;
    constexpr int THREADS = 256, PPT = 16;
    const int tmax = bt_tiles(S, pn2::tune_get("bt_per_tile", 16));
    const long long nwg = (long long)B * tmax;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    const int cap = 2 * THREADS;
    int dp = 0;                                                                       // wider rows are gathered from memory
    if (fused && D > 0 && D <= 4 * BT_MAXNQ) dp = (D & 1) ? D : D + 1;                // odd pitch: conflict-free column reads
    const BtSmem L = bt_layout(THREADS, cap, dp, K);
    if (L.total > 64 * 1024) return PN2_ERR_UNSUPPORTED;
    const int qpr = Cg >> 2;
    const unsigned magic = qpr > 1 ? (unsigned)((1ULL << 32) / (unsigned)qpr) + 1u : 0u;
#define PN2_BT(NQ)                                                                                                            \
    do {                                                                                                                      \
        if (N <= THREADS * PPT)                                                                                               \
            hipLaunchKernelGGL((ball_tile_kernel<THREADS, PPT, NQ, false>), dim3((unsigned)nwg), dim3(THREADS), (size_t)L.total, stream, xyz, \
                               new_xyz, points, N, S, K, D, r2, tmax, cap, dp, magic, idx, fused ? grouped : nullptr, err_count);  \
        else                                                                                                                  \
            hipLaunchKernelGGL((ball_tile_kernel<THREADS, PPT, NQ, true>), dim3((unsigned)nwg), dim3(THREADS), (size_t)L.total, stream, xyz, \
                               new_xyz, points, N, S, K, D, r2, tmax, cap, dp, magic, idx, fused ? grouped : nullptr, err_count);  \
    } while (0)
    switch (dp > 0 ? (D + 3) >> 2 : 0) {
        case 0: PN2_BT(0); break;
        case 1: PN2_BT(1); break;
        case 2: PN2_BT(2); break;
        case 3: PN2_BT(3); break;
        default: PN2_BT(4); break;
    }
#undef PN2_BT
    return PN2_LAUNCH_RC();
}
}  // namespace pn2
