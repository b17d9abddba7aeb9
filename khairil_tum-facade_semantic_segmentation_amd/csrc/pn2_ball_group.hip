// query_ball_point fused with the grouping half of sample_and_group, on gfx950.
//
// Reference (models/pointnet2_utils.py:87-107, 127-132) materialises a [B,S,N] distance
// matrix, masks it, fully sorts a [B,S,N] int64 tensor and gathers three times.  Here a
// workgroup stages one block's xyz (+|p|^2) in LDS once, every wave owns CPW centroids and
// scans the points 64 at a time in ascending index order (lane = point, centroid in SGPRs),
// appending hits with ballot + mbcnt so the "nsample lowest indices" rule holds by
// construction and a centroid stops as soon as it has nsample hits.  The same wave then
// writes idx (int64) and the grouped rows [xyz-centroid, feats] with 256-B coalesced stores.
// Bound: HBM (the grouped tensor write); algorithmic bytes in DESIGN.md.
#include <math.h>

#include "pn2_common.h"

namespace {

constexpr int BQ_TILE = 4096;                 // points staged in LDS per pass (64 KiB as float4)
constexpr int BQ_UNROLL = 8;                  // independent gathers in flight per lane in the group phase

// THREADS per workgroup, CPW centroids per wave.
template <int THREADS, int CPW>
__global__ __launch_bounds__(THREADS) void ball_query_group_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ points,
    int B, int N, int S, int K, int D, float r2, int tiles_per_block, unsigned cg_magic,
    int64_t *__restrict__ idx, float *__restrict__ grouped, int32_t *err_count)
{
    constexpr int WAVES = THREADS / PN2_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *pts = reinterpret_cast<float4 *>(smem);                                   // [min(N,BQ_TILE) up to x64]
    const int tile_pts = min(BQ_TILE, (N + PN2_WAVE - 1) & ~(PN2_WAVE - 1));
    int *lists = reinterpret_cast<int *>(smem + (size_t)tile_pts * sizeof(float4));   // [WAVES][CPW][K]

    const unsigned logical = pn2::xcd_remap(blockIdx.x, gridDim.x);
    const int b = (int)(logical / (unsigned)tiles_per_block);
    const int tile = (int)(logical % (unsigned)tiles_per_block);
    const int tid = threadIdx.x;
    const int lane = tid & (PN2_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / PN2_WAVE);
    const int s0 = (tile * WAVES + wave) * CPW;
    int *mylist = lists + (size_t)wave * CPW * K;

    const float *bx = xyz + (size_t)b * N * 3;
    const float *bc = new_xyz + (size_t)b * S * 3;

    float cx[CPW], cy[CPW], cz[CPW], cn[CPW];
    int cnt[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int s = min(s0 + c, S - 1);
        cx[c] = bc[s * 3 + 0];
        cy[c] = bc[s * 3 + 1];
        cz[c] = bc[s * 3 + 2];
        cn[c] = pn2::norm3(cx[c], cy[c], cz[c]);
        cnt[c] = (s0 + c < S) ? 0 : K;        // centroids past S are "done"
    }

    for (int n0 = 0; n0 < N; n0 += BQ_TILE) {
        if (n0) __syncthreads();
        const int npad = min(tile_pts, (N - n0 + PN2_WAVE - 1) & ~(PN2_WAVE - 1));
        for (int j = tid; j < npad; j += THREADS) {
            const int g = n0 + j;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, INFINITY);   // padding: d = +inf, never a hit
            if (g < N) {
                v.x = bx[g * 3 + 0];
                v.y = bx[g * 3 + 1];
                v.z = bx[g * 3 + 2];
                v.w = pn2::norm3(v.x, v.y, v.z);
            }
            pts[j] = v;
        }
        __syncthreads();
        const int nchunks = npad / PN2_WAVE;
        for (int ch = 0; ch < nchunks; ++ch) {
            bool all_done = true;
#pragma unroll
            for (int c = 0; c < CPW; ++c) all_done = all_done && (cnt[c] >= K);
            if (all_done) break;
            const float4 p = pts[ch * PN2_WAVE + lane];
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                if (cnt[c] < K) {
                    // src = new_xyz (centroid), dst = xyz (point): pointnet2_utils.py:101
                    const float d = pn2::pair_sqdist(cx[c], cy[c], cz[c], cn[c], p.x, p.y, p.z, p.w);
                    const bool hit = !(d > r2);                                  // :102
                    const unsigned long long m = __ballot(hit);
                    if (m) {
                        const int pos = cnt[c] + pn2::mbcnt(m);
                        if (hit && pos < K) mylist[c * K + pos] = n0 + ch * PN2_WAVE + lane;
                        cnt[c] += __builtin_popcountll(m);
                    }
                }
            }
        }
    }

    // ---- pad + idx (int64) -------------------------------------------------------------
    const int Cg = 3 + D;
    const int row_elems = K * Cg;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int s = s0 + c;
        if (s >= S) break;
        const int n = min(cnt[c], K);
        int *lst = mylist + c * K;
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
    // Elements of one centroid's [K, 3+D] block are walked 64 at a time (256-B coalesced
    // stores); BQ_UNROLL independent gathers are issued before the first one is consumed.
    const float *bp = points ? points + (size_t)b * N * D : nullptr;
#pragma unroll 1
    for (int c = 0; c < CPW; ++c) {
        const int s = s0 + c;
        if (s >= S) break;
        const int *lst = mylist + c * K;
        float *g = grouped + ((size_t)b * S + s) * row_elems;
        const float ccx = cx[c], ccy = cy[c], ccz = cz[c];
#pragma unroll 1
        for (int f0 = 0; f0 < row_elems; f0 += PN2_WAVE * BQ_UNROLL) {
            float v[BQ_UNROLL];
#pragma unroll
            for (int u = 0; u < BQ_UNROLL; ++u) {
                const int f = f0 + u * PN2_WAVE + lane;
                v[u] = 0.0f;
                if (f < row_elems) {
                    const int k = (int)__umulhi((unsigned)f, cg_magic);          // f / Cg (exact: host checks the range)
                    const int col = f - k * Cg;
                    const int j = lst[k];
                    if (j >= 0) {
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
                            int D, float r2, int64_t *idx, float *grouped, int32_t *err_count, hipStream_t stream)
{
    constexpr int WAVES = THREADS / PN2_WAVE;
    const int per_wg = WAVES * CPW;
    const int tiles = (S + per_wg - 1) / per_wg;
    const int tile_pts = N < BQ_TILE ? ((N + PN2_WAVE - 1) & ~(PN2_WAVE - 1)) : BQ_TILE;
    const size_t lds = (size_t)tile_pts * sizeof(float4) + (size_t)WAVES * CPW * K * sizeof(int);
    const long long nwg = (long long)B * tiles;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    const int Cg = 3 + D;
    if ((long long)K * Cg + PN2_WAVE * BQ_UNROLL >= (1LL << 32) / Cg) return PN2_ERR_UNSUPPORTED;
    const unsigned magic = (unsigned)((1ULL << 32) / (unsigned)Cg) + 1u;   // umulhi(f, magic) == f / Cg for f*Cg < 2^32
    auto kern = ball_query_group_kernel<THREADS, CPW>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(THREADS), lds, stream, xyz, new_xyz, points, B, N, S, K, D,
                       r2, tiles, magic, idx, grouped, err_count);
    return PN2_LAUNCH_RC();
}

}  // namespace

PN2_EXPORT int pn2_ball_query_group(double radius, int nsample, const float *xyz, const float *new_xyz,
                                    const float *points, int B, int N, int S, int D, int64_t *idx,
                                    float *grouped, int32_t *err_count, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz);
    PN2_REQUIRE_PTR(new_xyz);
    PN2_REQUIRE_PTR(idx);
    if (B < 0 || N <= 0 || S <= 0 || D < 0 || nsample <= 0) return PN2_ERR_SHAPE;
    if (D > 0 && points == nullptr) return PN2_ERR_NULL;
    if (nsample > 64) return PN2_ERR_UNSUPPORTED;
    if (B == 0) return PN2_OK;
    const float r2 = (float)(radius * radius);          // python `radius ** 2` (double), compared in fp32
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // Many centroids per block: 8 waves x 4 centroids share one LDS image of the block.
    // Few centroids (deep levels): one centroid per wave, small workgroups, for parallelism.
    const int cfg = pn2::tune_get("bq_cfg", (long long)B * S >= 8192 ? 0 : 1);
    switch (cfg) {
        case 0: return launch_ball_query_group<512, 4>(xyz, new_xyz, points, B, N, S, nsample, D, r2, idx, grouped, err_count, stream);
        case 1: return launch_ball_query_group<256, 1>(xyz, new_xyz, points, B, N, S, nsample, D, r2, idx, grouped, err_count, stream);
        case 2: return launch_ball_query_group<512, 2>(xyz, new_xyz, points, B, N, S, nsample, D, r2, idx, grouped, err_count, stream);
        case 3: return launch_ball_query_group<1024, 4>(xyz, new_xyz, points, B, N, S, nsample, D, r2, idx, grouped, err_count, stream);
        case 4: return launch_ball_query_group<256, 4>(xyz, new_xyz, points, B, N, S, nsample, D, r2, idx, grouped, err_count, stream);
        case 5: return launch_ball_query_group<512, 8>(xyz, new_xyz, points, B, N, S, nsample, D, r2, idx, grouped, err_count, stream);
        default: return PN2_ERR_UNSUPPORTED;
    }
}
