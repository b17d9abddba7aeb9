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

constexpr int BQ_THREADS = 512;               // 8 waves
constexpr int BQ_WAVES = BQ_THREADS / PN2_WAVE;
constexpr int BQ_CPW = 4;                     // centroids per wave
constexpr int BQ_TILE = 4096;                 // points staged in LDS per pass (64 KiB as float4)

__global__ __launch_bounds__(BQ_THREADS, 4) void ball_query_group_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ points,
    int B, int N, int S, int K, int D, float r2, int tiles_per_block, int64_t *__restrict__ idx,
    float *__restrict__ grouped, int32_t *err_count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *pts = reinterpret_cast<float4 *>(smem);                                   // [BQ_TILE]
    int *lists = reinterpret_cast<int *>(smem + (size_t)BQ_TILE * sizeof(float4));    // [WAVES][CPW][K]

    const unsigned logical = pn2::xcd_remap(blockIdx.x, gridDim.x);
    const int b = (int)(logical / (unsigned)tiles_per_block);
    const int tile = (int)(logical % (unsigned)tiles_per_block);
    const int tid = threadIdx.x;
    const int lane = tid & (PN2_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / PN2_WAVE);
    const int s0 = (tile * BQ_WAVES + wave) * BQ_CPW;
    int *mylist = lists + (size_t)wave * BQ_CPW * K;

    const float *bx = xyz + (size_t)b * N * 3;
    const float *bc = new_xyz + (size_t)b * S * 3;

    float cx[BQ_CPW], cy[BQ_CPW], cz[BQ_CPW], cn[BQ_CPW];
    int cnt[BQ_CPW];
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) {
        const int s = min(s0 + c, S - 1);
        cx[c] = bc[s * 3 + 0];
        cy[c] = bc[s * 3 + 1];
        cz[c] = bc[s * 3 + 2];
        cn[c] = pn2::norm3(cx[c], cy[c], cz[c]);
        cnt[c] = (s0 + c < S) ? 0 : K;        // centroids past S are "done"
    }

    for (int n0 = 0; n0 < N; n0 += BQ_TILE) {
        if (n0) __syncthreads();
        for (int j = tid; j < BQ_TILE; j += BQ_THREADS) {
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
        const int nchunks = (min(BQ_TILE, N - n0) + PN2_WAVE - 1) / PN2_WAVE;
        for (int ch = 0; ch < nchunks; ++ch) {
            bool all_done = true;
#pragma unroll
            for (int c = 0; c < BQ_CPW; ++c) all_done = all_done && (cnt[c] >= K);
            if (all_done) break;
            const float4 p = pts[ch * PN2_WAVE + lane];
#pragma unroll
            for (int c = 0; c < BQ_CPW; ++c) {
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

    const int Cg = 3 + D;
    const float *bp = points ? points + (size_t)b * N * D : nullptr;
#pragma unroll 1
    for (int c = 0; c < BQ_CPW; ++c) {
        const int s = s0 + c;
        if (s >= S) break;
        const int n = min(cnt[c], K);
        int *lst = mylist + c * K;
        int64_t *orow = idx + ((size_t)b * S + s) * K;
        if (n == 0) {                                       // reference: IndexError at :59
            if (lane == 0 && err_count) atomicAdd(err_count, 1);
            for (int k = lane; k < K; k += PN2_WAVE) orow[k] = N;
            if (grouped) {
                float *g = grouped + ((size_t)b * S + s) * K * Cg;
                for (int f = lane; f < K * Cg; f += PN2_WAVE) g[f] = 0.0f;
            }
            continue;
        }
        const int first = lst[0];
        for (int k = lane; k < K; k += PN2_WAVE) {          // :104-106 pad with the first hit
            const int v = k < n ? lst[k] : first;
            lst[k] = v;
            orow[k] = v;
        }
        if (!grouped) continue;
        float *g = grouped + ((size_t)b * S + s) * K * Cg;
        // element f of the [K, 3+D] row block: k = f / Cg, col = f % Cg, walked incrementally
        int k = lane / Cg, col = lane - k * Cg;
        const int dk = PN2_WAVE / Cg, dcol = PN2_WAVE - dk * Cg;
        for (int f = lane; f < K * Cg; f += PN2_WAVE) {
            const int j = lst[k];
            float v;
            if (col < 3) {
                const float ctr = col == 0 ? cx[c] : (col == 1 ? cy[c] : cz[c]);
                v = bx[(size_t)j * 3 + col] - ctr;                               // :128
            } else {
                v = bp[(size_t)j * D + (col - 3)];                               // :131-132
            }
            g[f] = v;
            k += dk;
            col += dcol;
            if (col >= Cg) { col -= Cg; k += 1; }
        }
    }
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
    const int per_wg = BQ_WAVES * BQ_CPW;
    const int tiles = (S + per_wg - 1) / per_wg;
    const size_t lds = (size_t)BQ_TILE * sizeof(float4) + (size_t)BQ_WAVES * BQ_CPW * nsample * sizeof(int);
    static bool attr_set = false;                       // idempotent; a benign race sets it twice
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ball_query_group_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long long nwg = (long long)B * tiles;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ball_query_group_kernel, dim3((unsigned)nwg), dim3(BQ_THREADS), lds,
                       static_cast<hipStream_t>(stream_), xyz, new_xyz, points, B, N, S, nsample, D, r2, tiles,
                       idx, grouped, err_count);
    return PN2_LAUNCH_RC();
}
