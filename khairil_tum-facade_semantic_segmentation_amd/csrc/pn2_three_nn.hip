// three-NN search + inverse-distance weights, interpolation and its backward, on gfx950.
// Reference: PointNetFeaturePropagation.forward, models/pointnet2_utils.py:296-303 -- a full
// [B,N,S] distance matrix plus a full sort per row to pick 3 entries.  Here each lane owns one
// query point, the S sources are staged once in LDS (xyz + |p|^2) and streamed past it; the
// S range is split over the 4 waves of a workgroup and the partial top-3 lists are merged in
// range order, so exact ties resolve to the lowest index.
#include <math.h>

#include "pn2_common.h"

namespace {

constexpr int NN_THREADS = 256;
constexpr int NN_WAVES = NN_THREADS / PN2_WAVE;     // S split
constexpr int NN_TILE = 2048;                       // sources per LDS pass (32 KiB)

struct Top3 {
    float d0, d1, d2;
    int i0, i1, i2;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// pn2::pair_sqdist for two sources at once (v_pk_mul / v_pk_fma / v_pk_add: each half is the scalar chain, bit for bit)
__device__ __forceinline__ f32x2 pair_sqdist2(f32x2 ax, f32x2 ay, f32x2 az, f32x2 na, f32x2 minus2, f32x2 bx, f32x2 by, f32x2 bz,
                                             f32x2 nb)
{
    const f32x2 dot = __builtin_elementwise_fma(az, bz, __builtin_elementwise_fma(ay, by, ax * bx));
    const f32x2 d = __builtin_elementwise_fma(minus2, dot, na);
    return d + nb;
}

__device__ __forceinline__ void top3_insert(Top3 &t, float d, int j)
{
    // strict '<': an equal distance never displaces an earlier (lower) index
    if (d < t.d2) {
        if (d < t.d1) {
            t.d2 = t.d1; t.i2 = t.i1;
            if (d < t.d0) { t.d1 = t.d0; t.i1 = t.i0; t.d0 = d; t.i0 = j; }
            else { t.d1 = d; t.i1 = j; }
        } else { t.d2 = d; t.i2 = j; }
    }
}

__device__ __forceinline__ void three_nn_body(
    unsigned block, unsigned nblocks, const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int S, int qtiles,
    int64_t *__restrict__ idx3, float *__restrict__ dist3, float *__restrict__ weight3)
{
    // sources as four arrays (x, y, z, |p|^2): four consecutive sources are one 16-byte broadcast read per array, and two
    // of them fill the halves of a packed-fp32 operand
    __shared__ __attribute__((aligned(16))) float srcx[NN_TILE], srcy[NN_TILE], srcz[NN_TILE], srcn[NN_TILE];
    __shared__ float md[NN_WAVES][3][PN2_WAVE];
    __shared__ int mi[NN_WAVES][3][PN2_WAVE];

    const unsigned logical = pn2::xcd_remap(block, nblocks);
    const int b = (int)(logical / (unsigned)qtiles);
    const int qt = (int)(logical % (unsigned)qtiles);
    const int tid = threadIdx.x;
    const int lane = tid & (PN2_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / PN2_WAVE);
    const int q = qt * PN2_WAVE + lane;                    // query handled by this lane (all 4 waves)
    const float *b1 = xyz1 + (size_t)b * N * 3;
    const float *b2 = xyz2 + (size_t)b * S * 3;

    const int qq = min(q, N - 1);
    const float ax = b1[qq * 3 + 0], ay = b1[qq * 3 + 1], az = b1[qq * 3 + 2];
    const float na = pn2::norm3(ax, ay, az);
    Top3 t = {INFINITY, INFINITY, INFINITY, 0, 0, 0};

    for (int s0 = 0; s0 < S; s0 += NN_TILE) {
        if (s0) __syncthreads();
        const int ns = min(NN_TILE, S - s0);
        {   // every source of the tile is loaded before the first is stored (one memory round trip, not one per pass)
            constexpr int PASSES = NN_TILE / NN_THREADS;
            float sx[PASSES], sy[PASSES], sz[PASSES];
#pragma unroll
            for (int u = 0; u < PASSES; ++u) {
                const size_t o = (size_t)(s0 + min(tid + u * NN_THREADS, ns - 1)) * 3;
                sx[u] = b2[o]; sy[u] = b2[o + 1]; sz[u] = b2[o + 2];
            }
#pragma unroll
            for (int u = 0; u < PASSES; ++u) {
                const int j = tid + u * NN_THREADS;
                if (j < ns) { srcx[j] = sx[u]; srcy[j] = sy[u]; srcz[j] = sz[u]; srcn[j] = pn2::norm3(sx[u], sy[u], sz[u]); }
            }
        }
        __syncthreads();
        // wave w scans the w-th contiguous quarter of this tile (quarters start on multiples of four sources).  Four sources
        // per pass: four broadcast reads, the four distances as two packed-fp32 chains (each component is pair_sqdist's own
        // k-ordered fma chain: the same bits), ONE test of the smallest against the current third best -- v_min ignores a
        // NaN as the insertion does -- and the (rare) insertions in index order.
        const int per = ((ns + NN_WAVES - 1) / NN_WAVES + 3) & ~3;
        const int jb = min(ns, wave * per), je = min(ns, jb + per);
        const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, na2 = {na, na}, m2 = {-2.0f, -2.0f};
        const float4 *x4 = reinterpret_cast<const float4 *>(srcx), *y4 = reinterpret_cast<const float4 *>(srcy);
        const float4 *z4 = reinterpret_cast<const float4 *>(srcz), *n4 = reinterpret_cast<const float4 *>(srcn);
        int j = jb;
        for (; j + 4 <= je; j += 4) {
            const int q4 = j >> 2;                                               // jb is a multiple of four
            const float4 px = x4[q4], py = y4[q4], pz = z4[q4], pn = n4[q4];     // LDS broadcasts
            // src = xyz1 (query), dst = xyz2 (source): pointnet2_utils.py:296
            const f32x2 dlo = pair_sqdist2(ax2, ay2, az2, na2, m2, f32x2{px.x, px.y}, f32x2{py.x, py.y}, f32x2{pz.x, pz.y}, f32x2{pn.x, pn.y});
            const f32x2 dhi = pair_sqdist2(ax2, ay2, az2, na2, m2, f32x2{px.z, px.w}, f32x2{py.z, py.w}, f32x2{pz.z, pz.w}, f32x2{pn.z, pn.w});
            const float least = fminf(fminf(dlo.x, dlo.y), fminf(dhi.x, dhi.y));
            if (__ballot(least < t.d2)) {
                top3_insert(t, dlo.x, s0 + j);
                top3_insert(t, dlo.y, s0 + j + 1);
                top3_insert(t, dhi.x, s0 + j + 2);
                top3_insert(t, dhi.y, s0 + j + 3);
            }
        }
        for (; j < je; ++j) {
            const float d = pn2::pair_sqdist(ax, ay, az, na, srcx[j], srcy[j], srcz[j], srcn[j]);
            if (__ballot(d < t.d2)) top3_insert(t, d, s0 + j);
        }
    }
    md[wave][0][lane] = t.d0; md[wave][1][lane] = t.d1; md[wave][2][lane] = t.d2;
    mi[wave][0][lane] = t.i0; mi[wave][1][lane] = t.i1; mi[wave][2][lane] = t.i2;
    __syncthreads();
    if (wave != 0 || q >= N) return;
    // Merge in ascending source-range order.  With several LDS tiles the ranges interleave
    // (wave w holds quarter w of every tile), so order candidates by (distance, index).
    Top3 r = {INFINITY, INFINITY, INFINITY, 0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll
    for (int w = 0; w < NN_WAVES; ++w) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = md[w][k][lane];
            const int j = mi[w][k][lane];
            const bool lt2 = d < r.d2 || (d == r.d2 && j < r.i2);
            if (lt2 && d < INFINITY) {
                const bool lt1 = d < r.d1 || (d == r.d1 && j < r.i1);
                const bool lt0 = d < r.d0 || (d == r.d0 && j < r.i0);
                if (lt1) {
                    r.d2 = r.d1; r.i2 = r.i1;
                    if (lt0) { r.d1 = r.d0; r.i1 = r.i0; r.d0 = d; r.i0 = j; }
                    else { r.d1 = d; r.i1 = j; }
                } else { r.d2 = d; r.i2 = j; }
            }
        }
    }
    const size_t o = ((size_t)b * N + q) * 3;
    idx3[o] = r.i0; idx3[o + 1] = r.i1; idx3[o + 2] = r.i2;
    if (dist3) { dist3[o] = r.d0; dist3[o + 1] = r.d1; dist3[o + 2] = r.d2; }
    const float r0 = 1.0f / (r.d0 + 1e-8f);                                      // :300
    const float r1 = 1.0f / (r.d1 + 1e-8f);
    const float r2 = 1.0f / (r.d2 + 1e-8f);
    const float nrm = (r0 + r1) + r2;                                            // :301
    weight3[o] = r0 / nrm; weight3[o + 1] = r1 / nrm; weight3[o + 2] = r2 / nrm;  // :302
}

__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int S, int qtiles,
    int64_t *__restrict__ idx3, float *__restrict__ dist3, float *__restrict__ weight3)
{
    three_nn_body(blockIdx.x, gridDim.x, xyz1, xyz2, N, S, qtiles, idx3, dist3, weight3);
}

// Several (queries, sources) pairs in one launch: the four interpolation levels of the network need nothing but the four
// levels' coordinates, and their launches in a row only add up (5 + 6 + 13 + 36 us).  Jobs by value, workgroup ranges per job.
constexpr int NN_MANY_MAX = 8;
struct ThreeNnMany {
    int n;
    unsigned first[NN_MANY_MAX + 1];
    const float *xyz1[NN_MANY_MAX], *xyz2[NN_MANY_MAX];
    int64_t *idx3[NN_MANY_MAX];
    float *weight3[NN_MANY_MAX];
    int N[NN_MANY_MAX], S[NN_MANY_MAX], qtiles[NN_MANY_MAX];
};
__global__ __launch_bounds__(NN_THREADS) void three_nn_many_kernel(ThreeNnMany m)
{
    int j = 0;
    while (j + 1 < m.n && blockIdx.x >= m.first[j + 1]) ++j;
    three_nn_body(blockIdx.x - m.first[j], m.first[j + 1] - m.first[j], m.xyz1[j], m.xyz2[j], m.N[j], m.S[j], m.qtiles[j], m.idx3[j],
                  nullptr, m.weight3[j]);
}

// out[b,i,:] = (p0*w0 + p1*w1) + p2*w2, one thread per VEC consecutive channels.
template <int VEC>
__global__ __launch_bounds__(256) void three_interpolate_kernel(
    const float *__restrict__ points2, const int64_t *__restrict__ idx3, const float *__restrict__ weight3,
    long long total, int N, int S, int D, float *__restrict__ out)
{
    PN2_MAIN_BRANCH_PRIORITY();
    const int dv = D / VEC;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long row = t / dv;                  // b*N + i
        const int c = (int)(t - row * dv) * VEC;
        const long long b = row / N;
        const int64_t j0 = idx3[row * 3], j1 = idx3[row * 3 + 1], j2 = idx3[row * 3 + 2];
        const float w0 = weight3[row * 3], w1 = weight3[row * 3 + 1], w2 = weight3[row * 3 + 2];
        const float *p0 = points2 + ((size_t)b * S + j0) * D + c;
        const float *p1 = points2 + ((size_t)b * S + j1) * D + c;
        const float *p2 = points2 + ((size_t)b * S + j2) * D + c;
        float *o = out + (size_t)row * D + c;
        if (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4 *>(p0);
            const float4 bb = *reinterpret_cast<const float4 *>(p1);
            const float4 cc = *reinterpret_cast<const float4 *>(p2);
            float4 r;
            r.x = (a.x * w0 + bb.x * w1) + cc.x * w2;
            r.y = (a.y * w0 + bb.y * w1) + cc.y * w2;
            r.z = (a.z * w0 + bb.z * w1) + cc.z * w2;
            r.w = (a.w * w0 + bb.w * w1) + cc.w * w2;
            pn2::store_rows4(out, (size_t)row * D + c, r, (size_t)total * VEC * sizeof(float));   // read next by a GEMM
        } else {
            o[0] = (p0[0] * w0 + p1[0] * w1) + p2[0] * w2;
        }
    }
}

// grad_points2[b, idx3[b,i,k], c] += w_k * grad_out[b,i,c]; lanes run along c so that each
// atomic wave-instruction covers contiguous bytes of one row (MI355X_MICROARCH float atomics).
__global__ __launch_bounds__(256) void three_interpolate_backward_kernel(
    const float *__restrict__ grad_out, const int64_t *__restrict__ idx3, const float *__restrict__ weight3,
    long long total, int N, int S, int D, float *__restrict__ grad_points2)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long row = t / D;
        const int c = (int)(t - row * D);
        const long long b = row / N;
        const float g = grad_out[t];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int64_t j = idx3[row * 3 + k];
            atomicAdd(grad_points2 + ((size_t)b * S + j) * D + c, g * weight3[row * 3 + k]);
        }
    }
}

inline unsigned grid_for(long long total, int threads)
{
    long long blocks = (total + threads - 1) / threads;
    const long long cap = 256LL * 16;                  // grid-stride beyond 16 blocks per CU
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

}  // namespace

PN2_EXPORT int pn2_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int64_t *idx3,
                            float *dist3, float *weight3, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz1);
    PN2_REQUIRE_PTR(xyz2);
    PN2_REQUIRE_PTR(idx3);
    PN2_REQUIRE_PTR(weight3);
    if (B < 0 || N <= 0 || S < 3) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    const int qtiles = (N + PN2_WAVE - 1) / PN2_WAVE;
    const long long nwg = (long long)B * qtiles;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(three_nn_kernel, dim3((unsigned)nwg), dim3(NN_THREADS), 0,
                       static_cast<hipStream_t>(stream_), xyz1, xyz2, N, S, qtiles, idx3, dist3, weight3);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_three_nn_many(int n, const float *const *xyz1, const float *const *xyz2, int B, const int *N, const int *S,
                                 int64_t *const *idx3, float *const *weight3, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz1); PN2_REQUIRE_PTR(xyz2); PN2_REQUIRE_PTR(N); PN2_REQUIRE_PTR(S); PN2_REQUIRE_PTR(idx3); PN2_REQUIRE_PTR(weight3);
    if (n <= 0 || n > NN_MANY_MAX || B < 0) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    ThreeNnMany m;
    m.n = n;
    long long blocks = 0;
    for (int j = 0; j < NN_MANY_MAX; ++j) {
        const int i = j < n ? j : n - 1;
        if (!xyz1[i] || !xyz2[i] || !idx3[i] || !weight3[i]) return PN2_ERR_NULL;
        if (N[i] <= 0 || S[i] < 3) return PN2_ERR_SHAPE;
        m.xyz1[j] = xyz1[i]; m.xyz2[j] = xyz2[i]; m.idx3[j] = idx3[i]; m.weight3[j] = weight3[i];
        m.N[j] = N[i]; m.S[j] = S[i]; m.qtiles[j] = (N[i] + PN2_WAVE - 1) / PN2_WAVE;
        m.first[j] = (unsigned)blocks;
        if (j < n) blocks += (long long)B * m.qtiles[j];
        if (blocks > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    }
    m.first[NN_MANY_MAX] = (unsigned)blocks;
    for (int j = n; j < NN_MANY_MAX; ++j) m.first[j] = (unsigned)blocks;
    hipLaunchKernelGGL(three_nn_many_kernel, dim3((unsigned)blocks), dim3(NN_THREADS), 0, static_cast<hipStream_t>(stream_), m);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_three_interpolate(const float *points2, const int64_t *idx3, const float *weight3, int B,
                                     int N, int S, int D, float *out, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(points2);
    PN2_REQUIRE_PTR(idx3);
    PN2_REQUIRE_PTR(weight3);
    PN2_REQUIRE_PTR(out);
    if (B < 0 || N <= 0 || S <= 0 || D <= 0) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(points2) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    if (vec4) {
        const long long total = (long long)B * N * (D / 4);
        hipLaunchKernelGGL(three_interpolate_kernel<4>, dim3(grid_for(total, 256)), dim3(256), 0, stream, points2,
                           idx3, weight3, total, N, S, D, out);
    } else {
        const long long total = (long long)B * N * D;
        hipLaunchKernelGGL(three_interpolate_kernel<1>, dim3(grid_for(total, 256)), dim3(256), 0, stream, points2,
                           idx3, weight3, total, N, S, D, out);
    }
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_three_interpolate_backward(const float *grad_out, const int64_t *idx3, const float *weight3,
                                              int B, int N, int S, int D, float *grad_points2,
                                              pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(grad_out);
    PN2_REQUIRE_PTR(idx3);
    PN2_REQUIRE_PTR(weight3);
    PN2_REQUIRE_PTR(grad_points2);
    if (B < 0 || N <= 0 || S <= 0 || D <= 0) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    const long long total = (long long)B * N * D;
    hipLaunchKernelGGL(three_interpolate_backward_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), grad_out, idx3, weight3, total, N, S, D, grad_points2);
    return PN2_LAUNCH_RC();
}
