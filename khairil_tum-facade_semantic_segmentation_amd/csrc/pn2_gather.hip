// index_points / grouping gathers, their scatter-add backward, and the materialising
// square_distance, on gfx950.  Reference: models/pointnet2_utils.py:43-60 (advanced indexing),
// :127-132 (grouping), :19-40 (square_distance).  All HBM-bound streaming kernels: lanes run
// along the contiguous channel axis so every wave-instruction touches whole rows.
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#include "pn2_common.h"

namespace {

inline unsigned grid_for(long long total, int threads)
{
    long long blocks = (total + threads - 1) / threads;
    const long long cap = 256LL * 16;
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

__global__ __launch_bounds__(256) void index_points_kernel(const float *__restrict__ points,
                                                           const int64_t *__restrict__ idx, long long total,
                                                           int N, int C, long long M, float *__restrict__ out,
                                                           int32_t *err_count)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long row = t / C;                   // b*M + m
        const int c = (int)(t - row * C);
        const long long b = row / M;
        const int64_t j = idx[row];
        float v = 0.0f;
        if (j >= 0 && j < N) v = points[((size_t)b * N + j) * C + c];
        else if (c == 0 && err_count) atomicAdd(err_count, 1);
        out[t] = v;
    }
}

__global__ __launch_bounds__(256) void index_points_backward_kernel(const float *__restrict__ grad_out,
                                                                    const int64_t *__restrict__ idx,
                                                                    long long total, int N, int D, long long M,
                                                                    int Cg, int col0, float *__restrict__ grad_points)
{
    PN2_MAIN_BRANCH_PRIORITY();
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long row = t / D;
        const int c = (int)(t - row * D);
        const long long b = row / M;
        const int64_t j = idx[row];
        if (j < 0 || j >= N) continue;
        atomicAdd(grad_points + ((size_t)b * N + j) * D + c, grad_out[(size_t)row * Cg + col0 + c]);
    }
}

// dst[r][c] = c < cols_src ? src[r][c] : 0 for c < cols_dst (row pitches lds / ldd): zero-padding of a weight's
// columns to the padded row width of its input, and (cols_dst < cols_src) the slice back for the gradient.
__global__ __launch_bounds__(256) void copy_pad_cols_kernel(const float *__restrict__ src, int lds, int cols_src,
                                                            float *__restrict__ dst, int ldd, int cols_dst, long long total)
{
    PN2_MAIN_BRANCH_PRIORITY();
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long r = t / cols_dst;
    const int c = (int)(t - r * cols_dst);
    dst[r * ldd + c] = c < cols_src ? src[r * lds + c] : 0.0f;
}

// One centroid's [K, 3+D] block per blockIdx.x, 256 consecutive elements per blockIdx.y: all
// index arithmetic is 32-bit (a magic multiply for f / Cg), every gather independent.
__global__ __launch_bounds__(256) void group_points_kernel(const float *__restrict__ xyz,
                                                           const float *__restrict__ new_xyz,
                                                           const float *__restrict__ points,
                                                           const int64_t *__restrict__ idx, int N, int S, int K,
                                                           int D, int ldg, unsigned cg_magic, float *__restrict__ grouped,
                                                           int32_t *err_count)
{
    const int Cg = 3 + D;                              // logical width; stored pitch ldg >= Cg, pad zero-filled
    const int row_elems = K * ldg;
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (f >= row_elems) return;
    const unsigned bs = blockIdx.x;                    // b*S + s
    const unsigned b = bs / (unsigned)S;
    const int k = (int)__umulhi((unsigned)f, cg_magic);
    const int c = f - k * ldg;
    const int64_t j = idx[(size_t)bs * K + k];
    float v = 0.0f;
    if (c >= Cg) {
    } else if (j >= 0 && j < N) {
        if (c < 3) v = xyz[((size_t)b * N + j) * 3 + c] - new_xyz[(size_t)bs * 3 + c];   // :128
        else v = points[((size_t)b * N + j) * D + (c - 3)];                              // :131-132
    } else if (c == 0 && err_count) atomicAdd(err_count, 1);
    grouped[(size_t)bs * row_elems + f] = v;
}

// The same with one float4 of the output row per thread (pitch ldg % 4 == 0, 16-byte aligned output): quad 0 is
// [xyz - centroid, feat 0], quad q > 0 is feats 4q-3 .. 4q (a 16-byte load that is only 4-byte aligned), columns
// past 3+D are the zero pad.  Four times fewer threads, 16-byte stores.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));

__global__ __launch_bounds__(256) void group_points_vec4_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                const float *__restrict__ points, const int64_t *__restrict__ idx,
                                                                int N, int S, int K, int D, int ldg, unsigned q_magic,
                                                                float *__restrict__ grouped, int32_t *err_count)
{
    const int Cg = 3 + D, qpr = ldg >> 2;              // quads per row
    const int quads = K * qpr;
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (f >= quads) return;
    const unsigned bs = blockIdx.x;                    // b*S + s
    const unsigned b = bs / (unsigned)S;
    const int k = qpr == 1 ? f : (int)__umulhi((unsigned)f, q_magic);
    const int q = f - k * qpr;
    const int64_t j = idx[(size_t)bs * K + k];
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (j >= 0 && j < N) {
        const float *row = points ? points + ((size_t)b * N + j) * D : nullptr;
        if (q == 0) {
            const f32x3u p3 = *reinterpret_cast<const f32x3u *>(xyz + ((size_t)b * N + j) * 3);
            const f32x3u c3 = *reinterpret_cast<const f32x3u *>(new_xyz + (size_t)bs * 3);
            v = make_float4(p3.x - c3.x, p3.y - c3.y, p3.z - c3.z, D > 0 ? row[0] : 0.0f);      // :128, :131
        } else {
            const int c0 = 4 * q - 3;                  // first feature of this quad
            if (c0 + 4 <= D) {
                const f32x4u t = *reinterpret_cast<const f32x4u *>(row + c0);
                v = make_float4(t.x, t.y, t.z, t.w);
            } else {                                   // tail quad: some columns are pad
                if (c0 < D) v.x = row[c0];
                if (c0 + 1 < D) v.y = row[c0 + 1];
                if (c0 + 2 < D) v.z = row[c0 + 2];
            }
        }
    } else if (q == 0 && err_count) atomicAdd(err_count, 1);
    (void)Cg;
    reinterpret_cast<float4 *>(grouped + (size_t)bs * K * ldg)[f] = v;
}

// The same quads numbered flat over one batch's [S, K, ldg/4] block (blockIdx.y = batch): no partly filled last
// workgroup per centroid (K*ldg/4 = 544 at SA2 left every third workgroup 7/8 idle), the gathers issued before any
// branch on their result, and 16-byte write-through stores (the block is read next by another kernel, possibly on
// another XCD, and never again by this one).
typedef int gp_v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void group_points_flat_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                const float *__restrict__ points, const int64_t *__restrict__ idx,
                                                                int N, int S, int K, int D, int ldg, unsigned q_magic,
                                                                unsigned k_magic, unsigned batch_quads,
                                                                float *__restrict__ grouped, int32_t *err_count)
{
    PN2_MAIN_BRANCH_PRIORITY();
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= batch_quads) return;
    const unsigned b = blockIdx.y;
    const int qpr = ldg >> 2;
    const unsigned row = qpr == 1 ? t : __umulhi(t, q_magic);      // s*K + k
    const int q = (int)(t - row * (unsigned)qpr);
    const unsigned s = K == 1 ? row : __umulhi(row, k_magic);
    const int64_t j = idx[(size_t)b * S * K + row];
    const bool ok = j >= 0 && j < N;
    const unsigned jj = ok ? (unsigned)j : 0u;
    const float *prow = points + ((size_t)b * N + jj) * D;         // only dereferenced when D > 0
    const int c0 = q == 0 ? 0 : 4 * q - 3;                         // first feature this quad reads
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (c0 + 4 <= D) {                                             // a full quad of features (quad 0 keeps only the first)
        const f32x4u t4 = *reinterpret_cast<const f32x4u *>(prow + c0);
        v = make_float4(t4.x, t4.y, t4.z, t4.w);
    } else {
        if (c0 < D) v.x = prow[c0];
        if (c0 + 1 < D) v.y = prow[c0 + 1];
        if (c0 + 2 < D) v.z = prow[c0 + 2];
    }
    if (q == 0) {
        const f32x3u p3 = *reinterpret_cast<const f32x3u *>(xyz + ((size_t)b * N + jj) * 3);
        const f32x3u c3 = *reinterpret_cast<const f32x3u *>(new_xyz + ((size_t)b * S + s) * 3);
        v = make_float4(p3.x - c3.x, p3.y - c3.y, p3.z - c3.z, v.x);                          // :128, :131
    }
    if (!ok) {
        v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (q == 0 && err_count) atomicAdd(err_count, 1);
    }
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(grouped + (size_t)b * batch_quads * 4, 0,
                                                                         (int)(batch_quads * 16u), 0x00020000);
    gp_v4i w;
    w.x = __float_as_int(v.x); w.y = __float_as_int(v.y); w.z = __float_as_int(v.z); w.w = __float_as_int(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(w, grs, (int)(t * 16u), 0, 16);   // aux 16 = sc1: write-through
}

__global__ __launch_bounds__(256) void square_distance_kernel(const float *__restrict__ src,
                                                              const float *__restrict__ dst, long long total,
                                                              int N, int M, float *__restrict__ out)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long row = t / M;                   // b*N + i
        const int j = (int)(t - row * M);
        const long long b = row / N;
        const float *a = src + (size_t)row * 3;
        const float *q = dst + ((size_t)b * M + j) * 3;
        out[t] = pn2::pair_sqdist(a[0], a[1], a[2], pn2::norm3(a[0], a[1], a[2]), q[0], q[1], q[2],
                                  pn2::norm3(q[0], q[1], q[2]));
    }
}

}  // namespace

extern char **environ;

namespace {
struct TuneTable {
    struct Entry { char key[48]; int value; };
    Entry e[64];
    int n = 0;
    TuneTable()
    {
        for (char **p = environ; p && *p && n < 64; ++p) {
            if (strncmp(*p, "PN2_TUNE_", 9) != 0) continue;
            const char *eq = strchr(*p, '=');
            if (!eq || eq - (*p + 9) >= (long)sizeof(e[0].key)) continue;
            memset(e[n].key, 0, sizeof(e[n].key));
            for (const char *c = *p + 9; c < eq; ++c) e[n].key[c - (*p + 9)] = (char)toupper((unsigned char)*c);
            e[n].value = atoi(eq + 1);
            ++n;
        }
    }
};
}  // namespace

int pn2::tune_get(const char *name, int dflt)
{
    static const TuneTable table;                    // built once, thread-safe (C++11 static initialisation)
    if (table.n == 0) return dflt;
    char key[48] = {0};
    for (int i = 0; name[i] && i < 47; ++i) key[i] = (char)toupper((unsigned char)name[i]);
    for (int i = 0; i < table.n; ++i)
        if (strcmp(table.e[i].key, key) == 0) return table.e[i].value;
    return dflt;
}

PN2_EXPORT int pn2_abi_version(void) { return PN2_ABI_VERSION; }

PN2_EXPORT const char *pn2_error_string(int rc)
{
    switch (rc) {
        case PN2_OK: return "ok";
        case PN2_ERR_NULL: return "required pointer is NULL";
        case PN2_ERR_SHAPE: return "invalid or inconsistent size";
        case PN2_ERR_UNSUPPORTED: return "size outside the supported range";
        default: return rc > 0 ? hipGetErrorString((hipError_t)rc) : "unknown pn2 error";
    }
}

PN2_EXPORT int pn2_index_points(const float *points, const int64_t *idx, int B, int N, int C, int64_t M, float *out,
                                int32_t *err_count, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(points);
    PN2_REQUIRE_PTR(idx);
    PN2_REQUIRE_PTR(out);
    if (B < 0 || N <= 0 || C <= 0 || M < 0) return PN2_ERR_SHAPE;
    const long long total = (long long)B * M * C;
    if (total == 0) return PN2_OK;
    hipLaunchKernelGGL(index_points_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), points, idx, total, N, C, (long long)M, out, err_count);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_index_points_backward(const float *grad_out, const int64_t *idx, int B, int N, int D, int64_t M,
                                         int Cg, int col0, float *grad_points, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(grad_out);
    PN2_REQUIRE_PTR(idx);
    PN2_REQUIRE_PTR(grad_points);
    if (B < 0 || N <= 0 || D <= 0 || M < 0 || col0 < 0 || col0 + D > Cg) return PN2_ERR_SHAPE;
    const long long total = (long long)B * M * D;
    if (total == 0) return PN2_OK;
    hipLaunchKernelGGL(index_points_backward_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), grad_out, idx, total, N, D, (long long)M, Cg, col0,
                       grad_points);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_copy_pad_cols(const float *src, int lds, int cols_src, float *dst, int ldd, int cols_dst, long long rows,
                                 pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(src);
    PN2_REQUIRE_PTR(dst);
    if (rows < 0 || cols_src <= 0 || cols_dst <= 0 || lds < cols_src || ldd < cols_dst) return PN2_ERR_SHAPE;
    const long long total = rows * cols_dst;
    if (total == 0) return PN2_OK;
    if ((total + 255) / 256 > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(copy_pad_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), src,
                       lds, cols_src, dst, ldd, cols_dst, total);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_group_points(const float *xyz, const float *new_xyz, const float *points, const int64_t *idx,
                                int B, int N, int S, int K, int D, float *grouped, int ldg, int32_t *err_count,
                                pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz);
    PN2_REQUIRE_PTR(new_xyz);
    PN2_REQUIRE_PTR(idx);
    PN2_REQUIRE_PTR(grouped);
    if (B < 0 || N <= 0 || S <= 0 || K <= 0 || D < 0) return PN2_ERR_SHAPE;
    if (D > 0 && points == nullptr) return PN2_ERR_NULL;
    if (ldg == 0) ldg = 3 + D;
    if (ldg < 3 + D) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    const int Cg = ldg;
    const long long row_elems = (long long)K * Cg;
    const long long rows = (long long)B * S;
    const long long ny = (row_elems + 255) / 256;
    if (rows > 0x7fffffffLL || ny > 65535 || row_elems + 256 >= (1LL << 32) / Cg) return PN2_ERR_UNSUPPORTED;
    if ((ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(grouped) & 15) == 0 && pn2::tune_get("group_vec4", 1)) {
        const int qpr = ldg >> 2;
        const long long quads = (long long)K * qpr;
        const long long nyq = (quads + 255) / 256;
        const long long batch_quads = (long long)S * quads;
        // flat numbering: magic divisions are exact below 2^32 / divisor, the store offset is 32-bit
        if (B <= 65535 && batch_quads * 16 < (1LL << 31) && batch_quads < (1LL << 32) / (qpr > K ? qpr : K) &&
            pn2::tune_get("group_flat", 1)) {
            const unsigned qm = qpr > 1 ? (unsigned)((1ULL << 32) / (unsigned)qpr) + 1u : 0u;
            const unsigned km = K > 1 ? (unsigned)((1ULL << 32) / (unsigned)K) + 1u : 0u;
            hipLaunchKernelGGL(group_points_flat_kernel, dim3((unsigned)((batch_quads + 255) / 256), (unsigned)B), dim3(256), 0,
                               static_cast<hipStream_t>(stream_), xyz, new_xyz, points, idx, N, S, K, D, ldg, qm, km,
                               (unsigned)batch_quads, grouped, err_count);
            return PN2_LAUNCH_RC();
        }
        if (nyq <= 65535) {
            const unsigned qmagic = qpr > 1 ? (unsigned)((1ULL << 32) / (unsigned)qpr) + 1u : 0u;
            hipLaunchKernelGGL(group_points_vec4_kernel, dim3((unsigned)rows, (unsigned)nyq), dim3(256), 0,
                               static_cast<hipStream_t>(stream_), xyz, new_xyz, points, idx, N, S, K, D, ldg, qmagic, grouped,
                               err_count);
            return PN2_LAUNCH_RC();
        }
    }
    const unsigned magic = (unsigned)((1ULL << 32) / (unsigned)Cg) + 1u;     // umulhi(f, magic) == f / Cg
    hipLaunchKernelGGL(group_points_kernel, dim3((unsigned)rows, (unsigned)ny), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), xyz, new_xyz, points, idx, N, S, K, D, ldg, magic, grouped,
                       err_count);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_square_distance(const float *src, const float *dst, int B, int N, int M, float *out,
                                   pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(src);
    PN2_REQUIRE_PTR(dst);
    PN2_REQUIRE_PTR(out);
    if (B < 0 || N <= 0 || M <= 0) return PN2_ERR_SHAPE;
    const long long total = (long long)B * N * M;
    if (total == 0) return PN2_OK;
    hipLaunchKernelGGL(square_distance_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), src, dst, total, N, M, out);
    return PN2_LAUNCH_RC();
}
