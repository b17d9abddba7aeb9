/*
 * pn2_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Loop-form restatement of the PointNet++ sampling / grouping / interpolation
 * operators of the reference (models/pointnet2_utils.py).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the shipped path is the HIP library behind include/pn2_hip.h.
 *
 * Parity status: PINNED.  The reference holds no tests/golden vectors of its
 * own, so this file is pinned against outputs of the
 * reference itself, produced in the build container by oracle/make_golden.py
 * (which imports /root/reference on CPU) and committed under tests/golden/.
 * tests/test_oracle_golden.py checks every function below against them.
 *
 * Arithmetic rules (probed against torch 2.10 CPU + MKL):
 *   - FPS distance      : ((dx*dx + dy*dy) + dz*dz), every op rounded, no FMA.
 *   - square_distance   : dot = fma(a2,b2, fma(a1,b1, a0*b0));
 *                         d   = ((-2*dot) + |a|^2) + |b|^2, |p|^2 = ((x*x+y*y)+z*z).
 *   - ball membership   : !(d > (float)(r*r computed in double)).
 * Build with -ffp-contract=off so the compiler never fuses on its own.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

static inline float orc_norm3(const float *p)
{
    /* torch.sum(src ** 2, -1): pointnet2_utils.py:38-39 */
    float xx = p[0] * p[0];
    float yy = p[1] * p[1];
    float zz = p[2] * p[2];
    return (xx + yy) + zz;
}

static inline float orc_pair_sqdist(const float *a, float na, const float *b, float nb)
{
    /* pointnet2_utils.py:37-39: -2*matmul, += |src|^2, += |dst|^2 */
    float dot = fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
    float d = -2.0f * dot;
    d = d + na;
    d = d + nb;
    return d;
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* square_distance(src[B,N,3], dst[B,M,3]) -> out[B,N,M]; pointnet2_utils.py:19-40 */
ORC_API void orc_square_distance(const float *src, const float *dst, int B, int N, int M, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < N; ++i) {
            const float *a = src + ((size_t)b * N + i) * 3;
            float na = orc_norm3(a);
            float *row = out + ((size_t)b * N + i) * M;
            for (int j = 0; j < M; ++j) {
                const float *q = dst + ((size_t)b * M + j) * 3;
                row[j] = orc_pair_sqdist(a, na, q, orc_norm3(q));
            }
        }
    }
}

/* farthest_point_sample(xyz[B,N,3], npoint) with injected start indices;
 * pointnet2_utils.py:63-84 (start drawn at :75, loop :77-83). */
ORC_API int orc_farthest_point_sample(const float *xyz, int B, int N, int npoint,
                                      const int64_t *start, int64_t *out_idx)
{
    if (B < 0 || N <= 0 || npoint < 0) return -1;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        float *mind = (float *)malloc((size_t)N * sizeof(float));
        for (int j = 0; j < N; ++j) mind[j] = 1e10f;          /* :74 */
        int64_t far = start[b];
        if (far < 0 || far >= N) { bad += 1; far = 0; }
        for (int i = 0; i < npoint; ++i) {
            out_idx[(size_t)b * npoint + i] = far;             /* :78 */
            float cx = p[far * 3 + 0], cy = p[far * 3 + 1], cz = p[far * 3 + 2];
            float best = -INFINITY;
            int64_t besti = 0;
            for (int j = 0; j < N; ++j) {
                float dx = p[j * 3 + 0] - cx;
                float dy = p[j * 3 + 1] - cy;
                float dz = p[j * 3 + 2] - cz;
                float d = (dx * dx + dy * dy) + dz * dz;       /* :80 */
                if (d < mind[j]) mind[j] = d;                  /* :81-82 */
                if (mind[j] > best) { best = mind[j]; besti = j; } /* :83, ties -> lowest index */
            }
            far = besti;
        }
        free(mind);
    }
    return bad ? -2 : 0;
}

/* query_ball_point(radius, nsample, xyz[B,N,3], new_xyz[B,S,3]) -> idx[B,S,nsample];
 * pointnet2_utils.py:87-107.  The reference builds arange, masks d>r^2 to N, sorts and
 * keeps the first nsample; equivalently: the nsample lowest indices that pass, tail
 * padded with the first hit.  Returns the number of centroids with NO hit (the
 * reference would raise IndexError at :59 for those); their rows are filled with N. */
ORC_API int64_t orc_query_ball_point(double radius, int nsample, const float *xyz, const float *new_xyz,
                                     int B, int N, int S, int64_t *idx)
{
    const float r2 = (float)(radius * radius);                 /* :102 radius ** 2 */
    int64_t empty = 0;
#pragma omp parallel for collapse(2) schedule(static) reduction(+ : empty)
    for (int b = 0; b < B; ++b) {
        for (int s = 0; s < S; ++s) {
            const float *c = new_xyz + ((size_t)b * S + s) * 3;
            float nc = orc_norm3(c);
            int64_t *row = idx + ((size_t)b * S + s) * nsample;
            int cnt = 0;
            for (int j = 0; j < N && cnt < nsample; ++j) {
                const float *q = xyz + ((size_t)b * N + j) * 3;
                float d = orc_pair_sqdist(c, nc, q, orc_norm3(q));   /* :101 (src=new_xyz) */
                if (!(d > r2)) row[cnt++] = j;
            }
            if (cnt == 0) {
                empty += 1;
                for (int k = 0; k < nsample; ++k) row[k] = N;
            } else {
                for (int k = cnt; k < nsample; ++k) row[k] = row[0];  /* :104-106 */
            }
        }
    }
    return empty;
}

/* index_points(points[B,N,C], idx[B,M]) -> out[B,M,C]; pointnet2_utils.py:43-60.
 * Returns -3 if any index is out of range (reference: IndexError). */
ORC_API int orc_index_points(const float *points, const int64_t *idx, int B, int N, int C, int64_t M, float *out)
{
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (int b = 0; b < B; ++b) {
        for (int64_t m = 0; m < M; ++m) {
            int64_t j = idx[(size_t)b * M + m];
            float *o = out + ((size_t)b * M + m) * C;
            if (j < 0 || j >= N) { bad += 1; memset(o, 0, (size_t)C * sizeof(float)); continue; }
            memcpy(o, points + ((size_t)b * N + j) * C, (size_t)C * sizeof(float));
        }
    }
    return bad ? -3 : 0;
}

/* Grouping half of sample_and_group: pointnet2_utils.py:127-132.
 * out[b,s,k,:] = [xyz[idx]-new_xyz[s] (3), points[idx] (D)]; points may be NULL (D=0). */
ORC_API int orc_group_points(const float *xyz, const float *new_xyz, const float *points, const int64_t *idx,
                             int B, int N, int S, int K, int D, float *out)
{
    int bad = 0;
    const int C = 3 + D;
#pragma omp parallel for collapse(2) schedule(static) reduction(+ : bad)
    for (int b = 0; b < B; ++b) {
        for (int s = 0; s < S; ++s) {
            const float *c = new_xyz + ((size_t)b * S + s) * 3;
            for (int k = 0; k < K; ++k) {
                int64_t j = idx[((size_t)b * S + s) * K + k];
                float *o = out + (((size_t)b * S + s) * K + k) * C;
                if (j < 0 || j >= N) { bad += 1; memset(o, 0, (size_t)C * sizeof(float)); continue; }
                const float *q = xyz + ((size_t)b * N + j) * 3;
                o[0] = q[0] - c[0];                               /* :128 */
                o[1] = q[1] - c[1];
                o[2] = q[2] - c[2];
                if (D > 0) memcpy(o + 3, points + ((size_t)b * N + j) * D, (size_t)D * sizeof(float));
            }
        }
    }
    return bad ? -3 : 0;
}

/* Backward of index_points / grouping w.r.t. points: scatter-add of grad rows
 * (autograd of pointnet2_utils.py:59).  grad_out[B,M,Cg] uses columns
 * [col0, col0+D) ; grad_points[B,N,D] must be zeroed by the caller. Serial over M
 * per batch, in index order, so the float sum order is defined. */
ORC_API void orc_index_points_backward(const float *grad_out, const int64_t *idx, int B, int N, int D,
                                       int64_t M, int Cg, int col0, float *grad_points)
{
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int64_t m = 0; m < M; ++m) {
            int64_t j = idx[(size_t)b * M + m];
            if (j < 0 || j >= N) continue;
            const float *g = grad_out + ((size_t)b * M + m) * Cg + col0;
            float *o = grad_points + ((size_t)b * N + j) * D;
            for (int c = 0; c < D; ++c) o[c] += g[c];
        }
    }
}

/* three-NN search + inverse-distance weights of PointNetFeaturePropagation;
 * pointnet2_utils.py:296-302.  dist = square_distance(xyz1, xyz2) (src = xyz1);
 * the reference sorts each row and keeps 3; ties are unpinned in the reference
 * (unstable sort) -- here: lowest index first.  Requires S >= 3. */
ORC_API int orc_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S,
                         int64_t *idx3, float *dist3, float *weight3)
{
    if (S < 3) return -1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < N; ++i) {
            const float *a = xyz1 + ((size_t)b * N + i) * 3;
            float na = orc_norm3(a);
            float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
            int64_t i0 = -1, i1 = -1, i2 = -1;
            for (int j = 0; j < S; ++j) {
                const float *q = xyz2 + ((size_t)b * S + j) * 3;
                float d = orc_pair_sqdist(a, na, q, orc_norm3(q));
                if (d < d0) { d2 = d1; i2 = i1; d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else if (d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = j; }
                else if (d < d2) { d2 = d; i2 = j; }
            }
            size_t o = ((size_t)b * N + i) * 3;
            idx3[o] = i0; idx3[o + 1] = i1; idx3[o + 2] = i2;
            dist3[o] = d0; dist3[o + 1] = d1; dist3[o + 2] = d2;
            float r0 = 1.0f / (d0 + 1e-8f);                     /* :300 */
            float r1 = 1.0f / (d1 + 1e-8f);
            float r2 = 1.0f / (d2 + 1e-8f);
            float nrm = (r0 + r1) + r2;                         /* :301 */
            weight3[o] = r0 / nrm; weight3[o + 1] = r1 / nrm; weight3[o + 2] = r2 / nrm; /* :302 */
        }
    }
    return 0;
}

/* interpolated[b,i,:] = sum_k points2[b, idx3[b,i,k], :] * weight3[b,i,k]; pointnet2_utils.py:303 */
ORC_API void orc_three_interpolate(const float *points2, const int64_t *idx3, const float *weight3,
                                   int B, int N, int S, int D, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < N; ++i) {
            size_t o = ((size_t)b * N + i) * 3;
            const float *p0 = points2 + ((size_t)b * S + idx3[o]) * D;
            const float *p1 = points2 + ((size_t)b * S + idx3[o + 1]) * D;
            const float *p2 = points2 + ((size_t)b * S + idx3[o + 2]) * D;
            float w0 = weight3[o], w1 = weight3[o + 1], w2 = weight3[o + 2];
            float *dst = out + ((size_t)b * N + i) * D;
            for (int c = 0; c < D; ++c) dst[c] = (p0[c] * w0 + p1[c] * w1) + p2[c] * w2;
        }
    }
}

/* Backward of three_interpolate w.r.t. points2 (autograd of :303): grad_points2[B,S,D]
 * (zeroed by caller) += w_k * grad_out[b,i,:] at row idx_k; serial in i per batch. */
ORC_API void orc_three_interpolate_backward(const float *grad_out, const int64_t *idx3, const float *weight3,
                                            int B, int N, int S, int D, float *grad_points2)
{
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < N; ++i) {
            size_t o = ((size_t)b * N + i) * 3;
            const float *g = grad_out + ((size_t)b * N + i) * D;
            for (int k = 0; k < 3; ++k) {
                float *dst = grad_points2 + ((size_t)b * S + idx3[o + k]) * D;
                float w = weight3[o + k];
                for (int c = 0; c < D; ++c) dst[c] += g[c] * w;
            }
        }
    }
}
