// Query plan of one block of points for the binned ball query (gfx950).
//
// query_ball_point (models/pointnet2_utils.py:87-107) keeps, per centroid, the nsample lowest
// indices among the points with dist <= r^2.  Those points lie within r of the centroid, so a uniform
// grid with cells at least R' wide (R' = r plus the worst-case rounding of the reference's fp32
// distance expression) confines them to the 27 cells around the centroid's cell.  Everything that
// depends only on the block's geometry is prepared ONCE per block -- by the tail of the
// farthest-point-sampling kernel, which holds the block in registers and has just produced the
// centroids, or by the stand-alone kernel in pn2_ball_binned.hip -- and read by the query kernel's
// workgroups from L2:
//
//   byte 0     BinHeader (64 B; bookkeeping, the query kernel does not read it)
//   byte 128   plan    [S][16 words]             per centroid: words 0..8 = the nine runs of the cell-sorted array that
//                                                hold its candidates (the cells (cx-1..cx+1, cy+dy, cz+dz) are one
//                                                contiguous run), as first slot | (one past the last) << 16, 0 for a
//                                                row outside the grid; word 9 = 1 when the centroid must test every
//                                                point (non-finite coordinates, or a norm the cell width was not sized for)
//   then       sorted  float4 [Npad + 32]        (x, y, z, original index as int bits), cells in x-fastest order; the
//                                                entries from N on lie far away (never members): the query reads up to
//                                                31 slots past a run without masking.  Npad = N rounded up to 32
//   then       rows    [N][RP bytes]             index order: the grouped row of point j before the centroid is
//                                                subtracted, [x, y, z, feats(D)], padded to RP = the power of two >=
//                                                4*(3+D) (multiples of 128 beyond that), so that a gathered row is one
//                                                cache line, not the three of separate xyz / feature rows
//
// bin_block<T, P> is the device routine shared by both producers: T threads, thread t holds the P
// consecutive points t*P .. t*P+P-1 in registers.
#pragma once
#include <math.h>

#include "pn2_common.h"

namespace pn2 {

constexpr int BIN_GMAX = 16;                                   // cells per axis at most
constexpr int BIN_MAXCELLS = BIN_GMAX * BIN_GMAX * BIN_GMAX;   // 4096
constexpr int BIN_MAXN = 65535;                                // slots are 16-bit
constexpr size_t BIN_HDR_BYTES = 64;
constexpr size_t BIN_PLAN_OFF = 128;
constexpr int BIN_PLAN_WORDS = 16;
constexpr float BIN_FAR = 1.0e18f;                             // coordinates of the padding entries

struct BinHeader {
    float mnx, mny, mnz;      // lower corner of the bounding box
    float ihx, ihy, ihz;      // cells per unit length (0: one cell on that axis)
    int Gx, Gy, Gz;           // cells per axis
    float m2;                 // largest |p|^2 of the block (the rounding slack of R' was sized for it)
    float r2;                 // the squared radius the plan was built for
    int nonfinite;            // 1: some coordinate is NaN/Inf -> every centroid tests every point
    int N, S;
    int rp;                   // pitch of a packed row in bytes
    int pad;
};
static_assert(sizeof(BinHeader) == BIN_HDR_BYTES, "header layout");

__host__ __device__ inline int bin_npad(int N) { return (N + 31) & ~31; }
__host__ __device__ inline int bin_row_pitch(int D)
{
    const int bytes = (3 + D) * 4;
    if (bytes > 128) return (bytes + 127) & ~127;
    int p = 16;
    while (p < bytes) p <<= 1;
    return p;
}
__host__ __device__ inline size_t bin_sorted_off(int S) { return BIN_PLAN_OFF + (size_t)S * BIN_PLAN_WORDS * 4; }
__host__ __device__ inline size_t bin_rows_off(int N, int S) { return (bin_sorted_off(S) + (size_t)(bin_npad(N) + 32) * 16 + 127) & ~(size_t)127; }
__host__ __device__ inline size_t bin_block_bytes(int N, int S, int D)
{
    return (bin_rows_off(N, S) + (size_t)N * bin_row_pitch(D) + 127) & ~(size_t)127;
}

// LDS the routine needs: histogram / starts + reduction scratch
template <int T>
constexpr size_t bin_lds_bytes() { return (BIN_MAXCELLS + 4) * sizeof(unsigned) + (T / 64) * 8 * sizeof(float) + 64; }

// max over each 16-lane row, result in every lane of the row (4 DPP steps)
__device__ __forceinline__ float bin_row_max(float v)
{
    v = fmaxf(v, __int_as_float(dpp_i32<0xB1>(__float_as_int(v))));    // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(dpp_i32<0x4E>(__float_as_int(v))));    // quad_perm [2,3,0,1]
    v = fmaxf(v, __int_as_float(dpp_i32<0x141>(__float_as_int(v))));   // row_half_mirror
    v = fmaxf(v, __int_as_float(dpp_i32<0x140>(__float_as_int(v))));   // row_mirror
    return v;
}
__device__ __forceinline__ float bin_wave_max(float v)
{
    v = bin_row_max(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Cell width rule shared by producer and consumer.  |computed dist - true dist| <= 20 u M2 (u = 2^-24, M2 the
// largest squared norm involved: 3 roundings in the dot, 3 per norm, 2 in the sums, on values <= 4 M2); 2^-19 M2
// covers it, and 0.1 % on R' covers the rounding of the cell coordinates themselves.
__device__ __forceinline__ float bin_cell_width(float r2, float m2) { return sqrtf(r2 + m2 * 1.9073486328125e-06f) * 1.001f; }

__device__ __forceinline__ int bin_axis_cell(float x, float mn, float ih, int G)
{
    return min(G - 1, max(0, (int)((x - mn) * ih)));
}

// Builds the plan of one block.  px/py/pz: this thread's P points (index tid*P + k; entries past N are
// ignored); cent [S][3] the block's centroids; D only sizes the layout (the packed rows are written by
// ball_pack_rows_kernel, which needs many workgroups).  smem: bin_lds_bytes<T>() bytes, 16-byte aligned.
// Contains workgroup barriers: every thread of the workgroup must call it.
template <int T, int P>
__device__ __forceinline__ void bin_block(const float (&px)[P], const float (&py)[P], const float (&pz)[P], int N, float r2,
                                          int D, const float *cent, int S, char *smem, char *table)
{
    static_assert(BIN_MAXCELLS % T == 0 && (BIN_MAXCELLS / T) % 4 == 0, "cells per thread must be a multiple of 4");
    constexpr int W = T / 64;
    constexpr int CPT = BIN_MAXCELLS / T;     // cells per thread in the scan
    unsigned *start = reinterpret_cast<unsigned *>(smem);                       // [BIN_MAXCELLS + 4]
    float *red = reinterpret_cast<float *>(start + BIN_MAXCELLS + 4);           // [W][8]
    unsigned *wsum = reinterpret_cast<unsigned *>(red + W * 8);                 // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < (BIN_MAXCELLS + 4) / 4; i += T) reinterpret_cast<uint4 *>(start)[i] = make_uint4(0u, 0u, 0u, 0u);
    // bounding box (as maxima of +-coordinate), largest squared norm, non-finite flag
    float q[8] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < P; ++k) {
        if (tid * P + k < N) {
            const float pn = norm3(px[k], py[k], pz[k]);
            q[0] = fmaxf(q[0], -px[k]); q[1] = fmaxf(q[1], -py[k]); q[2] = fmaxf(q[2], -pz[k]);
            q[3] = fmaxf(q[3], px[k]);  q[4] = fmaxf(q[4], py[k]);  q[5] = fmaxf(q[5], pz[k]);
            q[6] = fmaxf(q[6], pn);
            q[7] = fmaxf(q[7], (pn < INFINITY) ? 0.0f : 1.0f);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = bin_wave_max(q[k]);
    if (lane < 8) {
        float v = q[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) v = lane == k ? q[k] : v;
        red[wave * 8 + lane] = v;
    }
    __syncthreads();
    {
        // lane -> (wave lane & 15, quantity lane >> 4 and 4 + lane >> 4); 16-lane row maxima, then one lane per row
        const int wv = (lane & 15) < W ? (lane & 15) : 0;
        const float va = bin_row_max(red[wv * 8 + (lane >> 4)]);
        const float vb = bin_row_max(red[wv * 8 + 4 + (lane >> 4)]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            q[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(va), 16 * k));
            q[4 + k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vb), 16 * k));
        }
    }
    const float mnx = -q[0], mny = -q[1], mnz = -q[2];
    const float Rp = bin_cell_width(r2, q[6]);
    int Gx = 1, Gy = 1, Gz = 1;
    float ihx = 0.0f, ihy = 0.0f, ihz = 0.0f;
    const float ex = q[3] - mnx, ey = q[4] - mny, ez = q[5] - mnz;
    const bool finite = q[7] == 0.0f && Rp > 0.0f && Rp < INFINITY;
    if (finite) {
        if (ex > 0.0f) { Gx = (int)fminf(fmaxf(floorf(ex / Rp), 1.0f), (float)BIN_GMAX); ihx = (float)Gx / ex; }
        if (ey > 0.0f) { Gy = (int)fminf(fmaxf(floorf(ey / Rp), 1.0f), (float)BIN_GMAX); ihy = (float)Gy / ey; }
        if (ez > 0.0f) { Gz = (int)fminf(fmaxf(floorf(ez / Rp), 1.0f), (float)BIN_GMAX); ihz = (float)Gz / ez; }
    }
    PN2_STAMP(2);
    // histogram; the atomic's return value is the point's rank inside its cell
    int pcell[P];
    unsigned prank[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        pcell[k] = 0;
        prank[k] = 0;
        if (tid * P + k < N) {
            const int ix = bin_axis_cell(px[k], mnx, ihx, Gx), iy = bin_axis_cell(py[k], mny, ihy, Gy),
                      iz = bin_axis_cell(pz[k], mnz, ihz, Gz);
            pcell[k] = (iz * Gy + iy) * Gx + ix;          // x fastest: a cell and its x-neighbours are adjacent runs
            prank[k] = atomicAdd(&start[pcell[k]], 1u);
        }
    }
    __syncthreads();
    PN2_STAMP(3);
    // exclusive scan of the cell counts (CPT cells per thread)
    {
        const int c0 = tid * CPT;
        unsigned a[CPT];
        unsigned s = 0;
#pragma unroll
        for (int v4 = 0; v4 < CPT / 4; ++v4) {
            const uint4 t = *reinterpret_cast<const uint4 *>(&start[c0 + 4 * v4]);
            a[4 * v4 + 0] = t.x; a[4 * v4 + 1] = t.y; a[4 * v4 + 2] = t.z; a[4 * v4 + 3] = t.w;
            s += t.x + t.y + t.z + t.w;
        }
        unsigned inc = s;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned off = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) off += w < wave ? wsum[w] : 0u;
        unsigned e = off + inc - s;
#pragma unroll
        for (int v4 = 0; v4 < CPT / 4; ++v4) {
            const unsigned e0 = e, e1 = e0 + a[4 * v4], e2 = e1 + a[4 * v4 + 1], e3 = e2 + a[4 * v4 + 2];
            e = e3 + a[4 * v4 + 3];
            *reinterpret_cast<uint4 *>(&start[c0 + 4 * v4]) = make_uint4(e0, e1, e2, e3);
        }
        if (tid == 0) start[BIN_MAXCELLS] = (unsigned)N;
    }
    __syncthreads();
    PN2_STAMP(4);
    // scatter into cell order
    float4 *gsorted = reinterpret_cast<float4 *>(table + bin_sorted_off(S));
#pragma unroll
    for (int k = 0; k < P; ++k) {
        if (tid * P + k < N) {
            const unsigned pos = start[pcell[k]] + prank[k];
            gsorted[pos] = make_float4(px[k], py[k], pz[k], __int_as_float(tid * P + k));
        }
    }
    for (int i = N + tid; i < bin_npad(N) + 32; i += T) gsorted[i] = make_float4(BIN_FAR, BIN_FAR, BIN_FAR, __int_as_float(0));
    // the plan: per centroid, the nine runs of its 27 neighbouring cells
    {
        uint4 *gplan = reinterpret_cast<uint4 *>(table + BIN_PLAN_OFF);
        for (int c = tid; c < S; c += T) {
            const float cx = cent[c * 3 + 0], cy = cent[c * 3 + 1], cz = cent[c * 3 + 2];
            const float cn = norm3(cx, cy, cz);
            // a centroid whose norm exceeds the block's largest (new_xyz is not a subset of xyz) or that is not
            // finite is outside what the cell width was sized for: it tests every point, like the reference does
            const bool full = !finite || !(cn <= q[6]);
            const int ccx = bin_axis_cell(cx, mnx, ihx, Gx), ccy = bin_axis_cell(cy, mny, ihy, Gy), ccz = bin_axis_cell(cz, mnz, ihz, Gz);
            const int x0 = max(ccx - 1, 0), x1 = min(ccx + 1, Gx - 1);
            unsigned r[9];
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int z = ccz + dz - 1, y = ccy + dy - 1;
                    const bool in = z >= 0 && z < Gz && y >= 0 && y < Gy && !full;
                    const int base = in ? (z * Gy + y) * Gx : 0;
                    const unsigned rs = start[base + x0], re = start[base + x1 + 1];
                    r[dz * 3 + dy] = in ? rs | (re << 16) : 0u;
                }
            }
            gplan[c * 4 + 0] = make_uint4(r[0], r[1], r[2], r[3]);
            gplan[c * 4 + 1] = make_uint4(r[4], r[5], r[6], r[7]);
            gplan[c * 4 + 2] = make_uint4(r[8], full ? 1u : 0u, 0u, 0u);
            gplan[c * 4 + 3] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    const int rp = bin_row_pitch(D);
    if (tid == 0) {
        BinHeader h;
        h.mnx = mnx; h.mny = mny; h.mnz = mnz;
        h.ihx = ihx; h.ihy = ihy; h.ihz = ihz;
        h.Gx = Gx; h.Gy = Gy; h.Gz = Gz;
        h.m2 = q[6]; h.r2 = r2; h.nonfinite = finite ? 0 : 1; h.N = N; h.S = S;
        h.rp = rp; h.pad = 0;
        *reinterpret_cast<BinHeader *>(table) = h;
    }
    PN2_STAMP(5);
}

}  // namespace pn2
