// query_ball_point + grouping with cell pruning (gfx950).
//
// The reference masks the full [S,N] distance matrix and sorts it (models/pointnet2_utils.py:98-103);
// its result per centroid is "the nsample lowest indices among the points with dist <= r^2, padded
// with the lowest".  That set only contains points within r of the centroid, so this kernel bins the
// block's points into a uniform grid whose cells are at least R' wide (R' = r plus the worst-case
// rounding of the reference's fp32 distance expression, so a point the reference accepts can never
// sit outside the 27 neighbouring cells), tests only those candidates with the reference's exact
// expression (pn2::pair_sqdist), and orders the few hits by index.  Same output bit for bit, ~3 % of
// the pair tests at SA1 (N=4096, r=0.1); what remains is the HBM traffic of the grouped tensor.
//
// One 1024-thread workgroup = one block's points (<= 4096, counting-sorted by cell into LDS) x 64
// centroids, 16 lanes per centroid.  Members set their bit in a per-centroid bitmap indexed by the
// ORIGINAL point index, so "the nsample lowest indices" are simply the first set bits: no list, no
// capacity, no sort.  A block with non-finite coordinates gets a single cell (every point is a
// candidate), so NaN/Inf behave exactly as in the reference's full matrix.
#include <math.h>

#include "pn2_common.h"

namespace {

constexpr int GR_THREADS = 1024;
constexpr int GR_WAVES = 16;
constexpr int GR_CENT = 64;                   // centroids per workgroup (16 lanes each)
constexpr int GR_MAXN = 4096;
constexpr int GR_GMAX = 16;                   // cells per axis (16^3 = 4096 cells at most)
constexpr int GR_MAXCELLS = GR_GMAX * GR_GMAX * GR_GMAX;
constexpr int GR_BMW = GR_MAXN / 32;          // bitmap words per centroid
constexpr int GR_PT = GR_MAXN / GR_THREADS;   // points per thread

// vectors that are only 4-byte aligned in memory (rows of 3 / D floats): global loads of 12 / 16 bytes
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// max over each 16-lane row, result in every lane of the row (4 DPP steps)
__device__ __forceinline__ float row_max_f(float v)
{
    v = fmaxf(v, __int_as_float(pn2::dpp_i32<0xB1>(__float_as_int(v))));    // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(pn2::dpp_i32<0x4E>(__float_as_int(v))));    // quad_perm [2,3,0,1]
    v = fmaxf(v, __int_as_float(pn2::dpp_i32<0x141>(__float_as_int(v))));   // row_half_mirror
    v = fmaxf(v, __int_as_float(pn2::dpp_i32<0x140>(__float_as_int(v))));   // row_mirror
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
    v = row_max_f(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// max over the 64 lanes with the DPP operand fused into v_max_f32 (the builtin form canonicalises both operands of every
// fmaxf: three instructions a step); IEEE maxNum: a NaN lane is ignored.  Result uniform.
__device__ __forceinline__ float wave_max_f_dpp(float v)
{
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// SAMPLED = false: the grid spans the bounding box of the block's points (a workgroup-wide reduction of eight quantities).
// SAMPLED = true (round 4): no block-wide reduction.  ANY box gives a correct grid -- cell_of() clamps, the clamp is
// monotone, and two points at most one cell width apart land at most one cell apart wherever the box sits -- so the box
// is that of the block's first 64 centroids (farthest point sampling picks the extremes first: it is nearly the block's),
// reduced inside each wave, identically in all of them.  The rounding slack of the reference's distance expression comes
// from each centroid's own norm instead of the block's largest: a pair the reference accepts has true distance t with
// t^2 <= r^2 + 2^-19 (|c| + t)^2, hence t <= R'_c = sqrt(r^2 + 2^-19 (|c| + 2r)^2) whenever R'_c <= 2r; a centroid whose
// R'_c exceeds the cell width (or 2r, or is not finite) tests every point, and so does the whole workgroup when the block
// holds a non-finite coordinate (a flag in LDS).
template <bool SAMPLED>
__global__ __launch_bounds__(GR_THREADS) void ball_query_group_grid_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ points,
    int B, int N, int S, int K, int D, int ldg, float r2, int tiles_per_block, unsigned ldg_magic,
    int64_t *__restrict__ idx, float *__restrict__ grouped, int32_t *err_count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *sP = reinterpret_cast<float4 *>(smem);                // [4096] cell-sorted (x, y, z, |p|^2)
    unsigned *bm = reinterpret_cast<unsigned *>(sP + GR_MAXN);    // [64][128] member bitmaps (bit = original index)
    unsigned *start = bm + GR_CENT * GR_BMW;                      // [GR_MAXCELLS + 4] histogram, then exclusive starts
    float *red = reinterpret_cast<float *>(start + GR_MAXCELLS + 4);      // [16][8] reduction scratch
    unsigned *wsum = reinterpret_cast<unsigned *>(red + GR_WAVES * 8);    // [16]
    unsigned short *sI = reinterpret_cast<unsigned short *>(wsum + GR_WAVES);  // [4096] original index of sorted slot
    unsigned short *mIdx = sI + GR_MAXN;                          // [64][K] result indices, ascending

    const unsigned logical = pn2::xcd_remap(blockIdx.x, gridDim.x);
    const int b = (int)(logical / (unsigned)tiles_per_block);
    const int tile = (int)(logical % (unsigned)tiles_per_block);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s_base = tile * GR_CENT;
    const float *bx = xyz + (size_t)b * N * 3;
    const float *bc = new_xyz + (size_t)b * S * 3;
    PN2_STAMP(0);

    // ---- (1) this thread's 4 points (12 consecutive floats when the block is 16-byte aligned) ---------
    float px[GR_PT], py[GR_PT], pz[GR_PT];
    int pj[GR_PT];
    if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(xyz) & 15) == 0) {
        const int j0 = tid * GR_PT;
        const float4 *src = reinterpret_cast<const float4 *>(bx + (size_t)(j0 < N ? j0 : 0) * 3);
        const float4 q0 = src[0], q1 = src[1], q2 = src[2];
        px[0] = q0.x; py[0] = q0.y; pz[0] = q0.z;
        px[1] = q0.w; py[1] = q1.x; pz[1] = q1.y;
        px[2] = q1.z; py[2] = q1.w; pz[2] = q2.x;
        px[3] = q2.y; py[3] = q2.z; pz[3] = q2.w;
#pragma unroll
        for (int i = 0; i < GR_PT; ++i) pj[i] = j0 + i;
    } else {
#pragma unroll
        for (int i = 0; i < GR_PT; ++i) {
            const int j = tid + i * GR_THREADS;
            const int jj = j < N ? j : 0;
            pj[i] = j;
            px[i] = bx[jj * 3 + 0];
            py[i] = bx[jj * 3 + 1];
            pz[i] = bx[jj * 3 + 2];
        }
    }
    // this thread's centroid (16 lanes share one)
    const int cl = tid >> 4, l16 = tid & 15;
    const int my_s = s_base + cl;
    const bool s_ok = my_s < S;
    const int scl = s_ok ? my_s : S - 1;
    const float cx = bc[scl * 3 + 0], cy = bc[scl * 3 + 1], cz = bc[scl * 3 + 2];
    const float cn = pn2::norm3(cx, cy, cz);
    PN2_STAMP(1);

    for (int i = tid; i < GR_MAXCELLS + 4; i += GR_THREADS) start[i] = 0;
    {
        uint4 *z4 = reinterpret_cast<uint4 *>(bm);
        for (int i = tid; i < GR_CENT * GR_BMW / 4; i += GR_THREADS) z4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    float pn[GR_PT];
    float q[8];
    if (SAMPLED) {
        // this lane's sample: centroid `lane` of the block (loaded before the barrier: its latency hides behind the zeroing)
        const int sl = lane < S ? lane : S - 1;
        const float sx = bc[sl * 3 + 0], sy = bc[sl * 3 + 1], sz = bc[sl * 3 + 2];
        if (tid == 0) red[0] = 0.0f;                       // the block's "a coordinate is not finite" flag (the box scratch is free)
        __syncthreads();                                   // zeroed histogram / bitmaps / flag before anybody adds to them
        PN2_STAMP(2);
        q[0] = wave_max_f_dpp(-sx); q[1] = wave_max_f_dpp(-sy); q[2] = wave_max_f_dpp(-sz);
        q[3] = wave_max_f_dpp(sx);  q[4] = wave_max_f_dpp(sy);  q[5] = wave_max_f_dpp(sz);
        const float sn = pn2::norm3(sx, sy, sz);
        q[6] = wave_max_f_dpp(sn);
        // a sample that is not finite: v_max ignored its NaNs, an infinity shows in the extent below
        q[7] = 0.0f;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < GR_PT; ++i) {
            pn[i] = pn2::norm3(px[i], py[i], pz[i]);
            bad |= pj[i] < N && !(pn[i] < INFINITY);
        }
        if (bad) *reinterpret_cast<volatile float *>(red) = 1.0f;
        PN2_STAMP(3);
    } else {
    // bounding box (as maxima of +-coordinate), largest squared norm, non-finite flag
    q[0] = q[1] = q[2] = q[3] = q[4] = q[5] = -INFINITY;
    q[6] = cn;
    q[7] = (cn < INFINITY) ? 0.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < GR_PT; ++i) {
        pn[i] = pn2::norm3(px[i], py[i], pz[i]);
        if (pj[i] < N) {
            q[0] = fmaxf(q[0], -px[i]); q[1] = fmaxf(q[1], -py[i]); q[2] = fmaxf(q[2], -pz[i]);
            q[3] = fmaxf(q[3], px[i]);  q[4] = fmaxf(q[4], py[i]);  q[5] = fmaxf(q[5], pz[i]);
            q[6] = fmaxf(q[6], pn[i]);
            q[7] = fmaxf(q[7], (pn[i] < INFINITY) ? 0.0f : 1.0f);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = wave_max_f(q[k]);
    if (lane < 8) {
        float v = q[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) v = lane == k ? q[k] : v;
        red[wave * 8 + lane] = v;
    }
    PN2_STAMP(2);
    __syncthreads();
    PN2_STAMP(3);
    {
        // lane -> (wave lane & 15, quantity lane >> 4 and 4 + lane >> 4); 16-lane row maxima, then one lane per row
        const float va = row_max_f(red[(lane & 15) * 8 + (lane >> 4)]);
        const float vb = row_max_f(red[(lane & 15) * 8 + 4 + (lane >> 4)]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            q[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(va), 16 * k));
            q[4 + k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vb), 16 * k));
        }
    }
    }
    const float mnx = -q[0], mny = -q[1], mnz = -q[2];
    // ---- (2) grid: cells at least R' wide.  |computed dist - true dist| <= 20 u M2 (u = 2^-24, M2 the
    //      largest squared norm: 3 roundings in the dot, 3 per norm, 2 in the sums, on values <= 4 M2);
    //      2^-19 M2 covers it, and 0.1 % on R' covers the rounding of the cell coordinates themselves.
    // SAMPLED: (|c| + 2r)^2 <= 2 |c|^2 + 8 r^2 bounds the squared norm of anything in a sampled centroid's ball (no root)
    const float Rp = sqrtf(r2 + (SAMPLED ? 2.0f * q[6] + 8.0f * r2 : q[6]) * 1.9073486328125e-06f) * 1.001f;
    int Gx = 1, Gy = 1, Gz = 1;
    float ihx = 0.0f, ihy = 0.0f, ihz = 0.0f;
    const float ex = q[3] - mnx, ey = q[4] - mny, ez = q[5] - mnz;
    float hmin2 = INFINITY;                                // SAMPLED: square of the narrowest cell (an axis with one cell constrains nothing)
    if (q[7] == 0.0f && Rp > 0.0f && Rp < INFINITY) {
        if (SAMPLED) {
            // hardware reciprocals (1 ulp) behind a 2^-10 safety factor: G <= ex / R' holds, and the cell width 1 / ih the
            // binning uses differs from ex / G by ulps, inside the 0.1 % already on R'
            const float iR = __builtin_amdgcn_rcpf(Rp) * 0.9990234375f;
            if (ex > 0.0f) { Gx = (int)fminf(fmaxf(floorf(ex * iR), 1.0f), (float)GR_GMAX); ihx = (float)Gx * __builtin_amdgcn_rcpf(ex); }
            if (ey > 0.0f) { Gy = (int)fminf(fmaxf(floorf(ey * iR), 1.0f), (float)GR_GMAX); ihy = (float)Gy * __builtin_amdgcn_rcpf(ey); }
            if (ez > 0.0f) { Gz = (int)fminf(fmaxf(floorf(ez * iR), 1.0f), (float)GR_GMAX); ihz = (float)Gz * __builtin_amdgcn_rcpf(ez); }
            const float wx = ex * __builtin_amdgcn_rcpf((float)Gx), wy = ey * __builtin_amdgcn_rcpf((float)Gy), wz = ez * __builtin_amdgcn_rcpf((float)Gz);
            if (Gx > 1) hmin2 = fminf(hmin2, wx * wx);
            if (Gy > 1) hmin2 = fminf(hmin2, wy * wy);
            if (Gz > 1) hmin2 = fminf(hmin2, wz * wz);
        } else {
            if (ex > 0.0f) { Gx = (int)fminf(fmaxf(floorf(ex / Rp), 1.0f), (float)GR_GMAX); ihx = (float)Gx / ex; }
            if (ey > 0.0f) { Gy = (int)fminf(fmaxf(floorf(ey / Rp), 1.0f), (float)GR_GMAX); ihy = (float)Gy / ey; }
            if (ez > 0.0f) { Gz = (int)fminf(fmaxf(floorf(ez / Rp), 1.0f), (float)GR_GMAX); ihz = (float)Gz / ez; }
        }
    }
    auto cell_of = [&](float x, float y, float z, int &ix, int &iy, int &iz) {
        ix = min(Gx - 1, max(0, (int)((x - mnx) * ihx)));
        iy = min(Gy - 1, max(0, (int)((y - mny) * ihy)));
        iz = min(Gz - 1, max(0, (int)((z - mnz) * ihz)));
    };

    // ---- (3) histogram; the atomic's return value is the point's rank inside its cell ---------------
    int pcell[GR_PT];
    unsigned prank[GR_PT];
#pragma unroll
    for (int i = 0; i < GR_PT; ++i) {
        pcell[i] = 0;
        prank[i] = 0;
        if (pj[i] < N) {
            int ix, iy, iz;
            cell_of(px[i], py[i], pz[i], ix, iy, iz);
            pcell[i] = (iz * Gy + iy) * Gx + ix;
            prank[i] = atomicAdd(&start[pcell[i]], 1u);
        }
    }
    PN2_STAMP(4);
    __syncthreads();
    PN2_STAMP(5);
    // ---- (4) exclusive scan of the cell counts (4 cells per thread) -------------------------------
    {
        const int c0 = tid * 4;
        const uint4 a = *reinterpret_cast<const uint4 *>(&start[c0]);
        const unsigned s = a.x + a.y + a.z + a.w;
        unsigned inc = s;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned off = 0;
#pragma unroll
        for (int w4 = 0; w4 < GR_WAVES / 4; ++w4) {
            const uint4 ws = *reinterpret_cast<const uint4 *>(&wsum[4 * w4]);
            off += (4 * w4 + 0 < wave ? ws.x : 0u) + (4 * w4 + 1 < wave ? ws.y : 0u) + (4 * w4 + 2 < wave ? ws.z : 0u) +
                   (4 * w4 + 3 < wave ? ws.w : 0u);
        }
        const unsigned e0 = off + inc - s;
        *reinterpret_cast<uint4 *>(&start[c0]) = make_uint4(e0, e0 + a.x, e0 + a.x + a.y, e0 + a.x + a.y + a.z);
        if (tid == 0) start[GR_MAXCELLS] = (unsigned)N;
    }
    __syncthreads();
    PN2_STAMP(6);
    // ---- (5) scatter into cell order (cells past the grid are empty, their start is N) ---------------
#pragma unroll
    for (int i = 0; i < GR_PT; ++i) {
        if (pj[i] < N) {
            const unsigned pos = start[pcell[i]] + prank[i];
            sP[pos] = make_float4(px[i], py[i], pz[i], pn[i]);
            sI[pos] = (unsigned short)pj[i];
        }
    }
    __syncthreads();
    PN2_STAMP(7);

    // From here on a centroid's 16 lanes depend on nothing but their own bitmap row, so there is no
    // workgroup barrier any more: groups that are still testing candidates (LDS / vector work) run
    // beside groups that already gather and store their rows (memory work).
    if (!s_ok) return;
    // ---- (6) candidates of the 27 neighbouring cells (the 3 x-neighbours are one contiguous run of the
    //      cell-sorted array), 16 per step; members set their bit ------------------------------------
    unsigned *mybm = bm + cl * GR_BMW;
    {
        int ccx, ccy, ccz;
        cell_of(cx, cy, cz, ccx, ccy, ccz);
        const int x0 = max(ccx - 1, 0), x1 = min(ccx + 1, Gx - 1);
        const int z0 = max(ccz - 1, 0), z1 = min(ccz + 1, Gz - 1);
        const int y0 = max(ccy - 1, 0), y1 = min(ccy + 1, Gy - 1);
        // The (up to) 9 runs as ONE index space: all 18 range reads are in flight together and the 16 lanes walk
        // the concatenation (ceil(total / 16) steps instead of a partly filled last step per run).
        int off[9], cum[9];
        int tot = 0;
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int z = z0 + dz, y = y0 + dy;
                const bool ok = z <= z1 && y <= y1;
                const int base = ((ok ? z : z0) * Gy + (ok ? y : y0)) * Gx;
                const int rs = (int)start[base + x0], re = (int)start[base + x1 + 1];
                off[dz * 3 + dy] = rs - tot;                   // slot = off + position inside the concatenation
                tot += ok ? re - rs : 0;
                cum[dz * 3 + dy] = tot;
            }
        }
        bool all = false;
        if (SAMPLED) {
            // this centroid's own rounding slack ((|c| + 2r)^2 <= 2 |c|^2 + 8 r^2) against the cell width, 0.3 % on the
            // squares for the roundings; a non-finite coordinate anywhere in the block
            const float Rc2 = r2 + (2.0f * cn + 8.0f * r2) * 1.9073486328125e-06f;
            all = !(Rc2 <= 4.0f * r2) || !(Rc2 * 1.003f <= hmin2) || *reinterpret_cast<volatile float *>(red) != 0.0f;
            tot = all ? N : tot;                               // every point of the block, in sorted order
        }
#pragma unroll 2
        for (int pos = l16; pos < tot; pos += 16) {
            int o = off[8];
#pragma unroll
            for (int i = 7; i >= 0; --i) o = pos < cum[i] ? off[i] : o;
            if (SAMPLED) o = all ? 0 : o;
            const int j = o + pos;
            const float4 p = sP[j];
            const float d = pn2::pair_sqdist(cx, cy, cz, cn, p.x, p.y, p.z, p.w);
            if (!(d > r2)) {                                   // reference :102 masks d > r^2
                const unsigned i = sI[j];
                atomicOr(&mybm[i >> 5], 1u << (i & 31u));
            }
        }
    }
    // LDS operations of one wave complete in order; the fence keeps the compiler from moving the reads up
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    PN2_STAMP(8);
    // ---- (7) the K lowest set bits, in order: lane l16 owns words 8*l16 .. 8*l16+7 ----------------------
    unsigned short *oi = mIdx + cl * K;
    int n;
    {
        unsigned w[8];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { w[k] = mybm[l16 * 8 + k]; mine += __builtin_popcount(w[k]); }
        int inc = mine;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int t = __shfl_up(inc, o, 16);
            if (l16 >= o) inc += t;
        }
        const int total = __shfl(inc, 15, 16);
        int pos = inc - mine;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned bits = w[k];
            while (bits && pos < K) {
                const int bit = __builtin_ctz(bits);
                bits &= bits - 1u;
                oi[pos++] = (unsigned short)((l16 * 8 + k) * 32 + bit);
            }
        }
        n = min(total, K);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    PN2_STAMP(9);
    // ---- (8) idx [b, s, 0..K) (padded with the first member, :104-106) -------------------------------
    {
        int64_t *orow = idx + ((size_t)b * S + my_s) * K;
        for (int k = l16; k < K; k += 16) orow[k] = n > 0 ? (int64_t)oi[k < n ? k : 0] : (int64_t)N;   // empty: IndexError at :59
        if (n == 0 && l16 == 0 && err_count) atomicAdd(err_count, 1);
    }
    PN2_STAMP(10);
    if (!grouped) return;
    // ---- (9) grouped rows [xyz - centroid, feats] of this centroid: K rows of qpr float4, contiguous;
    //      lane l16 writes float4 number l16 + 16 i ------------------------------------------------------
    const int Cg = 3 + D;
    const int qpr = Cg >> 2;
    const int E = K * qpr;
    const float *bp = points ? points + (size_t)b * N * D : nullptr;
    float4 *g4 = reinterpret_cast<float4 *>(grouped + ((size_t)b * S + my_s) * (size_t)K * ldg);
    // Branch-free gathers (a load under a divergent branch is waited for at the end of the branch, which
    // would serialise them): every element is one 16-byte load (xyz through the same path: 4 floats that
    // contain the point's 3) plus the row's first feature, 6 elements in flight per lane.
    constexpr int U = 6;
    for (int e0 = l16; e0 < E; e0 += U * 16) {
        f32x4u f[U];
        float f0[U];
        int part[U], shift[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u * 16, E - 1);
            const int k = qpr == 1 ? e : (int)__umulhi((unsigned)e, ldg_magic);             // e / qpr
            part[u] = e - k * qpr;
            const int j = n > 0 ? (int)oi[k < n ? k : 0] : 0;
            const float *row = bp ? bp + (size_t)j * D : bx;
            shift[u] = (j == N - 1) ? 1 : 0;                      // the last point is read as [.., x, y, z]
            const float *src = part[u] == 0 ? bx + (size_t)j * 3 - shift[u] : row + (4 * part[u] - 3);
            f[u] = *reinterpret_cast<const f32x4u *>(src);
            f0[u] = row[0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * 16;
            if (e < E) {
                float4 v = make_float4(f[u].x, f[u].y, f[u].z, f[u].w);
                if (part[u] == 0) {
                    const float x = shift[u] ? f[u].y : f[u].x, y = shift[u] ? f[u].z : f[u].y, z = shift[u] ? f[u].w : f[u].z;
                    v = make_float4(x - cx, y - cy, z - cz, D > 0 ? f0[u] : 0.0f);           // :128, :131
                }
                if (n <= 0) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                g4[e] = v;
            }
        }
    }
    PN2_STAMP(11);
    PN2_STAMP_DRAIN();                   // lab builds only: stamps 11 -> 12 = drain of the row stores
    PN2_STAMP(12);
}

size_t gr_lds_bytes(int K)
{
    return (size_t)GR_MAXN * sizeof(float4) + (size_t)GR_CENT * GR_BMW * sizeof(unsigned) + (GR_MAXCELLS + 4) * sizeof(unsigned) +
           GR_WAVES * 8 * sizeof(float) + GR_WAVES * sizeof(unsigned) + GR_MAXN * sizeof(unsigned short) +
           (size_t)GR_CENT * K * sizeof(unsigned short);
}

}  // namespace

namespace pn2 {

// Returns PN2_ERR_UNSUPPORTED when the shape is outside what this kernel is built for (the caller then
// uses the matrix-core or the vector-unit kernel).
int launch_ball_query_grid(const float *xyz, const float *new_xyz, const float *points, int B, int N, int S, int K,
                           int D, int ldg, float r2, int64_t *idx, float *grouped, int32_t *err_count,
                           hipStream_t stream, bool sampled_box)
{
    // N >= 2: the coordinate gather is one 16-byte load per point, shifted back by a float for the block's LAST point
    // (line "shift"); a one-point block has only 12 bytes, and for b == 0 the shifted load would start at xyz[-1]
    if (N > GR_MAXN || N < 2 || K > 64 || K < 1) return PN2_ERR_UNSUPPORTED;
    if (grouped && !(ldg == 3 + D && ((3 + D) & 3) == 0 && (reinterpret_cast<uintptr_t>(grouped) & 15) == 0))
        return PN2_ERR_UNSUPPORTED;
    if (grouped && D > 0 && !points) return PN2_ERR_NULL;
    const size_t lds = gr_lds_bytes(K);
    if (lds > 160 * 1024) return PN2_ERR_UNSUPPORTED;
    const int tiles = (S + GR_CENT - 1) / GR_CENT;
    const long long nwg = (long long)B * tiles;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    const int qpr = (3 + D) >> 2;
    const unsigned magic = qpr > 1 ? (unsigned)((1ULL << 32) / (unsigned)qpr) + 1u : 0u;     // e/qpr exact for e*qpr < 2^32
    if (sampled_box) {
        static pn2::PerDevice lds_memo;
        if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(ball_query_group_grid_kernel<true>), 160 * 1024, lds_memo)) return e;
        hipLaunchKernelGGL(ball_query_group_grid_kernel<true>, dim3((unsigned)nwg), dim3(GR_THREADS), lds, stream, xyz, new_xyz, points,
                           B, N, S, K, D, ldg, r2, tiles, magic, idx, grouped, err_count);
    } else {
        static pn2::PerDevice lds_memo;
        if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(ball_query_group_grid_kernel<false>), 160 * 1024, lds_memo)) return e;
        hipLaunchKernelGGL(ball_query_group_grid_kernel<false>, dim3((unsigned)nwg), dim3(GR_THREADS), lds, stream, xyz, new_xyz, points,
                           B, N, S, K, D, ldg, r2, tiles, magic, idx, grouped, err_count);
    }
    return PN2_LAUNCH_RC();
}

}  // namespace pn2
