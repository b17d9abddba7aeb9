// LAB KERNEL 2 (tools/bqlab/lab5.hip includes it): query_ball_point + grouping in ONE launch, no workspace -- spatial tiles
// again, but built from what the round's measurements say a phase costs on gfx950 ((vector + scalar instructions) x 4 clocks
// x waves per SIMD; a burst of row stores that nothing overlaps; 16-wave barriers):
//   * 512-thread workgroups, ~66 KB of LDS: TWO per CU, so one workgroup's stores drain under the other's compute;
//   * a workgroup owns one of S/32 spatial tiles of its block: it classifies the block's points against the tile's box
//     widened by R' (6 compares per point), compacts the candidates (~580 of 4096) in INDEX order (ballots give the ranks)
//     and cell-sorts only THOSE into a local grid (cells >= R' wide) -- one histogram pass whose atomics return the rank in
//     the cell, one scan, one scatter;
//   * per centroid 16 lanes (8 when a tile holds more than 32 centroids) walk the 27 neighbouring local cells as nine runs,
//     test with the reference's exact expression and set bit `local rank` -- the candidate's position in index order -- in
//     a 768-bit bitmap: "the nsample lowest indices, padded with the first" are its first set bits, 1.5 words per lane;
//   * the candidates' rows [x, y, z, feats] are staged in LDS by local rank (the feature loads are issued before the tests
//     and land under them), so the grouped rows are assembled without touching memory again.
// Centroids the tile argument does not cover (a norm above the block's largest, non-finite coordinates anywhere) and
// crowded tiles scan every point of the block from memory: same result, slower.
#include <math.h>

#include "pn2_ball_bin.h"

namespace {

constexpr int T2_THREADS = 512;
constexpr int T2_W = T2_THREADS / 64;
constexpr int T2_PPT = 8;                   // points per thread (N <= 4096)
constexpr int T2_G = T2_PPT / 4;
constexpr int T2_CAP = 768;                 // candidates a tile stages
constexpr int T2_BMW = 32;                  // bitmap words per centroid (>= CAP / 32, a power of two)
constexpr int T2_LG = 10;                   // local cells per axis at most
constexpr int T2_LC = T2_LG * T2_LG * T2_LG;
constexpr int T2_HIST = 2 * T2_THREADS + 8;  // the scan covers 2 cells per thread; one more entry = the total
static_assert(T2_LC + 1 <= T2_HIST, "local grid");
constexpr int T2_CL = 256;                  // centroids of a tile per scan round
constexpr int T2_MAXK = 64;
constexpr int T2_RP = 12;                   // floats per staged row: x, y, z, feats (D <= 9 ... rows of 48 bytes)
constexpr float T2_UPAD = 1.0e-4f;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v3i __attribute__((ext_vector_type(3)));

__device__ __forceinline__ float t2_vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float t2_vmax_neg(float a, float b) { float r; asm("v_max_f32_e64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int t2_med3i(int v, int lo, int hi) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi)); return r; }

#define T2_DPP6(CTRL) \
    "v_max_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\tv_max_f32_dpp %2, %2, %2 " CTRL "\n\t" \
    "v_max_f32_dpp %3, %3, %3 " CTRL "\n\tv_max_f32_dpp %4, %4, %4 " CTRL "\n\tv_max_f32_dpp %5, %5, %5 " CTRL "\n\t" \
    "v_max_u32_dpp %6, %6, %6 " CTRL "\n\t"
// wave maxima of six floats and one unsigned, results in lane 63 (interleaved chains: no DPP read of a just-written register)
__device__ __forceinline__ void t2_wave_max7(float &a, float &b, float &c, float &d, float &e, float &f, unsigned &u)
{
    asm("s_nop 1\n\t"
        T2_DPP6("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        T2_DPP6("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        T2_DPP6("row_half_mirror row_mask:0xf bank_mask:0xf")
        T2_DPP6("row_mirror row_mask:0xf bank_mask:0xf")
        T2_DPP6("row_bcast:15 row_mask:0xa bank_mask:0xf")
        T2_DPP6("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 1"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(u));
}
#define T2_ROW3(CTRL) "v_max_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\tv_max_u32_dpp %2, %2, %2 " CTRL "\n\t"
__device__ __forceinline__ void t2_row_max3(float &a, float &b, unsigned &u)
{
    asm("s_nop 1\n\t"
        T2_ROW3("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        T2_ROW3("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        T2_ROW3("row_half_mirror row_mask:0xf bank_mask:0xf")
        T2_ROW3("row_mirror row_mask:0xf bank_mask:0xf")
        "s_nop 1"
        : "+v"(a), "+v"(b), "+v"(u));
}

struct T2Smem { int red, wtot, misc, hist, clist, ccoord, sorted4, srank, cidx, frow, bm, mIdx, total; };
__host__ __device__ inline T2Smem t2_layout(int K)
{
    T2Smem s;
    int o = 0;
    s.red = o;     o += T2_W * 8 * 4;
    s.wtot = o;    o += 128;
    s.misc = o;    o += 64;
    s.hist = o;    o += T2_HIST * 4;
    s.clist = o;   o += T2_CL * 2;
    s.ccoord = o;  o += T2_CL * 16;
    s.sorted4 = o; o += T2_CAP * 16;
    s.srank = o;   o += T2_CAP * 2;
    s.cidx = o;    o += T2_CAP * 4;
    s.frow = o;    o += T2_CAP * T2_RP * 4;
    s.bm = o;      o += 64 * T2_BMW * 4;            // 64 centroid slots (8 lanes each) or 32 (16 lanes each, half used)
    s.mIdx = o;    o += 64 * K * 2;
    s.total = (o + 15) & ~15;
    return s;
}

__global__ __launch_bounds__(T2_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ball_tile2_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ points, int N, int S, int K, int D,
    float r2, int tmax, unsigned qpr_magic, int64_t *__restrict__ idx, float *__restrict__ grouped, int32_t *err_count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const T2Smem L = t2_layout(K);
    float *red = reinterpret_cast<float *>(smem + L.red);
    unsigned *wtot = reinterpret_cast<unsigned *>(smem + L.wtot);
    unsigned *misc = reinterpret_cast<unsigned *>(smem + L.misc);
    unsigned *hist = reinterpret_cast<unsigned *>(smem + L.hist);
    unsigned short *clist = reinterpret_cast<unsigned short *>(smem + L.clist);
    float4 *ccoord = reinterpret_cast<float4 *>(smem + L.ccoord);
    float4 *sorted4 = reinterpret_cast<float4 *>(smem + L.sorted4);
    unsigned short *srank = reinterpret_cast<unsigned short *>(smem + L.srank);
    unsigned *cidx = reinterpret_cast<unsigned *>(smem + L.cidx);
    float *frow = reinterpret_cast<float *>(smem + L.frow);
    unsigned *bm = reinterpret_cast<unsigned *>(smem + L.bm);
    unsigned short *mIdx = reinterpret_cast<unsigned short *>(smem + L.mIdx);

    const unsigned logical = pn2::xcd_remap(blockIdx.x, gridDim.x);
    const int b = (int)(logical / (unsigned)tmax);
    const int tile = (int)(logical - (unsigned)b * (unsigned)tmax);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xyz + (size_t)b * N * 3), 0, N * 12, 0x00020000);
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(new_xyz + (size_t)b * S * 3), 0, S * 12, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(points ? points + (size_t)b * N * D : xyz), 0, points ? (int)((unsigned)N * (unsigned)D * 4u) : 0, 0x00020000);
    PN2_STAMP(0);

    // ---- loads: 8 points per thread (wave w holds indices [512 w, 512 w + 512): (w, g, lane, k) order = index order) and
    //      the first round's centroids (2 per thread)
    float px[T2_PPT], py[T2_PPT], pz[T2_PPT];
    {
        const int voff = (wave * T2_PPT * 64 + lane * 4) * 12;
#pragma unroll
        for (int g = 0; g < T2_G; ++g) {
            const v4i a = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, g * 256 * 12, 0);
            const v4i c = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, g * 256 * 12 + 16, 0);
            const v4i d = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, g * 256 * 12 + 32, 0);
            px[4 * g + 0] = __int_as_float(a.x); py[4 * g + 0] = __int_as_float(a.y); pz[4 * g + 0] = __int_as_float(a.z);
            px[4 * g + 1] = __int_as_float(a.w); py[4 * g + 1] = __int_as_float(c.x); pz[4 * g + 1] = __int_as_float(c.y);
            px[4 * g + 2] = __int_as_float(c.z); py[4 * g + 2] = __int_as_float(c.w); pz[4 * g + 2] = __int_as_float(d.x);
            px[4 * g + 3] = __int_as_float(d.y); py[4 * g + 3] = __int_as_float(d.z); pz[4 * g + 3] = __int_as_float(d.w);
        }
    }
    auto index_of = [&](int i) { return wave * T2_PPT * 64 + (i >> 2) * 256 + lane * 4 + (i & 3); };
    v3i cq[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) cq[r] = __builtin_amdgcn_raw_buffer_load_b96(crs, tid * 12, r * T2_THREADS * 12, 0);
    // clear the local histogram and the counters while the loads fly
    for (int i = tid; i < T2_HIST / 4; i += T2_THREADS) reinterpret_cast<uint4 *>(hist)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 16) misc[tid] = 0u;

    // ---- 1. bounding box + non-finite flag (x * 0 is NaN for an infinite or NaN x): identical in every workgroup
    float q0 = -INFINITY, q1 = -INFINITY, q2 = -INFINITY, q3 = -INFINITY, q4 = -INFINITY, q5 = -INFINITY, nanacc = 0.0f;
    const bool whole = T2_THREADS * T2_PPT <= N;
#pragma unroll
    for (int i = 0; i < T2_PPT; ++i) {
        float x = px[i], y = py[i], z = pz[i];
        if (!whole) {                                // an index past N reads as this lane's first point (or nothing at all)
            const bool ok = index_of(i) < N;
            x = ok ? x : px[0]; y = ok ? y : py[0]; z = ok ? z : pz[0];
        }
        q0 = t2_vmax_neg(q0, x); q1 = t2_vmax_neg(q1, y); q2 = t2_vmax_neg(q2, z);
        q3 = t2_vmax(q3, x);     q4 = t2_vmax(q4, y);     q5 = t2_vmax(q5, z);
        nanacc = __builtin_fmaf(x, 0.0f, nanacc); nanacc = __builtin_fmaf(y, 0.0f, nanacc); nanacc = __builtin_fmaf(z, 0.0f, nanacc);
    }
    if (!whole && !(index_of(0) < N)) { q0 = q1 = q2 = q3 = q4 = q5 = -INFINITY; nanacc = 0.0f; }
    unsigned q6 = __float_as_uint(nanacc) & 0x7fffffffu;        // 0, or the bits of a NaN
    t2_wave_max7(q0, q1, q2, q3, q4, q5, q6);
    if (lane == 63) {
        reinterpret_cast<float4 *>(red + wave * 8)[0] = make_float4(q0, q1, q2, q3);
        reinterpret_cast<float4 *>(red + wave * 8)[1] = make_float4(q4, q5, __uint_as_float(q6), 0.0f);
    }
    PN2_STAMP(1);
    __syncthreads();
    float mn0, mn1, mn2, mx0, mx1, mx2;
    bool finite;
    {
        float va = red[(lane & 7) * 8 + (lane >> 4)];                         // rows: -x, -y, -z, x
        float vb = red[(lane & 7) * 8 + 4 + ((lane >> 4) & 1)];               // rows: y, z, y, z
        unsigned vu = __float_as_uint(red[(lane & 7) * 8 + 6]);
        t2_row_max3(va, vb, vu);
        mn0 = -__int_as_float(__builtin_amdgcn_readlane(__float_as_int(va), 0));
        mn1 = -__int_as_float(__builtin_amdgcn_readlane(__float_as_int(va), 16));
        mn2 = -__int_as_float(__builtin_amdgcn_readlane(__float_as_int(va), 32));
        mx0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(va), 48));
        mx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vb), 0));
        mx2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vb), 16));
        finite = __builtin_amdgcn_readlane((int)vu, 0) == 0;
    }
    PN2_STAMP(2);
    // an upper bound of every |p|^2 of the block, rounded like the reference rounds a norm (fp32 rounding is monotone)
    const float m2 = pn2::norm3(fmaxf(-mn0, mx0), fmaxf(-mn1, mx1), fmaxf(-mn2, mx2));
    const float Rp = pn2::bin_cell_width(r2, m2);
    finite = finite && Rp > 0.0f && Rp < INFINITY && m2 < INFINITY;

    // ---- 2. tiles: halve the axis whose tiles are widest until there are tmax of them
    int t0 = 1, t1 = 1, t2 = 1;
    float w0 = mx0 - mn0, w1 = mx1 - mn1, w2 = mx2 - mn2;
    if (finite) {
        while (t0 * t1 * t2 < tmax) {
            if (w0 >= w1 && w0 >= w2) {
                if (!(w0 > 0.0f)) break;
                t0 *= 2; w0 *= 0.5f;
            } else if (w1 >= w2) {
                t1 *= 2; w1 *= 0.5f;
            } else {
                t2 *= 2; w2 *= 0.5f;
            }
        }
    }
    const int ntile = finite ? t0 * t1 * t2 : tmax;
    if (tile >= ntile) return;                                   // uniform for the workgroup
    const int tc0 = tile % t0, tc1 = (tile / t0) % t1, tc2 = tile / (t0 * t1);
    // ownership box [olo, ohi) (outermost tiles reach to infinity) and the finite candidates' box [blo, bhi]
    const float lo0 = mn0 + (float)tc0 * w0, hi0 = tc0 == t0 - 1 ? mx0 : mn0 + (float)(tc0 + 1) * w0;
    const float lo1 = mn1 + (float)tc1 * w1, hi1 = tc1 == t1 - 1 ? mx1 : mn1 + (float)(tc1 + 1) * w1;
    const float lo2 = mn2 + (float)tc2 * w2, hi2 = tc2 == t2 - 1 ? mx2 : mn2 + (float)(tc2 + 1) * w2;
    const float olo0 = tc0 == 0 ? -INFINITY : lo0, ohi0 = tc0 == t0 - 1 ? INFINITY : hi0;
    const float olo1 = tc1 == 0 ? -INFINITY : lo1, ohi1 = tc1 == t1 - 1 ? INFINITY : hi1;
    const float olo2 = tc2 == 0 ? -INFINITY : lo2, ohi2 = tc2 == t2 - 1 ? INFINITY : hi2;
    const float h0 = Rp + T2_UPAD * w0, h1 = Rp + T2_UPAD * w1, h2 = Rp + T2_UPAD * w2;
    const float blo0 = lo0 - h0, bhi0 = hi0 + h0, blo1 = lo1 - h1, bhi1 = hi1 + h1, blo2 = lo2 - h2, bhi2 = hi2 + h2;
    // local grid over the candidates' box: cells at least R' wide (reciprocals with a safety factor: never narrower)
    int g0 = 1, g1 = 1, g2 = 1;
    float ih0 = 0.0f, ih1 = 0.0f, ih2 = 0.0f;
    if (finite) {
        const float ir = __builtin_amdgcn_rcpf(Rp) * 0.99999f;
        const float e0 = bhi0 - blo0, e1 = bhi1 - blo1, e2 = bhi2 - blo2;
        { const float g = __builtin_amdgcn_fmed3f(floorf(e0 * ir), 1.0f, (float)T2_LG); g0 = (int)g; ih0 = g * __builtin_amdgcn_rcpf(e0) * 0.999999f; }
        { const float g = __builtin_amdgcn_fmed3f(floorf(e1 * ir), 1.0f, (float)T2_LG); g1 = (int)g; ih1 = g * __builtin_amdgcn_rcpf(e1) * 0.999999f; }
        { const float g = __builtin_amdgcn_fmed3f(floorf(e2 * ir), 1.0f, (float)T2_LG); g2 = (int)g; ih2 = g * __builtin_amdgcn_rcpf(e2) * 0.999999f; }
    }
    const int g0m = g0 - 1, g1m = g1 - 1, g2m = g2 - 1;
    auto cell_of = [&](float x, float y, float z, int &ix, int &iy, int &iz) {
        ix = t2_med3i((int)((x - blo0) * ih0), 0, g0m);
        iy = t2_med3i((int)((y - blo1) * ih1), 0, g1m);
        iz = t2_med3i((int)((z - blo2) * ih2), 0, g2m);
    };

    // ---- 3. the centroids of this tile (first round: 2 per thread; later rounds reload), unordered list in LDS
    const int nround = (S + 2 * T2_THREADS - 1) / (2 * T2_THREADS);
    auto scan_centroids = [&](int cbase) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = cbase + r * T2_THREADS + tid;
            const float cx = __int_as_float(cq[r].x), cy = __int_as_float(cq[r].y), cz = __int_as_float(cq[r].z);
            const float cn = pn2::norm3(cx, cy, cz);
            const bool full = !finite || !(cn <= m2);            // outside what R' was sized for: tests every point
            const bool mine = c < S && (full ? (c & (ntile - 1)) == tile
                                             : (cx >= olo0 && cx < ohi0 && cy >= olo1 && cy < ohi1 && cz >= olo2 && cz < ohi2));
            const unsigned long long bmk = __builtin_amdgcn_ballot_w64(mine);
            if (bmk != 0ull) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&misc[0], (unsigned)__builtin_popcountll(bmk));
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                const unsigned pos = base + (unsigned)pn2::mbcnt(bmk);
                if (mine && pos < (unsigned)T2_CL) {
                    clist[pos] = (unsigned short)((r * T2_THREADS + tid) | (full ? 0x8000 : 0));
                    ccoord[pos] = make_float4(cx, cy, cz, cn);
                }
            }
        }
    };
    scan_centroids(0);

    // ---- 4. candidates: classify, histogram into the local grid (the atomic returns the rank in the cell), ranks in index order
    unsigned inmask = 0, mine = 0;
    unsigned pcell[T2_PPT];                                       // local cell | rank in the cell << 10
    if (finite) {
#pragma unroll
        for (int i = 0; i < T2_PPT; ++i) {
            const float x = px[i], y = py[i], z = pz[i];
            bool in = x >= blo0 && x <= bhi0 && y >= blo1 && y <= bhi1 && z >= blo2 && z <= bhi2;
            if (!whole) in = in && index_of(i) < N;
            pcell[i] = 0u;
            if (in) {
                int ix, iy, iz;
                cell_of(x, y, z, ix, iy, iz);
                const unsigned cell = (unsigned)((iz * g1 + iy) * g0 + ix);
                pcell[i] = cell | (atomicAdd(&hist[cell], 1u) << 10);
            }
            inmask |= in ? (1u << i) : 0u;
            mine += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
        }
    }
    if (lane == 0) wtot[wave] = mine;
    PN2_STAMP(3);
    __syncthreads();                                               // histogram, wave totals and the centroid list are complete
    unsigned pre = 0, ncand_total = 0;
#pragma unroll
    for (int w = 0; w < T2_W; ++w) {
        const unsigned c = wtot[w];
        pre += w < wave ? c : 0u;
        ncand_total += c;
    }
    const bool overloaded = ncand_total > (unsigned)T2_CAP;
    const int ncand = overloaded ? 0 : (int)ncand_total;
    // exclusive scan of the local histogram: 2 cells per thread, wave scan, 8 wave sums
    {
        const uint2 a = reinterpret_cast<const uint2 *>(hist)[tid];
        const unsigned s = a.x + a.y;
        unsigned inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wtot[16 + wave] = inc;
        __syncthreads();
        unsigned off = 0;
#pragma unroll
        for (int w = 0; w < T2_W; ++w) off += w < wave ? wtot[16 + w] : 0u;
        const unsigned e0 = off + inc - s;
        reinterpret_cast<uint2 *>(hist)[tid] = make_uint2(e0, e0 + a.x);       // hist[c] = first slot of cell c (c < 1024)
    }
    __syncthreads();
    PN2_STAMP(4);
    // scatter: sorted slot = start[cell] + rank in cell; local rank (index order) = wave base + lower lanes of the group + own lower k
    if (!overloaded) {
#pragma unroll
        for (int g = 0; g < T2_G; ++g) {
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
                if ((own >> k) & 1u) {
                    const unsigned lr = pre + lower + (unsigned)__builtin_popcount(own & ((1u << k) - 1u));
                    const unsigned pos = hist[pcell[i] & 1023u] + (pcell[i] >> 10);
                    const float x = px[i], y = py[i], z = pz[i];
                    sorted4[pos] = make_float4(x, y, z, pn2::norm3(x, y, z));
                    srank[pos] = (unsigned short)lr;
                    cidx[lr] = (unsigned)index_of(i);
                    frow[lr * T2_RP + 0] = x; frow[lr * T2_RP + 1] = y; frow[lr * T2_RP + 2] = z;
                }
            }
            pre += gsum;
        }
    }
    __syncthreads();
    PN2_STAMP(5);
    const int Cg = 3 + D;
    const int qpr = Cg >> 2;
    const bool rows = grouped != nullptr;
    const bool staged = rows && !overloaded;
    // the features of the candidates by local rank: loads issued now, landed after the first tests
    v4i fq[3][2];
    if (staged) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int lr = s2 * T2_THREADS + tid;
            const unsigned o = lr < ncand ? cidx[lr] * (unsigned)D * 4u : 0xfffffff0u;
#pragma unroll
            for (int u = 0; u < 3; ++u) fq[u][s2] = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)(o + 16u * (unsigned)u), 0, 0);
        }
    }
    bool features_staged = false;

    for (int round = 0; round < nround; ++round) {
        const int cbase = round * 2 * T2_THREADS;
        if (round > 0) {
            __syncthreads();
            if (tid == 0) misc[0] = 0u;
#pragma unroll
            for (int r = 0; r < 2; ++r) cq[r] = __builtin_amdgcn_raw_buffer_load_b96(crs, (cbase + tid) * 12, r * T2_THREADS * 12, 0);
            __syncthreads();
            scan_centroids(cbase);
            __syncthreads();
        }
        const int ncent = min((int)misc[0], T2_CL);
        if (ncent == 0) continue;
        // 16 lanes per centroid (32 per pass), or 8 (64 per pass) when the tile holds more than 32
        const int lpc = ncent > 32 ? 8 : 16;
        const int cpp = T2_THREADS / lpc;
        const int grp = tid / lpc, ll = tid % lpc;
        const int wpl = T2_BMW / lpc;                               // bitmap words per lane: 2 or 4
        unsigned *mybm = bm + grp * T2_BMW;
        unsigned short *oi = mIdx + grp * K;
        for (int pass0 = 0; pass0 < ncent; pass0 += cpp) {
            const int k = pass0 + grp;
            const bool have = k < ncent;
            for (int wdx = 0; wdx < wpl; ++wdx) mybm[ll * wpl + wdx] = 0u;
            const unsigned entry = have ? clist[k] : 0u;
            const int c = cbase + (int)(entry & 0x7fffu);
            const float4 cc = ccoord[have ? k : 0];
            const float cx = cc.x, cy = cc.y, cz = cc.z, cn = cc.w;
            const bool brute = (entry & 0x8000u) != 0u || overloaded;
            int n = 0;
            if (have && !brute) {
                // ---- 5. candidates of the 27 neighbouring local cells: nine runs of the cell-sorted array as one index space
                int ccx, ccy, ccz;
                cell_of(cx, cy, cz, ccx, ccy, ccz);
                const int x0 = max(ccx - 1, 0), x1 = min(ccx + 1, g0m);
                const int z0 = max(ccz - 1, 0), z1 = min(ccz + 1, g2m);
                const int y0 = max(ccy - 1, 0), y1 = min(ccy + 1, g1m);
                int off[9], cum[9];
                int tot = 0;
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int z = z0 + dz, y = y0 + dy;
                        const bool ok = z <= z1 && y <= y1;
                        const int base = ((ok ? z : z0) * g1 + (ok ? y : y0)) * g0;
                        const int rs = (int)hist[base + x0], re = (int)hist[base + x1 + 1];
                        off[dz * 3 + dy] = rs - tot;
                        tot += ok ? re - rs : 0;
                        cum[dz * 3 + dy] = tot;
                    }
                }
                for (int pos = ll; pos < tot; pos += lpc) {
                    int o = off[8];
#pragma unroll
                    for (int i = 7; i >= 0; --i) o = pos < cum[i] ? off[i] : o;
                    const int j = o + pos;
                    const float4 p = sorted4[j];
                    const float d = pn2::pair_sqdist(cx, cy, cz, cn, p.x, p.y, p.z, p.w);
                    if (!(d > r2)) {
                        const unsigned lr = srank[j];
                        atomicOr(&mybm[lr >> 5], 1u << (lr & 31u));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (have && !brute) {
                // ---- 6. the K lowest set bits = the K lowest indices: lane ll owns words wpl*ll .. wpl*ll + wpl-1
                unsigned w[4];
                int cntw = 0;
#pragma unroll
                for (int wdx = 0; wdx < 4; ++wdx) { w[wdx] = wdx < wpl ? mybm[ll * wpl + wdx] : 0u; cntw += __builtin_popcount(w[wdx]); }
                int inc = cntw;
                for (int o = 1; o < lpc; o <<= 1) {
                    const int t = __shfl_up(inc, o, lpc);
                    if (ll >= o) inc += t;
                }
                const int total = __shfl(inc, lpc - 1, lpc);
                int pos = inc - cntw;
#pragma unroll
                for (int wdx = 0; wdx < 4; ++wdx) {
                    unsigned bits = w[wdx];
                    while (bits && pos < K) {
                        const int bit = __builtin_ctz(bits);
                        bits &= bits - 1u;
                        oi[pos++] = (unsigned short)((ll * wpl + wdx) * 32 + bit);
                    }
                }
                n = min(total, K);
            } else if (have) {
                // every point of the block, 16 / 8 at a time in index order (rare: foreign centroids, crowded tiles)
                int cnt = 0;
                const unsigned long long gmask = (lpc == 16 ? 0xffffull : 0xffull) << (unsigned)((lane / lpc) * lpc);
                for (int base = 0; base < N && cnt < K; base += lpc) {
                    const int j = base + ll;
                    const v3i p = __builtin_amdgcn_raw_buffer_load_b96(xrs, min(j, N - 1) * 12, 0, 0);
                    const float x = __int_as_float(p.x), y = __int_as_float(p.y), z = __int_as_float(p.z);
                    const float d = pn2::pair_sqdist(cx, cy, cz, cn, x, y, z, pn2::norm3(x, y, z));
                    const bool hit = j < N && !(d > r2);
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit) & gmask;
                    const int rank = cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (hit && rank < K) oi[rank] = (unsigned short)j;
                    cnt += __builtin_popcountll(m);
                }
                n = min(cnt, K);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            PN2_STAMP(6);
            // ---- 7. idx [b, c, 0..K), padded with the first member (:104-106); an empty ball stores N (IndexError at :59)
            if (have) {
                int64_t *orow = idx + ((size_t)b * S + c) * K;
                for (int kk = ll; kk < K; kk += lpc) {
                    int a = N;
                    if (n > 0) { const unsigned e = oi[kk < n ? kk : 0]; a = (int)(brute ? e : cidx[e]); }
                    orow[kk] = (int64_t)a;
                }
                if (n == 0 && ll == 0 && err_count) atomicAdd(err_count, 1);
            }
            if (!rows) continue;
            // ---- 8. the staged features land in LDS (once)
            if (staged && !features_staged) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int lr = s2 * T2_THREADS + tid;
                    if (lr < ncand) {
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
                            const int f0 = 4 * u;
                            if (f0 + 0 < D) frow[lr * T2_RP + 3 + f0 + 0] = __int_as_float(fq[u][s2].x);
                            if (f0 + 1 < D) frow[lr * T2_RP + 3 + f0 + 1] = __int_as_float(fq[u][s2].y);
                            if (f0 + 2 < D) frow[lr * T2_RP + 3 + f0 + 2] = __int_as_float(fq[u][s2].z);
                            if (f0 + 3 < D) frow[lr * T2_RP + 3 + f0 + 3] = __int_as_float(fq[u][s2].w);
                        }
                    }
                }
                features_staged = true;
                __syncthreads();
            }
            PN2_STAMP(7);
            // ---- 9. grouped rows [xyz - centroid, feats]: K rows of qpr float4, contiguous; lane ll writes float4 ll + lpc i
            if (have) {
                const int E = K * qpr;
                const __amdgpu_buffer_rsrc_t grs =
                    __builtin_amdgcn_make_buffer_rsrc(grouped + ((size_t)b * S + c) * K * Cg, 0, E * 16, 0x00020000);
                const float gx = n > 0 ? cx : 0.0f, gy = n > 0 ? cy : 0.0f, gz = n > 0 ? cz : 0.0f;
                for (int e = ll; e < E; e += lpc) {
                    const int kk = qpr == 1 ? e : (int)__umulhi((unsigned)e, qpr_magic);
                    const int part = e - kk * qpr;
                    v4i v;
                    v.x = v.y = v.z = v.w = 0;
                    if (n > 0) {
                        const unsigned m = oi[kk < n ? kk : 0];
                        if (staged && !brute) {
                            const float4 f = reinterpret_cast<const float4 *>(frow + m * T2_RP)[part];
                            v.x = __float_as_int(part == 0 ? f.x - gx : f.x); v.y = __float_as_int(part == 0 ? f.y - gy : f.y);
                            v.z = __float_as_int(part == 0 ? f.z - gz : f.z); v.w = __float_as_int(f.w);
                        } else {
                            const unsigned j = brute ? m : cidx[m];
                            float col[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int colno = 4 * part + q;
                                if (colno < 3) {
                                    const float x = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(j * 12u) + colno * 4, 0, 0));
                                    col[q] = x - (colno == 0 ? gx : (colno == 1 ? gy : gz));
                                } else {
                                    col[q] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(prs, (int)((j * (unsigned)D + (unsigned)(colno - 3)) * 4u), 0, 0));
                                }
                            }
                            v.x = __float_as_int(col[0]); v.y = __float_as_int(col[1]); v.z = __float_as_int(col[2]); v.w = __float_as_int(col[3]);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(v, grs, e * 16, 0, 0);
                }
            }
            PN2_STAMP(8);
        }
    }
    PN2_STAMP_DRAIN();
    PN2_STAMP(9);
}

}  // namespace

namespace pn2 {
// rc PN2_ERR_UNSUPPORTED: nothing was launched
int launch_ball_query_tile2(const float *xyz, const float *new_xyz, const float *points, int B, int N, int S, int K, int D, int ldg,
                            float r2, int64_t *idx, float *grouped, int32_t *err_count, hipStream_t stream)
{
    if (K > T2_MAXK || N > T2_THREADS * T2_PPT || S > 32767 * 2) return PN2_ERR_UNSUPPORTED;
    const int Cg = 3 + D;
    const bool fused = grouped && ldg == Cg && (Cg & 3) == 0 && D >= 1 && D <= 9 && (reinterpret_cast<uintptr_t>(grouped) & 15) == 0;
    if (grouped && !fused) return PN2_ERR_UNSUPPORTED;
    int tmax = 1;
    while (tmax < 64 && tmax * 2 * 32 <= S) tmax *= 2;
    const long long nwg = (long long)B * tmax;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    const T2Smem L = t2_layout(K);
    static pn2::PerDevice memo;
    if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(ball_tile2_kernel), L.total, memo)) return e;
    const int qpr = Cg >> 2;
    const unsigned magic = qpr > 1 ? (unsigned)((1ULL << 32) / (unsigned)qpr) + 1u : 0u;
    hipLaunchKernelGGL(ball_tile2_kernel, dim3((unsigned)nwg), dim3(T2_THREADS), (size_t)L.total, stream, xyz, new_xyz, points, N, S, K, D,
                       r2, tmax, magic, idx, fused ? grouped : nullptr, err_count);
    return PN2_LAUNCH_RC();
}
}  // namespace pn2
