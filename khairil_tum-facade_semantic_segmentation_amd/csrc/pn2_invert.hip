// Transposed form of the two gather operators' backward passes.
//
// index_points / grouping (reference models/pointnet2_utils.py:43-60, :127-132) and the 3-NN
// interpolation (:296-303) read rows through an index; their autograd is a scatter-add of the incoming
// gradient rows into the indexed rows.  A scatter-add needs float atomics (arbitrary summation order,
// a zero-filled target).  The index tables depend on the coordinates only, so they can be transposed
// once per batch, off the critical path, into "which entries point at row j" lists
// (pn2_invert_index); the backward is then a plain gather-sum per row (pn2_gather_sum): no atomics,
// no zero fill, and a fixed summation order (entries ascending) -> run-to-run identical gradients.
#include "pn2_common.h"

namespace {

constexpr int INV_THREADS = 1024;
constexpr int INV_MAX_KEYS = 8192;            // keys per batch handled in LDS
constexpr int INV_MAX_ENTRIES = 24576;        // entries per batch handled in LDS (96 KB of int32)

// 16-byte loads from rows that are only 4-byte aligned (gradient columns behind the 3 xyz columns)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// One workgroup per batch: histogram of the keys, exclusive scan, scatter of the entry ids, then every
// key's list is put in ascending order (lists are short: 3N/S resp. SK/N entries on average).
__device__ __forceinline__ void invert_index_body(const int64_t *__restrict__ idx, long long E, int Nkeys,
                                                  int32_t *__restrict__ offsets, int32_t *__restrict__ entries)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *cnt = reinterpret_cast<int *>(smem);                      // [Nkeys + 1] counts -> starts
    int *fill = cnt + (Nkeys + 1);                                 // [Nkeys] next free slot
    int *list = fill + Nkeys;                                      // [E]
    __shared__ int wsum[INV_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t *bi = idx + (size_t)b * E;
    for (int i = tid; i <= Nkeys; i += INV_THREADS) cnt[i] = 0;
    __syncthreads();
    // four keys per pass, loaded before the first is counted: the workgroup is alone on its CU and a pass is one memory
    // round trip whatever it holds
    for (long long e0 = tid; e0 < E; e0 += 4 * INV_THREADS) {
        int64_t k4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k4[u] = bi[min(e0 + u * INV_THREADS, E - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + u * INV_THREADS < E && k4[u] >= 0 && k4[u] < Nkeys) atomicAdd(&cnt[(int)k4[u]], 1);
    }
    __syncthreads();
    // exclusive scan over Nkeys counts: chunk per thread, wave scan, wave offsets
    const int per = (Nkeys + INV_THREADS - 1) / INV_THREADS;
    const int k0 = tid * per;
    int s = 0;
    for (int i = 0; i < per; ++i)
        if (k0 + i < Nkeys) s += cnt[k0 + i];
    int inc = s;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    int run = off + inc - s;
    for (int i = 0; i < per; ++i)
        if (k0 + i < Nkeys) {
            const int c = cnt[k0 + i];
            cnt[k0 + i] = run;
            fill[k0 + i] = run;
            run += c;
        }
    if (tid == INV_THREADS - 1) cnt[Nkeys] = run;                  // total number of valid entries
    __syncthreads();
    // entries are appended batch by batch in ascending order (a barrier between batches), so a list is
    // ascending except among entries of one batch; the insertion sort below only repairs those
    int64_t knext = bi[min((long long)tid, E - 1)];                // the next batch's key travels during this batch
    for (long long e0 = 0; e0 < E; e0 += INV_THREADS) {
        const long long e = e0 + tid;
        const int64_t k = knext;
        knext = bi[min(e + INV_THREADS, E - 1)];
        if (e < E && k >= 0 && k < Nkeys) list[atomicAdd(&fill[(int)k], 1)] = (int)e;
        __syncthreads();
    }
    // ascending order inside every list (insertion sort, one thread per key)
    for (int k = tid; k < Nkeys; k += INV_THREADS) {
        const int a = cnt[k], z = cnt[k + 1];
        for (int i = a + 1; i < z; ++i) {
            const int v = list[i];
            int j = i - 1;
            while (j >= a && list[j] > v) { list[j + 1] = list[j]; --j; }
            list[j + 1] = v;
        }
    }
    __syncthreads();
    int32_t *bo = offsets + (size_t)b * (Nkeys + 1);
    int32_t *be = entries + (size_t)b * E;
    for (int i = tid; i <= Nkeys; i += INV_THREADS) bo[i] = cnt[i];
    const int total = cnt[Nkeys];
    for (int e = tid; e < E; e += INV_THREADS) be[e] = e < total ? list[e] : -1;
}

__global__ __launch_bounds__(INV_THREADS) void invert_index_kernel(const int64_t *__restrict__ idx, long long E, int Nkeys,
                                                                  int32_t *__restrict__ offsets, int32_t *__restrict__ entries)
{
    invert_index_body(idx, E, Nkeys, offsets, entries);
}

// Several tables in one launch (blockIdx.y = table): the four interpolation levels of the network are independent once their
// 3-NN indices exist, and one-workgroup-per-block launches in a row only add up their latencies (15-27 us each).
constexpr int INV_MANY_MAX = 8;
struct InvertMany {
    const int64_t *idx[INV_MANY_MAX];
    int32_t *offsets[INV_MANY_MAX], *entries[INV_MANY_MAX];
    long long E[INV_MANY_MAX];
    int Nkeys[INV_MANY_MAX];
};
__global__ __launch_bounds__(INV_THREADS) void invert_index_many_kernel(InvertMany m)
{
    const int j = blockIdx.y;
    invert_index_body(m.idx[j], m.E[j], m.Nkeys[j], m.offsets[j], m.entries[j]);
}

// out[b][key][c] = addend[b][key][c] + sum over the key's entries e of w[b][e] * src[b][e / ediv][col0 + c].
// LPR lanes per (b, key) row (a lane owns 4 consecutive columns per pass of 4*LPR columns) times SL slices: slice s sums the
// entries a + s, a + s + SL, ... (ascending, four in flight), the slices are combined in order s = 0, 1, ... -- a fixed
// summation order whatever the launch.  The per-key lists have a heavy tail (grouping: 8 entries on average, up to ~100:
// a point of a dense region sits in many balls); with one lane group walking a whole list, a wave waits for its longest
// list (the form with SL = 1: 88 us for the three grouping levels against 71 us of float atomics).  64 / (LPR SL) rows per wave.
template <int LPR, int SL>
__global__ __launch_bounds__(256) void gather_sum_kernel(const float *__restrict__ src, long long rows_src, int lds, int col0,
                                                         const int32_t *__restrict__ offsets, const int32_t *__restrict__ entries,
                                                         const float *__restrict__ weight, long long E, int ediv, int B, int Nkeys,
                                                         int D, const float *__restrict__ addend, float *__restrict__ out)
{
    PN2_MAIN_BRANCH_PRIORITY();
    constexpr int LPK = LPR * SL;                                  // lanes per key row
    constexpr int RPB = 256 / LPK;                                 // rows per workgroup
    static_assert(LPK <= 64, "a key row stays inside one wave");
    const int sub = threadIdx.x % LPR, sl = (threadIdx.x / LPR) % SL;
    long long row = (long long)blockIdx.x * RPB + threadIdx.x / LPK;
    const bool live = row < (long long)B * Nkeys;
    if (!live) row = (long long)B * Nkeys - 1;                    // (the shuffles below need every lane of the wave)
    const int b = (int)(row / Nkeys), key = (int)(row - (long long)b * Nkeys);
    const int32_t *bo = offsets + (size_t)b * (Nkeys + 1);
    const int a = bo[key], z = bo[key + 1];
    const int32_t *be = entries + (size_t)b * E;
    const float *bw = weight ? weight + (size_t)b * E : nullptr;
    const float *bs = src + (size_t)b * rows_src * lds + col0;
    // blockIdx.y = block of 4 * LPR columns (wide rows: D up to 768 at the deep levels).  As a loop inside the workgroup the
    // column blocks ran one after the other, each with its own entry -> weight / row round trips.
    {
        const int c0 = (int)blockIdx.y * 4 * LPR;
        const int c = c0 + sub * 4;
        const bool col_ok = c < D;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col_ok) {
            int i = a + sl;
            for (; i + 3 * SL < z; i += 4 * SL) {
                int e[4];
                float w[4];
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    e[u] = be[i + u * SL];
                    w[u] = bw ? bw[e[u]] : 1.0f;
                    const f32x4u t = *reinterpret_cast<const f32x4u *>(bs + (size_t)(e[u] / ediv) * lds + c);
                    v[u] = make_float4(t.x, t.y, t.z, t.w);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc.x += w[u] * v[u].x; acc.y += w[u] * v[u].y; acc.z += w[u] * v[u].z; acc.w += w[u] * v[u].w;
                }
            }
            for (; i < z; i += SL) {
                const int e = be[i];
                const float w = bw ? bw[e] : 1.0f;
                const f32x4u v = *reinterpret_cast<const f32x4u *>(bs + (size_t)(e / ediv) * lds + c);
                acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
            }
        }
        if (SL > 1) {                                              // slice 0 collects the others in order
#pragma unroll
            for (int s2 = 1; s2 < SL; ++s2) {
                const float ox = __shfl_down(acc.x, LPR * s2, 64), oy = __shfl_down(acc.y, LPR * s2, 64);
                const float oz = __shfl_down(acc.z, LPR * s2, 64), ow = __shfl_down(acc.w, LPR * s2, 64);
                if (sl == 0) { acc.x += ox; acc.y += oy; acc.z += oz; acc.w += ow; }
            }
        }
        if (live && col_ok && sl == 0) {
            if (addend) {
                const float4 ad = *reinterpret_cast<const float4 *>(addend + (size_t)row * D + c);
                acc.x += ad.x; acc.y += ad.y; acc.z += ad.z; acc.w += ad.w;
            }
            pn2::store_rows4(out, (size_t)row * D + c, acc, (size_t)B * Nkeys * D * sizeof(float));
        }
    }
}

// any width / alignment: one wave per row, one column per lane and pass
__global__ __launch_bounds__(256) void gather_sum_scalar_kernel(const float *__restrict__ src, long long rows_src, int lds, int col0,
                                                                const int32_t *__restrict__ offsets,
                                                                const int32_t *__restrict__ entries, const float *__restrict__ weight,
                                                                long long E, int ediv, int B, int Nkeys, int D,
                                                                const float *__restrict__ addend, float *__restrict__ out)
{
    PN2_MAIN_BRANCH_PRIORITY();
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * Nkeys) return;
    const int b = (int)(row / Nkeys), key = (int)(row - (long long)b * Nkeys);
    const int32_t *bo = offsets + (size_t)b * (Nkeys + 1);
    const int a = bo[key], z = bo[key + 1];
    const int32_t *be = entries + (size_t)b * E;
    const float *bw = weight ? weight + (size_t)b * E : nullptr;
    const float *bs = src + (size_t)b * rows_src * lds + col0;
    float *orow = out + (size_t)row * D;
    for (int c = lane; c < D; c += 64) {
        float acc = 0.f;
        for (int i = a; i < z; ++i) {
            const int e = be[i];
            acc += (bw ? bw[e] : 1.0f) * bs[(size_t)(e / ediv) * lds + c];
        }
        orow[c] = addend ? acc + addend[(size_t)row * D + c] : acc;
    }
}

}  // namespace

PN2_EXPORT int pn2_invert_index(const int64_t *idx, int B, long long E, int Nkeys, int32_t *offsets, int32_t *entries,
                                pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(idx); PN2_REQUIRE_PTR(offsets); PN2_REQUIRE_PTR(entries);
    if (B < 0 || E <= 0 || Nkeys <= 0) return PN2_ERR_SHAPE;
    if (E > INV_MAX_ENTRIES || Nkeys > INV_MAX_KEYS) return PN2_ERR_UNSUPPORTED;
    if (B == 0) return PN2_OK;
    const size_t lds = ((size_t)(2 * Nkeys + 1) + (size_t)E) * sizeof(int);
    if (lds > 150 * 1024) return PN2_ERR_UNSUPPORTED;
    static pn2::PerDevice lds_memo;
    if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(invert_index_kernel), 150 * 1024, lds_memo)) return e;
    hipLaunchKernelGGL(invert_index_kernel, dim3((unsigned)B), dim3(INV_THREADS), lds, static_cast<hipStream_t>(stream_), idx, E,
                       Nkeys, offsets, entries);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_invert_index_many(int n, const int64_t *const *idx, int B, const long long *E, const int *Nkeys,
                                     int32_t *const *offsets, int32_t *const *entries, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(idx); PN2_REQUIRE_PTR(E); PN2_REQUIRE_PTR(Nkeys); PN2_REQUIRE_PTR(offsets); PN2_REQUIRE_PTR(entries);
    if (n <= 0 || n > INV_MANY_MAX || B < 0) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    InvertMany m;
    size_t lds = 0;
    for (int j = 0; j < INV_MANY_MAX; ++j) {
        const int i = j < n ? j : n - 1;
        if (!idx[i] || !offsets[i] || !entries[i]) return PN2_ERR_NULL;
        if (E[i] <= 0 || Nkeys[i] <= 0) return PN2_ERR_SHAPE;
        if (E[i] > INV_MAX_ENTRIES || Nkeys[i] > INV_MAX_KEYS) return PN2_ERR_UNSUPPORTED;
        m.idx[j] = idx[i]; m.offsets[j] = offsets[i]; m.entries[j] = entries[i]; m.E[j] = E[i]; m.Nkeys[j] = Nkeys[i];
        const size_t need = ((size_t)(2 * Nkeys[i] + 1) + (size_t)E[i]) * sizeof(int);
        lds = need > lds ? need : lds;
    }
    if (lds > 150 * 1024) return PN2_ERR_UNSUPPORTED;
    static pn2::PerDevice lds_memo;
    if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(invert_index_many_kernel), 150 * 1024, lds_memo)) return e;
    hipLaunchKernelGGL(invert_index_many_kernel, dim3((unsigned)B, (unsigned)n), dim3(INV_THREADS), lds, static_cast<hipStream_t>(stream_), m);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_gather_sum(const float *src, long long rows_src, int lds, int col0, const int32_t *offsets,
                              const int32_t *entries, const float *weight, long long E, int ediv, int B, int Nkeys, int D,
                              float *out, pn2_stream_t stream_)
{
    return pn2_gather_sum_add(src, rows_src, lds, col0, offsets, entries, weight, E, ediv, B, Nkeys, D, nullptr, out, stream_);
}

PN2_EXPORT int pn2_gather_sum_add(const float *src, long long rows_src, int lds, int col0, const int32_t *offsets,
                                  const int32_t *entries, const float *weight, long long E, int ediv, int B, int Nkeys, int D,
                                  const float *addend, float *out, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(src); PN2_REQUIRE_PTR(offsets); PN2_REQUIRE_PTR(entries); PN2_REQUIRE_PTR(out);
    if (B < 0 || E <= 0 || Nkeys <= 0 || D <= 0 || ediv <= 0 || rows_src <= 0 || col0 < 0 || lds < col0 + D) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    const long long rows = (long long)B * Nkeys;
    const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && ((reinterpret_cast<uintptr_t>(addend) & 15) == 0);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int lpr = !vec4 ? 0 : (D <= 32 ? 8 : (D <= 64 ? 16 : (D <= 128 ? 32 : 64)));
    // slices per key row: lists of four entries and more on average are split (up to 64 lanes per row)
    int sl = (lpr && E / Nkeys >= 4 && pn2::tune_get("gather_slices", 1)) ? 64 / lpr : 1;
    if (sl > 4) sl = 4;
    const long long rpb = lpr ? 256 / (lpr * sl) : 4;
    const long long blocks = (rows + rpb - 1) / rpb;
    if (blocks > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    const unsigned colblocks = lpr ? (unsigned)((D + 4 * lpr - 1) / (4 * lpr)) : 1u;
#define PN2_GS(L, S) hipLaunchKernelGGL((gather_sum_kernel<L, S>), dim3((unsigned)blocks, colblocks), dim3(256), 0, stream, src, rows_src, lds, col0, \
                                        offsets, entries, weight, E, ediv, B, Nkeys, D, addend, out)
    if (lpr == 8) { if (sl == 4) PN2_GS(8, 4); else PN2_GS(8, 1); }
    else if (lpr == 16) { if (sl == 4) PN2_GS(16, 4); else PN2_GS(16, 1); }
    else if (lpr == 32) { if (sl == 2) PN2_GS(32, 2); else PN2_GS(32, 1); }
    else if (lpr == 64) PN2_GS(64, 1);
    else
        hipLaunchKernelGGL(gather_sum_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, rows_src, lds, col0, offsets,
                           entries, weight, E, ediv, B, Nkeys, D, addend, out);
#undef PN2_GS
    return PN2_LAUNCH_RC();
}
