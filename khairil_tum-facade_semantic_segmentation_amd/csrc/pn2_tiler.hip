// Sliding-window tiler of whole-scene inference on the device.
//
// Reference: TestCustomDataset.__getitem__ (sem_seg_testing.py:182-254) walks a grid of block_size windows at `stride`
// over the scene; per window an np.where over the ENTIRE scene (:202-203), the points topped up to a multiple of
// block_points with a random sample of themselves (:205-211: without replacement when the top-up is at most the
// population, else with), one shuffle (:212), then the block features [x - cx, y - cy, z, xyz / scene_max, extras]
// (:214-239), labels, label weights and point indices.  At GPU speeds that host loop, not the network, bounds test-time
// throughput (measured: 107 blocks/s against 12-21 k blocks/s of the network).
//
// Here the scene lives on the device bucketed into the 2-D grid of the training sampler (pn2_sampler.hip):
//   pn2_tile_windows   one workgroup per window: the window's points (closed window, double compares like numpy's) are
//                      counted (members == NULL) or written as a list in a deterministic order (grid rows in order,
//                      cell order inside a row);
//   pn2_tile_fill      one thread per output slot: "members + random top-up, shuffled" is two keyed pseudo-random
//                      PERMUTATIONS (a balanced Feistel network over the next even power of two, cycle-walked into the
//                      range): slot p of a window of `size` slots takes pooled element perm_a(p); pooled elements below
//                      the population are the members themselves, the others the first `size - population` elements of
//                      perm_b over the members (a uniformly random subset without replacement), or independent uniform
//                      draws when the top-up exceeds the population.  No sort, no atomics, every slot independent;
//                      the same seed gives the same tiling.  With `srcpos` the slot -> member map is GIVEN: the host
//                      replays numpy's choice() / shuffle() stream from the window populations alone, and the blocks
//                      equal the reference's bit for bit (tests/golden/scene_tiler.npz).
// Arithmetic that defines a block (window bounds, centring, xyz / scene_max) is double like the reference's numpy code
// and rounded to float once (the loop's torch.Tensor(...) of the float64 batch, localfunctions.py:393-395).
#include "pn2_common.h"

namespace {

constexpr int TW_THREADS = 256;

__device__ __forceinline__ unsigned long long tmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// keyed permutation of [0, n): four Feistel rounds on the next even number of bits, cycle-walked (a value that leaves
// the range is permuted again: the walk visits each value of the cycle once, so the map stays a bijection on [0, n))
__device__ __forceinline__ unsigned perm_u32(unsigned x, unsigned n, unsigned long long key)
{
    if (n <= 1u) return 0u;
    int bits = 32 - __builtin_clz(n - 1u);
    if (bits & 1) ++bits;
    if (bits < 2) bits = 2;
    const int hb = bits >> 1;
    const unsigned mask = (1u << hb) - 1u;
    do {
        unsigned l = x >> hb, r = x & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const unsigned f = (unsigned)(tmix64(key + (unsigned long long)round * 0xD6E8FEB86659FD93ull + r) >> 20) & mask;
            const unsigned t = l ^ f;
            l = r;
            r = t;
        }
        x = (l << hb) | r;
    } while (x >= n);
    return x;
}

struct WindowArgs {
    const double *xyz;            // [P][3]
    const int *order;             // [P] point indices sorted by cell (row-major cells, ascending index inside a cell)
    const int *cell_start;        // [nx*ny + 1]
    double x0, y0, cell;
    int nx, ny;
    const double *win;            // [W][4]: xmin, xmax, ymin, ymax of the closed window (padding included)
    const long long *member_off;  // [W] start of each window's list in `members` (members pass)
    int *counts;                  // [W]
    int *members;                 // NULL: count only
};

__global__ __launch_bounds__(TW_THREADS) void tile_windows_kernel(WindowArgs a)
{
    __shared__ int wsum[TW_THREADS / 64];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double xmin = a.win[w * 4 + 0], xmax = a.win[w * 4 + 1], ymin = a.win[w * 4 + 2], ymax = a.win[w * 4 + 3];
    const int i0 = max((int)floor((xmin - a.x0) / a.cell), 0), i1 = min((int)floor((xmax - a.x0) / a.cell), a.nx - 1);
    const int j0 = max((int)floor((ymin - a.y0) / a.cell), 0), j1 = min((int)floor((ymax - a.y0) / a.cell), a.ny - 1);
    int *out = a.members ? a.members + a.member_off[w] : nullptr;
    int total = 0;                                    // uniform: members written so far
    for (int j = j0; j <= j1 && i1 >= i0; ++j) {
        const int e0 = a.cell_start[j * a.nx + i0], e1 = a.cell_start[j * a.nx + i1 + 1];
        for (int base = e0; base < e1; base += TW_THREADS) {
            const int e = base + tid;
            int p = -1;
            bool in = false;
            if (e < e1) {
                p = a.order[e];
                const double x = a.xyz[(size_t)p * 3], y = a.xyz[(size_t)p * 3 + 1];
                in = x >= xmin && x <= xmax && y >= ymin && y <= ymax;            // :202-203
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
            if (lane == 0) wsum[wave] = __builtin_popcountll(m);
            __syncthreads();
            int pre = 0, all = 0;
#pragma unroll
            for (int k = 0; k < TW_THREADS / 64; ++k) {
                const int c = wsum[k];
                pre += k < wave ? c : 0;
                all += c;
            }
            if (in && out) out[total + pre + pn2::mbcnt(m)] = p;
            total += all;
            __syncthreads();
        }
    }
    if (tid == 0 && !a.members) a.counts[w] = total;
}

struct FillArgs {
    const double *xyz;            // [P][3]
    const float *extra;           // [E][P], already scaled, nullable
    const long long *labels;      // [P]
    const float *labelweights;    // [classes], nullable (weight 1)
    int P, E, num_classes;
    double max_x, max_y, max_z;
    const int *members;
    const long long *member_off;  // [W]
    const int *counts;            // [W]
    const double *centre;         // [W][2]
    const long long *block_off;   // [W + 1], in blocks
    int W, block_points;
    const int *srcpos;            // nullable: [slots] position in the window's member list (exact replay)
    unsigned long long seed;
    float *data;                  // [blocks][block_points][6 + E]
    long long *out_labels;
    float *out_weight;
    long long *out_index;
    long long slots;
};

__global__ __launch_bounds__(256) void tile_fill_kernel(FillArgs a)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= a.slots) return;
    const long long blk = g / a.block_points;
    int lo = 0, hi = a.W;                             // the window whose block range holds blk
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.block_off[mid] <= blk) lo = mid; else hi = mid;
    }
    const int w = lo;
    const unsigned count = (unsigned)a.counts[w];
    const unsigned size = (unsigned)((a.block_off[w + 1] - a.block_off[w]) * a.block_points);
    const unsigned p = (unsigned)(g - a.block_off[w] * a.block_points);
    unsigned m;
    if (a.srcpos) {
        m = (unsigned)a.srcpos[g];
    } else {
        const unsigned long long kw = tmix64(a.seed ^ ((unsigned long long)(unsigned)w << 32));
        const unsigned q = perm_u32(p, size, kw);                                                   // the shuffle (:212)
        if (q < count) {
            m = q;
        } else {
            const unsigned r = q - count, fill = size - count;
            if (fill <= count) m = perm_u32(r, count, kw ^ 0xA5A5A5A55A5A5A5Aull);                   // :209-210, without replacement
            else m = (unsigned)(((tmix64(kw + 0x1234567ull + r) >> 32) * (unsigned long long)count) >> 32);   // with replacement
        }
    }
    const int pt = a.members[a.member_off[w] + m];
    const double x = a.xyz[(size_t)pt * 3], y = a.xyz[(size_t)pt * 3 + 1], z = a.xyz[(size_t)pt * 3 + 2];
    const int F = 6 + a.E;
    float *o = a.data + (size_t)g * F;
    o[0] = (float)(x - a.centre[w * 2 + 0]);          // :220-221
    o[1] = (float)(y - a.centre[w * 2 + 1]);
    o[2] = (float)z;
    o[3] = (float)(x / a.max_x);                      // :217-219
    o[4] = (float)(y / a.max_y);
    o[5] = (float)(z / a.max_z);
    for (int k = 0; k < a.E; ++k) o[6 + k] = a.extra[(size_t)k * a.P + pt];       // :227-239
    const long long lab = a.labels[pt];
    a.out_labels[g] = lab;
    a.out_weight[g] = a.labelweights ? ((lab >= 0 && lab < a.num_classes) ? a.labelweights[lab] : 0.0f) : 1.0f;   // :225
    a.out_index[g] = pt;
}

}  // namespace

PN2_EXPORT int pn2_tile_windows(const double *xyz, const int *order, const int *cell_start, double x0, double y0, double cell, int nx,
                                int ny, const double *windows, int W, const long long *member_off, int *counts, int *members,
                                pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz); PN2_REQUIRE_PTR(order); PN2_REQUIRE_PTR(cell_start); PN2_REQUIRE_PTR(windows);
    if (W < 0 || nx <= 0 || ny <= 0 || !(cell > 0.0)) return PN2_ERR_SHAPE;
    if (members == nullptr) PN2_REQUIRE_PTR(counts); else PN2_REQUIRE_PTR(member_off);
    if (W == 0) return PN2_OK;
    WindowArgs a;
    a.xyz = xyz; a.order = order; a.cell_start = cell_start; a.x0 = x0; a.y0 = y0; a.cell = cell; a.nx = nx; a.ny = ny;
    a.win = windows; a.member_off = member_off; a.counts = counts; a.members = members;
    hipLaunchKernelGGL(tile_windows_kernel, dim3(W), dim3(TW_THREADS), 0, static_cast<hipStream_t>(stream_), a);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_tile_fill(const double *xyz, const float *extra, const long long *labels, const float *labelweights, int P, int E,
                             int num_classes, const double *coord_max, const int *members, const long long *member_off,
                             const int *counts, const double *centre, const long long *block_off, int W, long long blocks,
                             int block_points, const int *srcpos, unsigned long long seed, float *data, long long *out_labels,
                             float *out_weight, long long *out_index, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz); PN2_REQUIRE_PTR(labels); PN2_REQUIRE_PTR(coord_max); PN2_REQUIRE_PTR(members); PN2_REQUIRE_PTR(member_off);
    PN2_REQUIRE_PTR(counts); PN2_REQUIRE_PTR(centre); PN2_REQUIRE_PTR(block_off); PN2_REQUIRE_PTR(data); PN2_REQUIRE_PTR(out_labels);
    PN2_REQUIRE_PTR(out_weight); PN2_REQUIRE_PTR(out_index);
    if (P <= 0 || E < 0 || W < 0 || blocks < 0 || block_points <= 0 || num_classes < 0) return PN2_ERR_SHAPE;
    if (E > 0 && extra == nullptr) return PN2_ERR_NULL;
    if (W == 0 || blocks == 0) return PN2_OK;
    FillArgs a;
    a.xyz = xyz; a.extra = extra; a.labels = labels; a.labelweights = labelweights; a.P = P; a.E = E; a.num_classes = num_classes;
    a.max_x = coord_max[0]; a.max_y = coord_max[1]; a.max_z = coord_max[2];
    a.members = members; a.member_off = member_off; a.counts = counts; a.centre = centre; a.block_off = block_off; a.W = W;
    a.block_points = block_points; a.srcpos = srcpos; a.seed = seed; a.data = data; a.out_labels = out_labels;
    a.out_weight = out_weight; a.out_index = out_index;
    a.slots = blocks * (long long)block_points;
    const long long nwg = (a.slots + 255) / 256;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(tile_fill_kernel, dim3((unsigned)nwg), dim3(256), 0, static_cast<hipStream_t>(stream_), a);
    return PN2_LAUNCH_RC();
}
