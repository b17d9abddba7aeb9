// Training block sampler on the device.
//
// Reference: TrainCustomDataset.__getitem__ (sem_seg_training.py:200-259) draws a random point of the scene as the
// centre of a block_size x block_size column, re-draws until the column holds more than 1024 points (np.where over
// the ENTIRE scene per attempt), samples num_point of them (without replacement when there are enough, with
// replacement otherwise), centres x / y on the block and appends xyz / room_max and the extra features.  Eight
// DataLoader workers of that cannot feed one MI355X (~5 700 blocks/s at the measured step time).
//
// Here the scene lives on the device, bucketed once into a 2-D grid (cell-sorted index + cell starts, built by the
// host side with a sort), and one workgroup draws one block: a window touches only the cell rows it overlaps; "a
// uniformly random num_point-subset in uniformly random order" is taken as the num_point smallest of independent
// random keys -- keys below a threshold sized for ~num_point + 6 sigma survivors are collected in LDS and sorted
// there -- and the with-replacement case indexes the (sorted, hence reproducible) candidate list with independent
// uniform draws.  All randomness is a counter-based hash of (seed, block, attempt, item): the same seed gives the
// same blocks.  Arithmetic that defines the block (window bounds, centring, xyz / room_max) is done in double like
// the reference's numpy code and rounded to float once (the loop's `points.float()`, localfunctions.py:208).
#include "pn2_common.h"

namespace {

constexpr int SB_THREADS = 1024;
constexpr int SB_CAP = 8192;                       // LDS list capacity (64 KB of u64)

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ unsigned rnd32(unsigned long long seed, unsigned a, unsigned b, unsigned c)
{
    return (unsigned)(mix64(mix64(seed ^ ((unsigned long long)a << 32 | b)) + c) >> 32);
}

struct SamplerArgs {
    const double *xyz;            // [P][3] original order
    const int *order;             // [P] point indices sorted by cell (row-major cells, ascending index inside a cell)
    const int *cell_start;        // [nx*ny + 1]
    const float *extra;           // [E][P] extra feature columns, already scaled (rgb / 255), nullable
    const long long *labels;      // [P]
    double x0, y0, cell;          // grid origin and cell size
    int nx, ny, P, E;
    double half;                  // block_size / 2
    double max_x, max_y, max_z;   // room_coord_max
    int num_point, min_points, max_attempts;
    unsigned long long seed;
    float *feats;                 // [B][num_point][6 + E]
    long long *out_labels;        // [B][num_point]
    int *info;                    // [B][4]: centre index, points in the window, attempts used, 0 ok / 1 gave up
    int *sel;                     // [B][num_point] chosen point indices (nullable)
};

// one workgroup draws output block `b` from the scene described by `a`
__device__ __forceinline__ void sample_one_block(const SamplerArgs &a, const int b, unsigned long long *list, int &s_cnt, int &s_n)
{
    const int tid = threadIdx.x;
    int attempt = 0, cnt = 0, centre = 0;
    double cx = 0.0, cy = 0.0, xmin = 0, xmax = 0, ymin = 0, ymax = 0;
    int i0 = 0, i1 = -1, j0 = 0, j1 = -1;
    auto inside = [&](int p) {
        const double x = a.xyz[(size_t)p * 3], y = a.xyz[(size_t)p * 3 + 1];
        return x >= xmin && x <= xmax && y >= ymin && y <= ymax;                  // :211-214, closed window
    };
    for (;; ++attempt) {
        if (attempt >= a.max_attempts) break;
        centre = (int)(((unsigned long long)rnd32(a.seed, (unsigned)b, (unsigned)attempt, 0u) * (unsigned long long)a.P) >> 32);   // :207
        cx = a.xyz[(size_t)centre * 3];
        cy = a.xyz[(size_t)centre * 3 + 1];
        xmin = cx - a.half; xmax = cx + a.half; ymin = cy - a.half; ymax = cy + a.half;                                       // :208-209
        i0 = max((int)floor((xmin - a.x0) / a.cell), 0); i1 = min((int)floor((xmax - a.x0) / a.cell), a.nx - 1);
        j0 = max((int)floor((ymin - a.y0) / a.cell), 0); j1 = min((int)floor((ymax - a.y0) / a.cell), a.ny - 1);
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int mine = 0;
        for (int j = j0; j <= j1; ++j) {
            const int e0 = a.cell_start[j * a.nx + i0], e1 = a.cell_start[j * a.nx + i1 + 1];
            for (int e = e0 + tid; e < e1; e += SB_THREADS) mine += inside(a.order[e]) ? 1 : 0;
        }
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        if ((tid & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
        __syncthreads();
        cnt = s_cnt;
        __syncthreads();
        if (cnt > a.min_points) break;                                                                                        // :215-216
    }
    const bool gave_up = !(cnt > a.min_points);
    if (tid == 0) { a.info[b * 4 + 0] = centre; a.info[b * 4 + 1] = cnt; a.info[b * 4 + 2] = attempt + (gave_up ? 0 : 1); a.info[b * 4 + 3] = gave_up ? 1 : 0; }
    const int F = 6 + a.E;
    if (gave_up) {                                    // no column with enough points within max_attempts: zero block, flagged
        for (int i = tid; i < a.num_point * F; i += SB_THREADS) a.feats[(size_t)b * a.num_point * F + i] = 0.0f;
        for (int i = tid; i < a.num_point; i += SB_THREADS) {
            a.out_labels[(size_t)b * a.num_point + i] = 0;
            if (a.sel) a.sel[(size_t)b * a.num_point + i] = -1;
        }
        return;
    }
    const bool without = cnt >= a.num_point;          // :218-221
    // collect: the candidates whose random key falls below the threshold (without replacement), or all of them
    double frac = without ? ((double)a.num_point + 6.0 * sqrt((double)a.num_point) + 32.0) / (double)cnt : 1.0;
    int n = 0;
    for (int round = 0; round < 12; ++round) {
        const unsigned thr = frac >= 1.0 ? 0xffffffffu : (unsigned)(frac * 4294967296.0);
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (int j = j0; j <= j1; ++j) {
            const int e0 = a.cell_start[j * a.nx + i0], e1 = a.cell_start[j * a.nx + i1 + 1];
            for (int e = e0 + tid; e < e1; e += SB_THREADS) {
                const int p = a.order[e];
                if (!inside(p)) continue;
                const unsigned key = without ? rnd32(a.seed, (unsigned)b, (unsigned)attempt, 0x40000000u + (unsigned)p) : 0u;
                if (without && key > thr) continue;
                const int pos = atomicAdd(&s_n, 1);
                if (pos < SB_CAP) list[pos] = ((unsigned long long)key << 32) | (unsigned)p;
            }
        }
        __syncthreads();
        n = s_n;
        __syncthreads();
        if (!without || (n >= a.num_point && n <= SB_CAP)) break;
        frac = n < a.num_point ? frac * 1.5 : frac * 0.75;      // a >6 sigma event (or a tiny window population): redo
    }
    if (n > SB_CAP) n = SB_CAP;
    // bitonic sort of the list by (key, index): random order for the subset, ascending index for the candidate list
    int m = 1;
    while (m < n) m <<= 1;
    for (int i = n + tid; i < m; i += SB_THREADS) list[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m; i += SB_THREADS) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long x = list[i], y = list[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { list[i] = y; list[l] = x; }
                }
            }
            __syncthreads();
        }
    }
    // emit the block
    for (int i = tid; i < a.num_point; i += SB_THREADS) {
        int p;
        if (without) {
            p = (int)(unsigned)list[i < n ? i : n - 1];
        } else {
            const unsigned r = rnd32(a.seed, (unsigned)b, (unsigned)attempt, 0x80000000u + (unsigned)i);
            p = (int)(unsigned)list[(int)(((unsigned long long)r * (unsigned long long)n) >> 32)];
        }
        const double x = a.xyz[(size_t)p * 3], y = a.xyz[(size_t)p * 3 + 1], z = a.xyz[(size_t)p * 3 + 2];
        float *o = a.feats + ((size_t)b * a.num_point + i) * F;
        o[0] = (float)(x - cx);                                   // :229-231
        o[1] = (float)(y - cy);
        o[2] = (float)z;
        o[3] = (float)(x / a.max_x);                              // :226-228
        o[4] = (float)(y / a.max_y);
        o[5] = (float)(z / a.max_z);
        for (int k = 0; k < a.E; ++k) o[6 + k] = a.extra[(size_t)k * a.P + p];     // :236-252
        a.out_labels[(size_t)b * a.num_point + i] = a.labels[p];
        if (a.sel) a.sel[(size_t)b * a.num_point + i] = p;
    }
}

__global__ __launch_bounds__(SB_THREADS) void sample_blocks_kernel(SamplerArgs a)
{
    __shared__ unsigned long long list[SB_CAP];
    __shared__ int s_cnt, s_n;
    sample_one_block(a, blockIdx.x, list, s_cnt, s_n);
}

// Several scenes ("rooms") in ONE launch: the reference's loader mixes rooms inside a batch (room_idxs replicated by point
// share and shuffled, sem_seg_training.py:184-193), and a launch per room that contributes to a batch put 0.1 ms each in
// front of every step.  rooms = a device table of per-room descriptors (the scene-specific fields of SamplerArgs),
// room_of_block [B] = the room every output block is drawn from.
struct SamplerRoom {
    const double *xyz;
    const int *order;
    const int *cell_start;
    const float *extra;
    const long long *labels;
    double x0, y0, cell;
    double max_x, max_y, max_z;
    int nx, ny, P, pad;
};
static_assert(sizeof(SamplerRoom) == 104, "room descriptor layout (scene.MultiRoomSampler packs it)");

struct RoomIds { unsigned char id[256]; };          // room of each block, travelling in the kernel arguments (no upload)

__global__ __launch_bounds__(SB_THREADS) void sample_blocks_multi_kernel(SamplerArgs a, const SamplerRoom *__restrict__ rooms,
                                                                        const int *__restrict__ room_of_block, RoomIds ids, int nrooms)
{
    __shared__ unsigned long long list[SB_CAP];
    __shared__ int s_cnt, s_n;
    const int r = min(max(room_of_block ? room_of_block[blockIdx.x] : (int)ids.id[blockIdx.x & 255], 0), nrooms - 1);
    const SamplerRoom m = rooms[r];
    a.xyz = m.xyz; a.order = m.order; a.cell_start = m.cell_start; a.extra = m.extra; a.labels = m.labels;
    a.x0 = m.x0; a.y0 = m.y0; a.cell = m.cell; a.nx = m.nx; a.ny = m.ny; a.P = m.P;
    a.max_x = m.max_x; a.max_y = m.max_y; a.max_z = m.max_z;
    sample_one_block(a, blockIdx.x, list, s_cnt, s_n);
}

}  // namespace

PN2_EXPORT int pn2_sample_blocks_multi(const void *rooms, int nrooms, const int *room_of_block, int room_ids_on_host, int E,
                                       double block_size, int num_point, int min_points, unsigned long long seed, int B, float *feats,
                                       long long *out_labels, int *info, int *sel_idx, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(rooms); PN2_REQUIRE_PTR(room_of_block); PN2_REQUIRE_PTR(feats); PN2_REQUIRE_PTR(out_labels); PN2_REQUIRE_PTR(info);
    if (B < 0 || nrooms <= 0 || E < 0 || num_point <= 0 || min_points < 0 || !(block_size > 0.0)) return PN2_ERR_SHAPE;
    if (num_point > SB_CAP / 2) return PN2_ERR_UNSUPPORTED;
    if (B == 0) return PN2_OK;
    SamplerArgs a = {};
    a.E = E;
    a.half = block_size / 2.0;
    a.num_point = num_point; a.min_points = min_points; a.max_attempts = 256; a.seed = seed;
    a.feats = feats; a.out_labels = out_labels; a.info = info; a.sel = sel_idx;
    RoomIds ids = {};
    const int *dev_ids = room_of_block;
    if (room_ids_on_host) {                            // a HOST array: copied into the launch's arguments
        if (B > 256 || nrooms > 256) return PN2_ERR_UNSUPPORTED;
        for (int i = 0; i < B; ++i) {
            if (room_of_block[i] < 0 || room_of_block[i] >= nrooms) return PN2_ERR_SHAPE;
            ids.id[i] = (unsigned char)room_of_block[i];
        }
        dev_ids = nullptr;
    }
    hipLaunchKernelGGL(sample_blocks_multi_kernel, dim3(B), dim3(SB_THREADS), 0, static_cast<hipStream_t>(stream_), a,
                       static_cast<const SamplerRoom *>(rooms), dev_ids, ids, nrooms);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_sample_blocks(const double *xyz, const int *order, const int *cell_start, const float *extra,
                                 const long long *labels, double x0, double y0, double cell, int nx, int ny, int P, int E,
                                 double block_size, const double *coord_max, int num_point, int min_points, unsigned long long seed,
                                 int B, float *feats, long long *out_labels, int *info, int *sel_idx, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz); PN2_REQUIRE_PTR(order); PN2_REQUIRE_PTR(cell_start); PN2_REQUIRE_PTR(labels);
    PN2_REQUIRE_PTR(coord_max); PN2_REQUIRE_PTR(feats); PN2_REQUIRE_PTR(out_labels); PN2_REQUIRE_PTR(info);
    if (B < 0 || P <= 0 || nx <= 0 || ny <= 0 || E < 0 || num_point <= 0 || min_points < 0 || !(cell > 0.0) || !(block_size > 0.0))
        return PN2_ERR_SHAPE;
    if (E > 0 && extra == nullptr) return PN2_ERR_NULL;
    if (num_point > SB_CAP / 2) return PN2_ERR_UNSUPPORTED;
    if (B == 0) return PN2_OK;
    SamplerArgs a;
    a.xyz = xyz; a.order = order; a.cell_start = cell_start; a.extra = extra; a.labels = labels;
    a.x0 = x0; a.y0 = y0; a.cell = cell; a.nx = nx; a.ny = ny; a.P = P; a.E = E;
    a.half = block_size / 2.0;
    a.max_x = coord_max[0]; a.max_y = coord_max[1]; a.max_z = coord_max[2];
    a.num_point = num_point; a.min_points = min_points; a.max_attempts = 256; a.seed = seed;
    a.feats = feats; a.out_labels = out_labels; a.info = info; a.sel = sel_idx;
    hipLaunchKernelGGL(sample_blocks_kernel, dim3(B), dim3(SB_THREADS), 0, static_cast<hipStream_t>(stream_), a);
    return PN2_LAUNCH_RC();
}
