// Segmentation head tail and loss of pointnet2_sem_seg:
//   x = conv2(x); x = F.log_softmax(x, dim=1)          reference models/pointnet2_sem_seg.py:37-38
//   F.nll_loss(pred, target, weight=weight)            reference models/pointnet2_sem_seg.py:48
// Per-point rows [M][K] (K = 128 features) against a [C][K] weight with C <= 32 classes: far too
// narrow for the 32-column MFMA tiles of pn2_mlp.hip, and HBM-bound anyway (M*K*4 bytes in, M*C*4
// out), so these are VALU kernels that stream 64-row tiles through LDS with coalesced float4 rows.
#include "pn2_common.h"

namespace {

constexpr int HD_THREADS = 256;
constexpr int HD_ROWS = 64;        // rows per tile
constexpr int HD_KMAX = 128;       // feature width the kernels are built for
constexpr int HD_CMAX = 32;        // classes
constexpr int HD_LDY = HD_KMAX + 4;

// Dropout in front of conv2 (reference models/pointnet2_sem_seg.py:36, nn.Dropout(0.5)): the Bernoulli keep-mask is a
// counter-based hash of (seed, row, column), regenerated wherever it is needed (forward staging, backward staging
// and the gy store) instead of being stored; kept values are scaled by 1/(1-p).  seed == nullptr: no dropout.
struct DropArgs {
    const unsigned long long *seed;
    unsigned thresh;              // p * 2^32: an element is dropped when its hash is below
    float scale;                  // 1 / (1 - p)
    // counted form (forward only): the seed is hashed from state[0] (base) and state[1] (calls so far), written to
    // *seed_out for the backward, and the workgroup that finishes last counts the call (state[2] = ticket)
    unsigned long long *state, *seed_out;
};
__device__ __forceinline__ bool drop_on(const DropArgs &d) { return d.seed != nullptr || d.state != nullptr; }
__device__ __forceinline__ unsigned long long drop_seed_value(const DropArgs &d)
{
    if (d.state) {
        unsigned long long x = d.state[0] + (d.state[1] + 1ull) * 0x9E3779B97F4A7C15ull;      // splitmix64 finaliser
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    }
    return d.seed ? *d.seed : 0ull;
}
// End of a counted forward launch.  `seed` is the value the workgroup read at its START (stage_rows received the same
// register): that load has long completed -- its value masked every staged element -- so nothing of this workgroup still
// reads the state when it takes its ticket, and the workgroup that takes the last one may count the call.  The counter
// and the ticket change through device-scope atomics (no plain store races a later launch's first read: the kernel
// boundary orders those).
__device__ __forceinline__ void drop_count_call(const DropArgs &d, unsigned long long seed)
{
    if (!d.state || threadIdx.x != 0) return;
    if (blockIdx.x == 0) *d.seed_out = seed;
    unsigned *ticket = reinterpret_cast<unsigned *>(d.state + 2);
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
        atomicExch(ticket, 0u);
        atomicAdd(d.state + 1, 1ull);
    }
}
__device__ __forceinline__ bool drop_keep(unsigned long long seed, unsigned row, unsigned col, unsigned thresh)
{
    unsigned h = ((unsigned)seed ^ (row * 0x9E3779B1u)) + (((unsigned)(seed >> 32)) ^ (col * 0x85EBCA77u));
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h >= thresh;
}
__device__ __forceinline__ float4 drop4(float4 v, const DropArgs &d, unsigned long long seed, unsigned row, unsigned col)
{
    v.x = drop_keep(seed, row, col + 0, d.thresh) ? v.x * d.scale : 0.f;
    v.y = drop_keep(seed, row, col + 1, d.thresh) ? v.y * d.scale : 0.f;
    v.z = drop_keep(seed, row, col + 2, d.thresh) ? v.z * d.scale : 0.f;
    v.w = drop_keep(seed, row, col + 3, d.thresh) ? v.w * d.scale : 0.f;
    return v;
}

// 64 rows x K floats -> LDS tile: the (up to) 8 float4 of a thread are all loaded before the first LDS store,
// unconditionally (clamped row, masked value): a load-store loop would wait for every load in turn.
__device__ __forceinline__ void stage_rows(const float *__restrict__ y, int ldy, int row0, int M, int k4n, float *__restrict__ sY, int tid,
                                           const DropArgs &d, unsigned long long seed)
{
    constexpr int NI = HD_ROWS * (HD_KMAX / 4) / HD_THREADS;          // 8
    const int total = HD_ROWS * k4n;
    float4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = min(tid + i * HD_THREADS, total - 1);
        const int r = e / k4n, q = e - r * k4n;
        v[i] = *reinterpret_cast<const float4 *>(&y[(size_t)min(row0 + r, M - 1) * ldy + 4 * q]);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * HD_THREADS;
        if (e < total) {
            const int r = e / k4n, q = e - r * k4n;
            float4 o = row0 + r < M ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (drop_on(d)) o = drop4(o, d, seed, (unsigned)(row0 + r), (unsigned)(4 * q));
            *reinterpret_cast<float4 *>(&sY[r * HD_LDY + 4 * q]) = o;
        }
    }
}

// The class weights [C][K] on their way into LDS (row pitch ld): every load of a thread is issued before its first store
// -- a `load; store` loop waits for one memory round trip per pass at the head of the launch -- and the caller stages its
// first tile of rows between load() and store(), so that both travel in one round trip.  Rows past C and columns past K
// are stored as zeros when `pad`.
struct HeadWeights {
    static constexpr int PASSES = HD_CMAX * (HD_KMAX / 4) / HD_THREADS;      // 4
    float4 v[PASSES];
    // element e = (class j = e / k4n, four columns at 4 * (e % k4n)); columns from 4 * kvalid on do not exist in w
    __device__ __forceinline__ void load(const float *__restrict__ w, int K, int C, int k4n, int kvalid, int tid)
    {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int e = tid + i * HD_THREADS;
            const int j = e / k4n, q = e - j * k4n;
            v[i] = *reinterpret_cast<const float4 *>(&w[(size_t)min(j, C - 1) * K + 4 * (q < kvalid ? q : 0)]);
        }
    }
    __device__ __forceinline__ void store(float *__restrict__ sW, int ld, int C, int k4n, int kvalid, int rows, int tid) const
    {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int e = tid + i * HD_THREADS;
            const int j = e / k4n, q = e - j * k4n;
            if (j < rows)
                *reinterpret_cast<float4 *>(&sW[j * ld + 4 * q]) = (j < C && q < kvalid) ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
};

// ---- forward -----------------------------------------------------------------------------------------
// tile -> LDS, thread (row = t & 63, jq = t >> 6) accumulates classes jq, jq+4, ... over k in the
// order k = 0..K-1 (one fma chain per class), then one thread per row does the log-softmax.
template <int CQ>   // classes per thread = ceil(C / 4)
__global__ __launch_bounds__(HD_THREADS) void head_logits_kernel(const float *__restrict__ y, int ldy,
                                                                const float *__restrict__ w,
                                                                const float *__restrict__ bias,
                                                                float *__restrict__ logp, int M, int K, int C, DropArgs drop)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ __attribute__((aligned(16))) float sY[HD_ROWS * HD_LDY];
    __shared__ __attribute__((aligned(16))) float sW[HD_CMAX * HD_KMAX];
    __shared__ float sL[HD_ROWS][HD_CMAX + 1];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * HD_ROWS;
    const int k4n = K >> 2;
    HeadWeights hw;
    hw.load(w, K, C, k4n, k4n, tid);
    const unsigned long long seed0 = drop_seed_value(drop);      // read ONCE per workgroup (see drop_count_call)
    stage_rows(y, ldy, row0, M, k4n, sY, tid, drop, seed0);
    hw.store(sW, HD_KMAX, C, k4n, k4n, C, tid);
    __syncthreads();
    const int r = tid & 63, jq = tid >> 6;
    float acc[CQ];
#pragma unroll
    for (int i = 0; i < CQ; ++i) acc[i] = 0.f;
    for (int q = 0; q < k4n; ++q) {
        const float4 a = *reinterpret_cast<const float4 *>(&sY[r * HD_LDY + 4 * q]);
#pragma unroll
        for (int i = 0; i < CQ; ++i) {
            const int j = jq + 4 * i;
            if (j < C) {                                      // uniform per wave
                const float4 b = *reinterpret_cast<const float4 *>(&sW[j * HD_KMAX + 4 * q]);
                acc[i] = fmaf(a.x, b.x, acc[i]);
                acc[i] = fmaf(a.y, b.y, acc[i]);
                acc[i] = fmaf(a.z, b.z, acc[i]);
                acc[i] = fmaf(a.w, b.w, acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        const int j = jq + 4 * i;
        if (j < C) sL[r][j] = acc[i] + (bias ? bias[j] : 0.f);
    }
    __syncthreads();
    {   // log-softmax: 4 lanes per row (classes part, part+4, ...), then the [rows][C] tile leaves coalesced
        const int rr = tid >> 2, part = tid & 3;
        float mx = -INFINITY;
        for (int j = part; j < C; j += 4) mx = fmaxf(mx, sL[rr][j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        float sum = 0.f;
        for (int j = part; j < C; j += 4) sum += expf(sL[rr][j] - mx);
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const float lse = mx + logf(sum);
        for (int j = part; j < C; j += 4) sL[rr][j] -= lse;
    }
    __syncthreads();
    const int nout = min(HD_ROWS, M - row0) * C;
    float *o = logp + (size_t)row0 * C;
    for (int e = tid; e < nout; e += HD_THREADS) {
        const int rr = e / C;
        o[e] = sL[rr][e - rr * C];
    }
    if (drop.state) drop_count_call(drop, seed0);
}

// K = 128 (the model's head): the 64 x C products of a tile on the matrix cores.  Wave w takes row block w & 1 and
// K half w >> 1 (32 MFMAs of v_mfma_f32_32x32x2_f32: lane l supplies y[row l & 31][k] and w[class l & 31][k], k from
// its half's 32 values); the two K halves meet in sL.  Classes >= C read unstaged weight rows: a column of the
// product depends on its own B column only, and those columns are never used.  The vector-unit loop above is bound by
// its LDS reads (6 ds_read_b128 per 20 fma): 30.6 us per 65536 rows, 13.7 us without the loop.
typedef float hd_f32x16 __attribute__((ext_vector_type(16)));
constexpr int HD_LDW = HD_KMAX + 4;

__global__ __launch_bounds__(HD_THREADS) void head_logits_mfma_kernel(const float *__restrict__ y, int ldy,
                                                                     const float *__restrict__ w,
                                                                     const float *__restrict__ bias,
                                                                     float *__restrict__ logp, int M, int C, DropArgs drop)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ __attribute__((aligned(16))) float sY[HD_ROWS * HD_LDY];
    __shared__ __attribute__((aligned(16))) float sW[HD_CMAX * HD_LDW];
    __shared__ float sL[HD_ROWS][HD_CMAX + 1];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * HD_ROWS;
    constexpr int k4n = HD_KMAX / 4;
    HeadWeights hw;
    hw.load(w, HD_KMAX, C, k4n, k4n, tid);
    const unsigned long long seed0 = drop_seed_value(drop);      // read ONCE per workgroup (see drop_count_call)
    stage_rows(y, ldy, row0, M, k4n, sY, tid, drop, seed0);
    hw.store(sW, HD_LDW, C, k4n, k4n, C, tid);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int rb = wave & 1, kh = wave >> 1;
    hd_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *aRow = &sY[(rb * 32 + l31) * HD_LDY + kh * 64 + half * 32];
    const float *bRow = &sW[l31 * HD_LDW + kh * 64 + half * 32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 a = *reinterpret_cast<const float4 *>(aRow + 4 * q);
        const float4 b = *reinterpret_cast<const float4 *>(bRow + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    // C/D layout: column (class) = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 half
    if (kh == 0 && l31 < C) {
        const float bv = bias ? bias[l31] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sL[rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r] + bv;
    }
    __syncthreads();
    if (kh == 1 && l31 < C) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sL[rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half][l31] += acc[r];
    }
    __syncthreads();
    {   // log-softmax: 4 lanes per row (classes part, part+4, ...), then the [rows][C] tile leaves coalesced
        const int rr = tid >> 2, part = tid & 3;
        float mx = -INFINITY;
        for (int j = part; j < C; j += 4) mx = fmaxf(mx, sL[rr][j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        float sum = 0.f;
        for (int j = part; j < C; j += 4) sum += expf(sL[rr][j] - mx);
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const float lse = mx + logf(sum);
        for (int j = part; j < C; j += 4) sL[rr][j] -= lse;
    }
    __syncthreads();
    const int nout = min(HD_ROWS, M - row0) * C;
    float *o = logp + (size_t)row0 * C;
    for (int e = tid; e < nout; e += HD_THREADS) {
        const int rr = e / C;
        o[e] = sL[rr][e - rr * C];
    }
    if (drop.state) drop_count_call(drop, seed0);
}

// ---- backward ----------------------------------------------------------------------------------------
// dz = g - exp(logp) * sum_j g_j ;  gy = dz * W ;  dW += dz^T * y ;  db += sum dz.
// Workgroups walk tiles grid-stride and keep their dW slab in registers; slabs are summed in
// fixed order by head_dw_reduce_kernel.
__global__ __launch_bounds__(HD_THREADS) void head_logits_backward_kernel(
    const float *__restrict__ g, const float *__restrict__ logp, const float *__restrict__ y, int ldy,
    const float *__restrict__ w, float *__restrict__ gy, int ldgy, float *__restrict__ partial, int M, int K, int C, DropArgs drop)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ __attribute__((aligned(16))) float sY[HD_ROWS * HD_LDY];
    __shared__ __attribute__((aligned(16))) float sW[HD_CMAX * HD_KMAX];
    __shared__ __attribute__((aligned(16))) float sD[HD_ROWS][HD_CMAX];
    __shared__ float sG[HD_ROWS * HD_CMAX], sP[HD_ROWS * HD_CMAX];      // g / logp rows of the tile, pitch C
    const int tid = threadIdx.x;
    const int k4n = K >> 2;
    const int ntiles = (M + HD_ROWS - 1) / HD_ROWS;
    {
        HeadWeights hw;
        hw.load(w, K, C, HD_KMAX / 4, k4n, tid);
        hw.store(sW, HD_KMAX, C, HD_KMAX / 4, k4n, HD_CMAX, tid);
    }
    // dW ownership: k = tid & 127, classes 16*(tid >> 7) .. +15
    const int wk = tid & 127, wj0 = (tid >> 7) * 16;
    float dwa[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dwa[i] = 0.f;
    float dba = 0.f;                                          // tid < HD_CMAX: db of class tid
    // gy ownership: 4 columns at gk4, rows (tid >> 5) + 8 i
    const int gk4 = (tid & 31) * 4, gr0 = tid >> 5;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * HD_ROWS;
        __syncthreads();                                      // previous tile's readers are done (also covers sW)
        // g / logp rows of the tile: [rows][C] contiguous, loaded coalesced and unconditionally (clamped), in flight
        // together with the feature rows
        constexpr int NG = HD_ROWS * HD_CMAX / HD_THREADS;           // 8
        const int nin = min(HD_ROWS, M - row0) * C;
        float gv[NG], lv[NG];
        {
            const float *gt = g + (size_t)row0 * C, *lt = logp + (size_t)row0 * C;
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int e = min(tid + i * HD_THREADS, nin - 1);
                gv[i] = gt[e];
                lv[i] = lt[e];
            }
        }
        stage_rows(y, ldy, row0, M, k4n, sY, tid, drop, drop_seed_value(drop));
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int e = tid + i * HD_THREADS;
            if (e < HD_ROWS * C) { sG[e] = e < nin ? gv[i] : 0.f; sP[e] = e < nin ? lv[i] : 0.f; }
        }
        __syncthreads();
        {   // dz = g - exp(logp) * sum_j g_j: 4 lanes per row
            const int rr = tid >> 2, part = tid & 3;
            float gs = 0.f;
            for (int j = part; j < C; j += 4) gs += sG[rr * C + j];
            gs += __shfl_xor(gs, 1);
            gs += __shfl_xor(gs, 2);
            for (int j = part; j < HD_CMAX; j += 4)
                sD[rr][j] = j < C ? sG[rr * C + j] - expf(sP[rr * C + j]) * gs : 0.f;     // rows >= M: g = 0 -> dz = 0
        }
        __syncthreads();
        if (gy && gk4 < K) {
            const bool full = row0 + HD_ROWS <= M;            // straight-line stores: a branch per row makes the
#pragma unroll                                                // compiler wait for each store's acknowledgement
            for (int i = 0; i < HD_ROWS / 8; ++i) {
                const int r = gr0 + 8 * i;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int j = 0; j < C; ++j) {
                    const float d = sD[r][j];
                    const float4 b = *reinterpret_cast<const float4 *>(&sW[j * HD_KMAX + gk4]);
                    a.x = fmaf(d, b.x, a.x);
                    a.y = fmaf(d, b.y, a.y);
                    a.z = fmaf(d, b.z, a.z);
                    a.w = fmaf(d, b.w, a.w);
                }
                if (drop.seed) a = drop4(a, drop, *drop.seed, (unsigned)(row0 + r), (unsigned)gk4);
                if (full || row0 + r < M) *reinterpret_cast<float4 *>(&gy[(size_t)(row0 + r) * ldgy + gk4]) = a;
            }
        }
        if (wj0 < C) {                                        // uniform per wave
            for (int r = 0; r < HD_ROWS; ++r) {
                const float yv = sY[r * HD_LDY + wk];
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const float4 d = *reinterpret_cast<const float4 *>(&sD[r][wj0 + 4 * i4]);
                    dwa[4 * i4 + 0] = fmaf(d.x, yv, dwa[4 * i4 + 0]);
                    dwa[4 * i4 + 1] = fmaf(d.y, yv, dwa[4 * i4 + 1]);
                    dwa[4 * i4 + 2] = fmaf(d.z, yv, dwa[4 * i4 + 2]);
                    dwa[4 * i4 + 3] = fmaf(d.w, yv, dwa[4 * i4 + 3]);
                }
            }
        }
        if (tid < HD_CMAX)
            for (int r = 0; r < HD_ROWS; ++r) dba += sD[r][tid];
    }
    // slab [C][K+1] of this workgroup
    float *slab = partial + (size_t)blockIdx.x * C * (K + 1);
    if (wk < K) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (wj0 + i < C) slab[(size_t)(wj0 + i) * (K + 1) + wk] = dwa[i];
    }
    if (tid < C) slab[(size_t)tid * (K + 1) + K] = dba;
}

// K = 128: both products of the backward on the matrix cores.  Per 64-row tile wave w owns feature columns
// [32 w, 32 w + 32): gy = dz * W as two 32 x 32 blocks (reduction over the 32 zero-padded classes, 16 MFMAs each) and
// its 32 x 32 block of dW = dz^T * y (reduction over the tile's 64 rows, 32 MFMAs), accumulated in registers over
// the workgroup's tiles.
constexpr int HD_LDD = HD_CMAX + 4;

__global__ __launch_bounds__(HD_THREADS) void head_logits_backward_mfma_kernel(
    const float *__restrict__ g, const float *__restrict__ logp, const float *__restrict__ y, int ldy,
    const float *__restrict__ w, float *__restrict__ gy, int ldgy, float *__restrict__ partial, int M, int C, DropArgs drop)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ __attribute__((aligned(16))) float sY[HD_ROWS * HD_LDY];
    __shared__ __attribute__((aligned(16))) float sW[HD_CMAX * HD_LDW];
    __shared__ __attribute__((aligned(16))) float sD[HD_ROWS * HD_LDD];
    __shared__ float sG[HD_ROWS * HD_CMAX], sP[HD_ROWS * HD_CMAX];      // g / logp rows of the tile, pitch C
    constexpr int K = HD_KMAX, k4n = HD_KMAX / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int ntiles = (M + HD_ROWS - 1) / HD_ROWS;
    {
        HeadWeights hw;
        hw.load(w, K, C, k4n, k4n, tid);
        hw.store(sW, HD_LDW, C, k4n, k4n, HD_CMAX, tid);
    }
    hd_f32x16 accW;                                           // dW[class][32 wave + (lane & 31)]
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[r] = 0.f;
    float dba = 0.f;                                          // tid < HD_CMAX: db of class tid
    const int col = wave * 32 + l31;                          // this lane's feature column

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * HD_ROWS;
        __syncthreads();                                      // previous tile's readers are done (also covers sW)
        constexpr int NG = HD_ROWS * HD_CMAX / HD_THREADS;           // 8
        const int nin = min(HD_ROWS, M - row0) * C;
        float gv[NG], lv[NG];
        {
            const float *gt = g + (size_t)row0 * C, *lt = logp + (size_t)row0 * C;
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int e = min(tid + i * HD_THREADS, nin - 1);
                gv[i] = gt[e];
                lv[i] = lt[e];
            }
        }
        stage_rows(y, ldy, row0, M, k4n, sY, tid, drop, drop_seed_value(drop));
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int e = tid + i * HD_THREADS;
            if (e < HD_ROWS * C) { sG[e] = e < nin ? gv[i] : 0.f; sP[e] = e < nin ? lv[i] : 0.f; }
        }
        __syncthreads();
        {   // dz = g - exp(logp) * sum_j g_j: 4 lanes per row; classes >= C are zero columns
            const int rr = tid >> 2, part = tid & 3;
            float gs = 0.f;
            for (int j = part; j < C; j += 4) gs += sG[rr * C + j];
            gs += __shfl_xor(gs, 1);
            gs += __shfl_xor(gs, 2);
            for (int j = part; j < HD_CMAX; j += 4)
                sD[rr * HD_LDD + j] = j < C ? sG[rr * C + j] - expf(sP[rr * C + j]) * gs : 0.f;
        }
        __syncthreads();
        if (gy) {
            const bool full = row0 + HD_ROWS <= M;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                hd_f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float *aRow = &sD[(rb * 32 + l31) * HD_LDD + half * 16];
                const float *bCol = &sW[(half * 16) * HD_LDW + col];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = *reinterpret_cast<const float4 *>(aRow + 4 * q);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bCol[(4 * q + 0) * HD_LDW], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bCol[(4 * q + 1) * HD_LDW], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bCol[(4 * q + 2) * HD_LDW], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bCol[(4 * q + 3) * HD_LDW], acc, 0, 0, 0);
                }
                if (drop.seed) {
                    const unsigned long long seed = *drop.seed;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned row = (unsigned)(row0 + rb * 32 + 4 * half + (r & 3) + 8 * (r >> 2));
                        acc[r] = drop_keep(seed, row, (unsigned)col, drop.thresh) ? acc[r] * drop.scale : 0.f;
                    }
                }
                float *o = gy + (size_t)(row0 + rb * 32 + 4 * half) * ldgy + col;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) pn2::store_rows(&o[(size_t)((r & 3) + 8 * (r >> 2)) * ldgy], acc[r]);   // read next by the stack's backward
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = (r & 3) + 8 * (r >> 2);
                        if (row0 + rb * 32 + 4 * half + rl < M) pn2::store_rows(&o[(size_t)rl * ldgy], acc[r]);
                    }
                }
            }
        }
        {   // dW block: A[class][row] = dz^T, B[row][col] = y; lane half h reduces rows [32 h, 32 h + 32)
            const float *dBase = &sD[(half * 32) * HD_LDD + l31];
            const float *yBase = &sY[(half * 32) * HD_LDY + col];
#pragma unroll 8
            for (int t = 0; t < 32; ++t)
                accW = __builtin_amdgcn_mfma_f32_32x32x2f32(dBase[t * HD_LDD], yBase[t * HD_LDY], accW, 0, 0, 0);
        }
        if (tid < HD_CMAX)
            for (int r = 0; r < HD_ROWS; ++r) dba += sD[r * HD_LDD + tid];
    }
    // slab [C][K+1] of this workgroup: C/D layout row (class) = (r & 3) + 8 (r >> 2) + 4 half, column = col
    float *slab = partial + (size_t)blockIdx.x * C * (K + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cls = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (cls < C) slab[(size_t)cls * (K + 1) + col] = accW[r];
    }
    if (tid < C) slab[(size_t)tid * (K + 1) + K] = dba;
}

// ---- weighted negative log-likelihood ---------------------------------------------------------------------
// loss = sum_i -w[t_i] * logp[i][t_i] / sum_i w[t_i]   (rows with t_i == ignore_index are skipped,
// F.nll_loss reduction='mean').  Partials per workgroup in double, combined in order.
__global__ __launch_bounds__(256) void nll_partial_kernel(const float *__restrict__ logp, const int64_t *__restrict__ target,
                                                         const float *__restrict__ weight, long long M, int C,
                                                         long long ignore_index, double *__restrict__ partial,
                                                         int32_t *__restrict__ err_count, unsigned *__restrict__ ticket,
                                                         float *__restrict__ loss, float *__restrict__ wsum)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ double sN[256], sDn[256];
    __shared__ bool sLast;
    double num = 0.0, den = 0.0;
    int bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long long)gridDim.x * 256) {
        const long long t = target[i];
        if (t == ignore_index) continue;
        if (t < 0 || t >= C) { bad = 1; continue; }
        const float wv = weight ? weight[t] : 1.f;
        num -= (double)(wv * logp[i * C + t]);
        den += (double)wv;
    }
    if (bad && err_count) atomicAdd(err_count, 1);
    sN[threadIdx.x] = num;
    sDn[threadIdx.x] = den;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sN[threadIdx.x] += sN[threadIdx.x + s]; sDn[threadIdx.x] += sDn[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sN[0]; partial[2 * blockIdx.x + 1] = sDn[0]; }
    if (!ticket) return;
    // With a ticket word the workgroup that finishes last sums the partials (in index order: the same result whoever
    // is last) -- no second launch.  Release our partial, take a ticket, and as the last one acquire everybody's.
    if (threadIdx.x == 0) {
        __threadfence();
        sLast = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!sLast) return;
    __threadfence();
    // all partials in one memory round trip (gridDim.x <= 256 = one per thread), then summed by one thread in index
    // order from LDS -- the order of nll_finalize_kernel
    const volatile double *vp = partial;
    sN[threadIdx.x] = threadIdx.x < gridDim.x ? vp[2 * threadIdx.x] : 0.0;
    sDn[threadIdx.x] = threadIdx.x < gridDim.x ? vp[2 * threadIdx.x + 1] : 0.0;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double n = 0.0, d = 0.0;
    for (unsigned i = 0; i < gridDim.x; ++i) { n += sN[i]; d += sDn[i]; }
    *loss = (float)(n / d);                                   // 0/0 -> nan, as torch
    *wsum = (float)d;
    *ticket = 0u;                                             // ready for the next launch
}

__global__ __launch_bounds__(64) void nll_finalize_kernel(const double *__restrict__ partial, int P, float *__restrict__ loss,
                                                         float *__restrict__ wsum)
{
    if (threadIdx.x != 0) return;
    double num = 0.0, den = 0.0;
    for (int i = 0; i < P; ++i) { num += partial[2 * i]; den += partial[2 * i + 1]; }
    *loss = (float)(num / den);                               // 0/0 -> nan, as torch
    *wsum = (float)den;
}

__global__ __launch_bounds__(256) void nll_backward_kernel(const float *__restrict__ gloss, const int64_t *__restrict__ target,
                                                          const float *__restrict__ weight, const float *__restrict__ wsum,
                                                          long long M, int C, long long ignore_index, float *__restrict__ glogp)
{
    PN2_MAIN_BRANCH_PRIORITY();
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * C) return;
    const long long i = e / C;
    const int j = (int)(e - i * C);
    const long long t = target[i];
    float v = 0.f;
    if (t == j && t != ignore_index) v = -(weight ? weight[t] : 1.f) / *wsum * *gloss;
    glogp[e] = v;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(DropArgs d, long long M, int K, unsigned char *__restrict__ mask)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * K) return;
    const long long r = e / K;
    mask[e] = drop_keep(*d.seed, (unsigned)r, (unsigned)(e - r * K), d.thresh) ? 1 : 0;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

DropArgs make_drop(const unsigned long long *seed, float p)
{
    DropArgs d;
    d.state = d.seed_out = nullptr;
    d.seed = (seed && p > 0.f) ? seed : nullptr;
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
    d.scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
    return d;
}

}  // namespace

PN2_EXPORT int pn2_head_logits(const float *y, int ldy, const float *w, const float *bias, float *logp, int M, int K,
                               int C, pn2_stream_t stream)
{
    return pn2_head_logits_dropout(y, ldy, w, bias, logp, M, K, C, nullptr, 0.f, stream);
}

PN2_EXPORT int pn2_dropout_mask(const unsigned long long *seed, float p, long long M, int K, unsigned char *mask,
                                pn2_stream_t stream)
{
    PN2_REQUIRE_PTR(seed); PN2_REQUIRE_PTR(mask);
    if (M <= 0 || K <= 0 || p < 0.f || p > 1.f) return PN2_ERR_SHAPE;
    DropArgs d = make_drop(seed, p);
    d.seed = seed;
    const long long n = M * K;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, M, K, mask);
    return PN2_LAUNCH_RC();
}

static int head_logits_impl(const float *y, int ldy, const float *w, const float *bias, float *logp, int M, int K, int C,
                            const DropArgs &drop, pn2_stream_t stream);

PN2_EXPORT int pn2_head_logits_dropout(const float *y, int ldy, const float *w, const float *bias, float *logp, int M,
                                       int K, int C, const unsigned long long *drop_seed, float drop_p,
                                       pn2_stream_t stream)
{
    if (drop_p < 0.f || drop_p > 1.f) return PN2_ERR_SHAPE;
    return head_logits_impl(y, ldy, w, bias, logp, M, K, C, make_drop(drop_seed, drop_p), stream);
}

PN2_EXPORT int pn2_head_logits_dropout_counted(const float *y, int ldy, const float *w, const float *bias, float *logp, int M,
                                               int K, int C, unsigned long long *state, unsigned long long *seed_out,
                                               float drop_p, pn2_stream_t stream)
{
    PN2_REQUIRE_PTR(state); PN2_REQUIRE_PTR(seed_out);
    if (!(drop_p > 0.f) || drop_p > 1.f) return PN2_ERR_SHAPE;
    DropArgs drop = make_drop(nullptr, drop_p);
    drop.state = state;
    drop.seed_out = seed_out;
    return head_logits_impl(y, ldy, w, bias, logp, M, K, C, drop, stream);
}

static int head_logits_impl(const float *y, int ldy, const float *w, const float *bias, float *logp, int M, int K, int C,
                            const DropArgs &drop, pn2_stream_t stream)
{
    PN2_REQUIRE_PTR(y); PN2_REQUIRE_PTR(w); PN2_REQUIRE_PTR(logp);
    if (M <= 0 || K <= 0 || C <= 0 || ldy < K) return PN2_ERR_SHAPE;
    if (K > HD_KMAX || C > HD_CMAX || (K & 3) || (ldy & 3) || !aligned16(y) || !aligned16(w)) return PN2_ERR_UNSUPPORTED;
    const dim3 grid((M + HD_ROWS - 1) / HD_ROWS);
    hipStream_t s = (hipStream_t)stream;
    if (K == HD_KMAX && pn2::tune_get("hd_mfma", 1)) {
        hipLaunchKernelGGL(head_logits_mfma_kernel, grid, dim3(HD_THREADS), 0, s, y, ldy, w, bias, logp, M, C, drop);
        return PN2_LAUNCH_RC();
    }
    const int cq = (C + 3) / 4;
#define PN2_HD(Q) hipLaunchKernelGGL((head_logits_kernel<Q>), grid, dim3(HD_THREADS), 0, s, y, ldy, w, bias, logp, M, K, C, drop)
    if (cq <= 2) PN2_HD(2); else if (cq <= 4) PN2_HD(4); else PN2_HD(8);
#undef PN2_HD
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_head_logits_partials(int M)
{
    const int tiles = (M + HD_ROWS - 1) / HD_ROWS;
    return tiles < 1 ? 1 : (tiles > 512 ? 512 : tiles);
}

PN2_EXPORT int pn2_head_logits_backward(const float *glogp, const float *logp, const float *y, int ldy, const float *w,
                                        float *gy, int ldgy, float *partial, float *dw, float *db, int M, int K, int C,
                                        pn2_stream_t stream)
{
    return pn2_head_logits_dropout_backward(glogp, logp, y, ldy, w, gy, ldgy, partial, dw, db, M, K, C, nullptr, 0.f, stream);
}

PN2_EXPORT int pn2_head_logits_dropout_backward(const float *glogp, const float *logp, const float *y, int ldy,
                                                const float *w, float *gy, int ldgy, float *partial, float *dw, float *db,
                                                int M, int K, int C, const unsigned long long *drop_seed, float drop_p,
                                                pn2_stream_t stream)
{
    if (drop_p < 0.f || drop_p > 1.f) return PN2_ERR_SHAPE;
    const DropArgs drop = make_drop(drop_seed, drop_p);
    PN2_REQUIRE_PTR(glogp); PN2_REQUIRE_PTR(logp); PN2_REQUIRE_PTR(y); PN2_REQUIRE_PTR(w);
    PN2_REQUIRE_PTR(partial);                        // dw NULL: slabs only (summed later, pn2_mlp_dw_reduce_many)
    if (M <= 0 || K <= 0 || C <= 0 || ldy < K || (gy && ldgy < K)) return PN2_ERR_SHAPE;
    if (K > HD_KMAX || C > HD_CMAX || (K & 3) || (ldy & 3) || !aligned16(y) || !aligned16(w)) return PN2_ERR_UNSUPPORTED;
    if (gy && ((ldgy & 3) || !aligned16(gy))) return PN2_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int P = pn2_head_logits_partials(M);
    if (K == HD_KMAX && pn2::tune_get("hd_mfma", 1))
        hipLaunchKernelGGL(head_logits_backward_mfma_kernel, dim3(P), dim3(HD_THREADS), 0, s, glogp, logp, y, ldy, w, gy, ldgy,
                           partial, M, C, drop);
    else
        hipLaunchKernelGGL(head_logits_backward_kernel, dim3(P), dim3(HD_THREADS), 0, s, glogp, logp, y, ldy, w, gy, ldgy,
                           partial, M, K, C, drop);
    int rc = PN2_LAUNCH_RC();
    if (rc != PN2_OK || !dw) return rc;
    return pn2::launch_dw_reduce(partial, P, C, K, dw, db, s);       // same slab layout as the MLP's dW partials
}

PN2_EXPORT int pn2_nll_loss_partials(long long M)
{
    // (256 workgroups of one row per thread were tried: 12.5 against 9.5 us -- the ticketed form ends every workgroup with
    // a release fence, and 64 of them cost less than three more dependent round trips per thread)
    const long long b = (M + 1023) / 1024;
    return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

PN2_EXPORT int pn2_nll_loss(const float *logp, const int64_t *target, const float *weight, long long M, int C,
                            long long ignore_index, double *partial, float *loss, float *wsum, int32_t *err_count,
                            pn2_stream_t stream)
{
    PN2_REQUIRE_PTR(logp); PN2_REQUIRE_PTR(target); PN2_REQUIRE_PTR(partial); PN2_REQUIRE_PTR(loss); PN2_REQUIRE_PTR(wsum);
    if (M <= 0 || C <= 0) return PN2_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int P = pn2_nll_loss_partials(M);
    hipLaunchKernelGGL(nll_partial_kernel, dim3(P), dim3(256), 0, s, logp, target, weight, M, C, ignore_index, partial,
                       err_count, static_cast<unsigned *>(nullptr), loss, wsum);
    hipLaunchKernelGGL(nll_finalize_kernel, dim3(1), dim3(64), 0, s, partial, P, loss, wsum);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_nll_loss_ticketed(const float *logp, const int64_t *target, const float *weight, long long M, int C,
                                     long long ignore_index, double *partial, float *loss, float *wsum, int32_t *err_count,
                                     unsigned int *ticket, pn2_stream_t stream)
{
    PN2_REQUIRE_PTR(logp); PN2_REQUIRE_PTR(target); PN2_REQUIRE_PTR(partial); PN2_REQUIRE_PTR(loss); PN2_REQUIRE_PTR(wsum);
    PN2_REQUIRE_PTR(ticket);
    if (M <= 0 || C <= 0) return PN2_ERR_SHAPE;
    const int P = pn2_nll_loss_partials(M);
    hipLaunchKernelGGL(nll_partial_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, logp, target, weight, M, C, ignore_index,
                       partial, err_count, ticket, loss, wsum);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_nll_loss_backward(const float *gloss, const int64_t *target, const float *weight, const float *wsum,
                                     long long M, int C, long long ignore_index, float *glogp, pn2_stream_t stream)
{
    PN2_REQUIRE_PTR(gloss); PN2_REQUIRE_PTR(target); PN2_REQUIRE_PTR(wsum); PN2_REQUIRE_PTR(glogp);
    if (M <= 0 || C <= 0) return PN2_ERR_SHAPE;
    const long long n = M * C;
    hipLaunchKernelGGL(nll_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gloss,
                       target, weight, wsum, M, C, ignore_index, glogp);
    return PN2_LAUNCH_RC();
}
