// farthest_point_sample on gfx950.  Replaces the npoint-iteration torch loop of the reference
// (models/pointnet2_utils.py:63-84): one workgroup per 4096-point block, the points and their
// running min-distance live in registers for the whole kernel, one workgroup barrier per
// iteration.  Latency-bound by construction (npoint dependent argmax steps); see DESIGN.md.
#include "pn2_common.h"

namespace pn2 {
int launch_ball_bin(const float *xyz, const float *new_xyz, int B, int N, int S, int D, float r2, char *plans, hipStream_t stream,
                    const float *pack_points = nullptr, bool pack = false);
}

namespace {

// The per-iteration arg-max is what this kernel's time is made of (two waves per SIMD, ~170 instructions per iteration
// before this form): reductions with the DPP operand fused into v_max_i32 (the builtin form costs copy + wait + move + max
// per step), the last two steps across rows with row_bcast instead of four read-lanes, and the running minimum as one
// v_min_f32 (`d < md ? d : md` compiles to compare + wait + select; for the values that occur -- d >= 0 or NaN -- the
// IEEE minimum gives the same bits: a NaN d keeps md).
__device__ __forceinline__ int fps_row_max(int v)            // max over each 16-lane row, in every lane of the row
{
    asm("s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ int fps_wave_max(int v)           // max over the 64 lanes, uniform
{
    v = fps_row_max(v);
    asm("v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float fps_min(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// T threads, P consecutive points per thread (thread t owns points t*P .. t*P+P-1, so lane
// order == index order and "first lane with the max" == "lowest index with the max", the tie
// rule of torch.max (:83)).
template <int T, int P, bool LDS_XYZ>
__global__ __launch_bounds__(T) void fps_kernel(const float *__restrict__ xyz, int N, int npoint,
                                                const int64_t *__restrict__ start,
                                                int64_t *__restrict__ out_idx, float *__restrict__ new_xyz,
                                                int32_t *err_count)
{
    constexpr int W = T / PN2_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(smem);     // [2][W]
    float4 *pts = reinterpret_cast<float4 *>(smem + 2 * 16 * sizeof(unsigned long long));  // [T*P] if LDS_XYZ

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & (PN2_WAVE - 1);
    const int wave = tid / PN2_WAVE;
    const float *p = xyz + (size_t)b * N * 3;

    float px[P], py[P], pz[P], md[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int j = tid * P + k;
        const bool in = j < N;
        px[k] = in ? p[j * 3 + 0] : 0.0f;
        py[k] = in ? p[j * 3 + 1] : 0.0f;
        pz[k] = in ? p[j * 3 + 2] : 0.0f;
        md[k] = in ? 1e10f : -1.0f;          // :74; padding can never be the max (real values >= 0)
        if (LDS_XYZ) pts[j] = make_float4(px[k], py[k], pz[k], 0.0f);
    }

    long long far = start[b];                // :75 (drawn by the caller)
    if (far < 0 || far >= N) {
        if (tid == 0 && err_count) atomicAdd(err_count, 1);
        far = 0;
    }
    if (LDS_XYZ) __syncthreads();

    for (int i = 0; i < npoint; ++i) {
        float cx, cy, cz;
        if (LDS_XYZ) {
            const float4 c = pts[far];
            cx = c.x; cy = c.y; cz = c.z;
        } else {
            cx = p[far * 3 + 0]; cy = p[far * 3 + 1]; cz = p[far * 3 + 2];
        }
        if (tid == 0) {
            out_idx[(size_t)b * npoint + i] = far;                               // :78
            if (new_xyz) {
                float *o = new_xyz + ((size_t)b * npoint + i) * 3;
                o[0] = cx; o[1] = cy; o[2] = cz;
            }
        }
        if (i + 1 == npoint) break;

        int best = __float_as_int(-1.0f);
        int bestj = 0;
        float dk[P];
        if (P % 2 == 0) {
            // two points per packed instruction (v_pk_add / v_pk_mul: 8 for a pair instead of 12 scalar ones)
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
            for (int k = 0; k + 1 < P; k += 2) {
                const f2 x = {px[k], px[k + 1]}, y = {py[k], py[k + 1]}, z = {pz[k], pz[k + 1]};
                const f2 dx = x - c2x, dy = y - c2y, dz = z - c2z;
                const f2 d = (dx * dx + dy * dy) + dz * dz;                      // :80, un-fused
                dk[k] = d.x; dk[k + 1] = d.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
                dk[k] = (dx * dx + dy * dy) + dz * dz;
            }
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            md[k] = fps_min(dk[k], md[k]);                                       // :81-82 (mask = dist < distance)
            // non-negative floats (and the -1 padding) order like their bit patterns as signed ints
            const int bits = __float_as_int(md[k]);
            if (bits > best) { best = bits; bestj = tid * P + k; }               // strict: lowest k wins
        }
        const int wmax = fps_wave_max(best);
        const unsigned long long m = __ballot(best == wmax);
        const int wl = __builtin_ctzll(m);
        int widx = __builtin_amdgcn_readlane(bestj, wl);
        if (W > 1) {
            unsigned long long *s = slot + (i & 1) * W;
            if (lane == 0) s[wave] = ((unsigned long long)(unsigned)wmax << 32) | (unsigned)widx;
            __syncthreads();
            const unsigned long long e = s[lane & (W - 1)];
            const int ev = (int)(e >> 32);
            const int gmax = fps_row_max(ev);                                    // W <= 16 entries repeat in every row
            const unsigned long long gm = __ballot(ev == gmax);
            const int gl = __builtin_ctzll(gm);                                  // lowest wave == lowest index
            widx = __builtin_amdgcn_readlane((int)(unsigned)e, gl);
        }
        far = widx;                                                              // :83
    }
}

// LAB kernel (PN2_TUNE_lab_fps_dummy=<microseconds>, N >= 4096 only; wrong results by design): the footprint of
// fps_kernel<512,8> -- 512 threads, 64 registers per lane, its dynamic LDS -- held for the given time by sleeping waves, with the
// block's first npoint points handed out as "centroids".  It separates what the sampling kernel costs the step beside it by
// OCCUPYING 16 CUs from what its instructions cost.  0 us: the footprint for an instant.
template <bool REGS>
__global__ __launch_bounds__(512) void fps_dummy_kernel(const float *__restrict__ xyz, int N, int npoint, int64_t *__restrict__ out_idx,
                                                        float *__restrict__ new_xyz, int hold_us)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (REGS) asm volatile("v_mov_b32 v63, 0" ::: "v63");            // the register allocation of the real kernel
    if (tid == 0) smem[0] = 0;
    for (int i = tid; i < npoint; i += (int)blockDim.x) {
        const int j = i < N ? i : N - 1;
        if (out_idx) out_idx[(size_t)b * npoint + i] = j;
        if (new_xyz)
            for (int c = 0; c < 3; ++c) new_xyz[((size_t)b * npoint + i) * 3 + c] = xyz[((size_t)b * N + j) * 3 + c];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)hold_us * 100ull) __builtin_amdgcn_s_sleep(32);
}

template <int T, int P>
int launch_fps(const float *xyz, int B, int N, int npoint, const int64_t *start, int64_t *out_idx,
               float *new_xyz, int32_t *err_count, hipStream_t stream)
{
    if (T == 512 && P == 8) {
        const int hold = pn2::tune_get("lab_fps_dummy", -1);
        if (hold >= 0) {
            // which part of the footprint: PN2_TUNE_lab_fps_dummy_lds (bytes, default the kernel's 64 KB), _threads (512), _regs (1)
            const size_t lds = (size_t)pn2::tune_get("lab_fps_dummy_lds", (int)(2 * 16 * sizeof(unsigned long long) + (size_t)T * P * sizeof(float4)));
            const int threads = pn2::tune_get("lab_fps_dummy_threads", 512);
            static pn2::PerDevice memo, memo2;
            if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(fps_dummy_kernel<true>), 160 * 1024, memo)) return e;
            if (const int e = pn2::ensure_dynamic_lds(reinterpret_cast<const void *>(fps_dummy_kernel<false>), 160 * 1024, memo2)) return e;
            if (pn2::tune_get("lab_fps_dummy_regs", 1))
                hipLaunchKernelGGL(fps_dummy_kernel<true>, dim3(B), dim3(threads), lds, stream, xyz, N, npoint, out_idx, new_xyz, hold);
            else
                hipLaunchKernelGGL(fps_dummy_kernel<false>, dim3(B), dim3(threads), lds, stream, xyz, N, npoint, out_idx, new_xyz, hold);
            return PN2_LAUNCH_RC();
        }
    }
    const size_t slots = 2 * 16 * sizeof(unsigned long long);
    const bool lds_xyz = (size_t)T * P * sizeof(float4) <= 128 * 1024;
    if (lds_xyz) {
        const size_t lds = slots + (size_t)T * P * sizeof(float4);
        auto k = fps_kernel<T, P, true>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(k, dim3(B), dim3(T), lds, stream, xyz, N, npoint, start, out_idx, new_xyz, err_count);
    } else {
        hipLaunchKernelGGL((fps_kernel<T, P, false>), dim3(B), dim3(T), slots, stream, xyz, N, npoint, start,
                           out_idx, new_xyz, err_count);
    }
    return PN2_LAUNCH_RC();
}

}  // namespace

PN2_EXPORT int pn2_farthest_point_sample(const float *xyz, int B, int N, int npoint, const int64_t *start,
                                         int64_t *out_idx, float *new_xyz, int32_t *err_count,
                                         pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(xyz);
    PN2_REQUIRE_PTR(start);
    PN2_REQUIRE_PTR(out_idx);
    if (B < 0 || N <= 0 || npoint < 0) return PN2_ERR_SHAPE;
    if (N > 32768) return PN2_ERR_UNSUPPORTED;
    if (B == 0 || npoint == 0) return PN2_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
#define PN2_FPS_CASE(T, P) \
    if (N <= (T) * (P)) return launch_fps<T, P>(xyz, B, N, npoint, start, out_idx, new_xyz, err_count, stream)
    PN2_FPS_CASE(64, 1);
    PN2_FPS_CASE(64, 2);
    PN2_FPS_CASE(64, 4);
    PN2_FPS_CASE(128, 4);
    PN2_FPS_CASE(256, 4);
    PN2_FPS_CASE(512, 4);
    // (N <= 4096 as 256 threads x 16 points -- one wave per SIMD -- and as 1024 x 4 -- four per SIMD -- measured: 599 / 593 us
    // against 550 us for 512 x 8)
    PN2_FPS_CASE(512, 8);
    PN2_FPS_CASE(1024, 8);
    PN2_FPS_CASE(1024, 16);
    PN2_FPS_CASE(1024, 32);
#undef PN2_FPS_CASE
    return PN2_ERR_UNSUPPORTED;
}

// farthest_point_sample + the ball-query plan of every block (csrc/pn2_ball_bin.h) for `radius`: the sampling
// kernel, then the one-workgroup-per-block binning kernel on the same stream.  (Building the plan in the tail of the
// sampling kernel itself -- the block is in its registers -- was measured: the tail takes 8.6 us, but the extra
// live state changes how the compiler schedules the latency-bound sampling loop, +0.028 us on each of its 1024
// iterations = +29 us per call against 11 us for the separate kernel.)
PN2_EXPORT int pn2_farthest_point_sample_plan(const float *xyz, int B, int N, int npoint, const int64_t *start, int64_t *out_idx,
                                              float *new_xyz, double radius, int D, void *plans, int32_t *err_count,
                                              pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(new_xyz);
    PN2_REQUIRE_PTR(plans);
    if (D < 0 || npoint <= 0) return PN2_ERR_SHAPE;
    if (N > 8192) return PN2_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(plans) & 127) != 0) return PN2_ERR_SHAPE;
    const int rc = pn2_farthest_point_sample(xyz, B, N, npoint, start, out_idx, new_xyz, err_count, stream_);
    if (rc != PN2_OK || B == 0) return rc;
    const float r2 = (float)(radius * radius);          // python `radius ** 2` (double), compared in fp32
    return pn2::launch_ball_bin(xyz, new_xyz, B, N, npoint, D, r2, static_cast<char *>(plans), static_cast<hipStream_t>(stream_));
}
