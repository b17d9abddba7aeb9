// Shared device helpers for libpn2hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "pn2_hip.h"

#pragma clang fp contract(off)

#define PN2_EXPORT extern "C" __attribute__((visibility("default")))
#define PN2_WAVE 64

#define PN2_REQUIRE_PTR(p) \
    do { if ((p) == nullptr) return PN2_ERR_NULL; } while (0)
#define PN2_LAUNCH_RC() ((int)hipGetLastError())

// Phase stamps for the lab programs under tools/ (which define PN2_STAMP before including a kernel source);
// nothing in the library build.
#ifndef PN2_STAMP
#define PN2_STAMP(i) do { } while (0)
#define PN2_STAMP_DRAIN() do { } while (0)
#endif

// The kernels of the training step's main branch raise their waves' issue priority: the geometry branch of the same graph
// (farthest point sampling: two vector-ALU-bound waves per SIMD on 16 CUs for a fifth of the step) shares SIMDs with
// their workgroups, and a main-branch kernel ends with its slowest workgroup.  The sampling chain has slack.
#define PN2_MAIN_BRANCH_PRIORITY() __builtin_amdgcn_s_setprio(3)

namespace pn2 {
// Rows of a tensor the NEXT kernel reads (on any XCD) are written through to memory -- agent-scope atomic stores / sc1
// buffer stores -- so that they stream out while the kernel computes instead of leaving the launch to end with the
// write-back of its dirty L2 lines.  Unconditional on purpose: a run-time switch around the store inside the unrolled
// epilogues cost 80 us per training step whichever way it was set (the stores were no longer straight-line code).
typedef int wt_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_rows4(float *base, size_t elem, float4 v, size_t total_bytes)
{
    if (total_bytes < 0xffffff00ull) {                // uniform; 16-byte form through a buffer descriptor (aux 16 = sc1)
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(unsigned)total_bytes, 0x00020000);
        wt_v4i w;
        w.x = __float_as_int(v.x); w.y = __float_as_int(v.y); w.z = __float_as_int(v.z); w.w = __float_as_int(v.w);
        __builtin_amdgcn_raw_buffer_store_b128(w, r, (int)(unsigned)(elem * 4), 0, 16);
    } else {
        *reinterpret_cast<float4 *>(base + elem) = v;
    }
}

__device__ __forceinline__ void store_rows(float *dst, float v)
{
    __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Developer tuning knob: PN2_TUNE_<NAME>=<int> in the environment overrides a launch heuristic.  The environment
// is scanned ONCE, when the first launcher asks (pn2_gather.hip); later changes of the environment are not seen,
// and a launch costs no getenv.  Benchmarks only, never set in production.
int tune_get(const char *name, int dflt);

// Per-device memo of a small host-side fact about a kernel (its dynamic-LDS attribute has been raised, its
// occupancy): the library keeps no other state, and this one is idempotent -- two threads racing on the same
// slot compute the same value.  Devices beyond 64 share slots modulo 64 (the memo is then merely recomputed).
struct PerDevice {
    std::atomic<int> v[64];
    PerDevice() { for (auto &x : v) x.store(-1, std::memory_order_relaxed); }
    static int device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? (d & 63) : 0; }
    int get() const { return v[device()].load(std::memory_order_acquire); }
    void set(int x) { v[device()].store(x, std::memory_order_release); }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device); `memo` is the call site's static PerDevice.
inline int ensure_dynamic_lds(const void *fn, int bytes, PerDevice &memo)
{
    if (memo.get() >= bytes) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    memo.set(bytes);
    return 0;
}

// dW[n][k] = sum_p partial[p][n][k], db[n] = sum_p partial[p][n][K] over P slabs [N][K+1], fixed order
// (pn2_mlp.hip); shared by the weight-gradient launchers.
int launch_dw_reduce(const float *partial, int P, int N, int K, float *dw, float *db, hipStream_t stream);
// the same plus, in the same launch, the BatchNorm-backward finalize of stat_partial [P][2][C] (pn2_bn_bwd_finalize)
int launch_bwd_post(const float *dw_partial, int P, int N, int K, float *dw, float *db, const float *stat_partial, int Ps,
                    int C, double count, float *dgamma, float *dbeta, float *c1, float *c2, hipStream_t stream);

// |p|^2 exactly as torch.sum(p ** 2, -1) evaluates it: ((x*x + y*y) + z*z), every op rounded
// (reference models/pointnet2_utils.py:38-39; rule SURVEY.md 8a-2).
__device__ __forceinline__ float norm3(float x, float y, float z)
{
    float xx = x * x;
    float yy = y * y;
    float zz = z * z;
    return (xx + yy) + zz;
}

// square_distance(src=a, dst=b) for one pair, bit-for-bit the reference CPU result
// (models/pointnet2_utils.py:37-39): dot is a k-ordered fma chain (sgemm with K=3), then
// (-2*dot + |a|^2) + |b|^2.  -2*dot is exact, so fma(-2,dot,na) rounds once like the add does.
__device__ __forceinline__ float pair_sqdist(float ax, float ay, float az, float na,
                                             float bx, float by, float bz, float nb)
{
    float dot = __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
    float d = __builtin_fmaf(-2.0f, dot, na);
    return d + nb;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (PN2_WAVE - 1)); }

// 64-lane prefix popcount of a ballot mask (number of set bits below this lane).
__device__ __forceinline__ int mbcnt(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}

// max over the 64 lanes of a signed int, result uniform in every lane.
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, dpp_i32<0xB1>(v));   // quad_perm [1,0,3,2]
    v = max(v, dpp_i32<0x4E>(v));   // quad_perm [2,3,0,1]
    v = max(v, dpp_i32<0x141>(v));  // row_half_mirror
    v = max(v, dpp_i32<0x140>(v));  // row_mirror: every lane now holds its 16-lane row max
    int r0 = __builtin_amdgcn_readlane(v, 0);
    int r1 = __builtin_amdgcn_readlane(v, 16);
    int r2 = __builtin_amdgcn_readlane(v, 32);
    int r3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(r0, r1), max(r2, r3));
}

// Bijective XCD-aware remap of a 1-D grid (cdna_hip_programming.md T1): workgroups that share
// `id % 8` (one XCD under round-robin dispatch) get a contiguous range of logical ids, so that
// workgroups working on the same 4096-point block hit the same L2.  Speed only.
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned nwg)
{
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = id & 7u, k = id >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + k;
}

}  // namespace pn2
