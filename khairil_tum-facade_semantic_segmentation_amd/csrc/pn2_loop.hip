// Host -> device glue of the reference's training / evaluation loops, on the device.
//
// pn2_input_blocks: the per-step input preparation of modelTraining (localfunctions.py:205-209) -- rotate the xyz
// columns of every block about the up axis by its own angle (provider.rotate_point_cloud_z, provider.py:66-84:
// row-vector @ [[c, s, 0], [-s, c, 0], [0, 0, 1]]) and hand the network its layout -- as ONE pass: channel-first
// [B, C, N] (what the loop passes to the classifier) or channel-last [B, N, C] in, channel-last rows [B, N, C] and the
// contiguous xyz [B, N, 3] the sampling kernels read out.  The reference rotates on the host in numpy (float64
// matrix product rounded to float32) between two host<->device copies; here the angles stay on the device.
//
// pn2_seg_metrics: the per-batch accuracy / IoU bookkeeping (localfunctions.py:214, 220-223 for training;
// :271-283 for evaluation) accumulated into int64 device counters that the loop reads once per epoch, instead of a
// .cpu() of the whole prediction tensor every step: counters[0] = correct points, [1] = seen points, then per class c:
// [2 + c] = labels == c, [2 + C + c] = pred == c && label == c, [2 + 2C + c] = pred == c || label == c.
#include "pn2_common.h"

namespace {

__global__ __launch_bounds__(256) void input_blocks_kernel(const float *__restrict__ in, int channel_first, int N, int C,
                                                           const float *__restrict__ angles, float *__restrict__ pts,
                                                           float *__restrict__ xyz, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;       // one point
    if (t >= total) return;
    const long long b = t / N;
    const int n = (int)(t - b * N);
    const float *src = in + (size_t)b * N * C;
    float c = 1.0f, s = 0.0f;
    if (angles) { const float a = angles[b]; c = cosf(a); s = sinf(a); }
    float *o = pts + (size_t)t * C;
    float x = 0.f, y = 0.f, z = 0.f;
    // twelve channels per pass, every load issued before the first store (a `load; store` loop over the channels waits for
    // one memory round trip per channel: 9 of them for the 11.8 us this kernel took)
    for (int k0 = 0; k0 < C; k0 += 12) {
        float v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int k = k0 + u < C ? k0 + u : C - 1;
            v[u] = channel_first ? src[(size_t)k * N + n] : src[(size_t)n * C + k];
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int k = k0 + u;
            if (k < C) {
                if (k == 0) x = v[u];
                else if (k == 1) y = v[u];
                else if (k == 2) z = v[u];
                else o[k] = v[u];
            }
        }
    }
    const float xr = x * c - y * s, yr = x * s + y * c;          // provider.py:78-81
    o[0] = xr;
    if (C > 1) o[1] = yr;
    if (C > 2) o[2] = z;
    if (xyz) {
        float *q = xyz + (size_t)t * 3;
        q[0] = xr; q[1] = yr; q[2] = z;
    }
}

constexpr int MET_MAXC = 64;

__global__ __launch_bounds__(256) void seg_metrics_kernel(const float *__restrict__ logp, const int64_t *__restrict__ target,
                                                          long long M, int C, unsigned long long *__restrict__ counters)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ unsigned cnt[2 + 3 * MET_MAXC];
    for (int i = threadIdx.x; i < 2 + 3 * C; i += 256) cnt[i] = 0u;
    __syncthreads();
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long long)gridDim.x * 256) {
        const float *row = logp + (size_t)r * C;
        int best = 0;
        float bv = row[0];
        for (int k0 = 1; k0 < C; k0 += 8) {               // eight classes per pass, loaded before the first compare
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[k0 + u < C ? k0 + u : C - 1];
#pragma unroll
            for (int u = 0; u < 8; ++u)                   // first maximum wins, like Tensor.max(1)[1] / np.argmax
                if (k0 + u < C && v[u] > bv) { bv = v[u]; best = k0 + u; }
        }
        const long long lab = target[r];
        atomicAdd(&cnt[1], 1u);
        if (lab == best) atomicAdd(&cnt[0], 1u);
        if (lab >= 0 && lab < C) {
            atomicAdd(&cnt[2 + (int)lab], 1u);
            atomicAdd(&cnt[2 + 2 * C + (int)lab], 1u);    // union: label == c ...
            if (lab == best) atomicAdd(&cnt[2 + C + best], 1u);
        }
        if (lab != best) atomicAdd(&cnt[2 + 2 * C + best], 1u);   // ... or pred == c (counted once when they coincide)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 + 3 * C; i += 256)
        if (cnt[i]) atomicAdd(&counters[i], (unsigned long long)cnt[i]);
}

}  // namespace

PN2_EXPORT int pn2_input_blocks(const float *in, int channel_first, int B, int N, int C, const float *angles, float *pts,
                                float *xyz, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(in);
    PN2_REQUIRE_PTR(pts);
    if (B < 0 || N <= 0 || C < 3) return PN2_ERR_SHAPE;
    if (B == 0) return PN2_OK;
    const long long total = (long long)B * N;
    const long long nwg = (total + 255) / 256;
    if (nwg > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(input_blocks_kernel, dim3((unsigned)nwg), dim3(256), 0, static_cast<hipStream_t>(stream_), in, channel_first, N,
                       C, angles, pts, xyz, total);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_seg_metrics(const float *logp, const int64_t *target, long long M, int C, long long *counters,
                               pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(logp);
    PN2_REQUIRE_PTR(target);
    PN2_REQUIRE_PTR(counters);
    if (M < 0 || C <= 0) return PN2_ERR_SHAPE;
    if (C > MET_MAXC) return PN2_ERR_UNSUPPORTED;
    if (M == 0) return PN2_OK;
    long long nwg = (M + 256 * 8 - 1) / (256 * 8);        // ~8 rows per thread: few global atomics per launch
    if (nwg > 1024) nwg = 1024;
    hipLaunchKernelGGL(seg_metrics_kernel, dim3((unsigned)nwg), dim3(256), 0, static_cast<hipStream_t>(stream_), logp, target, M, C,
                       reinterpret_cast<unsigned long long *>(counters));
    return PN2_LAUNCH_RC();
}
