// Adam over ONE flat fp32 parameter buffer (the optimiser of the reference loop:
// torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=decay_rate), sem_seg_training.py:576-582).
// The model's 92 parameter tensors are views of one buffer and their gradients are packed into one
// buffer for the data-parallel all-reduce anyway, so the update is a single elementwise pass instead of
// three multi-tensor kernels over 92 small tensors.  Same arithmetic as torch's Adam (L2 weight decay added to
// the gradient, bias-corrected moments, eps added after the square root), fp32 state, step counter and
// learning rate live on the device so that the launch can be replayed from a hipGraph.
#include <math.h>

#include "pn2_common.h"

namespace {

// state[0] = step count (as float), state[1] = 1 - beta1^t, state[2] = sqrt(1 - beta2^t) for the step about to run
__global__ void adam_tick_kernel(float *__restrict__ state, double beta1, double beta2)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double t = (double)state[0] + 1.0;
    state[0] = (float)t;
    state[1] = (float)(1.0 - pow(beta1, t));
    state[2] = (float)sqrt(1.0 - pow(beta2, t));
}

__global__ __launch_bounds__(256) void adam_step_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                        float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                        long long n, const float *__restrict__ lr, const float *__restrict__ state,
                                                        float beta1, float beta2, float eps, float weight_decay, float grad_scale)
{
    PN2_MAIN_BRANCH_PRIORITY();
    const float step_size = *lr / state[1];
    const float bc2_sqrt = state[2];
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n) {
        float4 p = *reinterpret_cast<float4 *>(param + i4);
        const float4 g0 = *reinterpret_cast<const float4 *>(grad + i4);
        float4 m = *reinterpret_cast<float4 *>(exp_avg + i4);
        float4 v = *reinterpret_cast<float4 *>(exp_avg_sq + i4);
#define PN2_ADAM(f)                                                             \
    {                                                                           \
        const float g = g0.f * grad_scale + weight_decay * p.f;                 \
        m.f = m.f + (1.0f - beta1) * (g - m.f);                                 \
        v.f = beta2 * v.f + (1.0f - beta2) * g * g;                             \
        p.f = p.f - step_size * (m.f / (sqrtf(v.f) / bc2_sqrt + eps));          \
    }
        PN2_ADAM(x) PN2_ADAM(y) PN2_ADAM(z) PN2_ADAM(w)
#undef PN2_ADAM
        *reinterpret_cast<float4 *>(param + i4) = p;
        *reinterpret_cast<float4 *>(exp_avg + i4) = m;
        *reinterpret_cast<float4 *>(exp_avg_sq + i4) = v;
    } else {
        for (long long i = i4; i < n; ++i) {
            const float g = grad[i] * grad_scale + weight_decay * param[i];
            const float m = exp_avg[i] + (1.0f - beta1) * (g - exp_avg[i]);
            const float v = beta2 * exp_avg_sq[i] + (1.0f - beta2) * g * g;
            exp_avg[i] = m;
            exp_avg_sq[i] = v;
            param[i] = param[i] - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
        }
    }
}

// The same update with the gradients left where backward wrote them: one pointer per parameter tensor (null = no
// gradient = zeros) and the tensors' element offsets in the flat parameter buffer, passed BY VALUE (a captured launch
// keeps them: the graph's tensors do not move).  A single process has no all-reduce to pack for, so this drops the
// concatenation kernel from the step.
constexpr int ADAM_MAX_TENSORS = 192;
struct AdamTensors {
    int n;
    unsigned off[ADAM_MAX_TENSORS + 1];          // off[n] = total elements
    const float *grad[ADAM_MAX_TENSORS];
};

__global__ __launch_bounds__(256) void adam_step_scattered_kernel(float *__restrict__ param, AdamTensors t,
                                                                  float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                                  const float *__restrict__ lr, const float *__restrict__ state,
                                                                  float beta1, float beta2, float eps, float weight_decay,
                                                                  float grad_scale)
{
    PN2_MAIN_BRANCH_PRIORITY();
    __shared__ unsigned sOff[ADAM_MAX_TENSORS + 1];
    __shared__ const float *sGrad[ADAM_MAX_TENSORS];
    for (int i = threadIdx.x; i <= t.n; i += 256) sOff[i] = t.off[i];
    for (int i = threadIdx.x; i < t.n; i += 256) sGrad[i] = t.grad[i];
    __syncthreads();
    const unsigned n = sOff[t.n];
    const float step_size = *lr / state[1];
    const float bc2_sqrt = state[2];
    const unsigned i4 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (i4 >= n) return;
    int lo = 0, hi = t.n - 1;                    // the tensor that holds element i4
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (sOff[mid] <= i4) lo = mid; else hi = mid - 1;
    }
    int j = lo;
    float g0[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned i = i4 + e;
        while (j + 1 < t.n && i >= sOff[j + 1]) ++j;
        const float *gp = sGrad[j];
        g0[e] = (i < n && gp) ? gp[i - sOff[j]] : 0.0f;
    }
#define PN2_ADAM1(P, G, M, V)                                                   \
    {                                                                           \
        const float g = (G) * grad_scale + weight_decay * (P);                  \
        (M) = (M) + (1.0f - beta1) * (g - (M));                                 \
        (V) = beta2 * (V) + (1.0f - beta2) * g * g;                             \
        (P) = (P) - step_size * ((M) / (sqrtf(V) / bc2_sqrt + eps));            \
    }
    if (i4 + 4 <= n) {
        float4 p = *reinterpret_cast<float4 *>(param + i4);
        float4 m = *reinterpret_cast<float4 *>(exp_avg + i4);
        float4 v = *reinterpret_cast<float4 *>(exp_avg_sq + i4);
        PN2_ADAM1(p.x, g0[0], m.x, v.x) PN2_ADAM1(p.y, g0[1], m.y, v.y) PN2_ADAM1(p.z, g0[2], m.z, v.z) PN2_ADAM1(p.w, g0[3], m.w, v.w)
        *reinterpret_cast<float4 *>(param + i4) = p;
        *reinterpret_cast<float4 *>(exp_avg + i4) = m;
        *reinterpret_cast<float4 *>(exp_avg_sq + i4) = v;
    } else {
        for (unsigned i = i4; i < n; ++i) {
            float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
            PN2_ADAM1(p, g0[i - i4], m, v)
            param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
        }
    }
#undef PN2_ADAM1
}

}  // namespace

PN2_EXPORT int pn2_adam_step_scattered(float *param, int n_tensors, const float *const *grads, const long long *offsets,
                                       float *exp_avg, float *exp_avg_sq, const float *lr, float *state, double beta1,
                                       double beta2, double eps, double weight_decay, double grad_scale, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(param); PN2_REQUIRE_PTR(grads); PN2_REQUIRE_PTR(offsets); PN2_REQUIRE_PTR(exp_avg);
    PN2_REQUIRE_PTR(exp_avg_sq); PN2_REQUIRE_PTR(lr); PN2_REQUIRE_PTR(state);
    if (n_tensors <= 0) return PN2_ERR_SHAPE;
    if (n_tensors > ADAM_MAX_TENSORS || offsets[n_tensors] >= (1LL << 32) - 4) return PN2_ERR_UNSUPPORTED;
    const uintptr_t al = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) |
                         reinterpret_cast<uintptr_t>(exp_avg_sq);
    if (al & 15) return PN2_ERR_UNSUPPORTED;
    AdamTensors t;
    t.n = n_tensors;
    for (int i = 0; i <= ADAM_MAX_TENSORS; ++i) {
        const int k = i <= n_tensors ? i : n_tensors;
        if (offsets[k] < 0 || (k > 0 && offsets[k] < offsets[k - 1])) return PN2_ERR_SHAPE;
        t.off[i] = (unsigned)offsets[k];
        if (i < ADAM_MAX_TENSORS) {
            t.grad[i] = i < n_tensors ? grads[i] : nullptr;
            if (reinterpret_cast<uintptr_t>(t.grad[i]) & 3) return PN2_ERR_UNSUPPORTED;
        }
    }
    if (offsets[0] != 0 || offsets[n_tensors] <= 0) return PN2_ERR_SHAPE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, stream, state, beta1, beta2);
    const long long blocks = (offsets[n_tensors] + 1023) / 1024;
    hipLaunchKernelGGL(adam_step_scattered_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, param, t, exp_avg, exp_avg_sq, lr,
                       state, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)grad_scale);
    return PN2_LAUNCH_RC();
}

PN2_EXPORT int pn2_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n,
                             const float *lr, float *state, double beta1, double beta2, double eps, double weight_decay,
                             double grad_scale, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(param); PN2_REQUIRE_PTR(grad); PN2_REQUIRE_PTR(exp_avg); PN2_REQUIRE_PTR(exp_avg_sq);
    PN2_REQUIRE_PTR(lr); PN2_REQUIRE_PTR(state);
    if (n <= 0) return PN2_ERR_SHAPE;
    const uintptr_t al = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
                         reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq);
    if (al & 15) return PN2_ERR_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, stream, state, beta1, beta2);
    const long long blocks = (n + 1023) / 1024;
    if (blocks > 0x7fffffffLL) return PN2_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                       state, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)grad_scale);
    return PN2_LAUNCH_RC();
}
