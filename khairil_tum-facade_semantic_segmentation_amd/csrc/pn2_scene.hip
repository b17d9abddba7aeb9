// Whole-scene inference aggregation on gfx950.
// Reference: add_vote, localfunctions.py:339-346 -- a Python double loop over B x N that adds one
// vote per predicted label into vote_label_pool[point_idx, pred] when the sample weight is
// neither 0 nor inf -- fed by `seg_pred.max(2)[1]` computed on the host (:399).  Here one kernel
// takes the log-probabilities as they leave the network, does the arg-max (first maximum wins,
// like torch.max) and scatters the votes with integer atomics (order-independent, exact).
#include <math.h>

#include "pn2_common.h"

namespace {

__global__ __launch_bounds__(256) void vote_kernel(const float *__restrict__ logp, const int64_t *__restrict__ pred,
                                                   const int64_t *__restrict__ point_idx,
                                                   const float *__restrict__ weight, long long M, int C,
                                                   long long P, int32_t *__restrict__ pool, int32_t *err_count)
{
    for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M;
         m += (long long)gridDim.x * blockDim.x) {
        const float w = weight ? weight[m] : 1.0f;
        if (w == 0.0f || isinf(w)) continue;                       // :344
        int label;
        if (logp) {
            const float *row = logp + (size_t)m * C;
            float best = row[0];
            label = 0;
            for (int c = 1; c < C; ++c) {
                const float v = row[c];
                if (v > best) { best = v; label = c; }             // first maximum wins (torch.max)
            }
        } else {
            label = (int)pred[m];
        }
        const int64_t p = point_idx[m];
        if (p < 0 || p >= P || label < 0 || label >= C) {
            if (err_count) atomicAdd(err_count, 1);
            continue;
        }
        atomicAdd(pool + (size_t)p * C + label, 1);                // :345
    }
}

}  // namespace

PN2_EXPORT int pn2_add_vote(const float *logp, const int64_t *pred_label, const int64_t *point_idx,
                            const float *weight, long long M, int C, long long P, int32_t *vote_pool,
                            int32_t *err_count, pn2_stream_t stream_)
{
    PN2_REQUIRE_PTR(point_idx);
    PN2_REQUIRE_PTR(vote_pool);
    if (logp == nullptr && pred_label == nullptr) return PN2_ERR_NULL;
    if (M < 0 || C <= 0 || P <= 0) return PN2_ERR_SHAPE;
    if (M == 0) return PN2_OK;
    long long blocks = (M + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vote_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), logp,
                       pred_label, point_idx, weight, M, C, P, vote_pool, err_count);
    return PN2_LAUNCH_RC();
}
