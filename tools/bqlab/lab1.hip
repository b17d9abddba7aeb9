// Lab 1: where do the 19.6 us of the SA1 ball-query kernel go?
//   E1 empty kernels at the candidate launch shapes, E2 store-only kernels (29 MB, plain / nt / sc1),
//   E3 the production cell-pruned kernel with phase stamps (s_memrealtime, 100 MHz).
// Build: tools/bqlab/build.sh lab1 ; run on the GPU box: tools/bqlab/lab1 [cube|facade]
#include "lab_common.h"

__device__ unsigned long long *pn2_stamp_buf;
#define PN2_STAMP(i) do { if (threadIdx.x == 0 && pn2_stamp_buf) pn2_stamp_buf[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PN2_STAMP_DRAIN() __builtin_amdgcn_s_waitcnt(0)
namespace pn2 { int tune_get(const char *, int d) { return d; } }
#include "../../khairil_tum-facade_semantic_segmentation_amd/csrc/pn2_ball_grid.hip"

// ---- E1 -----------------------------------------------------------------------------------------
__global__ void empty_kernel(int *p) { extern __shared__ char sm[]; if (p && threadIdx.x == 12345) p[0] = sm[0]; }

// ---- E2: every workgroup writes its contiguous share of `bytes` as float4, coalesced -------------------
template <int MODE>
__global__ void store_kernel(float4 *out, size_t n4_per_wg)
{
    float4 *o = out + (size_t)blockIdx.x * n4_per_wg;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f vv = {v.x, v.y, v.z, v.w};
    for (size_t i = threadIdx.x; i < n4_per_wg; i += blockDim.x) {
        if (MODE == 0) o[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(vv, reinterpret_cast<v4f *>(&o[i]));
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(&o[i]), "v"(vv) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(&o[i]), "v"(vv) : "memory");
    }
}

int main(int argc, char **argv)
{
    const bool facade = argc > 1 && !strcmp(argv[1], "facade");
    const int B = 16, N = 4096, C = 9, S = 1024, K = 32, D = 9;
    auto blocks = make_blocks(B, N, C, facade, 20231003);
    std::vector<float> xyz((size_t)B * N * 3);
    for (size_t i = 0; i < (size_t)B * N; ++i) for (int c = 0; c < 3; ++c) xyz[i * 3 + c] = blocks[i * C + c];
    std::vector<int64_t> start(B);
    for (int b = 0; b < B; ++b) start[b] = (b * 977) % N;
    float *d_pts, *d_xyz, *d_new, *d_grouped; int64_t *d_start, *d_fps, *d_idx; int32_t *d_err;
    CK(hipMalloc(&d_pts, blocks.size() * 4)); CK(hipMalloc(&d_xyz, xyz.size() * 4)); CK(hipMalloc(&d_new, (size_t)B * S * 3 * 4));
    CK(hipMalloc(&d_grouped, (size_t)B * S * K * (3 + D) * 4)); CK(hipMalloc(&d_start, B * 8)); CK(hipMalloc(&d_fps, (size_t)B * S * 8));
    CK(hipMalloc(&d_idx, (size_t)B * S * K * 8)); CK(hipMalloc(&d_err, 4));
    CK(hipMemcpy(d_pts, blocks.data(), blocks.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_xyz, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_start, start.data(), B * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_err, 0, 4));
    int rc = pn2_farthest_point_sample(d_xyz, B, N, S, d_start, d_fps, d_new, d_err, nullptr);
    if (rc) { fprintf(stderr, "fps rc %d\n", rc); return 1; }
    CK(hipDeviceSynchronize());

    if (!getenv("LAB_only_e3")) {
    // E1
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    struct { int g, t, lds; } shapes[] = {{256, 1024, 125 * 1024}, {256, 1024, 0}, {512, 512, 60 * 1024}, {1024, 256, 30 * 1024}, {2048, 256, 30 * 1024}, {256, 256, 0}};
    for (auto &s : shapes) {
        double us = time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(s.g), dim3(s.t), s.lds, 0, (int *)nullptr); });
        printf("E1 empty grid %4d x %4d threads, LDS %6d B: %6.2f us/launch\n", s.g, s.t, s.lds, us);
    }
    // E2
    const size_t out_bytes = (size_t)B * S * K * (3 + D) * 4 + (size_t)B * S * K * 8;   // grouped + idx = 29.4 MB
    float4 *d_out; CK(hipMalloc(&d_out, out_bytes));
    const char *mname[] = {"plain", "nt", "sc1", "sc0sc1"};
    struct { int g, t; } sshapes[] = {{256, 1024}, {512, 512}, {1024, 256}, {2048, 256}, {4096, 256}};
    for (auto &s : sshapes) {
        const size_t n4 = out_bytes / 16 / s.g;
        double us[4];
        us[0] = time_us([&] { hipLaunchKernelGGL(store_kernel<0>, dim3(s.g), dim3(s.t), 0, 0, d_out, n4); });
        us[1] = time_us([&] { hipLaunchKernelGGL(store_kernel<1>, dim3(s.g), dim3(s.t), 0, 0, d_out, n4); });
        us[2] = time_us([&] { hipLaunchKernelGGL(store_kernel<2>, dim3(s.g), dim3(s.t), 0, 0, d_out, n4); });
        us[3] = time_us([&] { hipLaunchKernelGGL(store_kernel<3>, dim3(s.g), dim3(s.t), 0, 0, d_out, n4); });
        for (int m = 0; m < 4; ++m)
            printf("E2 store %5.1f MB grid %4d x %4d %-6s: %6.2f us (%5.2f TB/s)\n", out_bytes / 1e6, s.g, s.t, mname[m], us[m], out_bytes / us[m] / 1e6);
    }
    }
    // E3: production kernel, timing + stamps
    {
        double us_api = time_us([&] { pn2_ball_query_group(0.1, K, d_xyz, d_new, d_pts, B, N, S, D, d_idx, d_grouped, 0, d_err, nullptr); });
        printf("E3 library pn2_ball_query_group (SA1 %s): %6.2f us/launch\n", facade ? "facade" : "cube", us_api);
        const size_t lds = gr_lds_bytes(K);
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ball_query_group_grid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const int tiles = S / GR_CENT, nwg = B * tiles;
        const float r2 = (float)(0.1 * 0.1);
        const unsigned magic = (unsigned)((1ULL << 32) / 3u) + 1u;
        auto launch = [&] {
            hipLaunchKernelGGL(ball_query_group_grid_kernel, dim3(nwg), dim3(GR_THREADS), lds, 0, d_xyz, d_new, d_pts, B, N, S, K, D, 3 + D, r2,
                               tiles, magic, d_idx, d_grouped, d_err);
        };
        unsigned long long *d_st, *nul = nullptr;
        CK(hipMalloc(&d_st, (size_t)nwg * 16 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(pn2_stamp_buf), &nul, sizeof(nul)));
        printf("E3 lab copy of the kernel, stamps off: %6.2f us/launch\n", time_us(launch));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(pn2_stamp_buf), &d_st, sizeof(d_st)));
        printf("E3 lab copy of the kernel, stamps on : %6.2f us/launch\n", time_us(launch));
        CK(hipMemset(d_st, 0, (size_t)nwg * 16 * 8));
        launch();
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st((size_t)nwg * 16);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int w = 0; w < nwg; ++w) { t0 = std::min(t0, st[w * 16]); t1 = std::max(t1, st[w * 16 + 12]); }
        printf("E3 stamps (us after the first workgroup's entry; 10 ns ticks): first entry -> last exit %.2f us\n", (t1 - t0) * 0.01);
        const char *names[13] = {"entry", "loads issued", "bbox partials", "barrier 1 (bbox)", "histogram done", "barrier 2", "scan done (2 barriers)",
                                 "scatter + barrier", "candidates", "first-K", "idx stores issued", "row gathers+stores issued", "stores drained"};
        for (int i = 0; i < 13; ++i) {
            std::vector<double> v;
            for (int w = 0; w < nwg; ++w) v.push_back((st[w * 16 + i] - t0) * 0.01);
            std::sort(v.begin(), v.end());
            printf("   %2d %-28s min %6.2f  median %6.2f  max %6.2f\n", i, names[i], v[0], v[nwg / 2], v[nwg - 1]);
        }
    }
    return 0;
}
