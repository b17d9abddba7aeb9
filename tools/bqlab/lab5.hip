// Lab 5: the second spatial-tile ball query (tools/bqlab/pn2_ball_tile2.hip; NOT part of the library): 512-thread workgroups,
// two per CU, a local cell grid over the tile's candidates, bitmaps by index-order rank.  Checks against the library's
// cell-pruned kernel, times it, prints phase stamps (profiles/r03/lab5_*.log).
#include "lab_common.h"

__device__ unsigned long long *pn2_stamp_buf;
#define PN2_STAMP(i) do { if ((threadIdx.x & 511) == 0 && pn2_stamp_buf) pn2_stamp_buf[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PN2_STAMP_DRAIN() __builtin_amdgcn_s_waitcnt(0)
namespace pn2 { int tune_get(const char *name, int d) { std::string k = std::string("LAB_") + name; const char *v = getenv(k.c_str()); return v ? atoi(v) : d; } }
#include "pn2_ball_tile2.hip"

static void stamp_report(const char *what, unsigned long long *d_st, int nwg, int nst, const char **names)
{
    std::vector<unsigned long long> st((size_t)nwg * 16);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < nwg; ++w) { if (st[w * 16]) t0 = std::min(t0, st[w * 16]); for (int i = 0; i < nst; ++i) t1 = std::max(t1, st[w * 16 + i]); }
    printf("%s: first entry -> last exit %.2f us\n", what, (t1 - t0) * 0.01);
    for (int i = 0; i < nst; ++i) {
        std::vector<double> v;
        for (int w = 0; w < nwg; ++w) if (st[w * 16 + i]) v.push_back((st[w * 16 + i] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("   %2d %-26s n %5zu  min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f\n", i, names[i], v.size(), v[0], v[v.size() / 10],
               v[v.size() / 2], v[v.size() * 9 / 10], v.back());
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
    float *d_pts, *d_xyz, *d_new, *d_grouped, *d_grouped2; int64_t *d_start, *d_fps, *d_idx, *d_idx2; int32_t *d_err;
    const size_t gbytes = (size_t)B * S * K * (3 + D) * 4, ibytes = (size_t)B * S * K * 8;
    CK(hipMalloc(&d_pts, blocks.size() * 4)); CK(hipMalloc(&d_xyz, xyz.size() * 4)); CK(hipMalloc(&d_new, (size_t)B * S * 3 * 4));
    CK(hipMalloc(&d_grouped, gbytes)); CK(hipMalloc(&d_grouped2, gbytes)); CK(hipMalloc(&d_start, B * 8)); CK(hipMalloc(&d_fps, (size_t)B * S * 8));
    CK(hipMalloc(&d_idx, ibytes)); CK(hipMalloc(&d_idx2, ibytes)); CK(hipMalloc(&d_err, 4));
    CK(hipMemcpy(d_pts, blocks.data(), blocks.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_xyz, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_start, start.data(), B * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_err, 0, 4));
    int rc = pn2_farthest_point_sample(d_xyz, B, N, S, d_start, d_fps, d_new, d_err, nullptr);
    if (rc) { fprintf(stderr, "fps rc %d\n", rc); return 1; }
    rc = pn2_ball_query_group_select(1, 0.1, K, d_xyz, d_new, d_pts, B, N, S, D, d_idx, d_grouped, 0, d_err, nullptr);
    if (rc) { fprintf(stderr, "ref rc %d\n", rc); return 1; }
    CK(hipDeviceSynchronize());
    const float r2 = (float)(0.1 * 0.1);
    auto lab = [&]() { return pn2::launch_ball_query_tile2(d_xyz, d_new, d_pts, B, N, S, K, D, 3 + D, r2, d_idx2, d_grouped2, d_err, nullptr); };
    CK(hipMemset(d_idx2, 0xff, ibytes)); CK(hipMemset(d_grouped2, 0xff, gbytes));
    rc = lab();
    if (rc) { fprintf(stderr, "lab rc %d\n", rc); return 1; }
    CK(hipDeviceSynchronize());
    {
        std::vector<int64_t> a((size_t)B * S * K), b2((size_t)B * S * K);
        std::vector<float> ga(gbytes / 4), gb(gbytes / 4);
        CK(hipMemcpy(a.data(), d_idx, ibytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(b2.data(), d_idx2, ibytes, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ga.data(), d_grouped, gbytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(gb.data(), d_grouped2, gbytes, hipMemcpyDeviceToHost));
        size_t bad = 0, gbad = 0;
        for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b2[i];
        for (size_t i = 0; i < ga.size(); ++i) gbad += memcmp(&ga[i], &gb[i], 4) != 0;
        printf("CHECK %s: idx mismatches %zu / %zu, grouped mismatching floats %zu / %zu\n", facade ? "facade" : "cube", bad, a.size(), gbad, ga.size());
    }
    if (argc > 2 && !strcmp(argv[2], "prof")) {
        for (int i = 0; i < 3; ++i) lab();
        CK(hipDeviceSynchronize());
        return 0;
    }
    printf("tile kernel: %.2f us per launch (back to back)\n", time_us([&]() { lab(); }));
    int tmax = 1;
    while (tmax < 64 && tmax * 2 * 32 <= S) tmax *= 2;
    const int nwg = B * tmax;
    unsigned long long *d_st;
    CK(hipMalloc(&d_st, (size_t)nwg * 16 * 8));
    CK(hipMemset(d_st, 0, (size_t)nwg * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(pn2_stamp_buf), &d_st, sizeof(d_st)));
    lab();
    CK(hipDeviceSynchronize());
    const char *names[] = {"entry", "loads landed, wave max", "box known", "classified + histogram", "scan done", "scattered", "tests + first K",
                           "idx + features in LDS", "rows stored", "drained"};
    stamp_report("tile2 kernel", d_st, nwg, 10, names);
    return 0;
}
