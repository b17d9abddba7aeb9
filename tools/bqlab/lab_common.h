// Shared helpers of the ball-query lab programs (developer probes; GPU box only, not part of the product).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include <string>
#include <functional>

#include "pn2_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    float uni() { return (float)((next() >> 40) * (1.0 / 16777216.0)); }
    float normal() { float u1 = uni() + 1e-7f, u2 = uni(); return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2); }
};

// blocks [B][N][C]: cube x,y~U(-.5,.5) z~U(0,1); facade x~U(-.5,.5) y~N(0,.02) z~U(0,3); other columns U(0,1)
inline std::vector<float> make_blocks(int B, int N, int C, bool facade, uint64_t seed)
{
    std::vector<float> v((size_t)B * N * C);
    Rng r(seed);
    for (size_t i = 0; i < (size_t)B * N; ++i) {
        float *p = &v[i * C];
        p[0] = r.uni() - 0.5f;
        p[1] = facade ? 0.02f * r.normal() : r.uni() - 0.5f;
        p[2] = facade ? 3.0f * r.uni() : r.uni();
        for (int c = 3; c < C; ++c) p[c] = r.uni();
    }
    return v;
}

// average microseconds per call of fn over `reps` back-to-back calls, median of `rounds`
inline double time_us(const std::function<void()> &fn, int reps = 50, int rounds = 7)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) fn();
    CK(hipDeviceSynchronize());
    std::vector<double> t;
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms * 1e3 / reps);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return t[t.size() / 2];
}
