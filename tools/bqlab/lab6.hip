// Lab 6: the clock the chip holds in an fp32-MFMA loop (developer probe; GPU box only, not part of the product).
// Every wave runs `iters` x 4 independent v_mfma_f32_32x32x2_f32 on operands from memory (random or zero), re-reading
// its operands from LDS every step like the MLP kernels' multiplier waves do; around the loop it reads s_memtime (core
// clock ticks) and s_memrealtime (100 MHz): in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, "DVFS give-back", item 6).  Output: median clock over workgroups, ns per MFMA, TFLOP/s.
#include "lab_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_clock_kernel(const float *__restrict__ src, int iters, int mode, float *__restrict__ sink,
                                                        unsigned long long *__restrict__ stamps)
{
    __shared__ __attribute__((aligned(16))) float tile[64 * 68 + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 68; i += 256) tile[i] = src[(blockIdx.x * 64 * 68 + i) & ((1 << 20) - 1)];
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) {                       // read -> wait -> multiply
        for (int it = 0; it < iters; ++it) {
            const float *row = &tile[(it & 31) * 68];
            const float a0 = row[lane & 31], a1 = row[32 + (lane & 31)], b0 = row[68 + (lane & 31)], b1 = row[68 + 32 + (lane & 31)];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    } else if (mode == 1) {                // the next step's operands read before this step's MFMAs (the MLP kernels' loop)
        const float *row = &tile[0];
        float a0 = row[lane & 31], a1 = row[32 + (lane & 31)], b0 = row[68 + (lane & 31)], b1 = row[68 + 32 + (lane & 31)];
        for (int it = 0; it < iters; ++it) {
            const float *nx = &tile[((it + 1) & 31) * 68];
            const float a0n = nx[lane & 31], a1n = nx[32 + (lane & 31)], b0n = nx[68 + (lane & 31)], b1n = nx[68 + 32 + (lane & 31)];
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
        }
    } else if (mode == 2) {                // operands in registers: the issue rate of the matrix pipe alone
        const float a0 = tile[lane], a1 = tile[64 + lane], b0 = tile[128 + lane], b1 = tile[192 + lane];
        for (int it = 0; it < iters; ++it) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    } else if (mode == 4 || mode == 5) {   // 4 ds_read_b128 feed 16 MFMAs (k-contiguous operands); 5: next group read ahead, ping-pong registers
        const float4 *t4 = reinterpret_cast<const float4 *>(tile);
        const int o = (lane & 31) * 17 + (lane >> 5);                 // pitch 68 floats = 17 float4: conflict-free b128
        auto rd = [&](int g, float4 &A0, float4 &A1, float4 &B0, float4 &B1) {
            const float4 *q = t4 + ((g & 7) * 2);
            A0 = q[o]; A1 = q[o + 4]; B0 = q[o + 8]; B1 = q[o + 12];
        };
#define MM16(A0, A1, B0, B1) \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x, B0.x, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x, B1.x, acc[1], 0, 0, 0); \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x, B0.x, acc[2], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x, B1.x, acc[3], 0, 0, 0); \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.y, B0.y, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.y, B1.y, acc[1], 0, 0, 0); \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.y, B0.y, acc[2], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.y, B1.y, acc[3], 0, 0, 0); \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.z, B0.z, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.z, B1.z, acc[1], 0, 0, 0); \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.z, B0.z, acc[2], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.z, B1.z, acc[3], 0, 0, 0); \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.w, B0.w, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.w, B1.w, acc[1], 0, 0, 0); \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.w, B0.w, acc[2], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.w, B1.w, acc[3], 0, 0, 0);
        if (mode == 4) {
            for (int it = 0; it < iters / 4; ++it) {
                float4 A0, A1, B0, B1;
                rd(it, A0, A1, B0, B1);
                MM16(A0, A1, B0, B1)
            }
        } else {
            float4 A0, A1, B0, B1, C0, C1, D0, D1;
            rd(0, A0, A1, B0, B1);
            for (int it = 0; it < iters / 4; it += 2) {
                rd(it + 1, C0, C1, D0, D1);
                __builtin_amdgcn_sched_barrier(0);
                MM16(A0, A1, B0, B1)
                __builtin_amdgcn_sched_barrier(0);
                rd(it + 2, A0, A1, B0, B1);
                __builtin_amdgcn_sched_barrier(0);
                MM16(C0, C1, D0, D1)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (mode == 6 || mode == 7 || mode == 8) {
        // the forward GEMMs' loop: one 32 x 32 block per wave, 2 ds_read_b128 (A, B: four k each) feed 4 MFMAs.
        // 6: read-ahead with register moves (a4 = an; b4 = bn), as the kernels are written; 7: the same unrolled by two
        // with alternating registers (no moves); 8: read -> multiply, no read-ahead at all
        const float4 *t4 = reinterpret_cast<const float4 *>(tile);
        const int o = (lane & 31) * 17 + (lane >> 5);
#define MM4(A, B) \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, B.x, acc[0], 0, 0, 0); acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, B.y, acc[0], 0, 0, 0); \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, B.z, acc[0], 0, 0, 0); acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, B.w, acc[0], 0, 0, 0);
        if (mode == 6) {
            float4 a4 = t4[o], b4 = t4[o + 8];
            for (int it = 0; it < iters; ++it) {
                const float4 an = t4[o + 2 * ((it + 1) & 7)], bn = t4[o + 8 + 2 * ((it + 1) & 7)];
                __builtin_amdgcn_sched_barrier(0);
                MM4(a4, b4)
                a4 = an; b4 = bn;
            }
        } else if (mode == 7) {
            float4 a4 = t4[o], b4 = t4[o + 8], an, bn;
            for (int it = 0; it < iters; it += 2) {
                an = t4[o + 2 * ((it + 1) & 7)]; bn = t4[o + 8 + 2 * ((it + 1) & 7)];
                __builtin_amdgcn_sched_barrier(0);
                MM4(a4, b4)
                __builtin_amdgcn_sched_barrier(0);
                a4 = t4[o + 2 * ((it + 2) & 7)]; b4 = t4[o + 8 + 2 * ((it + 2) & 7)];
                __builtin_amdgcn_sched_barrier(0);
                MM4(an, bn)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                const float4 a4 = t4[o + 2 * (it & 7)], b4 = t4[o + 8 + 2 * (it & 7)];
                MM4(a4, b4)
            }
        }
    } else {                               // one accumulator: every MFMA waits for the one before it
        const float a0 = tile[lane], b0 = tile[128 + lane];
        for (int it = 0; it < iters; ++it) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, b0, acc[0], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int nwg = 256 * (argc > 2 ? atoi(argv[2]) : 1);        // workgroups of 4 waves: 1 or 2 per CU
    std::vector<float> h(1 << 20);
    Rng r(7);
    float *d_src, *d_sink; unsigned long long *d_st;
    CK(hipMalloc(&d_src, h.size() * 4)); CK(hipMalloc(&d_sink, (size_t)nwg * 256 * 4)); CK(hipMalloc(&d_st, (size_t)nwg * 16));
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    printf("mode %d (0 read-wait-multiply, 1 read-ahead, 2 register operands, 3 one dependent accumulator, 4 b128 reads x 16 MFMAs, 5 ... read ahead, 6 2 b128 x 4 MFMAs with moves, 7 ... ping-pong, 8 ... no read-ahead)\n", mode);
    for (int zero = 0; zero < 1; ++zero) {
        for (auto &v : h) v = zero ? 0.f : r.normal();
        CK(hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        // ~2 s of back-to-back launches first
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        float ms = 0.f; int launches = 0;
        CK(hipEventRecord(a, 0));
        while (ms < 700.f) {
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(mfma_clock_kernel, dim3(nwg), dim3(256), 0, 0, d_src, iters, mode, d_sink, d_st);
            launches += 20;
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        std::vector<unsigned long long> st((size_t)nwg * 2);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> clk, nsm;
        for (int w = 0; w < nwg; ++w) {
            const double ticks = (double)st[w * 2], real = (double)st[w * 2 + 1];       // real: 10 ns units
            clk.push_back(ticks / real * 100.0);                                           // MHz
            nsm.push_back(real * 10.0 / (4.0 * iters));
        }
        std::sort(clk.begin(), clk.end()); std::sort(nsm.begin(), nsm.end());
        const double ns = nsm[nsm.size() / 2];
        const double per_cu_waves = 4.0 * nwg / 256.0;
        printf("%s operands, %d workgroups x 4 waves (%g waves per SIMD), %d x 4 MFMAs per wave, %d launches in %.1f s:\n"
               "   in-kernel clock: median %.0f MHz (min %.0f, max %.0f); %.1f ns per MFMA per wave = %.1f clocks; "
               "chip: %.1f TFLOP/s of fp32 MFMA\n",
               zero ? "zero  " : "random", nwg, per_cu_waves / 4.0, iters, launches, ms / 1e3, clk[clk.size() / 2], clk.front(), clk.back(), ns,
               ns * clk[clk.size() / 2] * 1e-3, 4096.0 / ns * 1e-3 * (4.0 * nwg) );
    }
    return 0;
}
