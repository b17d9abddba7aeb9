// Lab 3: what does one vector-memory load instruction cost a CU when the data is L2-resident?
// 16 waves per CU, every wave issues LOADS loads (batches of 8) from a 64 KB table; variants: all lanes valid,
// 3/4 of the lanes outside the descriptor's range, 3/4 of the lanes switched off by EXEC; widths 16/12/4/2 bytes.
#include "lab_common.h"
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v3i __attribute__((ext_vector_type(3)));

template <int WIDTH, int MODE>   // MODE 0 all lanes, 1 lanes >= 16 out of range, 2 lanes >= 16 EXEC-off, 3: 16 lanes contiguous 256 B (like one centroid group), others oob
__global__ __launch_bounds__(256) void load_kernel(const char *tab, int bytes, int loads, int *out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(tab), 0, bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    int acc = 0;
    if (MODE == 2 && lane >= 16) { out[blockIdx.x * 256 + threadIdx.x] = 0; return; }
    for (int i = 0; i < loads; i += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            unsigned o = (h >> 8) % (unsigned)(bytes - 16) & ~15u;
            if (MODE == 4) o = (((h >> 8) % (unsigned)(bytes - 1024)) & ~1023u) + lane * 16;      // a wave reads 1 KB contiguous
            if (MODE == 1 && lane >= 16) o = 0xfffffff0u;
            if (WIDTH == 16) { v4i t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, 0); v[u] = t.x ^ t.w; }
            else if (WIDTH == 12) { v3i t = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)o, 0, 0); v[u] = t.x ^ t.z; }
            else if (WIDTH == 4) { v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)o, 0, 0); }
            else { v[u] = __builtin_amdgcn_raw_buffer_load_b16(rs, (int)o, 0, 0); }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WIDTH>
__global__ __launch_bounds__(256) void lds_kernel(int loads, int *out)
{
    __shared__ __attribute__((aligned(16))) int sm[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = i;
    __syncthreads();
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    int acc = 0;
    for (int i = 0; i < loads; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            const unsigned o = ((h >> 8) % 16380u) & ~3u;
            if (WIDTH == 16) { const int4 t = *reinterpret_cast<const int4 *>(&sm[o]); acc ^= t.x ^ t.w; }
            else acc ^= sm[o];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const int bytes = 80 * 1024, loads = 256;
    char *tab; int *out;
    CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 1, bytes)); CK(hipMalloc(&out, 1024 * 256 * 4));
    const double base = time_us([&] { hipLaunchKernelGGL((load_kernel<16, 0>), dim3(1024), dim3(256), 0, 0, tab, bytes, 0, out); });
    printf("launch + prologue only: %.2f us\n", base);
#define RUN(W, M, name) { const double t = time_us([&] { hipLaunchKernelGGL((load_kernel<W, M>), dim3(1024), dim3(256), 0, 0, tab, bytes, loads, out); }); \
    printf("%-44s %7.2f us  -> %6.1f clk per load instruction per CU (16 waves x %d loads, 2.4 GHz)\n", name, t, (t - base) * 2400.0 / (16.0 * loads), loads); }
    RUN(16, 0, "16 B, 64 random lanes");
    RUN(16, 4, "16 B, 64 lanes = 1 KB contiguous");
    RUN(16, 1, "16 B, 16 lanes valid, 48 out of range");
    RUN(16, 2, "16 B, 16 lanes, 48 EXEC-off");
    RUN(12, 0, "12 B, 64 random lanes");
    RUN(12, 4, "12 B, 64 lanes contiguous stride 16");
    RUN(4, 0, "4 B, 64 random lanes");
    RUN(4, 4, "4 B, 64 lanes stride 16");
    RUN(2, 0, "2 B, 64 random lanes");
    const double lb = time_us([&] { hipLaunchKernelGGL((lds_kernel<16>), dim3(1024), dim3(256), 0, 0, 0, out); });
    { const double t = time_us([&] { hipLaunchKernelGGL((lds_kernel<16>), dim3(1024), dim3(256), 0, 0, loads, out); });
      printf("%-44s %7.2f us  -> %6.1f clk per instruction per CU\n", "LDS 16 B random", t, (t - lb) * 2400.0 / (16.0 * loads)); }
    { const double t = time_us([&] { hipLaunchKernelGGL((lds_kernel<4>), dim3(1024), dim3(256), 0, 0, loads, out); });
      printf("%-44s %7.2f us  -> %6.1f clk per instruction per CU\n", "LDS 4 B random", t, (t - lb) * 2400.0 / (16.0 * loads)); }
    return 0;
}
