// Diagnostic (not part of the library): what does a c3-shaped launch cost before any env logic?
//   empty      1024 workgroups x 64 lanes, 13 KiB dynamic LDS, no work: the per-launch floor inside a hipGraph
//   traffic    the same grid; every wave reads 6 KiB and writes 13 KiB in 16-byte coalesced accesses (the c3 step's
//              HBM volume: 6.2 MB in, 13.3 MB out per launch), write-through stores like the observation stream
//   traffic-p  the same with plain stores
// Build: hipcc -O3 --offload-arch=gfx950 -o launch_floor tools/launch_floor.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_empty(int *sink) {
    extern __shared__ unsigned char lds[];
    if (sink == (int *)1) lds[threadIdx.x] = 1;  // never true; keeps the LDS allocation
}

template <bool WT>
__global__ __launch_bounds__(64) void k_traffic(const uint4 *__restrict__ in, float *out, unsigned out_bytes, int rd16, int wr16) {
    // rd16 / wr16: 16-byte accesses per lane
    const int lane = threadIdx.x;
    const uint4 *src = in + (size_t)blockIdx.x * rd16 * 64 + lane;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int k = 0; k < rd16; k++) {
        const uint4 v = src[k * 64];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, out_bytes, 0x00020000);
    const unsigned off0 = ((unsigned)blockIdx.x * wr16 * 64 + lane) * 16u;
    for (int k = 0; k < wr16; k++) {
        const v4u w = {acc.x + k, acc.y, acc.z, acc.w};
        if (WT) __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, (int)(off0 + k * 1024u), 0, 16);
        else __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, (int)(off0 + k * 1024u), 0, 0);
    }
}

template <class F>
static int time_graph(const char *name, hipStream_t s, F launch, int per_graph, int replays, double bytes) {
    hipGraph_t g; hipGraphExec_t ge;
    for (int i = 0; i < 3; i++) launch();
    CK(hipStreamSynchronize(s));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < per_graph; i++) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < replays; i++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / (per_graph * replays);
    printf("{\"kernel\": \"%s\", \"us_per_launch\": %.3f, \"GBps\": %.1f}\n", name, us, bytes > 0 ? bytes / (us * 1e-6) / 1e9 : 0.0);
    return 0;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int WG = 1024, LDS = 13 * 1024;
    const int rd16 = 6, wr16 = 13;  // 6 KiB in, 13 KiB out per wave
    uint4 *in; float *out;
    const size_t in_bytes = (size_t)WG * 2 * rd16 * 1024, out_bytes = (size_t)WG * 2 * wr16 * 1024;
    CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, out_bytes));
    CK(hipMemset(in, 1, in_bytes)); CK(hipMemset(out, 0, out_bytes));
    const double bytes = (double)WG * (rd16 + wr16) * 1024;
    time_graph("empty_1024wg_13KiB_lds", s, [&] { hipLaunchKernelGGL(k_empty, dim3(WG), dim3(64), LDS, s, (int *)nullptr); }, 100, 20, 0);
    time_graph("empty_1024wg_no_lds", s, [&] { hipLaunchKernelGGL(k_empty, dim3(WG), dim3(64), 0, s, (int *)nullptr); }, 100, 20, 0);
    time_graph("empty_64wg", s, [&] { hipLaunchKernelGGL(k_empty, dim3(64), dim3(64), LDS, s, (int *)nullptr); }, 100, 20, 0);
    time_graph("traffic_wt_1024wg", s, [&] { hipLaunchKernelGGL(k_traffic<true>, dim3(WG), dim3(64), LDS, s, in, out, (unsigned)out_bytes, rd16, wr16); }, 100, 20, bytes);
    time_graph("traffic_plain_1024wg", s, [&] { hipLaunchKernelGGL(k_traffic<false>, dim3(WG), dim3(64), LDS, s, in, out, (unsigned)out_bytes, rd16, wr16); }, 100, 20, bytes);
    // the same bytes from twice as many, half-sized waves (two per SIMD)
    time_graph("traffic_wt_2048wg_half", s, [&] { hipLaunchKernelGGL(k_traffic<true>, dim3(2 * WG), dim3(64), LDS / 2, s, in, out, (unsigned)out_bytes, rd16 / 2, (wr16 + 1) / 2); }, 100, 20, (double)2 * WG * (rd16 / 2 + (wr16 + 1) / 2) * 1024);
    // pure write stream over a buffer far larger than the 256 MB Infinity Cache: what HBM takes from the fused
    // kernel's observation stream (every step writes a fresh 8.65 MB slice of a [T][B][N][L] tensor)
    {
        float *big; const size_t big_bytes = (size_t)2 << 30;
        CK(hipMalloc(&big, big_bytes));
        CK(hipMemset(big, 0, big_bytes));
        const int wr = 64;  // 64 KiB per wave
        const int wgs = (int)(big_bytes / ((size_t)wr * 1024));
        for (int wt = 0; wt < 2; wt++) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 4; i++) {
                    if (wt) hipLaunchKernelGGL(k_traffic<true>, dim3(wgs), dim3(64), 0, s, in, big, 0xFFFFFFFFu, 0, wr);
                    else hipLaunchKernelGGL(k_traffic<false>, dim3(wgs), dim3(64), 0, s, in, big, 0xFFFFFFFFu, 0, wr);
                }
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"kernel\": \"fill_2GiB_%s\", \"GBps\": %.1f}\n", wt ? "wt" : "plain", 4.0 * big_bytes / (ms * 1e-3) / 1e9);
        }
        // copy 1 GiB -> 1 GiB (read + write)
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(big, (char *)big + (big_bytes >> 1), big_bytes >> 1, hipMemcpyDeviceToDevice, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
        }
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"memcpy_d2d_1GiB\", \"GBps_read_plus_write\": %.1f}\n", 4.0 * big_bytes / (ms * 1e-3) / 1e9);
    }
    return 0;
}
