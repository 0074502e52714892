// ubench_gather.hip -- how fast can a CU serve lane-divergent 16-B loads?  (measurement tool, not product)
//
// k_trace's node step makes every lane read 7 x 16 B of "its" 128-B node.  This microbenchmark reproduces that
// access pattern in isolation: every lane picks a pseudo-random 128-B record per iteration and reads K x 16 B
// of it (K = 1, 2, 4, 7, 8), over tables of different sizes (L1-, L2-, MALL-, HBM-resident), at the same
// occupancy as k_trace (4 blocks x 256 threads per CU).  Output: CU cycles per wave-level load instruction
// and per record.  If the cost follows K (instructions) rather than the number of distinct lines (1 per lane
// whatever K), the L1's per-lane address/tag rate is the bound and smaller nodes pay directly.
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_gather tools/ubench_gather.hip && gpurun_out/ubench_gather
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

__device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// ACTIVE: lanes [0, ACTIVE) of every wave take part (the others skip the loads: k_trace runs at ~45 % lanes)
template <int K, int STRIDE16>
__global__ __launch_bounds__(256, 4) void k_gather(const float4* __restrict__ tab, uint32_t mask, int iters,
                                                   int active, float* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    uint32_t h = mix(tid * 2654435761u + 12345u);
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            // dependent chain like a traversal: the next record follows from the loaded data
            const float4* p = tab + (size_t)(h & mask) * 8u;
            float4 v[K];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = p[k * STRIDE16];
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) s += v[k].x + v[k].w;
            acc += s;
            h = mix(h + __float_as_uint(s));
        }
    }
    if (acc == 123.456f) out[tid] = acc;
}

template <int K, int STRIDE16>
static int run(const float4* tab, uint32_t n_rec, int active, float* out, int cus, double mhz, const char* what) {
    const int blocks = cus * 4, iters = 2000;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    k_gather<K, STRIDE16><<<blocks, 256>>>(tab, n_rec - 1, 200, active, out);
    CK(hipEventRecord(a));
    k_gather<K, STRIDE16><<<blocks, 256>>>(tab, n_rec - 1, iters, active, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double cyc = ms * 1e-3 * mhz * 1e6;                 // CU cycles of the launch
    const double wave_steps_per_cu = 16.0 * iters;            // 16 waves per CU, one record per lane per iter
    const double recs = (double)blocks * 256.0 * active / 64.0 * iters;
    printf("%-6s K=%d stride=%3dB lanes=%2d  %8.3f ms  %7.1f cyc/wave-step/CU  %6.1f cyc/load-instr/CU  %7.1f Grec/s  %7.1f GB/s(used)\n",
           what, K, STRIDE16 * 16, active, ms, cyc / wave_steps_per_cu, cyc / wave_steps_per_cu / K,
           recs / (ms * 1e-3) / 1e9, recs * K * 16.0 / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const double mhz = pr.clockRate / 1000.0;
    printf("%s: %d CUs, %.0f MHz\n", pr.name, cus, mhz);
    const size_t max_rec = (size_t)1 << 22;  // 4 M records x 128 B = 512 MB
    float4* tab;
    float* out;
    CK(hipMalloc(&tab, max_rec * 128));
    CK(hipMalloc(&out, (size_t)cus * 4 * 256 * 4));
    {
        std::vector<float> h(max_rec * 32);
        uint32_t s = 1u;
        for (auto& x : h) {
            s = s * 1664525u + 1013904223u;
            x = (float)(s >> 8) * (1.0f / 16777216.0f);
        }
        CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    struct { uint32_t n; const char* what; } sizes[] = {
        {128, "16KB"}, {8192, "1MB"}, {1u << 18, "32MB"}, {1u << 21, "256MB"}};
    for (auto& sz : sizes) {
        for (int active : {64, 29}) {
            if (run<1, 1>(tab, sz.n, active, out, cus, mhz, sz.what)) return 1;
            if (run<2, 1>(tab, sz.n, active, out, cus, mhz, sz.what)) return 1;
            if (run<4, 1>(tab, sz.n, active, out, cus, mhz, sz.what)) return 1;
            if (run<7, 1>(tab, sz.n, active, out, cus, mhz, sz.what)) return 1;
            if (run<8, 1>(tab, sz.n, active, out, cus, mhz, sz.what)) return 1;
        }
        // 4 loads spread over the whole 128-B line vs packed in its first half
        if (run<4, 2>(tab, sz.n, 64, out, cus, mhz, sz.what)) return 1;
    }
    return 0;
}
