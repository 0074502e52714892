// ubench_fetchsize.hip -- what does rocprofv3's FETCH_SIZE report for the access patterns of k_trace / k_shade?
// (measurement tool, not product)
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads (it reports exactly half of the
// bytes there) and says "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern".  This program does that.  Every kernel below touches each record of a 2 GiB table EXACTLY ONCE (a
// bijective hash of the global thread index picks the record), so nothing can be served twice from L2 or the
// Infinity Cache and the bytes that have to leave HBM are known in closed form:
//   stream16     coalesced 16 B per lane (the guide's calibration case)
//   stream8      coalesced 8 B per lane  (the SoA path-state streams of k_shade / k_generate / k_trace's ray fetch)
//   node7        lane-divergent: 7 x 16 B of "its own" 128-B record (k_trace's node step)
//   node1        lane-divergent: 16 B of a 128-B record            (a single dwordx4 of a node)
//   node2halves  lane-divergent: 16 B at offset 0 and at offset 64 (both 64-B halves of a line)
//   tri72        lane-divergent: 9 x 8 B of a 72-B record          (k_trace's leaf step; records straddle lines)
// Run under   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- ./ubench_fetchsize
// (a second pass with TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum if the counters exist) and feed the CSV plus this
// program's stdout to tools/fetchsize_factors.py.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));         \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

// odd multiplier -> bijection on [0, 2^k)
__device__ inline uint64_t pick(uint64_t i, uint64_t mask) { return (i * 0x9E3779B97F4A7C15ull + 0x1234567ull) & mask; }

__global__ __launch_bounds__(256) void k_stream16(const float4* __restrict__ t, uint64_t n16, float* out) {
    float acc = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
        const float4 v = t[i];
        acc += v.x + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream8(const double* __restrict__ t, uint64_t n8, float* out) {
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * 256) acc += t[i];
    if (acc == 123.456) out[0] = (float)acc;
}
template <int K, int STRIDE16>
__global__ __launch_bounds__(256) void k_node(const float4* __restrict__ t, uint64_t n_rec, float* out) {
    float acc = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_rec; i += (uint64_t)gridDim.x * 256) {
        const float4* p = t + pick(i, n_rec - 1) * 8u;
        float4 v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = p[k * STRIDE16];
#pragma unroll
        for (int k = 0; k < K; k++) acc += v[k].x + v[k].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_tri72(const double* __restrict__ t, uint64_t n_rec, uint64_t n_pow2, float* out) {
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_pow2; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = pick(i, n_pow2 - 1);
        if (r >= n_rec) continue;
        const double* p = t + r * 9u;
#pragma unroll
        for (int k = 0; k < 9; k++) acc += p[k];
    }
    if (acc == 123.456) out[0] = (float)acc;
}

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const uint64_t bytes = 1ull << 31;  // 2 GiB: 8x the Infinity Cache
    const uint64_t n_rec = bytes / 128;  // power of two
    void* tab;
    float* out;
    CK(hipMalloc(&tab, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(tab, 0, bytes));
    CK(hipDeviceSynchronize());
    const int blocks = pr.multiProcessorCount * 8;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float ms;
#define RUN(name, launch, req_bytes, lines128, sectors64, sectors32)                                                     \
    CK(hipEventRecord(a));                                                                                               \
    launch;                                                                                                              \
    CK(hipEventRecord(b));                                                                                               \
    CK(hipEventSynchronize(b));                                                                                          \
    CK(hipEventElapsedTime(&ms, a, b));                                                                                  \
    printf("UBENCH %s requested_bytes %llu bytes_as_128B_lines %llu bytes_as_64B_sectors %llu bytes_as_32B_sectors %llu ms %.3f\n", \
           name, (unsigned long long)(req_bytes), (unsigned long long)(lines128), (unsigned long long)(sectors64),      \
           (unsigned long long)(sectors32), ms);
    RUN("k_stream16", (k_stream16<<<blocks, 256>>>((const float4*)tab, bytes / 16, out)), bytes, bytes, bytes, bytes)
    RUN("k_stream8", (k_stream8<<<blocks, 256>>>((const double*)tab, bytes / 8, out)), bytes, bytes, bytes, bytes)
    RUN("k_node<7, 1>", (k_node<7, 1><<<blocks, 256>>>((const float4*)tab, n_rec, out)), n_rec * 112, n_rec * 128, n_rec * 128,
        n_rec * 128)
    RUN("k_node<1, 1>", (k_node<1, 1><<<blocks, 256>>>((const float4*)tab, n_rec, out)), n_rec * 16, n_rec * 128, n_rec * 64,
        n_rec * 32)
    RUN("k_node<2, 4>", (k_node<2, 4><<<blocks, 256>>>((const float4*)tab, n_rec, out)), n_rec * 32, n_rec * 128, n_rec * 128,
        n_rec * 64)
    {
        const uint64_t n72 = bytes / 72, n_pow2 = 1ull << 25;  // 2^25 > n72 = 29.8 M: every record once
        // 72-B records: 7/16 of them straddle a 128-B line boundary (72 mod 128 pattern over 16 records = 9 lines)
        RUN("k_tri72", (k_tri72<<<blocks, 256>>>((const double*)tab, n72, n_pow2, out)), n72 * 72, n72 * 72, n72 * 72, n72 * 72)
    }
    printf("UBENCH_DONE table_bytes %llu\n", (unsigned long long)bytes);
    return 0;
}
