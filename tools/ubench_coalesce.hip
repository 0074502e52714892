// ubench_coalesce.hip -- what does the vector L1 charge for 8-B loads whose 64 lane addresses are a PERMUTATION of a
// contiguous window?  (measurement tool; companion of ubench_gather.hip)
// Patterns, 64 lanes x 8 B per wave instruction, every wave on its own 512-B / 2-KB window of a large array:
//   ident   lane i -> element i                      (fully coalesced: 4 lines of 128 B)
//   rev     lane i -> element 63 - i
//   g8/g16  groups of 8 / 16 consecutive lanes keep consecutive elements, the groups are shuffled
//   perm64  a fixed pseudo-random permutation inside the wave's 512 B
//   perm256 a permutation inside the block's 2 KB (what k_shade's class dealing produces)
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_coalesce tools/ubench_coalesce.hip && gpurun_out/ubench_coalesce
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

__global__ __launch_bounds__(256, 4) void k(const double* __restrict__ tab, const uint16_t* __restrict__ perm,
                                            uint32_t n_win, int iters, int fields, double* out) {
    const uint32_t p = perm[threadIdx.x];  // element of the block's 256-element window this lane reads
    double acc = 0.0;
    uint32_t w = blockIdx.x;
    for (int i = 0; i < iters; ++i) {
        w = (w * 1664525u + 1013904223u) % n_win;  // next window (blocks hop around a large table)
        const double* base = tab + (size_t)w * 256u * (size_t)fields;
        for (int f = 0; f < fields; ++f) acc += base[(size_t)f * 256u + p];  // SoA fields, like the path state
    }
    if (acc == 123.456) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const double mhz = pr.clockRate / 1000.0;
    const int fields = 16, iters = 400;
    const uint32_t n_win_max = 1u << 15;  // 32768 windows x 16 fields x 2 KB = 1 GB
    double *tab, *out;
    uint16_t* d_perm;
    CK(hipMalloc(&tab, (size_t)n_win_max * 256 * fields * 8));
    CK(hipMemset(tab, 0, (size_t)n_win_max * 256 * fields * 8));
    CK(hipMalloc(&out, (size_t)cus * 4 * 256 * 8));
    CK(hipMalloc(&d_perm, 512));
    struct Pat { const char* name; std::vector<uint16_t> p; };
    std::vector<Pat> pats;
    auto ident = [] { std::vector<uint16_t> v(256); for (int i = 0; i < 256; i++) v[i] = (uint16_t)i; return v; };
    pats.push_back({"ident", ident()});
    { auto v = ident(); for (int w = 0; w < 4; w++) for (int i = 0; i < 64; i++) v[w * 64 + i] = (uint16_t)(w * 64 + 63 - i); pats.push_back({"rev", v}); }
    for (int g : {16, 8, 4}) {
        auto v = ident();
        for (int w = 0; w < 4; w++) {
            const int ng = 64 / g;
            for (int k = 0; k < ng; k++) {
                const int src = (k * 5 + 3) % ng;  // shuffled group order (5 is odd: a permutation for ng = 4, 8, 16)
                for (int i = 0; i < g; i++) v[w * 64 + k * g + i] = (uint16_t)(w * 64 + src * g + i);
            }
        }
        char nm[16];
        snprintf(nm, sizeof nm, "g%d", g);
        pats.push_back({g == 16 ? "g16" : (g == 8 ? "g8" : "g4"), v});
    }
    { auto v = ident(); for (int w = 0; w < 4; w++) for (int i = 0; i < 64; i++) v[w * 64 + i] = (uint16_t)(w * 64 + (i * 37 + 11) % 64); pats.push_back({"perm64", v}); }
    { auto v = ident(); for (int i = 0; i < 256; i++) v[i] = (uint16_t)((i * 149 + 57) % 256); pats.push_back({"perm256", v}); }
    const int blocks = cus * 4;
    printf("%s: %d CUs, %.0f MHz; %d SoA fields of 8 B per element, %d windows per block\n", pr.name, cus, mhz, fields, iters);
    for (uint32_t n_win : {8u, 512u, n_win_max}) {  // 256 KB (cache resident), 16 MB (L2 / Infinity Cache), 1 GB (HBM)
        printf("table of %u windows (%.1f MB)\n", n_win, (double)n_win * 256 * fields * 8 / 1e6);
        for (auto& pt : pats) {
            CK(hipMemcpy(d_perm, pt.p.data(), 512, hipMemcpyHostToDevice));
            hipEvent_t a, b;
            CK(hipEventCreate(&a));
            CK(hipEventCreate(&b));
            k<<<blocks, 256>>>(tab, d_perm, n_win, 20, fields, out);
            CK(hipEventRecord(a));
            k<<<blocks, 256>>>(tab, d_perm, n_win, iters, fields, out);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            const double bytes = (double)blocks * 256 * 8.0 * fields * iters;
            const double instr_per_cu = 16.0 * fields * iters;  // wave-level loads per CU (16 waves)
            printf("  %-8s %8.3f ms  %7.1f GB/s  %6.1f cycles per wave-load per CU\n", pt.name, ms,
                   bytes / (ms * 1e-3) / 1e9, ms * 1e-3 * mhz * 1e6 / instr_per_cu);
        }
    }
    return 0;
}
