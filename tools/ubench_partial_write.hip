// ubench_partial_write.hip -- what does HBM charge for partly written atoms / lines?  (round 4: layout of the path records)
// A 4 GiB table of 128-B lines, far beyond L2 + Infinity Cache.  Every lane owns one line per iteration (random-ish, all
// distinct: a multiplicative permutation) and writes / reads part of it.  Prints time and lines per second per variant.
//   build: hipcc -O3 --offload-arch=gfx950 tools/ubench_partial_write.hip -o /tmp/ubench_partial_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// mode: 0 write 8 B   1 write 24 B (one 16 + one 8)   2 write 64 B (one atom, 4 x 16)   3 write 128 B (8 x 16, own line)
//       4 write 72 B spread over both atoms (48 + 24)   5 read 64 B   6 read 128 B   7 read 16 B
//       8 write 128 B, eight lanes per line (whole-line stores)   9 read 128 + write 128 same line (own line)
__global__ __launch_bounds__(256) void k(u64x2* tab, u64 n_lines, u64 mul, int mode, u64* sink) {
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 n_thr = (u64)gridDim.x * blockDim.x;
    u64 acc = 0;
    for (u64 i = tid; i < n_lines; i += n_thr) {
        u64 line = (i * mul) % n_lines;  // a permutation (mul odd, n_lines a power of two)
        if (mode == 8) {  // eight consecutive lanes share a line (i counts 16-B parts here)
            const u64 part = i & 7;
            line = ((i >> 3) * mul) % (n_lines >> 3);
            u64x2 v; v.x = i; v.y = mul;
            tab[line * 8 + part] = v;
            continue;
        }
        u64x2* p = tab + line * 8;
        u64x2 v; v.x = i; v.y = mul;
        switch (mode) {
            case 0: reinterpret_cast<u64*>(p)[1] = i; break;
            case 1: p[1] = v; reinterpret_cast<u64*>(p)[4] = i; break;
            case 2: p[0] = v; p[1] = v; p[2] = v; p[3] = v; break;
            case 3: for (int q = 0; q < 8; q++) p[q] = v; break;
            case 4: p[0] = v; p[1] = v; p[2] = v; p[5] = v; reinterpret_cast<u64*>(p)[12] = i; break;
            case 5: { u64x2 a = p[0], b = p[1], c = p[2], d = p[3]; acc += a.x + b.y + c.x + d.y; } break;
            case 6: for (int q = 0; q < 8; q++) { u64x2 a = p[q]; acc += a.x ^ a.y; } break;
            case 7: { u64x2 a = p[3]; acc += a.x; } break;
            case 9: { u64x2 a[8]; for (int q = 0; q < 8; q++) a[q] = p[q]; for (int q = 0; q < 8; q++) { a[q].x += i; p[q] = a[q]; } } break;
        }
    }
    if (acc == 0x1234567) sink[0] = acc;
}

int main() {
    const u64 n_lines = 1ull << 25;  // 4 GiB
    u64x2* tab; u64* sink;
    hipMalloc(&tab, n_lines * 128); hipMalloc(&sink, 8);
    hipMemset(tab, 0, n_lines * 128);
    const char* names[] = {"write 8 B of a line", "write 24 B of one atom", "write one whole 64-B atom", "write the whole 128-B line (8 x 16 B, own line)",
                           "write 72 B over both atoms", "read one 64-B atom", "read the whole line", "read 16 B", "write whole lines, 8 lanes per line",
                           "read + rewrite the whole line"};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 10; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 0, 0, tab, mode == 8 ? n_lines * 8 : n_lines, 0x9E3779B97F4A7C15ull | 1ull, mode, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-52s %8.2f ms  %7.2f G lines/s  (%6.2f TB/s if every touched line moved whole)\n", names[mode], best, n_lines / best / 1e6,
               n_lines * 128.0 / best / 1e9);
    }
    return 0;
}
