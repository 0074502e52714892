// Does gfx950 execute scalar atomics (s_atomic_add, returning through lgkmcnt), and are they coherent with vector
// atomics on the same address?   hipcc --offload-arch=gfx950 -O3 -o /tmp/satomic_test satomic_test.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void k(unsigned* ctr, unsigned* out, unsigned add) {
    unsigned ret = add;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(ret) : "s"(ctr) : "memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = ret;
    if (threadIdx.x == 0) atomicAdd(ctr + 1, 1u);  // a vector atomic next to it
}

int main() {
    unsigned *ctr, *out;
    const int blocks = 1024, waves = blocks * 4;
    hipMalloc(&ctr, 64);
    hipMalloc(&out, waves * 4);
    hipMemset(ctr, 0, 64);
    k<<<blocks, 256>>>(ctr, out, 3u);
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    unsigned h[2];
    hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost);
    std::vector<unsigned> o(waves);
    hipMemcpy(o.data(), out, waves * 4, hipMemcpyDeviceToHost);
    std::vector<char> seen(waves, 0);
    int bad = 0;
    for (unsigned v : o) {
        if (v % 3 || v / 3 >= (unsigned)waves || seen[v / 3]) bad++;
        else seen[v / 3] = 1;
    }
    printf("counter %u (expect %u), vector counter %u (expect %d), bad/duplicate returns %d\n", h[0], 3u * waves, h[1], blocks, bad);
    return bad || h[0] != 3u * waves;
}
