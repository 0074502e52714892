// Compile-only experiment (candidate b): would k_shade's two halves fit 4 waves per SIMD as separate
// kernels?  The A half (state fetch, fold of the pending light terms, hit record, emitted-light rule) and the B half
// (BSDF, one-light NEE + MIS, continuation) as kernels of their own, the ShadeA record going through memory between
// them.  Never launched: the register / spill figures come from the compiler's resource remarks.
//   cd rustraytracer_amd/csrc && hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 \
//       -Rpass-analysis=kernel-resource-usage -c ../../tools/experiments/shade_split_resources.hip -o /tmp/x.o
#include "../../rustraytracer_amd/csrc/device/kernels.hip"

namespace rtd {

template <int FEAT, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_split_a(DevScene sc, PathState in, ShadeA* as, uint32_t n, uint32_t max_depth) {
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    ShadeA a;
    shade_a<FEAT>(sc, in, slot, slot < n, max_depth, a);
    if (slot < n) as[slot] = a;
}

template <int FEAT, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_split_b(DevScene sc, PathState in, PathState out, const ShadeA* as, uint32_t n,
                                                         uint32_t* os_list) {
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= n) return;
    ShadeA a = as[slot];
    ShadeOut r{false, false, false, false};
    if (a.will_shade) r = shade_b<FEAT>(sc, in, out, slot, os_list[slot], a);
    os_list[slot] = (r.keep ? 1u : 0u) | (r.emit_ext ? 2u : 0u) | (r.emit_sh ? 4u : 0u) | (r.emit_pr ? 8u : 0u);
}

template __global__ void k_split_a<3, 4>(DevScene, PathState, ShadeA*, uint32_t, uint32_t);
template __global__ void k_split_a<3, 5>(DevScene, PathState, ShadeA*, uint32_t, uint32_t);
template __global__ void k_split_a<3, 8>(DevScene, PathState, ShadeA*, uint32_t, uint32_t);
template __global__ void k_split_b<3, 3>(DevScene, PathState, PathState, const ShadeA*, uint32_t, uint32_t*);
template __global__ void k_split_b<3, 4>(DevScene, PathState, PathState, const ShadeA*, uint32_t, uint32_t*);
template __global__ void k_split_b<3, 1>(DevScene, PathState, PathState, const ShadeA*, uint32_t, uint32_t*);
template __global__ void k_split_a<3, 1>(DevScene, PathState, ShadeA*, uint32_t, uint32_t);

__global__ void k_sizeof_shade_a(uint32_t* o) { *o = (uint32_t)sizeof(ShadeA); }

}  // namespace rtd
