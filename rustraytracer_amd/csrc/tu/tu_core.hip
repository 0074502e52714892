// tu_core.hip -- the kernels that do not depend on the shading-feature variant, one precision per compile
// (-DRT_TU_F32=0: binary64 + the precision-independent ones; =1: the generated binary32 copies).
#define RT_KERNELS_CORE 1
#if RT_TU_F32
#include "../build/f32/kernels.hip"
namespace K = rtd32;
#else
#include "../device/kernels.hip"
namespace K = rtd;
#endif
#include "../kernels_api.h"

namespace {
struct Reg {
    Reg() {
        rtk::KernelTable& t = rtk::kernel_table();
        constexpr int P = RT_TU_F32 ? 1 : 0;
        t.trace[P][0] = K::k_trace<false, true>;
        t.trace[P][1] = K::k_trace<false, false>;
        t.trace[P][2] = K::k_trace<true, false>;
        t.generate[P] = K::k_generate;
        t.intersect[P] = K::k_intersect_batch;
        t.classify_count = K::k_classify_count<0>;
        t.classify_scan = K::k_classify_scan<0>;
        t.classify_scatter = K::k_classify_scatter<0>;
        t.shade_light[P][0] = K::k_shade_light<0>;
        t.shade_light[P][1] = K::k_shade_light<K::kFeatEnv>;
#if !RT_TU_F32
        t.plan = K::k_plan;
        t.resolve = K::k_resolve;
        t.tonemap = K::k_tonemap;
#endif
    }
} reg;
}  // namespace
