// tu_shade.hip -- the class kernels and the fused tail of ONE shading-feature variant (-DRT_TU_V=0..8) in one precision
// (-DRT_TU_F32=0/1).  Eighteen small compiles in parallel instead of one two-minute translation unit.
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
        constexpr int P = RT_TU_F32 ? 1 : 0, V = RT_TU_V, F = K::kFeatVariants[RT_TU_V];
        t.shade_cls[P][V][rtc::kKindAny] = K::k_shade_cls<F, rtc::kKindAny>;
        t.shade_cls[P][V][rtc::kKindMesh] = K::k_shade_cls<F, rtc::kKindMesh>;
        t.shade_cls[P][V][rtc::kKindOther] = K::k_shade_cls<F, rtc::kKindOther>;
#if RT_TU_V == 0
        t.shade_cls_w2[P][rtc::kKindAny] = K::k_shade_cls<0, rtc::kKindAny, 2>;
        t.shade_cls_w2[P][rtc::kKindMesh] = K::k_shade_cls<0, rtc::kKindMesh, 2>;
        t.shade_cls_w2[P][rtc::kKindOther] = K::k_shade_cls<0, rtc::kKindOther, 2>;
#endif
        t.tail[P][V][0] = K::k_tail<F, false>;
        t.tail[P][V][1] = K::k_tail<F, true>;
    }
} reg;
}  // namespace
