// kernels_api.h -- how abi.hip reaches the device kernels (round 4).
// The kernels are templates in device/kernels.hip (binary64, namespace rtd) and in the copy tools/make_f32_sources.py
// generates from it (binary32, namespace rtd32).  They are instantiated in small translation units of their own (tu/*.hip,
// one per shading-feature variant and precision, compiled in parallel) which register the kernels' host-side handles
// here; abi.hip only sees these function-pointer types.
#pragma once
#include <hip/hip_runtime.h>

#include "device/scene_dev.h"

namespace rtk {
using namespace rtc;

constexpr int kVariants = 9;  // shading.h: kNumFeatVariants

typedef void (*ShadeClsKernel)(DevScene, PathState, PathState, Ctl*, uint32_t, uint32_t, Lists, uint32_t, uint32_t*, uint32_t,
                               uint32_t, double*, DevStats*);
typedef void (*ShadeLightKernel)(DevScene, PathState, Ctl*, uint32_t, uint32_t, Lists, double*);
typedef void (*TailKernel)(DevScene, PathState, PathState, Ctl*, uint32_t, uint32_t, const uint32_t*, Lists, double*, DevStats*);
typedef void (*TraceKernel)(DevScene, PathState, const uint32_t*, Ctl*, uint32_t, DevStats*, TraceTune, MirrorEntry*, uint32_t,
                            const BatchCtl*, unsigned long long, uint32_t*);
typedef void (*ClassifyCountKernel)(const uint32_t*, const uint32_t*, const Ctl*, uint32_t, uint32_t*);
typedef void (*ClassifyScanKernel)(uint32_t*, uint32_t, Ctl*, uint32_t);
typedef void (*ClassifyScatterKernel)(const uint32_t*, const uint32_t*, const Ctl*, uint32_t, const uint32_t*, Lists);
typedef void (*GenKernel)(PathState, rt_camera, ChunkDesc, const uint32_t*, uint32_t*, const Ctl*);
typedef void (*PlanKernel)(Ctl*, BatchCtl*, uint32_t, uint32_t, unsigned long long, DevStats*);
typedef void (*IntersectKernel)(DevScene, const rt_ray*, uint64_t, rt_hit*);
typedef void (*ResolveKernel)(const double*, ChunkDesc, const uint32_t*, double*, uint32_t*);
typedef void (*TonemapKernel)(const double*, const uint32_t*, uint64_t, uint8_t*);

struct KernelTable {
    // [precision: 0 = binary64, 1 = binary32]
    ShadeClsKernel shade_cls[2][kVariants][3];  // [variant][kind: any / mesh / other]
    ShadeClsKernel shade_cls_w2[2][3];          // the Lambert-only instance compiled for 2 waves/SIMD (all-Lambert scenes)
    ShadeLightKernel shade_light[2][2];         // [environment light]
    TailKernel tail[2][kVariants][2];           // [variant][counting]
    TraceKernel trace[2][3];                    // 0: simple scenes, 1: general, 2: counting
    GenKernel generate[2];
    IntersectKernel intersect[2];
    PlanKernel plan;
    ClassifyCountKernel classify_count;  // (the same integer code in both precisions: registered twice, one wins)
    ClassifyScanKernel classify_scan;
    ClassifyScatterKernel classify_scatter;
    ResolveKernel resolve;
    TonemapKernel tonemap;
};
KernelTable& kernel_table();  // abi.hip (filled by the static initialisers of tu/*.hip)

}  // namespace rtk
