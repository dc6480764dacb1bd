// The persistent trace kernel for one feature set (see rt_variants.h).
#define RT_VARIANT_TRACE_ONLY
#include "../rt_variants.h"
LaunchStatus rt_launch_trace_area(bool count, uint32_t grid, hipStream_t stream, const DScene& ds, const DRenderParams& rp, const TraceArgs& ta) {
    return launch_trace_variant<V_AREA>(count, grid, stream, ds, rp, ta);
}
