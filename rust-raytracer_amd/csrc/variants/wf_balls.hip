// The queue pipeline's analytic / settle / shade kernels for one feature set (see rt_variants.h).
#include "../rt_variants.h"
LaunchStatus rt_launch_wf_balls(int which, bool count, uint32_t grid, hipStream_t stream, const DScene& ds, const DRenderParams& rp, const WfArgs& wa) {
    return launch_wf_variant<V_BALLS>(which, count, grid, stream, ds, rp, wa);
}
