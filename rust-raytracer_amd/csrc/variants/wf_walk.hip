// The queue pipeline's BVH walk kernel (k_wf_walk): LDS-resident and global-memory BVH.
#include <cstdlib>

#include "../rt_variants.h"
LaunchStatus rt_launch_wf_walk(bool count, uint32_t num_cus, hipStream_t stream, const DScene& ds, const WfArgs& wa) {
    const bool lds_bvh = ds.lds_blob_bytes != 0;
    // LDS walk: one workgroup per CU (the blob fills most of its LDS).  RT_WF_WALK_BLOCK: fewer waves per
    // workgroup leave registers for the other chains' kernels on the same CU (512, 768 or 1024 threads).
    static const uint32_t lds_block = [] { const char* e = std::getenv("RT_WF_WALK_BLOCK"); long v = e ? std::atol(e) : 0; return v >= 256 && v <= 1024 && v % 256 == 0 ? (uint32_t)v : 1024u; }();
    const uint32_t block = lds_bvh ? lds_block : 256u, grid = lds_bvh ? num_cus : num_cus * 8u;
    const uint32_t lds = RT_WF_WALK_FIXED_BYTES(block) + (lds_bvh ? ds.lds_blob_bytes : 0u);   // walk stacks, list starts (+ blob)
    const void* fn = lds_bvh ? (count ? (const void*)k_wf_walk<true, true> : (const void*)k_wf_walk<true, false>)
                             : (count ? (const void*)k_wf_walk<false, true> : (const void*)k_wf_walk<false, false>);
    {
        hipError_t ea = rt_grant_lds(fn, lds);
        if (ea != hipSuccess) return {ea, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)"};
    }
    if (lds_bvh) {
        if (count) hipLaunchKernelGGL((k_wf_walk<true, true>), dim3(grid), dim3(block), lds, stream, ds, wa);
        else hipLaunchKernelGGL((k_wf_walk<true, false>), dim3(grid), dim3(block), lds, stream, ds, wa);
    } else {
        if (count) hipLaunchKernelGGL((k_wf_walk<false, true>), dim3(grid), dim3(block), lds, stream, ds, wa);
        else hipLaunchKernelGGL((k_wf_walk<false, false>), dim3(grid), dim3(block), lds, stream, ds, wa);
    }
    return {hipGetLastError(), "k_wf_walk launch"};
}
