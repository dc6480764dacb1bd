// rt_variants.h — the feature variants of the big kernels, one translation unit each.
//
// The trace kernel (rt_kernels.h) and the queue pipeline's analytic / settle / shade kernels
// (rt_wavefront.h) are compiled for a handful of feature sets; a scene runs on the smallest one that
// covers what it contains (fewer registers and less scratch than the all-features build).  Every
// variant lives in a source file of its own under variants/ (they compile side by side); the frame
// driver in rt_api.hip only sees the launch functions declared here.
#pragma once
#include <map>
#include <mutex>
#include <utility>
// (the translation units of the persistent trace kernel define RT_VARIANT_TRACE_ONLY: they do not depend on
// the queue pipeline's header, and are not rebuilt when it changes)
#ifdef RT_VARIANT_TRACE_ONLY
#include "rt_kernels.h"
struct WfArgs;
#else
#include "rt_wavefront.h"
#endif

constexpr uint32_t V_MESH = RTF_MESH;                                         // rooms + meshes, point lights (glass_monkey)
constexpr uint32_t V_AREA = RTF_FACED | RTF_INLINE_LIGHTS | RTF_IMAGES;       // square light, cube, textures (light_room)
constexpr uint32_t V_BALLS = RTF_WRAP | RTF_IMAGES | RTF_TLAS;                // one-way spheres, skybox, maps, many instances (many_balls)
constexpr uint32_t V_CSG = RTF_CSG | RTF_FACED | RTF_IMAGES;                  // composite shapes under a skybox
constexpr uint32_t V_AREA_MESH = V_AREA | RTF_MESH;                           // square light + mesh + maps (monkey_1..3)
constexpr uint32_t V_EVERYTHING = RTF_ALL | RTF_TLAS;
constexpr uint32_t V_NO_INLINE = (RTF_ALL | RTF_TLAS) & ~RTF_INLINE_LIGHTS;

// persistent trace kernel
#define RT_TRACE_VARIANTS(X) \
    X(plain, 0u) X(mesh_lds, V_MESH | RTF_LDS_BVH) X(mesh, V_MESH) X(balls, V_BALLS) X(area, V_AREA) \
    X(area_mesh_lds, V_AREA_MESH | RTF_LDS_BVH) X(area_mesh, V_AREA_MESH) X(csg, V_CSG) X(everything, V_EVERYTHING)
// queue pipeline (no inline lights there)
#define RT_WF_VARIANTS(X) X(plain, 0u) X(mesh, V_MESH) X(balls, V_BALLS) X(csg, V_CSG) X(everything, V_NO_INLINE)

struct LaunchStatus {             // hipSuccess, or the failing call
    hipError_t code;
    const char* what;
};

// Raises a kernel's dynamic-LDS limit on the current device when a launch needs more than it has been granted so
// far.  The attribute sticks, so remembering the grant keeps the call out of every later launch: the frame's
// launch sequence is then made of stream operations only (it can be captured into a HIP graph once it has run once).
inline hipError_t rt_grant_lds(const void* fn, uint32_t lds) {
    if (lds <= 48u * 1024u) return hipSuccess;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, uint32_t> granted;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    uint32_t& g = granted[{dev, fn}];
    if (lds <= g) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) g = lds;
    return e;
}

#define RT_DECLARE_TRACE(name, F) \
    LaunchStatus rt_launch_trace_##name(bool count, uint32_t grid, hipStream_t stream, const DScene& ds, const DRenderParams& rp, const TraceArgs& ta);
RT_TRACE_VARIANTS(RT_DECLARE_TRACE)
#undef RT_DECLARE_TRACE
// which: 0 analytic, 1 shade, 2 settle, 3 analytic redo (scenes with composite shapes), 4 / 5 shade on the kernel for
// material class 1 / 2 (k_wf_shade; only built for scenes whose tables fit LDS, else the general kernel runs), 6 the tie
// decisions noted by the analytic kernel (k_wf_ties; `grid` is ignored: one workgroup per sub-queue of the launch)
#define RT_DECLARE_WF(name, F) \
    LaunchStatus rt_launch_wf_##name(int which, bool count, uint32_t grid, hipStream_t stream, const DScene& ds, const DRenderParams& rp, const WfArgs& wa);
RT_WF_VARIANTS(RT_DECLARE_WF)
#undef RT_DECLARE_WF
LaunchStatus rt_launch_wf_walk(bool count, uint32_t num_cus, hipStream_t stream, const DScene& ds, const WfArgs& wa);   // num_cus: the chain's share of the CUs

#define RT_TRACE_HOT_LIMIT (12u * 1024u)   // largest table copy per 64-thread workgroup of the persistent kernel (eight of them share a CU's LDS)
// ---- bodies, instantiated by the variant files only
template <uint32_t F>
LaunchStatus launch_trace_variant(bool count, uint32_t grid, hipStream_t stream, const DScene& ds, const DRenderParams& rp, const TraceArgs& ta) {
    uint32_t block = 64, lds = 0;
    if (F & RTF_LDS_BVH) { block = 512; grid = grid / 8 > 1u ? grid / 8 : 1u; }   // one 512-thread workgroup per CU, the BVH blob in its LDS
    if (F & RTF_MESH) lds = RT_LDS_FIXED_BYTES(block) + ((F & RTF_LDS_BVH) ? ds.lds_blob_bytes : 0u);   // walk stacks, job lists (+ blob)
    if (lds > 48u * 1024u) {
        auto fn = count ? (const void*)k_trace<F, true> : (const void*)k_trace<F, false>;
        hipError_t ea = rt_grant_lds(fn, lds);
        if (ea != hipSuccess) return {ea, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)"};
    }
    // Scenes whose lights are sampled inside the shader and that have no meshes: every shadow test reads the
    // instance and primitive tables, one dependent read after another; out of LDS (eight 64-thread
    // workgroups per CU, a copy each) the wait is a fraction of what it is out of L2.
    if constexpr ((F & RTF_INLINE_LIGHTS) && !(F & RTF_MESH)) {
        if (ds.hot_bytes > 0 && ds.hot_bytes <= RT_TRACE_HOT_LIMIT) {
            const uint32_t tables = (ds.hot_bytes + 15u) & ~15u;
            if (count) hipLaunchKernelGGL((k_trace<F, true, true>), dim3(grid), dim3(block), tables, stream, ds, rp, ta);
            else hipLaunchKernelGGL((k_trace<F, false, true>), dim3(grid), dim3(block), tables, stream, ds, rp, ta);
            return {hipGetLastError(), "k_trace launch"};
        }
    }
    if (count) hipLaunchKernelGGL((k_trace<F, true>), dim3(grid), dim3(block), lds, stream, ds, rp, ta);
    else hipLaunchKernelGGL((k_trace<F, false>), dim3(grid), dim3(block), lds, stream, ds, rp, ta);
    return {hipGetLastError(), "k_trace launch"};
}

#ifndef RT_VARIANT_TRACE_ONLY
template <uint32_t F>
LaunchStatus launch_wf_variant(int which, bool count, uint32_t grid, hipStream_t stream, const DScene& ds, const DRenderParams& rp, const WfArgs& wa) {
    const bool hot = ds.hot_bytes > 0;                                 // the scene's small tables fit into LDS
    const uint32_t tables = hot ? (ds.hot_bytes + 15u) & ~15u : 0u;
    uint32_t lds = tables;
    if (which == 1 || which == 4 || which == 5) lds = tables + 4u * 640u * 8u;   // + accumulation staging of the four waves
    if (which == 0 && ds.has_background) lds = tables + 4u * 640u * 8u + 4u * 128u * 4u;   // + staging and the waves' lists of rays that only see the sky box
    if (which == 2) lds = 4u * 640u * 8u + 4u * 128u * 4u;            // staging + the waves' lists of rays that left the scene
    const void* fn = nullptr;
    if (which == 0) fn = hot ? (count ? (const void*)k_wf_analytic<F, true, true, false> : (const void*)k_wf_analytic<F, false, true, false>)
                             : (count ? (const void*)k_wf_analytic<F, true, false, false> : (const void*)k_wf_analytic<F, false, false, false>);
    else if (which == 1) fn = hot ? (count ? (const void*)k_wf_shade<F, true, true> : (const void*)k_wf_shade<F, false, true>)
                                  : (count ? (const void*)k_wf_shade<F, true, false> : (const void*)k_wf_shade<F, false, false>);
    else if (which == 4 && hot) fn = count ? (const void*)k_wf_shade<F, true, true, 1> : (const void*)k_wf_shade<F, false, true, 1>;
    else if (which == 5 && hot) fn = count ? (const void*)k_wf_shade<F, true, true, 2> : (const void*)k_wf_shade<F, false, true, 2>;
    else if (which == 4 || which == 5) fn = count ? (const void*)k_wf_shade<F, true, false> : (const void*)k_wf_shade<F, false, false>;
    else if (which == 2) fn = count ? (const void*)k_wf_settle<F, true> : (const void*)k_wf_settle<F, false>;
    else if constexpr ((F & RTF_CSG) != 0) {                            // 3: the analytic kernel's second run (composite shapes only)
        if (which == 3)
            fn = hot ? (count ? (const void*)k_wf_analytic<F, true, true, true> : (const void*)k_wf_analytic<F, false, true, true>)
                     : (count ? (const void*)k_wf_analytic<F, true, false, true> : (const void*)k_wf_analytic<F, false, false, true>);
    }
    if (which == 6) {                                                    // the tie decisions the analytic kernel noted: one small workgroup per sub-queue
        void* tie_params[3] = {(void*)&ds, (void*)&rp, (void*)&wa};
        (void)hipLaunchKernel((const void*)k_wf_ties<F>, dim3(wa.sq_count), dim3(64), tie_params, 0, stream);
        return {hipGetLastError(), "k_wf_ties launch"};
    }
    if (!fn) return {hipErrorInvalidValue, "queue-pipeline kernel selection"};
    (void)rt_grant_lds(fn, lds);
    void* params[3] = {(void*)&ds, (void*)&rp, (void*)&wa};
    (void)hipLaunchKernel(fn, dim3(grid), dim3(256), params, lds, stream);
    return {hipGetLastError(), "queue-pipeline kernel launch"};
}
#endif  // RT_VARIANT_TRACE_ONLY
