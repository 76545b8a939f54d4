// mapf_engine.h -- internal to libmapfstep.so: what the translation units of the library share.
//
// The library is built from one host translation unit (mapf_step.hip: the C ABI of include/mapf_step.h) and a set of
// LAUNCH units (mapf_launch.hip compiled once per -DMAPF_TU_* selection), each of which instantiates the kernels of one
// group -- one prebuilt specialisation, the runtime-config kernels of one group width and window-mask width, the
// single-agent kernels of one group width -- behind plain functions declared here.  The units compile in parallel
// (dl_reference_models_amd/build.py); a cold build of the whole library is the longest unit, not the sum.
//
// Device code lives in mapf_kernels.inl, compiled under the named namespace `mapfk` so that the types the units
// exchange (Io, Params, ManyPolicy, ...) are the same types in every unit.

#ifndef MAPF_ENGINE_H
#define MAPF_ENGINE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mapf_step.h"

#ifndef MAPF_NS
#define MAPF_NS mapfk
#endif
#include "mapf_kernels.inl"

namespace mapfk {

// Group widths and window-mask widths the library holds.  Reduced builds (development / checking: dl_reference_models_amd/
// build.py selects the launch units to match) cut them down; mapf_create refuses what a reduced build does not hold.
#if defined(MAPF_DEV_C3)  // the headline shape only
#define MAPF_FOR_LPE(X) X(8)
#define MAPF_FOR_MW(X, L) X(L, 32)
#elif defined(MAPF_DEV_CTE)  // the single-agent env at 8 and 64 lanes per env
#define MAPF_FOR_LPE(X) X(8) X(64)
#define MAPF_FOR_MW(X, L) X(L, 32)
#elif defined(MAPF_DEV_N16)  // groups of 16 lanes, 7 x 7 windows: the reference's training setup
#define MAPF_FOR_LPE(X) X(16)
#define MAPF_FOR_MW(X, L) X(L, 64)
#elif defined(MAPF_DEV_C5)  // the c5 shape only -- one wavefront per env, 5 x 5 windows
#define MAPF_FOR_LPE(X) X(64)
#define MAPF_FOR_MW(X, L) X(L, 32)
#elif defined(MAPF_SMALL_SHAPES)  // the checking build: groups of 4, 8 and 16 lanes, windows up to 7 x 7
#define MAPF_FOR_LPE(X) X(4) X(8) X(16)
#define MAPF_FOR_MW(X, L) X(L, 32) X(L, 64)
#else
#define MAPF_FOR_LPE(X) X(4) X(8) X(16) X(32) X(64)
#define MAPF_FOR_MW(X, L) X(L, 32) X(L, 64) X(L, 128)
#endif
#if defined(MAPF_DEV_C3) || defined(MAPF_DEV_N16) || defined(MAPF_DEV_C5) || defined(MAPF_SMALL_SHAPES)
#define MAPF_NO_CTE_KERNELS 1
#endif

// What a launch unit needs to know about a handle (the host unit fills it from mapf_engine).
struct LaunchPlan {
    const Params *d_params;
    int blocks;          // env workgroups
    int sampler_blocks;  // k_step only: workgroups of the grid that pre-draw next-episode placements
    int lds_bytes;
    int dense;           // k_step: the 128-register build (more than three waves per SIMD in one launch)
    int many_dense;      // k_step_many: idem (more than two)
    int three_wave;      // k_step3
    int rt_sliced;       // runtime-config kernels with the sliced background draw (KRuntimeSliced)
    int wide3;           // k_stepw: the three-wave kernel of 64-lane groups (one env per workgroup, bit rows in LDS)
    int wide_lds_bytes;  // its dynamic LDS
};

enum { KIND_RESET = 0, KIND_STEP = 1, KIND_OBSERVE = 2 };

// observation-window mask width of a sensor range
constexpr int mask_width_for(int sr) {
    return (2 * sr + 1) * (2 * sr + 1) <= 32 ? 32 : ((2 * sr + 1) * (2 * sr + 1) <= 64 ? 64 : 128);
}

// ---- the launch units' entry points (mapf_launch.hip) ----------------------------------------------------------------
// prebuilt specialisations: one unit per id of MAPF_SPECIALIZATIONS
#define MAPF_DECLARE_SPECIAL(ID, N_, SR_, FLAGS_, DW_, LW_, NEARBY_, MINN_, LPE_)                            \
    hipError_t launch_special_step_##ID(const LaunchPlan &lp, const Io &io, hipStream_t s);                   \
    hipError_t launch_special_many_##ID(const LaunchPlan &lp, const Io &io, int T, int obs_mode, const ManyPolicy &pol, hipStream_t s);
MAPF_SPECIALIZATIONS(MAPF_DECLARE_SPECIAL)
#undef MAPF_DECLARE_SPECIAL
// runtime-config kernels: one unit per (lanes per env, window-mask width)
#define MAPF_DECLARE_RUNTIME(L, MW)                                                                           \
    hipError_t launch_runtime_##L##_##MW(int kind, const LaunchPlan &lp, const Io &io, hipStream_t s);        \
    hipError_t launch_runtime_many_##L##_##MW(const LaunchPlan &lp, const Io &io, int T, int obs_mode, const ManyPolicy &pol, hipStream_t s);
#define MAPF_DECLARE_RUNTIME_L(L) MAPF_FOR_MW(MAPF_DECLARE_RUNTIME, L)
MAPF_FOR_LPE(MAPF_DECLARE_RUNTIME_L)
#undef MAPF_DECLARE_RUNTIME_L
#undef MAPF_DECLARE_RUNTIME
// single-agent (CTE) kernels: one unit per lanes per env
#ifndef MAPF_NO_CTE_KERNELS
#define MAPF_DECLARE_CTE(L) hipError_t launch_cte_##L(const LaunchPlan &lp, const CteIo &io, bool step, hipStream_t s, CteMany many);
MAPF_FOR_LPE(MAPF_DECLARE_CTE)
#undef MAPF_DECLARE_CTE
#endif

// Status of the launch just made.  hipGetLastError() also returns (and clears) an error some earlier, unrelated call
// left on this thread (torch, RCCL, an event query), so stale state is dropped right before the launch and only what
// the launch itself raised is reported.
#define LAUNCH_CHECKED(...)                          \
    do {                                             \
        (void)hipGetLastError();                     \
        hipLaunchKernelGGL(__VA_ARGS__);             \
        return hipGetLastError();                    \
    } while (0)

// the step kernels take the head of Io as individual (preloadable) arguments
#define IO_HEAD_ARGS(io) (io).agents, (io).scal, (io).grid_rows, (io).actions, (io).B, (io).H, (io).W, (io).bn8, \
                         static_cast<const IoTail &>(io)

}  // namespace mapfk

#endif  // MAPF_ENGINE_H
