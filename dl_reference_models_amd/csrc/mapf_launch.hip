// mapf_launch.hip -- one LAUNCH unit of libmapfstep.so (mapf_engine.h says how the library is cut up).
//
// Compiled once per selection; exactly one of these is defined on the command line (dl_reference_models_amd/build.py):
//   -DMAPF_TU_SPECIAL=<id>                 the step kernels of prebuilt specialisation <id> (MAPF_SPECIALIZATIONS)
//   -DMAPF_TU_LPE=<L> -DMAPF_TU_MW=<MW>    the runtime-config kernels for groups of L lanes and MW-bit window masks
//   -DMAPF_TU_CTE=<L>                      the single-agent (CTE) kernels for groups of L lanes
// Each unit instantiates its kernels where it launches them and exports plain functions (declared in mapf_engine.h);
// nothing here touches a handle.

#include "mapf_engine.h"

#if !defined(MAPF_TU_SPECIAL) && !defined(MAPF_TU_LPE) && !defined(MAPF_TU_CTE)
#define MAPF_TU_SPECIAL 1  // (compiled without a selection -- e.g. a bare `hipcc -c` -- the unit holds the headline shape)
#endif

namespace mapfk {

#define MAPF_CAT_(a, b) a##b
#define MAPF_CAT(a, b) MAPF_CAT_(a, b)
#define MAPF_CAT4_(a, b, c, d) a##b##c##d
#define MAPF_CAT4(a, b, c, d) MAPF_CAT4_(a, b, c, d)

// small groups: both register budgets are built (k_step's WPS), the plan says which one its grid needs
template <class K, int LPE, int MW>
static hipError_t launch_fixed_step(const LaunchPlan &lp, const Io &io, hipStream_t s) {
    if constexpr (LPE == 64) {
        if (lp.wide3)
            LAUNCH_CHECKED((k_stepw<K, MW>), dim3(lp.blocks + lp.sampler_blocks), dim3(192), lp.wide_lds_bytes, s, lp.d_params,
                           IO_HEAD_ARGS(io));
    }
    if constexpr (K::kSlicedDraw && LPE <= 16) {
        if (lp.three_wave)
            LAUNCH_CHECKED((k_step3<K, LPE, MW, 0>), dim3(lp.blocks + lp.sampler_blocks), dim3(192), lp.lds_bytes, s, lp.d_params,
                           IO_HEAD_ARGS(io));
    }
    if constexpr (LPE < 32) {
        if (lp.dense)
            LAUNCH_CHECKED((k_step<K, LPE, MW, 4>), dim3(lp.blocks + lp.sampler_blocks), dim3(step_threads(LPE)), lp.lds_bytes, s,
                           lp.d_params, IO_HEAD_ARGS(io));
    }
    LAUNCH_CHECKED((k_step<K, LPE, MW, 0>), dim3(lp.blocks + lp.sampler_blocks), dim3(step_threads(LPE)), lp.lds_bytes, s,
                   lp.d_params, IO_HEAD_ARGS(io));
}

template <class K, int LPE, int MW>
static hipError_t launch_many(const LaunchPlan &lp, const Io &io, int T, int obs_mode, const ManyPolicy &pol, hipStream_t s) {
    if constexpr (LPE < 32) {  // both register budgets, as for k_step
        if (lp.many_dense)
            LAUNCH_CHECKED((k_step_many<K, LPE, MW, 4>), dim3(lp.blocks), dim3(many_threads(LPE)), lp.lds_bytes, s, lp.d_params,
                           IO_HEAD_ARGS(io), T, obs_mode, pol);
    }
    LAUNCH_CHECKED((k_step_many<K, LPE, MW>), dim3(lp.blocks), dim3(many_threads(LPE)), lp.lds_bytes, s, lp.d_params,
                   IO_HEAD_ARGS(io), T, obs_mode, pol);
}

#if defined(MAPF_TU_SPECIAL)
// ---- one prebuilt specialisation ---------------------------------------------------------------------------------------
template <int ID>
struct SpecOf;
#define MAPF_SPEC_TRAITS(ID, N_, SR_, FLAGS_, DW_, LW_, NEARBY_, MINN_, LPE_)  \
    template <>                                                               \
    struct SpecOf<ID> {                                                       \
        using K = KFixed<N_, SR_, (uint32_t)(FLAGS_), DW_, LW_, NEARBY_, MINN_>; \
        static constexpr int LPE = LPE_, MW = mask_width_for(SR_);            \
    };
MAPF_SPECIALIZATIONS(MAPF_SPEC_TRAITS)
#undef MAPF_SPEC_TRAITS
using Spec = SpecOf<MAPF_TU_SPECIAL>;

hipError_t MAPF_CAT(launch_special_step_, MAPF_TU_SPECIAL)(const LaunchPlan &lp, const Io &io, hipStream_t s) {
    return launch_fixed_step<Spec::K, Spec::LPE, Spec::MW>(lp, io, s);
}
hipError_t MAPF_CAT(launch_special_many_, MAPF_TU_SPECIAL)(const LaunchPlan &lp, const Io &io, int T, int obs_mode,
                                                           const ManyPolicy &pol, hipStream_t s) {
    return launch_many<Spec::K, Spec::LPE, Spec::MW>(lp, io, T, obs_mode, pol, s);
}

#elif defined(MAPF_TU_LPE)
// ---- the runtime-config kernels of one group width and window-mask width -------------------------------------------------
#ifndef MAPF_TU_MW
#error "-DMAPF_TU_LPE needs -DMAPF_TU_MW"
#endif
hipError_t MAPF_CAT4(launch_runtime_, MAPF_TU_LPE, _, MAPF_TU_MW)(int kind, const LaunchPlan &lp, const Io &io, hipStream_t s) {
    constexpr int LPE = MAPF_TU_LPE, MW = MAPF_TU_MW;
    if (kind == KIND_STEP) {
        if constexpr (LPE <= 16) {
            if (lp.rt_sliced) return launch_fixed_step<KRuntimeSliced, LPE, MW>(lp, io, s);
        }
        return launch_fixed_step<KRuntime, LPE, MW>(lp, io, s);
    }
    if (kind == KIND_RESET)
        LAUNCH_CHECKED((k_reset<KRuntime, LPE, MW>), dim3(lp.blocks), dim3(64), lp.lds_bytes, s, lp.d_params, io);
    LAUNCH_CHECKED((k_observe<KRuntime, LPE, MW>), dim3(lp.blocks), dim3(64), lp.lds_bytes, s, lp.d_params, io);
}
hipError_t MAPF_CAT4(launch_runtime_many_, MAPF_TU_LPE, _, MAPF_TU_MW)(const LaunchPlan &lp, const Io &io, int T, int obs_mode,
                                                                      const ManyPolicy &pol, hipStream_t s) {
    return launch_many<KRuntime, MAPF_TU_LPE, MAPF_TU_MW>(lp, io, T, obs_mode, pol, s);
}

#elif defined(MAPF_TU_CTE)
// ---- the single-agent (CTE) kernels of one group width -------------------------------------------------------------------
hipError_t MAPF_CAT(launch_cte_, MAPF_TU_CTE)(const LaunchPlan &lp, const CteIo &io, bool step, hipStream_t s, CteMany many) {
    constexpr int L = MAPF_TU_CTE;
    if (step && many.T > 1) LAUNCH_CHECKED((k_cte_step<L, true>), dim3(lp.blocks), dim3(128), lp.lds_bytes, s, lp.d_params, io, many);
    if (step) LAUNCH_CHECKED((k_cte_step<L, false>), dim3(lp.blocks + io.sampler_blocks), dim3(128), lp.lds_bytes, s, lp.d_params, io, many);
    LAUNCH_CHECKED((k_cte_reset<L>), dim3(lp.blocks), dim3(64), lp.lds_bytes, s, lp.d_params, io);
}
#endif

}  // namespace mapfk
