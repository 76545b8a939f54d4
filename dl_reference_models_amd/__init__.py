"""dl_reference_models_amd -- MI355X-native vectorized step engine for the multi-agent grid env.

Only the hot path of the reference's ``src/environments`` MultiAgentEnv lives here:

    csrc/mapf_step.hip     HIP kernels (gfx950) + the C ABI declared in include/mapf_step.h
    _lib.py / build.py     ctypes binding and in-tree hipcc build of libmapfstep.so
    vec_env.py             VecReferenceModel: batched tensor API (B envs per GPU)
    reference_model_multi_agent.py   ReferenceModel: the reference's MultiAgentEnv / gymnasium dict API
    get_grid.py, actions.py          named grids and action ids (host-side data)
    sharding.py            one-process-per-GPU env sharding for bench / rollout workers

There is no CPU implementation in this package; the HIP library must be built.
"""

__version__ = "0.1.0"

from .actions import DOWN, LEFT, NO_OP, RIGHT, UP  # noqa: F401


def __getattr__(name):
    # lazy: importing the package must not require torch / a GPU
    if name == "VecReferenceModel":
        from .vec_env import VecReferenceModel

        return VecReferenceModel
    if name == "ReferenceModel":
        from .reference_model_multi_agent import ReferenceModel

        return ReferenceModel
    raise AttributeError(name)
