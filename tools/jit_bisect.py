import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from trace_util import EngineStepper, OracleStepper, compare_steppers, synth_grids
base = {'env_name': 'synthetic', 'num_agents': 64, 'sensor_range': 4, 'steps_per_episode': 61, 'normalize_goal_delta': False, 'include_goal_distance': False, 'include_action_mask_in_obs': True, 'include_blocking_pressure_in_obs': True, 'lifelong_mapf': False, 'enable_lock_metrics': True, 'deadlock_window_steps': 37, 'livelock_window_steps': 1, 'lock_nearby_manhattan': 3, 'lock_min_neighbors': 3, 'lock_progress_epsilon': 3.7}
variants = {"as failed": {}, "sr2": {"sensor_range": 2}, "sr3": {"sensor_range": 3}, "lifelong": {"lifelong_mapf": True}, "lw16 dw8": {"deadlock_window_steps": 8, "livelock_window_steps": 16},
            "nearby2 minn1": {"lock_nearby_manhattan": 2, "lock_min_neighbors": 1}, "N40": {"num_agents": 40}, "N33 sr4": {"num_agents": 33}, "N20 sr4": {"num_agents": 20}, "N12 sr4": {"num_agents": 12},
            "sr5": {"sensor_range": 5}, "sr4 no mask": {"include_action_mask_in_obs": False}}
for name, ch in variants.items():
    cfg = dict(base, **ch); N = cfg["num_agents"]; B = 15
    grids = synth_grids(B, 28, 28, 0.0, N, base_seed=1); seeds = list(range(B))
    acts = np.random.default_rng(3).integers(0, 5, size=(40, B, N)).astype(np.int8)
    for jit in (True, False):
        try:
            eng = EngineStepper(grids, cfg, seeds=seeds, jit_specialize=jit)
            tag = "jit" if eng.env.launch_info()["jit"] else "runtime (" + eng.env.launch_info()["jit_note"][:40] + ")"
            compare_steppers(eng, OracleStepper(grids, cfg, seeds=seeds), acts)
            print(f"{name:16s} {tag:12s} ok", flush=True)
        except AssertionError as e:
            print(f"{name:16s} {tag:12s} FAIL {str(e)[:110]}", flush=True)
