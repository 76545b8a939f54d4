#!/usr/bin/env python3
"""Soak of the SPECIALISED step kernels (BASELINE.json shapes: N = 8 / 4, sensor
range 2, default lock windows, L = 33 / 28), whose episode boundaries go through pre-drawn placement slots, the sliced
background draw in the observation wave and the fast reset: random grid sizes, obstacle densities, batch sizes (ragged
last wave), episode lengths (down to 1), staggered phases, action distributions (incl. goal seeking, so that episodes
also end by success at arbitrary steps), with and without the terminal observation; engine vs oracle, bit-exact, and
positions, goals and generator words every few steps.  tests/test_soak_gpu.py runs a short one (run_soak below); the
long ones are recorded in DESIGN.md section 2.
Usage: python tools/soak_specialized.py [master_seed] [cases]
Environment: SOAK_GENERIC=1 (the runtime-config kernels on the same shapes), SOAK_DENSE=1 (the 128-register builds), SOAK_N=4|8|16 (one agent count only; 16 = specialisation 6, sensor_range 3), SOAK_FINAL=0|1 (terminal observation off / on), SOAK_KNOBS=key=value,... (engine knobs of the env config for every handle), SOAK_SEQ=1 (sequential
reset, the A/B), SOAK_ONLY=<case> (run one case of the sequence), SOAK_WATCH=<case>:<env> (print that env's placement slot
and staging buffer before every step).  Cases are NOT independent on the GPU side: what a kernel finds in LDS depends on
the launches before it, so a failure is reported with its case number in the sequence."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from trace_util import EngineStepper, OracleStepper, _eq, synth_grids


def run_soak(master=2026, cases=200, only_n=None, final=None, sequential=False, only=-1, watch=None, log=print,
             poll_errors=False, knobs=None):
    """Returns None when every case matched, else the failure text.  knobs: engine knobs for every handle (e.g.
    {"register_budget": "dense"}, {"background_draw": "sampler_workgroups"})."""
    rng = np.random.default_rng(master)
    t0 = time.time()
    for case in range(cases):
        N = int(rng.choice([8, 4]))
        if only_n: N = int(only_n)
        H, W = int(rng.integers(3, 65)), int(rng.integers(3, 65))
        while H * W < 4 * N:
            H, W = int(rng.integers(3, 65)), int(rng.integers(3, 65))
        cfg = {"env_name": "synthetic", "num_agents": N, "sensor_range": 2,
               "steps_per_episode": int(rng.choice([1, 2, 3, 5, 9, 17, 40, 100])), "include_action_mask_in_obs": bool(rng.integers(0, 2))}
        if os.environ.get("SOAK_MASK"):  # (development builds that only hold the L = 33 specialisations)
            cfg["include_action_mask_in_obs"] = True
        if N == 16:  # specialisation 6: the reference's own training setup (main.py:55-67), sampler workgroups in front
            cfg.update(sensor_range=3, include_action_mask_in_obs=False)
        B = int(rng.choice([1, 7, 8, 9, 63, 64, 65, 200, 513]))
        density = float(rng.choice([0.0, 0.05, 0.2, 0.4]))
        grids = synth_grids(B, H, W, density, N, base_seed=int(rng.integers(0, 10**6)))
        seeds = [int(x) for x in rng.integers(0, 10**6, size=B)]
        want_final = bool(rng.integers(0, 2))
        if final is not None: want_final = bool(final)
        orc = OracleStepper(grids, cfg, seeds=seeds)
        if only >= 0 and case != only:
            class _Null:  # keeps the generator in step without touching the GPU
                def reset(self): return orc.reset()
                def set_step_counts(self, c): pass
                def step(self, a): return None
            eng = _Null()
            orc.reset()
        else:
            kw = {"force_sequential_reset": True} if sequential else {}
            kw.update(knobs or {})
            if os.environ.get("SOAK_DENSE"):
                kw["register_budget"] = "dense"
            if os.environ.get("SOAK_GENERIC"):  # the same shapes on the runtime-config kernels (full groups of 4 / 8 agents:
                kw["force_generic_kernel"] = True  # sliced draw + three-wave kernel since round 3; background_draw knob: sampler workgroups)
            eng = EngineStepper(grids, cfg, seeds=seeds, want_final_obs=want_final, **kw)
            assert eng.env.launch_info()["specialized_kernel"] in ((0,) if os.environ.get("SOAK_GENERIC") else (1, 2, 4, 5, 6)), eng.env.launch_info()
            _eq("reset obs", eng.reset(), orc.reset())
        counts = rng.integers(0, cfg["steps_per_episode"], size=B)
        eng.set_step_counts(counts); orc.set_step_counts(counts)
        greedy = float(rng.choice([0.0, 0.5, 0.9]))
        T = 110
        try:
            for t in range(T):
                pos, goal = orc.positions().astype(int), orc.goals().astype(int)
                d = goal - pos
                vert = np.where(d[..., 0] < 0, 1, 3); horz = np.where(d[..., 1] > 0, 2, 4)
                g = np.where(np.abs(d[..., 0]) >= np.abs(d[..., 1]), vert, horz)
                g = np.where((d == 0).all(-1), 0, g)
                a = np.where(rng.random(g.shape) < greedy, g, rng.integers(0, 5, g.shape)).astype(np.int8)
                if watch and case == watch[0] and not isinstance(eng, type(None)) and hasattr(eng, "env"):
                    import ctypes as C
                    sl = np.zeros((B, N), np.uint32); sg = np.zeros((B, 4 * N + 4), np.uint32)
                    eng.env._lib.mapf_debug_slots(eng.env._h, sl.ctypes.data_as(C.c_void_p), sg.ctypes.data_as(C.c_void_p), None)
                    log(f"  WATCH before step {t}: slots {[hex(x) for x in sl[watch[1]]]} stage {[hex(x) for x in sg[watch[1]][:8]]} "
                        f"ctr {eng.env.get_state()['counters'][watch[1]][:2].tolist()}")
                ra, rb = eng.step(a), orc.step(a)
                if ra is None:
                    continue
                if poll_errors:  # (checking build: an index that left its LDS region is latched like a device error)
                    eng.env.poll_error()
                bad = np.argwhere(ra["obs"] != rb["obs"])
                if len(bad):
                    e = int(bad[0][0])
                    st = eng.env.get_state()
                    dn = (rb["terminated"] | rb["truncated"]).astype(bool)
                    log(f"  DIAG step {t}: mismatching envs {sorted(set(bad[:, 0].tolist()))} done envs {np.flatnonzero(dn).tolist()}")
                    log(f"  DIAG engine starts {st['starts'][e].tolist()} goals {st['goals'][e].tolist()} pos {st['positions'][e].tolist()}")
                    log(f"  DIAG oracle starts {orc.batch.envs[e].starts.tolist()} goals {orc.goals()[e].tolist()} pos {orc.positions()[e].tolist()}")
                    log(f"  DIAG free cells {int((grids[e] == 0).sum())} grid {grids[e].shape} counters {st['counters'][e].tolist()}")
                for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
                    _eq(k, ra[k], rb[k], t)
                done = (rb["terminated"] | rb["truncated"]).astype(bool)
                if want_final and done.any():
                    _eq("final_obs", ra["final_obs"][done], rb["final_obs"][done], t)
                if t % 9 == 0 or t == T - 1:
                    _eq("positions", eng.positions(), orc.positions(), t)
                    _eq("goals", eng.goals(), orc.goals(), t)
                    _eq("rng words", eng.rng_words(), orc.rng_words(), t)
        except (AssertionError, RuntimeError) as exc:
            return (f"FAIL case {case} of master seed {master}: cfg={cfg} B={B} HxW={H}x{W} density={density} "
                    f"want_final={want_final} greedy={greedy}: {exc}")
        if case % 20 == 0:
            log(f"case {case} ok ({time.time() - t0:.0f} s)")
    log(f"soak ok: {cases} cases, master seed {master}, {time.time() - t0:.0f} s")
    return None


if __name__ == "__main__":
    env = os.environ
    w = tuple(int(x) for x in env["SOAK_WATCH"].split(":")) if env.get("SOAK_WATCH") else None
    err = run_soak(int(sys.argv[1]) if len(sys.argv) > 1 else 2026, int(sys.argv[2]) if len(sys.argv) > 2 else 200,
                   only_n=int(env["SOAK_N"]) if env.get("SOAK_N") else None,
                   final=(env["SOAK_FINAL"] == "1") if env.get("SOAK_FINAL") else None, sequential=bool(env.get("SOAK_SEQ")),
                   only=int(env.get("SOAK_ONLY", "-1")), watch=w, log=lambda m: print(m, flush=True),
                   knobs=dict(kv.split("=", 1) for kv in filter(None, env.get("SOAK_KNOBS", "").split(","))) or None)
    if err:
        print(err, flush=True)
        sys.exit(1)
