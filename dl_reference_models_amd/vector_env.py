"""Vector-env adapter: the envs of one RLlib env runner served by ONE batched engine handle.

The reference builds ``num_envs_per_env_runner`` independent ``ReferenceModel`` objects per runner
(``src/agents/ppo.py:96-100``: 8 runners x 4 envs; created through ``main.py:80-82,396``) and RLlib steps them
one after the other with per-env action dicts (``MultiAgentEnvWrapper.send_actions`` -> ``env.step`` for every
sub-env).  Doing that with one B = 1 engine per object costs a kernel launch and several device->host copies
per env per step -- as slow as the pure-Python reference.  Here the runner's envs are rows of one
``VecReferenceModel``:

    vec = ReferenceModelVectorEnv(env_config, num_envs=4)
    vec.envs[b]                       -> ``ReferenceModelRow``: the reference's attribute surface for row b
    vec.vector_reset()                -> [(obs_dict, info_dict)] * num_envs
    vec.vector_step([action_dict]*B)  -> (obs, rewards, terminateds, truncateds, infos), each a list of the dicts
                                         ``ReferenceModel.step`` returns (MA-env:695)
    vec.reset_at(b)                   -> (obs_dict, info_dict) of row b alone

and the ``BaseEnv``-style trio RLlib's OLD-stack sampler drives (``poll`` / ``send_actions`` / ``try_reset``).

The reference itself runs RLlib's NEW stack (``src/agents/ppo.py:92-100``, ``impala.py``:
``enable_env_runner_and_connector_v2=True``, ``num_envs_per_env_runner=4``), whose multi-agent env runner drives a vector
object with ``num_envs``, ``envs``, ``reset(*, seed, options) -> (observations, infos)`` and
``step(list_of_action_dicts) -> (observations, rewards, terminateds, truncateds, infos)`` with NEXT-STEP autoreset (a
row that finished is reset by the following ``step`` call, which returns its reset observation, zero rewards and
``False`` flags for it and ignores the action sent for it).  ``ReferenceModelAutoresetVectorEnv`` below offers that
surface over the same single handle.  ``ray`` is not installed where this was written and tested, so both adapters are
duck-typed against the documented interfaces; nobody has handed them to a live runner yet (INTEGRATION.md says so).

Cost model: ONE kernel launch, ONE stream sync and TWO device->host copies (observations; one blob with
rewards / done flags / info) per *vector* step, whatever ``num_envs`` is; per-env Python work is building
the dicts.  State attributes of a row (``_positions_arr``, ``step_count`` ...) come from one batched
``get_state`` that is fetched lazily, at most once per vector step, and only if somebody reads them
(``src/trainers/callbacks.py:111-131,265-307`` do, at episode end).

Episode ends: rows are NOT reset inside the step (``auto_reset`` off), exactly like the reference object --
the caller resets a finished row (``reset_at`` / ``try_reset``), which is how RLlib drives sub-envs.
"""

from __future__ import annotations

import logging

import numpy as np
import torch

from . import _lib as L
from .actions import LEFT, NO_OP
from .reference_model_multi_agent import ReferenceModel as _Facade
from .vec_env import VecReferenceModel

logger = logging.getLogger(__name__)


class ReferenceModelRow:
    """Row ``b`` of a ``ReferenceModelVectorEnv`` with the attribute surface callers of the reference env read
    (``callbacks.py:111-131,265-307``, ``main.py:154,265,290,314``).  Read-only views: state is written through
    ``ReferenceModelVectorEnv.set_row_state``."""

    def __init__(self, vec: "ReferenceModelVectorEnv", b: int):
        self._vec, self._b = vec, b
        t = vec._template
        # static surface, shared with the template facade object (spaces, config, constants)
        for name in ("possible_agents", "agents", "observation_spaces", "action_spaces", "observation_space",
                     "action_space", "_obs_slices", "sensor_range", "steps_per_episode", "lifelong_mapf", "deterministic",
                     "info_mode", "seed", "_num_agents", "_coord_dtype", "_agent_index", "_action_deltas",
                     "normalize_goal_delta", "include_goal_distance", "include_action_mask_in_obs",
                     "include_blocking_pressure_in_obs", "enable_lock_metrics", "deadlock_window_steps",
                     "livelock_window_steps", "lock_nearby_manhattan", "lock_progress_epsilon", "lock_min_neighbors",
                     "EMPTY_CELL", "OBSTACLE_CELL", "OTHER_AGENT_CELL", "OWN_GOAL_CELL", "OTHER_GOAL_CELL",
                     "UNASSIGNED_OWNER", "TRAVERSABLE_LOCAL_VALUES"):
            setattr(self, name, getattr(t, name))
        self.grid = vec._grids[b if vec._grids.shape[0] > 1 else 0]
        self._free_positions = np.argwhere(self.grid == 0).astype(np.int16, copy=False)

    # ---- state views (one batched get_state per vector step, fetched on first use) ----------------------
    def _s(self):
        return self._vec._state()

    _positions_arr = property(lambda self: self._s()["positions"][self._b])
    _goals_arr = property(lambda self: self._s()["goals"][self._b])
    _starts_arr = property(lambda self: self._s()["starts"][self._b])
    _reached_arr = property(lambda self: self._s()["reached"][self._b].astype(np.bool_))
    _completed_once_arr = property(lambda self: self._s()["completed_once"][self._b].astype(np.bool_))
    _blocking_pressure_prev_arr = property(lambda self: self._s()["pressure_prev"][self._b].astype(np.float32))

    def _ctr(self, idx):
        return self._s()["counters"][self._b, idx]

    step_count = property(lambda self: int(self._ctr(L.CTR_STEP_COUNT)))
    _episode_blocking_count = property(lambda self: float(self._ctr(L.CTR_BLOCKING_COUNT)))
    _episode_goals_reached_total = property(lambda self: float(self._ctr(L.CTR_GOALS_REACHED_TOTAL)))
    _episode_deadlock_events = property(lambda self: float(self._ctr(L.CTR_DEADLOCK_EVENTS)))
    _episode_livelock_events = property(lambda self: float(self._ctr(L.CTR_LIVELOCK_EVENTS)))
    _episode_deadlock_steps = property(lambda self: float(self._ctr(L.CTR_DEADLOCK_STEPS)))
    _episode_livelock_steps = property(lambda self: float(self._ctr(L.CTR_LIVELOCK_STEPS)))

    @property
    def goal_reached_once(self):
        done = self._completed_once_arr
        return {aid: bool(done[i]) for i, aid in enumerate(self.agents)}

    @property
    def positions(self):
        p = self._positions_arr
        return {aid: p[i] for i, aid in enumerate(self.agents)}

    @property
    def goals(self):
        g = self._goals_arr
        return {aid: g[i] for i, aid in enumerate(self.agents)}

    @property
    def starts(self):
        s = self._starts_arr
        return {aid: s[i] for i, aid in enumerate(self.agents)}

    @property
    def unwrapped(self):
        return self

    def get_agent_ids(self):
        return set(self.agents)

    # ---- per-row calls (each is a launch of its own: use the vector calls on the hot path) ---------------
    def reset(self, *, seed=None, options=None):
        return self._vec.reset_at(self._b)

    def step(self, action_dict):
        """Steps this row ALONE (the other rows do not move): one launch per call, for callers that insist on the
        per-object protocol.  A runner should call ``vector_step`` instead."""
        return self._vec._step_rows([self._b], [action_dict])[0]

    def render(self, mode="human"):
        return None


class ReferenceModelVectorEnv:
    def __init__(self, env_config: dict, num_envs: int):
        cfg = dict(env_config)
        self.num_envs = B = int(num_envs)
        if B < 1:
            raise ValueError("num_envs must be >= 1")
        # a B = 1 facade object supplies (and validates) everything static: spaces, layout, config clamps
        tcfg = dict(cfg)
        tcfg.pop("seeds", None)
        self._template = _Facade(tcfg)
        t = self._template
        self.agents, self._n = t.agents, t._num_agents
        ecfg = dict(cfg)
        ecfg["num_envs"] = B
        ecfg.setdefault("grid", t.grid)
        if "seeds" not in ecfg and "rng_words" not in ecfg:
            seed = cfg.get("seed", None)
            ecfg["seeds"] = [None] * B if seed is None else [int(seed) + b for b in range(B)]
        self._engine = VecReferenceModel(ecfg)
        self._grids = self._engine.grids
        self.device = self._engine.device
        self._state_cache = None
        self._live = np.ones(B, dtype=bool)  # rows that have been reset and are not done
        e = self._engine
        # pinned host mirrors: one copy for the observations, one for the blob of small outputs
        self._h_obs = torch.empty(e._obs.shape, dtype=torch.float32).pin_memory()
        self._h_blob = torch.empty(e._out_blob.shape, dtype=torch.uint8).pin_memory()
        off = lambda tns: tns.data_ptr() - e._out_blob.data_ptr()
        nb = lambda tns: tns.numel() * tns.element_size()
        hb = self._h_blob.numpy()
        self._v_rew = hb[off(e._rewards):off(e._rewards) + nb(e._rewards)].view(np.float32).reshape(tuple(e._rewards.shape))
        self._v_ia = hb[off(e._info_all):off(e._info_all) + nb(e._info_all)].view(np.float32).reshape(tuple(e._info_all.shape))
        self._v_iag = hb[off(e._info_agent):off(e._info_agent) + nb(e._info_agent)].reshape(tuple(e._info_agent.shape))
        self._v_term = hb[off(e._terminated):off(e._terminated) + nb(e._terminated)]
        self._v_trunc = hb[off(e._truncated):off(e._truncated) + nb(e._truncated)]
        self._acts = torch.zeros((B, self._n), dtype=torch.int8).pin_memory()
        self._acts_dev = torch.zeros((B, self._n), dtype=torch.int8, device=self.device)
        self._mask_dev = torch.zeros((B,), dtype=torch.uint8, device=self.device)
        self.envs = [ReferenceModelRow(self, b) for b in range(B)]
        self._pending = None  # BaseEnv-style poll()/send_actions() hand-over

    # ------------------------------------------------------------------------------------------------------
    def _state(self):
        if self._state_cache is None:
            self._state_cache = self._engine.get_state()
        return self._state_cache

    def set_row_state(self, b: int, **kw):
        """Overwrite state arrays of row b (keys of ``VecReferenceModel.set_state`` without the batch axis)."""
        s = self._engine.get_state()
        for k, v in kw.items():
            s[k][b] = np.asarray(v, dtype=s[k].dtype).reshape(s[k][b].shape)
        self._engine.set_state(**{k: s[k] for k in kw})
        self._state_cache = None

    def close(self):
        self._engine.close()
        self._template.close()

    # ---- building the reference's dicts from the host mirrors ----------------------------------------------
    def _obs_dicts(self, rows, obs_np):
        agents, full, t = self.agents, self._template.info_mode == "full", self._template
        out_obs, out_info = [], []
        if full:
            st = self._state()
        for b in rows:
            ob = obs_np[b]
            out_obs.append(dict(zip(agents, ob)))  # rows of a fresh array (copied out of the pinned mirror per step)
            if full:
                info = {}
                for i, aid in enumerate(agents):
                    sl = t._obs_slices
                    local = ob[i][sl["local_obs"]].astype(np.uint8).reshape(t._view_side, t._view_side)
                    mask = (ob[i][sl["action_mask"]].astype(np.int8) if "action_mask" in sl else t.get_action_mask(local))
                    info[aid] = {"position": np.asarray(st["positions"][b, i]), "goal": np.asarray(st["goals"][b, i]),
                                 "goal_delta": np.asarray(ob[i][sl["goal_delta"]], dtype=np.float32), "action_mask": mask,
                                 "local_obs": local}
                out_info.append(info)
            else:
                out_info.append({aid: {} for aid in agents})
        if t.validate_observation_space:
            for b, od in zip(rows, out_obs):
                for aid, o in od.items():
                    t._check_obs(aid, o, "step")
        return out_obs, out_info

    def _fetch(self, want_small=True):
        """Device -> pinned host: observations (+ the blob of small outputs), one sync."""
        e = self._engine
        stream = torch.cuda.current_stream(self.device)
        self._h_obs.copy_(e._obs, non_blocking=True)
        if want_small:
            self._h_blob.copy_(e._out_blob, non_blocking=True)
        stream.synchronize()
        return self._h_obs.numpy().copy()

    # ---- vector API ---------------------------------------------------------------------------------------
    def vector_reset(self):
        self._engine.reset()
        self._state_cache = None
        self._live[:] = True
        obs_np = self._fetch(want_small=False)
        obs, infos = self._obs_dicts(range(self.num_envs), obs_np)
        return list(zip(obs, infos))

    def reset_at(self, b: int):
        self._mask_dev.zero_()
        self._mask_dev[b] = 1
        self._engine.reset(self._mask_dev)
        self._state_cache = None
        self._live[b] = True
        obs_np = self._fetch(want_small=False)
        obs, infos = self._obs_dicts([b], obs_np)
        return obs[0], infos[0]

    def vector_step(self, action_dicts):
        """One step of EVERY row: ``action_dicts[b]`` is the reference's ``action_dict`` for row b."""
        if len(action_dicts) != self.num_envs:
            raise ValueError(f"need one action dict per env ({self.num_envs})")
        res = self._step_rows(range(self.num_envs), action_dicts)
        return tuple(list(col) for col in zip(*res))

    def _step_rows(self, rows, action_dicts):
        rows = list(rows)
        agents, n = self.agents, self._n
        acts = self._acts.numpy()
        all_rows = len(rows) == self.num_envs
        first_bad = None
        for b, ad in zip(rows, action_dicts):
            if not ad or any(aid not in ad for aid in agents):  # MA-env:498-500
                ad = dict.fromkeys(agents, NO_OP)
                logger.warning("No actions provided or missing agent actions. Defaulting to no-op actions: %s", ad)
            row = acts[b]
            for i, aid in enumerate(agents):
                a = int(ad[aid])
                if a < NO_OP or a > LEFT:
                    if first_bad is None:
                        first_bad = (b, a, aid)
                    a = 5  # the kernel stops that env's agent loop there, like the reference (MA-env:504-506)
                row[i] = a
        e = self._engine
        self._acts_dev.copy_(self._acts, non_blocking=True)
        if all_rows:
            e.step(self._acts_dev, auto_reset=False)
        else:
            # rows stepped alone (per-object callers, rows waiting for a reset): the engine's per-env step mask leaves
            # every other row untouched -- state, generator, counters, episode statistics, error latch
            sel = np.zeros(self.num_envs, dtype=np.uint8)
            sel[rows] = 1
            self._mask_dev.copy_(torch.from_numpy(sel), non_blocking=False)
            e.step(self._acts_dev, auto_reset=False, env_mask=self._mask_dev)
        self._state_cache = None
        obs_np = self._fetch()
        if first_bad is not None:
            try:
                e.poll_error()
            except ValueError:
                pass
            raise ValueError(f"Invalid action {first_bad[1]} for {first_bad[2]} (env {first_bad[0]})")
        e.poll_error()
        lifelong = self._template.lifelong_mapf
        rew, ia, iag = self._v_rew.tolist(), self._v_ia, self._v_iag.tolist()
        term, trunc = self._v_term.astype(bool).tolist(), self._v_trunc.astype(bool).tolist()
        ia64 = ia.astype(np.float64)
        obs_l, info_l = self._obs_dicts(rows, obs_np)
        need_state = lifelong  # completion_ratio / throughput are float64 quotients of integers (MA-env:638,653-655)
        st = self._state() if need_state else None
        out = []
        for k, b in enumerate(rows):
            info = info_l[k]
            r = dict(zip(agents, rew[b]))
            row_ia = ia64[b]
            grt, bct = float(row_ia[1]), float(row_ia[3])
            for i, aid in enumerate(agents):
                d = info[aid]
                d["blocking"] = float(iag[b][i][0])
                d["goal_reached_step"] = float(iag[b][i][1])
                d["goals_reached_total"] = grt
                d["blocking_count_total"] = bct
            info_all = {
                "goals_reached_step": float(row_ia[0]), "goals_reached_total": grt,
                "blocking_count_step": float(row_ia[2]), "blocking_count_total": bct,
                "deadlock_step": float(row_ia[4]), "livelock_step": float(row_ia[5]),
                "deadlock_event_step": float(row_ia[6]), "livelock_event_step": float(row_ia[7]),
                "deadlock_events_total": float(row_ia[8]), "livelock_events_total": float(row_ia[9]),
                "deadlock_steps_total": float(row_ia[10]), "livelock_steps_total": float(row_ia[11]),
            }
            if lifelong:
                info_all["completion_ratio"] = float(np.mean(st["completed_once"][b].astype(np.bool_)))
                info_all["throughput"] = grt / float(max(int(st["counters"][b, L.CTR_STEP_COUNT]), 1))
            info["__all__"] = info_all
            tb, ub = term[b], trunc[b]
            terminated = dict.fromkeys(agents, tb)
            truncated = dict.fromkeys(agents, ub)
            terminated["__all__"] = tb
            truncated["__all__"] = ub
            if tb or ub:
                self._live[b] = False
            out.append((obs_l[k], r, terminated, truncated, info))
        return out

    # ---- BaseEnv-style protocol (ray.rllib.env.base_env.BaseEnv: poll / send_actions / try_reset) ----------
    def poll(self):
        """({env_id: obs}, {env_id: rewards}, {env_id: terminateds}, {env_id: truncateds}, {env_id: infos}, {})"""
        if self._pending is None:
            res = self.vector_reset()
            z = dict.fromkeys(self.agents, 0.0)
            f = dict.fromkeys(list(self.agents) + ["__all__"], False)
            self._pending = ({b: o for b, (o, _) in enumerate(res)}, {b: dict(z) for b in range(self.num_envs)},
                             {b: dict(f) for b in range(self.num_envs)}, {b: dict(f) for b in range(self.num_envs)},
                             {b: i for b, (_, i) in enumerate(res)})
        out, self._pending = self._pending, None
        return (*out, {})

    def send_actions(self, action_dict):
        rows = sorted(action_dict)  # all rows: one launch; a subset (some rows wait for a reset): the snapshot path
        res = self._step_rows(rows, [action_dict[b] for b in rows])
        cols = list(zip(*res))
        self._pending = tuple({b: c[k] for k, b in enumerate(rows)} for c in cols)

    def try_reset(self, env_id=None, *, seed=None, options=None):
        if env_id is None:
            res = self.vector_reset()
            return {b: o for b, (o, _) in enumerate(res)}, {b: i for b, (_, i) in enumerate(res)}
        obs, info = self.reset_at(int(env_id))
        return {env_id: obs}, {env_id: info}

    def get_sub_environments(self):
        return self.envs


class ReferenceModelAutoresetVectorEnv(ReferenceModelVectorEnv):
    """The vector surface RLlib's new-stack multi-agent env runner drives (see the module docstring): ``num_envs``,
    ``envs``, ``reset(*, seed=None, options=None)`` and ``step(action_dicts)`` with next-step autoreset, over ONE engine
    handle.  A step in which no row restarts is one launch; a step in which some rows restart is two (the masked step of
    the live rows, the masked reset of the finished ones)."""

    def __init__(self, env_config: dict, num_envs: int):
        super().__init__(env_config, num_envs)
        self._needs_reset = np.zeros(self.num_envs, dtype=bool)

    def reset(self, *, seed=None, options=None):
        """``seed`` and ``options`` are ignored, as by the reference env (MA-env:440)."""
        res = self.vector_reset()
        self._needs_reset[:] = False
        return [o for o, _ in res], [i for _, i in res]

    def step(self, action_dicts):
        if len(action_dicts) != self.num_envs:
            raise ValueError(f"need one action dict per env ({self.num_envs})")
        restart = np.flatnonzero(self._needs_reset).tolist()
        live = [b for b in range(self.num_envs) if not self._needs_reset[b]]
        out = [None] * self.num_envs
        if live:
            res = self._step_rows(live, [action_dicts[b] for b in live])
            for b, r in zip(live, res):
                out[b] = r
                if r[2]["__all__"] or r[3]["__all__"]:
                    self._needs_reset[b] = True
        if restart:
            sel = np.zeros(self.num_envs, dtype=np.uint8)
            sel[restart] = 1
            self._mask_dev.copy_(torch.from_numpy(sel), non_blocking=False)
            self._engine.reset(self._mask_dev)
            self._state_cache = None
            obs_np = self._fetch(want_small=False)
            obs, infos = self._obs_dicts(restart, obs_np)
            zeros = dict.fromkeys(self.agents, 0.0)
            flags = dict.fromkeys(list(self.agents) + ["__all__"], False)
            for k, b in enumerate(restart):
                out[b] = (obs[k], dict(zeros), dict(flags), dict(flags), infos[k])
                self._needs_reset[b] = False
                self._live[b] = True
        return tuple(list(col) for col in zip(*out))
