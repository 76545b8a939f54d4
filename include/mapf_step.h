/*
 * mapf_step.h -- C ABI of libmapfstep.so, the MI355X (gfx950) vectorized step engine that replaces
 * the hot path of the reference's multi-agent grid environment
 * (/root/reference/src/environments/reference_model_multi_agent.py, "MA-env" below).
 *
 * One handle owns B independent env instances resident in HBM on one GPU.  Plain pointers and sizes
 * only; no torch / C++ types cross this boundary.  A handle is NOT thread-safe (the reference env is
 * single-threaded and not re-entrant either); use one handle per host thread / per GPU.
 *
 * Pointer convention
 *   - "host"   : ordinary host memory, copied synchronously inside the call (setup / inspection calls)
 *   - "device" : HIP device memory on the handle's GPU, caller-owned, used asynchronously on `stream`
 *                (the hot-path calls mapf_reset / mapf_step).  `stream` is a hipStream_t passed as void*
 *                (NULL = the default stream).
 *
 * Every function returns MAPF_OK (0) or a negative MAPF_ERR_* code; mapf_last_error() gives the text.
 * Errors the reference raises as Python exceptions *inside* step() (bad action -> ValueError MA-env:504-506,
 * no respawn cell -> RuntimeError MA-env:296-298) happen per env on the device; they are latched in a
 * device-side error record that mapf_poll_error() reads back.
 *
 * Reference interface each entry point replaces (a maintainer binds these with ctypes):
 *   mapf_create + mapf_set_grids + mapf_set_rng_state (+ mapf_set_fixed_starts_goals)
 *                         <- ReferenceModel.__init__            MA-env:34-184 (config keys :38-61, state block :82-120,
 *                                                               RNG :74-78, grid :80, fixed tables :124-132)
 *   mapf_reset            <- ReferenceModel.reset               MA-env:440-472 (+ generate_starts_goals :267-282)
 *   mapf_step             <- ReferenceModel.step                MA-env:474-695 (+ get_obs :707-747, get_action_mask :749-773,
 *                                                               _flatten_observation :306-328, _assign_new_goal :284-304,
 *                                                               lock detector :374-438)
 *   mapf_get_state / mapf_set_state
 *                         <- the private arrays callers and tests read or poke: _positions_arr, _goals_arr, _starts_arr,
 *                            _reached_arr, _completed_once_arr, _blocking_pressure_prev_arr, step_count, _episode_* counters
 *                            (MA-env:83-89, :63-69; read by src/trainers/callbacks.py:111-131,265-307 and main.py:265,314)
 *   mapf_step_many        <- T x step() (+ reset() on done) for an action stream known up front: the loop of
 *                                                               scripts/benchmark_multi_agent_env.py:85-95
 *   mapf_get_episode_stats, mapf_episode_stats_async
 *                         <- what ReferenceModelCallbacks.on_episode_end reads from the env   src/trainers/callbacks.py:236-345
 *   mapf_observe          <- get_obs / get_action_mask / _flatten_observation called on a static state
 *                                                               MA-env:707-773, :306-328
 *   mapf_obs_len          <- _build_obs_layout                  MA-env:238-265
 *   mapf_assign_new_goal  <- _assign_new_goal(agent_idx) called by itself (the reference's lifelong tests do,
 *                            tests/test_reference_model_lifelong.py:132-173)                     MA-env:284-304
 */
#ifndef MAPF_STEP_H
#define MAPF_STEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAPF_VERSION_MAJOR 0
#define MAPF_VERSION_MINOR 1

/* limits of this build (checked by mapf_create) */
#define MAPF_MAX_DIM 64          /* grid height and width */
#define MAPF_MAX_AGENTS 64       /* agents per env (one wavefront lane per agent) */
#define MAPF_MAX_SENSOR_RANGE 5  /* view side 2*sr+1 <= 11 */
#define MAPF_MAX_LOCK_WINDOW 64  /* deadlock / livelock window steps */

/* config flags (defaults of the reference in brackets, MA-env:41-61) */
#define MAPF_FLAG_NORMALIZE_GOAL_DELTA 1u /* normalize_goal_delta [on]  */
#define MAPF_FLAG_GOAL_DISTANCE 2u        /* include_goal_distance [off] */
#define MAPF_FLAG_ACTION_MASK 4u          /* include_action_mask_in_obs [off] */
#define MAPF_FLAG_BLOCKING_PRESSURE 8u    /* include_blocking_pressure_in_obs [on] */
#define MAPF_FLAG_LIFELONG 16u            /* lifelong_mapf [off] */
#define MAPF_FLAG_LOCK_METRICS 32u        /* enable_lock_metrics [on] */
#define MAPF_FLAG_DETERMINISTIC 64u       /* deterministic [off]: reset() re-places agents on fixed starts, no RNG */
#define MAPF_FLAG_SINGLE_AGENT 256u        /* the handle runs the single-agent (CTE) sibling env: use the mapf_cte_* calls */
#define MAPF_FLAG_JIT_SPECIALIZE 0x10000000u /* opt-in: when no prebuilt specialisation of the step kernels matches, compile
                                               one for exactly this configuration at mapf_create (hiprtc, a few seconds;
                                               cached per process).  Falls back to the runtime-config kernels -- with the
                                               reason in mapf_jit_status() -- when hiprtc or the kernel source is not there */
#define MAPF_FLAG_FORCE_DENSE 0x08000000u   /* engine knob (tests): the 128-register builds of the small-group step kernels
                                             * whatever the size of the grid (mapf_create picks by grid size otherwise) */
#define MAPF_FLAG_FORCE_SPARSE 0x04000000u  /* engine knob (tests): never the 128-register builds */
#define MAPF_FLAG_SAMPLER_WORKGROUPS 0x02000000u /* engine knob (tests): the runtime-config kernels pre-draw placements in
                                             * sampler workgroups of the step grid instead of slices inside the env workgroups */
#define MAPF_FLAG_TWO_WAVE_WIDE 0x00800000u /* engine knob (tests / A-B): 64-lane groups step on the two-wave kernel with the
                                             * word-per-cell LDS map (rounds 1-3) instead of the three-wave kernel with bit rows */
#define MAPF_FLAG_TABLE_WALK_OBS 0x00400000u /* engine knob (tests / A-B): the observation wave of the three-wave small-group
                                             * kernel walks the agent table after the moves (round 3) instead of preparing
                                             * both outcomes of its agent's move from bit rows before them (round 4) */
#define MAPF_FLAG_NO_BIT_ROWS 0x00200000u /* engine knob (tests / A-B): the three-wave small-group kernel as of round 3 -- no
                                           * goal / occupancy / intent bit rows, per-agent outputs by the aux wave */
#define MAPF_FLAG_SEQUENTIAL_RESET 0x20000000u /* engine knob (tests): in-kernel resets always take the sequential
                                                 * sampler (otherwise only after a Lemire rejection or when F = 2N) */
#define MAPF_FLAG_NO_CELL_MAP 0x40000000u   /* engine knob (tests / A-B): never use the LDS cell-map path of wide groups */
#define MAPF_FLAG_GENERIC_KERNEL 0x80000000u /* engine knob (tests): never pick a compile-time specialised step kernel */

/* status codes */
#define MAPF_OK 0
#define MAPF_ERR_BAD_ACTION (-1)  /* reference: ValueError "Invalid action ..." MA-env:504-506 */
#define MAPF_ERR_FEW_FREE (-2)    /* reference: ValueError, fewer than 2N free cells MA-env:270-275 */
#define MAPF_ERR_NO_RESPAWN (-3)  /* reference: RuntimeError, no cell for lifelong goal MA-env:296-298 */
#define MAPF_ERR_CONFIG (-4)      /* config outside this build's limits / inconsistent arguments */
#define MAPF_ERR_HIP (-5)         /* a HIP runtime call failed */
#define MAPF_ERR_STATE (-6)       /* call sequence error (e.g. step before grids were set) */
#define MAPF_ERR_INTERNAL (-8)    /* checking build only (-DMAPF_CHECK): an index left the LDS region it belongs to; env, site
                                   * id and the offending value are latched like the device errors above */
#define MAPF_ERR_RNG_GUARD (-7)   /* device: a bounded draw was rejected 4096 times in a row (cannot happen with a sound
                                   * stream state, p < 1e-24000): the env's RNG state is corrupt; latched like the others */

/* info_all[...] column order: the reference's info["__all__"] keys MA-env:639-655 */
#define MAPF_INFO_ALL 14
#define MAPF_INFO_GOALS_REACHED_STEP 0
#define MAPF_INFO_GOALS_REACHED_TOTAL 1
#define MAPF_INFO_BLOCKING_COUNT_STEP 2
#define MAPF_INFO_BLOCKING_COUNT_TOTAL 3
#define MAPF_INFO_DEADLOCK_STEP 4
#define MAPF_INFO_LIVELOCK_STEP 5
#define MAPF_INFO_DEADLOCK_EVENT_STEP 6
#define MAPF_INFO_LIVELOCK_EVENT_STEP 7
#define MAPF_INFO_DEADLOCK_EVENTS_TOTAL 8
#define MAPF_INFO_LIVELOCK_EVENTS_TOTAL 9
#define MAPF_INFO_DEADLOCK_STEPS_TOTAL 10
#define MAPF_INFO_LIVELOCK_STEPS_TOTAL 11
#define MAPF_INFO_COMPLETION_RATIO 12 /* emitted by the reference only in lifelong mode; always computed here */
#define MAPF_INFO_THROUGHPUT 13       /* idem */

/* per-env counters of mapf_state.counters[B][MAPF_NUM_COUNTERS] */
#define MAPF_NUM_COUNTERS 16
#define MAPF_CTR_STEP_COUNT 0           /* step_count                         MA-env:37  */
#define MAPF_CTR_HIST_ROWS 1            /* rows appended to the lock history since its reset (unsaturated) */
#define MAPF_CTR_BLOCKING_COUNT 2       /* _episode_blocking_count            MA-env:63  */
#define MAPF_CTR_GOALS_REACHED_TOTAL 3  /* _episode_goals_reached_total       MA-env:88  */
#define MAPF_CTR_DEADLOCK_EVENTS 4      /* _episode_deadlock_events           MA-env:64  */
#define MAPF_CTR_LIVELOCK_EVENTS 5
#define MAPF_CTR_DEADLOCK_STEPS 6
#define MAPF_CTR_LIVELOCK_STEPS 7
#define MAPF_CTR_LOCK_STATE_PREV 8      /* bit0 _deadlock_state_prev, bit1 _livelock_state_prev MA-env:68-69 */
#define MAPF_CTR_EPISODES_DONE 9        /* episodes finished by this env since create (auto-reset bookkeeping) */
#define MAPF_CTR_MAY_FINISH 10          /* engine-internal hint, nonzero = the episode may end in the next step (step limit
                                         * reached, or every agent within one cell of its goal): the background sampler
                                         * leaves such envs alone.  Written by every step; mapf_set_state forces it on. */

/* per-env lifetime sums over finished episodes, mapf_get_episode_stats() adds them up over the envs:
 * the quantities the reference's RLlib callbacks log at episode end (src/trainers/callbacks.py:135-345) */
#define MAPF_NUM_EPISODE_ACC 12
#define MAPF_ACC_EPISODES 0
#define MAPF_ACC_SUCCESSES 1        /* terminated and not truncated (SuccessRateCallback, finite mode) */
#define MAPF_ACC_GOALS_REACHED 2    /* sum of _episode_goals_reached_total at episode end */
#define MAPF_ACC_BLOCKING_COUNT 3
#define MAPF_ACC_DEADLOCK_COUNT 4   /* rising-edge events */
#define MAPF_ACC_LIVELOCK_COUNT 5
#define MAPF_ACC_DEADLOCK_STEPS 6
#define MAPF_ACC_LIVELOCK_STEPS 7
#define MAPF_ACC_COMPLETED_AGENTS 8 /* agents with _completed_once_arr set at episode end (completion_ratio numerator) */
#define MAPF_ACC_EPISODE_STEPS 9    /* sum of step_count at episode end */

typedef struct mapf_config {
    int32_t num_envs;           /* B >= 1 */
    int32_t height, width;      /* grid shape, each <= MAPF_MAX_DIM */
    int32_t num_agents;         /* N <= MAPF_MAX_AGENTS */
    int32_t sensor_range;       /* MA-env:40 */
    int32_t steps_per_episode;  /* MA-env:38 */
    uint32_t flags;             /* MAPF_FLAG_* */
    int32_t deadlock_window_steps; /* MA-env:56, clamped to >= 1 */
    int32_t livelock_window_steps; /* MA-env:57 */
    int32_t lock_nearby_manhattan; /* MA-env:58 */
    int32_t lock_min_neighbors;    /* MA-env:60 */
    double lock_progress_epsilon;  /* MA-env:59 */
    int32_t device;             /* HIP device ordinal */
    int32_t lanes_per_env;      /* 0 = auto (smallest power of two >= max(N,4)); else 4/8/16/32/64 */
} mapf_config;

typedef struct mapf_engine *mapf_handle;

/* host views for mapf_get_state / mapf_set_state; NULL members are skipped */
typedef struct mapf_state {
    int16_t *positions;      /* [B][N][2] (row, col)  _positions_arr */
    int16_t *goals;          /* [B][N][2]             _goals_arr */
    int16_t *starts;         /* [B][N][2]             _starts_arr */
    uint8_t *reached;        /* [B][N]                _reached_arr (sticky) */
    uint8_t *completed_once; /* [B][N]                _completed_once_arr */
    uint8_t *pressure_prev;  /* [B][N]  0/1           _blocking_pressure_prev_arr */
    int32_t *counters;       /* [B][MAPF_NUM_COUNTERS] */
    uint64_t *rng_words;     /* [B][6] numpy PCG64: state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger */
    uint64_t *lock_history;  /* [B][N][3] shift registers, bit k = flag k steps ago: moved, failed_move, goal_progress */
    int16_t *distance_ring;  /* [B][livelock_window][N], slot = history row index mod livelock_window */
} mapf_state;

uint32_t mapf_version(void);
/* flat observation length L for a config (MA-env:214-236): V*V + 2 [+1] [+1] [+5] */
int32_t mapf_obs_len(const mapf_config *cfg);

/* mapf_create also picks the build of the step kernel by the size of its grid: small-group configurations (at most 16
 * lanes per env) exist for two register budgets, and a grid of more than three waves per SIMD gets the 128-register
 * one.  MAPF_FLAG_FORCE_DENSE / MAPF_FLAG_FORCE_SPARSE override the choice (test knobs; results are
 * identical either way).  The library reads no environment variable except MAPF_JIT_CACHE_DIR (and XDG_CACHE_HOME /
 * HOME behind it) for the on-disk cache of MAPF_FLAG_JIT_SPECIALIZE. */
int mapf_create(const mapf_config *cfg /* host */, mapf_handle *out);
int mapf_destroy(mapf_handle h);
const char *mapf_last_error(mapf_handle h); /* h may be NULL: error of the last failed mapf_create */

/* grids: host uint8 [B][H][W] (shared == 0) or one [H][W] used by every env (shared != 0); 0 free, 1 obstacle.
 * Fails with MAPF_ERR_FEW_FREE when an env has fewer than 2N free cells (reference ctor, MA-env:270-275). */
int mapf_set_grids(mapf_handle h, const uint8_t *grids /* host */, int32_t shared);

/* numpy Generator(PCG64) state per env, host uint64 [B][6] (see mapf_state.rng_words).  Seed expansion
 * (SeedSequence) is the caller's job: np.random.default_rng(seed).bit_generator.state (MA-env:74-78). */
int mapf_set_rng_state(mapf_handle h, const uint64_t *rng_words /* host */);

/* deterministic mode (MA-env:124-132): fixed starts / goals, host int16 [B][N][2] each; also places agents. */
int mapf_set_fixed_starts_goals(mapf_handle h, const int16_t *starts /* host */, const int16_t *goals /* host */);

int mapf_get_state(mapf_handle h, mapf_state *out /* host views */);
int mapf_set_state(mapf_handle h, const mapf_state *in /* host views */);

/* reset (MA-env:440-472).  env_mask: device uint8 [B], nonzero = reset that env; NULL = all.
 * obs: device float32 [B][N][L], rows of reset envs are written; may be NULL (state only -- the
 * reference ctor's own generate_starts_goals() draw, MA-env:133-134, is mapf_reset with obs NULL). */
int mapf_reset(mapf_handle h, const uint8_t *env_mask /* device */, float *obs /* device */, void *stream);

/* one step of every env (MA-env:474-695).  All pointers device; any output may be NULL.
 *   actions     int8   [B][N]      0 NO_OP, 1 UP, 2 RIGHT, 3 DOWN, 4 LEFT (actions.py:1-5)
 *   obs         float32[B][N][L]   per-agent flat observation, reference layout (MA-env:306-328)
 *   rewards     float32[B][N]
 *   terminated  uint8  [B]         terminated["__all__"]   (per-agent flags equal it, MA-env:668-690)
 *   truncated   uint8  [B]         truncated["__all__"]
 *   info_all    float32[B][14]     MAPF_INFO_* columns
 *   info_agent  uint8  [B][N][2]   {blocking, goal_reached_step}  (MA-env:627-629)
 *   final_obs   float32[B][N][L]   only with auto_reset: terminal observation of envs that finished
 * auto_reset != 0: an env whose episode ended is reset() inside the same launch, exactly as the reference
 * harness does right after the step (scripts/benchmark_multi_agent_env.py:89-95); its `obs` rows then hold
 * the reset observation. */
int mapf_step(mapf_handle h, const int8_t *actions, float *obs, float *rewards, uint8_t *terminated, uint8_t *truncated,
              float *info_all, uint8_t *info_agent, float *final_obs, int32_t auto_reset, void *stream);

/* mapf_step for a SUBSET of the envs: env_mask (device uint8 [B]) selects the envs that step; every other env is not
 * touched at all -- state, generator, counters, episode statistics, error latch -- and its rows of the outputs are
 * left as they are.  This is what stepping ONE of several reference env objects means (RLlib's runners step their
 * sub-envs one by one, and skip the ones that wait for a reset: MultiAgentEnv.step per object, MA-env:474); the vector
 * adapters use it for rows stepped alone and for the next-step autoreset of the new-stack vector protocol. */
int mapf_step_masked(mapf_handle h, const int8_t *actions, const uint8_t *env_mask, float *obs, float *rewards,
                     uint8_t *terminated, uint8_t *truncated, float *info_all, uint8_t *info_agent, float *final_obs,
                     int32_t auto_reset, void *stream);

/* T consecutive steps in ONE launch (state stays in registers, obstacle rows in LDS): what the reference's
 * env-only benchmark loop does when the actions do not depend on the observations
 * (scripts/benchmark_multi_agent_env.py:85-95, mode "random"), or any scripted / pre-sampled action stream.
 *   actions  device int8 [T][B][N]
 *   obs      device float32; obs_mode 0: unused (may be NULL), 1: [B][N][L] observation after the last step,
 *            2: [T][B][N][L] every step
 *   rewards [T][B][N], terminated / truncated [T][B], info_all [T][B][14], info_agent [T][B][N][2]; any may be NULL
 * Finished envs are reset inside the loop (auto_reset semantics of mapf_step: the observation of a step that
 * ended an episode is the reset observation). */
int mapf_step_many(mapf_handle h, int32_t T, const int8_t *actions, float *obs, int32_t obs_mode, float *rewards,
                   uint8_t *terminated, uint8_t *truncated, float *info_all, uint8_t *info_agent, void *stream);

/* The same fused launch with a DEVICE-SIDE action source, for loops whose actions depend on the observations: the
 * masked-random policy of the reference's benchmark (scripts/benchmark_multi_agent_env.py:42-57, mode "masked": every
 * agent picks uniformly among the actions its action mask allows) evaluated in-kernel, step after step, on the
 * observation the previous step produced -- obs_in [B][N][L] (device) for the first step.  The generator is
 * counter-based (a hash of seed, env, agent and step index), so a run is reproducible and has no state; it is NOT
 * NumPy's stream: the actions taken are returned in actions_out [T][B][N] (device) and replaying them through
 * mapf_step / mapf_step_many gives the same transitions.  Needs MAPF_FLAG_ACTION_MASK; obs [T][B][N][L] is written for
 * every step (it is what the policy reads); the other outputs are as in mapf_step_many. */
int mapf_step_many_sampled(mapf_handle h, int32_t T, const float *obs_in, uint64_t seed, int8_t *actions_out, float *obs,
                           float *rewards, uint8_t *terminated, uint8_t *truncated, float *info_all, uint8_t *info_agent,
                           void *stream);

/* ---- single-agent (CTE) sibling env: reference src/environments/reference_model_single_agent.py ("SA-env") ----
 * One policy drives all N agents (gym.Env, MultiDiscrete([5]*N) action).  Create the handle with
 * MAPF_FLAG_SINGLE_AGENT (only MAPF_FLAG_DETERMINISTIC is meaningful besides it; sensor_range and the lock
 * settings are unused), then set grids / RNG / fixed tables as usual.  Flat observation = H*W cell codes
 * (0 free, 1 obstacle, 2+2i agent i, 3+2i goal i; SA-env:407-441) followed by the joint 5N action mask
 * (SA-env:443-495); mapf_obs_len() returns H*W + 5N.
 *   mapf_cte_configure  <- blocking_penalty / move_after_goal_penalty of the ctor   SA-env:92-93 (defaults -0.2, -0.05)
 *   mapf_cte_reset      <- reset()   SA-env:222-244 (+ generate_starts_goals :158-191); obs NULL = the ctor's draw :113-114
 *   mapf_cte_step       <- step()    SA-env:246-363:  reward double [B] (the reference's float64 sum, same order of
 *                          additions), terminated / truncated uint8 [B], info float32 [B][4] =
 *                          {blocking_count_step, goals_reached_step, goals_reached_total, blocking_count_total};
 *                          info["action_mask"] is the tail of the observation. */
int mapf_cte_configure(mapf_handle h, double blocking_penalty, double move_after_goal_penalty);
int mapf_cte_reset(mapf_handle h, const uint8_t *env_mask /* device */, float *obs /* device */, void *stream);
int mapf_cte_step(mapf_handle h, const int8_t *actions, float *obs, double *reward, uint8_t *terminated,
                  uint8_t *truncated, float *info, float *final_obs, int32_t auto_reset, void *stream);

/* T consecutive steps of the single-agent env in ONE launch for an action stream known up front (the CTE counterpart of
 * mapf_step_many; the reference's loop is `for t: obs, r, term, trunc, info = env.step(a[t]); if term or trunc:
 * env.reset()` around SA-env:246-363 / :222-244): positions stay in registers, the obstacle part of the full-grid
 * observation row is written once per launch, per step only the actions are read and the outputs written.
 *   actions device int8 [T][B][N];  obs_mode 0: no observation, 1: [B][H*W+5N] after the last step, 2: [T][B][H*W+5N];
 *   reward [T][B] float64, terminated / truncated [T][B], info [T][B][4]; any may be NULL.
 * Finished envs are reset inside the loop; the observation of a step that ended an episode is the reset observation. */
int mapf_cte_step_many(mapf_handle h, int32_t T, const int8_t *actions, float *obs, int32_t obs_mode, double *reward,
                       uint8_t *terminated, uint8_t *truncated, float *info, void *stream);

/* The same step for callers that pay per ARGUMENT (ctypes from a Python rollout loop: ~0.4 us each): the output buffers of
 * mapf_step are bound to the handle once, mapf_step_bound(h, actions, auto_reset, stream) then is mapf_step with those
 * pointers and final_obs = NULL.  The buffers stay the caller's; binding again replaces them. */
int mapf_bind_outputs(mapf_handle h, float *obs, float *rewards, uint8_t *terminated, uint8_t *truncated, float *info_all,
                      uint8_t *info_agent);
int mapf_step_bound(mapf_handle h, const int8_t *actions, int32_t auto_reset, void *stream);

/* observation of every agent from the CURRENT state, nothing is modified: what the reference returns when
 * get_obs / get_action_mask / _flatten_observation (MA-env:707-773, :306-328) are called outside step().
 * obs: device float32 [B][N][L]. */
int mapf_observe(mapf_handle h, float *obs /* device */, void *stream);

/* one goal respawn of one agent OUTSIDE step(): `_assign_new_goal(agent_idx)` (MA-env:284-304) -- clears the agent's goal,
 * counts the k free cells (row-major `_free_positions`) that hold no agent and no goal, draws rng.integers(k) from the
 * env's stream (nothing is drawn when k == 1), stores the r-th such cell as the new goal and returns it in
 * new_goal (host int16 [2]: row, col).  Runs on `stream` and waits for it.  MAPF_ERR_NO_RESPAWN when k == 0 (the
 * reference's RuntimeError, :296-298; also latched for mapf_poll_error).  Inside mapf_step the respawns of lifelong
 * mode run in the step kernel; this entry point is the helper by itself, as the reference's tests call it. */
int mapf_assign_new_goal(mapf_handle h, int32_t env, int32_t agent, int16_t *new_goal /* host */, void *stream);

/* sums of the per-env episode accumulators over all envs of the handle: host int64 out[MAPF_NUM_EPISODE_ACC];
 * reset != 0 clears them afterwards.  Synchronizes the device.  (Off the hot path; for a multi-GPU job add the
 * vectors of the ranks, e.g. one RCCL all-reduce of this 96-byte buffer per reporting interval.) */
int mapf_get_episode_stats(mapf_handle h, int64_t *out /* host */, int32_t reset);

/* the same sums without a host round trip: one small launch on `stream` adds the accumulators up into DEVICE memory
 * (int64 out[MAPF_NUM_EPISODE_ACC]); nothing is synchronized and nothing is cleared.  For a training loop that logs the
 * callbacks' metrics (src/trainers/callbacks.py:236-345) from a tensor every so often, and for bench.py, which must not
 * leave the GPU idle between its warm-up launches and the timed region. */
int mapf_episode_stats_async(mapf_handle h, int64_t *out /* device */, void *stream);

/* read (and clear) the device error record; synchronizes `stream`.  Returns MAPF_OK when no env has
 * failed, else the code of the first failure with its env / agent / offending value. */
int mapf_poll_error(mapf_handle h, void *stream, int32_t *env, int32_t *agent, int32_t *value);

/* diagnostic builds only (-DMAPF_STAMPS; the shipped library returns MAPF_ERR_STATE): copies the per-wave
 * s_memtime stamps of the last mapf_step to host uint64 out[workgroups][32]; returns the number of words. */
int mapf_debug_stamps(mapf_handle h, uint64_t *out /* host */, int32_t max_words);

/* diagnostic: the placement slots [B][N] (mapf_kernels.inl: kSlotInvalid / kSlotStaged*), the staging buffer of the
 * background draw [B][4N+4] and the visible stream states [B][6] as they are on the device (host outputs, any may be
 * NULL).  Synchronizes the device. */
int mapf_debug_slots(mapf_handle h, uint32_t *slots /* host */, uint32_t *stage /* host */, uint64_t *vis /* host */);

/* 1 = this handle steps with a kernel compiled for its configuration at mapf_create (MAPF_FLAG_JIT_SPECIALIZE), 0 = not;
 * *why (may be NULL) gets a static or handle-owned string: the reason when 0, the compile time when 1. */
int mapf_jit_status(mapf_handle h, const char **why);

/* dynamic-LDS bytes and grid size the step kernel is launched with (for DESIGN.md / profiling notes).
 * Returns >= 0: the id of the compile-time specialisation of the step kernel in use (0 = runtime-config kernel). */
int mapf_launch_info(mapf_handle h, int32_t *blocks, int32_t *threads, int32_t *lds_bytes, int32_t *lanes_per_env);
/* the same for the fused launches of a MAPF_FLAG_SINGLE_AGENT handle (mapf_cte_step_many with T > 1), which pick their
 * own group width; MAPF_ERR_STATE for other handles. */
int mapf_cte_many_launch_info(mapf_handle h, int32_t *blocks, int32_t *threads, int32_t *lds_bytes, int32_t *lanes_per_env);

#ifdef __cplusplus
}
#endif
#endif
