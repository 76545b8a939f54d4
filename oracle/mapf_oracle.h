/*
 * mapf_oracle.h -- CPU restatement of the reference MultiAgentEnv step()/reset() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported CPU baseline.  The product path (dl_reference_models_amd/) never
 * links, imports or falls back to it.
 *
 * What it restates (all line numbers: /root/reference/src/environments/reference_model_multi_agent.py,
 * "MA-env" below): ReferenceModel.__init__ state block :82-120, generate_starts_goals :267-282,
 * _assign_new_goal :284-304, _flatten_observation :306-328, _get_goal_delta :330-335,
 * lock tracking :360-438, reset :440-472, step :474-695, get_obs :707-747, get_action_mask :749-773,
 * plus the NumPy Generator(PCG64) draws those functions make (numpy is a third-party dependency,
 * unpinned in requirements.txt:4; algorithm restated from numpy 2.2.6:
 * numpy/random/src/pcg64/pcg64.h, src/distributions/distributions.c (buffered_bounded_lemire_uint32),
 * _generator.pyx Generator.choice (Floyd branch) and _shuffle_int).
 *
 * Parity pinning: tests/test_oracle_golden.py checks this restatement against
 *   - both SHA-256 digests and episode summaries of the reference's own
 *     tests/test_reference_model_multi_agent_parity.py:12-24,
 *   - golden traces recorded from the unmodified reference by oracle/gen_golden.py
 *     (the .npz files under tests/golden), and the reference's micro-case tests (lock metrics, blocking
 *     pressure, lifelong, tests/get_obs.py known-answer arrays).
 *
 * Structure: one env at a time, agents strictly in index order against LIVE owner maps,
 * observation taken inside the move loop -- i.e. the reference's own order of operations.
 * (The GPU engine uses a different, parallel formulation; agreement between the two is the
 * point of the parity tests.)
 */
#ifndef MAPF_ORACLE_H
#define MAPF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MO_FLAG_NORMALIZE_GOAL_DELTA 1u
#define MO_FLAG_GOAL_DISTANCE 2u
#define MO_FLAG_ACTION_MASK 4u
#define MO_FLAG_BLOCKING_PRESSURE 8u
#define MO_FLAG_LIFELONG 16u
#define MO_FLAG_LOCK_METRICS 32u
#define MO_FLAG_DETERMINISTIC 64u /* reset() re-places agents on fixed starts, no RNG (MA-env:452-455) */

#define MO_INFO_ALL 14

/* error codes */
#define MO_OK 0
#define MO_ERR_BAD_ACTION -1   /* ValueError MA-env:504-506 (state partially mutated, like the reference) */
#define MO_ERR_FEW_FREE -2     /* ValueError MA-env:270-275 */
#define MO_ERR_NO_RESPAWN -3   /* RuntimeError MA-env:296-298 */
#define MO_ERR_CONFIG -4

typedef struct mo_config {
    int32_t height, width;
    int32_t num_agents;
    int32_t sensor_range;
    int32_t steps_per_episode;
    uint32_t flags;
    int32_t deadlock_window_steps;
    int32_t livelock_window_steps;
    int32_t lock_nearby_manhattan;
    int32_t lock_min_neighbors;
    double lock_progress_epsilon;
} mo_config;

typedef struct mo_env mo_env;

/* observation length L for a config (MA-env:214-236) */
int mo_obs_len(const mo_config *cfg);

/* one env; grid is uint8[H][W] row-major, 0 = free, 1 = obstacle */
mo_env *mo_create(const mo_config *cfg, const uint8_t *grid);
void mo_destroy(mo_env *e);

/* PCG64 state exactly as numpy's bit_generator.state dict holds it */
void mo_set_rng(mo_env *e, uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi, uint64_t inc_lo,
                int32_t has_uint32, uint32_t uinteger);
void mo_get_rng(const mo_env *e, uint64_t out[6]);

/* MA-env:267-282; returns MO_OK or MO_ERR_FEW_FREE */
int mo_generate_starts_goals(mo_env *e);
/* deterministic mode: install fixed starts/goals (MA-env:124-132): int16[N][2] each */
void mo_set_fixed_starts_goals(mo_env *e, const int16_t *starts, const int16_t *goals);

/* MA-env:440-472; obs float32[N][L] (may be NULL) */
int mo_reset(mo_env *e, float *obs);

/* MA-env:474-695.  actions int32[N].  Outputs (any may be NULL):
 *   obs float32[N][L], rewards float32[N], done uint8[2] = {terminated["__all__"], truncated["__all__"]},
 *   info_all float32[14], info_agent uint8[N][2] = {blocking, goal_reached_step}. */
int mo_step(mo_env *e, const int32_t *actions, float *obs, float *rewards, uint8_t *done, float *info_all,
            uint8_t *info_agent);

/* helpers mirroring get_obs :707 / get_action_mask :749 for known-answer tests */
void mo_get_obs(const mo_env *e, int agent, uint8_t *local /* [V][V] */);
void mo_get_action_mask(const mo_env *e, const uint8_t *local, int8_t mask[5]);
/* MA-env:284-304 exposed for RNG/ordering tests; returns MO_OK / MO_ERR_NO_RESPAWN */
int mo_assign_new_goal(mo_env *e, int agent);

/* ---- state access (what the reference's tests poke, tests/...invariants.py:28-38) ---- */
typedef struct mo_state_view {
    int16_t *positions;       /* [N][2] */
    int16_t *goals;           /* [N][2] */
    int16_t *starts;          /* [N][2] */
    uint8_t *reached;         /* [N] */
    uint8_t *completed_once;  /* [N] */
    float *pressure_prev;     /* [N] */
    int16_t *occupancy_owner; /* [H][W] */
    int16_t *goal_owner;      /* [H][W] */
    uint8_t *hist_goal_progress; /* [Hs][N] */
    uint8_t *hist_moved;
    uint8_t *hist_failed_move;
    int16_t *hist_distance;   /* [Hs][N] */
    int32_t *hist_count, *hist_head;
    int32_t *step_count;
    double *episode_blocking_count, *episode_goals_reached_total;
    double *episode_deadlock_events, *episode_livelock_events, *episode_deadlock_steps, *episode_livelock_steps;
    uint8_t *deadlock_state_prev, *livelock_state_prev;
    int32_t n_free;
    const int16_t *free_positions; /* [F][2] */
    int32_t hist_size;
} mo_state_view;
void mo_view(mo_env *e, mo_state_view *out);
void mo_rebuild_owner_maps(mo_env *e);   /* MA-env:200-212 */
void mo_reset_lock_tracking(mo_env *e);  /* MA-env:360-372 */

/* ---- batch driver (parity at size + cpu_baseline): B independent envs, scalar loop ---- */
typedef struct mo_batch mo_batch;
mo_batch *mo_batch_create(const mo_config *cfg, int32_t num_envs, const uint8_t *grids /* [B][H][W] */);
void mo_batch_destroy(mo_batch *b);
mo_env *mo_batch_env(mo_batch *b, int32_t i);
/* step every env; when auto_reset != 0 a finished env is reset() right after its step, exactly like
 * the reference harness loop scripts/benchmark_multi_agent_env.py:89-95: obs then holds the reset
 * observation and final_obs (optional) the terminal one.  actions int8[B][N].  Returns first error. */
int mo_batch_step(mo_batch *b, const int8_t *actions, int auto_reset, float *obs, float *rewards, uint8_t *terminated,
                  uint8_t *truncated, float *info_all, uint8_t *info_agent, float *final_obs, int32_t *err_env);
int mo_batch_reset(mo_batch *b, float *obs);

/* raw RNG known-answer hooks */
uint64_t mo_rng_bounded(mo_env *e, uint64_t rng_inclusive);
void mo_rng_choice_noreplace(mo_env *e, int64_t pop, int64_t size, int64_t *out);

/* ---- single-agent (CTE) sibling env: reference src/environments/reference_model_single_agent.py ---- */
typedef struct moc_env moc_env;
moc_env *moc_create(int H, int W, int N, int steps_per_episode, int deterministic, double blocking_penalty,
                    double move_after_goal_penalty, const uint8_t *grid);
void moc_destroy(moc_env *e);
int moc_obs_len(const moc_env *e);
void moc_set_rng(moc_env *e, uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi, uint64_t inc_lo, int32_t has_uint32,
                 uint32_t uinteger);
void moc_get_rng(const moc_env *e, uint64_t out[6]);
void moc_set_fixed_starts_goals(moc_env *e, const int32_t *starts, const int32_t *goals);
void moc_generate_starts_goals(moc_env *e);
void moc_reset(moc_env *e, float *obs);
int moc_step(moc_env *e, const int32_t *action, float *obs, double *reward, uint8_t *done, float *info);
/* timing helper (bench.py cpu_baseline of the single-agent env): `steps` steps of B envs with reset-on-done, one call */
long moc_run(moc_env **envs, int B, const int8_t *actions /* [P][B][N] */, int P, int steps, float *obs_scratch);
void moc_view(moc_env *e, int32_t **positions, int32_t **goals, int32_t **starts, uint8_t **reached_once,
              int32_t **step_count, double **blocking_count);

#ifdef __cplusplus
}
#endif
#endif
