/*
 * mapf_oracle.c -- CPU restatement of the reference env hot path.  TEST INFRASTRUCTURE ONLY
 * (see mapf_oracle.h for the scope statement and what pins it).
 *
 * "MA-env:N" = /root/reference/src/environments/reference_model_multi_agent.py line N.
 */
#include "mapf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* cell codes, MA-env:26-32 */
enum { EMPTY_CELL = 0, OBSTACLE_CELL = 1, OTHER_AGENT_CELL = 2, OWN_GOAL_CELL = 3, OTHER_GOAL_CELL = 4 };
#define UNASSIGNED_OWNER (-1)

/* MA-env:104-113 / actions.py:1-5: NO_OP, UP, RIGHT, DOWN, LEFT as (d_row, d_col) */
static const int ACTION_DELTAS[5][2] = {{0, 0}, {-1, 0}, {0, 1}, {1, 0}, {0, -1}};

/* numpy Generator(PCG64) state */
typedef struct mo_rng {
    u128 rng_state, rng_inc;
    int32_t has_uint32;
    uint32_t uinteger;
} mo_rng;

struct mo_env {
    mo_config cfg;
    int H, W, N, V, L, Hs;
    uint8_t *grid;            /* [H][W] */
    int16_t *free_positions;  /* [F][2], row-major argwhere(grid==0), MA-env:82 */
    int n_free;
    int16_t *starts, *positions, *goals; /* [N][2] */
    uint8_t *reached, *completed_once;   /* MA-env:86-87 */
    float *pressure_prev;                /* MA-env:89 */
    int16_t *occupancy_owner, *goal_owner; /* MA-env:102-103 */
    uint8_t *hist_goal_progress, *hist_moved, *hist_failed_move; /* MA-env:115-117 */
    int16_t *hist_distance;                                       /* MA-env:118 */
    int32_t hist_count, hist_head;
    int32_t step_count;
    double episode_blocking_count, episode_goals_reached_total;
    double episode_deadlock_events, episode_livelock_events, episode_deadlock_steps, episode_livelock_steps;
    uint8_t deadlock_state_prev, livelock_state_prev;
    /* scratch, MA-env:90-101 */
    int16_t *prev_positions, *intended_next;
    uint8_t *reached_goal;
    float *goal_reached_step_flags, *blocking_flags;
    int8_t *actions_taken;
    uint8_t *moved_flags, *failed_move_flags, *goal_progress_flags, *prev_on_goal, *current_on_goal;
    int16_t *distance_to_goal;
    int *participants; /* [N][N+1]: count + members */
    mo_rng rng; /* numpy PCG64 */
};

/* ------------------------------------------------------------------------------------------
 * NumPy Generator(PCG64) restatement (numpy 2.2.6)
 * ------------------------------------------------------------------------------------------ */
#define PCG_MULT_HI 0x2360ED051FC65DA4ULL
#define PCG_MULT_LO 0x4385DF649FCCF645ULL

static uint64_t pcg64_next64(mo_rng *e) {
    /* pcg64.h: pcg_setseq_128_step_r then pcg_output_xsl_rr_128_64 on the NEW state */
    const u128 mult = ((u128)PCG_MULT_HI << 64) | PCG_MULT_LO;
    e->rng_state = e->rng_state * mult + e->rng_inc;
    uint64_t hi = (uint64_t)(e->rng_state >> 64), lo = (uint64_t)e->rng_state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(e->rng_state >> 122);
    return (x >> rot) | (x << ((-rot) & 63));
}

static uint32_t pcg64_next32(mo_rng *e) {
    /* pcg64.h pcg64_next32: hand out the low half first, cache the high half */
    if (e->has_uint32) {
        e->has_uint32 = 0;
        return e->uinteger;
    }
    uint64_t next = pcg64_next64(e);
    e->has_uint32 = 1;
    e->uinteger = (uint32_t)(next >> 32);
    return (uint32_t)(next & 0xffffffffu);
}

/* distributions.c random_bounded_uint64(off=0, rng, mask, use_masked=false), 32-bit branch only:
 * every call site on this path has rng < 2^32 - 1 (rng <= free cells <= 4096). */
static uint64_t rng_bounded(mo_rng *e, uint64_t rng) {
    if (rng == 0) return 0; /* no draw */
    if (rng == 0xFFFFFFFFu) return pcg64_next32(e);
    /* buffered_bounded_lemire_uint32 */
    const uint32_t rng_excl = (uint32_t)rng + 1u;
    uint64_t m = (uint64_t)pcg64_next32(e) * rng_excl;
    uint32_t leftover = (uint32_t)(m & 0xFFFFFFFFu);
    if (leftover < rng_excl) {
        const uint32_t threshold = (uint32_t)((0xFFFFFFFFu - (uint32_t)rng) % rng_excl);
        while (leftover < threshold) {
            m = (uint64_t)pcg64_next32(e) * rng_excl;
            leftover = (uint32_t)(m & 0xFFFFFFFFu);
        }
    }
    return m >> 32;
}

/* _generator.pyx Generator.choice(pop, size, replace=False), p=None, shuffle=True: Floyd branch.
 * (The tail-shuffle branch needs pop > 10000, impossible for grids up to 64x64 = 4096 cells.) */
static void rng_choice_noreplace(mo_rng *e, int64_t pop, int64_t size, int64_t *out) {
    uint64_t set_size = (uint64_t)(1.2 * (double)size);
    uint64_t mask = set_size; /* _gen_mask: smear to 2^p - 1 */
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    mask |= mask >> 32;
    set_size = mask + 1;
    uint64_t *hash_set = (uint64_t *)malloc(sizeof(uint64_t) * set_size);
    for (uint64_t i = 0; i < set_size; i++) hash_set[i] = (uint64_t)-1;
    for (int64_t j = pop - size; j < pop; j++) {
        uint64_t val = rng_bounded(e, (uint64_t)j);
        uint64_t loc = val & mask;
        while (hash_set[loc] != (uint64_t)-1 && hash_set[loc] != val) loc = (loc + 1) & mask;
        if (hash_set[loc] == (uint64_t)-1) {
            hash_set[loc] = val;
            out[j - pop + size] = (int64_t)val;
        } else {
            loc = (uint64_t)j & mask;
            while (hash_set[loc] != (uint64_t)-1) loc = (loc + 1) & mask;
            hash_set[loc] = (uint64_t)j;
            out[j - pop + size] = j;
        }
    }
    /* _shuffle_int(bitgen, n=size, first=1, data) */
    for (int64_t i = size - 1; i >= 1; i--) {
        int64_t j = (int64_t)rng_bounded(e, (uint64_t)i);
        int64_t t = out[j];
        out[j] = out[i];
        out[i] = t;
    }
    free(hash_set);
}

uint64_t mo_rng_bounded(mo_env *e, uint64_t rng_inclusive) { return rng_bounded(&e->rng, rng_inclusive); }
void mo_rng_choice_noreplace(mo_env *e, int64_t pop, int64_t size, int64_t *out) {
    rng_choice_noreplace(&e->rng, pop, size, out);
}

static void rng_set(mo_rng *g, uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi, uint64_t inc_lo, int32_t has_uint32,
                    uint32_t uinteger) {
    g->rng_state = ((u128)state_hi << 64) | state_lo;
    g->rng_inc = ((u128)inc_hi << 64) | inc_lo;
    g->has_uint32 = has_uint32;
    g->uinteger = uinteger;
}

static void rng_get(const mo_rng *g, uint64_t out[6]) {
    out[0] = (uint64_t)(g->rng_state >> 64);
    out[1] = (uint64_t)g->rng_state;
    out[2] = (uint64_t)(g->rng_inc >> 64);
    out[3] = (uint64_t)g->rng_inc;
    out[4] = (uint64_t)g->has_uint32;
    out[5] = (uint64_t)g->uinteger;
}

void mo_set_rng(mo_env *e, uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi, uint64_t inc_lo, int32_t has_uint32,
                uint32_t uinteger) {
    rng_set(&e->rng, state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger);
}

void mo_get_rng(const mo_env *e, uint64_t out[6]) { rng_get(&e->rng, out); }

/* ------------------------------------------------------------------------------------------
 * construction, MA-env:34-184
 * ------------------------------------------------------------------------------------------ */
int mo_obs_len(const mo_config *cfg) {
    /* _build_obs_component_spaces MA-env:214-236 */
    int V = 2 * cfg->sensor_range + 1;
    int L = V * V + 2;
    if (cfg->flags & MO_FLAG_GOAL_DISTANCE) L += 1;
    if (cfg->flags & MO_FLAG_BLOCKING_PRESSURE) L += 1;
    if (cfg->flags & MO_FLAG_ACTION_MASK) L += 5;
    return L;
}

#define OWN(e, map, r, c) ((e)->map[(r) * (e)->W + (c)])

void mo_rebuild_owner_maps(mo_env *e) {
    /* _rebuild_goal_owner MA-env:207-212, _rebuild_occupancy_owner MA-env:200-205 */
    for (int i = 0; i < e->H * e->W; i++) e->goal_owner[i] = UNASSIGNED_OWNER;
    for (int idx = 0; idx < e->N; idx++) OWN(e, goal_owner, e->goals[2 * idx], e->goals[2 * idx + 1]) = (int16_t)idx;
    for (int i = 0; i < e->H * e->W; i++) e->occupancy_owner[i] = UNASSIGNED_OWNER;
    for (int idx = 0; idx < e->N; idx++)
        OWN(e, occupancy_owner, e->positions[2 * idx], e->positions[2 * idx + 1]) = (int16_t)idx;
}

void mo_reset_lock_tracking(mo_env *e) {
    /* MA-env:360-372 */
    size_t n = (size_t)e->Hs * e->N;
    memset(e->hist_goal_progress, 0, n);
    memset(e->hist_moved, 0, n);
    memset(e->hist_failed_move, 0, n);
    memset(e->hist_distance, 0, n * sizeof(int16_t));
    e->hist_count = 0;
    e->hist_head = 0;
    e->episode_deadlock_events = 0.0;
    e->episode_livelock_events = 0.0;
    e->episode_deadlock_steps = 0.0;
    e->episode_livelock_steps = 0.0;
    e->deadlock_state_prev = 0;
    e->livelock_state_prev = 0;
}

mo_env *mo_create(const mo_config *cfg, const uint8_t *grid) {
    if (cfg->height < 1 || cfg->width < 1 || cfg->num_agents < 1 || cfg->num_agents > 4096 || cfg->sensor_range < 0 ||
        cfg->sensor_range > 15)
        return NULL;
    mo_env *e = (mo_env *)calloc(1, sizeof(mo_env));
    e->cfg = *cfg;
    /* clamps MA-env:56-60 */
    if (e->cfg.deadlock_window_steps < 1) e->cfg.deadlock_window_steps = 1;
    if (e->cfg.livelock_window_steps < 1) e->cfg.livelock_window_steps = 1;
    if (e->cfg.lock_nearby_manhattan < 1) e->cfg.lock_nearby_manhattan = 1;
    if (e->cfg.lock_min_neighbors < 1) e->cfg.lock_min_neighbors = 1;
    e->H = cfg->height;
    e->W = cfg->width;
    e->N = cfg->num_agents;
    e->V = 2 * cfg->sensor_range + 1;
    e->L = mo_obs_len(cfg);
    e->Hs = e->cfg.deadlock_window_steps > e->cfg.livelock_window_steps ? e->cfg.deadlock_window_steps
                                                                        : e->cfg.livelock_window_steps; /* :114 */
    int HW = e->H * e->W, N = e->N;
    e->grid = (uint8_t *)malloc(HW);
    memcpy(e->grid, grid, HW);
    e->free_positions = (int16_t *)malloc(sizeof(int16_t) * 2 * HW);
    e->n_free = 0;
    for (int r = 0; r < e->H; r++)
        for (int c = 0; c < e->W; c++)
            if (grid[r * e->W + c] == EMPTY_CELL) {
                e->free_positions[2 * e->n_free] = (int16_t)r;
                e->free_positions[2 * e->n_free + 1] = (int16_t)c;
                e->n_free++;
            }
    e->starts = (int16_t *)calloc(2 * N, sizeof(int16_t));
    e->positions = (int16_t *)calloc(2 * N, sizeof(int16_t));
    e->goals = (int16_t *)calloc(2 * N, sizeof(int16_t));
    e->reached = (uint8_t *)calloc(N, 1);
    e->completed_once = (uint8_t *)calloc(N, 1);
    e->pressure_prev = (float *)calloc(N, sizeof(float));
    e->occupancy_owner = (int16_t *)malloc(sizeof(int16_t) * HW);
    e->goal_owner = (int16_t *)malloc(sizeof(int16_t) * HW);
    e->hist_goal_progress = (uint8_t *)calloc((size_t)e->Hs * N, 1);
    e->hist_moved = (uint8_t *)calloc((size_t)e->Hs * N, 1);
    e->hist_failed_move = (uint8_t *)calloc((size_t)e->Hs * N, 1);
    e->hist_distance = (int16_t *)calloc((size_t)e->Hs * N, sizeof(int16_t));
    e->prev_positions = (int16_t *)calloc(2 * N, sizeof(int16_t));
    e->intended_next = (int16_t *)calloc(2 * N, sizeof(int16_t));
    e->reached_goal = (uint8_t *)calloc(N, 1);
    e->goal_reached_step_flags = (float *)calloc(N, sizeof(float));
    e->blocking_flags = (float *)calloc(N, sizeof(float));
    e->actions_taken = (int8_t *)calloc(N, 1);
    e->moved_flags = (uint8_t *)calloc(N, 1);
    e->failed_move_flags = (uint8_t *)calloc(N, 1);
    e->goal_progress_flags = (uint8_t *)calloc(N, 1);
    e->prev_on_goal = (uint8_t *)calloc(N, 1);
    e->current_on_goal = (uint8_t *)calloc(N, 1);
    e->distance_to_goal = (int16_t *)calloc(N, sizeof(int16_t));
    e->participants = (int *)calloc((size_t)N * (N + 1), sizeof(int));
    for (int i = 0; i < HW; i++) e->occupancy_owner[i] = e->goal_owner[i] = UNASSIGNED_OWNER;
    return e;
}

void mo_destroy(mo_env *e) {
    if (!e) return;
    free(e->grid);
    free(e->free_positions);
    free(e->starts);
    free(e->positions);
    free(e->goals);
    free(e->reached);
    free(e->completed_once);
    free(e->pressure_prev);
    free(e->occupancy_owner);
    free(e->goal_owner);
    free(e->hist_goal_progress);
    free(e->hist_moved);
    free(e->hist_failed_move);
    free(e->hist_distance);
    free(e->prev_positions);
    free(e->intended_next);
    free(e->reached_goal);
    free(e->goal_reached_step_flags);
    free(e->blocking_flags);
    free(e->actions_taken);
    free(e->moved_flags);
    free(e->failed_move_flags);
    free(e->goal_progress_flags);
    free(e->prev_on_goal);
    free(e->current_on_goal);
    free(e->distance_to_goal);
    free(e->participants);
    free(e);
}

void mo_view(mo_env *e, mo_state_view *v) {
    v->positions = e->positions;
    v->goals = e->goals;
    v->starts = e->starts;
    v->reached = e->reached;
    v->completed_once = e->completed_once;
    v->pressure_prev = e->pressure_prev;
    v->occupancy_owner = e->occupancy_owner;
    v->goal_owner = e->goal_owner;
    v->hist_goal_progress = e->hist_goal_progress;
    v->hist_moved = e->hist_moved;
    v->hist_failed_move = e->hist_failed_move;
    v->hist_distance = e->hist_distance;
    v->hist_count = &e->hist_count;
    v->hist_head = &e->hist_head;
    v->step_count = &e->step_count;
    v->episode_blocking_count = &e->episode_blocking_count;
    v->episode_goals_reached_total = &e->episode_goals_reached_total;
    v->episode_deadlock_events = &e->episode_deadlock_events;
    v->episode_livelock_events = &e->episode_livelock_events;
    v->episode_deadlock_steps = &e->episode_deadlock_steps;
    v->episode_livelock_steps = &e->episode_livelock_steps;
    v->deadlock_state_prev = &e->deadlock_state_prev;
    v->livelock_state_prev = &e->livelock_state_prev;
    v->n_free = e->n_free;
    v->free_positions = e->free_positions;
    v->hist_size = e->Hs;
}

/* ------------------------------------------------------------------------------------------
 * generate_starts_goals MA-env:267-282
 * ------------------------------------------------------------------------------------------ */
int mo_generate_starts_goals(mo_env *e) {
    int N = e->N, required = 2 * N;
    if (e->n_free < required) return MO_ERR_FEW_FREE; /* :270-275 */
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * required);
    rng_choice_noreplace(&e->rng, e->n_free, required, idx); /* :277 */
    for (int i = 0; i < N; i++) {
        e->starts[2 * i] = e->free_positions[2 * idx[i]];
        e->starts[2 * i + 1] = e->free_positions[2 * idx[i] + 1];
        e->goals[2 * i] = e->free_positions[2 * idx[N + i]];
        e->goals[2 * i + 1] = e->free_positions[2 * idx[N + i] + 1];
    }
    memcpy(e->positions, e->starts, sizeof(int16_t) * 2 * N); /* :279 */
    free(idx);
    mo_rebuild_owner_maps(e); /* :281-282 */
    return MO_OK;
}

void mo_set_fixed_starts_goals(mo_env *e, const int16_t *starts, const int16_t *goals) {
    /* deterministic ctor branch MA-env:124-132 */
    memcpy(e->starts, starts, sizeof(int16_t) * 2 * e->N);
    memcpy(e->goals, goals, sizeof(int16_t) * 2 * e->N);
    memcpy(e->positions, e->starts, sizeof(int16_t) * 2 * e->N);
    mo_rebuild_owner_maps(e);
}

/* ------------------------------------------------------------------------------------------
 * _assign_new_goal MA-env:284-304
 * ------------------------------------------------------------------------------------------ */
int mo_assign_new_goal(mo_env *e, int a) {
    OWN(e, goal_owner, e->goals[2 * a], e->goals[2 * a + 1]) = UNASSIGNED_OWNER; /* :286-288 */
    /* candidates: free cells (row-major) with no occupant and no goal, :290-295 */
    int k = 0;
    for (int f = 0; f < e->n_free; f++) {
        int r = e->free_positions[2 * f], c = e->free_positions[2 * f + 1];
        if (OWN(e, occupancy_owner, r, c) == UNASSIGNED_OWNER && OWN(e, goal_owner, r, c) == UNASSIGNED_OWNER) k++;
    }
    if (k == 0) return MO_ERR_NO_RESPAWN; /* :296-298 */
    int64_t sel = (int64_t)rng_bounded(&e->rng, (uint64_t)(k - 1)); /* rng.integers(k), :300 */
    int seen = 0;
    for (int f = 0; f < e->n_free; f++) {
        int r = e->free_positions[2 * f], c = e->free_positions[2 * f + 1];
        if (OWN(e, occupancy_owner, r, c) == UNASSIGNED_OWNER && OWN(e, goal_owner, r, c) == UNASSIGNED_OWNER) {
            if (seen == sel) {
                e->goals[2 * a] = (int16_t)r; /* :301-303 */
                e->goals[2 * a + 1] = (int16_t)c;
                OWN(e, goal_owner, r, c) = (int16_t)a;
                return MO_OK;
            }
            seen++;
        }
    }
    return MO_ERR_NO_RESPAWN; /* unreachable */
}

/* ------------------------------------------------------------------------------------------
 * get_obs MA-env:707-747, get_action_mask MA-env:749-773, _flatten_observation MA-env:306-328
 * ------------------------------------------------------------------------------------------ */
void mo_get_obs(const mo_env *e, int a, uint8_t *local) {
    int V = e->V, sr = e->cfg.sensor_range;
    for (int i = 0; i < V * V; i++) local[i] = OBSTACLE_CELL; /* :711-715 */
    int base_r = e->positions[2 * a] - sr, base_c = e->positions[2 * a + 1] - sr;
    for (int i = 0; i < V; i++) {
        int r = base_r + i;
        if (r < 0 || r >= e->H) continue;
        for (int j = 0; j < V; j++) {
            int c = base_c + j;
            if (c < 0 || c >= e->W) continue;
            if (e->grid[r * e->W + c] == OBSTACLE_CELL) {
                local[i * V + j] = OBSTACLE_CELL;
                continue;
            }
            int occ = OWN(e, occupancy_owner, r, c);
            if (occ != UNASSIGNED_OWNER && occ != a) { /* :734-737 */
                local[i * V + j] = OTHER_AGENT_CELL;
                continue;
            }
            int g = OWN(e, goal_owner, r, c);
            if (g == a)
                local[i * V + j] = OWN_GOAL_CELL;
            else if (g != UNASSIGNED_OWNER)
                local[i * V + j] = OTHER_GOAL_CELL;
            else
                local[i * V + j] = EMPTY_CELL;
        }
    }
}

static int traversable(uint8_t v) { return v == EMPTY_CELL || v == OWN_GOAL_CELL || v == OTHER_GOAL_CELL; }

void mo_get_action_mask(const mo_env *e, const uint8_t *local, int8_t mask[5]) {
    int V = e->V, x = e->cfg.sensor_range, y = e->cfg.sensor_range;
    mask[0] = 1; /* :756 */
    mask[1] = mask[2] = mask[3] = mask[4] = 0;
    if (x > 0 && traversable(local[(x - 1) * V + y])) mask[1] = 1;     /* UP    :761 */
    if (y < V - 1 && traversable(local[x * V + y + 1])) mask[2] = 1;   /* RIGHT :764 */
    if (x < V - 1 && traversable(local[(x + 1) * V + y])) mask[3] = 1; /* DOWN  :767 */
    if (y > 0 && traversable(local[x * V + y - 1])) mask[4] = 1;       /* LEFT  :770 */
}

static void observe(const mo_env *e, int a, float *out) {
    /* get_obs + get_action_mask + _flatten_observation for one agent */
    uint8_t local[31 * 31];
    int V = e->V;
    mo_get_obs(e, a, local);
    float *p = out;
    for (int i = 0; i < V * V; i++) *p++ = (float)local[i];
    /* _get_goal_delta :330-335: int16 difference -> float32, then float32 / float32 denominator :152-155 */
    float gd_r = (float)(int16_t)(e->goals[2 * a] - e->positions[2 * a]);
    float gd_c = (float)(int16_t)(e->goals[2 * a + 1] - e->positions[2 * a + 1]);
    if (e->cfg.flags & MO_FLAG_NORMALIZE_GOAL_DELTA) {
        float den_r = (float)(e->H - 1 > 1 ? e->H - 1 : 1), den_c = (float)(e->W - 1 > 1 ? e->W - 1 : 1);
        gd_r = gd_r / den_r;
        gd_c = gd_c / den_c;
    }
    *p++ = gd_r;
    *p++ = gd_c;
    if (e->cfg.flags & MO_FLAG_GOAL_DISTANCE) *p++ = fabsf(gd_r) + fabsf(gd_c); /* :320, float32 sum */
    if (e->cfg.flags & MO_FLAG_BLOCKING_PRESSURE) *p++ = e->pressure_prev[a];   /* :322 */
    if (e->cfg.flags & MO_FLAG_ACTION_MASK) {
        int8_t mask[5];
        mo_get_action_mask(e, local, mask);
        for (int k = 0; k < 5; k++) *p++ = (float)mask[k];
    }
}

/* ------------------------------------------------------------------------------------------
 * reset MA-env:440-472
 * ------------------------------------------------------------------------------------------ */
int mo_reset(mo_env *e, float *obs) {
    e->step_count = 0;
    e->episode_blocking_count = 0.0;
    e->episode_goals_reached_total = 0.0;
    mo_reset_lock_tracking(e);
    memset(e->reached, 0, e->N);
    memset(e->completed_once, 0, e->N);
    for (int i = 0; i < e->N; i++) e->pressure_prev[i] = 0.0f;
    if (e->cfg.flags & MO_FLAG_DETERMINISTIC) {
        /* :452-455 -- positions go back to starts; goals are whatever _goals_arr holds now */
        memcpy(e->positions, e->starts, sizeof(int16_t) * 2 * e->N);
        mo_rebuild_owner_maps(e);
    } else {
        int rc = mo_generate_starts_goals(e);
        if (rc != MO_OK) return rc;
    }
    if (obs)
        for (int a = 0; a < e->N; a++) observe(e, a, obs + (size_t)a * e->L);
    return MO_OK;
}

/* ------------------------------------------------------------------------------------------
 * lock detector MA-env:374-438
 * ------------------------------------------------------------------------------------------ */
static void append_lock_history(mo_env *e) {
    int row = e->hist_head, N = e->N; /* :381-387 */
    for (int i = 0; i < N; i++) {
        e->hist_goal_progress[row * N + i] = e->goal_progress_flags[i];
        e->hist_moved[row * N + i] = e->moved_flags[i];
        e->hist_failed_move[row * N + i] = e->failed_move_flags[i];
        e->hist_distance[row * N + i] = e->distance_to_goal[i];
    }
    e->hist_head = (e->hist_head + 1) % e->Hs;
    e->hist_count = e->hist_count + 1 < e->Hs ? e->hist_count + 1 : e->Hs;
}

static int window_row(const mo_env *e, int window, int k) {
    /* idxs = (head - arange(window, 0, -1)) % Hs, k-th entry (chronological), :409-411 */
    int v = (e->hist_head - (window - k)) % e->Hs;
    if (v < 0) v += e->Hs;
    return v;
}

static void detect_lock_step(mo_env *e, const uint8_t *current_off_goal, int *deadlock, int *livelock) {
    int N = e->N;
    *deadlock = *livelock = 0;
    int any = 0;
    for (int i = 0; i < N; i++) any |= current_off_goal[i];
    if (!any) return; /* :401-402 */
    /* _get_focal_participants :389-398 */
    int n_sets = 0;
    for (int f = 0; f < N; f++) {
        if (!current_off_goal[f]) continue;
        int *set = e->participants + (size_t)n_sets * (N + 1);
        int cnt = 0;
        set[1 + cnt++] = f;
        int nbrs = 0;
        for (int j = 0; j < N; j++) {
            int d = abs(e->positions[2 * j] - e->positions[2 * f]) + abs(e->positions[2 * j + 1] - e->positions[2 * f + 1]);
            if (d <= e->cfg.lock_nearby_manhattan && d > 0) {
                set[1 + cnt++] = j;
                nbrs++;
            }
        }
        if (nbrs < e->cfg.lock_min_neighbors) continue;
        set[0] = cnt;
        n_sets++;
    }
    if (n_sets == 0) return; /* :405-406 */
    int dw = e->cfg.deadlock_window_steps, lw = e->cfg.livelock_window_steps;
    if (e->hist_count >= dw) { /* :408-420 */
        for (int s = 0; s < n_sets; s++) {
            const int *set = e->participants + (size_t)s * (N + 1);
            double gp = 0, mv = 0, fm = 0;
            for (int k = 0; k < dw; k++) {
                int row = window_row(e, dw, k);
                for (int m = 0; m < set[0]; m++) {
                    int a = set[1 + m];
                    gp += e->hist_goal_progress[row * N + a];
                    mv += e->hist_moved[row * N + a];
                    fm += e->hist_failed_move[row * N + a];
                }
            }
            if (gp <= 0.0 && mv <= 0.0 && fm > 0.0) {
                *deadlock = 1;
                return;
            }
        }
    }
    if (e->hist_count >= lw) { /* :422-436 */
        int first = window_row(e, lw, 0), last = window_row(e, lw, lw - 1);
        for (int s = 0; s < n_sets; s++) {
            const int *set = e->participants + (size_t)s * (N + 1);
            double gp = 0, mv = 0, d0 = 0, d1 = 0;
            for (int k = 0; k < lw; k++) {
                int row = window_row(e, lw, k);
                for (int m = 0; m < set[0]; m++) {
                    int a = set[1 + m];
                    gp += e->hist_goal_progress[row * N + a];
                    mv += e->hist_moved[row * N + a];
                }
            }
            for (int m = 0; m < set[0]; m++) {
                d0 += e->hist_distance[first * N + set[1 + m]];
                d1 += e->hist_distance[last * N + set[1 + m]];
            }
            if (gp <= 0.0 && mv > 0.0 && (d0 - d1) <= e->cfg.lock_progress_epsilon) {
                *livelock = 1;
                return;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * step MA-env:474-695
 * ------------------------------------------------------------------------------------------ */
int mo_step(mo_env *e, const int32_t *actions, float *obs, float *rewards_out, uint8_t *done, float *info_all,
            uint8_t *info_agent) {
    const int N = e->N, H = e->H, W = e->W;
    const int lifelong = (e->cfg.flags & MO_FLAG_LIFELONG) != 0;
    double rewards[64 * 64]; /* N <= 4096 */
    e->step_count += 1; /* :475 */
    int goal_reassigned = 0;
    for (int i = 0; i < N; i++) {
        rewards[i] = 0.0;
        e->reached_goal[i] = 0;
        e->goal_reached_step_flags[i] = 0.0f;
        e->blocking_flags[i] = 0.0f;
        e->actions_taken[i] = 0;
    }
    memcpy(e->prev_positions, e->positions, sizeof(int16_t) * 2 * N); /* :484 */

    for (int a = 0; a < N; a++) { /* :502 */
        int action = actions[a];
        if (action < 0 || action > 4) return MO_ERR_BAD_ACTION; /* :504-506, state left as is */
        e->actions_taken[a] = (int8_t)action;
        int pos_r = e->positions[2 * a], pos_c = e->positions[2 * a + 1];
        int next_r = pos_r + ACTION_DELTAS[action][0], next_c = pos_c + ACTION_DELTAS[action][1];
        e->intended_next[2 * a] = (int16_t)next_r;
        e->intended_next[2 * a + 1] = (int16_t)next_c;
        int valid = next_r >= 0 && next_r < H && next_c >= 0 && next_c < W && e->grid[next_r * W + next_c] == EMPTY_CELL &&
                    (OWN(e, occupancy_owner, next_r, next_c) == UNASSIGNED_OWNER ||
                     OWN(e, occupancy_owner, next_r, next_c) == a); /* :516-521 */
        if (valid && (next_r != pos_r || next_c != pos_c)) {       /* :522-526 */
            OWN(e, occupancy_owner, pos_r, pos_c) = UNASSIGNED_OWNER;
            e->positions[2 * a] = (int16_t)next_r;
            e->positions[2 * a + 1] = (int16_t)next_c;
            OWN(e, occupancy_owner, next_r, next_c) = (int16_t)a;
        }
        if (obs) observe(e, a, obs + (size_t)a * e->L); /* :528-534, inside the loop */

        int on_goal = e->positions[2 * a] == e->goals[2 * a] && e->positions[2 * a + 1] == e->goals[2 * a + 1];
        e->reached_goal[a] = (uint8_t)on_goal; /* :542-543 */
        if (!on_goal) continue;
        if (lifelong) { /* :547-556 */
            rewards[a] += 0.5;
            e->goal_reached_step_flags[a] = 1.0f;
            e->episode_goals_reached_total += 1.0;
            e->completed_once[a] = 1;
            e->reached[a] = 0;
            int rc = mo_assign_new_goal(e, a);
            if (rc != MO_OK) return rc;
            e->reached_goal[a] = 0;
            goal_reassigned = 1;
        } else if (!e->reached[a]) { /* :557-563 */
            e->reached[a] = 1;
            e->completed_once[a] = 1;
            rewards[a] += 0.5;
            e->goal_reached_step_flags[a] = 1.0f;
            e->episode_goals_reached_total += 1.0;
        }
    }

    if (goal_reassigned && obs) /* :565-575 */
        for (int a = 0; a < N; a++) observe(e, a, obs + (size_t)a * e->L);

    int deadlock_step = 0, livelock_step = 0;
    double deadlock_event_step = 0.0, livelock_event_step = 0.0;
    if (e->cfg.flags & MO_FLAG_LOCK_METRICS) { /* :581-606 */
        uint8_t off_goal[4096];
        for (int i = 0; i < N; i++) {
            e->moved_flags[i] = e->positions[2 * i] != e->prev_positions[2 * i] ||
                                e->positions[2 * i + 1] != e->prev_positions[2 * i + 1];
            e->failed_move_flags[i] = (e->actions_taken[i] != 0) && !e->moved_flags[i];
            e->prev_on_goal[i] = lifelong ? 0
                                          : (e->prev_positions[2 * i] == e->goals[2 * i] &&
                                             e->prev_positions[2 * i + 1] == e->goals[2 * i + 1]);
            e->current_on_goal[i] =
                e->positions[2 * i] == e->goals[2 * i] && e->positions[2 * i + 1] == e->goals[2 * i + 1];
            e->goal_progress_flags[i] =
                lifelong ? (e->goal_reached_step_flags[i] > 0.0f) : (!e->prev_on_goal[i] && e->current_on_goal[i]);
            e->distance_to_goal[i] = (int16_t)(abs(e->goals[2 * i] - e->positions[2 * i]) +
                                               abs(e->goals[2 * i + 1] - e->positions[2 * i + 1]));
            off_goal[i] = !e->current_on_goal[i];
        }
        append_lock_history(e);
        detect_lock_step(e, off_goal, &deadlock_step, &livelock_step);
        if (deadlock_step) livelock_step = 0;
        deadlock_event_step = (double)(deadlock_step && !e->deadlock_state_prev);
        livelock_event_step = (double)(livelock_step && !e->livelock_state_prev);
        e->deadlock_state_prev = (uint8_t)deadlock_step;
        e->livelock_state_prev = (uint8_t)livelock_step;
        e->episode_deadlock_steps += (double)deadlock_step;
        e->episode_livelock_steps += (double)livelock_step;
        e->episode_deadlock_events += deadlock_event_step;
        e->episode_livelock_events += livelock_event_step;
    }

    /* intent-based blocking :608-625 */
    double blocking_sum = 0.0;
    for (int b = 0; b < N; b++) {
        if (!e->reached[b]) continue;
        int br = e->positions[2 * b], bc = e->positions[2 * b + 1];
        if (br != e->prev_positions[2 * b] || bc != e->prev_positions[2 * b + 1]) continue;
        for (int o = 0; o < N; o++) {
            if (o == b || e->reached[o]) continue;
            if (e->intended_next[2 * o] == br && e->intended_next[2 * o + 1] == bc) {
                e->blocking_flags[b] = 1.0f;
                break;
            }
        }
    }
    for (int i = 0; i < N; i++) {
        e->pressure_prev[i] = e->blocking_flags[i]; /* :624 */
        blocking_sum += e->blocking_flags[i];
    }
    e->episode_blocking_count += blocking_sum; /* :625 */

    /* info :627-656 */
    double goals_reached_total;
    if (lifelong) {
        goals_reached_total = e->episode_goals_reached_total;
    } else {
        goals_reached_total = 0.0;
        for (int i = 0; i < N; i++) goals_reached_total += e->reached[i];
    }
    double step_goal_sum = 0.0, completed = 0.0;
    for (int i = 0; i < N; i++) {
        step_goal_sum += e->goal_reached_step_flags[i];
        completed += e->completed_once[i];
        if (info_agent) {
            info_agent[2 * i] = (uint8_t)(e->blocking_flags[i] != 0.0f);
            info_agent[2 * i + 1] = (uint8_t)(e->goal_reached_step_flags[i] != 0.0f);
        }
    }
    if (info_all) {
        info_all[0] = (float)step_goal_sum;
        info_all[1] = (float)goals_reached_total;
        info_all[2] = (float)blocking_sum;
        info_all[3] = (float)e->episode_blocking_count;
        info_all[4] = (float)deadlock_step;
        info_all[5] = (float)livelock_step;
        info_all[6] = (float)deadlock_event_step;
        info_all[7] = (float)livelock_event_step;
        info_all[8] = (float)e->episode_deadlock_events;
        info_all[9] = (float)e->episode_livelock_events;
        info_all[10] = (float)e->episode_deadlock_steps;
        info_all[11] = (float)e->episode_livelock_steps;
        info_all[12] = (float)(completed / (double)N);                                                    /* :638 */
        info_all[13] = (float)(goals_reached_total / (double)(e->step_count > 1 ? e->step_count : 1)); /* :655 */
    }

    /* collision penalty :658-666 (unreachable by invariant, kept) */
    for (int i = 0; i < N; i++)
        for (int j = i + 1; j < N; j++)
            if (e->positions[2 * i] == e->positions[2 * j] && e->positions[2 * i + 1] == e->positions[2 * j + 1]) {
                rewards[i] -= 1;
                rewards[j] -= 1;
            }

    /* termination :668-690 */
    int all_reached = 1;
    for (int i = 0; i < N; i++) all_reached &= e->reached_goal[i];
    int term = 0, trunc = 0;
    if (!lifelong && all_reached) {
        for (int i = 0; i < N; i++) rewards[i] += 1;
        term = 1;
        trunc = 0;
    } else if (e->step_count >= e->cfg.steps_per_episode) {
        for (int i = 0; i < N; i++)
            if (!lifelong && !e->reached_goal[i]) rewards[i] -= 1;
        term = 1;
        trunc = 1;
    }
    if (done) {
        done[0] = (uint8_t)term;
        done[1] = (uint8_t)trunc;
    }
    if (rewards_out)
        for (int i = 0; i < N; i++) rewards_out[i] = (float)rewards[i];
    return MO_OK;
}

/* ------------------------------------------------------------------------------------------
 * batch driver
 * ------------------------------------------------------------------------------------------ */
struct mo_batch {
    int32_t B;
    mo_env **envs;
    int32_t *tmp_actions;
};

mo_batch *mo_batch_create(const mo_config *cfg, int32_t num_envs, const uint8_t *grids) {
    mo_batch *b = (mo_batch *)calloc(1, sizeof(mo_batch));
    b->B = num_envs;
    b->envs = (mo_env **)calloc(num_envs, sizeof(mo_env *));
    b->tmp_actions = (int32_t *)calloc(cfg->num_agents, sizeof(int32_t));
    size_t HW = (size_t)cfg->height * cfg->width;
    for (int i = 0; i < num_envs; i++) {
        b->envs[i] = mo_create(cfg, grids + HW * i);
        if (!b->envs[i]) {
            mo_batch_destroy(b);
            return NULL;
        }
    }
    return b;
}

void mo_batch_destroy(mo_batch *b) {
    if (!b) return;
    for (int i = 0; i < b->B; i++) mo_destroy(b->envs[i]);
    free(b->envs);
    free(b->tmp_actions);
    free(b);
}

mo_env *mo_batch_env(mo_batch *b, int32_t i) { return b->envs[i]; }

int mo_batch_reset(mo_batch *b, float *obs) {
    for (int i = 0; i < b->B; i++) {
        mo_env *e = b->envs[i];
        int rc = mo_reset(e, obs ? obs + (size_t)i * e->N * e->L : NULL);
        if (rc != MO_OK) return rc;
    }
    return MO_OK;
}

int mo_batch_step(mo_batch *b, const int8_t *actions, int auto_reset, float *obs, float *rewards, uint8_t *terminated,
                  uint8_t *truncated, float *info_all, uint8_t *info_agent, float *final_obs, int32_t *err_env) {
    int first_err = MO_OK;
    for (int i = 0; i < b->B; i++) {
        mo_env *e = b->envs[i];
        const int N = e->N, L = e->L;
        for (int a = 0; a < N; a++) b->tmp_actions[a] = actions[(size_t)i * N + a];
        uint8_t done[2] = {0, 0};
        float *o = obs ? obs + (size_t)i * N * L : NULL;
        int rc = mo_step(e, b->tmp_actions, o, rewards ? rewards + (size_t)i * N : NULL, done,
                         info_all ? info_all + (size_t)i * MO_INFO_ALL : NULL,
                         info_agent ? info_agent + (size_t)i * N * 2 : NULL);
        if (rc != MO_OK) {
            if (first_err == MO_OK) {
                first_err = rc;
                if (err_env) *err_env = i;
            }
            continue;
        }
        if (terminated) terminated[i] = done[0];
        if (truncated) truncated[i] = done[1];
        if (auto_reset && (done[0] || done[1])) {
            if (final_obs && o) memcpy(final_obs + (size_t)i * N * L, o, sizeof(float) * N * L);
            rc = mo_reset(e, o);
            if (rc != MO_OK && first_err == MO_OK) {
                first_err = rc;
                if (err_env) *err_env = i;
            }
        }
    }
    return first_err;
}

/* ==========================================================================================
 * Single-agent (CTE) sibling env: /root/reference/src/environments/reference_model_single_agent.py
 * ("SA-env:N").  One policy controls all agents; full-grid observation; scalar reward.
 * Restates __init__ state :84-114, generate_starts_goals :158-191, reset :222-244, step :246-363,
 * get_obs :407-441, get_action_mask :443-495.  Pinned by golden traces recorded from the reference
 * (tests/golden/gs_*.npz) -- the reference's own tests only check dtype/bounds for this env.
 * ========================================================================================== */
struct moc_env {
    int H, W, N, steps_per_episode, deterministic;
    double blocking_penalty, move_after_goal_penalty; /* SA-env:92-93 */
    uint8_t *grid;
    int16_t *free_positions;
    int n_free;
    int32_t *starts, *positions, *goals; /* [N][2] */
    uint8_t *reached_once;               /* goal_reached_once SA-env:91 */
    int32_t step_count;
    double episode_blocking_count;
    mo_rng rng;
};

moc_env *moc_create(int H, int W, int N, int steps_per_episode, int deterministic, double blocking_penalty,
                    double move_after_goal_penalty, const uint8_t *grid) {
    if (H < 1 || W < 1 || N < 1 || N > 4096) return NULL;
    moc_env *e = (moc_env *)calloc(1, sizeof(moc_env));
    e->H = H; e->W = W; e->N = N;
    e->steps_per_episode = steps_per_episode;
    e->deterministic = deterministic;
    e->blocking_penalty = blocking_penalty;
    e->move_after_goal_penalty = move_after_goal_penalty;
    e->grid = (uint8_t *)malloc((size_t)H * W);
    memcpy(e->grid, grid, (size_t)H * W);
    e->free_positions = (int16_t *)malloc(sizeof(int16_t) * 2 * H * W);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++)
            if (grid[r * W + c] == 0) { /* np.argwhere(self.grid == 0) SA-env:171 */
                e->free_positions[2 * e->n_free] = (int16_t)r;
                e->free_positions[2 * e->n_free + 1] = (int16_t)c;
                e->n_free++;
            }
    e->starts = (int32_t *)calloc(2 * N, sizeof(int32_t));
    e->positions = (int32_t *)calloc(2 * N, sizeof(int32_t));
    e->goals = (int32_t *)calloc(2 * N, sizeof(int32_t));
    e->reached_once = (uint8_t *)calloc(N, 1);
    return e;
}

void moc_destroy(moc_env *e) {
    if (!e) return;
    free(e->grid); free(e->free_positions); free(e->starts); free(e->positions); free(e->goals); free(e->reached_once);
    free(e);
}

int moc_obs_len(const moc_env *e) { return e->H * e->W + 5 * e->N; } /* SA-env:124-141 */

void moc_set_rng(moc_env *e, uint64_t a, uint64_t b, uint64_t c, uint64_t d, int32_t has32, uint32_t u) {
    rng_set(&e->rng, a, b, c, d, has32, u);
}
void moc_get_rng(const moc_env *e, uint64_t out[6]) { rng_get(&e->rng, out); }

void moc_set_fixed_starts_goals(moc_env *e, const int32_t *starts, const int32_t *goals) { /* SA-env:109-112 */
    memcpy(e->starts, starts, sizeof(int32_t) * 2 * e->N);
    memcpy(e->positions, starts, sizeof(int32_t) * 2 * e->N);
    memcpy(e->goals, goals, sizeof(int32_t) * 2 * e->N);
}

/* SA-env:158-191: one rng.choice(F) (= integers(0, F)) per attempt, rejection until unique */
void moc_generate_starts_goals(moc_env *e) {
    const int N = e->N;
    for (int i = 0; i < N; i++) {
        for (;;) {
            int idx = (int)rng_bounded(&e->rng, (uint64_t)(e->n_free - 1));
            int r = e->free_positions[2 * idx], c = e->free_positions[2 * idx + 1];
            int clash = 0;
            for (int j = 0; j < i; j++) clash |= (e->starts[2 * j] == r && e->starts[2 * j + 1] == c);
            if (!clash) {
                e->starts[2 * i] = r;
                e->starts[2 * i + 1] = c;
                break;
            }
        }
    }
    memcpy(e->positions, e->starts, sizeof(int32_t) * 2 * N);
    for (int i = 0; i < N; i++) {
        for (;;) {
            int idx = (int)rng_bounded(&e->rng, (uint64_t)(e->n_free - 1));
            int r = e->free_positions[2 * idx], c = e->free_positions[2 * idx + 1];
            int clash = 0;
            for (int j = 0; j < i; j++) clash |= (e->goals[2 * j] == r && e->goals[2 * j + 1] == c);
            for (int j = 0; j < N; j++) clash |= (e->starts[2 * j] == r && e->starts[2 * j + 1] == c);
            if (!clash) {
                e->goals[2 * i] = r;
                e->goals[2 * i + 1] = c;
                break;
            }
        }
    }
}

static void moc_observe(const moc_env *e, float *out) {
    const int H = e->H, W = e->W, N = e->N;
    uint8_t *obs = (uint8_t *)malloc((size_t)H * W);
    memcpy(obs, e->grid, (size_t)H * W);                                                    /* SA-env:427 */
    for (int i = 0; i < N; i++) obs[e->goals[2 * i] * W + e->goals[2 * i + 1]] = (uint8_t)(i * 2 + 3);         /* :431-434 */
    for (int i = 0; i < N; i++) obs[e->positions[2 * i] * W + e->positions[2 * i + 1]] = (uint8_t)(i * 2 + 2); /* :436-439 */
    for (int k = 0; k < H * W; k++) out[k] = (float)obs[k];
    float *m = out + H * W;
    for (int i = 0; i < N; i++) { /* get_action_mask SA-env:474-493: "== 0 or odd" (an obstacle, 1, is odd) */
        int x = e->positions[2 * i], y = e->positions[2 * i + 1];
        m[i * 5 + 0] = 1.0f;
        m[i * 5 + 1] = (x > 0 && (obs[(x - 1) * W + y] == 0 || obs[(x - 1) * W + y] % 2 == 1)) ? 1.0f : 0.0f;
        m[i * 5 + 2] = (y < W - 1 && (obs[x * W + y + 1] == 0 || obs[x * W + y + 1] % 2 == 1)) ? 1.0f : 0.0f;
        m[i * 5 + 3] = (x < H - 1 && (obs[(x + 1) * W + y] == 0 || obs[(x + 1) * W + y] % 2 == 1)) ? 1.0f : 0.0f;
        m[i * 5 + 4] = (y > 0 && (obs[x * W + y - 1] == 0 || obs[x * W + y - 1] % 2 == 1)) ? 1.0f : 0.0f;
    }
    free(obs);
}

void moc_reset(moc_env *e, float *obs) { /* SA-env:222-244 */
    e->step_count = 0;
    e->episode_blocking_count = 0.0;
    memset(e->reached_once, 0, e->N);
    if (e->deterministic) memcpy(e->positions, e->starts, sizeof(int32_t) * 2 * e->N);
    else moc_generate_starts_goals(e);
    if (obs) moc_observe(e, obs);
}

/* SA-env:246-363.  info = {blocking_count_step, goals_reached_step, goals_reached_total, blocking_count_total} */
int moc_step(moc_env *e, const int32_t *action, float *obs, double *reward_out, uint8_t *done, float *info) {
    const int N = e->N, H = e->H, W = e->W;
    e->step_count += 1;
    double reward = 0;
    int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * 2 * N), *intended = (int32_t *)malloc(sizeof(int32_t) * 2 * N);
    uint8_t *reached_goal = (uint8_t *)calloc(N, 1);
    memcpy(prev, e->positions, sizeof(int32_t) * 2 * N);
    double blocking_count_step = 0.0, goals_reached_step = 0.0;
    for (int i = 0; i < N; i++) {
        int a = action[i];
        if (a < 0 || a > 4) { /* get_next_position raises ValueError("Invalid action") SA-env:401-403 */
            free(prev); free(intended); free(reached_goal);
            return MO_ERR_BAD_ACTION;
        }
        int nr = e->positions[2 * i] + ACTION_DELTAS[a][0], nc = e->positions[2 * i + 1] + ACTION_DELTAS[a][1];
        intended[2 * i] = nr;
        intended[2 * i + 1] = nc;
        int ok = nr >= 0 && nr < H && nc >= 0 && nc < W && e->grid[nr * W + nc] == 0;
        for (int j = 0; ok && j < N; j++)
            if (j != i && e->positions[2 * j] == nr && e->positions[2 * j + 1] == nc) ok = 0; /* :270-272 */
        if (ok) {
            e->positions[2 * i] = nr;
            e->positions[2 * i + 1] = nc;
        }
        if (e->positions[2 * i] == e->goals[2 * i] && e->positions[2 * i + 1] == e->goals[2 * i + 1]) { /* :281-286 */
            reached_goal[i] = 1;
            if (!e->reached_once[i]) {
                e->reached_once[i] = 1;
                reward += 0.5;
                goals_reached_step += 1.0;
            }
        }
    }
    if (obs) moc_observe(e, obs); /* :288-293, after ALL moves */
    for (int i = 0; i < N; i++) /* :296-300 */
        for (int j = i + 1; j < N; j++)
            if (e->positions[2 * i] == e->positions[2 * j] && e->positions[2 * i + 1] == e->positions[2 * j + 1]) reward -= 1;
    for (int b = 0; b < N; b++) { /* intent-based blocking penalty :303-317 */
        if (!e->reached_once[b]) continue;
        if (e->positions[2 * b] != prev[2 * b] || e->positions[2 * b + 1] != prev[2 * b + 1]) continue;
        for (int o = 0; o < N; o++) {
            if (o == b || e->reached_once[o]) continue;
            if (intended[2 * o] == e->positions[2 * b] && intended[2 * o + 1] == e->positions[2 * b + 1]) {
                reward += e->blocking_penalty;
                blocking_count_step += 1.0;
                break;
            }
        }
    }
    e->episode_blocking_count += blocking_count_step;
    for (int i = 0; i < N; i++) { /* moving after having reached the goal :320-325 */
        if (!e->reached_once[i]) continue;
        if (e->positions[2 * i] != prev[2 * i] || e->positions[2 * i + 1] != prev[2 * i + 1])
            reward += e->move_after_goal_penalty;
    }
    int all = 1;
    for (int i = 0; i < N; i++) all &= reached_goal[i];
    int term = 0, trunc = 0;
    if (all) { /* :328-331 */
        reward += N;
        term = 1;
    } else if (e->step_count >= e->steps_per_episode) { /* :340-348 */
        for (int i = 0; i < N; i++)
            if (!reached_goal[i]) reward -= 1;
        term = 1;
        trunc = 1;
    }
    double total = 0;
    for (int i = 0; i < N; i++) total += e->reached_once[i];
    if (info) {
        info[0] = (float)blocking_count_step;
        info[1] = (float)goals_reached_step;
        info[2] = (float)total;
        info[3] = (float)e->episode_blocking_count;
    }
    if (done) { done[0] = (uint8_t)term; done[1] = (uint8_t)trunc; }
    if (reward_out) *reward_out = reward;
    free(prev); free(intended); free(reached_goal);
    return MO_OK;
}

void moc_view(moc_env *e, int32_t **positions, int32_t **goals, int32_t **starts, uint8_t **reached_once,
              int32_t **step_count, double **blocking_count) {
    *positions = e->positions; *goals = e->goals; *starts = e->starts; *reached_once = e->reached_once;
    *step_count = &e->step_count; *blocking_count = &e->episode_blocking_count;
}

/* Timing helper for bench.py's cpu_baseline leg of the single-agent env: `steps` steps of B envs with the loop the
 * reference's callers run (obs, r, term, trunc, info = env.step(a); if term or trunc: env.reset(), around SA-env:246-363 /
 * :222-244), in one call so that no Python sits between the steps.  actions int8 [P][B][N], step t uses row t mod P;
 * obs_scratch float [moc_obs_len].  Returns the number of episodes finished, or a negative error code. */
long moc_run(moc_env **envs, int B, const int8_t *actions, int P, int steps, float *obs_scratch) {
    long episodes = 0;
    int32_t act[64];
    for (int t = 0; t < steps; t++) {
        const int8_t *row = actions + (size_t)(t % P) * B * envs[0]->N;
        for (int b = 0; b < B; b++) {
            moc_env *e = envs[b];
            for (int i = 0; i < e->N && i < 64; i++) act[i] = row[(size_t)b * e->N + i];
            double r;
            uint8_t done[2];
            float info[4];
            int rc = moc_step(e, act, obs_scratch, &r, done, info);
            if (rc != MO_OK) return rc;
            if (done[0] || done[1]) {
                moc_reset(e, obs_scratch);
                episodes++;
            }
        }
    }
    return episodes;
}
