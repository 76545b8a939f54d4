// mapf_kernels.inl -- device code of the MI355X step engine (included by mapf_step.hip).
//
// "MA-env:N" = /root/reference/src/environments/reference_model_multi_agent.py line N.
//
// Execution shape: an env owns a GROUP of LPE consecutive lanes (LPE = power of two >= N); lane a of
// the group is agent a, so a 64-lane wave steps 64/LPE envs (N = 64: one wavefront per env; N = 8: eight
// envs per wave, no idle lanes in the agent phases).  The reset / observe / single-agent kernels run one
// wave per workgroup; the step kernels add a second wave over the same agents for N <= 16 (state wave +
// observation wave, see step_body / obs_wave_step).  Inside a wave every LDS hand-off is intra-wave: LDS
// executes a wave's DS instructions in order, so a compiler-level fence is all the synchronisation that is
// needed (wave_lds_sync); the two waves of a step workgroup meet at LDS-only barriers (wg_sync).
//
// Cross-lane traffic goes through two LDS tables per step (one entry per lane, written once, then
// read by every lane of the group with 16-byte reads), instead of one broadcast per agent pair:
//   move table     {old cell, wanted cell}                                         (8 B / agent)
//   pair table     {old | new<<16, goal | reached<<16 | (dist delta+256)<<17,
//                   intended cell (+1,+1) or ~0 when the agent has reached its goal}   (16 B / agent)
// Lanes that hold no agent publish sentinel entries that fail every test, so the pair loops carry no
// "is this a real agent" predicate.
//
// At the headline shape the launch is one or two waves per SIMD and a step is bound by the waves' chains of
// dependent instructions (measured: profiles/r01), so the env configuration is a
// template policy: KFixed<...> turns agent count, sensor range, observation layout,
// flags and lock windows into compile-time constants for the BASELINE.json shapes, KRuntime keeps
// every other configuration working from the same source.

// MAPF_NS: empty (anonymous namespace) in the library build; the run-time specialisation (mapf_step.hip: jit_specialize)
// compiles this file once more under a named namespace so that its one instantiation has a linkable name.
#ifndef MAPF_NS
#define MAPF_NS
#endif
namespace MAPF_NS {

// ------------------------------------------------------------------------------------------------
// device-side data layout
// ------------------------------------------------------------------------------------------------
// Agent state: 48 B per agent as FOUR PLANES over the agent index i = env * N + a (structure of arrays, one allocation,
// plane k starts agent_plane_off(k, bn8) bytes behind plane 0; env-major, so a wave's 64 agents are one contiguous run
// in every plane and every load / store of a wave is fully coalesced):
//   plane 0,  8 B  "hot": w0 = pos (row<<8 | col) | goal << 16,  w1 = start | flags << 16 | pass << 24
//                  -- all the move phase needs: the state wave of k_step3 waits for these 512 bytes per wave only
//   plane 1, 16 B  moved, failed          lock-history shift registers, bit k = flag k steps ago
//   plane 2, 16 B  progress, dist[0..1]   dist[4]: goal-distance history as 16 x uint8, byte k = distance k steps ago
//   plane 3,  8 B  dist[2..3]             (used when the livelock window is <= 16 steps; longer windows: the int16 ring)
// pass = which neighbours of `pos` an agent can step on as far as the GRID goes (bit 0 up, 1 right, 2 down, 3 left: no
// obstacle, inside the grid), a cache derived from pos and the obstacle rows by whoever stores a position
// (agent_pass_bits); it saves the state wave of k_step3 the obstacle rows altogether.
// (Rounds 1-2 kept 48-byte records; their stores needed a transpose through LDS to be coalesced, and any wave that
// wanted the position pulled all 3 KiB of a wave's records through the memory system.)
struct AgentRec {  // host-side image of one agent (mapf_get_state / mapf_set_state assemble it from the planes)
    uint32_t w0, w1;
    uint64_t moved, failed, progress;
    uint32_t dist[4];
};
static_assert(sizeof(AgentRec) == 48, "AgentRec must be 48 bytes");
__host__ __device__ constexpr uint32_t agent_plane_stride(int B, int N) { return ((uint32_t)B * (uint32_t)N * 8u + 255u) & ~255u; }
__host__ __device__ constexpr size_t agent_plane_off(int k, uint32_t bn8) {  // plane sizes 1, 2, 2, 1 x bn8
    return (size_t)bn8 * (k == 0 ? 0 : (k == 1 ? 1 : (k == 2 ? 3 : 5)));
}
__host__ __device__ constexpr size_t agent_state_bytes(uint32_t bn8) { return (size_t)bn8 * 6; }

constexpr int kFlagReached = 1, kFlagCompleted = 2, kFlagPressure = 4;
constexpr int kScalInts = MAPF_NUM_COUNTERS;  // 16 int32 = 64 B per env
constexpr uint32_t kNoCell = 0xFFFFu;         // "no target"
constexpr uint32_t kIdleCell = 0xFFFEu;       // cell of a lane that holds no agent (row 255: far from any grid)
constexpr uint32_t kIdleGoal = 0xFFFDu;
// Obstacle rows are padded so that no bounds logic is needed: kRowPad all-ones sentinel rows above and below
// each env's rows in LDS, and (when W <= 64 - 2*kRowPad: Io::col_pad = kRowPad) kRowPad sentinel bits at the
// low end of every row; bits beyond the grid width are ones as well.
constexpr int kRowPad = 5;  // = MAPF_MAX_SENSOR_RANGE
// Pre-sampled placement ("slot") of an env's next episode.  In finite mode the env's stream is consumed by reset()
// alone (generate_starts_goals, MA-env:267-282), so the next rng.choice(F, 2N) can be drawn at any time before the
// episode ends: sampler workgroups appended to the k_step grid do it in the background (sampler_wave), and the
// step that ends the episode only installs the placement.  Invariant: slot valid <=> next_sg[env] holds the cells
// choice() yields from the env's visible stream state, which then sits in vis_rng[env], while Params::rng[env] already
// holds the state after that draw; slot invalid <=> Params::rng[env] is the visible state.  Consuming the slot is
// therefore just invalidating it (no copy at the episode boundary); host setters of the stream invalidate it too, and
// mapf_get_state reports vis_rng for envs whose slot is pending.
constexpr uint32_t kSlotInvalid = 0xFFFFFFFFu;
// The background draw is split over two launches (sampler_wave): after its first half the env's word 0 reads kSlotStaged
// (the stream is advanced and vis_rng set as for a pending slot, the bounded draws wait in Params::stage_vals); the
// second half turns it into a valid slot.  Consumers treat a staged slot as "none" and draw inline from vis_rng.
constexpr uint32_t kSlotStaged = 0xFFFFFFFEu;
// The specialised finite kernels slice the draw finer still and run it in the observation wave of the env's own
// workgroup (draw_slice), seven slices of at most ~1.5 k cycles: kSlotStaged = later half of the raw outputs done
// (stream advanced), kSlotStaged2 = all of them, kSlotStaged3 = bounded draws, kSlotStaged4 / 5 = Floyd (two
// halves), kSlotStaged6 = shuffle; the free-cell gather then makes the slot valid.
constexpr uint32_t kSlotStaged2 = 0xFFFFFFFDu, kSlotStaged3 = 0xFFFFFFFCu, kSlotStaged4 = 0xFFFFFFFBu,
                   kSlotStaged5 = 0xFFFFFFFAu, kSlotStaged6 = 0xFFFFFFF9u,
                   kSlotStageFailed = 0xFFFFFFF8u;  // (a bounded draw might be rejected: the env will draw inline)
__host__ __device__ constexpr int stage_dwords(int N) { return 4 * N + 4; }  // Params::stage_vals per env
__device__ __forceinline__ bool slot_word_valid(uint32_t w) { return w < 0xFFFFFFF0u; }  // cells are < 0x4000
__device__ __forceinline__ bool slot_word_staged(uint32_t w) { return !slot_word_valid(w) && w != kSlotInvalid; }
constexpr int kDbgRow = 32;  // diagnostic build: s_memtime stamps per workgroup (Params::dbg)
// next_sg lives behind the env scalars in one allocation, so the step kernel reaches it from preloaded arguments
__host__ __device__ __forceinline__ uint32_t *slots_of(int *scal, int B) {
    return reinterpret_cast<uint32_t *>(scal + (size_t)B * MAPF_NUM_COUNTERS);
}
// ... and behind the slots: the env streams [B][6] uint64 (Params::rng) and the free-cell counts [B] (Params::n_free),
// so that a sampler wave can fetch them in its first round trip without waiting for Params
__host__ __device__ __forceinline__ uint64_t *streams_of(int *scal, int B, int N) {
    const size_t slot_bytes = ((size_t)B * N * sizeof(uint32_t) + 15) & ~(size_t)15;
    return reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(slots_of(scal, B)) + slot_bytes);
}
__host__ __device__ __forceinline__ int *free_counts_of(int *scal, int B, int N) {
    return reinterpret_cast<int *>(streams_of(scal, B, N) + (size_t)B * 6);
}
__host__ __device__ inline size_t scal_block_bytes(int B, int N) {
    return (size_t)B * MAPF_NUM_COUNTERS * sizeof(int) + (((size_t)B * N * sizeof(uint32_t) + 15) & ~(size_t)15) +
           (size_t)B * 6 * sizeof(uint64_t) + (size_t)B * sizeof(int);
}

// Engine constants, resident in device memory and read through a __restrict__ pointer (scalar loads
// at the point of use; a by-value struct this size is held in SGPRs for the whole kernel and spills).
struct Params {
    int B, H, W, N;
    int sr, V, L, steps_per_episode;
    uint32_t flags;
    int dw, lw, nearby, min_nbrs, eps_floor, hs, ring_stride;
    float den_r, den_c;
    int HW;
    int hash_cap;      // Floyd hash-set size (power of two), numpy: 1 + gen_mask(uint64(1.2 * 2N))
    int scratch_i16;   // int16 entries of reset scratch per group
    int lds_tab_off, lds_stage_off, lds_scratch_off;  // byte offsets into dynamic LDS (rows start at 0)
    // cold state (reset / respawn / error paths)
    uint64_t *rng;
    const uint16_t *free_cells;  // [B][HW], k-th free cell (row-major) as row<<8|col
    const uint16_t *free_rank;   // [B][HW], row-major rank of a free cell among free cells
    const int *n_free;           // [B]
    int *err;                    // [4] code, env, agent, value
    int *ep_acc;                 // [B][MAPF_NUM_EPISODE_ACC] lifetime per-env sums over finished episodes
    uint32_t *next_sg;           // [B][N] pre-sampled placement of the NEXT episode: start | goal << 16, kSlotInvalid = none
    uint32_t *stage_vals;        // [B][stage_dwords(N)] intermediate data of a staged background draw (kSlotStaged*)
    uint64_t *vis_rng;           // [B][6] while a slot is pending: the env's stream state BEFORE the pre-draw, i.e. what
                                 // NumPy's bit_generator.state shows at that point (Params::rng already holds the state after)
    unsigned long long *dbg;     // diagnostic build only (-DMAPF_STAMPS): [blocks][16] s_memtime stamps
};

// kernel arguments passed by value: the hot state arrays (as kernel arguments they are known to be
// global-address-space pointers; pointers read out of Params are generic and compile to flat_* ops)
// and the per-launch io tensors
// IoHead: what the first global loads of a wave need.  The step kernels take these 12 dwords (plus the Params
// pointer: 14, the preload budget next to the kernarg segment pointer -- a 15th argument is fetched with a scalar load
// and, if the first addresses depend on it, delays every load of the wave: measured, +0.2 us per step) as individual kernel arguments, because
// only scalar and pointer arguments can be preloaded into SGPRs at wave launch (a by-value struct cannot): the
// state loads then leave in the wave's first cycles instead of behind a scalar-load round trip (build.py,
// -amdgpu-kernarg-preload-count).
struct IoHead {
    uint2 *agents;               // plane 0 of the agent state (the other planes: agent_plane_off)
    int *scal;
    const uint64_t *grid_rows;   // [B][H], bit c = obstacle, bits >= W set
    const int8_t *actions;
    int B, H, W;
    uint32_t bn8;                // agent_plane_stride(B, N)
};
struct IoTail {
    int16_t *dist_ring;          // [B][N][ring_stride], slot = history row index mod lw (only when lw > 16)
    // hot scalars (copies of the Params fields every launch needs before its first memory access)
    int eps_floor, steps_per_episode;
    int use_map;  // large-N path: per-env cell map in LDS instead of the all-pairs walk (lds_map_off valid)
    int lds_map_off;
    float den_r, den_c;
    int lds_tab_off, lds_stage_off, lds_scratch_off;
    float *obs, *rewards;
    uint8_t *terminated, *truncated;
    float *info_all;
    uint8_t *info_agent;
    float *final_obs;
    const uint8_t *env_mask;
    int auto_reset;
    // sliced background draw (draw_slice): as kernel arguments these are global-address-space pointers; read through
    // Params they are generic and their loads (flat_*) would sit in the LDS wait counter of the observation wave
    uint32_t *stage_vals;
    const uint16_t *free_cells;
    const uint16_t *free_rank;   // (k_stepw's lifelong respawn: ranks are gathered before the move phase, and a pointer read out
                                 //  of Params would be one more scalar round trip in front of those loads)
    uint64_t *vis_rng;
    const uint64_t *jump_c;      // [B][32][2] per env: S_q * inc mod 2^128 (hi, lo), q = 1 .. 32 -- the increment of a
                                 // PCG64 stream never changes, so this half of the jump-ahead (state_q = A_q * state +
                                 // S_q * inc) is a table written when the stream is set (mapf_set_rng_state / set_state)
};
struct Io : IoHead, IoTail {
    int col_pad;  // kRowPad when the rows carry low sentinel bits (W <= 64 - 2 * kRowPad), else 0; a function of W
};
__host__ __device__ constexpr int col_pad_for(int W) { return W <= 64 - 2 * kRowPad ? kRowPad : 0; }
#define MAPF_IO_HEAD_PARAMS                                                                                          \
    uint2 *a_agents, int *a_scal, const uint64_t *a_grid_rows, const int8_t *a_actions, const int a_B, const int a_H, \
        const int a_W, const uint32_t a_bn8
__device__ __forceinline__ Io join_io(uint2 *agents, int *scal, const uint64_t *grid_rows, const int8_t *actions, int B,
                                      int H, int W, uint32_t bn8, const IoTail &tail) {
    Io io;
    static_cast<IoTail &>(io) = tail;
    io.agents = agents;
    io.scal = scal;
    io.grid_rows = grid_rows;
    io.actions = actions;
    io.B = B;
    io.H = H;
    io.W = W;
    io.bn8 = bn8;
    io.col_pad = col_pad_for(W);
    return io;
}
#define MAPF_IO_JOIN join_io(a_agents, a_scal, a_grid_rows, a_actions, a_B, a_H, a_W, a_bn8, tail)

// Scalar-cache warm-up, first statement of the hot kernels.  The compiler fetches kernel arguments and Params
// fields lazily, one scalar load (and one full wait) per first use, so a wave would pay a chain of scalar-cache
// misses spread through its prologue.  Demanding one dword of every 64-byte line here makes all those loads go
// out in the first batch, under a single wait (the Params pointer itself arrives preloaded in SGPRs: build.py).
__device__ __forceinline__ void warm_scalar_cache(const Params *__restrict__ pp, const IoTail &io) {
    asm volatile("" ::"s"(io.eps_floor), "s"(io.truncated), "s"(io.auto_reset), "s"(pp->N), "s"(pp->HW), "s"(pp->ep_acc));
}

// In-kernel stamps (diagnostic build only; never in the shipped library): lane 0 of each wave records
// s_memtime at phase boundaries of k_step into Params::dbg, which nothing else reads.
#ifdef MAPF_STAMPS
#define MAPF_STAMP(k)                                                                   \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        unsigned long long _t;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)(env0 / (64 / LPE)) * kDbgRow + (k)] = _t; \
    } while (0)
// wave entry time, taken before the first scalar load is waited for; written to slot 15 by MAPF_STAMP_ENTRY_STORE
#define MAPF_STAMP_ENTRY()                                                              \
    unsigned long long _t_entry;                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t_entry)::"memory")
#define MAPF_STAMP_ENTRY_STORE()                                                        \
    do {                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              \
        if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)(env0 / (64 / LPE)) * kDbgRow + 15] = _t_entry; \
    } while (0)
// the same for the observation wave (lane 64 of a two-wave workgroup), slots 10..14
#define MAPF_STAMP_W1(k)                                                                \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        unsigned long long _t;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if (p.dbg && threadIdx.x == 64) p.dbg[(size_t)(env0 / (64 / LPE)) * kDbgRow + (k)] = _t; \
    } while (0)
// the aux wave of k_step3 (lane 128 of a three-wave workgroup), slots 21, 22, 29..31
#define MAPF_STAMP_W2(k)                                                                \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        unsigned long long _t;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if (p.dbg && threadIdx.x == 128) p.dbg[(size_t)(env0 / (64 / LPE)) * kDbgRow + (k)] = _t; \
    } while (0)
// sampler waves (wave 0 of a sampler workgroup): slots 0 entry, 1 need known, 2 draw done, 3 placement stored, 4 = active
#define MAPF_STAMP_SW(k)                                                                \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        unsigned long long _t;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)sw_row * kDbgRow + (k)] = _t;            \
    } while (0)
#else
#define MAPF_STAMP_SW(k) do { } while (0)
#define MAPF_STAMP(k) do { } while (0)
#define MAPF_STAMP_W1(k) do { } while (0)
#define MAPF_STAMP_W2(k) do { } while (0)
#define MAPF_STAMP_ENTRY() do { } while (0)
#define MAPF_STAMP_ENTRY_STORE() do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// configuration policies
// ------------------------------------------------------------------------------------------------
constexpr int obs_len_of(int sr, uint32_t flags) {
    return (2 * sr + 1) * (2 * sr + 1) + 2 + ((flags & MAPF_FLAG_GOAL_DISTANCE) ? 1 : 0) +
           ((flags & MAPF_FLAG_BLOCKING_PRESSURE) ? 1 : 0) + ((flags & MAPF_FLAG_ACTION_MASK) ? 5 : 0);
}

// Sampler workgroups (sampler_wave) of a k_step launch: how many, and where in the grid.  Each of their waves looks
// after 64 envs; the count is rounded up to a multiple of 8 so that env workgroup b stays on XCD b mod 8 when they lead
// the grid.  kSamplerFront: a compile-time finite / sampled-placement configuration puts them FIRST (blockIdx < count:
// they start with the launch and their draw, about as long as an env step, ends with it instead of after it); the
// runtime-config kernel only knows its mode after a scalar load, so there they trail the env workgroups.
__host__ __device__ constexpr int sampler_blocks_for(int B, int waves_per_wg) {
    return ((((B + 63) / 64) + waves_per_wg - 1) / waves_per_wg + 7) & ~7;
}

struct KRuntime {
    static constexpr bool kFixed = false;
    static constexpr bool kSamplerFront = false;
    static constexpr bool kSlicedDraw = false;
    static constexpr bool kMapAlways = false;
    static constexpr int kL = 0, kV = 0;  // (observation length / window side when known at compile time)
    static constexpr uint32_t kFlags = 0;
    __device__ static __forceinline__ int N(const Params &p) { return p.N; }
    __device__ static __forceinline__ int sr(const Params &p) { return p.sr; }
    __device__ static __forceinline__ int V(const Params &p) { return p.V; }
    __device__ static __forceinline__ int L(const Params &p) { return p.L; }
    __device__ static __forceinline__ uint32_t flags(const Params &p) { return p.flags; }
    __device__ static __forceinline__ int dw(const Params &p) { return p.dw; }
    __device__ static __forceinline__ int lw(const Params &p) { return p.lw; }
    __device__ static __forceinline__ int hs(const Params &p) { return p.hs; }
    __device__ static __forceinline__ int nearby(const Params &p) { return p.nearby; }
    __device__ static __forceinline__ int min_nbrs(const Params &p) { return p.min_nbrs; }
    __device__ static __forceinline__ int ring_stride(const Params &p) { return p.ring_stride; }
};

// The runtime-config kernels for FULL groups of 4 or 8 agents with finite episodes and sampled placements: everything
// still comes from Params, but the background draw runs in slices inside the env workgroups and small grids get the
// three-wave kernel, like the prebuilt shapes (round 3; mapf_create: rt_sliced).  The host guarantees N == lanes per env.
// (also full groups of 16: their draw is sliced, the kernel stays the two-wave one)
struct KRuntimeSliced : KRuntime {
    static constexpr bool kSlicedDraw = true;
};

template <int N_, int SR_, uint32_t FLAGS_, int DW_, int LW_, int NEARBY_, int MINN_>
struct KFixed {
    static constexpr bool kFixed = true;
    // finite episodes with sampled placements and full groups of 4, 8 or 16 agents: the background draw runs in slices
    // inside the env's own workgroup (draw_slice), no sampler workgroups
    // (the slices assume N = lanes per env, two values per lane; other N take the sampler workgroups)
    static constexpr bool kSlicedDraw = (FLAGS_ & (MAPF_FLAG_LIFELONG | MAPF_FLAG_DETERMINISTIC)) == 0 && (N_ == 4 || N_ == 8 || N_ == 16);
    static constexpr bool kSamplerFront = (FLAGS_ & (MAPF_FLAG_LIFELONG | MAPF_FLAG_DETERMINISTIC)) == 0 && !kSlicedDraw;
    // wide groups (N > 16): the specialisation is only used with the LDS cell map (mapf_create falls back to the
    // runtime-config kernel otherwise), so the all-pairs walk is not compiled in: at N = 64 its unrolled loops were
    // a third of the code (98 KB against a 64 KB instruction cache) and much of the register pressure
    static constexpr bool kMapAlways = N_ > 16;
    static constexpr int kL = obs_len_of(SR_, FLAGS_), kV = 2 * SR_ + 1;
    static constexpr uint32_t kFlags = FLAGS_;
    __device__ static __forceinline__ int N(const Params &) { return N_; }
    __device__ static __forceinline__ int sr(const Params &) { return SR_; }
    __device__ static __forceinline__ int V(const Params &) { return 2 * SR_ + 1; }
    __device__ static __forceinline__ int L(const Params &) { return obs_len_of(SR_, FLAGS_); }
    __device__ static __forceinline__ uint32_t flags(const Params &) { return FLAGS_; }
    __device__ static __forceinline__ int dw(const Params &) { return DW_; }
    __device__ static __forceinline__ int lw(const Params &) { return LW_; }
    __device__ static __forceinline__ int hs(const Params &) { return DW_ > LW_ ? DW_ : LW_; }
    __device__ static __forceinline__ int nearby(const Params &) { return NEARBY_; }
    __device__ static __forceinline__ int min_nbrs(const Params &) { return MINN_; }
    __device__ static __forceinline__ int ring_stride(const Params &) { return (LW_ + 7) & ~7; }
};

// ------------------------------------------------------------------------------------------------
// group (sub-wave) primitives
// ------------------------------------------------------------------------------------------------
template <int LPE>
__device__ __forceinline__ uint64_t group_mask() {
    return LPE == 64 ? ~0ull : ((1ull << (LPE & 63)) - 1ull);
}

template <int LPE>
__device__ __forceinline__ uint64_t gballot(bool pred, int lane) {
    uint64_t b = __ballot(pred);
    if (LPE == 64) return b;
    return (b >> (lane & ~(LPE - 1))) & group_mask<LPE>();
}

// group ballot in the narrowest register type that holds one bit per lane of the group
template <int LPE>
struct GMask {
    using type = uint32_t;
};
template <>
struct GMask<64> {
    using type = uint64_t;
};
template <int LPE>
__device__ __forceinline__ typename GMask<LPE>::type gballot_n(bool pred, int lane) {
    return (typename GMask<LPE>::type)gballot<LPE>(pred, lane);
}

// OR the per-group bit sets of a wave ballot together (bit i = "some group has agent i set")
template <int LPE>
__device__ __forceinline__ uint64_t fold_groups(uint64_t m) {
    if (LPE <= 32) m |= m >> 32;
    if (LPE <= 16) m |= m >> 16;
    if (LPE <= 8) m |= m >> 8;
    if (LPE <= 4) m |= m >> 4;
    return m & group_mask<LPE>();
}

// broadcast lane j of every group (j wave-uniform); only used on rare paths (lifelong respawn)
template <int LPE>
__device__ __forceinline__ uint32_t gshfl(uint32_t v, int j) {
    if (LPE == 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, j);
    return (uint32_t)__shfl((int)v, j, LPE);
}

// intra-wave LDS hand-off point (DS ops of one wave execute in order)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// LDS hand-off between the two waves of a workgroup.  Not __syncthreads(): that also waits for the wave's global
// stores (vmcnt(0)), i.e. for the write-through observation stream to reach memory, once per barrier.  Only LDS
// contents are exchanged here, so the wave waits for its own DS operations and then joins the barrier.
__device__ __forceinline__ void wg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// V x V window bit masks (bit dr*V+dc); MW = 32 (V <= 5), 64 (V <= 7), 128 (V <= 11)
// ------------------------------------------------------------------------------------------------
template <int MW>
struct WMask;
template <>
struct WMask<32> {
    uint32_t lo;
    __device__ __forceinline__ void clear() { lo = 0; }
    __device__ __forceinline__ void set_if(bool c, int b) { lo |= c ? (1u << (b & 31)) : 0u; }
    __device__ __forceinline__ void clear_bit(int b) { lo &= ~(1u << b); }
    __device__ __forceinline__ bool get(int b) const { return (lo >> b) & 1u; }
    __device__ __forceinline__ void or_row(uint32_t w, int shift) { lo |= w << shift; }
    __device__ __forceinline__ void toggle_if(bool c, int b) { lo ^= c ? (1u << (b & 31)) : 0u; }
    static __device__ __forceinline__ WMask pick(bool c, const WMask &x, const WMask &y) { return WMask{c ? x.lo : y.lo}; }
    __device__ __forceinline__ uint32_t nib(int t0) const { return (lo >> t0) & 15u; }
    __device__ __forceinline__ WMask operator|(const WMask &o) const { return WMask{lo | o.lo}; }
    __device__ __forceinline__ WMask andnot(const WMask &o) const { return WMask{lo & ~o.lo}; }
    __device__ __forceinline__ bool any() const { return lo != 0; }
    __device__ __forceinline__ int first() const { return (int)__builtin_ctz(lo); }  // (any() must hold)
};
template <>
struct WMask<64> {
    uint64_t lo;
    __device__ __forceinline__ void clear() { lo = 0; }
    __device__ __forceinline__ void set_if(bool c, int b) { lo |= c ? (1ull << (b & 63)) : 0ull; }
    __device__ __forceinline__ void clear_bit(int b) { lo &= ~(1ull << b); }
    __device__ __forceinline__ bool get(int b) const { return (lo >> b) & 1ull; }
    __device__ __forceinline__ void or_row(uint32_t w, int shift) { lo |= (uint64_t)w << shift; }
    __device__ __forceinline__ void toggle_if(bool c, int b) { lo ^= c ? (1ull << (b & 63)) : 0ull; }
    static __device__ __forceinline__ WMask pick(bool c, const WMask &x, const WMask &y) { return WMask{c ? x.lo : y.lo}; }
    __device__ __forceinline__ uint32_t nib(int t0) const { return (uint32_t)(lo >> t0) & 15u; }
    __device__ __forceinline__ WMask operator|(const WMask &o) const { return WMask{lo | o.lo}; }
    __device__ __forceinline__ WMask andnot(const WMask &o) const { return WMask{lo & ~o.lo}; }
    __device__ __forceinline__ bool any() const { return lo != 0; }
    __device__ __forceinline__ int first() const { return (int)__builtin_ctzll(lo); }
};
template <>
struct WMask<128> {
    uint64_t lo, hi;
    __device__ __forceinline__ void clear() { lo = hi = 0; }
    __device__ __forceinline__ void set_if(bool c, int b) {
        const uint64_t v = c ? 1ull : 0ull;
        if (b < 64) lo |= v << (b & 63); else hi |= v << ((b - 64) & 63);
    }
    __device__ __forceinline__ void clear_bit(int b) {
        if (b < 64) lo &= ~(1ull << b); else hi &= ~(1ull << (b - 64));
    }
    __device__ __forceinline__ bool get(int b) const { return b < 64 ? ((lo >> b) & 1ull) : ((hi >> (b - 64)) & 1ull); }
    __device__ __forceinline__ void or_row(uint32_t w, int shift) {
        if (shift < 64) {
            lo |= (uint64_t)w << shift;
            if (shift > 0) hi |= (uint64_t)w >> (64 - shift);
        } else {
            hi |= (uint64_t)w << (shift - 64);
        }
    }
    __device__ __forceinline__ void toggle_if(bool c, int b) {
        const uint64_t v = c ? 1ull : 0ull;
        if (b < 64) lo ^= v << (b & 63); else hi ^= v << ((b - 64) & 63);
    }
    static __device__ __forceinline__ WMask pick(bool c, const WMask &x, const WMask &y) {
        return WMask{c ? x.lo : y.lo, c ? x.hi : y.hi};
    }
    // t0 is a multiple of 4, so a nibble never straddles the two words
    __device__ __forceinline__ uint32_t nib(int t0) const {
        return (uint32_t)(t0 < 64 ? (lo >> t0) : (hi >> (t0 - 64))) & 15u;
    }
    __device__ __forceinline__ WMask operator|(const WMask &o) const { return WMask{lo | o.lo, hi | o.hi}; }
    __device__ __forceinline__ WMask andnot(const WMask &o) const { return WMask{lo & ~o.lo, hi & ~o.hi}; }
    __device__ __forceinline__ bool any() const { return (lo | hi) != 0; }
    __device__ __forceinline__ int first() const { return lo ? (int)__builtin_ctzll(lo) : 64 + (int)__builtin_ctzll(hi); }
};

// V bits of an obstacle row starting at column c0; outside the grid = 1.
// Padded rows (col_pad != 0, W <= 54): the row carries col_pad sentinel bits below column 0 and ones above the
// grid, so one shift does it.
__device__ __forceinline__ uint32_t row_window_padded(uint64_t ext, int c0, int V, int col_pad) {
    return (uint32_t)(ext >> ((c0 + col_pad) & 63)) & ((1u << V) - 1u);
}
// Unpadded rows (W > 54): c0 may be negative or the window may run past bit 63.  Branch-free: both shift
// directions are computed and selected (the compiler turns an if/else over 64-bit shifts into exec-mask branches).
__device__ __forceinline__ uint32_t row_window_wide(uint64_t ext, int c0, int V) {
    const int n = -c0;
    const uint64_t right = (ext >> (c0 & 63)) | ((c0 > 0) ? (~0ull << ((64 - c0) & 63)) : 0ull);
    const uint64_t left = (ext << (n & 63)) | ((1ull << (n & 63)) - 1ull);
    const uint64_t w = (c0 >= 0) ? right : left;
    return (uint32_t)w & ((1u << V) - 1u);
}

// |r - r'| + |c - c'| of two packed cells (row<<8 | col, upper half zero): one v_sad_u8
__device__ __forceinline__ int cell_l1(uint32_t x, uint32_t y) { return (int)__builtin_amdgcn_sad_u8(x, y, 0u); }

// ------------------------------------------------------------------------------------------------
// NumPy Generator(PCG64) on device (numpy 2.2.6: pcg64.h, distributions.c, _generator.pyx)
// ------------------------------------------------------------------------------------------------
struct Pcg {
    uint64_t shi, slo, ihi, ilo;
    uint32_t has32, uinteger;
};
__device__ __forceinline__ void pcg_load(Pcg &g, const uint64_t *w) {
    g.shi = w[0]; g.slo = w[1]; g.ihi = w[2]; g.ilo = w[3];
    g.has32 = (uint32_t)w[4]; g.uinteger = (uint32_t)w[5];
}
__device__ __forceinline__ void pcg_store(const Pcg &g, uint64_t *w) {
    w[0] = g.shi; w[1] = g.slo; w[2] = g.ihi; w[3] = g.ilo; w[4] = g.has32; w[5] = g.uinteger;
}
__device__ __forceinline__ uint64_t pcg_next64(Pcg &g) {
    // state = state * 0x2360ED051FC65DA44385DF649FCCF645 + inc (mod 2^128); XSL-RR output of the NEW state
    const uint64_t MH = 0x2360ED051FC65DA4ull, ML = 0x4385DF649FCCF645ull;
    uint64_t lo = g.slo * ML;
    uint64_t hi = __umul64hi(g.slo, ML) + g.slo * MH + g.shi * ML;
    uint64_t nlo = lo + g.ilo;
    uint64_t nhi = hi + g.ihi + (nlo < lo ? 1ull : 0ull);
    g.slo = nlo; g.shi = nhi;
    uint64_t x = nhi ^ nlo;
    unsigned rot = (unsigned)(nhi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
}
__device__ __forceinline__ uint32_t pcg_next32(Pcg &g) {
    if (g.has32) { g.has32 = 0; return g.uinteger; }
    uint64_t n = pcg_next64(g);
    g.has32 = 1;
    g.uinteger = (uint32_t)(n >> 32);
    return (uint32_t)n;
}
// random_bounded_uint64(off=0, rng, use_masked=false) for rng < 2^32-1: Lemire with rejection.
// The rejection probability per draw is thr / 2^32 < 1e-6 for the bounds of this engine (< 2^13), so 4096 rejections
// in a row cannot happen (p < 1e-24000); the cap makes termination structural, and if it ever trips `stuck` is set
// and the caller latches MAPF_ERR_RNG_GUARD instead of continuing on a stream that no longer matches NumPy's.
__device__ __forceinline__ uint32_t pcg_bounded(Pcg &g, uint32_t rng, bool &stuck) {
    if (rng == 0) return 0;  // no draw
    const uint32_t excl = rng + 1u;
    uint64_t m = (uint64_t)pcg_next32(g) * excl;
    uint32_t left = (uint32_t)m;
    if (left < excl) {
        const uint32_t thr = (0xFFFFFFFFu - rng) % excl;
        int guard = 0;
        for (; left < thr && guard < 4096; guard++) {
            m = (uint64_t)pcg_next32(g) * excl;
            left = (uint32_t)m;
        }
        stuck |= left < thr;
    }
    return (uint32_t)(m >> 32);
}

// ---- PCG64 jump-ahead: state_j = A_j * state_0 + S_j * inc (mod 2^128), A_j = M^j, S_j = 1 + M + ... + M^(j-1).
//      Lets the lanes of a group produce the next outputs of ONE stream side by side (parallel reset sampling). ----
__device__ const uint64_t kPcgJumpA[65][2] = {  // [j] = {hi, lo}
    {0x0000000000000000ull, 0x0000000000000001ull},
    {0x2360ED051FC65DA4ull, 0x4385DF649FCCF645ull},
    {0x17BCE35BDF69743Cull, 0x529ED9EB20E0AE99ull},
    {0x25F041404BD80E82ull, 0xEB5AE837ED42153Dull},
    {0xF4DD417327DB7A9Bull, 0xD194DFBE42D45771ull},
    {0x16C406E9FBE6C01Full, 0x1712DD28EC4E2775ull},
    {0x19B2ADD48DEFCDA8ull, 0x81AB1C97E7371089ull},
    {0x9C49933E0E7F5995ull, 0x3D56ECCA7FE71AEDull},
    {0x6347AF777A7898F6ull, 0xD1A2D6F33505FFE1ull},
    {0x18F8F9F7734932A8ull, 0x16509219B4CC2DA5ull},
    {0xE0928447FDA002F4ull, 0x3520872BC960DB79ull},
    {0x7FFF9B7984D6798Cull, 0x36814613656D6D9Dull},
    {0x5673A2CDBFB514C0ull, 0xC86DE54D59EF6951ull},
    {0xD008599933890BF1ull, 0xFCCE321A884738D5ull},
    {0x8D76620EAAEC2FFDull, 0x3B7E935CC08AFF69ull},
    {0xBB65D723FCF3D618ull, 0x2B8DD86F3591BD4Dull},
    {0xB6A4239F3B315F84ull, 0xF6EF6D3D288C03C1ull},
    {0x122FCCEAFED32C57ull, 0xEEA812BE56247905ull},
    {0xA2575AAF626FC8D2ull, 0x96C0E142CF1B6C59ull},
    {0x5C6A218569CC205Cull, 0x59A6D5EFCA6DB9FDull},
    {0x01FC86B51903EBA0ull, 0x64AC95EF58E83F31ull},
    {0xE2266B713B195F27ull, 0xCEC4DF67ED5E1E35ull},
    {0x8F37EE576D4F4D2Eull, 0xC299A89268A11249ull},
    {0x49225B7C80456050ull, 0x27D4463AE42813ADull},
    {0x2EAB8AF607E4C633ull, 0xCB029BDA22918BA1ull},
    {0x9B752C7238207889ull, 0xDDE45A9B70B35865ull},
    {0xF8D77BDD4FC33988ull, 0x98027F76E2C3E139ull},
    {0x7227137533ADDE3Eull, 0xEC0E973139A47A5Dull},
    {0x9C66D9841EFEA4D2ull, 0x6766F40DDC065911ull},
    {0xFBA09424C6129045ull, 0x16DDF92EEFD85795ull},
    {0x7B053022297A0D47ull, 0xD3DBC41FDF34C929ull},
    {0x4FFF222CCDBFD619ull, 0xC6E06F18A6339E0Dull},
    {0x2C82901AD1CB0CD1ull, 0x82B631BA6B261781ull},
    {0x09B2F524AD4778E2ull, 0xBA5E228D55A64BC5ull},
    {0x3340B19703FE2BEBull, 0x3EB4DDED9E9DBA19ull},
    {0x444E12C9D609DD48ull, 0xB795FC8124432EBDull},
    {0x43DBDCEDB0981F9Cull, 0x77A5AC0BF6A136F1ull},
    {0x155A595EF55BD10Full, 0x5EBBECF3DB4B64F5ull},
    {0x6C7E40C3C318A6EDull, 0x4E8DA0F12C91A409ull},
    {0x2D8782C3E5472BA9ull, 0x2A270549450DDC6Dull},
    {0xC29A3AF9625E59F8ull, 0xABED85A248692761ull},
    {0x365CA55913F9E80Eull, 0x987FDA77307AD325ull},
    {0x149559E09D8FB49Cull, 0x639C0B2E547C76F9ull},
    {0xFA4B8066E6321F2Bull, 0xE1D08A03D54B571Dull},
    {0xAEDBCDE0C6862989ull, 0x69BDEF77512058D1ull},
    {0xC92A38F6FEE662F3ull, 0xA85BB0B2889CC655ull},
    {0x48358C36996266D1ull, 0x64CAEFB64F9322E9ull},
    {0xE769316CEE909844ull, 0xCCAF328D5EE04ECDull},
    {0xD6E61FE52FDFEB98ull, 0xA825C00A3C8A3B41ull},
    {0x652EF16D5960C4C7ull, 0xD7DA7647BCFE6E85ull},
    {0x1C48D99EF19B5BCDull, 0x9AD9794B1BC397D9ull},
    {0x5215BFDE6E4E50BFull, 0xEDAED0DD378E737Dull},
    {0x071D5483DFDC0147ull, 0x63CC9E1586FB3EB1ull},
    {0xE155A077C32C68B3ull, 0x58C293CCB401FBB5ull},
    {0xAB4C9B79448B3C1Cull, 0x007735C46BA4C5C9ull},
    {0xE33ACD710F626603ull, 0x7641B50ACCA4752Dull},
    {0x9D86EF706106C4CAull, 0xD7145CF783C8D321ull},
    {0x15FB139EF60C3088ull, 0x176ADE673D4E9DE5ull},
    {0xF601543BB8EA7814ull, 0xC016FDE81F669CB9ull},
    {0xAAC2E61C7727D111ull, 0xDC609E1CDBAE03DDull},
    {0x00147DF1B7837868ull, 0x67928145C4B96891ull},
    {0xD3234B0DE825F8E7ull, 0xAC55F62D93008515ull},
    {0xA8A94DEA2829AFDBull, 0x0B63E3D136C20CA9ull},
    {0x0E60F8FE78F6FAE5ull, 0xC632C3854823CF8Dull},
    {0xDAB03F988288676Eull, 0xE49E66C4D2746F01ull},
};
__device__ const uint64_t kPcgJumpS[65][2] = {  // [j] = {hi, lo}
    {0x0000000000000000ull, 0x0000000000000000ull},
    {0x0000000000000000ull, 0x0000000000000001ull},
    {0x2360ED051FC65DA4ull, 0x4385DF649FCCF646ull},
    {0x3B1DD060FF2FD1E0ull, 0x9624B94FC0ADA4DFull},
    {0x610E11A14B07E063ull, 0x817FA187ADEFBA1Cull},
    {0x55EB531472E35AFFull, 0x53148145F0C4118Dull},
    {0x6CAF59FE6ECA1B1Eull, 0x6A275E6EDD123902ull},
    {0x866207D2FCB9E8C6ull, 0xEBD27B06C449498Bull},
    {0x22AB9B110B39425Cull, 0x292967D144306478ull},
    {0x85F34A8885B1DB52ull, 0xFACC3EC479366459ull},
    {0x9EEC447FF8FB0DFBull, 0x111CD0DE2E0291FEull},
    {0x7F7EC8C7F69B10EFull, 0x463D5809F7636D77ull},
    {0xFF7E64417B718A7Bull, 0x7CBE9E1D5CD0DB14ull},
    {0x55F2070F3B269F3Cull, 0x452C836AB6C04465ull},
    {0x25FA60A86EAFAB2Eull, 0x41FAB5853F077D3Aull},
    {0xB370C2B7199BDB2Bull, 0x7D7948E1FF927CA3ull},
    {0x6ED699DB168FB143ull, 0xA9072151352439F0ull},
    {0x257ABD7A51C110C8ull, 0x9FF68E8E5DB03DB1ull},
    {0x37AA8A6550943D20ull, 0x8E9EA14CB3D4B6B6ull},
    {0xDA01E514B30405F3ull, 0x255F828F82F0230Full},
    {0x366C069A1CD0264Full, 0x7F06587F4D5DDD0Cull},
    {0x38688D4F35D411EFull, 0xE3B2EE6EA6461C3Dull},
    {0x1A8EF8C070ED7117ull, 0xB277CDD693A43A72ull},
    {0xA9C6E717DE3CBE46ull, 0x75117668FC454CBBull},
    {0xF2E942945E821E96ull, 0x9CE5BCA3E06D6068ull},
    {0x2194CD8A6666E4CAull, 0x67E8587E02FEEC09ull},
    {0xBD09F9FC9E875D54ull, 0x45CCB31973B2446Eull},
    {0xB5E175D9EE4A96DCull, 0xDDCF3290567625A7ull},
    {0x2808894F21F8751Bull, 0xC9DDC9C1901AA004ull},
    {0xC46F62D340F719EEull, 0x3144BDCF6C20F915ull},
    {0xC00FF6F80709AA33ull, 0x4822B6FE5BF950AAull},
    {0x3B15271A3083B77Bull, 0x1BFE7B1E3B2E19D3ull},
    {0x8B144946FE438D94ull, 0xE2DEEA36E161B7E0ull},
    {0xB796D961D00E9A66ull, 0x65951BF14C87CF61ull},
    {0xC149CE867D561349ull, 0x1FF33E7EA22E1B26ull},
    {0xF48A801D81543F34ull, 0x5EA81C6C40CBD53Full},
    {0x38D892E7575E1C7Dull, 0x163E18ED650F03FCull},
    {0x7CB46FD507F63C19ull, 0x8DE3C4F95BB03AEDull},
    {0x920EC933FD520D28ull, 0xEC9FB1ED36FB9FE2ull},
    {0xFE8D09F7C06AB416ull, 0x3B2D52DE638D43EBull},
    {0x2C148CBBA5B1DFBFull, 0x65545827A89B2058ull},
    {0xEEAEC7B5081039B8ull, 0x1141DDC9F10447B9ull},
    {0x250B6D0E1C0A21C6ull, 0xA9C1B841217F1ADEull},
    {0x39A0C6EEB999D663ull, 0x0D5DC36F75FB91D7ull},
    {0x33EC47559FCBF58Eull, 0xEF2E4D734B46E8F4ull},
    {0xE2C8153666521F18ull, 0x58EC3CEA9C6741C5ull},
    {0xABF24E2D6538820Cull, 0x0147ED9D2504081Aull},
    {0xF427DA63FE9AE8DDull, 0x6612DD5374972B03ull},
    {0xDB910BD0ED2B8122ull, 0x32C20FE0D37779D0ull},
    {0xB2772BB61D0B6CBAull, 0xDAE7CFEB1001B511ull},
    {0x17A61D23766C3182ull, 0xB2C24632CD002396ull},
    {0x33EEF6C268078D50ull, 0x4D9BBF7DE8C3BB6Full},
    {0x8604B6A0D655DE10ull, 0x3B4A905B20522EECull},
    {0x8D220B24B631DF57ull, 0x9F172E70A74D6D9Dull},
    {0x6E77AB9C795E480Aull, 0xF7D9C23D5B4F6952ull},
    {0x19C44715BDE98426ull, 0xF850F801C6F42F1Bull},
    {0xFCFF1486CD4BEA2Aull, 0x6E92AD0C9398A448ull},
    {0x9A8603F72E52AEF5ull, 0x45A70A0417617769ull},
    {0xB0811796245EDF7Dull, 0x5D11E86B54B0154Eull},
    {0xA6826BD1DD495792ull, 0x1D28E6537416B207ull},
    {0x514551EE547128A3ull, 0xF98984704FC4B5E4ull},
    {0x5159CFE00BF4A10Cull, 0x611C05B6147E1E75ull},
    {0x247D1AEDF41A99F4ull, 0x0D71FBE3A77EA38Aull},
    {0xCD2668D81C4449CFull, 0x18D5DFB4DE40B033ull},
    {0xDB8761D6953B44B4ull, 0xDF08A33A26647FC0ull},
};
struct U128 {
    uint64_t hi, lo;
};
__device__ __forceinline__ U128 mul128(U128 x, U128 y) {  // low 128 bits of the product
    U128 r;
    r.lo = x.lo * y.lo;
    r.hi = __umul64hi(x.lo, y.lo) + x.lo * y.hi + x.hi * y.lo;
    return r;
}
__device__ __forceinline__ U128 add128(U128 x, U128 y) {
    U128 r;
    r.lo = x.lo + y.lo;
    r.hi = x.hi + y.hi + (r.lo < x.lo ? 1ull : 0ull);
    return r;
}
__device__ __forceinline__ uint64_t pcg_output(U128 st) {  // XSL-RR of a state
    const uint64_t x = st.hi ^ st.lo;
    const unsigned rot = (unsigned)(st.hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
}

// ------------------------------------------------------------------------------------------------
// per-lane register image of an agent
// ------------------------------------------------------------------------------------------------
struct Lane {
    uint32_t pos, goal, start;  // row<<8|col
    uint32_t flags;
    uint64_t moved, failed, progress;
    uint4 dist;  // 16 x uint8 goal-distance history
};

// Index `idx` must be readable for every lane (callers clamp the index of idle lanes to a real agent): the four loads
// are unconditional, so they issue back to back with the wave's other loads instead of sitting in an exec-masked
// branch with its own wait; idle lanes then replace what they read by the sentinels.
// Two halves so that a kernel can put other work (issuing more loads, waiting for scalar loads) between the
// issue and the first use.
struct LaneRaw {
    uint2 h;       // plane 0
    uint4 q1, q2;  // planes 1, 2
    uint2 d;       // plane 3
};
__device__ __forceinline__ const uint4 *agent_plane16(const uint2 *hot, uint32_t bn8, int k) {
    return reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(hot) + agent_plane_off(k, bn8));
}
__device__ __forceinline__ void lane_issue_hist(const uint2 *hot, uint32_t bn8, size_t idx, LaneRaw &r) {
    r.q1 = agent_plane16(hot, bn8, 1)[idx];
    r.q2 = agent_plane16(hot, bn8, 2)[idx];
    r.d = reinterpret_cast<const uint2 *>(agent_plane16(hot, bn8, 3))[idx];
}
__device__ __forceinline__ void lane_issue(const uint2 *hot, uint32_t bn8, size_t idx, LaneRaw &r) {
    r.h = hot[idx];
    lane_issue_hist(hot, bn8, idx, r);
}
__device__ __forceinline__ void lane_unpack(const LaneRaw &r, bool is_agent, Lane &st) {
    st.dist = is_agent ? make_uint4(r.q2.z, r.q2.w, r.d.x, r.d.y) : make_uint4(0, 0, 0, 0);
    st.pos = is_agent ? (r.h.x & 0xFFFFu) : (uint32_t)kIdleCell;
    st.goal = is_agent ? (r.h.x >> 16) : (uint32_t)kIdleGoal;
    st.start = is_agent ? (r.h.y & 0xFFFFu) : (uint32_t)kIdleCell;
    st.flags = is_agent ? ((r.h.y >> 16) & 0xFFu) : 0u;
    st.moved = is_agent ? ((uint64_t)r.q1.x | ((uint64_t)r.q1.y << 32)) : 0ull;
    st.failed = is_agent ? ((uint64_t)r.q1.z | ((uint64_t)r.q1.w << 32)) : 0ull;
    st.progress = is_agent ? ((uint64_t)r.q2.x | ((uint64_t)r.q2.y << 32)) : 0ull;
}
__device__ __forceinline__ void load_lane(const uint2 *hot, uint32_t bn8, size_t idx, bool is_agent, Lane &st) {
    LaneRaw r;
    lane_issue(hot, bn8, idx, r);
    lane_unpack(r, is_agent, st);
}

// the hot plane alone (the single-agent env keeps no lock history: it never touches planes 1-3)
__device__ __forceinline__ uint32_t load_lane_hot(const uint2 *hot, size_t idx, bool is_agent, Lane &st) {
    const uint2 h = hot[idx];
    st.pos = is_agent ? (h.x & 0xFFFFu) : (uint32_t)kIdleCell;
    st.goal = is_agent ? (h.x >> 16) : (uint32_t)kIdleGoal;
    st.start = is_agent ? (h.y & 0xFFFFu) : (uint32_t)kIdleCell;
    st.flags = is_agent ? ((h.y >> 16) & 0xFFu) : 0u;
    st.moved = st.failed = st.progress = 0ull;
    st.dist = make_uint4(0, 0, 0, 0);
    return is_agent ? (h.y >> 24) : 0u;  // pass bits
}

// 16-byte state store (plain: write-through and nontemporal variants were measured and are slower)
__device__ __forceinline__ void store_state16(void *dst, const uint4 v) { *reinterpret_cast<uint4 *>(dst) = v; }

// Which neighbours of `cell` can be stepped on as far as the grid goes (bit 0 up, 1 right, 2 down, 3 left): exactly the
// test of the move phase (MA-env:508-512: inside the grid and not an obstacle) for the four targets.  myrows = the env's
// row 0 in LDS (sentinel rows on either side: rows -1 and H read as all obstacles).
__device__ __forceinline__ uint32_t agent_pass_bits(const uint64_t *myrows, uint32_t cell, int col_pad, int W) {
    const int r = (int)(cell >> 8), c = (int)(cell & 255u);
    const uint64_t up = myrows[r - 1], mid = myrows[r], dn = myrows[r + 1];
    const bool lf_ok = col_pad != 0 || c > 0, rt_ok = col_pad != 0 || c + 1 < W;
    const uint32_t b_up = (uint32_t)(up >> ((c + col_pad) & 63)) & 1u;
    const uint32_t b_dn = (uint32_t)(dn >> ((c + col_pad) & 63)) & 1u;
    const uint32_t b_rt = rt_ok ? ((uint32_t)(mid >> ((c + 1 + col_pad) & 63)) & 1u) : 1u;
    const uint32_t b_lf = lf_ok ? ((uint32_t)(mid >> ((c - 1 + col_pad) & 63)) & 1u) : 1u;
    return (b_up | (b_rt << 1) | (b_dn << 2) | (b_lf << 3)) ^ 15u;
}

__device__ __forceinline__ void store_lane_hot(uint2 *hot, size_t idx, const Lane &st, uint32_t pass) {
    hot[idx] = make_uint2((st.pos & 0xFFFFu) | (st.goal << 16), (st.start & 0xFFFFu) | ((st.flags & 0xFFu) << 16) | (pass << 24));
}
__device__ __forceinline__ void store_lane_hist(uint2 *hot, uint32_t bn8, size_t idx, const Lane &st) {
    uint4 *p1 = const_cast<uint4 *>(agent_plane16(hot, bn8, 1)), *p2 = const_cast<uint4 *>(agent_plane16(hot, bn8, 2));
    uint2 *p3 = reinterpret_cast<uint2 *>(const_cast<uint4 *>(agent_plane16(hot, bn8, 3)));
    store_state16(p1 + idx, make_uint4((uint32_t)st.moved, (uint32_t)(st.moved >> 32), (uint32_t)st.failed, (uint32_t)(st.failed >> 32)));
    store_state16(p2 + idx, make_uint4((uint32_t)st.progress, (uint32_t)(st.progress >> 32), st.dist.x, st.dist.y));
    p3[idx] = make_uint2(st.dist.z, st.dist.w);
}
// the same as <wave-uniform first agent> + <lane>: the plane addresses stay scalar, the lane offset 32 bits
__device__ __forceinline__ void store_lane_hist(uint2 *hot, uint32_t bn8, size_t idx0, int lane, const Lane &st) {
    uint4 *p1 = const_cast<uint4 *>(agent_plane16(hot, bn8, 1)) + idx0, *p2 = const_cast<uint4 *>(agent_plane16(hot, bn8, 2)) + idx0;
    uint2 *p3 = reinterpret_cast<uint2 *>(const_cast<uint4 *>(agent_plane16(hot, bn8, 3))) + idx0;
    store_state16(p1 + lane, make_uint4((uint32_t)st.moved, (uint32_t)(st.moved >> 32), (uint32_t)st.failed, (uint32_t)(st.failed >> 32)));
    store_state16(p2 + lane, make_uint4((uint32_t)st.progress, (uint32_t)(st.progress >> 32), st.dist.x, st.dist.y));
    p3[lane] = make_uint2(st.dist.z, st.dist.w);
}
// the whole agent; `myrows` = the env's obstacle rows in LDS (for the pass bits of st.pos)
__device__ __forceinline__ void store_lane(uint2 *hot, uint32_t bn8, size_t idx, const Lane &st, const uint64_t *myrows,
                                           int col_pad, int W) {
    store_lane_hot(hot, idx, st, agent_pass_bits(myrows, st.pos, col_pad, W));
    store_lane_hist(hot, bn8, idx, st);
}

__device__ __forceinline__ void load_scal(const int *scal, int env, int *sc) {
    const int4 *sp = reinterpret_cast<const int4 *>(scal + (size_t)env * kScalInts);
    int4 s0 = sp[0], s1 = sp[1], s2 = sp[2];
    sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w;
    sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
    sc[8] = s2.x; sc[9] = s2.y; sc[10] = s2.z; sc[11] = s2.w;
}
__device__ __forceinline__ void store_scal(int *scal, int env, const int *sc) {
    int4 *sp = reinterpret_cast<int4 *>(scal + (size_t)env * kScalInts);
    store_state16(sp, make_uint4(sc[0], sc[1], sc[2], sc[3]));
    store_state16(sp + 1, make_uint4(sc[4], sc[5], sc[6], sc[7]));
    store_state16(sp + 2, make_uint4(sc[8], sc[9], sc[10], sc[11]));
}

// obstacle rows of the wave's envs -> LDS, env g at lrows[g * (H + 2*kRowPad) + kRowPad + r]; the pad rows are
// all ones.  Lane (g, a) fetches rows a, a + LPE, ... of its own env.  Split in two so that a kernel can issue
// these loads together with its other state loads and only then wait: rows_issue() fetches the first 4*LPE rows
// into registers, rows_commit() writes them (and any further rows of taller grids) to LDS.
struct RowRegs {
    uint64_t t[4];
};
template <int LPE>
__device__ __forceinline__ void rows_issue(const uint64_t *grid_rows, int H, int lane, int env0, int ngroups, RowRegs &rr) {
    const int grp = lane / LPE, a = lane % LPE;
    const uint64_t *src = grid_rows + (size_t)(env0 + min(grp, ngroups - 1)) * H;
#pragma unroll
    for (int u = 0; u < 4; u++) rr.t[u] = src[min(u * LPE + a, H - 1)];  // clamped, unconditional; commit selects
}
template <int LPE>
__device__ __forceinline__ void rows_commit(const uint64_t *grid_rows, int H, uint64_t *lrows, int lane, int env0,
                                            int ngroups, const RowRegs &rr) {
    const int grp = lane / LPE, a = lane % LPE;
    const int stride = H + 2 * kRowPad;
    uint64_t *dst = lrows + grp * stride;
    const bool ok = grp < ngroups;
    const uint64_t *src = grid_rows + (size_t)(env0 + (ok ? grp : 0)) * H;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int r = u * LPE + a;
        if (r < H) dst[kRowPad + r] = ok ? rr.t[u] : ~0ull;
    }
    for (int r = 4 * LPE + a; r < H; r += LPE) dst[kRowPad + r] = ok ? src[r] : ~0ull;  // H > 4*LPE only
    for (int k = a; k < 2 * kRowPad; k += LPE) dst[k < kRowPad ? k : H + k] = ~0ull;
}
template <int LPE>
__device__ __forceinline__ void load_rows_to_lds(const uint64_t *grid_rows, int H, uint64_t *lrows, int lane, int env0,
                                                 int ngroups) {
    RowRegs rr;
    rows_issue<LPE>(grid_rows, H, lane, env0, ngroups, rr);
    rows_commit<LPE>(grid_rows, H, lrows, lane, env0, ngroups, rr);
}

// per-env step mask of mapf_step_masked: an env whose byte is zero is not touched at all (state, generator, counters,
// statistics, outputs, error latch); its lane group behaves like the idle groups of a ragged last wave
__device__ __forceinline__ bool env_live(const Io &io, int env) {
    return io.env_mask == nullptr || io.env_mask[min(env, io.B - 1)] != 0;
}

__device__ __forceinline__ void raise_error(const Params &p, int code, int env, int agent, int value) {
    if (atomicCAS(&p.err[0], 0, code) == 0) {
        p.err[1] = env;
        p.err[2] = agent;
        p.err[3] = value;
    }
}

// Checking build (-DMAPF_CHECK; dl_reference_models_amd/build.py: build_check(), selected with MAPF_CHECK_BUILD=1): every
// scatter / gather index into the LDS regions that the draw, the inline reset, the cell map and the staging rows compute
// from DATA is range-checked against the region of its lane group; a violation latches MAPF_ERR_INTERNAL with the env,
// the site id below and the offending value (mapf_poll_error).  Round 2's one real kernel bug (draw_shuffle16 storing for
// lane groups that were not drawing, out[inv[..]] landing in a neighbour group's scratch) only showed as a wrong goal
// cell 200 soak cases later; this build reports it at the store.  Sites: 1 raw outputs, 2 bounded draws, 3 shuffle
// scatter (16 values), 4 shuffle scatter (128 values), 5 sequential Floyd hash / output, 6 free-cell gather index,
// 7 cell-map index, 8 staging row, 9 slice staging, 10 scratch layout, 11 bit-row index (k_step3 with bit rows: goal / old-cell /
// intent rows and the window rows read from them).
#ifdef MAPF_CHECK
#define MAPF_CHK(P, cond, site, env, val)                                                   \
    do {                                                                                    \
        if (!(cond)) raise_error((P), MAPF_ERR_INTERNAL, (env), (site), (int)(val));        \
    } while (0)
#else
#define MAPF_CHK(P, cond, site, env, val) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// move phase (MA-env:502-526): agents move in index order against live occupancy.  Restated as a
// dependency problem: agent i (target T) is blocked iff
//   - a higher-index agent still stands on T (it has not had its turn yet), or
//   - a lower-index agent stood on T and did not move away, or
//   - a lower-index agent with the same target moved into T.
// Outcomes depend on lower indices only, so they resolve in rounds of two ballots; almost every
// agent has no dependency at all and the loop ends after one round.
// ------------------------------------------------------------------------------------------------
template <class K, int LPE>
__device__ __forceinline__ uint32_t resolve_moves(const Params &p, uint2 *tabg, int lane, int a, uint32_t old,
                                                  uint32_t tgt) {
    using gm_t = typename GMask<LPE>::type;
    constexpr int C = LPE < 8 ? LPE : 8;
    const int N = K::N(p);
    tabg[a] = make_uint2(old, tgt);  // idle lanes publish {kIdleCell, kNoCell}: they match nothing
    wave_lds_sync();
    gm_t occ_bit = 0;  // the agent standing on my target (positions are unique: at most one)
    gm_t cont = 0;     // lower-index contenders for my target
    const bool want = tgt != kNoCell;
    const gm_t below = ((gm_t)1 << a) - 1;  // indices < a
    for (int j0 = 0; j0 < N; j0 += C) {
        uint2 e[C];
#pragma unroll
        for (int u = 0; u < C; u++) e[u] = tabg[j0 + u];
#pragma unroll
        for (int u = 0; u < C; u++) {
            const gm_t bit = (gm_t)1 << (j0 + u);
            occ_bit |= (e[u].x == tgt) ? bit : 0;
            cont |= (e[u].y == tgt) ? bit : 0;
        }
    }
    cont = want ? (cont & below) : 0;
    const gm_t occ_low = occ_bit & below;    // occupant has a lower index: blocked unless it moved away
    const bool occ_high = (occ_bit & ~below) != 0;  // occupant has a higher index (never me: tgt != old): blocked
    const gm_t dep = cont | occ_low;
    bool resolved = !want, moved = false;
    gm_t R = gballot_n<LPE>(resolved, lane), M = 0;
#pragma unroll 1
    for (int it = 0; it <= N; it++) {
        if (__all(resolved)) break;
        if (!resolved && (dep & ~R) == 0) {
            moved = !(occ_high || (occ_low & ~M) != 0 || (cont & M) != 0);
            resolved = true;
        }
        R = gballot_n<LPE>(resolved, lane);
        M = gballot_n<LPE>(moved, lane);
    }
    wave_lds_sync();  // the table region is rewritten after this point
    return moved ? tgt : old;
}

// Cell word of the LDS occupancy map (large-N path).  Every agent ORs its fields in with LDS atomics:
//   bits 0-6   index+1 of the agent standing on the cell after the move ("new")      bits 22-31 its delta+256
//   bits 7-13  index+1 of the agent that stood on the cell before the move ("old")
//   bits 14-20 index+1 of the agent whose goal the cell is            bit 21  some not-yet-reached agent intends to enter
// The map has a kRowPad border of never-occupied cells, so window / neighbourhood reads need no bounds checks.
__device__ __forceinline__ int map_index(uint32_t cell, int map_w) {
    return ((int)(cell >> 8) + kRowPad) * map_w + (int)(cell & 255u) + kRowPad;
}

// The same resolution for wide groups with the LDS cell map (owner-old fields already written): the occupant of
// the target is one map read, and a per-cell contender count (LDS atomic add into the not-yet-used delta field)
// tells which agents are contended at all; only those are walked to build the lower-index contender sets.
template <class K, int LPE>
__device__ __forceinline__ uint32_t resolve_moves_map(const Params &p, uint32_t *mapg, int map_w, int lane, int a,
                                                      uint32_t old, uint32_t tgt) {
    using gm_t = typename GMask<LPE>::type;
    const bool want = tgt != kNoCell;
    uint32_t *tcell = mapg + (want ? map_index(tgt, map_w) : 0);
    if (want) atomicAdd(tcell, 1u << 22);
    wave_lds_sync();
    const uint32_t w = want ? *tcell : 0u;
    const uint32_t occ1 = (w >> 7) & 127u;  // index+1 of the agent standing on my target (0 = none)
    const bool contended = want && ((w >> 22) & 7u) > 1u;
    gm_t cont = 0;
    uint64_t u = fold_groups<LPE>(__ballot(contended));
    while (u) {  // typically a handful of agents, not N
        const int j = (int)__builtin_ctzll(u);
        u &= u - 1;
        const uint32_t tj = gshfl<LPE>(contended ? tgt : kNoCell, j);
        cont |= (want && j < a && tj == tgt) ? ((gm_t)1 << j) : 0;
    }
    const gm_t below = ((gm_t)1 << a) - 1;
    const gm_t occ_bit = occ1 ? ((gm_t)1 << (occ1 - 1)) : 0;
    const gm_t occ_low = occ_bit & below;
    const bool occ_high = (occ_bit & ~below) != 0;
    const gm_t dep = cont | occ_low;
    bool resolved = !want, moved = false;
    gm_t R = gballot_n<LPE>(resolved, lane), M = 0;
#pragma unroll 1
    for (int it = 0; it <= K::N(p); it++) {
        if (__all(resolved)) break;
        if (!resolved && (dep & ~R) == 0) {
            moved = !(occ_high || (occ_low & ~M) != 0 || (cont & M) != 0);
            resolved = true;
        }
        R = gballot_n<LPE>(resolved, lane);
        M = gballot_n<LPE>(moved, lane);
    }
    if (want) atomicAnd(tcell, ~(7u << 22));  // give the field back: it receives the distance deltas next
    wave_lds_sync();
    return moved ? tgt : old;
}

// ------------------------------------------------------------------------------------------------
// observation of every agent lane -> LDS staging row (MA-env:707-747 get_obs, :749-773 mask,
// :306-335 flatten), fused with the other all-pairs work of a step (MODE below): neighbour sets of
// the lock detector (MA-env:389-398), intent blocking (:608-623).  (The coincidence penalty MA-env:658-666 is
// identically zero: agents never share a cell -- the move rule keeps it so and mapf_set_state rejects states that
// violate it -- so nothing is computed for it.)
// Pair-table entry of agent j:
//   x = old | new<<16      y = goal | reached<<16 | (delta+256)<<17      z = intended cell (+1,+1) or ~0
// final_state: everybody at their new cell (reset, or after a lifelong respawn MA-env:565-575);
// otherwise agent i sees agents <= i at their new cell and agents > i at their old one (MA-env:528).
// ------------------------------------------------------------------------------------------------
// ---- pieces of an agent's observation, shared by observe() and the precomputing observation wave ----------------
// obstacle / out-of-bounds bits of the V x V window whose top-left cell is (r0, c0); rows[d] = obstacle row r0 + d
template <int MW, int MAXV>
__device__ __forceinline__ WMask<MW> window_obstacles(const uint64_t *rows, int c0, int V, int col_pad) {
    WMask<MW> obst;
    obst.clear();
    if (col_pad) {  // one wave-uniform branch around the whole window, not one per row
#pragma unroll
        for (int d = 0; d < MAXV; d++) {
            if (d < V) obst.or_row(row_window_padded(rows[d], c0, V, col_pad), d * V);
        }
    } else {
#pragma unroll
        for (int d = 0; d < MAXV; d++) {
            if (d < V) obst.or_row(row_window_wide(rows[d], c0, V), d * V);
        }
    }
    return obst;
}
// window bit of `cell` (row<<8 | col) or nothing if it lies outside
template <int MW>
__device__ __forceinline__ void window_set(WMask<MW> &m, uint32_t cell, int r0, int c0, int V) {
    const int r = (int)((cell >> 8) & 255u) - r0, c = (int)(cell & 255u) - c0;
    m.set_if(max((unsigned)r, (unsigned)c) < (unsigned)V, __mul24(r, V) + c);
}
// goal delta component (MA-env:330-335): correctly rounded fp32 quotient when normalised
__device__ __forceinline__ float goal_delta(int d, float den, bool normalise) {
    const float f = (float)d;
    return normalise ? f / den : f;
}
// the flat observation row (MA-env:306-328) from the window masks: cell codes by priority obstacle / out-of-bounds
// 1 > other agent 2 > own goal 3 > other goal 4 > empty 0, as three bit planes, four cells per round (bit spread +
// byte->float converts); then goal delta, optional goal distance, pressure flag, action mask (MA-env:749-773)
// from the three bit planes of the cell codes (code = bit0 | bit1 << 1 | g4 << 2) and the blocked cells oa
template <class K, int MW, int MAXV>
__device__ __forceinline__ void emit_obs_planes(const Params &p, float *srow, const WMask<MW> &bit0, const WMask<MW> &bit1,
                                                const WMask<MW> &g4, const WMask<MW> &oa, float gd_r, float gd_c,
                                                bool pressure) {
    const int V = K::V(p), sr = K::sr(p), VV = V * V, ctr = sr * V + sr;
    const uint32_t flags = K::flags(p);
    constexpr uint32_t KS = 0x00204081u, MS = 0x01010101u;  // bit i of a nibble -> LSB of byte i
    if constexpr (K::kFixed && K::kL % 4 == 0 && K::kV <= MAXV) {
        // Rows of a multiple of four floats (L = 28, 52: the reference-default layouts) go to the staging buffer as 16-byte
        // writes: with a row stride of L words, 32-bit writes of the same element by 64 lanes fall on 16 banks (gcd(L, 64) =
        // 4: four lanes per bank, every write replayed four times), 16-byte writes of 16 lanes at a time cover all 64.
        constexpr int L = K::kL, CV = K::kV * K::kV, CTR = (K::kV / 2) * K::kV + K::kV / 2;
        float f[L];
#pragma unroll
        for (int t0 = 0; t0 < CV; t0 += 4) {
            const uint32_t by = ((bit0.nib(t0) * KS) & MS) | (((bit1.nib(t0) * KS) & MS) << 1) | (((g4.nib(t0) * KS) & MS) << 2);
            f[t0] = (float)(by & 0xFFu);
            if (t0 + 1 < CV) f[t0 + 1] = (float)((by >> 8) & 0xFFu);
            if (t0 + 2 < CV) f[t0 + 2] = (float)((by >> 16) & 0xFFu);
            if (t0 + 3 < CV) f[t0 + 3] = (float)(by >> 24);
        }
        int q = CV;
        f[q++] = gd_r;
        f[q++] = gd_c;
        if constexpr ((K::kFlags & MAPF_FLAG_GOAL_DISTANCE) != 0) f[q++] = fabsf(gd_r) + fabsf(gd_c);
        if constexpr ((K::kFlags & MAPF_FLAG_BLOCKING_PRESSURE) != 0) f[q++] = pressure ? 1.0f : 0.0f;
        if constexpr ((K::kFlags & MAPF_FLAG_ACTION_MASK) != 0) {
            f[q++] = 1.0f;
            f[q++] = (K::kV > 1 && !oa.get(CTR - K::kV)) ? 1.0f : 0.0f;
            f[q++] = (K::kV > 1 && !oa.get(CTR + 1)) ? 1.0f : 0.0f;
            f[q++] = (K::kV > 1 && !oa.get(CTR + K::kV)) ? 1.0f : 0.0f;
            f[q++] = (K::kV > 1 && !oa.get(CTR - 1)) ? 1.0f : 0.0f;
        }
        float4 *dst = reinterpret_cast<float4 *>(srow);
#pragma unroll
        for (int i = 0; i < L; i += 4) dst[i >> 2] = make_float4(f[i], f[i + 1], f[i + 2], f[i + 3]);
        return;
    }
#pragma unroll
    for (int t0 = 0; t0 < MAXV * MAXV; t0 += 4) {
        if (t0 < VV) {
            const uint32_t by = ((bit0.nib(t0) * KS) & MS) | (((bit1.nib(t0) * KS) & MS) << 1) |
                                (((g4.nib(t0) * KS) & MS) << 2);
            srow[t0] = (float)(by & 0xFFu);
            if (t0 + 1 < VV) srow[t0 + 1] = (float)((by >> 8) & 0xFFu);
            if (t0 + 2 < VV) srow[t0 + 2] = (float)((by >> 16) & 0xFFu);
            if (t0 + 3 < VV) srow[t0 + 3] = (float)(by >> 24);
        }
    }
    float *q = srow + VV;
    *q++ = gd_r;
    *q++ = gd_c;
    if (flags & MAPF_FLAG_GOAL_DISTANCE) *q++ = fabsf(gd_r) + fabsf(gd_c);
    if (flags & MAPF_FLAG_BLOCKING_PRESSURE) *q++ = pressure ? 1.0f : 0.0f;
    if (flags & MAPF_FLAG_ACTION_MASK) {
        bool up = false, rt = false, dn = false, lf = false;
        if (sr > 0) {
            up = !oa.get(ctr - V);
            rt = !oa.get(ctr + 1);
            dn = !oa.get(ctr + V);
            lf = !oa.get(ctr - 1);
        }
        q[0] = 1.0f;
        q[1] = up ? 1.0f : 0.0f;
        q[2] = rt ? 1.0f : 0.0f;
        q[3] = dn ? 1.0f : 0.0f;
        q[4] = lf ? 1.0f : 0.0f;
    }
}
template <class K, int MW, int MAXV>
__device__ __forceinline__ void emit_obs_row(const Params &p, float *srow, const WMask<MW> &obst, const WMask<MW> &agm,
                                             const WMask<MW> &goals, const WMask<MW> &own, float gd_r, float gd_c,
                                             bool pressure) {
    const WMask<MW> oa = obst | agm;
    const WMask<MW> g3 = own.andnot(oa);
    const WMask<MW> g4 = goals.andnot(oa | own);
    const WMask<MW> bit0 = obst | g3;
    const WMask<MW> bit1 = agm.andnot(obst) | g3;
    emit_obs_planes<K, MW, MAXV>(p, srow, bit0, bit1, g4, oa, gd_r, gd_c, pressure);
}

struct PairOut {
    uint64_t nbr;   // agents within lock_nearby_manhattan (final positions), self excluded
    int sum_delta;  // sum over {self} U nbr of (distance at window start - distance now)
    bool blocks;    // some not-yet-reached agent intended to enter my cell
};

// MODE: kObsEmit = observation only (reset / observe kernels, and the observation wave of a step),
//       kObsBoth = observation + pair outputs in one walk, kObsPairs = pair outputs only (the state wave of a step).
constexpr int kObsEmit = 0, kObsBoth = 1, kObsPairs = 2;
template <class K, int LPE, int MW, int MODE, bool USE_MAP = false>
__device__ __forceinline__ void observe(const Params &p, const Io &io, const uint64_t *lrows, const uint4 *tabg, float *srow,
                                        bool is_agent, int a, uint32_t cur, uint32_t goal, bool final_state,
                                        bool pressure, int my_delta, PairOut &po, const uint32_t *map = nullptr,
                                        const float *gd_lut = nullptr) {
    constexpr bool FULL = MODE != kObsEmit, EMIT = MODE != kObsPairs;
    constexpr int MAXV = MW == 32 ? 5 : (MW == 64 ? 7 : 11);
    constexpr int C = LPE < 8 ? LPE : 8;
    const int N = K::N(p), V = K::V(p), sr = K::sr(p);
    const uint32_t flags = K::flags(p);
    const int myr = (int)(cur >> 8), myc = (int)(cur & 255u);
    const int r0 = myr - sr, c0 = myc - sr;
    const uint32_t mycell1 = cur + 0x0101u;  // (row+1)<<8 | (col+1): the encoding of intended cells

    WMask<MW> obst, agm, goals;
    obst.clear(); agm.clear(); goals.clear();

    // lrows points at the env's row 0; kRowPad sentinel rows sit on either side, so no bounds checks
    uint64_t rows[MAXV];
    const int rbase = is_agent ? r0 : 0;
    if (EMIT) {
#pragma unroll
        for (int d = 0; d < MAXV; d++) rows[d] = (d < V) ? lrows[rbase + d] : ~0ull;
    }

    uint32_t nbr_lo = 0, nbr_hi = 0;
    int sum_biased = 0;
    bool blocks = false;
    if (USE_MAP) {
        // ---- large N: read my window and my lock neighbourhood from the env's cell map ----
        const int map_w = io.W + 2 * kRowPad;
        const uint32_t me1 = (uint32_t)a + 1u;
        // occupant at "time a": agents <= a at their new cell, agents > a at their old one.  As two thresholds (no
        // per-cell branch on final_state): a new-cell owner o counts if 1 <= o < new_lim, i.e. (o - 1) < (new_lim - 1)
        // unsigned, an old-cell owner if > old_lim.  My own cell (the centre) is cleared afterwards.
        const uint32_t new_lim1 = final_state ? 127u : me1 - 1u;
        const uint32_t old_lim = final_state ? 127u : me1;
        const uint32_t *win = map + (is_agent ? (r0 + kRowPad) * map_w + (c0 + kRowPad) : 0);
        for (int d = 0; EMIT && d < V; d++) {
            uint32_t wrow[MAXV];  // one window row per round trip: the V reads are issued back to back
#pragma unroll
            for (int e = 0; e < MAXV; e++) wrow[e] = (e < V) ? win[d * map_w + e] : 0u;
#pragma unroll
            for (int e = 0; e < MAXV; e++) {
                if (e < V) {
                    const uint32_t w = wrow[e];
                    const uint32_t on = w & 127u, oo = (w >> 7) & 127u, go = (w >> 14) & 127u;
                    const bool occ = (on - 1u) < new_lim1 || oo > old_lim;
                    agm.set_if(occ, d * V + e);
                    goals.set_if(go != 0, d * V + e);
                }
            }
        }
        if (FULL) {
            const int nb = K::nearby(p);
            const uint32_t *ctr = map + (is_agent ? map_index(cur, map_w) : kRowPad * map_w + kRowPad);
            auto visit = [&](const uint32_t w) {  // w = cell word of a neighbourhood cell (centre excluded by the caller)
                const uint32_t on = w & 127u;
                const bool isn = on != 0;
                const int j = (int)on - 1;
                if (LPE <= 32 || j < 32) nbr_lo |= isn ? (1u << (j & 31)) : 0u;
                if (LPE > 32 && j >= 32) nbr_hi |= isn ? (1u << (j & 31)) : 0u;
                sum_biased += isn ? (int)(w >> 22) : 0;
            };
            if (K::kFixed) {
                // compile-time radius: the diamond is fully unrolled and pruned, all its reads are issued together
#pragma unroll
                for (int dr = -kRowPad; dr <= kRowPad; dr++) {
#pragma unroll
                    for (int dc = -kRowPad; dc <= kRowPad; dc++) {
                        if (abs(dr) + abs(dc) <= nb && (dr | dc) != 0) visit(ctr[dr * map_w + dc]);
                    }
                }
            } else {
                for (int dr = -nb; dr <= nb; dr++) {
                    const int span = nb - abs(dr);
                    for (int dc = -span; dc <= span; dc++) {
                        if ((dr | dc) != 0) visit(ctr[dr * map_w + dc]);
                    }
                }
            }
            blocks = (ctr[0] >> 21) & 1u;
        }
    } else
    for (int j0 = 0; j0 < N; j0 += C) {
        uint4 e[C];
#pragma unroll
        for (int u = 0; u < C; u++) e[u] = tabg[j0 + u];
#pragma unroll
        for (int u = 0; u < C; u++) {
            const int j = j0 + u;
            const uint32_t Aj = e[u].x, Bj = e[u].y;
            const uint32_t newj = Aj >> 16;
            if (EMIT) {
                const uint32_t pj = (final_state || j <= a) ? newj : (Aj & 0xFFFFu);
                // occupancy (self included: it sets the centre bit, cleared below) and goals (own included)
                const int pr = (int)(pj >> 8) - r0, pc = (int)(pj & 255u) - c0;
                agm.set_if(max((unsigned)pr, (unsigned)pc) < (unsigned)V, __mul24(pr, V) + pc);
                const int gr = (int)((Bj >> 8) & 255u) - r0, gc = (int)(Bj & 255u) - c0;
                goals.set_if(max((unsigned)gr, (unsigned)gc) < (unsigned)V, __mul24(gr, V) + gc);
            }
            if (FULL) {
                const int d = cell_l1(newj, cur);
                const bool isn = (unsigned)(d - 1) < (unsigned)K::nearby(p);  // 0 < d <= nearby
                if (LPE <= 32 || j < 32) nbr_lo |= isn ? (1u << (j & 31)) : 0u;
                else nbr_hi |= isn ? (1u << (j & 31)) : 0u;
                sum_biased += isn ? (int)((Bj >> 17) & 1023u) : 0;
                blocks |= e[u].z == mycell1;
            }
        }
    }
    if (FULL) {
        po.nbr = (uint64_t)nbr_lo | ((uint64_t)nbr_hi << 32);
        po.sum_delta = my_delta + sum_biased - 256 * (__popc(nbr_lo) + __popc(nbr_hi));
        po.blocks = blocks;
    }
    if (!EMIT || !is_agent) return;

    obst = window_obstacles<MW, MAXV>(rows, c0, V, io.col_pad);
    agm.clear_bit(sr * V + sr);  // my own cell: "occ not in (UNASSIGNED, self)" MA-env:735
    WMask<MW> own;
    own.clear();
    window_set<MW>(own, goal, r0, c0, V);
    const bool norm = (flags & MAPF_FLAG_NORMALIZE_GOAL_DELTA) != 0;
    const int gdr = (int)((goal >> 8) & 255u) - myr, gdc = (int)(goal & 255u) - myc;
    // gd_lut (k_step3's observation wave): the quotients of every possible delta, computed -- with the same correctly
    // rounded divide -- while the wave waited for the moves; two LDS reads instead of two divide sequences behind B1
    const float gd_r = gd_lut ? gd_lut[gdr + 63] : goal_delta(gdr, io.den_r, norm);
    const float gd_c = gd_lut ? gd_lut[128 + gdc + 63] : goal_delta(gdc, io.den_c, norm);
    emit_obs_row<K, MW, MAXV>(p, srow, obst, agm, goals, own, gd_r, gd_c, pressure);
}

// copy the wave's staged observations to global memory.  sel (per lane, uniform inside a group):
// 0 -> io.obs, 1 -> io.final_obs, 2 -> skip.  Flat 16-byte stores when every valid group goes to
// the same tensor, otherwise one contiguous run per group.
// Observation stream stores.  The 8.6 MB of observations a launch writes are not re-read by this engine, so
// they are pushed towards HBM while the kernel runs (write-through) instead of sitting dirty in L2 until the
// end-of-kernel write-back (measured against plain and nontemporal stores: DESIGN.md section 5).  The sc1 form is a
// buffer store with the cache-policy bit (aux 16), a store hipcc can see: an inline-asm store reads its data registers
// asynchronously and the hazard recognizer does not protect them (that variant corrupted 16 floats per wave in the
// L = 28 specialisation).
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
struct ObsSink {
    __amdgpu_buffer_rsrc_t rsrc;
    float *base;
};
__device__ __forceinline__ ObsSink make_obs_sink(float *base, unsigned bytes) {
    ObsSink s;
    s.base = base;
    s.rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);  // wave-uniform operands only
    return s;
}
// store 16 bytes at byte offset `off` of the sink
__device__ __forceinline__ void store_obs4(const ObsSink &s, unsigned off, const float4 v) {
    const v4u_t w = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(w, s.rsrc, (int)off, 0, 16);  // aux 16 = sc1
}

// Write-through stores of <wave-uniform base> + <lane index> for the OTHER outputs nobody on the device reads again
// (rewards, per-agent info, info rows; k_step3's aux wave): like the observation stream they leave L2 while the kernel
// runs instead of at its end.  Measured per class (tools/ab_inproc.py, in phase / staggered): rewards alone -1.15 % /
// -1.2 %; the STATE planes written through: +1.0 .. +1.5 % / +0.2 .. +0.7 % -- the next launch reads them back.
typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
template <int BYTES>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, 64 * BYTES, 0x00020000);  // (64 lanes; wave-uniform operands only)
}
__device__ __forceinline__ void store_wt8(void *base, int lane, const uint2 v) {
    const v2u_t w = {v.x, v.y};
    __builtin_amdgcn_raw_buffer_store_b64(w, wt_rsrc<8>(base), lane * 8, 0, 16);  // aux 16 = sc1
}
__device__ __forceinline__ void store_wt4(void *base, int lane, const uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, wt_rsrc<4>(base), lane * 4, 0, 16);
}
__device__ __forceinline__ void store_wt2(void *base, int lane, const uint16_t v) {
    __builtin_amdgcn_raw_buffer_store_b16(v, wt_rsrc<2>(base), lane * 2, 0, 16);
}

// full wave, one destination, compile-time shape: straight-line 16-byte copies
template <class K, int LPE>
__device__ __forceinline__ void flush_obs_full(const Params &p, const Io &io, float *flat, const float *stage, int lane,
                                               int env0) {
    constexpr int G = 64 / LPE;
    const int NL = K::N(p) * K::L(p);
    const int n = G * NL;
    float *dst = flat + (size_t)env0 * NL;
    if ((n & 3) == 0) {
        const ObsSink sink = make_obs_sink(dst, (unsigned)n * 4u);  // wave-relative: offsets stay small
        const unsigned off0 = (unsigned)lane * 16u;
        const int n4 = n >> 2;
        const float4 *s4 = reinterpret_cast<const float4 *>(stage);
        if (K::kFixed) {
            const int full = n4 >> 6;  // rounds in which all 64 lanes copy (a compile-time constant here: unrolled whole.
                                       // It used to be capped at 16 rounds = 4 096 floats per wave, which every prebuilt
                                       // shape fits; 64 agents with 9x9 windows, compiled at run time, do not)
#pragma unroll
            for (int r = 0; r < full; r++) store_obs4(sink, off0 + (unsigned)r * 1024u, s4[r * 64 + lane]);
            if ((full << 6) + lane < n4) store_obs4(sink, off0 + (unsigned)full * 1024u, s4[(full << 6) + lane]);
        } else {
            for (int k = lane; k < n4; k += 64) store_obs4(sink, off0 + (unsigned)(k - lane) * 16u, s4[k]);
        }
    } else {
        for (int k = lane; k < n; k += 64) dst[k] = stage[k];
    }
}

template <class K, int LPE>
__device__ __forceinline__ void flush_rows(const Io &io, const float *stage, int lane, int env0, int ngroups, int sel,
                                           const int NL);

template <class K, int LPE>
__device__ __forceinline__ void flush_obs(const Params &p, const Io &io, const float *stage, int lane, int env0,
                                          int ngroups, int sel) {
    flush_rows<K, LPE>(io, stage, lane, env0, ngroups, sel, K::N(p) * K::L(p));
}

// NL = floats per env
template <class K, int LPE>
__device__ __forceinline__ void flush_rows(const Io &io, const float *stage, int lane, int env0, int ngroups, int sel,
                                           const int NL) {
    constexpr int G = 64 / LPE;
    const uint64_t valid = __ballot((lane / LPE) < ngroups);
    const uint64_t m0 = __ballot((lane / LPE) < ngroups && sel == 0);
    const uint64_t m1 = __ballot((lane / LPE) < ngroups && sel == 1);
    if ((m0 | m1) == 0) return;
    if (m0 == valid || m1 == valid) {
        float *flat = m0 == valid ? io.obs : io.final_obs;
        if (!flat) return;
        const int n = ngroups * NL;
        float *dst = flat + (size_t)env0 * NL;
        if (((G * NL) & 3) == 0) {
            const int n4 = n >> 2;
            const float4 *s4 = reinterpret_cast<const float4 *>(stage);
            // write-through stream like flush_obs_full; the sink is addressed from the wave's own first byte so that
            // the 32-bit buffer offsets stay small whatever the size of the tensor
            const ObsSink sink = make_obs_sink(dst, (unsigned)n * 4u);
            // rounds of 4 x 1 KiB: the four LDS reads are issued back to back, then the four stores
            for (int k0 = 0; k0 < n4; k0 += 256) {
                // unconditional (clamped) reads keep v[] in registers; only the stores are predicated
                const float4 v0 = s4[min(k0 + lane, n4 - 1)];
                const float4 v1 = s4[min(k0 + 64 + lane, n4 - 1)];
                const float4 v2 = s4[min(k0 + 128 + lane, n4 - 1)];
                const float4 v3 = s4[min(k0 + 192 + lane, n4 - 1)];
                const unsigned off = (unsigned)(k0 + lane) * 16u;
                if (k0 + lane < n4) store_obs4(sink, off, v0);
                if (k0 + 64 + lane < n4) store_obs4(sink, off + 1024u, v1);
                if (k0 + 128 + lane < n4) store_obs4(sink, off + 2048u, v2);
                if (k0 + 192 + lane < n4) store_obs4(sink, off + 3072u, v3);
            }
            for (int k = (n4 << 2) + lane; k < n; k += 64) dst[k] = stage[k];
        } else {
            for (int k = lane; k < n; k += 64) dst[k] = stage[k];
        }
        return;
    }
    // mixed destinations (typically: one env of the wave resets): one run per group, 16 bytes per lane when the rows
    // are 16-byte multiples (then every row start is aligned as well)
    for (int g = 0; g < ngroups; g++) {
        const int sg = __shfl(sel, g * LPE, 64);
        float *base = sg == 0 ? io.obs : (sg == 1 ? io.final_obs : nullptr);
        if (!base) continue;
        float *dst = base + (size_t)(env0 + g) * NL;
        const float *src = stage + g * NL;
        if ((NL & 3) == 0) {
            const ObsSink sink = make_obs_sink(dst, (unsigned)NL * 4u);
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            for (int k = lane; k < (NL >> 2); k += 64) store_obs4(sink, (unsigned)k * 16u, s4[k]);
        } else {
            for (int k = lane; k < NL; k += 64) dst[k] = src[k];
        }
    }
}

// pair-table entry for a static state (reset / observe kernels): nothing moves, nobody intends anything
__device__ __forceinline__ uint4 static_entry(uint32_t pos, uint32_t goal) {
    return make_uint4(pos | (pos << 16), goal | (256u << 17), 0xFFFFFFFFu, 0u);
}

// ------------------------------------------------------------------------------------------------
// generate_starts_goals (MA-env:267-282), idx = rng.choice(F, 2N, replace=False), spread over the lanes of the group.
// NumPy's algorithm is sequential -- Floyd's sampling (2N bounded draws, "already chosen -> take j instead") and a
// tail shuffle (2N - 1 bounded draws) on ONE PCG64 stream -- and a wave that resets an env sets the duration of the
// launch for everybody.  What is sequential about it is small, though:
//   * the raw 64-bit outputs only depend on the stream: lane q computes state_q by jump-ahead (kPcgJump*);
//   * every draw's bound is known up front (Floyd: j = F - 2N + k, shuffle: i = 2N - 1, ..., 1), so all Lemire
//     products are formed side by side; a rejection (probability < 2N / 2^32 per draw) or F = 2N (a draw with bound 0
//     consumes nothing) sends the group to the sequential restatement instead;
//   * Floyd's membership test is a ballot over the lanes that hold the chosen values (no hash set);
//   * the shuffle swaps run on the LDS copy.
// Scratch of the group: raw32[4N + 2] | vals[4N] (uint16) | out[2N] (int16).  Returns false for a group that has to
// take the sequential path; on true, out (at scratch + kSampleOutOff(N)) holds idx and the stream state after the
// draws is stored to Params::rng[env].  The two halves below are separate functions because the background sampler
// runs them in different launches (sampler_wave).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sample_out_off_i16(int N) { return 2 * (4 * N + 2) + 4 * N; }

struct PcgPre {  // fetched ahead by the caller (background sampler): stream state, free-cell count and this lane's
                 // jump-ahead constants A[a+1], S[a+1]; or nothing
    bool have;
    Pcg g;
    int pop;
    uint4 ja, js;  // {hi.lo32, hi.hi32, lo.lo32, lo.hi32} as stored in kPcgJumpA / kPcgJumpS
    bool have_cq = false;  // sliced draw: `ja` is A_q of the ONE output this lane computes in this pass and `cq` the env's
    uint4 cq = {0, 0, 0, 0};  // S_q * inc from Io::jump_c (same word order): one 128-bit product per output instead of two
    bool have_j = true;  // false: stream and count only (an inline reset whose caller holds the stream in registers)
};
// First half: the raw outputs (jump-ahead) and all bounded draws -> vals[4N - 1] in the group's scratch.  Returns
// whether the lane-parallel draw stands (else: sequential restatement); then the stream after the draws is in
// Params::rng[env] and, with vis_dst, the stream before them in vis_dst[env].  rng_src: where the stream is read from.
template <int LPE>
__device__ __forceinline__ bool draw_stage_a(const Params &p, int16_t *scr, int lane, int a, int env, bool env_ok,
                                             bool do_reset, int N, uint64_t *vis_dst, const PcgPre &pre,
                                             const uint64_t *rng_src, int &pop_out, const int sw_row = -1,
                                             const int mode = 0, uint64_t *rng_dst = nullptr, const int qpass = -1,
                                             uint32_t *raw_dst = nullptr) {
    // qpass (mode 1 only): -1 every output; 0 the lane's first output only (q = a + 1); 1 its second only
    // (q = a + 1 + LPE, straight from the jump tables; this pass holds the last output and stores the streams).
    // raw_dst: the outputs also go to this (global) array.
    if (!rng_dst) rng_dst = p.rng;  // (the sliced draw passes the same array as a global-address-space pointer)
    // mode 0: both parts; 1: outputs only (raw[] left in the scratch, the streams are stored right away: a later
    // rejection is then resolved from vis_dst); 2: bounded draws only (raw[] already in the scratch, pop in pre.pop)
    (void)sw_row;
#ifdef MAPF_STAMPS
#define MAPF_STAMP_A(k) do { if (sw_row >= 0) { MAPF_STAMP_SW(k); } } while (0)
#else
#define MAPF_STAMP_A(k) do { } while (0)
#endif
    const int size = 2 * N, D = 2 * size - 1;  // bounded draws of one reset
    uint32_t *raw = reinterpret_cast<uint32_t *>(scr);
    uint16_t *vals = reinterpret_cast<uint16_t *>(raw + 4 * N + 2);
    Pcg g;
    g.shi = g.slo = g.ihi = g.ilo = 0;
    g.has32 = g.uinteger = 0;
    int pop = size + 1;
    if (pre.have) {
        if (do_reset) {
            g = pre.g;
            pop = pre.pop;
        }
    } else if (do_reset) {
        pcg_load(g, rng_src + (size_t)env * 6);
        pop = p.n_free[env];
    }
    pop_out = pop;
    bool ok = do_reset && pop > size && !(p.flags & MAPF_FLAG_SEQUENTIAL_RESET);
    const int has = (int)g.has32;
    const int nout = (D - has + 1) >> 1;  // 64-bit outputs consumed; the stream of 32-bit halves is
                                          // [buffered half if has] lo(o1) hi(o1) lo(o2) hi(o2) ...
    const U128 s0 = {g.shi, g.slo}, inc = {g.ihi, g.ilo};
    const U128 stride_a = {kPcgJumpA[LPE][0], kPcgJumpA[LPE][1]};
    const U128 stride_c = mul128(U128{kPcgJumpS[LPE][0], kPcgJumpS[LPE][1]}, inc);
    U128 st = s0, fin = s0;
    uint32_t fin_hi32 = 0;
    for (int q = a + 1 + (qpass == 1 ? LPE : 0); mode != 2 && q <= nout; q += LPE) {
        if (pre.have_cq) {
            const U128 ja = {(uint64_t)pre.ja.x | ((uint64_t)pre.ja.y << 32), (uint64_t)pre.ja.z | ((uint64_t)pre.ja.w << 32)};
            const U128 cq = {(uint64_t)pre.cq.x | ((uint64_t)pre.cq.y << 32), (uint64_t)pre.cq.z | ((uint64_t)pre.cq.w << 32)};
            st = add128(mul128(ja, s0), cq);
        } else if (q == a + 1 || qpass == 1) {
            U128 ja = {kPcgJumpA[q][0], kPcgJumpA[q][1]}, js = {kPcgJumpS[q][0], kPcgJumpS[q][1]};
            if (pre.have && pre.have_j && qpass != 1) {
                ja = U128{(uint64_t)pre.ja.x | ((uint64_t)pre.ja.y << 32), (uint64_t)pre.ja.z | ((uint64_t)pre.ja.w << 32)};
                js = U128{(uint64_t)pre.js.x | ((uint64_t)pre.js.y << 32), (uint64_t)pre.js.z | ((uint64_t)pre.js.w << 32)};
            }
            st = add128(mul128(ja, s0), mul128(js, inc));
        } else {
            st = add128(mul128(stride_a, st), stride_c);
        }
        const uint64_t o = pcg_output(st);
        MAPF_CHK(p, has + 2 * (q - 1) + 1 < 4 * N + 2, 1, env, has + 2 * (q - 1) + 1);
        raw[has + 2 * (q - 1)] = (uint32_t)o;
        raw[has + 2 * (q - 1) + 1] = (uint32_t)(o >> 32);
        if (raw_dst && do_reset) {
            raw_dst[has + 2 * (q - 1)] = (uint32_t)o;
            raw_dst[has + 2 * (q - 1) + 1] = (uint32_t)(o >> 32);
        }
        if (q == nout) {
            fin = st;
            fin_hi32 = (uint32_t)(o >> 32);
        }
        if (qpass == 0) break;
    }
    if (mode != 2 && has && a == 0) {
        raw[0] = g.uinteger;
        if (raw_dst && do_reset && qpass != 1) raw_dst[0] = g.uinteger;
    }
    const bool store_streams_now = mode == 1 && ok && qpass != 0;
    if (store_streams_now && env_ok && ((nout - 1) % LPE) == a) {
        Pcg f;
        f.shi = fin.hi; f.slo = fin.lo; f.ihi = g.ihi; f.ilo = g.ilo;
        f.has32 = (uint32_t)(has + 2 * nout - D);
        f.uinteger = fin_hi32;
        if (vis_dst) pcg_store(g, vis_dst + (size_t)env * 6);
        pcg_store(f, rng_dst + (size_t)env * 6);
    }
    wave_lds_sync();
    if (mode == 1) return ok;
    MAPF_STAMP_A(6);
    // all bounded draws at once: vals[k] = (raw[k] * (bound_k + 1)) >> 32 unless Lemire might reject.  Lemire rejects
    // when the low half `left` is below 2^32 mod excl, which is itself below excl: `left < excl` (probability
    // excl / 2^32 < 1e-6) is taken as "might" and sends the group to the sequential restatement, which decides exactly;
    // no modulo here (the compiler evaluated it unconditionally: 3 k cycles of a sampler wave's 7 k).
    bool rej = false;
    for (int k = a; k < D; k += LPE) {
        const uint32_t rng = (uint32_t)(k < size ? pop - size + k : size - 1 - (k - size));
        const uint32_t excl = rng + 1u;
        const uint64_t m = (uint64_t)raw[k] * excl;
        rej |= (uint32_t)m < excl;
        MAPF_CHK(p, k < 4 * N, 2, env, k);
        vals[k] = (uint16_t)(m >> 32);
    }
    ok = ok && gballot<LPE>(rej, lane) == 0;
    MAPF_STAMP_A(7);
    // the stream after D draws: state_nout; a half is left in the buffer when the number of halves used is odd
    if (mode == 0 && ok && env_ok && ((nout - 1) % LPE) == a) {
        Pcg f;
        f.shi = fin.hi; f.slo = fin.lo; f.ihi = g.ihi; f.ilo = g.ilo;
        f.has32 = (uint32_t)(has + 2 * nout - D);
        f.uinteger = fin_hi32;  // NumPy keeps the last high half in the buffer field even once it has been handed out
        if (vis_dst) pcg_store(g, vis_dst + (size_t)env * 6);
        pcg_store(f, p.rng + (size_t)env * 6);
    }
    wave_lds_sync();
    MAPF_STAMP_A(8);
    return ok;
}

// Floyd's sampling for at most 16 values (N <= 8): lane t ends up with the chosen values number t (c0) and t + LPE (c1).
// The drawn values live in registers (static indices: the loops are unrolled to 16 and predicated), so the dependent
// chain of an iteration is compare -> ballot -> select with no LDS access in it.
// half: -1 = all iterations (c0, c1 start empty); 0 = iterations k < LPE (fills c0); 1 = iterations k >= LPE (c0 is
// given, fills c1).
template <int LPE>
__device__ __forceinline__ void draw_floyd16(int16_t *scr, int lane, int a, int N, int pop, int &c0, int &c1,
                                             const int half = -1) {
    const int size = 2 * N;
    const uint32_t *v32 = reinterpret_cast<const uint32_t *>(scr) + 4 * N + 2;  // vals[] as packed pairs
    uint32_t vw[8];
#pragma unroll
    for (int q = 0; q < 8; q++) vw[q] = v32[q];
    if (half != 1) c0 = -1;
    c1 = -1;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < size && (half < 0 || (half == 0) == (k < LPE))) {
            const int val = (int)((vw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu), j = pop - size + k;
            const bool hit = (a < k && c0 == val) || (a + LPE < k && c1 == val);
            const int chosen = gballot<LPE>(hit, lane) != 0 ? j : val;
            if (a == (k & (LPE - 1))) {
                if (k < LPE) c0 = chosen;
                else c1 = chosen;
            }
        }
    }
}
// _shuffle_int tail shuffle with the precomputed indices vals[2N ..): the permutation fits one 64-bit register as
// nibbles; every lane applies the swaps to it and then places its chosen values in out[] (group scratch).
template <int LPE>
__device__ __forceinline__ void draw_shuffle16(const Params &p, int env, int16_t *scr, int lane, int a, int N, int c0, int c1,
                                               bool on) {
    (void)lane; (void)p; (void)env;
    // `on`: groups that are not drawing run along on whatever their scratch holds; their swap indices are junk, the
    // nibbles no bijection and inv[] partly unwritten, so they must not store (out[inv[..]] would land in another
    // group's scratch)
    const int size = 2 * N;
    const uint16_t *vals = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint32_t *>(scr) + 4 * N + 2);
    const uint32_t *j32 = reinterpret_cast<const uint32_t *>(vals + size);  // size is even: 4-byte aligned
    int16_t *out = scr + sample_out_off_i16(N);
    uint32_t jw[8];
#pragma unroll
    for (int q = 0; q < 8; q++) jw[q] = j32[q];
    uint64_t perm = 0xFEDCBA9876543210ull;  // nibble x = which chosen value ends up at position x
#pragma unroll
    for (int t = 0; t < 15; t++) {
        if (t < size - 1) {
            const int i = size - 1 - t;
            const int j = (int)((jw[t >> 1] >> (16 * (t & 1))) & 0xFFFFu);
            const uint64_t d = ((perm >> (4 * i)) ^ (perm >> (4 * j))) & 15ull;
            perm ^= (d << (4 * i)) | (d << (4 * j));
        }
    }
    // chosen value t sits at the position x with nibble x == t: scatter through LDS, then every position is read
    uint8_t *inv = reinterpret_cast<uint8_t *>(out + size);  // [16] position of chosen value t
    if (on)
        for (int x = a; x < 16; x += LPE) inv[(perm >> (4 * x)) & 15ull] = (uint8_t)x;  // all 16 nibbles: a bijection
    wave_lds_sync();
    if (on && a < size) {
        MAPF_CHK(p, (inv[a] & 15) < size, 3, env, inv[a]);
        out[inv[a] & 15] = (int16_t)c0;
    }
    if (on && a + LPE < size) {
        MAPF_CHK(p, (inv[a + LPE] & 15) < size, 3, env, inv[a + LPE]);
        out[inv[a + LPE] & 15] = (int16_t)c1;
    }
}

// Floyd's sampling and the tail shuffle for lists of more than LPE (up to 2 LPE) values WITHOUT a loop over the elements:
// element k and k + LPE live in lane k of the group; all masks are the group's (gballot).  NumPy writes both as 2N
// dependent iterations (40 k cycles of one wave at N = 64, during which the other 1 023 wait for the launch to end).
//
// ---- Floyd.  chosen_k = j_k = base + k if val_k is in the set when its turn comes, else val_k.  A value v is in the
//      set at time k iff it was DRAWN before (val_s == v, s < k: whoever drew it first put it there, unless it already
//      was) or it is the j of an earlier element that collided (v = j_u, u < k, element u took j_u); nothing else ever
//      enters the set.  So
//          coll_k = dup_k | (u_k < k & coll[u_k]),   u_k = val_k - base,
//      with dup_k from equality masks built by ballots over the 12 value bits (cells of a 64 x 64 grid), and the second
//      term a chain through LOWER indices that only exists for values in the j range (2N / F of them): iterated to its
//      fixed point, one or two rounds of two ballots in practice.
template <int LPE>
__device__ __forceinline__ void draw_floyd_par(const uint16_t *vals, int lane, int a, int size, int pop, int &c0, int &c1) {
    const uint64_t gm = group_mask<LPE>();
    const int base = pop - size;
    const bool e1 = a + LPE < size;
    const int v0 = (int)vals[min(a, size - 1)], v1 = (int)vals[min(a + LPE, size - 1)];
    const uint64_t below = (1ull << a) - 1ull;
    uint64_t q00 = gm, q10 = gm, q11 = gballot<LPE>(e1, lane);  // elements of set s equal to my v_x: q<x><s>
#pragma unroll 1  // (rare path inside the step kernels: unrolled, the two loops are 12 KB of a 70 KB kernel and every step
                  // of c5 pays 2 % for them in instruction-cache misses)
    for (int b = 0; b < 12; b++) {
        const bool x0 = (v0 >> b) & 1, x1 = (v1 >> b) & 1;
        const uint64_t B0 = gballot<LPE>(x0, lane), B1 = gballot<LPE>(x1, lane);
        q00 &= x0 ? B0 : ~B0;
        q10 &= x1 ? B0 : ~B0;
        q11 &= x1 ? B1 : ~B1;
    }
    const bool dup0 = (q00 & below) != 0, dup1 = e1 && (q10 != 0 || (q11 & below) != 0);
    const int u0 = v0 - base, u1 = v1 - base;
    const bool r0 = u0 >= 0 && u0 < a, r1 = e1 && u1 >= 0 && u1 < a + LPE;
    bool k0 = dup0, k1 = dup1;
    for (int it = 0; it < size; it++) {  // (the chain is shorter than the list: the bound is structural)
        const uint64_t C0 = gballot<LPE>(k0, lane), C1 = gballot<LPE>(k1, lane);
        const bool n0 = dup0 || (r0 && ((C0 >> (u0 & (LPE - 1))) & 1ull));
        const bool n1 = dup1 || (r1 && (((u1 < LPE ? C0 : C1) >> (u1 & (LPE - 1))) & 1ull));
        const bool changed = n0 != k0 || n1 != k1;
        k0 = n0;
        k1 = n1;
        if (!__any(changed)) break;
    }
    c0 = k0 ? base + a : v0;
    c1 = k1 ? base + a + LPE : v1;
}
// ---- tail shuffle: for i = size-1 .. 1: swap(idx[i], idx[J_i]), J_i <= i.  Position i is final after step i and receives
//      what position J_i holds then.  A position q <= i holds, before step i, what the NEXT step after i aiming at q put
//      there (the steps run downwards: that is the latest write), else its original element; and what step x puts
//      somewhere is what position x held before step x, i.e. what the first step above x aiming at x put there -- up(x) --
//      and so on: a forest of pointers to higher indices.  final[i] = orig[root(nx(i))], nx(i) = the next step above i
//      with the same aim (or orig[J_i] if there is none).  nx and up are lowest-set-bit queries on equality masks over
//      the bits of J; the roots come from pointer doubling on a 2 LPE-byte table (<= log2(2 LPE) rounds, 2-4 in practice).
//      c0 / c1: the chosen values number a and a + LPE; raw: the group's scratch (its raw outputs were consumed by
//      stage a); out[2N] receives idx.  (A group that is not drawing runs along on junk: its stores are guarded by ok.)
template <int LPE>
__device__ __forceinline__ void draw_shuffle_par(const Params &p, int env, uint32_t *raw, const uint16_t *vals, int16_t *out,
                                                 int lane, int a, int size, int c0, int c1, bool ok) {
    (void)p; (void)env;
    constexpr int kBits = LPE == 64 ? 7 : (LPE == 32 ? 6 : (LPE == 16 ? 5 : (LPE == 8 ? 4 : 3)));
    const uint64_t gm = group_mask<LPE>();
    const bool e1 = a + LPE < size;
    const uint64_t above = (a == LPE - 1) ? 0ull : ((~0ull << (a + 1)) & gm);
    const bool s0 = a >= 1, s1 = e1;  // element i is a step (i = 0 is not: J_0 := 0)
    const int J0 = s0 ? (int)vals[size + size - 1 - a] & (2 * LPE - 1) : 0;
    const int J1 = s1 ? (int)vals[size + size - 1 - min(a + LPE, size - 1)] & (2 * LPE - 1) : 0;
    const uint64_t V0 = gm & ~1ull, V1 = gballot<LPE>(s1, lane);
    uint64_t n00 = V0, n01 = V1, n11 = V1;  // steps of set s aiming where element x aims: n<x><s>
    uint64_t w00 = V0, w01 = V1, w11 = V1;  // steps of set s aiming AT element x (index a / a + LPE): w<x><s>
#pragma unroll 1
    for (int b = 0; b < kBits; b++) {
        const bool y0 = (J0 >> b) & 1, y1 = (J1 >> b) & 1;
        const uint64_t B0 = gballot<LPE>(y0, lane), B1 = gballot<LPE>(y1, lane);
        const bool i0 = (a >> b) & 1, i1 = ((a + LPE) >> b) & 1;  // bits of my indices a and a + LPE
        n00 &= y0 ? B0 : ~B0;
        n01 &= y0 ? B1 : ~B1;
        n11 &= y1 ? B1 : ~B1;
        w00 &= i0 ? B0 : ~B0;
        w01 &= i0 ? B1 : ~B1;
        w11 &= i1 ? B1 : ~B1;
    }
    auto lowest = [&](uint64_t m0, uint64_t m1, int none) -> int {  // lowest step in (set 0 | set 1), else `none`
        return m0 ? (int)__builtin_ctzll(m0) : (m1 ? LPE + (int)__builtin_ctzll(m1) : none);
    };
    const int nx0 = lowest(n00 & above, n01, -1), nx1 = lowest(0ull, n11 & above, -1);
    int p0 = lowest(w00 & above, w01, a), p1 = lowest(0ull, w11 & above, a + LPE);  // up(x), or x itself: a root
    uint8_t *ptr = reinterpret_cast<uint8_t *>(raw);
    int16_t *chosen = reinterpret_cast<int16_t *>(raw) + LPE;  // [2 LPE] behind the 2 LPE pointer bytes
    chosen[a] = (int16_t)c0;
    chosen[a + LPE] = (int16_t)c1;
    for (int it = 0; it <= kBits; it++) {
        ptr[a] = (uint8_t)p0;
        ptr[a + LPE] = (uint8_t)p1;
        wave_lds_sync();
        const int t0 = ptr[p0], t1 = ptr[p1];
        const bool changed = t0 != p0 || t1 != p1;
        p0 = t0;
        p1 = t1;
        wave_lds_sync();
        if (!__any(changed)) break;
    }
    const int g0 = nx0 >= 0 ? (int)ptr[nx0] : J0, g1 = nx1 >= 0 ? (int)ptr[nx1] : J1;
    MAPF_CHK(p, !ok || (unsigned)g0 < (unsigned)size, 4, env, g0);
    MAPF_CHK(p, !(ok && e1) || (unsigned)g1 < (unsigned)size, 4, env, g1);
    if (ok) out[a] = chosen[g0];
    if (ok && e1) out[a + LPE] = chosen[g1];
}

// Second half: Floyd's sampling and the tail shuffle on vals[] (group scratch) -> out[2N] (group scratch).
template <int LPE>
__device__ __forceinline__ void draw_stage_b(const Params &p, int env, int16_t *scr, int lane, int a, bool ok, int N, int pop) {
    (void)p; (void)env;
    const int size = 2 * N;
    uint32_t *raw = reinterpret_cast<uint32_t *>(scr);
    uint16_t *vals = reinterpret_cast<uint16_t *>(raw + 4 * N + 2);
    int16_t *out = scr + sample_out_off_i16(N);
    // Floyd: lane t holds the chosen values number t and t + LPE
    int c0 = -1, c1 = -1;
    if (size <= 16) {
        draw_floyd16<LPE>(scr, lane, a, N, pop, c0, c1);
        draw_shuffle16<LPE>(p, env, scr, lane, a, N, c0, c1, ok);
    } else {
        // Longer lists (N > 8: up to 128 values at N = 64).  Floyd's sampling and the tail shuffle are sequential as
        // NumPy writes them, 2N dependent iterations each (40 k cycles of one wave at N = 64, during which the other
        // 1 023 wait for the launch to end).  With one group per wave (N > 32) both are restated with short chains:
        if (size > LPE) {  // two elements per lane: element k and k + LPE in lane k of the group (N = 9 .. 16 in groups of
                           // 16 lanes, 17 .. 32 in groups of 32, 33 .. 64 in one wave); fewer values in a wide group --
                           // lanes_per_env forced wider than the agents need -- take the loop below
            draw_floyd_par<LPE>(vals, lane, a, size, pop, c0, c1);
            draw_shuffle_par<LPE>(p, env, raw, vals, out, lane, a, size, c0, c1, ok);
        } else {
            for (int k = 0; k < size; k++) {
                const int val = (int)vals[k], j = pop - size + k;
                const bool hit = (a < k && c0 == val) || (a + LPE < k && c1 == val);
                const int chosen = gballot<LPE>(hit, lane) != 0 ? j : val;
                if (a == (k & (LPE - 1))) {
                    if (k < LPE) c0 = chosen;
                    else c1 = chosen;
                }
            }
            if (a < size) out[a] = (int16_t)c0;
            if (a + LPE < size) out[a + LPE] = (int16_t)c1;
            wave_lds_sync();
            if (ok && a == 0) {  // several groups per wave: the swaps run on the LDS copy, one lane per group
                for (int i = size - 1; i >= 1; i--) {
                    const int j = (int)vals[size + (size - 1 - i)];
                    const int16_t t = out[j];
                    out[j] = out[i];
                    out[i] = t;
                }
            }
        }
    }
    wave_lds_sync();
}

// both halves back to back (the inline reset)
template <int LPE>
__device__ __forceinline__ bool sample_starts_goals_parallel(const Params &p, int16_t *scr, int lane, int a, int env,
                                                             bool env_ok, bool do_reset, int N, const uint64_t *rng_src,
                                                             const PcgPre &pre = PcgPre{false, {}, 0, {}, {}}) {
    int pop = 0;
    const bool ok = draw_stage_a<LPE>(p, scr, lane, a, env, env_ok, do_reset, N, nullptr, pre, rng_src, pop);
    draw_stage_b<LPE>(p, env, scr, lane, a, ok, N, pop);
    return ok;
}

// zero the wave's cell maps (16-byte LDS stores; in k_step this runs under the latency of the state loads)
template <int LPE>
__device__ __forceinline__ void clear_cell_maps(const Io &io, uint32_t *map, int lane) {
    constexpr int G = 64 / LPE;
    const int n4 = (G * (io.H + 2 * kRowPad) * (io.W + 2 * kRowPad) + 3) >> 2;  // region is padded to 16 bytes
    uint4 *m4 = reinterpret_cast<uint4 *>(map);
    for (int k = lane; k < n4; k += 64) m4[k] = make_uint4(0u, 0u, 0u, 0u);
}

// ------------------------------------------------------------------------------------------------
// reset() of the groups with do_reset set (MA-env:440-472).  Group-uniform inputs; called under a
// wave-uniform branch.  Updates lane state + scalars; stages the reset observation when want_obs.
// ------------------------------------------------------------------------------------------------
template <class K, int LPE, int MW>
__device__ __forceinline__ void reset_groups(const Params &p, const Io &io, const uint64_t *lrows, uint4 *tab, float *stage,
                                             int16_t *scratch, int lane, int a, int grp, int env, bool env_ok,
                                             bool is_agent, bool do_reset, Lane &st, int *sc, bool want_obs, uint32_t &nsg,
                                             bool obs_wave_barrier = false, uint32_t *wave_map = nullptr,
                                             const bool have_stream = false, const Pcg &stream = Pcg{}, int stream_pop = 0) {
    // wave_map: the wave's LDS cell map when the group IS the wave (64 lanes) and the kernel has one: the reset observation
    // of 33 .. 64 agents then reads its windows from the map (25 reads) instead of walking all pairs (7.5 k cycles at N = 64).
    // stream / stream_pop: the env's generator and free-cell count when the caller holds them in registers already
    // (lifelong steps: loaded before the move phase, advanced by this step's respawns; else requested when the end of
    // the episode was decided): the draw does not start with a round trip to memory.
    const int N = K::N(p);
#ifdef MAPF_STAMPS  // (stamps build: slots 24..28 of the workgroup's row time an inline reset)
#define MAPF_STAMP_RG(k)                                                                                   \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        unsigned long long _t;                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)((env - grp) / (64 / LPE)) * kDbgRow + (k)] = _t;      \
    } while (0)
#else
#define MAPF_STAMP_RG(k) do { } while (0)
#endif
    MAPF_STAMP_RG(24);
    if (!(K::flags(p) & MAPF_FLAG_DETERMINISTIC)) {
        // generate_starts_goals MA-env:267-282: idx = rng.choice(F, 2N, replace=False).  A pre-drawn placement (see
        // kSlotInvalid) IS that draw; otherwise it is made here.
        const bool slot_ok = do_reset && gballot<LPE>(is_agent && !slot_word_valid(nsg), lane) == 0;
        const bool draw = do_reset && !slot_ok;
        if (__any(draw)) {
            // a staged background draw has advanced Params::rng already: the visible stream is in vis_rng then
            const bool staged = gballot<LPE>(is_agent && a == 0 && slot_word_staged(nsg), lane) != 0;
            const uint64_t *rng_src = staged ? p.vis_rng : p.rng;
            int16_t *hs = scratch + grp * p.scratch_i16;
            const int16_t *out = hs + sample_out_off_i16(N);
            MAPF_CHK(p, sample_out_off_i16(N) + 2 * N + 8 <= p.scratch_i16 && p.hash_cap + 2 * N <= p.scratch_i16, 10, env, p.scratch_i16);
            PcgPre pre{false, {}, 0, {}, {}};
            if (have_stream) {  // (by value through the inlined call: a pointer to the caller's copy would pin it in scratch)
                pre.have = true;
                pre.have_j = false;
                pre.g = stream;
                pre.pop = stream_pop;
            }
            const bool sampled = sample_starts_goals_parallel<LPE>(p, hs, lane, a, env, env_ok, draw, N, rng_src, pre);
            MAPF_STAMP_RG(25);
            if (__any(draw && !sampled)) {  // F = 2N or a Lemire rejection: the sequential restatement
                int16_t *outs = hs + p.hash_cap;
                if (draw && !sampled && a == 0) {
                    Pcg g;
                    pcg_load(g, rng_src + (size_t)env * 6);
                    const int hash_cap = p.hash_cap, mask = hash_cap - 1, size = 2 * N, pop = p.n_free[env];
                    bool stuck = false;
                    for (int k = 0; k < hash_cap; k++) hs[k] = -1;
                    for (int j = pop - size; j < pop; j++) {  // Floyd
                        int val = (int)pcg_bounded(g, (uint32_t)j, stuck);
                        int loc = val & mask;
                        // the set holds at most 2N < hash_cap entries, so an empty slot always exists; the
                        // probe counters only make termination structural
                        for (int pr = 0; hs[loc] != -1 && hs[loc] != val && pr < hash_cap; pr++) loc = (loc + 1) & mask;
                        MAPF_CHK(p, (unsigned)loc < (unsigned)hash_cap && (unsigned)(j - pop + size) < (unsigned)size, 5, env, loc);
                        if (hs[loc] == -1) {
                            hs[loc] = (int16_t)val;
                            outs[j - pop + size] = (int16_t)val;
                        } else {
                            loc = j & mask;
                            for (int pr = 0; hs[loc] != -1 && pr < hash_cap; pr++) loc = (loc + 1) & mask;
                            hs[loc] = (int16_t)j;
                            outs[j - pop + size] = (int16_t)j;
                        }
                    }
                    for (int i = size - 1; i >= 1; i--) {  // _shuffle_int tail shuffle
                        int j = (int)pcg_bounded(g, (uint32_t)i, stuck);
                        int16_t t = outs[j];
                        outs[j] = outs[i];
                        outs[i] = t;
                    }
                    if (stuck) raise_error(p, MAPF_ERR_RNG_GUARD, env, 0, 0);
                    if (env_ok) pcg_store(g, p.rng + (size_t)env * 6);
                }
                wave_lds_sync();
                if (draw && !sampled) out = outs;
            }
            if (draw && is_agent) {
                const uint16_t *fc = p.free_cells + (size_t)env * p.HW;
                const int top = p.HW - 1;  // idx entries are ranks < F <= HW; the clamp only bounds the address
                MAPF_CHK(p, (unsigned)out[a] < (unsigned)p.n_free[env] && (unsigned)out[N + a] < (unsigned)p.n_free[env], 6, env, out[a]);
                st.start = fc[min(max((int)out[a], 0), top)];
                st.goal = fc[min(max((int)out[N + a], 0), top)];
            }
            wave_lds_sync();
        }
        if (slot_ok && is_agent) {
            st.start = nsg & 0xFFFFu;
            st.goal = nsg >> 16;
        }
        // consumed, or overtaken by the inline draw (staged): Params::rng is the visible stream again
        if (do_reset && is_agent && nsg != kSlotInvalid) slots_of(io.scal, io.B)[(size_t)env * N + a] = kSlotInvalid;
        if (do_reset) nsg = kSlotInvalid;
    }
    if (do_reset) {
        st.pos = st.start;  // MA-env:279 / :453
        st.flags = 0;       // _reached_arr, _completed_once_arr, _blocking_pressure_prev_arr MA-env:447-449
        st.moved = st.failed = st.progress = 0;  // _reset_lock_tracking MA-env:360-372
        st.dist = make_uint4(0, 0, 0, 0);
        sc[MAPF_CTR_STEP_COUNT] = 0;
        sc[MAPF_CTR_HIST_ROWS] = 0;
        sc[MAPF_CTR_BLOCKING_COUNT] = 0;
        sc[MAPF_CTR_GOALS_REACHED_TOTAL] = 0;
        sc[MAPF_CTR_DEADLOCK_EVENTS] = 0;
        sc[MAPF_CTR_LIVELOCK_EVENTS] = 0;
        sc[MAPF_CTR_DEADLOCK_STEPS] = 0;
        sc[MAPF_CTR_LIVELOCK_STEPS] = 0;
        sc[MAPF_CTR_LOCK_STATE_PREV] = 0;
        sc[MAPF_CTR_MAY_FINISH] = 1;  // conservative; the next step computes the real hint
    }
    // Two-wave step kernel: the sampling above only touched the group's scratch, so it ran beside the observation wave;
    // pair table and staging rows are that wave's until it has passed B2.
    MAPF_STAMP_RG(26);
    if (obs_wave_barrier) wg_sync();  // B2
    MAPF_STAMP_RG(27);
    if (want_obs) {
        uint4 *tabg = tab + grp * LPE;
        PairOut po;
        if (LPE == 64 && wave_map) {
            const int map_w = io.W + 2 * kRowPad;
            clear_cell_maps<LPE>(io, wave_map, lane);  // (a step's fields are of no use once its observation has left)
            wave_lds_sync();
            if (is_agent && do_reset) {
                MAPF_CHK(p, (unsigned)map_index(st.pos, map_w) < (unsigned)((io.H + 2 * kRowPad) * map_w) &&
                                (unsigned)map_index(st.goal, map_w) < (unsigned)((io.H + 2 * kRowPad) * map_w), 7, env, st.pos);
                atomicOr(&wave_map[map_index(st.pos, map_w)], (uint32_t)a + 1u);           // owner-new field
                atomicOr(&wave_map[map_index(st.goal, map_w)], ((uint32_t)a + 1u) << 14);  // goal-owner field
            }
            wave_lds_sync();
            observe<K, LPE, MW, kObsEmit, LPE == 64>(p, io, lrows + kRowPad, tabg, stage + (size_t)a * K::L(p), is_agent && do_reset, a,
                                                    st.pos, st.goal, true, false, 0, po, wave_map);
        } else {
            tabg[a] = static_entry(st.pos, st.goal);
            wave_lds_sync();
            observe<K, LPE, MW, kObsEmit>(p, io, lrows + grp * (io.H + 2 * kRowPad) + kRowPad, tabg, stage + (size_t)(grp * N + a) * K::L(p),
                                       is_agent && do_reset, a, st.pos, st.goal, true, false, 0, po);
        }
        wave_lds_sync();
    }
    MAPF_STAMP_RG(28);
}

// LDS carve-up shared by the three kernels
struct Lds {
    uint64_t *rows;
    uint4 *tab;
    uint4 *otab;  // two-wave step kernels: the entries the observation wave reads (x old|new<<16, y goal, w kObsW*);
                  // two copies, used alternately by consecutive steps of the fused kernel (carve_lds + set_parity)
    uint4 *xpose;  // 3 KiB of staging: info rows and counters of the wave's envs on their way out (step_body)
    float *stage;
    int16_t *scratch;
    uint32_t *map;  // [G][H + 2*kRowPad][W + 2*kRowPad] cell words (only when Io::use_map)
};
__device__ __forceinline__ Lds carve_lds(const Io &io, unsigned char *raw) {
    Lds l;
    l.rows = reinterpret_cast<uint64_t *>(raw);
    l.tab = reinterpret_cast<uint4 *>(raw + io.lds_tab_off);
    l.otab = l.tab + 64;
    l.xpose = l.tab + 192;
    l.stage = reinterpret_cast<float *>(raw + io.lds_stage_off);
    l.scratch = reinterpret_cast<int16_t *>(raw + io.lds_scratch_off);
    l.map = reinterpret_cast<uint32_t *>(raw + io.lds_map_off);
    return l;
}

// ------------------------------------------------------------------------------------------------
// reset kernel
// ------------------------------------------------------------------------------------------------
template <class K, int LPE, int MW>
__global__ __launch_bounds__(64) void k_reset(const Params *__restrict__ pp, const Io io) {
    const Params &p = *pp;
    constexpr int G = 64 / LPE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const Lds l = carve_lds(io, lds_raw);
    const int lane = threadIdx.x, grp = lane / LPE, a = lane % LPE;
    const int env0 = blockIdx.x * G;
    const int ngroups = min(G, io.B - env0);
    const bool env_ok = grp < ngroups;
    const int env = env_ok ? env0 + grp : io.B - 1;
    const int N = K::N(p);
    const bool is_agent = env_ok && a < N;

    load_rows_to_lds<LPE>(io.grid_rows, io.H, l.rows, lane, env0, ngroups);
    Lane st;
    load_lane(io.agents, io.bn8, (size_t)env * N + min(a, N - 1), is_agent, st);
    int sc[12];
    load_scal(io.scal, env, sc);
    const bool do_reset = env_ok && (io.env_mask == nullptr || io.env_mask[env] != 0);
    uint32_t nsg = p.next_sg[(size_t)env * N + min(a, N - 1)];
    wave_lds_sync();

    reset_groups<K, LPE, MW>(p, io, l.rows, l.tab, l.stage, l.scratch, lane, a, grp, env, env_ok, is_agent, do_reset, st, sc,
                             io.obs != nullptr, nsg, false, (LPE == 64 && io.use_map) ? l.map : nullptr);
    if (io.obs) flush_obs<K, LPE>(p, io, l.stage, lane, env0, ngroups, do_reset ? 0 : 2);
    if (do_reset) {
        if (is_agent)
            store_lane(io.agents, io.bn8, (size_t)env * N + a, st, l.rows + grp * (io.H + 2 * kRowPad) + kRowPad, io.col_pad, io.W);
        if (a == 0) store_scal(io.scal, env, sc);
    }
}

// ------------------------------------------------------------------------------------------------
// observe kernel: observation of every agent from the current (static) state, nothing is modified.
// What the reference computes when get_obs / _flatten_observation are called outside step()
// (its tests do: tests/test_reference_model_multi_agent_invariants.py:76-95).
// ------------------------------------------------------------------------------------------------
template <class K, int LPE, int MW>
__global__ __launch_bounds__(64) void k_observe(const Params *__restrict__ pp, const Io io) {
    const Params &p = *pp;
    constexpr int G = 64 / LPE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const Lds l = carve_lds(io, lds_raw);
    const int lane = threadIdx.x, grp = lane / LPE, a = lane % LPE;
    const int env0 = blockIdx.x * G;
    const int ngroups = min(G, io.B - env0);
    const bool env_ok = grp < ngroups;
    const int env = env_ok ? env0 + grp : io.B - 1;
    const int N = K::N(p);
    const bool is_agent = env_ok && a < N;
    load_rows_to_lds<LPE>(io.grid_rows, io.H, l.rows, lane, env0, ngroups);
    Lane st;
    load_lane(io.agents, io.bn8, (size_t)env * N + min(a, N - 1), is_agent, st);
    uint4 *tabg = l.tab + grp * LPE;
    tabg[a] = static_entry(st.pos, st.goal);
    wave_lds_sync();
    PairOut po;
    observe<K, LPE, MW, kObsEmit>(p, io, l.rows + grp * (io.H + 2 * kRowPad) + kRowPad, tabg, l.stage + (size_t)(grp * N + a) * K::L(p), is_agent, a,
                               st.pos, st.goal, true, (st.flags & kFlagPressure) != 0, 0, po);
    wave_lds_sync();
    flush_obs<K, LPE>(p, io, l.stage, lane, env0, ngroups, env_ok ? 0 : 2);
}

// ------------------------------------------------------------------------------------------------
// step kernel (MA-env:474-695)
// ------------------------------------------------------------------------------------------------
// FAST = the wave is full (every group is a live env, every lane an agent) and no lane carries an invalid
// action: every validity predicate below is then a compile-time constant and the error bookkeeping vanishes.
// The general body handles ragged batches, N < LPE and the reference's mid-loop ValueError.
//
// DUAL = the workgroup has a second wave, the observation wave (obs_wave_step below): this function is then the
// state wave.  It publishes what the observation needs in the pair table (entry word w, kObsW*), releases the
// observation wave with a workgroup barrier and carries on with the lock detector, the per-step outputs and the
// state image while the other wave builds, stages and streams out the observations.  With one wave per SIMD
// (c3: 1024 workgroups on 1024 SIMDs) a step is bound by one wave's dependent-instruction latency, not by
// bandwidth or issue slots; the split takes the observation (about a third of the instructions) off that path.
// (The post-B1 episode-end blocks of the state wave -- record image, end of body -- are cold code: laid out inline their
// selects ran in every wave)
constexpr uint32_t kObsWAgent = 1u, kObsWPressure = 2u, kObsWFinal = 4u, kObsWSelShift = 3u, kObsWFast = 32u,
                   kObsWReset = 64u,       // the state wave builds this group's reset observation itself, after B2
                   kObsWResetFast = 128u;  // the observation wave builds it (second pass) from entry word z

// store_inside (single-step kernel, full wave): the wave's 64 agents and the envs' counters are stored
// from inside the body, as soon as they are final, unless an env of the wave resets in this launch; returns
// whether that happened.
template <class K, int LPE, int MW, bool FAST, bool DUAL>
__device__ __forceinline__ bool step_body(const Params &p, const Io &io, const Lds &l, const int lane, const int env0,
                                          const int ngroups, int act, Lane &st, int *sc, uint32_t &nsg,
                                          const bool store_inside = false, const bool nsg_lazy = false) {
    constexpr int G = 64 / LPE;
    const int grp = lane / LPE, a = lane % LPE;
    const bool env_ok = FAST ? true : (grp < ngroups && env_live(io, env0 + grp));
    const int env = env_ok ? env0 + grp : io.B - 1;
    const int N = K::N(p), H = io.H, W = io.W;
    const uint32_t flags = K::flags(p);
    const bool is_agent = FAST ? true : (env_ok && a < N);
    const bool lifelong = (flags & MAPF_FLAG_LIFELONG) != 0;
    const bool lock_on = (flags & MAPF_FLAG_LOCK_METRICS) != 0;
    const int lw = K::lw(p), dw = K::dw(p), ring_stride = K::ring_stride(p);
    uint4 *tabg = l.tab + grp * LPE;
    const uint64_t *myrows = l.rows + grp * (H + 2 * kRowPad) + kRowPad;
    const bool dist_in_rec = lw <= 16;  // goal-distance history lives in the agent record

    // ---- invalid action: the reference raises mid-loop, after the agents before the bad one were
    //      processed (MA-env:502-506); reproduce the partial mutation and latch the error ---------
    bool errored = false, live = is_agent;
    if (!FAST) {
        const bool bad = is_agent && (act < 0 || act > 4);
        const uint64_t badm = gballot<LPE>(bad, lane);
        errored = badm != 0;
        const int n_live = errored ? (int)__builtin_ctzll(badm) : N;
        live = is_agent && a < n_live;
        if (bad && a == n_live) raise_error(p, MAPF_ERR_BAD_ACTION, env, a, act);
        if (!live) act = 0;
    }

    sc[MAPF_CTR_STEP_COUNT] += 1;  // MA-env:475

    // ---- move phase (MA-env:502-526) -----------------------------------------------------------
    const uint32_t old = st.pos;
    const int r_old = (int)(old >> 8), c_old = (int)(old & 255u);
    const int dr = (act == 1) ? -1 : ((act == 3) ? 1 : 0);
    const int dc = (act == 2) ? 1 : ((act == 4) ? -1 : 0);
    const int tr = r_old + dr, tc = c_old + dc;
    // rows -1 and H are sentinel rows; with col_pad the columns -1 and W are sentinel bits too
    const uint64_t trow = live ? myrows[tr] : ~0ull;
    const bool col_ok = io.col_pad != 0 || (tc >= 0 && tc < W);
    const bool want = live && act != 0 && col_ok && !((trow >> ((tc + io.col_pad) & 63)) & 1ull);
    const uint32_t tgt = want ? (uint32_t)((tr << 8) | tc) : kNoCell;
    // intended_next (MA-env:514-515) in the (+1,+1) encoding; may lie outside the grid
    const uint32_t intended1 = (uint32_t)(((tr + 1) << 8) | (tc + 1));
    uint32_t cur = old;
    // Lifelong mode: what a respawn (below) reads from global memory -- the stream state, the free-cell count and the
    // row-major ranks of my old cell, my target cell and my goal -- is requested here, a whole move phase before it is
    // needed: a wave in which an agent arrives (most waves of c5, every step) sets the launch's duration, and it used
    // to start this round trip only after the move.
    Pcg g_ll;
    g_ll.shi = g_ll.slo = g_ll.ihi = g_ll.ilo = 0;
    g_ll.has32 = g_ll.uinteger = 0;
    int F_ll = 0, rankOld = 0x7FFFFFFF, rankTgt = 0x7FFFFFFF, rankGoal = 0x7FFFFFFF;
    if (lifelong) {
        pcg_load(g_ll, p.rng + (size_t)env * 6);
        F_ll = p.n_free[env];
        const uint16_t *frank = p.free_rank + (size_t)env * p.HW;
        if (is_agent) {
            rankOld = (int)frank[r_old * W + c_old];
            rankTgt = (int)frank[want ? tr * W + tc : r_old * W + c_old];  // (independent of the load above: no wait here)
            rankGoal = (int)frank[(st.goal >> 8) * W + (st.goal & 255u)];
        }
    }
    MAPF_STAMP(16);  // (sub-stamp: target cell known)
    constexpr bool MAP_OK = LPE >= 32;  // the cell-map path is only built for wide groups (N > 16)
    const bool use_map = MAP_OK && (K::kMapAlways || io.use_map);
    const int map_w = W + 2 * kRowPad;
    uint32_t *mapg = l.map + grp * (H + 2 * kRowPad) * map_w;
    if (use_map) {
        MAPF_CHK(p, !is_agent || (unsigned)map_index(old, map_w) < (unsigned)((H + 2 * kRowPad) * map_w), 7, env, old);
        if (is_agent) atomicOr(&mapg[map_index(old, map_w)], ((uint32_t)a + 1u) << 7);  // owner-old field
        if (__any(want)) cur = resolve_moves_map<K, LPE>(p, mapg, map_w, lane, a, old, tgt);
        else wave_lds_sync();
    } else if (__any(want)) {
        cur = resolve_moves<K, LPE>(p, reinterpret_cast<uint2 *>(tabg), lane, a, old, tgt);
    }
    const bool moved = cur != old;
    MAPF_STAMP(2);

    // ---- goal / reward logic (MA-env:538-563) --------------------------------------------------
    bool reached = (st.flags & kFlagReached) != 0;
    bool completed = (st.flags & kFlagCompleted) != 0;
    const bool pressure_prev = (st.flags & kFlagPressure) != 0;
    float reward = 0.0f;
    bool grs = false;                      // goal_reached_step flag
    bool on_goal = live && cur == st.goal;  // reached_goal[i], evaluated at agent i's own turn
    bool reassigned = false;                // group-uniform: any lifelong respawn this step
    // termination (MA-env:668-690) only needs on_goal and the step counter, so it is decided right after the move:
    // the observation can then be built (and its stores leave) while the rest of the step is still being computed.
    // Success check precedes the step-limit check.  (Lifelong never terminates on goals and resets on_goal below.)
    int term = 0, trunc = 0;
    float term_reward = 0.0f;
    {
        const int n_on_goal = __popcll(gballot<LPE>(on_goal, lane));
        if (!lifelong && n_on_goal == N) {
            term_reward = 1.0f;
            term = 1;
        } else if (sc[MAPF_CTR_STEP_COUNT] >= io.steps_per_episode) {
            if (!lifelong && !on_goal) term_reward = -1.0f;
            term = 1;
            trunc = 1;
        }
    }
    const bool done = env_ok && !errored && (term | trunc);
    const bool do_reset = done && io.auto_reset;
    // How a finished env gets its next episode:
    //   fast  the placement is known already -- a pre-drawn slot (kSlotInvalid) or the fixed starts of deterministic
    //         mode -- so the reset is a register image and the OBSERVATION WAVE builds the reset observation, instead
    //         of the terminal one when nobody asked for that, else in a second pass; this wave never waits for it;
    //   slow  reset_groups() at the end of the body: draws the placement, builds the observation after B2.
    const bool deterministic = (flags & MAPF_FLAG_DETERMINISTIC) != 0;
    const bool want_any_obs = io.obs || io.final_obs;
    const bool obs_wave = DUAL && want_any_obs;  // the other wave works this step (workgroup-uniform)
    // the placement a fast reset installs, start | goal << 16 (deterministic: reset() keeps the goals, MA-env:452-455)
    auto reset_placement = [&]() -> uint32_t {
        if (!is_agent) return kIdleCell | (kIdleGoal << 16);
        return deterministic ? ((st.start & 0xFFFFu) | (st.goal << 16)) : nsg;
    };
    // which tensor a group's observation goes to: 0 io.obs, 1 io.final_obs (terminal observation of an env that is
    // reset right away), 2 nowhere;  obs_w0: the kObsW* flags of this lane's table entry
    int sel = (!env_ok || errored) ? 2 : (io.obs ? 0 : 2);
    uint32_t obs_w0 = (is_agent ? kObsWAgent : 0u) | (pressure_prev ? kObsWPressure : 0u) | (FAST ? kObsWFast : 0u);
    bool fast_reset = false, slow_reset = false, subst = false;
    // Everything about episode ends sits behind ONE wave-uniform branch: a step in which no env of the wave finishes
    // (the common case) pays a ballot and a scalar branch for it.
    // ('unlikely': measured both ways -- laid out inline the episode-end blocks cost every wave more (B1 is reached
    // 80 cycles later, 5.5 -> 5.7 us synchronised, 6.2 -> 6.4 staggered) than the jumps to the far end of the kernel
    // cost the waves that take them)
    // Lifelong steps hold the env's generator and free-cell count in registers since before the move phase (g_ll, advanced
    // by the respawns below): an inline draw at the end of this body starts from them instead of a round trip to memory.
    // They wait in LDS meanwhile (kept in registers they stretched g_ll's live range over the whole body: +3 % on
    // every step for the sake of the rare one).
    bool have_rs = false;
    uint4 *rs_stash = l.xpose + 128;  // 64 bytes past the info / counter staging of the wave's envs
    if (__builtin_expect(__any(do_reset), 0)) {
        if (nsg_lazy && !deterministic && !lifelong)  // (A/B: the slot is only fetched when an env of the wave finishes)
            nsg = slots_of(io.scal, io.B)[(size_t)env * N + min(a, N - 1)];
        bool slot_ok = deterministic;
        if (!deterministic && !lifelong) slot_ok = gballot<LPE>(is_agent && !slot_word_valid(nsg), lane) == 0;
        fast_reset = do_reset && !use_map && slot_ok && (obs_wave || !want_any_obs);
        slow_reset = do_reset && !fast_reset;
        have_rs = LPE == 64 && lifelong && slow_reset && !slot_ok;  // (one group per wave: wave-uniform)
        subst = fast_reset && io.final_obs == nullptr;  // reset observation in place of the terminal one
        if (do_reset) sel = io.final_obs ? 1 : ((subst && io.obs) ? 0 : 2);
        if (subst) obs_w0 = (obs_w0 & ~kObsWPressure) | kObsWFinal;
        obs_w0 |= (slow_reset ? kObsWReset : 0u) | ((fast_reset && !subst) ? kObsWResetFast : 0u);
    }
    obs_w0 |= (uint32_t)sel << kObsWSelShift;
    uint4 *otabg = l.otab + grp * LPE;
    // entry the observation wave reads: x old | new << 16, y goal, z reset placement, w kObsW* flags
    auto obs_entry = [&](bool all_final) -> uint4 {
        uint4 e = make_uint4(old | (cur << 16), st.goal & 0xFFFFu, 0u, obs_w0 | (all_final ? kObsWFinal : 0u));
        if (__builtin_expect(__any(fast_reset), 0)) {
            const uint32_t rs = reset_placement(), rs_pos = rs & 0xFFFFu;
            e.z = rs;
            if (subst) e = make_uint4(rs_pos | (rs_pos << 16), rs >> 16, rs, obs_w0);
        }
        return e;
    };
    if (obs_wave && !lifelong && !use_map) {  // finite episodes: goals are fixed, the observation only waited for the moves
        otabg[a] = obs_entry(false);
        wg_sync();  // B1
        MAPF_STAMP(19);  // (sub-stamp: finite mode, observation wave released)
    }

    if (!lifelong) {
        if (on_goal && !reached) {
            reached = true;
            completed = true;
            reward += 0.5f;
            grs = true;
        }
    } else {
        const uint64_t arr_wave = __ballot(on_goal);
        if (arr_wave) {  // wave-uniform, rare
            const uint64_t garr = gballot<LPE>(on_goal, lane);
            reassigned = garr != 0;
            // (stream state, free-cell count and ranks were requested before the move phase.)  Everything in the loop
            // works on ranks: the cell of a goal drawn here is only loaded, never waited for.
            Pcg g = g_ll;
            const int F = F_ll;
            const int rankCur = moved ? rankTgt : rankOld;
            // index + 1 of the agent standing on my goal cell after (on) / before (oo) its move, 0 = nobody
            int on = 0, oo = 0;
            if (use_map) {  // owner-new ids (the full field is ORed in later anyway), then one map read
                if (is_agent) atomicOr(&mapg[map_index(cur, map_w)], (uint32_t)a + 1u);
                wave_lds_sync();
                const uint32_t mw = is_agent ? mapg[map_index(st.goal, map_w)] : 0u;
                on = (int)(mw & 127u);
                oo = (int)((mw >> 7) & 127u);
            }
            uint64_t u = fold_groups<LPE>(arr_wave);
            while (u) {  // respawns happen in agent order, each sees the state "at time i" (MA-env:554)
                const int i = (int)__builtin_ctzll(u);
                u &= u - 1;
                const bool gact = (garr >> i) & 1ull;
                // occupied cells at time i; goals of everybody else (own old goal is released first, MA-env:286-288)
                const bool Gact = is_agent && a != i;
                const int rankP = (a <= i) ? rankCur : rankOld;
                const int rankG = Gact ? rankGoal : 0x7FFFFFFF;
                bool dup = false;  // my goal cell is also occupied -> count it once
                if (use_map) {
                    dup = (on != 0 && on - 1 <= i) || (oo != 0 && oo - 1 > i);
                } else {
                    for (int q = 0; q < N; q++) dup |= (gshfl<LPE>(rankP, q) == rankGoal);
                }
                dup = dup && Gact;
                const int overlap = __popcll(gballot<LPE>(dup, lane));
                const int k = F - N - (N - 1) + overlap;  // candidate_indices.size MA-env:295
                uint32_t r = 0;
                if (gact) {
                    if (k <= 0) {
                        if (a == i) raise_error(p, MAPF_ERR_NO_RESPAWN, env, i, k);
                    } else {
                        bool stuck = false;
                        r = pcg_bounded(g, (uint32_t)(k - 1), stuck);  // rng.integers(k) MA-env:300
                        if (stuck && a == i) raise_error(p, MAPF_ERR_RNG_GUARD, env, i, k);
                    }
                }
                // r-th candidate in row-major order = free-rank y with y = r + #{excluded ranks <= y}
                int y = (int)r;
                for (int it = 0; it <= 2 * N; it++) {  // converges in <= #excluded + 1 rounds
                    int cnt = __popcll(gballot<LPE>(is_agent && rankP <= y, lane)) +
                              __popcll(gballot<LPE>(Gact && !dup && rankG <= y, lane));
                    int y2 = (int)r + cnt;
                    bool changed = gact && k > 0 && y2 != y;
                    y = y2;
                    if (!__any(changed)) break;
                }
                if (__any(gact && k > 0)) {
                    // who stands on the chosen cell later in this step (it is free of agents at time i, not after)
                    const uint64_t bc = gballot<LPE>(is_agent && rankCur == y, lane);
                    const uint64_t bo = gballot<LPE>(is_agent && rankOld == y, lane);
                    if (gact && k > 0 && a == i) {
                        st.goal = p.free_cells[(size_t)env * p.HW + y];  // MA-env:301-303
                        rankGoal = y;                                     // the rank of the cell just chosen
                        on = bc ? (int)__builtin_ctzll(bc) + 1 : 0;
                        oo = bo ? (int)__builtin_ctzll(bo) + 1 : 0;
                    }
                }
            }
            if (reassigned && a == 0 && env_ok) pcg_store(g, p.rng + (size_t)env * 6);
            if (reassigned) g_ll = g;
            if (on_goal) {  // MA-env:547-556
                reward += 0.5f;
                grs = true;
                completed = true;
                reached = false;
                on_goal = false;  // reached_goal[i] = False after the respawn
            }
        }
        if (LPE == 64 && __builtin_expect(__any(have_rs), 0)) {
            if (a == 0) {
                rs_stash[0] = make_uint4((uint32_t)g_ll.shi, (uint32_t)(g_ll.shi >> 32), (uint32_t)g_ll.slo, (uint32_t)(g_ll.slo >> 32));
                rs_stash[1] = make_uint4((uint32_t)g_ll.ihi, (uint32_t)(g_ll.ihi >> 32), (uint32_t)g_ll.ilo, (uint32_t)(g_ll.ilo >> 32));
                rs_stash[2] = make_uint4(g_ll.has32, g_ll.uinteger, (uint32_t)F_ll, 0u);
            }
        }
    }
    const int goals_step = __popcll(gballot<LPE>(grs, lane));
    sc[MAPF_CTR_GOALS_REACHED_TOTAL] += goals_step;  // _episode_goals_reached_total MA-env:550,563

    // ---- everything below is skipped by the reference when the ValueError fired; such groups keep
    //      only the mutations made before the exception ----------------------------------------------
    int sc_keep[12];
    uint64_t h_moved = 0, h_failed = 0, h_progress = 0;
    const uint4 h_dist = st.dist;
    if (!FAST) {
#pragma unroll
        for (int k = 0; k < 12; k++) sc_keep[k] = sc[k];
        h_moved = st.moved;
        h_failed = st.failed;
        h_progress = st.progress;
    }

    // lock flags (MA-env:581-594) and distance ring
    const bool cur_on_goal = is_agent && cur == st.goal;
    const bool prev_on_goal = !lifelong && old == st.goal;
    const bool progress = lifelong ? grs : (!prev_on_goal && cur_on_goal);
    const bool failed = act != 0 && !moved;
    const int dist = cell_l1(cur, st.goal);
    int delta = 0;
    bool dl_ok = false, ll_ok = false;
    if (lock_on) {
        const int t = sc[MAPF_CTR_HIST_ROWS];
        const int count = min(t + 1, K::hs(p));
        dl_ok = count >= dw;
        ll_ok = count >= lw;
        st.moved = (st.moved << 1) | (moved ? 1ull : 0ull);  // _append_lock_history_step MA-env:374-387
        st.failed = (st.failed << 1) | (failed ? 1ull : 0ull);
        st.progress = (st.progress << 1) | (progress ? 1ull : 0ull);
        int d_old = dist;
        if (dist_in_rec) {
            // shift the 16-byte history by one step and read the oldest row of the window (byte lw-1)
            st.dist.w = (st.dist.w << 8) | (st.dist.z >> 24);
            st.dist.z = (st.dist.z << 8) | (st.dist.y >> 24);
            st.dist.y = (st.dist.y << 8) | (st.dist.x >> 24);
            st.dist.x = (st.dist.x << 8) | (uint32_t)dist;
            const int ob = lw - 1, od = ob >> 2;
            const uint32_t w = od == 0 ? st.dist.x : (od == 1 ? st.dist.y : (od == 2 ? st.dist.z : st.dist.w));
            if (ll_ok) d_old = (int)((w >> ((ob & 3) * 8)) & 0xFFu);
        } else {
            int16_t *ring = io.dist_ring + ((size_t)env * N + a) * ring_stride;
            const int slot_new = t % lw;
            const int slot_old = (slot_new + 1 == lw) ? 0 : slot_new + 1;  // oldest row of the livelock window
            if (ll_ok && is_agent) d_old = ring[slot_old];
            if (is_agent && !errored) ring[slot_new] = (int16_t)dist;
        }
        delta = d_old - dist;
        sc[MAPF_CTR_HIST_ROWS] = t + 1;
    }
    // hint for the background sampler (sampler_wave): may this episode end in the NEXT step?  Only then could the
    // env's stream be consumed by a reset while the sampler draws from it.  Finite episodes end when every agent
    // stands on its goal -- impossible next step while some agent is two or more cells away -- or at the step limit.
    if (!lifelong)
        sc[MAPF_CTR_MAY_FINISH] = (gballot<LPE>(is_agent && dist > 1, lane) == 0 ||
                                   sc[MAPF_CTR_STEP_COUNT] + 1 >= io.steps_per_episode) ? 1 : 0;

    // observations (MA-env:528-534 staggered, or :565-575 all-final after a respawn) fused with the
    // neighbour / blocking pass
    PairOut po;
    float *srow = l.stage + (size_t)(grp * N + a) * K::L(p);
    MAPF_CHK(p, !is_agent || (grp * N + a + 1) * K::L(p) * 4 <= io.lds_scratch_off - io.lds_stage_off, 8, env, grp * N + a);
    // this wave builds the observation itself unless the other wave does, or nobody asked for one (fused steps
    // without observations)
    const bool emit_here = !obs_wave && (io.obs || io.final_obs);
    if (obs_wave && lifelong && !use_map) {  // respawned goals are part of the observation: publish after the goal logic
        otabg[a] = obs_entry(reassigned);
        wg_sync();  // B1
    }
    if (use_map) {
        // large N: every agent ORs its remaining fields into the env's cell map (owner-old went in before the move),
        // then reads only its window and lock neighbourhood from it (single-wave workgroups only: dual_for())
        if (is_agent) {
            const uint32_t me1 = (uint32_t)a + 1u;
            MAPF_CHK(p, (unsigned)map_index(cur, map_w) < (unsigned)((H + 2 * kRowPad) * map_w) &&
                            (unsigned)map_index(st.goal, map_w) < (unsigned)((H + 2 * kRowPad) * map_w) &&
                            (unsigned)((tr + kRowPad) * map_w + tc + kRowPad) < (unsigned)((H + 2 * kRowPad) * map_w), 7, env, cur);
            atomicOr(&mapg[map_index(cur, map_w)], me1 | ((uint32_t)(delta + 256) << 22));
            atomicOr(&mapg[map_index(st.goal, map_w)], me1 << 14);
            // intended_next may lie one cell outside the grid: that is inside the map's border
            if (!reached) atomicOr(&mapg[(tr + kRowPad) * map_w + tc + kRowPad], 1u << 21);
        }
        if (obs_wave) {  // the observation wave reads its window from the map: release it when the map is complete
            otabg[a] = obs_entry(reassigned);
            wg_sync();  // B1
        } else {
            wave_lds_sync();
        }
        MAPF_STAMP(3);
        if (emit_here)
            observe<K, LPE, MW, kObsBoth, MAP_OK>(p, io, myrows, tabg, srow, is_agent, a, cur, st.goal, reassigned,
                                                  pressure_prev, delta, po, mapg);
        else
            observe<K, LPE, MW, kObsPairs, MAP_OK>(p, io, myrows, tabg, srow, is_agent, a, cur, st.goal, reassigned,
                                                   pressure_prev, delta, po, mapg);
    } else {
        uint4 ent;
        ent.x = old | (cur << 16);
        ent.y = (st.goal & 0xFFFFu) | ((reached ? 1u : 0u) << 16) | ((uint32_t)(delta + 256) << 17);
        ent.z = (is_agent && !reached) ? intended1 : 0xFFFFFFFFu;
        ent.w = 0u;
        tabg[a] = ent;
        wave_lds_sync();
        MAPF_STAMP(3);
        if (emit_here)
            observe<K, LPE, MW, kObsBoth, false>(p, io, myrows, tabg, srow, is_agent, a, cur, st.goal, reassigned,
                                                 pressure_prev, delta, po);
        else
            observe<K, LPE, MW, kObsPairs, false>(p, io, myrows, tabg, srow, is_agent, a, cur, st.goal, reassigned,
                                                  pressure_prev, delta, po);
    }
    MAPF_STAMP(4);

    // ---- observations leave the wave as one contiguous stream (single-wave mode; otherwise the observation
    //      wave does this, concurrently with everything below) --------------------------------------------
    if (emit_here) {
        wave_lds_sync();
        if (FAST && !__any(do_reset)) {  // (no fast resets on this path: they need the observation wave)
            if (io.obs) flush_obs_full<K, LPE>(p, io, io.obs, l.stage, lane, env0);
        } else {
            flush_obs<K, LPE>(p, io, l.stage, lane, env0, ngroups, sel);
        }
    }
    MAPF_STAMP(5);

    reward += term_reward;
    // blocking flags feed NEXT step's observation (MA-env:608-625)
    const bool blocking = is_agent && reached && !moved && po.blocks;

    // ---- everything that does not depend on the lock detector leaves now, ahead of the observation stream of the
    //      other wave: rewards, per-agent info flags, done flags and (FAST, nobody resets) the agent records ------
    // (plain stores here: written through -- as k_step3's aux wave does, which only runs on grids of at most three waves
    // per SIMD -- these small outputs cost the launches that are bound by bandwidth rather than by one wave's latency:
    // 65 536 envs 27.6 -> 30.0 us per step, fused launches without observations 2.33 -> 2.61)
    if (is_agent && !errored) {
        if (io.rewards) io.rewards[(size_t)env * N + a] = reward;
        if (io.info_agent) {
            uchar2 ia;
            ia.x = blocking ? 1 : 0;
            ia.y = grs ? 1 : 0;
            reinterpret_cast<uchar2 *>(io.info_agent)[(size_t)env * N + a] = ia;
        }
    }
    if (env_ok && !errored && a == 0) {
        if (io.terminated) io.terminated[env] = (uint8_t)term;
        if (io.truncated) io.truncated[env] = (uint8_t)trunc;
    }
    MAPF_STAMP(17);  // (sub-stamp: rewards / per-agent info / done flags issued)
    bool records_stored = false;
    if (FAST && store_inside && !__any(slow_reset)) {
        st.pos = cur;
        st.flags = (reached ? kFlagReached : 0) | (completed ? kFlagCompleted : 0) | (blocking ? kFlagPressure : 0);
        Lane img = st;
        if (__builtin_expect(__any(fast_reset), 0)) {  // re-placed envs store the image reset() leaves (MA-env:440-455)
            const uint32_t rs = reset_placement();
            img.start = fast_reset ? (rs & 0xFFFFu) : st.start;
            img.goal = fast_reset ? (rs >> 16) : st.goal;
            img.pos = fast_reset ? img.start : st.pos;
            img.flags = fast_reset ? 0u : st.flags;
            img.moved = fast_reset ? 0ull : st.moved;
            img.failed = fast_reset ? 0ull : st.failed;
            img.progress = fast_reset ? 0ull : st.progress;
            img.dist = fast_reset ? make_uint4(0, 0, 0, 0) : st.dist;
        }
        store_lane(io.agents, io.bn8, (size_t)env0 * N + lane, img, myrows, io.col_pad, W);
        records_stored = true;
    }

    MAPF_STAMP(18);  // (sub-stamp: agent records issued)
    // lock detector (MA-env:400-438): deadlock has priority over livelock
    using gm_t = typename GMask<LPE>::type;
    int deadlock = 0, livelock = 0, dl_event = 0, ll_event = 0;
    if (lock_on) {
        const uint64_t mdw = dw >= 64 ? ~0ull : ((1ull << dw) - 1ull);
        const uint64_t mlw = lw >= 64 ? ~0ull : ((1ull << lw) - 1ull);
        const gm_t nbr = (gm_t)po.nbr;
        const gm_t members = nbr | ((gm_t)1 << a);
        const bool focal = is_agent && !cur_on_goal && __popcll((uint64_t)nbr) >= K::min_nbrs(p);
        const gm_t prog_dw_nz = gballot_n<LPE>(is_agent && (st.progress & mdw) != 0, lane);
        const gm_t moved_dw_nz = gballot_n<LPE>(is_agent && (st.moved & mdw) != 0, lane);
        const gm_t fail_dw_nz = gballot_n<LPE>(is_agent && (st.failed & mdw) != 0, lane);
        const gm_t prog_lw_nz = gballot_n<LPE>(is_agent && (st.progress & mlw) != 0, lane);
        const gm_t moved_lw_nz = gballot_n<LPE>(is_agent && (st.moved & mlw) != 0, lane);
        const bool dead_me = focal && dl_ok && (members & (prog_dw_nz | moved_dw_nz)) == 0 && (members & fail_dw_nz) != 0;
        const bool live_me = focal && ll_ok && (members & prog_lw_nz) == 0 && (members & moved_lw_nz) != 0 &&
                             po.sum_delta <= io.eps_floor;
        deadlock = gballot_n<LPE>(dead_me, lane) != 0;
        livelock = !deadlock && gballot_n<LPE>(live_me, lane) != 0;
        const int prev = sc[MAPF_CTR_LOCK_STATE_PREV];
        dl_event = deadlock && !(prev & 1);  // rising edges MA-env:599-600
        ll_event = livelock && !(prev & 2);
        sc[MAPF_CTR_LOCK_STATE_PREV] = deadlock | (livelock << 1);
        sc[MAPF_CTR_DEADLOCK_STEPS] += deadlock;
        sc[MAPF_CTR_LIVELOCK_STEPS] += livelock;
        sc[MAPF_CTR_DEADLOCK_EVENTS] += dl_event;
        sc[MAPF_CTR_LIVELOCK_EVENTS] += ll_event;
    }

    const int blocking_step = __popcll(gballot<LPE>(blocking, lane));
    sc[MAPF_CTR_BLOCKING_COUNT] += blocking_step;

    MAPF_STAMP(6);
    // ---- per-step outputs that need the lock detector: info (MA-env:627-656) ------------------------------
    {
        const int reached_cnt = __popcll(gballot<LPE>(is_agent && reached, lane));
        const int completed_cnt = __popcll(gballot<LPE>(is_agent && completed, lane));
        const int goals_total = lifelong ? sc[MAPF_CTR_GOALS_REACHED_TOTAL] : reached_cnt;
        const int steps = max(sc[MAPF_CTR_STEP_COUNT], 1);
        const float2 i0 = make_float2((float)goals_step, (float)goals_total);
        const float2 i1 = make_float2((float)blocking_step, (float)sc[MAPF_CTR_BLOCKING_COUNT]);
        const float2 i2 = make_float2((float)deadlock, (float)livelock);
        const float2 i3 = make_float2((float)dl_event, (float)ll_event);
        const float2 i4 = make_float2((float)sc[MAPF_CTR_DEADLOCK_EVENTS], (float)sc[MAPF_CTR_LIVELOCK_EVENTS]);
        const float2 i5 = make_float2((float)sc[MAPF_CTR_DEADLOCK_STEPS], (float)sc[MAPF_CTR_LIVELOCK_STEPS]);
        const float2 i6 = make_float2((float)completed_cnt / (float)N,        // completion_ratio MA-env:638
                                      (float)goals_total / (float)steps);      // throughput MA-env:655
        if (FAST) {
            // full wave: the G envs own G*56 contiguous bytes of info_all and G*64 of the counters.  Lane 0 of each
            // group drops its env's values in LDS (the transpose region is free again: same wave, DS ops in order)
            // and the wave writes each tensor with one store instead of seven (three) partial ones per env.
            constexpr int G = 64 / LPE;
            float2 *xi = reinterpret_cast<float2 *>(l.xpose);
            uint4 *xs = l.xpose + 64;
            if (a == 0) {
                float2 *q = xi + grp * 7;
                q[0] = i0; q[1] = i1; q[2] = i2; q[3] = i3; q[4] = i4; q[5] = i5; q[6] = i6;
                if (records_stored) {
                    xs[grp * 3] = make_uint4(sc[0], sc[1], sc[2], sc[3]);
                    xs[grp * 3 + 1] = make_uint4(sc[4], sc[5], sc[6], sc[7]);
                    xs[grp * 3 + 2] = make_uint4(sc[8], sc[9], sc[10], sc[11]);
                    if (fast_reset) {  // a re-placed env stores the counters reset() leaves
                        xs[grp * 3] = xs[grp * 3 + 1] = make_uint4(0, 0, 0, 0);
                        xs[grp * 3 + 2] = make_uint4(0, sc[MAPF_CTR_EPISODES_DONE] + 1, 1, sc[11]);
                    }
                }
            }
            wave_lds_sync();
            MAPF_STAMP(20);  // (sub-stamp: info / counters staged in LDS)
            if (io.info_all) {
                float2 *dst = reinterpret_cast<float2 *>(io.info_all + (size_t)env0 * MAPF_INFO_ALL);
                for (int k = lane; k < G * 7; k += 64) dst[k] = xi[k];
            }
            if (records_stored && lane < 3 * G) {  // the counters of an env are its first 48 of 64 bytes
                const int g = lane / 3, j = lane - 3 * g;
                store_state16(io.scal + (size_t)(env0 + g) * kScalInts + j * 4, xs[lane]);
            }
        } else if (io.info_all && env_ok && !errored && a == 0) {
            float2 *ia2 = reinterpret_cast<float2 *>(io.info_all + (size_t)env * MAPF_INFO_ALL);  // 56 B per env
            ia2[0] = i0; ia2[1] = i1; ia2[2] = i2; ia2[3] = i3; ia2[4] = i4; ia2[5] = i5; ia2[6] = i6;
        }
    }

    MAPF_STAMP(7);
    // ---- state image after the step -----------------------------------------------------------
    st.pos = cur;
    if (!FAST && errored) {
#pragma unroll
        for (int k = 0; k < 12; k++) sc[k] = sc_keep[k];
        st.moved = h_moved;
        st.failed = h_failed;
        st.progress = h_progress;
        st.dist = h_dist;
        st.flags = (reached ? kFlagReached : 0) | (completed ? kFlagCompleted : 0) | (pressure_prev ? kFlagPressure : 0);
        sc[MAPF_CTR_MAY_FINISH] = 1;  // agents before the bad one did move: the hint of the previous step is stale
    } else {
        st.flags = (reached ? kFlagReached : 0) | (completed ? kFlagCompleted : 0) | (blocking ? kFlagPressure : 0);
    }

    // ---- episode statistics (what the reference's RLlib callbacks read from the env when an episode ends,
    //      src/trainers/callbacks.py:236-345): per-env lifetime sums, touched only on the finishing step ----
    if (__any(done)) {
        const int completed_now = __popcll(gballot<LPE>(is_agent && completed, lane));
        if (done && a == 0) {
            // adds without a return value: nothing waits for them (a load-modify-store here put a memory round trip
            // into the tail of every wave in which an episode ends)
            int *acc = p.ep_acc + (size_t)env * MAPF_NUM_EPISODE_ACC;
            atomicAdd(acc + MAPF_ACC_EPISODES, 1);
            if (term && !trunc) atomicAdd(acc + MAPF_ACC_SUCCESSES, 1);                  // SuccessRateCallback
            atomicAdd(acc + MAPF_ACC_GOALS_REACHED, sc[MAPF_CTR_GOALS_REACHED_TOTAL]);  // <- _episode_goals_reached_total
            atomicAdd(acc + MAPF_ACC_BLOCKING_COUNT, sc[MAPF_CTR_BLOCKING_COUNT]);      // <- _episode_blocking_count
            atomicAdd(acc + MAPF_ACC_DEADLOCK_COUNT, sc[MAPF_CTR_DEADLOCK_EVENTS]);
            atomicAdd(acc + MAPF_ACC_LIVELOCK_COUNT, sc[MAPF_CTR_LIVELOCK_EVENTS]);
            atomicAdd(acc + MAPF_ACC_DEADLOCK_STEPS, sc[MAPF_CTR_DEADLOCK_STEPS]);
            atomicAdd(acc + MAPF_ACC_LIVELOCK_STEPS, sc[MAPF_CTR_LIVELOCK_STEPS]);
            atomicAdd(acc + MAPF_ACC_COMPLETED_AGENTS, completed_now);  // completion_ratio numerator (_completed_once_arr)
            atomicAdd(acc + MAPF_ACC_EPISODE_STEPS, sc[MAPF_CTR_STEP_COUNT]);           // episode length
        }
    }

    // ---- auto-reset of finished envs (reference harness loop scripts/benchmark_multi_agent_env.py:89-95:
    //      reset() right after a done step) ------------------------------------------------------------
    if (__builtin_expect(__any(fast_reset), 0)) {  // placement known: reset() is a register image (MA-env:440-455), the observation wave
                              // has (or is building) the reset observation
        if (fast_reset) {
            const uint32_t rs = reset_placement();
            st.start = rs & 0xFFFFu;
            st.goal = rs >> 16;
            st.pos = st.start;
            st.flags = 0;
            st.moved = st.failed = st.progress = 0;
            st.dist = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k <= MAPF_CTR_LOCK_STATE_PREV; k++) sc[k] = 0;
            sc[MAPF_CTR_EPISODES_DONE] += 1;
            sc[MAPF_CTR_MAY_FINISH] = 1;
        }
        if (fast_reset && !deterministic) {  // the slot is consumed (Params::rng already is the stream after its draw)
            if (is_agent) slots_of(io.scal, io.B)[(size_t)env * N + a] = kSlotInvalid;
            nsg = kSlotInvalid;
        }
    }
    if (__builtin_expect(__any(slow_reset), 0)) {
        if (slow_reset) sc[MAPF_CTR_EPISODES_DONE] += 1;
        wave_lds_sync();
        Pcg g_rs = Pcg{};
        int F_rs = 0;
        if (LPE == 64 && have_rs) {
            const uint4 s0 = rs_stash[0], s1 = rs_stash[1], s2 = rs_stash[2];
            g_rs.shi = (uint64_t)s0.x | ((uint64_t)s0.y << 32); g_rs.slo = (uint64_t)s0.z | ((uint64_t)s0.w << 32);
            g_rs.ihi = (uint64_t)s1.x | ((uint64_t)s1.y << 32); g_rs.ilo = (uint64_t)s1.z | ((uint64_t)s1.w << 32);
            g_rs.has32 = s2.x; g_rs.uinteger = s2.y;
            F_rs = (int)s2.z;
        }
        reset_groups<K, LPE, MW>(p, io, l.rows, l.tab, l.stage, l.scratch, lane, a, grp, env, env_ok, is_agent, slow_reset,
                                 st, sc, io.obs != nullptr, nsg, obs_wave,  // B2 inside: after the draw, before the observation
                                 (LPE == 64 && use_map) ? l.map : nullptr, have_rs, g_rs, F_rs);
        if (io.obs) flush_obs<K, LPE>(p, io, l.stage, lane, env0, ngroups, slow_reset ? 0 : 2);
    }
    return records_stored;
}

// The fused kernel alternates between the two copies of the observation table, so that the state wave may publish
// step t+1 while the observation wave still reads step t (no end-of-step barrier).
__device__ __forceinline__ Lds with_parity(const Lds &l, int t) {
    Lds r = l;
    r.otab = l.otab + (t & 1) * 64;
    return r;
}

// One step of the observation wave (wave 1 of a DUAL workgroup).  Everything it needs arrives through LDS: the
// obstacle rows and the observation table Lds::otab (its own entry carries cur / goal / kObsW* flags).  Barrier
// protocol per step with observations, identical in both waves: B1 (observation table published), then B2 only if
// some env of the workgroup resets (the state wave re-uses table and staging rows for the reset observation; its
// sampling, which only touches the group's scratch, runs before B2).
template <class K, int LPE, int MW>
__device__ __forceinline__ void obs_wave_step(const Params &p, const Io &io, const Lds &l, const int lane, const int env0,
                                              const int ngroups, const bool past_b1 = false,
                                              const float *gd_lut = nullptr) {
    const int grp = lane / LPE, a = lane % LPE;
    const int N = K::N(p), H = io.H;
    const uint4 *otabg = l.otab + grp * LPE;
    const uint64_t *myrows = l.rows + grp * (H + 2 * kRowPad) + kRowPad;
    float *srow = l.stage + (size_t)(grp * N + min(a, N - 1)) * K::L(p);
    if (!past_b1) wg_sync();  // B1
    MAPF_STAMP_W1(11);
    const uint4 ent = otabg[a];
    const uint32_t w = ent.w;
    const bool is_agent = (w & kObsWAgent) != 0;
    PairOut po;
    constexpr bool MAP_OK = LPE >= 32;
    if (MAP_OK && (K::kMapAlways || io.use_map)) {
        const int map_w = io.W + 2 * kRowPad;
        observe<K, LPE, MW, kObsEmit, MAP_OK>(p, io, myrows, otabg, srow, is_agent, a, ent.x >> 16, ent.y & 0xFFFFu,
                                              (w & kObsWFinal) != 0, (w & kObsWPressure) != 0, 0, po,
                                              l.map + grp * (H + 2 * kRowPad) * map_w, gd_lut);
    } else {
        observe<K, LPE, MW, kObsEmit, false>(p, io, myrows, otabg, srow, is_agent, a, ent.x >> 16, ent.y & 0xFFFFu,
                                             (w & kObsWFinal) != 0, (w & kObsWPressure) != 0, 0, po, nullptr, gd_lut);
    }
    wave_lds_sync();
    MAPF_STAMP_W1(12);
    const bool any_slow = __any((w & kObsWReset) != 0);
    const bool any_fast = __any((w & kObsWResetFast) != 0);
    if (__all((w & kObsWFast) != 0) && !any_slow && !any_fast) {
        if (io.obs) flush_obs_full<K, LPE>(p, io, io.obs, l.stage, lane, env0);
    } else {
        flush_obs<K, LPE>(p, io, l.stage, lane, env0, ngroups, (int)((w >> kObsWSelShift) & 3u));
    }
    MAPF_STAMP_W1(13);
    if (any_fast) {
        // Second pass: groups that were re-placed in this step and whose terminal observation was wanted (it just
        // went to final_obs).  Their table entries are rewritten to the reset state (everybody on its start, the
        // new goals: entry word z) and the reset observation goes to io.obs.
        const bool fr = (w & kObsWResetFast) != 0;
        const uint32_t rpos = ent.z & 0xFFFFu, rgoal = ent.z >> 16;
        uint4 *mine = l.otab + grp * LPE;
        wave_lds_sync();
        if (fr) mine[a] = make_uint4(rpos | (rpos << 16), rgoal, ent.z, w);
        wave_lds_sync();
        observe<K, LPE, MW, kObsEmit, false>(p, io, myrows, otabg, srow, is_agent && fr, a, rpos, rgoal, true, false, 0, po,
                                             nullptr, gd_lut);
        wave_lds_sync();
        flush_obs<K, LPE>(p, io, l.stage, lane, env0, ngroups, fr ? 0 : 2);
    }
#ifdef MAPF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MAPF_STAMP_W1(14);
#endif
    if (any_slow) wg_sync();  // B2
}

// ------------------------------------------------------------------------------------------------
// Sliced background draw (K::kSlicedDraw: specialised finite kernels, N = LPE <= 8).  The next episode's placement of
// an env is drawn by the OBSERVATION WAVE of the env's own workgroup, a third per launch, in the tail of the wave
// (after its observation stream is out it has ~2 k cycles to spare while the state wave finishes):
//     slot word 0  kSlotInvalid --[outputs + bounded draws]--> kSlotStaged --[Floyd]--> kSlotStaged2
//                  --[shuffle + free-cell gather]--> valid placement
// Intermediate data waits in Params::stage_vals; what a slice needs from memory is requested right after B0 (the env's
// slot word and hint arrive with the wave's first loads) and is there long before the tail.  Lane layout = the env
// groups of the step, so every group works on its own env: no picking, no hand-over, and an env is only touched
// when its MAY_FINISH hint is clear (nobody else reads or writes its stream or slot in this launch, see sampler_wave).
// ------------------------------------------------------------------------------------------------
struct DrawReq {
    uint32_t w0;       // slot word 0 of the group's env
    int hint;          // its MAY_FINISH hint
    uint4 rq;          // slices 1, 2: the stream the outputs are computed from, 16 bytes each in lanes 0 .. 2 of the group
    int pop;           // free-cell count
    uint32_t sv[5];    // slices 3..7: stage_vals dwords a + i * LPE
    uint4 ja[2], cq[2];  // slices 1, 2: A_q (kPcgJumpA) and the env's S_q * inc (Io::jump_c) for q = a + 1 + LPE and q = a + 1
};
__device__ __forceinline__ void draw_request_jump(const Io &io, int a, int env, int LPE, int which, DrawReq &d) {
    const int q = a + 1 + (which == 0 ? LPE : 0);  // index 0: slice 1 (the later half), 1: slice 2
    d.ja[which] = *reinterpret_cast<const uint4 *>(&kPcgJumpA[min(q, 64)][0]);
    d.cq[which] = reinterpret_cast<const uint4 *>(io.jump_c)[(size_t)env * 32 + min(q, 32) - 1];
}
// With the wave's first loads: the env's slot word and hint, and -- speculatively, 116 bytes per env and launch, so that
// a slice starts computing at B0 instead of a memory round trip later (it has to be done by B1) -- the free-cell count,
// the first two staging dwords of every lane (all that slices 4 .. 7 read) and the visible stream (slice 2).
// SPEC = false (the dense build of k_step, WPS != 0: several waves per SIMD hide a round trip, and at 65 536 envs the
// launch is HBM-bound, where 116 B per env count): only slot word and hint here, the rest once the slice is known.
template <class K, int LPE, bool SPEC>
__device__ __forceinline__ void draw_request_head(const Io &io, int N, int a, int env, DrawReq &d) {
    d.w0 = slots_of(io.scal, io.B)[(size_t)env * N];
    d.hint = io.scal[(size_t)env * kScalInts + MAPF_CTR_MAY_FINISH];
    d.pop = 2 * N + 1;
    d.sv[0] = d.sv[1] = 0;
    d.rq = make_uint4(0, 0, 0, 0);
    d.ja[0] = d.ja[1] = d.cq[0] = d.cq[1] = make_uint4(0, 0, 0, 0);
    if (SPEC) {
        draw_request_jump(io, a, env, LPE, 0, d);
        draw_request_jump(io, a, env, LPE, 1, d);
        d.pop = free_counts_of(io.scal, io.B, N)[env];
        const uint32_t *sv = io.stage_vals + (size_t)env * stage_dwords(N);
        d.sv[0] = sv[a];
        d.sv[1] = sv[a + LPE];
        d.rq = reinterpret_cast<const uint4 *>(io.vis_rng + (size_t)env * 6)[min(a, 2)];
    }
}
// after B0: which slice (0 = none) this group runs in this launch; issues the loads it still needs
constexpr int kDrawSlices = 7;
template <class K, int LPE, bool SPEC>
__device__ __forceinline__ int draw_request_body(const Params &p, const Io &io, int N, int a, int env, bool env_ok, DrawReq &d) {
    const bool idle = env_ok && d.hint == 0;
    int stage = 0;
    if (idle) {
        const uint32_t w = d.w0;
        stage = w == kSlotInvalid ? 1 : ((w >= kSlotStaged6 && w <= kSlotStaged) ? (int)(kSlotStaged - w) + 2 : 0);
    }
#pragma unroll
    for (int i = 2; i < 5; i++) d.sv[i] = 0;
    if (__any(stage != 0)) {
        // ONE kind of slice per wave and launch -- the most advanced one present (drains the pipeline; envs at the same
        // stage run side by side in their groups, the others wait for a later launch).  Without this a wave whose
        // envs finished in consecutive steps runs all the slices back to back and becomes the launch's long pole.
        int run = 0;
#pragma unroll
        for (int k = 1; k <= kDrawSlices; k++) run = __any(stage == k) ? k : run;
        if (stage != run) stage = 0;
        if (!SPEC && stage != 0) d.pop = free_counts_of(io.scal, io.B, N)[env];
        if (stage == 1) {
            // (F = 2N, where the first bounded draw consumes nothing, and the test knob never start a lane-parallel draw)
            if (d.pop <= 2 * N || (p.flags & MAPF_FLAG_SEQUENTIAL_RESET)) stage = 0;
        }
        if (__any(stage == 1)) {  // the stream before the draw is the env's own (slice 2: vis_rng, fetched with the head)
            const uint4 *rw = reinterpret_cast<const uint4 *>(streams_of(io.scal, io.B, N) + (size_t)env * 6);
            if (stage == 1) d.rq = rw[min(a, 2)];
            if (!SPEC) draw_request_jump(io, a, env, LPE, 0, d);
        }
        if (!SPEC && __any(stage == 2)) {
            const uint4 *rw = reinterpret_cast<const uint4 *>(io.vis_rng + (size_t)env * 6);
            if (stage == 2) d.rq = rw[min(a, 2)];
            draw_request_jump(io, a, env, LPE, 1, d);
        }
        if (__any(stage >= 3)) {
            const uint32_t *sv = io.stage_vals + (size_t)env * stage_dwords(N);
#pragma unroll
            for (int i = SPEC ? 2 : 0; i < 5; i++)
                if ((stage == 3 || (i < 2 && stage > 3)) && a + i * LPE < stage_dwords(N)) d.sv[i] = sv[a + i * LPE];
        }
    }
    return stage;
}
// the slice itself (tail of the observation wave)
template <class K, int LPE>
__device__ __forceinline__ void draw_slice(const Params &p, const Io &io, int16_t *scratch, int lane, int env, int stage,
                                           const DrawReq &d) {
    const int N = K::N(p);
    const int grp = lane / LPE, a = lane % LPE;
    int16_t *hs = scratch + grp * p.scratch_i16;
    uint32_t *raw = reinterpret_cast<uint32_t *>(hs);
    uint32_t *vals32 = raw + 4 * N + 2;  // vals[] of the group's scratch, as dwords
    uint16_t *vals = reinterpret_cast<uint16_t *>(vals32);
    uint32_t *out32 = reinterpret_cast<uint32_t *>(hs + sample_out_off_i16(N));  // out[2N] (int16) as dwords
    uint32_t *sv = io.stage_vals + (size_t)env * stage_dwords(N);
    uint32_t *slot = slots_of(io.scal, io.B) + (size_t)env * N;
    if (__any(stage == 1 || stage == 2)) {
        // ---- raw outputs (jump-ahead), half of them per slice: first the later half, whose last output carries the
        //      stream state after the draw (the stream advances, vis_rng keeps the visible state), then the earlier half
        const bool on = stage == 1 || stage == 2;
        PcgPre pre;
        pre.have = true;
        // (the 48 bytes of the stream sit in lanes 0 .. 2 of the group)
        pre.g.shi = (uint64_t)gshfl<LPE>(d.rq.x, 0) | ((uint64_t)gshfl<LPE>(d.rq.y, 0) << 32);
        pre.g.slo = (uint64_t)gshfl<LPE>(d.rq.z, 0) | ((uint64_t)gshfl<LPE>(d.rq.w, 0) << 32);
        pre.g.ihi = (uint64_t)gshfl<LPE>(d.rq.x, 1) | ((uint64_t)gshfl<LPE>(d.rq.y, 1) << 32);
        pre.g.ilo = (uint64_t)gshfl<LPE>(d.rq.z, 1) | ((uint64_t)gshfl<LPE>(d.rq.w, 1) << 32);
        pre.g.has32 = gshfl<LPE>(d.rq.x, 2);
        pre.g.uinteger = gshfl<LPE>(d.rq.z, 2);
        pre.pop = d.pop;
        // (one kind of slice per wave and launch: `stage` is wave-uniform among the groups that are on)
        const int qpass = __any(stage == 1) ? 1 : 0;
        pre.js = make_uint4(0, 0, 0, 0);
        pre.have_cq = true;  // one 128-bit product per output: A_q * state + (S_q * inc from the env's table)
        pre.ja = qpass == 1 ? d.ja[0] : d.ja[1];  // (a select of values: indexed with a run-time value the request lives in
        pre.cq = qpass == 1 ? d.cq[0] : d.cq[1];  //  scratch memory under the older clang of a host process's hiprtc)
        int pop_unused;
        const bool ok = draw_stage_a<LPE>(p, hs, lane, a, env, on, on, N, io.vis_rng, pre, p.rng, pop_unused, -1, 1,
                                          streams_of(io.scal, io.B, N), qpass, sv);
        if (on && ok && a == 0) slot[0] = qpass == 1 ? kSlotStaged : kSlotStaged2;
    }
    if (__any(stage == 3)) {  // ---- bounded draws (Lemire) on the raw outputs
        const bool on = stage == 3;
        if (on) {
            MAPF_CHK(p, (hs - scratch) + 2 * (4 * N + 2) <= (long)(64 / LPE) * p.scratch_i16, 9, env, grp);
#pragma unroll
            for (int i = 0; i < 5; i++)
                if (a + i * LPE < 4 * N + 2) raw[a + i * LPE] = d.sv[i];
        }
        wave_lds_sync();
        PcgPre pre;
        pre.have = true;
        pre.g.shi = pre.g.slo = pre.g.ihi = pre.g.ilo = 0;
        pre.g.has32 = pre.g.uinteger = 0;
        pre.pop = d.pop;
        pre.ja = pre.js = make_uint4(0, 0, 0, 0);
        int pop_unused;
        const bool ok = draw_stage_a<LPE>(p, hs, lane, a, env, on, on, N, nullptr, pre, p.rng, pop_unused, -1, 2);
        if (on) {
            if (ok) {
                sv[a] = vals32[a];
                sv[a + LPE] = vals32[a + LPE];
            }
            if (a == 0) slot[0] = ok ? kSlotStaged3 : kSlotStageFailed;
        }
    }
    if (__any(stage == 4 || stage == 5)) {  // ---- Floyd, half per slice; the chosen values replace the draws
        const bool on = stage == 4 || stage == 5;
        const int half = __any(stage == 4) ? 0 : 1;
        if (on) {
            vals32[a] = d.sv[0];
            vals32[a + LPE] = d.sv[1];
        }
        wave_lds_sync();
        int c0 = (int)vals[a], c1 = -1;  // (second half: the first half left this lane's chosen value in its place)
        const bool whole = 2 * N > 16;   // 32 values (N = 16): the loop-free formulation does both halves in this slice
        if (whole) draw_floyd_par<LPE>(vals, lane, a, 2 * N, d.pop, c0, c1);
        else draw_floyd16<LPE>(hs, lane, a, N, d.pop, c0, c1, half);
        wave_lds_sync();
        if (on) {
            if (half == 0 || whole) vals[a] = (uint16_t)c0;
            if (half == 1 || whole) vals[a + LPE] = (uint16_t)c1;
        }
        wave_lds_sync();
        if (on) {
            sv[a] = vals32[a];  // dwords 0 .. LPE-1 hold the 2N chosen values; the shuffle indices behind them stay
            if (a == 0) slot[0] = (half == 0 && !whole) ? kSlotStaged4 : kSlotStaged5;
        }
    }
    if (__any(stage == 6)) {  // ---- tail shuffle -> idx[2N]
        const bool on = stage == 6;
        if (on) {
            vals32[a] = d.sv[0];
            vals32[a + LPE] = d.sv[1];
        }
        wave_lds_sync();
        const int c0 = (int)vals[a], c1 = (int)vals[a + LPE];
        if (2 * N > 16) draw_shuffle_par<LPE>(p, env, raw, vals, hs + sample_out_off_i16(N), lane, a, 2 * N, c0, c1, on);
        else draw_shuffle16<LPE>(p, env, hs, lane, a, N, c0, c1, on);
        wave_lds_sync();
        if (on) {
            sv[a] = out32[a];  // N dwords = idx[2N]
            if (a == 0) slot[0] = kSlotStaged6;
        }
    }
    if (__any(stage == 7)) {  // ---- free-cell gather: the slot becomes valid
        const bool on = stage == 7;
        if (on) out32[a] = d.sv[0];
        wave_lds_sync();
        if (on) {
            const int16_t *out = hs + sample_out_off_i16(N);
            const int HW = io.H * io.W;
            const uint16_t *fc = io.free_cells + (size_t)env * HW;
            const int top = HW - 1;  // idx entries are ranks < F <= HW; the clamp only bounds the address
            const uint32_t st = fc[min(max((int)out[a], 0), top)];
            const uint32_t gl = fc[min(max((int)out[N + a], 0), top)];
            slot[a] = st | (gl << 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Background sampler: workgroups appended to the k_step grid (blockIdx >= number of env workgroups) whose waves
// pre-draw the NEXT episode's placement of envs that have none (kSlotInvalid), so that the step which ends the episode
// finds it ready (step_body: fast reset).  Lane i of sampler wave sw looks at env 64*sw + i; envs in need are then
// handled G at a time by the wave's lane groups with the same lane-parallel restatement of rng.choice(F, 2N) the
// inline reset uses, writing the slot (next_sg, vis_rng, and the advanced stream) instead of the env's state.
// No race with the env's own workgroup: an env is only touched when its MAY_FINISH hint (written by the previous
// step) is clear, i.e. when this launch cannot reset it, so nobody else reads or writes its stream or slot now.
// ------------------------------------------------------------------------------------------------
// LDS of a sampler workgroup (nothing of the env workgroups' layout is used): per wave a 4 KiB hand-over area (one
// 64-byte row per lane) followed by the draw scratch of its G groups.
constexpr int kSamplerHandover = 64 * 64;
__host__ __device__ constexpr int sampler_lds_bytes_per_wave(int groups, int scratch_i16) {
    return kSamplerHandover + ((groups * scratch_i16 * 2 + 15) & ~15);
}
template <class K, int LPE>
__device__ __forceinline__ void sampler_wave(const Params &p, const Io &io, unsigned char *lds, const int sw, const int lane,
                                             const int sw_row) {
    (void)sw_row;
    uint4 *pf = reinterpret_cast<uint4 *>(lds);
    int16_t *scratch = reinterpret_cast<int16_t *>(lds + kSamplerHandover);
    constexpr int G = 64 / LPE;
    const int N = K::N(p);
    const int grp = lane / LPE, a = lane % LPE;
    MAPF_STAMP_SW(0);
    const int e_l = sw * 64 + lane;
    const bool in = e_l < io.B;
    const int e_c = in ? e_l : io.B - 1;
    // One round trip fetches everything a draw needs, for the env each lane looks at: slot word, hint, stream state
    // and free-cell count (3 KiB per wave whether or not anything is to be done; a dependent second trip would sit on
    // the one path of the launch that has no slack).
    const uint32_t slot0 = slots_of(io.scal, io.B)[(size_t)e_c * N];
    const int hint = io.scal[(size_t)e_c * kScalInts + MAPF_CTR_MAY_FINISH];
    const uint4 *rw = reinterpret_cast<const uint4 *>(streams_of(io.scal, io.B, N) + (size_t)e_c * 6);
    const uint4 r0 = rw[0], r1 = rw[1], r2 = rw[2];
    const int pop_l = free_counts_of(io.scal, io.B, N)[e_c];
    const uint4 ja_l = *reinterpret_cast<const uint4 *>(&kPcgJumpA[a + 1][0]);  // this lane's jump-ahead constants
    const uint4 js_l = *reinterpret_cast<const uint4 *>(&kPcgJumpS[a + 1][0]);
    // (bitwise: all loads are in flight before any value is looked at)
    const uint64_t idle_ok = __ballot((int)in & (int)(hint == 0));
    const uint64_t need_b = idle_ok & __ballot(slot0 == kSlotStaged);   // second half pending
    const uint64_t need_a = idle_ok & __ballot(slot0 == kSlotInvalid);  // nothing drawn yet
    MAPF_STAMP_SW(1);
#ifdef MAPF_STAMPS
    if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)sw_row * kDbgRow + 4] = need_b ? 2 : (need_a ? 1 : 0);
#endif
    // ONE round per launch and HALF a draw per round: the first G envs in need are served (second halves first), the
    // others in a later launch (an episode that ends before its slot is ready draws inline, like any env without
    // one).  A sampler wave's work per launch is thereby bounded by half a draw -- about a third of an env workgroup's
    // step -- whatever happened to the batch (e.g. every env finishing in the same step).
    if (need_a | need_b) {
        uint64_t nb = need_b, na = need_a;
        int pick = -1;
        bool is_b = false;
#pragma unroll
        for (int g = 0; g < G; g++) {
            uint64_t &m = nb ? nb : na;
            if (m) {
                const int bit = (int)__builtin_ctzll(m);
                const bool from_b = nb != 0;
                m &= m - 1;
                pick = (grp == g) ? bit : pick;
                is_b = (grp == g) ? from_b : is_b;
            }
        }
        const bool act = pick >= 0, act_a = act && !is_b, act_b = act && is_b;
        const int env = act ? sw * 64 + pick : io.B - 1;
        const int src = act ? pick : lane;  // the lane that fetched the picked env's stream
        // hand-over through LDS (one 64-byte row per lane: three writes + one, then the picked lane's row back) instead
        // of a dozen cross-lane shuffles
        pf[lane * 4] = r0;
        pf[lane * 4 + 1] = r1;
        pf[lane * 4 + 2] = make_uint4(r2.x, r2.z, (uint32_t)pop_l, 0u);
        wave_lds_sync();
        const uint4 q0 = pf[src * 4], q1 = pf[src * 4 + 1], q2 = pf[src * 4 + 2];
        const int pop = (int)q2.z;
        int16_t *hs = scratch + grp * p.scratch_i16;
        uint32_t *vals32 = reinterpret_cast<uint32_t *>(hs) + 4 * N + 2;  // vals[] of the group's scratch, as dwords
        uint32_t *stage = p.stage_vals + (size_t)env * stage_dwords(N);
        if (__any(act_a)) {  // ---- first half: raw outputs + bounded draws -> stage_vals, stream advanced
            PcgPre pre;
            pre.have = true;
            pre.g.shi = (uint64_t)q0.x | ((uint64_t)q0.y << 32);
            pre.g.slo = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
            pre.g.ihi = (uint64_t)q1.x | ((uint64_t)q1.y << 32);
            pre.g.ilo = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
            pre.g.has32 = q2.x;
            pre.g.uinteger = q2.y;
            pre.pop = pop;
            pre.ja = ja_l;
            pre.js = js_l;
            int pop_unused;
            MAPF_STAMP_SW(5);
            const bool ok = draw_stage_a<LPE>(p, hs, lane, a, env, act_a, act_a, N, p.vis_rng, pre, p.rng, pop_unused, sw_row);
            if (act_a && ok) {
                for (int k = a; k < 2 * N; k += LPE) stage[k] = vals32[k];
                if (a == 0) slots_of(io.scal, io.B)[(size_t)env * N] = kSlotStaged;
            }
        }
        MAPF_STAMP_SW(2);
        if (__any(act_b)) {  // ---- second half: Floyd + shuffle on the staged draws -> the placement
            if (act_b) {
                for (int k = a; k < 2 * N; k += LPE) vals32[k] = stage[k];
            }
            wave_lds_sync();
            draw_stage_b<LPE>(p, env, hs, lane, a, act_b, N, pop);
            if (act_b && a < N) {
                const int16_t *out = hs + sample_out_off_i16(N);
                const uint16_t *fc = p.free_cells + (size_t)env * p.HW;
                const int top = p.HW - 1;  // idx entries are ranks < F <= HW; the clamp only bounds the address
                const uint32_t s = fc[min(max((int)out[a], 0), top)];
                const uint32_t g = fc[min(max((int)out[N + a], 0), top)];
                slots_of(io.scal, io.B)[(size_t)env * N + a] = s | (g << 16);
            }
        }
        wave_lds_sync();
#ifdef MAPF_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        MAPF_STAMP_SW(3);
    }
}

// Workgroup shape of the step kernels: kStepWaves waves of 64 lanes.  Wave 0 is the state wave, wave 1 (when
// present) the observation wave.
// k_step uses the split for every group width (c5, N = 64 with the LDS cell map: 10.2 -> 8.8 us).  The fused
// kernel keeps wide groups single-wave: its cell map is single-buffered, so the waves would have to meet at the
// end of every step, and the loop body then spills several hundred SGPRs (measured: 5.9 -> 8.8 us per step).
constexpr bool dual_for(int lpe) { return lpe >= 4; }
constexpr bool dual_many_for(int lpe) { return lpe < 32; }
constexpr int step_threads(int lpe) { return dual_for(lpe) ? 128 : 64; }
constexpr int many_threads(int lpe) { return dual_many_for(lpe) ? 128 : 64; }

// Second launch bound = minimum waves per SIMD the register budget must allow.  Wide groups: every SIMD has to hold a
// state wave and an observation wave (c5 launches exactly two waves per SIMD; past 256 registers half of the workgroups
// wait for a second round: 7.3 -> 11.4 us per step, measured).  The others: WPS, 0 meaning "no limit".  Without a limit the small-group
// kernels take ~145 VGPRs (three waves per SIMD), which is right while a launch has at most three waves per SIMD
// (c3: 8 192 envs = two) -- that build is 0.3 us per step faster there than one squeezed into 128 registers, whose
// spills sit on the episode-end paths -- and wrong beyond: 16 384 envs of the c3 shape are 4 096 waves, a quarter of
// them waited for a second round (10.0 us per step against 7.7).  The specialised kernels are therefore built both
// ways and mapf_create picks by the size of the grid (mapf_engine::dense).
template <class K, int LPE, int MW, int WPS = 0>
__global__ __launch_bounds__(step_threads(LPE), (LPE >= 32 ? 2 : (WPS ? WPS : 1))) void k_step(const Params *__restrict__ pp, MAPF_IO_HEAD_PARAMS,
                                                            const IoTail tail) {
    MAPF_STAMP_ENTRY();
    const Params &p = *pp;
    const Io io = MAPF_IO_JOIN;
    constexpr int G = 64 / LPE;
    constexpr bool kDual = dual_for(LPE);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int wv = kDual ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int lane = threadIdx.x & 63, grp = lane / LPE, a = lane % LPE;
    constexpr int kWavesPerWg = kDual ? 2 : 1;
    const int main_blocks = (io.B + G - 1) / G;
    // leading sampler workgroups shift the env workgroups (sampler_blocks_for: a function of the preloaded B alone)
    const int lead = K::kSamplerFront ? sampler_blocks_for(io.B, kWavesPerWg) : 0;
    const int env0 = ((int)blockIdx.x - lead) * G;
    const int ngroups = min(G, io.B - env0);
    const int N = K::N(p);

    // (unlikely: keeps the env workgroups' path the fall-through -- as the first block of the kernel the sampler code
    // pushed their entry 5 KB down and their first scalar loads behind a taken branch: +0.2 us per step)
    if (__builtin_expect(env0 < 0 || env0 >= io.B, 0)) {  // a sampler workgroup (the host adds them in finite mode
                                                          // with sampled placements)
        const int si = K::kSamplerFront ? (int)blockIdx.x : (int)blockIdx.x - main_blocks;
        // independent sampler waves, each with its own slice of the workgroup's LDS
        sampler_wave<K, LPE>(p, io, lds_raw + wv * sampler_lds_bytes_per_wave(G, p.scratch_i16), si * kWavesPerWg + wv, lane,
                             main_blocks + si);
        return;
    }

    // env workgroups issue ahead of a sampler wave that shares their SIMD (both of their waves at the same priority:
    // ranking state against observation wave was measured and is a loss)
    __builtin_amdgcn_s_setprio(2);
    // Both waves issue their global loads from the preloaded arguments alone, and only then wait for the scalar
    // loads (rest of the arguments, Params) in one batch.
    if (kDual && wv == 1) {
        // ---- observation wave: fetches the obstacle rows for both waves, then builds and streams the observations
        RowRegs rr;
        rows_issue<LPE>(io.grid_rows, io.H, lane, env0, ngroups, rr);
        DrawReq dreq;
        const bool d_env_ok = grp < ngroups;
        const int d_env = d_env_ok ? env0 + grp : io.B - 1;
        if (K::kSlicedDraw) draw_request_head<K, LPE, WPS == 0>(io, N, a, d_env, dreq);
        __builtin_amdgcn_sched_barrier(0);
        warm_scalar_cache(pp, tail);
        const Lds l = carve_lds(io, lds_raw);
        if (LPE >= 32 && (K::kMapAlways || io.use_map)) clear_cell_maps<LPE>(io, l.map, lane);
        rows_commit<LPE>(io.grid_rows, io.H, l.rows, lane, env0, ngroups, rr);
        wg_sync();  // B0: rows visible to the state wave
        MAPF_STAMP_W1(10);
        int d_stage = 0;
        if (K::kSlicedDraw) d_stage = draw_request_body<K, LPE, WPS == 0>(p, io, N, a, d_env, d_env_ok, dreq);
        // The background slice runs HERE, in the window in which this wave would only wait for the state wave's moves
        // (B0 .. B1, about 2700 cycles at c3): at the wave's tail, behind the observation stores, it outlasted the
        // state wave and set the launch's duration (6.2 against 5.5 us per step at c3 with staggered episodes).  Its
        // LDS is the draw scratch of its lane group, which the state wave only touches in a slow reset at the end of
        // its body -- and never for this env in this launch (MAY_FINISH hint).
        {
            if (K::kSlicedDraw && __builtin_expect(__any(d_stage != 0), 0)) {
                MAPF_STAMP_W1(21);
                draw_slice<K, LPE>(p, io, l.scratch, lane, d_env, d_stage, dreq);
#ifdef MAPF_STAMPS
                MAPF_STAMP_W1(22);
                {
                    int run = 0;
                    for (int k = 1; k <= kDrawSlices; k++) run = __any(d_stage == k) ? k : run;
                    if (p.dbg && threadIdx.x == 64) p.dbg[(size_t)(env0 / (64 / LPE)) * kDbgRow + 23] = run;
                }
#endif
            } else {
#ifdef MAPF_STAMPS
                if (K::kSlicedDraw && p.dbg && threadIdx.x == 64) p.dbg[(size_t)(env0 / (64 / LPE)) * kDbgRow + 23] = 0;
#endif
            }
        }
        if (io.obs || io.final_obs) obs_wave_step<K, LPE, MW>(p, io, l, lane, env0, ngroups);
        return;
    }

    bool full = ngroups == G && N == LPE;  // wave-uniform
    bool env_ok = grp < ngroups;
    const int env = env_ok ? env0 + grp : io.B - 1;
    bool is_agent = env_ok && a < N;

    // ---- loads: agent record, action, env scalars (and the obstacle rows in the single-wave build)
    // the slot dword comes with the state loads; the dense build fetches it only when an env finishes (32 B per env and
    // launch less)
    constexpr int kNsgMode = WPS != 0 ? 1 : 0;
    uint32_t nsg = kSlotInvalid;  // pre-drawn placement of the next episode
    LaneRaw raw;
    lane_issue(io.agents, io.bn8, (size_t)env * N + min(a, N - 1), raw);
    int act = (int)io.actions[(size_t)env * N + min(a, N - 1)];
    int sc[12];
    load_scal(io.scal, env, sc);
    if (kNsgMode == 0) nsg = slots_of(io.scal, io.B)[(size_t)env * N + min(a, N - 1)];
    RowRegs rr;
    if (!kDual) rows_issue<LPE>(io.grid_rows, io.H, lane, env0, ngroups, rr);
    __builtin_amdgcn_sched_barrier(0);  // nothing below may be hoisted between the loads
    warm_scalar_cache(pp, tail);
    __builtin_amdgcn_sched_barrier(0);
    const Lds l = carve_lds(io, lds_raw);
    if (__builtin_expect(io.env_mask != nullptr, 0)) {  // mapf_step_masked: masked-out envs are idle groups
        env_ok = env_ok && io.env_mask[env] != 0;
        is_agent = is_agent && env_ok;
        full = full && __all(env_ok);
    }
    Lane st;
    lane_unpack(raw, full || is_agent, st);
    act = (full || is_agent) ? act : 0;
    MAPF_STAMP(0);
    if (kDual) {
        wg_sync();  // B0
    } else {
        if (LPE >= 32 && (K::kMapAlways || io.use_map)) clear_cell_maps<LPE>(io, l.map, lane);
        rows_commit<LPE>(io.grid_rows, io.H, l.rows, lane, env0, ngroups, rr);
        wave_lds_sync();
    }
#ifdef MAPF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // attribute the load latency to phase 0->1
#endif
    MAPF_STAMP(1);
    bool records_stored = false;
    if (full && !__any(act < 0 || act > 4))
        records_stored = step_body<K, LPE, MW, true, kDual>(p, io, l, lane, env0, ngroups, act, st, sc, nsg, true, kNsgMode == 1);
    else
        step_body<K, LPE, MW, false, kDual>(p, io, l, lane, env0, ngroups, act, st, sc, nsg, false, kNsgMode == 1);
    if (!records_stored) {  // otherwise agents and counters left from inside the body
        if (full || is_agent)
            store_lane(io.agents, io.bn8, (size_t)env * N + a, st, l.rows + grp * (io.H + 2 * kRowPad) + kRowPad, io.col_pad, io.W);
        if (env_ok && a == 0) store_scal(io.scal, env, sc);
    }
    MAPF_STAMP(8);
#ifdef MAPF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // how long the trailing stores take to drain
#endif
    MAPF_STAMP(9);
    MAPF_STAMP_ENTRY_STORE();
}

// ================================================================================================
// Three-wave step kernel (round 3): the specialised finite shapes with full small groups (K::kSlicedDraw: N = lanes
// per env = 4 or 8), launches of at most three waves per SIMD.
//
// In k_step the state wave's chain is  loads -> B0 (rows from the observation wave) -> target cell -> move table
// through LDS -> B1 -> goal logic -> rewards -> records -> lock detector -> info -> stores, and the observation wave
// idles until B1 (stamps: profiles/r02/stamps_v22*.txt; B1 5.5 k cycles after wave entry, both waves end ~5 k cycles
// later).  Here the work after the moves is dealt to three waves that all start from their own loads:
//   wave 0 (state)  own copy of the obstacle rows (no B0), move table exchanged by DPP inside the lane group (no LDS
//                   round trip), publishes the moves (B1), then goal / reward logic, per-agent outputs, record store;
//   wave 1 (obs)    as in k_step: rows, B1, observation build, stream out;
//   wave 2 (aux)    loads the records and the env counters itself, takes the moves at B1, owns lock flags + detector
//                   (MA-env:577-606), info / counter outputs (:627-656), episode statistics, the MAY_FINISH hint, and
//                   runs the background draw slice (draw_slice) in its tail instead of in the observation wave's window
//                   before B1, which this layout closes.
// Workgroups that are not FAST (a ragged last workgroup, an invalid action) run the two-wave code of k_step: wave 2
// leaves before any barrier (a barrier only counts the waves that are still alive), wave 0 calls step_body.
// ================================================================================================
// value of lane (a ^ k) of the lane group, k = 1 .. LPE-1, for groups of 4 or 8 lanes: quad permutes and the
// half-row mirror (lane i <-> 7 - i = i ^ 7), all full-rate DPP moves
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_get(uint32_t v) {
    // (mov_dpp: no tied `old` operand, so no register copy in front of every DPP move; all source lanes of these
    //  patterns are live, bound_ctrl never matters)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int LPE>
__device__ __forceinline__ void group_xchg(const uint32_t v, uint32_t (&o)[LPE - 1]) {
    static_assert(LPE == 4 || LPE == 8 || LPE == 16, "DPP exchange is written for groups of 4, 8 or 16 lanes (one DPP row)");
    o[0] = dpp_get<0xB1>(v);  // quad_perm [1,0,3,2]: lane ^ 1
    o[1] = dpp_get<0x4E>(v);  // quad_perm [2,3,0,1]: lane ^ 2
    o[2] = dpp_get<0x1B>(v);  // quad_perm [3,2,1,0]: lane ^ 3
    if constexpr (LPE >= 8) {
        const uint32_t m = dpp_get<0x141>(v);  // row_half_mirror: lane ^ 7
        o[6] = m;
        o[5] = dpp_get<0xB1>(m);  // lane ^ 6
        o[4] = dpp_get<0x4E>(m);  // lane ^ 5
        o[3] = dpp_get<0x1B>(m);  // lane ^ 4
    }
    if constexpr (LPE == 16) {
        const uint32_t r = dpp_get<0x140>(v);  // row_mirror: lane ^ 15
        o[14] = r;
        o[13] = dpp_get<0xB1>(r);  // lane ^ 14
        o[12] = dpp_get<0x4E>(r);  // lane ^ 13
        o[11] = dpp_get<0x1B>(r);  // lane ^ 12
        const uint32_t h = dpp_get<0x141>(r);  // (lane ^ 7) ^ 15 = lane ^ 8
        o[7] = h;
        o[8] = dpp_get<0xB1>(h);   // lane ^ 9
        o[9] = dpp_get<0x4E>(h);   // lane ^ 10
        o[10] = dpp_get<0x1B>(h);  // lane ^ 11
    }
}

// resolve_moves (same rule, same outcome) without a table in LDS, without a loop and without a branch: the {old, target}
// pairs travel by DPP, every lane works out who stands on its target and who else wants it, and then EVERY lane replays
// the group's decisions in index order on one word per agent (broadcast inside the group by ds_swizzle: the LDS
// crossbar, no memory access): agent j moves iff no lower-index occupant of its target stayed and no lower-index
// contender got in (MA-env:502-526 is sequential; dependencies only point to lower indices).
template <int J, int LPE>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v) {  // lane J of every lane group
    constexpr int pattern = (J << 5) | (LPE == 16 ? 0x10 : (LPE == 8 ? 0x18 : 0x1C));  // bit mode: lane' = (lane & and_mask) | or_mask
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, pattern);
}
template <int LPE>
__device__ __forceinline__ uint32_t resolve_moves_dpp(int a, uint32_t old, uint32_t tgt) {
    static_assert(LPE == 4 || LPE == 8, "written for groups of 4 or 8 lanes");
    uint32_t x[LPE - 1];
    group_xchg<LPE>(old | (tgt << 16), x);
    uint32_t occ = 0, cont = 0;
#pragma unroll
    for (int k = 1; k < LPE; k++) {
        const uint32_t bit = 1u << (a ^ k);
        occ |= ((x[k - 1] & 0xFFFFu) == tgt) ? bit : 0u;
        cont |= ((x[k - 1] >> 16) == tgt) ? bit : 0u;
    }
    const uint32_t below = (1u << a) - 1u;
    // free to move as far as higher indices go: a target, and no higher-index agent standing on it (it has not had its
    // turn yet).  Word of this agent: bits 0-7 lower-index occupants of the target (bit 8: cannot move at all),
    // bits 16-23 lower-index contenders.
    const bool ok = tgt != kNoCell && (occ & ~below) == 0;
    const uint32_t mine = (ok ? (occ & below) : 0x100u) | ((cont & below) << 16);
    uint32_t d[LPE];
    d[0] = group_bcast<0, LPE>(mine);
    d[1] = group_bcast<1, LPE>(mine);
    d[2] = group_bcast<2, LPE>(mine);
    d[3] = group_bcast<3, LPE>(mine);
    if constexpr (LPE == 8) {
        d[4] = group_bcast<4, LPE>(mine);
        d[5] = group_bcast<5, LPE>(mine);
        d[6] = group_bcast<6, LPE>(mine);
        d[7] = group_bcast<7, LPE>(mine);
    }
    uint32_t M = 0;  // who has moved
#pragma unroll
    for (int j = 0; j < LPE; j++) {
        const uint32_t blocked = ((d[j] >> 16) & M) | ((d[j] & 0xFFFFu) & ~M);  // a contender got in | an occupant stayed
        M |= blocked == 0 ? (1u << j) : 0u;
    }
    return ((M >> a) & 1u) ? tgt : old;
}

// what an agent's step means for its own flags, shared by the state wave and the aux wave of k_step3 (finite mode,
// every lane an agent, no invalid action): MA-env:538-563 goal logic, :581-594 lock flags
struct AgentStep {
    bool moved, reached, completed, grs, cur_on_goal, progress, failed;
    int dist;
    uint32_t intended1;  // intended_next in the (+1,+1) encoding
};
__device__ __forceinline__ AgentStep agent_step(const Lane &st, int act, uint32_t cur) {
    AgentStep s;
    const uint32_t old = st.pos;
    const int dr = (act == 1) ? -1 : ((act == 3) ? 1 : 0);
    const int dc = (act == 2) ? 1 : ((act == 4) ? -1 : 0);
    const int tr = (int)(old >> 8) + dr, tc = (int)(old & 255u) + dc;
    s.intended1 = (uint32_t)(((tr + 1) << 8) | (tc + 1));
    s.moved = cur != old;
    s.reached = (st.flags & kFlagReached) != 0;
    s.completed = (st.flags & kFlagCompleted) != 0;
    s.grs = false;
    const bool on_goal = cur == st.goal;
    if (on_goal && !s.reached) {
        s.reached = true;
        s.completed = true;
        s.grs = true;
    }
    s.cur_on_goal = on_goal;
    s.progress = old != st.goal && on_goal;  // !prev_on_goal && cur_on_goal
    s.failed = act != 0 && !s.moved;
    s.dist = cell_l1(cur, st.goal);
    return s;
}

// How the step ends for an env (MA-env:668-690 and the auto-reset), from what ALL three waves of k_step3 hold -- the new
// cells, the goals, the step counter, the env's placement slot -- so that each of them decides for itself, identically,
// and the state wave publishes nothing but the moves: success check before the step limit; a finished env is re-placed
// from its pre-drawn slot (fast) or draws inline (slow); without a caller for the terminal observation the reset
// observation takes its place (subst).
struct EndDecision {
    int term, trunc;
    bool on_goal, done, do_reset, fast_reset, slow_reset, subst;
};
template <int LPE>
__device__ __forceinline__ EndDecision decide_end(const Io &io, int N, int lane, uint32_t cur, uint32_t goal,
                                                  int step_count, uint32_t nsg) {
    EndDecision d;
    d.on_goal = cur == goal;
    d.term = d.trunc = 0;
    if (__popcll(gballot<LPE>(d.on_goal, lane)) == N) {
        d.term = 1;
    } else if (step_count >= io.steps_per_episode) {
        d.term = 1;
        d.trunc = 1;
    }
    d.done = (d.term | d.trunc) != 0;
    d.do_reset = d.done && io.auto_reset;
    d.fast_reset = d.slow_reset = d.subst = false;
    if (__builtin_expect(__any(d.do_reset), 0)) {
        const bool slot_ok = gballot<LPE>(!slot_word_valid(nsg), lane) == 0;
        d.fast_reset = d.do_reset && slot_ok;
        d.slow_reset = d.do_reset && !slot_ok;
        d.subst = d.fast_reset && io.final_obs == nullptr;
    }
    return d;
}
// which tensor a group's observation goes to (0 io.obs, 1 io.final_obs, 2 nowhere) and the kObsW* flags of an entry
__device__ __forceinline__ uint32_t obs_flags_for(const Io &io, const EndDecision &d, bool pressure_prev) {
    int sel = io.obs ? 0 : 2;
    uint32_t w = kObsWAgent | (pressure_prev ? kObsWPressure : 0u) | kObsWFast;
    if (d.do_reset) sel = io.final_obs ? 1 : ((d.subst && io.obs) ? 0 : 2);
    if (d.subst) w = (w & ~kObsWPressure) | kObsWFinal;
    w |= (d.slow_reset ? kObsWReset : 0u) | ((d.fast_reset && !d.subst) ? kObsWResetFast : 0u);
    return w | ((uint32_t)sel << kObsWSelShift);
}

// ---- wave 1 of k_step3 -------------------------------------------------------------------------------------------
// The observation wave: waits for the moves (B1), decides for itself how the step ends for each of its envs
// (decide_end), completes the entries the state wave published -- flags; for an env whose reset observation replaces the
// terminal one: everybody on its start, the new goals (step_body: subst) -- and walks them (obs_wave_step).
// (Tried and dropped, round 3: building the bit planes of a (V+2) x (V+2) superset window around the OLD position before
// B1 -- obstacles, goals, the others on their old cells, per lower-index agent the two-bit difference between old and
// wanted cell -- and only patching after B1 (xor of the movers' differences, window shifted by the agent's own move).
// It takes the table walk off the path behind B1 but costs ~250 more vector instructions per wave than it saves, and
// after B1 this kernel is bound by instruction issue on the SIMD, not by any wave's latency: 5.67 against 5.40 us.)
template <class K, int LPE, int MW>
__device__ __forceinline__ void obs3_wave(const Params &p, const Io &io, const Lds &l, const int lane, const int env0,
                                          const uint2 hot, const uint32_t nsg, const int step_count_in, float *gd_lut) {
    constexpr int G = 64 / LPE;
    const int N = K::N(p);
    const uint32_t goal_real = hot.x >> 16;
    const bool pressure_real = ((hot.y >> 16) & kFlagPressure) != 0;
    wg_sync();  // B1: the moves (and, LDS being drained before the barrier, the caller's goal-delta table)
    // After B1 the workgroup's SIMDs are issue-bound (three busy waves each).  The observation wave goes first: its
    // stream is what the launch ends with, and the aux wave's work then fills the slots under that stream's drain.
    __builtin_amdgcn_s_setprio(3);  // (measured, us per step staggered / synchronised: obs 3, aux 1, slice 0: 5.52 / 5.08;
                                    //  all at 2: 5.76 / 5.31; obs 3, aux 2, slice 1: 5.87 / 5.40; obs 1, aux 3: 5.61 / 5.40)
    const uint4 ent = l.otab[lane];
    const uint32_t cur = ent.y >> 16;
    const EndDecision dec = decide_end<LPE>(io, N, lane, cur, goal_real, step_count_in + 1, nsg);
    const uint32_t w = obs_flags_for(io, dec, pressure_real);
    uint4 mine = make_uint4(ent.x, ent.y, 0u, w);  // (word y's upper half, the real new cell, is the aux wave's and stays)
    if (__builtin_expect(__any(dec.fast_reset), 0)) {
        const uint32_t rs = nsg, rs_pos = rs & 0xFFFFu;
        if (dec.fast_reset) mine.z = rs;
        if (dec.subst) mine = make_uint4(rs_pos | (rs_pos << 16), (rs >> 16) | (cur << 16), rs, w);
    }
    l.otab[lane] = mine;
    wave_lds_sync();
    obs_wave_step<K, LPE, MW>(p, io, l, lane, env0, G, true, gd_lut);
}

// ---- wave 1 of k_step3, prepared before B1 (round 4; grids of at most 64 - 2 kRowPad columns) ------------------------------
// Before B1 the observation wave knows everything about its agent's observation except whether the move succeeds and who
// else moved: the agent ends the step on its old cell or on its target.  While the state wave resolves the moves, this
// wave builds for BOTH cells the obstacle / goal / occupancy windows -- from bit rows: the obstacle rows as loaded, and two
// row sets of its own into which every agent ors its goal and its OLD cell -- plus its own goal's bit and the goal delta.
// After B1 it picks one of the two, and what is left of the staggered view of MA-env:528 (agents <= i at their new cell,
// agents > i at their old one) is a toggle of two window bits per successful mover of LOWER index anywhere near: every
// move of the sequential loop takes an agent off a cell that held it and puts it on a cell that was empty at that moment,
// so the occupancy after turn i is the initial occupancy XOR the toggles of the turns <= i.  Which lower-index agents can
// matter at all (old cell within sr + 2 of mine: both of us move one cell at most) is a 16-bit mask made before B1 too;
// the group's ballot of who moved cuts it down to a handful of table reads per wave.
template <int MW>
struct ObsCand {
    WMask<MW> obst, gls, occ, own;
    float gd_r, gd_c;
};
// LDS bit rows of k_step3 (grids with sentinel columns), behind the aux wave's staging and gd_lut (relative to
// Io::lds_map_off): [goals | old cells | intents][G][H + 2 kRowPad], bit (column + col_pad) of row (row + kRowPad); cleared
// and filled by the aux wave before B1
// which instantiations of k_step3 hold the bit-row code at all (the host's choice, mapf_create, is the same): 16-lane
// groups with windows up to 7 x 7.  (Groups of 4 and 8 lanes measured slower with the rows -- c3 5.63 against 5.42 us
// staggered, c2 4.08 against 3.73: short table walks, cheap pair pass -- and merely compiling the paths in cost c3 4 %.)
template <int LPE, int MW>
constexpr bool k3_rows() { return LPE == 16 && MW <= 64; }
__host__ __device__ constexpr int obs_rows_lds_off() { return 3072; }
__host__ __device__ constexpr int obs_rows_lds_bytes(int G, int H) { return 3 * G * (H + 2 * kRowPad) * 8; }
// the observation wave's half: what only needs the obstacle rows and the agent's own goal
template <class K, int MW>
__device__ __forceinline__ void obs_candidate_static(const Params &p, const Io &io, const uint64_t *obrows, const float *gd_lut,
                                                     uint32_t cell, uint32_t goal, ObsCand<MW> &c) {
    constexpr int MAXV = MW == 32 ? 5 : (MW == 64 ? 7 : 11);
    const int V = K::V(p), sr = K::sr(p);
    const int myr = (int)(cell >> 8), myc = (int)(cell & 255u);
    const int r0 = myr - sr, c0 = myc - sr;
    const int sh = c0 + io.col_pad;  // 0 .. 63 - V: the rows carry col_pad = kRowPad >= sr bits below column 0
    const uint32_t vm = (1u << V) - 1u;
    uint64_t ro[MAXV];
#pragma unroll
    for (int d = 0; d < MAXV; d++) ro[d] = (d < V) ? obrows[r0 + d] : 0ull;
    c.obst.clear(); c.own.clear();
#pragma unroll
    for (int d = 0; d < MAXV; d++) {
        if (d < V) c.obst.or_row((uint32_t)(ro[d] >> sh) & vm, d * V);
    }
    window_set<MW>(c.own, goal, r0, c0, V);
    const int gdr = (int)((goal >> 8) & 255u) - myr, gdc = (int)(goal & 255u) - myc;
    c.gd_r = gd_lut[gdr + 63];
    c.gd_c = gd_lut[128 + gdc + 63];
}
// the aux wave's half: everybody's goals and everybody's old cell in the window around `cell`
template <class K, int MW>
__device__ __forceinline__ void obs_candidate_rows(const Params &p, const Io &io, const uint64_t *goalb, const uint64_t *occO,
                                                   uint32_t cell, WMask<MW> &gls, WMask<MW> &occ) {
    constexpr int MAXV = MW == 32 ? 5 : (MW == 64 ? 7 : 11);
    const int V = K::V(p), sr = K::sr(p);
    const int r0 = (int)(cell >> 8) - sr, sh = (int)(cell & 255u) - sr + io.col_pad;
    const uint32_t vm = (1u << V) - 1u;
    MAPF_CHK(p, r0 >= -kRowPad && r0 + V <= io.H + kRowPad && sh >= 0 && sh + V <= 64, 11, -1, cell);
    uint64_t rg[MAXV], rq[MAXV];
#pragma unroll
    for (int d = 0; d < MAXV; d++) {  // (all reads of the window issue back to back)
        rg[d] = (d < V) ? goalb[r0 + d] : 0ull;
        rq[d] = (d < V) ? occO[r0 + d] : 0ull;
    }
    gls.clear(); occ.clear();
#pragma unroll
    for (int d = 0; d < MAXV; d++) {
        if (d < V) {
            gls.or_row((uint32_t)(rg[d] >> sh) & vm, d * V);
            occ.or_row((uint32_t)(rq[d] >> sh) & vm, d * V);
        }
    }
}
// the two cells an agent can end the step on: its old cell, or the neighbour its action names when that is a free cell
// of the grid (the pass bits of the hot plane, as state3_wave)
__device__ __forceinline__ uint32_t step_target_or_old(uint32_t old, uint32_t pass, int act) {
    const int dr = (act == 1) ? -1 : ((act == 3) ? 1 : 0);
    const int dc = (act == 2) ? 1 : ((act == 4) ? -1 : 0);
    const bool want = act != 0 && ((pass >> ((act - 1) & 3)) & 1u) != 0;
    return want ? (uint32_t)(((int)old + (dr << 8) + dc) & 0xFFFF) : old;
}
// ---- intent blocking (MA-env:608-623) from a bit row set ------------------------------------------------------------------
// An agent that has reached its goal and did not move is "blocking" when some OTHER agent that has not reached its goal
// (after this step's goal logic) intended to enter its cell.  Instead of exchanging every agent's intended cell inside the
// group, the aux wave ors the intents into bit rows BEFORE B1 -- an agent publishes iff it has not reached its goal before
// the step and does not stand on it afterwards, which is known ahead of the moves unless its old cell or its target IS its
// goal -- and whoever needs the flag (state wave: per-agent outputs; aux wave: counters) reads one row behind B1 and adds,
// by broadcast, the few agents whose publishing depended on the move.  (An agent that may be blocking has reached its goal
// and publishes nothing itself, so its own intent never counts.)
struct IntentOf {
    uint32_t cell1;  // intended cell in the (+1, +1) encoding of AgentStep::intended1
    int row, bit;    // its place in the rows (row index relative to row 0)
    bool sure, unsure;
};
__device__ __forceinline__ IntentOf intent_of(const Io &io, uint32_t old, uint32_t goal, uint32_t flags0, uint32_t pass, int act) {
    IntentOf t;
    const int dr = (act == 1) ? -1 : ((act == 3) ? 1 : 0);
    const int dc = (act == 2) ? 1 : ((act == 4) ? -1 : 0);
    const int tr = (int)(old >> 8) + dr, tc = (int)(old & 255u) + dc;
    t.cell1 = (uint32_t)(((tr + 1) << 8) | (tc + 1));
    t.row = tr;
    t.bit = tc + io.col_pad;  // >= col_pad - 1 >= 0
    const bool want = act != 0 && ((pass >> ((act - 1) & 3)) & 1u) != 0;
    const uint32_t tgt = want ? (uint32_t)((tr << 8) | tc) : old;
    const bool unreached = (flags0 & kFlagReached) == 0;
    t.unsure = unreached && (old == goal || tgt == goal);
    t.sure = unreached && !t.unsure;
    return t;
}
template <int LPE>
__device__ __forceinline__ bool intent_blocks(const Io &io, const uint64_t *irow0, int lane, const IntentOf &t, uint32_t goal,
                                              uint32_t cur) {
    bool blocks = ((irow0[cur >> 8] >> ((cur & 255u) + io.col_pad)) & 1ull) != 0;
    const bool late = t.unsure && cur != goal;  // publishes after all
    uint64_t u = fold_groups<LPE>(__ballot(late));
    const uint32_t mycell1 = cur + 0x0101u;
    while (u) {  // (rare)
        const int j = (int)__builtin_ctzll(u);
        u &= u - 1;
        blocks |= gshfl<LPE>(late ? t.cell1 : 0xFFFFFFFFu, j) == mycell1;
    }
    return blocks;
}

// An env that reaches its step limit in this step, is re-placed from its pre-drawn slot and has no taker for the terminal
// observation shows its RESET observation instead (decide_end: subst) -- whatever the moves turn out to be.  Both preparing
// waves see that coming from the same inputs and prepare that observation (everybody on its new start, the new goals)
// in place of the two outcomes of the move.  (An env that ends by success before the limit is not foreseen: the
// observation wave falls back to the table walk for it.)
template <int LPE>
__device__ __forceinline__ bool foresee_subst(const Io &io, int lane, int step_count_in, uint32_t nsg) {
    const bool at_limit = step_count_in + 1 >= io.steps_per_episode && io.auto_reset && io.final_obs == nullptr;
    return at_limit && gballot<LPE>(!slot_word_valid(nsg), lane) == 0;
}
// aux wave, before B1: the bit rows (cleared by this wave at entry) and its half of both candidates -> LDS
template <class K, int LPE, int MW>
__device__ __forceinline__ void aux3_prepare_rows(const Params &p, const Io &io, uint64_t *brows, const int lane, const uint2 hot,
                                                  const int act, const int step_count_in, const uint32_t nsg, const int env) {
    constexpr int G = 64 / LPE;
    const int grp = lane / LPE;
    (void)env;
    const int RS = io.H + 2 * kRowPad;
    (void)p;
    const bool fs = foresee_subst<LPE>(io, lane, step_count_in, nsg);
    const uint32_t old = fs ? (nsg & 0xFFFFu) : (hot.x & 0xFFFFu), goal = fs ? (nsg >> 16) : (hot.x >> 16);
    uint64_t *goalb = brows + grp * RS + kRowPad, *occO = brows + (G + grp) * RS + kRowPad;
    MAPF_CHK(p, (int)(goal >> 8) < io.H && (int)(old >> 8) < io.H && (int)((goal & 255u) + io.col_pad) < 64 && (int)((old & 255u) + io.col_pad) < 64,
             11, env, (goal << 16) | old);
    atomicOr(reinterpret_cast<unsigned long long *>(&goalb[goal >> 8]), 1ull << ((goal & 255u) + io.col_pad));
    atomicOr(reinterpret_cast<unsigned long long *>(&occO[old >> 8]), 1ull << ((old & 255u) + io.col_pad));
    {   // intents of the step itself (never of the placement a foreseen reset shows)
        const IntentOf t = intent_of(io, hot.x & 0xFFFFu, hot.x >> 16, (hot.y >> 16) & 0xFFu, hot.y >> 24, act);
        uint64_t *irow0 = brows + (2 * G + grp) * RS + kRowPad;
        MAPF_CHK(p, !t.sure || (t.row >= -kRowPad && t.row < io.H + kRowPad && t.bit >= 0 && t.bit < 64), 11, env, (t.row << 8) | (t.bit & 255));
        if (t.sure) atomicOr(reinterpret_cast<unsigned long long *>(&irow0[t.row]), 1ull << t.bit);
    }
}
// The per-agent outputs of a step -- rewards (MA-env:538-563, :668-690), {blocking, goal_reached_step}, done flags, the hot
// plane incl. the image of an env re-placed from its slot -- by the STATE wave (round 4, with the intent bit rows): behind B1
// it has nothing else to do, and the aux wave's chain (lock detector, info row, counters, history planes) is what the
// launch ends with.  Same arithmetic as aux3_wave's block for workgroups without the rows.
template <class K, int LPE>
__device__ __forceinline__ void state3_outputs(const Params &p, const Io &io, const Lds &l, const uint64_t *irow0, const int lane,
                                               const int env0, const int act, const uint2 hot, const uint32_t cur,
                                               const EndDecision &dec, const uint32_t nsg) {
    const int grp = lane / LPE, a = lane % LPE;
    const int N = K::N(p);
    const size_t idx0 = (size_t)env0 * N;
    const uint64_t *myrows = l.rows + grp * (io.H + 2 * kRowPad) + kRowPad;  // (the observation wave's; valid after B1)
    Lane st;
    st.pos = hot.x & 0xFFFFu;
    st.goal = hot.x >> 16;
    st.start = hot.y & 0xFFFFu;
    st.flags = (hot.y >> 16) & 0xFFu;
    const AgentStep as = agent_step(st, act, cur);
    const IntentOf t = intent_of(io, st.pos, st.goal, st.flags, hot.y >> 24, act);
    const bool blocks = intent_blocks<LPE>(io, irow0, lane, t, st.goal, cur);  // (every lane: it ballots and broadcasts)
    const bool blocking = as.reached && !as.moved && blocks;
    const bool done = dec.done;
    const float term_reward = !done ? 0.0f : (!dec.trunc ? 1.0f : (dec.on_goal ? 0.0f : -1.0f));
    const float reward = (as.grs ? 0.5f : 0.0f) + term_reward;
    if (io.rewards) store_wt4(io.rewards + idx0, lane, __float_as_uint(reward));
    if (io.info_agent)  // {blocking, goal_reached_step} as two bytes
        store_wt2(reinterpret_cast<uchar2 *>(io.info_agent) + idx0, lane, (uint16_t)((blocking ? 1u : 0u) | (as.grs ? 0x100u : 0u)));
    if (a == 0) {
        if (io.terminated) (io.terminated + env0)[grp] = (uint8_t)dec.term;
        if (io.truncated) (io.truncated + env0)[grp] = (uint8_t)dec.trunc;
    }
    Lane img;
    img.pos = cur;
    img.goal = st.goal;
    img.start = st.start;
    img.flags = (as.reached ? kFlagReached : 0) | (as.completed ? kFlagCompleted : 0) | (blocking ? kFlagPressure : 0);
    if (__builtin_expect(__any(dec.fast_reset), 0)) {  // re-placed envs store the image reset() leaves (MA-env:440-455)
        if (dec.fast_reset) {
            img.start = nsg & 0xFFFFu;
            img.goal = nsg >> 16;
            img.pos = img.start;
            img.flags = 0u;
            (slots_of(io.scal, io.B) + idx0)[lane] = kSlotInvalid;  // consumed
        }
    }
    // (an env that draws inline -- slow reset -- gets its hot plane below, after the draw)
    if (!dec.slow_reset) store_lane_hot(io.agents + idx0, (size_t)lane, img, agent_pass_bits(myrows, img.pos, io.col_pad, io.W));
}

template <int MW>
__device__ __forceinline__ void window_toggle(WMask<MW> &m, bool on, uint32_t cell, int r0, int c0, int V) {
    const int r = (int)((cell >> 8) & 255u) - r0, c = (int)(cell & 255u) - c0;
    m.toggle_if(on && max((unsigned)r, (unsigned)c) < (unsigned)V, __mul24(r, V) + c);
}
template <class K, int LPE, int MW>
__device__ __forceinline__ void obs3_wave_prepared(const Params &p, const Io &io, const Lds &l, uint64_t *brows, const int lane,
                                                   const int env0, const uint2 hot, const int act, const uint32_t nsg,
                                                   const int step_count_in, float *gd_lut) {
    constexpr int G = 64 / LPE;
    constexpr int MAXV = MW == 32 ? 5 : (MW == 64 ? 7 : 11);
    const int grp = lane / LPE, a = lane % LPE;
    const int N = K::N(p), H = io.H, V = K::V(p), sr = K::sr(p);
    const int RS = H + 2 * kRowPad;
    const uint32_t old = hot.x & 0xFFFFu, goal_real = hot.x >> 16;
    const bool pressure_real = ((hot.y >> 16) & kFlagPressure) != 0;
    const uint64_t *myrows = l.rows + grp * RS + kRowPad;
    // ---- before B1 ----
    const bool fs = foresee_subst<LPE>(io, lane, step_count_in, nsg);  // (the reset observation instead: see there)
    uint32_t near = 0;  // lower-index agents of my env whose old cell lies within sr + 2 (rows and columns) of mine
    {
        uint32_t xo[LPE - 1];
        group_xchg<LPE>(old, xo);
        const int R = sr + 2;
        const unsigned rlo = (old >> 8) - R, clo = (old & 255u) - R;
#pragma unroll
        for (int k = 1; k < LPE; k++) {
            const unsigned tr = (xo[k - 1] >> 8) - rlo, tc = (xo[k - 1] & 255u) - clo;
            near |= (max(tr, tc) <= 2u * R) ? (1u << (a ^ k)) : 0u;
        }
        near &= (1u << a) - 1u;
    }
    asm volatile("" ::"v"(near));  // (in a register BEFORE the barrier: left alone, the compiler sinks this to its first use)
    MAPF_STAMP_W1(26);
    wg_sync();  // B1: the moves
    __builtin_amdgcn_s_setprio(3);  // (as obs3_wave; with the bit rows, obs / aux: 3 / 1 6.32 us staggered, 2 / 2 6.47, 3 / 2 6.52, 2 / 3 6.54, 1 / 3 6.63)
    MAPF_STAMP_W1(11);
    const uint4 ent = l.otab[lane];
    const uint32_t cur = ent.y >> 16;
    const EndDecision dec = decide_end<LPE>(io, N, lane, cur, goal_real, step_count_in + 1, nsg);
    const uint32_t w = obs_flags_for(io, dec, pressure_real);
    const bool any_fast_reset = __any(dec.fast_reset);
    if (__builtin_expect(any_fast_reset, 0)) {  // the entries of re-placed groups, as obs3_wave leaves them (second pass / subst below)
        uint4 mine = make_uint4(ent.x, ent.y, 0u, w);
        const uint32_t rs = nsg, rs_pos = rs & 0xFFFFu;
        if (dec.fast_reset) mine.z = rs;
        if (dec.subst) mine = make_uint4(rs_pos | (rs_pos << 16), (rs >> 16) | (cur << 16), rs, w);
        // (the loop below reads word x of LOWER-index entries of groups that are not substituted: unchanged for those)
        if (dec.fast_reset) l.otab[lane] = mine;
    }
    const bool moved = cur != old;
    const bool active = !dec.subst || fs;       // (a substitution that was not foreseen: the table walk below)
    const bool staggered = !dec.subst;          // (the reset observation shows everybody where the placement puts them)
    float *srow = l.stage + (size_t)(grp * N + a) * K::L(p);
    {
        ObsCand<MW> c;
        const uint32_t cell = fs ? (nsg & 0xFFFFu) : cur, goal_shown = fs ? (nsg >> 16) : goal_real;
        obs_candidate_static<K, MW>(p, io, myrows, gd_lut, cell, goal_shown, c);
        obs_candidate_rows<K, MW>(p, io, brows + grp * RS + kRowPad, brows + (G + grp) * RS + kRowPad, cell, c.gls, c.occ);
        const WMask<MW> obst = c.obst, own = c.own, gls = c.gls;
        WMask<MW> occ = c.occ;
        const float gd_r = c.gd_r, gd_c = c.gd_c;
        const int r0 = (int)(cur >> 8) - sr, cc0 = (int)(cur & 255u) - sr;
        const uint32_t M = (uint32_t)gballot_n<LPE>(moved, lane);
        const uint4 *otabg = l.otab + grp * LPE;
        uint32_t todo = staggered ? (near & M) : 0u;
        while (__any(todo != 0u)) {
            const bool on = todo != 0u;
            const int j = on ? (int)__builtin_ctz(todo) : 0;
            todo &= todo - 1u;
            const uint32_t e = otabg[j].x;  // old | new << 16 of a lower-index agent that moved
            window_toggle<MW>(occ, on, e & 0xFFFFu, r0, cc0, V);
            window_toggle<MW>(occ, on, e >> 16, r0, cc0, V);
        }
        window_toggle<MW>(occ, moved && staggered, old, r0, cc0, V);  // the cell I left myself
        MAPF_STAMP_W1(27);
        if (active) {
            occ.clear_bit(sr * V + sr);  // my own cell: "occ not in (UNASSIGNED, self)" MA-env:735
            emit_obs_row<K, MW, MAXV>(p, srow, obst, occ, gls, own, gd_r, gd_c, pressure_real && staggered);
        }
    }
    PairOut po;
    if (__builtin_expect(__any(dec.subst && !fs), 0)) {
        wave_lds_sync();
        const uint4 e2 = l.otab[lane];
        observe<K, LPE, MW, kObsEmit, false>(p, io, myrows, l.otab + grp * LPE, srow, dec.subst && !fs, a, e2.x >> 16, e2.y & 0xFFFFu, true,
                                             false, 0, po, nullptr, gd_lut);
    }
    wave_lds_sync();
    MAPF_STAMP_W1(12);
    const bool any_slow = __any(dec.slow_reset);
    const bool any_fast = __any(dec.fast_reset && !dec.subst);
    if (!any_fast && !any_slow) {
        if (io.obs) flush_obs_full<K, LPE>(p, io, io.obs, l.stage, lane, env0);
    } else {
        flush_obs<K, LPE>(p, io, l.stage, lane, env0, G, (int)((w >> kObsWSelShift) & 3u));
    }
    MAPF_STAMP_W1(13);
    if (any_fast) {  // second pass, as obs_wave_step: the reset observation of groups whose terminal one just went to final_obs
        const bool fr = dec.fast_reset && !dec.subst;
        const uint32_t rpos = nsg & 0xFFFFu, rgoal = nsg >> 16;
        uint4 *mine = l.otab + grp * LPE;
        wave_lds_sync();
        if (fr) mine[a] = make_uint4(rpos | (rpos << 16), rgoal, nsg, w);
        wave_lds_sync();
        observe<K, LPE, MW, kObsEmit, false>(p, io, myrows, mine, srow, fr, a, rpos, rgoal, true, false, 0, po, nullptr, gd_lut);
        wave_lds_sync();
        flush_obs<K, LPE>(p, io, l.stage, lane, env0, G, fr ? 0 : 2);
    }
    // The per-agent outputs (rewards, {blocking, goal_reached_step}, done flags, the hot plane), behind the stream: the aux
    // wave's chain is what the launch ends with, and the state wave may have a slice of the background draw to run.  (Measured,
    // us per step staggered / in phase: here 6.31 / 6.23; by the state wave right behind B1 6.47 / 6.16; by the state wave
    // unless its env's slot says a slice may be due, else here: 6.58 / 6.11; by the state wave unless an env of the workgroup
    // is re-placed in this step, else here: 6.41 / 6.09; round 3's kernel: 6.67 / 6.58.)
    state3_outputs<K, LPE>(p, io, l, brows + (2 * G + grp) * RS + kRowPad, lane, env0, act, hot, cur, dec, nsg);
#ifdef MAPF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MAPF_STAMP_W1(14);
#endif
    if (any_slow) wg_sync();  // B2
}

// ---- wave 2 of k_step3 -------------------------------------------------------------------------------------------
// Everything of the step that follows the moves and is not the observation: goal / reward logic (MA-env:538-563), lock
// flags and detector (:577-606), intent blocking (:608-623), info (:627-656), termination outputs (:668-690), the state
// image (all four planes, counters) incl. the image of a re-placed env, episode statistics, the MAY_FINISH hint.
template <class K, int LPE, int MW>
__device__ __forceinline__ void aux3_wave(const Params &p, const Io &io, const Lds &l, unsigned char *aux_lds, const int lane,
                                          const int env0, const int act, const LaneRaw &raw, int *sc, const uint32_t nsg,
                                          uint64_t *brows) {
    constexpr int G = 64 / LPE;
    using gm_t = typename GMask<LPE>::type;
    const int grp = lane / LPE, a = lane % LPE;
    const int env = env0 + grp;
    const int N = K::N(p);
    const uint32_t flags = K::flags(p);
    const bool lock_on = (flags & MAPF_FLAG_LOCK_METRICS) != 0;
    const int lw = K::lw(p), dw = K::dw(p), ring_stride = K::ring_stride(p);
    const bool dist_in_rec = lw <= 16;
    const size_t idx0 = (size_t)env0 * N;  // wave-uniform: the stores below are <uniform base> + <lane> (no 64-bit vector math)
    const uint64_t *myrows = l.rows + grp * (io.H + 2 * kRowPad) + kRowPad;  // (the observation wave's; valid after B1)

    MAPF_STAMP_W2(21);
    if constexpr (k3_rows<LPE, MW>()) {
        if (brows) aux3_prepare_rows<K, LPE, MW>(p, io, brows, lane, raw.h, act, sc[MAPF_CTR_STEP_COUNT], nsg, env);
    }
    MAPF_STAMP_W2(20);
    wg_sync();  // B1: the moves are published
    __builtin_amdgcn_s_setprio(1);  // (behind the observation wave: obs3_wave)
    MAPF_STAMP_W2(22);
    // (Tried: unpacking behind the barrier, with the loaded registers re-defined there by an empty asm so that this wave
    // arrives without waiting for its history planes and counters -- as written the compiler hoists the step counter's + 1
    // above the barrier and with it a wait for every load.  Slower, 5.19 against 5.09 us: the wait then sits at the head
    // of this wave's post-B1 chain, which is the one the launch ends with.)
    Lane st;
    lane_unpack(raw, true, st);
    const uint32_t cur = l.otab[lane].y >> 16;  // (the rest of the entry is the observation wave's)
    sc[MAPF_CTR_STEP_COUNT] += 1;  // MA-env:475
    const EndDecision dec = decide_end<LPE>(io, N, lane, cur, st.goal, sc[MAPF_CTR_STEP_COUNT], nsg);
    const bool done = dec.done, do_reset = dec.do_reset;
    const AgentStep as = agent_step(st, act, cur);
    const int goals_step = __popcll(gballot<LPE>(as.grs, lane));
    sc[MAPF_CTR_GOALS_REACHED_TOTAL] += goals_step;

    int delta = 0;
    bool dl_ok = false, ll_ok = false;
    if (lock_on) {  // _append_lock_history_step MA-env:374-387 (same arithmetic as step_body)
        const int t = sc[MAPF_CTR_HIST_ROWS];
        const int count = min(t + 1, K::hs(p));
        dl_ok = count >= dw;
        ll_ok = count >= lw;
        st.moved = (st.moved << 1) | (as.moved ? 1ull : 0ull);
        st.failed = (st.failed << 1) | (as.failed ? 1ull : 0ull);
        st.progress = (st.progress << 1) | (as.progress ? 1ull : 0ull);
        int d_old = as.dist;
        if (dist_in_rec) {
            st.dist.w = (st.dist.w << 8) | (st.dist.z >> 24);
            st.dist.z = (st.dist.z << 8) | (st.dist.y >> 24);
            st.dist.y = (st.dist.y << 8) | (st.dist.x >> 24);
            st.dist.x = (st.dist.x << 8) | (uint32_t)as.dist;
            const int ob = lw - 1, od = ob >> 2;
            const uint32_t wd = od == 0 ? st.dist.x : (od == 1 ? st.dist.y : (od == 2 ? st.dist.z : st.dist.w));
            if (ll_ok) d_old = (int)((wd >> ((ob & 3) * 8)) & 0xFFu);
        } else {
            int16_t *ring = io.dist_ring + ((size_t)env * N + a) * ring_stride;
            const int slot_new = t % lw;
            const int slot_old = (slot_new + 1 == lw) ? 0 : slot_new + 1;
            if (ll_ok) d_old = ring[slot_old];
            ring[slot_new] = (int16_t)as.dist;
        }
        delta = d_old - as.dist;
        sc[MAPF_CTR_HIST_ROWS] = t + 1;
    }
    sc[MAPF_CTR_MAY_FINISH] = (gballot<LPE>(as.dist > 1, lane) == 0 ||
                               sc[MAPF_CTR_STEP_COUNT] + 1 >= io.steps_per_episode) ? 1 : 0;

    // pair pass (MA-env:389-398 neighbour sets, :608-623 intent blocking) on DPP-exchanged words:
    // xa = new cell | reached << 16 | (distance delta + 256) << 17,  xz = intended cell (+1,+1) or ~0 once reached
    uint32_t xa[LPE - 1];
    group_xchg<LPE>(cur | ((as.reached ? 1u : 0u) << 16) | ((uint32_t)(delta + 256) << 17), xa);
    const uint32_t mycell1 = cur + 0x0101u;
    gm_t nbr = 0;
    int sum_biased = 0;
    bool blocks = false;
    const bool rows = k3_rows<LPE, MW>() && brows != nullptr;
    if (rows) {  // intents from the bit rows (intent_blocks); the state wave stores the per-agent outputs (state3_outputs)
#pragma unroll
        for (int k = 1; k < LPE; k++) {
            const int d = cell_l1(xa[k - 1] & 0xFFFFu, cur);
            const bool isn = (unsigned)(d - 1) < (unsigned)K::nearby(p);
            nbr |= isn ? ((gm_t)1 << (a ^ k)) : 0;
            sum_biased += isn ? (int)(xa[k - 1] >> 17) : 0;
        }
        const IntentOf t = intent_of(io, st.pos, st.goal, st.flags, raw.h.y >> 24, act);
        blocks = intent_blocks<LPE>(io, brows + (2 * G + grp) * (io.H + 2 * kRowPad) + kRowPad, lane, t, st.goal, cur);
    } else {
        uint32_t xz[LPE - 1];
        group_xchg<LPE>(as.reached ? 0xFFFFFFFFu : as.intended1, xz);
#pragma unroll
        for (int k = 1; k < LPE; k++) {
            const int d = cell_l1(xa[k - 1] & 0xFFFFu, cur);
            const bool isn = (unsigned)(d - 1) < (unsigned)K::nearby(p);
            nbr |= isn ? ((gm_t)1 << (a ^ k)) : 0;
            sum_biased += isn ? (int)(xa[k - 1] >> 17) : 0;
            blocks |= xz[k - 1] == mycell1;
        }
    }
    const int sum_delta = delta + sum_biased - 256 * __popc((uint32_t)nbr);
    const bool blocking = as.reached && !as.moved && blocks;

    // ---- what does not need the lock detector leaves first: rewards, per-agent info, done flags, the hot plane ----
    if (!rows) {
        const float term_reward = !done ? 0.0f : (!dec.trunc ? 1.0f : (dec.on_goal ? 0.0f : -1.0f));
        const float reward = (as.grs ? 0.5f : 0.0f) + term_reward;
        if (io.rewards) store_wt4(io.rewards + idx0, lane, __float_as_uint(reward));
        if (io.info_agent)  // {blocking, goal_reached_step} as two bytes
            store_wt2(reinterpret_cast<uchar2 *>(io.info_agent) + idx0, lane, (uint16_t)((blocking ? 1u : 0u) | (as.grs ? 0x100u : 0u)));
        if (a == 0) {
            if (io.terminated) (io.terminated + env0)[grp] = (uint8_t)dec.term;
            if (io.truncated) (io.truncated + env0)[grp] = (uint8_t)dec.trunc;
        }
        Lane img;
        img.pos = cur;
        img.goal = st.goal;
        img.start = st.start;
        img.flags = (as.reached ? kFlagReached : 0) | (as.completed ? kFlagCompleted : 0) | (blocking ? kFlagPressure : 0);
        if (__builtin_expect(__any(dec.fast_reset), 0)) {  // re-placed envs store the image reset() leaves (MA-env:440-455)
            if (dec.fast_reset) {
                img.start = nsg & 0xFFFFu;
                img.goal = nsg >> 16;
                img.pos = img.start;
                img.flags = 0u;
                (slots_of(io.scal, io.B) + idx0)[lane] = kSlotInvalid;  // consumed
            }
        }
        // (an env that draws inline -- slow reset -- gets its hot plane from the state wave, which makes the draw)
        if (!dec.slow_reset) store_lane_hot(io.agents + idx0, (size_t)lane, img, agent_pass_bits(myrows, img.pos, io.col_pad, io.W));
    }
    MAPF_STAMP_W2(24);

    int deadlock = 0, livelock = 0, dl_event = 0, ll_event = 0;
    if (lock_on) {  // MA-env:400-438, as step_body
        const uint64_t mdw = dw >= 64 ? ~0ull : ((1ull << dw) - 1ull);
        const uint64_t mlw = lw >= 64 ? ~0ull : ((1ull << lw) - 1ull);
        const gm_t members = nbr | ((gm_t)1 << a);
        const bool focal = !as.cur_on_goal && __popc((uint32_t)nbr) >= K::min_nbrs(p);
        const gm_t prog_dw_nz = gballot_n<LPE>((st.progress & mdw) != 0, lane);
        const gm_t moved_dw_nz = gballot_n<LPE>((st.moved & mdw) != 0, lane);
        const gm_t fail_dw_nz = gballot_n<LPE>((st.failed & mdw) != 0, lane);
        const gm_t prog_lw_nz = gballot_n<LPE>((st.progress & mlw) != 0, lane);
        const gm_t moved_lw_nz = gballot_n<LPE>((st.moved & mlw) != 0, lane);
        const bool dead_me = focal && dl_ok && (members & (prog_dw_nz | moved_dw_nz)) == 0 && (members & fail_dw_nz) != 0;
        const bool live_me = focal && ll_ok && (members & prog_lw_nz) == 0 && (members & moved_lw_nz) != 0 &&
                             sum_delta <= io.eps_floor;
        deadlock = gballot_n<LPE>(dead_me, lane) != 0;
        livelock = !deadlock && gballot_n<LPE>(live_me, lane) != 0;
        const int prev = sc[MAPF_CTR_LOCK_STATE_PREV];
        dl_event = deadlock && !(prev & 1);
        ll_event = livelock && !(prev & 2);
        sc[MAPF_CTR_LOCK_STATE_PREV] = deadlock | (livelock << 1);
        sc[MAPF_CTR_DEADLOCK_STEPS] += deadlock;
        sc[MAPF_CTR_LIVELOCK_STEPS] += livelock;
        sc[MAPF_CTR_DEADLOCK_EVENTS] += dl_event;
        sc[MAPF_CTR_LIVELOCK_EVENTS] += ll_event;
    }
    const int blocking_step = __popcll(gballot<LPE>(blocking, lane));
    sc[MAPF_CTR_BLOCKING_COUNT] += blocking_step;
    MAPF_STAMP_W2(29);

    // info (MA-env:627-656) and counters, one coalesced store each (as step_body's FAST branch)
    const int reached_cnt = __popcll(gballot<LPE>(as.reached, lane));
    const int completed_cnt = __popcll(gballot<LPE>(as.completed, lane));
    {
        const int goals_total = reached_cnt;
        const int steps = max(sc[MAPF_CTR_STEP_COUNT], 1);
        float2 *xi = reinterpret_cast<float2 *>(aux_lds);
        uint4 *xs = reinterpret_cast<uint4 *>(aux_lds + 1024);  // (info: 16 groups x 56 bytes at most)
        if (a == 0) {
            float2 *q = xi + grp * 7;
            q[0] = make_float2((float)goals_step, (float)goals_total);
            q[1] = make_float2((float)blocking_step, (float)sc[MAPF_CTR_BLOCKING_COUNT]);
            q[2] = make_float2((float)deadlock, (float)livelock);
            q[3] = make_float2((float)dl_event, (float)ll_event);
            q[4] = make_float2((float)sc[MAPF_CTR_DEADLOCK_EVENTS], (float)sc[MAPF_CTR_LIVELOCK_EVENTS]);
            q[5] = make_float2((float)sc[MAPF_CTR_DEADLOCK_STEPS], (float)sc[MAPF_CTR_LIVELOCK_STEPS]);
            q[6] = make_float2((float)completed_cnt / (float)N, (float)goals_total / (float)steps);
            if (do_reset) {  // a re-placed env stores the counters reset() leaves (MA-env:440-455)
                xs[grp * 3] = xs[grp * 3 + 1] = make_uint4(0, 0, 0, 0);
                xs[grp * 3 + 2] = make_uint4(0, sc[MAPF_CTR_EPISODES_DONE] + 1, 1, sc[11]);
            } else {
                xs[grp * 3] = make_uint4(sc[0], sc[1], sc[2], sc[3]);
                xs[grp * 3 + 1] = make_uint4(sc[4], sc[5], sc[6], sc[7]);
                xs[grp * 3 + 2] = make_uint4(sc[8], sc[9], sc[10], sc[11]);
            }
        }
        wave_lds_sync();
        if (io.info_all) {
            float2 *dst = reinterpret_cast<float2 *>(io.info_all + (size_t)env0 * MAPF_INFO_ALL);
#pragma unroll
            for (int k0 = 0; k0 < G * 7; k0 += 64) {  // (56 float2 for groups of 8 lanes, 112 for groups of 4)
                if (k0 + lane < G * 7) {
                    const float2 v = xi[k0 + lane];
                    store_wt8(dst + k0, lane, make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)));
                }
            }
        }
        if (lane < 3 * G) {
            const int g = lane / 3, j = lane - 3 * g;
            store_state16(io.scal + (size_t)(env0 + g) * kScalInts + j * 4, xs[lane]);
        }
    }
    {   // the history planes (a re-placed env: _reset_lock_tracking MA-env:360-372, whichever wave places it)
        Lane img = st;
        if (do_reset) {
            img.moved = img.failed = img.progress = 0ull;
            img.dist = make_uint4(0, 0, 0, 0);
        }
        store_lane_hist(io.agents, io.bn8, idx0, lane, img);
    }
    MAPF_STAMP_W2(30);
    // episode statistics (callbacks.py:236-345), as step_body
    if (__builtin_expect(__any(done), 0)) {
        if (done && a == 0) {
            int *acc = p.ep_acc + (size_t)env * MAPF_NUM_EPISODE_ACC;
            atomicAdd(acc + MAPF_ACC_EPISODES, 1);
            if (dec.term && !dec.trunc) atomicAdd(acc + MAPF_ACC_SUCCESSES, 1);
            atomicAdd(acc + MAPF_ACC_GOALS_REACHED, sc[MAPF_CTR_GOALS_REACHED_TOTAL]);
            atomicAdd(acc + MAPF_ACC_BLOCKING_COUNT, sc[MAPF_CTR_BLOCKING_COUNT]);
            atomicAdd(acc + MAPF_ACC_DEADLOCK_COUNT, sc[MAPF_CTR_DEADLOCK_EVENTS]);
            atomicAdd(acc + MAPF_ACC_LIVELOCK_COUNT, sc[MAPF_CTR_LIVELOCK_EVENTS]);
            atomicAdd(acc + MAPF_ACC_DEADLOCK_STEPS, sc[MAPF_CTR_DEADLOCK_STEPS]);
            atomicAdd(acc + MAPF_ACC_LIVELOCK_STEPS, sc[MAPF_CTR_LIVELOCK_STEPS]);
            atomicAdd(acc + MAPF_ACC_COMPLETED_AGENTS, completed_cnt);
            atomicAdd(acc + MAPF_ACC_EPISODE_STEPS, sc[MAPF_CTR_STEP_COUNT]);
        }
    }
    if (__builtin_expect(__any(dec.slow_reset), 0)) wg_sync();  // B2 (the state wave's slow reset; all waves of the workgroup meet)
}

// ---- wave 0 of k_step3 (FAST workgroups): the move phase, nothing else on its path -----------------------------------
template <class K, int LPE, int MW>
__device__ __forceinline__ void state3_wave(const Params &p, const Io &io, const Lds &l, const int lane, const int env0,
                                            const int act, const uint2 hot, const int step_count_in, uint32_t nsg,
                                            const uint64_t *brows, DrawReq &dreq, int &d_stage) {
    constexpr int G = 64 / LPE;
    const int grp = lane / LPE, a = lane % LPE;
    const int env = env0 + grp;
    const int N = K::N(p), H = io.H, W = io.W;
    const uint32_t old = hot.x & 0xFFFFu, goal = hot.x >> 16, pass = hot.y >> 24;

    // ---- move phase (MA-env:502-526): "inside the grid and no obstacle" is the agent's pass bit for the action ----
    const int dr = (act == 1) ? -1 : ((act == 3) ? 1 : 0);
    const int dc = (act == 2) ? 1 : ((act == 4) ? -1 : 0);
    const bool want = act != 0 && ((pass >> ((act - 1) & 3)) & 1u) != 0;
    const uint32_t tgt = want ? (uint32_t)(((int)old + (dr << 8) + dc) & 0xFFFF) : kNoCell;
    MAPF_STAMP(16);
    uint32_t cur = old;
    if (__any(want)) {
        if constexpr (LPE <= 8) cur = resolve_moves_dpp<LPE>(a, old, tgt);
        else cur = resolve_moves<K, LPE>(p, reinterpret_cast<uint2 *>(l.tab + grp * LPE), lane, a, old, tgt);  // (16 lanes: the LDS
            // table and two ballots per round; replaying sixteen decisions on DPP-exchanged words is the longer chain)
    }
    MAPF_STAMP(2);
    // ---- publish the moves: x old | new << 16, y goal | new << 16 (the aux wave reads the new cell there), z the
    //      placement slot word; how the step ends every wave decides for itself (decide_end) ----
    l.otab[lane] = make_uint4(old | (cur << 16), goal | (cur << 16), nsg, 0u);
    wg_sync();  // B1
    MAPF_STAMP(19);
    // ---- an env of the wave ends its episode without a pre-drawn placement (rare): draw inline (reset_groups, B2 inside)
    const EndDecision dec = decide_end<LPE>(io, N, lane, cur, goal, step_count_in + 1, nsg);
    // which slice of the background draw this wave runs in this launch (below, in the kernel); what it reads from memory is
    // requested HERE, so that the round trip passes under the per-agent outputs
    dreq.w0 = group_bcast<0, LPE>(nsg);
    if constexpr (k3_rows<LPE, MW>()) {
        d_stage = draw_request_body<K, LPE, false>(p, io, N, a, env, true, dreq);
        // (the observation wave stores the per-agent outputs when it runs its prepared path: obs3_wave_prepared)
        const bool w1_outputs = (io.use_map & 6) == 2 && (io.obs || io.final_obs);
        if (brows && !w1_outputs)
            state3_outputs<K, LPE>(p, io, l, brows + (2 * G + grp) * (H + 2 * kRowPad) + kRowPad, lane, env0, act, hot, cur, dec, nsg);
    }
    if (__builtin_expect(__any(dec.slow_reset), 0)) {
        const uint64_t *myrows = l.rows + grp * (H + 2 * kRowPad) + kRowPad;  // (the observation wave's; valid after B1)
        Lane st;
        st.pos = cur;
        st.goal = goal;
        st.start = hot.y & 0xFFFFu;
        st.flags = 0u;
        st.moved = st.failed = st.progress = 0ull;
        st.dist = make_uint4(0, 0, 0, 0);
        int sc_unused[12];  // (the aux wave owns the counters and the history planes; it stores their reset image)
#pragma unroll
        for (int k = 0; k < 12; k++) sc_unused[k] = 0;
        wave_lds_sync();
        reset_groups<K, LPE, MW>(p, io, l.rows, l.tab, l.stage, l.scratch, lane, a, grp, env, true, true, dec.slow_reset, st,
                                 sc_unused, io.obs != nullptr, nsg, true);  // B2: the aux wave is always there
        if (io.obs) flush_obs<K, LPE>(p, io, l.stage, lane, env0, G, dec.slow_reset ? 0 : 2);
        if (dec.slow_reset)
            store_lane_hot(io.agents, (size_t)env0 * N + lane, st, agent_pass_bits(myrows, st.pos, io.col_pad, W));
    }
}

template <class K, int LPE, int MW, int WPS = 0>
__global__ __launch_bounds__(192, (WPS ? WPS : 1)) void k_step3(const Params *__restrict__ pp, MAPF_IO_HEAD_PARAMS,
                                                                const IoTail tail) {
    static_assert(K::kSlicedDraw, "k_step3 is for finite shapes with the sliced draw: full groups of 4, 8 or 16 lanes");
    MAPF_STAMP_ENTRY();
    const Params &p = *pp;
    const Io io = MAPF_IO_JOIN;
    constexpr int G = 64 / LPE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, grp = lane / LPE, a = lane % LPE;
    const int env0 = (int)blockIdx.x * G;
    const int N = K::N(p);  // == LPE
    const int ngroups = min(G, io.B - env0);
    const bool full = ngroups == G;  // wave-uniform
    const bool env_ok = grp < ngroups;
    const int env = env_ok ? env0 + grp : io.B - 1;
    const size_t idx = (size_t)env * N + a;
    __builtin_amdgcn_s_setprio(2);  // (until B1; afterwards: observation wave 3, aux wave 1, background slice 0)

    // every wave reads the actions: FAST (full workgroup, no invalid action) is decided by each of them from the same bytes
    if (wv == 1) {
        // ---- observation wave ----
        RowRegs rr;
        rows_issue<LPE>(io.grid_rows, io.H, lane, env0, ngroups, rr);
        const uint2 hot1 = io.agents[idx];
        int act1 = (int)io.actions[idx];
        const uint32_t nsg1 = slots_of(io.scal, io.B)[idx];
        const int step1 = io.scal[(size_t)env * kScalInts + MAPF_CTR_STEP_COUNT];
        __builtin_amdgcn_sched_barrier(0);
        warm_scalar_cache(pp, tail);
        const Lds l = carve_lds(io, lds_raw);
        float *gd_lut = reinterpret_cast<float *>(lds_raw + io.lds_map_off + 2048);
        {   // under the latency of the loads above: the goal-delta quotients (MA-env:330-335) of every delta a <= 64 x 64 grid
            // has, with the correctly rounded divide (behind the loads this was 0.4 k cycles in front of B1, and the
            // observation wave is the last to arrive there)
            const bool norm = (K::flags(p) & MAPF_FLAG_NORMALIZE_GOAL_DELTA) != 0;
#pragma unroll
            for (int k = lane; k < 128; k += 64) {
                gd_lut[k] = goal_delta(k - 63, io.den_r, norm);
                gd_lut[128 + k] = goal_delta(k - 63, io.den_c, norm);
            }
        }
        uint64_t *brows = reinterpret_cast<uint64_t *>(lds_raw + io.lds_map_off + obs_rows_lds_off());
        const bool prepared = k3_rows<LPE, MW>() && (io.use_map & 6) == 2 && (io.obs || io.final_obs);  // (host: grids with sentinel columns)
        __builtin_amdgcn_sched_barrier(0);
        rows_commit<LPE>(io.grid_rows, io.H, l.rows, lane, env0, ngroups, rr);
        wave_lds_sync();
        MAPF_STAMP_W1(10);
        act1 = env_ok ? act1 : 0;
        const bool fast3 = full && io.env_mask == nullptr && !__any(act1 < 0 || act1 > 4);
        if (__builtin_expect(!fast3, 0)) {  // the two-wave code
            wg_sync();  // B0
            if (io.obs || io.final_obs) obs_wave_step<K, LPE, MW>(p, io, l, lane, env0, ngroups);
            return;
        }
        if constexpr (k3_rows<LPE, MW>()) {
            if (prepared) {
                obs3_wave_prepared<K, LPE, MW>(p, io, l, brows, lane, env0, hot1, act1, nsg1, step1, gd_lut);
                return;
            }
        }
        if (io.obs || io.final_obs)
            obs3_wave<K, LPE, MW>(p, io, l, lane, env0, hot1, nsg1, step1, gd_lut);
        else wg_sync();  // B1 (the other waves read the rows behind it)
        return;
    }

    // ---- waves 0 and 2 ----
    LaneRaw raw;
    raw.h = io.agents[idx];
    int act = (int)io.actions[idx];
    int sc[12];
    uint32_t nsg = slots_of(io.scal, io.B)[idx];
    DrawReq dreq;
    if (wv == 0) {
        sc[0] = io.scal[(size_t)env * kScalInts + MAPF_CTR_STEP_COUNT];
        // for the background draw (below): the env's hint (same 64 bytes as the step counter); its slot word 0 is lane
        // 0's nsg.  What a slice reads beyond that is fetched after B1 (this wave has the time; ahead of B1 every
        // byte delays the loads the move phase waits for)
        dreq.hint = io.scal[(size_t)env * kScalInts + MAPF_CTR_MAY_FINISH];
        dreq.pop = 2 * N + 1;
        dreq.sv[0] = dreq.sv[1] = 0;
        dreq.rq = make_uint4(0, 0, 0, 0);
        dreq.ja[0] = dreq.ja[1] = dreq.cq[0] = dreq.cq[1] = make_uint4(0, 0, 0, 0);
    } else {
        lane_issue_hist(io.agents, io.bn8, idx, raw);
        load_scal(io.scal, env, sc);
    }
    __builtin_amdgcn_sched_barrier(0);
    warm_scalar_cache(pp, tail);
    __builtin_amdgcn_sched_barrier(0);
    const Lds l = carve_lds(io, lds_raw);
    unsigned char *aux_lds = lds_raw + io.lds_map_off;  // (k_step3: 2 KiB for the aux wave's info / counter staging)
    // prepared observation (obs3_wave_prepared): the aux wave owns the bit rows; they start a step empty (cleared under the
    // latency of the loads above)
    uint64_t *brows = nullptr;
    if constexpr (k3_rows<LPE, MW>()) {
        if ((io.use_map & 2) != 0) {
            brows = reinterpret_cast<uint64_t *>(aux_lds + obs_rows_lds_off());
            if (wv == 2) {
                uint4 *z = reinterpret_cast<uint4 *>(brows);
                const int n4 = obs_rows_lds_bytes(G, io.H) >> 4;
                for (int k = lane; k < n4; k += 64) z[k] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    bool is_agent = env_ok;  // (N == LPE)
    act = (full || is_agent) ? act : 0;
#ifdef MAPF_STAMPS
    if (wv == 0) {
        asm volatile("" ::"v"(raw.h.x), "v"(act));
        MAPF_STAMP(0);
    }
#endif
    const bool fast3 = full && io.env_mask == nullptr && !__any(act < 0 || act > 4);
    if (__builtin_expect(!fast3, 0)) {
        // ---- a ragged last workgroup, an invalid action or a masked step: the two-wave code of k_step ----
        if (wv == 2) return;  // before any barrier
        is_agent = is_agent && env_live(io, env);
        act = is_agent ? act : 0;
        lane_issue_hist(io.agents, io.bn8, idx, raw);
        load_scal(io.scal, env, sc);
        Lane st;
        lane_unpack(raw, is_agent, st);
        wg_sync();  // B0: the observation wave's rows
        step_body<K, LPE, MW, false, true>(p, io, l, lane, env0, ngroups, act, st, sc, nsg, false, false);
        if (is_agent)
            store_lane(io.agents, io.bn8, idx, st, l.rows + grp * (io.H + 2 * kRowPad) + kRowPad, io.col_pad, io.W);
        if (is_agent && a == 0) store_scal(io.scal, env, sc);
        return;
    }
    if (wv == 0) {
        MAPF_STAMP(1);
#ifdef MAPF_STAMPS
        {   // (stamps build: which XCD / CU / SIMD the state wave runs on, slot 3 of the workgroup's row)
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            if (p.dbg && lane == 0) p.dbg[(size_t)(env0 / G) * kDbgRow + 3] = (unsigned long long)hwid | ((unsigned long long)(xcc & 0xFu) << 32) | (1ull << 40);
        }
#endif
        int d_stage = 0;
        state3_wave<K, LPE, MW>(p, io, l, lane, env0, act, raw.h, sc[0], nsg, brows, dreq, d_stage);
        // The background draw of the next placement (draw_slice), one slice per launch, runs HERE: this wave has nothing
        // else to do once the moves are out.  (The window before B1, where round 2 ran it in the observation wave, closed
        // when B1 moved from 5.5 k to under 3 k cycles: run there by the aux wave -- inputs fetched with its first loads,
        // outputs at one 128-bit product each -- a slice still takes 1.7-2.6 k cycles of LDS staging and round trips and
        // holds B1 up for its workgroup: 6.25 us per staggered step against 5.6 here.)
        {
            if constexpr (!k3_rows<LPE, MW>()) d_stage = draw_request_body<K, LPE, false>(p, io, N, a, env, true, dreq);
            if (__builtin_expect(__any(d_stage != 0), 0)) {
                __builtin_amdgcn_s_setprio(0);  // (background work: behind everything that a step waits for; 1 or 2: -0.3 %)
                MAPF_STAMP(4);
                draw_slice<K, LPE>(p, io, l.scratch, lane, env, d_stage, dreq);
#ifdef MAPF_STAMPS
                int run = 0;
                for (int k = 1; k <= kDrawSlices; k++) run = __any(d_stage == k) ? k : run;
                if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)(env0 / (64 / LPE)) * kDbgRow + 23] = run;
#endif
            }
        }
        MAPF_STAMP(8);
#ifdef MAPF_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        MAPF_STAMP(9);
        MAPF_STAMP_ENTRY_STORE();
        return;
    }
    // ---- aux wave ----
    aux3_wave<K, LPE, MW>(p, io, l, aux_lds, lane, env0, act, raw, sc, nsg, brows);
#ifdef MAPF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    MAPF_STAMP_W2(31);
}

// ------------------------------------------------------------------------------------------------
// T steps in one launch (mapf_step_many): state stays in registers, obstacle rows stay in LDS, only the
// per-step action bytes are read and the per-step outputs written.  actions [T][B][N]; every non-null
// output is [T][...] except obs: obs_mode 0 = none, 1 = observation after the last step only, 2 = every step.
// Finished envs are reset inside the loop (auto_reset semantics of mapf_step).
// Two-wave workgroups: on every step that produces observations the waves meet at B1 (observation table
// published) and at B2 if an env resets.  There is no end-of-step barrier on the pair-table path: the observation
// table is double buffered, the staging rows belong to the observation wave (the state wave touches them only
// between B2 and the next B1), so the state wave runs up to one step ahead and the two waves overlap across step
// boundaries.  Wide groups (the single-buffered LDS cell map) run this kernel with one wave: dual_many_for().
// ------------------------------------------------------------------------------------------------
// Device-side action source of the fused kernel (mapf_step_many_sampled): the reference benchmark's masked-random
// policy (scripts/benchmark_multi_agent_env.py:42-57: uniform over the actions the agent's action mask allows)
// evaluated in-kernel on the observation the previous step produced, with a counter-based generator (a hash of
// seed, env, agent and step: reproducible, no state), so that T steps whose actions depend on the observations run in
// ONE launch.  The actions taken are written to actions_out.
struct ManyPolicy {
    const float *obs_in;   // [B][N][L] current observation (mask of the first step); nullptr = actions come from Io::actions
    int8_t *actions_out;   // [T][B][N]
    uint64_t seed;
    int mask_off;          // offset of the 5 mask floats in an observation row
};
__device__ __forceinline__ int masked_random_action(uint64_t seed, uint32_t agent_id, uint32_t t, float up, float rt, float dn,
                                                    float lf) {
    uint64_t x = seed ^ (((uint64_t)agent_id << 32) | t);  // splitmix64 finalizer
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    const uint32_t bits = 1u | (up > 0.5f ? 2u : 0u) | (rt > 0.5f ? 4u : 0u) | (dn > 0.5f ? 8u : 0u) | (lf > 0.5f ? 16u : 0u);
    const uint32_t r = (uint32_t)(((x >> 32) * (uint64_t)__popc(bits)) >> 32);  // index among the valid actions
    int act = 0, seen = 0;
#pragma unroll
    for (int d = 0; d < 5; d++) {
        const int v = (bits >> d) & 1u;
        act = (v && seen == (int)r) ? d : act;
        seen += v;
    }
    return act;
}

// WPS: as k_step's -- the register budget in waves per SIMD (0: none).  The fused kernel's loop keeps state, counters and
// both bodies live and takes ~220 VGPRs unconstrained (two waves per SIMD): right for grids of at most two waves per SIMD
// (c3: 8 192 envs), wrong beyond, where half of the workgroups waited for a second round (16 384 envs: 9.6 us per step
// fused against 7.7 with single launches).  mapf_create picks the 128-register build for such grids (mapf_engine::dense).
template <class K, int LPE, int MW, int WPS = 0>
__global__ __launch_bounds__(many_threads(LPE), (WPS ? WPS : 1)) void k_step_many(const Params *__restrict__ pp, MAPF_IO_HEAD_PARAMS,
                                                                 const IoTail tail, const int T, const int obs_mode,
                                                                 const ManyPolicy pol) {
    const Params &p = *pp;
    warm_scalar_cache(pp, tail);
    const Io io = MAPF_IO_JOIN;
    constexpr int G = 64 / LPE;
    constexpr bool kDual = dual_many_for(LPE);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const Lds l = carve_lds(io, lds_raw);
    const int wv = kDual ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int lane = threadIdx.x & 63, grp = lane / LPE, a = lane % LPE;
    const int env0 = blockIdx.x * G;
    const int ngroups = min(G, io.B - env0);
    const int N = K::N(p), L = K::L(p);
    const size_t BN = (size_t)io.B * N;
    // the observation tensor of step t (nullptr: this step produces none)
    auto obs_of = [&](int t) -> float * {
        return obs_mode == 2 ? io.obs + (size_t)t * BN * L : ((obs_mode == 1 && t == T - 1) ? io.obs : nullptr);
    };

    if (kDual && wv == 1) {
        load_rows_to_lds<LPE>(io.grid_rows, io.H, l.rows, lane, env0, ngroups);
        wg_sync();  // B0
        for (int t = 0; t < T; t++) {
            Io it = io;
            it.final_obs = nullptr;
            it.obs = obs_of(t);
            if (it.obs == nullptr) continue;
            obs_wave_step<K, LPE, MW>(p, it, with_parity(l, t), lane, env0, ngroups);
            // B3 (sampled actions only): the state wave picks the next step's actions from the masks staged above; it
            // comes after B2 in both waves
            if (pol.obs_in && t + 1 < T) wg_sync();
        }
        return;
    }

    const bool full = ngroups == G && N == LPE;
    const bool env_ok = grp < ngroups;
    const int env = env_ok ? env0 + grp : io.B - 1;
    const bool is_agent = env_ok && a < N;

    Lane st;
    load_lane(io.agents, io.bn8, (size_t)env * N + min(a, N - 1), full || is_agent, st);
    int sc[12];
    load_scal(io.scal, env, sc);
    // a pre-drawn placement serves the env's first reset of this launch; later ones draw inline (no sampler here)
    uint32_t nsg = slots_of(io.scal, io.B)[(size_t)env * N + min(a, N - 1)];
    if (kDual) {
        wg_sync();  // B0
    } else {
        load_rows_to_lds<LPE>(io.grid_rows, io.H, l.rows, lane, env0, ngroups);
        wave_lds_sync();
    }

    for (int t = 0; t < T; t++) {
        if (LPE >= 32 && (K::kMapAlways || io.use_map)) {
            clear_cell_maps<LPE>(io, l.map, lane);
            wave_lds_sync();
        }
        Io it = io;
        it.auto_reset = 1;
        it.final_obs = nullptr;
        it.obs = obs_of(t);
        if (io.rewards) it.rewards = io.rewards + (size_t)t * BN;
        if (io.terminated) it.terminated = io.terminated + (size_t)t * io.B;
        if (io.truncated) it.truncated = io.truncated + (size_t)t * io.B;
        if (io.info_all) it.info_all = io.info_all + (size_t)t * io.B * MAPF_INFO_ALL;
        if (io.info_agent) it.info_agent = io.info_agent + (size_t)t * BN * 2;
        int act;
        if (pol.obs_in) {
            // the agent's action mask: from the caller's observation before the first step, afterwards from the
            // observation row the previous step staged in LDS (reset observation included)
            const float *m;
            if (t == 0) {
                m = pol.obs_in + ((size_t)env * N + min(a, N - 1)) * L + pol.mask_off;
            } else {
                if (kDual) wg_sync();  // B3
                m = l.stage + (size_t)(grp * N + min(a, N - 1)) * L + pol.mask_off;
            }
            act = masked_random_action(pol.seed, (uint32_t)(env * N + a), (uint32_t)t, m[1], m[2], m[3], m[4]);
            if (full || is_agent) pol.actions_out[(size_t)t * BN + (size_t)env * N + a] = (int8_t)act;
        } else {
            act = (int)io.actions[(size_t)t * BN + (size_t)env * N + min(a, N - 1)];
        }
        act = (full || is_agent) ? act : 0;
        if (full && !__any(act < 0 || act > 4))
            step_body<K, LPE, MW, true, kDual>(p, it, with_parity(l, t), lane, env0, ngroups, act, st, sc, nsg);
        else
            step_body<K, LPE, MW, false, kDual>(p, it, with_parity(l, t), lane, env0, ngroups, act, st, sc, nsg);
        wave_lds_sync();  // this wave's staging / table regions are reused by the next step
    }
    if (full || is_agent)
        store_lane(io.agents, io.bn8, (size_t)env * N + a, st, l.rows + grp * (io.H + 2 * kRowPad) + kRowPad, io.col_pad, io.W);
    if (env_ok && a == 0) store_scal(io.scal, env, sc);
    // ---- tail: the NEXT placement of every env of the wave whose slot is empty (consumed by a reset of this launch, or
    //      never drawn) is drawn here, once per launch, all groups of the wave side by side: the single-step kernels
    //      spread that draw over the launches of an episode (slices, sampler workgroups); a fused launch has no
    //      background, and without this every episode end of the next launch drew inline (staggered c3: 4.13 us per
    //      step against 3.60 in phase).  Finite episodes only: there the env's stream is consumed by reset() alone, so
    //      drawing ahead changes nothing the reference would see (Params::vis_rng keeps the visible stream, as for any
    //      pre-drawn slot).  A slot that a single-step launch left half drawn (staged) is not touched.
    if (!(K::flags(p) & (MAPF_FLAG_DETERMINISTIC | MAPF_FLAG_LIFELONG))) {
        const bool need = env_ok && gballot<LPE>(is_agent && a == 0 && nsg == kSlotInvalid, lane) != 0;
        if (__any(need)) {
            int16_t *hs = l.scratch + grp * p.scratch_i16;
            int pop = 0;
            const bool ok = draw_stage_a<LPE>(p, hs, lane, a, env, env_ok, need, N, p.vis_rng, PcgPre{false, {}, 0, {}, {}}, p.rng, pop);
            draw_stage_b<LPE>(p, env, hs, lane, a, ok, N, pop);
            if (need && ok && is_agent) {
                const int16_t *out = hs + sample_out_off_i16(N);
                const uint16_t *fc = p.free_cells + (size_t)env * p.HW;
                const int top = p.HW - 1;  // idx entries are ranks < F <= HW; the clamp only bounds the address
                MAPF_CHK(p, (unsigned)out[a] < (unsigned)p.n_free[env] && (unsigned)out[N + a] < (unsigned)p.n_free[env], 6, env, out[a]);
                const uint32_t cs = fc[min(max((int)out[a], 0), top)], cg = fc[min(max((int)out[N + a], 0), top)];
                slots_of(io.scal, io.B)[(size_t)env * N + a] = cs | (cg << 16);
            }
        }
    }
}

// ================================================================================================
// Wide three-wave step kernel (round 4): ONE env per workgroup, 33 .. 64 agents (64 lanes per env, lane a = agent a).
//
// What round 3's profile of the c5 shape (1 024 envs x 64x64 x 64 agents, lifelong; k_step with the LDS cell map) said:
// 596 vector instructions per wave but 15.9 k cycles of wave lifetime, two thirds of them in wait states -- nothing is
// short of issue slots, each wave is one long chain of dependent LDS round trips: 21.9 KB of cell map cleared per env
// and launch (four workgroups of a CU share one LDS pipe), B1 only after move + goal logic + map fill (5.2 k cycles after
// the state is in registers), then 25 map reads and their decode per observation window (4.0 k cycles to the staged
// row) beside 12 map reads + lock detector + info + stores in the state wave (6.1 k).
//
// Here an env's step is dealt to three waves that all start from their own loads (as k_step3 does for small groups), and
// the word-per-cell map is replaced by BIT ROWS (one 64-bit word per grid row: free cells, occupancy before / after
// the move, cells a successful mover left or entered, goals, intents) plus two byte maps that name the agent on a cell
// and are only read where a bit row says there is one -- so nothing but ~5 KB of bit rows is cleared per launch, a
// V x V window is V reads per bit row and a funnel shift each, and index lookups are left for the few cells whose
// occupancy depends on the observer's turn (MA-env:528: agents <= i at their new cell, agents > i at their old one) and
// for the few neighbours of the lock detector:
//   wave 0 (state)  hot plane + actions -> move phase (pass bits: no obstacle rows needed; occupant of the target and
//                   contention through the bit rows) -> lifelong respawn (MA-env:284-304) -> publish, B1 -> rewards,
//                   per-agent info, done flags, hot plane; the inline draw of an env that ends its episode;
//   wave 1 (obs)    obstacle rows -> free-cell bit rows, goal-delta quotient table; after B1 the windows, the flat
//                   rows (MA-env:306-328) and the observation stream; reset observation of a re-placed env;
//   wave 2 (aux)    history planes + counters; after B1 lock flags, neighbour sets from the bit rows, lock detector
//                   (MA-env:374-438), info (:627-656), counters, history planes, episode statistics.
// Every wave decides for itself how the step ends (wide_decide: same inputs, same answer).  Idle lanes (N < 64), an
// invalid action (MA-env:502-506: agents before it are processed, nothing after the loop) and a masked-out env are
// handled in place: the kernel has no fallback path.
// ================================================================================================
typedef const __attribute__((address_space(1))) uint16_t global_u16;
__device__ __forceinline__ global_u16 *as_global(const uint16_t *q) { return (global_u16 *)q; }
constexpr int kWideBitmaps = 9;        // occO occN mov goalb intent wantb (cleared at entry), occR goalR (cleared by whoever
constexpr int kWideCleared = 6;        // re-places the env), freeb (written whole)
constexpr int kWideOwnBytes = 64 * 64;  // byte maps are indexed row * 64 + col
__host__ __device__ constexpr int wide_rows(int H) { return H + 2 * kRowPad; }
__host__ __device__ constexpr int wide_lds_bytes(int H, int NL, int scratch_i16) {
    return kWideBitmaps * wide_rows(H) * 8 + 16 /* alignment slack */ + 3 * kWideOwnBytes + 64 * 16 /* tab */ +
           64 /* ctl */ + 1024 /* gd_lut */ + 128 /* xinfo */ + ((NL * 4 + 15) & ~15) + ((scratch_i16 * 2 + 15) & ~15);
}
struct WideLds {
    uint64_t *occO, *occN, *mov, *goalb, *intent, *wantb, *occR, *goalR, *freeb;  // row r at index r + kRowPad
    uint8_t *ownO, *ownN;  // agent index on a cell before / after the move; valid where occO / occN has the bit
    uint4 *tab;            // per agent: x old | new << 16, y goal (after a respawn) | arrived << 16, z reset placement
    int8_t *dmap;          // per CELL: goal-distance delta (livelock window) of the agent standing there after the move (the aux
                           // wave's, for its neighbour sums: read together with ownN, valid where occN has the bit)
    uint32_t *ctl;         // wave-uniform words of the state wave: [0] 1 = some goal was respawned in this step
    float *gd_lut;
    unsigned char *xinfo;
    float *stage;
    int16_t *scratch;
};
__device__ __forceinline__ WideLds carve_wide(unsigned char *raw, int H, int NL) {
    WideLds l;
    const int rs = wide_rows(H);
    uint64_t *b = reinterpret_cast<uint64_t *>(raw);
    l.occO = b; l.occN = b + rs; l.mov = b + 2 * rs; l.goalb = b + 3 * rs; l.intent = b + 4 * rs; l.wantb = b + 5 * rs;
    l.occR = b + 6 * rs; l.goalR = b + 7 * rs; l.freeb = b + 8 * rs;
    unsigned char *q = raw + ((kWideBitmaps * rs * 8 + 15) & ~15);
    l.ownO = q; q += kWideOwnBytes;
    l.ownN = q; q += kWideOwnBytes;
    l.tab = reinterpret_cast<uint4 *>(q); q += 64 * 16;
    l.dmap = reinterpret_cast<int8_t *>(q); q += kWideOwnBytes;
    l.ctl = reinterpret_cast<uint32_t *>(q); q += 64;
    l.gd_lut = reinterpret_cast<float *>(q); q += 1024;
    l.xinfo = q; q += 128;
    l.stage = reinterpret_cast<float *>(q); q += (NL * 4 + 15) & ~15;
    l.scratch = reinterpret_cast<int16_t *>(q);
    return l;
}
__device__ __forceinline__ int wide_cell(uint32_t cell) { return (int)((cell >> 8) * 64u + (cell & 255u)); }
__device__ __forceinline__ uint64_t wide_bit(uint32_t cell) { return 1ull << (cell & 63u); }
__device__ __forceinline__ int wide_row(uint32_t cell) { return (int)(cell >> 8) + kRowPad; }
// pass bits (mapf_kernels.inl: agent_pass_bits) of a cell from the free-cell bit rows: bit 0 up, 1 right, 2 down, 3 left
__device__ __forceinline__ uint32_t wide_pass_bits(const uint64_t *freeb, uint32_t cell) {
    const int r = (int)(cell >> 8) + kRowPad, c = (int)(cell & 255u);
    const uint64_t up = freeb[r - 1], mid = freeb[r], dn = freeb[r + 1];
    const uint32_t f_up = (uint32_t)(up >> (c & 63)) & 1u, f_dn = (uint32_t)(dn >> (c & 63)) & 1u;
    const uint32_t f_rt = c + 1 < 64 ? ((uint32_t)(mid >> ((c + 1) & 63)) & 1u) : 0u;
    const uint32_t f_lf = c > 0 ? ((uint32_t)(mid >> ((c - 1) & 63)) & 1u) : 0u;
    return f_up | (f_rt << 1) | (f_dn << 2) | (f_lf << 3);
}
// V bits of a bit row starting at column c0 (which may be negative or run past column 63: zeros there)
__device__ __forceinline__ uint32_t wide_window(uint64_t row, int c0, int V) {
    const uint32_t right = (uint32_t)(row >> (c0 & 63));
    const uint32_t left = (uint32_t)row << ((-c0) & 31);
    return ((c0 >= 0) ? right : left) & ((1u << V) - 1u);
}

// How the step ends for the env (MA-env:668-690 + auto-reset), identically in all three waves: from the agents' new
// cells and goals (published before B1), the step counter and the placement slot.
struct WideEnd {
    int term, trunc;
    bool done, do_reset, fast_reset, slow_reset, subst;
    int sel;  // which tensor the step's observation goes to: 0 io.obs, 1 io.final_obs, 2 nowhere
};
template <class K>
__device__ __forceinline__ WideEnd wide_decide(const Params &p, const Io &io, int N, bool on_goal, bool is_agent, bool errored,
                                               int step_count, uint32_t nsg) {
    const uint32_t flags = K::flags(p);
    const bool lifelong = (flags & MAPF_FLAG_LIFELONG) != 0, deterministic = (flags & MAPF_FLAG_DETERMINISTIC) != 0;
    WideEnd d;
    d.term = d.trunc = 0;
    if (!lifelong && __popcll(__ballot(on_goal)) == N) {
        d.term = 1;
    } else if (step_count >= io.steps_per_episode) {
        d.term = 1;
        d.trunc = 1;
    }
    d.done = !errored && (d.term | d.trunc) != 0;
    d.do_reset = d.done && io.auto_reset;
    d.fast_reset = d.slow_reset = d.subst = false;
    d.sel = errored ? 2 : (io.obs ? 0 : 2);
    if (__builtin_expect(d.do_reset, 0)) {
        bool slot_ok = deterministic;
        if (!deterministic && !lifelong) slot_ok = __ballot(is_agent && !slot_word_valid(nsg)) == 0;
        d.fast_reset = slot_ok;
        d.slow_reset = !slot_ok;
        d.subst = d.fast_reset && io.final_obs == nullptr;
        d.sel = io.final_obs ? 1 : ((d.subst && io.obs) ? 0 : 2);
    }
    return d;
}

// ---- the observation of one agent from the bit rows -----------------------------------------------------------------------
// occ / goals: the bit rows to read occupancy and goals from (the step's, or the reset placement's); final_state: everybody
// at the cell `occ` shows (reset, or after a lifelong respawn MA-env:565-575), else the staggered view of MA-env:528.
template <class K, int MW>
__device__ __forceinline__ void wide_observe(const Params &p, const Io &io, const WideLds &l, const uint64_t *occ,
                                             const uint64_t *goals, float *srow, bool is_agent, int a, uint32_t cur,
                                             uint32_t goal, bool final_state, bool pressure, const int env0 = 0) {
    constexpr int MAXV = MW == 32 ? 5 : (MW == 64 ? 7 : 11);
    constexpr int LPE = 64;
    (void)env0; (void)LPE;
    const int V = K::V(p), sr = K::sr(p);
    const int myr = is_agent ? (int)(cur >> 8) : 0, myc = is_agent ? (int)(cur & 255u) : 0;
    const int r0 = myr - sr, c0 = myc - sr;
    uint64_t rf[MAXV], ro[MAXV], rg[MAXV], rm[MAXV], rb[MAXV];
#pragma unroll
    for (int d = 0; d < MAXV; d++) {  // all reads of the window issue back to back: one LDS round trip
        const int ri = r0 + d + kRowPad;
        rf[d] = (d < V) ? l.freeb[ri] : 0ull;
        ro[d] = (d < V) ? occ[ri] : 0ull;
        rg[d] = (d < V) ? goals[ri] : 0ull;
        rm[d] = (d < V && !final_state) ? l.mov[ri] : 0ull;
        rb[d] = (d < V && !final_state) ? l.occO[ri] : 0ull;
    }
#ifdef MAPF_STAMPS
    asm volatile("" ::"v"(rf[0]), "v"(ro[V - 1]), "v"(rg[V - 1]));
#endif
    MAPF_STAMP_W1(25);  // (sub-stamp: window rows in registers)
    WMask<MW> obst, agm, gls, pend, was;
    obst.clear(); agm.clear(); gls.clear(); pend.clear(); was.clear();
#pragma unroll
    for (int d = 0; d < MAXV; d++) {
        if (d < V) {
            const uint32_t vm = (1u << V) - 1u;
            obst.or_row(~wide_window(rf[d], c0, V) & vm, d * V);
            agm.or_row(wide_window(ro[d], c0, V), d * V);
            gls.or_row(wide_window(rg[d], c0, V), d * V);
            if (!final_state) {
                pend.or_row(wide_window(rm[d], c0, V), d * V);
                was.or_row(wide_window(rb[d], c0, V), d * V);
            }
        }
    }
    MAPF_STAMP_W1(26);  // (sub-stamp: window masks built)
    if (!final_state) {
        // A window cell in `mov` -- a successful mover left it and / or entered it -- is occupied for observer i iff the
        // agent that stood there has not had its turn (i < l) or the one that entered has (i >= e): MA-env:528 sits inside
        // the move loop.  One cell per lane and round; a lane's window holds a handful at most (all windows of a 64-agent
        // env: ~80 such cells on 4 096).
        const WMask<MW> isn = agm;         // occupied after the move
        agm = agm.andnot(pend);            // the cells nobody moved on: occupied whoever looks
        while (__any(pend.any())) {
            // up to four cells per lane and round: their eight index reads share one LDS round trip
            int b[4], lo[4], en[4];
            bool on[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                on[k] = pend.any();
                b[k] = on[k] ? pend.first() : 0;
                if (on[k]) pend.clear_bit(b[k]);
                const int d = b[k] / V, e = b[k] - d * V;
                const int ci = on[k] ? (r0 + d) * 64 + c0 + e : 0;  // (inside the grid: mov only has bits there)
                lo[k] = (int)l.ownO[ci];
                en[k] = (int)l.ownN[ci];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool occupied = (was.get(b[k]) && a < lo[k]) || (isn.get(b[k]) && a >= en[k]);
                agm.set_if(on[k] && occupied, b[k]);
            }
        }
    }
    MAPF_STAMP_W1(27);  // (sub-stamp: turn-dependent cells resolved)
    if (!is_agent) return;
    agm.clear_bit(sr * V + sr);  // my own cell: "occ not in (UNASSIGNED, self)" MA-env:735
    WMask<MW> own;
    own.clear();
    window_set<MW>(own, goal, r0, c0, V);
    const int gdr = (int)((goal >> 8) & 255u) - myr, gdc = (int)(goal & 255u) - myc;
    emit_obs_row<K, MW, MAXV>(p, srow, obst, agm, gls, own, l.gd_lut[gdr + 63], l.gd_lut[128 + gdc + 63], pressure);
}

// one contiguous write-through stream of the env's N * L floats
template <class K>
__device__ __forceinline__ void wide_flush(const Params &p, const WideLds &l, float *dst_tensor, int env, int lane) {
    const int NL = K::N(p) * K::L(p);
    float *dst = dst_tensor + (size_t)env * NL;
    if ((NL & 3) == 0) {
        const ObsSink sink = make_obs_sink(dst, (unsigned)NL * 4u);
        const float4 *s4 = reinterpret_cast<const float4 *>(l.stage);
        const int n4 = NL >> 2;
        if (K::kFixed) {
            const int full = n4 >> 6;
#pragma unroll
            for (int r = 0; r < full; r++) store_obs4(sink, (unsigned)(r * 64 + lane) * 16u, s4[r * 64 + lane]);
            if ((full << 6) + lane < n4) store_obs4(sink, (unsigned)((full << 6) + lane) * 16u, s4[(full << 6) + lane]);
        } else {
            for (int k = lane; k < n4; k += 64) store_obs4(sink, (unsigned)k * 16u, s4[k]);
        }
    } else {
        for (int k = lane; k < NL; k += 64) dst[k] = l.stage[k];
    }
}

template <class K, int MW>
__global__ __launch_bounds__(192, 3) void k_stepw(const Params *__restrict__ pp, MAPF_IO_HEAD_PARAMS, const IoTail tail) {
    constexpr int LPE = 64;
    MAPF_STAMP_ENTRY();
    const Params &p = *pp;
    const Io io = MAPF_IO_JOIN;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, a = lane;
    const int N = K::N(p), H = io.H, W = io.W;
    const uint32_t flags = K::flags(p);
    const bool lifelong = (flags & MAPF_FLAG_LIFELONG) != 0, deterministic = (flags & MAPF_FLAG_DETERMINISTIC) != 0;
    const bool lock_on = (flags & MAPF_FLAG_LOCK_METRICS) != 0;
    const int lead = K::kSamplerFront ? sampler_blocks_for(io.B, 3) : 0;
    const int env = (int)blockIdx.x - lead;
    const int env0 = env;  // (stamp macros: one workgroup per env, stamp rows are indexed by it)
    (void)env0;
    if (__builtin_expect(env < 0 || env >= io.B, 0)) {  // a sampler workgroup (finite episodes with sampled placements)
        const int si = K::kSamplerFront ? (int)blockIdx.x : (int)blockIdx.x - io.B;
        sampler_wave<K, LPE>(p, io, lds_raw + wv * sampler_lds_bytes_per_wave(1, p.scratch_i16), si * 3 + wv, lane, io.B + si);
        return;
    }
    const bool is_agent = a < N;
    const size_t idx0 = (size_t)env * N;
    const size_t idx = idx0 + min(a, N - 1);
    __builtin_amdgcn_s_setprio(2);

    // every wave reads the hot plane and the actions itself (512 + 64 bytes): no wave waits for another one's loads
    const uint2 hot = io.agents[idx];
    int act = (int)io.actions[idx];
    __builtin_amdgcn_sched_barrier(0);
#ifdef MAPF_STAMPS
    {   // (stamps build: which SIMD of which CU this wave runs on -- HW_REG_HW_ID -- in slots 3 / 4 / 5 of the workgroup's row)
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (p.dbg && lane == 0) p.dbg[(size_t)env * kDbgRow + 3 + wv] = (unsigned long long)hwid | ((unsigned long long)(xcc & 0xFu) << 32);
    }
#endif
    // mapf_step_masked: nothing of a masked-out env is touched -- all three waves leave before any barrier or store.  The mask
    // pointer is not among the preloaded arguments: each wave looks at it once ALL its loads are out (in front of them, the
    // wait for the kernel-argument tail delayed every later load of the wave by that round trip).
#define MAPF_WIDE_MASK_CHECK()                                   \
    do {                                                         \
        if (__builtin_expect(io.env_mask != nullptr, 0)) {       \
            if (io.env_mask[env] == 0) return;                   \
        }                                                        \
    } while (0)

    if (wv == 1) {
        // =================================== observation wave ===================================
        const uint64_t *src = io.grid_rows + (size_t)env * H;
        const uint64_t grow = src[min(a, H - 1)];
        const uint32_t nsg = (lifelong || deterministic) ? kSlotInvalid : slots_of(io.scal, io.B)[idx];
        const int step1 = io.scal[(size_t)env * kScalInts + MAPF_CTR_STEP_COUNT];
        __builtin_amdgcn_sched_barrier(0);
        warm_scalar_cache(pp, tail);
        MAPF_WIDE_MASK_CHECK();
        wg_sync();  // B0 (this wave has nothing to wait for here: the state wave must not find it missing)
        const WideLds l = carve_wide(lds_raw, H, N * K::L(p));
        {   // goal-delta quotients (MA-env:330-335) of every delta a <= 64 x 64 grid has, correctly rounded divide, computed
            // under the latency of the loads above
            const bool norm = (flags & MAPF_FLAG_NORMALIZE_GOAL_DELTA) != 0;
#pragma unroll
            for (int k = lane; k < 128; k += 64) {
                l.gd_lut[k] = goal_delta(k - 63, io.den_r, norm);
                l.gd_lut[128 + k] = goal_delta(k - 63, io.den_c, norm);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // free-cell bit rows: bit c of row r = cell (r, c) is inside the grid and no obstacle; pad rows are zero
            const uint64_t colmask = W >= 64 ? ~0ull : ((1ull << W) - 1ull);
            if (a < H) l.freeb[kRowPad + a] = ~(grow >> io.col_pad) & colmask;
            if (a < 2 * kRowPad) l.freeb[a < kRowPad ? a : H + a] = 0ull;
        }
        MAPF_STAMP_W1(10);
        const bool bad = is_agent && (act < 0 || act > 4);
        const uint64_t badm = __ballot(bad);
        const bool errored = badm != 0;
        const int n_live = errored ? (int)__builtin_ctzll(badm) : N;
        const uint32_t goal0 = hot.x >> 16;
        const bool pressure_prev = ((hot.y >> 16) & kFlagPressure) != 0;
        (void)goal0;
        // (everything this wave loaded is in registers before the barrier: behind it the other waves store the new state)
        asm volatile("" ::"v"(hot.x), "v"(hot.y), "v"(nsg), "s"(step1));
        wg_sync();  // B1
        __builtin_amdgcn_s_setprio(3);
        MAPF_STAMP_W1(11);
        const bool want_obs = io.obs != nullptr || io.final_obs != nullptr;
        const uint4 ent = l.tab[a];
        const uint32_t cur = ent.x >> 16, goal = ent.y & 0xFFFFu;
        const bool reassigned = l.ctl[0] != 0u;
        const bool on_goal = is_agent && a < n_live && cur == (hot.x >> 16);  // (against the goal BEFORE a respawn, like step_body)
        const WideEnd dec = wide_decide<K>(p, io, N, on_goal, is_agent, errored, step1 + 1, nsg);
        float *srow = l.stage + (size_t)min(a, N - 1) * K::L(p);
        // the placement a fast reset installs (deterministic: reset() keeps the goals, MA-env:452-455)
        const uint32_t rs = deterministic ? ((hot.y & 0xFFFFu) | (goal << 16)) : nsg;
        if (want_obs && dec.sel != 2 && !dec.subst) {
            wide_observe<K, MW>(p, io, l, l.occN, l.goalb, srow, is_agent, a, cur, goal, reassigned, pressure_prev, env0);
            wave_lds_sync();
            MAPF_STAMP_W1(12);
            wide_flush<K>(p, l, dec.sel == 0 ? io.obs : io.final_obs, env, lane);
            MAPF_STAMP_W1(13);
        }
        if (__builtin_expect(dec.do_reset, 0)) {
            if (dec.slow_reset) wg_sync();  // B2: the state wave has drawn the placement (tab[].z, occR / goalR)
            if (io.obs) {
                uint32_t place = rs;
                if (dec.fast_reset) {  // the placement is known here: its bit rows are built by this wave
                    for (int k = lane; k < 2 * wide_rows(H); k += 64) l.occR[k] = 0ull;  // (occR and goalR, adjacent)
                    if (is_agent) {
                        atomicOr(reinterpret_cast<unsigned long long *>(&l.occR[wide_row(rs & 0xFFFFu)]), wide_bit(rs & 0xFFFFu));
                        atomicOr(reinterpret_cast<unsigned long long *>(&l.goalR[wide_row(rs >> 16)]), wide_bit(rs >> 16));
                    }
                } else {
                    place = l.tab[a].z;
                }
                wave_lds_sync();  // (also: the staging row of the first pass has been read by its flush)
                wide_observe<K, MW>(p, io, l, l.occR, l.goalR, srow, is_agent, a, place & 0xFFFFu, place >> 16, true, false);
                wave_lds_sync();
                wide_flush<K>(p, l, io.obs, env, lane);
            }
        }
#ifdef MAPF_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MAPF_STAMP_W1(14);
        if (p.dbg && threadIdx.x == 64) p.dbg[(size_t)env * kDbgRow + 6] = _t_entry;
#endif
        return;
    }

    if (wv == 2) {
        // =================================== aux wave ===================================
        LaneRaw raw;
        raw.h = hot;
        lane_issue_hist(io.agents, io.bn8, idx, raw);
        int sc[12];
        load_scal(io.scal, env, sc);
        const uint32_t nsg = (lifelong || deterministic) ? kSlotInvalid : slots_of(io.scal, io.B)[idx];
        __builtin_amdgcn_sched_barrier(0);
        warm_scalar_cache(pp, tail);
        __builtin_amdgcn_sched_barrier(0);
        MAPF_WIDE_MASK_CHECK();
        const WideLds l = carve_wide(lds_raw, H, N * K::L(p));
        {   // the six bit rows that start a step empty (3.5 KB at H = 64), under the latency of this wave's loads: done by
            // the state wave it was 0.5 k cycles on the path every other phase waits for
            uint4 *z = reinterpret_cast<uint4 *>(lds_raw);
            const int n4 = (kWideCleared * wide_rows(H) * 8 + 15) >> 4;
            for (int k = lane; k < n4; k += 64) z[k] = make_uint4(0u, 0u, 0u, 0u);
        }
        wg_sync();  // B0
        const int lw = K::lw(p), dw = K::dw(p), ring_stride = K::ring_stride(p);
        const bool dist_in_rec = lw <= 16;
        const bool bad = is_agent && (act < 0 || act > 4);
        const uint64_t badm = __ballot(bad);
        const bool errored = badm != 0;
        const int n_live = errored ? (int)__builtin_ctzll(badm) : N;
        const bool live = is_agent && a < n_live;
        if (!live) act = 0;
        MAPF_STAMP_W2(21);
        wg_sync();  // B1
        __builtin_amdgcn_s_setprio(1);
        MAPF_STAMP_W2(22);
        Lane st;
        lane_unpack(raw, is_agent, st);
        const uint4 ent = l.tab[a];
        const uint32_t old = st.pos, cur = is_agent ? (ent.x >> 16) : (uint32_t)kIdleCell;
        const uint32_t goal_new = is_agent ? (ent.y & 0xFFFFu) : (uint32_t)kIdleGoal;
        sc[MAPF_CTR_STEP_COUNT] += 1;  // MA-env:475
        const bool on_goal = live && cur == st.goal;
        const WideEnd dec = wide_decide<K>(p, io, N, on_goal, is_agent, errored, sc[MAPF_CTR_STEP_COUNT], nsg);
        // goal / reward flags (MA-env:538-563)
        const bool moved = cur != old;
        bool reached = (st.flags & kFlagReached) != 0, completed = (st.flags & kFlagCompleted) != 0;
        bool grs = false;
        if (on_goal && (lifelong || !reached)) {
            grs = true;
            completed = true;
            reached = !lifelong;  // lifelong: reached_goal[i] = False after the respawn, _reached_arr never set
        }
        const int goals_step = __popcll(__ballot(grs));
        sc[MAPF_CTR_GOALS_REACHED_TOTAL] += goals_step;  // MA-env:550,563
        int sc_keep[12];
#pragma unroll
        for (int k = 0; k < 12; k++) sc_keep[k] = sc[k];
        const uint32_t goal_before = st.goal;
        st.goal = goal_new;
        // lock flags (MA-env:581-594) and distance history
        const bool cur_on_goal = is_agent && cur == st.goal;
        const bool prev_on_goal = !lifelong && old == goal_before;
        const bool progress = lifelong ? grs : (!prev_on_goal && cur_on_goal);
        const bool failed = act != 0 && !moved;
        const int dist = cell_l1(cur, st.goal);
        int delta = 0;
        bool dl_ok = false, ll_ok = false;
        if (lock_on) {
            const int t = sc[MAPF_CTR_HIST_ROWS];
            const int count = min(t + 1, K::hs(p));
            dl_ok = count >= dw;
            ll_ok = count >= lw;
            st.moved = (st.moved << 1) | (moved ? 1ull : 0ull);  // _append_lock_history_step MA-env:374-387
            st.failed = (st.failed << 1) | (failed ? 1ull : 0ull);
            st.progress = (st.progress << 1) | (progress ? 1ull : 0ull);
            int d_old = dist;
            if (dist_in_rec) {
                st.dist.w = (st.dist.w << 8) | (st.dist.z >> 24);
                st.dist.z = (st.dist.z << 8) | (st.dist.y >> 24);
                st.dist.y = (st.dist.y << 8) | (st.dist.x >> 24);
                st.dist.x = (st.dist.x << 8) | (uint32_t)dist;
                const int ob = lw - 1, od = ob >> 2;
                const uint32_t wd = od == 0 ? st.dist.x : (od == 1 ? st.dist.y : (od == 2 ? st.dist.z : st.dist.w));
                if (ll_ok) d_old = (int)((wd >> ((ob & 3) * 8)) & 0xFFu);
            } else {
                int16_t *ring = io.dist_ring + (idx0 + min(a, N - 1)) * ring_stride;
                const int slot_new = t % lw;
                const int slot_old = (slot_new + 1 == lw) ? 0 : slot_new + 1;
                if (ll_ok && is_agent) d_old = ring[slot_old];
                if (is_agent && !errored) ring[slot_new] = (int16_t)dist;
            }
            delta = d_old - dist;
            sc[MAPF_CTR_HIST_ROWS] = t + 1;
        }
        if (!lifelong)
            sc[MAPF_CTR_MAY_FINISH] = (__ballot(is_agent && dist > 1) == 0 || sc[MAPF_CTR_STEP_COUNT] + 1 >= io.steps_per_episode) ? 1 : 0;
        // neighbour sets (MA-env:389-398) from the occupancy rows after the move: the diamond of radius `nearby` around my
        // cell, then one index (and delta) lookup per neighbour -- there are few (64 agents on ~3 300 free cells)
        uint64_t nbr = 0;
        int sum_delta = delta;
        bool blocking = false;
        const int myr = is_agent ? (int)(cur >> 8) : 0, myc = is_agent ? (int)(cur & 255u) : 0;
        if (lock_on && is_agent) l.dmap[myr * 64 + myc] = (int8_t)delta;  // (|delta| <= 126 on a 64 x 64 grid)
        {
            // the diamond as ONE bit mask, row dr at bits (dr + nb) * (2 nb + 1) ..: a lane pops one neighbour per round, and
            // the rounds a wave runs are the most neighbours any of its agents has (2-3), not a sum over rows
            const int nb = K::nearby(p), side = 2 * nb + 1;
            uint64_t m_lo = 0, m_hi = 0;
#pragma unroll
            for (int dr = -kRowPad; dr <= kRowPad; dr++) {
                if (lock_on && abs(dr) <= nb) {
                    const int span = nb - abs(dr);
                    uint64_t bits = wide_window(l.occN[myr + dr + kRowPad], myc - span, 2 * span + 1);
                    if (dr == 0) bits &= ~(1ull << span);  // not myself
                    const int at = (dr + nb) * side + (nb - span);
                    if (at < 64) m_lo |= bits << at;
                    if (at + 2 * span + 1 > 64) m_hi |= at >= 64 ? (bits << (at - 64)) : (bits >> (64 - at));
                }
            }
            if (!is_agent) m_lo = m_hi = 0;
            const uint64_t irow = l.intent[myr + kRowPad];
            blocking = is_agent && reached && !moved && ((irow >> (myc & 63)) & 1ull) != 0;  // MA-env:608-623
            wave_lds_sync();  // (every lane's delta is in the map)
            while (__any((m_lo | m_hi) != 0)) {
                // two neighbours per lane and round (their four reads share one LDS round trip)
                int j[2], dj[2];
                bool on[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    on[k] = (m_lo | m_hi) != 0;
                    const int b = !on[k] ? 0 : (m_lo ? (int)__builtin_ctzll(m_lo) : 64 + (int)__builtin_ctzll(m_hi));
                    if (b < 64) m_lo &= m_lo - 1; else m_hi &= m_hi - 1;
                    const int dr = b / side - nb, dc = b - (dr + nb) * side - nb;
                    const int ci = on[k] ? (myr + dr) * 64 + myc + dc : 0;
                    j[k] = (int)l.ownN[ci];
                    dj[k] = (int)l.dmap[ci];
                }
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    nbr |= on[k] ? (1ull << (j[k] & 63)) : 0ull;
                    sum_delta += on[k] ? dj[k] : 0;
                }
            }
        }
        MAPF_STAMP_W2(24);
        int deadlock = 0, livelock = 0, dl_event = 0, ll_event = 0;
        if (lock_on) {  // MA-env:400-438: deadlock has priority over livelock
            const uint64_t mdw = dw >= 64 ? ~0ull : ((1ull << dw) - 1ull);
            const uint64_t mlw = lw >= 64 ? ~0ull : ((1ull << lw) - 1ull);
            const uint64_t members = nbr | (1ull << a);
            const bool focal = is_agent && !cur_on_goal && __popcll(nbr) >= K::min_nbrs(p);
            const uint64_t prog_dw_nz = __ballot(is_agent && (st.progress & mdw) != 0);
            const uint64_t moved_dw_nz = __ballot(is_agent && (st.moved & mdw) != 0);
            const uint64_t fail_dw_nz = __ballot(is_agent && (st.failed & mdw) != 0);
            const uint64_t prog_lw_nz = __ballot(is_agent && (st.progress & mlw) != 0);
            const uint64_t moved_lw_nz = __ballot(is_agent && (st.moved & mlw) != 0);
            const bool dead_me = focal && dl_ok && (members & (prog_dw_nz | moved_dw_nz)) == 0 && (members & fail_dw_nz) != 0;
            const bool live_me = focal && ll_ok && (members & prog_lw_nz) == 0 && (members & moved_lw_nz) != 0 &&
                                 sum_delta <= io.eps_floor;
            deadlock = __ballot(dead_me) != 0;
            livelock = !deadlock && __ballot(live_me) != 0;
            const int prev = sc[MAPF_CTR_LOCK_STATE_PREV];
            dl_event = deadlock && !(prev & 1);  // rising edges MA-env:599-600
            ll_event = livelock && !(prev & 2);
            sc[MAPF_CTR_LOCK_STATE_PREV] = deadlock | (livelock << 1);
            sc[MAPF_CTR_DEADLOCK_STEPS] += deadlock;
            sc[MAPF_CTR_LIVELOCK_STEPS] += livelock;
            sc[MAPF_CTR_DEADLOCK_EVENTS] += dl_event;
            sc[MAPF_CTR_LIVELOCK_EVENTS] += ll_event;
        }
        const int blocking_step = __popcll(__ballot(blocking));
        sc[MAPF_CTR_BLOCKING_COUNT] += blocking_step;
        MAPF_STAMP_W2(29);
        const int reached_cnt = __popcll(__ballot(is_agent && reached));
        const int completed_cnt = __popcll(__ballot(is_agent && completed));
        if (!errored) {
            // info (MA-env:627-656) and counters: staged by lane 0, stored as two coalesced runs
            const int goals_total = lifelong ? sc[MAPF_CTR_GOALS_REACHED_TOTAL] : reached_cnt;
            const int steps = max(sc[MAPF_CTR_STEP_COUNT], 1);
            float2 *xi = reinterpret_cast<float2 *>(l.xinfo);
            uint4 *xs = reinterpret_cast<uint4 *>(l.xinfo + 64);
            if (a == 0) {
                xi[0] = make_float2((float)goals_step, (float)goals_total);
                xi[1] = make_float2((float)blocking_step, (float)sc[MAPF_CTR_BLOCKING_COUNT]);
                xi[2] = make_float2((float)deadlock, (float)livelock);
                xi[3] = make_float2((float)dl_event, (float)ll_event);
                xi[4] = make_float2((float)sc[MAPF_CTR_DEADLOCK_EVENTS], (float)sc[MAPF_CTR_LIVELOCK_EVENTS]);
                xi[5] = make_float2((float)sc[MAPF_CTR_DEADLOCK_STEPS], (float)sc[MAPF_CTR_LIVELOCK_STEPS]);
                xi[6] = make_float2((float)completed_cnt / (float)N, (float)goals_total / (float)steps);
                if (dec.do_reset) {  // a re-placed env stores the counters reset() leaves (MA-env:440-455)
                    xs[0] = xs[1] = make_uint4(0, 0, 0, 0);
                    xs[2] = make_uint4(0, sc[MAPF_CTR_EPISODES_DONE] + 1, 1, sc[11]);
                } else {
                    xs[0] = make_uint4(sc[0], sc[1], sc[2], sc[3]);
                    xs[1] = make_uint4(sc[4], sc[5], sc[6], sc[7]);
                    xs[2] = make_uint4(sc[8], sc[9], sc[10], sc[11]);
                }
            }
            wave_lds_sync();
            if (io.info_all && lane < 7) reinterpret_cast<float2 *>(io.info_all + (size_t)env * MAPF_INFO_ALL)[lane] = xi[lane];
            if (lane < 3) store_state16(io.scal + (size_t)env * kScalInts + lane * 4, xs[lane]);
            Lane img = st;
            if (dec.do_reset) {  // _reset_lock_tracking MA-env:360-372
                img.moved = img.failed = img.progress = 0ull;
                img.dist = make_uint4(0, 0, 0, 0);
            }
            if (is_agent) store_lane_hist(io.agents, io.bn8, idx0, lane, img);
        } else {
            // the reference raised mid-loop (MA-env:502-506): nothing after the loop ran -- history, lock counters and
            // blocking keep their values; step_count and the goals counted before the exception stay
            sc_keep[MAPF_CTR_MAY_FINISH] = 1;  // agents before the bad one did move: the hint of the previous step is stale
            if (a == 0) {
                store_state16(io.scal + (size_t)env * kScalInts, make_uint4(sc_keep[0], sc_keep[1], sc_keep[2], sc_keep[3]));
                store_state16(io.scal + (size_t)env * kScalInts + 4, make_uint4(sc_keep[4], sc_keep[5], sc_keep[6], sc_keep[7]));
                store_state16(io.scal + (size_t)env * kScalInts + 8, make_uint4(sc_keep[8], sc_keep[9], sc_keep[10], sc_keep[11]));
            }
        }
        MAPF_STAMP_W2(30);
        // episode statistics (src/trainers/callbacks.py:236-345), as step_body
        if (__builtin_expect(dec.done, 0)) {
            if (a == 0) {
                int *acc = p.ep_acc + (size_t)env * MAPF_NUM_EPISODE_ACC;
                atomicAdd(acc + MAPF_ACC_EPISODES, 1);
                if (dec.term && !dec.trunc) atomicAdd(acc + MAPF_ACC_SUCCESSES, 1);
                atomicAdd(acc + MAPF_ACC_GOALS_REACHED, sc[MAPF_CTR_GOALS_REACHED_TOTAL]);
                atomicAdd(acc + MAPF_ACC_BLOCKING_COUNT, sc[MAPF_CTR_BLOCKING_COUNT]);
                atomicAdd(acc + MAPF_ACC_DEADLOCK_COUNT, sc[MAPF_CTR_DEADLOCK_EVENTS]);
                atomicAdd(acc + MAPF_ACC_LIVELOCK_COUNT, sc[MAPF_CTR_LIVELOCK_EVENTS]);
                atomicAdd(acc + MAPF_ACC_DEADLOCK_STEPS, sc[MAPF_CTR_DEADLOCK_STEPS]);
                atomicAdd(acc + MAPF_ACC_LIVELOCK_STEPS, sc[MAPF_CTR_LIVELOCK_STEPS]);
                atomicAdd(acc + MAPF_ACC_COMPLETED_AGENTS, completed_cnt);
                atomicAdd(acc + MAPF_ACC_EPISODE_STEPS, sc[MAPF_CTR_STEP_COUNT]);
            }
        }
        if (__builtin_expect(dec.slow_reset, 0)) wg_sync();  // B2 (all waves of the workgroup meet)
#ifdef MAPF_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p.dbg && threadIdx.x == 128) p.dbg[(size_t)env * kDbgRow + 7] = _t_entry;
#endif
        MAPF_STAMP_W2(31);
        return;
    }

    // =================================== state wave ===================================
    // The step counter, the stream and the free-cell count are wave-uniform, but they are fetched with VECTOR loads (an
    // opaque per-lane zero in the address): scalar loads return out of order and share their counter with LDS, so every
    // LDS result of the move phase would wait for the slowest of them (measured: -2 % of the c5 step)
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    const int step0 = io.scal[(size_t)env * kScalInts + MAPF_CTR_STEP_COUNT + vz];
    uint32_t nsg = (lifelong || deterministic) ? kSlotInvalid : slots_of(io.scal, io.B)[idx];
    // Lifelong mode: what a respawn reads from global memory -- the stream state, the free-cell count, and (below, once the
    // hot plane is here) the row-major ranks of my old cell, my target cell and my goal -- is requested a whole move phase
    // before it is needed
    Pcg g_ll;
    g_ll.shi = g_ll.slo = g_ll.ihi = g_ll.ilo = 0;
    g_ll.has32 = g_ll.uinteger = 0;
    int F_ll = 0;
    if (lifelong) {
        pcg_load(g_ll, streams_of(io.scal, io.B, N) + (size_t)env * 6 + vz);
        F_ll = free_counts_of(io.scal, io.B, N)[env + vz];
    }
    __builtin_amdgcn_sched_barrier(0);
    warm_scalar_cache(pp, tail);
    __builtin_amdgcn_sched_barrier(0);
    MAPF_WIDE_MASK_CHECK();
    const WideLds l = carve_wide(lds_raw, H, N * K::L(p));
    const bool bad = is_agent && (act < 0 || act > 4);
    const uint64_t badm = __ballot(bad);
    const bool errored = badm != 0;
    const int n_live = errored ? (int)__builtin_ctzll(badm) : N;
    const bool live = is_agent && a < n_live;
    if (__builtin_expect(errored, 0)) {
        if (bad && a == n_live) raise_error(p, MAPF_ERR_BAD_ACTION, env, a, act);
    }
    if (!live) act = 0;
    const uint32_t old = is_agent ? (hot.x & 0xFFFFu) : (uint32_t)kIdleCell;
    uint32_t goal = is_agent ? (hot.x >> 16) : (uint32_t)kIdleGoal;
    const uint32_t start = hot.y & 0xFFFFu, fl0 = (hot.y >> 16) & 0xFFu, pass = hot.y >> 24;
    MAPF_STAMP(0);
    // ---- move phase (MA-env:502-526): "inside the grid and no obstacle" is the agent's pass bit for the action ----
    const int dr = (act == 1) ? -1 : ((act == 3) ? 1 : 0);
    const int dc = (act == 2) ? 1 : ((act == 4) ? -1 : 0);
    const bool want = live && act != 0 && ((pass >> ((act - 1) & 3)) & 1u) != 0;
    const int r_old = (int)(old >> 8), c_old = (int)(old & 255u);
    const int tr = r_old + dr, tc = c_old + dc;
    const uint32_t tgt = want ? (uint32_t)((tr << 8) | tc) : kNoCell;
    int rankOld = 0x7FFFFFFF, rankTgt = 0x7FFFFFFF, rankGoal = 0x7FFFFFFF;
    if (lifelong && is_agent) {
        // (global-address-space loads: through the generic pointer of Params they would be FLAT loads, which also count in
        //  the LDS wait counter -- every LDS read of the move phase would wait for these gathers)
        const uint16_t *frank = io.free_rank + (size_t)env * (H * W);  // (a kernel argument: global address space, no wait)
        // (three independent loads: `want ? frank[target] : rankOld` made the second one wait for the first -- a global
        //  round trip in front of the move phase of every step, 0.9 k cycles in the stamps)
        rankOld = (int)frank[r_old * W + c_old];
        rankTgt = (int)frank[want ? tr * W + tc : r_old * W + c_old];
        rankGoal = (int)frank[(int)(goal >> 8) * W + (int)(goal & 255u)];
    }
    MAPF_STAMP(16);
    wg_sync();  // B0: the aux wave has cleared the bit rows (it and the observation wave arrive long before this wave's loads)
    // Who stands where before the move; who wants which cell -- ONE LDS round trip: the byte map and the occupancy row are
    // written, the `want` row is or-ed WITH RETURN (every contender for a cell but the first sees the bit set), and the
    // reads of the target's occupancy follow in the same batch (a wave's DS operations execute in order).
    if (is_agent) {
        l.ownO[wide_cell(old)] = (uint8_t)a;
        atomicOr(reinterpret_cast<unsigned long long *>(&l.occO[wide_row(old)]), wide_bit(old));
        atomicOr(reinterpret_cast<unsigned long long *>(&l.goalb[wide_row(goal)]), wide_bit(goal));  // (a respawn below puts it right)
    }
    unsigned long long seen = 0ull;
    if (want) seen = atomicOr(reinterpret_cast<unsigned long long *>(&l.wantb[tr + kRowPad]), 1ull << (tc & 63));
    wave_lds_sync();
    uint32_t cur = old;
    if (__any(want)) {
        const uint64_t orow = l.occO[want ? tr + kRowPad : kRowPad];
        const int occ = (int)l.ownO[want ? tr * 64 + tc : 0];
        const bool occupied = want && ((orow >> (tc & 63)) & 1ull) != 0;
        const bool later = want && ((seen >> (tc & 63)) & 1ull) != 0;  // somebody registered for my target before me
        // Contended cells are rare: the later contenders name their targets (one broadcast each), everybody compares; then
        // the first registrants of those cells -- who saw nothing -- do the same, so that every contender knows all the
        // lower-index ones.
        bool contended = later;
        uint64_t cont = 0;  // lower-index contenders for my target
        uint64_t u = __ballot(later);
        while (u) {
            const int j = (int)__builtin_ctzll(u);
            u &= u - 1;
            const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)tgt, j);
            const bool same = want && tj == tgt;
            contended |= same;
            cont |= (same && j < a) ? (1ull << j) : 0ull;
        }
        u = __ballot(contended && !later);
        while (u) {
            const int j = (int)__builtin_ctzll(u);
            u &= u - 1;
            const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)tgt, j);
            cont |= (want && tj == tgt && j < a) ? (1ull << j) : 0ull;
        }
        const uint64_t below = (1ull << a) - 1ull;
        const uint64_t occ_bit = occupied ? (1ull << (occ & 63)) : 0ull;
        const uint64_t occ_low = occ_bit & below;       // occupant has a lower index: blocked unless it moved away
        const bool occ_high = (occ_bit & ~below) != 0;  // occupant has a higher index (its turn comes later): blocked
        const uint64_t dep = cont | occ_low;
        bool resolved = !want, mv = false;
        uint64_t R = __ballot(resolved), M = 0;
#pragma unroll 1
        for (int it = 0; it <= N; it++) {
            if (__all(resolved)) break;
            if (!resolved && (dep & ~R) == 0) {
                mv = !(occ_high || (occ_low & ~M) != 0 || (cont & M) != 0);
                resolved = true;
            }
            R = __ballot(resolved);
            M = __ballot(mv);
        }
        cur = mv ? tgt : old;
    }
    const bool moved = cur != old;
    MAPF_STAMP(2);
    // occupancy after the move, and the cells whose occupancy depends on the observer's turn
    if (is_agent) {
        l.ownN[wide_cell(cur)] = (uint8_t)a;
        atomicOr(reinterpret_cast<unsigned long long *>(&l.occN[wide_row(cur)]), wide_bit(cur));
        if (moved) {
            atomicOr(reinterpret_cast<unsigned long long *>(&l.mov[wide_row(old)]), wide_bit(old));
            atomicOr(reinterpret_cast<unsigned long long *>(&l.mov[wide_row(cur)]), wide_bit(cur));
        }
    }
    // ---- goal logic of the arrivals (MA-env:538-563); lifelong: respawn in agent order (MA-env:284-304) ----
    bool reached = (fl0 & kFlagReached) != 0, completed = (fl0 & kFlagCompleted) != 0;
    const bool pressure_prev = (fl0 & kFlagPressure) != 0;
    const bool on_goal = live && cur == goal;
    bool grs = false;
    bool reassigned = false;
    if (!lifelong) {
        if (on_goal && !reached) {
            reached = true;
            completed = true;
            grs = true;
        }
    } else {
        const uint64_t arr = __ballot(on_goal);
        if (arr) {  // wave-uniform, one wave-step in twelve at c5
            reassigned = true;
            wave_lds_sync();
            Pcg g = g_ll;
            const int F = F_ll;
            const int rankCur = moved ? rankTgt : rankOld;
            // index + 1 of the agent standing on my goal cell after (on) / before (oo) its move, 0 = nobody
            int on = 0, oo = 0;
            if (is_agent) {
                const uint64_t nrow = l.occN[wide_row(goal)], orow = l.occO[wide_row(goal)];
                const int gi = wide_cell(goal);
                on = ((nrow >> (goal & 63u)) & 1ull) ? (int)l.ownN[gi] + 1 : 0;
                oo = ((orow >> (goal & 63u)) & 1ull) ? (int)l.ownO[gi] + 1 : 0;
            }
            uint64_t u = arr;
            while (u) {  // respawns happen in agent order, each sees the state "at time i" (MA-env:554)
                const int i = (int)__builtin_ctzll(u);
                u &= u - 1;
                // occupied cells at time i; goals of everybody else (own old goal is released first, MA-env:286-288)
                const bool Gact = is_agent && a != i;
                const int rankP = (a <= i) ? rankCur : rankOld;
                const int rankG = Gact ? rankGoal : 0x7FFFFFFF;
                bool dup = (on != 0 && on - 1 <= i) || (oo != 0 && oo - 1 > i);  // my goal cell is also occupied -> count it once
                dup = dup && Gact;
                const int overlap = __popcll(__ballot(dup));
                const int k = F - N - (N - 1) + overlap;  // candidate_indices.size MA-env:295
                uint32_t r = 0;
                if (k <= 0) {
                    if (a == i) raise_error(p, MAPF_ERR_NO_RESPAWN, env, i, k);
                } else {
                    bool stuck = false;
                    r = pcg_bounded(g, (uint32_t)(k - 1), stuck);  // rng.integers(k) MA-env:300
                    if (stuck && a == i) raise_error(p, MAPF_ERR_RNG_GUARD, env, i, k);
                }
                // r-th candidate in row-major order = free-rank y with y = r + #{excluded ranks <= y}
                int y = (int)r;
                for (int it = 0; it <= 2 * N; it++) {  // converges in <= #excluded + 1 rounds
                    const int cnt = __popcll(__ballot(is_agent && rankP <= y)) + __popcll(__ballot(Gact && !dup && rankG <= y));
                    const int y2 = (int)r + cnt;
                    const bool changed = k > 0 && y2 != y;
                    y = y2;
                    if (!changed) break;
                }
                if (k > 0) {
                    // who stands on the chosen cell later in this step (it is free of agents at time i, not after)
                    const uint64_t bc = __ballot(is_agent && rankCur == y);
                    const uint64_t bo = __ballot(is_agent && rankOld == y);
                    if (a == i) {
                        goal = as_global(p.free_cells)[(size_t)env * p.HW + y];  // MA-env:301-303
                        rankGoal = y;
                        on = bc ? (int)__builtin_ctzll(bc) + 1 : 0;
                        oo = bo ? (int)__builtin_ctzll(bo) + 1 : 0;
                    }
                }
            }
            if (a == 0) pcg_store(g, p.rng + (size_t)env * 6);
            g_ll = g;
            if (on_goal) {  // MA-env:547-556
                grs = true;
                completed = true;
                reached = false;
            }
            // the goal row: every released goal goes out first, then every new one comes in (a new goal may be a cell
            // that another arrival of this step released)
            const uint32_t goal_was = hot.x >> 16;
            if (on_goal) atomicAnd(reinterpret_cast<unsigned long long *>(&l.goalb[wide_row(goal_was)]), ~wide_bit(goal_was));
            if (on_goal) atomicOr(reinterpret_cast<unsigned long long *>(&l.goalb[wide_row(goal)]), wide_bit(goal));
        }
    }
    // intents (MA-env:608-623: intended_next of the agents that have not reached their goal)
    if (is_agent) {
        if (!errored && !reached && tr >= 0 && tr < H && tc >= 0 && tc < W)
            atomicOr(reinterpret_cast<unsigned long long *>(&l.intent[tr + kRowPad]), 1ull << (tc & 63));
    }
    l.tab[a] = make_uint4(old | (cur << 16), goal | ((grs ? 1u : 0u) << 16), 0u, 0u);
    if (a == 0) l.ctl[0] = reassigned ? 1u : 0u;
    wg_sync();  // B1
    MAPF_STAMP(19);
    // ---- after the moves: rewards, per-agent info, done flags, the hot plane ----
    const WideEnd dec = wide_decide<K>(p, io, N, on_goal, is_agent, errored, step0 + 1, nsg);
    const uint64_t irow = l.intent[wide_row(is_agent ? cur : 0u)];
    const bool blocking = is_agent && reached && !moved && ((irow >> (cur & 63u)) & 1ull) != 0;
    // The inline draw of an env that ends its episode comes FIRST: the observation wave waits for it at B2 (its reset
    // observation is what the launch then ends with), the per-agent outputs below do not depend on it
    uint32_t ns_drawn = kIdleCell, ng_drawn = kIdleGoal;
    if (__builtin_expect(dec.slow_reset, 0)) {
        // ---- the env ends its episode without a pre-drawn placement (lifelong: always): draw inline, MA-env:267-282 ----
        const bool staged = !lifelong && __ballot(is_agent && a == 0 && slot_word_staged(nsg)) != 0;
        const uint64_t *rng_src = staged ? p.vis_rng : p.rng;
        int16_t *hs = l.scratch;
        const int16_t *out = hs + sample_out_off_i16(N);
        PcgPre pre{false, {}, 0, {}, {}};
        if (lifelong) {  // the stream is in registers already (advanced by this step's respawns)
            pre.have = true;
            pre.have_j = false;
            pre.g = g_ll;
            pre.pop = F_ll;
        }
        const bool sampled = sample_starts_goals_parallel<LPE>(p, hs, lane, a, env, true, true, N, rng_src, pre);
        if (!sampled) {  // F = 2N or a Lemire rejection: the sequential restatement
            int16_t *outs = hs + p.hash_cap;
            if (a == 0) {
                Pcg g;
                if (lifelong) g = g_ll; else pcg_load(g, rng_src + (size_t)env * 6);
                const int hash_cap = p.hash_cap, mask = hash_cap - 1, size = 2 * N, pop = p.n_free[env];
                bool stuck = false;
                for (int k = 0; k < hash_cap; k++) hs[k] = -1;
                for (int j = pop - size; j < pop; j++) {  // Floyd
                    const int val = (int)pcg_bounded(g, (uint32_t)j, stuck);
                    int loc = val & mask;
                    for (int pr = 0; hs[loc] != -1 && hs[loc] != val && pr < hash_cap; pr++) loc = (loc + 1) & mask;
                    if (hs[loc] == -1) {
                        hs[loc] = (int16_t)val;
                        outs[j - pop + size] = (int16_t)val;
                    } else {
                        loc = j & mask;
                        for (int pr = 0; hs[loc] != -1 && pr < hash_cap; pr++) loc = (loc + 1) & mask;
                        hs[loc] = (int16_t)j;
                        outs[j - pop + size] = (int16_t)j;
                    }
                }
                for (int i = size - 1; i >= 1; i--) {  // _shuffle_int tail shuffle
                    const int j = (int)pcg_bounded(g, (uint32_t)i, stuck);
                    const int16_t t = outs[j];
                    outs[j] = outs[i];
                    outs[i] = t;
                }
                if (stuck) raise_error(p, MAPF_ERR_RNG_GUARD, env, 0, 0);
                pcg_store(g, p.rng + (size_t)env * 6);
            }
            wave_lds_sync();
            out = outs;
        }
        uint32_t ns = kIdleCell, ng = kIdleGoal;
        for (int k = lane; k < 2 * wide_rows(H); k += 64) l.occR[k] = 0ull;  // (occR and goalR, adjacent)
        if (is_agent) {
            const uint16_t *fc = p.free_cells + (size_t)env * p.HW;
            const int top = p.HW - 1;  // idx entries are ranks < F <= HW; the clamp only bounds the address
            ns = fc[min(max((int)out[a], 0), top)];
            ng = fc[min(max((int)out[N + a], 0), top)];
            atomicOr(reinterpret_cast<unsigned long long *>(&l.occR[wide_row(ns)]), wide_bit(ns));
            atomicOr(reinterpret_cast<unsigned long long *>(&l.goalR[wide_row(ng)]), wide_bit(ng));
            if (nsg != kSlotInvalid) slots_of(io.scal, io.B)[idx0 + a] = kSlotInvalid;  // overtaken (staged): Params::rng is the visible stream again
        }
        l.tab[a].z = ns | (ng << 16);
        wg_sync();  // B2
        ns_drawn = ns;
        ng_drawn = ng;
    }
    if (!errored) {
        // +1 each when all stand on their goals, -1 at the step limit for an agent off its goal (finite mode), MA-env:668-690
        const float term_reward = !(dec.term | dec.trunc) ? 0.0f : (!dec.trunc ? 1.0f : ((lifelong || on_goal) ? 0.0f : -1.0f));
        const float reward = (grs ? 0.5f : 0.0f) + term_reward;
        if (is_agent) {
            if (io.rewards) io.rewards[idx0 + a] = reward;
            if (io.info_agent) {
                uchar2 ia;
                ia.x = blocking ? 1 : 0;
                ia.y = grs ? 1 : 0;
                reinterpret_cast<uchar2 *>(io.info_agent)[idx0 + a] = ia;
            }
        }
        if (a == 0) {
            if (io.terminated) io.terminated[env] = (uint8_t)dec.term;
            if (io.truncated) io.truncated[env] = (uint8_t)dec.trunc;
        }
    }
    Lane img;
    img.pos = cur;
    img.goal = goal;
    img.start = start;
    img.flags = (reached ? kFlagReached : 0) | (completed ? kFlagCompleted : 0) |
                ((errored ? pressure_prev : blocking) ? kFlagPressure : 0);
    img.moved = img.failed = img.progress = 0ull;
    img.dist = make_uint4(0, 0, 0, 0);
    if (__builtin_expect(dec.fast_reset, 0)) {  // re-placed from the slot / the fixed starts: the image reset() leaves (MA-env:440-455)
        const uint32_t rs = deterministic ? (start | (goal << 16)) : nsg;
        img.start = rs & 0xFFFFu;
        img.goal = rs >> 16;
        img.pos = img.start;
        img.flags = 0u;
        if (!deterministic && is_agent) slots_of(io.scal, io.B)[idx0 + a] = kSlotInvalid;  // consumed
    }
    if (__builtin_expect(dec.slow_reset, 0)) {
        img.start = ns_drawn;
        img.goal = ng_drawn;
        img.pos = ns_drawn;
        img.flags = 0u;
    }
    if (is_agent) store_lane_hot(io.agents + idx0, (size_t)a, img, wide_pass_bits(l.freeb, img.pos));
    MAPF_STAMP(8);
#ifdef MAPF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    MAPF_STAMP(9);
    MAPF_STAMP_ENTRY_STORE();
}

// ================================================================================================
// Single-agent (CTE) sibling env: /root/reference/src/environments/reference_model_single_agent.py ("SA-env").
// One policy drives all agents: same sequential move rule, but a FULL-GRID observation (codes 0 free,
// 1 obstacle, 2+2i agent i, 3+2i goal i; agents drawn over goals, SA-env:407-441), a joint 5N action mask
// (incl. the reference's quirk that an obstacle, being odd, counts as enterable, SA-env:483-493), a scalar
// float64 reward (SA-env:246-363) and rejection-sampled starts/goals (SA-env:158-191).
// LDS staging holds one row of H*W + 5N floats per env.
// ================================================================================================
struct CteIo {
    uint2 *agents;  // plane 0 of the agent state
    uint32_t bn8;
    int *scal;
    const uint64_t *grid_rows;
    int B, H, W, steps_per_episode;
    int col_pad;
    int lds_tab_off, lds_stage_off, lds_scratch_off;
    double blocking_penalty, move_after_goal_penalty;
    const int8_t *actions;
    float *obs;
    double *reward;
    uint8_t *terminated, *truncated;
    float *info;  // [B][4] blocking_count_step, goals_reached_step, goals_reached_total, blocking_count_total
    float *final_obs;
    const uint8_t *env_mask;
    int auto_reset;
    int sampler_blocks;  // k_cte_step, single steps: workgroups at the FRONT of the grid that pre-draw next-episode placements
    int lds_scratch2_off;  // the draw scratch of their two waves (cte_sampler_groups groups each)
};

// full-grid observation + joint mask of the groups' current state -> staging rows, in two parts: the obstacle
// floats of the whole grid (static: the step kernel's observation wave writes them while the moves are resolved)
// and the agents / goals / mask on top of them
template <int LPE>
__device__ __forceinline__ void cte_fill_grid(const CteIo &io, const uint64_t *lrows, float *srow, bool env_ok, int a) {
    const int H = io.H, W = io.W;
    if (!env_ok) return;
    if ((W & 3) == 0) {
        // four cells per lane and round: the lanes of the group take consecutive 4-cell chunks of the row-major
        // grid; (row, chunk) advance incrementally, so there is no division in the loop
        const int cpr = W >> 2;                     // chunks per row
        const int dq_r = LPE / cpr, dq_c = LPE - dq_r * cpr;
        int r = a / cpr, c = a - r * cpr;
        constexpr uint32_t KS = 0x00204081u, MS = 0x01010101u;  // bit i of a nibble -> LSB of byte i
        for (int q = a; q < H * cpr; q += LPE) {
            const uint32_t nib = (uint32_t)(lrows[r] >> (io.col_pad + 4 * c)) & 15u;
            const uint32_t by = (nib * KS) & MS;
            float *d = srow + r * W + 4 * c;
            d[0] = (float)(by & 0xFFu);
            d[1] = (float)((by >> 8) & 0xFFu);
            d[2] = (float)((by >> 16) & 0xFFu);
            d[3] = (float)(by >> 24);
            r += dq_r;
            c += dq_c;
            if (c >= cpr) {
                c -= cpr;
                r++;
            }
        }
    } else {
        for (int r = 0; r < H; r++) {
            const uint64_t bits = lrows[r] >> io.col_pad;
            for (int c = a; c < W; c += LPE) srow[r * W + c] = (float)((bits >> c) & 1ull);
        }
    }
}
template <int LPE>
__device__ __forceinline__ void cte_overlay(const CteIo &io, float *srow, bool is_agent, int a, uint32_t pos,
                                            uint32_t goal) {
    const int H = io.H, W = io.W, HW = H * W;
    wave_lds_sync();
    if (is_agent) srow[(goal >> 8) * W + (goal & 255u)] = (float)(2 * a + 3);  // goals first ...
    wave_lds_sync();
    if (is_agent) srow[(pos >> 8) * W + (pos & 255u)] = (float)(2 * a + 2);   // ... agents overwrite goals
    wave_lds_sync();
    if (is_agent) {
        const int x = (int)(pos >> 8), y = (int)(pos & 255u);
        float *m = srow + HW + 5 * a;
        auto open = [&](int r, int c) {  // "== 0 or odd" on the cell code (SA-env:483)
            const int v = (int)srow[r * W + c];
            return (v == 0 || (v & 1)) ? 1.0f : 0.0f;
        };
        m[0] = 1.0f;
        m[1] = x > 0 ? open(x - 1, y) : 0.0f;
        m[2] = y < W - 1 ? open(x, y + 1) : 0.0f;
        m[3] = x < H - 1 ? open(x + 1, y) : 0.0f;
        m[4] = y > 0 ? open(x, y - 1) : 0.0f;
    }
    wave_lds_sync();
}
template <int LPE>
__device__ __forceinline__ void cte_observe(const CteIo &io, const int N, const uint64_t *lrows, float *srow,
                                            bool env_ok, bool is_agent, int a, uint32_t pos, uint32_t goal) {
    (void)N;
    cte_fill_grid<LPE>(io, lrows, srow, env_ok, a);
    cte_overlay<LPE>(io, srow, is_agent, a, pos, goal);
}

// SA-env:158-191 for the groups with do_reset: one rng.choice(F) (= bounded(F-1)) per attempt, rejection until unique.
// The sequential restatement: one lane, one PCG64 step and one free-cell gather per attempt -- 2N dependent global round
// trips, ~22 k cycles at N = 4, paid by every launch in which an env of the batch ends its episode (round 3 measured this
// env at 5.0 us per step with all episodes in phase and never with staggered ones: 13.1 us).
template <int LPE>
__device__ __forceinline__ void cte_sample_sequential(const Params &p, const int N, uint16_t *starts, uint16_t *goals, int env) {
    Pcg g;
    pcg_load(g, p.rng + (size_t)env * 6);
    const uint16_t *fc = p.free_cells + (size_t)env * p.HW;
    const uint32_t top = (uint32_t)(p.n_free[env] - 1);
    bool gave_up = false, stuck = false;
    for (int i = 0; i < N && !gave_up; i++) {
        int guard = 0;
        for (;; guard++) {
            const uint16_t cell = fc[pcg_bounded(g, top, stuck)];
            bool clash = false;
            for (int j = 0; j < i; j++) clash |= starts[j] == cell;
            if (!clash) { starts[i] = cell; break; }
            if (guard > (1 << 20)) { gave_up = true; break; }  // F >= 2N is checked at set_grids; guard only
        }
    }
    for (int i = 0; i < N && !gave_up; i++) {
        int guard = 0;
        for (;; guard++) {
            const uint16_t cell = fc[pcg_bounded(g, top, stuck)];
            bool clash = false;
            for (int j = 0; j < i; j++) clash |= goals[j] == cell;
            for (int j = 0; j < N; j++) clash |= starts[j] == cell;
            if (!clash) { goals[i] = cell; break; }
            if (guard > (1 << 20)) { gave_up = true; break; }
        }
    }
    if (gave_up) raise_error(p, MAPF_ERR_FEW_FREE, env, 0, 0);
    if (stuck) raise_error(p, MAPF_ERR_RNG_GUARD, env, 0, 0);
    pcg_store(g, p.rng + (size_t)env * 6);
}
// The same draw spread over the lanes of the group (round 4).  An attempt is accepted iff its cell has not come up before in
// the WHOLE sequence of attempts: a start clashes with the earlier starts, a goal with the earlier goals and with every start,
// and every rejected attempt repeats a value that an earlier attempt had accepted.  So with K = 2N + 8 attempts made at once
// -- raw outputs by PCG64 jump-ahead (one output = two attempts per lane), all Lemire products and free-cell gathers side by
// side -- the accepted attempts are the FIRST OCCURRENCES, the first N of them are the starts, the next N the goals, and the
// stream has advanced by the attempts up to the 2N-th first occurrence.  Fewer than 2N distinct cells among the K attempts,
// or an attempt Lemire's test might reject (`left < F`, probability F / 2^32), sends the group to the sequential restatement.
// Group scratch: raw[K + 1] uint32 | cand[K] uint16 | starts[N] | goals[N].
struct CtePre {  // the stream and the free-cell count of the lane's env, fetched ahead by the caller (lane 0 of each group), or nothing
    bool have;
    uint4 w0, w1, w2;  // the six 64-bit stream words
    int F;
};
template <int LPE>
__device__ __forceinline__ void cte_sample_starts_goals(const Params &p, const int N, int16_t *scratch, int grp, int a,
                                                        int env, bool do_reset, bool is_agent, Lane &st,
                                                        const CtePre &pre = CtePre{false, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, 0}) {
    const int lane = grp * LPE + a;
    const int K = 2 * N + 8;
    uint32_t *raw = reinterpret_cast<uint32_t *>(scratch + grp * p.scratch_i16);
    uint16_t *cand = reinterpret_cast<uint16_t *>(raw + K + 2);
    // (K + 2) * 2 + K + 1 + 2N int16 must fit the group's scratch; at most two outputs per lane, jump tables up to 64
    const bool par = (2 * (K + 2) + ((K + 1) & ~1) + 2 * N) <= p.scratch_i16 && (K + 1) / 2 <= 2 * LPE && (K + 1) / 2 <= 64 &&
                     !(p.flags & MAPF_FLAG_SEQUENTIAL_RESET);
    uint16_t *starts = par ? cand + ((K + 1) & ~1) : reinterpret_cast<uint16_t *>(scratch + grp * p.scratch_i16);
    uint16_t *goals = starts + N;
    bool ok = do_reset && par;
    if (__any(ok)) {
        Pcg g;
        g.shi = g.slo = g.ihi = g.ilo = 0;
        g.has32 = g.uinteger = 0;
        int F = 2;
        if (pre.have) {  // fetched with the wave's first loads by lane 0 of the group: one global round trip less behind the episode's end
            const uint32_t w[12] = {gshfl<LPE>(pre.w0.x, 0), gshfl<LPE>(pre.w0.y, 0), gshfl<LPE>(pre.w0.z, 0), gshfl<LPE>(pre.w0.w, 0),
                                    gshfl<LPE>(pre.w1.x, 0), gshfl<LPE>(pre.w1.y, 0), gshfl<LPE>(pre.w1.z, 0), gshfl<LPE>(pre.w1.w, 0),
                                    gshfl<LPE>(pre.w2.x, 0), gshfl<LPE>(pre.w2.y, 0), gshfl<LPE>(pre.w2.z, 0), gshfl<LPE>(pre.w2.w, 0)};
            const int Fg = (int)gshfl<LPE>((uint32_t)pre.F, 0);
            if (ok) {
                g.shi = (uint64_t)w[0] | ((uint64_t)w[1] << 32); g.slo = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
                g.ihi = (uint64_t)w[4] | ((uint64_t)w[5] << 32); g.ilo = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
                g.has32 = w[8]; g.uinteger = w[10];
                F = Fg;
            }
        } else if (ok) {
            pcg_load(g, p.rng + (size_t)env * 6);
            F = p.n_free[env];
        }
        const int has = (int)g.has32;
        const int nout = (K - has + 1) >> 1;  // 64-bit outputs behind the K attempts: [buffered half] lo(o1) hi(o1) lo(o2) ...
        const U128 s0 = {g.shi, g.slo}, inc = {g.ihi, g.ilo};
        U128 stq[2] = {s0, s0};
        uint32_t hiq[2] = {0u, 0u};
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int q = a + 1 + r * LPE;
            if (ok && q <= nout) {
                const U128 ja = {kPcgJumpA[q][0], kPcgJumpA[q][1]};
                const U128 js = {kPcgJumpS[q][0], kPcgJumpS[q][1]};
                stq[r] = add128(mul128(ja, s0), mul128(js, inc));
                const uint64_t o = pcg_output(stq[r]);
                hiq[r] = (uint32_t)(o >> 32);
                raw[has + 2 * (q - 1)] = (uint32_t)o;
                raw[has + 2 * (q - 1) + 1] = hiq[r];
            }
        }
        if (ok && has && a == 0) raw[0] = g.uinteger;
        wave_lds_sync();
        // all attempts at once: bounded draw (Lemire, conservative rejection test) and the gather of its free cell
        constexpr int kPerLane = 4;  // attempts per lane: K = 2N + 8 <= 2 LPE + 8 <= 4 LPE (LPE >= 4)
        const uint16_t *fc = p.free_cells + (size_t)env * p.HW;
        uint32_t rejm = 0;   // bit i: my i-th attempt might be rejected by Lemire's test
        uint16_t mine[kPerLane];
#pragma unroll
        for (int i = 0; i < kPerLane; i++) {
            const int k = a + i * LPE;
            mine[i] = 0xFFFFu;
            if (ok && k < K) {
                const uint64_t m = (uint64_t)raw[k] * (uint32_t)F;
                rejm |= ((uint32_t)m < (uint32_t)F) ? (1u << i) : 0u;
                mine[i] = fc[min((int)(m >> 32), F - 1)];
            }
        }
        wave_lds_sync();
        // first occurrences: attempt k is one iff no earlier attempt drew the same cell.  Every attempt is broadcast inside
        // the group once (ds_bpermute: the LDS crossbar, all of them in flight together) and every lane compares it with its
        // own: as a loop over the LDS copy -- one dependent read per earlier attempt -- this test alone was 3.2 us per reset
        bool dupf[kPerLane];
#pragma unroll
        for (int i = 0; i < kPerLane; i++) dupf[i] = false;
#pragma unroll
        for (int r = 0; r < kPerLane; r++) {
            if (r * LPE < K) {
#pragma unroll 8
                for (int j = 0; j < LPE; j++) {
                    if (r * LPE + j < K) {
                        const uint32_t cj = gshfl<LPE>((uint32_t)mine[r], j);
#pragma unroll
                        for (int i = r; i < kPerLane; i++)  // (attempt r * LPE + j comes before a + i * LPE iff i > r, or i == r and j < a)
                            dupf[i] |= cj == (uint32_t)mine[i] && (i > r || j < a);
                    }
                }
            }
        }
        // their running count in attempt order (rows of LPE attempts), the attempt that completes the draw
        int accepted_before = 0, used = 0;
        bool rejected = false;
#pragma unroll
        for (int i = 0; i < kPerLane; i++) {
            const int k = a + i * LPE;
            const bool dup = dupf[i];
            const bool first = ok && k < K && !dup;
            const uint64_t fm = gballot<LPE>(first, lane);
            const int idx = accepted_before + __popcll(fm & ((1ull << a) - 1ull));
            if (first && idx < N) starts[idx] = mine[i];
            if (first && idx >= N && idx < 2 * N) goals[idx - N] = mine[i];
            const uint64_t lastm = gballot<LPE>(first && idx == 2 * N - 1, lane);
            if (lastm) used = i * LPE + (int)__builtin_ctzll(lastm) + 1;
            // a possibly rejected attempt among those consumed (attempts of this row up to the completing one, or the whole row)
            const uint64_t rj = gballot<LPE>(((rejm >> i) & 1u) != 0, lane);
            const uint64_t upto = lastm ? ((2ull << __builtin_ctzll(lastm)) - 1ull) : ~0ull;
            if (used == 0 || lastm) rejected |= (rj & upto) != 0;
            accepted_before += __popcll(fm);
        }
        ok = ok && used > 0 && !rejected;
        if (ok) {  // the stream after `used` attempts: state of the last output consumed; an unused high half stays buffered
            const int nused = (used - has + 1) >> 1;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if (a + 1 + r * LPE == nused) {
                    Pcg f;
                    f.shi = stq[r].hi; f.slo = stq[r].lo; f.ihi = g.ihi; f.ilo = g.ilo;
                    f.has32 = (uint32_t)(has + 2 * nused - used);
                    f.uinteger = hiq[r];  // NumPy keeps the last high half in the field even once it has been handed out
                    pcg_store(f, p.rng + (size_t)env * 6);
                }
            }
        }
    }
    wave_lds_sync();
    if (do_reset && !ok && a == 0) cte_sample_sequential<LPE>(p, N, starts, goals, env);
    wave_lds_sync();
    if (do_reset && is_agent) {
        st.start = starts[a];
        st.goal = goals[a];
    }
    wave_lds_sync();
}

template <int LPE>
__global__ __launch_bounds__(64) void k_cte_reset(const Params *__restrict__ pp, const CteIo io) {
    const Params &p = *pp;
    constexpr int G = 64 / LPE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint64_t *lrows = reinterpret_cast<uint64_t *>(lds_raw);
    float *stage = reinterpret_cast<float *>(lds_raw + io.lds_stage_off);
    int16_t *scratch = reinterpret_cast<int16_t *>(lds_raw + io.lds_scratch_off);
    const int lane = threadIdx.x, grp = lane / LPE, a = lane % LPE;
    const int env0 = blockIdx.x * G, ngroups = min(G, io.B - env0), N = p.N;
    const bool env_ok = grp < ngroups;
    const int env = env_ok ? env0 + grp : io.B - 1;
    const bool is_agent = env_ok && a < N;
    const int row_len = io.H * io.W + 5 * N;
    load_rows_to_lds<LPE>(io.grid_rows, io.H, lrows, lane, env0, ngroups);
    Lane st;
    load_lane_hot(io.agents, (size_t)env * N + min(a, N - 1), is_agent, st);
    const bool do_reset = env_ok && (io.env_mask == nullptr || io.env_mask[env] != 0);
    wave_lds_sync();
    if (!(p.flags & MAPF_FLAG_DETERMINISTIC)) {
        // a placement pre-drawn for the env's next episode (k_cte_step) IS what rng.choice returns now: its draw is in the
        // stream already
        const size_t si = (size_t)env * N + min(a, N - 1);
        const uint32_t nsg = p.next_sg[si];
        const bool slot_ok = gballot<LPE>(is_agent && !slot_word_valid(nsg), lane) == 0;
        cte_sample_starts_goals<LPE>(p, N, scratch, grp, a, env, do_reset && !slot_ok, is_agent, st);
        if (do_reset && slot_ok && is_agent) {
            st.start = nsg & 0xFFFFu;
            st.goal = nsg >> 16;
            p.next_sg[si] = kSlotInvalid;
        }
    }
    if (do_reset) {  // SA-env:222-232
        st.pos = st.start;
        st.flags = 0;
    }
    if (io.obs) {
        cte_observe<LPE>(io, N, lrows + grp * (io.H + 2 * kRowPad) + kRowPad, stage + (size_t)grp * row_len, env_ok, is_agent, a, st.pos, st.goal);
        Io fio;
        fio.obs = io.obs;
        fio.final_obs = nullptr;
        flush_rows<KRuntime, LPE>(fio, stage, lane, env0, ngroups, do_reset ? 0 : 2, row_len);
    }
    if (do_reset) {
        if (is_agent)
            store_lane_hot(io.agents, (size_t)env * N + a, st,
                           agent_pass_bits(lrows + grp * (io.H + 2 * kRowPad) + kRowPad, st.pos, io.col_pad, io.W));
        if (a == 0) {
            int4 *sp = reinterpret_cast<int4 *>(io.scal + (size_t)env * kScalInts);
            sp[0] = make_int4(0, 0, 0, 0);  // step_count, -, _episode_blocking_count, -
            io.scal[(size_t)env * kScalInts + MAPF_CTR_MAY_FINISH] = 1;  // (conservative: the next step writes the real hint)
        }
    }
}

// Two waves per workgroup like k_step: wave 1, the observation wave, fetches the obstacle rows, writes the static
// part of every env's observation row (the obstacle floats of the whole grid: most of this env's work, and
// independent of the step) while wave 0 resolves the moves, then adds agents, goals and the joint mask once wave 0
// has published the new positions (B1) and streams the rows out.  kCteW* = flag bits of the published entries.
// Round 3: wave 0 waits for the hot plane and the actions only (8 + 1 bytes per agent; this env keeps no lock history),
// takes "inside the grid and not an obstacle" from the agent's pass bits instead of the rows (no B0) and, in groups of 4
// or 8 lanes, resolves the moves by DPP + swizzle (resolve_moves_dpp) instead of through a table in LDS.
// T > 1 (mapf_cte_step_many): the same two waves run T steps in one launch -- positions stay in registers, the obstacle
// floats are written ONCE (after a row has left, the 2N cells the overlay touched are set back to 0: agents and goals
// only ever stand on free cells), per step only the actions are read and the outputs written.
constexpr uint32_t kCteWAgent = 1u, kCteWSelShift = 1u, kCteWReset = 8u,
                   kCteWFast = 16u;  // (with kCteWReset) the new placement is in entry word w already: no B2 for this env
// lane groups of a sampler wave of k_cte_step: 8 or 16 lanes when the agents fit and the step's groups are wider, else the step's
__host__ __device__ constexpr int cte_sampler_lanes(int N, int lpe) { return (N <= 8 && lpe > 8) ? 8 : ((N <= 16 && lpe > 16) ? 16 : lpe); }
__host__ __device__ constexpr int cte_sampler_groups(int N, int lpe) { return 64 / cte_sampler_lanes(N, lpe); }
// one round of a sampler wave: the first 64 / LS envs of `todo` (bit i = env 64 sw + i needs a placement), one per group of LS
// lanes; s0..s2 / sF: stream words and free-cell count of env 64 sw + lane, fetched by the caller
template <int LS>
__device__ __forceinline__ void cte_sampler_round(const Params &p, int16_t *sscr, int lane, int sw, uint64_t todo, uint4 s0, uint4 s1,
                                                  uint4 s2, int sF) {
    constexpr int GS = 64 / LS;
    const int grp = lane / LS, a = lane % LS;
    int pick = -1;
    uint64_t m = todo;
    for (int g = 0; g < GS; g++) {
        const int j = m ? (int)__builtin_ctzll(m) : -1;
        if (m) m &= m - 1;
        if (g == grp) pick = j;
    }
    const bool on = pick >= 0;
    const int env_s = on ? sw * 64 + pick : 0;
    // the picked env's stream sits in lane `pick` of the wave: hand it to lane 0 of the group that draws it
    CtePre spre{true, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, 0};
    {
        const int src = on ? pick : 0;
        spre.w0 = make_uint4(__shfl(s0.x, src), __shfl(s0.y, src), __shfl(s0.z, src), __shfl(s0.w, src));
        spre.w1 = make_uint4(__shfl(s1.x, src), __shfl(s1.y, src), __shfl(s1.z, src), __shfl(s1.w, src));
        spre.w2 = make_uint4(__shfl(s2.x, src), __shfl(s2.y, src), __shfl(s2.z, src), __shfl(s2.w, src));
        spre.F = __shfl(sF, src);
    }
    if (on && a == 0) {  // the visible stream of an env whose placement is pending: the state before this draw
        uint4 *vw = reinterpret_cast<uint4 *>(p.vis_rng + (size_t)env_s * 6);
        vw[0] = spre.w0;
        vw[1] = spre.w1;
        vw[2] = spre.w2;
    }
    Lane drawn;
    drawn.start = drawn.goal = 0u;
    cte_sample_starts_goals<LS>(p, p.N, sscr, grp, a, env_s, on, on && a < p.N, drawn, spre);
    if (on && a < p.N) p.next_sg[(size_t)env_s * p.N + a] = (drawn.start & 0xFFFFu) | (drawn.goal << 16);
}

struct CteMany {
    int T;         // steps in this launch (1 = mapf_cte_step)
    int obs_mode;  // fused launches: 0 no observation, 1 after the last step only, 2 every step ([T][B][row])
};
template <int LPE, bool FUSED>
__global__ __launch_bounds__(128) void k_cte_step(const Params *__restrict__ pp, const CteIo io, const CteMany many) {
    const Params &p = *pp;
    constexpr int G = 64 / LPE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint64_t *lrows = reinterpret_cast<uint64_t *>(lds_raw);
    uint4 *tab = reinterpret_cast<uint4 *>(lds_raw + io.lds_tab_off);
    uint4 *otab = tab + 64;  // two copies: consecutive steps of a fused launch alternate (the state wave may be a step ahead)
    float *stage = reinterpret_cast<float *>(lds_raw + io.lds_stage_off);
    int16_t *scratch = reinterpret_cast<int16_t *>(lds_raw + io.lds_scratch_off);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, grp = lane / LPE, a = lane % LPE;
    const int lead = FUSED ? 0 : io.sampler_blocks;
    if (!FUSED && (int)blockIdx.x < lead) {
        // ---- sampler workgroup (single-step launches with sampled placements): wave w looks after envs 64 (2 b + w) .. + 63 --
        // an env without a placement for its next episode that CANNOT end its episode in this launch (the MAY_FINISH hint its
        // last step left: somebody further than one move from its goal, step limit not due) gets one drawn here, beside the
        // launch instead of inside its own workgroup's step: nobody else reads or writes its stream or slot in this launch.
        const int sw = (int)blockIdx.x * 2 + wv;
        const int e = sw * 64 + lane;
        bool need = false;
        // (the env's stream and free-cell count come with the same round trip, speculatively: 52 bytes per env and launch)
        uint4 s0 = make_uint4(0, 0, 0, 0), s1 = s0, s2 = s0;
        int sF = 0;
        if (e < io.B) {
            const uint32_t w0 = p.next_sg[(size_t)e * p.N];
            const int hint = io.scal[(size_t)e * kScalInts + MAPF_CTR_MAY_FINISH];
            need = w0 == kSlotInvalid && hint == 0;
            // (fetched whether needed or not: loading them for the envs in need only made no difference, 11.3 us at 8 192 envs)
            const uint4 *rw = reinterpret_cast<const uint4 *>(p.rng + (size_t)e * 6);
            s0 = rw[0];
            s1 = rw[1];
            s2 = rw[2];
            sF = p.n_free[e];
        }
        uint64_t todo = __ballot(need);
        // ONE round per launch -- one env per lane group; the others wait for the next launch: a round (~5 k cycles) ends well
        // inside the step beside it, a wave that drew all 64 of its envs in the launch after they were re-placed together
        // (episodes in phase) was the launch's last by 8 rounds (5.24 us per step in phase against 4.98).  The groups of a
        // sampler wave are as NARROW as the agent count allows, whatever width the step's own groups have (those follow the
        // H x W observation row): at 32 lanes per env a round of two envs left a wave 32 launches behind a common reset.
        if (todo) {
            int16_t *sscr = reinterpret_cast<int16_t *>(lds_raw + io.lds_scratch2_off) + wv * cte_sampler_groups(p.N, LPE) * p.scratch_i16;
            if (cte_sampler_lanes(p.N, LPE) == 8) cte_sampler_round<8>(p, sscr, lane, sw, todo, s0, s1, s2, sF);
            else if (cte_sampler_lanes(p.N, LPE) == 16) cte_sampler_round<16>(p, sscr, lane, sw, todo, s0, s1, s2, sF);
            else cte_sampler_round<LPE>(p, sscr, lane, sw, todo, s0, s1, s2, sF);
        }
        return;
    }
    const int env0 = ((int)blockIdx.x - lead) * G, ngroups = min(G, io.B - env0), N = p.N, H = io.H, W = io.W;
    const bool env_ok = grp < ngroups;
    const int env = env_ok ? env0 + grp : io.B - 1;
    const bool is_agent = env_ok && a < N;
    const int row_len = H * W + 5 * N;
    const int T = FUSED ? many.T : 1;
    constexpr bool fused = FUSED;
    const size_t BN = (size_t)io.B * N;
    uint4 *tabg = tab + grp * LPE;
    const uint64_t *myrows = lrows + grp * (H + 2 * kRowPad) + kRowPad;
    float *srow = stage + (size_t)grp * row_len;
    // the observation tensor of step t (nullptr: this step produces none)
    auto obs_of = [&](int t) -> float * {
        if (!fused) return io.obs;
        return many.obs_mode == 2 ? io.obs + (size_t)t * io.B * row_len : ((many.obs_mode == 1 && t == T - 1) ? io.obs : nullptr);
    };

    if (wv == 1) {
        load_rows_to_lds<LPE>(io.grid_rows, H, lrows, lane, env0, ngroups);
        wave_lds_sync();
        const bool any_obs = fused ? many.obs_mode != 0 : (io.obs || io.final_obs);
        if (any_obs) cte_fill_grid<LPE>(io, myrows, srow, env_ok, a);
        for (int t = 0; t < T; t++) {
            Io fio;
            fio.obs = obs_of(t);
            fio.final_obs = fused ? nullptr : io.final_obs;
            wg_sync();  // B1: rows in LDS (first step) / positions after the move are published
            if (!(fio.obs || fio.final_obs)) continue;
            const uint4 ent = (otab + (t & 1) * 64)[lane];
            const bool ag = (ent.z & kCteWAgent) != 0;
            const bool any_reset = __any((ent.z & kCteWReset) != 0);
            if (__builtin_expect(any_reset && fio.final_obs == nullptr && fio.obs != nullptr, 0)) {
                // An env of the workgroup is re-placed and nobody asked for its terminal observation: ONE pass -- the rows of
                // the other envs as usual, behind B2 the reset placement over the re-placed env's row, one stream.  (Two passes
                // -- the step's rows, then B2, overlay and a second stream with its own drain -- were 2.5 us of the 9.1 us a
                // staggered step took after the draw had been made lane-parallel.)
                const bool rs0 = (ent.z & kCteWReset) != 0;
                cte_overlay<LPE>(io, srow, ag && !rs0, a, ent.x, ent.y);
                // B2: the placement an env drew inline is in the table -- only when some env of the workgroup HAD to draw: a
                // pre-drawn placement (kCteWFast) came with the entry
                if (__any((ent.z & (kCteWReset | kCteWFast)) == kCteWReset)) wg_sync();
                const uint4 e2 = (otab + (t & 1) * 64)[lane];
                cte_overlay<LPE>(io, srow, ag && rs0, a, e2.w & 0xFFFFu, e2.w >> 16);
                flush_rows<KRuntime, LPE>(fio, stage, lane, env0, ngroups, rs0 ? 0 : (int)((ent.z >> kCteWSelShift) & 3u), row_len);
                if (fused) {  // put the touched cells back (free cells: 0) for the next overlay
                    wave_lds_sync();
                    if (ag) {
                        const uint32_t gq = rs0 ? (e2.w >> 16) : ent.y, pq = rs0 ? (e2.w & 0xFFFFu) : ent.x;
                        srow[(gq >> 8) * W + (gq & 255u)] = 0.0f;
                        srow[(pq >> 8) * W + (pq & 255u)] = 0.0f;
                    }
                    wave_lds_sync();
                }
                continue;
            }
            cte_overlay<LPE>(io, srow, ag, a, ent.x, ent.y);
            flush_rows<KRuntime, LPE>(fio, stage, lane, env0, ngroups, (int)((ent.z >> kCteWSelShift) & 3u), row_len);
            if (fused || any_reset) {  // put the touched cells back (free cells: 0) for the next overlay
                wave_lds_sync();
                if (ag) {
                    srow[(ent.y >> 8) * W + (ent.y & 255u)] = 0.0f;
                    srow[(ent.x >> 8) * W + (ent.x & 255u)] = 0.0f;
                }
                wave_lds_sync();
            }
            if (any_reset) {
                // An env of the workgroup was re-placed: its reset observation is built HERE -- the obstacle floats are in
                // place, so it is an overlay and a flush -- from the placement the state wave drew while this wave was
                // streaming the step's rows out (B2: that placement is in the table).  (Round 3: the state wave waited for
                // this wave's flush, drew, rebuilt the whole row itself and flushed: 13.1 us per step with staggered episodes
                // against 5.0 in phase.)
                if (__any((ent.z & (kCteWReset | kCteWFast)) == kCteWReset)) wg_sync();  // B2 (as above)
                const uint4 e2 = (otab + (t & 1) * 64)[lane];
                const bool rs = ag && (e2.z & kCteWReset) != 0;
                if (fio.obs) {
                    cte_overlay<LPE>(io, srow, rs, a, e2.w & 0xFFFFu, e2.w >> 16);
                    Io f2;
                    f2.obs = fio.obs;
                    f2.final_obs = nullptr;
                    flush_rows<KRuntime, LPE>(f2, stage, lane, env0, ngroups, (e2.z & kCteWReset) != 0 ? 0 : 2, row_len);
                    if (fused) {
                        wave_lds_sync();
                        if (rs) {
                            srow[(e2.w >> 24) * W + ((e2.w >> 16) & 255u)] = 0.0f;
                            srow[((e2.w >> 8) & 255u) * W + (e2.w & 255u)] = 0.0f;
                        }
                        wave_lds_sync();
                    }
                }
            }
        }
        return;
    }

    const size_t idx = (size_t)env * N + min(a, N - 1);
    Lane st;
    uint32_t pass = load_lane_hot(io.agents, idx, is_agent, st);
    int4 sc0 = *reinterpret_cast<const int4 *>(io.scal + (size_t)env * kScalInts);
    // the env's stream and free-cell count, by lane 0 of each group (52 B per env): a reset's draw then starts without a
    // global round trip (single-step launches: a fused launch draws from the stream as its own resets leave it)
    CtePre pre{false, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, 0};
    if (!fused && !(p.flags & MAPF_FLAG_DETERMINISTIC)) {
        pre.have = true;
        if (a == 0) {
            const uint4 *rw = reinterpret_cast<const uint4 *>(p.rng + (size_t)env * 6);
            pre.w0 = rw[0];
            pre.w1 = rw[1];
            pre.w2 = rw[2];
            pre.F = p.n_free[env];
        }
    }
    // the placement pre-drawn for the env's NEXT episode (k_cte_step draws it in the step after a reset, off every path a
    // step waits for): start | goal << 16 per agent, kSlotInvalid = none.  While it is pending the env's stream array holds
    // the state AFTER that draw and Params::vis_rng the visible one (the multi-agent engine's slots; mapf_get_state knows)
    uint32_t nsg = (p.flags & MAPF_FLAG_DETERMINISTIC) ? kSlotInvalid : p.next_sg[idx];
    int step_count = sc0.x, blocking_total = sc0.z;

    // (the actions of step t + 1 are requested while step t is computed: in a fused launch that load's latency would
    //  otherwise open every step)
    int act_next = (int)io.actions[(size_t)env * N + min(a, N - 1)];
    for (int t = 0; t < T; t++) {
        float *obs_t = obs_of(t);
        const bool want_obs = fused ? obs_t != nullptr : (io.obs || io.final_obs);
        int act = is_agent ? act_next : 0;
        if (t + 1 < T) act_next = (int)io.actions[(size_t)(t + 1) * BN + (size_t)env * N + min(a, N - 1)];
        // invalid action: get_next_position raises mid-loop (SA-env:262, :401-403); agents before it were processed
        const bool bad = is_agent && (act < 0 || act > 4);
        const uint64_t badm = gballot<LPE>(bad, lane);
        const bool errored = badm != 0;
        const int n_live = errored ? (int)__builtin_ctzll(badm) : N;
        const bool live = is_agent && a < n_live;
        if (bad && a == n_live) raise_error(p, MAPF_ERR_BAD_ACTION, env, a, act);
        if (!live) act = 0;
        const int step_now = step_count + 1;  // SA-env:247

        // move (SA-env:259-274): the same sequential rule as the multi-agent env; the target's passability is the pass bit
        const uint32_t old = st.pos;
        const int dr = (act == 1) ? -1 : ((act == 3) ? 1 : 0), dc = (act == 2) ? 1 : ((act == 4) ? -1 : 0);
        const int tr = (int)(old >> 8) + dr, tc = (int)(old & 255u) + dc;
        const bool want = live && act != 0 && ((pass >> ((act - 1) & 3)) & 1u) != 0;
        const uint32_t tgt = want ? (uint32_t)((tr << 8) | tc) : kNoCell;
        const uint32_t intended1 = (uint32_t)(((tr + 1) << 8) | (tc + 1));
        uint32_t cur = old;
        if (__any(want)) {
            if constexpr (LPE <= 8) cur = resolve_moves_dpp<LPE>(a, old, tgt);
            else cur = resolve_moves<KRuntime, LPE>(p, reinterpret_cast<uint2 *>(tabg), lane, a, old, tgt);
        }
        const bool moved = cur != old;

        // goal bookkeeping (SA-env:281-286)
        bool reached_once = (st.flags & kFlagReached) != 0;
        const bool on_goal = live && cur == st.goal;
        const bool first = on_goal && !reached_once;
        reached_once = reached_once || first;
        const int k_first = __popcll(gballot<LPE>(first, lane));
        const int n_on_goal = __popcll(gballot<LPE>(on_goal, lane));
        int term = 0, trunc = 0;
        if (n_on_goal == N) term = 1;
        else if (step_now >= io.steps_per_episode) term = trunc = 1;
        const bool done = env_ok && !errored && (term | trunc);
        const bool do_reset = done && (fused || io.auto_reset);
        const bool sampled = !(p.flags & MAPF_FLAG_DETERMINISTIC);
        const bool slot_ok = sampled && gballot<LPE>(is_agent && !slot_word_valid(nsg), lane) == 0;
        // the placement a reset installs is known HERE: the pre-drawn one, or (fixed starts) the start table's
        const bool fast = do_reset && (slot_ok || !sampled);
        const uint32_t place = slot_ok ? nsg : ((st.start & 0xFFFFu) | (st.goal << 16));

        // observation after ALL moves (SA-env:288-293): hand the new positions to the observation wave
        {
            const int sel = (!env_ok || errored) ? 2 : (fused ? 0 : (do_reset ? (io.final_obs ? 1 : 2) : (io.obs ? 0 : 2)));
            // (fused launches: a step that ends an episode shows the reset observation -- built below, after B2 -- so its
            //  terminal row goes nowhere)
            const int sel_f = (fused && do_reset) ? 2 : sel;
            (otab + (t & 1) * 64)[lane] = make_uint4(cur, st.goal, (is_agent ? kCteWAgent : 0u) | ((uint32_t)sel_f << kCteWSelShift) |
                                                                      ((do_reset && want_obs) ? kCteWReset : 0u) |
                                                                      ((fast && want_obs) ? kCteWFast : 0u),
                                                     fast ? place : 0u);
        }
        wg_sync();  // B1

        // all-pairs pass: intent blocking (SA-env:303-316) and coincidences (SA-env:296-300)
        {
            uint4 ent;
            ent.x = old | (cur << 16);
            ent.y = 0;
            ent.z = (is_agent && !reached_once) ? intended1 : 0xFFFFFFFFu;
            ent.w = 0u;
            tabg[a] = ent;
        }
        wave_lds_sync();
        bool blocks = false;
        int same = 0;
        const uint32_t mycell1 = cur + 0x0101u;
        for (int j = 0; j < N; j++) {
            const uint4 e = tabg[j];
            blocks |= e.z == mycell1;
            same += ((e.x >> 16) == cur) ? 1 : 0;
        }
        wave_lds_sync();
        const bool blocking = is_agent && reached_once && !moved && blocks;
        const int m_block = __popcll(gballot<LPE>(blocking, lane));
        const int q_move = __popcll(gballot<LPE>(is_agent && reached_once && moved, lane));  // SA-env:320-325
        int coll2 = is_agent ? same - 1 : 0;  // each coinciding pair is seen from both sides
        for (int o = LPE / 2; o > 0; o >>= 1) coll2 += __shfl_xor(coll2, o, LPE);

        // reward in float64, in the reference's order of additions (SA-env:253-346)
        double reward = 0.5 * (double)k_first - (double)(coll2 / 2);
        for (int i = 0; i < N; i++) if (i < m_block) reward += io.blocking_penalty;
        for (int i = 0; i < N; i++) if (i < q_move) reward += io.move_after_goal_penalty;
        if (term && !trunc) {
            reward += (double)N;
        } else if (trunc) {
            for (int i = 0; i < N; i++) if (i < N - n_on_goal) reward -= 1.0;
        }
        const int reached_total = __popcll(gballot<LPE>(is_agent && reached_once, lane));
        if (!errored) blocking_total += m_block;  // (the exception fires before the penalties are booked)
        if (env_ok && !errored && a == 0) {
            const size_t o = (size_t)t * io.B + env;
            if (io.reward) io.reward[o] = reward;
            if (io.terminated) io.terminated[o] = (uint8_t)term;
            if (io.truncated) io.truncated[o] = (uint8_t)trunc;
            if (io.info) {
                float4 v = make_float4((float)m_block, (float)k_first, (float)reached_total, (float)blocking_total);
                *reinterpret_cast<float4 *>(io.info + o * 4) = v;
            }
        }
        st.pos = cur;
        st.flags = reached_once ? kFlagReached : 0;
        step_count = step_now;  // (also when the ValueError fires: SA-env:247 increments first)

        // Who draws in this step: an env that ends its episode without a pre-drawn placement (inline: the observation wave
        // waits for it at B2), and -- single-step launches -- an env that has none for its NEXT episode and does not end one now
        // (background: nobody waits; the stream was fetched with the wave's first loads).  One call serves both.
        const bool slow = do_reset && !fast;
        // (round 4, first version: drawn here, by this wave behind its outputs -- 7.75 us per staggered step against 8.34 without
        //  any pre-draw: the draw still made its workgroup the launch's last.  The sampler workgroups at the front of the grid do
        //  it now; without them -- fused launches never have them -- nothing is pre-drawn.)
        const bool predraw = false;
        const bool any_slow = __any(slow);
        if (__any(do_reset) || __any(predraw)) {
            Lane drawn = st;
            if (__any(slow || predraw)) {
                if (predraw && a == 0) {  // the visible stream of an env whose placement is pending (before the draw advances it)
                    uint4 *vw = reinterpret_cast<uint4 *>(p.vis_rng + (size_t)env * 6);
                    vw[0] = pre.w0;
                    vw[1] = pre.w1;
                    vw[2] = pre.w2;
                }
                cte_sample_starts_goals<LPE>(p, N, scratch, grp, a, env, slow || predraw, is_agent, drawn, pre);
            }
            if (predraw && is_agent) {
                nsg = (drawn.start & 0xFFFFu) | (drawn.goal << 16);
                p.next_sg[idx] = nsg;
            }
            if (do_reset) {
                if (slot_ok) {  // the pre-drawn placement (its draw is in the stream already); the slot is consumed
                    if (is_agent) {  // (idle lanes keep their idle cell: they take part in the move table)
                        st.start = nsg & 0xFFFFu;
                        st.goal = nsg >> 16;
                        nsg = kSlotInvalid;
                        if (!fused) p.next_sg[idx] = kSlotInvalid;
                    }
                } else if (sampled) {
                    st.start = drawn.start;
                    st.goal = drawn.goal;
                }
                st.pos = st.start;
                st.flags = 0;
                step_count = 0;
                blocking_total = 0;
            }
            if (want_obs && any_slow) {
                // the reset observation is the observation wave's (it has the obstacle floats in place): an inline draw's
                // placement goes to it through the table, behind B2
                if (slow) (otab + (t & 1) * 64)[lane].w = (st.pos & 0xFFFFu) | (st.goal << 16);
                wg_sync();  // B2
            }
        }
        // pass bits of the cell the agent stands on now (the rows are in LDS since the first B1)
        pass = is_agent ? agent_pass_bits(myrows, st.pos, io.col_pad, W) : 0u;
    }
    if (is_agent) store_lane_hot(io.agents, (size_t)env * N + a, st, pass);
    if (fused && is_agent && !(p.flags & MAPF_FLAG_DETERMINISTIC)) p.next_sg[idx] = nsg;  // (consumed by a reset of this launch, or as it was)
    {   // MAY_FINISH hint for the sampler workgroups of the NEXT launch: the env can end its episode in its next step iff the
        // step limit is due or every agent is within one move of its goal (SA-env:327-336: all on their goals at once)
        const bool far = is_agent && cell_l1(st.pos, st.goal) > 1;
        const bool may = gballot<LPE>(far, lane) == 0 || step_count + 1 >= io.steps_per_episode;
        if (env_ok && a == 0) {
            int4 *sp = reinterpret_cast<int4 *>(io.scal + (size_t)env * kScalInts);
            sp[0] = make_int4(step_count, 0, blocking_total, 0);
            io.scal[(size_t)env * kScalInts + MAPF_CTR_MAY_FINISH] = may ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// compile-time specialisations of the step kernel for the BASELINE.json shapes (SURVEY 8(d) flags:
// sensor_range 2, lock windows 8/16, nearby 2, min_neighbors 1).  X(id, N, SR, FLAGS, DW, LW, NEARBY, MINN, LPE)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kFlagsHeadline = MAPF_FLAG_NORMALIZE_GOAL_DELTA | MAPF_FLAG_ACTION_MASK |
                                    MAPF_FLAG_BLOCKING_PRESSURE | MAPF_FLAG_LOCK_METRICS;  // L = 33
constexpr uint32_t kFlagsRefDefault = MAPF_FLAG_NORMALIZE_GOAL_DELTA | MAPF_FLAG_BLOCKING_PRESSURE |
                                      MAPF_FLAG_LOCK_METRICS;  // the reference's default obs, L = 28
#if defined(MAPF_DEV_C3) || defined(MAPF_DEV_CTE)  // development builds (mapf_step.hip): the headline shape only
#define MAPF_SPECIALIZATIONS(X) X(1, 8, 2, kFlagsHeadline, 8, 16, 2, 1, 8)
#elif defined(MAPF_DEV_C5)
#define MAPF_SPECIALIZATIONS(X) X(3, 64, 2, (kFlagsHeadline | MAPF_FLAG_LIFELONG), 8, 16, 2, 1, 64)
#elif defined(MAPF_DEV_N16)
#define MAPF_SPECIALIZATIONS(X) X(6, 16, 3, kFlagsRefDefault, 8, 16, 2, 1, 16)
#elif defined(MAPF_SMALL_SHAPES)  // the checking build
#define MAPF_SPECIALIZATIONS(X)                 \
    X(1, 8, 2, kFlagsHeadline, 8, 16, 2, 1, 8)   \
    X(2, 4, 2, kFlagsHeadline, 8, 16, 2, 1, 4)   \
    X(4, 8, 2, kFlagsRefDefault, 8, 16, 2, 1, 8) \
    X(5, 4, 2, kFlagsRefDefault, 8, 16, 2, 1, 4) \
    X(6, 16, 3, kFlagsRefDefault, 8, 16, 2, 1, 16)
#else
#define MAPF_SPECIALIZATIONS(X)                                            \
    X(1, 8, 2, kFlagsHeadline, 8, 16, 2, 1, 8)                              \
    X(2, 4, 2, kFlagsHeadline, 8, 16, 2, 1, 4)                              \
    X(3, 64, 2, (kFlagsHeadline | MAPF_FLAG_LIFELONG), 8, 16, 2, 1, 64)     \
    X(4, 8, 2, kFlagsRefDefault, 8, 16, 2, 1, 8)                            \
    X(5, 4, 2, kFlagsRefDefault, 8, 16, 2, 1, 4)                            \
    X(6, 16, 3, kFlagsRefDefault, 8, 16, 2, 1, 16)  /* the reference's own training setup, main.py:55-67: 16 agents, 7x7 */
#endif

}  // namespace
