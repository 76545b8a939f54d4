// mapf_step.hip -- MI355X (gfx950 / CDNA4) vectorized step engine for the multi-agent grid env.
//
// Replaces the hot path of /root/reference/src/environments/reference_model_multi_agent.py ("MA-env"):
// step() :474-695, reset() :440-472 and the helpers they call.  C ABI: include/mapf_step.h.
//
// Design (see DESIGN.md for the long form):
//   * one wavefront (64 lanes) per workgroup; an env instance owns a GROUP of LPE lanes of the wave
//     (LPE = smallest power of two >= max(N,4)); lane a of a group is agent a.  At N = 64 this is
//     one wavefront per env; at N = 8 one wave steps 8 envs, so no lane idles in the agent phases.
//   * the sequential move rule (MA-env:502-526, lower index wins) is resolved with a ballot per
//     contested agent index: lane i broadcasts its target, every lane compares its CURRENT cell
//     (already-moved lower indices hold their new cell, higher indices their old one).
//   * the observation of agent i is taken "at time i" (MA-env:528 sits inside the move loop):
//     occupant of a cell = new position of agents <= i, old position of agents > i.  Each lane builds
//     its V x V window as bit masks (obstacle / other agent / own goal / other goal) from the env's
//     bit-packed obstacle rows staged in LDS and N cross-lane broadcasts; no owner maps are needed.
//   * lock detector (MA-env:374-438): per-agent 64-step shift registers (moved / failed / progress)
//     replace the history ring; "sum over a participant set == 0" becomes a mask test against group
//     ballots; only the distance term needs a real sum.
//   * NumPy Generator(PCG64) (choice without replacement = Floyd + tail shuffle, integers = Lemire)
//     runs on device for reset() and lifelong goal respawn, so finished envs restart in-kernel.
//   * observations are assembled in LDS and leave the wave as one contiguous, 16-byte-vectorised
//     stream (the obs tensor is ~60% of the algorithmic bytes of a step).
// No MFMA: this is integer / indexing work.  No collective: envs are independent.
//
// Never compile with -ffast-math: goal_delta needs the correctly rounded fp32 divide (MA-env:332-334).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mapf_step.h"

namespace {

// ------------------------------------------------------------------------------------------------
// device-side data layout
// ------------------------------------------------------------------------------------------------
// Agent record, 32 B, array [B][N] (env-major: the N records of an env are contiguous, a wave reads
// 64 consecutive records = 2 KiB with two 16-byte loads per lane).
//   word0: pos (bits 0-15: row<<8 | col) | goal<<16
//   word1: start | flags<<16   (flags bit0 reached, bit1 completed_once, bit2 blocking_pressure_prev)
//   word2-3: moved history   (bit k = flag k steps ago)
//   word4-5: failed-move history
//   word6-7: goal-progress history
struct AgentRec {
    uint32_t w0, w1;
    uint64_t moved, failed, progress;
};
static_assert(sizeof(AgentRec) == 32, "AgentRec must be 32 bytes");

constexpr int kFlagReached = 1, kFlagCompleted = 2, kFlagPressure = 4;
constexpr int kScalInts = MAPF_NUM_COUNTERS;  // 16 int32 = 64 B per env

struct Params {
    int B, H, W, N, sr, V, L, steps_per_episode;
    uint32_t flags;
    int dw, lw, nearby, min_nbrs, eps_floor, hs;
    float den_r, den_c;
    int HW;
    int hash_cap;     // Floyd hash-set size (power of two), numpy: 1 + gen_mask(uint64(1.2 * 2N))
    int scratch_i16;  // int16 entries of reset scratch per group (hash_cap + 2N, rounded up to even)
    int lds_stage_off, lds_scratch_off;  // byte offsets into dynamic LDS (rows start at 0)
    // state
    AgentRec *agents;
    int *scal;
    int16_t *dist_ring;
    uint64_t *rng;
    const uint64_t *grid_rows;   // [B][H], bit c = obstacle, bits >= W set
    const uint16_t *free_cells;  // [B][HW], k-th free cell (row-major) as row<<8|col
    const uint16_t *free_rank;   // [B][HW], row-major rank of a free cell among free cells
    const int *n_free;           // [B]
    int *err;                    // [4] code, env, agent, value
    // io
    const int8_t *actions;
    float *obs, *rewards;
    uint8_t *terminated, *truncated;
    float *info_all;
    uint8_t *info_agent;
    float *final_obs;
    const uint8_t *env_mask;
    int auto_reset;
};

// ------------------------------------------------------------------------------------------------
// group (sub-wave) primitives: LPE consecutive lanes = one env
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

// OR the per-group bit sets of a wave ballot together (bit i = "some group has agent i set")
template <int LPE>
__device__ __forceinline__ uint64_t fold_groups(uint64_t m) {
    if (LPE <= 32) m |= m >> 32;
    if (LPE <= 16) m |= m >> 16;
    if (LPE <= 8) m |= m >> 8;
    if (LPE <= 4) m |= m >> 4;
    return m & group_mask<LPE>();
}

// broadcast lane j of every group (j must be wave-uniform)
template <int LPE>
__device__ __forceinline__ uint32_t gshfl(uint32_t v, int j) {
    if (LPE == 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, j);
    return (uint32_t)__shfl((int)v, j, LPE);
}

// ------------------------------------------------------------------------------------------------
// V x V window bit masks (bit dr*V+dc)
// ------------------------------------------------------------------------------------------------
template <bool WIDE>
struct WMask;
template <>
struct WMask<false> {  // V*V <= 49 (sensor_range <= 3)
    uint64_t lo;
    __device__ __forceinline__ void clear() { lo = 0; }
    __device__ __forceinline__ void set(int b) { lo |= 1ull << b; }
    __device__ __forceinline__ bool get(int b) const { return (lo >> b) & 1ull; }
    __device__ __forceinline__ void or_row(uint64_t w, int shift) { lo |= w << shift; }
};
template <>
struct WMask<true> {  // V*V <= 121 (sensor_range <= 5)
    uint64_t lo, hi;
    __device__ __forceinline__ void clear() { lo = hi = 0; }
    __device__ __forceinline__ void set(int b) {
        if (b < 64) lo |= 1ull << b; else hi |= 1ull << (b - 64);
    }
    __device__ __forceinline__ bool get(int b) const { return b < 64 ? ((lo >> b) & 1ull) : ((hi >> (b - 64)) & 1ull); }
    __device__ __forceinline__ void or_row(uint64_t w, int shift) {
        if (shift < 64) {
            lo |= w << shift;
            if (shift > 0) hi |= w >> (64 - shift);
        } else {
            hi |= w << (shift - 64);
        }
    }
};

// V bits of an obstacle row starting at column c0 (may be negative / run past 63); outside = 1
__device__ __forceinline__ uint64_t row_window(uint64_t ext, int c0, int V) {
    uint64_t w;
    if (c0 >= 0) {
        w = ext >> c0;
        if (c0 > 0) w |= ~0ull << (64 - c0);
    } else {
        w = (ext << (-c0)) | ((1ull << (-c0)) - 1ull);
    }
    return w & ((1ull << V) - 1ull);
}

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
// random_bounded_uint64(off=0, rng, use_masked=false) for rng < 2^32-1: Lemire with rejection
__device__ __forceinline__ uint32_t pcg_bounded(Pcg &g, uint32_t rng) {
    if (rng == 0) return 0;  // no draw
    const uint32_t excl = rng + 1u;
    uint64_t m = (uint64_t)pcg_next32(g) * excl;
    uint32_t left = (uint32_t)m;
    if (left < excl) {
        const uint32_t thr = (0xFFFFFFFFu - rng) % excl;
        // rejection probability per draw is thr / 2^32 < 1e-6 here; the cap only guards against a hang
        for (int guard = 0; left < thr && guard < 4096; guard++) {
            m = (uint64_t)pcg_next32(g) * excl;
            left = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}

// ------------------------------------------------------------------------------------------------
// observation of every agent lane -> LDS staging row (MA-env:707-747 get_obs, :749-773 mask,
// :306-335 flatten).  A = old_pos | new_pos<<16 of this lane; goal = this lane's goal.
// final_state: all agents at their new cell (reset, or after a lifelong respawn MA-env:565-575);
// otherwise agent i sees agents <= i at their new cell, > i at their old one (MA-env:528).
// ------------------------------------------------------------------------------------------------
template <int LPE, bool WIDE>
__device__ __forceinline__ void observe(const Params &p, const uint64_t *lrows, float *srow, bool is_agent, int a,
                                        uint32_t A, uint32_t goal, bool final_state, bool pressure) {
    constexpr int MAXV = WIDE ? 11 : 7;
    const int V = p.V, sr = p.sr;
    const uint32_t cur = A >> 16;
    const int myr = (int)(cur >> 8), myc = (int)(cur & 255u);
    const int r0 = myr - sr, c0 = myc - sr;

    WMask<WIDE> obst, agm, own, oth;
    obst.clear(); agm.clear(); own.clear(); oth.clear();

    uint64_t rows[MAXV];
#pragma unroll
    for (int d = 0; d < MAXV; d++) {
        int r = r0 + d;
        bool in = (d < V) && r >= 0 && r < p.H && is_agent;
        int rr = in ? r : 0;
        uint64_t v = lrows[rr];
        rows[d] = in ? v : ~0ull;
    }
#pragma unroll
    for (int d = 0; d < MAXV; d++) {
        if (d < V) obst.or_row(row_window(rows[d], c0, V), d * V);
    }

    for (int j = 0; j < p.N; j++) {
        uint32_t Aj = gshfl<LPE>(A, j);
        uint32_t Gj = gshfl<LPE>(goal, j);
        uint32_t pj = (final_state || j <= a) ? (Aj >> 16) : (Aj & 0xFFFFu);
        int pr = (int)(pj >> 8) - r0, pc = (int)(pj & 255u) - c0;
        if ((unsigned)pr < (unsigned)V && (unsigned)pc < (unsigned)V && j != a) agm.set(pr * V + pc);
        int gr = (int)((Gj >> 8) & 255u) - r0, gc = (int)(Gj & 255u) - c0;
        if ((unsigned)gr < (unsigned)V && (unsigned)gc < (unsigned)V) {
            if (j == a) own.set(gr * V + gc); else oth.set(gr * V + gc);
        }
    }

    if (!is_agent) return;
    const int VV = V * V;
    for (int t = 0; t < VV; t++) {
        // priority: obstacle/out-of-bounds 1 > other agent 2 > own goal 3 > other goal 4 > empty 0
        int code = obst.get(t) ? 1 : (agm.get(t) ? 2 : (own.get(t) ? 3 : (oth.get(t) ? 4 : 0)));
        srow[t] = (float)code;
    }
    float *q = srow + VV;
    float gd_r = (float)((int)((goal >> 8) & 255u) - myr);
    float gd_c = (float)((int)(goal & 255u) - myc);
    if (p.flags & MAPF_FLAG_NORMALIZE_GOAL_DELTA) {
        gd_r = gd_r / p.den_r;
        gd_c = gd_c / p.den_c;
    }
    *q++ = gd_r;
    *q++ = gd_c;
    if (p.flags & MAPF_FLAG_GOAL_DISTANCE) *q++ = fabsf(gd_r) + fabsf(gd_c);
    if (p.flags & MAPF_FLAG_BLOCKING_PRESSURE) *q++ = pressure ? 1.0f : 0.0f;
    if (p.flags & MAPF_FLAG_ACTION_MASK) {
        const int ctr = sr * V + sr;
        bool up = false, rt = false, dn = false, lf = false;
        if (sr > 0) {
            up = !(obst.get(ctr - V) || agm.get(ctr - V));
            rt = !(obst.get(ctr + 1) || agm.get(ctr + 1));
            dn = !(obst.get(ctr + V) || agm.get(ctr + V));
            lf = !(obst.get(ctr - 1) || agm.get(ctr - 1));
        }
        q[0] = 1.0f;
        q[1] = up ? 1.0f : 0.0f;
        q[2] = rt ? 1.0f : 0.0f;
        q[3] = dn ? 1.0f : 0.0f;
        q[4] = lf ? 1.0f : 0.0f;
    }
}

// copy the wave's staged observations to global memory.  sel (per lane, uniform inside a group):
// 0 -> p.obs, 1 -> p.final_obs, 2 -> skip.  Flat 16-byte stores when every valid group goes to
// the same tensor, otherwise one contiguous run per group.
template <int LPE>
__device__ __forceinline__ void flush_obs(const Params &p, const float *stage, int lane, int env0, int ngroups, int sel) {
    constexpr int G = 64 / LPE;
    const int NL = p.N * p.L;
    const uint64_t valid = __ballot((lane / LPE) < ngroups);
    const uint64_t m0 = __ballot((lane / LPE) < ngroups && sel == 0);
    const uint64_t m1 = __ballot((lane / LPE) < ngroups && sel == 1);
    float *flat = nullptr;
    if (m0 == valid) flat = p.obs;
    else if (m1 == valid) flat = p.final_obs;
    else if ((m0 | m1) == 0) return;
    if (m0 == valid || m1 == valid) {
        if (!flat) return;
        const int n = ngroups * NL;
        float *dst = flat + (size_t)env0 * NL;
        if (((G * NL) & 3) == 0) {
            const int n4 = n >> 2;
            const float4 *s4 = reinterpret_cast<const float4 *>(stage);
            float4 *d4 = reinterpret_cast<float4 *>(dst);
            for (int k = lane; k < n4; k += 64) d4[k] = s4[k];
            for (int k = (n4 << 2) + lane; k < n; k += 64) dst[k] = stage[k];
        } else {
            for (int k = lane; k < n; k += 64) dst[k] = stage[k];
        }
        return;
    }
    for (int g = 0; g < ngroups; g++) {
        const int sg = __shfl(sel, g * LPE, 64);
        float *base = sg == 0 ? p.obs : (sg == 1 ? p.final_obs : nullptr);
        if (!base) continue;
        float *dst = base + (size_t)(env0 + g) * NL;
        const float *src = stage + g * NL;
        for (int k = lane; k < NL; k += 64) dst[k] = src[k];
    }
}

__device__ __forceinline__ void raise_error(const Params &p, int code, int env, int agent, int value) {
    if (atomicCAS(&p.err[0], 0, code) == 0) {
        p.err[1] = env;
        p.err[2] = agent;
        p.err[3] = value;
    }
}

// per-lane register image of an env group
struct Lane {
    uint32_t pos, goal, start;  // row<<8|col
    uint32_t flags;
    uint64_t moved, failed, progress;
};

// ------------------------------------------------------------------------------------------------
// reset() of the groups with do_reset set (MA-env:440-472).  Group-uniform inputs; called under a
// wave-uniform branch.  Updates lane state + scalars; stages the reset observation when want_obs.
// ------------------------------------------------------------------------------------------------
template <int LPE, bool WIDE>
__device__ __forceinline__ void reset_groups(const Params &p, const uint64_t *lrows, float *stage, int16_t *scratch,
                                             int lane, int a, int grp, int env, bool env_ok, bool is_agent, bool do_reset,
                                             Lane &st, int *sc, bool want_obs) {
    if (!(p.flags & MAPF_FLAG_DETERMINISTIC)) {
        // generate_starts_goals MA-env:267-282: idx = rng.choice(F, 2N, replace=False)
        int16_t *hs = scratch + grp * p.scratch_i16;
        int16_t *out = hs + p.hash_cap;
        if (do_reset && a == 0) {
            Pcg g;
            pcg_load(g, p.rng + (size_t)env * 6);
            const int mask = p.hash_cap - 1, size = 2 * p.N, pop = p.n_free[env];
            for (int k = 0; k < p.hash_cap; k++) hs[k] = -1;
            for (int j = pop - size; j < pop; j++) {  // Floyd
                int val = (int)pcg_bounded(g, (uint32_t)j);
                int loc = val & mask;
                // the set holds at most 2N < hash_cap entries, so an empty slot always exists; the
                // probe counters only make termination structural
                for (int pr = 0; hs[loc] != -1 && hs[loc] != val && pr < p.hash_cap; pr++) loc = (loc + 1) & mask;
                if (hs[loc] == -1) {
                    hs[loc] = (int16_t)val;
                    out[j - pop + size] = (int16_t)val;
                } else {
                    loc = j & mask;
                    for (int pr = 0; hs[loc] != -1 && pr < p.hash_cap; pr++) loc = (loc + 1) & mask;
                    hs[loc] = (int16_t)j;
                    out[j - pop + size] = (int16_t)j;
                }
            }
            for (int i = size - 1; i >= 1; i--) {  // _shuffle_int tail shuffle
                int j = (int)pcg_bounded(g, (uint32_t)i);
                int16_t t = out[j];
                out[j] = out[i];
                out[i] = t;
            }
            if (env_ok) pcg_store(g, p.rng + (size_t)env * 6);
        }
        __syncthreads();
        if (do_reset && is_agent) {
            const uint16_t *fc = p.free_cells + (size_t)env * p.HW;
            st.start = fc[out[a]];
            st.goal = fc[out[p.N + a]];
        }
        __syncthreads();
    }
    if (do_reset) {
        st.pos = st.start;  // MA-env:279 / :453
        st.flags = 0;       // _reached_arr, _completed_once_arr, _blocking_pressure_prev_arr MA-env:447-449
        st.moved = st.failed = st.progress = 0;  // _reset_lock_tracking MA-env:360-372
        sc[MAPF_CTR_STEP_COUNT] = 0;
        sc[MAPF_CTR_HIST_ROWS] = 0;
        sc[MAPF_CTR_BLOCKING_COUNT] = 0;
        sc[MAPF_CTR_GOALS_REACHED_TOTAL] = 0;
        sc[MAPF_CTR_DEADLOCK_EVENTS] = 0;
        sc[MAPF_CTR_LIVELOCK_EVENTS] = 0;
        sc[MAPF_CTR_DEADLOCK_STEPS] = 0;
        sc[MAPF_CTR_LIVELOCK_STEPS] = 0;
        sc[MAPF_CTR_LOCK_STATE_PREV] = 0;
    }
    if (want_obs) {
        const uint32_t A = st.pos | (st.pos << 16);
        observe<LPE, WIDE>(p, lrows + grp * p.H, stage + (size_t)(grp * p.N + a) * p.L, is_agent && do_reset, a, A,
                           st.goal, true, false);
    }
}

template <int LPE>
__device__ __forceinline__ void load_rows_to_lds(const Params &p, uint64_t *lrows, int lane, int env0, int ngroups) {
    const int total = ngroups * p.H;
    const uint64_t *src = p.grid_rows + (size_t)env0 * p.H;
    for (int k0 = 0; k0 < total; k0 += 256) {
        uint64_t t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int idx = k0 + u * 64 + lane;
            t[u] = idx < total ? src[idx] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int idx = k0 + u * 64 + lane;
            if (idx < total) lrows[idx] = t[u];
        }
    }
}

__device__ __forceinline__ void load_lane(const Params &p, int env, int a, bool is_agent, Lane &st) {
    if (is_agent) {
        const uint4 *rp = reinterpret_cast<const uint4 *>(p.agents + (size_t)env * p.N + a);
        uint4 q0 = rp[0], q1 = rp[1];
        st.pos = q0.x & 0xFFFFu;
        st.goal = q0.x >> 16;
        st.start = q0.y & 0xFFFFu;
        st.flags = (q0.y >> 16) & 0xFFu;
        st.moved = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
        st.failed = (uint64_t)q1.x | ((uint64_t)q1.y << 32);
        st.progress = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
    } else {
        st.pos = 0xFFFEu;  // never equals a real cell or the 0xFFFF "no target" marker
        st.goal = 0xFFFDu;
        st.start = 0xFFFEu;
        st.flags = 0;
        st.moved = st.failed = st.progress = 0;
    }
}

__device__ __forceinline__ void store_lane(const Params &p, int env, int a, const Lane &st) {
    uint4 *rp = reinterpret_cast<uint4 *>(p.agents + (size_t)env * p.N + a);
    uint4 q0, q1;
    q0.x = (st.pos & 0xFFFFu) | (st.goal << 16);
    q0.y = (st.start & 0xFFFFu) | ((st.flags & 0xFFu) << 16);
    q0.z = (uint32_t)st.moved;
    q0.w = (uint32_t)(st.moved >> 32);
    q1.x = (uint32_t)st.failed;
    q1.y = (uint32_t)(st.failed >> 32);
    q1.z = (uint32_t)st.progress;
    q1.w = (uint32_t)(st.progress >> 32);
    rp[0] = q0;
    rp[1] = q1;
}

__device__ __forceinline__ void load_scal(const Params &p, int env, int *sc) {
    const int4 *sp = reinterpret_cast<const int4 *>(p.scal + (size_t)env * kScalInts);
    int4 s0 = sp[0], s1 = sp[1], s2 = sp[2];
    sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w;
    sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
    sc[8] = s2.x; sc[9] = s2.y; sc[10] = s2.z; sc[11] = s2.w;
}
__device__ __forceinline__ void store_scal(const Params &p, int env, const int *sc) {
    int4 *sp = reinterpret_cast<int4 *>(p.scal + (size_t)env * kScalInts);
    sp[0] = make_int4(sc[0], sc[1], sc[2], sc[3]);
    sp[1] = make_int4(sc[4], sc[5], sc[6], sc[7]);
    sp[2] = make_int4(sc[8], sc[9], sc[10], sc[11]);
}

// ------------------------------------------------------------------------------------------------
// reset kernel
// ------------------------------------------------------------------------------------------------
template <int LPE, bool WIDE>
__global__ __launch_bounds__(64) void k_reset(const Params p) {
    constexpr int G = 64 / LPE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint64_t *lrows = reinterpret_cast<uint64_t *>(lds_raw);
    float *stage = reinterpret_cast<float *>(lds_raw + p.lds_stage_off);
    int16_t *scratch = reinterpret_cast<int16_t *>(lds_raw + p.lds_scratch_off);

    const int lane = threadIdx.x, grp = lane / LPE, a = lane % LPE;
    const int env0 = blockIdx.x * G;
    const int ngroups = min(G, p.B - env0);
    const bool env_ok = grp < ngroups;
    const int env = env_ok ? env0 + grp : p.B - 1;
    const bool is_agent = env_ok && a < p.N;

    load_rows_to_lds<LPE>(p, lrows, lane, env0, ngroups);
    Lane st;
    load_lane(p, env, a, is_agent, st);
    int sc[12];
    load_scal(p, env, sc);
    const bool do_reset = env_ok && (p.env_mask == nullptr || p.env_mask[env] != 0);
    __syncthreads();

    reset_groups<LPE, WIDE>(p, lrows, stage, scratch, lane, a, grp, env, env_ok, is_agent, do_reset, st, sc,
                            p.obs != nullptr);
    __syncthreads();
    if (p.obs) flush_obs<LPE>(p, stage, lane, env0, ngroups, do_reset ? 0 : 2);
    if (do_reset) {
        if (is_agent) store_lane(p, env, a, st);
        if (a == 0) store_scal(p, env, sc);
    }
}

// ------------------------------------------------------------------------------------------------
// observe kernel: observation of every agent from the current (static) state, nothing is modified.
// What the reference computes when get_obs / _flatten_observation are called outside step()
// (its tests do: tests/test_reference_model_multi_agent_invariants.py:76-95).
// ------------------------------------------------------------------------------------------------
template <int LPE, bool WIDE>
__global__ __launch_bounds__(64) void k_observe(const Params p) {
    constexpr int G = 64 / LPE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint64_t *lrows = reinterpret_cast<uint64_t *>(lds_raw);
    float *stage = reinterpret_cast<float *>(lds_raw + p.lds_stage_off);
    const int lane = threadIdx.x, grp = lane / LPE, a = lane % LPE;
    const int env0 = blockIdx.x * G;
    const int ngroups = min(G, p.B - env0);
    const bool env_ok = grp < ngroups;
    const int env = env_ok ? env0 + grp : p.B - 1;
    const bool is_agent = env_ok && a < p.N;
    load_rows_to_lds<LPE>(p, lrows, lane, env0, ngroups);
    Lane st;
    load_lane(p, env, a, is_agent, st);
    __syncthreads();
    observe<LPE, WIDE>(p, lrows + grp * p.H, stage + (size_t)(grp * p.N + a) * p.L, is_agent, a, st.pos | (st.pos << 16),
                       st.goal, true, (st.flags & kFlagPressure) != 0);
    __syncthreads();
    flush_obs<LPE>(p, stage, lane, env0, ngroups, env_ok ? 0 : 2);
}

// ------------------------------------------------------------------------------------------------
// step kernel
// ------------------------------------------------------------------------------------------------
template <int LPE, bool WIDE>
__global__ __launch_bounds__(64) void k_step(const Params p) {
    constexpr int G = 64 / LPE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint64_t *lrows = reinterpret_cast<uint64_t *>(lds_raw);
    float *stage = reinterpret_cast<float *>(lds_raw + p.lds_stage_off);
    int16_t *scratch = reinterpret_cast<int16_t *>(lds_raw + p.lds_scratch_off);

    const int lane = threadIdx.x, grp = lane / LPE, a = lane % LPE;
    const int env0 = blockIdx.x * G;
    const int ngroups = min(G, p.B - env0);
    const bool env_ok = grp < ngroups;
    const int env = env_ok ? env0 + grp : p.B - 1;
    const bool is_agent = env_ok && a < p.N;
    const int N = p.N;
    const bool lifelong = (p.flags & MAPF_FLAG_LIFELONG) != 0;

    // ---- loads (all in flight together) -------------------------------------------------------
    load_rows_to_lds<LPE>(p, lrows, lane, env0, ngroups);
    Lane st;
    load_lane(p, env, a, is_agent, st);
    int sc[12];
    load_scal(p, env, sc);
    int act = is_agent ? (int)p.actions[(size_t)env * N + a] : 0;
    __syncthreads();
    const uint64_t *myrows = lrows + grp * p.H;

    // ---- invalid action: the reference raises mid-loop, after agents before the bad one were
    //      processed (MA-env:502-506); reproduce the partial mutation and latch the error ---------
    const bool bad = is_agent && (act < 0 || act > 4);
    const uint64_t badm = gballot<LPE>(bad, lane);
    const bool errored = badm != 0;
    const int n_live = errored ? (int)__builtin_ctzll(badm) : N;
    const bool live = is_agent && a < n_live;
    if (bad && a == n_live) raise_error(p, MAPF_ERR_BAD_ACTION, env, a, act);
    if (!live) act = 0;

    sc[MAPF_CTR_STEP_COUNT] += 1;  // MA-env:475

    // ---- move phase (MA-env:502-526) -----------------------------------------------------------
    const uint32_t old = st.pos;
    const int r_old = (int)(old >> 8), c_old = (int)(old & 255u);
    const int dr = (act == 1) ? -1 : ((act == 3) ? 1 : 0);
    const int dc = (act == 2) ? 1 : ((act == 4) ? -1 : 0);
    const int tr = r_old + dr, tc = c_old + dc;
    const bool inb = tr >= 0 && tr < p.H && tc >= 0 && tc < p.W;
    const uint64_t trow = (live && inb) ? myrows[tr] : ~0ull;
    const bool want = live && act != 0 && inb && !((trow >> tc) & 1ull);
    const uint32_t tgt = want ? (uint32_t)((tr << 8) | tc) : 0xFFFFu;
    uint32_t cur = old;
    {
        uint64_t u = fold_groups<LPE>(__ballot(want));
        while (u) {
            const int i = (int)__builtin_ctzll(u);
            u &= u - 1;
            const uint32_t ti = gshfl<LPE>(tgt, i);
            const bool occ = gballot<LPE>(cur == ti, lane) != 0;  // live occupancy: lower indices already moved
            if (a == i && ti != 0xFFFFu && !occ) cur = ti;
        }
    }
    const bool moved = cur != old;

    // ---- goal / reward logic (MA-env:538-563) --------------------------------------------------
    bool reached = (st.flags & kFlagReached) != 0;
    bool completed = (st.flags & kFlagCompleted) != 0;
    const bool pressure_prev = (st.flags & kFlagPressure) != 0;
    float reward = 0.0f;
    bool grs = false;                      // goal_reached_step flag
    bool on_goal = live && cur == st.goal;  // reached_goal[i], evaluated at agent i's own turn
    bool reassigned = false;                // group-uniform: any lifelong respawn this step
    if (!lifelong) {
        if (on_goal && !reached) {
            reached = true;
            completed = true;
            reward += 0.5f;
            grs = true;
        }
        sc[MAPF_CTR_GOALS_REACHED_TOTAL] += __popcll(gballot<LPE>(grs, lane));
    } else {
        const uint64_t arr_wave = __ballot(on_goal);
        if (arr_wave) {  // wave-uniform
            const uint64_t garr = gballot<LPE>(on_goal, lane);
            reassigned = garr != 0;
            Pcg g;
            pcg_load(g, p.rng + (size_t)env * 6);
            const int F = p.n_free[env];
            const uint16_t *frank = p.free_rank + (size_t)env * p.HW;
            uint64_t u = fold_groups<LPE>(arr_wave);
            while (u) {  // respawns happen in agent order, each sees the state "at time i" (MA-env:554)
                const int i = (int)__builtin_ctzll(u);
                u &= u - 1;
                const bool gact = (garr >> i) & 1ull;
                // occupied cells at time i; goals of everybody else (own old goal is released first, MA-env:286-288)
                const uint32_t P = is_agent ? ((a <= i) ? cur : old) : 0xFFFEu;
                const bool Gact = is_agent && a != i;
                const int rankP = is_agent ? (int)frank[(P >> 8) * p.W + (P & 255u)] : 0x7FFFFFFF;
                const int rankG = Gact ? (int)frank[(st.goal >> 8) * p.W + (st.goal & 255u)] : 0x7FFFFFFF;
                bool dup = false;  // my goal cell is also occupied -> count it once
                for (int l = 0; l < N; l++) dup |= (gshfl<LPE>(P, l) == st.goal);
                dup = dup && Gact;
                const int overlap = __popcll(gballot<LPE>(dup, lane));
                const int k = F - N - (N - 1) + overlap;  // candidate_indices.size MA-env:295
                uint32_t r = 0;
                if (gact) {
                    if (k <= 0) {
                        if (a == i) raise_error(p, MAPF_ERR_NO_RESPAWN, env, i, k);
                    } else {
                        r = pcg_bounded(g, (uint32_t)(k - 1));  // rng.integers(k) MA-env:300
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
                if (gact && k > 0 && a == i) st.goal = p.free_cells[(size_t)env * p.HW + y];  // MA-env:301-303
            }
            if (reassigned && a == 0 && env_ok) pcg_store(g, p.rng + (size_t)env * 6);
            if (on_goal) {  // MA-env:547-556
                reward += 0.5f;
                grs = true;
                completed = true;
                reached = false;
                on_goal = false;  // reached_goal[i] = False after the respawn
            }
            sc[MAPF_CTR_GOALS_REACHED_TOTAL] += __popcll(garr);
        }
    }

    // ---- everything below is skipped by the reference when the ValueError fired -----------------
    const uint32_t A = old | (cur << 16);
    int term = 0, trunc = 0;
    float blocking_f = 0.0f;
    // groups whose step raised keep only the mutations made before the exception: snapshot what the
    // (wave-wide) tail below would otherwise touch
    int sc_keep[12];
#pragma unroll
    for (int k = 0; k < 12; k++) sc_keep[k] = sc[k];
    const uint64_t h_moved = st.moved, h_failed = st.failed, h_progress = st.progress;
    if (!__all(errored || !env_ok)) {
        // observations (MA-env:528-534 staggered, or :565-575 all-final after a respawn)
        if (p.obs || p.final_obs)
            observe<LPE, WIDE>(p, myrows, stage + (size_t)(grp * N + a) * p.L, is_agent, a, A, st.goal, reassigned,
                               pressure_prev);

        // lock flags (MA-env:581-594)
        const bool lock_on = (p.flags & MAPF_FLAG_LOCK_METRICS) != 0;
        const int gr_ = (int)((st.goal >> 8) & 255u), gc_ = (int)(st.goal & 255u);
        const int r_new = (int)(cur >> 8), c_new = (int)(cur & 255u);
        const bool cur_on_goal = is_agent && cur == st.goal;
        const bool prev_on_goal = !lifelong && old == st.goal;
        const bool progress = lifelong ? grs : (!prev_on_goal && cur_on_goal);
        const bool failed = act != 0 && !moved;
        const int dist = abs(gr_ - r_new) + abs(gc_ - c_new);
        int delta = 0;
        bool dl_ok = false, ll_ok = false;
        if (lock_on) {
            const int t = sc[MAPF_CTR_HIST_ROWS];
            const int count = min(t + 1, p.hs);
            dl_ok = count >= p.dw;
            ll_ok = count >= p.lw;
            st.moved = (st.moved << 1) | (moved ? 1ull : 0ull);  // _append_lock_history_step MA-env:374-387
            st.failed = (st.failed << 1) | (failed ? 1ull : 0ull);
            st.progress = (st.progress << 1) | (progress ? 1ull : 0ull);
            if (is_agent && !errored) {
                int16_t *ring = p.dist_ring + (size_t)env * p.lw * N;
                int d_old = dist;
                if (p.lw > 1 && ll_ok) d_old = ring[((t + 1) % p.lw) * N + a];  // oldest row of the window
                ring[(t % p.lw) * N + a] = (int16_t)dist;
                delta = d_old - dist;
            }
            sc[MAPF_CTR_HIST_ROWS] = t + 1;
        }

        // one pass over the other agents: neighbour sets (MA-env:389-398), intent blocking (:608-623),
        // coincidence penalty (:658-666)
        const uint32_t Bw = (uint32_t)act | ((reached ? 1u : 0u) << 3) | ((uint32_t)(delta + 256) << 4);
        uint64_t nbr = 0;
        int sum_delta = delta;
        bool blocks = false;
        for (int j = 0; j < N; j++) {
            const uint32_t Aj = gshfl<LPE>(A, j);
            const uint32_t Bj = gshfl<LPE>(Bw, j);
            const uint32_t oldj = Aj & 0xFFFFu, newj = Aj >> 16;
            const int actj = (int)(Bj & 7u);
            const bool reachedj = (Bj >> 3) & 1u;
            const int dj = (int)((Bj >> 4) & 1023u) - 256;
            const int d = abs((int)(newj >> 8) - r_new) + abs((int)(newj & 255u) - c_new);
            if (d > 0 && d <= p.nearby) {
                nbr |= 1ull << j;
                sum_delta += dj;
            }
            if (j != a) {
                const int ir = (int)(oldj >> 8) + ((actj == 1) ? -1 : ((actj == 3) ? 1 : 0));
                const int ic = (int)(oldj & 255u) + ((actj == 2) ? 1 : ((actj == 4) ? -1 : 0));
                if (!reachedj && ir == r_new && ic == c_new) blocks = true;
                if (newj == cur) reward -= 1.0f;  // unreachable by invariant; kept like the reference
            }
        }

        // lock detector (MA-env:400-438): deadlock has priority over livelock
        int deadlock = 0, livelock = 0, dl_event = 0, ll_event = 0;
        if (lock_on) {
            const uint64_t mdw = p.dw >= 64 ? ~0ull : ((1ull << p.dw) - 1ull);
            const uint64_t mlw = p.lw >= 64 ? ~0ull : ((1ull << p.lw) - 1ull);
            const uint64_t members = nbr | (1ull << a);
            const bool focal = is_agent && !cur_on_goal && __popcll(nbr) >= p.min_nbrs;
            const uint64_t gm = group_mask<LPE>() & ((N >= 64) ? ~0ull : ((1ull << N) - 1ull));
            const uint64_t prog_dw_nz = gballot<LPE>((st.progress & mdw) != 0, lane) & gm;
            const uint64_t moved_dw_nz = gballot<LPE>((st.moved & mdw) != 0, lane) & gm;
            const uint64_t fail_dw_nz = gballot<LPE>((st.failed & mdw) != 0, lane) & gm;
            const uint64_t prog_lw_nz = gballot<LPE>((st.progress & mlw) != 0, lane) & gm;
            const uint64_t moved_lw_nz = gballot<LPE>((st.moved & mlw) != 0, lane) & gm;
            const bool dead_me = focal && dl_ok && (members & prog_dw_nz) == 0 && (members & moved_dw_nz) == 0 &&
                                 (members & fail_dw_nz) != 0;
            const bool live_me = focal && ll_ok && (members & prog_lw_nz) == 0 && (members & moved_lw_nz) != 0 &&
                                 sum_delta <= p.eps_floor;
            deadlock = gballot<LPE>(dead_me, lane) != 0;
            livelock = !deadlock && gballot<LPE>(live_me, lane) != 0;
            const int prev = sc[MAPF_CTR_LOCK_STATE_PREV];
            dl_event = deadlock && !(prev & 1);  // rising edges MA-env:599-600
            ll_event = livelock && !(prev & 2);
            sc[MAPF_CTR_LOCK_STATE_PREV] = deadlock | (livelock << 1);
            sc[MAPF_CTR_DEADLOCK_STEPS] += deadlock;
            sc[MAPF_CTR_LIVELOCK_STEPS] += livelock;
            sc[MAPF_CTR_DEADLOCK_EVENTS] += dl_event;
            sc[MAPF_CTR_LIVELOCK_EVENTS] += ll_event;
        }

        // blocking flags feed NEXT step's observation (MA-env:608-625)
        const bool blocking = is_agent && reached && !moved && blocks;
        blocking_f = blocking ? 1.0f : 0.0f;
        const int blocking_step = __popcll(gballot<LPE>(blocking, lane));
        sc[MAPF_CTR_BLOCKING_COUNT] += blocking_step;

        // termination (MA-env:668-690): success check precedes the step-limit check
        const int n_on_goal = __popcll(gballot<LPE>(on_goal, lane));
        if (!lifelong && n_on_goal == N) {
            reward += 1.0f;
            term = 1;
        } else if (sc[MAPF_CTR_STEP_COUNT] >= p.steps_per_episode) {
            if (!lifelong && !on_goal) reward -= 1.0f;
            term = 1;
            trunc = 1;
        }

        // info (MA-env:627-656)
        const int goals_step = __popcll(gballot<LPE>(grs, lane));
        const int reached_cnt = __popcll(gballot<LPE>(is_agent && reached, lane));
        const int completed_cnt = __popcll(gballot<LPE>(is_agent && completed, lane));
        const int goals_total = lifelong ? sc[MAPF_CTR_GOALS_REACHED_TOTAL] : reached_cnt;
        if (p.info_all && env_ok && !errored) {
            const int steps = max(sc[MAPF_CTR_STEP_COUNT], 1);
            for (int k = a; k < MAPF_INFO_ALL; k += LPE) {
                float v;
                switch (k) {
                    case 0: v = (float)goals_step; break;
                    case 1: v = (float)goals_total; break;
                    case 2: v = (float)blocking_step; break;
                    case 3: v = (float)sc[MAPF_CTR_BLOCKING_COUNT]; break;
                    case 4: v = (float)deadlock; break;
                    case 5: v = (float)livelock; break;
                    case 6: v = (float)dl_event; break;
                    case 7: v = (float)ll_event; break;
                    case 8: v = (float)sc[MAPF_CTR_DEADLOCK_EVENTS]; break;
                    case 9: v = (float)sc[MAPF_CTR_LIVELOCK_EVENTS]; break;
                    case 10: v = (float)sc[MAPF_CTR_DEADLOCK_STEPS]; break;
                    case 11: v = (float)sc[MAPF_CTR_LIVELOCK_STEPS]; break;
                    case 12: v = (float)completed_cnt / (float)N; break;  // completion_ratio MA-env:638
                    default: v = (float)goals_total / (float)steps; break;  // throughput MA-env:655
                }
                p.info_all[(size_t)env * MAPF_INFO_ALL + k] = v;
            }
        }
        if (is_agent && !errored) {
            if (p.rewards) p.rewards[(size_t)env * N + a] = reward;
            if (p.info_agent) {
                uchar2 ia;
                ia.x = blocking ? 1 : 0;
                ia.y = grs ? 1 : 0;
                reinterpret_cast<uchar2 *>(p.info_agent)[(size_t)env * N + a] = ia;
            }
        }
        if (env_ok && !errored && a == 0) {
            if (p.terminated) p.terminated[env] = (uint8_t)term;
            if (p.truncated) p.truncated[env] = (uint8_t)trunc;
        }
    }

    // ---- state image after the step -----------------------------------------------------------
    st.pos = cur;
    if (errored) {
#pragma unroll
        for (int k = 0; k < 12; k++) sc[k] = sc_keep[k];
        st.moved = h_moved;
        st.failed = h_failed;
        st.progress = h_progress;
        // reference state after the exception: moves + goal logic of the agents before the bad one
        st.flags = (reached ? kFlagReached : 0) | (completed ? kFlagCompleted : 0) | (pressure_prev ? kFlagPressure : 0);
    } else {
        st.flags = (reached ? kFlagReached : 0) | (completed ? kFlagCompleted : 0) | (blocking_f != 0.0f ? kFlagPressure : 0);
    }

    // ---- observations out; auto-reset of finished envs (reference harness loop
    //      scripts/benchmark_multi_agent_env.py:89-95: reset() right after a done step) ----------
    const bool done = env_ok && !errored && (term | trunc);
    const bool do_reset = done && p.auto_reset;
    __syncthreads();
    if (p.obs || p.final_obs) {
        const int sel = (!env_ok || errored) ? 2 : (do_reset ? (p.final_obs ? 1 : 2) : (p.obs ? 0 : 2));
        flush_obs<LPE>(p, stage, lane, env0, ngroups, sel);
    }
    if (__any(do_reset)) {
        if (do_reset) sc[MAPF_CTR_EPISODES_DONE] += 1;
        __syncthreads();
        reset_groups<LPE, WIDE>(p, lrows, stage, scratch, lane, a, grp, env, env_ok, is_agent, do_reset, st, sc,
                                p.obs != nullptr);
        __syncthreads();
        if (p.obs) flush_obs<LPE>(p, stage, lane, env0, ngroups, do_reset ? 0 : 2);
    }
    if (is_agent) store_lane(p, env, a, st);
    if (env_ok && a == 0) store_scal(p, env, sc);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local std::string g_create_error;

}  // namespace

struct mapf_engine {
    mapf_config cfg;
    Params p;
    int lpe = 0;
    bool wide = false;
    int blocks = 0;
    int lds_bytes = 0;
    bool grids_set = false;
    std::string err;
    // device allocations
    AgentRec *d_agents = nullptr;
    int *d_scal = nullptr;
    int16_t *d_ring = nullptr;
    uint64_t *d_rng = nullptr;
    uint64_t *d_rows = nullptr;
    uint16_t *d_free_cells = nullptr;
    uint16_t *d_free_rank = nullptr;
    int *d_n_free = nullptr;
    int *d_err = nullptr;
};

namespace {

int fail(mapf_engine *e, int code, const std::string &msg) {
    if (e) e->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(e, call)                                                                                  \
    do {                                                                                                  \
        hipError_t _s = (call);                                                                           \
        if (_s != hipSuccess)                                                                             \
            return fail((e), MAPF_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_s));            \
    } while (0)

int pick_lpe(int n) {
    int l = 4;
    while (l < n) l <<= 1;
    return l;
}

template <int LPE, bool WIDE>
hipError_t launch_step_t(const mapf_engine *e, const Params &p, hipStream_t s) {
    hipLaunchKernelGGL((k_step<LPE, WIDE>), dim3(e->blocks), dim3(64), e->lds_bytes, s, p);
    return hipGetLastError();
}
template <int LPE, bool WIDE>
hipError_t launch_reset_t(const mapf_engine *e, const Params &p, hipStream_t s) {
    hipLaunchKernelGGL((k_reset<LPE, WIDE>), dim3(e->blocks), dim3(64), e->lds_bytes, s, p);
    return hipGetLastError();
}
template <int LPE, bool WIDE>
hipError_t launch_observe_t(const mapf_engine *e, const Params &p, hipStream_t s) {
    hipLaunchKernelGGL((k_observe<LPE, WIDE>), dim3(e->blocks), dim3(64), e->lds_bytes, s, p);
    return hipGetLastError();
}

enum { KIND_RESET = 0, KIND_STEP = 1, KIND_OBSERVE = 2 };

template <int LPE, bool WIDE>
hipError_t launch_kind(int kind, const mapf_engine *e, const Params &p, hipStream_t s) {
    if (kind == KIND_STEP) return launch_step_t<LPE, WIDE>(e, p, s);
    if (kind == KIND_RESET) return launch_reset_t<LPE, WIDE>(e, p, s);
    return launch_observe_t<LPE, WIDE>(e, p, s);
}

hipError_t dispatch(int kind, const mapf_engine *e, const Params &p, hipStream_t s) {
#define MAPF_CASE(L)                                              \
    case L:                                                       \
        if (e->wide) return launch_kind<L, true>(kind, e, p, s);  \
        return launch_kind<L, false>(kind, e, p, s);
    switch (e->lpe) {
        MAPF_CASE(4)
        MAPF_CASE(8)
        MAPF_CASE(16)
        MAPF_CASE(32)
        MAPF_CASE(64)
    }
#undef MAPF_CASE
    return hipErrorInvalidValue;
}

}  // namespace

static int alloc_device_state(mapf_engine *e);

extern "C" {

uint32_t mapf_version(void) { return (MAPF_VERSION_MAJOR << 16) | MAPF_VERSION_MINOR; }

int32_t mapf_obs_len(const mapf_config *cfg) {
    const int V = 2 * cfg->sensor_range + 1;
    int L = V * V + 2;
    if (cfg->flags & MAPF_FLAG_GOAL_DISTANCE) L += 1;
    if (cfg->flags & MAPF_FLAG_BLOCKING_PRESSURE) L += 1;
    if (cfg->flags & MAPF_FLAG_ACTION_MASK) L += 5;
    return L;
}

const char *mapf_last_error(mapf_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mapf_create(const mapf_config *cfg, mapf_handle *out) {
    if (!cfg || !out) return fail(nullptr, MAPF_ERR_CONFIG, "null argument");
    *out = nullptr;
    mapf_config c = *cfg;
    // clamps of the reference ctor, MA-env:56-60
    if (c.deadlock_window_steps < 1) c.deadlock_window_steps = 1;
    if (c.livelock_window_steps < 1) c.livelock_window_steps = 1;
    if (c.lock_nearby_manhattan < 1) c.lock_nearby_manhattan = 1;
    if (c.lock_min_neighbors < 1) c.lock_min_neighbors = 1;
    if (c.num_envs < 1) return fail(nullptr, MAPF_ERR_CONFIG, "num_envs must be >= 1");
    if (c.height < 1 || c.width < 1 || c.height > MAPF_MAX_DIM || c.width > MAPF_MAX_DIM)
        return fail(nullptr, MAPF_ERR_CONFIG, "grid height/width must be in [1, 64] in this build");
    if (c.num_agents < 1 || c.num_agents > MAPF_MAX_AGENTS)
        return fail(nullptr, MAPF_ERR_CONFIG, "num_agents must be in [1, 64] in this build");
    if (c.sensor_range < 0 || c.sensor_range > MAPF_MAX_SENSOR_RANGE)
        return fail(nullptr, MAPF_ERR_CONFIG, "sensor_range must be in [0, 5] in this build");
    if (c.deadlock_window_steps > MAPF_MAX_LOCK_WINDOW || c.livelock_window_steps > MAPF_MAX_LOCK_WINDOW)
        return fail(nullptr, MAPF_ERR_CONFIG, "lock windows must be <= 64 steps in this build");
    int lpe = c.lanes_per_env ? c.lanes_per_env : pick_lpe(c.num_agents);
    if (!(lpe == 4 || lpe == 8 || lpe == 16 || lpe == 32 || lpe == 64) || lpe < c.num_agents)
        return fail(nullptr, MAPF_ERR_CONFIG, "lanes_per_env must be a power of two in [4,64] and >= num_agents");

    mapf_engine *e = new mapf_engine();
    e->cfg = c;
    e->lpe = lpe;
    e->wide = c.sensor_range > 3;
    const int G = 64 / lpe;
    const int B = c.num_envs, N = c.num_agents, H = c.height, W = c.width;
    e->blocks = (B + G - 1) / G;

    Params &p = e->p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.N = N;
    p.sr = c.sensor_range;
    p.V = 2 * c.sensor_range + 1;
    p.L = mapf_obs_len(&c);
    p.steps_per_episode = c.steps_per_episode;
    p.flags = c.flags;
    p.dw = c.deadlock_window_steps;
    p.lw = c.livelock_window_steps;
    p.nearby = c.lock_nearby_manhattan;
    p.min_nbrs = c.lock_min_neighbors;
    p.hs = p.dw > p.lw ? p.dw : p.lw;
    {   // integer distance_reduction <= eps  <=>  reduction <= floor(eps)   (MA-env:435)
        double eps = c.lock_progress_epsilon;
        if (std::isnan(eps)) p.eps_floor = INT32_MIN;
        else if (eps >= 1e9) p.eps_floor = 1000000000;
        else if (eps <= -1e9) p.eps_floor = -1000000000;
        else p.eps_floor = (int)std::floor(eps);
    }
    p.den_r = (float)(H - 1 > 1 ? H - 1 : 1);  // _goal_delta_denominator MA-env:152-155
    p.den_c = (float)(W - 1 > 1 ? W - 1 : 1);
    p.HW = H * W;
    {   // numpy Generator.choice Floyd hash set: set_size = 1 + gen_mask(uint64(1.2 * size))
        uint64_t m = (uint64_t)(1.2 * (double)(2 * N));
        m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16; m |= m >> 32;
        p.hash_cap = (int)(m + 1);
    }
    p.scratch_i16 = (p.hash_cap + 2 * N + 1) & ~1;
    const int rows_bytes = ((G * H * 8) + 15) & ~15;
    const int stage_bytes = ((G * N * p.L * 4) + 15) & ~15;
    const int scratch_bytes = ((G * p.scratch_i16 * 2) + 15) & ~15;
    p.lds_stage_off = rows_bytes;
    p.lds_scratch_off = rows_bytes + stage_bytes;
    e->lds_bytes = rows_bytes + stage_bytes + scratch_bytes;
    if (e->lds_bytes > 64 * 1024) {
        delete e;
        return fail(nullptr, MAPF_ERR_CONFIG, "config needs more than 64 KiB of LDS per wavefront");
    }

    const int rc = alloc_device_state(e);
    if (rc != MAPF_OK) {
        g_create_error = e->err;
        mapf_destroy(e);
        return rc;
    }
    *out = e;
    return MAPF_OK;
}

static int alloc_device_state(mapf_engine *e) {
    const mapf_config &c = e->cfg;
    Params &p = e->p;
    const int B = p.B, N = p.N, H = p.H;
    HIP_TRY(e, hipSetDevice(c.device));
    const size_t BN = (size_t)B * N;
    HIP_TRY(e, hipMalloc(&e->d_agents, BN * sizeof(AgentRec)));
    HIP_TRY(e, hipMalloc(&e->d_scal, (size_t)B * kScalInts * sizeof(int)));
    HIP_TRY(e, hipMalloc(&e->d_ring, BN * p.lw * sizeof(int16_t)));
    HIP_TRY(e, hipMalloc(&e->d_rng, (size_t)B * 6 * sizeof(uint64_t)));
    HIP_TRY(e, hipMalloc(&e->d_rows, (size_t)B * H * sizeof(uint64_t)));
    HIP_TRY(e, hipMalloc(&e->d_free_cells, (size_t)B * p.HW * sizeof(uint16_t)));
    HIP_TRY(e, hipMalloc(&e->d_free_rank, (size_t)B * p.HW * sizeof(uint16_t)));
    HIP_TRY(e, hipMalloc(&e->d_n_free, (size_t)B * sizeof(int)));
    HIP_TRY(e, hipMalloc(&e->d_err, 4 * sizeof(int)));
    HIP_TRY(e, hipMemset(e->d_agents, 0, BN * sizeof(AgentRec)));
    HIP_TRY(e, hipMemset(e->d_scal, 0, (size_t)B * kScalInts * sizeof(int)));
    HIP_TRY(e, hipMemset(e->d_ring, 0, BN * p.lw * sizeof(int16_t)));
    HIP_TRY(e, hipMemset(e->d_rng, 0, (size_t)B * 6 * sizeof(uint64_t)));
    HIP_TRY(e, hipMemset(e->d_err, 0, 4 * sizeof(int)));
    p.agents = e->d_agents;
    p.scal = e->d_scal;
    p.dist_ring = e->d_ring;
    p.rng = e->d_rng;
    p.grid_rows = e->d_rows;
    p.free_cells = e->d_free_cells;
    p.free_rank = e->d_free_rank;
    p.n_free = e->d_n_free;
    p.err = e->d_err;
    return MAPF_OK;
}

int mapf_destroy(mapf_handle e) {
    if (!e) return MAPF_OK;
    hipSetDevice(e->cfg.device);
    hipFree(e->d_agents);
    hipFree(e->d_scal);
    hipFree(e->d_ring);
    hipFree(e->d_rng);
    hipFree(e->d_rows);
    hipFree(e->d_free_cells);
    hipFree(e->d_free_rank);
    hipFree(e->d_n_free);
    hipFree(e->d_err);
    delete e;
    return MAPF_OK;
}

int mapf_set_grids(mapf_handle e, const uint8_t *grids, int32_t shared) {
    if (!e || !grids) return fail(e, MAPF_ERR_CONFIG, "null argument");
    const int B = e->p.B, H = e->p.H, W = e->p.W, HW = e->p.HW, N = e->p.N;
    std::vector<uint64_t> rows((size_t)B * H);
    std::vector<uint16_t> cells((size_t)B * HW, 0), rank((size_t)B * HW, 0);
    std::vector<int> nfree(B);
    const uint64_t hi = W >= 64 ? 0ull : (~0ull << W);  // columns >= W read as obstacle
    for (int b = 0; b < B; b++) {
        const uint8_t *g = grids + (shared ? 0 : (size_t)b * HW);
        int f = 0;
        for (int r = 0; r < H; r++) {
            uint64_t bits = hi;
            for (int c = 0; c < W; c++) {
                if (g[r * W + c] != 0) {
                    bits |= 1ull << c;
                } else {  // _free_positions = argwhere(grid == 0), row-major (MA-env:82)
                    cells[(size_t)b * HW + f] = (uint16_t)((r << 8) | c);
                    rank[(size_t)b * HW + r * W + c] = (uint16_t)f;
                    f++;
                }
            }
            rows[(size_t)b * H + r] = bits;
        }
        nfree[b] = f;
        if (f < 2 * N) {
            char buf[160];
            snprintf(buf, sizeof buf, "Environment has only %d free cells, but %d are required for starts and goals.", f, 2 * N);
            return fail(e, MAPF_ERR_FEW_FREE, buf);
        }
    }
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipMemcpy(e->d_rows, rows.data(), rows.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->d_free_cells, cells.data(), cells.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->d_free_rank, rank.data(), rank.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->d_n_free, nfree.data(), nfree.size() * sizeof(int), hipMemcpyHostToDevice));
    e->grids_set = true;
    return MAPF_OK;
}

int mapf_set_rng_state(mapf_handle e, const uint64_t *rng_words) {
    if (!e || !rng_words) return fail(e, MAPF_ERR_CONFIG, "null argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipMemcpy(e->d_rng, rng_words, (size_t)e->p.B * 6 * sizeof(uint64_t), hipMemcpyHostToDevice));
    return MAPF_OK;
}

int mapf_get_state(mapf_handle e, mapf_state *out) {
    if (!e || !out) return fail(e, MAPF_ERR_CONFIG, "null argument");
    const int B = e->p.B, N = e->p.N;
    const size_t BN = (size_t)B * N;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipDeviceSynchronize());
    std::vector<AgentRec> recs(BN);
    HIP_TRY(e, hipMemcpy(recs.data(), e->d_agents, BN * sizeof(AgentRec), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < BN; i++) {
        const AgentRec &r = recs[i];
        const uint32_t pos = r.w0 & 0xFFFFu, goal = r.w0 >> 16, start = r.w1 & 0xFFFFu, fl = (r.w1 >> 16) & 0xFFu;
        if (out->positions) { out->positions[2 * i] = (int16_t)(pos >> 8); out->positions[2 * i + 1] = (int16_t)(pos & 255u); }
        if (out->goals) { out->goals[2 * i] = (int16_t)(goal >> 8); out->goals[2 * i + 1] = (int16_t)(goal & 255u); }
        if (out->starts) { out->starts[2 * i] = (int16_t)(start >> 8); out->starts[2 * i + 1] = (int16_t)(start & 255u); }
        if (out->reached) out->reached[i] = (fl & kFlagReached) ? 1 : 0;
        if (out->completed_once) out->completed_once[i] = (fl & kFlagCompleted) ? 1 : 0;
        if (out->pressure_prev) out->pressure_prev[i] = (fl & kFlagPressure) ? 1 : 0;
        if (out->lock_history) {
            out->lock_history[3 * i] = r.moved;
            out->lock_history[3 * i + 1] = r.failed;
            out->lock_history[3 * i + 2] = r.progress;
        }
    }
    if (out->counters)
        HIP_TRY(e, hipMemcpy(out->counters, e->d_scal, (size_t)B * kScalInts * sizeof(int), hipMemcpyDeviceToHost));
    if (out->rng_words)
        HIP_TRY(e, hipMemcpy(out->rng_words, e->d_rng, (size_t)B * 6 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (out->distance_ring)
        HIP_TRY(e, hipMemcpy(out->distance_ring, e->d_ring, BN * e->p.lw * sizeof(int16_t), hipMemcpyDeviceToHost));
    return MAPF_OK;
}

int mapf_set_state(mapf_handle e, const mapf_state *in) {
    if (!e || !in) return fail(e, MAPF_ERR_CONFIG, "null argument");
    const int B = e->p.B, N = e->p.N, H = e->p.H, W = e->p.W;
    const size_t BN = (size_t)B * N;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipDeviceSynchronize());
    const bool touch_recs = in->positions || in->goals || in->starts || in->reached || in->completed_once ||
                            in->pressure_prev || in->lock_history;
    if (touch_recs) {
        std::vector<AgentRec> recs(BN);
        HIP_TRY(e, hipMemcpy(recs.data(), e->d_agents, BN * sizeof(AgentRec), hipMemcpyDeviceToHost));
        auto pack = [&](const int16_t *v, size_t i, uint32_t &dst) -> bool {
            int r = v[2 * i], c = v[2 * i + 1];
            if (r < 0 || r >= H || c < 0 || c >= W) return false;
            dst = (uint32_t)((r << 8) | c);
            return true;
        };
        for (size_t i = 0; i < BN; i++) {
            AgentRec &r = recs[i];
            uint32_t pos = r.w0 & 0xFFFFu, goal = r.w0 >> 16, start = r.w1 & 0xFFFFu, fl = (r.w1 >> 16) & 0xFFu;
            if (in->positions && !pack(in->positions, i, pos)) return fail(e, MAPF_ERR_CONFIG, "position outside the grid");
            if (in->goals && !pack(in->goals, i, goal)) return fail(e, MAPF_ERR_CONFIG, "goal outside the grid");
            if (in->starts && !pack(in->starts, i, start)) return fail(e, MAPF_ERR_CONFIG, "start outside the grid");
            if (in->reached) fl = (fl & ~kFlagReached) | (in->reached[i] ? kFlagReached : 0);
            if (in->completed_once) fl = (fl & ~kFlagCompleted) | (in->completed_once[i] ? kFlagCompleted : 0);
            if (in->pressure_prev) fl = (fl & ~kFlagPressure) | (in->pressure_prev[i] ? kFlagPressure : 0);
            r.w0 = pos | (goal << 16);
            r.w1 = start | (fl << 16);
            if (in->lock_history) {
                r.moved = in->lock_history[3 * i];
                r.failed = in->lock_history[3 * i + 1];
                r.progress = in->lock_history[3 * i + 2];
            }
        }
        HIP_TRY(e, hipMemcpy(e->d_agents, recs.data(), BN * sizeof(AgentRec), hipMemcpyHostToDevice));
    }
    if (in->counters)
        HIP_TRY(e, hipMemcpy(e->d_scal, in->counters, (size_t)B * kScalInts * sizeof(int), hipMemcpyHostToDevice));
    if (in->rng_words)
        HIP_TRY(e, hipMemcpy(e->d_rng, in->rng_words, (size_t)B * 6 * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (in->distance_ring)
        HIP_TRY(e, hipMemcpy(e->d_ring, in->distance_ring, BN * e->p.lw * sizeof(int16_t), hipMemcpyHostToDevice));
    return MAPF_OK;
}

int mapf_set_fixed_starts_goals(mapf_handle e, const int16_t *starts, const int16_t *goals) {
    if (!e || !starts || !goals) return fail(e, MAPF_ERR_CONFIG, "null argument");
    mapf_state s;
    memset(&s, 0, sizeof s);
    s.positions = const_cast<int16_t *>(starts);
    s.starts = const_cast<int16_t *>(starts);
    s.goals = const_cast<int16_t *>(goals);
    return mapf_set_state(e, &s);
}

int mapf_reset(mapf_handle e, const uint8_t *env_mask, float *obs, void *stream) {
    if (!e) return MAPF_ERR_CONFIG;
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_reset");
    Params p = e->p;
    p.env_mask = env_mask;
    p.obs = obs;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, dispatch(KIND_RESET, e, p, (hipStream_t)stream));
    return MAPF_OK;
}

int mapf_step(mapf_handle e, const int8_t *actions, float *obs, float *rewards, uint8_t *terminated, uint8_t *truncated,
              float *info_all, uint8_t *info_agent, float *final_obs, int32_t auto_reset, void *stream) {
    if (!e || !actions) return fail(e, MAPF_ERR_CONFIG, "null argument");
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_step");
    Params p = e->p;
    p.actions = actions;
    p.obs = obs;
    p.rewards = rewards;
    p.terminated = terminated;
    p.truncated = truncated;
    p.info_all = info_all;
    p.info_agent = info_agent;
    p.final_obs = final_obs;
    p.auto_reset = auto_reset;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, dispatch(KIND_STEP, e, p, (hipStream_t)stream));
    return MAPF_OK;
}

int mapf_observe(mapf_handle e, float *obs, void *stream) {
    if (!e || !obs) return fail(e, MAPF_ERR_CONFIG, "null argument");
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_observe");
    Params p = e->p;
    p.obs = obs;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, dispatch(KIND_OBSERVE, e, p, (hipStream_t)stream));
    return MAPF_OK;
}

int mapf_poll_error(mapf_handle e, void *stream, int32_t *env, int32_t *agent, int32_t *value) {
    if (!e) return MAPF_ERR_CONFIG;
    int rec[4] = {0, 0, 0, 0};
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(e, hipMemcpy(rec, e->d_err, sizeof rec, hipMemcpyDeviceToHost));
    if (rec[0] != 0) {
        HIP_TRY(e, hipMemset(e->d_err, 0, sizeof rec));
        if (env) *env = rec[1];
        if (agent) *agent = rec[2];
        if (value) *value = rec[3];
        char buf[160];
        if (rec[0] == MAPF_ERR_BAD_ACTION)
            snprintf(buf, sizeof buf, "Invalid action %d for agent_%d (env %d)", rec[3], rec[2], rec[1]);
        else if (rec[0] == MAPF_ERR_NO_RESPAWN)
            snprintf(buf, sizeof buf, "No valid cell available for lifelong goal reassignment. (env %d, agent_%d)", rec[1], rec[2]);
        else
            snprintf(buf, sizeof buf, "device error %d in env %d", rec[0], rec[1]);
        e->err = buf;
    }
    return rec[0];
}

int mapf_launch_info(mapf_handle e, int32_t *blocks, int32_t *threads, int32_t *lds_bytes, int32_t *lanes_per_env) {
    if (!e) return MAPF_ERR_CONFIG;
    if (blocks) *blocks = e->blocks;
    if (threads) *threads = 64;
    if (lds_bytes) *lds_bytes = e->lds_bytes;
    if (lanes_per_env) *lanes_per_env = e->lpe;
    return MAPF_OK;
}

}  // extern "C"
