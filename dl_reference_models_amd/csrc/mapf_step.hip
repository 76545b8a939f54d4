// mapf_step.hip -- MI355X (gfx950 / CDNA4) vectorized step engine for the multi-agent grid env.
//
// Replaces the hot path of /root/reference/src/environments/reference_model_multi_agent.py ("MA-env"):
// step() :474-695, reset() :440-472 and the helpers they call.  C ABI: include/mapf_step.h.
//
// Design (see DESIGN.md for the long form):
//   * one wavefront (64 lanes) per workgroup; an env instance owns a GROUP of LPE lanes of the wave
//     (LPE = smallest power of two >= max(N,4)); lane a of a group is agent a.  At N = 64 this is
//     one wavefront per env; at N = 8 one wave steps 8 envs, so no lane idles in the agent phases.
//   * the sequential move rule (MA-env:502-526, lower index wins) is restated as a dependency problem
//     (an agent is blocked by a higher index still standing on its target, by a lower index that stayed
//     there, or by a lower-index contender that got in) and resolved in rounds of two ballots.
//   * the observation of agent i is taken "at time i" (MA-env:528 sits inside the move loop):
//     occupant of a cell = new position of agents <= i, old position of agents > i.  Each lane builds
//     its V x V window as bit masks (obstacle / other agent / own goal / other goal) from the env's
//     bit-packed obstacle rows staged in LDS and an 8-byte-per-agent LDS table that every lane of
//     the group reads in 16-byte chunks; no owner maps are needed.
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
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <type_traits>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "mapf_engine.h"  // mapf_step.h, the device code (mapf_kernels.inl, namespace mapfk) and the launch units' entry points

using namespace mapfk;

namespace {

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local std::string g_create_error;

}  // namespace

struct mapf_engine {
    mapf_config cfg;
    Params p;
    int lpe = 0;
    int mask_w = 32;
    int special = 0;  // id in MAPF_SPECIALIZATIONS, 0 = runtime-config kernel
    int dense = 0;    // the step grid has more than three waves per SIMD: the 128-register build of k_step (WPS = 4)
    int many_dense = 0;  // the fused launch has more than two waves per SIMD: the 128-register build of k_step_many
    int obs_prepared = 0;  // k_step3 with bit rows (goals / old cells / intents: obs3_wave_prepared, state3_outputs; Io::use_map bit 1)
    int three_wave = 0;  // k_step3 (state / observation / aux wave): specialised finite shapes with N = 4 or 8, not dense
    int rt_sliced = 0;   // runtime-config kernels with the sliced background draw (KRuntimeSliced): full groups of 4 / 8 agents
    int wide3 = 0;       // k_stepw: 64-lane groups with the cell-map conditions met step on the three-wave kernel with bit rows
    int wide_lds_bytes = 0;
    // step kernels compiled for exactly this configuration at mapf_create (MAPF_FLAG_JIT_SPECIALIZE), else null
    hipFunction_t jit_step = nullptr, jit_many = nullptr;
    std::string jit_note = "not requested (MAPF_FLAG_JIT_SPECIALIZE)";
    bool cte = false;  // single-agent (CTE) variant
    int col_pad = 0;   // kRowPad when W <= 64 - 2*kRowPad
    int use_map = 0, lds_map_off = 0;  // LDS cell-map path of wide groups
    double cte_blocking_penalty = -0.2, cte_move_after_goal_penalty = -0.05;  // SA-env:92-93
    // single-agent env, fused launches (mapf_cte_step_many, T > 1): their own group width and LDS layout (mapf_create)
    int cte_scratch2_off = 0;
    struct CteManyPlan { int lpe = 0, blocks = 0, lds_bytes = 0, tab_off = 0, stage_off = 0, scratch_off = 0; } cte_many;
    int blocks = 0;
    int sampler_blocks = 0;  // k_step only: workgroups appended to the grid that pre-draw next-episode placements
    int lds_bytes = 0;
    bool grids_set = false;
    std::vector<uint64_t> h_rows;  // host copy of the obstacle rows (mapf_set_state validates injected positions against it)
    std::string err;
    // device allocations
    uint2 *d_agents = nullptr;  // the four planes of the agent state (mapf_kernels.inl: agent_plane_off)
    uint32_t bn8 = 0;           // agent_plane_stride(B, N)
    int *d_scal = nullptr;
    int16_t *d_ring = nullptr;
    uint64_t *d_rng = nullptr;
    uint64_t *d_rows = nullptr;
    uint16_t *d_free_cells = nullptr;
    uint16_t *d_free_rank = nullptr;
    int *d_n_free = nullptr;
    int *d_err = nullptr;
    int *d_ep_acc = nullptr;
    uint32_t *d_stage_vals = nullptr;  // [B][2N] bounded draws of a staged background draw (Params::stage_vals)
    uint64_t *d_jump_c = nullptr;   // [B][32][2] S_q * inc of every env's stream (Io::jump_c), rewritten whenever the streams are set
    uint64_t *d_vis_rng = nullptr;  // [B][6] visible stream state of envs whose placement slot is pending (Params::vis_rng)
    Params *d_params = nullptr;  // device copy of `p`, read by the kernels through a pointer
    unsigned long long *d_dbg = nullptr;  // stamps buffer (diagnostic build only)
    // mapf_bind_outputs: the caller's output buffers for mapf_step_bound
    float *b_obs = nullptr, *b_rewards = nullptr, *b_info_all = nullptr;
    uint8_t *b_terminated = nullptr, *b_truncated = nullptr, *b_info_agent = nullptr;
    bool bound = false;
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

// Makes the handle's GPU current for the duration of an ABI call and puts the caller's device back on every exit
// path: a process that holds envs on several GPUs (or whose torch current device differs) must not find its
// thread's device switched by a step() or by a destructor run from the garbage collector.
struct DeviceScope {
    int prev = -1;
    hipError_t status = hipSuccess;
    explicit DeviceScope(int dev) {
        status = hipGetDevice(&prev);
        if (status != hipSuccess) { prev = -1; return; }
        if (prev != dev) status = hipSetDevice(dev); else prev = -1;  // hot path: nothing to do, nothing to undo
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
#define ON_DEVICE(e)                                                                                       \
    DeviceScope _dev_scope((e)->cfg.device);                                                               \
    if (_dev_scope.status != hipSuccess)                                                                   \
        return fail((e), MAPF_ERR_HIP, std::string("selecting the handle's device: ") + hipGetErrorString(_dev_scope.status))

#define LAUNCH_TRY(e, call)                                                                               \
    do {                                                                                                  \
        hipError_t _s = (call);                                                                           \
        if (_s != hipSuccess)                                                                             \
            return fail((e), MAPF_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(_s)); \
    } while (0)

// engine knobs of mapf_config.flags that choose among builds of the same kernel (never part of the env's configuration)
constexpr uint32_t kKernelChoiceFlags = MAPF_FLAG_FORCE_DENSE | MAPF_FLAG_FORCE_SPARSE | MAPF_FLAG_SAMPLER_WORKGROUPS | MAPF_FLAG_TWO_WAVE_WIDE |
                                        MAPF_FLAG_TABLE_WALK_OBS | MAPF_FLAG_NO_BIT_ROWS;

// Development builds only (-DMAPF_DEV): knobs read from the environment for A/B timing.  The shipped library reads
// MAPF_JIT_CACHE_DIR (and the usual XDG / HOME variables behind it) and nothing else: which kernel a handle runs is a
// function of its mapf_config alone.
#ifdef MAPF_DEV
const char *dev_env(const char *name) { return getenv(name); }
#else
constexpr const char *dev_env(const char *) { return nullptr; }
#endif

int pick_lpe(int n) {
    int l = 4;
    while (l < n) l <<= 1;
    return l;
}

// the step kernel compiled for one of the BASELINE.json shapes (MAPF_SPECIALIZATIONS), if the config matches
int match_specialization(const mapf_config &c, int lpe, int nearby_clamped) {
    if (c.flags & MAPF_FLAG_GENERIC_KERNEL) return 0;
    const uint32_t cfg_flags = c.flags & ~(MAPF_FLAG_NO_CELL_MAP | MAPF_FLAG_SEQUENTIAL_RESET | MAPF_FLAG_JIT_SPECIALIZE | kKernelChoiceFlags);
#define MAPF_MATCH(ID, N_, SR_, FLAGS_, DW_, LW_, NEARBY_, MINN_, LPE_)                                            \
    if (c.num_agents == N_ && c.sensor_range == SR_ && cfg_flags == (uint32_t)(FLAGS_) &&                             \
        c.deadlock_window_steps == DW_ && c.livelock_window_steps == LW_ && nearby_clamped == NEARBY_ &&           \
        c.lock_min_neighbors == MINN_ && lpe == LPE_)                                                              \
        return ID;
    MAPF_SPECIALIZATIONS(MAPF_MATCH)
#undef MAPF_MATCH
    return 0;
}

LaunchPlan plan_of(const mapf_engine *e) {
    LaunchPlan lp;
    lp.d_params = e->d_params;
    lp.blocks = e->blocks;
    lp.sampler_blocks = e->sampler_blocks;
    lp.lds_bytes = e->lds_bytes;
    lp.dense = e->dense;
    lp.many_dense = e->many_dense;
    lp.three_wave = e->three_wave;
    lp.rt_sliced = e->rt_sliced;
    lp.wide3 = e->wide3;
    lp.wide_lds_bytes = e->wide_lds_bytes;
    return lp;
}

// ------------------------------------------------------------------------------------------------
// Run-time specialisation (MAPF_FLAG_JIT_SPECIALIZE).  KFixed<...> folds agent count, observation layout, flags and
// lock windows into the step kernels; the library carries that for the BASELINE.json shapes and the reference's
// training setup (MAPF_SPECIALIZATIONS).  For any other configuration the same template can be instantiated when
// the handle is created: hiprtc compiles mapf_kernels.inl (found next to this library) with the one k_step /
// k_step_many instantiation the handle needs -- 2-3 s each -- and the handle launches those through the module API.
// hiprtc is loaded with dlopen: no link-time dependency, and every failure (no hiprtc, no source, a compile error)
// leaves the handle on the runtime-config kernels with the reason in mapf_jit_status().  Compiled modules are kept
// per process, keyed by the instantiation.
// ------------------------------------------------------------------------------------------------
struct Hiprtc {
    void *lib = nullptr;
    int (*create)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*add_name)(void *, const char *) = nullptr;
    int (*compile)(void *, int, const char **) = nullptr;
    int (*log_size)(void *, size_t *) = nullptr;
    int (*log)(void *, char *) = nullptr;
    int (*code_size)(void *, size_t *) = nullptr;
    int (*code)(void *, char *) = nullptr;
    int (*lowered)(void *, const char *, const char **) = nullptr;
    int (*destroy)(void **) = nullptr;
    int (*version)(int *, int *) = nullptr;  // optional
    std::string ver = "?";  // "major.minor" of the hiprtc the process resolved.  Under PyTorch that is the wheel's bundled
                            // ROCm, not the toolkit this library was built with: on this image clang 20 (ROCm 7.0) against
                            // clang 22 (7.2), which is the whole difference between a kernel compiled at creation and the
                            // same kernel prebuilt ("Run-time specialisation")
    std::string build_id;   // path : size : mtime of the library file (part of the on-disk cache key)
    bool ok = false;
};
const Hiprtc &hiprtc_api() {
    static Hiprtc h;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            h.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h.lib) break;
        }
        if (!h.lib) return;
        bool all = true;
        auto sym = [&](auto &fn, const char *n) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h.lib, n));
            all = all && fn != nullptr;
        };
        sym(h.create, "hiprtcCreateProgram");
        sym(h.add_name, "hiprtcAddNameExpression");
        sym(h.compile, "hiprtcCompileProgram");
        sym(h.log_size, "hiprtcGetProgramLogSize");
        sym(h.log, "hiprtcGetProgramLog");
        sym(h.code_size, "hiprtcGetCodeSize");
        sym(h.code, "hiprtcGetCode");
        sym(h.lowered, "hiprtcGetLoweredName");
        sym(h.destroy, "hiprtcDestroyProgram");
        h.ok = all;
        h.version = reinterpret_cast<int (*)(int *, int *)>(dlsym(h.lib, "hiprtcVersion"));
        int major = 0, minor = 0;
        if (h.version && h.version(&major, &minor) == 0) h.ver = std::to_string(major) + "." + std::to_string(minor);
        // hiprtcVersion() stops at major.minor; the library file's identity (path, size, modification time) tells patch
        // levels and rebuilt toolchains apart for the on-disk cache
        Dl_info info;
        struct stat st;
        if (dladdr(reinterpret_cast<void *>(h.create), &info) && info.dli_fname && stat(info.dli_fname, &st) == 0)
            h.build_id = std::string(info.dli_fname) + ":" + std::to_string((long long)st.st_size) + ":" + std::to_string((long long)st.st_mtime);
    });
    return h;
}

struct JitModule {
    hipModule_t mod = nullptr;
    hipFunction_t step = nullptr, many = nullptr;
    double seconds = 0;
    bool from_disk = false;
};
std::mutex g_jit_mutex;
std::map<std::string, JitModule> g_jit_cache;  // key: device + the instantiation

// directory of this shared library (the kernel source is installed next to it, the header two levels up)
std::string library_dir() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<void *>(&mapf_version), &info) || !info.dli_fname) return "";
    std::string f = info.dli_fname;
    const size_t k = f.rfind('/');
    return k == std::string::npos ? "." : f.substr(0, k);
}

// On-disk cache of compiled code objects: MAPF_JIT_CACHE_DIR, else $XDG_CACHE_HOME/mapf_jit, else ~/.cache/mapf_jit (empty
// MAPF_JIT_CACHE_DIR = no cache).  A file is named by a hash of everything the code object depends on -- the kernel
// source and the ABI header as found next to the library, the instantiations, the target and the compile options -- so
// a rebuilt library or an edited source never meets a stale entry; the 0.5-2.5 s of hiprtc are paid once per machine and
// configuration instead of once per process.
static std::string jit_cache_dir() {
    if (const char *d = getenv("MAPF_JIT_CACHE_DIR")) return d;
    if (const char *x = getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/mapf_jit";
    if (const char *h = getenv("HOME")) if (*h) return std::string(h) + "/.cache/mapf_jit";
    return "";
}
static uint64_t fnv1a(const std::string &s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}
static std::string slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void mkdirs(const std::string &dir) {  // best effort (like `mkdir -p`), private to the user
    for (size_t k = 1; k <= dir.size(); k++)
        if (k == dir.size() || dir[k] == '/') (void)mkdir(dir.substr(0, k).c_str(), 0700);
}
// The cache holds executable code objects: it is only read from / written to a directory (and files) that belong to
// this user and that nobody else can write to.
static bool private_to_user(const std::string &path) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return false;
    return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}

// Tries to give `e` step kernels compiled for its configuration; on any failure e->jit_note says why.
void jit_specialize(mapf_engine *e) {
    const mapf_config &c = e->cfg;
    const int N = c.num_agents, lpe = e->lpe;
    if (e->cte) { e->jit_note = "single-agent env: runtime-config kernels only"; return; }
    if (c.flags & MAPF_FLAG_GENERIC_KERNEL) { e->jit_note = "MAPF_FLAG_GENERIC_KERNEL is set"; return; }
    // (MAPF_JIT_PREBUILT_TOO: development knob -- compile even when a prebuilt specialisation matches, so that an edited
    //  mapf_kernels.inl can be timed against the library's own kernels without rebuilding the library)
    if (e->special && !dev_env("MAPF_JIT_PREBUILT_TOO")) { e->jit_note = "a prebuilt specialisation matches"; return; }
    if (lpe != pick_lpe(N)) { e->jit_note = "lanes_per_env overrides the group width"; return; }
    if (N > 16 && !e->use_map) { e->jit_note = "wide group without the LDS cell map"; return; }
    const Hiprtc &rt = hiprtc_api();
    if (!rt.ok) { e->jit_note = "libhiprtc.so could not be loaded"; return; }
    const std::string dir = library_dir();
    if (dir.empty() || !std::ifstream(dir + "/mapf_kernels.inl").good()) {
        e->jit_note = "mapf_kernels.inl not found next to the library (" + dir + ")";
        return;
    }
    char inst[256];
    snprintf(inst, sizeof inst, "KFixed<%d, %d, %uu, %d, %d, %d, %d>", N, c.sensor_range, (unsigned)e->p.flags & ~MAPF_FLAG_SEQUENTIAL_RESET,
             c.deadlock_window_steps, c.livelock_window_steps, e->p.nearby, c.lock_min_neighbors);
    char tail_[64];
    snprintf(tail_, sizeof tail_, ", %d, %d", lpe, e->mask_w);
    const int wps = (lpe < 32 && e->dense) ? 4 : 0;
    const char *step_kernel = e->wide3 ? "k_stepw" : (e->three_wave ? "k_step3" : "k_step");  // (three_wave implies wps == 0)
    char wtail[32];
    snprintf(wtail, sizeof wtail, ", %d>", e->mask_w);  // k_stepw<K, MW>
    const std::string step_args = e->wide3 ? std::string(inst) + wtail : std::string(inst) + tail_ + ", " + std::to_string(wps) + ">";
    const std::string step_expr = std::string("mapfjit::") + step_kernel + "<mapfjit::" + step_args;
    const int many_wps = (lpe < 32 && e->many_dense) ? 4 : 0;
    const std::string many_expr = std::string("mapfjit::k_step_many<mapfjit::") + inst + tail_ + ", " + std::to_string(many_wps) + ">";
    const std::string key = std::to_string(c.device) + "|" + step_expr + "|" + many_expr;  // (the module holds both kernels)
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    auto it = g_jit_cache.find(key);
    if (it == g_jit_cache.end()) {
        const auto t0 = std::chrono::steady_clock::now();
        hipDeviceProp_t prop;
        std::string arch = "gfx950";
        if (hipGetDeviceProperties(&prop, c.device) == hipSuccess && prop.gcnArchName[0]) {
            arch = prop.gcnArchName;
            arch = arch.substr(0, arch.find(':'));
        }
        // ---- the on-disk cache first
        std::string cdir = jit_cache_dir();
        std::string cfile;
        if (!cdir.empty()) {
            mkdirs(cdir);
            if (!private_to_user(cdir)) cdir.clear();  // (someone else's or a world-writable directory: no cache)
        }
        if (!cdir.empty()) {
            uint64_t h = fnv1a(slurp(dir + "/mapf_kernels.inl"));
            h = fnv1a(slurp(dir + "/../../include/mapf_step.h"), h);
            h = fnv1a(step_expr + "|" + many_expr + "|" + arch + "|O3 c++17 kernarg-preload-16 v2|hiprtc " + rt.ver + "|" + rt.build_id, h);
            char name[40];
            snprintf(name, sizeof name, "/%016llx", (unsigned long long)h);
            cfile = cdir + name;
            const bool trusted = private_to_user(cfile + ".co") && private_to_user(cfile + ".names");
            const std::string code = trusted ? slurp(cfile + ".co") : std::string(), names = trusted ? slurp(cfile + ".names") : std::string();
            const size_t nl = names.find('\n');
            if (!code.empty() && nl != std::string::npos) {
                JitModule m;
                const std::string sn = names.substr(0, nl), mn = names.substr(nl + 1, names.find('\n', nl + 1) - nl - 1);
                hipError_t hs = hipModuleLoadData(&m.mod, code.data());
                if (hs == hipSuccess) hs = hipModuleGetFunction(&m.step, m.mod, sn.c_str());
                if (hs == hipSuccess) hs = hipModuleGetFunction(&m.many, m.mod, mn.c_str());
                if (hs == hipSuccess && m.step && m.many) {
                    m.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    m.from_disk = true;
                    it = g_jit_cache.emplace(key, m).first;
                } else if (m.mod) {
                    (void)hipModuleUnload(m.mod);  // (a damaged file: compile afresh and overwrite it)
                }
            }
        }
      if (it == g_jit_cache.end()) {
        std::string src = "#include \"mapf_step.h\"\n#include \"mapf_kernels.inl\"\nnamespace mapfjit {\n";
        src += std::string("template __global__ void ") + step_kernel + "<" + step_args + "(const Params *, MAPF_IO_HEAD_PARAMS, const IoTail);\n";
        src += std::string("template __global__ void k_step_many<") + inst + tail_ + ", " + std::to_string(many_wps) +
               ">(const Params *, MAPF_IO_HEAD_PARAMS, const IoTail, const int, const int, const ManyPolicy);\n}\n";
        void *prog = nullptr;
        if (rt.create(&prog, src.c_str(), "mapf_jit.hip", 0, nullptr, nullptr) != 0) { e->jit_note = "hiprtcCreateProgram failed"; return; }
        rt.add_name(prog, step_expr.c_str());
        rt.add_name(prog, many_expr.c_str());
        const std::string o_arch = "--offload-arch=" + arch, o_i1 = "-I" + dir, o_i2 = "-I" + dir + "/../../include";
        const char *opts[] = {o_arch.c_str(), "-O3", "-std=c++17", o_i1.c_str(), o_i2.c_str(), "-mllvm", "-amdgpu-kernarg-preload-count=16",
                              "-Wno-unused-value", "-DMAPF_NS=mapfjit", "-DMAPF_NS_IS_JIT"};
        const int rc = rt.compile(prog, (int)(sizeof opts / sizeof opts[0]), opts);
        if (rc != 0) {
            size_t n = 0;
            rt.log_size(prog, &n);
            std::string log(n, '\0');
            if (n) rt.log(prog, &log[0]);
            e->jit_note = "hiprtc compile error: " + log.substr(0, 600);
            rt.destroy(&prog);
            return;
        }
        size_t n = 0;
        rt.code_size(prog, &n);
        std::vector<char> code(n);
        rt.code(prog, code.data());
        const char *step_name = nullptr, *many_name = nullptr;
        rt.lowered(prog, step_expr.c_str(), &step_name);
        rt.lowered(prog, many_expr.c_str(), &many_name);
        const std::string step_sym = step_name ? step_name : "", many_sym = many_name ? many_name : "";  // (prog owns the names)
        JitModule m;
        hipError_t hs = hipModuleLoadData(&m.mod, code.data());
        if (hs == hipSuccess && step_name) hs = hipModuleGetFunction(&m.step, m.mod, step_name);
        if (hs == hipSuccess && many_name) hs = hipModuleGetFunction(&m.many, m.mod, many_name);
        rt.destroy(&prog);
        if (hs != hipSuccess || !m.step || !m.many) {
            e->jit_note = std::string("loading the compiled module failed: ") + hipGetErrorString(hs);
            if (m.mod) (void)hipModuleUnload(m.mod);
            return;
        }
        m.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (!cfile.empty() && !step_sym.empty() && !many_sym.empty()) {  // best effort: write to a temporary name, then rename
            const std::string tmp = cfile + ".tmp" + std::to_string((long)getpid());
            std::ofstream fc(tmp + ".co", std::ios::binary), fn(tmp + ".names");
            fc.write(code.data(), (std::streamsize)code.size());
            fn << step_sym << "\n" << many_sym << "\n";
            fc.close();
            fn.close();
            if (fc.good() && fn.good()) {
                (void)rename((tmp + ".co").c_str(), (cfile + ".co").c_str());
                (void)rename((tmp + ".names").c_str(), (cfile + ".names").c_str());
            }
        }
        it = g_jit_cache.emplace(key, m).first;
      }
    }
    e->jit_step = it->second.step;
    e->jit_many = it->second.many;
    char note[160];
    snprintf(note, sizeof note, it->second.from_disk ? "loaded from the on-disk cache in %.2f s (hiprtc %s): " : "compiled in %.1f s (hiprtc %s): ",
             it->second.seconds, rt.ver.c_str());
    e->jit_note = note + step_expr;
}

hipError_t launch_jit_step(const mapf_engine *e, const Io &io, hipStream_t s) {
    const Params *pp = e->d_params;
    IoTail tail = static_cast<const IoTail &>(io);
    Io h = io;
    void *args[] = {&pp, &h.agents, &h.scal, &h.grid_rows, &h.actions, &h.B, &h.H, &h.W, &h.bn8, &tail};
    return hipModuleLaunchKernel(e->jit_step, e->blocks + e->sampler_blocks, 1, 1, (e->three_wave || e->wide3) ? 192 : step_threads(e->lpe), 1, 1,
                                 e->wide3 ? e->wide_lds_bytes : e->lds_bytes, s, args, nullptr);
}
hipError_t launch_jit_many(const mapf_engine *e, const Io &io, int T, int obs_mode, const ManyPolicy &pol, hipStream_t s) {
    const Params *pp = e->d_params;
    IoTail tail = static_cast<const IoTail &>(io);
    Io h = io;
    ManyPolicy pl = pol;
    void *args[] = {&pp, &h.agents, &h.scal, &h.grid_rows, &h.actions, &h.B, &h.H, &h.W, &h.bn8, &tail, &T, &obs_mode, &pl};
    return hipModuleLaunchKernel(e->jit_many, e->blocks, 1, 1, many_threads(e->lpe), 1, 1, e->lds_bytes, s, args, nullptr);
}

// Which launch unit serves a handle: the prebuilt specialisation it matched, else the runtime-config kernels of its group
// width and window-mask width (mapf_launch.hip; a reduced build only declares -- and links -- the units it holds, and
// mapf_create has refused every configuration outside them).
#define MAPF_RUNTIME_CASE(L, MW) \
    if (e->lpe == L && e->mask_w == MW) return MAPF_RUNTIME_CALL(L, MW);
#define MAPF_RUNTIME_CASES(L) MAPF_FOR_MW(MAPF_RUNTIME_CASE, L)

hipError_t dispatch_many(const mapf_engine *e, const Io &io, int T, int obs_mode, const ManyPolicy &pol, hipStream_t s) {
    if (e->jit_many) return launch_jit_many(e, io, T, obs_mode, pol, s);
    const LaunchPlan lp = plan_of(e);
    switch (e->special) {
#define MAPF_LAUNCH(ID, N_, SR_, FLAGS_, DW_, LW_, NEARBY_, MINN_, LPE_) \
    case ID:                                                            \
        return launch_special_many_##ID(lp, io, T, obs_mode, pol, s);
        MAPF_SPECIALIZATIONS(MAPF_LAUNCH)
#undef MAPF_LAUNCH
    }
#define MAPF_RUNTIME_CALL(L, MW) launch_runtime_many_##L##_##MW(lp, io, T, obs_mode, pol, s)
    MAPF_FOR_LPE(MAPF_RUNTIME_CASES)
#undef MAPF_RUNTIME_CALL
    return hipErrorInvalidValue;
}

hipError_t dispatch(int kind, const mapf_engine *e, const Io &io, hipStream_t s) {
    if (kind == KIND_STEP && e->jit_step) return launch_jit_step(e, io, s);
    const LaunchPlan lp = plan_of(e);
    if (kind == KIND_STEP) {
        switch (e->special) {
#define MAPF_LAUNCH(ID, N_, SR_, FLAGS_, DW_, LW_, NEARBY_, MINN_, LPE_) \
    case ID:                                                            \
        return launch_special_step_##ID(lp, io, s);
            MAPF_SPECIALIZATIONS(MAPF_LAUNCH)
#undef MAPF_LAUNCH
        }
    }
#define MAPF_RUNTIME_CALL(L, MW) launch_runtime_##L##_##MW(kind, lp, io, s)
    MAPF_FOR_LPE(MAPF_RUNTIME_CASES)
#undef MAPF_RUNTIME_CALL
    return hipErrorInvalidValue;
}

}  // namespace

static int alloc_device_state(mapf_engine *e);

// Host writes to an env's stream or free-cell tables void the placements pre-drawn from them (kSlotInvalid).
static hipError_t invalidate_slots(mapf_engine *e) {
    return hipMemset(e->p.next_sg, 0xFF, (size_t)e->p.B * e->p.N * sizeof(uint32_t));
}
// What mapf_set_grids has to put right on the device once the new rows are there, one thread per agent:
//  * a pending (staged or valid) slot means that the env's stream array already holds the state AFTER the background draw
//    and the visible state sits in vis_rng (mapf_kernels.inl: kSlotInvalid).  Voiding such a slot without replacing the
//    stream (the free-cell tables changed, the stream did not) must put the visible state back, or the env would silently
//    skip one rng.choice draw;
//  * the agents' pass bits are a function of the grid (agent_pass_bits: which neighbours of the position can be stepped on).
// (Until round 4 both were host loops over downloaded copies of the slots, the streams and all four agent planes.)
static __global__ __launch_bounds__(256) void k_after_set_grids(uint2 *__restrict__ hot, const uint64_t *__restrict__ rows, int B, int N, int H,
                                                          int W, int col_pad, const uint32_t *__restrict__ slots,
                                                          const uint64_t *__restrict__ vis, uint64_t *__restrict__ rng) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= B * N) return;
    const int env = i / N, a = i - env * N;
    if (a == 0 && slots[(size_t)env * N] != kSlotInvalid)
        for (int k = 0; k < 6; k++) rng[(size_t)env * 6 + k] = vis[(size_t)env * 6 + k];
    uint2 w = hot[i];
    const int r = (int)((w.x >> 8) & 255u), c = (int)(w.x & 255u);
    const uint64_t *my = rows + (size_t)env * H;
    auto blocked = [&](int rr, int cc) -> uint32_t {
        if (rr < 0 || rr >= H || cc < 0 || cc >= W) return 1u;
        return (uint32_t)((my[rr] >> (cc + col_pad)) & 1ull);
    };
    const uint32_t pass = (blocked(r - 1, c) | (blocked(r, c + 1) << 1) | (blocked(r + 1, c) << 2) | (blocked(r, c - 1) << 3)) ^ 15u;
    w.y = (w.y & 0x00FFFFFFu) | (pass << 24);
    hot[i] = w;
}
// After host writes to positions / goals / counters the MAY_FINISH hint of the last step is stale: force it on
// (conservative: the sampler skips the env for one step, the next step writes the real hint).
static hipError_t force_may_finish(mapf_engine *e) {
    return hipMemset2D(e->d_scal + MAPF_CTR_MAY_FINISH, kScalInts * sizeof(int), 1, sizeof(int), (size_t)e->p.B);
}

// Io::jump_c: for every env the second half of the PCG64 jump-ahead, S_q * inc mod 2^128 for q = 1 .. 32, with
// S_1 = 1, S_(q+1) = S_q * M + 1 (state_q = M^q * state + S_q * inc).  words = [B][6] stream words (inc in words 2, 3).
static int upload_jump_table(mapf_engine *e, const uint64_t *words) {
    const int B = e->p.B;
    std::vector<uint64_t> tab((size_t)B * 64);
    const unsigned __int128 M = ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
    for (int b = 0; b < B; b++) {
        const unsigned __int128 inc = ((unsigned __int128)words[(size_t)b * 6 + 2] << 64) | words[(size_t)b * 6 + 3];
        unsigned __int128 S = 1;
        for (int q = 1; q <= 32; q++) {
            const unsigned __int128 c = S * inc;
            tab[(size_t)b * 64 + 2 * (q - 1)] = (uint64_t)(c >> 64);
            tab[(size_t)b * 64 + 2 * (q - 1) + 1] = (uint64_t)c;
            S = S * M + 1;
        }
    }
    HIP_TRY(e, hipMemcpy(e->d_jump_c, tab.data(), tab.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    return MAPF_OK;
}

// Host image of the agent state <-> the device planes (mapf_kernels.inl: agent_plane_off).
static int download_agents(mapf_engine *e, std::vector<AgentRec> &recs) {
    const size_t BN = (size_t)e->p.B * e->p.N;
    std::vector<unsigned char> raw(agent_state_bytes(e->bn8));
    HIP_TRY(e, hipMemcpy(raw.data(), e->d_agents, raw.size(), hipMemcpyDeviceToHost));
    const uint32_t *p0 = reinterpret_cast<const uint32_t *>(raw.data() + agent_plane_off(0, e->bn8));
    const uint32_t *p1 = reinterpret_cast<const uint32_t *>(raw.data() + agent_plane_off(1, e->bn8));
    const uint32_t *p2 = reinterpret_cast<const uint32_t *>(raw.data() + agent_plane_off(2, e->bn8));
    const uint32_t *p3 = reinterpret_cast<const uint32_t *>(raw.data() + agent_plane_off(3, e->bn8));
    recs.resize(BN);
    for (size_t i = 0; i < BN; i++) {
        AgentRec &r = recs[i];
        r.w0 = p0[2 * i];
        r.w1 = p0[2 * i + 1];
        r.moved = (uint64_t)p1[4 * i] | ((uint64_t)p1[4 * i + 1] << 32);
        r.failed = (uint64_t)p1[4 * i + 2] | ((uint64_t)p1[4 * i + 3] << 32);
        r.progress = (uint64_t)p2[4 * i] | ((uint64_t)p2[4 * i + 1] << 32);
        r.dist[0] = p2[4 * i + 2];
        r.dist[1] = p2[4 * i + 3];
        r.dist[2] = p3[2 * i];
        r.dist[3] = p3[2 * i + 1];
    }
    return MAPF_OK;
}
// the pass bits of w1 (which neighbours of the position the grid lets an agent step on) are recomputed here from the
// host copy of the obstacle rows: every writer of a position keeps them current (agent_pass_bits on the device)
static uint32_t host_pass_bits(const mapf_engine *e, int b, uint32_t cell) {
    if (!e->grids_set) return 0;
    const int H = e->p.H, W = e->p.W, pad = e->col_pad;
    const int r = (int)(cell >> 8), c = (int)(cell & 255u);
    auto blocked = [&](int rr, int cc) -> uint32_t {
        if (rr < 0 || rr >= H || cc < 0 || cc >= W) return 1u;
        return (uint32_t)((e->h_rows[(size_t)b * H + rr] >> (cc + pad)) & 1ull);
    };
    return (blocked(r - 1, c) | (blocked(r, c + 1) << 1) | (blocked(r + 1, c) << 2) | (blocked(r, c - 1) << 3)) ^ 15u;
}
static int upload_agents(mapf_engine *e, const std::vector<AgentRec> &recs) {
    const size_t BN = (size_t)e->p.B * e->p.N;
    const int N = e->p.N;
    std::vector<unsigned char> raw(agent_state_bytes(e->bn8), 0);
    uint32_t *p0 = reinterpret_cast<uint32_t *>(raw.data() + agent_plane_off(0, e->bn8));
    uint32_t *p1 = reinterpret_cast<uint32_t *>(raw.data() + agent_plane_off(1, e->bn8));
    uint32_t *p2 = reinterpret_cast<uint32_t *>(raw.data() + agent_plane_off(2, e->bn8));
    uint32_t *p3 = reinterpret_cast<uint32_t *>(raw.data() + agent_plane_off(3, e->bn8));
    for (size_t i = 0; i < BN; i++) {
        const AgentRec &r = recs[i];
        p0[2 * i] = r.w0;
        p0[2 * i + 1] = (r.w1 & 0x00FFFFFFu) | (host_pass_bits(e, (int)(i / N), r.w0 & 0xFFFFu) << 24);
        p1[4 * i] = (uint32_t)r.moved;
        p1[4 * i + 1] = (uint32_t)(r.moved >> 32);
        p1[4 * i + 2] = (uint32_t)r.failed;
        p1[4 * i + 3] = (uint32_t)(r.failed >> 32);
        p2[4 * i] = (uint32_t)r.progress;
        p2[4 * i + 1] = (uint32_t)(r.progress >> 32);
        p2[4 * i + 2] = r.dist[0];
        p2[4 * i + 3] = r.dist[1];
        p3[2 * i] = r.dist[2];
        p3[2 * i + 1] = r.dist[3];
    }
    HIP_TRY(e, hipMemcpy(e->d_agents, raw.data(), raw.size(), hipMemcpyHostToDevice));
    return MAPF_OK;
}

extern "C" {

uint32_t mapf_version(void) { return (MAPF_VERSION_MAJOR << 16) | MAPF_VERSION_MINOR; }

int32_t mapf_obs_len(const mapf_config *cfg) {
    if (cfg->flags & MAPF_FLAG_SINGLE_AGENT) return cfg->height * cfg->width + 5 * cfg->num_agents;  // SA-env:124-141
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
    // two cells of a <= 64x64 grid are at most 126 apart, so a larger radius means the same thing; the cap
    // keeps the idle-lane sentinel cell (row 255) outside every neighbourhood
    if (c.lock_nearby_manhattan > 126) c.lock_nearby_manhattan = 126;
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

    const bool cte = (c.flags & MAPF_FLAG_SINGLE_AGENT) != 0;
    // single-agent env: one staging row of H*W + 5N floats per env has to fit a wave's 64 KiB of LDS
    auto cte_fits = [&](int l) { return (64 / l) * (c.height * c.width + 5 * c.num_agents) * 4 <= 56 * 1024; };
    if (cte && !c.lanes_per_env) {
        // The work of this env is the H*W observation row, which spreads over however many lanes the group has.
        // Single-step launches: at most 32 cells per lane, then widen until the launch has one two-wave workgroup per
        // SIMD (half of that once a group has 32 lanes).  Measured, us per step in phase (profiles/r04/
        // cte_lanes_sweep.jsonl): 8192 x 16x16 x 4 agents 5.0 at 8 lanes, 5.2 at 16; 32x32 x 8 agents at 8 / 16 / 32 / 64
        // lanes: 2048 envs 7.2 / 5.5 / 4.8 / 5.5, 4096 envs 8.2 / 6.9 / 6.2 / 9.0, 8192 envs 16.5 / 12.4 / 10.8 / 14.4,
        // 16 384 envs 25.1 / 19.4 / 18.1 / 25.0.
        while (lpe < 64 && c.height * c.width > 32 * lpe) lpe <<= 1;
        while (lpe < 64 && !cte_fits(lpe)) lpe <<= 1;
        while (lpe < 64 && (int64_t)c.num_envs * lpe / 64 < (lpe >= 16 ? 512 : 1024)) lpe <<= 1;
    }
#if defined(MAPF_DEV_C5)
    if (lpe != 64 || c.sensor_range > 2 || cte)
        return fail(nullptr, MAPF_ERR_CONFIG, "this development build only holds 64-lane groups with windows up to 5 x 5");
#endif
#if defined(MAPF_DEV_N16)
    if (lpe != 16 || c.sensor_range != 3 || cte)
        return fail(nullptr, MAPF_ERR_CONFIG, "this development build only holds 16-lane groups with 7 x 7 windows");
#endif
#if defined(MAPF_DEV_C3)
    if ((lpe != 4 && lpe != 8) || c.sensor_range > 2 || cte)
        return fail(nullptr, MAPF_ERR_CONFIG, "this development build only holds groups of 4 and 8 lanes with windows up to 5 x 5");
#endif
#if defined(MAPF_SMALL_SHAPES)
    if (lpe > 16 || c.sensor_range > 3 || cte)
        return fail(nullptr, MAPF_ERR_CONFIG, "this reduced build (checking) only holds groups of 4, 8 and 16 lanes with windows up to 7 x 7");
#endif
    mapf_engine *e = new mapf_engine();
    e->cfg = c;
    e->cte = cte;
    e->col_pad = (c.width <= 64 - 2 * kRowPad) ? kRowPad : 0;
    e->lpe = lpe;
    {
        const int vv = (2 * c.sensor_range + 1) * (2 * c.sensor_range + 1);
        e->mask_w = vv <= 32 ? 32 : (vv <= 64 ? 64 : 128);
    }
    const int G = 64 / lpe;
    const int B = c.num_envs, N = c.num_agents, H = c.height, W = c.width;
    e->blocks = (B + G - 1) / G;
    // Launch shape of the step kernel.  Finite episodes with sampled placements pre-draw the next placement in the
    // background (the env's stream is consumed by reset() alone): kernels of the specialised small-group shapes
    // (KFixed::kSlicedDraw: prebuilt, or compiled at creation) do it in slices inside the env workgroups, the others in
    // sampler workgroups of the k_step grid (mapf_kernels.inl: sampler_wave; each of their waves looks after 64 envs).
    // plan_grid(sliced) is called again when a requested run-time specialisation does not come about.
    const bool finite_sampled = !cte && !(c.flags & (MAPF_FLAG_LIFELONG | MAPF_FLAG_DETERMINISTIC));
    const int special_id = cte ? 0 : match_specialization(c, lpe, c.lock_nearby_manhattan);
    const bool small_full = (c.num_agents == 4 || c.num_agents == 8 || c.num_agents == 16) && lpe == c.num_agents;
    const bool jit_sliced = !special_id && !cte && (c.flags & MAPF_FLAG_JIT_SPECIALIZE) && !(c.flags & MAPF_FLAG_GENERIC_KERNEL) &&
                            lpe == pick_lpe(N) && small_full && finite_sampled;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device) != hipSuccess || cus <= 0) cus = 256;
    auto plan_grid = [&](bool sliced) {
        e->sampler_blocks = 0;
#ifndef MAPF_NO_SAMPLER_WG  // (A/B builds: no background sampler, every reset draws inline)
        if (finite_sampled && !sliced) e->sampler_blocks = sampler_blocks_for(B, step_threads(lpe) / 64);
#endif
        // more than three waves per SIMD in one launch of the step kernel?  (k_step's WPS; 4 SIMDs per compute unit)
        const int64_t waves = (int64_t)(e->blocks + e->sampler_blocks) * (step_threads(lpe) / 64);
        e->dense = waves > (int64_t)3 * 4 * cus;
        if (c.flags & MAPF_FLAG_FORCE_DENSE) e->dense = 1;  // engine knobs (tests): either build on any grid
        if (c.flags & MAPF_FLAG_FORCE_SPARSE) e->dense = 0;
        // k_step3 (state / observation / aux wave): the sliced-draw shapes, while a launch has at most three waves per
        // SIMD with three waves per workgroup (beyond that the two-wave kernel's 128-register build is the one that fits)
        e->many_dense = (int64_t)e->blocks * (many_threads(lpe) / 64) > (int64_t)2 * 4 * cus;
        if (c.flags & MAPF_FLAG_FORCE_DENSE) e->many_dense = 1;
        if (c.flags & MAPF_FLAG_FORCE_SPARSE) e->many_dense = 0;
        e->three_wave = sliced && lpe <= 16 && (int64_t)e->blocks * 3 <= (int64_t)3 * 4 * cus && !e->dense;
        if (const char *f = dev_env("MAPF_THREE_WAVE")) e->three_wave = sliced && lpe <= 16 && atoi(f) != 0;  // (-DMAPF_DEV: A/B)
        if (e->three_wave) e->dense = 0;
    };
    // (round 3) the runtime-config kernels of such shapes take the sliced draw as well; MAPF_FLAG_SAMPLER_WORKGROUPS keeps
    // them on the sampler workgroups (engine knob: tests run both)
    const bool rt_sliced = small_full && finite_sampled && !special_id && !(c.flags & MAPF_FLAG_SAMPLER_WORKGROUPS);
    e->rt_sliced = rt_sliced && !jit_sliced;
    plan_grid(small_full && finite_sampled && (special_id != 0 || jit_sliced || rt_sliced));

    Params &p = e->p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.N = N;
    p.sr = c.sensor_range;
    p.V = 2 * c.sensor_range + 1;
    p.L = mapf_obs_len(&c);
    p.steps_per_episode = c.steps_per_episode;
    p.flags = c.flags & ~(MAPF_FLAG_GENERIC_KERNEL | MAPF_FLAG_NO_CELL_MAP | MAPF_FLAG_JIT_SPECIALIZE | kKernelChoiceFlags);
    e->special = special_id;
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
    // reset scratch per group: sequential path hash set + idx (hash_cap + 2N), parallel path raw32[4N + 2] | vals[4N] | idx[2N] | 16 bytes
    p.scratch_i16 = p.hash_cap + 2 * N + 1;
    if (p.scratch_i16 < 2 * (4 * N + 2) + 4 * N + 2 * N + 8) p.scratch_i16 = 2 * (4 * N + 2) + 4 * N + 2 * N + 8;
    p.scratch_i16 = (p.scratch_i16 + 7) & ~7;  // 16-byte multiple
    const int scratch_i16_alloc = p.scratch_i16;
#ifdef MAPF_CHECK  // test knob of the checking build: DECLARE the groups' draw scratch too small (the allocation keeps its size), so
                   // that tests/test_soak_gpu.py can see a check fire
    if (getenv("MAPF_CHECK_SHRINK")) p.scratch_i16 -= 16;
#endif
    p.ring_stride = (p.lw + 7) & ~7;  // 16-byte rows; <= 16 entries are preloaded whole by the step kernel
    const int rows_bytes = ((G * (H + 2 * kRowPad) * 8) + 15) & ~15;  // kRowPad sentinel rows on either side
    // one 16-byte entry per lane: pair table + two observation-wave tables, then 3 KiB for the record transpose
    const int tab_bytes = 3 * 64 * 16 + 64 * 48;
    const int stage_bytes = ((G * (cte ? (H * W + 5 * N) : N * p.L) * 4) + 15) & ~15;
    const int scratch_bytes = ((G * scratch_i16_alloc * 2) + 15) & ~15;
    p.lds_tab_off = rows_bytes;
    p.lds_stage_off = rows_bytes + tab_bytes;
    p.lds_scratch_off = rows_bytes + tab_bytes + stage_bytes;
    e->lds_bytes = rows_bytes + tab_bytes + stage_bytes + scratch_bytes;
    if (cte) {  // the draw scratch of the two waves of a sampler workgroup of k_cte_step (CteIo::lds_scratch2_off): their lane
                // groups are as narrow as the agent count allows (cte_sampler_lanes)
        // A sampler workgroup stages no observation: its scratch lies in the staging rows whenever they are large enough (a
        // region of its own cost the 32x32 shape a fifth of its workgroups per CU: 19.8 us against 18.0 at 16 384 envs)
        const int need2 = ((2 * cte_sampler_groups(N, lpe) * scratch_i16_alloc * 2) + 15) & ~15;
        if (stage_bytes >= need2) {
            e->cte_scratch2_off = p.lds_stage_off;
        } else {
            e->cte_scratch2_off = e->lds_bytes;
            e->lds_bytes += need2;
        }
    }
    {   // wide groups (N > 16): per-env cell map in LDS instead of the all-pairs walk, when it fits and the lock
        // neighbourhood stays inside the map's border
        const int map_bytes = ((G * (H + 2 * kRowPad) * (W + 2 * kRowPad) * 4) + 15) & ~15;
        if (!cte && lpe >= 32 && c.lock_nearby_manhattan <= kRowPad && !(c.flags & MAPF_FLAG_NO_CELL_MAP) &&
            e->lds_bytes + map_bytes <= 64 * 1024) {
            e->use_map = 1;
            e->lds_map_off = e->lds_bytes;
            e->lds_bytes += map_bytes;
        }
        // the wide specialisations are compiled for the cell-map path only (KFixed::kMapAlways)
        if (e->special && N > 16 && !e->use_map) e->special = 0;
    }
    // k_step3's aux wave stages the info rows and counters of its envs in 2 KiB of its own (no cell map at these widths:
    // lds_map_off is free)
    if (e->three_wave) {  // (only then: at 8 two-wave workgroups per CU -- 16 384 envs of the c3 shape -- these 2 KiB are the
                          // difference between fitting the CU's 160 KiB of LDS and a second round of workgroups)
        e->lds_map_off = e->lds_bytes;
        e->lds_bytes += 3072;  // + 1 KiB: the observation wave's goal-delta table (obs3_wave)
        // grids whose rows carry sentinel columns: the bit rows of the prepared observation (obs3_wave_prepared)
        // (16-lane groups, windows up to 7 x 7.  Groups of 4 and 8 lanes: measured slower with the rows -- c3 5.63 against
        //  5.42 us staggered, c2 4.08 against 3.73 -- their table walks are short and the aux wave's pair pass is cheap)
        if (e->col_pad && e->mask_w <= 64 && lpe == 16 && !(c.flags & MAPF_FLAG_NO_BIT_ROWS)) {
            e->obs_prepared = 1;
            e->lds_bytes += obs_rows_lds_bytes(G, H);
        }
    }
    if (finite_sampled) {  // the sampler workgroups of a k_step launch have their own LDS layout
        const int need = (step_threads(lpe) / 64) * sampler_lds_bytes_per_wave(G, p.scratch_i16);
        if (e->lds_bytes < need) e->lds_bytes = need;
    }
    if (e->lds_bytes > 64 * 1024) {
        delete e;
        return fail(nullptr, MAPF_ERR_CONFIG, "config needs more than 64 KiB of LDS per wavefront");
    }
    // 64-lane groups (one env per wavefront) whose configuration allows the cell-map path step on k_stepw: one env per
    // three-wave workgroup, bit rows instead of the word-per-cell map (mapf_kernels.inl).  Reset / observe / fused launches
    // keep the layout above.
    if (e->use_map && lpe == 64 && !(c.flags & MAPF_FLAG_TWO_WAVE_WIDE)) {
        e->wide3 = 1;
        e->wide_lds_bytes = wide_lds_bytes(H, N * p.L, scratch_i16_alloc);
        e->sampler_blocks = 0;
        if (finite_sampled) {
            e->sampler_blocks = sampler_blocks_for(B, 3);
            const int need = 3 * sampler_lds_bytes_per_wave(1, p.scratch_i16);
            if (e->wide_lds_bytes < need) e->wide_lds_bytes = need;
        }
    }
    if (cte) {
        // Fused launches keep the state in registers and write T rows per env: the fewer lanes per env, the fewer waves
        // do the same stores, so they take the narrowest group that leaves 256 workgroups (same sweep, T = 100, us per
        // step at 8 / 16 / 32 lanes: 2048 envs 2.5 / 2.9 / 2.8, 4096 envs 3.3 / 3.4 / 3.9, 8192 envs 8.4 / 8.6 / 8.9;
        // 16x16 x 4 agents at 4 / 8 / 16 lanes: 8192 envs 2.26 / 2.45 / 3.17, 32 768 envs 6.8 / 8.2 / 13.0).
        int lm = c.lanes_per_env ? c.lanes_per_env : pick_lpe(N);
        if (!c.lanes_per_env) {
            while (lm < 64 && !cte_fits(lm)) lm <<= 1;
            while (lm < 64 && (int64_t)B * lm / 64 < 256) lm <<= 1;
        }
        const int Gm = 64 / lm;
        auto &m = e->cte_many;
        m.lpe = lm;
        m.blocks = (B + Gm - 1) / Gm;
        m.tab_off = ((Gm * (H + 2 * kRowPad) * 8) + 15) & ~15;
        m.stage_off = m.tab_off + tab_bytes;
        m.scratch_off = m.stage_off + (((Gm * (H * W + 5 * N) * 4) + 15) & ~15);
        m.lds_bytes = m.scratch_off + (((Gm * scratch_i16_alloc * 2) + 15) & ~15);
        if (m.lds_bytes > 64 * 1024) {
            delete e;
            return fail(nullptr, MAPF_ERR_CONFIG, "config needs more than 64 KiB of LDS per wavefront");
        }
    }

    const int rc = alloc_device_state(e);
    if (rc != MAPF_OK) {
        g_create_error = e->err;
        mapf_destroy(e);
        return rc;
    }
    if (c.flags & MAPF_FLAG_JIT_SPECIALIZE) {
        DeviceScope scope(c.device);
        jit_specialize(e);
        if (jit_sliced && !e->jit_step) {  // no compiled kernel after all: the runtime-config kernels of the same launch shape
            if (rt_sliced) {
                e->rt_sliced = 1;  // (the grid, the LDS and the three-wave decision were planned for a sliced draw already)
                e->jit_note += " [runtime-config kernels, sliced background draw]";
            } else {
                plan_grid(false);
                e->jit_note += " [runtime-config kernels, background draw in sampler workgroups]";
            }
        }
    }
    *out = e;
    return MAPF_OK;
}

int mapf_jit_status(mapf_handle e, const char **why) {
    if (!e) return 0;
    if (why) *why = e->jit_note.c_str();
    return e->jit_step != nullptr;
}

static int alloc_device_state(mapf_engine *e) {
    const mapf_config &c = e->cfg;
    Params &p = e->p;
    const int B = p.B, N = p.N, H = p.H;
    ON_DEVICE(e);
    const size_t BN = (size_t)B * N;
    e->bn8 = agent_plane_stride(B, N);
    HIP_TRY(e, hipMalloc(&e->d_agents, agent_state_bytes(e->bn8)));
    // env scalars [B][16] followed by the next-episode placement slots [B][N] (slots_of(): the step kernel reaches
    // them from its preloaded arguments)
    // ... and by the env streams and free-cell counts (streams_of / free_counts_of)
    const size_t scal_bytes = (size_t)B * kScalInts * sizeof(int), slot_bytes = BN * sizeof(uint32_t);
    HIP_TRY(e, hipMalloc(&e->d_scal, scal_block_bytes(B, N)));
    e->d_rng = streams_of(e->d_scal, B, N);
    e->d_n_free = free_counts_of(e->d_scal, B, N);
    HIP_TRY(e, hipMalloc(&e->d_vis_rng, (size_t)B * 6 * sizeof(uint64_t)));
    HIP_TRY(e, hipMalloc(&e->d_jump_c, (size_t)B * 64 * sizeof(uint64_t)));
    HIP_TRY(e, hipMemset(e->d_jump_c, 0, (size_t)B * 64 * sizeof(uint64_t)));
    HIP_TRY(e, hipMalloc(&e->d_stage_vals, (size_t)B * stage_dwords(N) * sizeof(uint32_t)));
    HIP_TRY(e, hipMemset(e->d_stage_vals, 0, (size_t)B * stage_dwords(N) * sizeof(uint32_t)));
    p.stage_vals = e->d_stage_vals;
    HIP_TRY(e, hipMalloc(&e->d_ring, BN * p.ring_stride * sizeof(int16_t)));
    HIP_TRY(e, hipMalloc(&e->d_rows, (size_t)B * H * sizeof(uint64_t)));
    HIP_TRY(e, hipMalloc(&e->d_free_cells, (size_t)B * p.HW * sizeof(uint16_t)));
    HIP_TRY(e, hipMalloc(&e->d_free_rank, (size_t)B * p.HW * sizeof(uint16_t)));
    HIP_TRY(e, hipMalloc(&e->d_err, 8 * sizeof(int)));  // [0..3] the latched error record, [4..6] mapf_assign_new_goal's result
    HIP_TRY(e, hipMemset(e->d_agents, 0, agent_state_bytes(e->bn8)));
    HIP_TRY(e, hipMemset(e->d_scal, 0, scal_bytes));
    HIP_TRY(e, hipMemset(reinterpret_cast<char *>(e->d_scal) + scal_bytes, 0xFF, slot_bytes));  // kSlotInvalid
    HIP_TRY(e, hipMemset(e->d_vis_rng, 0, (size_t)B * 6 * sizeof(uint64_t)));
    HIP_TRY(e, hipMemset(e->d_ring, 0, BN * p.ring_stride * sizeof(int16_t)));
    HIP_TRY(e, hipMemset(e->d_rng, 0, (size_t)B * 6 * sizeof(uint64_t)));
    HIP_TRY(e, hipMemset(e->d_err, 0, 8 * sizeof(int)));
    HIP_TRY(e, hipMalloc(&e->d_ep_acc, (size_t)B * MAPF_NUM_EPISODE_ACC * sizeof(int)));
    HIP_TRY(e, hipMemset(e->d_ep_acc, 0, (size_t)B * MAPF_NUM_EPISODE_ACC * sizeof(int)));
    p.ep_acc = e->d_ep_acc;
    p.next_sg = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(e->d_scal) + scal_bytes);
    p.vis_rng = e->d_vis_rng;
    p.rng = e->d_rng;
    p.free_cells = e->d_free_cells;
    p.free_rank = e->d_free_rank;
    p.n_free = e->d_n_free;
    p.err = e->d_err;
#ifdef MAPF_STAMPS
    HIP_TRY(e, hipMalloc(&e->d_dbg, (size_t)(e->blocks + e->sampler_blocks) * kDbgRow * sizeof(unsigned long long)));
    HIP_TRY(e, hipMemset(e->d_dbg, 0, (size_t)(e->blocks + e->sampler_blocks) * kDbgRow * sizeof(unsigned long long)));
#endif
    p.dbg = e->d_dbg;
    HIP_TRY(e, hipMalloc(&e->d_params, sizeof(Params)));
    HIP_TRY(e, hipMemcpy(e->d_params, &p, sizeof(Params), hipMemcpyHostToDevice));
    return MAPF_OK;
}

int mapf_destroy(mapf_handle e) {
    if (!e) return MAPF_OK;
    // best effort: a failing free at teardown is reported through the return code, the handle goes away regardless
    DeviceScope scope(e->cfg.device);  // the caller's current device is restored when this returns (e.g. from __del__)
    hipError_t first = scope.status;
    void *const bufs[] = {e->d_jump_c, e->d_agents, e->d_scal, e->d_ring, e->d_rows, e->d_free_cells, e->d_free_rank, e->d_err, e->d_ep_acc, e->d_vis_rng, e->d_stage_vals, e->d_params, e->d_dbg};
    for (void *b : bufs) {
        const hipError_t rc = hipFree(b);
        if (first == hipSuccess) first = rc;
    }
    delete e;
    return first == hipSuccess ? MAPF_OK : MAPF_ERR_HIP;
}

int mapf_set_grids(mapf_handle e, const uint8_t *grids, int32_t shared) {
    if (!e || !grids) return fail(e, MAPF_ERR_CONFIG, "null argument");
    const int B = e->p.B, H = e->p.H, W = e->p.W, HW = e->p.HW, N = e->p.N;
    std::vector<uint64_t> rows((size_t)B * H);
    std::vector<uint16_t> cells((size_t)B * HW, 0), rank((size_t)B * HW, 0);
    std::vector<int> nfree(B);
    // low sentinel bits when they fit (Io::col_pad), ones above the grid: out-of-bounds columns read as obstacle
    const int pad = e->col_pad;
    const uint64_t hi = (W + pad >= 64 ? 0ull : (~0ull << (W + pad))) | ((1ull << pad) - 1ull);
    for (int b = 0; b < B; b++) {
        const uint8_t *g = grids + (shared ? 0 : (size_t)b * HW);
        int f = 0;
        for (int r = 0; r < H; r++) {
            uint64_t bits = hi;
            for (int c = 0; c < W; c++) {
                if (g[r * W + c] != 0) {
                    bits |= 1ull << (c + pad);
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
    ON_DEVICE(e);
    HIP_TRY(e, hipDeviceSynchronize());  // (no launch of this handle may still be reading the tables that change below)
    HIP_TRY(e, hipMemcpy(e->d_rows, rows.data(), rows.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->d_free_cells, cells.data(), cells.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->d_free_rank, rank.data(), rank.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->d_n_free, nfree.data(), nfree.size() * sizeof(int), hipMemcpyHostToDevice));
    e->h_rows = rows;
    e->grids_set = true;
    // visible streams back under the slots that are about to be voided, pass bits of every agent for the new rows: on
    // the device, stream-ordered behind the copies above (k_after_set_grids), then the slots
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_after_set_grids, dim3((unsigned)(((size_t)B * N + 255) / 256)), dim3(256), 0, (hipStream_t)0, e->d_agents,
                       e->d_rows, B, N, H, W, e->col_pad, e->p.next_sg, e->d_vis_rng, e->d_rng);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, invalidate_slots(e));
    HIP_TRY(e, hipDeviceSynchronize());
    return MAPF_OK;
}

int mapf_set_rng_state(mapf_handle e, const uint64_t *rng_words) {
    if (!e || !rng_words) return fail(e, MAPF_ERR_CONFIG, "null argument");
    ON_DEVICE(e);
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipMemcpy(e->d_rng, rng_words, (size_t)e->p.B * 6 * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(e, invalidate_slots(e));
    return upload_jump_table(e, rng_words);
}

int mapf_get_state(mapf_handle e, mapf_state *out) {
    if (!e || !out) return fail(e, MAPF_ERR_CONFIG, "null argument");
    const int B = e->p.B, N = e->p.N;
    const size_t BN = (size_t)B * N;
    ON_DEVICE(e);
    HIP_TRY(e, hipDeviceSynchronize());
    std::vector<AgentRec> recs;
    {
        const int rc = download_agents(e, recs);
        if (rc != MAPF_OK) return rc;
    }
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
    if (out->counters) {
        HIP_TRY(e, hipMemcpy(out->counters, e->d_scal, (size_t)B * kScalInts * sizeof(int), hipMemcpyDeviceToHost));
        for (int b = 0; b < B; b++) out->counters[(size_t)b * kScalInts + MAPF_CTR_MAY_FINISH] = 0;  // engine-internal
    }
    if (out->rng_words) {
        // the env's visible stream state: while a pre-drawn placement is pending, d_rng already holds the state after
        // that draw and the visible one is kept in d_vis_rng (mapf_kernels.inl: kSlotInvalid)
        HIP_TRY(e, hipMemcpy(out->rng_words, e->d_rng, (size_t)B * 6 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        std::vector<uint64_t> vis((size_t)B * 6);
        std::vector<uint32_t> slots(BN);
        HIP_TRY(e, hipMemcpy(vis.data(), e->d_vis_rng, vis.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIP_TRY(e, hipMemcpy(slots.data(), e->p.next_sg, slots.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int b = 0; b < B; b++) {
            // word 0 tells: kSlotInvalid = no background draw has touched the stream; staged or valid = it is advanced
            const bool pending = slots[(size_t)b * N] != kSlotInvalid;
            if (pending) memcpy(out->rng_words + (size_t)b * 6, vis.data() + (size_t)b * 6, 6 * sizeof(uint64_t));
        }
    }
    if (out->distance_ring) {  // ABI layout [B][lw][N], slot = history row index mod lw
        const int lw = e->p.lw, rs = e->p.ring_stride;
        memset(out->distance_ring, 0, BN * lw * sizeof(int16_t));
        if (lw <= 16) {  // 16 x uint8 shift register in the agent record: byte k = distance k steps ago
            std::vector<int> scal((size_t)B * kScalInts);
            HIP_TRY(e, hipMemcpy(scal.data(), e->d_scal, scal.size() * sizeof(int), hipMemcpyDeviceToHost));
            for (int b = 0; b < B; b++) {
                const int t = scal[(size_t)b * kScalInts + MAPF_CTR_HIST_ROWS];
                for (int n = 0; n < N; n++) {
                    const uint32_t *d = recs[(size_t)b * N + n].dist;
                    for (int k = 0; k < lw && k < t; k++) {
                        const int slot = ((t - 1 - k) % lw + lw) % lw;
                        out->distance_ring[((size_t)b * lw + slot) * N + n] = (int16_t)((d[k >> 2] >> ((k & 3) * 8)) & 0xFFu);
                    }
                }
            }
        } else {  // device layout [B][N][ring_stride]
            std::vector<int16_t> ring(BN * rs);
            HIP_TRY(e, hipMemcpy(ring.data(), e->d_ring, ring.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
            for (int b = 0; b < B; b++)
                for (int n = 0; n < N; n++)
                    for (int k = 0; k < lw; k++)
                        out->distance_ring[((size_t)b * lw + k) * N + n] = ring[((size_t)b * N + n) * rs + k];
        }
    }
    return MAPF_OK;
}

int mapf_set_state(mapf_handle e, const mapf_state *in) {
    if (!e || !in) return fail(e, MAPF_ERR_CONFIG, "null argument");
    const int B = e->p.B, N = e->p.N, H = e->p.H, W = e->p.W;
    const size_t BN = (size_t)B * N;
    ON_DEVICE(e);
    HIP_TRY(e, hipDeviceSynchronize());
    const bool ring_in_rec = e->p.lw <= 16;
    const bool touch_recs = in->positions || in->goals || in->starts || in->reached || in->completed_once ||
                            in->pressure_prev || in->lock_history || (in->distance_ring && ring_in_rec);
    if (touch_recs) {
        std::vector<AgentRec> recs;
        {
            const int rc = download_agents(e, recs);
            if (rc != MAPF_OK) return rc;
        }
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
        // The move rule keeps agents on distinct free cells and goals distinct (MA-env:502-526, :284-304); everything
        // downstream (time-indexed occupancy, cell maps, respawn ranks) relies on it, and the reference's own
        // behaviour with two agents in one cell is an artefact of its owner maps (the collision penalty MA-env:658-666
        // is dead code there too).  Injected states must therefore satisfy the invariant.
        if (in->positions || in->goals) {
            const int pad = e->col_pad;
            std::vector<uint32_t> seen;
            for (int b = 0; b < B; b++) {
                for (int pass = 0; pass < 2; pass++) {
                    if (pass == 0 ? !in->positions : !in->goals) continue;
                    seen.clear();
                    for (int n = 0; n < N; n++) {
                        const AgentRec &r = recs[(size_t)b * N + n];
                        const uint32_t cell = pass == 0 ? (r.w0 & 0xFFFFu) : (r.w0 >> 16);
                        for (uint32_t o : seen)
                            if (o == cell)
                                return fail(e, MAPF_ERR_CONFIG, pass == 0 ? "two agents of an env on the same cell"
                                                                          : "two agents of an env with the same goal cell");
                        seen.push_back(cell);
                        if (pass == 0 && e->grids_set &&
                            ((e->h_rows[(size_t)b * H + (cell >> 8)] >> ((cell & 255u) + pad)) & 1ull))
                            return fail(e, MAPF_ERR_CONFIG, "agent position on an obstacle cell");
                    }
                }
            }
        }
        if (in->distance_ring && ring_in_rec) {
            const int lw = e->p.lw;
            std::vector<int> scal((size_t)B * kScalInts);
            if (in->counters) memcpy(scal.data(), in->counters, scal.size() * sizeof(int));
            else HIP_TRY(e, hipMemcpy(scal.data(), e->d_scal, scal.size() * sizeof(int), hipMemcpyDeviceToHost));
            for (int b = 0; b < B; b++) {
                const int t = scal[(size_t)b * kScalInts + MAPF_CTR_HIST_ROWS];
                for (int n = 0; n < N; n++) {
                    uint32_t *d = recs[(size_t)b * N + n].dist;
                    d[0] = d[1] = d[2] = d[3] = 0;
                    for (int k = 0; k < lw && k < t; k++) {
                        const int slot = ((t - 1 - k) % lw + lw) % lw;
                        const uint32_t v = (uint32_t)in->distance_ring[((size_t)b * lw + slot) * N + n] & 0xFFu;
                        d[k >> 2] |= v << ((k & 3) * 8);
                    }
                }
            }
        }
        {
            const int rc = upload_agents(e, recs);
            if (rc != MAPF_OK) return rc;
        }
    }
    if (in->counters)
        HIP_TRY(e, hipMemcpy(e->d_scal, in->counters, (size_t)B * kScalInts * sizeof(int), hipMemcpyHostToDevice));
    if (in->rng_words) {
        HIP_TRY(e, hipMemcpy(e->d_rng, in->rng_words, (size_t)B * 6 * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(e, invalidate_slots(e));
        const int rc = upload_jump_table(e, in->rng_words);
        if (rc != MAPF_OK) return rc;
    }
    HIP_TRY(e, force_may_finish(e));
    if (in->distance_ring && !ring_in_rec) {
        const int lw = e->p.lw, rs = e->p.ring_stride;
        std::vector<int16_t> ring(BN * rs, 0);
        for (int b = 0; b < B; b++)
            for (int n = 0; n < N; n++)
                for (int k = 0; k < lw; k++)
                    ring[((size_t)b * N + n) * rs + k] = in->distance_ring[((size_t)b * lw + k) * N + n];
        HIP_TRY(e, hipMemcpy(e->d_ring, ring.data(), ring.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    }
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
    if (e->cte) return fail(e, MAPF_ERR_STATE, "handle was created with MAPF_FLAG_SINGLE_AGENT: use mapf_cte_reset");
    Io io;
    memset(&io, 0, sizeof io);
    io.agents = e->d_agents;
    io.scal = e->d_scal;
    io.dist_ring = e->d_ring;
    io.grid_rows = e->d_rows;
    io.B = e->p.B;
    io.H = e->p.H;
    io.W = e->p.W;
    io.col_pad = e->col_pad;
    io.bn8 = e->bn8;
    io.use_map = e->use_map;
    io.lds_map_off = e->lds_map_off;
    io.eps_floor = e->p.eps_floor;
    io.steps_per_episode = e->p.steps_per_episode;
    io.den_r = e->p.den_r;
    io.den_c = e->p.den_c;
    io.lds_tab_off = e->p.lds_tab_off;
    io.lds_stage_off = e->p.lds_stage_off;
    io.lds_scratch_off = e->p.lds_scratch_off;
    io.env_mask = env_mask;
    io.obs = obs;
    ON_DEVICE(e);
    LAUNCH_TRY(e, dispatch(KIND_RESET, e, io, (hipStream_t)stream));
    return MAPF_OK;
}

static int step_impl(mapf_handle e, const int8_t *actions, const uint8_t *env_mask, float *obs, float *rewards,
                     uint8_t *terminated, uint8_t *truncated, float *info_all, uint8_t *info_agent, float *final_obs,
                     int32_t auto_reset, void *stream) {
    if (!e || !actions) return fail(e, MAPF_ERR_CONFIG, "null argument");
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_step");
    if (e->cte) return fail(e, MAPF_ERR_STATE, "handle was created with MAPF_FLAG_SINGLE_AGENT: use mapf_cte_step");
    Io io;
    memset(&io, 0, sizeof io);
    io.agents = e->d_agents;
    io.scal = e->d_scal;
    io.dist_ring = e->d_ring;
    io.grid_rows = e->d_rows;
    io.B = e->p.B;
    io.H = e->p.H;
    io.W = e->p.W;
    io.col_pad = e->col_pad;
    io.bn8 = e->bn8;
    // (bits 1, 2: small groups only, where bit 0 is never looked at: bit rows in use; observation wave walks the table all the same)
    io.use_map = e->use_map | (e->obs_prepared << 1) | ((e->cfg.flags & MAPF_FLAG_TABLE_WALK_OBS) ? 4 : 0);
    io.lds_map_off = e->lds_map_off;
    io.eps_floor = e->p.eps_floor;
    io.steps_per_episode = e->p.steps_per_episode;
    io.den_r = e->p.den_r;
    io.den_c = e->p.den_c;
    io.lds_tab_off = e->p.lds_tab_off;
    io.lds_stage_off = e->p.lds_stage_off;
    io.lds_scratch_off = e->p.lds_scratch_off;
    io.actions = actions;
    io.obs = obs;
    io.rewards = rewards;
    io.terminated = terminated;
    io.truncated = truncated;
    io.info_all = info_all;
    io.info_agent = info_agent;
    io.final_obs = final_obs;
    io.auto_reset = auto_reset;
    io.env_mask = env_mask;
    io.stage_vals = e->d_stage_vals;
    io.free_cells = e->d_free_cells;
    io.free_rank = e->d_free_rank;
    io.vis_rng = e->d_vis_rng;
    io.jump_c = e->d_jump_c;
    ON_DEVICE(e);
    LAUNCH_TRY(e, dispatch(KIND_STEP, e, io, (hipStream_t)stream));
    return MAPF_OK;
}

int mapf_step(mapf_handle e, const int8_t *actions, float *obs, float *rewards, uint8_t *terminated, uint8_t *truncated,
              float *info_all, uint8_t *info_agent, float *final_obs, int32_t auto_reset, void *stream) {
    return step_impl(e, actions, nullptr, obs, rewards, terminated, truncated, info_all, info_agent, final_obs, auto_reset, stream);
}

int mapf_bind_outputs(mapf_handle e, float *obs, float *rewards, uint8_t *terminated, uint8_t *truncated, float *info_all,
                      uint8_t *info_agent) {
    if (!e) return MAPF_ERR_CONFIG;
    e->b_obs = obs; e->b_rewards = rewards; e->b_terminated = terminated; e->b_truncated = truncated;
    e->b_info_all = info_all; e->b_info_agent = info_agent;
    e->bound = true;
    return MAPF_OK;
}

int mapf_step_bound(mapf_handle e, const int8_t *actions, int32_t auto_reset, void *stream) {
    if (!e || !e->bound) return fail(e, MAPF_ERR_STATE, "mapf_bind_outputs must be called before mapf_step_bound");
    return step_impl(e, actions, nullptr, e->b_obs, e->b_rewards, e->b_terminated, e->b_truncated, e->b_info_all, e->b_info_agent,
                     nullptr, auto_reset, stream);
}

int mapf_step_masked(mapf_handle e, const int8_t *actions, const uint8_t *env_mask, float *obs, float *rewards,
                     uint8_t *terminated, uint8_t *truncated, float *info_all, uint8_t *info_agent, float *final_obs,
                     int32_t auto_reset, void *stream) {
    if (!env_mask) return fail(e, MAPF_ERR_CONFIG, "null env_mask (use mapf_step)");
    return step_impl(e, actions, env_mask, obs, rewards, terminated, truncated, info_all, info_agent, final_obs, auto_reset, stream);
}

static int step_many_impl(mapf_handle e, int32_t T, const int8_t *actions, const ManyPolicy &pol, float *obs,
                          int32_t obs_mode, float *rewards, uint8_t *terminated, uint8_t *truncated, float *info_all,
                          uint8_t *info_agent, void *stream) {
    Io io;
    memset(&io, 0, sizeof io);
    io.agents = e->d_agents;
    io.scal = e->d_scal;
    io.dist_ring = e->d_ring;
    io.grid_rows = e->d_rows;
    io.B = e->p.B;
    io.H = e->p.H;
    io.W = e->p.W;
    io.col_pad = e->col_pad;
    io.bn8 = e->bn8;
    io.use_map = e->use_map;
    io.lds_map_off = e->lds_map_off;
    io.eps_floor = e->p.eps_floor;
    io.steps_per_episode = e->p.steps_per_episode;
    io.den_r = e->p.den_r;
    io.den_c = e->p.den_c;
    io.lds_tab_off = e->p.lds_tab_off;
    io.lds_stage_off = e->p.lds_stage_off;
    io.lds_scratch_off = e->p.lds_scratch_off;
    io.actions = actions;
    io.obs = obs;
    io.rewards = rewards;
    io.terminated = terminated;
    io.truncated = truncated;
    io.info_all = info_all;
    io.info_agent = info_agent;
    io.auto_reset = 1;
    ON_DEVICE(e);
    LAUNCH_TRY(e, dispatch_many(e, io, T, obs_mode, pol, (hipStream_t)stream));
    return MAPF_OK;
}

int mapf_step_many(mapf_handle e, int32_t T, const int8_t *actions, float *obs, int32_t obs_mode, float *rewards,
                   uint8_t *terminated, uint8_t *truncated, float *info_all, uint8_t *info_agent, void *stream) {
    if (!e || !actions || T < 1) return fail(e, MAPF_ERR_CONFIG, "null argument or T < 1");
    if (obs_mode < 0 || obs_mode > 2 || (obs_mode != 0 && !obs)) return fail(e, MAPF_ERR_CONFIG, "bad obs_mode / obs");
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_step_many");
    if (e->cte) return fail(e, MAPF_ERR_STATE, "mapf_step_many is not available for the single-agent variant");
    ManyPolicy pol;
    memset(&pol, 0, sizeof pol);
    return step_many_impl(e, T, actions, pol, obs, obs_mode, rewards, terminated, truncated, info_all, info_agent, stream);
}

int mapf_step_many_sampled(mapf_handle e, int32_t T, const float *obs_in, uint64_t seed, int8_t *actions_out, float *obs,
                           float *rewards, uint8_t *terminated, uint8_t *truncated, float *info_all, uint8_t *info_agent,
                           void *stream) {
    if (!e || !obs_in || !actions_out || !obs || T < 1) return fail(e, MAPF_ERR_CONFIG, "null argument or T < 1");
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_step_many_sampled");
    if (e->cte) return fail(e, MAPF_ERR_STATE, "mapf_step_many_sampled is not available for the single-agent variant");
    if (!(e->p.flags & MAPF_FLAG_ACTION_MASK))
        return fail(e, MAPF_ERR_CONFIG, "masked sampling needs the action mask in the observation (include_action_mask_in_obs)");
    ManyPolicy pol;
    pol.obs_in = obs_in;
    pol.actions_out = actions_out;
    pol.seed = seed;
    pol.mask_off = e->p.L - 5;  // the mask is the tail of the observation row (MA-env:306-328)
    return step_many_impl(e, T, actions_out /* unused as input */, pol, obs, 2, rewards, terminated, truncated, info_all,
                          info_agent, stream);
}

// single-step launches of the single-agent env with sampled placements: one sampler workgroup per 64 envs at the front of the
// grid (k_cte_step: it pre-draws the next placement of the envs that have none and cannot end their episode in this launch)
static int cte_sampler_blocks(const mapf_engine *e) {  // (both waves of a sampler workgroup work: 128 envs each)
    return (e->cfg.flags & MAPF_FLAG_DETERMINISTIC) ? 0 : (e->p.B + 127) / 128;
}
static CteIo make_cte_io(const mapf_engine *e, bool fused = false) {
    CteIo io;
    memset(&io, 0, sizeof io);
    io.agents = e->d_agents;
    io.scal = e->d_scal;
    io.grid_rows = e->d_rows;
    io.B = e->p.B;
    io.H = e->p.H;
    io.W = e->p.W;
    io.col_pad = e->col_pad;
    io.bn8 = e->bn8;
    io.steps_per_episode = e->p.steps_per_episode;
    io.lds_tab_off = fused ? e->cte_many.tab_off : e->p.lds_tab_off;
    io.lds_stage_off = fused ? e->cte_many.stage_off : e->p.lds_stage_off;
    io.lds_scratch_off = fused ? e->cte_many.scratch_off : e->p.lds_scratch_off;
    io.lds_scratch2_off = e->cte_scratch2_off;
    io.blocking_penalty = e->cte_blocking_penalty;
    io.move_after_goal_penalty = e->cte_move_after_goal_penalty;
    return io;
}

static hipError_t launch_cte(const mapf_engine *e, const CteIo &io, bool step, hipStream_t s, CteMany many = CteMany{1, 2}) {
#ifndef MAPF_NO_CTE_KERNELS
    LaunchPlan lp = plan_of(e);
    const bool fused = step && many.T > 1;
    if (fused) { lp.blocks = e->cte_many.blocks; lp.lds_bytes = e->cte_many.lds_bytes; }
#define MAPF_CASE(L) \
    case L:          \
        return launch_cte_##L(lp, io, step, s, many);
    switch (fused ? e->cte_many.lpe : e->lpe) { MAPF_FOR_LPE(MAPF_CASE) }
#undef MAPF_CASE
#endif
    return hipErrorInvalidValue;
}

int mapf_cte_configure(mapf_handle e, double blocking_penalty, double move_after_goal_penalty) {
    if (!e || !e->cte) return fail(e, MAPF_ERR_STATE, "not a MAPF_FLAG_SINGLE_AGENT handle");
    e->cte_blocking_penalty = blocking_penalty;
    e->cte_move_after_goal_penalty = move_after_goal_penalty;
    return MAPF_OK;
}

int mapf_cte_reset(mapf_handle e, const uint8_t *env_mask, float *obs, void *stream) {
    if (!e || !e->cte) return fail(e, MAPF_ERR_STATE, "not a MAPF_FLAG_SINGLE_AGENT handle");
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_cte_reset");
    CteIo io = make_cte_io(e);
    io.env_mask = env_mask;
    io.obs = obs;
    ON_DEVICE(e);
    LAUNCH_TRY(e, launch_cte(e, io, false, (hipStream_t)stream));
    return MAPF_OK;
}

int mapf_cte_step(mapf_handle e, const int8_t *actions, float *obs, double *reward, uint8_t *terminated,
                  uint8_t *truncated, float *info, float *final_obs, int32_t auto_reset, void *stream) {
    if (!e || !e->cte) return fail(e, MAPF_ERR_STATE, "not a MAPF_FLAG_SINGLE_AGENT handle");
    if (!actions) return fail(e, MAPF_ERR_CONFIG, "null argument");
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_cte_step");
    CteIo io = make_cte_io(e);
    io.actions = actions;
    io.obs = obs;
    io.reward = reward;
    io.terminated = terminated;
    io.truncated = truncated;
    io.info = info;
    io.final_obs = final_obs;
    io.auto_reset = auto_reset;
    io.sampler_blocks = cte_sampler_blocks(e);
    ON_DEVICE(e);
    LAUNCH_TRY(e, launch_cte(e, io, true, (hipStream_t)stream));
    return MAPF_OK;
}

int mapf_cte_step_many(mapf_handle e, int32_t T, const int8_t *actions, float *obs, int32_t obs_mode, double *reward,
                       uint8_t *terminated, uint8_t *truncated, float *info, void *stream) {
    if (!e || !e->cte) return fail(e, MAPF_ERR_STATE, "not a MAPF_FLAG_SINGLE_AGENT handle");
    if (!actions || T < 1) return fail(e, MAPF_ERR_CONFIG, "null argument or T < 1");
    if (obs_mode < 0 || obs_mode > 2 || (obs_mode != 0 && !obs)) return fail(e, MAPF_ERR_CONFIG, "bad obs_mode / obs");
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_cte_step_many");
    CteIo io = make_cte_io(e, T > 1);
    io.actions = actions;
    io.obs = obs;
    io.reward = reward;
    io.terminated = terminated;
    io.truncated = truncated;
    io.info = info;
    io.auto_reset = 1;
    ON_DEVICE(e);
    if (T == 1) {  // (the single-step kernel shape: obs_mode 0 = no observation)
        if (obs_mode == 0) io.obs = nullptr;
        io.sampler_blocks = cte_sampler_blocks(e);
        LAUNCH_TRY(e, launch_cte(e, io, true, (hipStream_t)stream));
        return MAPF_OK;
    }
    LAUNCH_TRY(e, launch_cte(e, io, true, (hipStream_t)stream, CteMany{T, obs_mode}));
    return MAPF_OK;
}

int mapf_observe(mapf_handle e, float *obs, void *stream) {
    if (!e || !obs) return fail(e, MAPF_ERR_CONFIG, "null argument");
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_observe");
    if (e->cte) return fail(e, MAPF_ERR_STATE, "mapf_observe is not available for the single-agent variant");
    Io io;
    memset(&io, 0, sizeof io);
    io.agents = e->d_agents;
    io.scal = e->d_scal;
    io.dist_ring = e->d_ring;
    io.grid_rows = e->d_rows;
    io.B = e->p.B;
    io.H = e->p.H;
    io.W = e->p.W;
    io.col_pad = e->col_pad;
    io.bn8 = e->bn8;
    io.use_map = e->use_map;
    io.lds_map_off = e->lds_map_off;
    io.eps_floor = e->p.eps_floor;
    io.steps_per_episode = e->p.steps_per_episode;
    io.den_r = e->p.den_r;
    io.den_c = e->p.den_c;
    io.lds_tab_off = e->p.lds_tab_off;
    io.lds_stage_off = e->p.lds_stage_off;
    io.lds_scratch_off = e->p.lds_scratch_off;
    io.obs = obs;
    ON_DEVICE(e);
    LAUNCH_TRY(e, dispatch(KIND_OBSERVE, e, io, (hipStream_t)stream));
    return MAPF_OK;
}

namespace {
// _assign_new_goal (MA-env:284-304) for ONE agent of ONE env, outside step(): the old goal is cleared, the candidates are the
// free cells (row-major, _free_positions) that hold no agent and no other agent's goal, r = rng.integers(k) on the env's
// stream (no draw when k == 1), new goal = r-th candidate.  One thread: this is the reference's per-call helper, not the
// hot path (inside step() the respawn runs in the step kernels, in agent order).  out: {status, row, col}.
__global__ __launch_bounds__(64) void k_assign_new_goal(uint2 *__restrict__ hot, const Params *__restrict__ pp, int env, int agent,
                                                         int *__restrict__ out) {
    if (threadIdx.x != 0) return;
    const Params &p = *pp;
    const int N = p.N, F = p.n_free[env];
    uint2 *rec = hot + (size_t)env * N;
    const uint16_t *fc = p.free_cells + (size_t)env * p.HW;
    auto is_candidate = [&](uint32_t cell) {
        for (int b = 0; b < N; b++) {
            const uint32_t w0 = rec[b].x;
            if ((w0 & 0xFFFFu) == cell) return false;               // occupied (the agent's own cell included, :292)
            if (b != agent && (w0 >> 16) == cell) return false;     // somebody else's goal (the own old goal was cleared, :288)
        }
        return true;
    };
    int k = 0;
    for (int f = 0; f < F; f++) k += is_candidate(fc[f]) ? 1 : 0;
    if (k == 0) {
        raise_error(p, MAPF_ERR_NO_RESPAWN, env, agent, 0);
        out[0] = MAPF_ERR_NO_RESPAWN;
        return;
    }
    // the env's VISIBLE stream: while a pre-drawn placement is pending it sits in vis_rng (mapf_kernels.inl: kSlotInvalid);
    // that placement was drawn from a state this call leaves behind, so it is voided
    uint32_t *slots = p.next_sg + (size_t)env * N;
    const bool pending = slots[0] != kSlotInvalid;
    Pcg g;
    pcg_load(g, (pending ? p.vis_rng : p.rng) + (size_t)env * 6);
    bool stuck = false;
    const uint32_t r = pcg_bounded(g, (uint32_t)(k - 1), stuck);
    if (stuck) raise_error(p, MAPF_ERR_RNG_GUARD, env, agent, 0);
    pcg_store(g, p.rng + (size_t)env * 6);
    if (pending)
        for (int b = 0; b < N; b++) slots[b] = kSlotInvalid;
    uint32_t left = r, cell = 0;
    for (int f = 0; f < F; f++) {
        if (!is_candidate(fc[f])) continue;
        if (left == 0) { cell = fc[f]; break; }
        left--;
    }
    rec[agent].x = (rec[agent].x & 0xFFFFu) | (cell << 16);
    out[0] = MAPF_OK;
    out[1] = (int)(cell >> 8);
    out[2] = (int)(cell & 255u);
}
}  // namespace

int mapf_assign_new_goal(mapf_handle e, int32_t env, int32_t agent, int16_t *new_goal, void *stream) {
    if (!e || !new_goal) return fail(e, MAPF_ERR_CONFIG, "null argument");
    if (e->cte) return fail(e, MAPF_ERR_STATE, "mapf_assign_new_goal is not available for the single-agent variant");
    if (!e->grids_set) return fail(e, MAPF_ERR_STATE, "mapf_set_grids must be called before mapf_assign_new_goal");
    if (env < 0 || env >= e->p.B || agent < 0 || agent >= e->p.N) return fail(e, MAPF_ERR_CONFIG, "env / agent index out of range");
    ON_DEVICE(e);
    int *d_out = e->d_err + 4;  // (three ints behind the error record, same allocation)
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_assign_new_goal, dim3(1), dim3(64), 0, (hipStream_t)stream, e->d_agents, e->d_params, (int)env, (int)agent, d_out);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
    int res[3] = {0, 0, 0};
    HIP_TRY(e, hipMemcpy(res, d_out, sizeof res, hipMemcpyDeviceToHost));
    if (res[0] != MAPF_OK) return fail(e, res[0], "No valid cell available for lifelong goal reassignment.");
    new_goal[0] = (int16_t)res[1];
    new_goal[1] = (int16_t)res[2];
    return MAPF_OK;
}

int mapf_get_episode_stats(mapf_handle e, int64_t *out, int32_t reset) {
    if (!e || !out) return fail(e, MAPF_ERR_CONFIG, "null argument");
    const size_t n = (size_t)e->p.B * MAPF_NUM_EPISODE_ACC;
    std::vector<int> acc(n);
    ON_DEVICE(e);
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipMemcpy(acc.data(), e->d_ep_acc, n * sizeof(int), hipMemcpyDeviceToHost));
    for (int k = 0; k < MAPF_NUM_EPISODE_ACC; k++) out[k] = 0;
    for (size_t i = 0; i < n; i++) out[i % MAPF_NUM_EPISODE_ACC] += acc[i];
    if (reset) HIP_TRY(e, hipMemset(e->d_ep_acc, 0, n * sizeof(int)));
    return MAPF_OK;
}

namespace {
// sums of the per-env episode accumulators, one workgroup: every thread adds up a strided share of the envs, the
// workgroup's partial sums meet in LDS (64-bit adds)
__global__ __launch_bounds__(1024) void k_episode_sums(const int *__restrict__ acc, int B, long long *__restrict__ out) {
    __shared__ unsigned long long part[MAPF_NUM_EPISODE_ACC];
    if (threadIdx.x < MAPF_NUM_EPISODE_ACC) part[threadIdx.x] = 0ull;
    __syncthreads();
    long long mine[MAPF_NUM_EPISODE_ACC];
#pragma unroll
    for (int k = 0; k < MAPF_NUM_EPISODE_ACC; k++) mine[k] = 0;
    for (int env = (int)threadIdx.x; env < B; env += (int)blockDim.x) {
        const int4 *row = reinterpret_cast<const int4 *>(acc + (size_t)env * MAPF_NUM_EPISODE_ACC);
        const int4 a = row[0], b = row[1], c = row[2];
        mine[0] += a.x; mine[1] += a.y; mine[2] += a.z; mine[3] += a.w;
        mine[4] += b.x; mine[5] += b.y; mine[6] += b.z; mine[7] += b.w;
        mine[8] += c.x; mine[9] += c.y; mine[10] += c.z; mine[11] += c.w;
    }
#pragma unroll
    for (int k = 0; k < MAPF_NUM_EPISODE_ACC; k++) atomicAdd(&part[k], (unsigned long long)mine[k]);
    __syncthreads();
    if (threadIdx.x < MAPF_NUM_EPISODE_ACC) out[threadIdx.x] = (long long)part[threadIdx.x];
}
static_assert(MAPF_NUM_EPISODE_ACC == 12, "k_episode_sums reads a row as three int4");
}  // namespace

int mapf_episode_stats_async(mapf_handle e, int64_t *out, void *stream) {
    if (!e || !out) return fail(e, MAPF_ERR_CONFIG, "null argument");
    ON_DEVICE(e);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_episode_sums, dim3(1), dim3(1024), 0, (hipStream_t)stream, e->d_ep_acc, e->p.B, reinterpret_cast<long long *>(out));
    HIP_TRY(e, hipGetLastError());
    return MAPF_OK;
}

int mapf_poll_error(mapf_handle e, void *stream, int32_t *env, int32_t *agent, int32_t *value) {
    if (!e) return MAPF_ERR_CONFIG;
    int rec[4] = {0, 0, 0, 0};
    ON_DEVICE(e);
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
        else if (rec[0] == MAPF_ERR_INTERNAL)
            snprintf(buf, sizeof buf, "checking build: index %d left its LDS region at site %d (env %d)", rec[3], rec[2], rec[1]);
        else if (rec[0] == MAPF_ERR_RNG_GUARD)
            snprintf(buf, sizeof buf, "bounded draw rejected 4096 times in a row: RNG state of env %d is corrupt", rec[1]);
        else
            snprintf(buf, sizeof buf, "device error %d in env %d", rec[0], rec[1]);
        e->err = buf;
    }
    return rec[0];
}

int mapf_debug_stamps(mapf_handle e, uint64_t *out, int32_t max_words) {
    if (!e || !out) return MAPF_ERR_CONFIG;
#ifdef MAPF_STAMPS
    ON_DEVICE(e);
    HIP_TRY(e, hipDeviceSynchronize());
    size_t n = (size_t)(e->blocks + e->sampler_blocks) * kDbgRow;  // env workgroups first, then the sampler workgroups
    if ((size_t)max_words < n) n = (size_t)max_words;
    HIP_TRY(e, hipMemcpy(out, e->d_dbg, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return (int)n;
#else
    (void)max_words;
    return fail(e, MAPF_ERR_STATE, "library was built without -DMAPF_STAMPS");
#endif
}

int mapf_debug_slots(mapf_handle e, uint32_t *slots, uint32_t *stage, uint64_t *vis) {
    if (!e) return MAPF_ERR_CONFIG;
    ON_DEVICE(e);
    HIP_TRY(e, hipDeviceSynchronize());
    const size_t BN = (size_t)e->p.B * e->p.N;
    if (slots) HIP_TRY(e, hipMemcpy(slots, e->p.next_sg, BN * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (stage)
        HIP_TRY(e, hipMemcpy(stage, e->d_stage_vals, (size_t)e->p.B * stage_dwords(e->p.N) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (vis) HIP_TRY(e, hipMemcpy(vis, e->d_vis_rng, (size_t)e->p.B * 6 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MAPF_OK;
}

int mapf_launch_info(mapf_handle e, int32_t *blocks, int32_t *threads, int32_t *lds_bytes, int32_t *lanes_per_env) {
    if (!e) return MAPF_ERR_CONFIG;
    if (blocks) *blocks = e->blocks;
    if (threads) *threads = (e->three_wave || e->wide3) ? 192 : step_threads(e->lpe);  /* step kernels: state wave + observation wave (+ aux wave) */
    if (lds_bytes) *lds_bytes = e->lds_bytes;
    if (lanes_per_env) *lanes_per_env = e->lpe;
    return e->special;  /* >= 0: id of the compile-time specialisation in use (0 = runtime-config kernel) */
}

int mapf_cte_many_launch_info(mapf_handle e, int32_t *blocks, int32_t *threads, int32_t *lds_bytes, int32_t *lanes_per_env) {
    if (!e || !e->cte) return fail(e, MAPF_ERR_STATE, "not a MAPF_FLAG_SINGLE_AGENT handle");
    if (blocks) *blocks = e->cte_many.blocks;
    if (threads) *threads = 128;
    if (lds_bytes) *lds_bytes = e->cte_many.lds_bytes;
    if (lanes_per_env) *lanes_per_env = e->cte_many.lpe;
    return MAPF_OK;
}

}  // extern "C"
