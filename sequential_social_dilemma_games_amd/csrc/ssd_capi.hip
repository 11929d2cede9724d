// csrc/ssd_capi.hip -- host side of the C ABI declared in include/ssd.h.
// Owns the engine state in HBM (hipMalloc), derives the static per-map tables, stages host
// buffers when asked to, and launches the fused kernel of ssd_kernels.hip.  No CPU compute
// path exists here: every stepping call ends in a kernel launch or an error code.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ssd.h"
#include "ssd_internal.hpp"
#include "ssd_aql.hpp"

using ssd::Params;

static thread_local std::string g_create_error;

// One persistent host thread per extra rollout chain (ssd_rollout_random / ssd_rollout_actions through HIP streams).  Created on the first multi-chain call, bound
// to the handle's device once, then parked: a job is handed over through `state` (0 idle, 1 posted, 2 finished, 3 quit).
// After a job the thread keeps polling for the next one for a short while before it sleeps on the condition variable, so a
// loop of short rollout calls (rollout.py:58-70 called per training iteration) pays neither a thread creation + per-thread
// HIP initialisation (~250 us per call: round 1) nor a futex wake-up per call.
struct ChainJob {
    int chain = 0, e_begin = 0, e_end = 0;
    int32_t num_actions = 0, n_steps = 0, reset_every = 0, step0 = 0, ring = 1;
    // caller-supplied actions (ssd_rollout_actions): i32 [action_ring][E][N], step k reads slot (step0 + k) % action_ring;
    // null: every step draws its actions on the device (ssd_rollout_random)
    const int32_t *actions = nullptr; int32_t action_ring = 0;
    const uint8_t *order = nullptr;                  // [action_ring][E][N] action-dict orders (null: index order), same slots
    uint8_t *obs = nullptr; int32_t *rew = nullptr; uint8_t *done = nullptr;
    uint32_t flags = 0;
    hipStream_t s = nullptr;
};
struct ChainWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> state{0};
    ChainJob job;
    int rc = 0;
};

// ssd_rollout_random's own dispatch path (ssd_aql.hip): one HSA queue per chain; per (chain, ring slot) the kernel arguments of
// the step launch and of the reset launch, written once into device memory and reused by every packet ("sets", keyed by what
// the caller passed); a small ring of HSA signals for the fork from the caller's stream.
struct AqlState {
    bool tried = false, ok = false;
    ssd::aql::Queue *q[8] = {};
    int nq = 0;
    static constexpr int kForks = 16;
    uint64_t fork_sig[kForks] = {};
    uint64_t fork_used[kForks][8] = {};             // index of the packet that waits for the fork, per chain (+1; 0 = unused)
    int fork_next = 0;
    unsigned long long *fork_counter = nullptr;     // device memory: += 1 by a one-wave kernel on the caller's stream per forked call
    unsigned long long forks = 0;
    uint8_t *fork_kernarg = nullptr;                // host kernarg memory: kForks blocks of kForkBlock bytes (the chains' waiting kernel)
    static constexpr size_t kForkBlock = 256;
    ssd::aql::Kernel wait_kernel{};
    unsigned long long *join_counter = nullptr;     // device memory: += 1 by every chain's last packet of a call
    unsigned long long joins = 0;                   // value it reaches when everything enqueued so far is done
    void *flag_kernarg = nullptr;                   // host kernarg block: the flag kernel's argument (= join_counter)
    // the join's stream-side wait gave up (SSD_ST_WAIT_TIMEOUT): set by that kernel in host memory, so that the handle's next call
    // sees it without a device round trip; STICKY until ssd_synchronize or the next rollout call has reported it (after_timeout)
    uint32_t *timed_out_host = nullptr, *timed_out_dev = nullptr;
    struct Key {
        const void *obs = nullptr, *rew = nullptr, *done = nullptr;
        int32_t ring = 0, f32 = 0, num_actions = 0, chains = 0, horizon = 0, coherent = 0, split = 0;
        const void *actions = nullptr, *order = nullptr; int32_t action_ring = 0;  // caller-supplied actions: part of every step launch's arguments
        int32_t slots = 0;                                       // argument blocks per chain and kind: lcm(ring, action_ring)
        const void *stamps = nullptr; uint32_t dbg_skip = 0;     // (diagnostic builds: part of the kernel arguments too)
        const void *world = nullptr;                             // sets that are not split: the state buffer their blocks name
        bool operator==(const Key &o) const {
            return stamps == o.stamps && dbg_skip == o.dbg_skip && world == o.world && obs == o.obs && rew == o.rew && done == o.done && ring == o.ring && f32 == o.f32 && num_actions == o.num_actions &&
                   chains == o.chains && horizon == o.horizon && coherent == o.coherent && split == o.split && actions == o.actions && order == o.order &&
                   action_ring == o.action_ring && slots == o.slots;
        }
    };
    struct Geo { ssd::aql::Kernel k; uint32_t grid_x = 0, block_x = 0, lds = 0; };
    struct Set {
        bool valid = false;
        Key key;
        uint8_t *dev = nullptr;                     // [chains][slots][kKinds] blocks of kBlock bytes
        size_t cap = 0;
        Geo geo[8][12];                             // per chain and kind (kKinds: asserted below)
        uint64_t last_use[8] = {};                  // index of the last join packet after a use, per chain (+1)
        uint64_t stamp = 0;
    };
    // Kinds of launches a rollout is made of, each for both ORIENTATIONS o of the handle's pair of state buffers (o = which
    // of the two holds the current state; a launch of a split rollout reads buffer o and writes buffer 1 - o):
    //   kS   step with observations, state in place (rollouts that are not split)     kR   reset with observations, in place
    //   kRn  reset that renders nothing (inside a split rollout), in place
    //   kA   the first step of a split rollout: env waves only                        kAB  env waves + renderer workgroups for the step before
    //   kB   renderer workgroups only, for the step that produced buffer o (ends a split rollout, and precedes a reset inside one)
    enum Kind { kS = 0, kR = 1, kRn = 2, kA = 3, kAB = 4, kB = 5, kKindsPerO = 6, kKinds = 12 };
    static_assert(sizeof(Set::geo[0]) / sizeof(Geo) == kKinds, "Set::geo holds one entry per kind");
    uint8_t *world_buf[2] = {};                     // split rollouts: the pair of state buffers ([0]: the handle's original ones)
    uint32_t *agents_buf[2] = {};
    uint32_t *beam_list[2] = {};                    // [2][E][64] beam marks left by a launch of orientation o (second half: a second pass's)
    uint8_t *snap_grid[2] = {};                     // [E][S] overlay snapshot left by a launch of orientation o (rare steps)
    static constexpr int kSets = 4;
    static constexpr size_t kBlock = 512;           // >= sizeof(ssd::KernArgs), a multiple of 64
    Set sets[kSets];
    uint64_t clock = 0;
};

struct ssd_env {
    int game = 0, H = 0, W = 0, WP = 0, S = 0, E = 0, N = 0, view_len = 7, V = 15, beam_len = 5;
    int device = 0, keep_beams = 0, potential_waste = 0;
    uint64_t seed = 0;
    uint32_t env_base = 0;
    Params p{};                       // persistent part of the kernel parameters
    std::vector<void *> allocs;
    // staging for SSD_HOST_PTRS
    int32_t *st_actions = nullptr, *st_rew = nullptr, *st_actions_out = nullptr;
    float *st_obs_f32 = nullptr;
    // small handles (the dict API's single env): host-pinned, device-mapped staging -- the kernel reads / writes it
    // directly, so a host-pointer call is one launch + one synchronise instead of five copies
    bool st_mapped = false;
    uint8_t *st_host = nullptr;      // host view of the mapped block; the st_* pointers above are its device view
    std::vector<void *> host_allocs;
    uint8_t *st_order = nullptr, *st_obs = nullptr, *st_done = nullptr, *st_mask = nullptr, *st_rgb = nullptr;
    size_t st_rgb_frames = 0;                          // frames st_rgb holds
    // ssd_rollout_random: extra chains (streams + fork / join events), created on first use
    std::vector<hipStream_t> chain_streams;
    std::vector<hipEvent_t> chain_events;
    hipEvent_t fork_event = nullptr;
    int rollout_chains = 0;           // 0 = automatic
    std::vector<std::unique_ptr<ChainWorker>> workers;   // workers[c - 1] enqueues chain c
    std::unique_ptr<AqlState> aql;                       // the library's own dispatch path (ssd_aql.hip), set up on first use
    // The stream the last device-pointer call enqueued on.  A host-pointer call (which runs on the stream it is given, usually
    // the NULL stream, and returns finished results) first waits for that stream when it is another one: a caller that
    // mixes the two styles -- tensors on a non-blocking side stream, then a *_host call -- gets program order.
    hipStream_t last_stream = nullptr;
    bool last_stream_set = false;
    int last_path = 0;                                   // SSD_PATH_* of the last rollout call
    std::string err;
};

#ifndef SSD_RING_EVERY            // (experiment switch: 1 = a doorbell per step and chain)
#define SSD_RING_EVERY 4
#endif
namespace {

#define SSD_HIP(env, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (env)->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            return SSD_E_DEVICE;                                                                \
        }                                                                                       \
    } while (0)

// rand < p  <=>  k < ceil(p * 2^32) for rand = k / 2^32
uint64_t threshold(double p) {
    if (p <= 0.0) return 0;
    const double x = std::ceil(p * 4294967296.0);
    return x >= 4294967296.0 ? 4294967296ull : (uint64_t)x;
}

// cleanup.py:156-171 compute_probabilities for a map holding n_h cells of 'H' (constants :24-27)
void cleanup_probs(int potential, int n_h, double *p_apple, double *p_waste) {
    const double depletion = 0.4, restoration = 0.0, waste_p = 0.5, apple_p = 0.05;
    double density = 0;
    if (potential > 0) density = 1 - (double)(potential - n_h) / (double)potential;
    if (density >= depletion) { *p_apple = 0; *p_waste = 0; return; }
    *p_waste = waste_p;
    *p_apple = density <= restoration ? apple_p : (1 - (density - restoration) / (depletion - restoration)) * apple_p;
}

constexpr uint8_t kVoid = '0';      // the glyph utility_funcs.py:94-114 pads views with; fills the row padding of the grids

// dense [E][H][W] (the ABI's layout) <-> the engine's padded-row grids [E][S], row stride WP
void pack_grid(const ssd_env *env, const int8_t *dense, std::vector<uint8_t> &grid, uint8_t pad) {
    grid.assign((size_t)env->E * env->S, pad);
    for (int e = 0; e < env->E; ++e)
        for (int r = 0; r < env->H; ++r)
            std::memcpy(grid.data() + (size_t)e * env->S + (size_t)r * env->WP, dense + ((size_t)e * env->H + r) * env->W, env->W);
}
void unpack_grid(const ssd_env *env, const std::vector<uint8_t> &grid, int8_t *dense) {
    for (int e = 0; e < env->E; ++e)
        for (int r = 0; r < env->H; ++r)
            std::memcpy(dense + ((size_t)e * env->H + r) * env->W, grid.data() + (size_t)e * env->S + (size_t)r * env->WP, env->W);
}

template <typename T>
int dev_alloc(ssd_env *env, T **out, size_t count, bool zero = true) {
    void *ptr = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc(&ptr, bytes);
    if (e != hipSuccess) { env->err = std::string("hipMalloc: ") + hipGetErrorString(e); return SSD_E_NOMEM; }
    if (zero) {
        e = hipMemset(ptr, 0, bytes);
        if (e != hipSuccess) { env->err = std::string("hipMemset: ") + hipGetErrorString(e); return SSD_E_DEVICE; }
    }
    env->allocs.push_back(ptr);
    *out = static_cast<T *>(ptr);
    return SSD_OK;
}

template <typename T>
int upload(ssd_env *env, T **out, const std::vector<T> &host) {
    int rc = dev_alloc(env, out, host.size(), host.empty());
    if (rc) return rc;
    if (!host.empty()) SSD_HIP(env, hipMemcpy(*out, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return SSD_OK;
}

int fail_create(const std::string &msg, int code) {
    g_create_error = msg;
    return code;
}

#ifdef SSD_EXP_OBS768   // (experiment builds only: see ssd_kernels.hip)
size_t obs_bytes(const ssd_env *env, bool f32 = false) { return f32 ? (size_t)env->E * env->N * env->V * env->V * 3 * 4 : (size_t)env->E * env->N * SSD_EXP_OBS768; }
#else
size_t obs_bytes(const ssd_env *env, bool f32 = false) { return (size_t)env->E * env->N * env->V * env->V * 3 * (f32 ? 4 : 1); }
#endif

int ensure_staging(ssd_env *env) {
    if (env->st_obs) return SSD_OK;
    const size_t en = (size_t)env->E * env->N;
    int rc;
    const size_t a16 = 15;
    const size_t o_act = 0, o_aout = (o_act + en * 4 + a16) & ~a16, o_rew = (o_aout + en * 4 + a16) & ~a16,
                 o_ord = (o_rew + en * 4 + a16) & ~a16, o_done = (o_ord + en + a16) & ~a16, o_mask = (o_done + en + a16) & ~a16,
                 o_obs = (o_mask + (size_t)env->E + a16) & ~a16, o_f32 = (o_obs + obs_bytes(env) + a16) & ~a16,
                 total = o_f32 + obs_bytes(env, true);
    if (total <= (1u << 20)) {
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, total, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            std::memset(h, 0, total);
            env->host_allocs.push_back(h);
            env->st_mapped = true; env->st_host = static_cast<uint8_t *>(h);
            uint8_t *b = static_cast<uint8_t *>(d);
            env->st_actions = reinterpret_cast<int32_t *>(b + o_act); env->st_actions_out = reinterpret_cast<int32_t *>(b + o_aout);
            env->st_rew = reinterpret_cast<int32_t *>(b + o_rew); env->st_order = b + o_ord; env->st_done = b + o_done;
            env->st_mask = b + o_mask; env->st_obs = b + o_obs; env->st_obs_f32 = reinterpret_cast<float *>(b + o_f32);
            return SSD_OK;
        }
        if (h) (void)hipHostFree(h);
        (void)hipGetLastError();                 // fall back to device staging
    }
    if ((rc = dev_alloc(env, &env->st_actions, en))) return rc;
    if ((rc = dev_alloc(env, &env->st_actions_out, en))) return rc;
    if ((rc = dev_alloc(env, &env->st_rew, en))) return rc;
    if ((rc = dev_alloc(env, &env->st_order, en))) return rc;
    if ((rc = dev_alloc(env, &env->st_done, en))) return rc;
    if ((rc = dev_alloc(env, &env->st_mask, (size_t)env->E))) return rc;
    if ((rc = dev_alloc(env, &env->st_obs, obs_bytes(env)))) return rc;
    return SSD_OK;
}

// Runs one launch of the fused kernel with the per-call pointers filled in; handles host staging.
int run(ssd_env *env, int mode, const int32_t *actions, const uint8_t *order, const uint8_t *mask,
        int num_actions_random, int32_t *actions_out, void *obs_v, int32_t *rew, uint8_t *done, int rotate,
        uint32_t flags, void *stream) {
    {   // the stepping calls are launch-bound on the host: only switch devices when the thread is on another one
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != env->device) SSD_HIP(env, hipSetDevice(env->device));
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t en = (size_t)env->E * env->N;
    const bool host = (flags & SSD_HOST_PTRS) != 0, f32 = (flags & SSD_OBS_F32) != 0;
    uint8_t *obs = static_cast<uint8_t *>(obs_v);
    Params p = env->p;
    p.obs_f32 = f32 ? 1 : 0;
    p.mode = mode; p.rotate = rotate; p.num_actions_random = num_actions_random;
    if (!host) {
        if (obs && (reinterpret_cast<uintptr_t>(obs) & 3u)) { env->err = "obs must be 4-byte aligned"; return SSD_E_INVALID; }
        env->last_stream = s; env->last_stream_set = true;
        p.actions = actions; p.order = order; p.mask = mask; p.actions_out = actions_out;
        p.obs = obs; p.rew = rew; p.done = done;
        // (measured, round 2: the per-call step with the coherent kernel variant -- nothing left dirty for the launch's release -- 7.9
        // against 8.0 us per Python call.  Round 4: twenty such steps captured into a HIP graph replay at the pace of twenty calls,
        // 7.15 against 7.13 us per 4096-env step (bench.py, policy_step): back-to-back per-call steps are bound by the kernel --
        // one launch whose waves step AND render, each launch waiting for the one before -- not by the host's 5 us per launch)
        ssd::launch(p, env->game, stream);
        SSD_HIP(env, hipGetLastError());
        return SSD_OK;
    }
    if (env->last_stream_set && env->last_stream != s) {
        SSD_HIP(env, hipStreamSynchronize(env->last_stream));
        env->last_stream_set = false;
    }
    int rc = ensure_staging(env);
    if (rc) return rc;
    if (env->st_mapped) {
        // the staging block is host memory the GPU addresses directly: plain memcpy in, one launch, one wait, memcpy out
        auto hostp = [&](const void *dev) { return env->st_host + (static_cast<const uint8_t *>(dev) - reinterpret_cast<const uint8_t *>(env->st_actions)); };
        if (actions) { std::memcpy(hostp(env->st_actions), actions, en * sizeof(int32_t)); p.actions = env->st_actions; }
        if (order) { std::memcpy(hostp(env->st_order), order, en); p.order = env->st_order; }
        if (mask) { std::memcpy(hostp(env->st_mask), mask, (size_t)env->E); p.mask = env->st_mask; }
        if (actions_out) p.actions_out = env->st_actions_out;
        if (obs) {
            p.obs = f32 ? reinterpret_cast<uint8_t *>(env->st_obs_f32) : env->st_obs;
            if (mask) std::memcpy(hostp(p.obs), obs, obs_bytes(env, f32));      // rows of envs that are not reset come back unchanged
        }
        if (rew) p.rew = env->st_rew;
        if (done) p.done = env->st_done;
        ssd::launch(p, env->game, stream);
        SSD_HIP(env, hipGetLastError());
        SSD_HIP(env, hipStreamSynchronize(s));
        if (actions_out) std::memcpy(actions_out, hostp(env->st_actions_out), en * sizeof(int32_t));
        if (obs) std::memcpy(obs, hostp(p.obs), obs_bytes(env, f32));
        if (rew) std::memcpy(rew, hostp(env->st_rew), en * sizeof(int32_t));
        if (done) std::memcpy(done, hostp(env->st_done), en);
        return SSD_OK;
    }
    if (actions) { SSD_HIP(env, hipMemcpyAsync(env->st_actions, actions, en * sizeof(int32_t), hipMemcpyHostToDevice, s)); p.actions = env->st_actions; }
    if (order) { SSD_HIP(env, hipMemcpyAsync(env->st_order, order, en, hipMemcpyHostToDevice, s)); p.order = env->st_order; }
    if (mask) { SSD_HIP(env, hipMemcpyAsync(env->st_mask, mask, (size_t)env->E, hipMemcpyHostToDevice, s)); p.mask = env->st_mask; }
    if (actions_out) p.actions_out = env->st_actions_out;
    if (obs) {
        if (f32 && !env->st_obs_f32) { if ((rc = dev_alloc(env, &env->st_obs_f32, obs_bytes(env) /* floats */))) return rc; }
        p.obs = f32 ? reinterpret_cast<uint8_t *>(env->st_obs_f32) : env->st_obs;
        if (mask)   // rows of envs that are not reset must come back unchanged
            SSD_HIP(env, hipMemcpyAsync(p.obs, obs, obs_bytes(env, f32), hipMemcpyHostToDevice, s));
    }
    if (rew) p.rew = env->st_rew;
    if (done) p.done = env->st_done;
    ssd::launch(p, env->game, stream);
    SSD_HIP(env, hipGetLastError());
    if (actions_out) SSD_HIP(env, hipMemcpyAsync(actions_out, env->st_actions_out, en * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (obs) SSD_HIP(env, hipMemcpyAsync(obs, p.obs, obs_bytes(env, f32), hipMemcpyDeviceToHost, s));
    if (rew) SSD_HIP(env, hipMemcpyAsync(rew, env->st_rew, en * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (done) SSD_HIP(env, hipMemcpyAsync(done, env->st_done, en, hipMemcpyDeviceToHost, s));
    SSD_HIP(env, hipStreamSynchronize(s));
    return SSD_OK;
}

}  // namespace

extern "C" {

int ssd_abi_version(void) { return SSD_ABI_VERSION; }

const char *ssd_last_error(const ssd_env *env) { return env ? env->err.c_str() : g_create_error.c_str(); }

int ssd_create(const ssd_config *cfg, ssd_env **out) {
    if (!cfg || !out) return fail_create("null argument", SSD_E_INVALID);
    *out = nullptr;
    if (cfg->struct_size != sizeof(ssd_config)) return fail_create("ssd_config.struct_size mismatch", SSD_E_INVALID);
    const int H = cfg->height, W = cfg->width, N = cfg->num_agents, E = cfg->num_envs;
    const int V = 2 * cfg->view_len + 1;
    if (cfg->game != SSD_GAME_HARVEST && cfg->game != SSD_GAME_CLEANUP) return fail_create("unknown game", SSD_E_INVALID);
    if (!cfg->base_map || H < 3 || W < 3 || H > 4095 || W > 4095 || (long)H * W > ssd::kMaxCells)
        return fail_create("map must be 3..4095 on a side with at most 4096 cells", SSD_E_INVALID);
    if (E < 1 || N < 0 || N > ssd::kMaxAgents) return fail_create("need num_envs >= 1 and 0 <= num_agents <= 64", SSD_E_INVALID);
    if (cfg->view_len < 0 || cfg->view_len > 15) return fail_create("view_len must be 0..15", SSD_E_INVALID);
    if (cfg->beam_len < 0 || cfg->beam_len > ssd::kMaxBeamLen) return fail_create("beam_len must be 0..21", SSD_E_INVALID);
    const int hw = H * W;
    // Appendix C.11 of SURVEY.md: the reference indexes out of range on maps without a closed wall
    // border (agent.py:111); the engine rejects them.
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const char ch = cfg->base_map[r * W + c];
            if ((r == 0 || c == 0 || r == H - 1 || c == W - 1) && ch != '@') return fail_create("map border must be all '@'", SSD_E_INVALID);
            if (ch <= 0) return fail_create("map cells must be 7-bit ASCII", SSD_E_INVALID);
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail_create("no HIP device available: this engine has no CPU path", SSD_E_DEVICE);
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail_create("device_id out of range", SSD_E_INVALID);
    if (hipSetDevice(cfg->device_id) != hipSuccess) return fail_create("hipSetDevice failed", SSD_E_DEVICE);

    ssd_env *env = new ssd_env();
    env->game = cfg->game; env->H = H; env->W = W; env->E = E; env->N = N;
    env->view_len = cfg->view_len; env->V = V; env->beam_len = cfg->beam_len;
    env->device = cfg->device_id; env->keep_beams = cfg->keep_beams ? 1 : 0;
    env->seed = cfg->seed; env->env_base = cfg->env_index_base;
    // grid layout (ssd_kernels.hip): row stride WP = W + view_len, padding = void glyph; LDS aprons of view_len rows
    const int WP = W + cfg->view_len, S = (H * WP + 15) & ~15;
    env->WP = WP; env->S = S;

    // static per-map tables (map_env.py:93-101, harvest.py:22-26, cleanup.py:44-62)
    std::vector<uint8_t> reset_world(S, kVoid);
    std::vector<uint32_t> spawn_cells, apple_cells, waste_cells;   // grid index | dense index << 16
    const char apple_ch = cfg->game == SSD_GAME_HARVEST ? 'A' : 'B';
    for (int dc = 0; dc < hw; ++dc) {
        const char b = cfg->base_map[dc];
        const int c = (dc / W) * WP + dc % W;                   // grid index of the cell
        const uint32_t entry = (uint32_t)c | ((uint32_t)dc << 16);
        if (b == apple_ch) apple_cells.push_back(entry);
        if (cfg->game == SSD_GAME_CLEANUP && (b == 'H' || b == 'R')) { waste_cells.push_back(entry); env->potential_waste++; }
        if (b == 'P') spawn_cells.push_back(entry);
        char w = ' ';                         // reset_map (:560-564) + custom_reset (harvest.py:57-60, cleanup.py:84-92)
        if (b == '@') w = '@';
        else if (cfg->game == SSD_GAME_HARVEST && b == 'A') w = 'A';
        else if (cfg->game == SSD_GAME_CLEANUP && (b == 'H' || b == 'R' || b == 'S')) w = b;
        reset_world[c] = (uint8_t)w;
    }
    std::vector<uint32_t> lut(128, 0);
    if (cfg->color_lut) {
        for (int i = 0; i < 128; ++i)
            lut[i] = cfg->color_lut[3 * i] | (cfg->color_lut[3 * i + 1] << 8) | (cfg->color_lut[3 * i + 2] << 16);
    } else {                                  // map_env.py:24-41 DEFAULT_COLOURS + cleanup.py:15-18 CLEANUP_COLORS
        auto set = [&](char ch, int r, int g, int b) { lut[(int)ch] = r | (g << 8) | (b << 16); };
        set('@', 180, 180, 180); set('A', 0, 255, 0); set('F', 255, 255, 0); set('P', 159, 67, 255);
        set('1', 159, 67, 255); set('2', 2, 81, 154); set('3', 204, 0, 204); set('4', 216, 30, 54);
        set('5', 254, 151, 0); set('6', 100, 255, 255); set('7', 99, 99, 255); set('8', 250, 204, 255);
        set('9', 238, 223, 16); set('C', 100, 255, 255); set('S', 113, 75, 24); set('H', 99, 156, 194);
        set('R', 113, 75, 24);
    }
    const int n_thr = env->potential_waste + 1;
    std::vector<uint64_t> thr_ca(n_thr), thr_cw(n_thr);
    for (int n = 0; n < n_thr; ++n) {
        if (cfg->cleanup_apple_thresholds && cfg->cleanup_waste_thresholds) {
            thr_ca[n] = cfg->cleanup_apple_thresholds[n]; thr_cw[n] = cfg->cleanup_waste_thresholds[n];
        } else {
            double pa, pw;
            cleanup_probs(env->potential_waste, n, &pa, &pw);
            thr_ca[n] = threshold(pa); thr_cw[n] = threshold(pw);
        }
    }

    Params &p = env->p;
    p.E = E; p.E_total = E; p.ring = 1; p.N = N; p.H = H; p.W = W; p.WP = WP; p.S = S;
    p.A0 = (cfg->view_len * (WP + 1) + 15) & ~15;
    p.A1 = (cfg->view_len * WP + 15) & ~15;
    p.view_len = cfg->view_len; p.V = V; p.beam_len = cfg->beam_len;
    p.keep_beams = env->keep_beams;
    p.v_magic16 = (65536u + (uint32_t)V - 1u) / (uint32_t)V;
    p.seed_lo = (uint32_t)cfg->seed; p.seed_hi = (uint32_t)(cfg->seed >> 32); p.env_base = cfg->env_index_base;
    p.n_spawn = (int)spawn_cells.size(); p.n_thr = n_thr;
    p.n_apple = (int)apple_cells.size(); p.n_waste = (int)waste_cells.size();
    p.n_waste_reset = 0;
    for (uint8_t ch : reset_world) p.n_waste_reset += ch == 'H';
    const double sp[4] = {0, 0.005, 0.02, 0.05};   // harvest.py:13 SPAWN_PROB
    for (int i = 0; i < 4; ++i) {
        const uint64_t T = cfg->harvest_thresholds ? cfg->harvest_thresholds[i] : threshold(sp[i]);
        p.thr_h32[i] = T >= 4294967296ull ? 0xFFFFFFFFu : (uint32_t)T;
        if (T >= 4294967296ull) p.thr_h_always |= 1u << i;
    }

    auto bail = [&](int rc) {
        g_create_error = env->err;
        for (void *ptr : env->allocs) (void)hipFree(ptr);
        delete env;
        return rc;
    };
    int rc;
    if (ssd::lds_bytes(p, 1, true) > 64 * 1024) { env->err = "map too large for the 64 KiB LDS budget"; return bail(SSD_E_INVALID); }
    if ((rc = dev_alloc(env, &p.world, (size_t)E * S))) return bail(rc);
    if (env->keep_beams) { if ((rc = dev_alloc(env, &p.beam, (size_t)E * S))) return bail(rc); }
    if ((rc = dev_alloc(env, &p.agents, (size_t)E * N))) return bail(rc);
    if (N > 0) {   // before the first reset every agent sits on interior cell (1,1), facing UP: stepping an env that
                   // was never reset is then well defined and cannot index outside the grid
        std::vector<uint32_t> ag((size_t)E * N, (uint32_t)(WP + 1) | (2u << 16));
        if (hipMemcpy(p.agents, ag.data(), ag.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { env->err = "hipMemcpy(agents)"; return bail(SSD_E_DEVICE); }
    }
    if ((rc = dev_alloc(env, &p.status, 1))) return bail(rc);
    {   // header: episode = 0xFFFFFFFF ("never reset"; the first reset wraps it to 0)
        std::vector<uint4> hdr(E);
        for (int e = 0; e < E; ++e) hdr[e] = make_uint4(0, 0, 0xFFFFFFFFu, 0);
        if ((rc = upload(env, &p.hdr, hdr))) return bail(rc);
        std::vector<uint8_t> w0((size_t)E * S, kVoid);   // world starts blank (map_env.py:85)
        for (int e = 0; e < E; ++e)
            for (int r = 0; r < H; ++r) std::memset(w0.data() + (size_t)e * S + (size_t)r * WP, ' ', W);
        if (hipMemcpy(p.world, w0.data(), w0.size(), hipMemcpyHostToDevice) != hipSuccess) { env->err = "hipMemcpy(world)"; return bail(SSD_E_DEVICE); }
    }
    uint8_t *d8; uint32_t *d32; uint64_t *d64;
    if ((rc = upload(env, &d8, reset_world))) return bail(rc);
    p.reset_world = d8;
    if ((rc = upload(env, &d32, spawn_cells))) return bail(rc);
    p.spawn_cells = d32;
    if ((rc = upload(env, &d32, apple_cells))) return bail(rc);
    p.apple_cells = d32;
    if ((rc = upload(env, &d32, waste_cells))) return bail(rc);
    p.waste_cells = d32;
    if ((rc = upload(env, &d32, lut))) return bail(rc);
    p.lut = d32;
    {   // float32 of the reference's float64 normalisation (map_env.py:199), one entry per byte value
        // ... applied to the colour table: glyph -> (r, g, b, 0) as float32, one 16-byte entry per glyph
        std::vector<float> f32lut(128 * 4, 0.0f);
        for (int g = 0; g < 128; ++g)
            for (int c = 0; c < 3; ++c) f32lut[4 * g + c] = (float)(((double)((lut[g] >> (8 * c)) & 0xFFu) - 128.0) / 255.0);
        float *df;
        if ((rc = upload(env, &df, f32lut))) return bail(rc);
        p.f32lut = df;
    }
    if (cfg->view_len == 7) {   // the per-lane window offsets of the 15 x 15 view (ssd_kernels.hip, render_views_std): constants of the row stride
        std::vector<uint32_t> vt(64 * 8);
        for (int l = 0; l < 64; ++l) {
            const int pp0 = 4 * l > 221 ? 221 : 4 * l;
            for (int q = 0; q < 4; ++q) {
                const int pp = pp0 + q, i = pp / 15, j = pp % 15;
                vt[l * 8 + q] = (uint32_t)(i * WP + j);
                vt[l * 8 + 4 + q] = (uint32_t)(j * WP + (14 - i));
            }
        }
        if ((rc = upload(env, &d32, vt))) return bail(rc);
        p.view_tab = d32;
    }
    if ((rc = upload(env, &d64, thr_ca))) return bail(rc);
    p.thr_ca = d64;
    if ((rc = upload(env, &d64, thr_cw))) return bail(rc);
    p.thr_cw = d64;
    *out = env;
    return SSD_OK;
}

static void stop_workers(ssd_env *env);
static void aql_teardown(ssd_env *env);
static int rollout(ssd_env *env, const int32_t *actions, const uint8_t *order, int32_t action_ring, int32_t num_actions, int32_t n_steps,
                   int32_t reset_every, int32_t step0, void *obs, int32_t *rew, uint8_t *done, int32_t ring, uint32_t flags, void *stream);

int ssd_destroy(ssd_env *env) {
    if (!env) return SSD_E_INVALID;
    stop_workers(env);
    (void)hipSetDevice(env->device);
    (void)hipDeviceSynchronize();
    aql_teardown(env);
    for (void *ptr : env->allocs) (void)hipFree(ptr);
    for (void *ptr : env->host_allocs) (void)hipHostFree(ptr);
    for (hipStream_t cs : env->chain_streams) (void)hipStreamDestroy(cs);
    for (hipEvent_t ce : env->chain_events) (void)hipEventDestroy(ce);
    if (env->fork_event) (void)hipEventDestroy(env->fork_event);
    delete env;
    return SSD_OK;
}

int ssd_reset(ssd_env *env, const uint8_t *env_mask, void *obs, uint32_t flags, void *stream) {
    if (!env) return SSD_E_INVALID;
    return run(env, ssd::kModeReset, nullptr, nullptr, env_mask, 0, nullptr, obs, nullptr, nullptr, /*rotate=*/0, flags, stream);
}

int ssd_step(ssd_env *env, const int32_t *actions, const uint8_t *order, void *obs, int32_t *rew, uint8_t *done,
             uint32_t flags, void *stream) {
    if (!env) return SSD_E_INVALID;
    if (!actions && env->N > 0) { env->err = "actions is null"; return SSD_E_INVALID; }   // an env without agents has no actions
    if ((flags & SSD_AUTO_RESET) && (flags & SSD_OBS_F32)) { env->err = "an auto-reset step writes uint8 observations"; return SSD_E_INVALID; }
    return run(env, (flags & SSD_AUTO_RESET) ? ssd::kModeStepAuto : ssd::kModeStep, actions, order, nullptr, 0, nullptr, obs, rew, done, 1,
               flags, stream);
}

int ssd_step_random(ssd_env *env, int32_t num_actions, int32_t *actions_out, void *obs, int32_t *rew, uint8_t *done,
                    uint32_t flags, void *stream) {
    if (!env) return SSD_E_INVALID;
    const int na = env->game == SSD_GAME_HARVEST ? 8 : 9;
    if (num_actions < 1 || num_actions > na) { env->err = "num_actions outside the game's Discrete(n)"; return SSD_E_INVALID; }
    if ((flags & SSD_AUTO_RESET) && (flags & SSD_OBS_F32)) { env->err = "an auto-reset step writes uint8 observations"; return SSD_E_INVALID; }
    return run(env, (flags & SSD_AUTO_RESET) ? ssd::kModeStepAuto : ssd::kModeStep, nullptr, nullptr, nullptr, num_actions, actions_out, obs,
               rew, done, 1, flags, stream);
}

// The launches of step k of a plain (not fused) chain: a full reset when one is due, then the step.
struct ChainCursor {
    Params p;
    hipStream_t s;
    uint8_t *obs; int32_t *rew; uint8_t *done;
    const int32_t *actions; const uint8_t *order; int32_t action_ring;
    size_t en, ob;
    int32_t num_actions, reset_every, step0, ring;
};
static ChainCursor chain_cursor(const ssd_env *env, const ChainJob &j) {
    ChainCursor c;
    const bool f32 = (j.flags & SSD_OBS_F32) != 0;
    c.p = env->p;
    c.p.obs_f32 = f32 ? 1 : 0;
    c.p.e_begin = j.e_begin; c.p.E = j.e_end;
    c.s = j.s; c.obs = j.obs; c.rew = j.rew; c.done = j.done;
    c.actions = j.actions; c.order = j.order; c.action_ring = j.action_ring;
    c.en = (size_t)env->E * env->N; c.ob = obs_bytes(env, f32);
    // An output ring whose observations do not fit the device's 256-MB memory-side cache: the write-through stores bypass it
    // (non-temporal).  One slot rewritten every step lives in that cache (4096 envs: 13.8 MB, 5.43 us per step; 6.91 if it
    // bypassed it); 16 slots (221 MB) still do: 5.50 (7.06 bypassing); 32 slots (442 MB) thrash it: 8.07 us per step, with
    // non-temporal stores 7.03 (`python bench.py --ring R`, alternating fresh processes).
    c.p.obs_nt = (j.obs && (size_t)j.ring * c.ob > ((size_t)232 << 20)) ? 1 : 0;
    c.num_actions = j.num_actions; c.reset_every = j.reset_every; c.step0 = j.step0; c.ring = j.ring;
    return c;
}
// the step launch's action source: the caller's slot of the action ring, or the device's own draw
static inline void cursor_actions(ChainCursor &c, Params &p, size_t step_index) {
    if (c.actions) {
        const size_t at = (step_index % (size_t)c.action_ring) * c.en;
        p.actions = c.actions + at; p.order = c.order ? c.order + at : nullptr; p.num_actions_random = 0;
    } else { p.actions = nullptr; p.order = nullptr; p.num_actions_random = c.num_actions; }
}
static inline void chain_launch_step(const ssd_env *env, ChainCursor &c, int k) {
    Params &p = c.p;
    const size_t slot = (size_t)((c.step0 + k) % c.ring);
    p.obs = c.obs ? c.obs + slot * c.ob : nullptr;
    if (c.reset_every > 0 && (c.step0 + k) % c.reset_every == 0) {
        p.mode = ssd::kModeReset; p.rotate = 0; p.num_actions_random = 0; p.actions = nullptr; p.order = nullptr; p.rew = nullptr; p.done = nullptr;
        ssd::launch(p, env->game, c.s);
    }
    p.mode = ssd::kModeStep; p.rotate = 1;
    cursor_actions(c, p, (size_t)(c.step0 + k));
    p.rew = c.rew ? c.rew + slot * c.en : nullptr; p.done = c.done ? c.done + slot * c.en : nullptr;
    ssd::launch(p, env->game, c.s);
}

// One chain of a rollout: the launches of steps [0, n_steps) for envs [e_begin, e_end), enqueued on j.s by the calling thread
// (which is on the handle's device).
static int rollout_chain(ssd_env *env, const ChainJob &j) {
    const int32_t n_steps = j.n_steps;
    hipStream_t s = j.s;
    if (j.flags & SSD_ROLLOUT_FUSED) {
        // ONE launch for the whole chain: the kernel keeps each env in LDS / registers across its n_steps steps
        if (n_steps == 0) return SSD_OK;
        Params p = env->p;
        p.obs_f32 = 0;
        p.e_begin = j.e_begin; p.E = j.e_end;
        p.mode = ssd::kModeRollout; p.rotate = 1;
        p.num_actions_random = j.actions ? 0 : j.num_actions;
        p.actions = j.actions; p.order = j.actions ? j.order : nullptr; p.action_ring = j.actions ? j.action_ring : 0;
        p.n_steps = n_steps; p.reset_every = j.reset_every; p.step0 = j.step0; p.ring = j.ring;
        p.obs = j.obs; p.rew = j.rew; p.done = j.done;
        p.obs_nt = (j.obs && (size_t)j.ring * obs_bytes(env, false) > ((size_t)232 << 20)) ? 1 : 0;
        ssd::launch(p, env->game, s);
        return hipGetLastError() == hipSuccess ? SSD_OK : SSD_E_DEVICE;
    }
    ChainCursor c = chain_cursor(env, j);
    for (int k = 0; k < n_steps; ++k) chain_launch_step(env, c, k);
    return hipGetLastError() == hipSuccess ? SSD_OK : SSD_E_DEVICE;
}

// Body of a chain worker thread (see ChainWorker).
static void chain_worker_main(ssd_env *env, ChainWorker *w) {
    const bool on_device = hipSetDevice(env->device) == hipSuccess;       // once per thread
    static const int spin_us = SSD_HOOK("SSD_WORKER_SPIN_US", 200);
    for (;;) {
        int st = w->state.load(std::memory_order_acquire);
        if (st != 1 && st != 3) {
            // poll for a while (the caller of a rollout loop is back within microseconds), then sleep
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                st = w->state.load(std::memory_order_acquire);
                if (st == 1 || st == 3) break;
                __builtin_ia32_pause();
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
            }
            if (st != 1 && st != 3) {
                std::unique_lock<std::mutex> lk(w->mu);
                w->cv.wait(lk, [&] { const int v = w->state.load(std::memory_order_acquire); return v == 1 || v == 3; });
                st = w->state.load(std::memory_order_acquire);
            }
        }
        if (st == 3) return;
        w->rc = on_device ? rollout_chain(env, w->job) : SSD_E_DEVICE;
        w->state.store(2, std::memory_order_release);
    }
}
static void post_job(ChainWorker *w, const ChainJob &j) {
    w->job = j;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->state.store(1, std::memory_order_release);
    }
    w->cv.notify_one();
}
static int wait_job(ChainWorker *w) {
    // the worker finishes within microseconds of the caller's own chain: poll
    for (int spins = 0; w->state.load(std::memory_order_acquire) != 2; ++spins) {
        if (spins < 4096) __builtin_ia32_pause();
        else std::this_thread::yield();
    }
    w->state.store(0, std::memory_order_relaxed);
    return w->rc;
}
static void stop_workers(ssd_env *env) {
    for (auto &w : env->workers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->state.store(3, std::memory_order_release);
        }
        w->cv.notify_one();
        if (w->th.joinable()) w->th.join();
    }
    env->workers.clear();
}

// Everything the handle ever put into the library's queues has completed when this returns (or a queue has failed).
// hipDeviceSynchronize() does not cover queues the HIP runtime does not own, so whoever frees memory those packets name --
// ssd_destroy, an error path that leaves a call's work behind without a stream-side wait -- drains them on the host first.
static void aql_drain(ssd_env *env) {
    if (!env->aql) return;
    AqlState &A = *env->aql;
    if (!A.ok && !A.nq) return;
    if (!A.joins || !A.flag_kernarg) return;                        // (nothing was ever enqueued)
    std::lock_guard<std::mutex> enqueue_lock(ssd::aql::enqueue_mutex(env->device));
    for (int c = 0; c < A.nq; ++c) {
        if (!A.q[c] || ssd::aql::queue_failed(A.q[c])) continue;
        if (ssd::aql::join_and_wait(A.q[c], A.flag_kernarg)) A.joins += 1;
    }
}

// A wait of one of the handle's rollout calls timed out: the call had returned SSD_OK long before, the caller's
// stream went on, and the chains' packets may still be running and writing obs / rew / done and the state pair.  The first API
// call that notices -- the next rollout call, ssd_synchronize -- drains the library's queues on the host (bounded: a queue that
// is stuck for good is left alone), takes the handle off the library's own dispatch path (later calls issue the same launches
// through hipLaunchKernel) and reports SSD_E_DEVICE once.  Until then the call's buffers must not be freed or reused.
static int after_timeout(ssd_env *env) {
    if (!env->aql || !env->aql->timed_out_host) return SSD_OK;
    AqlState &A = *env->aql;
    if (!__atomic_load_n(A.timed_out_host, __ATOMIC_ACQUIRE)) return SSD_OK;
    bool drained = true;
    if (A.joins && A.flag_kernarg) {
        std::lock_guard<std::mutex> enqueue_lock(ssd::aql::enqueue_mutex(env->device));
        for (int c = 0; c < A.nq; ++c) {
            if (!A.q[c] || ssd::aql::queue_failed(A.q[c])) continue;
            if (ssd::aql::join_and_wait(A.q[c], A.flag_kernarg, 10.0)) A.joins += 1; else drained = false;
        }
    }
    __atomic_store_n(A.timed_out_host, 0u, __ATOMIC_RELEASE);
    A.ok = false;
    env->err = drained ? "a rollout call's wait for the library's dispatch queues timed out (the WAIT_TIMEOUT status bit): its outputs may be incomplete; "
                         "the queues have been drained and the handle now steps through hipLaunchKernel"
                       : "a rollout call's wait for the library's dispatch queues timed out (the WAIT_TIMEOUT status bit) and the queues did not "
                         "drain within 10 s: its buffers may still be written to";
    return SSD_E_DEVICE;
}

static void aql_teardown(ssd_env *env) {
    if (!env->aql) return;
    AqlState &A = *env->aql;
    // (the queues belong to the device's pool and stay; the handle's work in them must be over before its memory goes)
    aql_drain(env);
    for (uint64_t h : A.fork_sig) ssd::aql::signal_destroy(h);
    if (A.join_counter) (void)hipFree(A.join_counter);
    if (A.fork_counter) (void)hipFree(A.fork_counter);
    ssd::aql::host_kernarg_free(A.flag_kernarg);
    ssd::aql::host_kernarg_free(A.fork_kernarg);
    for (auto &st : A.sets) if (st.dev) (void)hipFree(st.dev);
    // (of the state pair, world_buf[0] / agents_buf[0] are the handle's original allocations: freed with env->allocs; if the
    // current state sits in the other pair, ssd_destroy frees whichever pointers env->p does not hold)
    for (int i = 0; i < 2; ++i) { if (A.snap_grid[i]) (void)hipFree(A.snap_grid[i]); if (A.beam_list[i]) (void)hipFree(A.beam_list[i]); }
    if (A.world_buf[1]) (void)hipFree(A.world_buf[1]);
    if (A.agents_buf[1]) (void)hipFree(A.agents_buf[1]);
    env->aql.reset();
}

// Queues for `chains` chains, fork signals: created once.  false: use hipLaunchKernel.
static bool aql_ready(ssd_env *env, int chains) {
    if (!env->aql) env->aql.reset(new AqlState());
    AqlState &A = *env->aql;
    if (A.tried && !A.ok) return false;
    if (!A.tried) {
        A.tried = true;
        if (!ssd::aql::available(env->device)) return false;
        for (int i = 0; i < AqlState::kForks; ++i) {
            A.fork_sig[i] = ssd::aql::signal_create(0);
            if (!A.fork_sig[i]) return false;
        }
        void *ptr = nullptr;
        if (hipMalloc(&ptr, 8) != hipSuccess || hipMemset(ptr, 0, 8) != hipSuccess) { (void)hipGetLastError(); return false; }
        A.join_counter = static_cast<unsigned long long *>(ptr);
        A.flag_kernarg = ssd::aql::host_kernarg_alloc(env->device, 64);
        if (!A.flag_kernarg) return false;
        std::memcpy(A.flag_kernarg, &A.join_counter, sizeof(void *));
        {
            void *h = nullptr, *d = nullptr;
            if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
            A.timed_out_host = static_cast<uint32_t *>(h); A.timed_out_dev = static_cast<uint32_t *>(d);
            *A.timed_out_host = 0;
            env->host_allocs.push_back(h);
        }
        ptr = nullptr;
        if (hipMalloc(&ptr, 8) != hipSuccess || hipMemset(ptr, 0, 8) != hipSuccess) { (void)hipGetLastError(); return false; }
        A.fork_counter = static_cast<unsigned long long *>(ptr);
        A.fork_kernarg = static_cast<uint8_t *>(ssd::aql::host_kernarg_alloc(env->device, AqlState::kForks * AqlState::kForkBlock));
        if (!A.fork_kernarg || !ssd::aql::lookup(env->device, ssd::wait_kernel_fn(), &A.wait_kernel) || A.wait_kernel.kernarg_size > AqlState::kForkBlock) return false;
        {   // first launches of the two helper kernels now (the runtime resolves a kernel on its first launch: ~50 us), not
            // inside somebody's first short rollout -- and before the pool's first queue is probed (the probe launches one of them)
            ssd::aql::signal_set(A.fork_sig[0], 1);
            ssd::launch_signal_kernel(ssd::aql::signal_value_ptr(A.fork_sig[0]), nullptr);
            ssd::launch_wait_counter_kernel(A.join_counter, 0, nullptr, 0, nullptr, nullptr, nullptr);
            ssd::launch_flag_kernel(A.fork_counter, nullptr); A.forks = 1;
            if (hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipGetLastError(); return false; }
        }
        A.q[0] = ssd::aql::pool_queue(env->device, 0);
        if (!A.q[0]) return false;
        A.nq = 1;
        A.ok = true;
    }
    if (chains > ssd::aql::pool_size(env->device)) return false;   // (more chains than the device's pool has queues: HIP streams)
    while (A.nq < chains) {
        // (a queue the probe turns down is not an error of the path: this call goes through HIP streams, later automatic
        // choices stay within the smaller pool)
        A.q[A.nq] = ssd::aql::pool_queue(env->device, A.nq);
        if (!A.q[A.nq]) return false;
        A.nq++;
    }
    for (int c = 0; c < A.nq; ++c) if (ssd::aql::queue_failed(A.q[c])) { A.ok = false; return false; }
    return true;
}

static inline void aql_wait_consumed(ssd::aql::Queue *q, uint64_t idx_plus_1) {
    // (bounded in practice: the packets in question belong to calls made long ago)
    while (idx_plus_1 && ssd::aql::read_index(q) < idx_plus_1 && !ssd::aql::queue_failed(q)) __builtin_ia32_pause();
}

// The second state buffers and the render side buffers of split rollouts, once per handle: all six or none.
static bool aql_split_buffers(ssd_env *env) {
    AqlState &A = *env->aql;
    if (A.world_buf[1] && A.agents_buf[1] && A.snap_grid[0] && A.snap_grid[1] && A.beam_list[0] && A.beam_list[1]) return true;
    const size_t sizes[6] = {(size_t)env->E * env->S, (size_t)env->E * (env->N ? env->N : 1) * 4, (size_t)env->E * env->S, (size_t)env->E * env->S,
                             (size_t)env->E * 128 * 4, (size_t)env->E * 128 * 4};     // (beam lists: [E][64] + a second [E][64] for steps whose beams took two passes)
    void *buf[6] = {};
    for (int i = 0; i < 6; ++i)
        if (hipMalloc(&buf[i], sizes[i]) != hipSuccess) {
            (void)hipGetLastError();
            for (int j = 0; j < i; ++j) (void)hipFree(buf[j]);
            return false;
        }
    A.world_buf[0] = env->p.world; A.agents_buf[0] = env->p.agents;
    A.world_buf[1] = static_cast<uint8_t *>(buf[0]); A.agents_buf[1] = static_cast<uint32_t *>(buf[1]);
    A.snap_grid[0] = static_cast<uint8_t *>(buf[2]); A.snap_grid[1] = static_cast<uint8_t *>(buf[3]);
    A.beam_list[0] = static_cast<uint32_t *>(buf[4]); A.beam_list[1] = static_cast<uint32_t *>(buf[5]);
    return true;
}

// The kernel-argument set for this call's parameters: found among the cached ones, or built, uploaded (synchronously) and cached.
// Argument block `i` of a chain serves the steps whose index is i modulo key.slots = lcm(output ring, action ring): its output slot
// is i % ring, its action slot i % action_ring.
static AqlState::Set *aql_set(ssd_env *env, const AqlState::Key &key, int chains, const ChainJob *jobs) {
    AqlState &A = *env->aql;
    A.clock++;
    for (auto &st : A.sets) if (st.valid && st.key == key) { st.stamp = A.clock; return &st; }
    AqlState::Set *victim = &A.sets[0];
    for (auto &st : A.sets) { if (!st.valid) { victim = &st; break; } if (st.stamp < victim->stamp) victim = &st; }
    AqlState::Set &st = *victim;
    if (st.valid) for (int c = 0; c < 8 && c < A.nq; ++c) aql_wait_consumed(A.q[c], st.last_use[c]);   // nobody reads the old blocks any more
    st.valid = false;
    constexpr int kKinds = AqlState::kKinds;
    const size_t need = (size_t)chains * key.slots * kKinds * AqlState::kBlock;
    if (need > st.cap) {
        if (st.dev) (void)hipFree(st.dev);
        st.dev = nullptr; st.cap = 0;
        void *ptr = nullptr;
        if (hipMalloc(&ptr, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        st.dev = static_cast<uint8_t *>(ptr); st.cap = need;
    }
    if (key.split && !aql_split_buffers(env)) return nullptr;
    std::vector<uint8_t> host(need, 0);
    for (int c = 0; c < chains; ++c) {
        ChainCursor cur = chain_cursor(env, jobs[c]);
        for (int i = 0; i < key.slots; ++i) {
            const int r = i % key.ring;
            for (int kind = 0; kind < kKinds; ++kind) {
                const int o = kind / AqlState::kKindsPerO, base = kind % AqlState::kKindsPerO;
                const bool split_kind = base >= AqlState::kRn;
                if (split_kind && !key.split) continue;
                if (!key.split && o == 1) continue;              // (without the pair there is one orientation)
                Params p = cur.p;
                if (key.split) { p.world = A.world_buf[o]; p.agents = A.agents_buf[o]; }     // the launch reads buffer o
                p.obs = cur.obs ? cur.obs + (size_t)r * cur.ob : nullptr;
                // (test hook: launches of orientation 1 use the other geometry)
                p.coherent = key.coherent ? ((key.coherent == 2 && o == 1) ? 2u : 1u) : 0u;
                if (base == AqlState::kR || base == AqlState::kRn) {
                    p.mode = ssd::kModeReset; p.rotate = 0; p.num_actions_random = 0; p.actions = nullptr; p.order = nullptr; p.rew = nullptr; p.done = nullptr;
                    // (every reset of a rollout is followed by a step that writes the same observation slot)
                    if (base == AqlState::kRn) p.obs = nullptr;
                } else {
                    p.mode = ssd::kModeStep; p.rotate = 1;
                    cursor_actions(cur, p, (size_t)i);
                    p.rew = cur.rew ? cur.rew + (size_t)r * cur.en : nullptr; p.done = cur.done ? cur.done + (size_t)r * cur.en : nullptr;
                    if (base >= AqlState::kA) {
                        // env waves: read buffer o, write buffer 1 - o, leave their beam marks in list o.  Renderer workgroups of
                        // a kAB launch: the step before produced buffer o (a launch of orientation 1 - o: its marks are in list
                        // 1 - o), its observations go to the ring slot before this one; of a kB launch: the same for THIS slot.
                        uint8_t *obs_r = p.obs;
                        uint8_t *obs_prev = cur.obs ? cur.obs + (size_t)((r + key.ring - 1) % key.ring) * cur.ob : nullptr;
                        p.obs = nullptr;
                        p.snap_mode = 1;
                        p.world_out = A.world_buf[1 - o]; p.agents_out = A.agents_buf[1 - o];
                        p.beam_list = A.beam_list[o]; p.snap = A.snap_grid[o];
                        if (base != AqlState::kA) {
                            p.snap_mode = base == AqlState::kB ? (2 | 4) : (1 | 2);
                            p.beam_list_in = A.beam_list[1 - o]; p.snap_in = A.snap_grid[1 - o];
                            p.obs_b = base == AqlState::kB ? obs_r : obs_prev;
                        }
                    }
                }
                ssd::Launch L;
                if (!ssd::select(p, env->game, &L)) return nullptr;
                AqlState::Geo &g = st.geo[c][kind];
                if (i == 0) {
                    if (!ssd::aql::lookup(env->device, L.fn, &g.k) || g.k.kernarg_size > AqlState::kBlock || g.k.kernarg_size < sizeof(ssd::KernArgs)) return nullptr;
                    g.grid_x = L.grid_x; g.block_x = L.block_x; g.lds = L.lds;
                }
                std::memcpy(host.data() + (((size_t)c * key.slots + i) * kKinds + kind) * AqlState::kBlock, &L.args, sizeof(ssd::KernArgs));
            }
        }
    }
    if (hipMemcpy(st.dev, host.data(), need, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    st.key = key; st.valid = true; st.stamp = A.clock;
    for (auto &u : st.last_use) u = 0;
    return &st;
}

static int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// A rollout through the library's own queues.  Returns SSD_OK, an error, or 1 = "not taken" (the caller then issues the same
// launches through hipLaunchKernel).
static int rollout_aql(ssd_env *env, int chains, const ChainJob *jobs, hipStream_t s) {
    if (!aql_ready(env, chains)) return 1;
    AqlState &A = *env->aql;
    const ChainJob &j0 = jobs[0];
    AqlState::Key key;
    key.obs = j0.obs; key.rew = j0.rew; key.done = j0.done; key.ring = j0.ring; key.f32 = (j0.flags & SSD_OBS_F32) ? 1 : 0;
    key.num_actions = j0.num_actions; key.chains = chains; key.horizon = env->p.horizon;
    key.actions = j0.actions; key.order = j0.actions ? j0.order : nullptr; key.action_ring = j0.actions ? j0.action_ring : 0;
    {   // one argument block per chain, kind and residue of the step index modulo lcm(output ring, action ring)
        const long long ar = key.action_ring > 0 ? key.action_ring : 1;
        const long long l = (long long)key.ring / gcd_i(key.ring, (int)ar) * ar;
        if (l * chains > 2048) return 1;                            // (12 x 512 B per chain and block index)
        key.slots = (int32_t)l;
    }
    key.stamps = env->p.stamps; key.dbg_skip = env->p.dbg_skip;
    // Coherent chains: with a map-specific uint8 kernel (and its write-through observation stores: up to 16 384 envs per launch)
    // state and outputs move with agent-scope accesses only, so the step packets need no release fence (ssd_kernels.hip, COH).
    // SSD_AQL_COHERENT=0 keeps the plain kernels with agent-scope acquire + release on every packet; SSD_AQL_ALTERNATE=1 (test
    // hook) makes every other launch use half the envs per workgroup -- an env then changes workgroup, and with it XCD and L2,
    // from one launch to the next that touches it.
    static const int env_coh = SSD_KNOB("SSD_AQL_COHERENT", 1);
    static const bool alternate = SSD_HOOK("SSD_AQL_ALTERNATE", 0) != 0;
    // (measured, us per step: 2048 envs per launch 5.5 split / 5.9 coherent / 6.2 plain; 2730: 9.7 / 8.6 / 8.7; 5461: 19.1 / 15.2 /
    // 14.3 -- once a launch is several rounds of waves the kernel is bandwidth-bound, and re-reading state from an L2 that still
    // holds it beats fetching it from memory: coherent chains up to 4096 envs per launch, split ones up to 2304)
    const int per_launch = (env->E + chains - 1) / chains;
    // (an explicit action order takes the general kernels: no coherent variant)
    const bool coherent = env_coh != 0 && !key.f32 && !key.order && ssd::fast_profile(env->p, env->game) > 0 && per_launch <= 4096;
    key.coherent = coherent ? (alternate ? 2 : 1) : 0;
    // Split rollouts (coherent chains with observations, 4 steps or more): the wave that steps an env does not render its
    // observations; the NEXT step's launch carries a second set of workgroups that render them while that step is being computed,
    // from the very state that launch steps from (the handle's state lives in a pair of buffers: a launch reads one and writes the
    // other) plus the step's beam marks, which travel as a list; a last launch of renderer workgroups alone delivers the last step's.
    // Still one launch per step and env range (+ 1 per call and per reset inside it), every step's observations in its ring slot
    // when the call's work is done; but the observation phase (1.2 of 5.7 us) is off the chain of dependent launches.  (Two
    // launches per step in one queue -- step, then an observe launch beside the next step -- do not work: kernels of one queue run
    // one after the other on this device even without the barrier bit: two chains' launches in ONE queue take 10.6 us per step, in
    // two queues 5.8.)  SSD_AQL_SPLIT=0 turns it off.
    static const int env_split = SSD_KNOB("SSD_AQL_SPLIT", 1);
    key.split = (coherent && env_split != 0 && j0.obs != nullptr && per_launch <= 2304) ? 1 : 0;   // (the argument set holds both forms' launches)
    const bool split = key.split && j0.n_steps >= 4;
    // (a split set holds its launches for both orientations of the state pair; any other set names the buffer that is current now)
    key.world = key.split ? nullptr : env->p.world;
    AqlState::Set *st = aql_set(env, key, chains, jobs);
    if (!st) return 1;
    // (only now is the call certain to take this path: ssd_rollout_path() must not name a path the launches did not take)
    env->last_path |= SSD_PATH_AQL | (coherent ? SSD_PATH_COHERENT : 0) | (split ? SSD_PATH_SPLIT : 0);
    if (j0.n_steps == 0) return SSD_OK;
    // (the queues are the device's, shared with the other handles on it: one call writes packets at a time)
    std::lock_guard<std::mutex> enqueue_lock(ssd::aql::enqueue_mutex(env->device));
    // FORK: the chains wait (barrier-AND) for a signal that a one-wave kernel on the caller's stream zeroes -- unless the
    // stream has nothing pending, in which case there is nothing to wait for and the first step can start at once
    static const bool always_fork = SSD_HOOK("SSD_AQL_ALWAYS_FORK", 0) != 0;
    // Sync mode: the call itself waits -- for the stream before, for the chains after -- and no kernel waits for another queue's
    // kernel.  Chosen automatically when a profiling tool is attached (aql_sync_mode), or with SSD_AQL_SYNC=1.
    const bool sync_mode = ssd::aql::sync_mode();
    // (measured: an empty barrier packet + doorbell on every chain's queue HERE, so that an idle queue wakes while the host still
    // looks at the stream and writes the first step's packets: 6.65 against 6.62 us per step of the driver's 20-step call, and no
    // better as a rank under torch.distributed.run, where that call follows an RCCL barrier and takes 7.1 - 7.2)
    if (sync_mode) { SSD_HIP(env, hipStreamSynchronize(s)); env->last_path |= SSD_PATH_SYNC; }
    bool stream_idle = sync_mode || (!always_fork && hipStreamQuery(s) == hipSuccess);
    if (!stream_idle && !always_fork) {
        // a marker or an event record just ahead of the call drains within microseconds: look again for a moment before paying
        // for a fork (a kernel launch on the stream + a barrier packet that polls its signal: ~20 us until the first step starts)
        static const int spin_us = SSD_HOOK("SSD_AQL_FORK_SPIN_US", 8);
        const auto t0 = std::chrono::steady_clock::now();
        while (!stream_idle && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us))
            stream_idle = hipStreamQuery(s) == hipSuccess;
    }
    if (!stream_idle) {
        (void)hipGetLastError();                                       // (hipErrorNotReady is not an error here)
        const int slot = A.fork_next;
        A.fork_next = (A.fork_next + 1) % AqlState::kForks;
        for (int c = 0; c < A.nq; ++c) { aql_wait_consumed(A.q[c], A.fork_used[slot][c]); A.fork_used[slot][c] = 0; }
        // Two forms.  (1) A one-wave kernel on the stream bumps a counter in device memory, and every chain starts with a one-wave
        // dispatch that polls it (the join's construct, the other way round).  (0) A kernel on the stream zeroes an HSA signal that a
        // barrier-AND packet at the head of every chain waits on: the command processor then polls the signal, at its own pace
        // (measured: a 20-step call from a stream with a 2 us kernel pending: 13.4 us per step against 6.5 from an idle stream).
        static const int fork_kind = SSD_HOOK("SSD_AQL_FORK_KIND", 1);
        if (fork_kind == 1) {
            A.forks++;
            ssd::launch_flag_kernel(A.fork_counter, s);
            // (nothing is in the library's queues yet: an error here leaves nothing behind)
            if (hipGetLastError() != hipSuccess) { A.forks--; env->err = "fork kernel launch failed"; return SSD_E_DEVICE; }
            // (the wait is bounded -- a minute: what the stream holds ahead of the call may be a whole training step)
            ssd::WaitArgs wa{A.fork_counter, A.forks, ssd::aql::abort_flag_dev(env->device), 6000000000ull, env->p.status, A.timed_out_dev};
            uint8_t *ka = A.fork_kernarg + (size_t)slot * AqlState::kForkBlock;
            std::memcpy(ka, &wa, sizeof wa);
            for (int c = 0; c < chains; ++c) {
                ssd::aql::dispatch(A.q[c], A.wait_kernel, 1, 64, 0, ka, /*barrier=*/true, 0, 0);
                A.fork_used[slot][c] = ssd::aql::write_index(A.q[c]);
            }
        } else {
            ssd::aql::signal_set(A.fork_sig[slot], 1);
            ssd::launch_signal_kernel(ssd::aql::signal_value_ptr(A.fork_sig[slot]), s);
            // (nothing is in the library's queues yet: an error here leaves nothing behind)
            if (hipGetLastError() != hipSuccess) { env->err = "fork kernel launch failed"; return SSD_E_DEVICE; }
            for (int c = 0; c < chains; ++c) {
                ssd::aql::barrier_and(A.q[c], A.fork_sig[slot]);
                A.fork_used[slot][c] = ssd::aql::write_index(A.q[c]);     // (= index of the barrier packet + 1)
            }
        }
        env->last_path |= SSD_PATH_FORKED;
    }
    // fence scopes of the step packets (hsa_fence_scope_t: 0 none, 1 agent, 2 system).  Test-hook build only: SSD_AQL_ACQ /
    // SSD_AQL_REL override them -- for the plain kernels anything weaker than agent / agent gives up the visibility a HIP stream
    // provides, which is why the product does not read them.
    static const int env_acq = SSD_HOOK("SSD_AQL_ACQ", -1);
    static const int env_rel = SSD_HOOK("SSD_AQL_REL", -1);
    // Coherent chains: only the first packet of the call acquires (what the caller's stream did before -- a set_state, another
    // kernel -- may sit in caches); between the chain's own launches nothing is read through a cache that could be stale
    // (-0.14 us per step); the renderer workgroups of a split rollout read state and beam lists with agent-scope loads as well.
    const int kAcq = env_acq >= 0 ? env_acq : (coherent ? 0 : 1), kRel = env_rel >= 0 ? env_rel : (coherent ? 0 : 1);
    const int32_t reset_every = j0.reset_every, step0 = j0.step0, slots = key.slots;
    const bool obs_wb = j0.obs && !key.f32 && (size_t)j0.ring * obs_bytes(env, false) > ((size_t)232 << 20);   // (chain_cursor's obs_nt)
    constexpr int kKinds = AqlState::kKinds;
    auto put = [&](int c, size_t i, int kind, bool barrier, int acq, int rel) {
        const AqlState::Geo &g = st->geo[c][kind];
        ssd::aql::dispatch(A.q[c], g.k, g.grid_x, g.block_x, g.lds, st->dev + (((size_t)c * slots + i) * kKinds + kind) * AqlState::kBlock,
                           barrier, acq, rel);
    };
    // orientation: which buffer of the pair holds the current state (always 0 until the first split rollout of the handle)
    int o = (A.world_buf[1] && env->p.world == A.world_buf[1]) ? 1 : 0;
    const int KO = AqlState::kKindsPerO;
    bool pending = false;                                 // a step's observations are still to be rendered (split rollouts)
    size_t i_prev = 0;
    for (int k = 0; k < j0.n_steps; ++k) {
        const size_t i = (size_t)((step0 + k) % slots);   // argument block of this step (output slot i % ring, action slot i % action_ring)
        const bool reset = reset_every > 0 && (step0 + k) % reset_every == 0;
        for (int c = 0; c < chains; ++c) {
            const int acq = k == 0 ? 1 : kAcq;                                 // (the call's first launch of the chain)
            if (reset) {
                // (a reset works in place: the step before it must have been rendered from that buffer first)
                if (split && pending) put(c, i_prev, o * KO + AqlState::kB, true, kAcq, kRel);
                put(c, i, o * KO + (split ? AqlState::kRn : AqlState::kR), true, acq, kRel);
            }
            const int base = !split ? AqlState::kS : (pending && !reset) ? AqlState::kAB : AqlState::kA;
            // (an output ring past the memory-side cache is written with write-back stores, ssd_kernels.hip select(): one launch per
            // round of the ring releases at agent scope, so that whatever of a slot still sits dirty in an L2 is in memory before
            // the slot's bytes are written again -- possibly through another XCD's L2)
            const int rel = (obs_wb && (step0 + k) % j0.ring == 0) ? 1 : kRel;
            put(c, i, o * KO + base, true, reset ? kAcq : acq, rel);
            if (split && k == j0.n_steps - 1) put(c, i, (1 - o) * KO + AqlState::kB, true, kAcq, kRel);   // (renders the buffer this step wrote)
            // (tried: the call's last step rendering itself, its renderer waves in the env's own workgroup behind a barrier -- with
            // a one-slot ring both write the same bytes -- instead of the renderer-only launch: 6.60 against 6.62 us per step of
            // a 20-step call, no gain)
            // (the doorbell -- an uncached write across the bus, 0.3 - 0.5 us -- after the first two steps, so that the device starts
            // at once, and then after every fourth: the device needs 5 us per step, the host under 1, it never runs dry; the join
            // rings for the rest)
#if SSD_RING_EVERY > 1
            if (!coherent || k < 2 || (k % SSD_RING_EVERY) == SSD_RING_EVERY - 1) ssd::aql::ring(A.q[c]);   // (large launches: every step, as ever)
#else
            ssd::aql::ring(A.q[c]);
#endif
        }
        if (split) { o = 1 - o; pending = true; i_prev = i; }
    }
    if (split) {                                          // the current state now sits in buffer o
        env->p.world = A.world_buf[o]; env->p.agents = A.agents_buf[o];
    }
    // JOIN: every chain ends by bumping the join counter; a one-wave kernel on the caller's stream sleeps until all have
    auto host_join = [&]() -> bool {                      // the synchronous form: the HOST waits for every chain
        bool ok = true;
        for (int c = 0; c < chains; ++c) ssd::aql::ring(A.q[c]);
        for (int c = 0; c < chains; ++c) { ok = ssd::aql::join_and_wait(A.q[c], A.flag_kernarg) && ok; st->last_use[c] = ssd::aql::write_index(A.q[c]); }
        A.joins += (unsigned long long)chains;
        return ok;
    };
    if (sync_mode) {
        if (!host_join()) { env->err = "the HSA runtime reported an error on a dispatch queue"; A.ok = false; return SSD_E_DEVICE; }
        return SSD_OK;
    }
    for (int c = 0; c < chains; ++c) {
        ssd::aql::join(A.q[c], A.flag_kernarg);
        st->last_use[c] = ssd::aql::write_index(A.q[c]);
    }
    A.joins += (unsigned long long)chains;
    // (the stream-side wait is bounded: seconds beyond any healthy call -- ssd_kernels.hip, ssd_wait_counter_kernel)
    unsigned long long timeout_ticks = (unsigned long long)((2.0 + 0.002 * (double)j0.n_steps) * 1e8);
    // (test hooks: wait for a count that never comes / with a short bound -- tests/test_hip_fullsize.py, test_join_wait_is_bounded)
    static const int test_lost = SSD_HOOK("SSD_AQL_TEST_LOST_JOIN", 0), test_timeout_ms = SSD_HOOK("SSD_AQL_TEST_TIMEOUT_MS", 0);
    if (test_timeout_ms > 0) timeout_ticks = (unsigned long long)test_timeout_ms * 100000ull;
    (void)hipGetLastError();                              // (an earlier, unrelated sticky error of the caller's must not be read as ours)
    ssd::launch_wait_counter_kernel(A.join_counter, A.joins + (test_lost ? 1000000ull : 0ull), ssd::aql::abort_flag_dev(env->device), timeout_ticks,
                                    env->p.status, A.timed_out_dev, s);
    if (hipGetLastError() != hipSuccess) {
        // the chains' packets are out and nothing on the stream waits for them: wait here, so that the call's work is over (and
        // its results in place) when the error is returned -- nobody frees or reuses memory under running kernels
        bool ok = true;
        for (int c = 0; c < chains; ++c) ok = ssd::aql::join_and_wait(A.q[c], A.flag_kernarg) && ok;
        A.joins += (unsigned long long)chains;
        env->err = "join kernel launch failed (the call's work was waited for on the host)"; A.ok = false;
        return SSD_E_DEVICE;
    }
    for (int c = 0; c < chains; ++c) if (ssd::aql::queue_failed(A.q[c])) { env->err = "the HSA runtime reported an error on a dispatch queue"; A.ok = false; return SSD_E_DEVICE; }
    return SSD_OK;
}

// ssd_rollout_random and ssd_rollout_actions: the same launches, the actions drawn on the device or read from the caller's ring.
static int rollout(ssd_env *env, const int32_t *actions, const uint8_t *order, int32_t action_ring, int32_t num_actions, int32_t n_steps,
                   int32_t reset_every, int32_t step0, void *obs, int32_t *rew, uint8_t *done, int32_t ring, uint32_t flags, void *stream) {
    if (obs && (reinterpret_cast<uintptr_t>(obs) & 3u)) { env->err = "obs must be 4-byte aligned"; return SSD_E_INVALID; }
    // SSD_ROLLOUT_AUTO: the form of the call is the library's choice.  The fused kernel is the fastest form there is (4096 Harvest
    // envs: 3.5 us per step of a long call against 5.4 through the chains, 4.5 against 6.1 - 6.6 in 20-step calls) wherever it
    // applies: uint8 observations, index action order, and more than one step (a single step is one launch either way, and the
    // per-step kernels are the shorter ones).  ssd_rollout_path() reports the form that ran.
    if (flags & SSD_ROLLOUT_AUTO) {
        flags &= ~(uint32_t)SSD_ROLLOUT_AUTO;
        if (!(flags & SSD_OBS_F32) && !order && n_steps >= 2) flags |= SSD_ROLLOUT_FUSED;
    }
    if ((flags & SSD_ROLLOUT_FUSED) && (flags & SSD_OBS_F32)) { env->err = "the fused rollout kernel writes uint8 observations"; return SSD_E_INVALID; }
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != env->device) SSD_HIP(env, hipSetDevice(env->device));
    }
    if (const int rc = after_timeout(env)) return rc;               // (an earlier call's join gave up: reported once, here or in ssd_synchronize)
    hipStream_t s = static_cast<hipStream_t>(stream);
    env->last_stream = s; env->last_stream_set = true;
    uint8_t *o = static_cast<uint8_t *>(obs);
    // Envs are independent, so a rollout is as many independent launch chains as we like.  Two chains in two queues keep the
    // GPU busy while the other chain's kernel drains and the next one is dispatched -- the ~2 us per launch that a single
    // chain of dependent kernels cannot hide.  SSD_ROLLOUT_CHAINS overrides.
    static const int forced = SSD_KNOB("SSD_ROLLOUT_CHAINS", 0);
    auto by_size = [&]() { return env->E < 2048 ? 1 : (env->E >= 6144 && env->E <= 24576) ? 3 : 2; };   // measured: profiles/r01_sweep_envs.txt
    int chains = env->rollout_chains > 0 ? env->rollout_chains : forced > 0 ? forced : by_size();
    if ((flags & SSD_ROLLOUT_FUSED) && env->rollout_chains <= 0 && forced <= 0) chains = 1;   // one launch already covers the whole rollout
    const bool automatic = env->rollout_chains <= 0 && forced <= 0;
    // (an automatic choice stays within the library's own dispatch queues -- ssd_aql.hip, pool_size() -- and where the probe found
    // the process past the hardware-queue cliff already, every further active queue or stream would be time-sliced against the
    // others: one chain on the caller's own stream)
    auto clamp_chains = [&]() {
        if (automatic && ssd::aql::pool_size(env->device) >= 1 && chains > ssd::aql::pool_size(env->device)) chains = ssd::aql::pool_size(env->device);
        if (automatic && ssd::aql::over_the_cliff(env->device)) chains = 1;
        if (chains > 8) chains = 8;
        if (chains > env->E) chains = env->E;
        if (chains < 1) chains = 1;
        env->last_path = (chains << 8) | ((flags & SSD_ROLLOUT_FUSED) ? SSD_PATH_FUSED : 0);
    };
    clamp_chains();
    auto range = [&](int c) { return (int)(((long long)env->E * c) / chains); };
    auto job_of = [&](int c, hipStream_t cs) {
        ChainJob j;
        j.chain = c; j.e_begin = range(c); j.e_end = range(c + 1);
        j.num_actions = num_actions; j.n_steps = n_steps; j.reset_every = reset_every; j.step0 = step0; j.ring = ring;
        j.actions = actions; j.order = order; j.action_ring = action_ring;
        j.obs = o; j.rew = rew; j.done = done; j.flags = flags; j.s = cs;
        return j;
    };
    if (!(flags & SSD_ROLLOUT_FUSED)) {
        // the library's own dispatch path: the same launches as below, written as AQL packets into its own queues
        for (int attempt = 0; attempt < 2; ++attempt) {
            ChainJob jobs[8];
            for (int c = 0; c < chains; ++c) jobs[c] = job_of(c, s);
            const int pool_before = ssd::aql::pool_size(env->device);
            const int rc = rollout_aql(env, chains, jobs, s);
            env->last_path |= ssd::aql::pool_report(env->device);
            if (rc <= 0) return rc;
            env->last_path &= ~(SSD_PATH_AQL | SSD_PATH_COHERENT | SSD_PATH_SPLIT | SSD_PATH_SYNC | SSD_PATH_FORKED);
            // (this call created a queue that failed its probe: an automatic chain count is chosen again for the pool that is left)
            if (!automatic || ssd::aql::pool_size(env->device) >= pool_before) break;
            chains = by_size();
            clamp_chains();
            env->last_path |= ssd::aql::pool_report(env->device);
        }
    }
    if (chains <= 1) {
        int rc = rollout_chain(env, job_of(0, s));
        if (rc) env->err = "kernel launch failed in a rollout call";
        return rc;
    }
    while ((int)env->chain_streams.size() < chains - 1) {
        hipStream_t ns; hipEvent_t ne;
        SSD_HIP(env, hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
        SSD_HIP(env, hipEventCreateWithFlags(&ne, hipEventDisableTiming));
        env->chain_streams.push_back(ns); env->chain_events.push_back(ne);
    }
    if (!env->fork_event) SSD_HIP(env, hipEventCreateWithFlags(&env->fork_event, hipEventDisableTiming));
    // Who enqueues.  The chains' launches either come from this thread, step by step across the chains (no other thread
    // involved: a short call costs its launches and nothing else), or each extra chain from its persistent worker thread
    // (long calls: two threads enqueue ~10 % faster than one).  (Test-hook build: SSD_ROLLOUT_THREADS=0 / 1 forces one or the other.)
    static const int env_threads = SSD_HOOK("SSD_ROLLOUT_THREADS", -1);
    static const int inline_steps = SSD_HOOK("SSD_ROLLOUT_INLINE_STEPS", 256);
    bool threads = !(flags & SSD_ROLLOUT_FUSED) && n_steps > inline_steps;
    if (env_threads == 0) threads = false;
    if (env_threads == 1) threads = true;
    // fork: the extra chains start after whatever the caller's stream holds so far
    SSD_HIP(env, hipEventRecord(env->fork_event, s));
    for (int c = 1; c < chains; ++c) SSD_HIP(env, hipStreamWaitEvent(env->chain_streams[c - 1], env->fork_event, 0));
    int rc_all = SSD_OK;
    if (threads) {
        while ((int)env->workers.size() < chains - 1) {
            env->workers.emplace_back(new ChainWorker());
            ChainWorker *w = env->workers.back().get();
            w->th = std::thread(chain_worker_main, env, w);
        }
        for (int c = 1; c < chains; ++c) post_job(env->workers[c - 1].get(), job_of(c, env->chain_streams[c - 1]));
        rc_all = rollout_chain(env, job_of(0, s));
        for (int c = 1; c < chains; ++c) { const int rc = wait_job(env->workers[c - 1].get()); if (rc && !rc_all) rc_all = rc; }
    } else if (flags & SSD_ROLLOUT_FUSED) {
        for (int c = 0; c < chains; ++c) { const int rc = rollout_chain(env, job_of(c, c ? env->chain_streams[c - 1] : s)); if (rc && !rc_all) rc_all = rc; }
    } else {
        ChainCursor cur[8];
        for (int c = 0; c < chains; ++c) cur[c] = chain_cursor(env, job_of(c, c ? env->chain_streams[c - 1] : s));
        for (int k = 0; k < n_steps; ++k)
            for (int c = 0; c < chains; ++c) chain_launch_step(env, cur[c], k);
        if (hipGetLastError() != hipSuccess) rc_all = SSD_E_DEVICE;
    }
    // join: the caller's stream continues after every chain
    for (int c = 1; c < chains; ++c) {
        SSD_HIP(env, hipEventRecord(env->chain_events[c - 1], env->chain_streams[c - 1]));
        SSD_HIP(env, hipStreamWaitEvent(s, env->chain_events[c - 1], 0));
    }
    if (rc_all) env->err = "kernel launch failed in a rollout call";
    return rc_all;
}

int ssd_rollout_random(ssd_env *env, int32_t num_actions, int32_t n_steps, int32_t reset_every, int32_t step0,
                       void *obs, int32_t *rew, uint8_t *done, int32_t ring, uint32_t flags, void *stream) {
    if (!env || n_steps < 0 || reset_every < 0 || step0 < 0 || ring < 1) return SSD_E_INVALID;
    if (flags & SSD_HOST_PTRS) { env->err = "ssd_rollout_random takes device pointers"; return SSD_E_INVALID; }
    const int na = env->game == SSD_GAME_HARVEST ? 8 : 9;
    if (num_actions < 1 || num_actions > na) { env->err = "num_actions outside the game's Discrete(n)"; return SSD_E_INVALID; }
    return rollout(env, nullptr, nullptr, 0, num_actions, n_steps, reset_every, step0, obs, rew, done, ring, flags, stream);
}

int ssd_rollout_actions(ssd_env *env, const int32_t *actions, const uint8_t *order, int32_t action_ring, int32_t n_steps, int32_t reset_every,
                        int32_t step0, void *obs, int32_t *rew, uint8_t *done, int32_t ring, uint32_t flags, void *stream) {
    if (!env || n_steps < 0 || reset_every < 0 || step0 < 0 || ring < 1 || action_ring < 1) return SSD_E_INVALID;
    if (flags & SSD_HOST_PTRS) { env->err = "ssd_rollout_actions takes device pointers"; return SSD_E_INVALID; }
    if (!actions && env->N > 0) { env->err = "actions is null"; return SSD_E_INVALID; }
    return rollout(env, actions, order, action_ring, 0, n_steps, reset_every, step0, obs, rew, done, ring, flags, stream);
}

int ssd_profiler_attached(void) { return ssd::aql::tool_attached_now(); }

int ssd_observe(ssd_env *env, void *obs, uint32_t flags, void *stream) {
    if (!env || !obs) return SSD_E_INVALID;
    return run(env, ssd::kModeObserve, nullptr, nullptr, nullptr, 0, nullptr, obs, nullptr, nullptr,
               (flags & SSD_NO_ROTATE) ? 0 : 1, flags, stream);
}

int ssd_get_state(ssd_env *env, int8_t *world, int8_t *beam, int16_t *pos, uint8_t *orient, uint32_t *episode, uint32_t *t) {
    if (!env) return SSD_E_INVALID;
    SSD_HIP(env, hipSetDevice(env->device));
    SSD_HIP(env, hipDeviceSynchronize());
    const int E = env->E, N = env->N, S = env->S, WP = env->WP;
    if (world || beam) {
        std::vector<uint8_t> buf((size_t)E * S);
        if (world) {
            SSD_HIP(env, hipMemcpy(buf.data(), env->p.world, buf.size(), hipMemcpyDeviceToHost));
            unpack_grid(env, buf, world);
        }
        if (beam) {
            if (!env->keep_beams) { env->err = "beam overlay is only kept with keep_beams"; return SSD_E_INVALID; }
            SSD_HIP(env, hipMemcpy(buf.data(), env->p.beam, buf.size(), hipMemcpyDeviceToHost));
            unpack_grid(env, buf, beam);
        }
    }
    if (pos || orient) {
        std::vector<uint32_t> ag((size_t)E * N);
        if (!ag.empty()) SSD_HIP(env, hipMemcpy(ag.data(), env->p.agents, ag.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ag.size(); ++i) {
            const uint32_t cell = ag[i] & 0xFFFFu;
            if (pos) { pos[2 * i] = (int16_t)(cell / WP); pos[2 * i + 1] = (int16_t)(cell % WP); }
            if (orient) orient[i] = (uint8_t)((ag[i] >> 16) & 3u);
        }
    }
    if (episode || t) {
        std::vector<uint4> hdr(E);
        SSD_HIP(env, hipMemcpy(hdr.data(), env->p.hdr, hdr.size() * sizeof(uint4), hipMemcpyDeviceToHost));
        for (int e = 0; e < E; ++e) { if (episode) episode[e] = hdr[e].z; if (t) t[e] = hdr[e].y; }
    }
    return SSD_OK;
}

int ssd_get_waste_count(ssd_env *env, uint32_t *waste_count) {
    if (!env || !waste_count) return SSD_E_INVALID;
    SSD_HIP(env, hipSetDevice(env->device));
    SSD_HIP(env, hipDeviceSynchronize());
    std::vector<uint4> hdr(env->E);
    SSD_HIP(env, hipMemcpy(hdr.data(), env->p.hdr, hdr.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    for (int e = 0; e < env->E; ++e) waste_count[e] = hdr[e].w & 0xFFFFu;   // (upper half: the grid's current count, kernel-internal)
    return SSD_OK;
}

static uint32_t host_mix32(uint32_t x) {
    x ^= x >> 17; x *= 0xED5AD4BBu; x ^= x >> 11; x *= 0xAC4C1B51u; x ^= x >> 15; x *= 0x31848BABu; x ^= x >> 14;
    return x;
}

int ssd_set_state(ssd_env *env, const int8_t *world, const int8_t *beam, const int16_t *pos, const uint8_t *orient,
                  const uint32_t *episode, const uint32_t *t) {
    if (!env) return SSD_E_INVALID;
    SSD_HIP(env, hipSetDevice(env->device));
    SSD_HIP(env, hipDeviceSynchronize());
    const int E = env->E, N = env->N, hw = env->H * env->W, W = env->W, H = env->H, WP = env->WP;
    if (world || beam) {
        std::vector<uint8_t> buf;
        // glyphs index the 128-entry colour table on the device: cells must be 7-bit ASCII
        for (size_t i = 0; world && i < (size_t)E * hw; ++i)
            if (world[i] <= 0) { env->err = "world cells must be 7-bit ASCII (1..127)"; return SSD_E_INVALID; }
        for (size_t i = 0; beam && i < (size_t)E * hw; ++i)
            if (beam[i] < 0) { env->err = "beam cells must be 0 or 7-bit ASCII"; return SSD_E_INVALID; }
        if (world) {
            // the wall border is what keeps agents and beams inside the grid (the kernel has no bounds tests)
            for (int e = 0; e < E; ++e)
                for (int r = 0; r < H; ++r)
                    for (int c = 0; c < W; c += (r == 0 || r == H - 1 || c == W - 1) ? 1 : W - 1)
                        if (world[((size_t)e * H + r) * W + c] != '@') { env->err = "the map border must stay '@'"; return SSD_E_INVALID; }
            pack_grid(env, world, buf, kVoid);
            SSD_HIP(env, hipMemcpy(env->p.world, buf.data(), buf.size(), hipMemcpyHostToDevice));
        }
        if (beam) {
            if (!env->keep_beams) { env->err = "beam overlay is only kept with keep_beams"; return SSD_E_INVALID; }
            pack_grid(env, beam, buf, 0);
            SSD_HIP(env, hipMemcpy(env->p.beam, buf.data(), buf.size(), hipMemcpyHostToDevice));
        }
    }
    std::vector<uint8_t> share;
    if ((pos || orient) && N > 0) {
        std::vector<uint32_t> ag((size_t)E * N);
        SSD_HIP(env, hipMemcpy(ag.data(), env->p.agents, ag.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ag.size(); ++i) {
            uint32_t cell = ag[i] & 0xFFFFu, o = (ag[i] >> 16) & 3u;
            if (pos) {
                const int r = pos[2 * i], c = pos[2 * i + 1];
                // agents live strictly inside the wall border; a position on the border could step out of the grid
                if (r < 1 || r >= H - 1 || c < 1 || c >= W - 1) { env->err = "agent position must be inside the map's wall border"; return SSD_E_INVALID; }
                cell = (uint32_t)(r * WP + c);
            }
            if (orient) { if (orient[i] > 3) { env->err = "orientation code must be 0..3"; return SSD_E_INVALID; } o = orient[i]; }
            ag[i] = cell | (o << 16);
        }
        SSD_HIP(env, hipMemcpy(env->p.agents, ag.data(), ag.size() * 4, hipMemcpyHostToDevice));
        if (pos) {                                  // hdr.w bit 31: two agents of the env share a cell (the kernel's move phase asks)
            share.assign(E, 0);
            for (int e = 0; e < E; ++e)
                for (int i = 0; i < N; ++i)
                    for (int j = i + 1; j < N; ++j)
                        if (((ag[(size_t)e * N + i] ^ ag[(size_t)e * N + j]) & 0xFFFFu) == 0) share[e] = 1;
        }
    }
    const bool recount = world && env->game == SSD_GAME_CLEANUP;      // the kernel keeps the grid's 'H' count in hdr.w
    if (episode || t || recount || !share.empty()) {
        std::vector<uint4> hdr(E);
        SSD_HIP(env, hipMemcpy(hdr.data(), env->p.hdr, hdr.size() * sizeof(uint4), hipMemcpyDeviceToHost));
        for (int e = 0; e < E; ++e) {
            if (recount) {
                uint32_t n = 0;
                for (int i = 0; i < hw; ++i) n += world[(size_t)e * hw + i] == 'H';
                hdr[e].w = (hdr[e].w & 0x8000FFFFu) | (n << 16);
            }
            if (!share.empty()) hdr[e].w = (hdr[e].w & 0x7FFFFFFFu) | ((uint32_t)share[e] << 31);
            if (t) hdr[e].y = t[e];
            if (episode) {
                hdr[e].z = episode[e];
                uint32_t h = 0x243F6A88u;         // env_key(seed, env, episode), prng.py
                h = host_mix32(h ^ (uint32_t)env->seed);
                h = host_mix32(h ^ (uint32_t)(env->seed >> 32));
                h = host_mix32(h ^ (env->env_base + (uint32_t)e));
                h = host_mix32(h ^ episode[e]);
                hdr[e].x = h;
            }
        }
        SSD_HIP(env, hipMemcpy(env->p.hdr, hdr.data(), hdr.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    return SSD_OK;
}

int ssd_render_frames(ssd_env *env, int32_t e_begin, int32_t count, uint8_t *rgb, uint32_t flags, void *stream) {
    if (!env || !rgb || e_begin < 0 || count < 0 || (int64_t)e_begin + count > env->E) return SSD_E_INVALID;
    if (flags & ~(uint32_t)SSD_HOST_PTRS) { env->err = "ssd_render_frames: only the host-pointer flag is meaningful here"; return SSD_E_INVALID; }
    if (count == 0) return SSD_OK;
    SSD_HIP(env, hipSetDevice(env->device));
    const size_t frame = (size_t)env->H * env->W * 3;
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint8_t *dst = rgb;
    if (flags & SSD_HOST_PTRS) {                       // frames are rendered into a device buffer of the handle and copied back
        if (env->st_rgb_frames < (size_t)count) {
            void *ptr = nullptr;
            hipError_t e = hipMalloc(&ptr, (size_t)count * frame);
            if (e != hipSuccess) { env->err = std::string("hipMalloc: ") + hipGetErrorString(e); return SSD_E_NOMEM; }
            if (env->st_rgb) {
                SSD_HIP(env, hipDeviceSynchronize());
                for (auto &a : env->allocs) if (a == env->st_rgb) a = ptr;
                SSD_HIP(env, hipFree(env->st_rgb));
            } else {
                env->allocs.push_back(ptr);
            }
            env->st_rgb = static_cast<uint8_t *>(ptr);
            env->st_rgb_frames = (size_t)count;
        }
        dst = env->st_rgb;
    }
    for (int32_t k = 0; k < count; k += 32768) {       // gridDim.y carries the env: at most 65535 per launch
        const int32_t n = count - k < 32768 ? count - k : 32768;
        ssd::launch_render_full(env->p, e_begin + k, n, dst + (size_t)k * frame, s);
    }
    SSD_HIP(env, hipGetLastError());
    if (flags & SSD_HOST_PTRS) {
        SSD_HIP(env, hipMemcpyAsync(rgb, dst, (size_t)count * frame, hipMemcpyDeviceToHost, s));
        SSD_HIP(env, hipStreamSynchronize(s));
    }
    return SSD_OK;
}

int ssd_render_full(ssd_env *env, int32_t e, uint8_t *rgb) {
    if (!env || !rgb || e < 0 || e >= env->E) return SSD_E_INVALID;
    SSD_HIP(env, hipSetDevice(env->device));
    SSD_HIP(env, hipDeviceSynchronize());              // (this entry point takes no stream: order it after everything enqueued)
    return ssd_render_frames(env, e, 1, rgb, SSD_HOST_PTRS, nullptr);
}

int ssd_agent_action_obs(ssd_env *env, const int32_t *actions, const uint8_t *done_mask, int64_t *other_actions, int64_t *visible,
                         uint32_t flags, void *stream) {
    if (!env) return SSD_E_INVALID;
    if (flags & ~(uint32_t)SSD_HOST_PTRS) { env->err = "ssd_agent_action_obs: only the host-pointer flag is meaningful here"; return SSD_E_INVALID; }
    const int E = env->E, N = env->N;
    if (N < 2 || (!other_actions && !visible)) return SSD_OK;           // (N - 1 == 0 columns)
    SSD_HIP(env, hipSetDevice(env->device));
    ssd::AgentOrder ord;
    {   // ids 'agent-<i>' sorted as strings (harvest.py:50, map_env.py:202)
        std::vector<std::pair<std::string, int>> ids;
        for (int i = 0; i < N; ++i) ids.emplace_back("agent-" + std::to_string(i), i);
        std::sort(ids.begin(), ids.end());
        std::memset(&ord, 0, sizeof(ord));
        for (int r = 0; r < N; ++r) { ord.sorted[r] = (uint8_t)ids[r].second; ord.rank[ids[r].second] = (uint8_t)r; }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t en = (size_t)E * N, out_n = en * (size_t)(N - 1);
    if (!(flags & SSD_HOST_PTRS)) {
        ssd::launch_agent_action_obs(actions, done_mask, reinterpret_cast<long long *>(other_actions), reinterpret_cast<long long *>(visible), ord, E, N, stream);
        SSD_HIP(env, hipGetLastError());
        return SSD_OK;
    }
    // host arrays: staged through scratch device buffers of this call (a convenience surface, not a fast path)
    int32_t *d_act = nullptr; uint8_t *d_mask = nullptr; long long *d_out = nullptr, *d_vis = nullptr;
    auto cleanup = [&]() { if (d_act) (void)hipFree(d_act); if (d_mask) (void)hipFree(d_mask); if (d_out) (void)hipFree(d_out); if (d_vis) (void)hipFree(d_vis); };
    auto fail = [&](const char *what) { env->err = what; cleanup(); return SSD_E_DEVICE; };
    if (actions) { if (hipMalloc(&d_act, en * 4) != hipSuccess || hipMemcpyAsync(d_act, actions, en * 4, hipMemcpyHostToDevice, s) != hipSuccess) return fail("staging actions"); }
    if (done_mask) { if (hipMalloc(&d_mask, en) != hipSuccess || hipMemcpyAsync(d_mask, done_mask, en, hipMemcpyHostToDevice, s) != hipSuccess) return fail("staging done_mask"); }
    if (other_actions && hipMalloc(&d_out, out_n * 8) != hipSuccess) return fail("staging other_actions");
    if (visible && hipMalloc(&d_vis, out_n * 8) != hipSuccess) return fail("staging visible");
    ssd::launch_agent_action_obs(d_act, d_mask, d_out, d_vis, ord, E, N, stream);
    if (hipGetLastError() != hipSuccess) return fail("kernel launch");
    if (other_actions && hipMemcpyAsync(other_actions, d_out, out_n * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return fail("copy back");
    if (visible && hipMemcpyAsync(visible, d_vis, out_n * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return fail("copy back");
    if (hipStreamSynchronize(s) != hipSuccess) return fail("synchronize");
    cleanup();
    return SSD_OK;
}

int ssd_set_horizon(ssd_env *env, int32_t horizon) {
    if (!env || horizon < 0) return SSD_E_INVALID;
    env->p.horizon = horizon;
    return SSD_OK;
}

int ssd_rollout_path(const ssd_env *env) { return env ? env->last_path : SSD_E_INVALID; }

int ssd_set_rollout_chains(ssd_env *env, int32_t chains) {
    if (!env || chains < 0 || chains > 8) return SSD_E_INVALID;
    env->rollout_chains = chains;
    return SSD_OK;
}

int ssd_potential_waste_area(const ssd_env *env) { return env ? env->potential_waste : SSD_E_INVALID; }

int ssd_device_status(ssd_env *env, uint32_t *status, int clear) {
    if (!env || !status) return SSD_E_INVALID;
    SSD_HIP(env, hipSetDevice(env->device));
    SSD_HIP(env, hipDeviceSynchronize());
    SSD_HIP(env, hipMemcpy(status, env->p.status, 4, hipMemcpyDeviceToHost));
    if (clear) SSD_HIP(env, hipMemset(env->p.status, 0, 4));
    return SSD_OK;
}

#ifdef SSD_STAMPS
int ssd_debug_set_skip(ssd_env *env, uint32_t mask) {
    if (!env) return SSD_E_INVALID;
    env->p.dbg_skip = mask;
    return SSD_OK;
}
// Diagnostic library only: a one-wave kernel writes the stamps' 100 MHz clock into a host-visible word `iters` times.
int ssd_debug_clock(ssd_env *env, void *host_word, int iters, void *stream) {
    if (!env || !host_word) return SSD_E_INVALID;
    ssd::launch_clock_kernel(static_cast<unsigned long long *>(host_word), iters, stream);
    return hipGetLastError() == hipSuccess ? SSD_OK : SSD_E_DEVICE;
}
// Diagnostic library only (make stamps): device buffer [E][16] u64 that the kernel fills with cycle stamps.
int ssd_debug_set_stamps(ssd_env *env, void *dev_ptr) {
    if (!env) return SSD_E_INVALID;
    env->p.stamps = static_cast<unsigned long long *>(dev_ptr);
    return SSD_OK;
}
#endif

int ssd_synchronize(ssd_env *env) {
    if (!env) return SSD_E_INVALID;
    SSD_HIP(env, hipSetDevice(env->device));
    SSD_HIP(env, hipDeviceSynchronize());
    return after_timeout(env);      // (SSD_E_DEVICE once if a rollout call's join wait gave up meanwhile: the queues are drained first)
}

}  // extern "C"
