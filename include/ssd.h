/* include/ssd.h -- C ABI of libssd_hip.so, the MI355X-native engine for the
 * MapEnv.step() hot path of the Harvest / Cleanup gridworlds.
 *
 * The reference is pure Python: it has no FFI.  The boundary this ABI replaces is the
 * RLlib MultiAgentEnv duck type that `MapEnv` implements (reference
 * social_dilemmas/envs/map_env.py:60,152,214) -- every entry point below cites the
 * reference method it stands in for.  The host-side mirror of that interface
 * (sequential_social_dilemma_games_amd/map_env.py) binds these symbols through ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: every call returns 0 on success or a negative SSD_E_* code and never
 * throws; the caller owns every buffer passed in (the engine only allocates its own
 * state); one handle = one device + one caller-chosen stream per call; calls on one
 * handle are not re-entrant, distinct handles are independent (a host-pointer call first waits for the stream of the handle's
 * last device-pointer call when that is another stream, so mixing the two styles keeps program order); there is no global state
 * (the reference's module-global RNGs become per-handle seed + counters).  There is no
 * CPU backend: without a usable HIP device ssd_create fails with SSD_E_DEVICE.
 *
 * Batched layouts (E = num_envs of the handle, N = num_agents, V = 2*view_len+1):
 *   actions i32 [E,N]   -1 = agent absent from the action dict this step
 *   order   u8  [E,N]   agent indices in action-dict order, 0xFF-terminated; NULL = index order
 *   obs     u8  [E,N,V,V,3]   RGB; the reference's float64 obs is (u8 - 128.0) / 255.0
 *           f32 [E,N,V,V,3]   with SSD_OBS_F32: float32(that float64 value), i.e. the reference observation cast to
 *                             its declared Box(dtype=float32) space (harvest.py:39-40), NHWC per agent
 *   rew     i32 [E,N]
 *   done    u8  [E,N]   0 (agent.py:174-175,209-210) unless a horizon is set (ssd_set_horizon)
 */
#ifndef SSD_H
#define SSD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSD_ABI_VERSION 3   /* 3: SSD_ROLLOUT_AUTO; SSD_STEP_CHAINS removed.  2: ssd_rollout_actions, ssd_profiler_attached; SSD_ROLLOUT_PIPELINED removed */

enum {
    SSD_OK = 0,
    SSD_E_INVALID = -1,   /* bad argument / configuration (open map, too many agents, ...) */
    SSD_E_DEVICE = -2,    /* no usable HIP device, or a HIP call failed (see ssd_last_error) */
    SSD_E_NOMEM = -3,
    SSD_E_STATE = -4      /* device status word is non-zero (see ssd_device_status) */
};

enum { SSD_GAME_HARVEST = 0, SSD_GAME_CLEANUP = 1 };

/* flags of the stepping calls */
enum {
    SSD_HOST_PTRS = 1u << 0, /* actions/order/obs/rew/done are host memory: the engine stages them
                                and the call returns after the results have landed.  Without it
                                they are device pointers on the handle's device and the call only
                                enqueues work on `stream`. */
    SSD_NO_ROTATE = 1u << 1, /* ssd_observe only: reset-form observation (map_env.py:239-240) */
    SSD_ROLLOUT_FUSED = 1u << 3, /* ssd_rollout_random only: ONE kernel launch for the whole call -- every env stays in LDS and
                                registers across its n_steps steps and only the per-step outputs (and, at the end, the state)
                                go to HBM.  Same results; no per-step launch, state reload or write-back.  uint8 obs only. */
    SSD_AUTO_RESET = 1u << 4,   /* ssd_step / ssd_step_random: an env whose step reaches the horizon (ssd_set_horizon; its done flags
                                are 1) starts its next episode in the same launch: MapEnv.reset (map_env.py:214-249) is applied to it
                                and its observation rows are the reset's (unrotated, :239-240), as if ssd_reset had been called with
                                the done flags as the mask.  uint8 obs only. */
    /* (1u << 5 was SSD_ROLLOUT_PIPELINED, rounds 1-2: launches of consecutive steps overlapped through per-env pass counters.
       It bought nothing once the library dispatched through its own queues -- 4.44 against 4.50 us per step at 2048 envs -- and
       was the one mode in which kernels waited on other kernels' flags: removed.  The bit is ignored.) */
    /* (1u << 6 was SSD_STEP_CHAINS, ABI 2: ssd_step dispatched like a one-step ssd_rollout_actions call.  Measured at the named
       batch -- 4096 envs -- it cost 21.7 us per call against 7.3 for the plain launch: a single launch has no chain of dependent
       launches to hide, and the fork / join cost more than two concurrent half-launches save.  Removed; the bit is ignored.) */
    SSD_ROLLOUT_AUTO = 1u << 7, /* ssd_rollout_random / ssd_rollout_actions: let the library pick the form of the call.  uint8
                                observations, index action order and n_steps >= 2 take the fused kernel (SSD_ROLLOUT_FUSED: 3.5 us per
                                4096-env step against 5.4 through the chains); anything else is dispatched as without the flag.
                                Same results either way; ssd_rollout_path() says which form ran. */
    SSD_OBS_F32 = 1u << 2    /* obs points at float32 [E,N,V,V,3] instead of uint8: the normalisation of map_env.py:199
                                fused into the kernel (4x the observation bytes; a separate, slower mode) */
};

/* bits of the device status word */
enum {
    SSD_ST_BAD_ACTION = 1u << 0,  /* action id outside the game's Discrete(n): KeyError in agent.action_map */
    SSD_ST_NO_SPAWN = 1u << 1,    /* not enough spawn points (assert at map_env.py:661) */
    SSD_ST_MOVE_LOOKUP = 1u << 2, /* agent_by_pos lookup miss (would be a KeyError at map_env.py:506) */
    SSD_ST_WAIT_TIMEOUT = 1u << 3 /* a rollout call's stream-side wait for the library's queues gave up (seconds: the queues' kernels
                                     never ran -- e.g. a tool that runs kernels one at a time, attached in a way the library did not
                                     notice); the call's outputs are not in place.  See SSD_AQL_SYNC below.  The condition is STICKY:
                                     the handle's next rollout call, or ssd_synchronize, drains the library's queues on the host
                                     (bounded), returns SSD_E_DEVICE once, and the handle steps through hipLaunchKernel from then
                                     on.  Until one of them has returned, the timed-out call's buffers must not be freed or reused:
                                     its launches may still be writing them */
};

typedef struct ssd_env ssd_env;

typedef struct ssd_config {
    uint32_t struct_size;        /* sizeof(ssd_config), for ABI evolution */
    int32_t game;                /* SSD_GAME_*: HarvestEnv (harvest.py:18) or CleanupEnv (cleanup.py:30) */
    int32_t height, width;       /* ascii_map shape (map_env.py:132-150) */
    const char *base_map;        /* height*width ASCII bytes, row-major, wall-closed */
    int32_t num_envs;            /* E: independent env copies held by this handle */
    int32_t num_agents;          /* N (ctor arg num_agents, map_env.py:62); 0..64 */
    int32_t view_len;            /* HARVEST_VIEW_SIZE / CLEANUP_VIEW_SIZE = 7 (harvest.py:15, cleanup.py:22) */
    int32_t beam_len;            /* ACTIONS['FIRE'] = ACTIONS['CLEAN'] = 5 (harvest.py:11, cleanup.py:11-12) */
    uint64_t seed;               /* replaces np.random.seed / random.seed */
    uint32_t env_index_base;     /* global index of env 0 of this handle (multi-GPU shards) */
    int32_t device_id;           /* HIP device ordinal */
    int32_t keep_beams;          /* 1: persist the beam overlay (map_env.py:86 beam_pos) between steps so that
                                    ssd_get_state / ssd_observe / ssd_render_full see it; 0: beams live only
                                    inside the step that draws them (they are already in that step's obs) */
    const uint8_t *color_lut;    /* 128*3 glyph -> RGB (map_env.py:24-41, cleanup.py:15-18); NULL = defaults */
    /* rand < p thresholds as ceil(p * 2^32); NULL = derived from the reference constants */
    const uint64_t *harvest_thresholds;        /* [4]  SPAWN_PROB[min(n,3)] (harvest.py:13,100) */
    const uint64_t *cleanup_apple_thresholds;  /* [potential_waste_area+1], index = current #'H' (cleanup.py:156-171) */
    const uint64_t *cleanup_waste_thresholds;  /* same indexing */
} ssd_config;

/* MapEnv.__init__ (map_env.py:62-102) + HarvestEnv/CleanupEnv.__init__ (harvest.py:20-28, cleanup.py:32-66)
 * for E env copies.  Unlike the reference constructor it does not spawn agents: call ssd_reset first. */
int ssd_create(const ssd_config *cfg, ssd_env **out);
int ssd_destroy(ssd_env *env);

/* MapEnv.reset (map_env.py:214-249) on the envs selected by env_mask (u8 [E], NULL = all; same memory
 * kind as obs).  obs may be NULL. */
int ssd_reset(ssd_env *env, const uint8_t *env_mask, void *obs, uint32_t flags, void *stream);

/* MapEnv.step (map_env.py:152-212) on every env.  obs / rew / done may be NULL. */
int ssd_step(ssd_env *env, const int32_t *actions, const uint8_t *order, void *obs, int32_t *rew,
             uint8_t *done, uint32_t flags, void *stream);

/* The random-action rollout step of rollout.py:62-70: actions are drawn on the device, uniformly over
 * Discrete(num_actions), from the ACTION stream; actions_out (i32 [E,N]) may be NULL. */
int ssd_step_random(ssd_env *env, int32_t num_actions, int32_t *actions_out, void *obs, int32_t *rew,
                    uint8_t *done, uint32_t flags, void *stream);

/* A whole random-action rollout (rollout.py:58-70: reset, then `horizon` steps of uniformly drawn actions) enqueued
 * by ONE call: n_steps launches of the step kernel, preceded by a full reset whenever (step0 + k) % reset_every == 0
 * (reset_every = 0: never).  Step k writes slot (step0 + k) % ring of obs [ring,E,N,V,V,3], rew [ring,E,N] and
 * done [ring,E,N] (ring = 1: every step overwrites the same buffers); any of the three may be NULL.  Device pointers
 * only; the call enqueues on `stream` and returns.  Exactly the launches that n_steps calls of ssd_step_random (and
 * ssd_reset) would make -- the point is the host: one library call instead of one per step keeps a launch-bound
 * rollout fed.  Envs are independent, so the library may split the batch into up to 8 env ranges ("chains") that it
 * enqueues on queues of its own, forked from and joined back into `stream`: the launches of one chain then overlap
 * the dispatch / drain gaps of the others.  Results do not depend on the number of chains.
 * How the launches are issued is the library's business and does not change what the caller sees on `stream`: up to 3
 * chains are written as AQL packets into HSA queues the library owns (one pool per device), with the observation rendering
 * of a step carried out by extra workgroups of the next step's launch where that pays; otherwise, and
 * with SSD_AQL=0 in the environment, through hipLaunchKernel on HIP streams.  Every step's outputs are in their ring slots
 * when the work enqueued by the call has completed on `stream`.
 * (An observation ring of more than 232 MB -- beyond what the device's 256-MB memory-side cache can hold next to the state -- is
 * written with non-temporal write-back stores, flushed by the call's closing release and by an agent-scope release on one launch
 * per round of the ring: a ring that thrashes that cache costs 8.1 us per 4096-env step, past it 5.8 (write-through: 7.0);
 * a single slot rewritten every step lives in it: 5.4.) */
int ssd_rollout_random(ssd_env *env, int32_t num_actions, int32_t n_steps, int32_t reset_every, int32_t step0,
                       void *obs, int32_t *rew, uint8_t *done, int32_t ring, uint32_t flags, void *stream);

/* The same call with CALLER-SUPPLIED actions -- what the reference's real callers do: env.step(policy actions)
 * (visuallizer_rllib.py:121-153; RLlib's sampler behind train_baseline.py:71-81), map_env.py:152-212 per step -- for open-loop
 * replay of recorded action sequences, action chunking, or a policy that emits several steps at once.
 *   actions i32 [action_ring,E,N] (device): step k of the call reads slot (step0 + k) % action_ring; -1 = the agent does not act.
 *           action_ring = n_steps with step0 = 0 is the plain [K,E,N] form; action_ring = 1 feeds every step the same actions.
 *   order   u8  [action_ring,E,N] (device) or NULL: per step, the agent indices in action-dict order, 0xFF-terminated, as ssd_step
 *           takes them; NULL = index order (what the map-specific kernels and the coherent chains need: an explicit order takes
 *           the general kernels).
 * Everything else -- resets, output ring, chains, flags (SSD_ROLLOUT_FUSED: one launch, the actions fetched a step ahead;
 * SSD_OBS_F32) -- as ssd_rollout_random, and dispatched the same way: the step launches' kernel arguments are static per
 * (output slot, action slot) pair, so a call enqueues 64-byte packets only (argument sets are cached by the buffers passed: reuse
 * the same action / output buffers from call to call; more than 2048 / chains distinct (slot, slot) pairs go through HIP streams).
 * The actions must be in place on `stream` before the call (the call orders itself after the stream's earlier work) and must not
 * change until the call's work has completed on `stream`.  An action outside the game's Discrete(n) sets SSD_ST_BAD_ACTION. */
int ssd_rollout_actions(ssd_env *env, const int32_t *actions, const uint8_t *order, int32_t action_ring, int32_t n_steps,
                        int32_t reset_every, int32_t step0, void *obs, int32_t *rew, uint8_t *done, int32_t ring, uint32_t flags,
                        void *stream);

/* How the last rollout call (ssd_rollout_random / ssd_rollout_actions) of the handle was
 * dispatched (a bit mask; 0 before the first call):
 *   SSD_PATH_AQL       the launches were written as AQL packets into the library's own queues (else: hipLaunchKernel)
 *   SSD_PATH_COHERENT  ... with the kernel variant that needs no cache write-back between a chain's launches
 *   SSD_PATH_SPLIT     ... and every step's observations rendered by extra workgroups of the next step's launch
 *   SSD_PATH_FUSED     the call ran as the fused rollout kernel
 *   SSD_PATH_SYNC      ... with host-side waits instead of waiting kernels (a profiling tool is attached, or SSD_AQL_SYNC=1)
 *   SSD_PATH_FORKED    ... behind a fork from `stream`, which had work pending when the call came
 *   SSD_PATH_QUEUE_DROPPED  a dispatch queue of the device's pool failed its probe and was destroyed again (the rule below)
 *   bits 8..11         number of chains        bits 12..14  dispatch queues the device's pool has settled on
 *   bits 16..17        how the library found the HSA agent of the handle's HIP device: 1 PCI address, 2 UUID, 3 ordinal (cross-
 *                      checked by architecture and compute-unit count); 0: not at all -- no dispatch path of its own on this device
 * So that a caller (a test, a benchmark) can tell a silent fallback from the path it meant to measure. */
enum { SSD_PATH_AQL = 1, SSD_PATH_COHERENT = 2, SSD_PATH_SPLIT = 4, SSD_PATH_FUSED = 8, SSD_PATH_SYNC = 16, SSD_PATH_QUEUE_DROPPED = 32,
       SSD_PATH_FORKED = 64 };
int ssd_rollout_path(const ssd_env *env);

/* Number of chains the rollout calls use: 1..8, or 0 = automatic (1 below 2048 envs, 3 from 6144 to 24576, else 2 -- and never
 * more than the device's pool has dispatch queues).
 *
 * THE QUEUE RULE.  A process has about four hardware queues before the device time-slices them, and past that EVERY kernel launch
 * of the process -- the host application's too -- takes ~30 us.  The HIP runtime takes up to GPU_MAX_HW_QUEUES of them (default 4,
 * one per stream in use), RCCL one more stream.  So: the library's pool holds SSD_AQL_QUEUES queues (1..3) if that is set; else
 * 4 - GPU_MAX_HW_QUEUES if the process sets that variable for the HIP runtime (at least 1); else 2.  And whatever the rule says,
 * every queue is PROBED when it is created (first rollout call that needs it; the device is synchronised once).  The cliff is about
 * queues that are ACTIVE at the same time, so the probe is a rollout in miniature: 16 dependent one-wave dispatches on every queue
 * of the pool at once, joined through the null stream the way a rollout call is joined, timed against the pool's first queue
 * alone (MI355X: ~50 us with a hardware queue slot each, ~140 us when time-sliced); and a burst of HIP launches against its figure
 * from before the pool existed.  A queue whose arrival makes the concurrent burst more than 1.6 x (+ 10 us) slower, or the HIP burst
 * more than 2.5 x (+ 20 us) -- in two measurements, the second after the device has drained -- is destroyed again and the pool stays at the size that was fine for the life of the process
 * (SSD_PATH_QUEUE_DROPPED, bits 12..14 of ssd_rollout_path); an automatic chain count follows the smaller pool.  If already the
 * FIRST queue's burst takes more than 100 us, the process is past the cliff without the library (a host application with four
 * busy streams): the library then holds no queue at all and an automatic chain count is 1 -- the launches go to the caller's own
 * stream.  Streams the host application starts using LATER are not seen by the probe: an application that knows it will hold
 * many should set SSD_AQL_QUEUES=1 or SSD_AQL=0.
 *
 * ENVIRONMENT (read once per process; these are all the variables the product library reads):
 *   SSD_AQL=0            no dispatch queues of the library's own: every launch through hipLaunchKernel
 *   SSD_AQL_QUEUES=n     size of the pool (1..3), see above
 *   SSD_AQL_COHERENT=0   plain kernels behind agent-scope fences instead of the coherent variant
 *   SSD_AQL_SPLIT=0      every step renders its own observations
 *   SSD_AQL_SYNC=1 / 0   host-side waits instead of waiting kernels: forced / forbidden (default: on when a profiling tool
 *                        is attached -- ssd_profiler_attached() -- because such tools may run kernels one at a time)
 *   SSD_AQL_VERBOSE=1    the dispatch layer says on stderr what it set up (agent match, probe figures, fallbacks)
 *   SSD_ROLLOUT_CHAINS=n chains of the rollout calls when ssd_set_rollout_chains is 0
 *   SSD_ENVS_PER_BLOCK=n envs (waves) per workgroup, 1..16
 * Nothing here changes results.  The knobs that CAN (fence scopes, alternating geometries, forced forks) exist only in the
 * test-hook build, libssd_hip_testhooks.so (make testhooks), which the product never loads. */
int ssd_set_rollout_chains(ssd_env *env, int32_t chains);

/* 1 when a profiling / tracing tool is attached to the process (the ROCm tools' environment variables, or their libraries
 * loaded): the rollout calls then use host-side waits (SSD_PATH_SYNC).  Needs no device. */
int ssd_profiler_attached(void);

/* Observation of the current state without stepping (the per-agent part of map_env.py:189-199). */
int ssd_observe(ssd_env *env, void *obs, uint32_t flags, void *stream);

/* State access (host pointers, synchronous; any pointer may be NULL).  Mirrors what the reference's tests
 * poke directly: world_map (map_env.py:85), beam_pos (:86), Agent.pos / .orientation (agent.py:37-38).
 *   world, beam: i8 [E,H,W] ASCII (beam: 0 = none; needs keep_beams)   pos: i16 [E,N,2] (row, col)
 *   orient: u8 [E,N] 0 LEFT 1 RIGHT 2 UP 3 DOWN (key order of ORIENTATIONS, map_env.py:19-22)
 *   episode, t: u32 [E] PRNG coordinates (resets so far - 1, steps since reset) */
int ssd_get_state(ssd_env *env, int8_t *world, int8_t *beam, int16_t *pos, uint8_t *orient,
                  uint32_t *episode, uint32_t *t);
int ssd_set_state(ssd_env *env, const int8_t *world, const int8_t *beam, const int16_t *pos,
                  const uint8_t *orient, const uint32_t *episode, const uint32_t *t);

/* Cleanup only: waste_count u32 [E] = number of 'H' cells from which the last step / reset computed
 * current_apple_spawn_prob and current_waste_spawn_prob (compute_probabilities, cleanup.py:115,156-171;
 * it runs after the beams and before the spawn).  Host pointer, synchronous. */
int ssd_get_waste_count(ssd_env *env, uint32_t *waste_count);

/* MapEnv.map_to_colors() on the full grid of env e (map_env.py:316-339): rgb u8 [H,W,3], host pointer, synchronous. */
int ssd_render_full(ssd_env *env, int32_t e, uint8_t *rgb);

/* The same for envs [e_begin, e_begin + count) in one go: rgb u8 [count,H,W,3] -- the frames rollout.py:77 and
 * visuallizer_rllib.py:161 take one env at a time with map_to_colors().  Device pointer (enqueued on `stream`, returns
 * at once) or, with SSD_HOST_PTRS, host pointer (returns when the frames have arrived).  No other flag applies. */
int ssd_render_frames(ssd_env *env, int32_t e_begin, int32_t count, uint8_t *rgb, uint32_t flags, void *stream);

/* The extra members of the observation dict of an env built with return_agent_actions=True (map_env.py:201-205 step,
 * :242-246 reset, find_visible_agents :749-770; consumers run_scripts/train_moa.py:70, models/moa_model.py:216-249), for the
 * whole batch, as device tensors (or host arrays with SSD_HOST_PTRS):
 *   other_actions i64 [E,N,N-1]  row (e,i): this step's actions of the agents other than i, in the order of their ids sorted AS
 *           STRINGS, i.e. `sorted(actions.keys())` ('agent-10' < 'agent-2').  An agent absent from the action dict (action -1)
 *           contributes -1 in its place -- the reference's array is shorter then, the dict API mirror drops these entries.
 *           actions == NULL is the reset form: all zeros.  Rows whose done_mask byte (u8 [E,N], may be NULL: ssd_step's done output)
 *           is non-zero are zeros too -- the env was reset by the step launch (SSD_AUTO_RESET) or is about to be, and its
 *           observation row is a reset's.
 *   visible       i64 [E,N,N-1]  all ones: the reference tests the agent's OWN position against its window (:767).
 * Either output may be NULL.  Needs no engine state: a pure function of `actions`. */
int ssd_agent_action_obs(ssd_env *env, const int32_t *actions, const uint8_t *done_mask, int64_t *other_actions,
                         int64_t *visible, uint32_t flags, void *stream);

/* Episode length.  The reference's agents never report done; episodes end through RLlib's `horizon`
 * (run_scripts/train_baseline.py:131, train_moa.py:122).  horizon > 0: a step whose t reaches it writes
 * done = 1 for that env's agents (the caller then resets those envs, e.g. ssd_reset with done as the mask);
 * 0 (default): done stays 0. */
int ssd_set_horizon(ssd_env *env, int32_t horizon);

/* Queries. */
int ssd_potential_waste_area(const ssd_env *env);             /* cleanup.py:36-38 */
int ssd_device_status(ssd_env *env, uint32_t *status, int clear); /* synchronises the handle's device work */
int ssd_synchronize(ssd_env *env);
const char *ssd_last_error(const ssd_env *env);               /* NULL env: error of the last failed ssd_create */
int ssd_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
