// csrc/ssd_internal.hpp -- kernel parameter block shared by ssd_kernels.hip and ssd_capi.hip.
#pragma once
#include <stdint.h>
#include <stdlib.h>

namespace ssd {

// Environment knobs.  The product library reads only the documented, safe ones (the table in include/ssd.h): SSD_KNOB.
// Everything else -- fence scopes, forced forks, alternating geometries, tuning overrides -- is a test hook: SSD_HOOK reads the
// variable only in the test-hook build (make testhooks: -DSSD_TESTHOOKS, libssd_hip_testhooks.so, loaded by the tests that need
// one) and is the constant default in the product.
inline int knob_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
#define SSD_KNOB(name, dflt) ::ssd::knob_int(name, dflt)
#ifdef SSD_TESTHOOKS
#define SSD_HOOK(name, dflt) ::ssd::knob_int(name, dflt)
#else
#define SSD_HOOK(name, dflt) (dflt)
#endif

constexpr int kWave = 64;          // CDNA wavefront width
constexpr int kMaxEnvsPerBlock = 16; // one wavefront per env; a workgroup holds 1..16 envs (chosen per launch, see launch())
constexpr int kMaxAgents = 64;     // lanes = agents in the move / beam phases
constexpr int kMaxCells = 4096;    // H*W, bounded by the u64 per-lane spawn bitmask and by LDS (grid indices stay below 2^16)
constexpr int kMaxBeamLen = 21;    // 3 rays * beam_len lanes must fit one wavefront

constexpr int kListRegsHarvest = 3, kListRegsCleanup = 8;   // per-lane registers holding the first 192 / 512 entries of a static
                                   // cell list (only as many as the list needs are touched: the loops skip on 64 * j >= n)

enum Mode : int32_t {
    kModeStep = 0, kModeReset = 1, kModeObserve = 2,
    kModeRollout = 3,      // n_steps random-action steps in ONE launch
    kModeStepAuto = 4      // a step that also resets, in the same launch, the envs that reach the horizon (SSD_AUTO_RESET)
};

// PRNG streams (sequential_social_dilemma_games_amd/prng.py)
enum Stream : uint32_t {
    kSpawnPoint = 1, kSpawnRot = 2, kMove = 3, kApple = 4, kWasteCoin = 5, kWasteOrder = 6, kAction = 7
};

struct Params {
    // dimensions
    int32_t E, N, H, W;            // E: one past the last env this launch covers (the handle's env count unless a sub-range is launched)
    int32_t e_begin;               // first env this launch covers (0 unless ssd_rollout_random splits the batch into chains)
    int32_t WP, S;                 // grid layout: row stride WP = W + view_len (row padding = '0'), S = H*WP rounded up to 16
                                   // (per-env stride of the grids in HBM); cell index = row * WP + col everywhere
    int32_t A0, A1;                // LDS aprons of the world layer (bytes, multiples of 16): view_len rows of '0' above / below
    int32_t view_len, V, beam_len;
    int32_t mode, rotate, keep_beams, num_actions_random;
    int32_t obs_wt;                // observation stores: 0 ordinary, 1 write-through (sc1), 2 write-through + non-temporal; set per launch by select()
    int32_t obs_nt;                // the call's observation ring does not fit the memory-side cache: ask for 2 where 1 would be chosen
    int32_t obs_f32;               // obs is float32 [E,N,V,V,3] (SSD_OBS_F32) instead of uint8
    int32_t n_steps, reset_every, step0, ring, E_total;   // kModeRollout: steps in this launch, reset period (0 = never), index of the
                                   // first step, slots in the output ring, env count of the handle (slot stride)
    int32_t horizon;               // > 0: done = (t >= horizon), RLlib's `horizon` (train_baseline.py:131); 0: never done
    uint32_t v_magic16;            // ceil(2^16 / V): pp / V == (pp * v_magic16) >> 16 for pp < V*V, V <= 31
    uint32_t seed_lo, seed_hi, env_base;
    int32_t n_spawn, n_thr, n_apple, n_waste;
    int32_t n_waste_reset;         // Cleanup: number of 'H' cells in reset_world
    // engine state in HBM
    uint8_t *world;                // [E][S]  ASCII cells
    uint8_t *beam;                 // [E][S]  beam overlay (keep_beams only), 0 = none
    uint32_t *agents;              // [E][N]  cell | orient << 16
    uint4 *hdr;                    // [E]     {key, t, episode, #'H' used by the last spawn pass (Cleanup) | #'H' in the grid << 16 |
                                   //          "two agents may share a cell" << 31}
    uint32_t *status;              // [1]     SSD_ST_* bits
    // static tables in HBM (L2-resident)
    const uint8_t *reset_world;    // [S]     world right after reset_map()
    // cell list entries: grid index (row * WP + col) | dense index (row * W + col, the PRNG's cell key) << 16
    const uint32_t *spawn_cells;   // [n_spawn] 'P' cells, row-major (map_env.py:96-99)
    const uint32_t *apple_cells;   // [n_apple] 'A' (harvest.py:22-26) / 'B' (cleanup.py:53-54) cells, row-major
    const uint32_t *waste_cells;   // [n_waste] 'H' or 'R' cells (cleanup.py:59-60), row-major
    const uint32_t *lut;           // [128]   r | g << 8 | b << 16
    const float *f32lut;           // [128][4] glyph -> float32((channel - 128.0) / 255.0) of r, g, b (exact, host-built), 0
    const uint32_t *view_tab;      // [64][8] (view_len 7 only, else null): lane l's window offsets of its four view cells pp = min(4 l, 221) + q,
                                   // (i, j) = (pp / 15, pp % 15): L0[q] = i * WP + j, then L1[q] = j * WP + (14 - i) (render_views_std)
    const uint64_t *thr_ca;        // [n_thr] Cleanup apple thresholds by #'H'
    const uint64_t *thr_cw;        // [n_thr] Cleanup waste thresholds by #'H'
    uint32_t thr_h32[4];           // Harvest apple thresholds by min(#neighbour apples, 3): rand < p  <=>  u32 draw < thr
    uint32_t thr_h_always;         // bit n set: threshold n is 2^32 (p >= 1), the compare always succeeds
    // per-call I/O (device pointers; any may be null)
    const int32_t *actions;        // [E][N]; a fused rollout with caller-supplied actions: [action_ring][E_total][N], see below
    const uint8_t *order;          // [E][N]
    const uint8_t *mask;           // [E]     reset only
    int32_t *actions_out;          // [E][N]
    uint8_t *obs;                  // [E][N][V][V][3] u8, or float32 with obs_f32
    int32_t *rew;                  // [E][N]
    uint8_t *done;                 // [E][N]
    uint32_t dbg_skip;             // diagnostic builds (-DSSD_STAMPS) only: bit mask of phases to skip (tools/variant_times.py)
    unsigned long long *stamps;    // [E][16] s_memtime stamps; diagnostic builds (-DSSD_STAMPS) only, else null
    // kModeRollout with caller-supplied actions (ssd_rollout_actions + SSD_ROLLOUT_FUSED): step k of the call takes its actions
    // from slot (step0 + k) % action_ring of `actions`; 0: the launch draws its actions (num_actions_random)
    int32_t action_ring;
    uint32_t coherent;             // the step launch moves state and outputs with agent-scope (sc1) accesses only (kernel: COH); set by
                                   // the library's own dispatch path for the map-specific uint8 kernels (ssd_aql.hip)
    // split rollouts (bits): 1 = the envs' waves do not render: they write grid and agents to `world_out` / `agents_out` (the other
    // buffer of the pair), this step's beam marks to `beam_list` (and, for the rare step whose marks are not in registers, the
    // overlay to `snap`); 2 = the launch carries renderer workgroups (behind its first blocks_a) that render the PREVIOUS step's
    // observations from the launch's input state + `beam_list_in` (or `snap_in`) into `obs_b`; 4 = renderer workgroups only
    int32_t snap_mode, blocks_a;
    uint8_t *world_out;            // [E][S]  null: state is written in place
    uint32_t *agents_out;          // [E][N]
    uint32_t *beam_list;           // [2][E][64] cell | mark << 16 per lane of the step's beam trace; 0 none ([1]: the second pass of a Cleanup step with
                                   // more shooters than one pass has slots, agent 0's bit 22)
    const uint32_t *beam_list_in;  // the previous step's
    uint8_t *snap;                 // [E][S]  overlay of the step (world <- agents <- beams), only after steps that could not list their marks
    const uint8_t *snap_in;        // the previous step's
    uint8_t *obs_b;                // [E][N][V][V][3]  where the renderer workgroups write
};

// The kernarg segment of ssd_env_kernel: its nine leading arguments (repeated Params fields, 56 bytes: preloaded into SGPRs
// by the command processor) followed by the parameter block.  Natural C layout == the code object's argument offsets.
struct KernArgs {
    uint4 *hdr; uint32_t *agents; uint8_t *world;
    int32_t E, e_begin, epb, n_apple;
    const uint32_t *apple_cells; const uint32_t *lut;
    Params p;
};
static_assert(sizeof(KernArgs) == 56 + sizeof(Params), "kernarg layout");

// One launch, fully resolved: instantiation (host stub of the __global__), geometry, dynamic LDS bytes, kernel arguments.
struct Launch {
    const void *fn = nullptr;
    uint32_t grid_x = 0, block_x = 0, lds = 0;
    KernArgs args;
};

// Agent indices in the order of their ids sorted as strings ('agent-0', 'agent-1', 'agent-10', 'agent-11', 'agent-2', ...: what
// `sorted(actions.keys())` gives, map_env.py:202), and each index's rank in that order.
struct AgentOrder { uint8_t sorted[kMaxAgents]; uint8_t rank[kMaxAgents]; };
void launch_agent_action_obs(const int32_t *actions, const uint8_t *done_mask, long long *other_actions, long long *visible,
                             const AgentOrder &order, int E, int N, void *stream);

size_t lds_bytes(const Params &p, int envs_per_block, bool f32);
int envs_per_block(const Params &p, bool f32);
int fast_profile(const Params &p, int game);    // which map-specific step kernel a launch gets (0: the general ones)
bool select(const Params &p, int game, Launch *out);   // resolve a launch without issuing it
void launch(const Launch &L, void *stream);            // hipLaunchKernel of a resolved launch
void launch(const Params &p, int game, void *stream);
const void *flag_kernel_fn();                          // ssd_flag_kernel's host stub (AQL join)
const void *wait_kernel_fn();                          // ssd_wait_counter_kernel's host stub (AQL fork: a chain's first packet)
void launch_flag_kernel(unsigned long long *counter, void *stream);   // AQL fork: bump a counter from a HIP stream
// the kernel arguments of ssd_wait_counter_kernel as its kernarg segment lays them out
struct WaitArgs { const unsigned long long *counter; unsigned long long target; const uint32_t *abort; unsigned long long timeout_ticks; uint32_t *status; uint32_t *timed_out; };
// the stream-side wait of the AQL join: gives up after `timeout_ticks` of the 100 MHz clock and then sets kStWaitTimeout in *status
// and 1 in *timed_out (host memory the device can write: the handle's sticky flag, looked at by the next API call)
void launch_wait_counter_kernel(const unsigned long long *counter, unsigned long long target, const uint32_t *abort,
                                unsigned long long timeout_ticks, uint32_t *status, uint32_t *timed_out, void *stream);
#ifdef SSD_STAMPS
void launch_clock_kernel(unsigned long long *out, int iters, void *stream);
#endif
void launch_signal_kernel(long long *signal_value, void *stream);   // AQL fork: zero an HSA signal from a HIP stream
void launch_render_full(const Params &p, int e0, int count, uint8_t *rgb_dev, void *stream);

}  // namespace ssd
