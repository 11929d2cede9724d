// csrc/ssd_kernels.hip -- the MapEnv.step() hot path as one fused gfx950 kernel.
//
// Mapping (CDNA4: 64-wide wavefronts, 160 KiB LDS per CU, 256 CUs):
//   * one wavefront owns one env for the whole launch; a workgroup holds 1..16 envs (envs_per_block());
//   * the env's grid (rows padded by view_len: 16 x (38 + 7) = 720 B for Harvest) is pulled from HBM once with
//     16 B/lane loads into LDS, every phase works on the LDS copy, and it is written back once;
//   * move / rotate / conflict resolution: lanes = agents, positions compared with wavefront ballots combined on
//     the scalar unit ("who stands on cell x" = ballot, last index = highest set bit); the order-dependent parts
//     of the reference algorithm run as wave-uniform code, chains of waiting agents by pointer jumping
//     (map_env.py:357-543);
//   * beams: lanes = (shooter, ray, step), stop positions from ballots (map_env.py:566-649);
//   * respawn: lanes = entries of the map's static apple / waste cell lists (held in registers),
//     3x3 stencil on the LDS grid, counter-based PRNG keyed on the cell (harvest.py:75-104, cleanup.py:132-171);
//   * observation: the wave renders its env's N agents from the LDS overlay, lane = 4 consecutive cells of the
//     15 x 15 view = 12 contiguous bytes per store, so a wavefront store covers up to 768 contiguous bytes of
//     the uint8 obs tensor; the padded layout makes a view cell one multiply-add away from its LDS address
//     (map_env.py:189-199);
//   * waves never touch each other's LDS: the kernel has no workgroup barrier.
// One source, four modes (template parameter): step, reset, observe -- one pass over the env per launch -- and
// rollout, which loops [reset pass,] step pass, ... with the env resident in LDS / registers (SSD_ROLLOUT_FUSED).
// At 4096 envs (4 waves per SIMD) a per-step launch lasts as long as its SLOWEST wave plus launch and store-drain
// time, so the code minimises a single wave's dynamic instruction count and wait chain --
// every global load is issued in the prologue, cross-lane reductions use DPP / scalar code instead of LDS-crossbar
// shuffles, per-map work lists replace grid scans, the common conflict-free move takes a short path -- and lets
// the rare long waves issue first (s_setprio).
// No MFMA: the path is integer / indexing work; its roofline is HBM traffic (4 724 B per env-step).
//
// Reference citations are file:line of the reference repository (social_dilemmas/envs/...).
#include <hip/hip_runtime.h>

#include "ssd_internal.hpp"
#include <type_traits>

#ifndef SSD_PIN_EARLY          // (experiment switch: 0 = the prologue's kernel arguments requested where they always were)
#define SSD_PIN_EARLY 1
#endif
#ifndef SSD_WB_UNROLL          // (experiment switch: 0 = the write-back of a known map's grid as a loop)
#define SSD_WB_UNROLL 1
#endif
#ifdef SSD_EXP_OBS768
#define SSD_OBS_STRIDE SSD_EXP_OBS768
#else
#define SSD_OBS_STRIDE 675
#endif

namespace ssd {

// SSD_ST_* (include/ssd.h)
constexpr uint32_t kStBadAction = 1u << 0;
constexpr uint32_t kStNoSpawn = 1u << 1;
constexpr uint32_t kStMoveLookup = 1u << 2;
constexpr uint32_t kStWaitTimeout = 1u << 3;    // the stream-side wait of a rollout's join gave up (ssd_wait_counter_kernel)

// ---------------------------------------------------------------------------------------------
// shared PRNG (prng.py): triple32 chain
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 17; x *= 0xED5AD4BBu;
    x ^= x >> 11; x *= 0xAC4C1B51u;
    x ^= x >> 15; x *= 0x31848BABu;
    x ^= x >> 14;
    return x;
}
__device__ __forceinline__ uint32_t env_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t env, uint32_t episode) {
    uint32_t h = 0x243F6A88u;
    h = mix32(h ^ seed_lo);
    h = mix32(h ^ seed_hi);
    h = mix32(h ^ env);
    h = mix32(h ^ episode);
    return h;
}
__device__ __forceinline__ uint32_t phase_key(uint32_t key, uint32_t t, uint32_t stream) {
    return mix32(mix32(key ^ t) ^ stream);
}
// (the two halves of phase_key: the first is the same for every stream of a step -- computed once per pass, on the scalar unit)
__device__ __forceinline__ uint32_t step_key(uint32_t key, uint32_t t) { return mix32(key ^ t); }
#ifdef SSD_EXP_NOSTREAMKEY   // (experiment switch, wrong results: the upper bound of what cheaper stream keys could give)
__device__ __forceinline__ uint32_t stream_key(uint32_t skey, uint32_t stream) { return skey ^ (stream * 0x9E3779B9u); }
#else
__device__ __forceinline__ uint32_t stream_key(uint32_t skey, uint32_t stream) { return mix32(skey ^ stream); }
#endif
__device__ __forceinline__ uint32_t draw(uint32_t pkey, uint32_t index) { return mix32(pkey ^ index); }
__device__ __forceinline__ uint32_t randint(uint32_t u, uint32_t n) { return __umulhi(u, n); }

// ---------------------------------------------------------------------------------------------
// wavefront helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint64_t bit(uint32_t i) { return 1ull << i; }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// Lanes of one wavefront communicate through LDS without a workgroup barrier (LDS operations of a
// wave complete in order); this only has to stop the compiler from moving LDS accesses across it.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Full-wave reduction: inclusive scan inside each row of 16 lanes with DPP row_shr (VALU latency,
// no LDS crossbar), then the four row results (lanes 15/31/47/63) are combined on the scalar unit.
#define SSD_DPP(old, v, ctrl, bc) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, 0xF, 0xF, bc))
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = umin(v, SSD_DPP(0xFFFFFFFFu, v, 0x111, false));   // out-of-row lanes keep `old` = identity
    v = umin(v, SSD_DPP(0xFFFFFFFFu, v, 0x112, false));
    v = umin(v, SSD_DPP(0xFFFFFFFFu, v, 0x114, false));
    v = umin(v, SSD_DPP(0xFFFFFFFFu, v, 0x118, false));
    return umin(umin(rl(v, 15), rl(v, 31)), umin(rl(v, 47), rl(v, 63)));
}
// Smallest (hi, lo) pair over the lanes with `has`; exact also when hi == 0xFFFFFFFF.  Returns "any lane has".
__device__ __forceinline__ bool wave_argmin_pair(bool has, uint32_t hi, uint32_t lo, uint32_t &out_hi, uint32_t &out_lo) {
    if (!ballot(has)) return false;
    const uint32_t mh = wave_min_u32(has ? hi : 0xFFFFFFFFu);
    const uint32_t ml = wave_min_u32((has && hi == mh) ? lo : 0xFFFFFFFFu);
    out_hi = mh; out_lo = ml;
    return true;
}

// map_env.py:290 + the '<U1' array dtype (:85): str(int(agent_id[-1]) + 1) truncated to one char.
__device__ __forceinline__ uint8_t agent_glyph(uint32_t i) {
    uint32_t d = i % 10u;
    return d == 9u ? (uint8_t)'1' : (uint8_t)('1' + d);
}

// map_env.py:719-737 update_rotation; orientation codes 0 LEFT 1 RIGHT 2 UP 3 DOWN.
__device__ __forceinline__ uint32_t turn(int act, uint32_t o) {
    // clockwise (5): LEFT->UP->RIGHT->DOWN->LEFT ; counter-clockwise (6): LEFT->DOWN->RIGHT->UP->LEFT
    const uint32_t cw = (2u << 0) | (3u << 2) | (1u << 4) | (0u << 6);    // [LEFT,RIGHT,UP,DOWN] -> UP,DOWN,RIGHT,LEFT
    const uint32_t ccw = (3u << 0) | (2u << 2) | (0u << 4) | (1u << 6);   // -> DOWN,UP,LEFT,RIGHT
    return ((act == 5 ? cw : ccw) >> (2u * o)) & 3u;
}

// orientation / MOVE_* vector (map_env.py:11-15,19-22): 0 (-1,0) 1 (1,0) 2 (0,-1) 3 (0,1) 4 (0,0)
__device__ __forceinline__ void unit_vec(int code, int &dr, int &dc) {
    dr = code == 0 ? -1 : (code == 1 ? 1 : 0);
    dc = code == 2 ? -1 : (code == 3 ? 1 : 0);
}

// Vector memory accesses with a cache policy go through the buffer instructions' builtins -- buffer_load / buffer_store with
// sc1 (agent scope: past the CU's L1 and not left dirty in the XCD's L2), nt (non-temporal: past the memory-side cache) or both in
// the instruction's cache-policy field -- NOT through inline asm: the compiler tracks these loads (its own s_waitcnt vmcnt before
// the first use, wherever that is, also under divergent control flow) and the hazards around the stores.  Round 3 had them as
// `asm volatile("global_load_dwordx4 ... sc1")`, which the compiler believes complete when the statement ends: one wrong grid and
// one memory fault came of it.  A buffer resource names a wave-uniform byte range; an access beyond it is dropped (stores) or
// returns zero (loads), which is also the bounds check of the lanes past a grid's end.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kAuxSc1 = 16, kAuxNt = 2, kAuxSc1Nt = 18;           // cache-policy bits of the gfx94x / gfx950 buffer builtins
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    // (wave-uniform by construction -- an env's or an agent's block -- and said so: the descriptor lives in four SGPRs)
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}
// Observation stores are write-through (sc1 = agent scope: the data leaves for HBM as it is produced).  With ordinary
// stores the 13.8 MB of a 4096-env step sit dirty in L2 until the end-of-kernel write-back, which the next launch has to
// wait for: 8.0 -> 6.9 us per step.  `r` names the env's observation block, `soff` (wave-uniform) the agent's offset in it,
// `off` the lane's byte offset; any alignment.
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// `wt` (wave-uniform, Params::obs_wt) picks the policy per launch: once a launch is many rounds of waves (above 16 384
// envs per launch, two launches in flight) the kernel is bandwidth-bound, L2 merging of the 12-byte pieces matters more than the final flush, and ordinary
// stores win (65 536 envs: 53.6 vs 69.4 us).
// wt = 2: `sc1 nt`, non-temporal on top: the stores bypass the 256-MB memory-side cache.  With ONE output slot (13.8 MB at 4096
// envs, rewritten every step) that cache absorbs the observation stores altogether -- 5.43 us per step, and 6.91 with `nt` -- but an
// output ring that does not fit in it (32 slots: 442 MB) thrashes it: 8.07 us per step with plain write-through stores.  So the host
// asks for `nt` when the ring's observation bytes exceed what the cache can hold (ssd_capi.hip: obs_nt).
// What then bounds WRITE-THROUGH stores is that the two ends of every agent block (675 bytes: any alignment) are 64-byte sectors
// written partly by one store instruction and partly by another: with the blocks padded to 704 / 768 bytes (NOT the output layout)
// a ring of 32 slots takes 5.95 / 5.74 us per step against 6.96; dword alignment alone (676) 6.81.
// wt = 3: `nt` alone -- write-BACK and non-temporal: the partial sectors meet in L2 and leave it as whole lines: 5.79 us per step in
// the real layout; what uint8 observations use for such a ring (select() below).  Before that was found, merging inside the wave
// was built and measured (commit ae4b8b6: the env's N x 675 bytes written as ONE stream of 16-byte pieces aligned in memory, eight
// whole lines per instruction, the block's two partial ends as 8 / 4 / 2 / 1-byte pieces; bit-exact): the memory system is relieved
// as hoped -- env waves 15 % shorter, stores land in 950 cycles instead of 2 700 -- but a piece's 16 bytes are 5 1/3 view cells of
// up to two agents, so cell coordinates become per-lane arithmetic where the 12-byte form has per-lane CONSTANTS: the renderer's
// render phase grows from 3 100 to 5 400 cycles and the renderer wave becomes the launch's critical path: 7.6 against 7.03 us.
__device__ __forceinline__ void store12_wt(rsrc_t r, uint32_t soff, uint32_t off, u32x3_t d, int wt) {
    // (the usual policy first: in the fused kernel's step loop every test in front of it showed, 3.75 -> 3.91 us per step)
    if (wt == 1) __builtin_amdgcn_raw_buffer_store_b96(d, r, (int)off, (int)soff, kAuxSc1);
    else if (wt == 3) __builtin_amdgcn_raw_buffer_store_b96(d, r, (int)off, (int)soff, kAuxNt);
    else if (wt == 2) __builtin_amdgcn_raw_buffer_store_b96(d, r, (int)off, (int)soff, kAuxSc1Nt);
    else __builtin_amdgcn_raw_buffer_store_b96(d, r, (int)off, (int)soff, 0);
}
__device__ __forceinline__ void store16_wt(rsrc_t r, uint32_t soff, uint32_t off, f32x4_t f, int wt) {
    const u32x4_t d = __builtin_bit_cast(u32x4_t, f);
    if (wt >= 2) __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)off, (int)soff, kAuxSc1Nt);
    else if (wt) __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)off, (int)soff, kAuxSc1);
    else __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)off, (int)soff, 0);
}
// 16 bytes with one agent-scope load / store
__device__ __forceinline__ u32x4_t load16_sc1(rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, kAuxSc1); }
__device__ __forceinline__ void store16_sc1(rsrc_t r, uint32_t off, u32x4_t d) { __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)off, 0, kAuxSc1); }

// bytes of x that are non-zero -> 0xFF, others 0x00
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) {
    uint32_t m = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    return (m >> 7) * 0xFFu;
}
// The observation phase of the reference configuration (view_len 7: V = 15) for NA agents, five per pass: lane = 4 consecutive
// cells of the 15 x 15 view = one 12-byte store (the lane holding the leftover 225th cell starts 4 cells before the end and
// re-renders 3 cells of its neighbour, so that every store is a full 12 bytes).  All grid reads of a pass go out together, then
// all colour-table reads, then the stores: two LDS round trips per pass.  `a_k` / `a_s0` hold, in lane = agent, the agent's
// quarter turns and the LDS offset of its window's first (k < 2) or last (k >= 2) cell; `view_lds` is the LDS byte address of grid
// cell 0 of the layer the views show.  (agent.py:76-78 -> utility_funcs.py:59-114, map_env.py:316-339, :669-689.)
// `tab` (the renderer workgroups of split rollouts): the lane's eight window offsets, L0[q] = i*WP + j and L1[q] = j*WP + (14 - i) of
// its four view cells (i, j) -- per-lane CONSTANTS of the map's row stride -- fetched from a table the host built (Params::view_tab,
// two 16-byte loads beside the grid's) instead of being computed by every wave of every launch: ~24 vector instructions of a
// renderer wave's 134.
template <int NA, bool TAB = false>
__device__ __forceinline__ void render_views_std(const int lane, const int WP, const uint32_t a_k, const uint32_t a_s0, const uint32_t view_lds,
                                                 const uint32_t *s_lut, uint8_t *out_env, const int wt, const uint4 tab0 = uint4{0, 0, 0, 0},
                                                 const uint4 tab1 = uint4{0, 0, 0, 0}) {
    typedef __attribute__((address_space(3))) const uint8_t lds_u8;
    constexpr int V = 15, VV = 225, kB = 5;
    const int pp_raw = 4 * lane;
    const bool lane_on = pp_raw < VV;
    const int pp0 = pp_raw > VV - 4 ? VV - 4 : pp_raw;                  // lanes past the end repeat the last one
    int L0[4], L1[4];
    if constexpr (TAB) {
        L0[0] = (int)tab0.x; L0[1] = (int)tab0.y; L0[2] = (int)tab0.z; L0[3] = (int)tab0.w;
        L1[0] = (int)tab1.x; L1[1] = (int)tab1.y; L1[2] = (int)tab1.z; L1[3] = (int)tab1.w;
    } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pp = pp0 + q;
        const int i = pp / 15, j = pp - i * V;
        // 24-bit multiply-adds (full rate; a plain `*` becomes a quarter-rate 32-bit multiply here)
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(L0[q]) : "v"(i), "s"(WP), "v"(j));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(L1[q]) : "v"(j), "s"(WP), "v"(V - 1 - i));
    }
    }
    uint32_t off3 = (uint32_t)pp0 * 3u;                                 // byte offset of the lane's cells in an agent's block
    asm volatile("" : "+v"(off3));                                      // keep it in a register (else re-derived per agent)
    const rsrc_t out_r = make_rsrc(out_env, (uint32_t)NA * SSD_OBS_STRIDE);
    for (int ag0 = 0; ag0 < NA; ag0 += kB) {
        uint32_t addr[kB][4], px[kB][4];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            const uint32_t k = rl(a_k, ag0 + u);
            const uint32_t s0 = rl(a_s0, ag0 + u) + view_lds;
            const int sgn = k >= 2 ? -1 : 1;
            // wave-uniform branch on the rotation's parity instead of a per-cell select (the asm is volatile so that the two
            // arms are not merged back into selects)
            if (k & 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(addr[u][q]) : "v"(L1[q]), "v"(sgn), "s"(s0));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(addr[u][q]) : "v"(L0[q]), "v"(sgn), "s"(s0));
            }
        }
        uint32_t gl[kB][4];
#pragma unroll
        for (int u = 0; u < kB; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) gl[u][q] = *(lds_u8 *)(uintptr_t)addr[u][q];
#pragma unroll
        for (int u = 0; u < kB; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) px[u][q] = s_lut[gl[u][q]];
        if (lane_on) {
            u32x3_t d[kB];
#pragma unroll
            for (int u = 0; u < kB; ++u) {
                // 4 x (r,g,b) -> 12 bytes with three byte permutes (v_perm_b32: selector bytes 0-3 pick from the second
                // operand, 4-7 from the first)
                d[u].x = __builtin_amdgcn_perm(px[u][1], px[u][0], 0x04020100u);   // r0 g0 b0 r1
                d[u].y = __builtin_amdgcn_perm(px[u][2], px[u][1], 0x05040201u);   // g1 b1 r2 g2
                d[u].z = __builtin_amdgcn_perm(px[u][3], px[u][2], 0x06050402u);   // b2 r3 g3 b3
            }
            // (measured with agent blocks 768 bytes apart instead -- every store then covers whole 128-byte lines: 5.30 against
            // 5.33 us per step; the partly written lines at the blocks' ends are not what bounds write-through stores in the
            // memory-side cache.)  The store policy is wave-uniform: ONE branch per pass of five agents, the five stores of a
            // policy behind each other (a test per store showed in the fused kernel's step loop: 3.75 -> 3.91 us per step).
            auto stores = [&](auto policy) {
#pragma unroll
                for (int u = 0; u < kB; ++u) store12_wt(out_r, (uint32_t)(ag0 + u) * SSD_OBS_STRIDE, off3, d[u], decltype(policy)::value);
            };
            if (wt == 1) stores(std::integral_constant<int, 1>{});
            else if (wt == 3) stores(std::integral_constant<int, 3>{});
            else if (wt == 2) stores(std::integral_constant<int, 2>{});
            else stores(std::integral_constant<int, 0>{});
        }
    }
}

#ifndef SSD_ROLL_VKEYS      // (1: the rollout kernel hashes its phase keys on the vector unit, all streams at once)
#define SSD_ROLL_VKEYS 1
#endif
// Per-phase cycle stamps for tools/phase_profile.py: compiled only into the diagnostic library
// (make stamps); the product build contains no stamp code.
#ifdef SSD_STAMPS
#define SSD_STAMP(i)                                                                               \
    do {                                                                                           \
        if (p.stamps && lane == 0 && e < p.E) p.stamps[(size_t)e * 16 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#define SSD_STAMP_RT(i) /* 100 MHz constant clock, the same on every XCD */                       \
    do {                                                                                           \
        if (p.stamps && lane == 0 && e < p.E) p.stamps[(size_t)e * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define SSD_NOTE(i, v)                                                                             \
    do {                                                                                           \
        if (p.stamps && lane == 0 && e < p.E) p.stamps[(size_t)e * 16 + (i)] = (unsigned long long)(v); \
    } while (0)
#define SSD_SKIP(bit) ((p.dbg_skip >> (bit)) & 1u)
#else
#define SSD_NOTE(i, v)
#define SSD_STAMP(i)
#define SSD_STAMP_RT(i)
#define SSD_SKIP(bit) false
#endif

// ---------------------------------------------------------------------------------------------
// LDS layout of one workgroup: per wave (= per env)  lut[128] u32 | scratch[64] u32 |
// glyph -> 3 floats table [128][4] (float32-obs kernels only) | apron[A0] | world[S] | apron[A1] | beam[S] | occ[S]
// Waves never read each other's LDS, so the kernel has no workgroup barrier.
//
// Grid layout (HBM and LDS alike): row stride WP = W + view_len, the view_len bytes after each row hold the
// void glyph '0'; in LDS the world layer additionally has view_len rows of '0' above and below (the aprons).
// An agent's whole (2*view_len+1)^2 window -- the map cells AND the '0' padding that
// utility_funcs.py:94-114 adds around the map -- is then plain memory around the agent's cell: the
// observation phase needs no bounds test and no row / column arithmetic at all.
// ---------------------------------------------------------------------------------------------
// The rollout kernel (kModeRollout) keeps the env in LDS across steps, so its overlay goes to a fourth layer (`view`,
// which then carries the aprons) instead of overwriting the world layer in place.
__host__ size_t lds_bytes(const Params &p, int envs_per_block, bool f32) {
    const int layers = p.mode == kModeRollout ? 4 : 3;
    return (size_t)envs_per_block * (128 * 4 + 256 + (f32 ? 128 * 16 : 0) + (size_t)p.A0 + (size_t)p.A1 + layers * (size_t)p.S);
}

// Envs (= waves) per workgroup.  Waves are independent, so this only changes dispatch granularity; measured on
// MI355X (Harvest, us per launch):  E=4096: 14.45 / 14.42 / 14.34 / 14.24 for 2 / 4 / 8 / 16 per block;
// E=8192: 21.9 / 22.0 / 21.7 / 22.8;  E=65536: 111 / 113 / 117 / 136.  So: big blocks while the whole batch is one
// round with at most one block per CU (256 CUs), small blocks once CUs run several rounds, and never fewer than
// 256 blocks when the batch is small.  SSD_ENVS_PER_BLOCK overrides (tuning).
static int forced_epb_early() {
    static const int forced = SSD_KNOB("SSD_ENVS_PER_BLOCK", 0);
    return forced;
}
__host__ int envs_per_block(const Params &p, bool f32) {
    const int E = p.E - p.e_begin;
    static const int forced = SSD_KNOB("SSD_ENVS_PER_BLOCK", 0);
    auto pow2floor = [](int x) { int p = 1; while (p * 2 <= x) p *= 2; return p; };
    int fill = pow2floor(E / 256 > 0 ? E / 256 : 1);            // keep >= 256 blocks
    int rounds = pow2floor(65536 / E > 2 ? 65536 / E : 2);      // 16 at 4096 envs, 8 at 8192, ... 2 from 32768 on
    int epb = fill < rounds ? fill : rounds;
    if (epb > kMaxEnvsPerBlock) epb = kMaxEnvsPerBlock;
    if (forced >= 1 && forced <= kMaxEnvsPerBlock) epb = forced;
    while (epb > 1 && lds_bytes(p, epb, f32) > 64 * 1024) epb /= 2;
    return epb;
}

// NA > 0: the number of agents is the compile-time constant NA; STD: view_len 7 / beam_len 5, the reference's module
// constants (harvest.py:11,15; cleanup.py:11-12,22).  Fixing them lets the compiler unroll the agent loops, fold the
// window arithmetic and address agents' lanes by immediate: 14.3 -> 12.6 us per 4096-env step (N, V, L fixed).
// NA = 0 / STD = false is the fully general kernel.
// FAST (1, 2) additionally fixes the map to a known one (kFastMap: the shipped maps, the enlarged ones) and the call to its plain
// form (index action order, beams not kept): another 3.8 %.
// Grid layout constants of the maps the FAST kernels are compiled for (view_len 7): FAST = 1 the game's shipped map
// (Harvest 16x38, Cleanup 25x18), FAST = 2 the enlarged maps of BASELINE.json's configurations (Harvest 25x38, Cleanup
// 48x36; constants.py builds them by the rule of SURVEY.md 8d).  launch_game() checks every one of these numbers.
struct FastMap { int H, W, WP, S, A0, A1, n_apple, n_waste; };
constexpr FastMap kFastMap[2][3] = {
    {{0, 0, 0, 0, 0, 0, 0, 0}, {16, 38, 45, 720, 336, 320, 155, 0}, {25, 38, 45, 1136, 336, 320, 252, 0}},
    {{0, 0, 0, 0, 0, 0, 0, 0}, {25, 18, 25, 640, 192, 176, 103, 119}, {48, 36, 43, 2064, 320, 304, 412, 476}}};

// COH ("coherent"): the env's state, rewards, dones and observations move with agent-scope (sc1) accesses only: nothing of a
// launch stays dirty in an XCD's L2 and nothing is read through a CU's L1, so consecutive launches of a chain need no cache
// write-back / invalidate between them -- the library's own dispatch queues (ssd_aql.hip) then order them with the packet's
// barrier bit alone (release fence NONE: -0.85 us per step).  (Round 2 also had a variant whose waves waited env by env on pass
// counters, "pipelined launches": 4.44 against 4.50 us per step at 2048 envs, nothing at 4096 -- removed in round 3.)
template <int GAME, int MODE, bool F32, int NA, bool STD, int FAST, bool COH = false, bool ACTS = false>
// The leading arguments repeat the Params fields the first global loads need (14 dwords).  Built with
// -mllvm -amdgpu-kernarg-preload-count=14 the command processor delivers them in SGPRs when the wave starts, so the
// loads of the env's state go out without first waiting ~0.3 us for a scalar load of the kernel arguments.
__global__ __launch_bounds__(64 * kMaxEnvsPerBlock) void ssd_env_kernel(uint4 *const a_hdr, uint32_t *const a_agents, uint8_t *const a_world,
                                                                        const int a_E, const int a_e_begin, const int a_epb, const int a_n_apple,
                                                                        const uint32_t *const a_apple_cells, const uint32_t *const a_lut,
                                                                        const Params p) {
    uint8_t *const a_obs = p.obs;
    extern __shared__ __align__(16) uint8_t smem[];
#ifdef SSD_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();    // before the first kernel-argument fetch
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index and everything derived from it (env index, LDS region, global offsets) is wave-uniform:
    // say so, and the per-env address arithmetic runs on the scalar unit instead of as 64-bit VALU multiplies
    const int wv = (int)rfl((uint32_t)tid >> 6);
    // FAST: the layout constants of a known map (kFastMap) instead of kernel arguments
    constexpr FastMap fm = kFastMap[GAME][FAST];
    const int WP = FAST ? fm.WP : p.WP;
    const int S = FAST ? fm.S : p.S;
    const int A0 = FAST ? fm.A0 : p.A0, A1 = FAST ? fm.A1 : p.A1;
    const int N = NA > 0 ? NA : p.N;
    // List registers per lane (64 * kLR entries of a static cell list live in registers, the rest is read from memory when
    // needed).  Measured: 8 pays for Cleanup's long lists (48x36 map: 412 apple / 476 waste points, -4 % per step, -10 %
    // fused), 3 is better for Harvest (25x38 map, 262 apple points: 8 costs +4 % per step).
    // The enlarged Harvest map's own kernel takes 4: its 252 points then all sit in registers and the usual respawn takes its
    // compact form (one stencil + draw per lane instead of four): 6.49 -> 6.05 us per 4096-env step.
    constexpr int kLR = GAME == 1 ? kListRegsCleanup : (FAST == 2 ? 4 : kListRegsHarvest);
    // sizes of the map's cell lists (FAST: the shipped maps' -- launch_game() checks them): lets the compiler drop the
    // unused third list register of Cleanup's 103 apple / 119 waste points
    const int n_apple = FAST ? fm.n_apple : a_n_apple, n_waste = FAST ? fm.n_waste : p.n_waste;
    const bool has_order = !FAST && p.order != nullptr, keep_beams = !FAST && p.keep_beams != 0;
    constexpr bool roll = MODE == kModeRollout;                     // many steps per launch, env resident in LDS
    // A step whose launch also resets the envs that reach the horizon (SSD_AUTO_RESET): the step pass, then -- for those
    // envs -- a reset pass whose (unrotated) observations replace the step's rows.  Laid out like a plain step.
    constexpr bool auto_mode = MODE == kModeStepAuto;
    constexpr bool stepping = MODE == kModeStep || auto_mode;       // the launch takes actions
    uint32_t *s_lut = reinterpret_cast<uint32_t *>(smem + (size_t)wv * (512 + 256 + (F32 ? 2048 : 0) + (size_t)A0 + (size_t)A1 +
                                                                            (roll ? 4 : 3) * (size_t)S));
    uint32_t *s_tmp = s_lut + 128;                                  // 64 list entries of scratch (respawn compaction)
    float *s_f32 = reinterpret_cast<float *>(s_lut + 192);          // float32-observation kernels only
    uint8_t *s_world = reinterpret_cast<uint8_t *>(s_lut + 192 + (F32 ? 512 : 0)) + (roll ? 0 : A0);
    uint8_t *s_beam = s_world + S + (roll ? 0 : A1);
    uint8_t *s_occ = s_beam + S;
    uint8_t *s_view = roll ? s_occ + S + A0 : s_world;              // what the observations read: world <- agents <- beams

    constexpr bool kCoh = COH;                                       // state through memory with sc1 accesses
    const int blk = blockIdx.x;
    // ---- renderer role (split rollouts): workgroups behind the launch's first p.blocks_a render the observations of the
    //      PREVIOUS step of their envs, nothing else.  What they show is the state this launch steps FROM (a_world / a_agents: the
    //      env waves of the launch read the same lines and write elsewhere), overlaid with the agents' glyphs and the previous
    //      step's beam marks (p.beam_list_in); or, for the rare step that left one, the overlay snapshot (p.snap_in) ----
    if constexpr (MODE == kModeStep && COH && STD && NA > 0 && NA % 5 == 0 && !F32 && FAST != 0) {
        // (measured: the renderer workgroups FIRST in the grid: 6.44 against 5.33 us per step; renderer waves that start their work
        // out of phase, by up to 0.3 / 0.7 us: 5.56 / 5.53 against 5.35 -- neither role of a launch has slack; the renderer waves
        // of an env in the env's own workgroup, behind its env waves: 5.65 against 5.30, with 2 envs per workgroup 5.40 against 5.33)
        if ((p.snap_mode & 2) && blk >= p.blocks_a) {
            const int eb = a_e_begin + (blk - p.blocks_a) * a_epb + wv;
#ifdef SSD_STAMPS   // renderer waves stamp into the second half of the buffer: [E_total + env][16]
#define SSD_BSTAMP(i, v) do { if (p.stamps && lane == 0 && eb < a_E) p.stamps[((size_t)p.E_total + eb) * 16 + (i)] = (v); } while (0)
#else
#define SSD_BSTAMP(i, v)
#endif
            SSD_BSTAMP(10, __builtin_amdgcn_s_memrealtime());
            SSD_BSTAMP(0, __builtin_readcyclecounter());
            if (eb < a_E) {
                typedef __attribute__((address_space(3))) const uint8_t lds_u8;
                const uint32_t lut_a = a_lut[lane], lut_b = a_lut[lane + 64];
                // (the lane's window offsets: constants of the map, render_views_std)
                const uint4 vt0 = reinterpret_cast<const uint4 *>(p.view_tab)[2 * lane], vt1 = reinterpret_cast<const uint4 *>(p.view_tab)[2 * lane + 1];
                // (the state was written through this very XCD's L2 a launch ago -- env and renderer workgroup numbers agree modulo
                // 8 -- but reading it there needs the CU's L1 out of the way: a buffer_inv sc1 per wave makes the step 33 us;
                // sc0 loads (L2, past L1 only in threadgroup-split mode) returned stale lines in the env waves)
                constexpr int kGridLoads = (fm.S + 1023) / 1024;       // the grid in pieces of 1 KiB, all fetched at once
                u32x4_t g0[kGridLoads];
                // (lanes past the grid's end address beyond the buffer: they get zeros and store nothing)
                const rsrc_t grid_r = make_rsrc(a_world + (size_t)eb * S, (uint32_t)S);
#pragma unroll
                for (int j = 0; j < kGridLoads; ++j) g0[j] = load16_sc1(grid_r, (uint32_t)(lane * 16 + j * 1024));
                uint32_t areg = 0;
                if (lane < N) areg = __hip_atomic_load(a_agents + (size_t)eb * N + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (what the step left beside its state: bits 20 / 21 of agent 0's word -- a list of beam marks / an overlay
                // snapshot; the list entry is fetched with the rest and ignored when the bit says there is none)
                const uint32_t entry = __hip_atomic_load(p.beam_list_in + (size_t)eb * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                {   // the aprons of the layer: '0' (void) cells
                    const uint4 z = make_uint4(0x30303030u, 0x30303030u, 0x30303030u, 0x30303030u);
                    const int n0 = A0 >> 4, n1 = A1 >> 4;
                    for (int i = lane; i < n0 + n1; i += 64)
                        *reinterpret_cast<uint4 *>(i < n0 ? s_world - A0 + i * 16 : s_world + S + (i - n0) * 16) = z;
                }
                s_lut[lane] = lut_a; s_lut[lane + 64] = lut_b;
                const uint32_t flags = rfl(areg) >> 20;
                const bool snapshot = (flags & 2u) != 0, marks = (flags & 1u) != 0;
                // (bit 22: the step's beams took two passes -- the later pass's marks are in the second list)
                uint32_t entry2 = 0;
                constexpr bool kTwoLists = GAME == 1 && NA > 5;        // (as kTwoPasses of the env role below)
                if (kTwoLists && (flags & 4u)) entry2 = __hip_atomic_load(p.beam_list_in + ((size_t)p.E_total + eb) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (snapshot) {                                         // (rare: the overlay as the step left it, instead of the state)
                    const rsrc_t snap_r = make_rsrc(p.snap_in + (size_t)eb * S, (uint32_t)S);
#pragma unroll
                    for (int j = 0; j < kGridLoads; ++j) g0[j] = load16_sc1(snap_r, (uint32_t)(lane * 16 + j * 1024));
                }
#pragma unroll
                for (int j = 0; j < kGridLoads; ++j) {
#ifdef SSD_EXP_RENDER_BLANK     // (experiment switch, wrong pictures: the grid is fetched as ever and then shown blank -- what of the no-fetch bound is the CONTENT's)
                    asm volatile("" : "+v"(g0[j]));
                    g0[j] = u32x4_t{0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u};
#endif
                    if (lane * 16 + j * 1024 < S) *reinterpret_cast<uint4 *>(s_world + lane * 16 + j * 1024) = make_uint4(g0[j].x, g0[j].y, g0[j].z, g0[j].w);
                }
                wave_sync();
                const uint32_t cellb = areg & 0xFFFFu, orientb = (areg >> 16) & 3u;
                if (!snapshot) {
                    // get_map_with_agents (map_env.py:280-302): agents in index order -- the highest index on a cell shows --
                    // then the beams over them
                    // (which agents those are the step's wave has said in bit 19 of their words: it had compared the cells anyway)
                    if (lane < N && (areg & (1u << 19))) s_world[cellb] = agent_glyph((uint32_t)lane);
                    wave_sync();
                    if (marks) {
                        if (entry) s_world[entry & 0xFFFFu] = (uint8_t)(entry >> 16);
                        wave_sync();
                        if constexpr (kTwoLists) {
                            if (entry2) s_world[entry2 & 0xFFFFu] = (uint8_t)(entry2 >> 16);
                            wave_sync();
                        }
                    }
                }
                SSD_BSTAMP(1, __builtin_readcyclecounter());                // layer ready
                const uint32_t kq = (0x8Du >> (2u * orientb)) & 3u;     // rotate_view: UP 0, LEFT 1, DOWN 2, RIGHT 3 (a 2-bit table by orientation code; as a chain of selects it compiled to exec-mask branches)
                const uint32_t s0 = (uint32_t)((int)cellb - 7 * (WP + 1) + (kq >= 2 ? 14 * (WP + 1) : 0));
                render_views_std<NA, true>(lane, WP, kq, s0, (uint32_t)(uintptr_t)(lds_u8 *)s_world, s_lut, p.obs_b + (size_t)eb * N * SSD_OBS_STRIDE, p.obs_wt,
                                           vt0, vt1);
                SSD_BSTAMP(2, __builtin_readcyclecounter());                // stores issued
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SSD_BSTAMP(3, __builtin_readcyclecounter());                // stores landed
                // (a renderer wave that takes TWO envs -- 3 waves per env pair instead of 4 -- made the renderer the launch's
                // critical path: 5.63 against 5.23 us per step)
            }
            SSD_BSTAMP(11, __builtin_amdgcn_s_memrealtime());
            return;
        }
    }
    const int e = a_e_begin + blk * a_epb + wv;             // (a_epb = blockDim.x / 64, without the implicit-argument load)
    constexpr int mode = MODE;                               // compile-time: step / reset / observe
    bool active = e < a_E;                                   // wave-uniform
    if (active && mode == kModeReset && p.mask) active = p.mask[e] != 0;
    SSD_STAMP_RT(10);
    SSD_NOTE(14, t_entry);
    SSD_STAMP(0);

    if (active) {
        const bool is_agent = lane < N;
        // ---- prologue: every global load of the env is issued before the first use, so the HBM / L2
        //      latency is paid once.  First the loads whose addresses come from the preloaded arguments alone (hdr, agents,
        //      the first 1 KiB of the grid = the whole grid of the shipped maps, the colour table, the apple list), then
        //      the ones that need further kernel arguments (actions, order, waste list).
        // (COH: the env's state may have been written a moment ago by a wave on another XCD, i.e. behind another L2: agent-
        // scope loads and stores, dword by dword, instead of cache write-backs / invalidations around ordinary ones.
        // Measured once more in round 3: ORDINARY loads of header, agents and grid -- through this XCD's L2, every packet
        // invalidating the CUs' L1s -- give bit-exact results over 1000 steps of 4096 envs (in practice an env's workgroup lands on
        // the same XCD launch after launch) and are no faster, 5.57 against 5.43 us per step: the write-through stores of the launch
        // before do not leave the lines in L2 to be hit, and the invalidate costs its 0.14 us.  Nothing to gain by speculating on it.
        // And the grid's loads as `sc1 nt`: 6.10 against 5.48 -- the state is served by the memory-side cache, which `nt` goes past.)
        auto cload = [](const uint32_t *ptr) -> uint32_t {
            return kCoh ? __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *ptr;
        };
        // (coherent variants: lanes 0..3 fetch the header's four words.  The v_readlanes right behind the load make the wave
        // wait for it before it issues the others -- two round trips to memory in a row.  Measured, not reasoned: reading the
        // lanes out after the common wait shortens the wave by 0.2 us (diagnostic build: load phase 1605 -> 1138 cycles) and
        // makes the 4096-env step SLOWER, 5.35 -> 5.53 us (Cleanup 5.79 -> 5.95; no difference at 1024 or 16 384 envs):
        // with two chains of launches in flight the step is not the wave's latency alone, and the staggered loads suit it.
        // Measured again at the end of round 3, on that round's kernels: 5.51 -> 5.66 us, Harvest 25 x 38 6.00 -> 6.18 -- a launch's
        // loads are a burst the memory side serves at its own rate, and a wave that asks for everything at once lengthens it.
        // And a third time on round 4's kernels, now behind a switch (-DSSD_EXP_HDR_LATE): Harvest 5.16 -> 5.40, Cleanup 5.48 -> 5.72,
        // Harvest 25 x 38 5.81 -> 5.88, Cleanup 48 x 36 6.60 -> 6.58: profiles/r04_ab/hdr_late_r04.txt)
        // kPre (Cleanup, the map-specific coherent step kernels).  Cleanup's spawn pass starts with a DEPENDENT fetch: the two
        // thresholds of the current waste count (cleanup.py:156-171 through the host's tables), a scalar load that can only go
        // out once the beams have said how many cells they cleaned -- a round trip to L2 in the middle of the wave, with nothing
        // to overlap it.  Here a window of both tables is requested in the prologue instead, lane l taking the entries of
        // (count in the header - l): the spawn pass then reads its pair out of lane `cells cleaned this step` (almost always
        // < 64; else the fetch as before).  (The compiler sinks the window's loads into the spawn pass -- they are vector loads
        // there instead of two scalar loads behind a wait -- and pinning them into the prologue changes nothing: 7.68 against 7.70.)
        // Measured (Cleanup 48 x 36, 10 agents, 2048 envs, alternating fresh processes): 8.18 -> 7.90 us per step; 25 x 18 x 4096:
        // 5.97 -> 5.90.  Tried on top of it and dropped: the respawn's keyed draws (up to 23 per lane on the 48 x 36 map) computed
        // in the prologue too, in the shadow of the grid's loads -- lists and window requested ahead of the grid as inline-asm
        // loads, s_waitcnt vmcnt(<grid pieces>) -- 8.30 us: the load phase grows by more than the spawn pass shrinks (every wave
        // of the launch multiplies at the same moment); the lists requested before the header has arrived: 8.72.  (And a lesson
        // kept: a load the compiler cannot see must not sit under a divergent branch, nor have much code between it and its
        // wait -- the compiler is free to copy or reuse its destination register before the data lands.)
        constexpr bool kPre = GAME == 1 && MODE == kModeStep && COH && FAST != 0;
        uint4 hdr;
        uint32_t hdr_lanes = 0;
        if (kCoh) {
            const uint32_t hv = cload(reinterpret_cast<const uint32_t *>(a_hdr + e) + (lane & 3));
            // (these kernel arguments are fetched while the header is on its way)
            if constexpr (kPre) asm volatile("" ::"s"(p.waste_cells), "s"(p.thr_ca), "s"(p.thr_cw), "s"(p.n_thr));
#if SSD_PIN_EARLY
            // ... and so are the ones the rest of the prologue and the respawn need (the pins further down): requested only after
            // the header had arrived they were two scalar-cache round trips in a row, each holding up the vector loads behind it
            asm volatile("" ::"s"(p.actions), "s"(p.order), "s"(p.num_actions_random), "s"(p.obs));
            if (GAME == 1) asm volatile("" ::"s"(p.waste_cells), "s"(n_waste), "s"(p.thr_ca), "s"(p.thr_cw), "s"(p.n_thr), "s"(p.rew), "s"(p.done), "s"(p.horizon));
            else asm volatile("" ::"s"(p.thr_h32[0]), "s"(p.thr_h32[1]), "s"(p.thr_h32[2]), "s"(p.thr_h32[3]), "s"(p.thr_h_always));
#endif
#ifndef SSD_EXP_HDR_LATE    // (experiment switch: the header's lanes read out after ALL the prologue's loads have been issued -- one round trip)
            hdr = make_uint4(rl(hv, 0), rl(hv, 1), rl(hv, 2), rl(hv, 3));
#else
            hdr_lanes = hv;
#endif
        } else {
            hdr = a_hdr[e];
        }
        uint32_t areg = 0;
        int act_in = -1;
        uint32_t ord_in = 0xFFu;
        // (the env's offset in every [E][N] array -- agents, actions, rewards, dones -- computed ONCE and kept in a scalar register: left
        // to the compiler it is re-derived inside every predicated block that uses it, a 64-bit multiply each)
        uint32_t eN32 = rfl((uint32_t)e * (uint32_t)N);
        asm volatile("" : "+s"(eN32));
        const size_t eN = eN32;
        if (mode != kModeReset && is_agent) areg = cload(a_agents + eN + lane);
        const uint8_t *gsrc = mode == kModeReset ? p.reset_world : a_world + (size_t)e * S;
        // (a known map's grid is kGridLoads pieces of 1 KiB, 16 bytes per lane each: ALL of them go out here -- fetched one after
        // the other behind the first, as the general kernel's loop below does, every piece past the first is another round trip
        // to memory on the wave's path: 25 x 38 Harvest 2 pieces, 48 x 36 Cleanup 3)
        constexpr int kGridLoads = FAST ? (fm.S + 1023) / 1024 : 1;
        uint4 w0[kGridLoads], b0 = make_uint4(0, 0, 0, 0);
        // (coherent variants: agent-scope buffer loads; lanes past the grid's end address beyond the buffer and get zeros)
        const rsrc_t grid_r = make_rsrc(gsrc, (uint32_t)S);              // (unused, and dropped, in the other variants)
        auto issue_grid = [&]() {
#pragma unroll
            for (int j = 0; j < kGridLoads; ++j) {
                w0[j] = make_uint4(0, 0, 0, 0);
                const int off = lane * 16 + j * 1024;
                if constexpr (kCoh) {
                    const u32x4_t v = load16_sc1(grid_r, (uint32_t)off);
                    w0[j] = make_uint4(v.x, v.y, v.z, v.w);
                } else if (off < S) {
                    w0[j] = *reinterpret_cast<const uint4 *>(gsrc + off);
                }
            }
        };
        issue_grid();
        if (lane * 16 < S && mode == kModeObserve && keep_beams) b0 = *reinterpret_cast<const uint4 *>(p.beam + (size_t)e * S + lane * 16);
        // glyph -> RGB table of the observation phase, one copy per wave
        // (a launch that renders nothing -- the env waves of a split rollout, a reset inside one -- needs no colour table)
        uint32_t lut_a = 0, lut_b = 0;
        if (a_obs) { lut_a = a_lut[lane]; lut_b = a_lut[lane + 64]; }
        const bool obs_f32 = F32 && a_obs;
        float4 flut = make_float4(0.f, 0.f, 0.f, 0.f), flut_b = flut;       // glyph -> (r, g, b, -) as float32, entries lane and lane + 64
        if (obs_f32) { flut = reinterpret_cast<const float4 *>(p.f32lut)[lane]; flut_b = reinterpret_cast<const float4 *>(p.f32lut)[lane + 64]; }
        // static cell lists of the map: the first 64 * kLR entries live in registers
        uint32_t alist[kLR], wlist[kLR];
#pragma unroll
        for (int j = 0; j < kLR; ++j) {
            const int idx = lane + 64 * j;
            alist[j] = 0u;
            if (kLR <= 3 || 64 * j < n_apple) alist[j] = (mode != kModeObserve && idx < n_apple) ? a_apple_cells[idx] : 0u;
        }
        // The other kernel arguments are fetched lazily by default, one scalar-cache round trip per basic block that needs
        // one.  Pin what the rest of the prologue and the respawn need into SGPRs here -- the loads above are in flight --
        // so that those fetches go out as one batch.
        // (which ones is measured, not reasoned: pinning rew / done / horizon too gains 1 % in Cleanup and costs 3 % in Harvest)
        asm volatile("" ::"s"(p.actions), "s"(p.order), "s"(p.num_actions_random), "s"(p.obs));
        if (GAME == 1) asm volatile("" ::"s"(p.waste_cells), "s"(n_waste), "s"(p.thr_ca), "s"(p.thr_cw), "s"(p.n_thr), "s"(p.rew), "s"(p.done), "s"(p.horizon));
        else asm volatile("" ::"s"(p.thr_h32[0]), "s"(p.thr_h32[1]), "s"(p.thr_h32[2]), "s"(p.thr_h32[3]), "s"(p.thr_h_always));
        if (stepping && is_agent) {
            // (caller-supplied actions: a coherent launch reads them past the caches too -- between a chain's launches nothing
            // invalidates a CU's L1, and the same action buffer may have held another call's actions a moment ago)
            if (p.num_actions_random <= 0) act_in = (int)cload(reinterpret_cast<const uint32_t *>(p.actions) + eN + lane);
            if (has_order) ord_in = p.order[eN + lane];
        }
#pragma unroll
        for (int j = 0; j < kLR; ++j) {
            const int idx = lane + 64 * j;
            wlist[j] = 0u;
            if (GAME == 1 && 64 * j < n_waste) wlist[j] = (mode != kModeObserve && idx < n_waste) ? p.waste_cells[idx] : 0u;
        }
#ifdef SSD_EXP_HDR_LATE
        if (kCoh) hdr = make_uint4(rl(hdr_lanes, 0), rl(hdr_lanes, 1), rl(hdr_lanes, 2), rl(hdr_lanes, 3));
#endif
        // kPre: thresholds by waste count, lane l holding those of (count in the header - l): the spawn pass will see the count
        // less what this step's CLEAN beams clean (cleanup.py:115 after :94-111), almost always within the window
        uint64_t thr_pa = 0, thr_pw = 0;
        if constexpr (kPre) {
            int ti = (int)((rfl(hdr.w) >> 16) & 0x7FFFu) - lane;
            ti = ti < 0 ? 0 : ti;
            ti = ti < p.n_thr ? ti : p.n_thr - 1;
            thr_pa = p.thr_ca[ti]; thr_pw = p.thr_cw[ti];
        }
        uint32_t key = rfl(hdr.x), t = rfl(hdr.y), episode = rfl(hdr.z);
        if (a_obs) { s_lut[lane] = lut_a; s_lut[lane + 64] = lut_b; }
        if (obs_f32) { reinterpret_cast<float4 *>(s_f32)[lane] = flut; reinterpret_cast<float4 *>(s_f32)[lane + 64] = flut_b; }
        uint32_t status = 0u;
        uint32_t cell = areg & 0xFFFFu, orient = mode == kModeReset ? 2u : (areg >> 16) & 3u;   // lane = agent index
        int rew = 0;
        // grid -> LDS (16 B per lane); beam and occupancy layers start empty
#pragma unroll
        for (int j = 0; j < kGridLoads; ++j)
            if (lane * 16 + j * 1024 < S) {
                *reinterpret_cast<uint4 *>(s_world + lane * 16 + j * 1024) = w0[j];
                *reinterpret_cast<uint4 *>(s_beam + lane * 16 + j * 1024) = j == 0 ? b0 : make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4 *>(s_occ + lane * 16 + j * 1024) = make_uint4(0, 0, 0, 0);
            }
        if (a_obs) {                                         // the aprons of the layer the observations read: '0' (void) cells
            const uint4 z = make_uint4(0x30303030u, 0x30303030u, 0x30303030u, 0x30303030u);
            const int n0 = A0 >> 4, n1 = A1 >> 4;
            for (int i = lane; i < n0 + n1; i += 64)
                *reinterpret_cast<uint4 *>(i < n0 ? s_view - A0 + i * 16 : s_view + S + (i - n0) * 16) = z;
        }
        for (int i = lane * 16 + 1024 * kGridLoads; i < S; i += 1024) {   // (general kernel) maps above 1024 cells
            uint4 bv = make_uint4(0, 0, 0, 0);
            if (mode == kModeObserve && keep_beams) bv = *reinterpret_cast<const uint4 *>(p.beam + (size_t)e * S + i);
            if constexpr (kCoh) {
                const u32x4_t wv = load16_sc1(grid_r, (uint32_t)i);
                *reinterpret_cast<uint4 *>(s_world + i) = make_uint4(wv.x, wv.y, wv.z, wv.w);
            } else {
                *reinterpret_cast<uint4 *>(s_world + i) = *reinterpret_cast<const uint4 *>(gsrc + i);
            }
            *reinterpret_cast<uint4 *>(s_beam + i) = bv;
            *reinterpret_cast<uint4 *>(s_occ + i) = make_uint4(0, 0, 0, 0);
        }
        wave_sync();

        // The grid, agents and header go back to HBM once per launch: right after the respawn of a step / reset launch,
        // after the last step of a rollout launch.
        uint32_t waste_last = 0;
        // Cleanup keeps the number of 'H' cells of the stored grid in the upper half of hdr.w (the lower half is the count
        // the last spawn pass used, ssd_get_waste_count): a step only changes it by the cells its CLEAN beams clean and the
        // one waste cell it may spawn, so compute_permitted_area (cleanup.py:173-179) need not recount the grid.
        uint32_t waste_cur = GAME == 1 ? (rfl(hdr.w) >> 16) & 0x7FFFu : 0u;
        // Bit 31 of hdr.w: "two agents may share a cell" in the stored state (possible only after a contested cell was entered
        // while its occupant was still there, map_env.py:480-483; never after a reset; ssd_set_state computes it).  Exact after
        // every step's consume phase; lets the move phase of a step that finds its agents apart skip the pair-by-pair look.
        bool share = (rfl(hdr.w) >> 31) != 0u;
        // (measured and dropped, round 3: the grid written back right after the beams -- its stores then overlap the spawn pass --
        // and only the 16-byte pieces the spawn pass changed written again at the end: Harvest 4096 envs 5.50 against 5.50 us per
        // step (a first pair of runs said 5.43 against 5.54: box noise), 2048 envs 4.68 against 4.61, 8192 envs 10.56 against
        // 10.37, Cleanup 25 x 18 5.96 against 5.87, 48 x 36 8.14 against 7.73: the stores are not what the wave's end waits for)
        // (`top_bit`, per lane: bit 19 of the agent's word = "the highest index on its cell" -- what shows on the cell,
        // map_env.py:289-297 -- for the renderer workgroups of the next launch: they then need not compare the agents' cells pair by
        // pair again)
        auto write_state = [&](const uint32_t render_flags, const uint32_t top_bit = 0u) {
            uint8_t *gw = a_world + (size_t)e * S;
            if constexpr (kCoh) {
                // (split rollouts: grid and agents go to the other buffer of the pair, this launch's renderer waves read the input)
                uint32_t *ga = a_agents;
                if (p.world_out) { gw = p.world_out + (size_t)e * S; ga = p.agents_out; }
                // (render_flags, split rollouts: what the renderer will find beside this state -- bits 20 / 21 of agent 0's
                // word: a list of beam marks / an overlay snapshot; readers of the word take bits 0..17)
                auto cstore = [](uint32_t *ptr, uint32_t v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
                const rsrc_t out_r = make_rsrc(gw, (uint32_t)S);         // (16-byte write-through stores; past the grid's end: dropped)
                if constexpr (FAST != 0 && SSD_WB_UNROLL) {
                    // a known map's grid is kGridLoads pieces of 1 KiB: all of them are read out of LDS first, into registers of their
                    // own, and the stores then follow each other -- as a loop (LDS read, wait, store, next piece into the same
                    // registers) every piece waited for the one before
                    uint4 v4[kGridLoads];
#pragma unroll
                    for (int j = 0; j < kGridLoads; ++j) {
                        const int off = lane * 16 + j * 1024;
                        v4[j] = *reinterpret_cast<const uint4 *>(s_world + (off < S ? off : 0));
                    }
#pragma unroll
                    for (int j = 0; j < kGridLoads; ++j) {
                        const int off = lane * 16 + j * 1024;
                        const u32x4_t v = {v4[j].x, v4[j].y, v4[j].z, v4[j].w};
                        store16_sc1(out_r, (uint32_t)off, v);
                    }
                } else
                for (int i = lane * 16; i < S; i += 64 * 16) {
                    const uint4 v4 = *reinterpret_cast<const uint4 *>(s_world + i);
                    const u32x4_t v = {v4.x, v4.y, v4.z, v4.w};
                    store16_sc1(out_r, (uint32_t)i, v);                 // one 16-byte write-through store
                }
                if (is_agent) cstore(ga + eN + lane, cell | (orient << 16) | top_bit | (lane == 0 ? render_flags : 0u));
                {   // the header's four words, one per lane: three v_writelane into a register that holds the fourth everywhere (as a
                    // chain of selects on the lane index the compiler made ~20 instructions of nested exec-mask branches of it)
                    uint32_t hw = waste_last | (waste_cur << 16) | (share ? 1u << 31 : 0u);
                    asm("v_writelane_b32 %0, %1, 0" : "+v"(hw) : "s"(key));
                    asm("v_writelane_b32 %0, %1, 1" : "+v"(hw) : "s"(t));
                    asm("v_writelane_b32 %0, %1, 2" : "+v"(hw) : "s"(episode));
                    if (lane < 4) cstore(reinterpret_cast<uint32_t *>(a_hdr + e) + lane, hw);
                }
                if (status && lane == 0) atomicOr(p.status, status);
                return;
            }
            for (int i = lane * 16; i < S; i += 64 * 16) {
                *reinterpret_cast<uint4 *>(gw + i) = *reinterpret_cast<const uint4 *>(s_world + i);
                if (keep_beams)
                    *reinterpret_cast<uint4 *>(p.beam + (size_t)e * S + i) = *reinterpret_cast<const uint4 *>(s_beam + i);
            }
            if (is_agent) a_agents[eN + lane] = cell | (orient << 16);
            if (lane == 0) a_hdr[e] = make_uint4(key, t, episode, waste_last | (waste_cur << 16) | (share ? 1u << 31 : 0u));
            if (status && lane == 0) atomicOr(p.status, status);
        };
        // ---- One pass = one reset or one step of the env.  A step / reset / observe launch makes one pass.  A rollout
        //      launch (rollout.py:58-70) loops: [reset pass when one is due,] step pass, ... with the env resident in LDS
        //      and registers; every step pass writes its observations / rewards / dones to its slot of the output ring. ----
        int k_step = 0;                                       // steps done so far (rollout)
        int to_reset = -1;                                    // steps until the next reset is due (rollout; < 0: never)
        uint32_t slot = 0;                                    // output ring slot of the current step (rollout)
        if (roll) {
            if (p.reset_every > 0) to_reset = (p.reset_every - p.step0 % p.reset_every) % p.reset_every;
            slot = (uint32_t)(p.step0 % p.ring);
        }
        bool in_reset = roll && to_reset == 0;
        // A fused rollout with caller-supplied actions (ssd_rollout_actions): step k reads slot (step0 + k) % action_ring of
        // p.actions [action_ring][E_total][N].  The load for the NEXT step pass goes out at the start of the current pass, so its
        // round trip to memory is off the env's path.
        // (ACTS: a specialisation of its own -- the slot counter and the prefetched actions / orders are loop-carried registers, and
        // the rollout kernels are short of them: carried by the random-action kernel as well they cost it 8 %, 3.7 -> 4.0 us per step)
        constexpr bool roll_acts = roll && ACTS;
        uint32_t aslot = 0;
        int act_next = -1;
        uint32_t ord_next = 0xFFu;
        auto fetch_action = [&]() {
            if (is_agent) {
                const size_t at = ((size_t)aslot * (size_t)p.E_total + (size_t)e) * N + lane;
                act_next = p.actions[at];
                if (has_order) ord_next = p.order[at];            // (the action dict's order, per step: same ring)
            }
            aslot = aslot + 1 == (uint32_t)p.action_ring ? 0u : aslot + 1;
        };
        if constexpr (roll_acts) { aslot = (uint32_t)(p.step0 % p.action_ring); fetch_action(); }
        // (rollout: the output slot's element offset as a running sum -- one 64-bit add per step instead of the 64-bit products of
        // slot x envs x agents and of that x the observation bytes: the rollout kernel is short of scalar issue slots)
        const size_t en_stride = (size_t)p.E_total * N;
        size_t slot_en_run = roll ? (size_t)slot * en_stride : 0;
        for (;;) {
            const bool is_reset = (roll || auto_mode) ? in_reset : (mode == kModeReset);
            const bool is_step = (roll || auto_mode) ? !in_reset : (mode == kModeStep);
            const size_t slot_en = roll ? slot_en_run : 0;   // element offset of the slot in rew / done (rollout: a running sum, below)
            if (roll || (auto_mode && in_reset)) {
                // a fresh pass over the resident env: default priority, empty beam / occupancy layers, and on a reset
                // the grid of reset_map() (:560-564) + custom_reset
                __builtin_amdgcn_s_setprio(0);
                rew = 0;
                for (int i = lane * 16; i < S; i += 64 * 16) {
                    if (is_reset) *reinterpret_cast<uint4 *>(s_world + i) = *reinterpret_cast<const uint4 *>(p.reset_world + i);
                    *reinterpret_cast<uint4 *>(s_beam + i) = make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4 *>(s_occ + i) = make_uint4(0, 0, 0, 0);
                }
                wave_sync();
            }
            uint32_t cleaned = 0;                                 // 'H' cells turned into 'R' by this pass's CLEAN beams
            if (is_reset) {
                // ---- MapEnv.reset (map_env.py:214-249) ----
                episode += 1; t = 0;
                waste_cur = (uint32_t)p.n_waste_reset;            // the grid is reset_world again
                key = env_key(p.seed_lo, p.seed_hi, p.env_base + (uint32_t)e, episode);
                // the grid loaded above is reset_map (:560-564) + custom_reset of the base map
                // setup_agents (harvest.py:46-55 / cleanup.py:118-130): spawn_point (map_env.py:651-662) takes
                // the LAST free point of a fresh shuffle = the free point with the largest (draw, cell);
                // spawn_rotation (:664-667) indexes [LEFT, RIGHT, UP, DOWN].
                const uint32_t pk_pt = phase_key(key, 0, kSpawnPoint), pk_rot = phase_key(key, 0, kSpawnRot);
                for (int i = 0; i < N; ++i) {
                    bool has = false;
                    uint32_t bh = 0, bl = 0;                     // lane-local min of (~draw, ~cell) = max of (draw, cell)
                    for (int s = lane; s < p.n_spawn; s += 64) {
                        const uint32_t ce = p.spawn_cells[s], c = ce & 0xFFFFu;         // grid index | dense index << 16
                        if (s_occ[c] == 0) {
                            const uint32_t kh = ~draw(pk_pt, ((uint32_t)i << 16) | (ce >> 16)), kl = ~c;
                            if (!has || kh < bh || (kh == bh && kl < bl)) { bh = kh; bl = kl; has = true; }
                        }
                    }
                    uint32_t oh, ol, chosen;
                    if (wave_argmin_pair(has, bh, bl, oh, ol)) chosen = ~ol;
                    else { status |= kStNoSpawn; chosen = p.n_spawn ? (p.spawn_cells[0] & 0xFFFFu) : (uint32_t)(WP + 1); }
                    if (lane == i) { cell = chosen; orient = randint(draw(pk_rot, (uint32_t)i), 4); }
                    s_occ[chosen] = agent_glyph((uint32_t)i);    // all lanes, same address, same value
                    wave_sync();
                }
                share = (status & kStNoSpawn) != 0u;             // spawn points are handed out one agent per point
            }

            SSD_STAMP(1);   // state loaded
            int act = -1;
            uint32_t ordv = (uint32_t)lane;                      // action order list, lane k = k-th acting agent
            int nord = N;
            bool all_apart = false;                              // known: no two agents share a cell after the moves
            if (is_step) t += 1;
            // (every keyed draw of this pass -- actions, move shuffle, apples, waste coins and order -- hashes (key, t) first: once,
            // here, instead of once per stream inside whatever branch needs one: 12 scalar instructions per further stream)
            const uint32_t skey = step_key(key, t);
            // (measured and dropped, round 4: the second half for ALL streams at once on the vector unit -- lane s hashing stream s, a
            // needed stream reading its lane: 12 vector instructions instead of 12 scalar ones per stream.  A hash is three 32-bit
            // multiplies, quarter rate on the vector unit: with the pairwise test below on DPP as well, Harvest 5.13 -> 5.33 us per
            // step; the scalar form stays)
            // (... and kept for the ROLLOUT kernel alone, like the pairwise test on DPP below: that kernel is bound by issue slots, the
            // scalar unit's first -- 24 scalar instructions per Harvest step become 12 vector ones and two v_readlane)
            uint32_t vkeys = 0;
            if constexpr (roll && SSD_ROLL_VKEYS) vkeys = stream_key(skey, (uint32_t)lane);       // lane s: the key of stream s
            auto stream_key_of = [&](uint32_t stream) -> uint32_t {
                if constexpr (roll && SSD_ROLL_VKEYS) return rl(vkeys, stream);
                else return stream_key(skey, stream);
            };
            if (is_step) {
                // ---- actions (map_env.py:171-173) ----
                constexpr int kNumActions = GAME == 0 ? 8 : 9;   // harvest.py:44, cleanup.py:70
                if constexpr (roll_acts) {                       // (fused rollout, caller-supplied actions: fetched a pass ago)
                    act = act_next; ord_in = ord_next;
                    fetch_action();                              // (the slot after the call's last step is read and ignored)
                    const bool bad = is_agent && (act < -1 || act >= kNumActions);
                    if (ballot(bad)) { status |= kStBadAction; if (bad) act = -1; }
                } else if (roll || p.num_actions_random > 0) {   // rollout.py:64-65 uniform random actions
                    const uint32_t pk = stream_key_of(kAction);
                    if (is_agent) {
                        act = (int)randint(draw(pk, (uint32_t)lane), (uint32_t)p.num_actions_random);
                        if (!roll && p.actions_out) p.actions_out[eN + lane] = act;
                    }
                } else {                                         // (drawn actions are valid by construction: ssd_step_random checks n)
                    act = act_in;
                    const bool bad = is_agent && (act < -1 || act >= kNumActions);
                    if (ballot(bad)) { status |= kStBadAction; if (bad) act = -1; }
                }
                if (has_order) {
                    ordv = ord_in;
                    if (ordv != 0xFFu && ordv >= (uint32_t)N) { ordv = 0xFFu; status |= kStBadAction; }
                    const uint64_t endm = ballot(ordv == 0xFFu);
                    nord = endm ? __builtin_ctzll(endm) : 64;
                    uint64_t acting = 0;
                    for (int k = 0; k < nord; ++k) acting |= bit(rl(ordv, k));
                    if (!((acting >> lane) & 1)) act = -1;       // agents absent from the action dict do nothing
                }

                // ---- update_moves (map_env.py:357-543) ----
                const bool mover = !SSD_SKIP(0) && is_agent && act >= 0 && act <= 4;              // :383
                if (is_agent && (act == 5 || act == 6)) orient = turn(act, orient);   // :390-392
                uint32_t tcell = cell;
                {   // every lane runs this (non-movers get a zero step), so the wall read is one unconditional LDS load.
                    // MOVE_* vectors (map_env.py:11-15) and rotate_action (:701-716: UP v, LEFT (vc,-vr), RIGHT (-vc,vr),
                    // DOWN -v) as arithmetic on a cyclic direction index (0 (-1,0), 1 (0,1), 2 (1,0), 3 (0,-1)): the
                    // move code picks an index, the orientation adds a quarter-turn count.  Tables are 5-bit fields
                    // holding 8 * index, so the sum (mod 32, which the bit-field extract applies by itself) is the
                    // bit offset of the result's byte.
                    constexpr uint32_t kMoveIdx8 = (0u << 0) | (16u << 5) | (24u << 10) | (8u << 15);   // MOVE_* codes 0..3
                    constexpr uint32_t kTurns8 = (8u << 0) | (24u << 5) | (0u << 10) | (16u << 15);     // LEFT, RIGHT, UP, DOWN
                    const uint32_t sum = __builtin_amdgcn_ubfe(kMoveIdx8, (uint32_t)act * 5u, 5u) + __builtin_amdgcn_ubfe(kTurns8, orient * 5u, 5u);
                    int step;
                    if (FAST) {                                  // row stride fits a signed byte: the table holds the cell offsets
                        const uint32_t off4 = ((uint32_t)(-WP) & 0xFFu) | (1u << 8) | ((uint32_t)WP << 16) | (0xFFu << 24);
                        step = __builtin_amdgcn_sbfe(off4, sum, 8u);
                    } else {
                        step = __builtin_amdgcn_sbfe(0x000100FFu, sum, 8u) * WP + __builtin_amdgcn_sbfe(0xFF000100u, sum, 8u);
                    }
                    const uint32_t cand = (uint32_t)((int)cell + ((mover & (act < 4)) ? step : 0));    // STAY (4) is a zero step
                    // agent.py:105-113 return_valid_pos (the agent's grid agrees with world_map on '@')
                    tcell = (mover & (s_world[cand] != '@')) ? cand : cell;
                }
                uint32_t mvcell = tcell;                         // agent_moves[id] (:410)
                const uint64_t M = ballot(mover);
                // Fast path.  If no mover's target is a cell some OTHER agent stands on and no two movers
                // share a target, every branch of :424-543 degenerates to "each mover takes its target"
                // whatever the shuffle says (STAY / wall-blocked movers target their own cell and stay), and
                // since draws are counter-keyed there is no RNG state to advance.  Otherwise run the
                // reference algorithm in full.
                const uint64_t agents_m = N >= 64 ? ~0ull : bit((uint32_t)N) - 1;
                uint64_t clashm = 0, dupm = 0;                   // lanes (!= j) whose target is agent j's cell or mover j's target
#ifndef SSD_EXP_NOCLASH   // (experiment switch, wrong results: the upper bound of what a cheaper clash test could give)
                // The pairwise comparison.  Up to 16 agents sit in one row of 16 lanes, so lane i can look at lane (i + k) mod 16's
                // cell and target with DPP row rotations, k = 1 .. 15 (never at itself; lanes that hold no agent / no mover carry
                // values no target can equal): its clash bit is "some xor came out zero", one min per partner, all on the vector
                // unit.  As a loop over the agents with v_readlane + scalar mask arithmetic the same test was 13 instructions per
                // agent, 9 of them scalar -- on the unit the CU's four SIMDs share; its upper bound (no test at all, no slow path)
                // measured 5.19 -> 5.00 us per step for Harvest and 7.05 -> 6.22 for Cleanup 48 x 36 with ten agents.  The exact masks
                // the slow path wants (who shares WHICH target) are still made by the loop, there.
                // In the ROLLOUT kernel only: there it pays (fused 3.39 -> 3.29 us per step: that kernel is bound by issue slots, the
                // scalar unit's first); in the per-step kernels of the chains the same change costs (Harvest 5.13 -> 5.32, Cleanup
                // 48 x 36 7.08 -> 7.21, alternating fresh processes): their env waves are a chain of dependent operations, and
                // 24 - 45 dependent vector instructions with DPP hazards are a longer one than the loop's scalar arithmetic.
                bool exact_loop = !roll || (NA == 0 && N > 16);
                if (!exact_loop) {
                    const uint32_t cellx = is_agent ? cell : 0xFFFFFFFEu, tcellm = mover ? tcell : 0xFFFFFFFFu;
                    uint32_t nearest = 0xFFFFFFFFu;
                    // partners at distance d = 1 .. N - 1 in either direction: rotations d and 16 - d (row_ror:k = DPP control 0x120 + k)
                    constexpr int kN = NA > 0 ? NA : 16;
#define SSD_LOOK(k)                                                                                                            \
    if constexpr ((k) < kN || 16 - (k) < kN) {                                                                                  \
        const uint32_t pc = SSD_DPP(0, cellx, 0x120 + (k), false), pt = SSD_DPP(0, tcellm, 0x120 + (k), false);                \
        nearest = umin(nearest, umin(tcell ^ pc, tcell ^ pt));                                                                  \
    }
                    SSD_LOOK(1) SSD_LOOK(2) SSD_LOOK(3) SSD_LOOK(4) SSD_LOOK(5) SSD_LOOK(6) SSD_LOOK(7) SSD_LOOK(8)
                    SSD_LOOK(9) SSD_LOOK(10) SSD_LOOK(11) SSD_LOOK(12) SSD_LOOK(13) SSD_LOOK(14) SSD_LOOK(15)
#undef SSD_LOOK
                    clashm = ballot(is_agent & (nearest == 0u));
                    exact_loop = (clashm & M) != 0;              // a mover clashes: the slow path, with exact masks
                }
                // The per-step kernels compiled for five or ten agents: the waves that take the contested path are their launch's tail
                // (ten agents on Cleanup 48 x 36: 2.5 % of the envs, i.e. some in every launch), so the loop also keeps what that
                // path would otherwise ask again with two more loops over the agents -- the two masks apart, and, per lane, who
                // stands on its target: + 2 instructions per agent for every wave, - 150 and more for the tail.  Cleanup 48 x 36 with
                // ten agents 7.05 -> 6.54 us per step, Harvest 5.18 -> 5.09, its 20-step call 6.41 -> 6.24 (alternating fresh processes).
                constexpr bool kFold = !roll && NA > 0;
                int occ_fold = -1;
                if (exact_loop) {
                    // (a target that is not a mover's is a value no target equals: the loop then has no "is j a mover" test -- a
                    // scalar bit test and a branch per agent -- and the second mask, which only the slow path reads, is made there:
                    // 7 instructions per agent instead of 13)
                    const uint32_t tcellm = mover ? tcell : 0xFFFFFFFFu;
                    clashm = 0;
                    if constexpr (kFold) {
                        for (int j = 0; j < N; ++j) {
                            const uint32_t cj = rl(cell, j), tj = rl(tcellm, j);
                            const bool on = tcell == cj;
                            clashm |= ballot(on) & ~bit(j);
                            dupm |= ballot(tcell == tj) & ~bit(j);
                            occ_fold = on ? j : occ_fold;
                        }
                        clashm |= dupm;
                    } else
                    for (int j = 0; j < N; ++j) {
                        const uint32_t cj = rl(cell, j), tj = rl(tcellm, j);
                        clashm |= (ballot(tcell == cj) | ballot(tcell == tj)) & ~bit(j);
                    }
                }
#endif
                // (1.75 % of the envs of a random-action Harvest step; upper bound of what a faster slow path could give -- every env
                // taking the fast path, wrong results --: 5.13 against 5.35 us per 4096-env step; Cleanup: no difference)
#ifdef SSD_EXP_NOSLOW       // (experiment switch, wrong results: every env takes the fast path; the pairwise test stays)
                const bool slow = false;
#elif defined(SSD_EXP_NOSLOW_KEEPCODE)   // (... and the same with the slow path's code still in the kernel, never entered)
                const bool slow = (clashm & M) != 0 && p.n_spawn < 0;
#else
                const bool slow = (clashm & M) != 0;
#endif
                SSD_NOTE(12, slow ? 1 : 0);
                // (likelihood hints: the rare arms -- contested moves, beams that land one after the other, long lists -- go out of line,
                // the common path falls through)
                // :494-543 for agents on cells of their own whose pending moves have targets of their own: "a waits for the agent on
                // its target" is a graph of disjoint paths and cycles, and the pass loop comes out as: a path moves as a whole iff its
                // head's target is free; it stays as a whole if it ends at an agent that is not moving; a 2-cycle (swap, :524-530)
                // stays; longer cycles rotate (:540-543).  Resolved by pointer jumping over lanes, ceil(log2 N) rounds.
                // `pend`: the lane has a pending move away from its cell; `occ`: the agent on its target (< 0: none).
                auto resolve_chains = [&](const bool pend, const int occ) {
                    int st = pend ? (occ < 0 ? 1 : 2) : 0;       // 0 stays, 1 moves, 2 waits for lane `nx`
                    // Most often nobody waits for an agent that is itself about to move (the agent in the way is firing,
                    // turning, blocked by a wall or staying): then free targets are taken and the rest stay, no jumping.
                    const uint64_t pendm = ballot(pend);
                    if (ballot((st == 2) & (((pendm >> (occ & 63)) & 1ull) != 0))) {
                        const int nx0 = (occ & 63) << 2;
                        int nx = nx0;
                        for (int r = 1; r < N; r <<= 1) {
                            const int s2 = __builtin_amdgcn_ds_bpermute(nx, st), n2 = __builtin_amdgcn_ds_bpermute(nx, nx);
                            const bool waiting = st == 2;
                            st = (waiting & (s2 != 2)) ? s2 : st;
                            nx = (waiting & (s2 == 2)) ? n2 : nx;
                            // (these waves are their launch's tail, and a chain of waiting agents is rarely longer than one: a
                            // round is two dependent trips through the LDS crossbar -- stop when nobody waits any more)
                            if (!ballot(st == 2)) break;
                        }
                        if (ballot(st == 2)) {                   // still waiting: on a cycle
                            const int back = __builtin_amdgcn_ds_bpermute(nx0, nx0);    // my target's target ...
                            st = st == 2 ? (back == (lane << 2) ? 0 : 1) : st;          // ... is me: a swap
                        }
                    }
                    return st == 1;
                };
                // (kFold) some mover's target is taken, but no two movers want the same cell and no cell holds two agents: the loop
                // above has everything the chains need -- no shuffle, no second and third loop over the agents, and the agents are
                // apart afterwards (the consume phase's loop over them goes too)
                const bool chains_only = kFold && slow && (dupm & M) == 0 && !share;
                if (__builtin_expect(!slow, 1)) {
                    if (mover) cell = tcell;
                    all_apart = (clashm & agents_m) == 0;        // nobody's target is anybody else's target or cell
#ifdef SSD_EXP_ALLAPART     // (experiment switch, wrong results: the consume phase never compares the agents' cells)
                    all_apart = true;
#endif
                } else if (chains_only) {
#if defined(SSD_EXP_MUT_CHAINS) && SSD_EXP_MUT_CHAINS == 1   // (mutation switch, wrong results: proves that the tests reach this path)
                    if (mover) cell = tcell;
#else
                    if (resolve_chains(mover & (tcell != cell), occ_fold)) cell = tcell;
#endif
                    all_apart = true;
                } else {                                         // :415 (M != 0 here)
                    __builtin_amdgcn_s_setprio(3);               // the slowest waves of a launch come through here (1-2 % of the envs)
                    uint64_t Hm = M;                             // ids that still have an entry in agent_moves
                    // :424-491 cells wanted by several agents, in lexicographic order (np.unique, axis=0): visited in
                    // ascending cell order (scalar min over the few lanes involved).  The shuffle of :421-423 only
                    // decides who wins such a cell, so it is only computed when there is one (draws are counter-keyed).
                    // These waves are the tail of their launch: forced onto the fast path (wrong results; the code still in the
                    // kernel) Cleanup 48 x 36 with ten agents steps in 6.47 instead of 6.98 us, Harvest in 5.12 instead of 5.15.
                    // (Measured and dropped, round 4: what the two loops below ask -- do two movers share a target, who stands on
                    // each agent's target, do two agents share a cell -- asked through LDS instead, every agent marking its cell
                    // and every mover its target in the two still-empty layers: ~12 instructions and ONE round trip instead of
                    // N x 15 instructions, bit-exact -- and slower: 7.00 -> 7.12 / 5.11 -> 5.24 us per step.  Under this load a
                    // dependent trip through LDS costs a wave more than a hundred scalar instructions do.)
                    if constexpr (!kFold)
                    for (int j = 0; j < N; ++j)                  // lanes (!= j) whose target is mover j's target
                        if ((M >> j) & 1) dupm |= ballot(tcell == rl(tcell, j)) & ~bit(j);
                    uint64_t todo = dupm & M;
                    SSD_NOTE(12, todo ? 2 : 1);
                    bool entered_taken = false;                  // a contested cell was entered while an agent stood on it
                    if (todo) {
                        const int nm = __builtin_popcountll(M);
                        uint32_t perm = 0;                       // lane k: k-th entry of the (shuffled) zipped list
                        const uint32_t pk = stream_key_of(kMove);
                        // draw i is keyed by i alone: lane i computes its own, all at once; only the swaps are sequential
                        const uint32_t jv = randint(draw(pk, (uint32_t)lane), (uint32_t)lane + 1);
                        if constexpr (kFold) {
                            // The unshuffled list: the movers in action order -- without an order array, mover a is entry
                            // (number of movers below a): sent there through the crossbar (the others send to lanes past the list).
                            if (!has_order) {
                                const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
                                perm = (uint32_t)__builtin_amdgcn_ds_permute((int)((mover ? r : 32u + ((uint32_t)lane & 31u)) << 2), lane);
                            } else {
                                int cnt = 0;
                                for (int k = 0; k < nord; ++k) {
                                    const uint32_t a = rl(ordv, k);
                                    if ((M >> a) & 1) { if (lane == cnt) perm = a; ++cnt; }
                                }
                            }
                            // :421-423 np.random.shuffle = Fisher-Yates from the end: for i = nm - 1 .. 1 swap entries i and j_i.  Lane k
                            // follows position k BACKWARDS through the swaps (the last one first) to the entry that ends there: a chain
                            // of vector selects against swap partners that are all read out beforehand -- as swaps of the list itself
                            // every round was three v_readlane + two selects, each waiting for the one before.
                            const uint32_t jeff = lane < nm ? jv : (uint32_t)lane;       // (past the list: a swap with itself)
                            uint32_t pos = (uint32_t)lane;
#pragma unroll
                            for (int i = 1; i < NA; ++i) {
                                const uint32_t j = rl(jeff, i);
                                pos = pos == (uint32_t)i ? j : (pos == j ? (uint32_t)i : pos);
                            }
                            perm = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pos << 2), (int)perm);
                        } else {
                            {
                                int cnt = 0;
                                for (int k = 0; k < nord; ++k) {
                                    const uint32_t a = has_order ? rl(ordv, k) : (uint32_t)k;
                                    if ((M >> a) & 1) { if (lane == cnt) perm = a; ++cnt; }
                                }
                            }
                            for (int i = nm - 1; i >= 1; --i) {  // :421-423 np.random.shuffle = Fisher-Yates from the end
                                const uint32_t j = rl(jv, i);
                                const uint32_t vi = rl(perm, i), vj = rl(perm, j);
                                if (lane == i) perm = vj;
                                if (lane == (int)j) perm = vi;
                            }
                        }
                        while (todo) {
                            uint32_t nxt = 0xFFFFFFFFu;
                            for (uint64_t m = todo; m; m &= m - 1) nxt = umin(nxt, rl(tcell, __builtin_ctzll(m)));
                            const uint64_t Cm = ballot(mover && tcell == nxt);          // contenders (:441-442)
                            todo &= ~Cm;
                            bool cell_free = true;
                            const uint64_t Pm = ballot(is_agent && cell == nxt);        // :449 move in self.agent_pos
                            if (Pm) {
                                const uint32_t occ = 63 - __builtin_clzll(Pm);          // agent_by_pos: last index wins
                                const uint32_t occ_mv = rl(mvcell, occ);
                                if ((Cm >> occ) & 1) cell_free = false;                 // (1) :460
                                else if (!((Hm >> occ) & 1) || occ_mv == nxt) cell_free = false;    // (2) :466-468
                                else if (ballot(__builtin_amdgcn_inverse_ballot_w64(Cm) && cell == occ_mv)) cell_free = false; // (3) :472-476
                            }
                            if (cell_free) {                     // :480-483 first contender in shuffled order moves NOW
                                uint32_t w = 0;
                                if constexpr (kFold) {           // (the list's first entry that is a contender)
                                    const uint64_t firstm = ballot(lane < nm && ((Cm >> (perm & 63u)) & 1ull) != 0);
                                    w = rl(perm, __builtin_ctzll(firstm));
#if defined(SSD_EXP_MUT_CHAINS) && SSD_EXP_MUT_CHAINS == 2   // (mutation switch, wrong results: the lowest contender wins, not the shuffle's first)
                                    w = (uint32_t)__builtin_ctzll(Cm);
#endif
                                    // the agent that waited for the winner's cell finds it empty
                                    occ_fold = occ_fold == (int)w ? -1 : occ_fold;
                                    entered_taken |= Pm != 0;
                                } else {
                                    for (int k = 0; k < nm; ++k) { w = rl(perm, k); if ((Cm >> w) & 1) break; }
                                }
                                if (lane == (int)w) cell = nxt;
                            }
                            if (__builtin_amdgcn_inverse_ballot_w64(Cm)) mvcell = cell;   // :486-491 every contender's move becomes "stay"
                        }
                    }
                    // (kFold) agents that started apart, no contested cell entered over its occupant: they are still apart, and
                    // "who stands on my target" is what the loop at the top found (less the winners' old cells): straight to the chains
                    if (kFold && !share && !entered_taken) {
                        if (resolve_chains(mover & (mvcell != cell), occ_fold)) cell = mvcell;
                        all_apart = true;
                    } else {
                    // :494-543 remaining moves: chains, swaps, cycles.  Who stands on each agent's target, and does any cell
                    // hold two agents (possible after a contested cell was entered while its occupant was still there)?
                    uint64_t overlap = 0;
                    int occ_of_target = -1;
                    for (int j = 0; j < N; ++j) {
                        const uint32_t cj = rl(cell, j);
                        occ_of_target = (mvcell == cj) ? j : occ_of_target;
                        overlap |= ballot(cell == cj) & agents_m & ~bit(j);
                    }
                    if (!overlap) {
                        // Usual case: every cell holds at most one agent and (after the step above) no two pending moves
                        // share a target: resolve_chains
                        if (resolve_chains(__builtin_amdgcn_inverse_ballot_w64(Hm) & (mvcell != cell), occ_of_target)) cell = mvcell;
                    } else
                    while (Hm) {
                        const uint32_t snap_cell = cell, snap_mv = mvcell;              // agent_by_pos (:495), moves_copy (:498)
                        const uint64_t snapH = Hm;
                        uint64_t del = 0;
                        const int n0 = __builtin_popcountll(Hm);
                        for (int k = 0; k < nord; ++k) {                                // agent_moves insertion order = action order
                            const uint32_t a = has_order ? rl(ordv, k) : (uint32_t)k;
                            if (!((snapH >> a) & 1) || ((del >> a) & 1)) continue;      // :500-502
                            const uint32_t m = rl(snap_mv, a);
                            if (ballot(is_agent && cell == m)) {                        // :503 (live positions)
                                const uint64_t sm = ballot(is_agent && snap_cell == m); // :506 (pass-start snapshot)
                                if (!sm) { status |= kStMoveLookup; Hm &= ~bit(a); del |= bit(a); continue; }
                                const uint32_t occ = 63 - __builtin_clzll(sm);
                                const uint32_t ccp = rl(cell, occ), occ_mv = rl(mvcell, occ);
                                const uint32_t cm = ((Hm >> occ) & 1) ? occ_mv : ccp;   // :509
                                const uint32_t pa = rl(cell, a);
                                if (a == occ) { Hm &= ~bit(a); del |= bit(a); }         // (1) :512-514
                                else if (!((snapH >> occ) & 1) || ccp == cm) { Hm &= ~bit(a); del |= bit(a); }  // (2) :518-521
                                else if (occ_mv == pa && m == ccp) {                    // (3) :524-530 swap: both give up
                                    Hm &= ~(bit(a) | bit(occ)); del |= bit(a) | bit(occ);
                                }
                            } else {                                                    // :532-535
                                if (lane == (int)a) cell = m;
                                Hm &= ~bit(a); del |= bit(a);
                            }
                        }
                        if (__builtin_popcountll(Hm) == n0) {    // :540-543 only cycles are left: rotate them
                            if (__builtin_amdgcn_inverse_ballot_w64(Hm)) cell = mvcell;
                            break;
                        }
                    }
                    }
                }
            }

            SSD_STAMP(2);   // moves resolved
            uint64_t highest = 0;                                // agents that are the highest index on their cell (agent_by_pos, :603)
            // this step's beam cells, when one parallel pass traced them all: lane = (shooter, ray, step) -> covered cell, mark
            bool beams_in_regs = true, b_cov = false;
            int b_idx = 0;
            uint32_t b_chr = 0;
            // Cleanup, more shooters than one pass has slots (10 agents: 4.5 % of random-action steps have five or more): the passes
            // follow each other against the map as the earlier ones left it, and the FIRST pass's cells and marks are kept in a
            // second set of registers -- the overlay patch and the split rollouts' beam list then take two entries per lane, the
            // later pass's over the earlier's (firing order, map_env.py:299-300), instead of the step falling back to the merged
            // overlay and its snapshot (+ 3 700 cycles on a wave that the launch then waits for).  Three passes: as before.
            // Measured (48 x 36, 10 agents, 2048 envs, alternating fresh processes): 7.71 -> 7.28 us per step.  Kernels compiled for five
            // agents leave it out (all five shooting: 0.05 % of steps; carrying the second set cost them 1.6 %, 6.05 -> 6.15).
            constexpr bool kTwoPasses = GAME == 1 && !roll && (NA == 0 || NA > 5);
            bool b_cov0 = false;
            int b_idx0 = 0;
            uint32_t b_chr0 = 0;
            if (!is_reset && !SSD_SKIP(1)) {
                // ---- consume (map_env.py:178-181, agent.py:177-183) + occupancy layer ----
                // Index order means: of several agents on one cell the LOWEST index eats the apple, and
                // agent_by_pos / the overlay show the HIGHEST index (:289-297, :603).
                // Wave masks on the scalar unit: agent j is the lowest (highest) index on its cell iff no lower (higher)
                // bit is set among the agents standing where it stands.
                const uint64_t agents = N >= 64 ? ~0ull : bit((uint32_t)N) - 1;
                uint64_t lowest = 0;
                if (all_apart) {                                 // (the move phase already compared every pair of cells)
                    lowest = highest = agents;
                } else
                for (int j = 0; j < N; ++j) {
                    const uint64_t here = ballot(cell == rl(cell, j)) & agents;
                    lowest |= (here & (bit(j) - 1)) ? 0ull : bit(j);
                    highest |= ((here >> j) >> 1) ? 0ull : bit(j);
                }
                share = highest != agents;                       // (the header's bit for the next step)
                const bool eats = is_step & __builtin_amdgcn_inverse_ballot_w64(lowest) & (s_world[cell] == 'A');
                if (eats) { s_world[cell] = ' '; rew += 1; }
                if (__builtin_amdgcn_inverse_ballot_w64(highest)) s_occ[cell] = agent_glyph((uint32_t)lane);
                wave_sync();
            }

            SSD_STAMP(3);   // consume + occupancy
            if (is_step) {
                // ---- update_custom_moves (map_env.py:545-552): beams in action order ----
                const int L = STD ? 5 : p.beam_len;
                const uint32_t rmask = (1u << L) - 1u;
                constexpr int kFire = 7, kClean = 8;
                uint64_t shooters = SSD_SKIP(2) ? 0ull : ballot(is_agent && (act == kFire || (GAME == 1 && act == kClean)));
                SSD_NOTE(13, __builtin_popcountll(shooters));
                SSD_NOTE(15, (GAME == 1 && __builtin_popcountll(shooters) > (STD ? 4 : 64 / (3 * L))) ? 1 : 0);
                if (is_agent && act == kFire) rew -= 1;                                 // agent.py:170-172 fire_beam('F')
                // shooters per pass: 64 / 3L lanes' worth; Cleanup keeps a mask of slots per cell in one byte (below): at most 8
                const int R = 3 * L, G = STD ? 4 : (GAME == 1 && 64 / R > 8) ? 8 : 64 / R;
                if (shooters && (GAME == 0 || !has_order || __builtin_popcountll(shooters) <= G)) {
                    // Harvest: a FIRE beam changes nothing another beam reads (no cell types, no blocking cells, harvest.py:62-67;
                    // 'F' marks and penalties commute), so the rays of up to 64 / 3L shooters are traced in ONE pass:
                    // lane = (shooter slot g, ray q, step kk).  A wave with three shooters costs what one shooter costs.
                    // Cleanup: a CLEAN beam turns the 'H' that stops it into 'R' (cleanup.py:94-111), which a later CLEAN beam
                    // would pass through, and 'F' / 'C' marks overwrite each other in action order -- but only where the beams
                    // of two shooters cover the same cell.  So: trace them all at once against the unchanged map, let every
                    // covered cell be claimed by its shooter's slot, and if any lane finds its cell claimed by another slot in
                    // a way that matters (below) let the slots land one after the other instead.  More shooters than slots
                    // (index action order): group after group, each against the map as the groups before left it.
                    const uint64_t all_shooters = shooters;
                    beams_in_regs = __builtin_popcountll(all_shooters) <= (kTwoPasses ? 2 * G : G);   // one pass covers them all (or two do)
                    int pass_no = 0;
                    const int g = STD ? lane / 15 : lane / R, r = lane - g * R;
                    const int q = (r >= L) + (r >= 2 * L), kk = r - q * L;
                    const int cq = q == 1 ? 1 : q == 2 ? -1 : 0, ck = kk + (q == 0);   // ray cell = pos + cq * right + ck * d (:608-609)
                    const int sh = g * R + q * L;                                       // first lane of this lane's ray
                    const uint32_t packed_agent = cell | (orient << 16) | ((GAME == 1 && act == kClean) ? 1u << 20 : 0u);
                    while (shooters) {
                        if (kTwoPasses && pass_no == 1) { b_cov0 = b_cov; b_idx0 = b_idx; b_chr0 = b_chr; }   // (the first pass's cells and marks)
                        ++pass_no;
                        int a = -1, taken = 0;                                          // slot g <- the g-th remaining shooter
                        for (; taken < G && shooters; ++taken) {
                            const int b = __builtin_ctzll(shooters);
                            a = (g == taken) ? b : a;
                            shooters &= shooters - 1;
                        }
                        const bool inray = a >= 0;
                        const uint32_t ar = (uint32_t)__builtin_amdgcn_ds_bpermute((inray ? a : 0) << 2, (int)packed_agent);
                        const int pc = (int)(ar & 0xFFFFu);
                        const bool clean = GAME == 1 && ((ar >> 20) & 1u) != 0;         // CLEAN beam (cleanup.py:94-111), else FIRE
                        const uint32_t o8 = ((ar >> 16) & 3u) << 3;                     // orientation code * 8: LEFT RIGHT UP DOWN
                        int dlin, rlin;                                                 // d and rotate_right(d) = (-dc, dr) (:607) as cell offsets
                        if (FAST) {
                            const uint32_t b = (uint32_t)WP & 0xFFu, nb = (uint32_t)(-WP) & 0xFFu;
                            dlin = __builtin_amdgcn_sbfe(nb | (b << 8) | (0xFFu << 16) | (1u << 24), o8, 8u);   // -WP, WP, -1, 1
                            rlin = __builtin_amdgcn_sbfe(0xFFu | (1u << 8) | (b << 16) | (nb << 24), o8, 8u);   // -1, 1, WP, -WP
                        } else {
                            const int dr = __builtin_amdgcn_sbfe(0x000001FFu, o8, 8u), dc = __builtin_amdgcn_sbfe(0x01FF0000u, o8, 8u);
                            dlin = dr * WP + dc; rlin = -dc * WP + dr;
                        }
                        // The map's border is wall and a ray ends at the first '@' (:616), so the in-bounds test of :615 can
                        // never be what stops it: cells past the wall are read (harmlessly) and ignored by the first-stop logic.
                        const int cidx = inray ? pc + __mul24(rlin, cq) + __mul24(dlin, ck) : WP + 1;
                        const uint8_t wch = s_world[cidx], och = s_occ[cidx];
                        const bool pass = inray & (wch != '@');                         // :616
                        const bool stopper = pass & ((och != 0) | (clean & (wch == 'H')));   // :621 agents absorb, :639 blocking cell
                        const uint64_t mf = ballot(inray & !pass), ms = ballot(stopper);
                        const uint32_t f = (uint32_t)(mf >> sh) & rmask, st = (uint32_t)(ms >> sh) & rmask;
                        const int ff = f ? __builtin_ctz(f) : L, fs = st ? __builtin_ctz(st) : L;
                        const int len = fs < ff ? fs + 1 : ff;                          // beam covers the stopping cell
                        const bool covered = inray & (kk < len);
                        bool landed = false;                                            // marks and cleaning already applied
                        bool top_clean = clean;                                         // kind of the mark that survives on this lane's cell
                        // (upper bound of what skipping this block could give -- no claims at all, wrong results: Cleanup 25 x 18
                        // 5.68 -> 5.53 us per 4096-env step, 48 x 36 with 10 agents 7.98 -> 7.62 per 2048-env step)
                        if (GAME == 1 && (all_shooters & (all_shooters - 1))) {         // two or more shooters
                            // Which slots cover each cell: one LDS atomic OR per covered lane into the cell's byte of the (still
                            // empty) beam layer.  Sharing a cell matters in two ways.  (1) Beams of different kind: the mark of
                            // the later one survives -- with index action order that is the highest slot in the cell's mask, so
                            // every lane knows the surviving mark without another pass.  (2) Two CLEAN beams on one 'H': the
                            // second finds it cleaned and goes on -- only then do the slots have to land one after the other.
                            // 'F' over 'F' and 'C' over 'C' are the same either way, and FIRE beams do not see waste.
                            const bool one_group = __builtin_popcountll(all_shooters) <= G;
                            uint32_t others = 0;
                            if (one_group || !roll) {
                                if (!one_group) {                                       // the layer holds the earlier groups' marks:
                                    if (covered) s_beam[cidx] = 0;                      // this group's cells start from an empty byte
                                    wave_sync();                                        // (they all get this group's marks below)
                                }
                                if (covered)
                                    __hip_atomic_fetch_or(reinterpret_cast<uint32_t *>(s_beam + (cidx & ~3)), (1u << g) << ((cidx & 3) * 8),
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                wave_sync();
                                const uint32_t m = s_beam[cidx];
                                others = m & ~(1u << g);
                                const uint64_t kb = ballot(clean);
                                uint32_t cs = 0;                                        // slots that hold a CLEAN beam
                                for (int t = 0; t < G; ++t) cs |= (uint32_t)((kb >> (t * R)) & 1u) << t;
                                top_clean = ((cs >> (31 - __builtin_clz(m | 1u))) & 1u) != 0;
                                others = covered ? others : 0u;
                                const bool two_on_waste = clean & (wch == 'H') & ((others & cs) != 0);
                                const bool kinds_differ = (others & (clean ? ~cs : cs)) != 0;
                                landed = ballot(two_on_waste | (has_order & kinds_differ)) != 0;
                            } else {
                                landed = true;      // (rollout kernel: later groups land slot by slot -- rare, and the kernel is short of SGPRs)
                            }
                            if (landed) SSD_NOTE(15, 2);
                            if (landed && has_order) {                                  // slots are not in action order:
                                if (covered) s_beam[cidx] = 0;
                                wave_sync();
                                shooters = all_shooters;                                // -> one after the other, below
                                beams_in_regs = false;
                                break;
                            }
                            if (__builtin_expect(landed, 0)) {
                                // Keep the geometry (cell of every lane, first wall of every ray -- walls do not move) and let
                                // the slots land in order: a CLEAN slot j re-reads its cells from the map as slots < j left it
                                // and finds its stops again; every slot marks, CLEAN ones clean.  (Every mask byte is overwritten:
                                // a ray only ever reaches further than in the first trace.)
                                const int nsl = taken;                                  // slots in this group
                                bool cov_fin = covered, did_clean = false;
                                for (int j = 0; j < nsl; ++j) {
                                    const bool mej = inray & (g == j);
                                    const uint8_t w2 = s_world[cidx];
                                    const uint64_t ms2 = ballot(mej & pass & ((och != 0) | (clean & (w2 == 'H'))));
                                    const uint32_t st2 = (uint32_t)(ms2 >> sh) & rmask;
                                    const int fs2 = st2 ? __builtin_ctz(st2) : L;
                                    const bool cov2 = mej & (kk < (fs2 < ff ? fs2 + 1 : ff));
                                    if (cov2) {
                                        s_beam[cidx] = clean ? 'C' : 'F';
                                        if (clean && w2 == 'H') s_world[cidx] = 'R';
                                    }
                                    cov_fin = mej ? cov2 : cov_fin;
                                    did_clean |= cov2 & clean & (w2 == 'H');
                                    wave_sync();
                                }
                                cleaned += (uint32_t)__builtin_popcountll(ballot(did_clean));
                                b_cov = cov_fin; b_idx = cidx; b_chr = s_beam[cidx];    // the mark that survived on this lane's cell
                            }
                        }
                        if (!landed) {
                            b_cov = covered; b_idx = cidx; b_chr = top_clean ? 'C' : 'F';
                            if (covered) {
                                s_beam[cidx] = (uint8_t)b_chr;                          // :624,:636 firing_points
                                if (clean && wch == 'H') s_world[cidx] = 'R';           // :625-634 cell_types ['H'] -> ['R']
                            }
                            if (GAME == 1) cleaned += (uint32_t)__builtin_popcountll(ballot(covered & clean & (wch == 'H')));
                        }
                        // agent.py:166-168 hit('F'): the last-index agent (:603) on a cell where a FIRE ray stopped loses 50 per ray
                        uint64_t hits = ballot(stopper & !clean & (och != 0) & (kk == fs) & (fs < ff));
                        const bool top = __builtin_amdgcn_inverse_ballot_w64(highest);
                        for (; hits; hits &= hits - 1) {
                            const uint32_t hit_cell = rl((uint32_t)cidx, (uint32_t)__builtin_ctzll(hits));
                            rew -= (top & (cell == hit_cell)) ? 50 : 0;
                        }
                    }
                    wave_sync();
                }
                if (GAME == 1 && shooters) beams_in_regs = false;
                for (int k = 0; GAME == 1 && shooters && k < nord; ++k) {
                    const uint32_t a = rl(ordv, k);
                    if (!((shooters >> a) & 1)) continue;
                    shooters &= ~bit(a);
                    const int aa = (int)rl((uint32_t)act, a);
                    const bool fire = aa == kFire, clean = !fire;                       // harvest.py:62-67, cleanup.py:94-111
                    // update_map_fire (map_env.py:566-649): lane = (ray q, step kk)
                    const int pc = (int)rl(cell, a);
                    int dr, dc;
                    unit_vec((int)rl(orient, a), dr, dc);
                    const int dlin = dr * WP + dc, rlin = -dc * WP + dr;                // d and rotate_right(d) = (-dc, dr) (:607) as cell offsets
                    const int q = (lane >= L) + (lane >= 2 * L), kk = lane - q * L;
                    const bool inray = lane < 3 * L;
                    // :608-609 rays start at pos, pos + right - d, pos - right - d; ray cell kk is start + (kk + 1) * d.
                    // The map's border is wall (ssd_create / ssd_set_state insist) and a ray ends at the first '@'
                    // (:616), so the in-bounds test of :615 can never be what stops it: cells past the wall are
                    // read (harmlessly, possibly outside the grid) and ignored by the first-stop logic below.
                    const int cidx = inray ? pc + (q == 1 ? rlin : q == 2 ? -rlin : 0) + dlin * (kk + (q == 0)) : pc;
                    const uint8_t wch = s_world[cidx], och = s_occ[cidx];
                    const bool pass = inray & (wch != '@');                             // :616
                    const bool stopper = pass & ((och != 0) | (clean & (wch == 'H')));  // :621 agents absorb, :639 blocking cell
                    const uint64_t mf = ballot(inray & !pass), ms = ballot(stopper);
                    const uint32_t f = (uint32_t)(mf >> (q * L)) & rmask, s = (uint32_t)(ms >> (q * L)) & rmask;
                    const int ff = f ? __builtin_ctz(f) : L, fs = s ? __builtin_ctz(s) : L;
                    const int len = fs < ff ? fs + 1 : ff;                              // beam covers the stopping cell
                    wave_sync();
                    if (inray && kk < len) {
                        s_beam[cidx] = clean ? 'C' : 'F';                               // :624,:636 firing_points
                        if (clean && wch == 'H') s_world[cidx] = 'R';                   // :625-634 cell_types ['H'] -> ['R']
                    }
                    cleaned += (uint32_t)__builtin_popcountll(ballot(inray & (kk < len) & clean & (wch == 'H')));
                    if (fire) {                                                         // agent.py:166-168 hit('F'): -50
                        for (int q2 = 0; q2 < 3; ++q2) {
                            const uint32_t f2 = (uint32_t)(mf >> (q2 * L)) & rmask, s2 = (uint32_t)(ms >> (q2 * L)) & rmask;
                            const int ff2 = f2 ? __builtin_ctz(f2) : L, fs2 = s2 ? __builtin_ctz(s2) : L;
                            if (fs2 < ff2) {
                                const uint32_t sl = (uint32_t)(q2 * L + fs2);
                                if (rl((uint32_t)och, sl)) {                            // an agent (not waste) stopped the ray
                                    const uint32_t hit_cell = rl((uint32_t)cidx, sl);
                                    const uint64_t vm = ballot(is_agent && cell == hit_cell);
                                    if (vm && lane == 63 - __builtin_clzll(vm)) rew -= 50;   // :603 last index wins
                                }
                            }
                        }
                    }
                    wave_sync();                                                        // :551-552 updates land before the next shooter
                }
            }

            SSD_STAMP(4);   // beams
            if (mode != kModeObserve) {
                // ---- custom_map_update (map_env.py:187 / :230): respawn ----
                // Lanes walk the map's static apple-point list (row-major, as the reference iterates it).  A list entry is
                // grid index | dense index << 16: the grid index addresses the padded-row layers, the dense index
                // (row * W + col, what prng.py keys the per-cell draws with) feeds the PRNG.
                // The LDS reads of one list entry are unconditional (padding entries point at an interior
                // cell), so they go out as one independent batch.
                uint64_t spawn_bits = 0;                                                // bit j: list entry lane + 64*j gets an apple
                const uint32_t pk_apple = stream_key_of(kApple);
                const int a_iters = (n_apple + 63) >> 6;
                const uint32_t safe = (uint32_t)(WP + 1);                               // cell (1,1)
                uint32_t waste_cell = 0xFFFFFFFFu;
                uint32_t waste_count = 0;                                               // #'H' the probabilities were computed from
                if (!SSD_SKIP(3)) {
                if (GAME == 0) {
                    // harvest.py:75-104 spawn_apples.  Apple points are interior cells (the border is wall),
                    // so the 3x3 neighbourhood (j*j + k*k <= 2 on the radius-2 box, :90-92) is always in bounds.
                    // Threshold by neighbour count as a sum of steps on n >= k.  (An `n == 0 ? a : n == 1 ? b : ...`
                    // chain is turned into a switch lookup table by the compiler: an indexed vector load from the
                    // kernarg buffer in memory -- an L2 round trip per list entry on the critical path.)
                    const uint32_t t0 = p.thr_h32[0], d1 = p.thr_h32[1] - t0, d2 = p.thr_h32[2] - p.thr_h32[1],
                                   d3 = p.thr_h32[3] - p.thr_h32[2];
                    // 3x3 apple count, threshold and keyed draw of one candidate cell (:90-103)
                    // (measured and dropped, round 4: the three cells of a row as ONE unaligned 4-byte LDS read and the apples among them by
                    // byte arithmetic -- 3 reads + 13 vector instructions instead of 8 + ~30; the compiler emits ds_read_b32 for it, and
                    // the hardware serves it slowly: 5.14 -> 5.35 us per 4096-env step, the fused kernel 3.39 -> 3.86)
                    auto wins = [&](uint32_t ce) -> bool {
                        const int c = (int)(ce & 0xFFFFu);
                        uint32_t n = 0;
#pragma unroll
                        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
                            for (int dc = -1; dc <= 1; ++dc)
                                if (dr != 0 || dc != 0) n += s_world[c + dr * WP + dc] == 'A';
                        const uint32_t thr = t0 + (n >= 1 ? d1 : 0u) + (n >= 2 ? d2 : 0u) + (n >= 3 ? d3 : 0u);   // SPAWN_PROB[min(n, 3)]
                        const bool always = ((p.thr_h_always >> (n < 3 ? n : 3u)) & 1u) != 0;
                        return (draw(pk_apple, ce >> 16) < thr) | always;
                    };
                    // Pass 1 (cheap): which list entries are candidates at all -- a cell without an apple (:88; "nobody stands
                    // on it" is checked for the few winners only).  The unused entries of the list registers are 0 = grid
                    // cell 0, a wall, and an apple point is never a wall: "neither 'A' nor '@'" needs no validity test.
                    bool el[kLR];
                    uint64_t em[kLR];
                    int total = 0;
#pragma unroll
                    for (int j = 0; j < kLR; ++j) {
                        el[j] = false; em[j] = 0;
                        if (kLR <= 3 || 64 * j < n_apple) {                             // (wave-uniform: skips unused list registers)
                            const uint8_t ch = s_world[alist[j] & 0xFFFFu];
                            el[j] = (ch != 'A') & (ch != '@');
                            em[j] = ballot(el[j]);
                            total += __builtin_popcountll(em[j]);
                        }
                    }
#ifdef SSD_EXP_NOCOMPACT    // (experiment switch: every lane evaluates its own list entries -- no compaction, one LDS round trip fewer)
                    if (false) {
#else
                    if (__builtin_expect(a_iters <= kLR && total <= 64, 1)) {
#endif
                        // Usual case: at most 64 candidates among the (up to 512) apple points.  Compact them through
                        // 128 B of LDS scratch so that ONE pass of lanes does the stencil + draw instead of three.
                        if (total) {
                            int base = 0;
#pragma unroll
                            for (int j = 0; j < kLR; ++j) {
                                if (kLR > 3 && 64 * j >= n_apple) continue;
                                const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(em[j] >> 32),
                                                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)em[j], 0u));
                                if (el[j]) s_tmp[slot] = alist[j];
                                base += __builtin_popcountll(em[j]);
                            }
                            wave_sync();
                            const bool mine = lane < total;
                            const uint32_t c = mine ? s_tmp[lane] : safe;
                            const bool hit = mine & wins(c) & (s_occ[c & 0xFFFFu] == 0);
                            wave_sync();                                                // every count used the pre-spawn map (:73)
                            if (hit) s_world[c & 0xFFFFu] = 'A';
                        }
                    } else {
                        // general form: every lane evaluates its own list entries
#pragma unroll
                        for (int j = 0; j < kLR; ++j) {
                            if (kLR > 3 && 64 * j >= n_apple) continue;
                            const uint32_t c = el[j] ? alist[j] : safe;
                            spawn_bits |= (el[j] & wins(c) & (s_occ[c & 0xFFFFu] == 0)) ? bit(j) : 0ull;
                        }
                        for (int j = kLR; j < a_iters; ++j) {
                            const int idx = lane + 64 * j;
                            const bool valid = idx < n_apple;
                            const uint32_t c = valid ? a_apple_cells[idx] : safe;
                            const bool cand = valid & (s_world[c & 0xFFFFu] != 'A') & (s_occ[c & 0xFFFFu] == 0);
                            spawn_bits |= (cand & wins(c)) ? bit(j) : 0ull;
                        }
                    }
                } else {
                    // cleanup.py:113-116: compute_probabilities (:156-171) from the current waste count, then
                    // spawn_apples_and_waste (:132-154).  Thresholds come from a host-computed table.
                    const uint32_t nh = waste_cur - cleaned;                            // compute_permitted_area (:173-179), kept incrementally
                    waste_count = nh;
                    uint64_t thr_a = 0, thr_w = 0;
                    bool thr_have = false;                                              // (wave-uniform) the prologue's window serves
                    if constexpr (kPre) {
                        const uint32_t back = rfl(cleaned);                             // the window's lane: cells cleaned this step
                        if (back < 64u) {
                            thr_a = (uint64_t)rl((uint32_t)thr_pa, back) | ((uint64_t)rl((uint32_t)(thr_pa >> 32), back) << 32);
                            thr_w = (uint64_t)rl((uint32_t)thr_pw, back) | ((uint64_t)rl((uint32_t)(thr_pw >> 32), back) << 32);
                            thr_have = true;
                        }
                    }
                    if (!thr_have) {                                                    // (the table's size is fetched only here)
                        const uint32_t ni = nh < (uint32_t)p.n_thr ? nh : (uint32_t)p.n_thr - 1;
                        thr_a = p.thr_ca[ni]; thr_w = p.thr_cw[ni];
                    }
                    auto apple = [&](int j, uint32_t c, bool valid) {                   // :135-141
                        c = valid ? c : safe;
                        const uint8_t w = s_world[c & 0xFFFFu], o = s_occ[c & 0xFFFFu];
                        const bool hit = valid & (w != 'A') & (o == 0) & ((uint64_t)draw(pk_apple, c >> 16) < thr_a);
                        spawn_bits |= hit ? bit(j) : 0ull;
                    };
                    if (thr_a) {                                                        // (density >= thresholdDepletion: nothing grows, :161-163)
#pragma unroll
                        for (int j = 0; j < kLR; ++j)
                            if (64 * j < n_apple) apple(j, alist[j], lane + 64 * j < n_apple);   // (wave-uniform: skips unused list registers)
                        for (int j = kLR; j < a_iters; ++j) {
                            const int idx = lane + 64 * j;
                            apple(j, idx < n_apple ? a_apple_cells[idx] : 0u, idx < n_apple);
                        }
                    }
                    if (thr_w) {
                        // :144-153 shuffled scan, first non-'H' point whose coin succeeds (at most one per step):
                        // order = ascending (ORDER draw, cell), coin keyed by cell.
                        const uint32_t pk_coin = stream_key_of(kWasteCoin), pk_ord = stream_key_of(kWasteOrder);
                        bool has = false;
                        uint32_t bh = 0, bl = 0;
                        auto waste = [&](uint32_t ce, bool valid) {
                            ce = valid ? ce : safe;
                            const uint32_t c = ce & 0xFFFFu;                             // ties break on the cell: grid and dense order agree
#ifdef SSD_EXP_WASTE_ONE_DRAW   // (experiment switch, results differ from the oracle's: one keyed draw per waste point instead of two)
                            const uint32_t kh = draw(pk_ord, ce >> 16);
                            const bool cand = valid & (s_world[c] != 'H') & ((uint64_t)kh < thr_w);
#else
                            const bool cand = valid & (s_world[c] != 'H') & ((uint64_t)draw(pk_coin, ce >> 16) < thr_w);
                            const uint32_t kh = draw(pk_ord, ce >> 16);
#endif
                            const bool better = cand & (!has | (kh < bh) | ((kh == bh) & (c < bl)));
                            bh = better ? kh : bh; bl = better ? c : bl; has = has | cand;
                        };
                        const int w_iters = (n_waste + 63) >> 6;
#pragma unroll
                        for (int j = 0; j < kLR; ++j)
                            if (64 * j < n_waste) waste(wlist[j], lane + 64 * j < n_waste);
                        for (int j = kLR; j < w_iters; ++j) {
                            const int idx = lane + 64 * j;
                            waste(idx < n_waste ? p.waste_cells[idx] : 0u, idx < n_waste);
                        }
                        uint32_t oh, ol;
                        if (wave_argmin_pair(has, bh, bl, oh, ol)) waste_cell = ol;
                    }
                }
                }
                wave_sync();                                                            // counts use the pre-spawn map (harvest.py:73)
#pragma unroll
                for (int j = 0; j < kLR; ++j)
                    if ((kLR <= 3 || 64 * j < n_apple) && ((spawn_bits >> j) & 1)) s_world[alist[j] & 0xFFFFu] = 'A';
                for (int j = kLR; j < a_iters; ++j)
                    if ((spawn_bits >> j) & 1) s_world[a_apple_cells[lane + 64 * j] & 0xFFFFu] = 'A';
                if (waste_cell != 0xFFFFFFFFu) s_world[waste_cell] = 'H';               // may land under an agent
                if (GAME == 1) waste_cur = waste_count + (waste_cell != 0xFFFFFFFFu ? 1u : 0u);
                wave_sync();

                SSD_STAMP(5);   // respawn
                // ---- write the env back (grid, agents, header: a rollout does that once, after its last step) and
                //      this step's rewards and dones ----
                waste_last = waste_count;
                if (!roll) {
                    uint32_t render_flags = 0;
                    if constexpr (stepping && COH) {
                        if ((p.snap_mode & 1) && is_step) {
                            render_flags = (!keep_beams && beams_in_regs) ? (ballot(b_cov | (kTwoPasses & b_cov0)) ? 1u << 20 : 0u) : 1u << 21;
                            if constexpr (kTwoPasses)
                                if (!keep_beams && beams_in_regs && ballot(b_cov0)) render_flags |= 1u << 22;   // (a second list: the later pass's marks)
                        }
                    }
                    write_state(render_flags, __builtin_amdgcn_inverse_ballot_w64(highest) ? 1u << 19 : 0u);
                }
                if (is_agent && is_step) {
                    // compute_reward (:208); get_done -> False (:209); with a horizon set, the episode ends after `horizon` steps
                    const uint8_t dn = (p.horizon > 0 && t >= (uint32_t)p.horizon) ? 1 : 0;
                    if constexpr (COH) {                                                // (write-through: nothing stays dirty in L2)
                        if (p.rew) __hip_atomic_store(p.rew + slot_en + eN + lane, rew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (p.done) __hip_atomic_store(p.done + slot_en + eN + lane, dn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        if (p.rew) p.rew[slot_en + eN + lane] = rew;
                        if (p.done) p.done[slot_en + eN + lane] = dn;
                    }
                }
                wave_sync();
            }

            SSD_STAMP(6);   // write-back issued
            // ---- get_map_with_agents (map_env.py:280-302): world <- agents <- beams (in place, except in a rollout, whose
            //      world layer lives on).  After a step whose beams one parallel pass traced, the cells that differ from the
            //      world are in registers -- agents' cells, beam cells -- so two predicated byte stores do it (a rollout copies
            //      the layer first, 16 B per lane); otherwise the three layers are merged 4 cells per op. ----
            const bool patch = is_step && !keep_beams && beams_in_regs;
            // Split rollouts (ssd_capi.hip): the env's wave does not render observations -- extra waves of the NEXT step's launch
            // do (the renderer role above), from the very state that launch steps from (this wave writes its state into the other
            // of two buffers, the next launch reads it, nobody writes it meanwhile) plus what the state does not hold: this step's
            // beam marks, one list entry per lane (cell | mark << 16; 0: none).  The overlay is then not built here at all.  Rare
            // steps whose beams did not stay in registers (several shooter groups, CLEAN conflicts) build the overlay as usual and
            // leave it in a snapshot instead; entry 0xFFFFFFFF tells the renderer so.  The observation phase (1.2 us of a 5.7 us
            // step) thereby leaves the chain of dependent launches.
            bool leave_overlay = false;                       // wave-uniform
            if constexpr (stepping && COH) {
                if ((p.snap_mode & 1) && is_step) {
                    leave_overlay = patch;
                    // (which of the two the step left is said by two spare bits of agent 0's state word, written with the state
                    // above; the list is only written when there is a mark at all.  One dword per lane: staging it through LDS
                    // for 16-byte stores cost more on this wave's path than it saved, 5.29 against 5.15 us per step)
                    // (two passes: the earlier pass's entries in the list proper, the later one's in a second list behind all envs' first)
                    const bool two = kTwoPasses && ballot(b_cov0) != 0;
                    const uint32_t ent = b_cov ? ((uint32_t)b_idx | (b_chr << 16)) : 0u, ent0 = b_cov0 ? ((uint32_t)b_idx0 | (b_chr0 << 16)) : 0u;
                    if (patch && ballot(b_cov | b_cov0))
                        __hip_atomic_store(p.beam_list + (size_t)e * 64 + lane, two ? ent0 : ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (patch && two)
                        __hip_atomic_store(p.beam_list + ((size_t)p.E_total + e) * 64 + lane, ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (leave_overlay) {
            } else if (patch && !SSD_SKIP(5)) {
                if (roll)
                    for (int i = lane * 16; i < S; i += 64 * 16)
                        *reinterpret_cast<uint4 *>(s_view + i) = *reinterpret_cast<const uint4 *>(s_world + i);
                if (__builtin_amdgcn_inverse_ballot_w64(highest)) s_view[cell] = agent_glyph((uint32_t)lane);   // :289-297
                if (kTwoPasses && b_cov0) s_view[b_idx0] = (uint8_t)b_chr0;                                     // (an earlier pass's marks first)
                if (b_cov) s_view[b_idx] = (uint8_t)b_chr;                                                      // :299-300, over the agents
            } else if ((!roll || is_step) && !SSD_SKIP(5))   // (diagnostic builds: skip bit 5 = no overlay, to price it)
            for (int i = lane * 4; i < S; i += 64 * 4) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(s_world + i);
                const uint32_t o = *reinterpret_cast<const uint32_t *>(s_occ + i);
                const uint32_t b = *reinterpret_cast<const uint32_t *>(s_beam + i);
                const uint32_t mo = nonzero_bytes(o), mb = nonzero_bytes(b);
                uint32_t v = (w & ~mo) | (o & mo);
                v = (v & ~mb) | (b & mb);
                *reinterpret_cast<uint32_t *>(s_view + i) = v;
            }
            wave_sync();
            if constexpr (stepping && COH) {
                if ((p.snap_mode & 1) && is_step && !patch) {                 // (the rare form: the overlay as a snapshot)
                    const rsrc_t snap_r = make_rsrc(p.snap + (size_t)e * S, (uint32_t)S);
                    for (int i = lane * 16; i < S; i += 64 * 16) {
                        const uint4 v4 = *reinterpret_cast<const uint4 *>(s_view + i);
                        store16_sc1(snap_r, (uint32_t)i, u32x4_t{v4.x, v4.y, v4.z, v4.w});
                    }
                }
            }
            SSD_STAMP(7);   // overlay built
            SSD_STAMP(8);

            // ---- per-agent observations (agent.py:76-78 -> utility_funcs.py:59-114 window with '0' padding,
            //      map_env.py:316-339 colour LUT, :669-689 rotate_view).  The wave renders its own env's agents
            //      one after the other: lane = 4 consecutive cells of the V x V view (12 contiguous output
            //      bytes), so the view coordinates are per-lane constants, the agent's position and rotation
            //      are scalars, and one wave store covers up to 768 contiguous bytes of the uint8 obs tensor.
            //      Thanks to the padded grid layout a view cell is ONE multiply-add away from its LDS address.
            //      An agent's block starts at a multiple of V*V*3 = 675 bytes, i.e. at any byte alignment:
            //      the 12-byte stores rely on gfx9's unaligned global access. ----
            if (a_obs && (!roll || is_step)) {
                typedef __attribute__((address_space(3))) const uint8_t lds_u8;
                const int V = STD ? 15 : p.V, v = STD ? 7 : p.view_len, VV = V * V;
                // (diagnostic builds, skip bit 4: all envs write the first 64 envs' blocks -- same instructions, no HBM write stream)
                uint8_t *out_env = a_obs + (slot_en + (size_t)(SSD_SKIP(4) ? (e & 63) : e) * N) * VV * 3;
                if constexpr (roll) {
                    // (wave-uniform by construction, but in the rollout kernels' loop the compiler does not always see it -- and the
                    // store helpers take their base address in scalar registers)
                    const uint64_t ob = reinterpret_cast<uint64_t>(out_env);
                    out_env = reinterpret_cast<uint8_t *>(((uint64_t)rfl((uint32_t)(ob >> 32)) << 32) | (uint64_t)rfl((uint32_t)ob));
                }
                // Per-agent constants, computed once with lane = agent and read back as scalars in the loop.
                // Window cell (a, b) of an agent on grid cell `cell` is grid cell cell + (a - v) * WP + (b - v).
                // The view is rot90^k of the window (rotate_view, map_env.py:669-689; UP 0, LEFT 1, DOWN 2,
                // RIGHT 3; reset observations are not rotated): view cell (i, j) shows window cell
                //   k=0 (i, j)   k=1 (j, V-1-i)   k=2 (V-1-i, V-1-j)   k=3 (V-1-j, i)
                // i.e. with lin0 = i*WP + j, lin1 = j*WP + (V-1-i) and C = (V-1)*(WP+1):
                //   address = base + lin0 | base + lin1 | base + C - lin0 | base + C - lin1,   base = cell - v*(WP+1).
                uint32_t a_s0 = 0, a_k = 0;
                if (is_agent) {
                    a_k = (is_step || (!roll && !auto_mode && p.rotate)) ? ((0x8Du >> (2u * orient)) & 3u) : 0u;
                    a_s0 = (uint32_t)((int)cell - v * (WP + 1) + (a_k >= 2 ? (V - 1) * (WP + 1) : 0));
                }
                const uint32_t world_lds = (uint32_t)(uintptr_t)(lds_u8 *)s_view;       // LDS byte address of grid cell 0
                if (F32 && a_obs && VV * 3 >= 4) {
                    // float32 observations: an agent's block is V*V*3 floats (2700 B).  Lane l of store k4 writes the four
                    // floats k4*256 + 4l .. +3 of it, so that every store instruction covers 1 KB of contiguous memory --
                    // with a cell's three floats kept together (48 B per lane, 16 B per store) every 64-byte line was written
                    // by three instructions, 16 B at a time, and the L2's write transactions, not HBM, bounded the mode
                    // (17.1 us per 4096-env step).  Four consecutive floats span two cells: both are looked up (glyph, then the
                    // glyph's float triple), and the lane picks its four of their six floats by f0 % 3.  The lane that would
                    // run past the block's end starts four floats before the end instead (same values, written twice).
                    const int nf = VV * 3, n_st = (nf + 255) >> 8;
                    float *base_env = reinterpret_cast<float *>(a_obs) + (slot_en + (size_t)e * N) * (size_t)nf;
                    const rsrc_t f32_r = make_rsrc(base_env, (uint32_t)(N * nf * 4));
                    for (int k4 = 0; k4 < n_st; ++k4) {
                        const int f_raw = k4 * 256 + 4 * lane;
                        const bool on = f_raw < nf;
                        const int f0 = f_raw > nf - 4 ? nf - 4 : f_raw;
                        const int c0 = (int)(((uint32_t)f0 * 21846u) >> 16);             // f0 / 3 (f0 < 2^15)
                        const int r = f0 - 3 * c0;
                        int M0[2], M1[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int pp = c0 + q;
                            const int i = STD ? pp / 15 : (int)(((uint32_t)pp * p.v_magic16) >> 16), j = pp - i * V;
                            M0[q] = i * WP + j;
                            M1[q] = j * WP + (V - 1 - i);
                        }
                        const bool r0 = r == 0, r1 = r == 1;
                        for (int ag = 0; ag < N; ++ag) {
                            const uint32_t k = rl(a_k, ag);
                            const int s0 = (int)(rl(a_s0, ag) + world_lds);
                            const int sgn = k >= 2 ? -1 : 1;
                            float4 t[2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const uint32_t ad = (uint32_t)(s0 + sgn * ((k & 1) ? M1[q] : M0[q]));
                                t[q] = reinterpret_cast<const float4 *>(s_f32)[*(lds_u8 *)(uintptr_t)ad];
                            }
                            f32x4_t o;
                            o.x = r0 ? t[0].x : r1 ? t[0].y : t[0].z;
                            o.y = r0 ? t[0].y : r1 ? t[0].z : t[1].x;
                            o.z = r0 ? t[0].z : r1 ? t[1].x : t[1].y;
                            o.w = r0 ? t[1].x : r1 ? t[1].y : t[1].z;
                            if (on) store16_wt(f32_r, (uint32_t)(ag * nf * 4), (uint32_t)f0 * 4u, o, p.obs_wt);
                        }
                    }
                } else if constexpr (STD && NA > 0 && NA % 5 == 0 && !F32) {
                    // Specialised kernels: five agents per pass (render_views_std)
                    render_views_std<NA>(lane, WP, a_k, a_s0, world_lds, s_lut, out_env, p.obs_wt);
                } else {
                // (the env's observation block as a buffer: uint8 [N][V][V][3], or the same cells as float32)
                const rsrc_t gen_r = obs_f32 ? make_rsrc(reinterpret_cast<float *>(a_obs) + (slot_en + (size_t)e * N) * VV * 3, (uint32_t)(N * VV * 12))
                                             : make_rsrc(out_env, (uint32_t)(N * VV * 3));
                for (int base = 0; base < VV; base += 256) {
                    // A lane renders 4 consecutive cells = one 12-byte store.  V*V is not a multiple of 4 (225 = 56*4 + 1):
                    // the lane holding the leftover cells starts 4 cells before the end instead, re-rendering up to 3
                    // cells of its neighbour (same bytes, written twice) so that EVERY store is a full 12 bytes and the
                    // wave never takes a divergent byte-store path.  Views under 4 cells (view_len 0) use byte stores.
                    const int pp_raw = base + 4 * lane;
                    const bool lane_on = pp_raw < VV;
                    const int pp0 = (VV >= 4 && pp_raw > VV - 4) ? VV - 4 : pp_raw;     // lanes past the end repeat the last one
                    int L0[4], L1[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int pp = pp0 + q;
                        const int i = STD ? pp / 15 : (int)(((uint32_t)pp * p.v_magic16) >> 16), j = pp - i * V;   // pp / V, pp % V
                        // 24-bit multiply-adds (full rate; a plain `*` becomes a quarter-rate 32-bit multiply here)
                        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(L0[q]) : "v"(i), "s"(WP), "v"(j));
                        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(L1[q]) : "v"(j), "s"(WP), "v"(V - 1 - i));
                    }
                    uint32_t off3 = (uint32_t)pp0 * 3u;                                 // byte offset of the lane's cells in an agent's block
                    asm volatile("" : "+v"(off3));                                      // keep it in a register (else re-derived per agent)
                    const int ncell = lane_on ? VV - pp0 : 0;                           // >= 4 whenever VV >= 4
                    for (int ag = 0; ag < N; ++ag) {
                        const uint32_t k = rl(a_k, ag);
                        const uint32_t s0 = rl(a_s0, ag) + world_lds;
                        const int sgn = k >= 2 ? -1 : 1;                                // one VGPR per agent: v_mad takes one scalar operand
                        uint32_t addr[4], px[4];
                        // wave-uniform branch on the rotation's parity instead of a per-cell select (the asm is volatile so
                        // that the two arms are not merged back into selects)
                        if (k & 1) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(addr[q]) : "v"(L1[q]), "v"(sgn), "s"(s0));
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(addr[q]) : "v"(L0[q]), "v"(sgn), "s"(s0));
                        }
                        // a cell outside the map reads the '0' of the row padding / aprons (utility_funcs.py:94-114)
                        uint32_t gl[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) gl[q] = *(lds_u8 *)(uintptr_t)addr[q];
                        const size_t cell0 = (size_t)ag * VV + pp0;                     // first of this lane's cells within the env
                        if (obs_f32) {
                            // float32 mode: 3 floats per cell through the glyph -> float32 colour table (one 16-byte LDS read
                            // per cell; the table holds float32 of the reference's float64 values, exact), 48 contiguous
                            // bytes per lane (three 16-byte stores; an agent block starts at a multiple of 2700 B)
                            float *dstf = reinterpret_cast<float *>(a_obs) + ((slot_en + (size_t)e * N) * VV + cell0) * 3;
                            float f[12];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 v = reinterpret_cast<const float4 *>(s_f32)[gl[q]];
                                f[q * 3 + 0] = v.x; f[q * 3 + 1] = v.y; f[q * 3 + 2] = v.z;
                            }
                            if (VV >= 4) {
                                if (lane_on) {
#pragma unroll
                                    for (int k4 = 0; k4 < 3; ++k4) {
                                        f32x4_t v4 = {f[4 * k4], f[4 * k4 + 1], f[4 * k4 + 2], f[4 * k4 + 3]};
                                        store16_wt(gen_r, (uint32_t)(ag * VV * 12), (uint32_t)pp0 * 12u + 16u * k4, v4, p.obs_wt);
                                    }
                                }
                            } else {
#pragma unroll
                                for (int q = 0; q < 3; ++q)
                                    if (q < ncell) { dstf[q * 3] = f[q * 3]; dstf[q * 3 + 1] = f[q * 3 + 1]; dstf[q * 3 + 2] = f[q * 3 + 2]; }
                            }
                            continue;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) px[q] = s_lut[gl[q]];
                        uint8_t *dst = out_env + (size_t)ag * VV * 3 + off3;
                        if (VV >= 4) {
                            if (lane_on) {
                                u32x3_t d;
                                // 4 x (r,g,b) -> 12 bytes with three byte permutes (v_perm_b32: selector bytes 0-3 pick from
                                // the second operand, 4-7 from the first)
                                d.x = __builtin_amdgcn_perm(px[1], px[0], 0x04020100u);   // r0 g0 b0 r1
                                d.y = __builtin_amdgcn_perm(px[2], px[1], 0x05040201u);   // g1 b1 r2 g2
                                d.z = __builtin_amdgcn_perm(px[3], px[2], 0x06050402u);   // b2 r3 g3 b3
                                store12_wt(gen_r, (uint32_t)(ag * VV * 3), off3, d, p.obs_wt);
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < 3; ++q)
                                if (q < ncell) {
                                    dst[q * 3 + 0] = (uint8_t)px[q];
                                    dst[q * 3 + 1] = (uint8_t)(px[q] >> 8);
                                    dst[q * 3 + 2] = (uint8_t)(px[q] >> 16);
                                }
                        }
                    }
                }
                }
            }
            if (auto_mode) {                                  // (done = t >= horizon: the env's next episode starts in this launch)
                if (is_step && p.horizon > 0 && t >= (uint32_t)p.horizon) { in_reset = true; continue; }
                break;
            }
            if (!roll) break;
            if (in_reset) { in_reset = false; continue; }     // the step this reset was due before comes next
            if (++k_step >= p.n_steps) break;
            slot = slot + 1 == (uint32_t)p.ring ? 0u : slot + 1;
            slot_en_run = slot == 0u ? 0 : slot_en_run + en_stride;
            if (to_reset >= 0) to_reset = (to_reset == 0 ? p.reset_every : to_reset) - 1;
            in_reset = to_reset == 0;
        }
        if (roll) write_state(0u);
        // (the launch that follows in the chain starts when this one has ended: every store of this wave has landed by then.
        // Measured alternative: the state left dirty in L2 and written back by an agent-scope release on the packet instead --
        // no wait here -- 5.48 against 5.45 us per step.  Without that release the results are wrong even while every env
        // stays on its XCD: an sc1 load does not return what an earlier launch left dirty in the same L2.)
#ifndef SSD_EXP_NO_END_WAIT
        if constexpr (COH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
    SSD_STAMP(9);       // observations issued
    SSD_STAMP_RT(11);
}

// MapEnv.map_to_colors() on the whole grids of envs e0 .. e0+gridDim.y-1 (map_env.py:316-339): one thread per cell,
// blockIdx.y = env; frame k of the launch goes to rgb[k][H][W][3].  Each lane writes its 3 bytes; a wave covers 192
// contiguous bytes, so the stores coalesce.
__global__ void ssd_render_full_kernel(const Params p, int e0, uint8_t *rgb) {
    const int hw = p.H * p.W;
    const int e = e0 + (int)blockIdx.y;
    uint8_t *out = rgb + (size_t)blockIdx.y * hw * 3;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < hw; c += gridDim.x * blockDim.x) {
        const int g = (c / p.W) * p.WP + c % p.W;                                    // dense cell -> padded-row grid index
        uint32_t ch = p.world[(size_t)e * p.S + g];
        for (int i = 0; i < p.N; ++i)                                                // agents, index order (:289-297)
            if ((p.agents[(size_t)e * p.N + i] & 0xFFFFu) == (uint32_t)g) ch = agent_glyph((uint32_t)i);
        if (p.keep_beams) { const uint32_t b = p.beam[(size_t)e * p.S + g]; if (b) ch = b; }   // :299-300
        const uint32_t px = p.lut[ch & 127u];
        out[c * 3 + 0] = (uint8_t)px; out[c * 3 + 1] = (uint8_t)(px >> 8); out[c * 3 + 2] = (uint8_t)(px >> 16);
    }
}

// Host-side handle (the __global__ stub) of one instantiation: what hipLaunchKernel takes, and what names the kernel's
// descriptor in the code object for the library's own AQL dispatches (ssd_aql.hip).
template <int GAME, int MODE, bool F32, int NA, bool STD, int FAST, bool COH = false, bool ACTS = false>
static const void *kernel_fn() {
    return reinterpret_cast<const void *>(&ssd_env_kernel<GAME, MODE, F32, NA, STD, FAST, COH, ACTS>);
}

template <int GAME, bool F32, int NA, bool STD, int FAST>
static const void *select_step(const Params &p) {
    if (p.mode == kModeRollout) {
        if constexpr (!F32) {
            if (p.action_ring > 0) return kernel_fn<GAME, kModeRollout, false, NA, STD, FAST, false, true>();
            return kernel_fn<GAME, kModeRollout, false, NA, STD, FAST>();
        }
        return nullptr;
    } else if (p.mode == kModeStepAuto) {
        if constexpr (!F32) return kernel_fn<GAME, kModeStepAuto, false, NA, STD, FAST>();
        return nullptr;
    } else {
        if constexpr (!F32 && FAST != 0) {           // (the kernels rollouts of the known maps use)
            if (p.coherent) return kernel_fn<GAME, kModeStep, false, NA, STD, FAST, true>();
        }
        return kernel_fn<GAME, kModeStep, F32, NA, STD, FAST>();
    }
}

// Which FAST kernel (0 = none) a step launch with these parameters gets: the map and call form must match kFastMap, and the
// (game, agents, profile) combination must be one that is instantiated below.
int fast_profile(const Params &p, int game) {
    const bool std_view = p.view_len == 7 && p.beam_len == 5;
    int fast = 0;
    for (int f = 1; f <= 2; ++f) {
        const FastMap &m = kFastMap[game][f];
        if (std_view && !p.order && !p.keep_beams && p.H == m.H && p.W == m.W && p.WP == m.WP && p.S == m.S && p.A0 == m.A0 &&
            p.A1 == m.A1 && p.n_apple == m.n_apple && (game == 0 || p.n_waste == m.n_waste))
            fast = f;
    }
    if (p.N != 5 && p.N != 10) fast = 0;
    if (fast == 2 && ((p.N == 5 && game != 0) || (p.N == 10 && game != 1))) fast = 0;
    return fast;
}

template <int GAME, bool F32>
static const void *select_game(const Params &p) {
    if (p.mode == kModeStep || p.mode == kModeRollout || p.mode == kModeStepAuto) {
        // specialised step kernels for the reference's configurations (view 7, beam 5; 5 or 10 agents), and
        // among those the FAST ones for the game's shipped map called in the plain way
        const bool std_view = p.view_len == 7 && p.beam_len == 5;
        const int fast = fast_profile(p, GAME);
        if (std_view && p.N == 5) {
            if (fast == 1) return select_step<GAME, F32, 5, true, 1>(p);
            if (GAME == 0 && fast == 2) { if constexpr (GAME == 0) return select_step<GAME, F32, 5, true, 2>(p); }
            return select_step<GAME, F32, 5, true, 0>(p);
        } else if (std_view && p.N == 10) {
            if (fast == 1) return select_step<GAME, F32, 10, true, 1>(p);
            if (GAME == 1 && fast == 2) { if constexpr (GAME == 1) return select_step<GAME, F32, 10, true, 2>(p); }
            return select_step<GAME, F32, 10, true, 0>(p);
        } else if (std_view) return select_step<GAME, F32, 0, true, 0>(p);
        return select_step<GAME, F32, 0, false, 0>(p);
    } else if (p.mode == kModeReset) {
        if constexpr (!F32) { if (p.coherent) return kernel_fn<GAME, kModeReset, false, 0, false, 0, true>(); }
        return kernel_fn<GAME, kModeReset, F32, 0, false, 0>();
    }
    return kernel_fn<GAME, kModeObserve, F32, 0, false, 0>();
}

// Everything one launch of the fused kernel needs: which instantiation, its geometry, and the kernel arguments laid out as the
// kernel's kernarg segment (KernArgs: 56 bytes of leading arguments the command processor preloads into SGPRs, then Params).
bool select(const Params &p_in, int game, Launch *out) {
    Params p = p_in;
    static const int forced_wt = SSD_HOOK("SSD_OBS_WT", -1);   // (test-hook build: tuning override)
    // per launch (rollouts run two launches at a time); float32 observations are 4x the bytes: a quarter of the envs
    p.obs_wt = forced_wt >= 0 ? forced_wt : ((p.E - p.e_begin) <= (p.obs_f32 ? 4096 : 16384) ? 1 : 0);
    if (p.coherent) p.obs_wt = 1;                   // (a coherent launch leaves nothing dirty in L2)
    static const int forced_nt = SSD_HOOK("SSD_OBS_NT", -1);   // (test-hook build: the non-temporal form whatever the ring's size)
    if (forced_nt >= 0) p.obs_nt = forced_nt;
    // An output ring beyond the memory-side cache.  uint8 observations: non-temporal write-BACK stores (3) -- the 12-byte pieces and
    // the partly covered sectors at the ends of the agents' 675-byte blocks merge in L2 and leave it as whole lines, past the
    // memory-side cache (ring 32: 5.79 us per step; write-through + non-temporal 6.86; ordinary write-back 7.04).  What a launch
    // leaves dirty in the L2s is written back by the call's closing system-scope release -- and, so that no byte of a slot can still
    // sit dirty in one XCD's L2 when another XCD writes it again a ring later, by an agent-scope release on one launch per round
    // of the ring (ssd_capi.hip; HIP-launched plain kernels release after every launch anyway).  float32 observations are whole
    // aligned lines per instruction already: write-through + non-temporal (2) as before.
    if (p.obs_nt) p.obs_wt = p.obs_f32 ? (p.obs_wt == 1 ? 2 : p.obs_wt) : 3;
#ifdef SSD_EXP_OBS_WB                               // (experiment: such a ring with another store policy)
    if (p.obs_nt) p.obs_wt = SSD_EXP_OBS_WB;
#endif
#ifdef SSD_EXP_OBS_WT_ALL                           // (experiment, unsafe: EVERY coherent launch's observation stores with this policy)
    if (p.coherent) p.obs_wt = SSD_EXP_OBS_WT_ALL;
#endif
    static const int forced_epb = SSD_KNOB("SSD_ENVS_PER_BLOCK", 0);
    const bool f32 = p.obs && p.obs_f32;            // the float32-observation variant is a separate instantiation
    int epb = envs_per_block(p, f32);
    static const int split_epb = SSD_HOOK("SSD_SPLIT_EPB", 4);   // (test-hook build: tuning override)
    // split rollouts run twice the waves per launch: smaller workgroups (measured: 4 envs per workgroup 5.23, 8: 5.78 us per step)
    if (p.snap_mode && forced_epb_early() <= 0 && epb > split_epb && split_epb >= 1) epb = split_epb;
    // (raising the issue priority of the env waves over the renderer waves changes nothing: 5.23 - 5.36 us)
    // test knob of the coherent chains (ssd_capi.hip, SSD_AQL_ALTERNATE): half the envs per workgroup, i.e. another env ->
    // workgroup -> XCD mapping than the launches before and after
    if (p.coherent == 2 && epb > 1 && forced_epb <= 0) epb /= 2;
    out->grid_x = (uint32_t)((p.E - p.e_begin + epb - 1) / epb);
    if (p.snap_mode & 2) {                          // renderer workgroups behind the env workgroups (4: no env workgroups at all)
        p.blocks_a = (p.snap_mode & 4) ? 0 : (int32_t)out->grid_x;
        out->grid_x = (uint32_t)p.blocks_a + out->grid_x;
    }
    out->block_x = (uint32_t)(64 * epb);
    out->lds = (uint32_t)lds_bytes(p, epb, f32);
    if (game == 0) out->fn = f32 ? select_game<0, true>(p) : select_game<0, false>(p);
    else out->fn = f32 ? select_game<1, true>(p) : select_game<1, false>(p);
    KernArgs &k = out->args;
    k.hdr = p.hdr; k.agents = p.agents; k.world = p.world; k.E = p.E; k.e_begin = p.e_begin; k.epb = epb; k.n_apple = p.n_apple;
    k.apple_cells = p.apple_cells; k.lut = p.lut; k.p = p;
    return out->fn != nullptr;
}

void launch(const Launch &L, void *stream) {
    KernArgs &k = const_cast<KernArgs &>(L.args);
    void *args[] = {&k.hdr, &k.agents, &k.world, &k.E, &k.e_begin, &k.epb, &k.n_apple, &k.apple_cells, &k.lut, &k.p};
    (void)hipLaunchKernel(L.fn, dim3(L.grid_x), dim3(L.block_x), args, L.lds, static_cast<hipStream_t>(stream));
}

void launch(const Params &p, int game, void *stream) {
    Launch L;
    if (select(p, game, &L)) launch(L, stream);
}

// Used by the library's own AQL queues (ssd_aql.hip): bump a counter in device memory once everything before this
// dispatch in its queue has completed (the dispatch carries the barrier bit): what the caller's HIP stream waits for ...
__global__ void ssd_flag_kernel(unsigned long long *counter) {
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// ... with this one-wave kernel, launched on that stream: it sleeps and polls (every ~0.4 us, one L2-served load) until the
// counter has reached `target`, i.e. until every chain of the rollout has finished.  A waiting kernel instead of
// hipStreamWaitValue64: the command processor polling host memory for the stream slowed the dispatch queues it shares the
// micro-engine with (6.7 against 6.1 us per 4096-env step).  The wait is BOUNDED: `abort` (host memory, set when the HSA runtime
// reports a queue error) ends it -- the work it waits for will then never come --, and so does `timeout_ticks` of the 100 MHz
// constant clock (the caller sizes it from the call: seconds, far beyond any healthy rollout).  A kernel that waits for another
// queue's kernel never ends under a tool that runs kernels one at a time (rocprofv3 --pmc): the library then uses host waits
// (ssd_capi.hip, sync mode, chosen automatically when a tool is attached); should it still get here, the wave gives up, sets
// SSD_ST_WAIT_TIMEOUT in the handle's status word -- the call's results are not in place -- and lets the stream go on.
__global__ void ssd_wait_counter_kernel(const unsigned long long *counter, unsigned long long target, const volatile uint32_t *abort,
                                        unsigned long long timeout_ticks, uint32_t *status, uint32_t *timed_out) {
    if (threadIdx.x != 0) return;
    uint32_t spins = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(16);
        if ((++spins & 1023u) == 0) {
            if (abort && *abort) break;
            if (timeout_ticks && __builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                if (status) atomicOr(status, kStWaitTimeout);
                // (host memory: the handle's next API call sees it without touching the device -- ssd_capi.hip, after_timeout)
                if (timed_out) __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// ... and the other direction: a kernel on the caller's HIP stream that releases the library's queues (they wait, in a
// barrier-AND packet, for this HSA signal's value to become 0) once the stream's earlier work is done.
__global__ void ssd_signal_kernel(long long *signal_value) {
    if (threadIdx.x == 0) __hip_atomic_store(signal_value, 0ll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#ifdef SSD_STAMPS
// Diagnostic library only: publishes the 100 MHz clock the stamps use to a host-visible word, over and over, so that a tool
// can place the stamps on the host's time axis (tools/call_timeline.py).
__global__ void ssd_clock_kernel(unsigned long long *out, int iters) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < iters; ++i) {
        __hip_atomic_store(out, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_s_sleep(4);
    }
}
void launch_clock_kernel(unsigned long long *out, int iters, void *stream) {
    hipLaunchKernelGGL(ssd_clock_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out, iters);
}
#endif
const void *flag_kernel_fn() { return reinterpret_cast<const void *>(&ssd_flag_kernel); }
const void *wait_kernel_fn() { return reinterpret_cast<const void *>(&ssd_wait_counter_kernel); }
void launch_flag_kernel(unsigned long long *counter, void *stream) {
    hipLaunchKernelGGL(ssd_flag_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), counter);
}
void launch_wait_counter_kernel(const unsigned long long *counter, unsigned long long target, const uint32_t *abort,
                                unsigned long long timeout_ticks, uint32_t *status, uint32_t *timed_out, void *stream) {
    hipLaunchKernelGGL(ssd_wait_counter_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), counter, target, abort,
                       timeout_ticks, status, timed_out);
}
void launch_signal_kernel(long long *signal_value, void *stream) {
    hipLaunchKernelGGL(ssd_signal_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), signal_value);
}

// The return_agent_actions extras for the batch (map_env.py:201-205, :242-246, :749-770): one thread per output element
// (e, i, j): the j-th other agent of agent i in string-sorted id order is sorted[j + (j >= rank[i])].
__global__ void ssd_agent_action_obs_kernel(const int32_t *actions, const uint8_t *done_mask, long long *other_actions, long long *visible,
                                            AgentOrder order, int E, int N) {
    const long long total = (long long)E * N * (N - 1);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % (N - 1));
        const long long ei = idx / (N - 1);
        const int i = (int)(ei % N);
        const long long e = ei / N;
        if (other_actions) {
            long long v = 0;
            if (actions && !(done_mask && done_mask[ei])) v = actions[e * N + order.sorted[j + (j >= (int)order.rank[i] ? 1 : 0)]];
            other_actions[idx] = v;
        }
        if (visible) visible[idx] = 1;
    }
}
void launch_agent_action_obs(const int32_t *actions, const uint8_t *done_mask, long long *other_actions, long long *visible,
                             const AgentOrder &order, int E, int N, void *stream) {
    const long long total = (long long)E * N * (N - 1);
    if (total <= 0) return;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(ssd_agent_action_obs_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), actions, done_mask,
                       other_actions, visible, order, E, N);
}

void launch_render_full(const Params &p, int e0, int count, uint8_t *rgb_dev, void *stream) {
    const int hw = p.H * p.W;
    hipLaunchKernelGGL(ssd_render_full_kernel, dim3((hw + 255) / 256, count), dim3(256), 0, static_cast<hipStream_t>(stream), p, e0, rgb_dev);
}

}  // namespace ssd
