// csrc/ssd_kernels.hip -- the MapEnv.step() hot path as one fused gfx950 kernel.
//
// Mapping (CDNA4: 64-wide wavefronts, 160 KiB LDS per CU, 256 CUs):
//   * one wavefront owns one env for the whole step; a 256-thread workgroup holds 4 envs;
//   * the env's grid (16x38 = 608 B for Harvest) is pulled from HBM once with 16 B/lane loads
//     into LDS, every phase works on the LDS copy, and it is written back once;
//   * move / rotate / conflict resolution: lanes = agents, positions compared with wavefront
//     ballots ("who stands on cell x" = ballot, last index = highest set bit), the order-dependent
//     parts of the reference algorithm run as wave-uniform loops (map_env.py:357-543);
//   * beams: lanes = (ray, step) pairs, stop positions from ballots (map_env.py:566-649);
//   * respawn: lanes = cells, 3x3 stencil on the LDS grid, counter-based PRNG keyed on the cell
//     (harvest.py:75-104, cleanup.py:132-171);
//   * observation: after a workgroup barrier all 256 lanes render the 4 envs' N x 15 x 15 x 3
//     windows from the LDS overlay, 4 cells = 12 contiguous bytes per lane per store, so a
//     wavefront store covers 768 contiguous bytes of the uint8 obs tensor (map_env.py:189-199).
// No MFMA: the path is integer / indexing work bounded by HBM traffic.
//
// Reference citations are file:line of the reference repository (social_dilemmas/envs/...).
#include <hip/hip_runtime.h>

#include "ssd_internal.hpp"

namespace ssd {

// SSD_ST_* (include/ssd.h)
constexpr uint32_t kStBadAction = 1u << 0;
constexpr uint32_t kStNoSpawn = 1u << 1;
constexpr uint32_t kStMoveLookup = 1u << 2;

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
__device__ __forceinline__ uint32_t draw(uint32_t pkey, uint32_t index) { return mix32(pkey ^ index); }
__device__ __forceinline__ uint32_t randint(uint32_t u, uint32_t n) { return __umulhi(u, n); }

// ---------------------------------------------------------------------------------------------
// wavefront helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint64_t bit(uint32_t i) { return 1ull << i; }

// Lanes of one wavefront communicate through LDS without a workgroup barrier (LDS operations of a
// wave complete in order); this only has to stop the compiler from moving LDS accesses across it.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { uint32_t w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return rfl(v);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return rfl(v);
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o, 64), hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
        uint64_t w = ((uint64_t)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v);
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) { return ~wave_min_u64(~v); }

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

// bytes of x that are non-zero -> 0xFF, others 0x00
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) {
    uint32_t m = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    return (m >> 7) * 0xFFu;
}

// ---------------------------------------------------------------------------------------------
// LDS layout of one workgroup
//   lut[128] u32 | agent[4][64] u32 | flag[4] u32 | cellinfo[S] | per wave: world[S] beam[S] occ[S]
// ---------------------------------------------------------------------------------------------
__host__ size_t lds_bytes(int S) { return 128 * 4 + kEnvsPerBlock * 64 * 4 + 16 + (size_t)S + (size_t)kEnvsPerBlock * 3 * S; }

template <int GAME>
__global__ __launch_bounds__(256) void ssd_env_kernel(const Params p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int S = p.S, N = p.N, W = p.W, H = p.H;
    uint32_t *s_lut = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_agent = s_lut + 128;
    uint32_t *s_flag = s_agent + kEnvsPerBlock * 64;
    uint8_t *s_info = reinterpret_cast<uint8_t *>(s_flag + 4);
    uint8_t *s_grids = s_info + S;
    uint8_t *s_world = s_grids + (size_t)wv * 3 * S;
    uint8_t *s_beam = s_world + S;
    uint8_t *s_occ = s_beam + S;

    // static tables: colour LUT and per-cell spawn info, shared by the 4 envs of the workgroup
    if (tid < 128) s_lut[tid] = p.lut[tid];
    for (int i = tid * 16; i < S; i += 256 * 16)
        *reinterpret_cast<uint4 *>(s_info + i) = *reinterpret_cast<const uint4 *>(p.cellinfo + i);

    const int e = blockIdx.x * kEnvsPerBlock + wv;
    const int mode = p.mode;
    bool active = e < p.E;                                   // wave-uniform
    if (active && mode == kModeReset && p.mask) active = p.mask[e] != 0;
    if (lane == 0) s_flag[wv] = active ? 1u : 0u;
    __syncthreads();                                         // s_info / s_lut visible to all waves

    if (active) {
        const bool is_agent = lane < N;
        const uint4 hdr = p.hdr[e];
        uint32_t key = rfl(hdr.x), t = rfl(hdr.y), episode = rfl(hdr.z);
        uint32_t status = 0;
        uint32_t cell = 0, orient = 2;                       // per-lane agent state (lane = agent index)
        int rew = 0;

        if (mode == kModeReset) {
            // ---- MapEnv.reset (map_env.py:214-249) ----
            episode += 1; t = 0;
            key = env_key(p.seed_lo, p.seed_hi, p.env_base + (uint32_t)e, episode);
            for (int i = lane * 16; i < S; i += 64 * 16) {   // reset_map (:560-564) + custom_reset
                *reinterpret_cast<uint4 *>(s_world + i) = *reinterpret_cast<const uint4 *>(p.reset_world + i);
                *reinterpret_cast<uint4 *>(s_beam + i) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4 *>(s_occ + i) = make_uint4(0, 0, 0, 0);
            }
            wave_sync();
            // setup_agents (harvest.py:46-55 / cleanup.py:118-130): spawn_point (map_env.py:651-662) takes
            // the LAST free point of a fresh shuffle = the free point with the largest (draw, cell);
            // spawn_rotation (:664-667) indexes [LEFT, RIGHT, UP, DOWN].
            const uint32_t pk_pt = phase_key(key, 0, kSpawnPoint), pk_rot = phase_key(key, 0, kSpawnRot);
            for (int i = 0; i < N; ++i) {
                uint64_t best = 0;
                for (int s = lane; s < p.n_spawn; s += 64) {
                    const uint32_t c = p.spawn_cells[s];
                    if (s_occ[c] == 0) {
                        const uint64_t k = (((uint64_t)draw(pk_pt, ((uint32_t)i << 16) | c) << 32) | c) + 1;
                        best = k > best ? k : best;
                    }
                }
                best = wave_max_u64(best);
                uint32_t chosen = 0;
                if (best == 0) { status |= kStNoSpawn; chosen = p.n_spawn ? p.spawn_cells[0] : (uint32_t)(W + 1); }
                else chosen = (uint32_t)(best - 1) & 0xFFFFu;
                if (lane == i) { cell = chosen; orient = randint(draw(pk_rot, (uint32_t)i), 4); }
                s_occ[chosen] = agent_glyph((uint32_t)i);    // all lanes, same address, same value
                wave_sync();
            }
        } else {
            // ---- load env state: grid -> LDS (16 B per lane), agents -> lanes ----
            const uint8_t *gw = p.world + (size_t)e * S;
            const bool load_beam = mode == kModeObserve && p.keep_beams;
            for (int i = lane * 16; i < S; i += 64 * 16) {
                *reinterpret_cast<uint4 *>(s_world + i) = *reinterpret_cast<const uint4 *>(gw + i);
                uint4 bv = make_uint4(0, 0, 0, 0);
                if (load_beam) bv = *reinterpret_cast<const uint4 *>(p.beam + (size_t)e * S + i);
                *reinterpret_cast<uint4 *>(s_beam + i) = bv;
                *reinterpret_cast<uint4 *>(s_occ + i) = make_uint4(0, 0, 0, 0);
            }
            if (is_agent) {
                const uint32_t a = p.agents[(size_t)e * N + lane];
                cell = a & 0xFFFFu; orient = (a >> 16) & 3u;
            }
            wave_sync();
        }

        int act = -1;
        uint32_t ordv = (uint32_t)lane;                      // action order list, lane k = k-th acting agent
        int nord = N;
        if (mode == kModeStep) {
            t += 1;
            // ---- actions (map_env.py:171-173) ----
            if (p.num_actions_random > 0) {                  // rollout.py:64-65 uniform random actions
                const uint32_t pk = phase_key(key, t, kAction);
                if (is_agent) {
                    act = (int)randint(draw(pk, (uint32_t)lane), (uint32_t)p.num_actions_random);
                    if (p.actions_out) p.actions_out[(size_t)e * N + lane] = act;
                }
            } else if (is_agent) {
                act = p.actions[(size_t)e * N + lane];
            }
            constexpr int kNumActions = GAME == 0 ? 8 : 9;   // harvest.py:44, cleanup.py:70
            const bool bad = is_agent && (act < -1 || act >= kNumActions);
            if (ballot(bad)) { status |= kStBadAction; if (bad) act = -1; }
            if (p.order) {
                ordv = is_agent ? (uint32_t)p.order[(size_t)e * N + lane] : 0xFFu;
                if (ordv != 0xFFu && ordv >= (uint32_t)N) { ordv = 0xFFu; status |= kStBadAction; }
                const uint64_t endm = ballot(ordv == 0xFFu);
                nord = endm ? __builtin_ctzll(endm) : 64;
                uint64_t acting = 0;
                for (int k = 0; k < nord; ++k) acting |= bit(rl(ordv, k));
                if (!((acting >> lane) & 1)) act = -1;       // agents absent from the action dict do nothing
            }

            // ---- update_moves (map_env.py:357-543) ----
            const bool mover = is_agent && act >= 0 && act <= 4;              // :383
            if (is_agent && (act == 5 || act == 6)) orient = turn(act, orient);   // :390-392
            uint32_t tcell = cell;
            if (mover) {
                int vr, vc, dr, dc;
                unit_vec(act, vr, vc);
                // rotate_action (:701-716): UP (v) LEFT (vc,-vr) RIGHT (-vc,vr) DOWN (-v)
                dr = orient == 2 ? vr : orient == 0 ? vc : orient == 1 ? -vc : -vr;
                dc = orient == 2 ? vc : orient == 0 ? -vr : orient == 1 ? vr : -vc;
                const uint32_t cand = (uint32_t)((int)cell + dr * W + dc);
                // agent.py:105-113 return_valid_pos (the agent's grid agrees with world_map on '@')
                tcell = s_world[cand] == '@' ? cell : cand;
            }
            uint32_t mvcell = tcell;                         // agent_moves[id] (:410)
            const uint64_t M = ballot(mover);
            if (M) {                                         // :415
                const int nm = __builtin_popcountll(M);
                uint32_t perm = 0;                           // lane k: k-th entry of the (shuffled) zipped list
                {
                    int cnt = 0;
                    for (int k = 0; k < nord; ++k) {
                        const uint32_t a = rl(ordv, k);
                        if ((M >> a) & 1) { if (lane == cnt) perm = a; ++cnt; }
                    }
                }
                const uint32_t pk = phase_key(key, t, kMove);
                for (int i = nm - 1; i >= 1; --i) {          // :421-423 np.random.shuffle = Fisher-Yates from the end
                    const uint32_t j = randint(draw(pk, (uint32_t)i), (uint32_t)i + 1);
                    const uint32_t vi = rl(perm, i), vj = rl(perm, j);
                    if (lane == i) perm = vj;
                    if (lane == (int)j) perm = vi;
                }
                uint64_t Hm = M;                             // ids that still have an entry in agent_moves
                // :424-491 cells wanted by several agents, in lexicographic order (np.unique, axis=0)
                int cur = -1;
                while (true) {
                    const uint32_t nxt = wave_min_u32((mover && (int)tcell > cur) ? tcell : 0xFFFFFFFFu);
                    if (nxt == 0xFFFFFFFFu) break;
                    cur = (int)nxt;
                    const uint64_t Cm = ballot(mover && tcell == nxt);
                    if (__builtin_popcountll(Cm) < 2) continue;                     // :436
                    bool cell_free = true;
                    const uint64_t Pm = ballot(is_agent && cell == nxt);            // :449 move in self.agent_pos
                    if (Pm) {
                        const uint32_t occ = 63 - __builtin_clzll(Pm);              // agent_by_pos: last index wins
                        const uint32_t occ_mv = rl(mvcell, occ);
                        if ((Cm >> occ) & 1) cell_free = false;                     // (1) :460
                        else if (!((Hm >> occ) & 1) || occ_mv == nxt) cell_free = false;    // (2) :466-468
                        else if (ballot(((Cm >> lane) & 1) && cell == occ_mv)) cell_free = false; // (3) :472-476
                    }
                    if (cell_free) {                         // :480-483 first contender in shuffled order moves NOW
                        uint32_t w = 0;
                        for (int k = 0; k < nm; ++k) { w = rl(perm, k); if ((Cm >> w) & 1) break; }
                        if (lane == (int)w) cell = nxt;
                    }
                    if ((Cm >> lane) & 1) mvcell = cell;     // :486-491 every contender's move becomes "stay"
                }
                // :494-543 remaining moves: chains, swaps, cycles
                while (Hm) {
                    const uint32_t snap_cell = cell, snap_mv = mvcell;              // agent_by_pos (:495), moves_copy (:498)
                    const uint64_t snapH = Hm;
                    uint64_t del = 0;
                    const int n0 = __builtin_popcountll(Hm);
                    for (int k = 0; k < nord; ++k) {                                // agent_moves insertion order = action order
                        const uint32_t a = rl(ordv, k);
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
                        if ((Hm >> lane) & 1) cell = mvcell;
                        break;
                    }
                }
            }
        }

        if (mode != kModeReset) {
            // ---- consume (map_env.py:178-181, agent.py:177-183) in index order + occupancy layer ----
            // Every lane runs the same LDS ops on the same address, so the loop is sequential per lane
            // and needs no cross-lane ordering.  s_occ holds the glyph of the LAST agent on a cell.
            for (int i = 0; i < N; ++i) {
                const uint32_t ci = rl(cell, i);
                if (mode == kModeStep && s_world[ci] == 'A') {
                    s_world[ci] = ' ';
                    if (lane == i) rew += 1;
                }
                s_occ[ci] = agent_glyph((uint32_t)i);
            }
            wave_sync();
        }

        if (mode == kModeStep) {
            // ---- update_custom_moves (map_env.py:545-552): beams in action order ----
            const int L = p.beam_len;
            const uint32_t rmask = (1u << L) - 1u;
            for (int k = 0; k < nord; ++k) {
                const uint32_t a = rl(ordv, k);
                const int aa = (int)rl((uint32_t)act, a);
                const bool fire = aa == 7, clean = GAME == 1 && aa == 8;            // harvest.py:62-67, cleanup.py:94-111
                if (!fire && !clean) continue;
                if (fire && lane == (int)a) rew -= 1;                               // agent.py:170-172 fire_beam('F')
                // update_map_fire (map_env.py:566-649): lane = (ray q, step kk)
                const uint32_t pc = rl(cell, a);
                const int pr = (int)__umulhi(pc, p.w_magic), pcc = (int)pc - pr * W;
                int dr, dc;
                unit_vec((int)rl(orient, a), dr, dc);
                const int rr = -dc, rc = dr;                                        // rotate_right(d) (:607)
                const int q = (lane >= L) + (lane >= 2 * L), kk = lane - q * L;
                const bool inray = lane < 3 * L;
                const int sr = pr + (q == 1 ? rr - dr : q == 2 ? -rr - dr : 0);     // :608-609 start positions
                const int sc = pcc + (q == 1 ? rc - dc : q == 2 ? -rc - dc : 0);
                const int r2 = sr + dr * (kk + 1), c2 = sc + dc * (kk + 1);
                const bool inb = inray && r2 >= 0 && r2 < H && c2 >= 0 && c2 < W;   // :615 test_if_in_bounds
                const int cidx = inb ? r2 * W + c2 : 0;
                const uint8_t wch = inb ? s_world[cidx] : (uint8_t)'@';
                const uint8_t och = inb ? s_occ[cidx] : (uint8_t)0;
                const bool pass = inb && wch != '@';                                // :616
                const bool stopper = pass && (och != 0 || (clean && wch == 'H'));   // :621 agents absorb, :639 blocking cell
                const uint64_t mf = ballot(inray && !pass), ms = ballot(stopper);
                const uint32_t f = (uint32_t)(mf >> (q * L)) & rmask, s = (uint32_t)(ms >> (q * L)) & rmask;
                const int ff = f ? __builtin_ctz(f) : L, fs = s ? __builtin_ctz(s) : L;
                const int len = fs < ff ? fs + 1 : ff;                              // beam covers the stopping cell
                wave_sync();
                if (inray && kk < len) {
                    s_beam[cidx] = clean ? 'C' : 'F';                               // :624,:636 firing_points
                    if (clean && wch == 'H') s_world[cidx] = 'R';                   // :625-634 cell_types ['H'] -> ['R']
                }
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

        if (mode != kModeObserve) {
            // ---- custom_map_update (map_env.py:187 / :230): respawn ----
            uint64_t spawn_bits = 0;                                                // bit j: cell lane + 64*j gets an apple
            const uint32_t pk_apple = phase_key(key, t, kApple);
            if (GAME == 0) {
                // harvest.py:75-104 spawn_apples.  Apple points are interior cells (the border is wall),
                // so the 3x3 neighbourhood (j*j + k*k <= 2 on the radius-2 box, :90-92) is always in bounds.
                int j = 0;
                for (int c = lane; c < S; c += 64, ++j) {
                    if ((s_info[c] & kInfoApple) && s_world[c] != 'A' && s_occ[c] == 0) {      // :88
                        int n = 0;
#pragma unroll
                        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
                            for (int dc = -1; dc <= 1; ++dc) n += s_world[c + dr * W + dc] == 'A';
                        const uint64_t thr = n == 0 ? p.thr_h[0] : n == 1 ? p.thr_h[1] : n == 2 ? p.thr_h[2] : p.thr_h[3];
                        if ((uint64_t)draw(pk_apple, (uint32_t)c) < thr) spawn_bits |= bit(j);  // :100-103
                    }
                }
                wave_sync();                                                        // counts use the pre-spawn map (:73)
                j = 0;
                for (int c = lane; c < S; c += 64, ++j)
                    if ((spawn_bits >> j) & 1) s_world[c] = 'A';
            } else {
                // cleanup.py:113-116: compute_probabilities (:156-171) from the current waste count, then
                // spawn_apples_and_waste (:132-154).  Thresholds come from a host-computed table.
                uint32_t nh = 0;
                for (int c = lane; c < S; c += 64) nh += s_world[c] == 'H';
                nh = wave_sum_u32(nh);
                nh = nh < (uint32_t)p.n_thr ? nh : (uint32_t)p.n_thr - 1;
                const uint64_t thr_a = p.thr_ca[nh], thr_w = p.thr_cw[nh];
                uint64_t best = ~0ull;
                const uint32_t pk_coin = phase_key(key, t, kWasteCoin), pk_ord = phase_key(key, t, kWasteOrder);
                int j = 0;
                for (int c = lane; c < S; c += 64, ++j) {
                    const uint8_t inf = s_info[c], w = s_world[c];
                    if ((inf & kInfoApple) && w != 'A' && s_occ[c] == 0 &&
                        (uint64_t)draw(pk_apple, (uint32_t)c) < thr_a) spawn_bits |= bit(j);   // :135-141
                    // :144-153 shuffled scan, first non-'H' point whose coin succeeds (at most one per step)
                    if (thr_w && (inf & kInfoWaste) && w != 'H' && (uint64_t)draw(pk_coin, (uint32_t)c) < thr_w) {
                        const uint64_t kx = ((uint64_t)draw(pk_ord, (uint32_t)c) << 32) | (uint32_t)c;
                        best = kx < best ? kx : best;
                    }
                }
                if (thr_w) best = wave_min_u64(best);
                wave_sync();
                j = 0;
                for (int c = lane; c < S; c += 64, ++j)
                    if ((spawn_bits >> j) & 1) s_world[c] = 'A';
                if (best != ~0ull) s_world[(uint32_t)best] = 'H';                   // may land under an agent
            }
            wave_sync();

            // ---- write the env back: grid, agents, header, rewards, dones ----
            uint8_t *gw = p.world + (size_t)e * S;
            for (int i = lane * 16; i < S; i += 64 * 16) {
                *reinterpret_cast<uint4 *>(gw + i) = *reinterpret_cast<const uint4 *>(s_world + i);
                if (p.keep_beams)
                    *reinterpret_cast<uint4 *>(p.beam + (size_t)e * S + i) = *reinterpret_cast<const uint4 *>(s_beam + i);
            }
            if (is_agent) {
                p.agents[(size_t)e * N + lane] = cell | (orient << 16);
                if (mode == kModeStep) {
                    if (p.rew) p.rew[(size_t)e * N + lane] = rew;                   // compute_reward (:208)
                    if (p.done) p.done[(size_t)e * N + lane] = 0;                   // get_done -> False (:209)
                }
            }
            if (lane == 0) p.hdr[e] = make_uint4(key, t, episode, 0);
            if (status && lane == 0) atomicOr(p.status, status);
            wave_sync();
        }

        // ---- get_map_with_agents (map_env.py:280-302): world <- agents <- beams, in place, 4 cells per op ----
        for (int i = lane * 4; i < S; i += 64 * 4) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(s_world + i);
            const uint32_t o = *reinterpret_cast<const uint32_t *>(s_occ + i);
            const uint32_t b = *reinterpret_cast<const uint32_t *>(s_beam + i);
            const uint32_t mo = nonzero_bytes(o), mb = nonzero_bytes(b);
            uint32_t v = (w & ~mo) | (o & mo);
            v = (v & ~mb) | (b & mb);
            *reinterpret_cast<uint32_t *>(s_world + i) = v;
        }
        if (is_agent) {
            const uint32_t r = __umulhi(cell, p.w_magic), c = cell - r * (uint32_t)W;
            s_agent[wv * 64 + lane] = r | (c << 12) | (orient << 24);
        }
    }
    __syncthreads();

    // ---- per-agent observations (agent.py:76-78 -> utility_funcs.py:59-114 window with '0' padding,
    //      map_env.py:316-339 colour LUT, :669-689 rotate_view), all 256 lanes over the 4 envs ----
    if (!p.obs) return;
    {
        const int V = p.V, v = p.view_len, VV = V * V, per_env = N * VV;
        const int env0 = blockIdx.x * kEnvsPerBlock;
        const int nenv = min(kEnvsPerBlock, p.E - env0);
        const int total = nenv * per_env;
        uint8_t *out = p.obs + (size_t)env0 * per_env * 3;
        const bool rotate = p.rotate != 0;
        for (int g = tid; g * 4 < total; g += 256) {
            const int f0 = g * 4;
            int el = (int)__umulhi((uint32_t)f0, p.per_env_magic);
            int rem = f0 - el * per_env;
            int ag = (int)__umulhi((uint32_t)rem, p.vv_magic);
            rem -= ag * VV;
            int i = (int)__umulhi((uint32_t)rem, p.v_magic);
            int j = rem - i * V;
            uint32_t px[4];
            uint32_t valid = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                px[q] = 0;
                if (f0 + q < total && s_flag[el]) {
                    valid |= 1u << q;
                    const uint32_t info = s_agent[el * 64 + ag];
                    const int r0 = info & 0xFFF, c0 = (info >> 12) & 0xFFF, o = (info >> 24) & 3;
                    // np.rot90 count k: UP 0, LEFT 1, DOWN 2, RIGHT 3; out[i,j] = view[a,b]
                    int a = i, b = j;
                    if (rotate) {
                        if (o == 0) { a = j; b = V - 1 - i; }
                        else if (o == 3) { a = V - 1 - i; b = V - 1 - j; }
                        else if (o == 1) { a = V - 1 - j; b = i; }
                    }
                    const int rr = r0 - v + a, cc = c0 - v + b;
                    const uint8_t *grid = s_grids + (size_t)el * 3 * S;
                    const uint32_t ch = (rr >= 0 && rr < H && cc >= 0 && cc < W) ? grid[rr * W + cc] : (uint32_t)'0';
                    px[q] = s_lut[ch & 127u];
                }
                if (++j == V) { j = 0; if (++i == V) { i = 0; if (++ag == N) { ag = 0; ++el; } } }
            }
            uint8_t *dst = out + (size_t)f0 * 3;
            if (valid == 0xFu) {
                typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                struct __attribute__((packed, aligned(4))) P3 { u32x3 v; };
                u32x3 d;
                d.x = px[0] | (px[1] << 24);
                d.y = (px[1] >> 8) | (px[2] << 16);
                d.z = (px[2] >> 16) | (px[3] << 8);
                reinterpret_cast<P3 *>(dst)->v = d;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((valid >> q) & 1) {
                        dst[q * 3 + 0] = (uint8_t)px[q];
                        dst[q * 3 + 1] = (uint8_t)(px[q] >> 8);
                        dst[q * 3 + 2] = (uint8_t)(px[q] >> 16);
                    }
            }
        }
    }
}

// MapEnv.map_to_colors() on the whole grid of one env (map_env.py:316-339), one thread per cell.
__global__ void ssd_render_full_kernel(const Params p, int e, uint8_t *rgb) {
    const int hw = p.H * p.W;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < hw; c += gridDim.x * blockDim.x) {
        uint32_t ch = p.world[(size_t)e * p.S + c];
        for (int i = 0; i < p.N; ++i)                                                // agents, index order (:289-297)
            if ((p.agents[(size_t)e * p.N + i] & 0xFFFFu) == (uint32_t)c) ch = agent_glyph((uint32_t)i);
        if (p.keep_beams) { const uint32_t b = p.beam[(size_t)e * p.S + c]; if (b) ch = b; }   // :299-300
        const uint32_t px = p.lut[ch & 127u];
        rgb[c * 3 + 0] = (uint8_t)px; rgb[c * 3 + 1] = (uint8_t)(px >> 8); rgb[c * 3 + 2] = (uint8_t)(px >> 16);
    }
}

void launch(const Params &p, int game, void *stream) {
    const dim3 grid((p.E + kEnvsPerBlock - 1) / kEnvsPerBlock), block(256);
    const size_t lds = lds_bytes(p.S);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (game == 0) hipLaunchKernelGGL(ssd_env_kernel<0>, grid, block, lds, s, p);
    else hipLaunchKernelGGL(ssd_env_kernel<1>, grid, block, lds, s, p);
}

void launch_render_full(const Params &p, int e, uint8_t *rgb_dev, void *stream) {
    const int hw = p.H * p.W;
    hipLaunchKernelGGL(ssd_render_full_kernel, dim3((hw + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), p, e, rgb_dev);
}

}  // namespace ssd
