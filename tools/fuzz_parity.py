#!/usr/bin/env python3
"""Randomised parity fuzz (GPU box; not part of the suite): random wall-closed maps, agent counts, views, beam lengths and
env counts, stepped five ways -- call by call with random action subsets / orders, ssd_rollout_random as chains, as the
fused rollout kernel, ssd_rollout_actions (caller-supplied action and order rings; chains and fused), and with SSD_AUTO_RESET --
against the C oracle, bit for bit.  First line: the library that ran and the SSD_* settings (tools/_label.py).

    python tools/fuzz_parity.py [n_configs] [seed]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402
import golden_util as G  # noqa: E402
from oracle import pyoracle  # noqa: E402
from sequential_social_dilemma_games_amd import constants as K  # noqa: E402
from sequential_social_dilemma_games_amd.engine import VecEngine  # noqa: E402
from test_hip_parity import _random_map  # noqa: E402


def check_state(eng, ora, where):
    a, b = eng.get_state(), ora.get_state()
    for k in ("world", "pos", "orient", "episode", "t"):
        assert np.array_equal(a[k], b[k]), "%s differs (%s)" % (k, where)


def draw(rng, idx, fixed=None):
    """Everything random about configuration `idx`, drawn from `rng` in the order the tool has always consumed it (so that
    configuration k of a seed is the same configuration in every version and can be drawn without running the ones before)."""
    c = {"idx": idx}
    game = int(rng.randint(0, 2))
    big = os.environ.get("FUZZ_BIG") == "1"               # the limits of the ABI: 64 agents, 4096 cells, 31 x 31 views, beams of 21
    H, W = (int(rng.randint(4, 65)), int(rng.randint(4, 65))) if big else (int(rng.randint(4, 26)), int(rng.randint(4, 31)))
    if H * W > 4096:
        W = 4096 // H
    free = (H - 2) * (W - 2)
    N = int(rng.randint(1, min(65 if big else 14, max(2, free // 3))))
    v, L = (int(rng.randint(0, 16)), int(rng.randint(1, 22))) if big else (int(rng.randint(0, 11)), int(rng.randint(1, 9)))
    E = int(rng.randint(1, 40 if big else 70))
    amap = _random_map(rng, H, W, game, n_spawn=N + int(rng.randint(0, 4)))
    keep = bool(rng.randint(0, 2))
    seed = int(rng.randint(0, 2 ** 31))
    if fixed is not None:                                     # a recorded configuration on a freshly drawn map
        game, H, W, N, v, L, E, keep, seed = fixed
        amap = _random_map(rng, H, W, game, n_spawn=N + 2)
    c.update(game=game, H=H, W=W, N=N, v=v, L=L, E=E, amap=amap, keep=keep, seed=seed)
    c["tag"] = "cfg %d: game %d %dx%d N=%d v=%d L=%d E=%d keep=%d seed=%d" % (idx, game, H, W, N, v, L, E, keep, seed)
    # the 64 KiB LDS rule of ssd_create (ssd_capi.hip): such a configuration is skipped before anything else is drawn
    WP = W + v
    S = (H * WP + 15) & ~15
    c["too_big"] = 512 + 256 + 2048 + ((v * (WP + 1) + 15) & ~15) + ((v * WP + 15) & ~15) + 3 * S > 64 * 1024
    if c["too_big"]:
        return c
    na = 8 if game == K.GAME_HARVEST else 9
    c["steps"] = []
    for s in range(12):
        if s % 2:
            act = rng.randint(0, na, size=(E, N)).astype(np.int32)
            order = np.full((E, N), 0xFF, np.uint8)
            for e in range(E):
                k = rng.randint(0, N + 1)
                perm = rng.permutation(N)[:k]
                order[e, :k] = perm
                act[e, np.setdiff1d(np.arange(N), perm)] = -1
            c["steps"].append((act, order))
        else:
            c["steps"].append(None)
    c["rollouts"] = [(int(rng.randint(1, 14)), int(rng.randint(1, 8)), int(rng.randint(1, 5))) for _ in range(2)]
    c["horizon"] = int(rng.randint(2, 7))
    c["part"] = (rng.rand(E) < 0.5).astype(np.uint8)
    return c


def run(c, quiet=False):
    game, N, v, L, E, amap, keep, seed, tag = c["game"], c["N"], c["v"], c["L"], c["E"], c["amap"], c["keep"], c["seed"], c["tag"]
    try:
        eng = VecEngine(game, amap, num_envs=E, num_agents=N, view_len=v, beam_len=L, seed=seed, keep_beams=keep)
    except Exception as ex:                                   # e.g. LDS budget: not a parity matter
        if not c["too_big"]:
            raise
        if not quiet:
            print(tag, "-> skipped:", str(ex)[:80])
        return
    assert not c["too_big"], tag
    ora = pyoracle.Oracle(game, amap, E, N, G.default_lut(), view_len=v, beam_len=L, seed=seed)
    V = 2 * v + 1
    assert np.array_equal(eng.reset_host(), ora.reset()), tag + " reset"
    # (1) call by call, explicit subsets / orders every other step
    for s, inp in enumerate(c["steps"]):
        if inp is not None:
            act, order = inp
            obs, rew, _ = eng.step_host(act, order)
            o_obs, o_rew, _ = ora.step(act, order)
        else:
            _, obs, rew, _ = eng.step_random_host()
            _, o_obs, o_rew, _ = ora.step_random()
        assert np.array_equal(rew, o_rew) and np.array_equal(obs, o_obs), tag + " step %d" % s
    check_state(eng, ora, tag + " after steps")
    # (2) rollout as chains, (3) fused
    ring = 3
    obs = torch.zeros((ring, E, N, V, V, 3), dtype=torch.uint8, device="cuda")
    rew = torch.zeros((ring, E, N), dtype=torch.int32, device="cuda")
    done = torch.zeros((ring, E, N), dtype=torch.uint8, device="cuda")
    step0 = 0
    for fused, (n, every, chains) in zip((False, True), c["rollouts"]):
        eng.set_rollout_chains(chains)
        want = {}
        for k in range(step0, step0 + n):
            if k % every == 0:
                ora.reset()
            _, o_obs, o_rew, _ = ora.step_random()
            want[k] = (o_obs, o_rew)
        eng.rollout_random(n, obs, rew, done, reset_every=every, step0=step0, fused=fused)
        g_obs, g_rew = obs.cpu().numpy(), rew.cpu().numpy()
        for k in range(max(step0, step0 + n - ring), step0 + n):
            assert np.array_equal(g_obs[k % ring], want[k][0]) and np.array_equal(g_rew[k % ring], want[k][1]), \
                tag + " rollout fused=%d step %d" % (fused, k)
        check_state(eng, ora, tag + " after rollout fused=%d" % fused)
        step0 += n
    # (3b) ssd_rollout_actions: caller-supplied actions, and every other call the action dicts' orders (a generator of its own, keyed
    #      by the configuration's seed: what `draw` consumes stays what it always was)
    rng2 = np.random.RandomState((seed ^ 0x5A5A5A) & 0x7FFFFFFF)
    na = 8 if game == K.GAME_HARVEST else 9
    for call, fused in enumerate((False, True, False, True)):
        n2, aring = int(rng2.randint(1, 8)), int(rng2.randint(1, 6))
        n2 = min(n2, aring)                                   # (a slot is read once per call)
        eng.set_rollout_chains(int(rng2.randint(1, 4)))
        a_host = rng2.randint(0, na, size=(aring, E, N)).astype(np.int32)
        o_host = None
        if call >= 2:
            o_host = np.full((aring, E, N), 0xFF, np.uint8)
            for r_ in range(aring):
                for e in range(E):
                    k = rng2.randint(0, N + 1)
                    perm = rng2.permutation(N)[:k]
                    o_host[r_, e, :k] = perm
                    a_host[r_, e, np.setdiff1d(np.arange(N), perm)] = -1
        else:
            a_host[rng2.rand(aring, E, N) < 0.2] = -1
        a_dev = torch.from_numpy(a_host).cuda()
        o_dev = torch.from_numpy(o_host).cuda() if o_host is not None else None
        want = {}
        for k in range(step0, step0 + n2):
            o_obs, o_rew, _ = ora.step(a_host[k % aring], None if o_host is None else o_host[k % aring])
            want[k] = (o_obs, o_rew)
        eng.rollout_actions(a_dev, n2, obs, rew, done, reset_every=0, step0=step0, fused=fused, order=o_dev)
        g_obs, g_rew = obs.cpu().numpy(), rew.cpu().numpy()
        for k in range(max(step0, step0 + n2 - ring), step0 + n2):
            assert np.array_equal(g_obs[k % ring], want[k][0]) and np.array_equal(g_rew[k % ring], want[k][1]), \
                tag + " rollout_actions call %d fused=%d step %d" % (call, fused, k)
        check_state(eng, ora, tag + " after rollout_actions call %d" % call)
        step0 += n2
    # (4) auto-reset in the step launch, envs out of phase
    Hz = c["horizon"]
    eng.set_horizon(Hz)
    out = eng.alloc_outputs()
    part = c["part"]
    eng.reset(mask=torch.from_numpy(part).cuda(), obs=out[0]); ora.reset(part)
    for s in range(9):
        o, r, d = eng.step_random(out=out, auto_reset=True)
        _, o_obs, o_rew, _ = ora.step_random()
        o_done = ora.get_state()["t"] >= Hz
        if o_done.any():
            r_obs = ora.reset(o_done.astype(np.uint8))
            o_obs[o_done] = r_obs[o_done]
        assert np.array_equal(d.cpu().numpy(), np.repeat(o_done[:, None], N, 1).astype(np.uint8)), tag + " auto done %d" % s
        assert np.array_equal(r.cpu().numpy(), o_rew) and np.array_equal(o.cpu().numpy(), o_obs), tag + " auto step %d" % s
    check_state(eng, ora, tag + " after auto-reset steps")
    assert eng.status() == 0, tag + " status"
    eng.close()


def one(rng, idx, quiet=False, fixed=None):
    run(draw(rng, idx, fixed), quiet)


def replay_cfg215(variants=12):
    """Round 1's only kept fuzz record ended in a parity failure: `cfg 215: game 1 24x8 N=12 v=8 L=1 E=49 keep=1
    seed=561232528 auto step 7` (Cleanup, 12 agents, beams of length 1, SSD_AUTO_RESET).  The map of that run was not kept, so
    the configuration is replayed on `variants` random maps of that shape."""
    for i in range(variants):
        one(np.random.RandomState(215000 + i), 215, quiet=True, fixed=(1, 24, 8, 12, 8, 1, 49, True, 561232528))


def main():
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from _label import label
    label("fuzz_parity " + " ".join(sys.argv[1:]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    for i in range(n):
        one(rng, i)
        if i % 20 == 19:
            print("%d configurations ok" % (i + 1), flush=True)
    print("fuzz ok: %d configurations" % n)


if __name__ == "__main__":
    main()
