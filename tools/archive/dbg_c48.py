import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import golden_util as G
from oracle import pyoracle
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine
amap = K.cleanup_map_48x36()
E, N = 256, 10
for chunk in (1, 2, 6):
    eng = VecEngine(K.GAME_CLEANUP, amap, num_envs=E, num_agents=N, seed=4)
    ora = pyoracle.Oracle(K.GAME_CLEANUP, amap, E, N, G.default_lut(), seed=4)
    eng.set_rollout_chains(1)
    obs = torch.zeros((1, E, N, 15, 15, 3), dtype=torch.uint8, device='cuda'); rew = torch.zeros((1, E, N), dtype=torch.int32, device='cuda')
    k = 0
    bad = False
    while k < 60 and not bad:
        eng.rollout_random(chunk, obs, rew, None, reset_every=1000, step0=k)
        torch.cuda.synchronize()
        for s in range(k, k + chunk):
            if s % 1000 == 0: ora.reset()
            _, o_obs, o_rew, _ = ora.step_random()
        k += chunk
        a, b = eng.get_state(), ora.get_state()
        msg = []
        for key in ('world', 'pos', 'orient', 't', 'episode'):
            if not np.array_equal(a[key], b[key]):
                d = np.argwhere(a[key] != b[key])
                msg.append('%s differs at %d places, first %s: got %r want %r' % (key, len(d), d[0], a[key][tuple(d[0])], b[key][tuple(d[0])]))
        if not np.array_equal(rew[0].cpu().numpy(), o_rew): msg.append('rew differs')
        go = obs[0].cpu().numpy()
        if not np.array_equal(go, o_obs):
            d = np.argwhere(go != o_obs)
            msg.append('obs differs at %d bytes in %d envs, first %s' % (len(d), len(set(d[:,0])), d[0]))
        if msg:
            print('chunk %d after step %d: %s path %s' % (chunk, k, '; '.join(msg), eng.rollout_path())); bad = True
    if not bad: print('chunk %d: 60 steps ok' % chunk, eng.rollout_path())
    print('status', eng.status())
