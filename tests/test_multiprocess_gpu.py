"""Several PROCESSES on one GPU (the closest a one-GPU box comes to the N-rank job, and the co-tenant case of the dispatch queues):
  * two ranks under torch.distributed (gloo; both on cuda:0) each step their shard of a 4096-env batch with the HIP engine
    (env_index_base 0 / 2048) and all-gather it: equal to one engine stepping all 4096 envs -- run_scripts/train_moa.py:127-128's
    workers, SURVEY.md 8e;
  * rollout calls through the library's own queues (device-drawn and caller-supplied actions) stay bit-exact, status word 0,
    while another process keeps the same GPU busy with plain step launches."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import golden_util as G
from oracle import pyoracle
from sequential_social_dilemma_games_amd import constants as K

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


_RANK_SCRIPT = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
from sequential_social_dilemma_games_amd import constants as K, parallel
from sequential_social_dilemma_games_amd.engine import VecEngine
dist, rank, world, local_rank = parallel.init_process_group("gloo")
assert world == 2
TOTAL, N, STEPS = 4096, 5, 6
game = K.GAME_HARVEST if os.environ["SSD_TEST_GAME"] == "0" else K.GAME_CLEANUP
eng, start, count = parallel.make_sharded_engine(game, None, TOTAL, N, rank, world, local_rank=0, seed=123)
assert (start, count) == (rank * 2048, 2048)
full = VecEngine(game, None, num_envs=TOTAL, num_agents=N, seed=123) if rank == 0 else None
out = eng.alloc_outputs()
obs = eng.reset(obs=out[0])
g = parallel.all_gather_batch(dist, obs.cpu(), TOTAL, world)
if rank == 0:
    assert torch.equal(g, full.reset().cpu()), "reset observations"
for s in range(STEPS):
    obs, rew, done = eng.step_random(out=out)
    g_obs = parallel.all_gather_batch(dist, obs.cpu(), TOTAL, world)
    g_rew = parallel.gather_batch(dist, rew.cpu(), TOTAL, world, rank, dst=0)
    if rank == 0:
        f_obs, f_rew, _ = full.step_random()
        assert torch.equal(g_obs, f_obs.cpu()), "observations of step %%d" %% s
        assert torch.equal(g_rew, f_rew.cpu()), "rewards of step %%d" %% s
# the rollout call (the library's own dispatch queues, in two processes at once) on the shards, ring of 2
ring = tuple(torch.zeros((2,) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in out)
eng.rollout_random(9, ring[0], ring[1], ring[2], reset_every=4, step0=STEPS)
torch.cuda.synchronize()
g_obs = parallel.all_gather_batch(dist, ring[0][(STEPS + 8) %% 2].cpu(), TOTAL, world)
st = eng.get_state()
g_world = parallel.all_gather_batch(dist, torch.from_numpy(st["world"]), TOTAL, world)
if rank == 0:
    fo = tuple(torch.zeros((2,) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in full.alloc_outputs())
    full.rollout_random(9, fo[0], fo[1], fo[2], reset_every=4, step0=STEPS)
    torch.cuda.synchronize()
    assert torch.equal(g_obs, fo[0][(STEPS + 8) %% 2].cpu()), "rollout observations"
    assert np.array_equal(g_world.numpy(), full.get_state()["world"]), "world after the rollout"
    assert full.status() == 0
assert eng.status() == 0
dist.barrier()
print("rank %%d ok" %% rank)
'''


@pytest.mark.parametrize("game", [0, 1])
def test_two_ranks_on_one_gpu_equal_one_engine(game, tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % {"root": ROOT})
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   SSD_TEST_GAME=str(game), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o.decode(errors="replace"))
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % rank) in o, "rank %d:\n%s" % (rank, o[-3000:])


def test_bench_with_two_ranks_rehearsed_on_one_gpu():
    """bench.py's multi-rank control flow (ranks, shards with env_index_base, barriers, the max-over-ranks reduction, rank 0's one
    JSON line) as torch.distributed.run starts it -- on this one GPU: SSD_BENCH_REHEARSAL=1 puts both ranks on the device there
    is and uses gloo.  Not a scaling measurement; the line says so."""
    import json
    port = _free_port()
    env = dict(os.environ, SSD_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--no-configs", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["rccl_ranks"] == 2 and "REHEARSAL" in res["data"]
    assert res["config"]["parallelism"] == "env-shard x2" and res["config"]["envs_per_gpu"] == 4096
    # both ranks' 20 steps of 4096 envs x 5 agents over the slower rank's time
    assert abs(res["value"] - 2 * 4096 * 5 * 20 / (res["ms_per_step"] * 1e-3 * 20)) < 1e-6 * res["value"]
    # every rank says how its rollout was dispatched -- a rank that fell back to hipLaunchKernel must show.
    # (Here two processes share ONE device: its hardware-queue slots are contended, and the dispatch-queue probe of either rank
    # may turn its pool down -- that is the probe working, and the rank says so (queue_dropped).  A rank that did not take the
    # library's queues WITHOUT that reason -- HSA agent not matched, code object not loaded -- is the failure this guards against;
    # on a real multi-GPU node every rank has its device to itself.)
    per_rank = res["config"]["dispatch_per_rank"]
    assert len(per_rank) == 2 and all(p["chains"] >= 1 for p in per_rank), per_rank
    for p in per_rank:
        assert p["aql"] or p["queue_dropped"], per_rank
    assert res["config"]["dispatch_fallback_ranks"] == [r for r, p in enumerate(per_rank) if not p["aql"]]
    assert res["fused_rollout"]["value"] > 0 and "call_overhead_us" in res   # (two ranks share the device: no claim about its sign)


def test_bench_line_survives_failing_legs():
    """VERDICT r02 #1 on the real thing: the driver's command with failures injected into the fused leg, the policy-step leg, the
    configs and the CPU baseline -- the headline comes out complete, every failed leg says so in its slot, exit status 0."""
    import json
    env = dict(os.environ, SSD_BENCH_FAIL_LEG="fused_rollout,policy_step,configs,cpu_baseline")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["value"] > 1e8 and res["n_gpus"] == 1 and res["roofline"]["frac"] > 0 and res["config"]["dispatch_fallback_ranks"] == []
    for leg in ("fused_rollout", "policy_step", "configs", "cpu_baseline"):
        assert "injected failure" in res[leg]["error"], (leg, res[leg])
    assert "call_overhead_us" in res and "us_per_call" in res["busy_stream_call"]       # the legs that were not told to fail ran


_LOAD_SCRIPT = r'''
import sys, time
sys.path.insert(0, %(root)r)
import torch
from sequential_social_dilemma_games_amd import constants as K
from sequential_social_dilemma_games_amd.engine import VecEngine
eng = VecEngine(K.GAME_CLEANUP, None, num_envs=4096, num_agents=5, seed=77)
out = eng.alloc_outputs()
eng.reset(obs=out[0])
print("load running", flush=True)
t0 = time.time()
while time.time() - t0 < %(seconds)f:
    for _ in range(200):
        eng.step_random(out=out)
    torch.cuda.synchronize()
print("load done", flush=True)
'''


@pytest.mark.parametrize("game", [K.GAME_HARVEST, K.GAME_CLEANUP])
def test_rollout_chains_with_another_process_on_the_gpu(game, tmp_path):
    """The stream-side join of a rollout call waits (bounded) for the library's queues, whose kernels share the device with
    whatever else runs there.  With a second process filling the same GPU with plain 4096-env step launches the rollout must still
    come out bit-exact and the wait must not have given up (status word 0: SSD_ST_WAIT_TIMEOUT not set)."""
    import torch
    from sequential_social_dilemma_games_amd.engine import VecEngine
    script = tmp_path / "load.py"
    script.write_text(_LOAD_SCRIPT % {"root": ROOT, "seconds": 6.0})
    load = subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        line = load.stdout.readline().decode()
        while line and "load running" not in line:
            line = load.stdout.readline().decode()
        assert "load running" in line, "the load process did not start"
        amap = K.HARVEST_MAP if game == K.GAME_HARVEST else K.CLEANUP_MAP
        E, N, ring, every, steps = 2048, 5, 2, 151, 400
        eng = VecEngine(game, amap, num_envs=E, num_agents=N, seed=21)
        ora = pyoracle.Oracle(game, amap, E, N, G.default_lut(), seed=21)
        obs = torch.zeros((ring, E, N, 15, 15, 3), dtype=torch.uint8, device="cuda")
        rew = torch.zeros((ring, E, N), dtype=torch.int32, device="cuda")
        for c0 in range(0, steps, 100):                      # several calls while the other process is at work
            eng.rollout_random(100, obs, rew, None, reset_every=every, step0=c0)
        torch.cuda.synchronize()
        assert load.poll() is None, "the load process ended before the rollout did: nothing was tested"
        want = {}
        for k in range(steps):
            if k % every == 0:
                ora.reset()
            _, o_obs, o_rew, _ = ora.step_random(want_obs=(k >= steps - ring))
            want[k] = (o_obs, o_rew)
        g_obs, g_rew = obs.cpu().numpy(), rew.cpu().numpy()
        for k in range(steps - ring, steps):
            np.testing.assert_array_equal(g_rew[k % ring], want[k][1], err_msg="rew of step %d" % k)
            assert np.array_equal(g_obs[k % ring], want[k][0]), "observations of step %d differ" % k
        a, b = eng.get_state(), ora.get_state()
        for key in ("world", "pos", "orient", "episode", "t"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=key)
        assert eng.status() == 0
    finally:
        try:
            load.communicate(timeout=60)
        except subprocess.TimeoutExpired:
            load.kill()
