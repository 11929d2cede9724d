#!/usr/bin/env python3
"""bench.py -- random-action rollout throughput of the MapEnv.step() hot path on MI355X.

A "step" is one pass of the hot path over one batch: every env of the batch advances one tick
(device-drawn uniform random actions, as rollout.py:62-70), i.e. ONE launch of the fused
kernel per env range that moves agents, resolves conflicts, fires beams, respawns apples / waste, writes the
uint8 observations [E,N,15,15,3], rewards and dones.  Inputs (all env state) are resident in
HBM when the timed region starts; `horizon` auto-resets are inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--game harvest|cleanup] [--envs E]

N > 1: one rank per GPU under torch.distributed.run (RCCL).  When the process is not already a rank of such a
job (WORLD_SIZE unset) it starts the N ranks itself -- as a CHILD process, before this process has touched
torch or the GPU -- and relays rank 0's JSON line.  Envs are sharded by global index with no data-path
collective (weak scaling: --envs is per GPU).  `--dry-run` is the launcher and sharding alone (gloo, no GPU, no
engine): what the CPU tests run.

Prints ONE JSON line (rank 0).  THE HEADLINE COMES FIRST AND CANNOT BE LOST: `value`, `ms_per_step`, `roofline` are
computed as soon as the K timed steps are done; everything after that -- the call-overhead leg, the fused leg, the
policy-step legs, the other configurations, the CPU baseline, the optional gather legs -- is an OPTIONAL LEG that
runs inside its own guard: whatever it raises (an exception, SystemExit, an out-of-memory error) lands as
{"error": "..."} in that leg's slot of the line, and the line is printed with exit status 0 all the same.  Legs that
contain collectives (the gather legs) are opt-in (`--gather-leg`): a rank that fails inside one cannot be waited for.

ONE clock: `value`, `ms_per_step` and `roofline.frac` all come from the wall time of the K timed steps (barrier +
synchronize on both sides, max over ranks); the HIP-event duration of the same region on the launch stream is
reported next to it (`roofline.hip_event_us_per_step`).  `roofline.achieved` = algorithmic bytes per step
(bytes / env-step x envs) / time per step.  `cpu_baseline` = the C oracle (a port of the reference
algorithm, oracle/ssd_oracle.c) timed on this host, one core, bounded sample.  `configs` = the other single-GPU
configurations of BASELINE.json, a few hundred steps each (N = 1 only).  `policy_step` = the step with
CALLER-SUPPLIED actions (what the reference's callers do: env.step(policy actions)): per-call stepping through the
batched adapter, and ssd_rollout_actions chunks.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# A rank of a process group (RCCL).  The HIP runtime multiplexes its streams (torch's, RCCL's) onto at most GPU_MAX_HW_QUEUES hardware
# queues (default 4), the library adds queues of its own, and beyond FOUR queues per process the hardware scheduler time-slices
# them: a rollout call that follows an RCCL barrier then takes 320 us instead of 130 (measured under torch.distributed.run: 3 or
# more here -> 16.2 us per step of a 20-step call, 1 or 2 -> 6.7).  With 2 for the runtime the library keeps to 2 of its own
# (include/ssd.h, "the queue rule").  Read by the runtime when it loads: set before torch is imported; an explicit setting wins.
if "MASTER_PORT" in os.environ or int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); ~6.3 TB/s achievable
XGMI_LINK_GBS = 153.0        # per direct peer link (7 per GPU)
HORIZON = 1000               # run_scripts/train_baseline.py:131
ROUND = "r04"


def _oracle_worker(game, amap, n_agents, E, seconds, seed, out, idx):
    from oracle import pyoracle
    from sequential_social_dilemma_games_amd import config
    o = pyoracle.Oracle(game, amap, E, n_agents, config.make_lut(), seed=seed, env_base=idx * E)
    o.reset()
    for _ in range(3):
        o.step_random()
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            o.step_random()
        steps += 10
    out[idx] = (steps, time.perf_counter() - t0)


def cpu_baseline(game, amap, n_agents, target_s=6.0, E=512):
    """The oracle on the host cores over a bounded sample of the same workload: one core (the headline `value` of this
    object), then every core the process may use (independent env shards, one thread each; the C call releases the GIL)."""
    import threading
    out = [None]
    _oracle_worker(game, amap, n_agents, E, target_s, 0, out, 0)
    steps, dt = out[0]
    res = {"value": E * n_agents * steps / dt, "unit": "agent-env-steps/s", "cores": 1, "kind": "port",
           "sample": "%d envs x %d steps (%.1f s), oracle/ssd_oracle.c single thread; the Python reference itself "
                     "measured 2289 agent-env-steps/s per core in the build container (BASELINE.md)" % (E, steps, dt)}
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 16)                         # a one-GPU box's CPU share is 16 cores, whatever the affinity mask says
    if cores > 1:
        outs = [None] * cores
        ths = [threading.Thread(target=_oracle_worker, args=(game, amap, n_agents, E, target_s / 2, 0, outs, i)) for i in range(cores)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        wall = time.perf_counter() - t0
        total = sum(E * n_agents * o[0] for o in outs if o)
        res["all_cores"] = {"value": total / wall, "unit": "agent-env-steps/s", "cores": cores,
                            "sample": "%d threads x %d envs for %.1f s, independent env shards" % (cores, E, wall)}
    return res


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv):
    """Start the N-rank job as a child process (torch.distributed.run, one rank per GPU) and relay rank 0's JSON line.
    Runs before this process imports torch or touches the GPU; this process never becomes a rank itself."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.lstrip().startswith("{")]
    rc = proc.returncode
    if not lines:
        print("bench.py: the %d-rank job printed no JSON line (exit status %d)" % (n, rc), file=sys.stderr)
        return rc or 1
    print(lines[-1])
    sys.stdout.flush()
    if rc:
        # the headline is out; a non-zero status of the job after that (a rank that died in an optional leg, a slow teardown)
        # must not make the caller throw the line away
        print("bench.py: the %d-rank job ended with status %d AFTER rank 0 had printed its line" % (n, rc), file=sys.stderr)
    return 0


GAMES = {"harvest": (0, "HARVEST_MAP", 5), "cleanup": (1, "CLEANUP_MAP", 5),
         "harvest25x38": (0, "harvest_map_25x38", 5), "cleanup48x36": (1, "cleanup_map_48x36", 10)}


def game_spec(name, agents=None):
    from sequential_social_dilemma_games_amd import constants as K
    game, m, n = GAMES[name]
    amap = getattr(K, m)
    if callable(amap):
        amap = amap()
    return game, amap, (n if agents is None else agents)


# ---------------------------------------------------------------------------------------------------------------
# optional legs: each in its own guard
# ---------------------------------------------------------------------------------------------------------------
def run_leg(res, key, fn):
    """One optional leg.  Its result goes to res[key]; whatever it raises goes there as {"error": ...} instead -- the headline
    that is already in `res` is never in danger.  SSD_BENCH_FAIL_LEG=<key>[,<key>...] makes the named legs fail (tests)."""
    try:
        if key in os.environ.get("SSD_BENCH_FAIL_LEG", "").split(","):
            raise RuntimeError("injected failure (SSD_BENCH_FAIL_LEG)")
        out = fn()
        if out is not None:
            res[key] = out
        return True
    except KeyboardInterrupt:
        raise
    except BaseException as exc:                   # SystemExit and MemoryError included
        res[key] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:400])}
        print("bench.py: optional leg %r failed: %s: %s" % (key, type(exc).__name__, exc), file=sys.stderr)
        return False


def time_rollout(torch, eng, ring, steps, warmup, step0=0, busy=None, **kw):
    """W untimed + K timed steps of eng.rollout_random (calls of at most 1000 steps).  Returns (wall s, HIP-event ms, enqueue s).
    busy: a callable that puts a (short) kernel on the stream right before the timed calls -- the call then forks from a stream
    with work pending, as a rollout inside a training loop does."""
    def run(k0, n):
        for c0 in range(k0, k0 + n, 1000):
            eng.rollout_random(min(1000, k0 + n - c0), ring[0], ring[1], ring[2], reset_every=HORIZON, step0=c0, **kw)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ev1.record()
    wchunk = max(4, warmup // 4)
    for w0 in range(0, warmup, wchunk):
        run(step0 + w0, min(wchunk, warmup - w0))
    if busy is not None:
        busy()                                     # (its first launch resolves the kernel: not inside the timed region)
    ev0.record()                                   # (before the opening synchronize: see main())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if busy is not None:
        busy()
    run(step0 + warmup, steps)
    ev1.record()
    enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall, ev0.elapsed_time(ev1), enq


def config_leg(torch, name, E, steps=400, warmup=50, obs_f32=False):
    """One of the other single-GPU workloads, stepped the way the headline is (ssd_rollout_random, automatic chains)."""
    from sequential_social_dilemma_games_amd.engine import VecEngine
    game, amap, n_agents = game_spec(name)
    eng = VecEngine(game, amap, num_envs=E, num_agents=n_agents, seed=0)
    try:
        out = eng.alloc_outputs(float32=obs_f32)
        ring = tuple(t.unsqueeze(0) for t in out)
        eng.set_rollout_chains(0)                  # the library's own choice
        wall, dev_ms, _ = time_rollout(torch, eng, ring, steps, warmup)
        path = eng.rollout_path()
        if eng.status() != 0:
            raise RuntimeError("device status word is non-zero (%s)" % name)
        bytes_env = eng.algorithmic_bytes_per_env_step() + (n_agents * eng.V * eng.V * 3 * 3 if obs_f32 else 0)
        us = wall * 1e6 / steps
        return {"workload": "%s %dx%d, %d agents, %d envs%s" % (name, eng.H, eng.W, n_agents, E, ", float32 obs" if obs_f32 else ""),
                "steps": steps, "warmup": warmup, "ms_per_step": us * 1e-3, "value": E * n_agents * steps / wall,
                "bytes_per_env_step": bytes_env, "frac": bytes_env * E / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "hip_event_us_per_step": dev_ms * 1e3 / steps, "chains": path["chains"], "split": path["split"]}
    finally:
        eng.close()


def policy_step_leg(torch, game, amap, n_agents, E, steps=300, K=20):
    """The step with CALLER-SUPPLIED actions: what the reference's callers do (visuallizer_rllib.py:121-153; RLlib's sampler behind
    train_baseline.py:71-81).  (a) SSDVectorEnv.step(device action tensor): one Python call, one hipLaunchKernel of the plain
    step kernel and, when the horizon comes, one reset launch per step; (b) the engine's own step call; (c) ssd_rollout_actions: chunks of K steps per call (synchronised after every call, as the driver's
    K-step region is) and one long call.  us per 4096-env step each; actions are a fixed random tensor on the device."""
    from sequential_social_dilemma_games_amd.vector_env import SSDVectorEnv
    na = 8 if game == 0 else 9
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    venv = SSDVectorEnv(game, E, n_agents, horizon=HORIZON, ascii_map=amap, seed=0)
    eng = venv.engine
    try:
        acts = torch.randint(0, na, (K, E, n_agents), dtype=torch.int32, device="cuda", generator=gen)
        venv.reset()
        out = {"label": "NOT the headline: steps with caller-supplied actions (a fixed random int32 tensor on the device), %d envs" % E,
               "unit": "us per step", "steps": steps}

        def per_call(fn, n):
            for i in range(20):
                fn(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e6 / n
        out["vector_env_step_us"] = per_call(lambda i: venv.step(acts[i % K]), steps)
        out["engine_step_us"] = per_call(lambda i: eng.step(acts[i % K], out=venv._out), steps)
        calls = max(3, steps // K)
        # (b') the same K per-call steps captured ONCE into a HIP graph (torch.cuda.CUDAGraph: ssd_step with device pointers is one
        # kernel launch on the caller's stream and nothing else, tests/test_hip_parity.py) and replayed: what a training loop that
        # captures policy + step gets -- the host's 5 - 6 us per hipLaunchKernel are gone, the kernels remain
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for k in range(K):
                    eng.step(acts[k], out=venv._out)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for k in range(K):
                    eng.step(acts[k], out=venv._out)
            torch.cuda.synchronize()
            for i in range(3):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(calls):
                graph.replay()
                torch.cuda.synchronize()
            out["engine_step_hip_graph_k%d_us_per_step" % K] = (time.perf_counter() - t0) * 1e6 / (calls * K)
            t0 = time.perf_counter()
            for i in range(calls):
                graph.replay()
            torch.cuda.synchronize()
            out["engine_step_hip_graph_back_to_back_us_per_step"] = (time.perf_counter() - t0) * 1e6 / (calls * K)
            del graph
        except Exception as exc:                     # (an optional figure: never the leg's failure)
            out["engine_step_hip_graph_error"] = "%s: %s" % (type(exc).__name__, exc)
        ring = tuple(t.unsqueeze(0) for t in venv._out)
        eng.set_rollout_chains(0)

        def chunk(i):
            eng.rollout_actions(acts, K, ring[0], ring[1], ring[2], reset_every=HORIZON, step0=(i * K) % HORIZON)
            torch.cuda.synchronize()
        for i in range(3):
            chunk(i)
        t0 = time.perf_counter()
        for i in range(calls):
            chunk(i)
        out["rollout_actions_k%d_us_per_step" % K] = (time.perf_counter() - t0) * 1e6 / (calls * K)
        out["rollout_actions_dispatch"] = eng.rollout_path()
        t0 = time.perf_counter()
        eng.rollout_actions(acts, 1000, ring[0], ring[1], ring[2], reset_every=HORIZON, step0=0)
        torch.cuda.synchronize()
        out["rollout_actions_long_call_us_per_step"] = (time.perf_counter() - t0) * 1e6 / 1000

        def fchunk(i):
            eng.rollout_actions(acts, K, ring[0], ring[1], ring[2], reset_every=HORIZON, step0=(i * K) % HORIZON, fused=True)
            torch.cuda.synchronize()
        for i in range(3):
            fchunk(i)
        t0 = time.perf_counter()
        for i in range(calls):
            fchunk(i)
        out["rollout_actions_fused_k%d_us_per_step" % K] = (time.perf_counter() - t0) * 1e6 / (calls * K)
        if eng.status() != 0:
            raise RuntimeError("device status word is non-zero")
        return out
    finally:
        eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--game", default="harvest", choices=sorted(GAMES))
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=None)
    ap.add_argument("--gather", action="store_true", help="the WHOLE run with an RCCL all-gather of obs/rew (one collective per --gather-steps steps)")
    ap.add_argument("--gather-leg", action="store_true", help="a process group's run: also report, separately, the rollout with the "
                    "observations and rewards (a) all-gathered to every rank, (b) gathered to rank 0 (opt-in: legs with collectives)")
    ap.add_argument("--gather-steps", type=int, default=32, help="steps per collective of the gather modes")
    ap.add_argument("--obs-f32", action="store_true", help="separate mode: the kernel writes float32 observations (4x the obs bytes)")
    ap.add_argument("--per-step-calls", action="store_true", help="one Python call per step instead of ssd_rollout_random")
    ap.add_argument("--ring", type=int, default=1, help="output ring slots of ssd_rollout_random (step k writes slot k %% ring)")
    ap.add_argument("--chains", type=int, default=0, help="env ranges stepped concurrently by ssd_rollout_random (0 = automatic)")
    ap.add_argument("--strict-dispatch", action="store_true", help="refuse to time anything if a rank's rollout did not take the "
                    "library's own dispatch queues (default: time it, and say so loudly in the line and on stderr)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` legs (the other single-GPU workloads)")
    ap.add_argument("--no-extras", action="store_true", help="only the headline: no optional leg at all")
    ap.add_argument("--dry-run", action="store_true", help="launcher + sharding only: gloo ranks, no GPU, no engine")
    ap.add_argument("--dry-run-legs", action="store_true", help="--dry-run: also run the legs that need no GPU (a short cpu_baseline) "
                    "through the same guards as the real run")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.no_extras:
        args.no_cpu_baseline = args.no_configs = True

    # ---- N ranks: if this process is not one of them yet, start them (as a child; nothing here has touched torch / HIP) ----
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    # The contract is ONE JSON line on stdout.  Libraries write to fd 1 behind Python's back (RCCL prints a
    # version banner when the first communicator is created), so park the real stdout and point fd 1 at
    # stderr for the rest of the run.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(res):
        print(json.dumps(res), file=real_stdout)
        real_stdout.flush()

    from sequential_social_dilemma_games_amd import parallel
    if args.dry_run:
        dist, rank, world, local_rank = parallel.init_process_group("gloo")
        if world != args.gpus:
            raise SystemExit("--gpus %d but the job has %d ranks" % (args.gpus, world))
        span = parallel.shard_range(world * args.envs, world, rank)
        spans = [span]
        if dist is not None:
            spans = [None] * world
            dist.all_gather_object(spans, span)
            dist.barrier()
        if rank == 0:
            res = {"dry_run": True, "n_gpus": world, "ranks": world, "envs_per_gpu": args.envs,
                   "shards": [list(s) for s in spans], "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}
            if args.dry_run_legs:                  # (the guards of the real run, on the legs that need no GPU)
                game, amap, n_agents = game_spec(args.game, args.agents)
                run_leg(res, "cpu_baseline", lambda: cpu_baseline(game, amap, n_agents, target_s=0.3, E=32))
                run_leg(res, "configs", lambda: [{"workload": "(dry run: no GPU)"}])
            emit(res)
        if dist is not None:
            dist.destroy_process_group()
        return

    import torch
    from sequential_social_dilemma_games_amd.engine import VecEngine  # noqa: F401
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # SSD_BENCH_REHEARSAL=1: the N ranks share the GPUs there are (rank r on device r % count) and talk over gloo -- the multi-rank
    # control flow of this file on a box with fewer GPUs than ranks (tests/test_multiprocess_gpu.py); its line says so.
    rehearsal = os.environ.get("SSD_BENCH_REHEARSAL") == "1"
    dist, rank, world, local_rank = parallel.init_process_group("gloo" if rehearsal else "nccl")
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if rehearsal else "cuda"        # where the timing reductions live (gloo: host tensors)
    if world != args.gpus:
        raise SystemExit("--gpus %d but the job has %d ranks (WORLD_SIZE)" % (args.gpus, world))
    # The communicator comes up HERE, not in the barrier that opens the timed region: RCCL builds it in the group's first
    # collective, and the rollout call that follows that one costs 185 us instead of 125 (the second: 137; then as without a
    # group -- tools/rccl_short_call_probe.py).  Set-up of the communication library, not of the steps.
    for _ in range(3):
        parallel.barrier(dist, local_rank)
    torch.cuda.synchronize()

    game, amap, n_agents = game_spec(args.game, args.agents)
    E = args.envs                                  # per GPU (weak scaling); global batch = world * E
    eng, start, count = parallel.make_sharded_engine(game, amap, world * E, n_agents, rank, world,
                                                     local_rank=local_rank, seed=0)
    assert (start, count) == (rank * E, E)
    # (the engine steps on the rank's own device -- a rank that silently stepped on device 0 would still produce a figure)
    assert eng.device == local_rank == torch.cuda.current_device(), (eng.device, local_rank, torch.cuda.current_device())
    out = eng.alloc_outputs(float32=args.obs_f32)
    do_gather = bool(args.gather and dist is not None and not rehearsal)
    GR = max(1, args.gather_steps)                 # gather modes: steps per collective (one RCCL collective moves GR steps' outputs)
    G = {}                                         # the gather's buffers, streams and events (made on first use)

    def ensure_gather(root_only=False):            # the batched tensors: [world, GR, E, ...] on every rank (all-gather) / on rank 0
        if "rings" not in G:
            # two rings: while the collective of one is in flight on its own stream, the rollout fills the other
            G["rings"] = [(torch.empty((GR,) + tuple(out[0].shape), dtype=out[0].dtype, device=out[0].device),
                           torch.empty((GR,) + tuple(out[1].shape), dtype=torch.int32, device=out[1].device),
                           torch.empty((GR,) + tuple(out[2].shape), dtype=torch.uint8, device=out[2].device)) for _ in range(2)]
            G["stream"] = torch.cuda.Stream()
            G["ready"] = [torch.cuda.Event() for _ in range(2)]     # the rollout has filled ring i
            G["free"] = [torch.cuda.Event() for _ in range(2)]      # the collective has read ring i
        need = world > 1 and (rank == 0 or not root_only)
        if need and "buf" not in G:
            G["buf"] = (torch.empty((world,) + tuple(G["rings"][0][0].shape), dtype=G["rings"][0][0].dtype, device=out[0].device),
                        torch.empty((world,) + tuple(G["rings"][0][1].shape), dtype=torch.int32, device=out[1].device))
    if do_gather:
        ensure_gather()

    def one_step(k):
        if k % HORIZON == 0:
            eng.reset(obs=out[0])
        eng.step_random(out=out)

    # Without --gather / --per-step-calls the steps are enqueued by ssd_rollout_random: the same launches (one step
    # kernel per step and env range into the same output buffers, a full reset every HORIZON steps), issued by one library
    # call per chunk instead of one Python call per step, so that the host never starves the kernels.
    ring = tuple(t.unsqueeze(0) for t in out) if args.ring <= 1 else \
        tuple(torch.empty((args.ring,) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in out)
    use_rollout = not args.per_step_calls
    chains = 1
    if use_rollout:                                # env ranges the library steps on queues of its own (envs are independent)
        eng.set_rollout_chains(args.chains)        # (0: the library's own choice; what it was is read back after the timed call)

    def run_steps(k0, n, gather=None):
        """gather: None = as the run was asked (--gather), "all" = all-gather to every rank, "root" = gather to rank 0."""
        mode = ("all" if do_gather else None) if gather is None else gather
        if mode:
            # GR steps into a ring of GR slots, then ONE RCCL collective of the ring's observations and one of its rewards over
            # xGMI, on a stream of their own, while the next GR steps fill the other ring
            main_s = torch.cuda.current_stream()
            grings, comm_stream, ring_ready, ring_free = G["rings"], G["stream"], G["ready"], G["free"]
            gbuf = G.get("buf")
            for c0 in range(k0, k0 + n, GR):
                m = min(GR, k0 + n - c0)
                i = (c0 // GR) % 2
                gring = grings[i]
                main_s.wait_event(ring_free[i])        # (no-op until the event has been recorded once)
                if use_rollout:
                    eng.rollout_random(m, gring[0], gring[1], gring[2], reset_every=HORIZON, step0=c0)
                else:
                    for k in range(c0, c0 + m):
                        if k % HORIZON == 0:
                            eng.reset(obs=gring[0][k % GR])
                        eng.step_random(out=(gring[0][k % GR], gring[1][k % GR], gring[2][k % GR]))
                ring_ready[i].record(main_s)
                with torch.cuda.stream(comm_stream):   # the collectives overlap the next chunk's steps
                    comm_stream.wait_event(ring_ready[i])
                    if mode == "all":
                        parallel.all_gather_ring(dist, gring[0], world, out=gbuf[0] if gbuf else None)
                        parallel.all_gather_ring(dist, gring[1], world, out=gbuf[1] if gbuf else None)
                    else:
                        parallel.gather_ring(dist, gring[0], world, rank, out=gbuf[0] if gbuf else None)
                        parallel.gather_ring(dist, gring[1], world, rank, out=gbuf[1] if gbuf else None)
                    ring_free[i].record(comm_stream)
            main_s.wait_stream(comm_stream)            # the K steps are not done before their last collective is
        elif use_rollout:
            for c0 in range(k0, k0 + n, 1000):
                eng.rollout_random(min(1000, k0 + n - c0), ring[0], ring[1], ring[2], reset_every=HORIZON, step0=c0)
        else:
            for k in range(k0, k0 + n):
                one_step(k)

    # (set-up, not steps: the two timing events exist and have been recorded once before anything is timed -- the runtime
    # builds its event machinery on first use, ~30 us)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ev1.record()
    torch.cuda.synchronize()
    # ---- how did every rank's rollout get dispatched?  Known BEFORE anything is timed, reported per rank.  A rank whose HSA agent
    #      could not be matched to its HIP device (device = local_rank != 0 has never run before the first multi-GPU job), or whose
    #      dispatch queues failed their probe, steps through hipLaunchKernel: same results, slower -- never silently. ----
    def exchange_paths():
        torch.cuda.synchronize()
        my_path = eng.rollout_path() if use_rollout and args.warmup > 0 else None
        if my_path is not None:
            # which physical device this rank stepped on (what the library matched its HSA agent against): eight ranks must show
            # eight different addresses, and "agent_match" how each was found
            pr = torch.cuda.get_device_properties(local_rank)
            my_path = dict(my_path, rank=rank, device=eng.device,
                           pci="%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)))
        paths = [my_path]
        if dist is not None:
            paths = [None] * world
            dist.all_gather_object(paths, my_path)
        want_aql = os.environ.get("SSD_AQL", "1") != "0"
        fallback_ranks = [r for r, p in enumerate(paths) if p is not None and want_aql and not p["aql"]]
        pcis = [p["pci"] for p in paths if p is not None]
        if not rehearsal and len(set(pcis)) != len(pcis):
            raise SystemExit("bench.py: two ranks stepped on the same physical device: %s" % (paths,))
        if fallback_ranks:
            msg = ("bench.py: rank(s) %s did NOT take the library's own dispatch queues (hipLaunchKernel fallback: SSD_AQL_VERBOSE=1 "
                   "says why); paths: %s" % (fallback_ranks, paths))
            print("=" * 100 + "\n" + msg + "\n" + "=" * 100, file=sys.stderr)
            if args.strict_dispatch:
                raise SystemExit(msg)
        return paths, fallback_ranks
    # The W warm-up steps, issued as a few calls rather than one (every call exercises the whole enqueue path).  The ranks' dispatch
    # paths are exchanged after the FIRST of those calls -- an object collective takes milliseconds, and a GPU left idle that long
    # ahead of the timed region starts it cold (tools/idle_gap_probe.py: a 20-step call 126 us right after other work, 131 after
    # 1 ms of nothing, 135 - 139 after 5 ms) -- so that the rest of the warm-up does what a warm-up is for: it ends at the opening
    # barrier.
    # (and no collector pause in or just ahead of a timed region of ~130 us -- the ranks report the MAX of theirs: a collection right
    # in front of it cost 30 us, 8.0 against 6.6 us per step: host caches and device both cold; so: collect NOW, before the warm-up,
    # and keep the collector off until the timed steps are done)
    import gc
    gc.collect()
    gc.disable()
    wchunk = max(4, args.warmup // 4)
    paths, fallback_ranks = None, []
    for w0 in range(0, args.warmup, wchunk):
        run_steps(w0, min(wchunk, args.warmup - w0))
        if paths is None:
            paths, fallback_ranks = exchange_paths()
    if paths is None:
        paths, fallback_ranks = exchange_paths()
    parallel.barrier(dist, local_rank)
    # The opening event is recorded BEFORE the opening synchronize: the timed region then starts on an idle stream, as a rollout
    # call after any synchronize does (an event record just ahead of the call would make the library fork from a "busy" stream:
    # ~20 us of a 20-step region; that case is the `busy_stream_call` leg below).  The HIP-event interval therefore also covers
    # that synchronize's return (~10 us): it is the secondary clock; `value`, `ms_per_step` and `roofline.frac` use the wall clock.
    ev0.record()                                   # torch's current stream == the stream the kernels are launched on
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    ev1.record()
    enq = time.perf_counter() - t0                 # host time to enqueue the K steps (must stay below the device time)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0                # this rank's K steps, start barrier to its own completion; MAX over ranks below
    gc.enable()
    parallel.barrier(dist, local_rank)             # (the closing barrier: everybody is done before anybody reports; its own
    torch.cuda.synchronize()                       #  latency, ~1 ms of RCCL, is not part of any rank's K steps)
    dev_ms = ev0.elapsed_time(ev1)
    path = eng.rollout_path() if use_rollout else None  # how the library dispatched the timed call (a fallback must not go unnoticed)
    if path:
        chains = path["chains"]
    status = eng.status()
    if dist is not None:
        tw = torch.tensor([wall, dev_ms, enq, float(status)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall, dev_ms, enq, status = float(tw[0]), float(tw[1]), float(tw[2]), int(tw[3])
    if status != 0:
        raise SystemExit("device status word is non-zero (%d) on some rank" % status)

    # ---- THE HEADLINE: complete before any optional leg runs ----
    res = None
    total_agent_steps = float(E) * n_agents * args.steps * world
    bytes_env = eng.algorithmic_bytes_per_env_step()
    if args.obs_f32:                               # the obs term becomes N * 2700 in this mode
        bytes_env += n_agents * eng.V * eng.V * 3 * 3
    if rank == 0:
        value = total_agent_steps / wall
        step_us = wall * 1e6 / args.steps          # THE clock: wall time of the K steps, per step
        achieved = bytes_env * E / (step_us * 1e-6) / 1e9

        def describe(p):
            if not p:
                return "one call per step (hipLaunchKernel)"
            return "%s, %d chain(s)%s%s%s" % ("AQL packets in the library's own HSA queues" if p["aql"] else "hipLaunchKernel on HIP streams",
                                               p["chains"], ", coherent kernel variant (no fence between a chain's launches)" if p["coherent"] else "",
                                               ", observations rendered by the next step's launch" if p["split"] else "",
                                               ", host-side waits (a profiling tool is attached / SSD_AQL_SYNC)" if p["sync"] else "")
        res = {
            "metric": "agent-env-steps/sec (random actions)", "value": value, "unit": "agent-env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_us * 1e-3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.obs_f32 else "u8",
            "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL: %d ranks on %d GPU(s), gloo -- not a scaling figure" % (world, torch.cuda.device_count()),
            "config": {"workload": "%s %dx%d, %d agents, %d envs per GPU, uniform random actions, reset every %d steps"
                                   % (args.game, eng.H, eng.W, n_agents, E, HORIZON),
                       "envs_per_gpu": E, "agents": n_agents, "obs": ("float32" if args.obs_f32 else "uint8") + " [E,N,15,15,3]", "launches_per_step": chains,
                       "enqueue": ("ssd_rollout_random, %d chain(s) of %d envs" % (chains, E // chains)) if use_rollout else "one call per step",
                       "gather": ("one RCCL all-gather of obs and one of rewards per %d steps, overlapped with the next %d steps" % (GR, GR)) if do_gather else False,
                       "parallelism": "env-shard x%d" % world, "rccl_ranks": dist.get_world_size() if dist is not None else 1,
                       "dispatch": describe(path), "dispatch_timed_call": path, "dispatch_per_rank": paths, "dispatch_fallback_ranks": fallback_ranks},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "ssd::ssd_env_kernel<%d, 0, %s, ...>" % (game, "true" if args.obs_f32 else "false"), "bytes_per_env_step": bytes_env,
                         "clock": "wall time of the K timed steps (the same clock as value and ms_per_step)",
                         "hip_event_us_per_step": dev_ms * 1e3 / args.steps, "concurrent_launches": chains, "envs_per_launch": E // chains,
                         "note": "achieved = algorithmic bytes per step (all concurrent launches) / time per step; each chain's launches "
                                 "run back to back in its own queue, so time per step = launch-to-launch duration of the step kernel",
                         "host_enqueue_us_per_step": enq * 1e6 / args.steps},
        }
        # HBM bytes per step from the PMC counters of the committed profile of this exact workload
        # (tools/profile_workloads.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH_SIZE correction)

        def traffic():
            tkey = "%s_%dx%d_n%d_e%d%s" % (args.game.rstrip("0123456789x"), eng.H, eng.W, n_agents, E, "_f32" if args.obs_f32 else "")
            for rnd in (ROUND, "r03", "r02", "r01"):
                tpath = os.path.join(REPO, "profiles", "%s_traffic.json" % rnd)
                if not os.path.exists(tpath):
                    continue
                ent = json.load(open(tpath)).get(tkey)
                if ent:
                    res["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
                    res["roofline"]["traffic_source"] = "profiles/%s_traffic.json[%s] (rocprofv3 PMC, same workload; sum over the step's concurrent launches)" % (rnd, tkey)
                    if "rocprof_avg_kernel_us" in ent:
                        # what rocprofv3 --kernel-trace --stats says one launch of that kernel lasts (committed summary; tracing slows
                        # the dispatch path down, so this is the kernel's duration in that regime, not the unprofiled step time)
                        res["roofline"]["rocprof_avg_kernel_us"] = ent["rocprof_avg_kernel_us"]
                        res["roofline"]["rocprof_source"] = ent.get("rocprof_source")
                    return
        run_leg(res, "_traffic", traffic)
        if "_traffic" in res:                      # (only ever an error slot)
            res["roofline"]["traffic_error"] = res.pop("_traffic")["error"]
    holder = res if res is not None else {}        # (every rank runs the legs; only rank 0's slots are printed)

    try:
        plain = use_rollout and not do_gather and not args.no_extras
        # ---- optional leg: what one rollout call costs beyond its steps.  The same K steps are timed again as part of ONE long
        #      call's rate (1000 steps, same launches); call_overhead_us = wall(K steps) - K x (per-step time of the long call) ----
        if plain and args.steps < 1000:
            def overhead():
                lw, _, _ = time_rollout(torch, eng, ring, 1000, 0, step0=args.warmup + args.steps)
                long_us = lw * 1e6 / 1000
                holder["long_call_us_per_step"] = long_us
                return wall * 1e6 - args.steps * long_us
            run_leg(holder, "call_overhead_us", overhead)

            # ---- optional leg: the same K-step call once the process is warm.  The timed region above is the process's FIRST call of
            #      its length, W steps (the driver's command: 5 steps = 30 us) after seconds of start-up with the device idle; a
            #      training loop issues such calls back to back.  Median of 30 further calls of K steps, each synchronised.
            def warm_calls():
                walls = []
                s0 = args.warmup + args.steps + 1000
                for i in range(30):
                    w, _, _ = time_rollout(torch, eng, ring, args.steps, 0, step0=s0 + i * args.steps)
                    walls.append(w)
                walls.sort()
                med = walls[len(walls) // 2] * 1e6 / args.steps
                return {"label": "NOT the headline: the same %d-step call, median of 30 issued back to back later in the same process" % args.steps,
                        "us_per_step": med, "fastest_us_per_step": walls[0] * 1e6 / args.steps, "slowest_us_per_step": walls[-1] * 1e6 / args.steps,
                        "roofline_frac": bytes_env * E / (med * 1e-6) / 1e9 / HBM_PEAK_GBS}
            run_leg(holder, "warm_process_call", warm_calls)
        # ---- optional leg: the same K steps from a stream that has work pending when the call comes (a rollout inside a training
        #      loop: policy kernels are still queued), so that the call forks from it -- the headline's region starts on an idle stream
        if plain:
            def busy_call():
                x = torch.zeros(1 << 24, device="cuda")       # 64 MB: the pending kernel lasts ~30 us, the call must fork from the stream
                x.add_(1.0)
                torch.cuda.synchronize()
                alone = []
                for _ in range(7):
                    t1 = time.perf_counter()
                    x.add_(1.0)
                    torch.cuda.synchronize()
                    alone.append(time.perf_counter() - t1)
                alone_us = sorted(alone)[len(alone) // 2] * 1e6
                walls, forked = [], False
                for rep_i in range(5):             # (median of 5: one shot of a ~170 us region is noisy)
                    bw, _, _ = time_rollout(torch, eng, ring, args.steps, 4 if rep_i == 0 else 0, step0=args.warmup + args.steps + 8 + rep_i * args.steps,
                                            busy=lambda: x.add_(1.0))
                    walls.append(bw)
                    forked = forked or eng.rollout_path()["forked"]
                    if args.steps > 200:
                        break
                bw = sorted(walls)[len(walls) // 2]
                return {"label": "NOT the headline: the same %d steps called while a 64-MB elementwise kernel is still pending on the stream, so "
                                 "that the call forks from the stream (a one-wave kernel on the stream bumps a counter the chains' first packet "
                                 "polls); median of %d runs; the pending kernel's own duration is subtracted in us_per_step_beyond_pending"
                                 % (args.steps, len(walls)),
                        "us_per_call": bw * 1e6, "pending_kernel_alone_us": alone_us,
                        "us_per_step_beyond_pending": (bw * 1e6 - alone_us) / args.steps, "forked": forked}
            run_leg(holder, "busy_stream_call", busy_call)
        # ---- optional leg: the rollout into a RING of 32 output slots (what a trainer that keeps a trajectory does).  The headline's one
        #      slot (13.8 MB, rewritten every step) lives in the device's 256-MB memory-side cache; 32 slots (442 MB) do not, and the
        #      observation stores then stream to HBM (non-temporal write-back: the library's choice past 232 MB) ----
        if plain and world == 1 and args.ring <= 1:
            def ring_leg():
                R = 32
                big = tuple(torch.empty((R,) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in out)
                n = max(256, args.steps)
                rw, _, _ = time_rollout(torch, eng, big, n, 64, step0=args.warmup + args.steps)
                us = rw * 1e6 / n
                return {"label": "NOT the headline: the same rollout into a ring of %d output slots (%d MB of observations: past the 256-MB "
                                 "memory-side cache that absorbs the headline's single slot)" % (R, R * big[0][0].numel() * big[0].element_size() >> 20),
                        "steps": n, "us_per_step": us, "roofline_frac": bytes_env * E / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
            run_leg(holder, "output_ring_32", ring_leg)
        # ---- optional leg, reported separately and labelled: the same K steps as ONE fused kernel launch --
        #      every env resident in LDS / registers across its steps (SSD_ROLLOUT_FUSED).  Not part of `value`. ----
        if plain and not args.obs_f32:
            def fused():
                eng.set_rollout_chains(0)              # the library's own choice (one launch per GPU: no second stream to fork and join)
                try:
                    fw, _, _ = time_rollout(torch, eng, ring, args.steps, args.warmup, fused=True)
                    fus = fw * 1e6 / args.steps
                    o = {"label": "NOT the headline: the same %d steps as ONE kernel launch per GPU (ssd_rollout_random + SSD_ROLLOUT_FUSED), "
                                  "envs resident in LDS across steps; per-step obs / rew / done still written to HBM%s"
                                  % (args.steps, "; this rank's figure" if world > 1 else ""),
                         "value": float(E) * n_agents * args.steps / fw, "unit": "agent-env-steps/s per GPU", "us_per_step": fus,
                         "roofline_frac": bytes_env * E / (fus * 1e-6) / 1e9 / HBM_PEAK_GBS}
                    # (the fused kernel does NOT move the env's state through HBM every step -- 2*H*W grid bytes, N*16 agent bytes and
                    # the 8-byte header of SURVEY.md 8d's per-step figure stay in LDS / registers: the fraction by the bytes it does
                    # move is the honest one to set against the headline's)
                    state_bytes = 2 * eng.H * eng.W + n_agents * 16 + 8
                    o["bytes_per_env_step_moved"] = bytes_env - state_bytes
                    o["roofline_frac_by_bytes_moved"] = (bytes_env - state_bytes) * E / (fus * 1e-6) / 1e9 / HBM_PEAK_GBS
                    if args.steps < 1000:
                        lw, _, _ = time_rollout(torch, eng, ring, 1000, 0, step0=args.warmup + args.steps, fused=True)
                        o["long_call_us_per_step"] = lw * 1e6 / 1000
                    if eng.status() != 0:
                        raise RuntimeError("device status word is non-zero")
                    return o
                finally:
                    eng.set_rollout_chains(args.chains)
            run_leg(holder, "fused_rollout", fused)

            # ---- optional leg, labelled: the same K steps with SSD_ROLLOUT_AUTO -- the library picks the form of the call (uint8
            #      observations, index action order, two steps or more: the fused kernel) and says which it picked.  Not `value`. ----
            def auto_leg():
                eng.set_rollout_chains(0)
                try:
                    aw, _, _ = time_rollout(torch, eng, ring, args.steps, args.warmup, fused="auto")
                    path = eng.rollout_path()
                    aus = aw * 1e6 / args.steps
                    if eng.status() != 0:
                        raise RuntimeError("device status word is non-zero")
                    return {"label": "NOT the headline: the same %d steps with SSD_ROLLOUT_AUTO (the library chooses the form of the call; "
                                     "`dispatch` says which form ran)" % args.steps,
                            "us_per_step": aus, "value": float(E) * n_agents * args.steps / aw, "unit": "agent-env-steps/s per GPU",
                            "roofline_frac": bytes_env * E / (aus * 1e-6) / 1e9 / HBM_PEAK_GBS, "dispatch": path}
                finally:
                    eng.set_rollout_chains(args.chains)
            run_leg(holder, "rollout_auto", auto_leg)
        # ---- optional legs WITH collectives (opt-in): north_star: "RCCL gather of obs/reward over xGMI only when a single batched
        #      tensor is requested".  The same rollout in chunks of GR steps, each chunk's observations and rewards (a) all-gathered to
        #      every rank, (b) gathered to rank 0 (the root ingests its 7 peers' shards over 7 direct links), one
        #      collective each, on a stream of their own, overlapped with the next chunk's steps.  Every rank takes part. ----
        if plain and args.gather_leg and dist is not None and not rehearsal:
            per_rank_bytes = out[0].numel() * out[0].element_size() + out[1].numel() * 4
            for mode, key in (("all", "with_gather"), ("root", "with_gather_to_root")):
                def gleg(mode=mode):
                    GK = 2 * GR
                    ensure_gather(root_only=(mode == "root"))
                    run_steps(0, GK, gather=mode)      # (warm-up: buffers touched, the communicator's channels for this size built)
                    torch.cuda.synchronize()
                    parallel.barrier(dist, local_rank)
                    torch.cuda.synchronize()
                    tg = time.perf_counter()
                    run_steps(GK, GK, gather=mode)
                    torch.cuda.synchronize()
                    gw = time.perf_counter() - tg
                    parallel.barrier(dist, local_rank)
                    tgw = torch.tensor([gw], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(tgw, op=dist.ReduceOp.MAX)
                    gw = float(tgw[0])
                    recv = per_rank_bytes * (world - 1)      # what the busiest receiver takes in per step (every rank / the root)
                    return {"label": "NOT the headline: the same rollout with every rank's observations and rewards %s (one RCCL collective "
                                     "of each per %d steps, overlapped with the next %d steps)%s"
                                     % ("all-gathered to every rank" if mode == "all" else "gathered to rank 0", GR, GR,
                                        "; one rank: the batch is already whole, nothing is copied" if world == 1 else ""),
                            "steps": GK, "ms_per_step": gw * 1e3 / GK, "value": float(E) * n_agents * GK * world / gw, "unit": "agent-env-steps/s",
                            "bytes_received_per_rank_per_step": int(recv), "receivers": world if mode == "all" else 1,
                            "xgmi_per_link_bound_us_per_step": per_rank_bytes / (XGMI_LINK_GBS * 1e9) * 1e6 if world > 1 else 0.0,
                            "xgmi_note": "every peer has a direct link to the receiver: per link and step one rank's %d bytes, whatever the world size "
                                         "(all-gather: on all %d x %d links at once; gather: on the root's %d)" % (per_rank_bytes, world, world - 1, world - 1)}
                run_leg(holder, key, gleg)
        if rank == 0 and world == 1 and not args.no_extras:
            # ---- optional leg: the step with caller-supplied actions ----
            if not args.obs_f32:
                run_leg(res, "policy_step", lambda: policy_step_leg(torch, game, amap, n_agents, E))
        if rank == 0 and world == 1 and not args.no_configs:
            # ---- optional legs: BASELINE.json's other single-GPU workloads (SURVEY.md 8d "state both": the 25x38 label; configs[2];
            #      configs[4]'s per-GPU share) and the float32-observation mode, a few hundred steps each ----
            eng.close()
            legs = [("cleanup", 4096, {}), ("harvest25x38", 4096, {}), ("cleanup48x36", 2048, {}), ("harvest", 4096, {"obs_f32": True})]
            res["configs"] = []
            for name, e, kw in legs:
                slot = {}
                run_leg(slot, "leg", lambda name=name, e=e, kw=kw: config_leg(torch, name, e, **kw))
                leg = slot.get("leg", {"error": "no result"})
                if "error" in leg:
                    leg["workload"] = "%s, %d envs%s" % (name, e, ", float32 obs" if kw.get("obs_f32") else "")
                res["configs"].append(leg)
            if os.environ.get("SSD_BENCH_FAIL_LEG", "") and "configs" in os.environ["SSD_BENCH_FAIL_LEG"].split(","):
                res["configs"] = {"error": "RuntimeError: injected failure (SSD_BENCH_FAIL_LEG)"}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            run_leg(res, "cpu_baseline", lambda: cpu_baseline(game, amap, n_agents))
    finally:
        if rank == 0:
            emit(res)                              # whatever happened after the headline: the line goes out
    if dist is not None:
        try:
            dist.destroy_process_group()
        except Exception as exc:                   # (the line is out: a teardown problem is not a failed run)
            print("bench.py: destroy_process_group: %s" % exc, file=sys.stderr)


if __name__ == "__main__":
    main()
