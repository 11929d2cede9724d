"""bench.py --gpus N must start N ranks by itself (the driver's N = 1 command is a plain `python bench.py`; VERDICT r01 #2:
it used to run ONE rank and report n_gpus 1).  --dry-run is the launcher and the sharding without GPU or engine: gloo ranks
under torch.distributed.run, started as a child of the process the user launched."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          env=env, timeout=300)


def test_gpus_2_starts_two_ranks_with_contiguous_shards():
    p = _run(["--gpus", "2", "--dry-run", "--envs", "4096"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # the contract: ONE JSON line on stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks"] == 2
    assert res["shards"] == [[0, 4096], [4096, 4096]]  # (start, count) of rank 0, rank 1: train_moa.py:127-128's workers
    # a rank of a process group holds the HIP runtime to two hardware queues (the library then keeps to two of its own: a
    # process has about four before the hardware scheduler time-slices them -- DESIGN.md, "Dispatch")
    assert res["gpu_max_hw_queues"] == "2"


def test_gpus_8_dry_run_is_eight_ranks_eight_shards_one_line():
    """The shape of the driver's scaling run (N = 8: one rank per GPU, weak scaling, shards (r * E, E)) on the CPU: eight gloo
    ranks under torch.distributed.run, the legs that need no GPU included, ONE line from rank 0."""
    p = _run(["--gpus", "8", "--dry-run", "--dry-run-legs", "--envs", "4096"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["ranks"] == 8
    assert res["shards"] == [[r * 4096, 4096] for r in range(8)]
    assert res["gpu_max_hw_queues"] == "2" and res["cpu_baseline"]["value"] > 0


def test_single_rank_needs_no_launcher():
    p = _run(["--gpus", "1", "--dry-run", "--envs", "100"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    res = json.loads(p.stdout.decode())
    assert res["shards"] == [[0, 100]]
    assert res["gpu_max_hw_queues"] is None            # a plain single process leaves the runtime's default alone


def test_rank_count_mismatch_is_an_error_not_a_warning():
    """Started as a rank of a 1-rank job but asked for 2 GPUs: refuse (it used to warn and report the 1-GPU figure)."""
    p = _run(["--gpus", "2", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert b"ranks" in p.stderr


def test_a_failing_optional_leg_does_not_cost_the_headline():
    """bench.py prints its one line whatever happens in an optional leg.  The legs run inside run_leg(): an
    exception (SSD_BENCH_FAIL_LEG injects one) lands as {"error": ...} in that leg's slot, the other legs run, the line goes out,
    exit status 0.  Here on the legs that need no GPU (--dry-run --dry-run-legs); tests/test_multiprocess_gpu.py does the same
    with the real legs on the GPU box."""
    p = _run(["--gpus", "1", "--dry-run", "--dry-run-legs", "--envs", "64"], env_extra={"SSD_BENCH_FAIL_LEG": "configs"})
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["shards"] == [[0, 64]]                                  # the "headline" of a dry run
    assert "injected failure" in res["configs"]["error"]
    assert res["cpu_baseline"]["value"] > 0 and res["cpu_baseline"]["kind"] == "port"   # the leg after the failing one still ran
    assert b"optional leg 'configs' failed" in p.stderr
    p = _run(["--gpus", "1", "--dry-run", "--dry-run-legs", "--envs", "64"], env_extra={"SSD_BENCH_FAIL_LEG": "cpu_baseline,configs"})
    res = json.loads(p.stdout.decode())
    assert p.returncode == 0 and "error" in res["cpu_baseline"] and "error" in res["configs"] and res["n_gpus"] == 1


def test_run_leg_catches_system_exit_and_memory_error():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    res = {"value": 1.0}

    def boom(exc):
        def f():
            raise exc
        return f
    assert bench.run_leg(res, "a", boom(SystemExit("status word"))) is False
    assert bench.run_leg(res, "b", boom(MemoryError("hipMalloc"))) is False
    assert bench.run_leg(res, "c", lambda: {"ok": 1}) is True
    assert res["value"] == 1.0 and "SystemExit" in res["a"]["error"] and "MemoryError" in res["b"]["error"] and res["c"] == {"ok": 1}
