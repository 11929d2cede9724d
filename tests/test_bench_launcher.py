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
