"""The C restatement under AddressSanitizer + UBSan on a crowded rollout (CPU build only: GPU sanitizers
are not available on the pool)."""
import os
import subprocess

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", ORACLE, "asan_driver"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(ORACLE, "asan_driver")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "game 0 ok" in r.stdout and "game 1 ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
