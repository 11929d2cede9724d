"""First line of every experiment / evidence log: which library ran, and under which SSD_* settings.

A failing log under gpurun_out/ must identify itself: the product library never fails its parity runs, the deliberate
unsafe experiments (fences removed, state left dirty in L2, ...) do -- VERDICT r02 #8.  `python tools/_label.py` prints the
line; tools import label()."""
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def label_line(what=""):
    pkg = os.path.join(REPO, "sequential_social_dilemma_games_amd")
    path = os.environ.get("SSD_LIB_PATH") or os.path.join(pkg, "libssd_hip.so")
    try:
        sha = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        sha = "missing"
    kind = "PRODUCT library" if os.path.abspath(path) == os.path.join(pkg, "libssd_hip.so") else "NOT the product library (test-hook / diagnostic / experiment build)"
    knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith(("SSD_", "GPU_MAX_HW_QUEUES", "SOAK_")))
    return "# %s lib=%s sha256/16=%s [%s] env: %s" % (what or os.path.basename(sys.argv[0]), os.path.relpath(path, REPO), sha, kind, knobs or "(no SSD_* variables)")


def label(what=""):
    print(label_line(what), flush=True)


if __name__ == "__main__":
    label(" ".join(sys.argv[1:]))
