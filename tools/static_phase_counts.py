#!/usr/bin/env python3
"""Static instruction mix per phase of a step-kernel instantiation (diagnostic build asm, phases delimited by the
s_memtime stamps; source lines in [--skip a:b] ranges, e.g. the slow move path, are left out).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -g -DSSD_STAMPS -S --cuda-device-only -o /tmp/k_st.s csrc/ssd_kernels.hip
    python tools/static_phase_counts.py /tmp/k_st.s [mangled-name-substring] [--skip 365:445 ...]
"""
import re
import sys

NAMES = ["pre", "load state", "actions+move", "consume+occ", "beams", "respawn", "write-back", "overlay", "-", "obs", "post"]


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer") or op.startswith("s_memtime") or op.startswith("s_memrealtime"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
        return "vmem"
    return "other"


def main():
    path = sys.argv[1]
    sub = "ILi0ELi0ELb0ELi5ELb1ELb1E"
    skips = []
    args = sys.argv[2:]
    i = 0
    while i < len(args):
        if args[i] == "--skip":
            a, b = args[i + 1].split(":")
            skips.append((int(a), int(b)))
            i += 2
        else:
            sub = args[i]
            i += 1
    lines = open(path).read().split("\n")
    start = [k for k, l in enumerate(lines) if l.startswith("_ZN3ssd14ssd_env_kernel") and sub in l and ":" in l][0]
    seg, cur = 0, None
    counts = {}
    slow = {}
    for l in lines[start:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"\s*\.loc\s+\d+\s+(\d+)", l)
        if m:
            cur = int(m.group(1))
            continue
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_memtime":
            seg += 1
            continue
        cls = classify(op)
        if any(a <= (cur or 0) <= b for a, b in skips):
            slow[cls] = slow.get(cls, 0) + 1
            continue
        d = counts.setdefault(seg, {})
        d[cls] = d.get(cls, 0) + 1
        if op in ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32"):
            d["quarter"] = d.get("quarter", 0) + 1
    cols = ["valu", "quarter", "salu", "lds", "vmem", "smem", "wait", "branch"]
    print("%-14s" % "phase" + "".join("%9s" % c for c in cols))
    tot = {}
    for sg in sorted(counts):
        nm = NAMES[sg] if sg < len(NAMES) else str(sg)
        print("%-14s" % nm + "".join("%9d" % counts[sg].get(c, 0) for c in cols))
        for c in cols:
            tot[c] = tot.get(c, 0) + counts[sg].get(c, 0)
    print("%-14s" % "total" + "".join("%9d" % tot.get(c, 0) for c in cols))
    if slow:
        print("%-14s" % "(skipped)" + "".join("%9d" % slow.get(c, 0) for c in cols))


if __name__ == "__main__":
    main()
