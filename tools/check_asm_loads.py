#!/usr/bin/env python3
"""Static check of the kernels' ISA (make -C csrc asm -> ssd_kernels.s): the destination registers of the inline-asm loads (the
agent-scope 16-byte loads the compiler cannot see as outstanding) must not be read or written by any instruction between the load
and the s_waitcnt vmcnt(0) that covers it -- the compiler believes they are written when the asm statement ends, and is free to
copy or reuse them (which is how a 48 x 36 grid once lost its third piece: a load under a divergent branch).  Exit status 1 if
any such use is found."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "sequential_social_dilemma_games_amd/csrc/ssd_kernels.s"
fn, pending, bad, loads = None, {}, 0, 0
for i, raw in enumerate(open(path), 1):
    l = raw.strip()
    m = re.match(r"(_ZN3ssd\S+):", l)
    if m:
        fn, pending = m.group(1), {}
        continue
    if not l or l.startswith((";", ".")):
        continue
    m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off sc1", l)
    if m:
        pending[(int(m.group(1)), int(m.group(2)))] = i
        loads += 1
        continue
    if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
        pending = {}
        continue
    if l.startswith(("s_endpgm", "s_setpc")):
        pending = {}
    if pending:
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", l):
            regs.update(range(int(a), int(b) + 1))
        regs.update(int(x) for x in re.findall(r"\bv(\d+)\b", l))
        for (a, b), ln in pending.items():
            if any(a <= r <= b for r in regs):
                bad += 1
                print("%s\n  line %d: %s   <- touches v[%d:%d] of the load at line %d" % ((fn or "?")[:90], i, l, a, b, ln))
print("asm loads checked: %d, uses before their wait: %d" % (loads, bad))
sys.exit(1 if bad else 0)
