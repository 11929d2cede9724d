#!/usr/bin/env python3
"""Static check of the kernels: no vector-memory instruction may be written as inline asm.

Round 3's coherent kernels fetched and stored an env's grid with `asm volatile("global_load_dwordx4 ... sc1")` -- loads the
compiler believes complete when the statement ends.  A load under a divergent branch lost a grid piece, loads with much code
before their wait cost a kernel its registers (a memory fault).  They are now the buffer builtins (__builtin_amdgcn_raw_buffer_
load / store_b128 / _b96 with the cache policy in the aux operand), which the compiler tracks.  This check keeps it that way:
  1. the kernel SOURCE holds no asm statement with a global_ / buffer_ / flat_ / scratch_ load, store or atomic in it;
  2. the ISA (make -C csrc asm -> ssd_kernels.s) of every coherent step kernel (template argument COH = true) fetches its grid
     with `buffer_load_dwordx4 ... sc1` and writes it back with `buffer_store_dwordx4 ... sc1`, and no kernel contains a
     `global_load_dwordx4 ... sc1` (the old asm form; the compiler itself never emits a 16-byte agent-scope global load).
Exit status 1 on a violation."""
import re
import sys

src_path = sys.argv[1] if len(sys.argv) > 1 else "sequential_social_dilemma_games_amd/csrc/ssd_kernels.hip"
isa_path = sys.argv[2] if len(sys.argv) > 2 else "sequential_social_dilemma_games_amd/csrc/ssd_kernels.s"
bad = 0
src = open(src_path).read()
code = re.sub(r"//[^\n]*", "", src)                       # (comments may, and do, talk about the old form)
for m in re.finditer(r"\basm\s*(volatile)?\s*\(\s*((?:\"(?:[^\"\\]|\\.)*\"\s*)+)", code):
    text = m.group(2)
    if re.search(r"\b(global|buffer|flat|scratch)_(load|store|atomic)", text):
        bad += 1
        print("inline-asm vector memory instruction in %s: %s" % (src_path, text.strip()[:100]))
fn, kernels = None, {}
for raw in open(isa_path):
    l = raw.strip()
    m = re.match(r"(_ZN3ssd14ssd_env_kernel\S+):", l)
    if m:
        fn = m.group(1)
        kernels[fn] = {"ld": 0, "st": 0, "old": 0}
        continue
    if fn is None or not l or l.startswith((";", ".")):
        continue
    if re.match(r"buffer_load_dwordx[234] .* sc1", l):
        kernels[fn]["ld"] += 1
    if re.match(r"buffer_store_dwordx4 .* sc1", l):
        kernels[fn]["st"] += 1
    if re.match(r"global_(load|store)_dwordx4 .* sc1", l):
        kernels[fn]["old"] += 1
coh = 0
for fn, k in kernels.items():
    # ssd_env_kernel<GAME, MODE, F32, NA, STD, FAST, COH, ACTS>: ILi<GAME>ELi<MODE>ELb<F32>ELi<NA>ELb<STD>ELi<FAST>ELb<COH>ELb<ACTS>E
    m = re.match(r"_ZN3ssd14ssd_env_kernelILi(\d)ELi(\d)ELb(\d)ELi(\d+)ELb(\d)ELi(\d)ELb(\d)ELb(\d)E", fn)
    if k["old"]:
        bad += 1
        print("%s: %d 16-byte agent-scope GLOBAL loads / stores (the old inline-asm form)" % (fn[:70], k["old"]))
    if m and m.group(7) == "1":
        coh += 1
        if not k["ld"] or not k["st"]:
            bad += 1
            print("%s: a coherent kernel without sc1 buffer loads (%d) / stores (%d)" % (fn[:70], k["ld"], k["st"]))
print("kernels: %d, coherent: %d, violations: %d" % (len(kernels), coh, bad))
sys.exit(1 if bad or not coh else 0)
