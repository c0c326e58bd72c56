#!/usr/bin/env python3
"""dev tool (no GPU needed): compile one fused-kernel source to ISA and rank its basic blocks by *exposed* memory waits.

An exposed wait = an `s_waitcnt vmcnt(0)` issued within a few instructions of the load it waits for (the latency of that load
is not overlapped with anything), or an `lgkmcnt(0)` right behind a `ds_read` / `ds_bpermute`.  hipcc produces these when it
sinks a load next to its use to shorten a live range, when a load sits under a per-lane condition (one exec-masked branch per
element), or when a conditional prefetch makes the outstanding-load count unknown at a join.  Each one costs a full L2 / HBM
(or LDS) round trip per execution; this round they were worth 15 % of the backward kernel.

usage: tools/isa_waits.py mop_amd/csrc/edgewise_fused_bwd.hip [kernel-name-substring] [-DMOPK_INST_NT=7 -DMOPK_INST_DK=64 ...]
"""
import os, re, subprocess, sys, tempfile

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
fused = "edgewise_fused" in os.path.basename(src)          # those are built per (NT, DK) instantiation and with relaxed fp
defs = [a for a in sys.argv[2:] if a.startswith("-")] or (["-DMOPK_INST_NT=7", "-DMOPK_INST_DK=64"] if fused and "16" not in os.path.basename(src) else [])
out = os.path.join(tempfile.gettempdir(), "isa_waits.s")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src] + defs
if fused:
    cmd += ["-ffast-math", "-fno-finite-math-only"]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l]
if not starts:
    sys.exit("no kernel matches")
start = starts[0]
end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
print("kernel:", lines[start].split(":")[0])
blocks, cur, name = [], [], "entry"
for l in (x.strip() for x in lines[start:end]):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); name, cur = m.group(1), []
    elif l and not l.startswith((";", ".")):
        cur.append(l)
blocks.append((name, cur))
idx = {n: i for i, (n, b) in enumerate(blocks)}
span = {}
for i, (n, b) in enumerate(blocks):
    for x in b:
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", x)
        if m and m.group(1) in idx and idx[m.group(1)] <= i:
            j = idx[m.group(1)]
            for k in range(j, i + 1):
                if k not in span or (i - j) < span[k][0]:
                    span[k] = (i - j, j, i)
rows = []
for i, (n, b) in enumerate(blocks):
    ev = el = 0
    lastv = lastl = -99
    for q, x in enumerate(b):
        if x.startswith(("global_load", "scratch_load", "buffer_load")):
            lastv = q
        if x.startswith(("ds_read", "ds_bpermute", "ds_swizzle")):
            lastl = q
        if x.startswith("s_waitcnt"):
            if "vmcnt(0)" in x and q - lastv <= 6:
                ev += 1
            if "lgkmcnt(0)" in x and q - lastl <= 2:
                el += 1
    if ev or el > 8:
        rows.append((ev * 6 + el, ev, el, i, n, len(b), sum(1 for x in b if x.startswith("v_mfma")), span.get(i, (None,))[0]))
rows.sort(reverse=True)
print("score  vm-exposed lgkm-exposed  block#  label  instrs  mfma  innermost-loop-span(blocks)")
for r in rows[:30]:
    print("%5d  %4d %4d  %4d  %-12s %5d %4d  %s" % r)
