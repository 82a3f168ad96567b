#!/usr/bin/env python3
"""development aid: the loops of one kernel's assembly (backward branches), with instruction counts by class
usage: tools/asm_loops.py <file.s> [min_instructions]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
minn = int(sys.argv[2]) if len(sys.argv) > 2 else 40
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.match(r"^\s*(s_cbranch_\w+|s_branch)\s+(\.LBB[0-9_]+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i))
def cls(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): return "vmem_ld"
    if op.startswith("global_store") or op.startswith("global_atomic") or op.startswith("flat_store"): return "vmem_st"
    if op.startswith("ds_"): return "lds"
    return "other"
for a, b in sorted(set(loops)):
    c = collections.Counter()
    ops = collections.Counter()
    for l in lines[a:b + 1]:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if m and not l.strip().startswith(";") and not l.strip().startswith("."):
            c[cls(m.group(1))] += 1
            ops[m.group(1)] += 1
    n = sum(c.values())
    if n >= minn:
        print("loop lines %d-%d: %d instructions %s" % (a, b, n, dict(c)))
        print("    top:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))
