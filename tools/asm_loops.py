#!/usr/bin/env python3
"""Where does a kernel's spill code sit?  Parses hipcc -S output: finds the loops (backward branches) of one kernel and reports, per scratch
load/store, the innermost loop that contains it (with that loop's instruction count), so that cold spills (straight-line code) can be told
from spills inside hot loops.   python tools/asm_loops.py file.s kernel-substring"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in label_at and label_at[m.group(2)] <= i:
            loops.append((label_at[m.group(2)], i))
    def is_inst(l):
        s = l.strip()
        return bool(s) and not s.startswith((";", ".", "//")) and not s.endswith(":")
    ninst = [0]
    for l in body:
        ninst.append(ninst[-1] + (1 if is_inst(l) else 0))
    print("kernel lines %d..%d, %d instructions, %d loops" % (start, end, ninst[-1], len(loops)))
    per_loop = {}
    cold = []
    for i, l in enumerate(body):
        if "scratch_" in l:
            inner = None
            for (a, b) in loops:
                if a <= i <= b and (inner is None or (b - a) < (inner[1] - inner[0])):
                    inner = (a, b)
            if inner is None:
                cold.append((i + start + 1, l.strip()[:70]))
            else:
                per_loop.setdefault(inner, []).append((i + start + 1, l.strip()[:70]))
    print("scratch ops outside any loop: %d" % len(cold))
    for (a, b), ops in sorted(per_loop.items()):
        depth = sum(1 for (x, y) in loops if x <= a and b <= y)
        print("loop lines %d..%d (%d instr, nesting depth %d): %d scratch ops" % (a + start + 1, b + start + 1, ninst[b + 1] - ninst[a], depth, len(ops)))
        for o in ops[:40]:
            print("      %d: %s" % o)


main()
