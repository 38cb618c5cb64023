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


if not (len(sys.argv) > 3 and sys.argv[3] == '--mix'):
    main()


def mix(path, pat):
    """instruction mix of every loop: python tools/asm_loops.py file.s kernel --mix"""
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    label_at = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in label_at and label_at[m.group(2)] <= i:
            loops.append((label_at[m.group(2)], i))
    def kind(l):
        s = l.strip().split()[0] if l.strip() else ""
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            return None
        if s.startswith("v_"):
            if "f64" in s:
                return "valu64"
            return "valu"
        if s.startswith("s_waitcnt"):
            return "wait"
        if s.startswith(("s_cbranch", "s_branch")):
            return "branch"
        if s.startswith("s_"):
            return "salu"
        if s.startswith("ds_"):
            return "lds"
        if s.startswith(("global_", "buffer_", "scratch_", "flat_")):
            return "vmem"
        return "other"
    print("%-22s %6s %6s %6s %6s %5s %5s %5s %5s" % ("loop (asm lines)", "instr", "valu32", "valu64", "salu", "brnch", "lds", "vmem", "wait"))
    for (a, b) in sorted(set(loops)):
        c = {}
        for l in body[a:b + 1]:
            k = kind(l)
            if k:
                c[k] = c.get(k, 0) + 1
        tot = sum(c.values())
        if tot < 25:
            continue
        print("%-22s %6d %6d %6d %6d %5d %5d %5d %5d" % ("%d..%d" % (a + start + 1, b + start + 1), tot, c.get("valu", 0), c.get("valu64", 0), c.get("salu", 0),
                                                      c.get("branch", 0), c.get("lds", 0), c.get("vmem", 0), c.get("wait", 0)))


if len(sys.argv) > 3 and sys.argv[3] == "--mix":
    mix(sys.argv[1], sys.argv[2])
