#!/usr/bin/env python3
"""Static instruction budget of one kernel BY SOURCE REGION (hipcc -S -gline-tables-only output: every instruction carries a .loc).

    python tools/asm_by_source.py file.s kernel-substring source.hip [regions.txt]

Buckets the kernel's instructions (valu32 / valu64 / salu / branch / lds / vmem / wait) by the source line of their .loc entry, grouped into
the regions given as `first-last name` lines (default: the phases of csrc/solver.hip, resolved from `// @region name` ... markers is NOT used --
the ranges are found from function / lambda names below).  A line table only names the INNERMOST inlined line, which is what a phase budget wants.
Static counts: weight them with the trip counts the PROFILE instantiation reports (tools/bench_solver.py)."""
import re
import sys


def kind(s):
    if s.startswith("v_"):
        return "valu64" if "f64" in s else "valu32"
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


def solver_regions(src):
    """(first, last, name) from the function heads of solver.hip"""
    lines = open(src).read().splitlines()
    heads = [
        (r"void project\(", "project (rotate, reciprocal, pixel)"),
        (r"double ln_pos\(", "ln_pos"),
        (r"^struct LogProd", "LogProd (cost product)"),
        (r"void eval_active\(", "eval_active (phase B: rows, corrector, normal equations)"),
        (r"void make_pre32\(", "make_pre32 (set-up)"),
        (r"void prefilter32\(", "prefilter32 (per-point fp32 test)"),
        (r"void make_box_abs\(", "make_box_abs (set-up)"),
        (r"int cluster_status\(", "cluster_status (box test)"),
        (r"void wave_min4_nonneg\(", "wave_min4_nonneg (DPP minima)"),
        (r"void sweep_clusters\(", "sweep_clusters: head"),
        (r"auto drain = ", "sweep_clusters: drain loop around eval_active"),
        (r"auto exact_active = ", "sweep_clusters: exact fp64 test"),
        (r"const bool use_pre = ", "sweep_clusters: cluster-test round + cache look-up"),
        (r"// ---- phase I\.", "sweep_clusters: phase I guard-only walk"),
        (r"if \(mA\) \{", "sweep_clusters: phase I classification walk"),
        (r"what phase I found, one entry per lane", "sweep_clusters: cache store"),
        (r"// ---- phase II", "sweep_clusters: phase II appends"),
        (r"^template <int CTRL> __device__ __forceinline__ int dpp_int", "dpp helpers (wave totals)"),
        (r"void sweep\(", "sweep: set-up + wave totals"),
        (r"bool chol_solve_inplace\(", "LM: cholesky"),
        (r"void plus_proj\(", "LM: projections / gradient norm"),
        (r"double poly_eval\(", "LM: polynomial minimiser (wave)"),
        (r"^struct LsSample", "LM: interpolating fit"),
        (r"^struct Bounds", "LM: state"),
        (r"void lm_begin_iteration\(", "LM: begin iteration"),
        (r"^enum \{ ACT_NONE", "LM: finish / trial / apply"),
        (r"void lm_poly_wave\(", "LM: poly wave glue"),
        (r"int lm_decide\(", "LM: decide"),
        (r"^struct SolveArgs", "kernel body (loop glue, combine, make_rot)"),
        (r"void angle_axis_to_R\(", "after"),
    ]
    found = []
    for pat, name in heads:
        for i, l in enumerate(lines):
            if re.search(pat, l):
                found.append((i + 1, name))
                break
        else:
            raise SystemExit("region head not found: " + pat)
    found.sort()
    regs = [(1, found[0][0] - 1, "helpers above project (make_rot, fast_rcp, lm_div/sqrt, prepare)")]
    for (a, name), (b, _) in zip(found, found[1:] + [(len(lines) + 1, "")]):
        regs.append((a, b - 1, name))
    return regs


def main():
    path, pat, src = sys.argv[1], sys.argv[2], sys.argv[3]
    regs = solver_regions(src)
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # file numbers of the source
    fileno = set()
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m and (m.group(2).endswith(src.split("/")[-1]) or (m.group(3) or "").endswith(src.split("/")[-1])):
            fileno.add(int(m.group(1)))
    cur = None
    cnt = {}
    other_files = {}
    for l in lines[start:end + 1]:
        s = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        k = kind(s.split()[0])
        if cur is None or cur[0] not in fileno:
            name = "(other files: libm / builtins)"
        else:
            name = next((n for a, b, n in regs if a <= cur[1] <= b), "?")
        c = cnt.setdefault(name, {})
        c[k] = c.get(k, 0) + 1
    cols = ["valu32", "valu64", "salu", "branch", "lds", "vmem", "wait", "other"]
    print("%-72s %s" % ("region (static instruction counts)", " ".join("%7s" % c for c in cols)))
    tot = {}
    order = [n for _, _, n in regs] + ["(other files: libm / builtins)", "?"]
    for name in order:
        if name not in cnt:
            continue
        c = cnt[name]
        print("%-72s %s" % (name, " ".join("%7d" % c.get(k, 0) for k in cols)))
        for k in cols:
            tot[k] = tot.get(k, 0) + c.get(k, 0)
    print("%-72s %s" % ("total", " ".join("%7d" % tot.get(k, 0) for k in cols)))


main()
