#!/bin/bash
# Wave-instruction counts of every kernel of one 32-frame step (GPU box; --pmc passes with kernel-trace only): how much VALU issue time
# and how much matrix-pipe time a step needs in total, next to the measured 8-stream step time.   tools/prof_step_instructions.sh r03
set -u
TAG=${1:-r06}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/si$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/si$i -- python $ROOT/bench.py --no-cpu-baseline --no-h2d-pass --no-graph --streams 1 --steps 4 --warmup 2 > /tmp/si$i.log 2>&1
  f=$(find /tmp/si$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f /tmp/si_pass$i.csv || { echo "pass $i ($set) failed"; tail -3 /tmp/si$i.log; }
done
python - $OUT/${TAG}_step_instructions.txt <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in sorted(glob.glob("/tmp/si_pass*.csv")):
    first = True
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if "pass1" in f and r["Counter_Name"] == "SQ_INSTS_VALU": calls[k] += 1
# bench.py ran 2 warm-up + 4 timed + 2 serial-pass steps + 1 capture-free warm-up; normalise by the solver's launch count
n_steps = max([1] + [v for k, v in calls.items() if k.startswith("solve_kernel")])
out = open(sys.argv[1], "w")
def p(*a):
    s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
p("wave-instruction counts per 32-frame step (%d steps counted; PMC passes serialise the kernels, the counts do not depend on that)" % n_steps)
p("%-62s %8s %10s %10s %10s %9s" % ("kernel", "calls", "VALU e6", "MFMA e6", "SALU e6", "LDS e6"))
S = collections.Counter()
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    v, m, s, l = (c.get(x, 0) / n_steps / 1e6 for x in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS"))
    S["v"] += v; S["m"] += m; S["s"] += s; S["l"] += l; S["mops"] += c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) / n_steps; S["gui"] += c.get("GRBM_GUI_ACTIVE", 0) / n_steps
    if v > 0.5: p("%-62s %8.1f %10.1f %10.1f %10.1f %9.1f" % (k, calls[k] / n_steps, v, m, s, l))
p("total: VALU %.0fe6 (of which MFMA %.0fe6), SALU %.0fe6, LDS %.0fe6 wave-instructions per step" % (S["v"], S["m"], S["s"], S["l"]))
for ghz in (2.1, 2.4):
    p("at %.1f GHz on 1024 SIMDs: VALU issue (4 cycles per non-MFMA instruction) %.2f ms; matrix pipe (MOPS_F32 x 512 flop... see DESIGN) ; GUI-active %.2f ms serial" % (
        ghz, (S["v"] - S["m"]) * 1e6 * 4 / 1024 / (ghz * 1e9) * 1e3, S["gui"] / 8 / (ghz * 1e9) * 1e3))
p("SQ_INSTS_VALU_MFMA_MOPS_F32 per step: %.3e" % S["mops"])
PY
