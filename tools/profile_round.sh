#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>
# Writes gpurun_out/<tag>_*.csv (kernel-trace statistics of the bench command, default streams ("pipelined") and serial; separate --pmc passes
# for the solver and index_max -- counters are never combined with other trace domains).  Copy what should be judged into profiles/.
set -u
TAG=${1:-r06}
WHAT=${2:-all}          # stats | pmc | all
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT/prof_$TAG
export TMPDIR=/tmp
cd /tmp
stats() {  # name, args...
  local name=$1; shift
  rm -rf $OUT/prof_$TAG/$name
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/$name -- python $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/prof_$TAG/$name.log 2>&1
  f=$(find $OUT/prof_$TAG/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_bench_kernel_stats_$name.csv
  grep '^{' $OUT/prof_$TAG/$name.log | tail -1 > $OUT/${TAG}_bench_line_$name.json
}
if [ "$WHAT" != "pmc" ]; then
stats pipelined --no-h2d-pass
stats serial --streams 1 --no-h2d-pass
fi
[ "$WHAT" = "stats" ] && { ls -la $OUT | grep ${TAG}_; exit 0; }
pmc() {  # counter, tool, name [, env assignment]
  local c=$1 tool=$2 name=$3
  rm -rf $OUT/prof_$TAG/pmc_${name}_$c
  env ${4:-_X=1} timeout 180 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/prof_$TAG/pmc_${name}_$c -- python $ROOT/tools/$tool > $OUT/prof_$TAG/pmc_${name}_$c.log 2>&1
  f=$(find $OUT/prof_$TAG/pmc_${name}_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $OUT/${TAG}_pmc_${name}_$c.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r.get("Kernel_Name", "")[:120], r.get("Counter_Name", ""))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r.get("Counter_Value", 0))
with open(sys.argv[2], "w") as fh:
    fh.write("kernel,counter,launches,sum,mean_per_launch\n")
    for (k, c), (n, s) in agg.items():
        fh.write('"%s",%s,%d,%.6g,%.6g\n' % (k, c, n, s, s / n))
PY
}
# (the solver's FETCH_SIZE / WRITE_SIZE are taken by tools/prof_solver_counters.sh together with its other counters: ONE source, the file
#  bench.py's `counters_file` names)
for c in FETCH_SIZE WRITE_SIZE; do pmc $c bench_conv.py conv; done
for c in FETCH_SIZE WRITE_SIZE; do pmc $c bench_index_max.py index_max_C64 CS=64; pmc $c bench_index_max.py index_max_C32 CS=32; done
ls -la $OUT | grep ${TAG}_
