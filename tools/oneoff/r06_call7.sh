#!/bin/bash
# round 6, call 7: multi-workgroup preparation, scatter without global atomics: tests + kernel times
set -u
OUT=gpurun_out; mkdir -p $OUT; ROOT=$(pwd)
LOG=$OUT/r06_c7_prepare.txt; : > $LOG
timeout 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -x 2>&1 | tail -2 >> $LOG
CASES=12 timeout 300 python tools/fuzz_solver_cull.py 2>&1 | tail -1 >> $LOG
python tools/dump_solve.py /tmp/new.npz > /dev/null 2>&1; DI2P_SOLVER_PREP_SINGLE=1 python tools/dump_solve.py /tmp/single.npz > /dev/null 2>&1
python -c "
import numpy as np
a,b=np.load('/tmp/new.npz'),np.load('/tmp/single.npz')
print('multi-workgroup preparation bit-identical to the single-workgroup kernel:', all(a[k].tobytes()==b[k].tobytes() for k in a.files))" >> $LOG 2>&1
export TMPDIR=/tmp; cd /tmp
for m in 0 1; do
  rm -rf /tmp/pp$m
  DI2P_SOLVER_PREP_SINGLE=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp$m -- python $ROOT/tools/bench_solver.py > /tmp/pp$m.log 2>&1
  f=$(find /tmp/pp$m -name "*kernel_stats.csv" | head -1)
  echo "== solver_prep_single=$m" >> $ROOT/$LOG
  [ -n "$f" ] && python - "$f" >> $ROOT/$LOG <<'PY'
import csv, sys, re
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(prep\w+|prepare_kernel)", r["Name"])
    if m:
        print("  %-24s calls %s  avg %.1f us" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3)); tot += float(r["AverageNs"]) / 1e3
print("  sum of the averages %.1f us" % tot)
PY
done
cd $ROOT; cat $LOG
