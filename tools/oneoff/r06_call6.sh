#!/bin/bash
# round 6, call 6: multi-workgroup frame preparation -- tests, batch of mixed frames (one degenerate), kernel times, headline / lone-batch latency A/B
set -u
OUT=gpurun_out; mkdir -p $OUT; ROOT=$(pwd)
LOG=$OUT/r06_c6_prepare.txt; : > $LOG
timeout 600 python -m pytest tests/test_gpu_solver.py -q -x 2>&1 | tail -4 >> $LOG
python tools/dump_solve.py /tmp/new.npz >> $LOG 2>&1; DI2P_SOLVER_PREP_SINGLE=1 python tools/dump_solve.py /tmp/single.npz >> $LOG 2>&1
python -c "
import numpy as np
a,b=np.load('/tmp/new.npz'),np.load('/tmp/single.npz')
print('multi-workgroup preparation bit-identical to the single-workgroup kernel:', all(a[k].tobytes()==b[k].tobytes() for k in a.files))" >> $LOG 2>&1
export TMPDIR=/tmp; cd /tmp
for m in 0 1; do
  rm -rf /tmp/pp$m
  DI2P_SOLVER_PREP_SINGLE=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp$m -- python $ROOT/tools/bench_solver.py > /tmp/pp$m.log 2>&1
  f=$(find /tmp/pp$m -name "*kernel_stats.csv" | head -1)
  echo "== solver_prep_single=$m (name, calls, total ns, avg ns)" >> $ROOT/$LOG
  [ -n "$f" ] && python - "$f" >> $ROOT/$LOG <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "prep" in n:
        import re
        m = re.search(r"(prep\w+|prepare_kernel)", n)
        print("  %-24s calls %s  avg %.1f us" % (m.group(1) if m else n[:24], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cd $ROOT
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f | 1-in-flight %.2f ms' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], l['latency_ms_per_batch']['one_step_in_flight']))"; }
for i in 1 2; do
  echo "multi : $(timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 6 2>/dev/null | line)" >> $LOG
  echo "single: $(DI2P_SOLVER_PREP_SINGLE=1 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 6 2>/dev/null | line)" >> $LOG
done
cat $LOG

# LM stages with batched LDS fetches: bit-identity, packed solver time, in-pipeline slope
L=$ROOT/deepi2p_amd/lib/variants/lmb/libdeepi2p_hip.so
DI2P_LIB=$L python tools/dump_solve.py /tmp/lmb.npz > /dev/null 2>&1
python -c "
import numpy as np
a,b=np.load('/tmp/new.npz'),np.load('/tmp/lmb.npz')
print('LMBATCH bit-identical to the main build:', all(a[k].tobytes()==b[k].tobytes() for k in a.files))" | tee -a $LOG
for rep in 1 2; do
  echo "main packed: $(F=128 timeout 300 python tools/bench_solver.py 2>&1 | tail -1)" | tee -a $LOG
  echo "lmb  packed: $(F=128 DI2P_LIB=$L timeout 300 python tools/bench_solver.py 2>&1 | tail -1)" | tee -a $LOG
done
PFC=2 PROF=1 DI2P_LIB=$L timeout 200 python tools/bench_solver.py 2>&1 | grep "per-sweep cycles\|LM stages\|LM cycles" | tee -a $LOG
PFC=2 PROF=1 timeout 200 python tools/bench_solver.py 2>&1 | grep "per-sweep cycles\|LM stages\|LM cycles" | tee -a $LOG
for rep in 1 2; do for v in main lmb; do
  LL=$L; [ $v = main ] && LL=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
  for r in 1 120; do
    DI2P_LIB=$LL timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 4 --restarts $r 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$v restarts $r: %.3f ms/step (%.1f frames/s)' % (l['ms_per_step'], l['value']))" | tee -a $LOG
  done
done; done
