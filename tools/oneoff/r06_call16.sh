#!/bin/bash
# round 6, call 16: take_bit as two scalar instructions (-14 % scalar instructions in the walk): bit-identity, counters, full-chip time, step time at 1 / 120 restarts
set -u
OUT=gpurun_out; mkdir -p $OUT; ROOT=$(pwd); LOG=$OUT/r06_c18_scalar_trims2.txt; : > $LOG
V=${V:-tb}
python tools/dump_solve.py /tmp/main.npz > /dev/null 2>&1
DI2P_LIB=$ROOT/deepi2p_amd/lib/variants/$V/libdeepi2p_hip.so python tools/dump_solve.py /tmp/v.npz > /dev/null 2>&1
python -c "
import numpy as np
a,b=np.load('/tmp/main.npz'),np.load('/tmp/v.npz')
print('$V bit-identical to the main build:', all(a[k].tobytes()==b[k].tobytes() for k in a.files))" >> $LOG
for rep in 1 2 3; do for v in main $V; do
  L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && L=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
  echo "$v packed: $(F=128 DI2P_LIB=$L timeout 300 python tools/bench_solver.py 2>&1 | tail -1)" >> $LOG
done; done
export TMPDIR=/tmp; cd /tmp
for v in main $V; do
  L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && L=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
  rm -rf /tmp/pc_$v
  DI2P_LIB=$L timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pc_$v -- python $ROOT/tools/bench_solver.py > /tmp/pc_$v.log 2>&1
  f=$(find /tmp/pc_$v -name "*counter_collection.csv" | head -1)
  python - $v "$f" >> $ROOT/$LOG <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "solve_kernel" in r.get("Kernel_Name", "")]
ids = sorted({int(r["Dispatch_Id"]) for r in rows}); keep = set(ids[-6:])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if int(r["Dispatch_Id"]) in keep:
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("%-8s per launch: " % sys.argv[1] + "  ".join("%s %.4g" % (k, v[1] / v[0]) for k, v in sorted(agg.items())))
PY
done
cd $ROOT
for rep in 1 2 3; do for v in main $V; do
  L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && L=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
  for r in 1 120; do
    DI2P_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 4 --restarts $r 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$v restarts $r: %.3f ms/step (%.1f frames/s)' % (l['ms_per_step'], l['value']))" >> $LOG
  done
done; done
cat $LOG
