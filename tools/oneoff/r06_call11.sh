#!/bin/bash
# round 6, call 11: pivot reciprocals reused in the LM's Cholesky solve (main) and the walk's first batches requested together (wpf), against the previous build
set -u
OUT=gpurun_out; mkdir -p $OUT; ROOT=$(pwd)
LOG=$OUT/r06_c12_chol_drainpf.txt; : > $LOG
python tools/dump_solve.py /tmp/main.npz > /dev/null 2>&1
for v in prev dpf; do
  DI2P_LIB=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so python tools/dump_solve.py /tmp/$v.npz > /dev/null 2>&1
  python - $v >> $LOG <<'PY'
import sys, numpy as np
a, b = np.load("/tmp/main.npz"), np.load("/tmp/%s.npz" % sys.argv[1])
print("%-5s bit-identical to the main build: %s" % (sys.argv[1], all(a[k].tobytes() == b[k].tobytes() for k in ("p", "c", "it", "sw"))))
PY
done
for rep in 1 2 3; do for v in prev main dpf; do
  L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && L=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
  echo "$v packed: $(F=128 DI2P_LIB=$L timeout 300 python tools/bench_solver.py 2>&1 | tail -1)" >> $LOG
done; done
for v in prev main dpf; do
  L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && L=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
  echo "$v: $(PFC=2 PROF=1 DI2P_LIB=$L timeout 200 python tools/bench_solver.py 2>&1 | grep 'per-sweep cycles\|LM stages\|inside the sweep' | tr '\n' ' ')" >> $LOG
done
for rep in 1 2; do for v in prev main dpf; do
  L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && L=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
  for r in 1 120; do
    DI2P_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 4 --restarts $r 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$v restarts $r: %.3f ms/step (%.1f frames/s)' % (l['ms_per_step'], l['value']))" >> $LOG
  done
done; done
cat $LOG
