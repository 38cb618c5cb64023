#!/bin/bash
# round 6, call 3: what the solver variants are worth IN THE PIPELINE: step time at 1 / 120 restarts (slope = the solver's chip time per restart)
set -u
OUT=gpurun_out; mkdir -p $OUT; ROOT=$(pwd)
LOG=$OUT/r06_c4_additivity_variants.txt; : > $LOG
for rep in 1 2; do
for v in ${VARS:-main rot bx bxrot v4rot flat}; do
  L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && L=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
  for r in ${RS:-1 60 120}; do
    DI2P_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 4 --restarts $r 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$v restarts $r: %.3f ms/step (%.1f frames/s)' % (l['ms_per_step'], l['value']))" >> $LOG
  done
done
done
cat $LOG
PFC=2 PROF=1 DI2P_LIB=$ROOT/deepi2p_amd/lib/variants/v4rot/libdeepi2p_hip.so timeout 200 python tools/bench_solver.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_c4_solver_phases_v4rot.txt | grep "walk\|inside"
python tools/dump_solve.py /tmp/main.npz
for v in rot v4rot; do
  DI2P_LIB=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so python tools/dump_solve.py /tmp/$v.npz > /dev/null 2>&1
  python - $v <<'PY'
import sys, numpy as np
a, b = np.load("/tmp/main.npz"), np.load("/tmp/%s.npz" % sys.argv[1])
print("%-8s bit-identical to the main build: %s" % (sys.argv[1], all(a[k].tobytes() == b[k].tobytes() for k in ("p", "c", "it", "sw"))))
PY
done
DI2P_LIB=$ROOT/deepi2p_amd/lib/variants/v4rot/libdeepi2p_hip.so timeout 600 python -m pytest tests/test_gpu_solver.py -x -q 2>&1 | tail -3
