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
PFC=2 PROF=1 DI2P_LIB=$ROOT/deepi2p_amd/lib/variants/v3/libdeepi2p_hip.so timeout 200 python tools/bench_solver.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_c4_solver_phases_v3.txt | grep "walk\|inside"
