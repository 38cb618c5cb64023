#!/bin/bash
# round 6, call 5: the solver ALONE on a full chip (128 frames x 60 hypotheses = 7.5 rounds of workgroups: little tail): time per launch per variant
set -u
OUT=gpurun_out; mkdir -p $OUT; ROOT=$(pwd)
LOG=$OUT/r06_c5_solver_packed.txt; : > $LOG
for rep in 1 2; do
for v in ${VARS:-old main bx flat v4 rot}; do
  L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && L=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
  echo "$v: $(F=128 DI2P_LIB=$L timeout 300 python tools/bench_solver.py 2>&1 | tail -1)" >> $LOG
done
done
cat $LOG
