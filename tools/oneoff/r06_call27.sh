#!/bin/bash
# round 6, call 27: ablations of the planes-source K loop (wrong results, timing only): 1 no barrier, 2 no staging, 3 no weight-fragment loads, 4 neither
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c27_x3p_ablations.txt; : > $LOG
for v in main x3p_abl1 x3p_abl2 x3p_abl3 x3p_abl4 main; do
  echo "== $v" >> $LOG
  lib=deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && lib=deepi2p_amd/lib/libdeepi2p_hip.so
  DI2P_LIB=$PWD/$lib REPS=20 PLANES=1 timeout 300 python tools/bench_pw_x3.py 2>&1 | grep -v amdgpu.ids | grep "plain" >> $LOG
done
cat $LOG
