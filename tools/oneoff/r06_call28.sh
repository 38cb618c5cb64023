#!/bin/bash
# round 6, call 28: split pointwise weights plane-major ([Kp/8][3][Mp] x 16 B: contiguous fragment requests) against the interleaved layout
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c28_x3_wlayout.txt; : > $LOG
timeout 900 python -m pytest tests/test_gpu_contractions.py tests/test_gpu_network.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3 >> $LOG
for v in main x3_wl0 main x3_wl0; do
  echo "== $v" >> $LOG
  lib=deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && lib=deepi2p_amd/lib/libdeepi2p_hip.so
  DI2P_LIB=$PWD/$lib REPS=20 PLANES=1 timeout 300 python tools/bench_pw_x3.py 2>&1 | grep -v amdgpu.ids | grep "plain\|gmax" >> $LOG
  DI2P_LIB=$PWD/$lib timeout 200 python tools/bench_pw_planes.py 2>&1 | grep -v amdgpu.ids | tail -2 >> $LOG
done
cat $LOG
