#!/bin/bash
# round 6, call 29: planes-source K loop -- weight fragments from an L1-resident range (diagnostic, wrong results) and requested a whole K-step ahead
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c29_x3p_weights.txt; : > $LOG
timeout 900 python -m pytest tests/test_gpu_contractions.py tests/test_gpu_network.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3 >> $LOG
DI2P_LIB=$PWD/deepi2p_amd/lib/variants/x3p_abl6/libdeepi2p_hip.so timeout 900 python -m pytest tests/test_gpu_contractions.py -x -q -k planes 2>&1 | tail -3 >> $LOG
for v in main x3p_abl5 x3p_abl6 main x3p_abl6; do
  echo "== $v" >> $LOG
  lib=deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && lib=deepi2p_amd/lib/libdeepi2p_hip.so
  DI2P_LIB=$PWD/$lib REPS=20 PLANES=1 timeout 300 python tools/bench_pw_x3.py 2>&1 | grep -v amdgpu.ids | grep "plain" >> $LOG
done
cat $LOG
