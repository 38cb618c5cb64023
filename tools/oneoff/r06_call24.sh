#!/bin/bash
# round 6, call 24: the failing planes test in full + K sweep of the planes-source kernel beside the fp32-source kernel
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c24_pw_planes.txt; : > $LOG
timeout 600 python -m pytest tests/test_gpu_contractions.py -q -k planes 2>&1 | grep -v "^$" | tail -60 >> $LOG
echo "== fp32 source" >> $LOG
REPS=20 timeout 300 python tools/bench_pw_x3.py 2>&1 | grep -v amdgpu.ids >> $LOG
echo "== planes source" >> $LOG
REPS=20 PLANES=1 timeout 300 python tools/bench_pw_x3.py 2>&1 | grep -v amdgpu.ids >> $LOG
cat $LOG
