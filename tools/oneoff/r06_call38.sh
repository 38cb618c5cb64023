#!/bin/bash
# round 6, call 38: training step -- vectorised dropout-mask application, interpolation with the table slice in LDS
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c38_train_elementwise.txt; : > $LOG
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_point_ops.py tests/test_gpu_network.py -x -q 2>&1 | tail -3 >> $LOG
for rep in 1 2 3; do
  timeout 200 python bench.py --mode train --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['calls_ms_per_step']
print('%.2f ms per step | ' % l['ms_per_step'] + ', '.join('%s %.2f' % (k.replace('di2p_',''), v['ms']) for k, v in c.items()))" >> $LOG
done
TOP=30 timeout 200 python tools/train_call_times.py 2>&1 | grep -v amdgpu.ids | grep "apply_mask\|interpolate\|per entry" >> $LOG
cat $LOG
