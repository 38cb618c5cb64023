#!/bin/bash
# round 6, call 41: training step -- 128 x 128 tiles for the big strided weight gradients
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c41_train_rc128.txt; : > $LOG
timeout 900 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -3 >> $LOG
DI2P_RC_TILE64=1 timeout 900 python -m pytest tests/test_gpu_training.py -x -q -k "linear or full" 2>&1 | tail -1 >> $LOG
for rep in 1 2 3; do
for x in 0 1; do
  DI2P_RC_TILE64=$x timeout 200 python bench.py --mode train --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['calls_ms_per_step']
print('rc_tile64=$x: %.2f ms per step | ' % (l['ms_per_step']) + ', '.join('%s %.2f' % (k.replace('di2p_',''), v['ms']) for k, v in list(c.items())[:8]))" >> $LOG
done
done
TOP=12 timeout 200 python tools/train_call_times.py bmm_rc 2>&1 | grep -v amdgpu.ids >> $LOG
cat $LOG
