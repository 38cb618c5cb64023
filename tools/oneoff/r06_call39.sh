#!/bin/bash
# round 6, call 39: training step -- forward / input-gradient contractions of the big point layers on the bf16x3 kernel (weights split per call)
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c39_train_x3.txt; : > $LOG
timeout 900 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -3 >> $LOG
for rep in 1 2; do
for x in 1 0; do
  DI2P_PW_X3=$x timeout 200 python bench.py --mode train --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['calls_ms_per_step']
print('pw_x3=$x: %.2f ms per step | loss %s | ' % (l['ms_per_step'], l['loss_first_last']) + ', '.join('%s %.2f' % (k.replace('di2p_',''), v['ms']) for k, v in list(c.items())[:6]))" >> $LOG
done
done
cat $LOG
