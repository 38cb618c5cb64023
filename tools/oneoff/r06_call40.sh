#!/bin/bash
# round 6, call 40: training step at batch 8 -- the 256- and 512-channel 3x3 layers (forward + input gradient) on di2p_conv3x3_x3 with the smallest tiles
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c40_train_convx3.txt; : > $LOG
timeout 900 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -3 >> $LOG
for rep in 1 2 3; do
for x in 31 0; do
  DI2P_CONV_X3=$x timeout 200 python bench.py --mode train --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['calls_ms_per_step']
print('conv_x3=$x: %.2f ms per step | ' % (l['ms_per_step']) + ', '.join('%s %.2f' % (k.replace('di2p_',''), v['ms']) for k, v in list(c.items())[:8]))" >> $LOG
done
done
cat $LOG
