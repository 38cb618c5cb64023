#!/bin/bash
# round 6, call 14: train-mode BatchNorm with the finalize step inside the elementwise kernels: tests, bit-identity of a step's gradients, step time A/B
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r06_c21_train_dgrad_filter.txt; : > $LOG
timeout 900 python -m pytest tests/test_gpu_training.py -q -x 2>&1 | tail -3 >> $LOG
for rep in 1 2 3; do for m in 0; do
  DI2P_BN_UNFUSED=$m timeout 300 python bench.py --mode train --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('bn_unfused=$m: %.2f ms per step (%.0f frames/s)' % (l['ms_per_step'], l['value']))" >> $LOG
done; done
cat $LOG
