#!/bin/bash
# round 6, call 37: stride-2 input gradient per parity class -- tests, training step A/B
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c37_dgrad_s2.txt; : > $LOG
timeout 900 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -3 >> $LOG
for rep in 1 2 3; do
for d in 0 1; do
  DI2P_CONV_DGRAD_DENSE=$d timeout 200 python bench.py --mode train --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['calls_ms_per_step']
d=c.get('di2p_conv2d_dgrad')
print('dense $d: %.2f ms per step | conv2d_dgrad %s' % (l['ms_per_step'], ('%.3f ms in %d calls' % (d['ms'], d['calls'])) if d else 'below the ten largest calls (< %.2f ms)' % min(v['ms'] for v in c.values())))" >> $LOG
done
done
cat $LOG
