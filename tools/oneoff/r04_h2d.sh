#!/bin/bash
for s in 8 12 16; do
GPU_MAX_HW_QUEUES=32 timeout 300 python bench.py --no-cpu-baseline --steps 32 --warmup 8 --streams $s 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('streams $s: resident %.1f  with_h2d %.1f  (%.3f)' % (l['value'], l['value_with_h2d'], l['value_with_h2d']/l['value']))"
done
