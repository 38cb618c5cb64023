#!/bin/bash
# round 6, call 13: hardware queues 16 against 32 at the default 8 streams, with and without the H2D copies (alternating)
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r06_c13_hw_queues.txt; : > $LOG
for rep in 1 2 3; do for q in 16 32; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --steps 48 --warmup 8 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('queues $q: resident %.1f fps, with h2d %.1f fps, %.1f ms per batch' % (l['value'], l['value_with_h2d'], l['latency_ms_per_batch']['streams_8']))" >> $LOG
done; done
cat $LOG
