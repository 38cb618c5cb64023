#!/bin/bash
# headline against the number of streams / hardware queues (GPU box)
for s in 4 6 8 10 12 16; do
  for q in 16 32; do
    [ $q -lt $s ] && continue
    GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 32 --warmup 6 --streams $s 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('streams $s queues $q: %.1f fps %.2f ms/step latency %.1f ms' % (l['value'], l['ms_per_step'], l['latency_ms_per_batch']['streams_$s']))"
  done
done
