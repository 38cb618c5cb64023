#!/bin/bash
# the headline against the length of the timed loop (the loop starts and ends with an empty pipeline: fill and drain are inside it)
for k in "20 5" "24 4" "32 6" "64 8"; do set -- $k
for s in ${STREAMS:-6 8}; do
timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps $1 --warmup $2 --streams $s 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('steps $1 warmup $2 streams $s: %.1f fps %.2f ms/step' % (l['value'], l['ms_per_step']))"
done; done
