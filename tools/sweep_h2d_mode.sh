#!/bin/bash
# where the host->device copies of a step run (GPU box): memcpy nodes of the graph / hipMemcpyAsync ahead of the replay / second stream per slot
for rep in 1 2 3; do for m in graph eager copy_stream; do
DI2P_H2D_MODE=$m timeout 200 python bench.py --no-cpu-baseline --steps 32 --warmup 6 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('h2d $m: resident %.1f  with h2d %.1f fps (%.3f)' % (l['value'], l['value_with_h2d'], l['value_with_h2d']/l['value']))"
done; done
