#!/bin/bash
# where the host->device copies of a step run (GPU box): memcpy nodes of the graph / hipMemcpyAsync ahead of the replay / second stream per
# slot, and how many hardware queues the 8 + 8 streams of the last mode get
for rep in 1 2; do for m in "DI2P_H2D_MODE=eager" "DI2P_H2D_MODE=copy_stream" "DI2P_H2D_MODE=copy_stream GPU_MAX_HW_QUEUES=24" "DI2P_H2D_MODE=copy_stream GPU_MAX_HW_QUEUES=32" "DI2P_H2D_MODE=graph"; do
env $m timeout 200 python bench.py --no-cpu-baseline --steps 32 --warmup 6 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('%-60s resident %.1f  with h2d %.1f fps (%.3f)' % ('$m', l['value'], l['value_with_h2d'], l['value_with_h2d']/l['value']))"
done; done
