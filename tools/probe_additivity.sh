# is the 8-stream step time the SUM of what the network and the pose solver need (no overlap benefit), or less?
for r in 1 30 60 120; do
timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 4 --restarts $r 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('restarts $r: %.2f ms/step (%.1f frames/s)' % (l['ms_per_step'], l['value']))"
done
