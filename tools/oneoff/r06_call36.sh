#!/bin/bash
# round 6, call 36 (final tree): soak -- the GPU suite three times, a long bench run, smoke()
OUT=gpurun_out; LOG=$OUT/r06_c36_soak.txt; : > $LOG
for i in 1 2 3; do timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -1 >> $LOG; done
timeout 600 python bench.py --no-cpu-baseline --steps 400 --warmup 16 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('400 steps: %.1f fps resident, %.1f with h2d, pose_check %s' % (l['value'], l['value_with_h2d'], l['pose_check']))" >> $LOG
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $LOG
cat $LOG
