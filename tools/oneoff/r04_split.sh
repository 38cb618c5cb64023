#!/bin/bash
# classifier and pose solve of a step as two graphs on streams of different priority (bench.py --split-solver) against one graph per step
# result (r04): 4563 / 4567 against 4638 / 4539 frames/s resident (level), 3856 / 3841 against 4357 / 4359 with the H2D copies, 63-66 against 52 ms per
# batch: not adopted (the option stays in the executor, default off)
OUT=gpurun_out/r04split; mkdir -p $OUT
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('%.1f fps  %.2f ms/step  with_h2d %.1f  latency %s' % (l['value'], l['ms_per_step'], l.get('value_with_h2d') or 0, l['latency_ms_per_batch']['streams_8']))"; }
for i in 1 2; do
  echo "one graph : $(timeout 200 python bench.py --no-cpu-baseline --steps 24 --warmup 6 2>$OUT/err.txt | line)" >> $OUT/ab.txt
  echo "split     : $(timeout 200 python bench.py --no-cpu-baseline --steps 24 --warmup 6 --split-solver 2>$OUT/err2.txt | line)" >> $OUT/ab.txt
done
cat $OUT/ab.txt; tail -3 $OUT/err2.txt
