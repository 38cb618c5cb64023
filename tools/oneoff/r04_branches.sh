#!/bin/bash
# image branch beside the point branch inside a step: headline / latency against the number of streams
OUT=gpurun_out/r04br; mkdir -p $OUT
for s in 1 2 4 8; do for c in 0 1; do
GPU_MAX_HW_QUEUES=16 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 6 --streams $s --concurrent-branches $c 2>$OUT/err_${s}_$c.txt | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('streams $s concurrent $c: %.1f fps %.2f ms/step latency %.1f ms (one in flight %s) graph %s' % (l['value'], l['ms_per_step'], l['latency_ms_per_batch']['streams_$s'], l['latency_ms_per_batch']['one_step_in_flight'], l['config']['hip_graph']))" >> $OUT/res.txt
done; done
cat $OUT/res.txt; tail -2 $OUT/err_1_1.txt
# Result (round 4): image encoder on a side stream beside the point encoder inside every step (fork / join captured into the slot's graph):
# streams 1: 10.86 -> 10.78 ms per step; streams 2: 3836 -> 3677 frames/s; 4: 4234 -> 4023; 8: 4357 -> 4329.  The image branch's kernels fill
# the chip by themselves; the fork only adds edges.  Not adopted (networks.py / bench.py went back).
