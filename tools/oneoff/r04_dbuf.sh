#!/bin/bash
# double-buffered device inputs per slot (DI2P_DOUBLE_BUFFER=1) against one input set (default): executor tests, H2D-inclusive rate
# result (r04): 4386 / 4382 against 4351 / 4390 frames/s with the H2D copies (0.95 of the resident rate either way), 48.5 against 51.5 ms per batch: option, off by default
OUT=gpurun_out/r04dbuf; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_bench_launch.py -q -m gpu 2>&1 | tail -4 > $OUT/tests.txt
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('%.1f fps resident  %.1f with_h2d (%.3f)  latency %.1f / %.1f ms' % (l['value'], l.get('value_with_h2d') or 0, (l.get('value_with_h2d') or 0) / l['value'], l['latency_ms_per_batch']['streams_8'], l['latency_ms_per_batch']['streams_8_with_h2d']))"; }
for i in 1 2; do
  echo "double buffer: $(DI2P_DOUBLE_BUFFER=1 timeout 200 python bench.py --no-cpu-baseline --steps 32 --warmup 8 2>$OUT/err.txt | line)" >> $OUT/ab.txt
  echo "single set   : $(timeout 200 python bench.py --no-cpu-baseline --steps 32 --warmup 8 2>$OUT/err2.txt | line)" >> $OUT/ab.txt
done
cat $OUT/tests.txt $OUT/ab.txt; tail -2 $OUT/err.txt
