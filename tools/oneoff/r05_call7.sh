#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_solver.py -q -x 2>&1 | grep -E "^E  |Error|FAILED|passed|failed" | cut -c1-300 | head -30 > $OUT/r05_c7_tests.txt
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps (h2d %.1f = %.3f) %.2f ms/step | lat %s' % (l['value'], l['value_with_h2d'], l['value_with_h2d']/l['value'], l['ms_per_step'], l['latency_ms_per_batch']))" >> $OUT/r05_c7_headline.txt 2>&1; done
cat $OUT/r05_c7_tests.txt $OUT/r05_c7_headline.txt
