#!/bin/bash
# round 6, call 26: bf16x3 pointwise epilogue with every tile's loads in front of every tile's stores (+ 16-byte row operands): tests, K sweep, chain, headline
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c26_pw_epilogue.txt; : > $LOG
timeout 900 python -m pytest tests/test_gpu_contractions.py tests/test_gpu_network.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_point_ops.py -x -q 2>&1 | tail -4 >> $LOG
echo "== fp32 source" >> $LOG
REPS=20 timeout 300 python tools/bench_pw_x3.py 2>&1 | grep -v amdgpu.ids >> $LOG
echo "== planes source" >> $LOG
REPS=20 PLANES=1 timeout 300 python tools/bench_pw_x3.py 2>&1 | grep -v amdgpu.ids >> $LOG
timeout 200 python tools/bench_pw_planes.py 2>&1 | grep -v amdgpu.ids >> $LOG
for rep in 1 2 3; do
for pl in 1 0; do
  DI2P_PW_X3_PLANES=$pl timeout 200 python bench.py --no-cpu-baseline --steps 48 --warmup 8 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('planes $pl: %.1f fps resident  %.1f with h2d | %.2f ms/step' % (l['value'], l.get('value_with_h2d', 0), l['ms_per_step']))" >> $LOG
done
done
cat $LOG
