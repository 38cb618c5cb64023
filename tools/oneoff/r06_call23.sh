#!/bin/bash
# round 6, call 23: bf16x3 pointwise layers handing their activations on as split planes -- tests, per-layer times, headline A/B
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c23_pw_planes.txt; : > $LOG
timeout 600 python -m pytest tests/test_gpu_contractions.py tests/test_gpu_network.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4 >> $LOG
timeout 200 python tools/bench_pw_planes.py 2>&1 | grep -v amdgpu.ids >> $LOG
for rep in 1 2 3; do
for pl in 1 0; do
  DI2P_PW_X3_PLANES=$pl timeout 200 python bench.py --no-cpu-baseline --steps 48 --warmup 8 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
x3=k['pointwise_gemm_kernel(+point_head)']['ms_per_step']
print('planes $pl: %.1f fps resident  %.1f with h2d | %.2f ms/step | pointwise family serial %.3f ms' % (l['value'], l.get('value_with_h2d', 0), l['ms_per_step'], x3))" >> $LOG
done
done
cat $LOG
