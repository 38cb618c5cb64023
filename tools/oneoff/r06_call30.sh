#!/bin/bash
# round 6, call 30: planes-source kernel with 256-row tiles and both operands through LDS (eight waves) against the 128-row kernel
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c30_x3p8.txt; : > $LOG
timeout 900 python -m pytest tests/test_gpu_contractions.py tests/test_gpu_network.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3 >> $LOG
for pl in 1 2 1 2; do
  echo "== pw_x3_planes=$pl" >> $LOG
  DI2P_PW_X3_PLANES=$pl REPS=20 PLANES=1 timeout 300 python tools/bench_pw_x3.py 2>&1 | grep -v amdgpu.ids | grep "plain\|gmax" >> $LOG
  DI2P_PW_X3_PLANES=$pl timeout 200 python tools/bench_pw_planes.py 2>&1 | grep -v amdgpu.ids | tail -2 >> $LOG
done
for rep in 1 2 3; do
for pl in 1 2 0; do
  DI2P_PW_X3_PLANES=$pl timeout 200 python bench.py --no-cpu-baseline --steps 48 --warmup 8 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('planes $pl: %.1f fps resident  %.1f with h2d | %.2f ms/step | pointwise family serial %.3f ms' % (l['value'], l.get('value_with_h2d', 0), l['ms_per_step'], l['kernels']['pointwise_gemm_kernel(+point_head)']['ms_per_step']))" >> $LOG
done
done
cat $LOG
