#!/bin/bash
OUT=gpurun_out/r04x3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_contractions.py tests/test_gpu_network.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py tests/test_gpu_training.py -q 2>&1 | tail -25 > $OUT/tests.txt
timeout 200 python tools/call_times.py 15 > $OUT/call_times_x3.txt 2>&1
DI2P_PW_X3=0 timeout 200 python tools/call_times.py 15 > $OUT/call_times_fp32.txt 2>&1
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
for i in 1 2; do
  echo "x3  : $(timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/err_x3.txt | line)" >> $OUT/ab.txt
  echo "fp32: $(DI2P_PW_X3=0 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/err_fp32.txt | line)" >> $OUT/ab.txt
done
cat $OUT/tests.txt; grep "pointwise_gemm" $OUT/call_times_x3.txt | head -30; echo ---; grep "pointwise_gemm  " $OUT/call_times_fp32.txt | tail -3; grep "per entry" -A6 $OUT/call_times_x3.txt; cat $OUT/ab.txt; tail -3 $OUT/err_x3.txt
