#!/bin/bash
OUT=gpurun_out/r04x3b; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_contractions.py tests/test_gpu_network.py tests/test_gpu_fullsize.py -q 2>&1 | tail -12 > $OUT/tests.txt
timeout 200 python tools/call_times.py 15 > $OUT/call_times.txt 2>&1
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
for i in 1 2; do
  echo "new: $(timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/err.txt | line)" >> $OUT/ab.txt
done
cat $OUT/tests.txt; grep "pointwise_gemm" $OUT/call_times.txt | head -40; cat $OUT/ab.txt
