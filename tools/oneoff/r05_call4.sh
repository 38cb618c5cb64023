#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv_x3.py -q 2>&1 | grep -E "^E  |Error|FAILED|passed|failed" | cut -c1-250 | head -40 > $OUT/r05_c4_test_conv_x3.txt
timeout 300 python tools/bench_conv_x3.py 2>&1 | grep -v amdgpu > $OUT/r05_c4_bench_conv_x3.txt
for m in 31 28 0; do
DI2P_CONV_X3=$m timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('conv_x3=$m %.1f fps (h2d %.1f) %.2f ms/step | solver %.2f conv %.2f pointwise %.2f | lat1 %s' % (l['value'], l['value_with_h2d'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step'], l['latency_ms_per_batch']['one_step_in_flight']))" >> $OUT/r05_c4_headline.txt 2>&1
done
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $OUT/r05_c4_gputest_tail.txt
cat $OUT/r05_c4_test_conv_x3.txt $OUT/r05_c4_bench_conv_x3.txt $OUT/r05_c4_headline.txt $OUT/r05_c4_gputest_tail.txt
