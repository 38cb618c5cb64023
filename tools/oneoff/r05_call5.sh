#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 > $OUT/r05_c5_gputest_tail.txt
for v in "31 -1" "31 3" "28 -1" "12 -1"; do set -- $v
DI2P_CONV_X3=$1 DI2P_CONV_X3_CFG=$2 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('conv_x3=$1 cfg=$2 %.1f fps %.2f ms/step | solver %.2f conv %.2f pointwise %.2f | lat1 %s' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step'], l['latency_ms_per_batch']['one_step_in_flight']))" >> $OUT/r05_c5_headline.txt 2>&1
done
cat $OUT/r05_c5_gputest_tail.txt $OUT/r05_c5_headline.txt
