#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
python tools/bench_head_x3.py 2>&1 | grep -v amdgpu > $OUT/r05_c11_bench_head.txt
for t in 1 0; do DI2P_HEAD_X3_TAB=$t python -m pytest tests/test_gpu_head_x3.py -q 2>&1 | tail -1 >> $OUT/r05_c11_bench_head.txt; done
for v in "1 1" "1 0" "0 1" "1 1"; do set -- $v
DI2P_HEAD_X3=$1 DI2P_HEAD_X3_TAB=$2 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 32 --warmup 6 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('head_x3=$1 tab=$2 %.1f fps %.2f ms/step | solver %.2f conv %.2f pointwise %.2f lat1 %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step'], l['latency_ms_per_batch']['one_step_in_flight']))" >> $OUT/r05_c11_headline.txt 2>&1
done
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $OUT/r05_c11_gputest_tail.txt
cat $OUT/r05_c11_bench_head.txt $OUT/r05_c11_headline.txt $OUT/r05_c11_gputest_tail.txt
