#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for f in "" 1 "" 1; do
DI2P_BENCH_H2D_FIRST=$f timeout 200 python bench.py --no-cpu-baseline --steps 48 --warmup 8 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('h2d_first=\"$f\" resident %.1f fps, with h2d %.1f (%.3f) | pointwise %.2f' % (l['value'], l['value_with_h2d'], l['value_with_h2d']/l['value'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))" >> $OUT/r05_c15_h2d_order.txt 2>&1
done
cat $OUT/r05_c15_h2d_order.txt
timeout 200 python tools/call_times.py 15 2>&1 | grep -v amdgpu | tail -32
