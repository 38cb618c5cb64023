#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for pad in 0 14000 8000 14000 0; do
DI2P_SOLVER_LDS_PAD=$pad timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 32 --warmup 6 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('solver_lds_pad=$pad %.1f fps %.2f ms/step | solver %.2f conv %.2f pointwise %.2f lat1 %.2f lat8 %.1f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step'], l['latency_ms_per_batch']['one_step_in_flight'], l['latency_ms_per_batch']['streams_8']))" >> $OUT/r05_c14_pad.txt 2>&1
done
cat $OUT/r05_c14_pad.txt
