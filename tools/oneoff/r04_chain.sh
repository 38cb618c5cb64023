#!/bin/bash
# fused narrow PointNet chains (di2p_point_chain) against the separate launches (DI2P_PW_NOCHAIN=1), same binary
OUT=gpurun_out/r04chain; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_contractions.py tests/test_gpu_fullsize.py -q 2>&1 | tail -12 > $OUT/tests.txt
timeout 200 python tools/call_times.py 5 > $OUT/call_times.txt 2>&1
DI2P_PW_NOCHAIN=1 timeout 200 python tools/call_times.py 5 > $OUT/call_times_nochain.txt 2>&1
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
for i in 1 2; do
  echo "chain  : $(timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/err.txt | line)" >> $OUT/ab.txt
  echo "nochain: $(DI2P_PW_NOCHAIN=1 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/err2.txt | line)" >> $OUT/ab.txt
done
cat $OUT/tests.txt; grep -i "chain\|M=32 \|M=64 " $OUT/call_times.txt | head; echo ---; grep -i "M=32 \|M=64 " $OUT/call_times_nochain.txt | head; cat $OUT/ab.txt
