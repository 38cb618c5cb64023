#!/bin/bash
# wave-autonomous head (DI2P_HEAD_REG=1) against the LDS-tile head (default), same binary: tests, micro-benchmark, bench A/B
# result (r04): alone 535 vs 642 us, but 4548 vs 4586 frames/s in the 8-stream pipeline (one persistent 8-wave workgroup with 119 KB of LDS
# per compute unit leaves no room for the other streams' workgroups) -> the LDS-tile kernel stays the default
OUT=gpurun_out/r04head2; mkdir -p $OUT
timeout 120 python tools/bench_head.py 2>&1 | grep -v amdgpu.ids > $OUT/res.txt
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_contractions.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q 2>&1 | tail -5 > $OUT/tests.txt
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
for i in 1 2; do
  echo "reg head: $(DI2P_HEAD_REG=1 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/err.txt | line)" >> $OUT/ab.txt
  echo "lds head: $(timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/err2.txt | line)" >> $OUT/ab.txt
done
cat $OUT/res.txt $OUT/tests.txt $OUT/ab.txt
